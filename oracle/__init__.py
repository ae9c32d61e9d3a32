"""CPU oracle for the mdctGAN hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference's hot path
(neoncloud/mdctGAN: models/mdct.py MDCT4/IMDCT4, Audio2MDCT in
models/pix2pixHD_model.py, the generator / discriminator stacks in
models/networks.py and the G/D step in train.py:160-202).  Every function
cites the reference file:line it follows.

It is the *checker*, never the product:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it;
* nothing under ``mdctgan_amd/`` imports it, and the product path raises when
  the HIP library is missing instead of falling back to this code.

Pinning: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, captured in the build container by ``oracle/gen_golden.py`` (which
imports /root/reference read-only) and committed as ``tests/golden/*.npz``.
The bottleneck-transformer block (third-party ``bottleneck_transformer_pytorch
==0.1.4``, not vendored by the reference) is restated from its published
algorithm; its parity is *unpinned* (shape / key fixtures only).
"""
