"""Data-parallel gradient reduction (mdctgan_amd/ddp.py) on CPU: 2 processes, gloo backend, 127.0.0.1.
The reducer only needs a flat gradient arena and 'gradient ready' notifications, so the oracle networks (torch CPU
autograd) stand in for the HIP wgrad kernels.  Claim under test: the average of per-rank gradients on
disjoint half-batches == the single-process gradient on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _arena(params):
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(total)
    slices = [(p, o, (p.numel() + 3) // 4 * 4) for p, o in zip(params, offs)]
    return flat, slices


def _loss(netD, x):
    from oracle import nets as onets
    feats = netD(x)
    return onets.lsgan_loss(feats, True) + sum(f.abs().mean() for sc in feats for f in sc[:-1])


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from mdctgan_amd import ddp
    from mdctgan_amd import functional as Fh
    from oracle import nets as onets
    torch.manual_seed(100 + rank)                      # deliberately different initial weights per rank
    netD = onets.init_weights(onets.MultiscaleDRef(3, ndf=4, n_layers=3, num_D=2))
    params = list(netD.parameters())
    flat_p, pslices = _arena(params)
    with torch.no_grad():
        for p, o, n in pslices:
            flat_p[o:o + p.numel()].copy_(p.reshape(-1))
            p.data = flat_p[o:o + p.numel()].view(p.shape)
    ddp.broadcast_arena(flat_p, 0)                     # C2
    ref0 = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(ref0, flat_p)
    assert torch.equal(ref0[0], ref0[1])
    flat_g, gslices = _arena(params)
    red = ddp.ArenaReducer(flat_g, gslices, writes_per_step=1, bucket_bytes=4096)
    assert len(red.buckets) > 2
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 32, 64, generator=g)         # the same global batch on both ranks
    mine = x[rank * 2:(rank + 1) * 2]
    for step in range(2):                              # two steps: reducer state resets correctly
        netD.zero_grad()
        _loss(netD, mine).backward()
        for p, o, n in reversed(gslices):              # wgrad kernels finish in reverse topological order
            flat_g[o:o + p.numel()].copy_(p.grad.reshape(-1))
            Fh._notify(p)
        red.finish()
        avg = flat_g / world
        netD.zero_grad()
        _loss(netD, x).backward()                      # single-process gradient on the concatenated batch
        gmax = max(float(p.grad.abs().max()) for p in params)
        for p, o, n in gslices:
            want = p.grad.reshape(-1)
            got = avg[o:o + p.numel()]
            err = float((got - want).abs().max())
            # biases ahead of an InstanceNorm have a true gradient of 0: what is compared there is rounding noise,
            # hence the absolute floor tied to the largest gradient in the network
            assert err <= 2e-6 * gmax + 2e-5 * float(want.abs().max()), (rank, step, o, err, float(want.abs().max()))
    # a bucket whose parameters never report (frozen layer) is flushed by finish()
    for p, o, n in gslices[:3]:
        Fh._notify(p)
    red.finish()
    assert red.pending == [b[2] for b in red.buckets]
    red.close()
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.ones(1))
    dist.destroy_process_group()


def test_two_rank_gradient_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d.npy" % r)) for r in range(2))


def test_bucket_layout_single_process():
    from mdctgan_amd import ddp
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 1000, 5, 3000, 7)]
    flat, slices = _arena(params)
    red = ddp.ArenaReducer(flat, slices, writes_per_step=2, bucket_bytes=4096)
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == flat.numel()
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:]))     # contiguous, no gaps
    assert sum(b[2] for b in red.buckets) == len(params)
    assert red.pending == [2 * b[2] for b in red.buckets]
    # the bucket holding the first parameters launches last (backward fills the arena from its end): it is capped at a
    # quarter of a bucket so the one all-reduce that cannot overlap with backward stays small
    first = red.buckets[0]
    assert first[1] - first[0] <= 4096 // 4 // 4 or first[2] == 1
    assert all(b[1] - b[0] <= 4096 // 4 or b[2] == 1 for b in red.buckets)
    red.close()
    # a module applied k times in one forward (the shared samplers of LocalEnhancer's attention sandwich) writes its
    # gradient k times per backward: its bucket waits for all of them
    params[1]._mg_writes = 3
    red = ddp.ArenaReducer(flat, slices, writes_per_step=1, bucket_bytes=1 << 20)
    assert sum(red.pending) == len(params) + 2
    from mdctgan_amd import functional as Fh
    red.force = False
    for p in params:
        for _ in range(int(getattr(p, "_mg_writes", 1))):
            Fh._notify(p)
    assert all(v == 0 for v in red.pending)
    red.finish()
    assert sum(red.pending) == len(params) + 2
    red.close()
