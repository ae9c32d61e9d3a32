"""History of generated discriminator inputs for ``--pool_size > 0`` (behaviour of util/image_pool.py:4-31).

Layout: one preallocated device slab ``[pool_size, C, H, W]`` instead of a Python list of one-image tensors.  A query is
split into a host half and a device half:

* host: ``_plan`` walks the incoming rows in order and draws from Python's ``random`` exactly the numbers the reference
  draws (one ``random.uniform(0, 1)`` per row once the slab is full, followed by one ``random.randint(0, pool_size - 1)``
  when the draw exceeds 0.5), so a seeded run hands out the same images (tests/golden/g11_image_pool.npz).  It tracks
  where every slot's content comes from *during* the walk -- a row can be swapped against a slot an earlier row of the same
  query has just filled -- and emits three index lists.
* device: at most one gather of old slab rows into the result, one gather of batch rows into the result, and one scatter
  of batch rows into the slab; reads of old content are issued before the slab is written.

No arithmetic happens here; images never leave HBM."""
from __future__ import annotations

import random

import torch


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = int(pool_size)
        self._slab = None           # [pool_size, C, H, W], allocated on the first query
        self._filled = 0

    # the reference exposes these two; tests/test_nets_gpu.py::test_image_pool_step reads them
    @property
    def num_imgs(self):
        return self._filled

    @property
    def images(self):
        return [] if self._slab is None else list(self._slab[:self._filled].split(1))

    def _plan(self, n):
        """-> (result_from_batch [n], [(result_row, slot)] served from old slab content, {slot: batch_row} final writes)."""
        origin = {}                 # slot -> batch row whose image sits there now (absent: content older than this query)
        from_batch = list(range(n))
        from_slab = []
        for row in range(n):
            if self._filled < self.pool_size:
                origin[self._filled] = row
                self._filled += 1
                continue
            if random.uniform(0, 1) > 0.5:
                slot = random.randint(0, self.pool_size - 1)
                if slot in origin:
                    from_batch[row] = origin[slot]
                else:
                    from_slab.append((row, slot))
                origin[slot] = row
        return from_batch, from_slab, origin

    def query(self, images):
        if self.pool_size == 0:
            return images
        batch = images.detach()
        n, dev = batch.shape[0], batch.device
        if self._slab is None:
            self._slab = torch.empty((self.pool_size,) + tuple(batch.shape[1:]), dtype=batch.dtype, device=dev)
        from_batch, from_slab, writes = self._plan(n)

        def idx(v):
            return torch.tensor(v, dtype=torch.long, device=dev)
        out = batch.clone() if from_batch == list(range(n)) else batch.index_select(0, idx(from_batch))
        if from_slab:
            rows, slots = zip(*from_slab)
            out.index_copy_(0, idx(rows), self._slab.index_select(0, idx(slots)))
        if writes:
            slots, rows = zip(*sorted(writes.items()))
            self._slab.index_copy_(0, idx(slots), batch.index_select(0, idx(rows)))
        return out
