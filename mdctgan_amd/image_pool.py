"""util/image_pool.py:4-31 (``--pool_size > 0``: the discriminator's fake pass sees a history of generated inputs).

Host-side bookkeeping over device tensors, statement for statement the reference's: the pool fills up first, afterwards a
query returns the new image or -- with probability 1/2 -- swaps it against a random stored one.  The random numbers come
from Python's ``random`` module in the reference's call order (``random.uniform`` then ``random.randint``), so a seeded run
selects the same images (tests/golden/g11_image_pool.npz).  Images stay in HBM; no arithmetic happens here."""
from __future__ import annotations

import random

import torch


class ImagePool:
    def __init__(self, pool_size):
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.num_imgs = 0
            self.images = []

    def query(self, images):
        if self.pool_size == 0:
            return images
        return_images = []
        for image in images.detach():
            image = torch.unsqueeze(image, 0)
            if self.num_imgs < self.pool_size:
                self.num_imgs = self.num_imgs + 1
                self.images.append(image)
                return_images.append(image)
            else:
                p = random.uniform(0, 1)
                if p > 0.5:
                    random_id = random.randint(0, self.pool_size - 1)
                    tmp = self.images[random_id].clone()
                    self.images[random_id] = image
                    return_images.append(tmp)
                else:
                    return_images.append(image)
        return torch.cat(return_images, 0)
