"""ISLES-2022 (DWI) data path with the reference's names (code/dataloaders/isles22.py; train_DyCON_ISLES22.py:28,165-190).

    ISLESDataset(h5_dir, split, transform)      :53-95    `<h5_dir>/<split>.list` of case ids, `<h5_dir>/<case>.h5` with 'image' and
                                                          'mask' (both float64, ISLES22_DataPreprocessing.py:208-210); sample key 'label'
    RandomCrop, RandomRotFlip, RandomRot, RandomNoise, CenterCrop, CreateOnehotLabel, ToTensor   :115-247
    TwoStreamBatchSampler, ThreeStreamBatchSampler                                              :250-308

Transforms shared with dataloaders/brats19.py / pancreas.py (identical np.random call order).  `RandomRot` rotates by a random
integer angle with scipy.ndimage (nearest neighbour, no reshape) -- host arrays only.  Parity unpinned (see pancreas.py).
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from .brats19 import CenterCrop, RandomCrop, RandomNoise, RandomRotFlip, TwoStreamBatchSampler   # noqa: F401  (re-exported)
from .pancreas import CreateOnehotLabel, Resize, ToTensor                                       # noqa: F401


class ISLESDataset(Dataset):
    def __init__(self, h5_dir, split="train", transform=None):
        self.h5_dir, self.split, self.transform = h5_dir, split, transform
        list_file = os.path.join(h5_dir, f"{split}.list")
        if not os.path.exists(list_file):
            raise FileNotFoundError(f"List file {list_file} not found.")
        with open(list_file, "r") as f:
            names = [f"{line.strip()}.h5" for line in f if line.strip()]
        self.sample_list = [n for n in names if os.path.exists(os.path.join(h5_dir, n))]    # listed cases without a file are skipped (:76)

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        import h5py   # lazy: absent in the build image
        with h5py.File(os.path.join(self.h5_dir, self.sample_list[idx]), "r") as h5f:
            sample = {"image": h5f["image"][:], "label": h5f["mask"][:]}
        return self.transform(sample) if self.transform else sample


class RandomRot(object):
    """rotation about the third axis by np.random.randint(-20, 20) degrees, order 0, shape kept (:24-28, :198-209)"""

    def __call__(self, sample):
        from scipy import ndimage
        angle = np.random.randint(-20, 20)
        f = lambda a: ndimage.rotate(np.asarray(a), angle, order=0, reshape=False)   # noqa: E731
        return {"image": f(sample["image"]), "label": f(sample["label"])}


class ThreeStreamBatchSampler(Sampler):
    """[primary | secondary | primary] batches (:280-308).  As the reference behaves: the first and the third group of a batch are the
    SAME primary indices (see __iter__), an epoch is one pass over the primary permutation = len(primary) // primary_batch_size batches."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        assert len(primary_indices) >= self.primary_batch_size > 0
        assert len(secondary_indices) >= secondary_batch_size > 0

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size

    def __iter__(self):
        # The reference zips grouper(primary_iter), grouper(secondary_iter), grouper(primary_iter) where primary_iter is an ndarray
        # (np.random.permutation), not an iterator: each grouper() takes its OWN iter() of it, so the first and the third group of a
        # batch are the same indices -- every batch is [primary group | secondary group | the same primary group], and an epoch has
        # len(primary) // primary_batch_size batches (dataloaders/isles22.py:296-305, 310-311, 321-325).  Reproduced as it is.
        labelled = list(np.random.permutation(self.primary_indices))

        def unlabelled_forever():
            while True:
                yield from np.random.permutation(self.secondary_indices)

        stream = unlabelled_forever()
        nl, nu = self.primary_batch_size, self.secondary_batch_size
        for pos in range(0, len(labelled) - nl + 1, nl):
            head = tuple(labelled[pos:pos + nl])
            mid = tuple(next(stream) for _ in range(nu))
            yield head + mid + head
