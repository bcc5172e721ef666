"""Data path of the DyCON runs with the reference's names (code/dataloaders/brats19.py).

    BraTS2019(base_dir, split, num, transform)                 :12-46    h5 cases listed in train.txt / test.txt
    CenterCrop, RandomCrop, RandomRotFlip, RandomNoise, ToTensor   :139-283
    TwoStreamBatchSampler(primary, secondary, batch, secondary_batch)   :286-335   labelled-first batches

The transforms draw from ``np.random`` in the reference's call order, so a seeded run reproduces the reference's augmentation
stream; they accept numpy arrays (the reference's DataLoader workers) and torch tensors alike -- on CUDA tensors crop / rot90 /
flip are index operations on the device, which is how a whole volume cache in HBM (288 GB) is augmented without a host round trip.
h5py is not available in this image: the dataset class imports it lazily and is exercised only where the data lives.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler


class BraTS2019(Dataset):
    """code/dataloaders/brats19.py:12-46: `<base_dir>/train.txt|test.txt` name lists, `<base_dir>/data/<name>.h5` with 'image'
    (float32) and 'label' datasets (BraTS19_DataPreprocessing.py:189-208)."""

    def __init__(self, base_dir=None, split="train", num=None, transform=None):
        self._base_dir = base_dir
        self.transform = transform
        path = self._base_dir + ("/train.txt" if split == "train" else "/test.txt")
        with open(path, "r") as f:
            self.image_list = [item.replace("\n", "").split(",")[0] for item in f.readlines()]
        if num is not None:
            self.image_list = self.image_list[:num]

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, idx):
        import h5py   # lazy: absent in the build image
        with h5py.File(self._base_dir + "/data/{}.h5".format(self.image_list[idx]), "r") as h5f:
            sample = {"image": h5f["image"][:], "label": h5f["label"][:].astype(np.uint8)}
        return self.transform(sample) if self.transform else sample


def _pad3(a, pw, ph, pd):
    if torch.is_tensor(a):
        return torch.nn.functional.pad(a, (pd, pd, ph, ph, pw, pw))
    return np.pad(a, [(pw, pw), (ph, ph), (pd, pd)], mode="constant", constant_values=0)


def _pad_to(sample_arrays, output_size):
    """the reference pads by (size - shape)//2 + 3 on both sides when ANY dimension is <= the crop (:146-156, :198-210)"""
    shape = sample_arrays[0].shape
    if shape[0] <= output_size[0] or shape[1] <= output_size[1] or shape[2] <= output_size[2]:
        pw = max((output_size[0] - shape[0]) // 2 + 3, 0)
        ph = max((output_size[1] - shape[1]) // 2 + 3, 0)
        pd = max((output_size[2] - shape[2]) // 2 + 3, 0)
        sample_arrays = [_pad3(a, pw, ph, pd) for a in sample_arrays]
    return sample_arrays


class SagittalToAxial(object):
    """First transform of the reference's BraTS training pipeline (train_DyCON_BraTS19.py:238-242, dataloaders/brats19.py:86-128):
    image and label (H, W, D) -> (D, W, H).  numpy arrays or (device) tensors."""

    def __call__(self, sample):
        image, label = sample["image"], sample["label"]
        if tuple(image.shape) != tuple(label.shape):
            raise ValueError("Shape mismatch between image and label")
        if torch.is_tensor(image):
            return {"image": image.permute(2, 1, 0), "label": label.permute(2, 1, 0)}
        return {"image": np.transpose(image, (2, 1, 0)), "label": np.transpose(label, (2, 1, 0))}


class CenterCrop(object):
    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = _pad_to([sample["image"], sample["label"]], self.output_size)
        (w, h, d) = image.shape
        o = self.output_size
        w1, h1, d1 = int(round((w - o[0]) / 2.)), int(round((h - o[1]) / 2.)), int(round((d - o[2]) / 2.))
        return {"image": image[w1:w1 + o[0], h1:h1 + o[1], d1:d1 + o[2]], "label": label[w1:w1 + o[0], h1:h1 + o[1], d1:d1 + o[2]]}


class RandomCrop(object):
    def __init__(self, output_size, with_sdf=False):
        self.output_size = output_size
        self.with_sdf = with_sdf

    def __call__(self, sample):
        keys = ["image", "label"] + (["sdf"] if self.with_sdf else [])
        arrs = _pad_to([sample[k] for k in keys], self.output_size)
        (w, h, d) = arrs[0].shape
        o = self.output_size
        w1 = np.random.randint(0, w - o[0])          # same draws, same order as :219-221
        h1 = np.random.randint(0, h - o[1])
        d1 = np.random.randint(0, d - o[2])
        return {k: a[w1:w1 + o[0], h1:h1 + o[1], d1:d1 + o[2]] for k, a in zip(keys, arrs)}


class RandomRotFlip(object):
    """np.rot90(k) about the first two axes, then a flip of axis 0 or 1 (:236-252)."""

    def __call__(self, sample):
        image, label = sample["image"], sample["label"]
        k = np.random.randint(0, 4)
        axis = np.random.randint(0, 2)
        if torch.is_tensor(image):
            f = lambda t: torch.flip(torch.rot90(t, k, dims=(0, 1)), dims=(axis,)).contiguous()   # noqa: E731
        else:
            f = lambda a: np.flip(np.rot90(a, k), axis=axis).copy()                               # noqa: E731
        return {"image": f(image), "label": f(label)}


class RandomNoise(object):
    def __init__(self, mu=0, sigma=0.1):
        self.mu = mu
        self.sigma = sigma

    def __call__(self, sample):
        image, label = sample["image"], sample["label"]
        noise = np.clip(self.sigma * np.random.randn(image.shape[0], image.shape[1], image.shape[2]), -2 * self.sigma, 2 * self.sigma)
        noise = noise + self.mu
        if torch.is_tensor(image):
            noise = torch.as_tensor(noise, dtype=image.dtype, device=image.device)
        return {"image": image + noise, "label": label}


class ToTensor(object):
    """image -> (1, w, h, d) float32 tensor, label -> long tensor (:272-283); tensors stay on their device."""

    def __call__(self, sample):
        image, label = sample["image"], sample["label"]
        if torch.is_tensor(image):
            return {"image": image.reshape(1, *image.shape).float(), "label": label.long()}
        image = image.reshape(1, image.shape[0], image.shape[1], image.shape[2]).astype(np.float32)
        return {"image": torch.from_numpy(image), "label": torch.from_numpy(np.ascontiguousarray(label)).long()}


class TwoStreamBatchSampler(Sampler):
    """Batches of `batch_size - secondary_batch_size` primary (labelled) indices FOLLOWED by `secondary_batch_size` secondary
    (unlabelled) ones -- the order the training step relies on (the first labeled_bs samples of a batch are the labelled ones).
    An epoch is one shuffled pass over the primary indices; the secondary indices are reshuffled and reused as often as needed
    (code/dataloaders/brats19.py:286-335).  np.random is consumed in the reference's order: the primary permutation first, then
    one secondary permutation each time the previous one is used up."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        if not (len(primary_indices) >= self.primary_batch_size > 0 and len(secondary_indices) >= secondary_batch_size > 0):
            raise AssertionError("each stream needs at least one batch worth of indices")

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size

    def __iter__(self):
        labelled = np.random.permutation(self.primary_indices)

        def unlabelled_forever():
            while True:
                yield from np.random.permutation(self.secondary_indices)

        stream = unlabelled_forever()
        nl, nu = self.primary_batch_size, self.secondary_batch_size
        for i in range(len(self)):
            head = tuple(labelled[i * nl:(i + 1) * nl])
            yield head + tuple(next(stream) for _ in range(nu))
