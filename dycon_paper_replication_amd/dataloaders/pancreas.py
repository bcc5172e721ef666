"""Pancreas-CT data path with the reference's names (code/dataloaders/pancreas.py; train_DyCON_Pancreas.py:20,150-167).

    Pancreas(base_dir, split, num, transform)      :12-47    `<base_dir>/train.list|test.list`, `<base_dir>/Pancreas_data/<name>`
    CenterCrop, RandomCrop(with_sdf), RandomRotFlip, RandomNoise, CreateOnehotLabel, ToTensor, TwoStreamBatchSampler   :66-238

The transforms are the ones of dataloaders/brats19.py (same np.random call order in the reference's three loader modules) and work on
numpy arrays and on device tensors alike.  `Resize` needs scikit-image, which neither this image nor the reference vendors: it is
imported when called.  h5py likewise (lazy): the reader runs where the data lives.  Parity unpinned: the reference module imports h5py
and skimage at load time and cannot be imported here; the transforms are restated from its text and tested on their own semantics.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from .brats19 import CenterCrop, RandomCrop, RandomNoise, RandomRotFlip, TwoStreamBatchSampler   # noqa: F401  (re-exported)


class Pancreas(Dataset):
    def __init__(self, base_dir=None, split="train", num=None, transform=None):
        self._base_dir = base_dir
        self.transform = transform
        path = self._base_dir + ("/train.list" if split == "train" else "/test.list")      # 'test' and 'val' share test.list (:24-29)
        with open(path, "r") as f:
            self.image_list = [item.replace("\n", "") for item in f.readlines()]
        if num is not None:
            self.image_list = self.image_list[:num]

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, idx):
        import h5py   # lazy: absent in the build image
        with h5py.File(self._base_dir + "/Pancreas_data/{}".format(self.image_list[idx]), "r") as h5f:
            sample = {"image": h5f["image"][:], "label": h5f["label"][:].astype(np.uint8)}
        return self.transform(sample) if self.transform else sample


class Resize(object):
    """:50-63 -- trilinear resize of the image, nearest of the (boolean) label; the result must stay a two-valued mask"""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        from skimage import transform as sk_trans   # lazy: not in this image
        image, label = sample["image"], sample["label"].astype(bool)
        image = sk_trans.resize(image, self.output_size, order=1, mode="constant", cval=0)
        label = sk_trans.resize(label, self.output_size, order=0)
        assert np.max(label) == 1 and np.min(label) == 0 and np.unique(label).shape[0] == 2
        return {"image": image, "label": label}


class CreateOnehotLabel(object):
    """adds 'onehot_label' (num_classes, w, h, d) float32 (:182-192)"""

    def __init__(self, num_classes):
        self.num_classes = num_classes

    def __call__(self, sample):
        label = sample["label"]
        if torch.is_tensor(label):
            onehot = torch.stack([(label == i) for i in range(self.num_classes)]).to(torch.float32)
        else:
            onehot = np.stack([(label == i) for i in range(self.num_classes)]).astype(np.float32)
        return {"image": sample["image"], "label": label, "onehot_label": onehot}


class ToTensor(object):
    """image -> (1, w, h, d) float32, label -> long, 'onehot_label' -> long when present (:195-207); tensors stay on their device"""

    def __call__(self, sample):
        image, label = sample["image"], sample["label"]
        if torch.is_tensor(image):
            out = {"image": image.reshape(1, *image.shape).float(), "label": label.long()}
        else:
            image = image.reshape(1, image.shape[0], image.shape[1], image.shape[2]).astype(np.float32)
            out = {"image": torch.from_numpy(image), "label": torch.from_numpy(np.ascontiguousarray(label)).long()}
        if "onehot_label" in sample:
            out["onehot_label"] = torch.as_tensor(sample["onehot_label"]).long()
        return out
