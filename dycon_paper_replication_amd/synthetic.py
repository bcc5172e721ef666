"""Synthetic DyCON batches: z-scored volumes ~ N(0,1) and lesion-like labels
(union of 1-3 random ellipsoids per volume, a few % foreground) for EVERY sample -- the reference
builds the FeCL mask from the labels of the unlabelled half too (train_DyCON_BraTS19.py:326)."""
import numpy as np
import torch


def blob_labels(rng: np.random.Generator, B, D, H, W):
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    lab = np.zeros((B, D, H, W), np.uint8)
    for b in range(B):
        for _ in range(int(rng.integers(1, 4))):
            c = rng.uniform(0.25, 0.75, 3) * (D, H, W)
            r = rng.uniform(0.12, 0.3, 3) * (D, H, W)
            lab[b] |= (((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 <= 1).astype(np.uint8)
    return lab


def make_batch(seed, B, patch=(96, 96, 96)):
    """Returns CPU tensors: volume (B,1,D,H,W) fp32, label (B,D,H,W) int64, noise (B,1,D,H,W) fp32."""
    rng = np.random.default_rng(seed)
    D, H, W = patch
    vol = torch.from_numpy(rng.standard_normal((B, 1, D, H, W)).astype(np.float32))
    lab = torch.from_numpy(blob_labels(rng, B, D, H, W).astype(np.int64))
    noise = torch.clamp(torch.from_numpy(rng.standard_normal((B, 1, D, H, W)).astype(np.float32)) * 0.1, -0.2, 0.2)
    return vol, lab, noise
