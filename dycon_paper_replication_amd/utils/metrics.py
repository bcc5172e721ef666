"""Train-time metrics with the reference's names (code/utils/metrics.py; used at code/train_DyCON_BraTS19.py:385-395) -- SURVEY 8f-4.

The reference computes, EVERY iteration, a batch Dice on the GPU and a medpy HD95 per sample on the host (a device->host copy
of the binary volumes plus two distance transforms per sample: more host time than the whole HIP step).  Here

  * `batch_dice_from_logits` / `compute_dice` / `compute_jaccard` count on the device (csrc/eval.hip) and return device tensors
    without a host sync;
  * `compute_hd95` keeps the reference's semantics (max_dist for an empty mask) on scipy ("parity unpinned": medpy absent);
  * `AsyncTrainMetrics` takes it off the critical path: every `every`-th iteration the binary maps are copied to pinned memory
    asynchronously and a worker thread computes HD95 while the GPU trains on.
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from .. import _lib
from .test_3d_patch import _surface_distances


def _counts(logits_ndhwc2, labels):
    B = logits_ndhwc2.shape[0]
    V = logits_ndhwc2.numel() // (2 * B)
    lab = labels.contiguous()
    if lab.dtype not in (torch.uint8, torch.int64):
        lab = lab.ne(0).to(torch.uint8)
    out = torch.zeros((B, 3), dtype=torch.int64, device=logits_ndhwc2.device)
    _lib.call("dycon_batch_overlap", logits_ndhwc2.data_ptr(), lab.data_ptr(), lab.element_size(), B, V, out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return out.double()


def batch_dice_from_logits(logits, labels):
    """logits (B, 2, D, H, W) as the HIP nets return them (channels-last-3D view) or (B, D, H, W, 2); labels (B, D, H, W).
    Per-sample Dice of (softmax(logits)[:, 1] > 0.5) against labels, = metrics.compute_dice(outputs_bin, label) of the reference
    (train_DyCON_BraTS19.py:387-389), as a device tensor (no sync)."""
    lg = logits if logits.shape[-1] == 2 and logits.dim() == 5 and logits.shape[1] != 2 else logits.permute(0, 2, 3, 4, 1)
    c = _counts(lg.contiguous().float(), labels)
    return 2.0 * c[:, 2] / (c[:, 0] + c[:, 1] + 1e-8)


def _binary_counts(output, label):
    out = torch.as_tensor(output)
    lab = torch.as_tensor(label)
    if not out.is_cuda:
        out, lab = out.cuda(), lab.cuda()
    B = out.shape[0]
    # binary maps -> two-channel "logits" whose argmax is the map: reuse the fused kernel
    lg = torch.stack([torch.zeros_like(out, dtype=torch.float32), out.float() - 0.5], dim=-1).reshape(B, -1, 2).contiguous()
    return _counts(lg, lab.reshape(B, -1))


def compute_dice(output, label):
    """Batch-wise Dice of binary volumes (B, ...): 2|A&B| / (|A| + |B| + 1e-8)  (utils/metrics.py:85-95)."""
    c = _binary_counts(output, label)
    return 2.0 * c[:, 2] / (c[:, 0] + c[:, 1] + 1e-8)


def compute_jaccard(output, label):
    """Batch-wise Jaccard / IoU (utils/metrics.py:98-109)."""
    c = _binary_counts(output, label)
    return c[:, 2] / (c[:, 0] + c[:, 1] - c[:, 2] + 1e-8)


def cal_dice(prediction, label, num=2):
    """Multi-class Dice on label maps, classes 1..num-1 (utils/metrics.py:14-27); numpy in, numpy out."""
    total = np.zeros(num - 1)
    for i in range(1, num):
        p, g = (np.asarray(prediction) == i), (np.asarray(label) == i)
        total[i - 1] += 2 * np.sum(p & g) / (np.sum(p) + np.sum(g) + 1e-8)
    return total


def compute_hd95(pred, target, max_dist):
    """Per-sample HD95 with the reference's conventions (utils/metrics.py:112-131): max_dist when either mask is empty."""
    pred = pred.cpu().numpy() if torch.is_tensor(pred) else np.asarray(pred)
    target = target.cpu().numpy() if torch.is_tensor(target) else np.asarray(target)
    scores = []
    for p, t in zip(pred, target):
        if np.sum(p) == 0 or np.sum(t) == 0:
            scores.append(max_dist)
        else:
            hd1, hd2 = _surface_distances(p != 0, t != 0), _surface_distances(t != 0, p != 0)
            scores.append(float(np.percentile(np.hstack((hd1, hd2)), 95)))
    return scores


class AsyncTrainMetrics:
    """Per-iteration Dice on the device, HD95 every `every` iterations on a worker thread.

        m = AsyncTrainMetrics(every=50)
        m.update(iter_num, out["s_logits"], label)     # never blocks on the GPU
        m.latest()  ->  {"iter": .., "dice": tensor (device), "hd95": (iter, mean) or None}
    """

    def __init__(self, every=50):
        self.every = every
        self._q = queue.Queue(maxsize=2)
        self._hd = None
        self._dice = None
        self._iter = -1
        self._t = threading.Thread(target=self._work, daemon=True)
        self._t.start()

    def _work(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            it, ev, pred, lab, max_dist = item
            ev.synchronize()
            self._hd = (it, float(np.mean(compute_hd95(pred.numpy(), lab.numpy(), max_dist))))

    def update(self, iter_num, logits, labels):
        self._iter = iter_num
        self._dice = batch_dice_from_logits(logits, labels)
        if self.every and iter_num % self.every == 0 and not self._q.full():
            lg = logits if logits.shape[-1] == 2 and logits.shape[1] != 2 else logits.permute(0, 2, 3, 4, 1)
            pred = (lg[..., 1] > lg[..., 0]).to(torch.uint8)
            ph = torch.empty(pred.shape, dtype=torch.uint8).pin_memory()
            lh = torch.empty(labels.shape, dtype=torch.uint8).pin_memory()
            ph.copy_(pred, non_blocking=True)
            lh.copy_(labels.ne(0).to(torch.uint8), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._q.put((iter_num, ev, ph, lh, float(np.linalg.norm(pred.shape[1:]))))

    def latest(self):
        return {"iter": self._iter, "dice": self._dice, "hd95": self._hd}

    def close(self):
        self._q.put(None)
