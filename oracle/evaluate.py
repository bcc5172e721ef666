"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's sliding-window evaluator and per-case metrics:
  * test_single_case          code/utils/test_3d_patch.py:293-351
  * calculate_metric_percase  code/utils/test_3d_patch.py:496-508 (dice, jaccard, hd95, asd through medpy)

Parity status: UNPINNED.  code/utils/test_3d_patch.py cannot be imported here (cv2, h5py, natsort, nibabel, medpy and
skimage are absent and its test_single_case calls ``.cuda()``), and the repository holds no evaluator fixtures; this
file follows the reference source line by line instead.  medpy (third party, not vendored, no version pinned by the
reference) is restated from its published algorithm (medpy.metric.binary: dc, jc, __surface_distances, hd95, asd)
on scipy.ndimage.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage


def test_single_case(net_logits, image, stride_xy, stride_z, patch_size, num_classes=1):
    """net_logits: callable (1,1,p0,p1,p2) float tensor -> logits (1,2,p0,p1,p2) (the reference's ``model(x)[1]``).

    Reproduces the reference including its quirk: the class-1 probability is added to EVERY channel of score_map (:338-339) and
    the label map is score_map[0] > 0.5 (:345)."""
    w, h, d = image.shape
    add_pad = False
    pads = []
    for size, p in zip((w, h, d), patch_size):          # :297-316
        if size < p:
            pad = p - size
            add_pad = True
        else:
            pad = 0
        pads.append((pad // 2, pad - pad // 2))
    if add_pad:
        image = np.pad(image, pads, mode="constant", constant_values=0)
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1   # :319-321
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score_map = np.zeros((num_classes,) + image.shape, dtype=np.float32)
    cnt = np.zeros(image.shape, dtype=np.float32)
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                zs = min(stride_z * z, dd - patch_size[2])
                patch = image[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]]
                t = torch.from_numpy(patch[None, None].astype(np.float32))
                with torch.no_grad():
                    prob = F.softmax(net_logits(t), dim=1).numpy()[0, 1]
                sl = (slice(xs, xs + patch_size[0]), slice(ys, ys + patch_size[1]), slice(zs, zs + patch_size[2]))
                score_map[(slice(None),) + sl] += prob
                cnt[sl] += 1
    score_map = score_map / cnt[None]
    label_map = (score_map[0] > 0.5).astype(int)
    if add_pad:
        sl = tuple(slice(lo, lo + n) for (lo, _), n in zip(pads, (w, h, d)))
        label_map = label_map[sl]
        score_map = score_map[(slice(None),) + sl]
    return label_map, score_map


# ---- medpy.metric.binary, restated ------------------------------------------------------------
def dc(result, reference):
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    s = np.count_nonzero(result) + np.count_nonzero(reference)
    return 2.0 * inter / float(s) if s else 0.0


def jc(result, reference):
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    union = np.count_nonzero(result | reference)
    return float(inter) / float(union)          # ZeroDivisionError for two empty masks, as medpy


def surface_distances(result, reference, voxelspacing=None, connectivity=1):
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    footprint = ndimage.generate_binary_structure(result.ndim, connectivity)
    if 0 == np.count_nonzero(result):
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if 0 == np.count_nonzero(reference):
        raise RuntimeError("The second supplied array does not contain any binary object.")
    result_border = result ^ ndimage.binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ ndimage.binary_erosion(reference, structure=footprint, iterations=1)
    dt = ndimage.distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95(result, reference, voxelspacing=None, connectivity=1):
    hd1 = surface_distances(result, reference, voxelspacing, connectivity)
    hd2 = surface_distances(reference, result, voxelspacing, connectivity)
    return np.percentile(np.hstack((hd1, hd2)), 95)


def asd(result, reference, voxelspacing=None, connectivity=1):
    return surface_distances(result, reference, voxelspacing, connectivity).mean()


def calculate_metric_percase(pred, gt):
    """code/utils/test_3d_patch.py:496-508: hd95 / asd are reported as 0 for an empty ground truth."""
    dice, jac = dc(pred, gt), jc(pred, gt)
    if gt.sum() == 0:
        return dice, jac, 0.0, 0.0
    return dice, jac, hd95(pred, gt), asd(pred, gt)
