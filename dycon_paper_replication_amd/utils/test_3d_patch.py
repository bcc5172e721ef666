"""Sliding-window evaluation with the reference's names and argument meaning (code/utils/test_3d_patch.py).

    test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=1)   :293-351
    calculate_metric_percase(pred, gt) -> (dice, jaccard, hd95, asd)                 :496-508
    test_all_case(model, cases, ...)                                                 :251-291

The windows run through the HIP network in batches; score / count maps, the final threshold and the overlap counts behind Dice
and Jaccard stay on the MI355X (csrc/eval.hip).  HD95 / ASD are surface-distance metrics of medpy (absent here, not vendored by
the reference): they are restated on scipy.ndimage on the host ("parity unpinned", see oracle/evaluate.py) -- they are
per-case diagnostics off the hot path.  The reference reads h5 files (h5py is not available in this image): `test_all_case`
takes (image, label) arrays, or h5 paths when h5py can be imported.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib, ops


def _windows(shape, patch_size, stride_xy, stride_z):
    ww, hh, dd = shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1      # :319-321
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    out = []
    for x in range(sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(sz):
                out.append((xs, ys, min(stride_z * z, dd - patch_size[2])))
    return out


def test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=1, batch_size=4, device=None):
    """image: (w, h, d) numpy array or tensor.  Returns (label_map int64 (w,h,d), score_map float32 (num_classes,w,h,d)) as numpy,
    like the reference; the class-1 probability fills every channel of score_map (the reference's broadcast, :338-339).

    ``batch_size`` windows go through the network per launch sequence (the reference runs them one by one)."""
    dev = torch.device(device) if device is not None else next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("test_single_case runs on the MI355X only: move the model to 'cuda'")
    img = torch.as_tensor(np.asarray(image), dtype=torch.float32)
    w, h, d = img.shape
    pads = []
    for size, p in zip((w, h, d), patch_size):                 # :297-316: centre-pad volumes smaller than the window
        pad = max(p - size, 0)
        pads.append((pad // 2, pad - pad // 2))
    add_pad = any(lo or hi for lo, hi in pads)
    vol = img.to(dev)
    if add_pad:
        vol = torch.nn.functional.pad(vol, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    vol = vol.contiguous()
    ww, hh, dd = vol.shape
    wins = _windows((ww, hh, dd), patch_size, stride_xy, stride_z)
    score = torch.zeros((ww, hh, dd), dtype=torch.float32, device=dev)
    cnt = torch.zeros_like(score)
    p0, p1, p2 = patch_size
    stream = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
    was_training = model.training
    was_static = getattr(model, "weights_static", None)
    model.eval()
    if was_static is not None:
        model.weights_static = True      # no parameter changes between the windows: pack the weights once (networks/_base.py)
    try:
        with torch.no_grad():
            for i in range(0, len(wins), batch_size):
                chunk = wins[i:i + batch_size]
                patches = torch.stack([vol[x:x + p0, y:y + p1, z:z + p2] for x, y, z in chunk]).unsqueeze(1).contiguous()
                logits = model(patches)[1]                                        # (nb, 2, p0, p1, p2), channels-last-3D view
                lg = logits.permute(0, 2, 3, 4, 1).contiguous().float()          # NDHWC (free for the HIP nets' outputs)
                org = torch.tensor(chunk, dtype=torch.int32).to(dev)
                _lib.call("dycon_sw_accumulate", lg.data_ptr(), len(chunk), p0, p1, p2, org.data_ptr(), score.data_ptr(),
                          cnt.data_ptr(), ww, hh, dd, stream())
    finally:
        model.train(was_training)
        if was_static is not None:
            model.weights_static = was_static
    label = torch.empty((ww, hh, dd), dtype=torch.uint8, device=dev)
    prob = torch.empty_like(score)
    _lib.call("dycon_sw_finalize", score.data_ptr(), cnt.data_ptr(), score.numel(), 0.5, label.data_ptr(), prob.data_ptr(), stream())
    if add_pad:
        sl = tuple(slice(lo, lo + n) for (lo, _), n in zip(pads, (w, h, d)))
        label, prob = label[sl], prob[sl]
    label_map = label.cpu().numpy().astype(np.int64)
    score_map = np.broadcast_to(prob.cpu().numpy()[None], (num_classes,) + label_map.shape).copy()
    return label_map, score_map


def overlap_counts(pred, gt):
    """(|pred|, |gt|, |pred & gt|) on the device.  pred / gt: CUDA tensors (uint8 / int64) or numpy arrays."""
    dev = pred.device if torch.is_tensor(pred) and pred.is_cuda else torch.device("cuda:0")
    p = torch.as_tensor(np.asarray(pred) if not torch.is_tensor(pred) else pred).to(dev).ne(0).to(torch.uint8).contiguous()
    g = torch.as_tensor(np.asarray(gt) if not torch.is_tensor(gt) else gt).to(dev)
    g = (g if g.dtype in (torch.uint8, torch.int64) else g.ne(0).to(torch.uint8)).contiguous()
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    _lib.call("dycon_binary_overlap", p.data_ptr(), g.data_ptr(), g.element_size(), p.numel(), out.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    return tuple(int(v) for v in out.tolist())


def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
    from scipy import ndimage
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    footprint = ndimage.generate_binary_structure(result.ndim, connectivity)
    if 0 == np.count_nonzero(result):
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if 0 == np.count_nonzero(reference):
        raise RuntimeError("The second supplied array does not contain any binary object.")
    result_border = result ^ ndimage.binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ ndimage.binary_erosion(reference, structure=footprint, iterations=1)
    return ndimage.distance_transform_edt(~reference_border, sampling=voxelspacing)[result_border]


def calculate_metric_percase(pred, gt):
    """(dice, jaccard, hd95, asd) of one case, medpy.metric.binary semantics (code/utils/test_3d_patch.py:496-508)."""
    n_p, n_g, n_i = overlap_counts(pred, gt)
    dice = 2.0 * n_i / float(n_p + n_g) if n_p + n_g else 0.0
    jc = float(n_i) / float(n_p + n_g - n_i)               # ZeroDivisionError for two empty masks, as medpy
    if n_g == 0:
        return dice, jc, 0.0, 0.0
    p = np.asarray(pred.cpu() if torch.is_tensor(pred) else pred) != 0
    g = np.asarray(gt.cpu() if torch.is_tensor(gt) else gt) != 0
    hd1, hd2 = _surface_distances(p, g), _surface_distances(g, p)
    return dice, jc, float(np.percentile(np.hstack((hd1, hd2)), 95)), float(hd1.mean())


def getLargestCC(segmentation):
    """Largest connected component of a label map (code/utils/test_3d_patch.py:19-26; skimage.measure.label's default full
    connectivity -- 26 neighbours in 3-D -- on scipy.ndimage, skimage is not vendored by the reference).  An empty map is returned
    unchanged.  The `nms` option of the test_all_case family."""
    from scipy import ndimage
    seg = np.asarray(segmentation)
    labels, n = ndimage.label(seg, structure=np.ones((3,) * seg.ndim, dtype=bool))
    if n == 0:
        return segmentation
    return labels == np.argmax(np.bincount(labels.flat)[1:]) + 1


def _read_case(case, label_key):
    if isinstance(case, (str, bytes)):
        import h5py   # not available in this image; present where the datasets are
        with h5py.File(case, "r") as f:
            return f["image"][:], f[label_key][:]
    return case


def test_all_case(model, cases, num_classes, patch_size=(96, 96, 64), stride_xy=16, stride_z=4, batch_size=4, metric_detail=0,
                  nms=0, preproc_fn=None, test_save_path=None, label_key="label"):
    """cases: iterable of (image, label) arrays, or of h5 paths with 'image' / `label_key` datasets (needs h5py).
    Returns the mean (dice, jaccard, hd95, asd) over the cases; nms keeps the largest connected component of each prediction,
    preproc_fn is applied to the image first, test_save_path receives `../performance.txt` (code/utils/test_3d_patch.py:251-291)."""
    total = np.zeros(4)
    n = 0
    for case in cases:
        image, label = _read_case(case, label_key)
        if preproc_fn is not None:
            image = preproc_fn(image)
        pred, _ = test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=num_classes, batch_size=batch_size)
        if nms:
            pred = getLargestCC(pred)
        m = (0.0, 0.0, 0.0, 0.0) if pred.sum() == 0 else calculate_metric_percase(pred, np.asarray(label))
        if metric_detail:
            print("%02d,\t%.5f, %.5f, %.5f, %.5f" % ((n,) + tuple(m)))
        total += np.asarray(m)
        n += 1
    avg = total / max(n, 1)
    if test_save_path is not None:
        with open(test_save_path + "../performance.txt", "w") as f:
            f.writelines("average metric is {} \n".format(avg))
    return avg


def _case_list(root_path, list_name, pattern):
    import os
    with open(os.path.join(root_path, list_name), "r") as f:
        return [pattern.format(root=root_path, case=line.strip()) for line in f if line.strip()]


def _mean_dice(model, cases, num_classes, patch_size, stride_xy, stride_z, label_key, transpose=None):
    """validation during training: mean Dice of the sliding-window predictions (the var_all_case_* family, :52-74, :120-141, :188-209).
    transpose: axis permutation applied to image AND label before the sliding window (BraTS19: (2, 1, 0), :64-65 -- the orientation
    the training patches have after dataloaders.brats19.SagittalToAxial)."""
    total, n = 0.0, 0
    for case in cases:
        image, label = _read_case(case, label_key)
        if transpose is not None:
            image, label = np.transpose(np.asarray(image), transpose), np.transpose(np.asarray(label), transpose)
        pred, _ = test_single_case(model, image, stride_xy, stride_z, patch_size, num_classes=num_classes)
        if pred.sum() != 0:
            n_p, n_g, n_i = overlap_counts(pred, np.asarray(label))
            total += 2.0 * n_i / float(n_p + n_g)
        n += 1
    return total / max(n, 1)


# per-dataset wrappers with the reference's names, defaults and on-disk layouts
def var_all_case_BraTS19(model, root_path, num_classes, patch_size=(96, 96, 64), stride_xy=16, stride_z=4):
    return _mean_dice(model, _case_list(root_path, "val.txt", "{root}/data/{case}.h5"), num_classes, patch_size, stride_xy, stride_z, "label",
                      transpose=(2, 1, 0))


def var_all_case_Pancreas(model, root_path, num_classes, patch_size=(112, 112, 80), stride_xy=18, stride_z=4):
    return _mean_dice(model, _case_list(root_path, "test1.list", "{root}/Pancreas_data/{case}"), num_classes, patch_size, stride_xy, stride_z, "label")


def var_all_case_ISLES22(root_path, model, num_classes, device=None, patch_size=(96, 96, 64), stride_xy=16, stride_z=4):
    return _mean_dice(model, _case_list(root_path, "val.list", "{root}/{case}.h5"), num_classes, patch_size, stride_xy, stride_z, "mask")


def test_all_case_BraTS19(model, image_list, num_classes, patch_size=(96, 96, 64), stride_xy=16, stride_z=4, save_result=True,
                          test_save_path=None, preproc_fn=None, metric_detail=0, nms=0):
    return test_all_case(model, image_list, num_classes, patch_size, stride_xy, stride_z, metric_detail=metric_detail, nms=nms,
                         preproc_fn=preproc_fn, test_save_path=test_save_path)


def test_all_case_Pancreas(model, image_list, num_classes, device=None, patch_size=(96, 96, 64), stride_xy=16, stride_z=4,
                           save_result=True, test_save_path=None, preproc_fn=None, metric_detail=0, nms=0):
    return test_all_case(model, image_list, num_classes, patch_size, stride_xy, stride_z, metric_detail=metric_detail, nms=nms,
                         preproc_fn=preproc_fn, test_save_path=test_save_path)


def test_all_case_ISLES22(model, image_list, num_classes, patch_size=(96, 96, 64), stride_xy=16, stride_z=4, save_result=True,
                          test_save_path=None, preproc_fn=None, metric_detail=0, nms=0):
    return test_all_case(model, image_list, num_classes, patch_size, stride_xy, stride_z, metric_detail=metric_detail, nms=nms,
                         preproc_fn=preproc_fn, test_save_path=test_save_path, label_key="mask")


for _f in (test_all_case, test_all_case_BraTS19, test_all_case_Pancreas, test_all_case_ISLES22):
    _f.__test__ = False        # reference names, not pytest cases
