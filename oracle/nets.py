"""Oracle networks: functional restatement of the reference's 3D V-Net and 3D U-Net.

Test infrastructure only (see oracle/__init__.py).  Both nets are written as
pure functions of a flat ``{state_dict key: tensor}`` mapping so that the very
same weights can be fed to the reference module (when generating fixtures) and
to the HIP path (when testing parity).

Reference sites followed:
  * V-Net blocks / topology ........ code/networks/VNet.py:5-31, 67-118, 145-239
  * U-Net blocks ................... code/networks/utils.py:99-123, 260-276
  * U-Net topology + feature head .. code/networks/UNet3D_contrastive.py:207-316
  * weight init .................... code/networks/networks_other.py:40-49

The reference V-Net has no DyCON head (VNet.forward returns one tensor and the
factory raises TypeError); the head wired on here is the
U-Net's projection head applied to the V-Net bottleneck x5 -- "parity unpinned
(wiring)", every layer of it is pinned individually.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

VNET_STAGES = {  # block name -> (n_stages, c_in, c_out); n_filters = 16
    "block_one": (1, None, 16), "block_two": (2, 32, 32), "block_three": (3, 64, 64),
    "block_four": (3, 128, 128), "block_five": (3, 256, 256), "block_six": (3, 128, 128),
    "block_seven": (3, 64, 64), "block_eight": (2, 32, 32), "block_nine": (1, 16, 16),
}
VNET_DOWN = {"block_one_dw": (16, 32), "block_two_dw": (32, 64), "block_three_dw": (64, 128),
             "block_four_dw": (128, 256)}
VNET_UP = {"block_five_up": (256, 128), "block_six_up": (128, 64), "block_seven_up": (64, 32),
           "block_eight_up": (32, 16)}
UNET_FILTERS = (16, 32, 64, 128, 256)


# --------------------------------------------------------------------------------------
# deterministic parameter construction (numpy PCG64: stable across torch versions)
# --------------------------------------------------------------------------------------
def _kaiming(rng, shape):
    fan_in = int(np.prod(shape[1:]))
    return torch.from_numpy((rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32))


def _bias(rng, n, fan_in):
    b = 1.0 / math.sqrt(fan_in)
    return torch.from_numpy(rng.uniform(-b, b, size=(n,)).astype(np.float32))


def _projection_params(rng, p: Params, c_in: int = 256):
    p["projection.0.weight"] = _kaiming(rng, (512, c_in, 1, 1, 1))
    p["projection.0.bias"] = _bias(rng, 512, c_in)
    p["projection.1.weight"] = torch.from_numpy((1.0 + 0.02 * rng.standard_normal(512)).astype(np.float32))
    p["projection.1.bias"] = torch.zeros(512)
    p["projection.1.running_mean"] = torch.zeros(512)
    p["projection.1.running_var"] = torch.ones(512)
    p["projection.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    p["projection.3.weight"] = _kaiming(rng, (256, 512, 1, 1, 1))
    p["projection.3.bias"] = _bias(rng, 256, 512)
    p["projection.4.weight"] = torch.from_numpy((1.0 + 0.02 * rng.standard_normal(256)).astype(np.float32))
    p["projection.4.bias"] = torch.zeros(256)
    p["projection.4.running_mean"] = torch.zeros(256)
    p["projection.4.running_var"] = torch.ones(256)
    p["projection.4.num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def make_vnet_params(seed: int, in_ch: int = 1, n_classes: int = 2, normalization: str = "groupnorm",
                     head: bool = True) -> Params:
    """Kaiming-normal conv weights, GN affine gamma=1+small noise / beta=small noise (so the
    affine path is exercised), in the reference V-Net's state_dict key order."""
    rng = np.random.default_rng(seed)
    p: Params = {}
    has_norm = normalization != "none"
    step = 3 if has_norm else 2

    def add_norm(prefix, c):
        if normalization in ("groupnorm", "batchnorm"):
            p[prefix + ".weight"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32))
            p[prefix + ".bias"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))

    def add_block(name):
        n, cin, cout = VNET_STAGES[name]
        cin = in_ch if cin is None else cin
        for i in range(n):
            ci = cin if i == 0 else cout
            p[f"{name}.conv.{step * i}.weight"] = _kaiming(rng, (cout, ci, 3, 3, 3))
            p[f"{name}.conv.{step * i}.bias"] = _bias(rng, cout, ci * 27)
            if has_norm:
                add_norm(f"{name}.conv.{step * i + 1}", cout)

    def add_down(name):
        cin, cout = VNET_DOWN[name]
        p[f"{name}.conv.0.weight"] = _kaiming(rng, (cout, cin, 2, 2, 2))
        p[f"{name}.conv.0.bias"] = _bias(rng, cout, cin * 8)
        if has_norm:
            add_norm(f"{name}.conv.1", cout)

    def add_up(name):
        cin, cout = VNET_UP[name]
        w = rng.standard_normal((cin, cout, 2, 2, 2)) * math.sqrt(2.0 / (cout * 8))
        p[f"{name}.conv.0.weight"] = torch.from_numpy(w.astype(np.float32))
        p[f"{name}.conv.0.bias"] = _bias(rng, cout, cout * 8)
        if has_norm:
            add_norm(f"{name}.conv.1", cout)

    for blk, dw in (("block_one", "block_one_dw"), ("block_two", "block_two_dw"),
                    ("block_three", "block_three_dw"), ("block_four", "block_four_dw")):
        add_block(blk)
        add_down(dw)
    for blk, up in (("block_five", "block_five_up"), ("block_six", "block_six_up"),
                    ("block_seven", "block_seven_up"), ("block_eight", "block_eight_up")):
        add_block(blk)
        add_up(up)
    add_block("block_nine")
    p["out_conv.weight"] = _kaiming(rng, (n_classes, 16, 1, 1, 1))
    p["out_conv.bias"] = _bias(rng, n_classes, 16)
    if head:
        _projection_params(rng, p)
    return p


def make_unet_params(seed: int, in_ch: int = 1, n_classes: int = 2) -> Params:
    rng = np.random.default_rng(seed)
    p: Params = {}
    f = UNET_FILTERS

    def add_unetconv(prefix, cin, cout):
        p[f"{prefix}.conv1.0.weight"] = _kaiming(rng, (cout, cin, 3, 3, 3))
        p[f"{prefix}.conv1.0.bias"] = _bias(rng, cout, cin * 27)
        p[f"{prefix}.conv2.0.weight"] = _kaiming(rng, (cout, cout, 3, 3, 3))
        p[f"{prefix}.conv2.0.bias"] = _bias(rng, cout, cout * 27)

    add_unetconv("conv1", in_ch, f[0])
    add_unetconv("conv2", f[0], f[1])
    add_unetconv("conv3", f[1], f[2])
    add_unetconv("conv4", f[2], f[3])
    add_unetconv("center", f[3], f[4])
    add_unetconv("up_concat4.conv", f[4] + f[3], f[3])
    add_unetconv("up_concat3.conv", f[3] + f[2], f[2])
    add_unetconv("up_concat2.conv", f[2] + f[1], f[1])
    add_unetconv("up_concat1.conv", f[1] + f[0], f[0])
    p["final.weight"] = _kaiming(rng, (n_classes, f[0], 1, 1, 1))
    p["final.bias"] = _bias(rng, n_classes, f[0])
    p["out_conv2.weight"] = _kaiming(rng, (n_classes, f[0], 1, 1, 1))
    p["out_conv2.bias"] = _bias(rng, n_classes, f[0])
    _projection_params(rng, p, f[4])
    return p


def trainable(p: Params) -> Dict[str, torch.Tensor]:
    """Parameters (registration order), i.e. the state_dict minus BatchNorm buffers."""
    return {k: v for k, v in p.items()
            if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))}


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _norm(x, p, prefix, normalization):
    if normalization == "groupnorm":      # VNet.py:19-20 -- GroupNorm(16, C), eps 1e-5, affine
        return F.group_norm(x, 16, p[prefix + ".weight"], p[prefix + ".bias"], 1e-5)
    if normalization == "instancenorm":   # VNet.py:21-22 -- InstanceNorm3d default: no affine
        return F.instance_norm(x, eps=1e-5)
    if normalization == "batchnorm":      # VNet.py:17-18 -- batch statistics in train mode
        return F.batch_norm(x, None, None, p[prefix + ".weight"], p[prefix + ".bias"], True, 0.1, 1e-5)
    assert normalization == "none"
    return x


def vnet_conv_block(x, p, name, normalization="groupnorm"):
    """VNet.ConvBlock (VNet.py:5-31): n x [conv3 pad1 -> norm -> ReLU]."""
    step = 3 if normalization != "none" else 2
    i = 0
    while f"{name}.conv.{step * i}.weight" in p:
        x = F.conv3d(x, p[f"{name}.conv.{step * i}.weight"], p[f"{name}.conv.{step * i}.bias"], padding=1)
        x = F.relu(_norm(x, p, f"{name}.conv.{step * i + 1}", normalization))
        i += 1
    return x


def vnet_down_block(x, p, name, normalization="groupnorm"):
    """VNet.DownsamplingConvBlock (VNet.py:67-91): conv k2 s2 -> norm -> ReLU."""
    x = F.conv3d(x, p[f"{name}.conv.0.weight"], p[f"{name}.conv.0.bias"], stride=2)
    return F.relu(_norm(x, p, f"{name}.conv.1", normalization))


def vnet_up_block(x, p, name, normalization="groupnorm"):
    """VNet.UpsamplingDeconvBlock (VNet.py:94-118): convT k2 s2 -> norm -> ReLU."""
    x = F.conv_transpose3d(x, p[f"{name}.conv.0.weight"], p[f"{name}.conv.0.bias"], stride=2)
    return F.relu(_norm(x, p, f"{name}.conv.1", normalization))


def projection_head(center, p, scale_factor: int, bn_training: bool = True, update_buffers: bool = False):
    """UNet3D_contrastive.py:261-267, 308-310: trilinear x scale (align_corners=True) ->
    conv1x1 256->512 -> BN -> ReLU -> conv1x1 512->256 -> BN."""
    x = F.interpolate(center, scale_factor=scale_factor, mode="trilinear", align_corners=True)

    def bn(x, prefix):
        rm = p[prefix + ".running_mean"]
        rv = p[prefix + ".running_var"]
        if bn_training and not update_buffers:
            rm, rv = rm.clone(), rv.clone()
        elif bn_training and prefix + ".num_batches_tracked" in p:
            p[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"], bn_training, 0.1, 1e-5)

    x = F.conv3d(x, p["projection.0.weight"], p["projection.0.bias"])
    x = F.relu(bn(x, "projection.1"))
    x = F.conv3d(x, p["projection.3.weight"], p["projection.3.bias"])
    return bn(x, "projection.4")


def dropout3d_mask(x, mask: Optional[torch.Tensor], p_drop: float):
    """Channel-wise dropout with an explicit keep-mask of shape (B, C) in {0,1}
    (nn.Dropout3d semantics: y = x * keep / (1-p)).  mask=None -> identity."""
    if mask is None:
        return x
    return x * (mask.to(x.dtype) / (1.0 - p_drop)).view(x.shape[0], x.shape[1], 1, 1, 1)


def dropout_mask(x, mask: Optional[torch.Tensor], p_drop: float):
    """Element-wise dropout with an explicit keep-mask shaped like x."""
    if mask is None:
        return x
    return x * (mask.to(x.dtype) / (1.0 - p_drop))


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
def vnet_forward(x, p: Params, scale_factor: int = 2, normalization: str = "groupnorm",
                 drop5: Optional[torch.Tensor] = None, drop9: Optional[torch.Tensor] = None,
                 bn_training: bool = True, update_buffers: bool = False
                 ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """V-Net encoder/decoder (VNet.py:180-228) + DyCON head.  Returns (tanh(logits), logits, features).

    drop5/drop9: explicit (B,C) keep masks for the two Dropout3d(0.5) sites (VNet.py:196, 226)."""
    nz = normalization
    x1 = vnet_conv_block(x, p, "block_one", nz)
    x2 = vnet_conv_block(vnet_down_block(x1, p, "block_one_dw", nz), p, "block_two", nz)
    x3 = vnet_conv_block(vnet_down_block(x2, p, "block_two_dw", nz), p, "block_three", nz)
    x4 = vnet_conv_block(vnet_down_block(x3, p, "block_three_dw", nz), p, "block_four", nz)
    x5 = vnet_conv_block(vnet_down_block(x4, p, "block_four_dw", nz), p, "block_five", nz)
    x5 = dropout3d_mask(x5, drop5, 0.5)
    u = vnet_up_block(x5, p, "block_five_up", nz) + x4
    u = vnet_up_block(vnet_conv_block(u, p, "block_six", nz), p, "block_six_up", nz) + x3
    u = vnet_up_block(vnet_conv_block(u, p, "block_seven", nz), p, "block_seven_up", nz) + x2
    u = vnet_up_block(vnet_conv_block(u, p, "block_eight", nz), p, "block_eight_up", nz) + x1
    x9 = dropout3d_mask(vnet_conv_block(u, p, "block_nine", nz), drop9, 0.5)
    logits = F.conv3d(x9, p["out_conv.weight"], p["out_conv.bias"])
    feats = None
    if "projection.0.weight" in p:
        feats = projection_head(x5, p, scale_factor, bn_training, update_buffers)
    return torch.tanh(logits), logits, feats


def _unet_conv3(x, p, prefix):
    """UnetConv3 with is_batchnorm=True == InstanceNorm3d, no affine (networks/utils.py:103-109)."""
    for c in ("conv1", "conv2"):
        x = F.conv3d(x, p[f"{prefix}.{c}.0.weight"], p[f"{prefix}.{c}.0.bias"], padding=1)
        x = F.relu(F.instance_norm(x, eps=1e-5))
    return x


def _unet_up(skip, low, p, prefix):
    """UnetUp3_CT (networks/utils.py:260-276): trilinear x2 (align_corners=False) -> cat([skip, up]) -> UnetConv3."""
    up = F.interpolate(low, scale_factor=2, mode="trilinear", align_corners=False)
    return _unet_conv3(torch.cat([skip, up], 1), p, prefix + ".conv")


def unet_forward(x, p: Params, scale_factor: int = 2, drop_center: Optional[torch.Tensor] = None,
                 drop_up1: Optional[torch.Tensor] = None, bn_training: bool = True, update_buffers: bool = False):
    """UNet3D.forward (UNet3D_contrastive.py:276-316).  drop_*: explicit element-wise keep masks (p=0.3)."""
    c1 = _unet_conv3(x, p, "conv1")
    c2 = _unet_conv3(F.max_pool3d(c1, 2), p, "conv2")
    c3 = _unet_conv3(F.max_pool3d(c2, 2), p, "conv3")
    c4 = _unet_conv3(F.max_pool3d(c3, 2), p, "conv4")
    center = dropout_mask(_unet_conv3(F.max_pool3d(c4, 2), p, "center"), drop_center, 0.3)
    u4 = _unet_up(c4, center, p, "up_concat4")
    u3 = _unet_up(c3, u4, p, "up_concat3")
    u2 = _unet_up(c2, u3, p, "up_concat2")
    u1 = dropout_mask(_unet_up(c1, u2, p, "up_concat1"), drop_up1, 0.3)
    feats = projection_head(center, p, scale_factor, bn_training, update_buffers)
    sdf = torch.tanh(F.conv3d(u1, p["final.weight"], p["final.bias"]))
    logits = F.conv3d(u1, p["out_conv2.weight"], p["out_conv2.bias"])
    return sdf, logits, feats


def forward(net_type: str, x, p: Params, **kw):
    if net_type == "vnet":
        return vnet_forward(x, p, **kw)
    if net_type == "unet_3D":
        kw.pop("normalization", None)
        return unet_forward(x, p, **kw)
    raise ValueError(net_type)
