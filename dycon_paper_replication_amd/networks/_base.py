"""nn.Module shell around the HIP executor (engine.Engine).

The module owns parameters/buffers under exactly the reference's state_dict keys and shapes
(code/networks/VNet.py:150-175, code/networks/UNet3D_contrastive.py:222-267), so checkpoints are
interchangeable with the reference's ``torch.save(model.state_dict())`` / ``load_state_dict``
(code/train_DyCON_BraTS19.py:411-418, code/test_BraTS19.py:62-63).  ``forward`` keeps the reference
contract ``(tanh_map, logits, features)`` with logical NCDHW shapes; physically the tensors are
channels-last-3D (the layout every HIP kernel works in).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import ops
from ..engine import DropoutSpec, Engine, net_buffers, param_spec


def _register(root: nn.Module, dotted: str, tensor, buffer=False):
    *path, leaf = dotted.split(".")
    m = root
    for part in path:
        if part not in m._modules:
            m.add_module(part, nn.Module())
        m = m._modules[part]
    if buffer:
        m.register_buffer(leaf, tensor)
    else:
        m.register_parameter(leaf, nn.Parameter(tensor))


def to_ndhwc(x: torch.Tensor) -> torch.Tensor:
    """(B,C,D,H,W) -> contiguous (B,D,H,W,C); free for C == 1 or channels_last_3d inputs."""
    if x.shape[1] == 1 and x.is_contiguous():
        return x.reshape(x.shape[0], *x.shape[2:], 1)
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncdhw_view(x: torch.Tensor) -> torch.Tensor:
    """(B,D,H,W,C) -> logical (B,C,D,H,W) view (channels_last_3d strides, no copy)."""
    return x.permute(0, 4, 1, 2, 3)


class _NetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are explicit HIP launch sequences."""

    @staticmethod
    def forward(ctx, x, net, *params):
        eng = net._engine()
        eng.repack()                   # all stale packs in one launch (no-op when nothing changed)
        record = any(ctx.needs_input_grad[2:])
        logits, feats, sdf = eng.forward(to_ndhwc(x), training=net.training, record=record, dropout=net._dropout_spec(),
                                         update_bn=net.training, want_sdf=True)
        ctx.net, ctx.eng = net, eng
        ctx.mark_non_differentiable(sdf)
        return to_ncdhw_view(sdf), to_ncdhw_view(logits), to_ncdhw_view(feats)

    @staticmethod
    def backward(ctx, _g_sdf, g_logits, g_feats):
        net, eng = ctx.net, ctx.eng
        names = net._param_names
        grads = {k: torch.empty_like(p) for k, p in zip(names, net.parameters())}
        touched = {k: False for k in names}
        eng.g = _Tracking(grads, touched)
        gl = to_ndhwc(g_logits).float() if g_logits is not None else None
        gf = to_ndhwc(g_feats).to(eng.dtype) if g_feats is not None else None
        eng.backward(gl, gf)
        return (None, None) + tuple(grads[k] if touched[k] else None for k in names)


class _Tracking(dict):
    """grad dict that remembers which entries the backward wrote (UNet3D ``final.*`` never is)."""

    def __init__(self, grads, touched):
        super().__init__(grads)
        self._touched = touched

    def __getitem__(self, k):
        self._touched[k] = True
        return super().__getitem__(k)


class HipSegNet(nn.Module):
    net_type = None

    def __init__(self, in_channels=1, n_classes=2, scale_factor=2, normalization="groupnorm", has_dropout=True,
                 dtype=torch.float32, seed=None):
        super().__init__()
        if n_classes != 2:
            raise NotImplementedError("the DyCON step hard-codes 2 classes (train_DyCON_BraTS19.py:146)")
        self.in_channels, self.n_classes, self.scale_factor = in_channels, n_classes, scale_factor
        self.normalization, self.has_dropout, self.compute_dtype = normalization, has_dropout, dtype
        spec = param_spec(self.net_type, in_channels, n_classes, normalization)
        self._param_names = list(spec)
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        for name, shape in spec.items():
            _register(self, name, self._init_tensor(name, shape, gen))
        for name, shape in net_buffers(self.net_type, normalization, spec).items():
            if name.endswith("num_batches_tracked"):
                t = torch.zeros((), dtype=torch.long)
            elif name.endswith("running_var"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            _register(self, name, t, buffer=True)
        self._eng = None
        self._eng_key = None
        self.weights_static = False    # True: parameters only change through version-counted in-place ops (see _engine)
        self._drop_calls = 0
        self._drop_seed = int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)

    # kaiming-normal fan_in conv weights, BN gamma ~ N(1, 0.02), beta = 0 (networks_other.py:40-49);
    # conv biases / GroupNorm affine keep torch's defaults (uniform(+-1/sqrt(fan_in)), ones / zeros)
    @staticmethod
    def _init_tensor(name, shape, gen):
        if len(shape) == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_in)
        if name.startswith("projection.1.") or name.startswith("projection.4."):
            return 1.0 + 0.02 * torch.randn(shape, generator=gen) if name.endswith("weight") else torch.zeros(shape)
        if name.endswith(".weight"):
            return torch.ones(shape)
        return torch.zeros(shape)

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self) -> Engine:
        ps = list(self.parameters())
        key = (tuple(p.data_ptr() for p in ps), self.compute_dtype, self.scale_factor)
        if self._eng is None or self._eng_key != key:
            params = {k: p.data for k, p in zip(self._param_names, ps)}
            bufs = dict(self.named_buffers())
            self._eng = Engine(self.net_type, params, None, bufs, self.compute_dtype, self.scale_factor, self.normalization)
            self._eng_key = key
            self._ver = None
        # Packed operands must follow the parameters.  optimizer.step() / load_state_dict bump the tensors' version counters, but
        # the reference's EMA writes through `.data` (train_DyCON_BraTS19.py:163-164), which does NOT: so unless the caller
        # declared the weights frozen (`weights_static`, the sliding-window evaluator), every forward refreshes all packs -- one
        # batched launch (Engine.repack) -- exactly as the fused trainer does each step.
        ver = sum(p._version for p in ps)
        if ver != self._ver or not self.weights_static:
            self._eng.params_changed()
            self._ver = ver
        return self._eng

    def params_changed(self):
        """Call after parameters were modified behind torch's back (the fused HIP optimiser)."""
        if self._eng is not None:
            self._eng.params_changed()

    def _dropout_spec(self):
        if not (self.training and self.has_dropout):
            return DropoutSpec("off")
        self._drop_calls += 1
        return DropoutSpec("philox", seed=self._drop_seed, offset=self._drop_calls * (1 << 24))

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("HipSegNet runs on the MI355X only (no CPU fallback): move the module and input to 'cuda'")
        if x.dim() != 5 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (B,{self.in_channels},D,H,W), got {tuple(x.shape)}")
        if any(s % 16 for s in x.shape[2:]):
            raise ValueError("D, H, W must be divisible by 16")
        return _NetFunction.apply(x.float().contiguous(), self, *self.parameters())
