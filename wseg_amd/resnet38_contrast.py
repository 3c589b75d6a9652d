"""Drop-in replacement of the reference model module: `--network wseg_amd.resnet38_contrast`.

Mirrors the public contract of network/resnet38_contrast.py:12-96 (+ network/resnet38d.py:104-214):
`Net()`, `forward(x) -> (cam, cam_rv, f_proj, cam_rv_down)`, `.normalize`, `.get_parameter_groups()`,
`.train()/.eval()`, and the exact 233 state_dict keys — but the sub-modules below are *parameter
containers only*: all compute runs in the HIP kernels of libwseg_hip.so through wseg_amd.engine.
There is no eager/CPU fallback: `forward` on a non-GPU tensor raises.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import arch
from . import engine


class _Block(nn.Module):
    """Parameter container with the attribute names of ResBlock / ResBlock_bot
    (network/resnet38d.py:7-25, 55-72).  Never called."""

    def __init__(self, spec):
        super().__init__()
        name, kind, cin, mid, cout, stride, fd, d, p = spec
        self.kind = kind
        self.same_shape = arch.block_same_shape(spec)
        bns = dict((k.split(".")[1], c) for k, c in arch.block_bns(spec))
        convs = dict((k.split(".")[1], (ci, co, kk, s, dd)) for (k, ci, co, kk, s, dd) in arch.block_convs(spec))
        order = (["bn_branch2a", "conv_branch2a", "bn_branch2b1", "conv_branch2b1", "conv_branch1"] if kind == "res" else
                 ["bn_branch2a", "conv_branch2a", "bn_branch2b1", "dropout_2b1", "conv_branch2b1",
                  "bn_branch2b2", "dropout_2b2", "conv_branch2b2", "conv_branch1"])
        for o in order:
            if o in bns:
                setattr(self, o, nn.BatchNorm2d(bns[o]))
            elif o in convs:
                ci, co, kk, s, dd = convs[o]
                setattr(self, o, nn.Conv2d(ci, co, kk, s, padding=dd * (kk // 2), dilation=dd, bias=False))
            elif o.startswith("dropout"):
                setattr(self, o, nn.Dropout2d(p))

    def forward(self, *a, **k):
        raise RuntimeError("wseg_amd blocks are parameter containers; call Net.forward")


class Normalize:
    """network/resnet38d.py:104-118 — host-side image normalisation (HWC uint8 -> HWC float32)."""

    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.mean = mean
        self.std = std

    def __call__(self, img):
        imgarr = np.asarray(img)
        proc_img = np.empty_like(imgarr, np.float32)
        for c in range(3):
            proc_img[..., c] = (imgarr[..., c] / 255. - self.mean[c]) / self.std[c]
        return proc_img


class Net(nn.Module):
    def __init__(self, precision=None):
        super().__init__()
        self.conv1a = nn.Conv2d(3, 64, 3, padding=1, bias=False)
        for spec in arch.BLOCKS:
            setattr(self, spec[0], _Block(spec))
        self.bn7 = nn.BatchNorm2d(4096)
        self.dropout7 = nn.Dropout2d(0.5)
        self.fc8 = nn.Conv2d(4096, 21, 1, bias=False)
        self.fc_proj = nn.Conv2d(4096, 128, 1, bias=False)
        self.f8_3 = nn.Conv2d(512, 64, 1, bias=False)
        self.f8_4 = nn.Conv2d(1024, 128, 1, bias=False)
        self.f9 = nn.Conv2d(192 + 3, 192, 1, bias=False)
        # inits of network/resnet38_contrast.py:22-26
        nn.init.xavier_uniform_(self.fc8.weight)
        nn.init.kaiming_normal_(self.f8_3.weight)
        nn.init.kaiming_normal_(self.f8_4.weight)
        nn.init.xavier_uniform_(self.f9.weight, gain=4)
        nn.init.xavier_uniform_(self.fc_proj.weight)
        self.from_scratch_layers = [self.f8_3, self.f8_4, self.f9, self.fc8, self.fc_proj]
        self.not_training = [self.conv1a, self.b2, self.b2_1, self.b2_2]
        self.normalize = Normalize()
        self.precision = precision or os.environ.get("WSEG_PRECISION", "bf16")
        assert self.precision in ("bf16", "fp32", "bf16x3")      # bf16x3: f32 storage, conv / wgrad products as split-bf16 (wseg_hip.h WSEG_F32X3)

    @property
    def _engine(self):
        """The engine of THIS module instance.  A shallow copy of the module (`nn.parallel.replicate`, contrast_infer.py:47)
        arrives with the original's engine in its __dict__: it gets one of its own, whose parent is the original's."""
        eng = self.__dict__.get("_wseg_engine")
        if eng is None or eng._net_ref() is not self:
            eng = engine.Engine(self, parent=eng)
            self.__dict__["_wseg_engine"] = eng
        return eng

    # ---- reference API -------------------------------------------------------------------
    def forward(self, x):
        """network/resnet38_contrast.py:31-61."""
        return self._engine.forward(x)

    def forward_lowres(self, x):
        """Fused-path entry used by wseg_amd.train_step: returns the stride-8 maps
        (cam_low [N,21,h,w], cam_rv_down [N,21,h,w], f_proj rows) without the two x8 upsamples."""
        return self._engine.forward(x, lowres=True)

    def set_dropout_masks(self, mask_sets):
        """Inject Dropout2d scale factors (list of dicts as wseg_amd.synth.synthetic_dropout_masks,
        consumed one per forward call) — used by the parity tests."""
        self._engine.injected_masks = list(mask_sets) if mask_sets is not None else None

    def get_parameter_groups(self):
        """network/resnet38_contrast.py:77-96."""
        groups = ([], [], [], [])
        print('======================================================')
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.modules.normalization.GroupNorm)):
                if m.weight.requires_grad:
                    (groups[2] if m in self.from_scratch_layers else groups[0]).append(m.weight)
                if m.bias is not None and m.bias.requires_grad:
                    (groups[3] if m in self.from_scratch_layers else groups[1]).append(m.bias)
        return groups

    def train(self, mode=True):
        """network/resnet38d.py:192-214: frozen prefix + every BatchNorm in eval and frozen."""
        super().train(mode)
        for layer in self.not_training:
            if isinstance(layer, nn.Conv2d):
                layer.weight.requires_grad = False
            elif isinstance(layer, nn.Module):
                for c in layer.children():
                    if getattr(c, "weight", None) is not None:
                        c.weight.requires_grad = False
                    if getattr(c, "bias", None) is not None:
                        c.bias.requires_grad = False
        for layer in self.modules():
            if isinstance(layer, nn.BatchNorm2d):
                layer.eval()
                layer.bias.requires_grad = False
                layer.weight.requires_grad = False
        return self
