"""ResNet-38d + contrast head layer table.

Single source of truth for parameter names / shapes of the hot path.  Mirrors the
constructor of the reference model (network/resnet38d.py:121-155 and
network/resnet38_contrast.py:12-29) as *data*, so the host code, the weight packer and
the synthetic-weight generator all agree on the 233 state_dict keys.
"""
from collections import OrderedDict

# (name, kind, cin, mid, cout, stride, first_dilation, dilation, dropout)
#   kind 'res' = ResBlock (two 3x3 convs), 'bot' = ResBlock_bot (1x1, 3x3, 1x1)
BLOCKS = [
    ("b2",   "res", 64,   128,  128,  2, 1, 1, 0.0),
    ("b2_1", "res", 128,  128,  128,  1, 1, 1, 0.0),
    ("b2_2", "res", 128,  128,  128,  1, 1, 1, 0.0),
    ("b3",   "res", 128,  256,  256,  2, 1, 1, 0.0),
    ("b3_1", "res", 256,  256,  256,  1, 1, 1, 0.0),
    ("b3_2", "res", 256,  256,  256,  1, 1, 1, 0.0),
    ("b4",   "res", 256,  512,  512,  2, 1, 1, 0.0),
    ("b4_1", "res", 512,  512,  512,  1, 1, 1, 0.0),
    ("b4_2", "res", 512,  512,  512,  1, 1, 1, 0.0),
    ("b4_3", "res", 512,  512,  512,  1, 1, 1, 0.0),
    ("b4_4", "res", 512,  512,  512,  1, 1, 1, 0.0),
    ("b4_5", "res", 512,  512,  512,  1, 1, 1, 0.0),
    ("b5",   "res", 512,  512,  1024, 1, 1, 2, 0.0),
    ("b5_1", "res", 1024, 512,  1024, 1, 2, 2, 0.0),
    ("b5_2", "res", 1024, 512,  1024, 1, 2, 2, 0.0),
    ("b6",   "bot", 1024, 512,  2048, 1, 4, 4, 0.3),
    ("b7",   "bot", 2048, 1024, 4096, 1, 4, 4, 0.5),
]

FROZEN_BLOCKS = ("b2", "b2_1", "b2_2")          # resnet38_contrast.py:29 (+ conv1a)
N_FROZEN_BLOCKS = len(FROZEN_BLOCKS)            # they are the first blocks: the forward pass up to here depends on no trainable weight
assert tuple(b[0] for b in BLOCKS[:N_FROZEN_BLOCKS]) == FROZEN_BLOCKS
HEAD_CONVS = OrderedDict([                       # resnet38_contrast.py:15-20
    ("fc8",     (21,  4096)),
    ("fc_proj", (128, 4096)),
    ("f8_3",    (64,  512)),
    ("f8_4",    (128, 1024)),
    ("f9",      (192, 195)),
])
NUM_CLASSES = 21
PROJ_DIM = 128
BN_EPS = 1e-5


def block_same_shape(b):
    name, kind, cin, mid, cout, stride, fd, d, p = b
    return kind == "res" and cin == cout and stride == 1


def block_convs(b):
    """[(param_prefix, cin, cout, k, stride, dilation)] in module-registration order."""
    name, kind, cin, mid, cout, stride, fd, d, p = b
    out = []
    if kind == "res":
        out.append((f"{name}.conv_branch2a", cin, mid, 3, stride, fd))
        out.append((f"{name}.conv_branch2b1", mid, cout, 3, 1, d))
        if not block_same_shape(b):
            out.append((f"{name}.conv_branch1", cin, cout, 1, stride, 1))
    else:
        out.append((f"{name}.conv_branch2a", cin, cout // 4, 1, stride, 1))
        out.append((f"{name}.conv_branch2b1", cout // 4, cout // 2, 3, 1, d))
        out.append((f"{name}.conv_branch2b2", cout // 2, cout, 1, 1, 1))
        out.append((f"{name}.conv_branch1", cin, cout, 1, stride, 1))
    return out


def block_bns(b):
    """[(param_prefix, channels)] in module-registration order."""
    name, kind, cin, mid, cout, stride, fd, d, p = b
    if kind == "res":
        return [(f"{name}.bn_branch2a", cin), (f"{name}.bn_branch2b1", mid)]
    return [(f"{name}.bn_branch2a", cin), (f"{name}.bn_branch2b1", cout // 4),
            (f"{name}.bn_branch2b2", cout // 2)]


def state_dict_spec():
    """OrderedDict key -> shape, in the reference's state_dict() order (233 keys)."""
    spec = OrderedDict()
    spec["conv1a.weight"] = (64, 3, 3, 3)

    def add_bn(prefix, c):
        spec[prefix + ".weight"] = (c,)
        spec[prefix + ".bias"] = (c,)
        spec[prefix + ".running_mean"] = (c,)
        spec[prefix + ".running_var"] = (c,)
        spec[prefix + ".num_batches_tracked"] = ()

    for b in BLOCKS:
        name, kind = b[0], b[1]
        convs = dict((p, (co, ci, k, k)) for (p, ci, co, k, s, d) in block_convs(b))
        bns = dict(block_bns(b))
        # registration order inside the reference blocks (resnet38d.py:15-25, 60-72)
        if kind == "res":
            order = ["bn_branch2a", "conv_branch2a", "bn_branch2b1", "conv_branch2b1", "conv_branch1"]
        else:
            order = ["bn_branch2a", "conv_branch2a", "bn_branch2b1", "conv_branch2b1",
                     "bn_branch2b2", "conv_branch2b2", "conv_branch1"]
        for o in order:
            p = f"{name}.{o}"
            if p in bns:
                add_bn(p, bns[p])
            elif p in convs:
                spec[p + ".weight"] = convs[p]
    add_bn("bn7", 4096)
    for hname, (co, ci) in HEAD_CONVS.items():
        spec[hname + ".weight"] = (co, ci, 1, 1)
    return spec


def _osz(h, k, s, d):
    p = d * (k // 2)
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def forward_macs(H, W):
    """Algorithmic multiply-accumulates of ONE Net.forward on an H x W image (conv MACs = Cin * Cout * k^2 * h_out * w_out
    for every conv of resnet38d.py:160-189 and resnet38_contrast.py:34-54, PCM = hw^2 * (192 + 21), resnet38_contrast.py:70-73).
    448 x 448 -> 403.697e9, 128 x 128 -> 32.798e9."""
    macs = 3 * 64 * 9 * H * W                                   # conv1a, stride 1, same size
    h, w = H, W
    for b in BLOCKS:
        name, kind, cin, mid, cout, stride, fd, d, p = b
        k0, d0 = (3, fd) if kind == "res" else (1, 1)
        oh, ow = _osz(h, k0, stride, d0), _osz(w, k0, stride, d0)
        for (_n, ci, co, k, s, dd) in block_convs(b):
            macs += ci * co * k * k * oh * ow                   # every conv of a block produces the block's output size
        h, w = oh, ow
    hw = h * w
    macs += hw * sum(co * ci for (co, ci) in HEAD_CONVS.values())
    macs += hw * hw * (192 + 21)
    return macs
