"""CPU restatement of the reference model forward (test infrastructure, see oracle/__init__.py).

Functional: everything is computed from a state_dict (reference key names) — no nn.Module —
so gradients w.r.t. any weight can be taken by passing leaf tensors that require grad.
Dropout2d masks are explicit inputs (per-(n,channel) scale factors 0 or 1/(1-p)); `None`
means eval mode (identity).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

# (name, kind, stride, first_dilation, dilation) — network/resnet38d.py:126-147
_BLOCKS = [
    ("b2", "res", 2, 1, 1), ("b2_1", "res", 1, 1, 1), ("b2_2", "res", 1, 1, 1),
    ("b3", "res", 2, 1, 1), ("b3_1", "res", 1, 1, 1), ("b3_2", "res", 1, 1, 1),
    ("b4", "res", 2, 1, 1), ("b4_1", "res", 1, 1, 1), ("b4_2", "res", 1, 1, 1),
    ("b4_3", "res", 1, 1, 1), ("b4_4", "res", 1, 1, 1), ("b4_5", "res", 1, 1, 1),
    ("b5", "res", 1, 1, 2), ("b5_1", "res", 1, 2, 2), ("b5_2", "res", 1, 2, 2),
    ("b6", "bot", 1, 4, 4), ("b7", "bot", 1, 4, 4),
]


def _bn(x, sd, p):
    # BatchNorm2d in eval mode (frozen by resnet38d.py:207-212)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _relu(x, gates, key):
    """ReLU — or, when `gates` holds a 0/1 tensor for this site, `x * gate`: the ReLU decision of ANOTHER implementation
    injected into this one (tests/test_gpu_loss.py uses it to show that two f32 implementations whose weight gradients differ
    only differ in which near-zero pre-activations they let through)."""
    if gates is None or key not in gates:
        return F.relu(x)
    return x * gates[key]


def _drop(x, masks, key):
    if masks is None:
        return x
    m = masks[key]
    return x * m.view(m.shape[0], m.shape[1], 1, 1)


def res_block(x, sd, name, stride, first_dilation, dilation, gates=None):
    """ResBlock.forward — network/resnet38d.py:27-49. Returns (out, x_bn_relu)."""
    t = _relu(_bn(x, sd, name + ".bn_branch2a"), gates, name + ".t")
    if (name + ".conv_branch1.weight") in sd:
        branch1 = F.conv2d(t, sd[name + ".conv_branch1.weight"], None, stride)
    else:
        branch1 = x
    y = F.conv2d(t, sd[name + ".conv_branch2a.weight"], None, stride, first_dilation, first_dilation)
    y = _relu(_bn(y, sd, name + ".bn_branch2b1"), gates, name + ".v")
    y = F.conv2d(y, sd[name + ".conv_branch2b1.weight"], None, 1, dilation, dilation)
    return branch1 + y, t


def bot_block(x, sd, name, stride, dilation, masks, gates=None):
    """ResBlock_bot.forward — network/resnet38d.py:74-99. Returns (out, x_bn_relu)."""
    t = _relu(_bn(x, sd, name + ".bn_branch2a"), gates, name + ".t")
    branch1 = F.conv2d(t, sd[name + ".conv_branch1.weight"], None, stride)
    y = F.conv2d(t, sd[name + ".conv_branch2a.weight"], None, stride)
    y = _relu(_bn(y, sd, name + ".bn_branch2b1"), gates, name + ".v1")
    y = _drop(y, masks, name + ".dropout_2b1")
    y = F.conv2d(y, sd[name + ".conv_branch2b1.weight"], None, 1, dilation, dilation)
    y = _relu(_bn(y, sd, name + ".bn_branch2b2"), gates, name + ".v2")
    y = _drop(y, masks, name + ".dropout_2b2")
    y = F.conv2d(y, sd[name + ".conv_branch2b2.weight"], None, 1)
    return branch1 + y, t


def backbone(x, sd, masks=None, gates=None):
    """forward_as_dict — network/resnet38d.py:160-189. Returns dict conv4, conv5, conv6."""
    x = F.conv2d(x, sd["conv1a.weight"], None, 1, 1)
    taps = {}
    for name, kind, stride, fd, d in _BLOCKS:
        if kind == "res":
            x, t = res_block(x, sd, name, stride, fd, d, gates)
        else:
            x, t = bot_block(x, sd, name, stride, d, masks, gates)
        if name == "b5":
            taps["conv4"] = t
        if name == "b6":
            taps["conv5"] = t
    taps["conv6"] = _relu(_bn(x, sd, "bn7"), gates, "conv6")
    return taps


def pcm(cam, f, sd):
    """Net.PCM — network/resnet38_contrast.py:63-75."""
    n, c, h, w = f.size()
    cam = F.interpolate(cam, (h, w), mode="bilinear", align_corners=True).view(n, -1, h * w)
    f = F.conv2d(f, sd["f9.weight"])
    f = f.view(n, -1, h * w)
    f = f / (torch.norm(f, dim=1, keepdim=True) + 1e-5)
    aff = F.relu(torch.matmul(f.transpose(1, 2), f))
    aff = aff / (torch.sum(aff, dim=1, keepdim=True) + 1e-5)
    return torch.matmul(cam, aff).view(n, -1, h, w)


def cam_normalize(cam):
    """no_grad CAM normalisation — network/resnet38_contrast.py:41-48."""
    with torch.no_grad():
        n, c, h, w = cam.size()
        cam_d = F.relu(cam.detach())
        cam_d_max = torch.max(cam_d.view(n, c, -1), dim=-1)[0].view(n, c, 1, 1) + 1e-5
        cam_d_norm = F.relu(cam_d - 1e-5) / cam_d_max
        cam_d_norm[:, 0, :, :] = 1 - torch.max(cam_d_norm[:, 1:, :, :], dim=1)[0]
        cam_max = torch.max(cam_d_norm[:, 1:, :, :], dim=1, keepdim=True)[0]
        cam_d_norm[:, 1:, :, :][cam_d_norm[:, 1:, :, :] < cam_max] = 0
    return cam_d_norm


def net_forward(x, sd, masks=None, return_lowres=False, gates=None):
    """Net.forward — network/resnet38_contrast.py:31-61.
    Returns (cam, cam_rv, f_proj, cam_rv_down) [+ (cam_lowres,) when return_lowres]."""
    N, C, H, W = x.size()
    d = backbone(x, sd, masks, gates)
    fea = _drop(d["conv6"], masks, "dropout7")
    f_proj = _relu(F.conv2d(fea, sd["fc_proj.weight"]), gates, "f_proj")
    cam_low = F.conv2d(fea, sd["fc8.weight"])
    n, c, h, w = cam_low.size()
    # (gates["cam_d_norm"]: ANOTHER implementation's gated, normalised CAM — the no_grad input of the PCM, resnet38_contrast.py:41-48 — in place of
    #  this one's: its arg-max gate is discontinuous, so a forward that differs by rounding picks other survivors)
    cam_d_norm = gates["cam_d_norm"] if (gates is not None and "cam_d_norm" in gates) else cam_normalize(cam_low)
    f8_3 = _relu(F.conv2d(d["conv4"].detach(), sd["f8_3.weight"]), gates, "f8_3")
    f8_4 = _relu(F.conv2d(d["conv5"].detach(), sd["f8_4.weight"]), gates, "f8_4")
    x_s = F.interpolate(x, (h, w), mode="bilinear", align_corners=True)
    f = torch.cat([x_s, f8_3, f8_4], dim=1)
    cam_rv_down = pcm(cam_d_norm, f, sd)
    cam_rv = F.interpolate(cam_rv_down, (H, W), mode="bilinear", align_corners=True)
    cam = F.interpolate(cam_low, (H, W), mode="bilinear", align_corners=True)
    if return_lowres:
        return cam, cam_rv, f_proj, cam_rv_down, cam_low
    return cam, cam_rv, f_proj, cam_rv_down


def trainable_keys(sd):
    """Parameters that receive gradients after Net.train() — resnet38d.py:192-214 with
    not_training = [conv1a, b2, b2_1, b2_2] (resnet38_contrast.py:29): all conv weights
    outside the frozen prefix; never BN."""
    keys = []
    for k, v in sd.items():
        if v.dim() != 4:
            continue
        if k.startswith(("conv1a.", "b2.", "b2_1.", "b2_2.")):
            continue
        keys.append(k)
    return keys
