"""Forward/backward orchestration of the ResNet-38d + CAM/PCM head on the HIP kernels.

Host-side plumbing only: allocates torch tensors (device memory), folds the frozen BatchNorms,
keeps the trainable weights in ONE flat f32 buffer (and their gradients in another — a single
RCCL all-reduce / fused SGD target), and issues the C-ABI kernels of libwseg_hip.so in the order
of network/resnet38d.py:160-189 and network/resnet38_contrast.py:31-75.  Activations are NHWC
("pixel rows") in the precision mode's dtype; 21-class maps are planar f32.

The whole network is one torch.autograd.Function: its backward runs the hand-written dgrad /
wgrad / PCM-backward kernels and accumulates straight into the flat gradient buffer.
"""
import torch

from . import arch
from . import _lib as L

HEAD_LD = 192          # fused head rows: [f_proj 128 | cam 21 | zero pad 43]
FEAT_LD = 256          # PCM feature rows: [f8_3 64 | f8_4 128 | x_s 3 | zero pad 61]


def _out_size(h, k, s, d):
    p = d * (k // 2)
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


class Engine:
    def __init__(self, net):
        self.__dict__["net"] = net          # plain attribute: not a sub-module
        self.flat_w = None
        self.flat_g = None
        self.packs = None
        self.pack_key = None
        self.injected_masks = None
        self._order = None
        self.flat_w_version = 0             # bumped by the fused SGD step (raw kernel writes)

    # ------------------------------------------------------------------ parameters
    def conv_param(self, name):
        mod = self.net
        for part in name.split("."):
            mod = getattr(mod, part)
        return mod.weight

    def bn_module(self, name):
        mod = self.net
        for part in name.split("."):
            mod = getattr(mod, part)
        return mod

    def trainable_order(self):
        """Flat-buffer order: backbone convs b3..b7, then fc_proj, fc8 (adjacent: the fused head
        GEMM's weight gradient is one [149,4096] block), f8_3, f8_4, f9."""
        if self._order is None:
            names = []
            for b in arch.BLOCKS:
                if b[0] in arch.FROZEN_BLOCKS:
                    continue
                names += [c[0] for c in arch.block_convs(b)]
            names += ["fc_proj", "fc8", "f8_3", "f8_4", "f9"]
            self._order = names
        return self._order

    def ensure_flat(self, device):
        """(Re)build the flat weight / gradient buffers when the parameters moved."""
        names = self.trainable_order()
        first = self.conv_param(names[0])
        if (self.flat_w is not None and self.flat_w.device == first.device
                and first.data_ptr() == self.flat_w.data_ptr() and first.device == device):
            return
        total = sum(self.conv_param(n).numel() for n in names)
        flat_w = torch.empty(total, device=device, dtype=torch.float32)
        flat_g = torch.zeros(total, device=device, dtype=torch.float32)
        off = 0
        self.offsets = {}
        for n in names:
            p = self.conv_param(n)
            oc, ic, kh, kw = p.shape
            view = flat_w[off:off + p.numel()].view(oc, kh, kw, ic).permute(0, 3, 1, 2)
            view.copy_(p.data.to(device))
            p.data = view                                        # logical [OC,IC,KH,KW], physical [OC][KH][KW][IC]
            p._wseg_flat = (self, off, p.numel())                # lets PolyOptimizer find the flat buffers
            self.offsets[n] = (off, p.numel())
            off += p.numel()
        for n in ["conv1a"] + [c[0] for b in arch.BLOCKS if b[0] in arch.FROZEN_BLOCKS for c in arch.block_convs(b)]:
            p = self.conv_param(n)
            p.data = p.data.to(device).contiguous(memory_format=torch.channels_last)
        self.flat_w, self.flat_g = flat_w, flat_g
        self.packs = None

    def grad_view(self, name):
        off, n = self.offsets[name]
        p = self.conv_param(name)
        oc, ic, kh, kw = p.shape
        return self.flat_g[off:off + n].view(oc, kh, kw, ic).permute(0, 3, 1, 2)

    def attach_grads(self):
        """Make every trainable p.grad a view of flat_g (zeroing it if the caller dropped the grads,
        e.g. optimizer.zero_grad(set_to_none=True))."""
        names = self.trainable_order()
        p0 = self.conv_param(names[0])
        if p0.grad is not None and p0.grad.data_ptr() == self.flat_g.data_ptr():
            return
        self.flat_g.zero_()
        for n in names:
            p = self.conv_param(n)
            if p.requires_grad:
                p.grad = self.grad_view(n)

    # ------------------------------------------------------------------ packs
    def _bn_fold(self, name, device):
        bn = self.bn_module(name)
        scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + arch.BN_EPS)).to(device)
        shift = (bn.bias.detach().float() - bn.running_mean.float() * scale).to(device)
        return scale.contiguous(), shift.contiguous()

    def ensure_packs(self, device, dt):
        """Packed (cast / transposed) weights + folded BatchNorms.  Frozen pieces (every BN, conv1a, b2*) are
        cached on their own key so a training step only re-packs the 40 trainable tensors."""
        net = self.net
        tdt = L.TORCH_DTYPE[dt]
        frozen_names = [c[0] for b in arch.BLOCKS if b[0] in arch.FROZEN_BLOCKS for c in arch.block_convs(b)]
        fkey = (dt, str(device)) + tuple(b._version for b in net.buffers()) + \
            tuple(p._version for n_, p in net.named_parameters() if ".bn" in n_ or n_.startswith("bn7")) + \
            tuple(self.conv_param(n_)._version for n_ in frozen_names)
        if getattr(self, "_frozen_packs", None) is None or self._frozen_key != fkey:
            F_ = {"w": {}, "bn": {}}
            for b in arch.BLOCKS:
                for (bname, c) in arch.block_bns(b):
                    F_["bn"][bname] = self._bn_fold(bname, device)
                if b[0] in arch.FROZEN_BLOCKS:
                    for (cname, ci, co, k, s, d) in arch.block_convs(b):
                        wf = torch.empty(co, k * k, ci, device=device, dtype=tdt)
                        L.pack_weights(self.conv_param(cname).detach(), wf, None, co, k * k, ci, co, ci, dt)
                        F_["w"][cname] = wf
            F_["bn"]["bn7"] = self._bn_fold("bn7", device)
            self._frozen_packs, self._frozen_key = F_, fkey
            self.packs = None
        names = self.trainable_order()
        key = (dt, str(device), self.flat_w_version) + tuple(self.conv_param(n_)._version for n_ in names)
        if self.packs is not None and key == self.pack_key:
            return self.packs
        P = {"w": dict(self._frozen_packs["w"]), "wt": {}, "bn": self._frozen_packs["bn"]}
        no_dgrad = {"b3.conv_branch1", "b3.conv_branch2a"}
        for b in arch.BLOCKS:
            if b[0] in arch.FROZEN_BLOCKS:
                continue
            for (cname, ci, co, k, s, d) in arch.block_convs(b):
                w = self.conv_param(cname).detach()
                T = k * k
                wf = torch.empty(co, T, ci, device=device, dtype=tdt)
                wt = torch.empty(ci, T, co, device=device, dtype=tdt) if cname not in no_dgrad else None
                L.pack_weights(w, wf, wt, co, T, ci, co, ci, dt)
                P["w"][cname], P["wt"][cname] = wf, wt
        # fused head: rows [fc_proj | fc8 | 0]; its transposed pack is made from the two f32 masters directly
        wh = torch.zeros(HEAD_LD, 1, 4096, device=device, dtype=tdt)
        wht = torch.zeros(4096, 1, HEAD_LD, device=device, dtype=tdt)
        L.pack_weights(net.fc_proj.weight.detach(), wh, None, 128, 1, 4096, 128, 4096, dt)
        L.pack_weights(net.fc8.weight.detach(), wh[128:], None, 21, 1, 4096, 21, 4096, dt)
        off, _ = self.offsets["fc_proj"]                     # fc_proj and fc8 are adjacent in flat_w: one [149,4096] master
        L.pack_weights(self.flat_w[off:off + 149 * 4096], None, wht, 149, 1, 4096, HEAD_LD, 4096, dt)
        P["w"]["head"], P["wt"]["head"] = wh, wht
        for nm, (co, ci) in (("f8_3", (64, 512)), ("f8_4", (128, 1024))):
            wf = torch.empty(co, 1, ci, device=device, dtype=tdt)
            L.pack_weights(getattr(net, nm).weight.detach(), wf, None, co, 1, ci, co, ci, dt)
            P["w"][nm] = wf
        # f9: input columns re-ordered to the internal feature layout [f8_3 | f8_4 | x_s | pad]
        w9 = net.f9.weight.detach().reshape(192, 195)
        w9p = torch.cat([w9[:, 3:67], w9[:, 67:195], w9[:, 0:3]], dim=1).contiguous()
        wf = torch.empty(192, 1, FEAT_LD, device=device, dtype=tdt)
        wt = torch.empty(FEAT_LD, 1, 192, device=device, dtype=tdt)
        L.pack_weights(w9p, wf, wt, 192, 1, 195, 192, FEAT_LD, dt)
        P["w"]["f9"], P["wt"]["f9"] = wf, wt
        self.packs, self.pack_key = P, key
        return P

    # ------------------------------------------------------------------ dropout
    def _masks(self, n, device):
        net = self.net
        if not net.training:
            return None
        if self.injected_masks:
            m = self.injected_masks.pop(0)
            return {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in m.items()}
        out = {}
        for key, (c, p) in (("b6.dropout_2b1", (512, 0.3)), ("b6.dropout_2b2", (1024, 0.3)),
                            ("b7.dropout_2b1", (1024, 0.5)), ("b7.dropout_2b2", (2048, 0.5)),
                            ("dropout7", (4096, 0.5))):
            out[key] = (torch.rand(n, c, device=device) >= p).float().div_(1.0 - p)
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, x, lowres=False):
        if not x.is_cuda:
            raise RuntimeError("wseg_amd.Net runs only on an MI355X (HIP) device; there is no CPU fallback")
        x = x.contiguous().float()
        self.ensure_flat(x.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        if need_grad:
            anchor = self.flat_w.new_zeros((), requires_grad=True)
            return _NetFunction.apply(x, anchor, self, lowres)
        outs, _ = self.run_forward(x, save=False, lowres=lowres)
        return outs

    def run_forward(self, x, save, lowres=False):
        net = self.net
        dev = x.device
        dt = L.BF16 if net.precision == "bf16" else L.F32
        tdt = L.TORCH_DTYPE[dt]
        P = self.ensure_packs(dev, dt)
        N, _, H, W = x.shape
        masks = self._masks(N, dev)
        S = {"masks": masks, "dims": {}, "N": N, "H": H, "W": W, "dt": dt, "x": x}

        def E(*shape):
            return torch.empty(shape, device=dev, dtype=tdt)

        def next_bn(i):
            if i + 1 < len(arch.BLOCKS):
                return P["bn"][arch.BLOCKS[i + 1][0] + ".bn_branch2a"], None
            return P["bn"]["bn7"], (masks["dropout7"] if masks else None)

        sc, sh = P["bn"]["b2.bn_branch2a"]
        t = E(N, H, W, 64)
        L.stem_conv(x, net.conv1a.weight.detach(), sc, sh, None, t, N, H, W, dt)
        xraw, h, w = None, H, W
        for i, b in enumerate(arch.BLOCKS):
            name, kind, cin, mid, cout, stride, fd, d, p = b
            same = arch.block_same_shape(b)
            (nsc, nsh), ndrop = next_bn(i)
            nxt_same = i + 1 < len(arch.BLOCKS) and arch.block_same_shape(arch.BLOCKS[i + 1])
            oh, ow = _out_size(h, 3 if kind == "res" else 1, stride, fd if kind == "res" else 1), \
                _out_size(w, 3 if kind == "res" else 1, stride, fd if kind == "res" else 1)
            geo = dict(N=N)
            if kind == "res":
                s1, sh1 = P["bn"][name + ".bn_branch2b1"]
                v = E(N, oh, ow, mid)
                L.conv_igemm(t, P["w"][name + ".conv_branch2a"], None, v, IH=h, IW=w, IC=cin, OH=oh, OW=ow, OC=mid,
                             KH=3, KW=3, stride=stride, dil=fd, pad=fd, scale=s1, shift=sh1, **geo)
                if same:
                    rpost = xraw
                else:
                    rpost = E(N, oh, ow, cout)
                    L.conv_igemm(t, P["w"][name + ".conv_branch1"], rpost, IH=h, IW=w, IC=cin, OH=oh, OW=ow, OC=cout,
                                 KH=1, KW=1, stride=stride, **geo)
                xn = E(N, oh, ow, cout) if nxt_same else None
                tn = E(N, oh, ow, cout)
                L.conv_igemm(v, P["w"][name + ".conv_branch2b1"], xn, tn, IH=oh, IW=ow, IC=mid, OH=oh, OW=ow, OC=cout,
                             KH=3, KW=3, dil=d, pad=d, r_post=rpost, scale=nsc, shift=nsh, drop=ndrop, **geo)
                if save:
                    S[name] = dict(t=t, v=v)
            else:
                c4, c2 = cout // 4, cout // 2
                s1, sh1 = P["bn"][name + ".bn_branch2b1"]
                s2, sh2 = P["bn"][name + ".bn_branch2b2"]
                d1 = masks[name + ".dropout_2b1"] if masks else None
                d2 = masks[name + ".dropout_2b2"] if masks else None
                v1 = E(N, oh, ow, c4)
                L.conv_igemm(t, P["w"][name + ".conv_branch2a"], None, v1, IH=h, IW=w, IC=cin, OH=oh, OW=ow, OC=c4,
                             KH=1, KW=1, stride=stride, scale=s1, shift=sh1, drop=d1, **geo)
                v2 = E(N, oh, ow, c2)
                L.conv_igemm(v1, P["w"][name + ".conv_branch2b1"], None, v2, IH=oh, IW=ow, IC=c4, OH=oh, OW=ow, OC=c2,
                             KH=3, KW=3, dil=d, pad=d, scale=s2, shift=sh2, drop=d2, **geo)
                b1 = E(N, oh, ow, cout)
                L.conv_igemm(t, P["w"][name + ".conv_branch1"], b1, IH=h, IW=w, IC=cin, OH=oh, OW=ow, OC=cout,
                             KH=1, KW=1, stride=stride, **geo)
                xn = None
                tn = E(N, oh, ow, cout)
                L.conv_igemm(v2, P["w"][name + ".conv_branch2b2"], xn, tn, IH=oh, IW=ow, IC=c2, OH=oh, OW=ow, OC=cout,
                             KH=1, KW=1, r_post=b1, scale=nsc, shift=nsh, drop=ndrop, **geo)
                if save:
                    S[name] = dict(t=t, v1=v1, v2=v2)
            S["dims"][name] = (h, w, oh, ow)
            if name == "b5":
                conv4 = t
            if name == "b6":
                conv5 = t
            t, xraw, h, w = tn, xn, oh, ow

        fea = t                                               # relu(bn7(x)) * dropout7   [N,h,w,4096]
        hw = h * w
        head = E(N, h, w, HEAD_LD)
        L.conv_igemm(fea, P["w"]["head"], head, N=N, IH=h, IW=w, IC=4096, OH=h, OW=w, OC=HEAD_LD, KH=1, KW=1, relu_lt=128)
        cam_low = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
        cmax = torch.empty(N, 21, device=dev, dtype=torch.float32)
        L.head_split(head, HEAD_LD, 128, cam_low, cmax, N, hw)
        G = torch.empty(N * hw, 32, device=dev, dtype=torch.float32)
        L.cam_gate(cam_low, cmax, G, N, hw)
        feat = E(N, h, w, FEAT_LD)
        L.conv_igemm(conv4, P["w"]["f8_3"], feat, N=N, IH=h, IW=w, IC=512, OH=h, OW=w, OC=64, KH=1, KW=1, epi=2, ld_out=FEAT_LD)
        L.conv_igemm(conv5, P["w"]["f8_4"], feat.view(-1)[64:], N=N, IH=h, IW=w, IC=1024, OH=h, OW=w, OC=128, KH=1, KW=1,
                     epi=2, ld_out=FEAT_LD)
        L.pcm_xs(x, feat, FEAT_LD, 192, FEAT_LD, N, H, W, h, w)
        Fm = E(N, h, w, 192)
        L.conv_igemm(feat, P["w"]["f9"], Fm, N=N, IH=h, IW=w, IC=FEAT_LD, OH=h, OW=w, OC=192, KH=1, KW=1)
        Fh = torch.empty(N * hw, 192, device=dev, dtype=torch.float32)
        nrm = torch.empty(N * hw, device=dev, dtype=torch.float32)
        L.l2norm_forward(Fm, 192, Fh, nrm, N * hw)
        rvd = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
        den = torch.empty(N, hw, device=dev, dtype=torch.float32)
        Fb = Gb = None
        if dt == L.BF16:                                      # bf16-MFMA PCM (throughput mode); fp32 mode keeps the exact-f32 kernel
            Fb = torch.empty(N * hw, 192, device=dev, dtype=torch.bfloat16)
            Gb = torch.empty(N * hw, 32, device=dev, dtype=torch.bfloat16)
            L.to_bf16(Fh, Fb)
            L.to_bf16(G, Gb)
            L.pcm_forward_bf16(Fb, Gb, rvd, den, N, hw)
        else:
            L.pcm_forward(Fh, G, rvd, den, N, hw)
        f_proj = head[..., :128].permute(0, 3, 1, 2)
        if lowres:
            outs = (cam_low, rvd, f_proj, head)
        else:
            cam = torch.empty(N, 21, H, W, device=dev, dtype=torch.float32)
            cam_rv = torch.empty(N, 21, H, W, device=dev, dtype=torch.float32)
            L.resize_planar_fwd(cam_low, cam, N * 21, h, w, H, W, True)
            L.resize_planar_fwd(rvd, cam_rv, N * 21, h, w, H, W, True)
            outs = (cam, cam_rv, f_proj.float() if dt == L.BF16 else f_proj, rvd)
        if save:
            S.update(fea=fea, head=head, G=G, feat=feat, Fm=Fm, Fh=Fh, Fb=Fb, Gb=Gb, nrm=nrm, rvd=rvd.clone(), den=den,
                     conv4=conv4, conv5=conv5, h=h, w=w, lowres=lowres)
        return outs, S

    # ------------------------------------------------------------------ backward
    def run_backward(self, S, g_cam, g_cam_rv, g_fproj, g_rvd, d_head_rows=None):
        """Gradients of the 4 outputs -> accumulates into flat_g.  In the fused (lowres) path
        g_cam / g_cam_rv are already gradients of the stride-8 maps and d_head_rows may be supplied
        directly (rows [f_proj | cam | pad])."""
        net = self.net
        P = self.packs
        dt = S["dt"]
        tdt = L.TORCH_DTYPE[dt]
        N, H, W, h, w = S["N"], S["H"], S["W"], S["h"], S["w"]
        hw = h * w
        dev = S["fea"].device
        masks = S["masks"]
        self.attach_grads()

        def E(*shape):
            return torch.empty(shape, device=dev, dtype=tdt)

        def trainable(nm):
            return self.conv_param(nm).requires_grad

        def wgrad(nm, x, dy, **kw):
            if trainable(nm):
                off, n = self.offsets[nm]
                L.conv_wgrad(x, dy, self.flat_g[off:off + n], **kw)

        # ---- upsample adjoints
        if S["lowres"]:
            d_cam_low, d_rvd = g_cam, g_cam_rv
        else:
            d_cam_low = None
            if g_cam is not None:
                d_cam_low = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
                L.resize_planar_bwd(g_cam.contiguous().float(), d_cam_low, N * 21, h, w, H, W, True)
            d_rvd = None
            if g_cam_rv is not None:
                d_rvd = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
                L.resize_planar_bwd(g_cam_rv.contiguous().float(), d_rvd, N * 21, h, w, H, W, True)
            if g_rvd is not None:
                d_rvd = g_rvd.contiguous().float() if d_rvd is None else d_rvd + g_rvd
        # ---- PCM branch -> f9, f8_3, f8_4
        if d_rvd is not None:
            DN = torch.empty(N * hw, 32, device=dev, dtype=torch.float32)
            dFh = torch.zeros(N * hw, 192, device=dev, dtype=torch.float32)
            if S["Fb"] is not None:
                DNb = torch.empty(N * hw, 32, device=dev, dtype=torch.bfloat16)
                L.pcm_backward_bf16(S["Fb"], S["Gb"], d_rvd.contiguous(), S["rvd"], S["den"], DN, DNb, dFh, N, hw)
            else:
                L.pcm_backward(S["Fh"], S["G"], d_rvd.contiguous(), S["rvd"], S["den"], DN, dFh, N, hw)
            dF = E(N, h, w, 192)
            L.l2norm_backward(S["Fm"], 192, dFh, S["nrm"], dF, 192, N * hw)
            if trainable("f9"):
                g9 = torch.zeros(192, 195, device=dev, dtype=torch.float32)
                L.conv_wgrad(S["feat"], dF, g9, N=N, IH=h, IW=w, IC=FEAT_LD, OH=h, OW=w, OC=192, KH=1, KW=1, IC_dw=195)
                gv = self.grad_view("f9").reshape(192, 195)
                gv[:, 3:67] += g9[:, 0:64]
                gv[:, 67:195] += g9[:, 64:192]
                gv[:, 0:3] += g9[:, 192:195]
            if trainable("f8_3") or trainable("f8_4"):
                d_feat = E(N, h, w, FEAT_LD)
                L.conv_igemm(dF, P["wt"]["f9"], d_feat, N=N, IH=h, IW=w, IC=192, OH=h, OW=w, OC=FEAT_LD, KH=1, KW=1,
                             mode=1, epi=1, mask=S["feat"])
                wgrad("f8_3", S["conv4"], d_feat, N=N, IH=h, IW=w, IC=512, OH=h, OW=w, OC=64, KH=1, KW=1, ld_dy=FEAT_LD)
                wgrad("f8_4", S["conv5"], d_feat.view(-1)[64:], N=N, IH=h, IW=w, IC=1024, OH=h, OW=w, OC=128, KH=1, KW=1, ld_dy=FEAT_LD)
        # ---- head
        if d_head_rows is None:
            if d_cam_low is None and g_fproj is None:
                return
            d_head_rows = E(N, h, w, HEAD_LD)
            gf = g_fproj.contiguous().float() if g_fproj is not None else None
            L.head_grad_rows(gf, d_cam_low.contiguous() if d_cam_low is not None else None, S["head"], d_head_rows, HEAD_LD, N, hw)
        if trainable("fc_proj") or trainable("fc8"):
            off, _ = self.offsets["fc_proj"]
            L.conv_wgrad(S["fea"], d_head_rows, self.flat_g[off:off + 149 * 4096], N=N, IH=h, IW=w, IC=4096, OH=h, OW=w,
                         OC=HEAD_LD, KH=1, KW=1, OC_dw=149)
        s7, _ = P["bn"]["bn7"]
        D = E(N, h, w, 4096)
        L.conv_igemm(d_head_rows, P["wt"]["head"], D, N=N, IH=h, IW=w, IC=HEAD_LD, OH=h, OW=w, OC=4096, KH=1, KW=1,
                     mode=1, epi=1, scale=s7, drop=masks["dropout7"] if masks else None, mask=S["fea"])
        # ---- blocks, last to first trainable
        for i in range(len(arch.BLOCKS) - 1, -1, -1):
            b = arch.BLOCKS[i]
            name, kind, cin, mid, cout, stride, fd, d, p = b
            if name in arch.FROZEN_BLOCKS:
                break
            same = arch.block_same_shape(b)
            ih, iw, oh, ow = S["dims"][name]
            sv = S[name]
            sa, _ = P["bn"][name + ".bn_branch2a"]
            first_trainable = name == "b3"               # its input comes from the frozen prefix
            if kind == "res":
                s1, _ = P["bn"][name + ".bn_branch2b1"]
                du = E(N, oh, ow, mid)
                L.conv_igemm(D, P["wt"][name + ".conv_branch2b1"], du, N=N, IH=oh, IW=ow, IC=cout, OH=oh, OW=ow, OC=mid,
                             KH=3, KW=3, dil=d, pad=d, mode=1, epi=1, scale=s1, mask=sv["v"])
                wgrad(name + ".conv_branch2b1", sv["v"], D, N=N, IH=oh, IW=ow, IC=mid, OH=oh, OW=ow, OC=cout, KH=3, KW=3, dil=d, pad=d)
                wgrad(name + ".conv_branch2a", sv["t"], du, N=N, IH=ih, IW=iw, IC=cin, OH=oh, OW=ow, OC=mid, KH=3, KW=3,
                      stride=stride, dil=fd, pad=fd)
                if not same:
                    wgrad(name + ".conv_branch1", sv["t"], D, N=N, IH=ih, IW=iw, IC=cin, OH=oh, OW=ow, OC=cout, KH=1, KW=1, stride=stride)
                if first_trainable:
                    break
                Din = E(N, ih, iw, cin)
                if same:
                    L.conv_igemm(du, P["wt"][name + ".conv_branch2a"], Din, N=N, IH=oh, IW=ow, IC=mid, OH=ih, OW=iw, OC=cin,
                                 KH=3, KW=3, stride=stride, dil=fd, pad=fd, mode=1, epi=1, scale=sa, mask=sv["t"], r_post=D)
                else:
                    tmp = E(N, ih, iw, cin)
                    L.conv_igemm(D, P["wt"][name + ".conv_branch1"], tmp, N=N, IH=oh, IW=ow, IC=cout, OH=ih, OW=iw, OC=cin,
                                 KH=1, KW=1, stride=stride, mode=1)
                    L.conv_igemm(du, P["wt"][name + ".conv_branch2a"], Din, N=N, IH=oh, IW=ow, IC=mid, OH=ih, OW=iw, OC=cin,
                                 KH=3, KW=3, stride=stride, dil=fd, pad=fd, mode=1, epi=1, scale=sa, mask=sv["t"], r_pre=tmp)
                D = Din
            else:
                c4, c2 = cout // 4, cout // 2
                s1, _ = P["bn"][name + ".bn_branch2b1"]
                s2, _ = P["bn"][name + ".bn_branch2b2"]
                d1 = masks[name + ".dropout_2b1"] if masks else None
                d2 = masks[name + ".dropout_2b2"] if masks else None
                du2 = E(N, oh, ow, c2)
                L.conv_igemm(D, P["wt"][name + ".conv_branch2b2"], du2, N=N, IH=oh, IW=ow, IC=cout, OH=oh, OW=ow, OC=c2,
                             KH=1, KW=1, mode=1, epi=1, scale=s2, drop=d2, mask=sv["v2"])
                wgrad(name + ".conv_branch2b2", sv["v2"], D, N=N, IH=oh, IW=ow, IC=c2, OH=oh, OW=ow, OC=cout, KH=1, KW=1)
                du1 = E(N, oh, ow, c4)
                L.conv_igemm(du2, P["wt"][name + ".conv_branch2b1"], du1, N=N, IH=oh, IW=ow, IC=c2, OH=oh, OW=ow, OC=c4,
                             KH=3, KW=3, dil=d, pad=d, mode=1, epi=1, scale=s1, drop=d1, mask=sv["v1"])
                wgrad(name + ".conv_branch2b1", sv["v1"], du2, N=N, IH=oh, IW=ow, IC=c4, OH=oh, OW=ow, OC=c2, KH=3, KW=3, dil=d, pad=d)
                wgrad(name + ".conv_branch1", sv["t"], D, N=N, IH=ih, IW=iw, IC=cin, OH=oh, OW=ow, OC=cout, KH=1, KW=1, stride=stride)
                wgrad(name + ".conv_branch2a", sv["t"], du1, N=N, IH=ih, IW=iw, IC=cin, OH=oh, OW=ow, OC=c4, KH=1, KW=1, stride=stride)
                tmp = E(N, ih, iw, cin)
                L.conv_igemm(D, P["wt"][name + ".conv_branch1"], tmp, N=N, IH=oh, IW=ow, IC=cout, OH=ih, OW=iw, OC=cin,
                             KH=1, KW=1, stride=stride, mode=1)
                Din = E(N, ih, iw, cin)
                L.conv_igemm(du1, P["wt"][name + ".conv_branch2a"], Din, N=N, IH=oh, IW=ow, IC=c4, OH=ih, OW=iw, OC=cin,
                             KH=1, KW=1, stride=stride, mode=1, epi=1, scale=sa, mask=sv["t"], r_pre=tmp)
                D = Din


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, eng, lowres):
        outs, S = eng.run_forward(x, save=True, lowres=lowres)
        ctx.eng, ctx.S = eng, S
        ctx.set_materialize_grads(False)
        if lowres:
            ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        S = ctx.S
        if S["lowres"]:
            # outs = (cam_low, rvd, f_proj view, head rows): g2 is d(f_proj) in NCHW-logical layout
            ctx.eng.run_backward(S, g0, g1, g2, None)
        else:
            ctx.eng.run_backward(S, g0, g1, g2, g3)
        ctx.S = None
        return None, None, None, None
