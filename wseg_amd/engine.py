"""Forward/backward orchestration of the ResNet-38d + CAM/PCM head on the HIP kernels.

Host-side plumbing only: allocates torch tensors (device memory), folds the frozen BatchNorms,
keeps the trainable weights in ONE flat f32 buffer (and their gradients in another — a single
RCCL all-reduce / fused SGD target), and issues the C-ABI kernels of libwseg_hip.so in the order
of network/resnet38d.py:160-189 and network/resnet38_contrast.py:31-75.  Activations are NHWC
("pixel rows") in the precision mode's dtype; 21-class maps are planar f32.

The whole network is one torch.autograd.Function: its backward runs the hand-written dgrad /
wgrad / PCM-backward kernels and accumulates straight into the flat gradient buffer.
"""
import os
import threading
import weakref

import torch

from . import arch
from . import _lib as L

HEAD_LD = 192          # fused head rows: [f_proj 128 | cam 21 | zero pad 43]  (a 256-wide row for the 256-tile kernels measured slower: profiles/HISTORY.md)
FUSE_SKIP = os.environ.get("WSEG_FUSE_SKIP", "1") != "0"   # bottleneck skip conv + last conv as one two-source launch (bf16)
FEAT_LD = 256          # PCM feature rows: [f8_3 64 | f8_4 128 | x_s 3 | zero pad 61]


DT_OF = {"bf16": L.BF16, "fp32": L.F32, "bf16x3": L.F32X3}   # bf16x3: f32 storage, conv / wgrad products as split-bf16 (3 bf16 MFMAs)


def _cdt(dt):
    """dtype override for the conv / wgrad launches: the split-bf16 mode runs on f32 tensors."""
    return L.F32X3 if dt == L.F32X3 else None


def _out_size(h, k, s, d):
    p = d * (k // 2)
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


class Engine:
    """One per Net INSTANCE (a replica made by nn.parallel.replicate gets its own, see Net._replicate_for_data_parallel).
    Re-entrancy (contrast_infer.py:69-73 calls one module from 8 threads): everything that mutates engine state — the flat
    buffers, the packs, the parameters' `.data` — happens under `self.lock`; a forward pass itself only reads them and
    allocates its own activations."""

    def __init__(self, net, parent=None):
        self._net_ref = weakref.ref(net)    # (no reference cycle; the Net owns the Engine)
        self.parent = parent                # replica: the engine of the module it was replicated from
        self.lock = threading.RLock()
        self.delegate = None                # set by ensure_flat on a replica that aliases its parent's flat buffer
        self.flat_w = None
        self.flat_g = None
        self.packs = None
        self.pack_key = None
        self.injected_masks = None
        self._order = None
        self.flat_w_version = 0             # bumped by the fused SGD step (raw kernel writes)
        self.flat_wb = None
        self.flat_wb_version = None
        self.block_done_hook = None         # called with the block name when all of its weight gradients are enqueued
        self.capture_ctx = False            # tests: keep the saved forward context of the last training pass in `last_ctx`
        self.last_ctx = None

    @property
    def net(self):
        net = self._net_ref()
        if net is None:
            raise RuntimeError("wseg_amd.Engine outlived its Net")
        return net

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

    def __deepcopy__(self, memo):
        return None                         # a copied Net builds its own engine on first use (Net._engine)

    def __reduce__(self):
        return (type(None), ())             # pickling a whole Net: the engine is derived state

    def ensure_flat(self, device):
        """(Re)build the flat weight / gradient buffers when the parameters moved, and return the engine whose buffers serve this
        module on `device`: itself, or the original's for an aliasing replica.  Serialised: eight inference threads may enter a fresh
        module at once (contrast_infer.py:69-73); callers use the RETURNED engine — `self.delegate` is written exactly once per call,
        under the lock, and never passes through None while another thread may be reading it."""
        with self.lock:
            return self._ensure_flat(device)

    def active(self, device):
        """The engine whose buffers serve this module on `device`: itself, or the original's for an aliasing replica."""
        return self.ensure_flat(device)

    def _ensure_flat(self, device):
        names = self.trainable_order()
        first = self.conv_param(names[0])
        par = self.parent
        if par is not None:
            # a replica on the original's device holds ALIASES of the original's parameters (nn.parallel.replicate hands device 0
            # the source tensors): when those already live in the original engine's flat buffer, its buffers and packs serve
            # this replica as they are; anything else (other device, original not flattened yet) gets buffers of its own below
            with par.lock:
                if (par.flat_w is not None and first.device == par.flat_w.device == device
                        and first.data_ptr() == par.flat_w.data_ptr()):
                    self.delegate = par
                    return par
        if (self.flat_w is not None and self.flat_w.device == first.device
                and first.data_ptr() == self.flat_w.data_ptr() and first.device == device):
            self.delegate = None
            return self
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
        self.flat_wb_version = None                          # the bf16 mirror (if any) is stale
        self.delegate = None
        return self

    def _mirror_fresh(self, names):
        """False when a parameter was modified through torch (load_state_dict, manual edits) since the mirror was written."""
        return getattr(self, "_mirror_pversions", None) == tuple(self.conv_param(n_)._version for n_ in names)

    def grad_buckets(self, after=("b7", "b5", "b4", "b3")):
        """Contiguous slices of flat_g in the order backward completes them: {block name: (begin, end)} — the slice is
        final once that block's weight gradients are enqueued (flat order = forward order; the heads sit behind b7)."""
        firsts = {}
        for b in arch.BLOCKS:
            if b[0] in arch.FROZEN_BLOCKS:
                continue
            firsts[b[0]] = min(self.offsets[c[0]][0] for c in arch.block_convs(b))
        out, end = {}, self.flat_g.numel()
        for name in after:
            out[name] = (firsts[name], end)
            end = firsts[name]
        assert end == 0, "the last bucket must reach the start of the flat buffer"
        return out

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

    @staticmethod
    def _x3(w32):
        """Split-bf16 pack of an f32 weight pack [rows][K] (K % 32 == 0): per 32 K-elements [32 hi | 32 lo] bf16 in the same bytes
        (kept in a float32-typed tensor of the same shape: only the conv kernels of dtype F32X3 read it)."""
        out = torch.empty_like(w32)
        L.pack_x3(w32.contiguous(), out)
        return out

    def frozen_key(self, device, dt):
        """Identity of everything the frozen prefix (conv1a, b2*, every folded BatchNorm) is computed from: precision, device, the versions of
        all buffers / BN parameters / frozen conv weights.  The frozen packs are rebuilt when it changes, and a lookahead prefix computed under
        another key is dropped (Trainer.step)."""
        net = self.net
        frozen_names = [c[0] for b in arch.BLOCKS if b[0] in arch.FROZEN_BLOCKS for c in arch.block_convs(b)]
        return (dt, str(device)) + tuple(b._version for b in net.buffers()) + \
            tuple(p._version for n_, p in net.named_parameters() if ".bn" in n_ or n_.startswith("bn7")) + \
            tuple((self.conv_param(n_)._version, self.conv_param(n_).data_ptr()) for n_ in frozen_names) + \
            (net.conv1a.weight._version, net.conv1a.weight.data_ptr())

    def ensure_packs(self, device, dt, defer_wt=False, late_stream=None):
        with self.lock:
            return self._ensure_packs(device, dt, defer_wt, late_stream)

    def _ensure_packs(self, device, dt, defer_wt=False, late_stream=None):
        """Packed (cast / transposed) weights + folded BatchNorms.  Frozen pieces (every BN, conv1a, b2*) are
        cached on their own key so a training step only re-packs the 40 trainable tensors.
        defer_wt: the transposed (dgrad) packs are only needed by the backward pass — the fused training step lets
        `finish_packs()` make them on a side stream during the loss phase, when the chip is mostly idle.
        late_stream: the packs the forward pass needs only from b5 on (the K-concatenated skip packs, the head, f9: ~10 small launches that would
        otherwise sit in front of every step) are made on that stream; the forward pass waits for them where it first uses one (`_join_late_packs`)."""
        net = self.net
        tdt = L.TORCH_DTYPE[dt]
        fkey = self.frozen_key(device, dt)
        if getattr(self, "_frozen_packs", None) is None or self._frozen_key != fkey:
            F_ = {"w": {}, "bn": {}}
            for b in arch.BLOCKS:
                for (bname, c) in arch.block_bns(b):
                    F_["bn"][bname] = self._bn_fold(bname, device)
                if b[0] in arch.FROZEN_BLOCKS:
                    for (cname, ci, co, k, s, d) in arch.block_convs(b):
                        wf = torch.empty(co, k * k, ci, device=device, dtype=tdt)
                        L.pack_weights(self.conv_param(cname).detach(), wf, None, co, k * k, ci, co, ci, L.F32 if dt == L.F32X3 else dt)
                        F_["w"][cname] = self._x3(wf) if dt == L.F32X3 else wf
            F_["bn"]["bn7"] = self._bn_fold("bn7", device)
            F_["w"]["conv1a_kc"] = net.conv1a.weight.detach().to(device).float().permute(2, 3, 1, 0).reshape(27, 64).contiguous()   # [k = (ky*3+kx)*3+ic][oc] for the packed-FMA stem
            self._frozen_packs, self._frozen_key = F_, fkey
            self.packs = None
        names = self.trainable_order()
        key = (dt, str(device), self.flat_w_version) + tuple(self.conv_param(n_)._version for n_ in names)
        if self.packs is not None and key == self.pack_key:
            return self.packs
        P = {"w": dict(self._frozen_packs["w"]), "wt": {}, "bn": self._frozen_packs["bn"]}
        no_dgrad = {"b3.conv_branch1", "b3.conv_branch2a"}
        # forward packs [OC][T][IC] have exactly the layout of the flat master buffer: in f32 mode they ARE views of
        # it, in bf16 mode they are views of ONE cast copy (a single launch instead of one per layer)
        if dt == L.BF16:
            if getattr(self, "flat_wb", None) is None or self.flat_wb.numel() != self.flat_w.numel() or self.flat_wb.device != device:
                self.flat_wb = torch.empty(self.flat_w.numel(), device=device, dtype=torch.bfloat16)
                self.flat_wb_version = None
            if self.flat_wb_version != self.flat_w_version or not self._mirror_fresh(names):
                L.to_bf16(self.flat_w, self.flat_wb)             # (normally the fused SGD step has already written it)
                self.flat_wb_version = self.flat_w_version
            self._mirror_pversions = tuple(self.conv_param(n_)._version for n_ in names)
            mirror = self.flat_wb
        elif dt == L.F32X3:
            if getattr(self, "flat_w3", None) is None or self.flat_w3.numel() != self.flat_w.numel() or self.flat_w3.device != device:
                self.flat_w3 = torch.empty_like(self.flat_w)
            L.pack_x3(self.flat_w, self.flat_w3)             # (every trainable tensor's rows are whole 32-element groups, f9 aside: it has its own pack)
            mirror = self.flat_w3
        else:
            mirror = self.flat_w
        # transposed (dgrad) packs [IC][T][OC]: ONE flat buffer, ONE launch over all layers (same offsets as flat_w)
        if (getattr(self, "flat_wt", None) is None or self.flat_wt.dtype != tdt or self.flat_wt.device != device
                or self.flat_wt.numel() != self.flat_w.numel()):
            self.flat_wt = torch.empty(self.flat_w.numel(), device=device, dtype=tdt)
            rows, tiles = [], 0
            for b in arch.BLOCKS:
                if b[0] in arch.FROZEN_BLOCKS:
                    continue
                for (cname, ci, co, k, s, d) in arch.block_convs(b):
                    if cname in no_dgrad:
                        continue
                    off, n = self.offsets[cname]
                    rows.append([tiles, off, off, co, k * k, ci])
                    tiles += ((co + 31) // 32) * ((ci + 31) // 32) * k * k
            self._wt_table = torch.tensor(rows, dtype=torch.int64, device=device)
            self._wt_tiles = tiles
            rows64, t64 = [], 0                              # the bf16 -> bf16 variant works on 64x64 tiles
            for (_t, off_i, off_o, co, T_, ci) in rows:
                rows64.append([t64, off_i, off_o, co, T_, ci])
                t64 += ((co + 63) // 64) * ((ci + 63) // 64) * T_
            self._wt_table64 = torch.tensor(rows64, dtype=torch.int64, device=device)
            self._wt_tiles64 = t64
        self._wt_pending = (dt, mirror)
        if dt == L.F32X3 and (getattr(self, "flat_wt3", None) is None or self.flat_wt3.numel() != self.flat_w.numel() or self.flat_wt3.device != device):
            self.flat_wt3 = torch.empty_like(self.flat_w)
        for b in arch.BLOCKS:
            if b[0] in arch.FROZEN_BLOCKS:
                continue
            for (cname, ci, co, k, s, d) in arch.block_convs(b):
                T = k * k
                off, n = self.offsets[cname]
                P["w"][cname] = mirror[off:off + n].view(co, T, ci)
                if cname not in no_dgrad:
                    P["wt"][cname] = (self.flat_wt3 if dt == L.F32X3 else self.flat_wt)[off:off + n].view(ci, T, co)
        def late():
            # bf16 mode: a block's skip conv and its last conv as ONE two-source product (K-concatenation): rows [W_a[oc] | W_b[oc]] — the skip output is
            # then neither written nor re-read (444 MB each way for b7).  b6 / b7: [W_branch1 | W_branch2b2]; b5 (the residual-block form, 3x3 last conv +
            # 1x1 skip conv at stride 1): [W_branch2b1 (9 taps) | W_branch1].  All pieces are copied from the bf16 mirror in ONE launch into buffers
            # that persist across steps (`_fused_plan`).
            if dt == L.BF16 and FUSE_SKIP:
                plan = self._fused_plan(device)
                for nm, buf in plan["fwd_bufs"].items():
                    P["w"][nm + ".skip_fused"] = buf
                L.copy2d_batch(mirror, plan["fwd_flat"], plan["fwd_table"], plan["fwd_table"].shape[0], plan["fwd_chunks"])
            # fused head: rows [fc_proj | fc8 | 0]; its transposed pack is made from the two f32 masters directly
            hb = getattr(self, "_head_bufs", None)               # persistent: the zero padding (rows / columns 149..191) is written once
            if hb is None or hb[0] != (dt, str(device)):
                # (the forward pack holds 256 rows — 107 of them zero: the head GEMM then runs on the 256-tile kernel, which reads whole 256-row weight
                #  tiles and masks the columns >= HEAD_LD; conv_igemm w_rows)
                hb = self._head_bufs = ((dt, str(device)), torch.zeros(256, 1, 4096, device=device, dtype=tdt),
                                        torch.zeros(4096, 1, HEAD_LD, device=device, dtype=tdt),
                                        torch.empty(192, 1, FEAT_LD, device=device, dtype=tdt), torch.empty(FEAT_LD, 1, 192, device=device, dtype=tdt))
            wh, wht = hb[1], hb[2]
            pdt = L.F32 if dt == L.F32X3 else dt                  # (split-bf16: f32 packs first, split below)
            L.pack_weights(net.fc_proj.weight.detach(), wh, None, 128, 1, 4096, 128, 4096, pdt)
            L.pack_weights(net.fc8.weight.detach(), wh[128:], None, 21, 1, 4096, 21, 4096, pdt)
            off, _ = self.offsets["fc_proj"]                     # fc_proj and fc8 are adjacent in flat_w: one [149,4096] master
            L.pack_weights(self.flat_w[off:off + 149 * 4096], None, wht, 149, 1, 4096, HEAD_LD, 4096, pdt)
            if dt == L.F32X3:
                wh, wht = self._x3(wh), self._x3(wht)
            P["w"]["head"], P["wt"]["head"] = wh, wht
            for nm, (co, ci) in (("f8_3", (64, 512)), ("f8_4", (128, 1024))):
                off, n = self.offsets[nm]
                P["w"][nm] = mirror[off:off + n].view(co, 1, ci)
            # f9: input columns re-ordered to the internal feature layout [f8_3 | f8_4 | x_s | pad] = the master's columns rotated by 3
            off9, n9 = self.offsets["f9"]
            wf, wt = hb[3], hb[4]
            L.pack_weights(self.flat_w[off9:off9 + n9], wf, wt, 192, 1, 195, 192, FEAT_LD, pdt, ic_rot=3)
            if dt == L.F32X3:
                wf, wt = self._x3(wf), self._x3(wt)
            P["w"]["f9"], P["wt"]["f9"] = wf, wt

        # (the names exist at once — the forward pass tests `in P["w"]` — their contents when `_late_ev` has been waited for)
        self._late_ev = None
        if late_stream is not None:
            cur = torch.cuda.current_stream(device)
            late_stream.wait_stream(cur)
            with torch.cuda.stream(late_stream):
                late()
            self._late_ev = late_stream.record_event()
        else:
            late()
        self.packs, self.pack_key = P, key
        if not defer_wt:
            self.finish_packs()
        return P

    def _join_late_packs(self, device):
        ev = getattr(self, "_late_ev", None)
        if ev is not None:
            torch.cuda.current_stream(device).wait_event(ev)
            self._late_ev = None

    def finish_packs(self):
        """The transposed (dgrad) packs [IC][T][OC] of the current weights: ONE launch over all layers into the flat buffer the
        `P["wt"]` views point into, plus the K-concatenated backward packs of the two-source launches.  Runs on the CURRENT
        stream (the fused step calls it on a side stream and joins before the backward pass); no-op when already done."""
        pend = getattr(self, "_wt_pending", None)
        if pend is None:
            return
        self._wt_pending = None
        dt, mirror = pend
        P = self.packs
        if dt == L.BF16:                                     # from the bf16 mirror (written by the fused SGD): a third of the traffic
            L.pack_transposed_batch_bf16(mirror, self.flat_wt, self._wt_table64, self._wt_table64.shape[0], self._wt_tiles64)
        else:
            L.pack_transposed_batch(self.flat_w, self.flat_wt, self._wt_table, self._wt_table.shape[0], self._wt_tiles, L.F32 if dt == L.F32X3 else dt)
            if dt == L.F32X3:
                L.pack_x3(self.flat_wt, self.flat_wt3)
        if dt == L.BF16 and any(k.endswith(".skip_fused") for k in P["w"]):
            # the K-concatenated backward packs of the two-source launches, one launch from the transposed packs just made.  b5 (res):
            # d_t = dgrad_3x3(du; W_2a) + D . W_branch1 — nine taps of du, then one K segment of D; b6 / b7 (bot): d_t = D . W_branch1 + du1 . W_branch2a
            plan = self._fused_plan(self.flat_wt.device)
            for nm, buf in plan["bwd_bufs"].items():
                P["wt"][nm + ".skip_fused"] = buf
            L.copy2d_batch(self.flat_wt, plan["bwd_flat"], plan["bwd_table"], plan["bwd_table"].shape[0], plan["bwd_chunks"])

    def _fused_blocks(self):
        """[(name, kind, cin, mid, cout)] of the blocks whose skip conv and last conv run as one two-source launch (bf16 mode)"""
        out = []
        for b in arch.BLOCKS:
            nm, kind, cin_, mid_, cout_, stride = b[0], b[1], b[2], b[3], b[4], b[5]
            if nm in arch.FROZEN_BLOCKS or stride != 1 or cout_ % 256:
                continue
            if kind == "res" and not arch.block_same_shape(b) and cin_ % 256 == 0:
                out.append((nm, kind, cin_, mid_, cout_))
            elif kind != "res" and cin_ == cout_ // 2:
                out.append((nm, kind, cin_, mid_, cout_))
        return out

    def _fused_plan(self, device):
        """Persistent K-concatenated pack buffers + the piece tables of wseg_copy2d_batch (16-byte = 8-element units), built once per device."""
        plan = getattr(self, "_fused", None)
        if plan is not None and plan["device"] == device:
            return plan

        def build(pieces_of):
            rows_tab, bufs, sizes, chunk0, base = [], {}, [], 0, 0
            for (nm, kind, cin_, mid_, cout_) in self._fused_blocks():
                nrows, width, pieces = pieces_of(nm, kind, cin_, mid_, cout_)       # pieces: (param name, cols, column offset)
                for (pname, cols, coff) in pieces:
                    off, n = self.offsets[pname]
                    assert n == nrows * cols and off % 8 == 0 and cols % 8 == 0 and coff % 8 == 0 and width % 8 == 0 and base % 8 == 0
                    rows_tab.append([chunk0, off // 8, (base + coff) // 8, nrows, cols // 8, cols // 8, width // 8])
                    chunk0 += nrows * (cols // 8)
                sizes.append((nm, base, nrows, width))
                base += nrows * width
            flat = torch.empty(base, device=device, dtype=torch.bfloat16)
            for nm, b0, nrows, width in sizes:
                bufs[nm] = flat[b0:b0 + nrows * width].view(nrows, width)
            return flat, bufs, torch.tensor(rows_tab, dtype=torch.int64, device=device), chunk0

        def fwd_pieces(nm, kind, cin_, mid_, cout_):
            if kind == "res":
                return cout_, 9 * mid_ + cin_, [(nm + ".conv_branch2b1", 9 * mid_, 0), (nm + ".conv_branch1", cin_, 9 * mid_)]
            return cout_, cin_ + cout_ // 2, [(nm + ".conv_branch1", cin_, 0), (nm + ".conv_branch2b2", cout_ // 2, cin_)]

        def bwd_pieces(nm, kind, cin_, mid_, cout_):          # transposed packs [cin][T][cout']: rows = cin
            if kind == "res":
                return cin_, 9 * mid_ + cout_, [(nm + ".conv_branch2a", 9 * mid_, 0), (nm + ".conv_branch1", cout_, 9 * mid_)]
            return cin_, cout_ + cout_ // 4, [(nm + ".conv_branch1", cout_, 0), (nm + ".conv_branch2a", cout_ // 4, cout_)]

        ff, fb, ft, fc = build(fwd_pieces)
        bf, bb, bt, bc = build(bwd_pieces)
        self._fused = dict(device=device, fwd_flat=ff, fwd_bufs=fb, fwd_table=ft, fwd_chunks=fc, bwd_flat=bf, bwd_bufs=bb, bwd_table=bt, bwd_chunks=bc)
        return self._fused

    # ------------------------------------------------------------------ dropout
    MASK_SPECS = (("b6.dropout_2b1", 512, 0.3), ("b6.dropout_2b2", 1024, 0.3),
                  ("b7.dropout_2b1", 1024, 0.5), ("b7.dropout_2b2", 2048, 0.5), ("dropout7", 4096, 0.5))

    def _masks(self, n, views, device):
        """Dropout2d scales [views*n, C] per dropout site (rows of view 1, then view 2): injected (parity tests) or drawn
        on the device — one uniform draw + one kernel for all five sites."""
        net = self.net
        if not net.training:
            return None
        if self.injected_masks:
            per_view = []
            for _ in range(views):
                m = self.injected_masks.pop(0)
                per_view.append({k: v.to(device=device, dtype=torch.float32) for k, v in m.items()})
            return {k: torch.cat([m[k] for m in per_view], dim=0).contiguous() for k in per_view[0]}
        rows = views * n
        total = sum(c for _, c, _ in self.MASK_SPECS)
        u = torch.rand(rows * total, device=device)
        flat = torch.empty_like(u)
        L.dropout_scale(u, flat, rows * (512 + 1024), 0.3, 0.5)          # the two b6 sites (p = 0.3) come first
        out, off = {}, 0
        for key, c, _p in self.MASK_SPECS:
            out[key] = flat[off:off + rows * c].view(rows, c)
            off += rows * c
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, x, lowres=False):
        if not x.is_cuda:
            raise RuntimeError("wseg_amd.Net runs only on an MI355X (HIP) device; there is no CPU fallback")
        x = x.contiguous().float()
        act = self.ensure_flat(x.device)
        if act is not self:                                   # replica whose parameters alias the original's flat buffer
            return act.forward(x, lowres)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        if need_grad:
            anchor = self.flat_w.new_zeros((), requires_grad=True)
            return _NetFunction.apply(x, anchor, self, lowres)
        outs, _ = self.run_forward([x], save=False, lowres=lowres)
        return outs[0]

    def _run_blocks(self, xs, state, first, last, save, S, final_t=None):
        """Blocks arch.BLOCKS[first:last] (first == 0: conv1a first) over the batched views.  state: the dict an earlier call returned."""
        net = self.net
        V = len(xs)
        dev = xs[0].device
        dt = DT_OF[net.precision]
        tdt = L.TORCH_DTYPE[dt]
        P = self.ensure_packs(dev, dt)
        N = xs[0].shape[0]
        masks = S["masks"] if S is not None else None

        def rows_of(dims):
            return sum(N * h * w for (h, w) in dims)

        def offs_of(dims):
            o, out = 0, []
            for (h, w) in dims:
                out.append(o)
                o += N * h * w
            return out

        def E(m, c):
            return torch.empty((m, c), device=dev, dtype=tdt)

        def conv(inp, wname, out, out2, cin, cout, k, stride, dil, din, dout, **kw):
            seg2 = (din[1][0], din[1][1], dout[1][0], dout[1][1]) if V == 2 else None
            L.conv_igemm(inp, P["w"][wname], out, out2, N=N, IH=din[0][0], IW=din[0][1], IC=cin, OH=dout[0][0], OW=dout[0][1],
                         OC=cout, KH=k, KW=k, stride=stride, dil=dil, pad=dil * (k // 2), seg2=seg2, dtype=_cdt(dt), **kw)

        def next_bn(i):
            if i + 1 < len(arch.BLOCKS):
                return P["bn"][arch.BLOCKS[i + 1][0] + ".bn_branch2a"], None
            return P["bn"]["bn7"], (masks["dropout7"] if masks else None)

        sdims = {}
        conv4 = conv5 = None
        if first == 0:
            dims = [(x.shape[2], x.shape[3]) for x in xs]
            sc, sh = P["bn"]["b2.bn_branch2a"]
            t = E(rows_of(dims), 64)
            for x, off, (H, W) in zip(xs, offs_of(dims), dims):
                L.stem_conv_kc(x, P["w"]["conv1a_kc"], sc, sh, None, t[off:], N, H, W, L.F32 if dt == L.F32X3 else dt)
            xraw = None
        else:
            t, xraw, dims = state["t"], state["xraw"], state["dims"]
        for i in range(first, last):
            b = arch.BLOCKS[i]
            name, kind, cin, mid, cout, stride, fd, d, p = b
            same = arch.block_same_shape(b)
            if (name + ".skip_fused") in P["w"]:
                self._join_late_packs(dev)
            (nsc, nsh), ndrop = next_bn(i)
            nxt_same = i + 1 < len(arch.BLOCKS) and arch.block_same_shape(arch.BLOCKS[i + 1])
            k0 = 3 if kind == "res" else 1
            odims = [(_out_size(h, k0, stride, fd if kind == "res" else 1), _out_size(w, k0, stride, fd if kind == "res" else 1)) for (h, w) in dims]
            Mo = rows_of(odims)
            if kind == "res":
                s1, sh1 = P["bn"][name + ".bn_branch2b1"]
                v = E(Mo, mid)
                conv(t, name + ".conv_branch2a", None, v, cin, mid, 3, stride, fd, dims, odims, scale=s1, shift=sh1)
                xn = E(Mo, cout) if nxt_same else None
                tn = final_t if (final_t is not None and i == last - 1) else E(Mo, cout)
                assert tn.shape == (Mo, cout) and tn.dtype == tdt
                if (name + ".skip_fused") in P["w"]:            # last conv + 1x1 skip conv as one two-source launch
                    conv(v, name + ".skip_fused", xn, tn, mid, cout, 3, 1, d, odims, odims, in2=t, IC2=cin, scale=nsc, shift=nsh, drop=ndrop)
                else:
                    if same:
                        rpost = xraw
                    else:
                        rpost = E(Mo, cout)
                        conv(t, name + ".conv_branch1", rpost, None, cin, cout, 1, stride, 1, dims, odims)
                    conv(v, name + ".conv_branch2b1", xn, tn, mid, cout, 3, 1, d, odims, odims, r_post=rpost, scale=nsc, shift=nsh, drop=ndrop)
                if save:
                    S[name] = dict(t=t, v=v)
            else:
                c4, c2 = cout // 4, cout // 2
                s1, sh1 = P["bn"][name + ".bn_branch2b1"]
                s2, sh2 = P["bn"][name + ".bn_branch2b2"]
                d1 = masks[name + ".dropout_2b1"] if masks else None
                d2 = masks[name + ".dropout_2b2"] if masks else None
                v1 = E(Mo, c4)
                conv(t, name + ".conv_branch2a", None, v1, cin, c4, 1, stride, 1, dims, odims, scale=s1, shift=sh1, drop=d1)
                v2 = E(Mo, c2)
                conv(v1, name + ".conv_branch2b1", None, v2, c4, c2, 3, 1, d, odims, odims, scale=s2, shift=sh2, drop=d2)
                xn = None
                tn = E(Mo, cout)
                if (name + ".skip_fused") in P["w"]:            # out = [t | v2] . [W_branch1 | W_branch2b2]^T in one launch
                    conv(t, name + ".skip_fused", xn, tn, cin, cout, 1, 1, 1, odims, odims, in2=v2, scale=nsc, shift=nsh, drop=ndrop)
                else:
                    b1 = E(Mo, cout)
                    conv(t, name + ".conv_branch1", b1, None, cin, cout, 1, stride, 1, dims, odims)
                    conv(v2, name + ".conv_branch2b2", xn, tn, c2, cout, 1, 1, 1, odims, odims, r_post=b1, scale=nsc, shift=nsh, drop=ndrop)
                if save:
                    S[name] = dict(t=t, v1=v1, v2=v2)
            sdims[name] = (dims, odims)
            if S is not None:
                S["dims"][name] = (dims, odims)
            if name == "b5":
                conv4 = t
            if name == "b6":
                conv5 = t
            t, xraw, dims = tn, xn, odims
        return dict(t=t, xraw=xraw, dims=dims, conv4=conv4, conv5=conv5, sdims=sdims, N=N, V=V, dt=dt,
                    in_dims=state["in_dims"] if state is not None else [tuple(x.shape[2:]) for x in xs])

    def prefix_out_shape(self, xs):
        """(rows, channels) of run_prefix's result for these views"""
        N = xs[0].shape[0]
        dims = [tuple(x.shape[2:]) for x in xs]
        for b in arch.BLOCKS[:arch.N_FROZEN_BLOCKS]:
            name, kind, cin, mid, cout, stride, fd, d, p = b
            k0 = 3 if kind == "res" else 1
            dims = [(_out_size(h, k0, stride, fd if kind == "res" else 1), _out_size(w, k0, stride, fd if kind == "res" else 1)) for (h, w) in dims]
        return sum(N * h * w for (h, w) in dims), arch.BLOCKS[arch.N_FROZEN_BLOCKS - 1][4]

    def run_prefix(self, xs, out=None):
        """The part of the forward pass that depends on no trainable weight: conv1a and the blocks Net.train() freezes (b2, b2_1, b2_2;
        resnet38_contrast.py:86-95 `not_training`), for one or two batched views.  Returns what run_forward continues from.  Because it depends only on the
        IMAGES, the fused step runs it for the NEXT batch on a side stream inside the loss phase of the current step (loss_hip.step `lookahead`).
        out: a [prefix_out_shape] tensor for the result — the fused step allocates it on the CALLER's stream, so that the one tensor that crosses from the
        prefix stream to the next step belongs to the caller's allocator pool (no record_stream bookkeeping: with it the reserved memory grew by 6 GB)."""
        return self._run_blocks(xs, None, 0, arch.N_FROZEN_BLOCKS, False, None, final_t=out)

    def run_forward(self, xs, save, lowres=False, prefix=None):
        """xs: list of one or two image batches (same N).  Two views are BATCHED: every activation is one row
        matrix [rows(view 1) ++ rows(view 2)][C] and every conv / wgrad is ONE launch over both row segments
        (the 128x128 view alone cannot fill the chip).  Returns ([per-view output tuple], saved context).
        prefix: the result of run_prefix(xs) when the caller has already computed it."""
        net = self.net
        V = len(xs)
        assert V in (1, 2)
        dev = xs[0].device
        dt = DT_OF[net.precision]
        tdt = L.TORCH_DTYPE[dt]
        P = self.ensure_packs(dev, dt)
        N = xs[0].shape[0]
        assert all(x.shape[0] == N for x in xs)
        masks = self._masks(N, V, dev)
        S = {"masks": masks, "dims": {}, "N": N, "V": V, "dt": dt, "xs": xs, "lowres": lowres}
        if prefix is None:                                    # (the frozen blocks' activations are kept only for the tests' gate capture)
            prefix = self._run_blocks(xs, None, 0, arch.N_FROZEN_BLOCKS, save and self.capture_ctx, S if (save and self.capture_ctx) else None)
        assert prefix["N"] == N and prefix["V"] == V and prefix["dt"] == dt and prefix["in_dims"] == [tuple(x.shape[2:]) for x in xs], \
            "run_forward: the prefix was computed for another batch shape / precision"
        S["dims"].update(prefix["sdims"])
        st = self._run_blocks(xs, prefix, arch.N_FROZEN_BLOCKS, len(arch.BLOCKS), save, S)
        t, dims, conv4, conv5 = st["t"], st["dims"], st["conv4"], st["conv5"]

        def rows_of(dims):
            return sum(N * h * w for (h, w) in dims)

        def offs_of(dims):
            o, out = 0, []
            for (h, w) in dims:
                out.append(o)
                o += N * h * w
            return out

        def E(m, c):
            return torch.empty((m, c), device=dev, dtype=tdt)

        def conv(inp, wname, out, out2, cin, cout, k, stride, dil, din, dout, **kw):
            seg2 = (din[1][0], din[1][1], dout[1][0], dout[1][1]) if V == 2 else None
            L.conv_igemm(inp, P["w"][wname], out, out2, N=N, IH=din[0][0], IW=din[0][1], IC=cin, OH=dout[0][0], OW=dout[0][1],
                         OC=cout, KH=k, KW=k, stride=stride, dil=dil, pad=dil * (k // 2), seg2=seg2, dtype=_cdt(dt), **kw)

        fea = t                                               # relu(bn7(x)) * dropout7   [M,4096]
        self._join_late_packs(dev)
        M = rows_of(dims)
        offs = offs_of(dims)
        head = E(M, HEAD_LD)
        conv(fea, "head", head, None, 4096, HEAD_LD, 1, 1, 1, dims, dims, relu_lt=128, w_rows=256)
        feat = E(M, FEAT_LD)
        conv(conv4, "f8_3", feat, None, 512, 64, 1, 1, 1, dims, dims, epi=2, ld_out=FEAT_LD)
        conv(conv5, "f8_4", feat.view(-1)[64:], None, 1024, 128, 1, 1, 1, dims, dims, epi=2, ld_out=FEAT_LD)
        G = torch.empty(M, 32, device=dev, dtype=torch.float32)
        views = []
        for x, off, (h, w) in zip(xs, offs, dims):
            hw = h * w
            cam_low = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
            cmax = torch.empty(N, 21, device=dev, dtype=torch.float32)
            L.head_split(head[off:], HEAD_LD, 128, cam_low, cmax, N, hw)
            L.cam_gate(cam_low, cmax, G[off:], N, hw)
            L.pcm_xs(x, feat[off:], FEAT_LD, 192, FEAT_LD, N, x.shape[2], x.shape[3], h, w)
            views.append(dict(h=h, w=w, H=x.shape[2], W=x.shape[3], off=off, rows=N * hw, cam_low=cam_low))
        Fm = E(M, 192)
        conv(feat, "f9", Fm, None, FEAT_LD, 192, 1, 1, 1, dims, dims)
        Fh = torch.empty(M, 192, device=dev, dtype=torch.float32)
        nrm = torch.empty(M, device=dev, dtype=torch.float32)
        L.l2norm_forward(Fm, 192, Fh, nrm, M)
        Fb = Gb = Gl = None
        if dt == L.BF16:                                      # bf16-MFMA PCM (throughput mode); fp32 mode keeps the exact-f32 kernel
            Fb = torch.empty(M, 192, device=dev, dtype=torch.bfloat16)
            Gb = torch.empty(M, 32, device=dev, dtype=torch.bfloat16)
            L.to_bf16(Fh, Fb)
            if save:                                          # the backward pass takes the gate map as hi + lo (split precision, csrc/pcm.hip)
                Gl = torch.empty(M, 32, device=dev, dtype=torch.bfloat16)
                L.split_bf16(G, Gb, Gl)
            else:
                L.to_bf16(G, Gb)
        outs = []
        for vw, x in zip(views, xs):
            h, w, off, hw = vw["h"], vw["w"], vw["off"], vw["h"] * vw["w"]
            rvd = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
            den = torch.empty(N, hw, device=dev, dtype=torch.float32)
            if dt == L.BF16:
                L.pcm_forward_bf16(Fb[off:], Gb[off:], rvd, den, N, hw)
            else:
                L.pcm_forward(Fh[off:], G[off:], rvd, den, N, hw)
            head_v = head[off:off + N * hw].view(N, h, w, HEAD_LD)
            f_proj = head_v[..., :128].permute(0, 3, 1, 2)
            if lowres:
                outs.append((vw["cam_low"], rvd, f_proj, head_v))
            else:
                H, W = vw["H"], vw["W"]
                cam = torch.empty(N, 21, H, W, device=dev, dtype=torch.float32)
                cam_rv = torch.empty(N, 21, H, W, device=dev, dtype=torch.float32)
                L.resize_planar_fwd(vw["cam_low"], cam, N * 21, h, w, H, W, True)
                L.resize_planar_fwd(rvd, cam_rv, N * 21, h, w, H, W, True)
                outs.append((cam, cam_rv, f_proj.float() if dt == L.BF16 else f_proj, rvd))
            # (the saved copy protects the PCM backward from a caller that edits its output in place; the fused step — lowres — only reads it)
            vw.update(rvd=(rvd if lowres else rvd.clone()) if save else None, den=den)
        if save and self.capture_ctx:
            self.last_ctx = S
        if save:
            S.update(fea=fea, head=head, G=G, feat=feat, Fm=Fm, Fh=Fh, Fb=Fb, Gb=Gb, Gl=Gl, nrm=nrm, conv4=conv4, conv5=conv5,
                     views=views, hdims=dims, M=M)
        return outs, S

    # ------------------------------------------------------------------ backward
    def run_backward(self, S, grads, d_head_rows=None):
        """grads: per view (g_cam, g_cam_rv, g_fproj, g_rvd) — gradients of the view's four outputs (any may be
        None); in the fused (lowres) path g_cam / g_cam_rv are gradients of the stride-8 maps and `d_head_rows`
        (joint rows [f_proj | cam | pad]) may be supplied directly.  Accumulates into flat_g."""
        self.finish_packs()                                  # (no-op when the caller already made the transposed packs)
        P = self.packs
        dt = S["dt"]
        tdt = L.TORCH_DTYPE[dt]
        N, V, M = S["N"], S["V"], S["M"]
        hdims = S["hdims"]
        dev = S["fea"].device
        masks = S["masks"]
        self.attach_grads()

        def E(m, c):
            return torch.empty((m, c), device=dev, dtype=tdt)

        def rows_of(dims):
            return sum(N * h * w for (h, w) in dims)

        def trainable(nm):
            return self.conv_param(nm).requires_grad

        def seg(din, dout):
            return (din[1][0], din[1][1], dout[1][0], dout[1][1]) if V == 2 else None

        # Weight gradients only feed flat_g: with WSEG_WGRAD_STREAM=1 they run on a second HIP stream so that the
        # partially filled last round of a dgrad launch (1 workgroup / CU kernels) is back-filled by wgrad
        # workgroups and vice versa.  Operands are kept alive until the streams join.
        main = torch.cuda.current_stream(dev)
        wstream = None
        if os.environ.get("WSEG_WGRAD_STREAM", "0") == "1":
            wstream = getattr(self, "_wgrad_stream", None)
            if wstream is None or wstream.device != dev:
                wstream = self._wgrad_stream = torch.cuda.Stream(dev)
            wstream.wait_stream(main)
        keep = []
        pcm_join = [None]                                    # the PCM branch's stream until main has waited for it

        planes = {}                                          # split-bf16 mode: (hi, lo) bf16 planes of an f32 operand, made once per tensor

        def split(t_):
            key = (t_.data_ptr(), t_.numel())
            if key not in planes:
                hi, lo = torch.empty(t_.shape, device=dev, dtype=torch.bfloat16), torch.empty(t_.shape, device=dev, dtype=torch.bfloat16)
                L.split_bf16(t_, hi, lo)
                planes[key] = (hi, lo, t_)                   # (keeps the source alive: the key is its address)
            return planes[key][:2]

        def wgrad_launch(x, dy, dw, big_x3, **args):
            if big_x3:
                # split-bf16 products on the bf16 pixel-reduction kernel: dW += X_lo^T dY_hi + X_hi^T dY_lo + X_hi^T dY_hi (it accumulates
                # with float atomics anyway) — 3 launches at the bf16 kernel's rate beat one launch of the f32-tile kernel that splits inside
                (xh, xl), (dh, dl) = split(x), split(dy)
                args = dict(args, dtype=None)
                L.conv_wgrad(xl, dh, dw, **args)
                L.conv_wgrad(xh, dl, dw, **args)
                L.conv_wgrad(xh, dh, dw, **args)
            else:
                L.conv_wgrad(x, dy, dw, **args)

        def wgrad(nm, x, dy, cin, cout, k, stride, dil, din, dout, **kw):
            if trainable(nm):
                off, n = self.offsets[nm]
                args = dict(N=N, IH=din[0][0], IW=din[0][1], IC=cin, OH=dout[0][0], OW=dout[0][1],
                            OC=cout, KH=k, KW=k, stride=stride, dil=dil, pad=dil * (k // 2), seg2=seg(din, dout), dtype=_cdt(dt), **kw)
                big_x3 = dt == L.F32X3 and cin >= 256 and cout >= 256 and not kw and x.is_contiguous() and dy.is_contiguous() \
                    and x.shape[1] == cin and dy.shape[1] == cout
                if wstream is None:
                    wgrad_launch(x, dy, self.flat_g[off:off + n], big_x3, **args)
                else:
                    keep.append((x, dy))
                    wstream.wait_event(torch.cuda.current_stream(dev).record_event())      # (main, or the PCM branch's stream)
                    with torch.cuda.stream(wstream):
                        wgrad_launch(x, dy, self.flat_g[off:off + n], big_x3, **args)

        def block_done(nm):
            """All weight gradients of block `nm` are enqueued: the data-parallel trainer may start reducing its bucket.  With a
            separate wgrad stream the collective (issued behind `main`) must first wait for the kernels on that stream."""
            if self.block_done_hook is not None:
                if wstream is not None:
                    main.wait_event(wstream.record_event())
                if pcm_join[0] is not None:                  # the last bucket (b7 + heads) holds the PCM branch's f9 / f8_3 / f8_4 gradients
                    main.wait_stream(pcm_join[0])
                    pcm_join[0] = None
                self.block_done_hook(nm)

        def pair_args(pair):
            """`pair_wgrad` of L.conv_igemm for the weight gradient (name, x, dy, cin, cout, k, stride, dil, din, dout), or None (then the caller
            launches it on its own): bf16 mode, no separate weight-gradient stream"""
            if pair is None:
                return None
            pnm, px, pdy, pcin, pcout, pk, pstride, pdil, pdin, pdout = pair
            if not (dt == L.BF16 and wstream is None and trainable(pnm)):
                return None
            off, n = self.offsets[pnm]
            return (px, pdy, self.flat_g[off:off + n],
                    dict(N=N, IH=pdin[0][0], IW=pdin[0][1], IC=pcin, OH=pdout[0][0], OW=pdout[0][1], OC=pcout, KH=pk, KW=pk, stride=pstride,
                         dil=pdil, pad=pdil * (pk // 2), seg2=seg(pdin, pdout), dtype=_cdt(dt)))

        def dgrad(dy, wname, out, conv_cin, conv_cout, k, stride, dil, din, dout, pair=None, **kw):
            # in = dY over the conv's OUTPUT dims (dout), out = dX over its INPUT dims (din)
            # pair = (name, x, dy, cin, cout, k, stride, dil, din, dout): a weight gradient of the same dY, launched in the same grid (bf16 mode)
            seg2 = (dout[1][0], dout[1][1], din[1][0], din[1][1]) if V == 2 else None
            pw = pair_args(pair)
            L.conv_igemm(dy, P["wt"][wname], out, None, N=N, IH=dout[0][0], IW=dout[0][1], IC=conv_cout, OH=din[0][0], OW=din[0][1],
                         OC=conv_cin, KH=k, KW=k, stride=stride, dil=dil, pad=dil * (k // 2), mode=1, seg2=seg2, dtype=_cdt(dt), pair_wgrad=pw, **kw)
            if pair is not None and pw is None:
                wgrad(*pair)

        # ---- per-view adjoints of the x8 upsamples / gather of the stride-8 gradients
        d_cam_low, d_rvd = [], []
        for vw, (g_cam, g_cam_rv, g_fproj, g_rvd) in zip(S["views"], grads):
            h, w, H, W = vw["h"], vw["w"], vw["H"], vw["W"]
            if S["lowres"]:
                d_cam_low.append(g_cam)
                d_rvd.append(g_cam_rv)
                continue
            dc = None
            if g_cam is not None:
                dc = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
                L.resize_planar_bwd(g_cam.contiguous().float(), dc, N * 21, h, w, H, W, True)
            dr = None
            if g_cam_rv is not None:
                dr = torch.empty(N, 21, h, w, device=dev, dtype=torch.float32)
                L.resize_planar_bwd(g_cam_rv.contiguous().float(), dr, N * 21, h, w, H, W, True)
            if g_rvd is not None:
                dr = g_rvd.contiguous().float() if dr is None else dr + g_rvd
            d_cam_low.append(dc)
            d_rvd.append(dr)
        # ---- PCM branch -> f9, f8_3, f8_4.  It ends in WEIGHT gradients only (f8_3 / f8_4 read conv4 / conv5 detached, resnet38_contrast.py:63-64),
        # so nothing of the backbone's backward pass waits for it: it runs on its own stream beside the first blocks (0.5 ms of small kernels + two PCM
        # launches that otherwise sit in front of the head's data gradient), joined before the gradients are consumed (WSEG_PCM_STREAM=0: in line).
        pcm_stream = None
        if any(d is not None for d in d_rvd):
            if os.environ.get("WSEG_PCM_STREAM", "1") != "0":   # (same-box A/B: 35.25 / 35.16 -> 34.94 / 34.94 ms per step; the b7 launches it overlaps slow down by ~1 %)
                pcm_stream = getattr(self, "_pcm_stream", None)
                if pcm_stream is None or pcm_stream.device != dev:
                    pcm_stream = self._pcm_stream = torch.cuda.Stream(dev)
                pcm_stream.wait_stream(main)
            with torch.cuda.stream(pcm_stream if pcm_stream is not None else main):
                DN = torch.empty(M, 32, device=dev, dtype=torch.float32)
                dFh = torch.zeros(M, 192, device=dev, dtype=torch.float32)
                DNb = torch.empty(M, 32, device=dev, dtype=torch.bfloat16) if S["Fb"] is not None else None
                DNl = torch.empty(M, 32, device=dev, dtype=torch.bfloat16) if S["Fb"] is not None else None
                for vw, dr in zip(S["views"], d_rvd):
                    if dr is None:
                        continue
                    off, hw = vw["off"], vw["h"] * vw["w"]
                    if S["Fb"] is not None:
                        L.pcm_backward_bf16(S["Fb"][off:], S["Gb"][off:], S["Gl"][off:], dr.contiguous(), vw["rvd"], vw["den"], DN[off:], DNb[off:],
                                            DNl[off:], dFh[off:], N, hw)
                    else:
                        L.pcm_backward(S["Fh"][off:], S["G"][off:], dr.contiguous(), vw["rvd"], vw["den"], DN[off:], dFh[off:], N, hw)
                dF = E(M, 192)
                L.l2norm_backward(S["Fm"], 192, dFh, S["nrm"], dF, 192, M)
                if trainable("f9"):
                    # feature rows are [f8_3 64 | f8_4 128 | x_s 3 | pad], f9.weight's columns [x_s | f8_3 | f8_4]: the weight-gradient kernel rotates
                    # its dW columns by 3 and accumulates straight into the flat gradient buffer (no staging tensor, no slice adds)
                    off9, n9 = self.offsets["f9"]
                    L.conv_wgrad(S["feat"], dF, self.flat_g[off9:off9 + n9], N=N, IH=hdims[0][0], IW=hdims[0][1], IC=FEAT_LD, OH=hdims[0][0],
                                 OW=hdims[0][1], OC=192, KH=1, KW=1, IC_dw=195, seg2=seg(hdims, hdims), dtype=_cdt(dt), dw_rot=3)
                if trainable("f8_3") or trainable("f8_4"):
                    d_feat = E(M, FEAT_LD)
                    dgrad(dF, "f9", d_feat, FEAT_LD, 192, 1, 1, 1, hdims, hdims, epi=1, mask=S["feat"])
                    wgrad("f8_3", S["conv4"], d_feat, 512, 64, 1, 1, 1, hdims, hdims, ld_dy=FEAT_LD)
                    wgrad("f8_4", S["conv5"], d_feat.view(-1)[64:], 1024, 128, 1, 1, 1, hdims, hdims, ld_dy=FEAT_LD)
        pcm_join[0] = pcm_stream
        # ---- head
        if d_head_rows is None:
            if all(dc is None for dc in d_cam_low) and all(g[2] is None for g in grads):
                if pcm_stream is not None:
                    main.wait_stream(pcm_stream)
                return
            d_head_rows = E(M, HEAD_LD)
            for vw, dc, g in zip(S["views"], d_cam_low, grads):
                off, hw = vw["off"], vw["h"] * vw["w"]
                gf = g[2].contiguous().float() if g[2] is not None else None
                L.head_grad_rows(gf, dc.contiguous() if dc is not None else None, S["head"][off:], d_head_rows[off:], HEAD_LD, N, hw)
        if trainable("fc_proj") or trainable("fc8"):
            off, _ = self.offsets["fc_proj"]
            L.conv_wgrad(S["fea"], d_head_rows, self.flat_g[off:off + 149 * 4096], N=N, IH=hdims[0][0], IW=hdims[0][1], IC=4096,
                         OH=hdims[0][0], OW=hdims[0][1], OC=HEAD_LD, KH=1, KW=1, OC_dw=149, seg2=seg(hdims, hdims), dtype=_cdt(dt))
        s7, _ = P["bn"]["bn7"]
        D = E(M, 4096)
        dgrad(d_head_rows, "head", D, 4096, HEAD_LD, 1, 1, 1, hdims, hdims, epi=1, scale=s7,
              drop=masks["dropout7"] if masks else None, mask=S["fea"])
        # ---- blocks, last to first trainable
        for i in range(len(arch.BLOCKS) - 1, -1, -1):
            b = arch.BLOCKS[i]
            name, kind, cin, mid, cout, stride, fd, d, p = b
            if name in arch.FROZEN_BLOCKS:
                break
            same = arch.block_same_shape(b)
            din, dout = S["dims"][name]
            Mi = rows_of(din)
            Mo = rows_of(dout)
            sv = S[name]
            sa, _ = P["bn"][name + ".bn_branch2a"]
            first_trainable = name == "b3"               # its input comes from the frozen prefix
            if kind == "res":
                s1, _ = P["bn"][name + ".bn_branch2b1"]
                du = E(Mo, mid)
                # (each data gradient takes the weight gradient of the same dY into its launch: wseg_conv_bwd_pair)
                dgrad(D, name + ".conv_branch2b1", du, mid, cout, 3, 1, d, dout, dout, epi=1, scale=s1, mask=sv["v"],
                      pair=(name + ".conv_branch2b1", sv["v"], D, mid, cout, 3, 1, d, dout, dout))
                pair2a = (name + ".conv_branch2a", sv["t"], du, cin, mid, 3, stride, fd, din, dout)
                if not (same and not first_trainable):
                    wgrad(*pair2a)
                pair1 = (name + ".conv_branch1", sv["t"], D, cin, cout, 1, stride, 1, din, dout)
                fused_skip = (name + ".skip_fused") in P["wt"] and not first_trainable
                if not same and not (fused_skip and pair_args(pair1) is not None):
                    wgrad(*pair1)
                if first_trainable:
                    block_done(name)
                    break
                Din = E(Mi, cin)
                if same:
                    dgrad(du, name + ".conv_branch2a", Din, cin, mid, 3, stride, fd, din, dout, epi=1, scale=sa, mask=sv["t"], r_post=D, pair=pair2a)
                elif (name + ".skip_fused") in P["wt"]:        # both data gradients into t: 9 taps of du + one K segment of D
                    seg2 = (dout[1][0], dout[1][1], din[1][0], din[1][1]) if V == 2 else None
                    L.conv_igemm(du, P["wt"][name + ".skip_fused"], Din, None, N=N, IH=dout[0][0], IW=dout[0][1], IC=mid, OH=din[0][0], OW=din[0][1],
                                 OC=cin, KH=3, KW=3, stride=1, dil=fd, pad=fd, mode=1, in2=D, IC2=cout, epi=1, scale=sa, mask=sv["t"], seg2=seg2,
                                 pair_wgrad=pair_args(pair1))        # (+ the skip conv's weight gradient: same D)
                else:
                    tmp = E(Mi, cin)
                    dgrad(D, name + ".conv_branch1", tmp, cin, cout, 1, stride, 1, din, dout)
                    dgrad(du, name + ".conv_branch2a", Din, cin, mid, 3, stride, fd, din, dout, epi=1, scale=sa, mask=sv["t"], r_pre=tmp)
                D = Din
                block_done(name)
            else:
                c4, c2 = cout // 4, cout // 2
                s1, _ = P["bn"][name + ".bn_branch2b1"]
                s2, _ = P["bn"][name + ".bn_branch2b2"]
                d1 = masks[name + ".dropout_2b1"] if masks else None
                d2 = masks[name + ".dropout_2b2"] if masks else None
                du2 = E(Mo, c2)
                dgrad(D, name + ".conv_branch2b2", du2, c2, cout, 1, 1, 1, dout, dout, epi=1, scale=s2, drop=d2, mask=sv["v2"],
                      pair=(name + ".conv_branch2b2", sv["v2"], D, c2, cout, 1, 1, 1, dout, dout))
                du1 = E(Mo, c4)
                dgrad(du2, name + ".conv_branch2b1", du1, c4, c2, 3, 1, d, dout, dout, epi=1, scale=s1, drop=d1, mask=sv["v1"],
                      pair=(name + ".conv_branch2b1", sv["v1"], du2, c4, c2, 3, 1, d, dout, dout))
                wgrad(name + ".conv_branch2a", sv["t"], du1, cin, c4, 1, stride, 1, din, dout)
                Din = E(Mi, cin)
                if (name + ".skip_fused") in P["wt"]:          # both 1x1 data gradients into t as ONE two-source product (+ the skip conv's weight gradient)
                    dgrad(D, name + ".skip_fused", Din, cin, cout, 1, 1, 1, din, dout, epi=1, scale=sa, mask=sv["t"], in2=du1, IC2=c4,
                          pair=(name + ".conv_branch1", sv["t"], D, cin, cout, 1, stride, 1, din, dout))
                else:
                    tmp = E(Mi, cin)
                    dgrad(D, name + ".conv_branch1", tmp, cin, cout, 1, stride, 1, din, dout,
                          pair=(name + ".conv_branch1", sv["t"], D, cin, cout, 1, stride, 1, din, dout))
                    dgrad(du1, name + ".conv_branch2a", Din, cin, c4, 1, stride, 1, din, dout, epi=1, scale=sa, mask=sv["t"], r_pre=tmp)
                D = Din
                block_done(name)
        planes.clear()
        if pcm_join[0] is not None:
            main.wait_stream(pcm_join[0])
        if wstream is not None:
            main.wait_stream(wstream)
            keep.clear()


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, eng, lowres):
        outs, S = eng.run_forward([x], save=True, lowres=lowres)
        ctx.eng, ctx.S = eng, S
        ctx.set_materialize_grads(False)
        if lowres:
            ctx.mark_non_differentiable(outs[0][3])
        return outs[0]

    @staticmethod
    def backward(ctx, g0, g1, g2, g3):
        S = ctx.S
        ctx.eng.run_backward(S, [(g0, g1, g2, None if S["lowres"] else g3)])
        ctx.S = None
        return None, None, None, None
