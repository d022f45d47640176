"""A small reverse-mode tape over the HIP kernels: what `loss.backward()` is for the reference's training step
(`train.py:119-131`), with every forward and backward computation a `cfp_*` kernel.

Values are NHWC row tensors `[rows, C]` on the device (`V.t`), gradients appear in `V.g`.  Every op runs its forward
kernel, pushes a closure on the tape, and `Tape.backward()` runs the closures in reverse.  PyTorch only owns the memory.
There is no CPU path: without the HIP library the first op raises.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import os
import torch

from . import hip, ops, train_ops


class V:
    """A value on the tape: `t` [rows, C] device tensor, `g` its gradient (same shape) once backward reached it."""
    __slots__ = ("t", "g", "needs_grad", "g_owned", "mom")

    def __init__(self, t: torch.Tensor, needs_grad: bool = True):
        self.t, self.g, self.needs_grad, self.g_owned = t, None, needs_grad, True
        self.mom = None        # (partials, nsplit, rows_per_split): channel moments of `t` left by the convolution that produced it

    @property
    def rows(self):
        return self.t.shape[0]

    @property
    def C(self):
        return self.t.shape[1]


class P(V):
    """A parameter: `t` in the layout the kernels want, `g` accumulated in float32 in the same layout;
    `to_torch(g)` converts a gradient back to the reference's state_dict layout."""
    __slots__ = ("name", "to_torch", "gview", "wt", "geom")

    def __init__(self, name: str, t: torch.Tensor, to_torch: Callable[[torch.Tensor], torch.Tensor], gview: Optional[torch.Tensor] = None):
        super().__init__(t, True)
        self.name, self.to_torch = name, to_torch
        self.gview = gview          # preallocated (zeroed) float32 gradient in the kernel layout: kernels write into it directly
        self.wt = None              # conv weights: the flipped copy for the data gradient, when the trainer refreshes all of them in one launch
        self.geom = None            # conv weights whose data gradient is needed: (Cout, KH, KW, Cin), recorded by Tape.conv


def _act(t: torch.Tensor) -> ops.Act:
    assert t.dim() == 2 and t.stride(1) == 1
    if t.stride(0) == t.shape[1]:
        return ops.Act(t, 0, t.shape[1])
    raise ValueError("tape tensors must be contiguous rows")


class Tape:
    def __init__(self, device, dtype=torch.float32, side: Optional[torch.cuda.Stream] = None):
        """`side`: a second stream for the parameter-gradient kernels (weight / bias gradients): nothing later in the backward
        reads them, so they leave the critical path dY -> dX -> ... and run beside it (forked and joined inside a captured step
        as graph edges).  Every tensor they read stays referenced by the tape until `backward()` has joined the streams."""
        self.dev, self.dtype = torch.device(device), dtype
        self.bw: List[Callable[[], None]] = []
        self._const: Dict[Tuple[str, int], torch.Tensor] = {}
        self.side, self._forked = side, False
        self.marks: Dict[str, int] = {}          # name -> tape position: `backward(stop=...)` runs the closures recorded after it
        # weight-gradient slab reductions batched into a few launches at the end of backward() (train_ops.WgradQueue; CFP_WGRAD_DEFER=0: per layer)
        self.wq = train_ops.WgradQueue() if os.environ.get("CFP_WGRAD_DEFER", "1") != "0" else None
        self.conv_stats = os.environ.get("CFP_CONV_STATS", "1") != "0"      # BatchNorm statistics from the producing conv's epilogue (16-bit modes)
        # `defer`: the parameter-gradient closures are queued instead of run (`run_deferred()` runs them, in recording order, wherever
        # the caller wants: the trainer captures them as graphs of their own and replays those on a second stream beside the next
        # segment of the backward -- two cross-stream edges per segment instead of one fork and join per layer)
        self.defer = False
        self.deferred: List[Callable[[], None]] = []

    # ------------------------------------------------------------------ helpers
    def new(self, rows: int, C: int, dtype=None) -> torch.Tensor:
        return torch.empty(rows, C, dtype=dtype or self.dtype, device=self.dev)

    def const(self, kind: str, C: int) -> torch.Tensor:
        k = (kind, C)
        if k not in self._const:
            self._const[k] = (torch.ones if kind == "ones" else torch.zeros)(C, dtype=torch.float32, device=self.dev)
        return self._const[k]

    def acc(self, v: V, g: torch.Tensor, own: bool = True) -> None:
        """v.g += g.  `own`: the caller hands the tensor over (no other reference to it).  A gradient that is only passed
        through (skip connections, identity ops) is shared, not copied: it is copied on the first accumulation into it."""
        if not v.needs_grad:
            return
        if v.g is None:
            v.g, v.g_owned = g, own
        elif v.g_owned:
            train_ops.axpby(v.g, g, 1.0, 1.0, out=v.g)
        else:
            v.g, v.g_owned = train_ops.axpby(v.g, g, 1.0, 1.0), True

    def pgrad(self, p: P, compute) -> None:
        """Parameter gradient: `compute(out, beta)` must produce beta*out + grad (out=None: return a fresh tensor).  With a
        preallocated view the kernel writes straight into the optimizer's flat gradient buffer (no copy afterwards)."""
        if p.gview is None:
            self.acc(p, compute(None, 0.0))
        elif p.g is None:
            compute(p.gview, 0.0)
            p.g = p.gview
        else:
            self.flush_wgrad()          # a postponed reduction may still owe this view its first term
            compute(p.gview, 1.0)

    def _queue(self, out, beta) -> Optional["train_ops.WgradQueue"]:
        """The reduction queue for a weight gradient written (not accumulated) into a preallocated view, else None."""
        return self.wq if (out is not None and beta == 0.0) else None

    def flush_wgrad(self) -> None:
        if self.wq is None or not self.wq.jobs:
            return
        cur = torch.cuda.current_stream(self.dev)
        if self.side is not None and cur != self.side:
            cur.wait_stream(self.side)          # the slab launches run on the side stream
        self.wq.flush()

    @staticmethod
    def _direct(p: P) -> Optional[torch.Tensor]:
        """The parameter's preallocated gradient view if a kernel may write it directly (first and only write so far)."""
        return p.gview if (p.gview is not None and p.g is None) else None

    def _pvec(self, p: P, g: torch.Tensor, direct: Optional[torch.Tensor]) -> None:
        if direct is not None:
            p.g = direct
        else:
            self.pgrad(p, lambda out, b: train_ops.into(g, out, b))

    def off_path(self, fn: Callable[[], None]) -> None:
        """Run `fn` (parameter-gradient kernels of the op whose backward is executing) on the side stream, after everything
        issued so far on the current stream."""
        if self.defer:
            self.deferred.append(fn)
            return
        if self.side is None:
            fn()
            return
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.side):
            fn()
        self._forked = True

    def run_deferred(self) -> List[Callable[[], None]]:
        """Run the queued parameter-gradient closures and the slab reductions they leave behind.  Returns the closures: they hold the
        activations / output gradients the kernels read, and a caller that captures this call into a graph replayed BESIDE later work
        must keep them referenced until everything that could reuse those blocks has been captured."""
        fns, self.deferred = self.deferred, []
        for fn in fns:
            fn()
        self.flush_wgrad()
        return fns

    def mark(self, name: str) -> None:
        """Remember the current tape position (the trainer splits the backward there: everything recorded AFTER the mark is
        differentiated first, e.g. head + decoder before the RGB encoder, so their gradient buckets can be reduced meanwhile)."""
        self.marks[name] = len(self.bw)

    def backward(self, stop: Optional[str] = None) -> None:
        """Run the recorded closures in reverse; with `stop` only those recorded after that mark (call again to finish)."""
        lo = self.marks[stop] if stop is not None else 0
        for f in reversed(self.bw[lo:]):
            f()
        if self._forked:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._forked = False
        self.flush_wgrad()
        self.bw = self.bw[:lo]

    # ------------------------------------------------------------------ dense conv / linear
    def conv(self, x: V, w: P, bias: Optional[P], B, H, W, k, stride, pt, pl, Ho, Wo, stats: bool = False) -> V:
        """w.t [Cout, k*k*Cin]; bias.t [Cout] f32.  `stats`: a batch-statistics BatchNorm follows -- in the 16-bit modes the conv kernel
        leaves the per-row-tile channel moments of its output beside it (V.mom) and bn_act merges them instead of reading the tensor."""
        Cout = w.t.shape[0]
        y = V(self.new(B * Ho * Wo, Cout, x.t.dtype))
        # split-K slabs (few rows, long K: the sr convs) in the 16-bit modes; float32 parity mode keeps its single summation chain
        nws = ops.conv2d_ws_bytes(B * Ho * Wo, Cout, w.t.shape[1], ops.DT[x.t.dtype]) if x.t.dtype != torch.float32 else 0
        ws = torch.empty(nws // 4, dtype=torch.float32, device=self.dev) if nws else None
        if stats and self.conv_stats and x.t.dtype != torch.float32:
            y.mom = ops.conv2d_moments(_act(x.t), w.t, bias.t if bias is not None else None, _act(y.t), B, H, W, k, k, stride, pt, pl, Ho, Wo, ws=None)
        else:
            ops.conv2d(_act(x.t), w.t, None, bias.t if bias is not None else None, _act(y.t), B, H, W, k, k, stride, pt, pl, Ho, Wo, ws=ws)

        def bw():
            g = y.g
            if g is None:
                return
            def param_grads():
                if bias is not None and g.dtype != torch.float32 and bias.gview is not None and w.gview is not None:
                    # 16-bit modes: the bias gradient comes out of the weight-gradient launch (one more matrix-core product per step)
                    bw_, bb_ = (0.0 if w.g is None else 1.0), (0.0 if bias.g is None else 1.0)
                    first = bw_ == 0.0 and bb_ == 0.0
                    if not first:
                        self.flush_wgrad()
                    train_ops.conv2d_wgrad(x.t, g, B, H, W, k, k, stride, pt, pl, Ho, Wo, dw=w.gview, beta=bw_, db=bias.gview, beta_b=bb_,
                                           queue=self.wq if first else None)
                    w.g, bias.g = w.gview, bias.gview
                    return
                if bias is not None:
                    self.pgrad(bias, lambda out, beta: train_ops.colsum(g, out=out, beta=beta))
                self.pgrad(w, lambda out, beta: train_ops.conv2d_wgrad(x.t, g, B, H, W, k, k, stride, pt, pl, Ho, Wo, dw=out, beta=beta,
                                                                       queue=self._queue(out, beta)))
            self.off_path(param_grads)
            if x.needs_grad and k == stride and k > 1 and pt == 0 and pl == 0:
                # non-overlapping patches (the GSA sub-sampling conv, kernel = stride = window size): every input pixel belongs to
                # ONE output pixel and one tap, so dX is a plain GEMM  dY [M_out, Cout] x W [Cout, k*k*Cin]  followed by a
                # depth-to-space row gather -- the generic route (stride-1 conv over the zero-stuffed dY) multiplies 1 - 1/k^2
                # zeros (k = 12: 369 us instead of ~20)
                Cin = x.C
                tmp = self.new(B * Ho * Wo, k * k * Cin, g.dtype)
                ops.conv2d(_act(g), w.t.t().contiguous(), None, None, _act(tmp), 1, 1, B * Ho * Wo, 1, 1, 1, 0, 0, 1, B * Ho * Wo)
                self.acc(x, train_ops.index_rows(tmp.view(B * Ho * Wo * k * k, Cin), self._patch_map(B, H, W, k, Ho, Wo)))
            elif x.needs_grad:
                w.geom = (Cout, k, k, x.C)
                wt = w.wt if w.wt is not None else train_ops.conv2d_weight_flip(w.t, Cout, k, k, x.C)
                if x.g is not None and x.g_owned:
                    # a gradient is already there (q / k / v projections of one input, a block and its skip): add inside the
                    # GEMM's epilogue instead of a separate pass
                    train_ops.conv2d_dgrad(g, wt, B, H, W, x.C, k, k, stride, pt, pl, Ho, Wo, dx=x.g, accumulate=True)
                elif x.g is not None and stride == 1:
                    # shared (not ours to overwrite): read it as the residual of the stride-1 data-gradient conv, write a fresh tensor
                    dx = self.new(B * H * W, x.C, g.dtype)
                    ops.conv2d(_act(g), wt, None, None, _act(dx), B, Ho, Wo, k, k, 1, k - 1 - pt, k - 1 - pl, H, W, residual=_act(x.g))
                    x.g, x.g_owned = dx, True
                else:
                    self.acc(x, train_ops.conv2d_dgrad(g, wt, B, H, W, x.C, k, k, stride, pt, pl, Ho, Wo))
        self.bw.append(bw)
        return y

    _patch_maps: Dict[tuple, torch.Tensor] = {}

    def _patch_map(self, B, H, W, k, Ho, Wo) -> torch.Tensor:
        """Row map of the depth-to-space step above: input pixel (b, y, x) <- row ((b, y // k, x // k), y % k, x % k) of the
        [B*Ho*Wo*k*k, Cin] GEMM result, -1 (zero gradient) for the pixels right / below the last full patch."""
        key = (str(self.dev), B, H, W, k, Ho, Wo)
        if key not in Tape._patch_maps:
            y, x = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            ho, wo = y // k, x // k
            idx = ((ho * Wo + wo) * k + y % k) * k + x % k
            idx = torch.where((ho < Ho) & (wo < Wo), idx, torch.full_like(idx, -1))
            full = torch.cat([torch.where(idx >= 0, idx + b * Ho * Wo * k * k, idx).reshape(-1) for b in range(B)])
            Tape._patch_maps[key] = full.to(torch.int32).to(self.dev)
        return Tape._patch_maps[key]

    def linear(self, x: V, w: P, bias: Optional[P] = None, stats: bool = False) -> V:
        return self.conv(x, w, bias, 1, 1, x.rows, 1, 1, 0, 0, 1, x.rows, stats=stats)

    # ------------------------------------------------------------------ normalisation / activation
    def bn_act(self, x: V, gamma: P, beta: P, running_mean, running_var, eps, momentum, act, residual: Optional[V] = None) -> V:
        """act(BatchNorm(x)) [+ residual]: the skip connection of a residual block rides on the apply pass instead of a separate add."""
        bn = train_ops.BatchNormTrain(x.C, self.dev, eps=eps, momentum=momentum)
        y = V(bn.forward(x.t, gamma.t, beta.t, running_mean, running_var, act, residual.t if residual is not None else None, mom=x.mom))

        def bw():
            if y.g is None:
                return
            if residual is not None:
                self.acc(residual, y.g, own=False)
            go, bo = self._direct(gamma), self._direct(beta)
            dx, dg, db = bn.backward(x.t, y.g, act, dgamma_out=go, dbeta_out=bo)
            self._pvec(gamma, dg, go); self._pvec(beta, db, bo); self.acc(x, dx)
        self.bw.append(bw)
        return y

    def act(self, x: V, kind: int) -> V:
        return self._affine_act(x, kind, None)

    def add_const(self, x: V, shift: torch.Tensor) -> V:
        """y = x + shift[c]  (per-channel float32 constant, not differentiated)."""
        return self._affine_act(x, hip.ACT_NONE, shift)

    def _affine_act(self, x: V, kind: int, shift: Optional[torch.Tensor]) -> V:
        y = V(torch.empty_like(x.t))
        sh = shift if shift is not None else self.const("zeros", x.C)
        hip.call("cfp_scale_shift_act", x.t.data_ptr(), x.t.stride(0), self.const("ones", x.C).data_ptr(), sh.data_ptr(), kind, y.t.data_ptr(),
                 y.t.stride(0), x.rows, x.C, ops.DT[x.t.dtype], hip.current_stream())

        def bw():
            if y.g is None:
                return
            if kind == hip.ACT_NONE:
                self.acc(x, y.g, own=False)
            else:
                self.acc(x, train_ops.act_bwd(x.t, y.g, kind))
        self.bw.append(bw)
        return y

    def layernorm(self, x: V, gamma: P, beta: P, eps, residual: Optional[V] = None) -> V:
        """LayerNorm(x) [+ residual] (the kernel's own residual input: transformer.py:71 `x + message`)."""
        y = V(torch.empty_like(x.t))
        ops.layernorm(_act(x.t), gamma.t, beta.t, eps, _act(y.t), x.rows, residual=_act(residual.t) if residual is not None else None)

        def bw():
            if y.g is None:
                return
            if residual is not None:
                self.acc(residual, y.g, own=False)
            go, bo = self._direct(gamma), self._direct(beta)
            # both parameter gradients written (not accumulated) into their views: their finishing sums wait for the batched reduction
            q = self.wq if (go is not None and bo is not None) else None
            dx, dg, db = train_ops.layernorm_bwd(x.t, y.g, gamma.t, eps, dgamma_out=go, dbeta_out=bo, queue=q)
            self._pvec(gamma, dg, go); self._pvec(beta, db, bo); self.acc(x, dx)
        self.bw.append(bw)
        return y

    # ------------------------------------------------------------------ depthwise
    def dw3x3(self, x: V, w: P, B, H, W, stride, pt, pl, Ho, Wo) -> V:
        """w.t [9, C] in the activation dtype."""
        C = x.C
        y = V(self.new(B * Ho * Wo, C, x.t.dtype))
        ops.dwconv3x3(_act(x.t), w.t, self.const("ones", C), self.const("zeros", C), _act(y.t), B, H, W, stride, pt, pl, Ho, Wo, hip.ACT_NONE)

        def bw():
            if y.g is None:
                return
            g = y.g          # bound now: a deferred closure runs after the tape has moved on
            self.off_path(lambda: self.pgrad(w, lambda out, beta: train_ops.dwconv3x3_wgrad(x.t, g, B, H, W, stride, pt, pl, Ho, Wo, dw=out, beta=beta,
                                                                                            queue=self._queue(out, beta))))
            self.acc(x, train_ops.dwconv3x3_dgrad(y.g, w.t, B, H, W, stride, pt, pl, Ho, Wo))
        self.bw.append(bw)
        return y

    def dwlarge(self, x: V, w: P, bias: P, B, H, W, k) -> V:
        """w.t [C, k, k] f32 in torch order (ky, kx); the forward kernel wants [C][kx][ky]."""
        C = x.C
        y = V(torch.empty_like(x.t))
        # 16-bit storage: the banded-Toeplitz matrix-core kernel of the inference path (4x faster than the VALU form at k = 31);
        # its band table is re-derived from the float32 master weights by one device gather per step
        mfma = x.t.dtype != torch.float32 and k in (7, 15, 31) and C % 16 == 0
        if mfma:
            ops.dwconv_large_mfma(_act(x.t), ops.toeplitz_bands_dev(w.t, x.t.dtype), self.const("ones", C), bias.t, _act(y.t), B, H, W, k, hip.ACT_NONE)
        else:
            wk = w.t.transpose(1, 2).contiguous().reshape(C, k * k)
            ops.dwconv_large(_act(x.t), wk, self.const("ones", C), bias.t, _act(y.t), B, H, W, k, hip.ACT_NONE)

        def bw():
            if y.g is None:
                return
            g = y.g
            def param_grads():
                self.pgrad(bias, lambda out, beta: train_ops.colsum(g, out=out, beta=beta))
                self.pgrad(w, lambda out, beta: train_ops.dwconv_large_wgrad(x.t, g, B, H, W, k, dw=out, beta=beta))
            self.off_path(param_grads)
            dx = torch.empty_like(x.t)                                                   # data gradient = correlation with the flipped kernel
            if mfma:
                ops.dwconv_large_mfma(_act(y.g), ops.toeplitz_bands_dev(w.t, x.t.dtype, flip=True), self.const("ones", C), self.const("zeros", C),
                                      _act(dx), B, H, W, k, hip.ACT_NONE)
            else:
                wf = w.t.flip(1, 2).transpose(1, 2).contiguous().reshape(C, k * k)
                ops.dwconv_large(_act(y.g), wf, self.const("ones", C), self.const("zeros", C), _act(dx), B, H, W, k, hip.ACT_NONE)
            self.acc(x, dx)
        self.bw.append(bw)
        return y

    # ------------------------------------------------------------------ structure
    def add(self, a: V, b: V) -> V:
        y = V(train_ops.axpby(a.t, b.t, 1.0, 1.0))

        def bw():
            if y.g is None:
                return
            self.acc(a, y.g, own=False); self.acc(b, y.g, own=False)
        self.bw.append(bw)
        return y

    def concat(self, a: V, b: V) -> V:
        Ca, Cb = a.C, b.C
        y = V(self.new(a.rows, Ca + Cb, a.t.dtype))
        full = ops.Act(y.t, 0, Ca + Cb)
        ops.copy_rows2(_act(a.t), full.slice(0, Ca), _act(b.t), full.slice(Ca, Cb), a.rows)

        def bw():
            if y.g is None:
                return
            gf = ops.Act(y.g, 0, Ca + Cb)
            if a.needs_grad and b.needs_grad:       # both halves in one launch
                ga, gb = self.new(a.rows, Ca, y.g.dtype), self.new(b.rows, Cb, y.g.dtype)
                ops.copy_rows2(gf.slice(0, Ca), _act(ga), gf.slice(Ca, Cb), _act(gb), a.rows)
                self.acc(a, ga); self.acc(b, gb)
                return
            for v, c0, c in ((a, 0, Ca), (b, Ca, Cb)):
                if v.needs_grad:
                    g = self.new(v.rows, c, y.g.dtype)
                    ops.copy_rows(gf.slice(c0, c), _act(g), v.rows)
                    self.acc(v, g)
        self.bw.append(bw)
        return y

    def gather(self, x: V, idx: torch.Tensor, inv: torch.Tensor) -> V:
        """y[i] = x[idx[i]] (zero rows where idx < 0); idx injective, inv its inverse (train_ops.inverse_index)."""
        y = V(train_ops.index_rows(x.t, idx))

        def bw():
            if y.g is not None:
                self.acc(x, train_ops.index_rows(y.g, inv))
        self.bw.append(bw)
        return y

    def resize(self, x: V, B, Hs, Ws, Hd, Wd) -> V:
        y = V(self.new(B * Hd * Wd, x.C, x.t.dtype))
        ops.resize_bilinear(_act(x.t), Hs, Ws, (0, 0, Hs, Ws), _act(y.t), Hd, Wd, (0, 0, Hd, Wd), B)

        def bw():
            if y.g is not None:
                self.acc(x, train_ops.resize_bilinear_bwd(y.g, B, Hs, Ws, Hd, Wd))
        self.bw.append(bw)
        return y

    def add_table(self, x: V, table: P, B, H, W, Ht, Wt, off) -> V:
        """y[b, y, x] = x[b, y, x] + table[(oy + y) * Wt + ox + x]  (learned positional encodings).  `off` = (oy, ox) on the
        host, or an int32[2] DEVICE tensor (the window can then change between replays of a captured step)."""
        y = V(torch.empty_like(x.t))
        dev_off = torch.is_tensor(off)
        if dev_off:
            hip.call("cfp_add_rowtable_dev", x.t.data_ptr(), x.t.stride(0), table.t.data_ptr(), y.t.data_ptr(), y.t.stride(0), x.rows, x.C, H, W,
                     Ht, Wt, off.data_ptr(), ops.DT[x.t.dtype], hip.current_stream())
        else:
            ops.add_rowtable(_act(x.t), table.t, _act(y.t), x.rows, H, W, Wt, off[0], off[1])

        def bw():
            if y.g is None:
                return
            def table_grad(out, beta):
                dt = torch.zeros_like(table.t) if out is None else out
                if out is not None and beta == 0.0:
                    out.zero_()                      # persistent view: last step's window (the offsets move) must not survive
                if dev_off:
                    hip.call("cfp_rowtable_grad_dev", y.g.data_ptr(), y.g.stride(0), dt.data_ptr(), B, H, W, x.C, Ht, Wt, off.data_ptr(), beta,
                             ops.DT[y.g.dtype], hip.current_stream())
                else:
                    train_ops.rowtable_grad(y.g, dt, B, H, W, Wt, off[0], off[1], beta=beta)
                return dt
            self.pgrad(table, table_grad)
            self.acc(x, y.g, own=False)
        self.bw.append(bw)
        return y

    # ------------------------------------------------------------------ attention / squeeze-excite / head
    def attention(self, q: V, k: V, v: V, N, L, S, heads, d) -> V:
        out, state = train_ops.linattn_fwd(q.t, k.t, v.t, N, L, S, heads, d)
        y = V(out)

        def bw():
            if y.g is None:
                return
            dq, dk, dv = train_ops.linattn_bwd(q.t, k.t, v.t, y.g, state, N, L, S, heads, d)
            self.acc(q, dq); self.acc(k, dk); self.acc(v, dv)
        self.bw.append(bw)
        return y

    def channel_mean(self, x: V, B, HW) -> V:
        """[B*HW, C] -> [B, C] float32."""
        ns = max(1, min(64, HW // 16, -(-1024 // B)))                # two-level sum: [B, ns, C] partials (~1000 workgroups), then over ns
        part = torch.empty(B * ns, x.C, dtype=torch.float32, device=self.dev)
        ops.channel_sum(_act(x.t), part, B, HW, ns)
        if ns > 1:
            tot = torch.empty(B, x.C, dtype=torch.float32, device=self.dev)
            ops.channel_sum(ops.Act(part, 0, x.C), tot, B, ns, 1)
            part = tot
        y = V(train_ops.axpby(part, None, 1.0 / HW, 0.0))

        def bw():
            if y.g is None:
                return
            add = train_ops.axpby(y.g, None, 1.0 / HW, 0.0)
            self.acc(x, train_ops.bcast_fma(x.t, self.new(B, x.C, torch.float32).zero_(), add, B, HW))
        self.bw.append(bw)
        return y

    def mul_bcast(self, x: V, gate: V, B, HW) -> V:
        """y[b, hw, c] = x[b, hw, c] * gate[b, c]  (gate float32)."""
        y = V(train_ops.bcast_fma(x.t, gate.t, None, B, HW))

        def bw():
            if y.g is None:
                return
            if gate.needs_grad:
                self.acc(gate, train_ops.channel_dot(x.t, y.g, B, HW))
            self.acc(x, train_ops.bcast_fma(y.g, gate.t, None, B, HW))
        self.bw.append(bw)
        return y

    def se_block(self, x: V, w1: P, b1: P, w2: P, b2: P, B, HW) -> V:
        """y = x * sigmoid(W2 silu(W1 mean_HW(x) + b1) + b2)  (timm SqueezeExcite) as 3 launches forward and 4 backward: channel sums,
        the gate kernel (cfp_se_train_fwd), the broadcast multiply; backward: channel dot, the two kernels of cfp_se_train_bwd, one
        broadcast multiply-add.  Same arithmetic as channel_mean -> linear -> act -> linear -> act -> mul_bcast on this tape."""
        assert w1.t.dtype == torch.float32 and w2.t.dtype == torch.float32
        ns = max(1, min(64, HW // 16, -(-1024 // B)))
        part = torch.empty(B * ns, x.C, dtype=torch.float32, device=self.dev)
        ops.channel_sum(_act(x.t), part, B, HW, ns)
        mean, z1, gate = train_ops.se_train_fwd(part, ns, 1.0 / HW, w1.t, b1.t, w2.t, b2.t, B)
        y = V(train_ops.bcast_fma(x.t, gate, None, B, HW))

        def bw():
            if y.g is None:
                return
            dgate = train_ops.channel_dot(x.t, y.g, B, HW)
            ps = (w1, b1, w2, b2)
            direct = [self._direct(p) for p in ps]
            if all(d is not None for d in direct):
                res = train_ops.se_train_bwd(dgate, gate, z1, mean, w1.t, w2.t, 1.0 / HW,
                                             outs=(direct[0].view(w1.t.shape), direct[1][:b1.t.numel()], direct[2].view(w2.t.shape), direct[3][:b2.t.numel()]))
                for p, d in zip(ps, direct):
                    p.g = d
            else:
                res = train_ops.se_train_bwd(dgate, gate, z1, mean, w1.t, w2.t, 1.0 / HW)
                for p, g in zip(ps, res[:4]):
                    self.pgrad(p, lambda out, beta, g=g: train_ops.into(g, out, beta))
            self.acc(x, train_ops.bcast_fma(y.g, gate, res[4], B, HW))
        self.bw.append(bw)
        return y

    def row_normalize(self, x: V) -> V:
        y = V(torch.empty_like(x.t))
        hip.call("cfp_row_normalize", x.t.data_ptr(), None, y.t.data_ptr(), x.rows, x.C, hip.current_stream())

        def bw():
            if y.g is None:
                return
            dx = torch.empty_like(x.t)
            hip.call("cfp_row_normalize", x.t.data_ptr(), y.g.data_ptr(), dx.data_ptr(), x.rows, x.C, hip.current_stream())
            self.acc(x, dx)
        self.bw.append(bw)
        return y

    def bin_centers(self, wn: V, min_val, max_val):
        edges, centers = train_ops.bin_centers(wn.t, min_val, max_val)
        y = V(centers)

        def bw():
            if y.g is not None:
                self.acc(wn, train_ops.bin_centers_bwd(y.g, min_val, max_val))
        self.bw.append(bw)
        return edges, y

    def softmax_expect(self, logits: V, centers: V, B, HW) -> V:
        """-> pred as [B*HW, 1]-shaped value (float32 vector inside)."""
        pred = train_ops.softmax_expect(logits.t, centers.t, B, HW)
        y = V(pred.reshape(-1, 1))

        def bw():
            if y.g is None:
                return
            dl, dc = train_ops.softmax_expect(logits.t, centers.t, B, HW, dpred=y.g.reshape(-1).contiguous())
            self.acc(logits, dl); self.acc(centers, dc)
        self.bw.append(bw)
        return y
