"""Training-step pieces: the SILog loss (forward + backward HIP kernels), the OneCycle
schedule, a flat-buffer AdamW, the gradient all-reduce plan for data-parallel training, and thin wrappers over the backward
kernels of the network ops (`autograd_hip.Tape` chains them; `train_model.TrainNet` is the model; `trainer.Trainer` the step).

torch is used for device memory, the stream and `torch.distributed` (RCCL) only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import ctypes
import torch

from . import hip


class SILogLoss:
    """`/root/reference/src/loss.py:4-19`: call with (pred [B,1,Hp,Wp] f32, target [B,1,Ht,Wt] f32, mask bool|None,
    interpolate).  `forward` returns the device scalar; `backward(grad_loss)` returns d loss / d pred."""

    name = "SILog"

    def __init__(self):
        self._ws: Optional[torch.Tensor] = None
        self._stats: Optional[torch.Tensor] = None
        self._shape = None

    def forward(self, pred: torch.Tensor, target: torch.Tensor, mask: Optional[torch.Tensor] = None, interpolate: bool = True):
        assert pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32
        B, _, Hp, Wp = pred.shape
        Bt, _, Ht, Wt = target.shape
        assert B == Bt
        pred, target = pred.contiguous(), target.contiguous()
        m8 = None
        if mask is not None:
            m8 = mask.to(device=pred.device).reshape(B, 1, Ht, Wt).to(torch.uint8).contiguous()
        need = int(hip.load().cfp_silog_ws_bytes(B, Ht, Wt))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=pred.device)
        self._stats = torch.empty(4, dtype=torch.float32, device=pred.device)
        hip.call("cfp_silog_loss_fwd", pred.data_ptr(), Hp, Wp, target.data_ptr(), hip.ptr(m8), Ht, Wt, B, int(interpolate),
                 self._ws.data_ptr(), self._ws.numel(), self._stats.data_ptr(), hip.current_stream())
        self._shape = (B, Hp, Wp, Ht, Wt, bool(interpolate))
        return self._stats[0]

    __call__ = forward

    def sync_moments(self, dist, group=None) -> torch.Tensor:
        """Make the loss the GLOBAL-batch SILog over all data-parallel ranks: all-reduce the three moments (sum g, sum g^2, n) and
        recompute mean / variance / loss from the totals -- what the reference gets from nn.DataParallel, which gathers the predictions
        and evaluates ONE loss on the whole batch (train.py:119-123); per-rank losses (the default of this trainer) are a slightly
        different objective.  Call between `forward` and `backward`, then pass `backward(grad_loss=world_size)`: the gradient averaging
        over the ranks divides by it again.  Returns the global loss (device scalar).  Three tiny torch kernels + one collective."""
        st = self._stats.double()
        mean, n, dg = st[1], st[2], st[3]
        # a rank whose shard has no valid pixel (n = 0; n = 1: no variance term) contributes ZERO moments instead of the NaN its local
        # mean / variance are -- the all-reduce would otherwise spread that NaN into every rank's loss and gradients
        zero = torch.zeros((), dtype=torch.float64, device=st.device)
        s = torch.where(n > 0, mean * n, zero)
        s2 = torch.where(n > 1, (dg - 0.15 * mean * mean) * (n - 1.0) + s * s / n.clamp(min=1.0), torch.where(n > 0, s * s, zero))   # var_unbiased = (s2 - s^2 / n) / (n - 1)
        m = torch.stack([s, s2, torch.where(n > 0, n, zero)])
        if dist.get_backend(group) == "gloo":                                  # host-staged, like the gradient buckets under gloo
            h = m.cpu()
            dist.all_reduce(h, group=group)
            m = h.to(self._stats.device)
        else:
            dist.all_reduce(m, group=group)
        S, S2, N = m[0], m[1], m[2]
        Nc = N.clamp(min=2.0)                                                  # N <= 1 over ALL ranks: no variance; loss 10 * sqrt(0.15) |mean|, finite
        gmean = S / N.clamp(min=1.0)
        gdg = torch.where(N > 1, (S2 - S * S / Nc) / (Nc - 1.0), torch.zeros_like(S)) + 0.15 * gmean * gmean
        self._stats.copy_(torch.stack([10.0 * torch.sqrt(gdg), gmean, N, gdg]).float())
        return self._stats[0]

    def backward(self, grad_loss: float = 1.0) -> torch.Tensor:
        B, Hp, Wp, Ht, Wt, interp = self._shape
        grad = torch.empty(B, 1, Hp, Wp, dtype=torch.float32, device=self._stats.device)
        hip.call("cfp_silog_loss_bwd", self._ws.data_ptr(), self._stats.data_ptr(), float(grad_loss), Hp, Wp, Ht, Wt, B, int(interp),
                 grad.data_ptr(), hip.current_stream())
        return grad


class OneCycle:
    """torch.optim.lr_scheduler.OneCycleLR as train.py:90-94 configures it: cosine annealing, pct_start 0.3,
    two phases, momentum (AdamW beta1) cycled inversely between base_momentum and max_momentum.  Note the reference
    passes the scalar `lr` as max_lr, so BOTH parameter groups follow the same schedule (the lr/10 of the encoder
    group in train.py:79 is overwritten by the scheduler's initial_lr = max_lr / div_factor)."""

    def __init__(self, max_lr: float, total_steps: int, div_factor: float = 25.0, final_div_factor: float = 100.0,
                 pct_start: float = 0.3, base_momentum: float = 0.85, max_momentum: float = 0.95):
        assert total_steps > 0
        self.total_steps = total_steps
        self.max_lr = max_lr
        self.initial_lr = max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.base_m, self.max_m = base_momentum, max_momentum
        self.up_end = float(pct_start * total_steps) - 1.0
        self.down_end = float(total_steps) - 1.0

    @staticmethod
    def _cos(start: float, end: float, pct: float) -> float:
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1.0)

    def at(self, step_num: int) -> Tuple[float, float]:
        """(lr, beta1) in effect for optimizer step number `step_num` (0-based: the values OneCycleLR has set after
        `step_num` calls of scheduler.step())."""
        if step_num > self.total_steps:
            raise ValueError(f"step {step_num} beyond total_steps {self.total_steps}")
        if step_num <= self.up_end:
            pct = step_num / self.up_end if self.up_end > 0 else 1.0
            return self._cos(self.initial_lr, self.max_lr, pct), self._cos(self.max_m, self.base_m, pct)
        pct = (step_num - self.up_end) / (self.down_end - self.up_end)
        return self._cos(self.max_lr, self.min_lr, pct), self._cos(self.base_m, self.max_m, pct)


@dataclass
class Segment:
    name: str
    start: int
    numel: int
    shape: Tuple[int, ...]
    group: int          # 0 = "1x" group (encoder [+ hist encoder]), 1 = "10x" group, 2 = dead (never gets a gradient)


class FlatParams:
    """All parameters in ONE flat f32 device buffer, ordered [group 0 | group 1 | dead], every tensor start aligned to
    4 elements, so that (i) AdamW is one 16-byte-vector sweep per group, (ii) the data-parallel gradient all-reduce is
    a few large RCCL calls over contiguous ranges, and (iii) the 48 tensors the forward never touches
    (transformer.py:183-194, convnext.py:38) sit in a tail that is neither reduced nor stepped (their .grad is None in
    the reference, so torch.optim.AdamW skips them too)."""

    def __init__(self, named_shapes: Sequence[Tuple[str, Tuple[int, ...]]], group_of, device="cpu", align: int = 4):
        self.segments: List[Segment] = []
        off = 0
        self.group_range: Dict[int, Tuple[int, int]] = {}
        for grp in (0, 1, 2):
            g0 = off
            for name, shape in named_shapes:
                if group_of(name) != grp:
                    continue
                n = int(math.prod(shape)) if len(shape) else 1
                self.segments.append(Segment(name, off, n, tuple(shape), grp))
                off += (n + align - 1) // align * align
            self.group_range[grp] = (g0, off)
        self.total = off
        self.live = self.group_range[1][1]          # [0, live) gets gradients
        self.device = torch.device(device)
        self.param = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self._by_name = {s.name: s for s in self.segments}

    def view(self, name: str, which: str = "param") -> torch.Tensor:
        s = self._by_name[name]
        return getattr(self, which)[s.start:s.start + s.numel].view(s.shape)

    def load(self, state_dict: Dict[str, torch.Tensor]):
        for s in self.segments:
            self.view(s.name).copy_(state_dict[s.name].to(torch.float32))

    def buckets(self, bucket_elems: int, lo: int = 0, hi: Optional[int] = None) -> List[Tuple[int, int]]:
        """Contiguous [start, end) ranges covering the live gradients in [lo, hi) (default: all of them), walked from the END
        of the range to its beginning: the flat order is encoder -> ... -> head, backward produces gradients head first, so
        the first bucket can be reduced while the encoder's backward is still running."""
        out, end = [], self.live if hi is None else min(hi, self.live)
        while end > lo:
            start = max(lo, end - bucket_elems)
            out.append((start, end))
            end = start
        return out


def _is_gloo(dist) -> bool:
    try:
        return dist.get_backend() == "gloo"
    except Exception:
        return False


def allreduce_range(flat: FlatParams, dist, world: int, lo: int, hi: int, bucket_elems: int = 8 * 1024 * 1024, async_op: bool = False):
    """Average flat.grad[lo:hi) over the data-parallel ranks, one all-reduce per bucket on the CURRENT stream's timeline
    (RCCL over xGMI with the "nccl" backend).  With the "gloo" backend (CPU tests, and the single-GPU 2-rank test) the bucket
    is staged through host memory: gloo reduces host buffers.  `async_op` (RCCL only): returns [(work, chunk)] for `finish_allreduce`."""
    if dist is None:
        return []
    handles = []
    gloo = _is_gloo(dist)
    for (a, b) in flat.buckets(bucket_elems, lo, hi):
        chunk = flat.grad[a:b]
        if gloo and chunk.is_cuda:
            host = chunk.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            chunk.copy_(host.div_(world))
            continue
        h = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=async_op)
        handles.append((h, chunk))
    if not async_op:
        for _, chunk in handles:
            chunk.div_(world)
        return []
    return handles


def finish_allreduce(handles, world: int) -> None:
    """Wait (stream-wise: the current stream waits for the collective's stream) and scale."""
    for h, chunk in handles:
        h.wait()
        chunk.div_(world)


def allreduce_gradients(flat: FlatParams, dist, world: int, bucket_elems: int = 8 * 1024 * 1024, async_op: bool = False):
    """Average the live gradients over the data-parallel ranks: one all-reduce (RCCL over xGMI with the "nccl"
    backend; gloo in the CPU tests) per bucket of the flat buffer, no per-tensor calls, dead tail excluded.
    21.4 M live f32 gradients = 86 MB = 3 buckets of 32 MB."""
    if dist is None or world <= 1:
        return []
    return allreduce_range(flat, dist, world, 0, flat.live, bucket_elems, async_op)


class FlatAdamW:
    """torch.optim.AdamW(params=[1x group, 10x group], weight_decay=wd) on a FlatParams buffer: per step one
    `cfp_adamw_step` launch per group (+ optionally the clip-factor kernels), lr / beta1 from OneCycle."""

    def __init__(self, flat: FlatParams, schedule: OneCycle, weight_decay: float = 0.1, beta2: float = 0.999, eps: float = 1e-8,
                 clip_grad_norm: Optional[float] = None, overflow_guard: bool = False):
        assert flat.param.is_cuda, "FlatAdamW runs the HIP kernel: parameters must be on the GPU"
        self.flat, self.sched = flat, schedule
        self.wd, self.beta2, self.eps, self.clip = weight_decay, beta2, eps, clip_grad_norm
        # 16-bit training: the gradient norm is always formed and a step whose norm is not finite is skipped on the device
        # (cfp_grad_clip_factor / cfp_adamw_step); `skipped_steps()` reads the counter
        self.guard = bool(overflow_guard)
        self.m = torch.zeros_like(flat.param)
        self.v = torch.zeros_like(flat.param)
        self.step_count = 0
        self._clip_ws = torch.empty(int(hip.load().cfp_grad_clip_ws_bytes()) // 8, dtype=torch.float64, device=flat.param.device)
        self._clip_out = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=flat.param.device)

    def step(self):
        lr, beta1 = self.sched.at(self.step_count)
        self.step_count += 1
        f = self.flat
        scale_ptr = 0
        if self.clip is not None or self.guard:
            hip.call("cfp_grad_clip_factor", f.grad.data_ptr(), f.live, float(self.clip) if self.clip is not None else 3.0e38, self._clip_ws.data_ptr(),
                     self._clip_ws.numel() * 8, self._clip_out.data_ptr(), hip.current_stream())
            scale_ptr = self._clip_out.data_ptr()
        for grp in (0, 1):
            a, b = f.group_range[grp]
            if b <= a:
                continue
            es = 4
            hip.call("cfp_adamw_step", f.param.data_ptr() + a * es, f.grad.data_ptr() + a * es, self.m.data_ptr() + a * es,
                     self.v.data_ptr() + a * es, b - a, float(lr), float(beta1), float(self.beta2), float(self.eps), float(self.wd),
                     self.step_count, scale_ptr, hip.current_stream())
        return lr, beta1

    def skipped_steps(self) -> int:
        """Steps the overflow guard has skipped so far (device counter; synchronises)."""
        return int(self._clip_out[2].item())


def lr_group_of(hist_encoder_10x: bool):
    """Parameter-group rule of deltar.py:68-82 plus the dead-tensor rule of SURVEY 2.2."""
    from . import spec

    def group(name: str) -> int:
        if spec.is_dead_param(name):
            return 2
        top = name.split(".")[0]
        if top == "img_encoder" or (top == "hist_encoder" and not hist_encoder_10x):
            return 0
        return 1
    return group


# ------------------------------------------------------------------------------------------------------------------
# Backward building blocks of the conv / BatchNorm layers (csrc/conv_bwd.hip, csrc/bn_train.hip).  Thin wrappers over
# the C ABI on NHWC row tensors [rows, C]; `autograd_hip.Tape` chains them into the backward of the whole network.
# ------------------------------------------------------------------------------------------------------------------
class WgradQueue:
    """Weight-gradient reductions put off until `flush()`: cfp_conv2d_wgrad_deferred leaves its slabs in `ws` and a job record here,
    cfp_wgrad_reduce_jobs finishes up to 48 layers per launch (bit-identical to the per-layer reduction).  The queue keeps the slab
    tensors alive until they have been reduced."""

    def __init__(self):
        self.jobs: list = []
        self.keep: list = []

    def flush(self) -> None:
        if self.jobs:
            from . import hip
            arr = (hip.WgradJob * len(self.jobs))(*self.jobs)
            hip.call("cfp_wgrad_reduce_jobs", ctypes.addressof(arr), len(self.jobs), hip.current_stream())
            # the reduction may run on another stream than the one the slabs were allocated on (side-stream weight gradients flushed on
            # main at the end of the backward and the reverse): tell the caching allocator, or it may hand a block back to its own stream
            # while this launch is still pending
            cur = torch.cuda.current_stream()
            for item in self.keep:
                for t in (item if isinstance(item, (tuple, list)) else (item,)):
                    if isinstance(t, torch.Tensor) and t.is_cuda:
                        t.record_stream(cur)
        self.jobs, self.keep = [], []


def conv2d_wgrad(x2d: torch.Tensor, dy2d: torch.Tensor, B, H, W, KH, KW, stride, pad_t, pad_l, Ho, Wo, dw: Optional[torch.Tensor] = None,
                 beta: float = 0.0, db: Optional[torch.Tensor] = None, beta_b: float = 0.0, queue: Optional[WgradQueue] = None) -> torch.Tensor:
    """x2d [B*H*W, Cin], dy2d [B*Ho*Wo, Cout] (same dtype: f32 / bf16 / f16) -> dw [Cout, KH*KW*Cin] f32 (= beta*dw + grad).
    `db` (16-bit dtypes): float32 [>= Cout], receives beta_b*db + the bias gradient (column sums of dy) from the same launch.
    `queue`: leave the reduction of the split slabs to `queue.flush()` (dw / db hold the result only after it)."""
    from . import hip, ops
    Cin, Cout = x2d.shape[1], dy2d.shape[1]
    K, M = KH * KW * Cin, B * Ho * Wo
    if dw is None:
        dw = torch.empty(Cout, K, dtype=torch.float32, device=x2d.device)
        beta = 0.0
    nbytes = hip.load().cfp_conv2d_wgrad_ws_bytes(Cout, K, M)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x2d.device)
    if db is not None:
        assert x2d.dtype != torch.float32 and db.dtype == torch.float32 and db.numel() >= Cout
    if queue is not None:
        job = hip.WgradJob()
        hip.call("cfp_conv2d_wgrad_deferred", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), hip.ptr(db), B, H, W,
                 Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, beta, beta_b, ops.DT[x2d.dtype], ws.data_ptr(), nbytes, ctypes.addressof(job),
                 hip.current_stream())
        if job.nsplit > 0:
            queue.jobs.append(job)
            queue.keep.append((ws, dw, db))
        return dw
    if db is not None:
        hip.call("cfp_conv2d_wgrad_bias", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), db.data_ptr(), B, H, W,
                 Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, beta, beta_b, ops.DT[x2d.dtype], ws.data_ptr(), nbytes, hip.current_stream())
        return dw
    hip.call("cfp_conv2d_wgrad", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), B, H, W, Cin, Cout, KH, KW,
             stride, pad_t, pad_l, Ho, Wo, beta, ops.DT[x2d.dtype], ws.data_ptr(), nbytes, hip.current_stream())
    return dw


def conv2d_weight_flip(w2d: torch.Tensor, Cout, KH, KW, Cin) -> torch.Tensor:
    """w [Cout, KH*KW*Cin] -> wt [Cin, KH*KW*Cout] with both kernel axes reversed."""
    from . import hip, ops
    wt = torch.empty(Cin, KH * KW * Cout, dtype=w2d.dtype, device=w2d.device)
    hip.call("cfp_conv2d_weight_flip", w2d.data_ptr(), wt.data_ptr(), Cout, KH, KW, Cin, ops.DT[w2d.dtype], hip.current_stream())
    return wt


def conv2d_dgrad(dy2d: torch.Tensor, wt: torch.Tensor, B, H, W, Cin, KH, KW, stride, pad_t, pad_l, Ho, Wo, dx: Optional[torch.Tensor] = None,
                 accumulate: bool = False) -> torch.Tensor:
    from . import hip, ops
    Cout = dy2d.shape[1]
    if dx is None:
        dx = torch.empty(B * H * W, Cin, dtype=dy2d.dtype, device=dy2d.device)
        accumulate = False
    hip.call("cfp_conv2d_dgrad", dy2d.data_ptr(), dy2d.stride(0), wt.data_ptr(), dx.data_ptr(), dx.stride(0), B, H, W, Cin, Cout, KH, KW,
             stride, pad_t, pad_l, Ho, Wo, int(accumulate), ops.DT[dy2d.dtype], None, 0, hip.current_stream())
    return dx


_BN_WS = {}


def bn_workspace(C: int, device, stream: int):
    """The reduction workspace of cfp_bn_train_stats / cfp_bn_train_bwd for a channel count: ONE per (device, stream, C), shared
    by every layer of that width -- the calls of a stream run one after the other and nothing in it outlives a call.  (Round 2 gave
    every layer call a workspace of its own: 32 KB x C each, 122 of them alive over a training step.)"""
    from . import hip
    key = (str(device), int(stream), C)
    ws = _BN_WS.get(key)
    if ws is None:
        ws = _BN_WS[key] = torch.empty(hip.load().cfp_bn_ws_bytes(C) // 4, dtype=torch.float32, device=device)
    return ws


class BatchNormTrain:
    """act(BatchNorm(x)) with batch statistics on [rows, C] NHWC rows: forward keeps what backward needs."""

    def __init__(self, C: int, device, eps: float = 1e-5, momentum: float = 0.1):
        from . import hip
        self.C, self.eps, self.momentum = C, eps, momentum
        f = lambda: torch.empty(C, dtype=torch.float32, device=device)
        self.mean, self.var, self.invstd, self.scale, self.shift = f(), f(), f(), f(), f()
        self.nbytes = hip.load().cfp_bn_ws_bytes(C)
        self.device = device

    @property
    def ws(self):
        from . import hip
        return bn_workspace(self.C, self.device, hip.current_stream())

    def forward(self, x2d, gamma, beta, running_mean, running_var, act: int, residual: Optional[torch.Tensor] = None, mom=None):
        """`residual` [rows, C]: added after the activation in the apply pass (the block's skip connection).
        `mom` = (partials, nsplit, rows_per_split) from the producing convolution (ops.conv2d_moments): the statistics are then merged
        from them in one small launch instead of a pass over x2d + that launch."""
        from . import hip, ops
        rows = x2d.shape[0]
        if mom is not None and mom[1] > 0:
            hip.call("cfp_bn_train_stats_partials", mom[0].data_ptr(), mom[1], rows, mom[2], self.C, hip.ptr(gamma), hip.ptr(beta), self.eps,
                     self.momentum, hip.ptr(running_mean), hip.ptr(running_var), self.mean.data_ptr(), self.var.data_ptr(),
                     self.invstd.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), hip.current_stream())
        else:
            hip.call("cfp_bn_train_stats", x2d.data_ptr(), x2d.stride(0), rows, self.C, ops.DT[x2d.dtype], hip.ptr(gamma), hip.ptr(beta), self.eps,
                     self.momentum, hip.ptr(running_mean), hip.ptr(running_var), self.mean.data_ptr(), self.var.data_ptr(), self.invstd.data_ptr(),
                     self.scale.data_ptr(), self.shift.data_ptr(), self.ws.data_ptr(), self.nbytes, hip.current_stream())
        y = torch.empty_like(x2d)
        hip.call("cfp_scale_shift_act_res", x2d.data_ptr(), x2d.stride(0), self.scale.data_ptr(), self.shift.data_ptr(), act,
                 hip.ptr(residual), residual.stride(0) if residual is not None else 0, y.data_ptr(), y.stride(0), rows, self.C,
                 ops.DT[x2d.dtype], hip.current_stream())
        return y

    def backward(self, x2d, dy2d, act: int, dgamma_out: Optional[torch.Tensor] = None, dbeta_out: Optional[torch.Tensor] = None):
        """-> (dx, dgamma, dbeta); the two parameter gradients are written into the given float32 tensors when provided."""
        from . import hip, ops
        rows = x2d.shape[0]
        dgamma = dgamma_out if dgamma_out is not None else torch.empty(self.C, dtype=torch.float32, device=x2d.device)
        dbeta = dbeta_out if dbeta_out is not None else torch.empty(self.C, dtype=torch.float32, device=x2d.device)
        dx = torch.empty_like(x2d)
        hip.call("cfp_bn_train_bwd", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), rows, self.C, ops.DT[x2d.dtype],
                 self.mean.data_ptr(), self.invstd.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), act, dgamma.data_ptr(),
                 dbeta.data_ptr(), dx.data_ptr(), dx.stride(0), self.ws.data_ptr(), self.nbytes, hip.current_stream())
        return dx, dgamma, dbeta


def into(src: torch.Tensor, out: Optional[torch.Tensor], beta: float) -> torch.Tensor:
    """out = beta*out + src over float32 vectors (out=None: src itself).  `out` may be longer than `src` (zero-padded layouts)."""
    if out is None:
        return src
    n = src.numel()
    dst = out.reshape(-1)[:n].reshape(1, n)
    axpby(src.reshape(1, n), dst if beta != 0.0 else None, 1.0, beta, out=dst)
    return out


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None, beta: float = 0.0) -> torch.Tensor:
    """Sum over rows per channel (f32): the bias gradient (out = beta*out + sum when `out` is given)."""
    from . import hip, ops
    C = x2d.shape[1]
    nbytes = hip.load().cfp_bn_ws_bytes(C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    direct = out is not None and beta == 0.0 and out.numel() == C and out.is_contiguous()
    res = out if direct else torch.empty(C, dtype=torch.float32, device=x2d.device)
    hip.call("cfp_colsum", x2d.data_ptr(), x2d.stride(0), x2d.shape[0], C, ops.DT[x2d.dtype], res.data_ptr(), ws.data_ptr(), nbytes,
             hip.current_stream())
    return out if direct else into(res, out, beta)


def act_bwd(z2d: torch.Tensor, dy2d: torch.Tensor, act: int) -> torch.Tensor:
    from . import hip, ops
    dz = torch.empty_like(z2d)
    hip.call("cfp_act_bwd", z2d.data_ptr(), z2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), act, dz.data_ptr(), dz.stride(0), z2d.shape[0],
             z2d.shape[1], ops.DT[z2d.dtype], hip.current_stream())
    return dz


def layernorm_bwd(x2d: torch.Tensor, dy2d: torch.Tensor, gamma: torch.Tensor, eps: float, dx: Optional[torch.Tensor] = None,
                  accumulate: bool = False, dgamma_out: Optional[torch.Tensor] = None, dbeta_out: Optional[torch.Tensor] = None,
                  queue: Optional["WgradQueue"] = None):
    """-> (dx, dgamma, dbeta).  `queue`: the finishing sum of the parameter-gradient partials joins the batched reduction (WgradQueue);
    dgamma / dbeta then hold the result only after queue.flush()."""
    from . import hip, ops
    rows, C = x2d.shape
    if dx is None:
        dx, accumulate = torch.empty_like(x2d), False
    nbytes = hip.load().cfp_layernorm_bwd_ws_bytes(rows, C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    dgamma = dgamma_out if dgamma_out is not None else torch.empty(C, dtype=torch.float32, device=x2d.device)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(C, dtype=torch.float32, device=x2d.device)
    if queue is not None:
        job = hip.WgradJob()
        hip.call("cfp_layernorm_bwd_deferred", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), gamma.data_ptr(), eps, dx.data_ptr(),
                 dx.stride(0), int(accumulate), dgamma.data_ptr(), dbeta.data_ptr(), rows, C, ops.DT[x2d.dtype], ws.data_ptr(), nbytes,
                 ctypes.addressof(job), hip.current_stream())
        queue.jobs.append(job)
        queue.keep.append((ws, dgamma, dbeta))
        return dx, dgamma, dbeta
    hip.call("cfp_layernorm_bwd", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), gamma.data_ptr(), eps, dx.data_ptr(),
             dx.stride(0), int(accumulate), dgamma.data_ptr(), dbeta.data_ptr(), rows, C, ops.DT[x2d.dtype], ws.data_ptr(), nbytes,
             hip.current_stream())
    return dx, dgamma, dbeta


def axpby(x2d, y2d, a: float, b: float, out=None):
    from . import hip, ops
    out = torch.empty_like(x2d) if out is None else out
    hip.call("cfp_axpby", x2d.data_ptr(), x2d.stride(0), hip.ptr(y2d), y2d.stride(0) if y2d is not None else 0, a, b, out.data_ptr(),
             out.stride(0), x2d.shape[0], x2d.shape[1], ops.DT[x2d.dtype], hip.current_stream())
    return out


def rowtable_grad(dx2d, dtable, B, H, W, Wt, oy, ox, beta: float = 0.0):
    from . import hip, ops
    hip.call("cfp_rowtable_grad", dx2d.data_ptr(), dx2d.stride(0), dtable.data_ptr(), B, H, W, dx2d.shape[1], Wt, oy, ox, beta,
             ops.DT[dx2d.dtype], hip.current_stream())
    return dtable


def channel_dot(x2d, y2d, B, HW):
    from . import hip, ops
    out = torch.empty(B, x2d.shape[1], dtype=torch.float32, device=x2d.device)
    hip.call("cfp_channel_dot", x2d.data_ptr(), x2d.stride(0), y2d.data_ptr(), y2d.stride(0), out.data_ptr(), B, HW, x2d.shape[1],
             ops.DT[x2d.dtype], hip.current_stream())
    return out


def se_train_fwd(part, ns, inv_hw, w1, b1, w2, b2, B):
    """-> (mean [B, C], z1 [B, Rp], gate [B, C]) float32 from the channel-sum partials [B * ns, C] (cfp_se_train_fwd)."""
    from . import hip
    C, Rp = w1.shape[1], w1.shape[0]
    f = lambda n: torch.empty(B, n, dtype=torch.float32, device=part.device)
    mean, z1, gate = f(C), f(Rp), f(C)
    hip.call("cfp_se_train_fwd", part.data_ptr(), ns, float(inv_hw), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), mean.data_ptr(),
             z1.data_ptr(), gate.data_ptr(), B, C, Rp, hip.current_stream())
    return mean, z1, gate


def se_train_bwd(dgate, gate, z1, mean, w1, w2, inv_hw, outs=None):
    """-> (dW1, db1, dW2, db2, add [B, C]); `outs`: four float32 tensors the parameter gradients are written into (beta = 0)."""
    from . import hip
    B, C = gate.shape
    Rp = w1.shape[0]
    dev = gate.device
    if outs is None:
        outs = (torch.empty(Rp, C, dtype=torch.float32, device=dev), torch.empty(Rp, dtype=torch.float32, device=dev),
                torch.empty(C, Rp, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev))
    add = torch.empty(B, C, dtype=torch.float32, device=dev)
    ws = torch.empty(int(hip.load().cfp_se_train_ws_floats(B, C, Rp)), dtype=torch.float32, device=dev)
    hip.call("cfp_se_train_bwd", dgate.data_ptr(), gate.data_ptr(), z1.data_ptr(), mean.data_ptr(), w1.data_ptr(), w2.data_ptr(),
             outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), add.data_ptr(), ws.data_ptr(), float(inv_hw), 0.0,
             B, C, Rp, hip.current_stream())
    return outs + (add,)


def bcast_fma(dy2d, gate, add, B, HW, dx=None):
    from . import hip, ops
    dx = torch.empty_like(dy2d) if dx is None else dx
    hip.call("cfp_bcast_fma", dy2d.data_ptr(), dy2d.stride(0), gate.data_ptr(), hip.ptr(add), dx.data_ptr(), dx.stride(0), B, HW,
             dy2d.shape[1], ops.DT[dy2d.dtype], hip.current_stream())
    return dx


def dwconv3x3_dgrad(dy2d, w9c, B, H, W, stride, pad_t, pad_l, Ho, Wo, dx=None, accumulate=False):
    from . import hip, ops
    C = dy2d.shape[1]
    if dx is None:
        dx, accumulate = torch.empty(B * H * W, C, dtype=dy2d.dtype, device=dy2d.device), False
    hip.call("cfp_dwconv3x3_dgrad", dy2d.data_ptr(), dy2d.stride(0), w9c.data_ptr(), dx.data_ptr(), dx.stride(0), B, H, W, C, stride, pad_t,
             pad_l, Ho, Wo, int(accumulate), ops.DT[dy2d.dtype], hip.current_stream())
    return dx


def dwconv3x3_wgrad(x2d, dy2d, B, H, W, stride, pad_t, pad_l, Ho, Wo, dw=None, beta: float = 0.0, queue: Optional["WgradQueue"] = None):
    from . import hip, ops
    C = x2d.shape[1]
    if dw is None:
        dw, beta = torch.empty(9, C, dtype=torch.float32, device=x2d.device), 0.0
    nbytes = hip.load().cfp_dwconv3x3_wgrad_ws_bytes(C)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    if queue is not None:
        job = hip.WgradJob()
        hip.call("cfp_dwconv3x3_wgrad_deferred", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), B, H, W, C, stride,
                 pad_t, pad_l, Ho, Wo, beta, ops.DT[x2d.dtype], ws.data_ptr(), nbytes, ctypes.addressof(job), hip.current_stream())
        queue.jobs.append(job)
        queue.keep.append((ws, dw))
        return dw
    hip.call("cfp_dwconv3x3_wgrad", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), B, H, W, C, stride, pad_t,
             pad_l, Ho, Wo, beta, ops.DT[x2d.dtype], ws.data_ptr(), nbytes, hip.current_stream())
    return dw


def index_rows(x2d, idx, n_out=None, out=None, accumulate=False):
    """out[i] = x2d[idx[i]] (zeros where idx[i] < 0); idx: int32 device tensor."""
    from . import hip, ops
    n_out = idx.numel() if n_out is None else n_out
    if out is None:
        out, accumulate = torch.empty(n_out, x2d.shape[1], dtype=x2d.dtype, device=x2d.device), False
    hip.call("cfp_index_rows", x2d.data_ptr(), x2d.stride(0), idx.data_ptr(), out.data_ptr(), out.stride(0), n_out, x2d.shape[1],
             int(accumulate), ops.DT[x2d.dtype], hip.current_stream())
    return out


def inverse_index(idx: torch.Tensor, n_src: int) -> torch.Tensor:
    """Inverse of an injective partial row map (host-side integer plumbing): inv[idx[i]] = i, -1 elsewhere."""
    inv = torch.full((n_src,), -1, dtype=torch.int32, device=idx.device)
    valid = idx >= 0
    inv[idx[valid].long()] = torch.arange(idx.numel(), dtype=torch.int32, device=idx.device)[valid]
    return inv


def resize_bilinear_bwd(dy2d, B, Hs, Ws, Hd, Wd, dx=None, accumulate=False):
    from . import hip, ops
    C = dy2d.shape[1]
    if dx is None:
        dx, accumulate = torch.empty(B * Hs * Ws, C, dtype=dy2d.dtype, device=dy2d.device), False
    hip.call("cfp_resize_bilinear_bwd", dy2d.data_ptr(), dy2d.stride(0), dx.data_ptr(), dx.stride(0), B, Hs, Ws, Hd, Wd, C, int(accumulate),
             ops.DT[dy2d.dtype], hip.current_stream())
    return dx


def bin_centers(widths_normed, min_val, max_val):
    from . import hip
    B, NB = widths_normed.shape
    edges = torch.empty(B, NB + 1, dtype=torch.float32, device=widths_normed.device)
    centers = torch.empty(B, NB, dtype=torch.float32, device=widths_normed.device)
    hip.call("cfp_bin_centers", widths_normed.data_ptr(), min_val, max_val, edges.data_ptr(), centers.data_ptr(), B, NB, hip.current_stream())
    return edges, centers


def bin_centers_bwd(dcenters, min_val, max_val):
    from . import hip
    B, NB = dcenters.shape
    dw = torch.empty_like(dcenters)
    hip.call("cfp_bin_centers_bwd", dcenters.data_ptr(), min_val, max_val, dw.data_ptr(), B, NB, hip.current_stream())
    return dw


def softmax_expect(logits2d, centers, B, HW, dpred=None):
    """forward: pred [B*HW] f32; with dpred: (dlogits, dcenters)."""
    from . import hip, ops
    NB = centers.shape[1]
    dev = logits2d.device
    if dpred is None:
        pred = torch.empty(B * HW, dtype=torch.float32, device=dev)
        hip.call("cfp_softmax_expect", logits2d.data_ptr(), logits2d.stride(0), centers.data_ptr(), pred.data_ptr(), None, None, 0, None, B, HW,
                 NB, ops.DT[logits2d.dtype], None, 0, hip.current_stream())
        return pred
    nbytes = hip.load().cfp_softmax_expect_ws_bytes(B, HW, NB)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits2d)
    dcenters = torch.empty(B, NB, dtype=torch.float32, device=dev)
    hip.call("cfp_softmax_expect", logits2d.data_ptr(), logits2d.stride(0), centers.data_ptr(), None, dpred.data_ptr(), dlogits.data_ptr(),
             dlogits.stride(0), dcenters.data_ptr(), B, HW, NB, ops.DT[logits2d.dtype], ws.data_ptr(), nbytes, hip.current_stream())
    return dlogits, dcenters


def _linattn_ws(N, L, S, heads, d, device, split: bool):
    from . import hip
    nbytes = hip.load().cfp_linattn_ws_bytes(N, L, S, heads, d) if split else 0
    return (torch.empty(nbytes // 4, dtype=torch.float32, device=device), nbytes) if nbytes else (None, 0)


def linattn_fwd(q2d, k2d, v2d, N, L, S, heads, d, eps=1e-6, split: bool = True):
    """-> (out [N*L, heads*d], state) for linattn_bwd.  `split=False`: one workgroup per (group, head) whatever the shape."""
    from . import hip, ops
    out = torch.empty(N * L, heads * d, dtype=q2d.dtype, device=q2d.device)
    state = torch.empty(hip.load().cfp_linattn_state_bytes(N, heads, d) // 4, dtype=torch.float32, device=q2d.device)
    ws, nbytes = _linattn_ws(N, L, S, heads, d, q2d.device, split)
    hip.call("cfp_linattn_fwd", q2d.data_ptr(), q2d.stride(0), k2d.data_ptr(), k2d.stride(0), v2d.data_ptr(), v2d.stride(0), out.data_ptr(),
             out.stride(0), state.data_ptr(), N, L, S, heads, d, eps, ops.DT[q2d.dtype], hip.ptr(ws), nbytes, hip.current_stream())
    return out, state


def linattn_bwd(q2d, k2d, v2d, dout2d, state, N, L, S, heads, d, eps=1e-6, split: bool = True):
    from . import hip, ops
    dq, dk, dv = torch.empty_like(q2d), torch.empty_like(k2d), torch.empty_like(v2d)
    ws, nbytes = _linattn_ws(N, L, S, heads, d, q2d.device, split)
    hip.call("cfp_linattn_bwd", q2d.data_ptr(), q2d.stride(0), k2d.data_ptr(), k2d.stride(0), v2d.data_ptr(), v2d.stride(0), dout2d.data_ptr(),
             dout2d.stride(0), state.data_ptr(), dq.data_ptr(), dq.stride(0), dk.data_ptr(), dk.stride(0), dv.data_ptr(), dv.stride(0), N, L, S,
             heads, d, eps, ops.DT[q2d.dtype], hip.ptr(ws), nbytes, hip.current_stream())
    return dq, dk, dv


def dwconv_large_wgrad(x2d, dy2d, B, H, W, k, dw=None, beta: float = 0.0):
    """-> dw [C, k, k] f32 (weights laid out [C][ky][kx] like the forward kernel's)."""
    from . import hip, ops
    C = x2d.shape[1]
    if dw is None:
        dw, beta = torch.empty(C, k, k, dtype=torch.float32, device=x2d.device), 0.0
    nbytes = hip.load().cfp_dwconv_large_wgrad_ws_bytes(B, H, W, C, k)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x2d.device)
    hip.call("cfp_dwconv_large_wgrad", x2d.data_ptr(), x2d.stride(0), dy2d.data_ptr(), dy2d.stride(0), dw.data_ptr(), B, H, W, C, k, beta,
             ops.DT[x2d.dtype], ws.data_ptr(), nbytes, hip.current_stream())
    return dw
