"""One optimisation step of the reference's training loop (`train.py:104-135`) on the HIP path:
training forward + SILog + backward (`train_model.TrainNet`), gradient averaging over the data-parallel ranks
(`train_ops.allreduce_gradients`: a few large RCCL all-reduces over the flat gradient buffer), AdamW with the OneCycle
schedule (`train_ops.FlatAdamW`).  The parameters live once, in the flat float32 optimizer buffer; the network reads them
through views, so nothing is copied back after the update.

`kernel_layout=True` (default) keeps that buffer in the KERNELS' layouts ([Cout_pad, kh*kw*Cin_pad] GEMM operands, [9, C]
depthwise taps, zero-padded vectors): the per-step work outside the network is then ONE cast of the buffer to the 16-bit
storage type, the backward kernels write every parameter gradient at its final address, and the ~330 per-parameter
re-layouts / casts / gradient copies of the reference-layout mode disappear (AdamW, weight decay and the gradient norm are
elementwise, so the update is the same numbers; padding elements are and stay zero).  `state_dict()` converts back.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch

from . import spec, train_ops
from .train_model import TrainNet


class Trainer:
    def __init__(self, state_dict: Dict[str, torch.Tensor], layer_names: Sequence[str], *, lr: float, total_steps: int, weight_decay: float = 0.1,
                 div_factor: float = 25.0, final_div_factor: float = 100.0, hist_encoder_10x: bool = True, clip_grad_norm: Optional[float] = None,
                 device="cuda:0", dist=None, world: int = 1, n_bins: int = 256, min_val: float = 1e-3, max_val: float = 10.0,
                 change_embedding: bool = True, dtype=torch.float32, no_skip_inside: bool = False, norm: str = "linear", kernel_layout: bool = True,
                 base_resolution=spec.BASE_RESOLUTION, overlap_param_grads: bool = False, comm: str = "overlap", sync_loss: bool = False):
        self.dev = torch.device(device)
        self.dtype, self.kernel_layout, self._hist10 = dtype, kernel_layout, hist_encoder_10x
        self._net_kw = dict(n_bins=n_bins, min_val=min_val, max_val=max_val, change_embedding=change_embedding, dtype=dtype,
                            no_skip_inside=no_skip_inside, norm=norm, base_resolution=base_resolution)
        self._layers = list(layer_names)
        self._opt_kw = dict(lr=lr, total_steps=total_steps, div_factor=div_factor, final_div_factor=final_div_factor, weight_decay=weight_decay,
                            clip_grad_norm=clip_grad_norm)
        self._shadow, self._to_torch = None, None
        self._pending_opt = None         # optimizer state loaded before the first batch fixed the kernel layouts
        self._flip_jobs = None           # [(source buffer, flipped buffer, device descriptors, n, workgroups, dtype code)]
        self.dist, self.world = dist, world
        # sync_loss: the SILog loss over the GLOBAL batch (its three moments all-reduced), as the reference's nn.DataParallel loop
        # computes it; default is the per-rank loss of DistributedDataParallel-style training (SURVEY 8e documents the difference)
        self.sync_loss = bool(sync_loss) and dist is not None and world > 1
        # gradient averaging over the data-parallel ranks: "overlap" = the captured step is split where the RGB encoder's backward
        # begins and the non-encoder bucket is reduced on a second stream while the encoder's backward runs;
        # "sequential" = all buckets after the whole backward; "off" = skipped (measurement of the step without communication)
        assert comm in ("overlap", "sequential", "off")
        self.comm = comm
        self._comm_stream = None
        self._graph2 = None
        self._segments = None
        names = [(k, tuple(v.shape)) for k, v in state_dict.items()
                 if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
        self.flat = train_ops.FlatParams(names, train_ops.lr_group_of(hist_encoder_10x), device=self.dev)
        self.flat.load(state_dict)
        sd = dict(state_dict)
        for name, _ in names:
            sd[name] = self.flat.view(name)                      # the network sees the optimizer's buffer
        self.net = TrainNet(sd, layer_names, self.dev, **self._net_kw)
        for name, _ in names:                                     # TrainNet.__init__ keeps device tensors as they are: still views
            assert self.net.sd[name].data_ptr() == self.flat.view(name).data_ptr()
        self.opt = train_ops.FlatAdamW(self.flat, train_ops.OneCycle(lr, total_steps, div_factor, final_div_factor), weight_decay=weight_decay,
                                       clip_grad_norm=clip_grad_norm, overflow_guard=dtype == torch.float16)
        self.min_val = min_val
        self._graph = None
        if overlap_param_grads:        # measured SLOWER inside a captured step (48.8 vs 45.2 ms): off by default, see DESIGN 4.0
            self.net.side_stream = torch.cuda.Stream(device=self.dev)

    def _bind_kernel_layout(self, input_data: dict, offs) -> None:
        """First batch: one forward on a scratch copy of the network records every live parameter's kernel layout (they
        depend on how the forward uses the tensor, not on the batch); the optimizer state moves to a flat buffer of those."""
        from .autograd_hip import Tape
        scratch = TrainNet(self.net.sd, self._layers, self.dev, **self._net_kw)              # own copy of the running statistics
        scratch.discovered = {}
        scratch.forward(Tape(self.dev, self.dtype), input_data, offs)
        found = scratch.discovered
        kflat = train_ops.FlatParams([(n, tuple(t32.shape)) for n, (t32, _, _) in found.items()], train_ops.lr_group_of(self._hist10),
                                     device=self.dev, align=128)                                # 128 elements: every tensor starts on a 256-byte line in the 16-bit shadow too
        assert kflat.group_range[2][0] == kflat.group_range[2][1], "a dead tensor was used by the forward"
        for n, (t32, _, _) in found.items():
            kflat.view(n).copy_(t32)
        self._to_torch = {n: f for n, (_, f, _) in found.items()}
        self._ref_flat, self.flat = self.flat, kflat
        self._shadow = torch.empty(kflat.total, dtype=self.dtype, device=self.dev) if self.dtype != torch.float32 else None
        self.net.bind(kflat, self._shadow)
        k = self._opt_kw
        # the pre-bind optimizer never stepped (its moments are zero); a schedule position set before the first batch (train.py's
        # resume without optimizer state) carries over
        assert not bool(self.opt.m.any()) and not bool(self.opt.v.any()), "optimizer moments exist before the kernel layout was bound"
        resumed_at = self.opt.step_count
        self.opt = train_ops.FlatAdamW(kflat, train_ops.OneCycle(k["lr"], k["total_steps"], k["div_factor"], k["final_div_factor"]),
                                       weight_decay=k["weight_decay"], clip_grad_norm=k["clip_grad_norm"], overflow_guard=self.dtype == torch.float16)
        self.opt.step_count = resumed_at
        if self._pending_opt is not None:
            pending, self._pending_opt = self._pending_opt, None
            self.load_optimizer_state_dict(pending)

    def _plan_weight_flips(self) -> None:
        """After the first bound step every convolution whose data gradient is needed has recorded its geometry: from now on
        their flipped copies ([Cin][KH][KW][Cout], both kernel axes reversed) are refreshed by ONE launch per storage type at the
        start of the step instead of one launch per convolution in the backward."""
        from . import hip, ops
        lib = hip.load()
        jobs = []
        sources = ([self._shadow] if self._shadow is not None else []) + [self.flat.param]      # 16-bit operands / float32 ones
        for src in sources:
            rows, off, blocks = [], 0, 0
            for name, p in self.net.P.items():
                if p.geom is None or p.t.dtype != src.dtype:
                    continue
                co, kh, kw, ci = p.geom
                n = co * kh * kw * ci
                assert n == p.t.numel(), name
                nb = int(lib.cfp_weight_flip_blocks(n))
                rows.append((name, [self.flat._by_name[name].start, off, co, kh, kw, ci, blocks, nb], (ci, kh * kw * co)))
                off += (n + 7) // 8 * 8
                blocks += nb
            if not rows:
                continue
            dst = torch.empty(off, dtype=src.dtype, device=self.dev)
            desc = torch.tensor([r[1] for r in rows], dtype=torch.int64).to(self.dev)
            for name, d, shape in rows:
                self.net.flipped[name] = dst[d[1]:d[1] + shape[0] * shape[1]].view(shape)
            jobs.append((src, dst, desc, len(rows), blocks, ops.DT[src.dtype]))
        self._flip_jobs = jobs

    def _refresh_weight_flips(self) -> None:
        from . import hip
        for src, dst, desc, n, blocks, dt in self._flip_jobs:
            hip.call("cfp_conv2d_weight_flip_batch", src.data_ptr(), dst.data_ptr(), desc.data_ptr(), n, blocks, dt, hip.current_stream())

    def param(self, name: str) -> torch.Tensor:
        """Parameter `name` in the reference's layout (a copy when the master is kept in kernel layout)."""
        if self._to_torch is not None and name in self._to_torch:
            return self._to_torch[name](self.flat.view(name)).contiguous()
        return self.net.sd[name]

    def draw_pos_offsets(self, H: int, W: int) -> Dict[str, tuple]:
        """fusion.py:87-91: a random window into the learned positional table whenever the token map is smaller than it."""
        offs = {}
        for name, (_, (Hm, Wm), _) in self.net.fusion.items():
            s = self.net.base_resolution[1] // Wm
            h, w = H // s, W // s
            oy = int(torch.randint(0, Hm - h + 1, [1])) if h < Hm else 0
            ox = int(torch.randint(0, Wm - w + 1, [1])) if w < Wm else 0
            offs[name] = (oy, ox)
        return offs

    def _loss_sync(self):
        return (self.dist, self.world) if self.sync_loss else None

    def _grads_to_flat(self, input_data, target, offs, stop_before_encoder: bool = False, stop: Optional[str] = None, defer: bool = False):
        if self.kernel_layout and self._to_torch is None:
            self._bind_kernel_layout(input_data, offs)
        self.net.zero_grad()
        if self._to_torch is not None:
            if self._shadow is not None:
                self._shadow.copy_(self.flat.param)                 # the one cast of the step
            if self._flip_jobs is not None:
                self._refresh_weight_flips()
            loss, _, _ = self.net.forward_backward(input_data, target, target > self.min_val, pos_offsets=offs,
                                                   stop_before_encoder=stop_before_encoder, loss_sync=self._loss_sync(), stop=stop,
                                                   defer_param_grads=defer)
            if self._flip_jobs is None:
                self._plan_weight_flips()
            return loss                                             # every gradient is already at its flat address
        loss, pred, _ = self.net.forward_backward(input_data, target, target > self.min_val, pos_offsets=offs, loss_sync=self._loss_sync())
        self.flat.grad.zero_()
        for name, g in self.net.grads().items():
            self.flat.view(name, "grad").copy_(g)
        return loss

    def capture(self, input_data: dict, target: torch.Tensor, split: Optional[bool] = None, wgrad_beside: Optional[bool] = None):
        """Record forward + loss + backward + gradient gathering for this batch shape into ONE HIP graph (the eager step is
        ~5 600 launches and host-bound in 16-bit mode).  Inputs are copied into static buffers at every `step`; the random
        positional-encoding windows are read by the kernels from a device buffer, so they still change per step.

        `split` (default: when there is a process group and comm == "overlap"): TWO graphs instead, cut where the backward of the
        RGB encoder begins.  After the first one every non-encoder gradient (the "10x" group of the flat buffer, 2/3 of the bytes)
        is final and its all-reduce runs on a communication stream beside the second graph; same kernels in the same order as
        the single graph, so the results are bit-identical.

        `wgrad_beside` (default OFF; CFP_TRAIN_WGRAD_BESIDE=1 turns it on -- measured 30.00 vs 30.00 ms on the benched shard: the step is
        the SUM of its kernels' durations, rocprofv3 span 31.9 ms vs 31.3 ms of kernel time, i.e. every kernel already fills the chip and
        a second queue only interleaves them; kept because it is the tested way to move work off the critical path): the weight / bias gradient kernels
        of the dense and depthwise convolutions leave the critical path  dY -> dX -> dY ...: the backward is cut at the tape marks
        (TrainNet.BACKWARD_MARKS) into segments, each segment's parameter-gradient kernels are captured as a graph of their own and
        replayed on a second hardware queue while the NEXT segment's data gradients run on the first.  Nothing on the critical path
        reads a weight gradient before the optimizer, so only the last segment's stay exposed.  Same kernels, same order within
        each stream, same operands: bit-identical to the single graph (tests/test_train_gpu.py)."""
        if wgrad_beside is None:
            wgrad_beside = self.kernel_layout and os.environ.get("CFP_TRAIN_WGRAD_BESIDE", "0") == "1"
        if wgrad_beside and not self.kernel_layout:
            raise ValueError("wgrad_beside needs kernel_layout=True (gradients written at their final flat addresses)")
        if split is None:
            split = self.dist is not None and self.comm == "overlap"
        if split and not self.kernel_layout:
            raise ValueError("the split step needs kernel_layout=True (gradients written at their final flat addresses)")
        if self.sync_loss and self.dist.get_backend() != "nccl":
            # the moments' all-reduce sits between the loss forward and backward, inside the captured region: RCCL collectives can be
            # captured, gloo's host-staged ones cannot (untested on a multi-GPU node from this single-GPU box: eager is the tested path)
            raise ValueError("sync_loss with a captured step needs the nccl (RCCL) backend; use the eager Trainer.step under gloo")
        dev = self.dev
        add = input_data["additional"]
        self._sin = {"rgb": input_data["rgb"].to(dev, torch.float32).contiguous().clone(),
                     "additional": {"hist_data": add["hist_data"].to(dev, torch.float32).contiguous().clone(),
                                    "mask": add["mask"].to(dev).contiguous().clone(), "rect_data": add.get("rect_data"),
                                    "patch_info": add["patch_info"]}}
        self._starget = target.to(dev, torch.float32).contiguous().clone()
        self._soffs = torch.zeros(3, 2, dtype=torch.int32, device=dev)
        self._offs_dev = {name: self._soffs[i] for i, name in enumerate(("cross_atten3", "cross_atten2", "cross_atten1"))}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        saved = {k: v.clone() for k, v in self.net.buf.items()}       # the warm-up passes and the capture pass itself are real
        with torch.cuda.stream(side):                                 # training forwards: they must not move the running statistics
            for _ in range(2):                                       # warm-up: builds the index maps, sets kernel attributes
                self._grads_to_flat(self._sin, self._starget, self._offs_dev)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._segments = None
        if wgrad_beside:
            self._capture_segments()
            for k, v in saved.items():
                self.net.buf[k].copy_(v)
            return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._sloss = self._grads_to_flat(self._sin, self._starget, self._offs_dev, stop_before_encoder=split)
        self._graph2 = None
        if split:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool()):                # same memory pool: always replayed right after g
                self.net.finish_backward()
            self._graph2 = g2
        for k, v in saved.items():
            self.net.buf[k].copy_(v)
        self._graph = g

    def _capture_segments(self) -> None:
        """The captured step as 2 n graphs: data-path segment i (main pool, main stream) and its parameter gradients (own pool, the
        second stream).  A parameter-gradient graph runs beside the NEXT data-path segment, so nothing it reads may be handed out
        again by the allocator while that segment is captured: the queued closures (which hold those tensors) stay referenced until
        every graph exists; after that the pools' block assignments are fixed."""
        from .engine import concurrent_streams
        dev = self.dev
        marks = list(self.net.BACKWARD_MARKS)
        # both from the probed set: a stream picked blindly may share its hardware queue with the caller's stream, and then nothing overlaps
        self._mstream, self._wstream = concurrent_streams(dev, want=2)
        keep, segs = [], []
        wpool = None
        main = torch.cuda.CUDAGraph()
        with torch.cuda.graph(main):
            self._sloss = self._grads_to_flat(self._sin, self._starget, self._offs_dev, stop=marks[0], defer=True)
        pool = main.pool()
        for i in range(len(marks) + 1):
            if i > 0:
                main = torch.cuda.CUDAGraph()
                with torch.cuda.graph(main, pool=pool):
                    self.net.finish_backward(stop=marks[i] if i < len(marks) else None)
            wg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(wg, **({"pool": wpool} if wpool is not None else {})):
                keep.append(self.net.run_deferred_param_grads(last=i == len(marks)))
            wpool = wg.pool()
            segs.append((main, wg, marks[i] if i < len(marks) else None))
        del keep
        self._segments = segs
        self._graph = segs[0][0]

    def _replay_segments(self) -> None:
        """main: D0 D1 ... Dn | second stream: W0 after D0, W1 after D1 and W0, ...; with a process group and comm == "overlap" the
        non-encoder bucket is reduced on the communication stream once the last segment above the encoder mark has its gradients."""
        cur, ms, ws = torch.cuda.current_stream(self.dev), self._mstream, self._wstream
        overlap = self.dist is not None and self.comm == "overlap"
        ms.wait_stream(cur)
        ws.wait_stream(cur)
        for main, wg, mark in self._segments:
            with torch.cuda.stream(ms):
                main.replay()
            ws.wait_stream(ms)
            with torch.cuda.stream(ws):
                wg.replay()
                if overlap and mark == "encoder":
                    self._reduce_group(1, beside=True)       # ordered after this segment's parameter gradients (current stream = ws)
        cur.wait_stream(ms)
        cur.wait_stream(ws)
        if overlap:
            self._reduce_group(0, beside=True)
            cur.wait_stream(self._comm_stream)
        else:
            self._reduce_group(None)

    def step(self, input_data: dict, target: torch.Tensor, pos_offsets: Optional[dict] = None):
        """-> (loss as a device scalar, lr, beta1).  `target` [B,1,H,W]; the loss mask is target > min_depth (train.py:121)."""
        H, W = input_data["rgb"].shape[-2:]
        offs = pos_offsets if pos_offsets is not None else self.draw_pos_offsets(H, W)
        if self._graph is not None:
            add = input_data["additional"]
            self._sin["rgb"].copy_(input_data["rgb"], non_blocking=True)
            self._sin["additional"]["hist_data"].copy_(add["hist_data"], non_blocking=True)
            self._sin["additional"]["mask"].copy_(add["mask"], non_blocking=True)
            self._starget.copy_(target, non_blocking=True)
            self._soffs.copy_(torch.tensor([offs[n] for n in ("cross_atten3", "cross_atten2", "cross_atten1")], dtype=torch.int32), non_blocking=True)
            loss = self._sloss
            if self._segments is not None:
                self._replay_segments()
                lr, beta1 = self.opt.step()
                return loss, lr, beta1
            self._graph.replay()
            if self._graph2 is not None and self.comm != "overlap":
                self._graph2.replay()
                self._reduce_group(None)
            elif self._graph2 is not None:
                self._reduce_group(1, beside=True)                   # head / decoder / fusion / ToF-encoder gradients: final now
                self._graph2.replay()                                # RGB encoder backward, beside that all-reduce
                self._reduce_group(0, beside=True)
                torch.cuda.current_stream(self.dev).wait_stream(self._comm_stream)
            else:
                self._reduce_group(None)
        else:
            loss = self._grads_to_flat(input_data, target, offs)
            self._reduce_group(None)
        lr, beta1 = self.opt.step()
        return loss, lr, beta1

    def _reduce_group(self, grp: Optional[int], beside: bool = False) -> None:
        """Average the gradients of lr group `grp` (None: all live ones) over the ranks.  `beside`: on the communication stream,
        ordered after everything issued so far on the current stream; the caller joins the stream before the optimizer."""
        if self.dist is None or self.comm == "off":
            if beside and self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.dev)
            return
        lo, hi = (0, self.flat.live) if grp is None else self.flat.group_range[grp]
        if not beside:
            train_ops.allreduce_range(self.flat, self.dist, self.world, lo, hi)
            return
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.dev)
        cur = torch.cuda.current_stream(self.dev)
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            handles = train_ops.allreduce_range(self.flat, self.dist, self.world, lo, hi, async_op=not train_ops._is_gloo(self.dist))
            train_ops.finish_allreduce(handles, self.world)

    # ------------------------------------------------------------------ optimizer state (checkpoint "optimizer" entry)
    def _layout_signature(self):
        return [(sg.name, sg.start, sg.numel) for sg in self.flat.segments]

    def optimizer_state_dict(self) -> dict:
        """AdamW moments + step counter for `model_io.save_checkpoint`'s "optimizer" entry (train.py:150).  The moments are
        saved as the flat buffers they live in, with the (name, start, numel) table of that layout: a resumed run must lay its
        parameters out the same way (same code, same attention_layer list), which `load_optimizer_state_dict` checks."""
        if self.kernel_layout and self._to_torch is None:
            raise RuntimeError("optimizer state exists only after the first training step fixed the kernel layouts")
        return {"format": "cfpnet_amd.FlatAdamW/1", "kernel_layout": self._to_torch is not None, "step_count": self.opt.step_count,
                "layout": self._layout_signature(), "exp_avg": self.opt.m.detach().cpu().clone(), "exp_avg_sq": self.opt.v.detach().cpu().clone()}

    def load_optimizer_state_dict(self, state: dict) -> None:
        if state.get("format") != "cfpnet_amd.FlatAdamW/1":
            raise ValueError("not a cfpnet_amd optimizer state (the reference's torch AdamW state is not restored by train.py:83-84 either)")
        if state["kernel_layout"] and self._to_torch is None:
            if not self.kernel_layout:
                raise ValueError("optimizer state was saved in kernel layout; this Trainer keeps the reference layout")
            self._pending_opt = state                       # applied by _bind_kernel_layout on the first batch
            return
        if not state["kernel_layout"] and self.kernel_layout:
            # a reference-layout state into a Trainer that will re-lay its parameters on the first batch: the moments would be written
            # into the pre-bind optimizer and dropped by the bind -- refuse instead of losing them silently
            raise ValueError("optimizer state was saved in the reference layout (kernel_layout=False); this Trainer keeps kernel layouts -- "
                             "resume it with Trainer(kernel_layout=False)")
        if [tuple(x) for x in state["layout"]] != [tuple(x) for x in self._layout_signature()]:
            raise ValueError("optimizer state was saved for a different parameter layout")
        self.opt.m.copy_(state["exp_avg"])
        self.opt.v.copy_(state["exp_avg_sq"])
        self.opt.step_count = int(state["step_count"])

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Parameters and running statistics in the reference's layout (for `model_io.save_checkpoint`)."""
        out = {}
        for k, v in self.net.sd.items():
            if self._to_torch is not None and k in self._to_torch:
                v = self.param(k)
            out[k] = (self.net.buf[k] if k in self.net.buf else v).detach().clone().cpu() if torch.is_tensor(v) else v
        return out
