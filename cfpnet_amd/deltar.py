"""`Deltar`: the drop-in model object (boundary of SURVEY.md §8b).

Mirrors what the reference's callers use (`/root/reference/src/models/deltar.py:8-82`,
`src/utils/utils.py:7-11`, `train.py:78-80`, `src/utils/model_io.py:5-55`):

  * `make_model(args)` -> `Deltar(n_bins, min_val, max_val, norm)`
  * `forward(input_data)` with `input_data = {'rgb', 'additional': {hist_data, rect_data, mask,
    patch_info}}` returning `(bin_edges, pred)` in training mode and
    `(bin_edges, pred, prob, None)` in eval mode
  * `state_dict()` / `load_state_dict(strict)` with the reference's key names and shapes,
    including the 48 dead tensors, so the authors' checkpoints load unchanged
  * `get_1x_lr_params()` / `get_10x_lr_params()` parameter groups

It is a parameter container: no submodule has arithmetic of its own.  In eval mode `forward` hands the
parameters to `cfpnet_amd.engine.Engine` (inference kernels, HIP graphs); in training mode it is a
`torch.autograd.Function` over `cfpnet_amd.train_model.TrainNet` (training-mode forward and the backward of
every op as HIP kernels).  There is no PyTorch fallback: without a GPU and the built extension `forward` raises.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import torch
import torch.nn as nn

from . import spec, weights
from .config import args as _global_args


def _flag(ns, name, default):
    try:
        return getattr(ns, name)
    except AttributeError:
        return default


class _Store(nn.Module):
    """Registers parameters/buffers under dotted reference key names by building the matching
    tree of (empty) nn.Module containers."""

    def _put(self, dotted: str, tensor: torch.Tensor, is_buffer: bool):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Store())
            mod = mod._modules[p]
        if is_buffer:
            mod.register_buffer(parts[-1], tensor)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(tensor))


class Deltar(_Store):
    def __init__(self, n_bins: int = 100, min_val: float = 0.1, max_val: float = 10, norm: str = "linear", *,
                 args=None, dtype="f32x3", stem_act: bool = False, init: str = "deterministic",
                 base_resolution=spec.BASE_RESOLUTION, prob_dtype=torch.float32):
        """`dtype` (not in the reference) picks the numerics of the HIP path:
          "f32x3" (DEFAULT)  float32 storage, split-precision (f16 x 3) matrix math: the mode whose depth maps stay within the reference
                             tolerance (1e-3 relative L1 against the reference's float32 forward, deltar.py:34-67) on every weight family
                             -- measured ~2e-6 (tests/test_forward_gpu.py::test_f32x3_meets_the_gate_on_every_image_of_every_weight_family);
                             `prob` comes out of the kernels in float32 like the reference's.  Training-mode forwards run in float32.
                             DYNAMIC RANGE: a value is held as two IEEE halves -- full precision up to |x| = 6.5e4, 11 bits up to 1.31e5,
                             non-finite beyond (the float32 reference has no such limit; BatchNorm networks sit at O(1) ... O(100)).
                             `model.check_finite = True` makes every eval forward verify its depth map and raise FloatingPointError
                             (one host sync per forward); dtype=torch.float32 is the mode without the limit.
          torch.float32      float32 storage and float32 matrix cores: the bit-level parity mode (1/16 of the 16-bit matrix rate)
          torch.float16 / torch.bfloat16   16-bit storage, opt-in SPEED modes: 2.4 x the default's throughput, but outside the tolerance
                             on ill-conditioned (confident-head) networks -- fp16 0.8e-3 ... 1.2e-2, bf16 ~6e-3."""
        super().__init__()
        a = args if args is not None else _global_args
        self.num_classes = n_bins
        self.min_val, self.max_val, self.norm = min_val, max_val, norm
        # model-affecting flags are snapshotted at construction (the reference reads the global
        # `args` at construction AND at forward: fusion.py:16,25,134,154; deltar.py:69,77)
        self.layer_names: List[str] = list(_flag(a, "attention_layer", spec.BASELINE_LAYERS))
        self.zone_sample_num: int = int(_flag(a, "zone_sample_num", 16))
        self.change_embedding: bool = bool(_flag(a, "change_embedding", False))
        self.no_skip_inside: bool = bool(_flag(a, "no_skip_inside", False))
        self.hist_encoder_10x: bool = bool(_flag(a, "hist_encoder_10x", False))
        for ln in self.layer_names:
            if ln not in ("hist2image", "image", "combine1"):
                raise NotImplementedError(ln)      # fusion.py:37
        self.x3 = isinstance(dtype, str) and dtype.lower() in ("f32x3", "x3")
        if isinstance(dtype, str):
            dtype = {"f32x3": torch.float32, "x3": torch.float32, "f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype.lower()]
        self.compute_dtype = dtype
        # the reference returns `prob` in float32 (deltar.py:51,64-67); the engine writes it in its storage type (2 bytes per element at
        # batch 8 = 315 MB instead of 630 MB).  At THIS boundary the reference's type is the default (one cast); prob_dtype=None hands
        # out the engine's tensor as it is
        self.prob_dtype = prob_dtype
        self.stem_act = stem_act
        self.base_resolution = tuple(base_resolution)     # decoder.py:82-88 hard-codes 480x640; larger tables are a generalisation
        self._manifest = spec.model_manifest(self.layer_names, n_bins, self.zone_sample_num, self.base_resolution)
        for key, shape, kind in self._manifest:
            if init == "deterministic":
                t = torch.from_numpy(weights.make_tensor(key, shape, kind).copy())
            else:
                t = torch.zeros(shape, dtype=torch.int64 if kind == "bn_count" else torch.float32)
            self._put(key, t, is_buffer=kind in ("bn_mean", "bn_var", "bn_count"))
        self._engine = None
        self._engine_version = -1
        self._version_counter = 0
        self.train_graphs = True          # .train() forward / backward replayed as HIP graphs (captured per batch geometry)
        self._train_captures: Dict = {}
        # .eval() forwards are replayed as HIP graphs too, captured per input geometry (what evaluate_time.py:73-82 of the reference times is
        # `model(input_data)`; eager, a forward is ~260 launches and host-bound).  The results live in a RING of `eval_out_ring` output
        # sets: the tensors a forward returns stay valid until `eval_out_ring` forwards later (the reference's loops consume a batch's
        # outputs before the next forward: evaluate_all.py:42-60, train.py validate).  eval_graphs = False: eager launches, fresh tensors.
        self.eval_graphs = True
        self.eval_out_ring = 2
        # Default: every forward returns FRESH tensors like the reference (deltar.py:64-67) -- the graph's static outputs are cloned, so
        # `preds.append(model(x)[1])` over a dataset is safe.  eval_static_outputs = True is the opt-in fast path of a latency loop
        # (evaluate_time.py, bench.py): the returned tensors ARE the ring's buffers and are overwritten `eval_out_ring` forwards later.
        self.eval_static_outputs = False
        self.check_finite = False         # True: eval forwards raise FloatingPointError on a non-finite depth map (f32x3 range limit, see __init__)
        self._eval_caps: Dict = {}
        self._eval_offs = None            # int32[3, 2] on the device: the positional-table windows the captured graphs read
        self._eval_offs_host = None
        self._sig_cache = None

    # -- reference API ------------------------------------------------------------------
    def _get_name(self):
        return "Deltar"

    def _group(self, prefixes) -> Iterator[nn.Parameter]:
        for name, p in self.named_parameters():
            if name.split(".")[0] in prefixes:
                yield p

    def get_1x_lr_params(self):   # lr / 10 (deltar.py:68-74)
        return self._group(("img_encoder",) if self.hist_encoder_10x else ("img_encoder", "hist_encoder"))

    def get_10x_lr_params(self):  # lr (deltar.py:76-82)
        g = ("decoder", "depth_head", "conv_out") + (("hist_encoder",) if self.hist_encoder_10x else ())
        return self._group(g)

    def load_state_dict(self, state_dict, strict: bool = True):
        r = super().load_state_dict(state_dict, strict=strict)
        self._version_counter += 1
        return r

    def invalidate(self):
        """Call after mutating parameters in place so the packed device copies are rebuilt."""
        self._version_counter += 1

    # -- execution ----------------------------------------------------------------------
    def engine(self, device=None):
        from .engine import Engine
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        if dev.type != "cuda":
            dev = torch.device("cuda:0")
        if self._engine is None or self._engine_version != self._version_counter or self._engine.device != dev:
            sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
            self._engine = Engine(sd, layer_names=self.layer_names, n_bins=self.num_classes, min_val=self.min_val,
                                  max_val=self.max_val, norm=self.norm, change_embedding=self.change_embedding,
                                  no_skip_inside=self.no_skip_inside, stem_act=self.stem_act, dtype=self.compute_dtype,
                                  device=dev, zone_sample_num=self.zone_sample_num, base_resolution=self.base_resolution, x3=self.x3)
            self._engine_version = self._version_counter
            self._eval_caps = {}          # graphs of the previous engine read its packed parameters
        return self._engine

    def draw_pos_offsets(self, H: int, W: int) -> Dict[str, tuple]:
        """Window into the learned positional tables.  When the feature map is smaller than the
        table (416x544 crops) the reference draws the offset with `torch.randint` on the CPU
        generator, y then x, coarsest scale first (fusion.py:87-91); same draws here."""
        out = {}
        for name, s in (("cross_atten3", 16), ("cross_atten2", 8), ("cross_atten1", 4)):
            Hm, Wm = spec.fusion_table(self.base_resolution)[name][1]
            h, w = H // s, W // s
            oy = int(torch.randint(0, Hm - h + 1, [1])) if h < Hm else 0
            ox = int(torch.randint(0, Wm - w + 1, [1])) if w < Wm else 0
            out[name] = (oy, ox)
        return out

    def forward(self, input_data: Dict, **kwargs):
        if self.training:
            return self._forward_train(input_data, kwargs.get("pos_offsets"))
        eng = self.engine(input_data["rgb"].device if input_data["rgb"].is_cuda else None)
        pos_offsets = kwargs.get("pos_offsets")
        if pos_offsets is None:
            pos_offsets = self.draw_pos_offsets(input_data["rgb"].shape[-2], input_data["rgb"].shape[-1])
        return_prob = kwargs.get("return_prob", True)
        if self.eval_graphs:
            edges, pred, prob = self._forward_eval_graph(eng, input_data, pos_offsets, return_prob)
            if not self.eval_static_outputs:
                edges, pred = edges.clone(), pred.clone()
                if prob is not None and (self.prob_dtype is None or prob.dtype == self.prob_dtype):
                    prob = prob.clone()                    # (a dtype cast below is a fresh tensor already)
        else:
            edges, pred, prob = eng.forward(input_data, return_prob=return_prob, pos_offsets=pos_offsets)
        # the reference returns `prob` in float32 (deltar.py:51,64-67).  The default mode (f32x3) and the float32 mode write it in that
        # type themselves; only the opt-in 16-bit speed modes store 2 bytes per element and pay this cast (prob_dtype=None: as stored)
        if prob is not None and self.prob_dtype is not None and prob.dtype != self.prob_dtype:
            prob = prob.to(self.prob_dtype)
        if self.check_finite and not bool(torch.isfinite(pred).all()):
            raise FloatingPointError("cfpnet_amd: non-finite depth map" + (" -- the f32x3 mode holds values as two IEEE halves (|activation| < 1.31e5); "
                                     "build the model with dtype=torch.float32 for inputs / weights of this magnitude" if self.x3 else ""))
        return edges, pred, prob, None

    def _forward_eval_graph(self, eng, input_data: Dict, pos_offsets, return_prob: bool):
        """Eval forward as a HIP-graph replay.  Two kinds of graph per input geometry and output-ring slot: one that reads PRIVATE copies
        of the inputs (any caller: three device copies, then the replay) and -- from the second consecutive call with the very same
        device tensors on -- one that reads the CALLER'S tensors in place: that replay is pure host logic + one graph launch, no torch
        kernel, and sees whatever the tensors hold at that moment (the reference's latency loop, evaluate_time.py:73-82)."""
        rgb, add = input_data["rgb"], input_data["additional"]
        pinfo = add["patch_info"]
        if self._sig_cache is None or self._sig_cache[0] is not pinfo:      # the same dict object again (kept alive here): same integers
            self._sig_cache = (pinfo, _patch_signature(pinfo))
        # the positional windows (random per forward below the table size: fusion.py:87-91) are NOT part of the key: the graphs read them
        # from a device buffer, rewritten only when the drawn values change (never at 480x640, where they are all zero)
        key = (tuple(rgb.shape), tuple(add["hist_data"].shape), self._sig_cache[1], bool(return_prob))
        if self._eval_offs is None or self._eval_offs.device != eng.device:
            self._eval_offs = torch.zeros(3, 2, dtype=torch.int32, device=eng.device)
            self._eval_offs_host = (0,) * 6
            self._eval_offs_views = {n: self._eval_offs[i] for i, n in enumerate(_OFFSET_NAMES)}     # made once: no aten::select per forward
        host = tuple(int(v) for n in _OFFSET_NAMES for v in pos_offsets.get(n, (0, 0)))
        if host != self._eval_offs_host:
            self._eval_offs.copy_(torch.tensor(host, dtype=torch.int32).view(3, 2))
            self._eval_offs_host = host
        pos_offsets = self._eval_offs_views
        st = self._eval_caps.get(key)
        if st is None:
            if len(self._eval_caps) >= 4:                      # a handful of geometries at most: each pins its output ring
                self._eval_caps.pop(next(iter(self._eval_caps)))
            st = self._eval_caps[key] = {"calls": 0, "last": None, "owned": {}, "adopted": {}}
        m = add["mask"]
        ptrs = None
        if (rgb.is_cuda and rgb.device == eng.device and rgb.dtype == torch.float32 and rgb.is_contiguous() and add["hist_data"].device == eng.device
                and add["hist_data"].dtype == torch.float32 and add["hist_data"].is_contiguous() and m.device == eng.device
                and m.dtype in (torch.bool, torch.uint8) and m.is_contiguous()):
            ptrs = (rgb.data_ptr(), add["hist_data"].data_ptr(), m.data_ptr())
        slot = st["calls"] % max(1, int(self.eval_out_ring))
        st["calls"] += 1
        same = ptrs is not None and ptrs == st["last"]
        st["last"] = ptrs
        if same:
            h = st["adopted"].get((ptrs, slot))
            if h is None:
                if len(st["adopted"]) >= 2 * max(1, int(self.eval_out_ring)):
                    st["adopted"].clear()
                eng.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets, adopt_inputs=True)
                h = st["adopted"][(ptrs, slot)] = eng._graph
            eng._graph, eng._slots = h, None
            return eng.replay()
        h = st["owned"].get(slot)
        if h is None:
            eng.capture(input_data, return_prob=return_prob, pos_offsets=pos_offsets)
            h = st["owned"][slot] = eng._graph
        eng._graph, eng._slots = h, None
        return eng.replay(input_data)


_OFFSET_NAMES = ("cross_atten3", "cross_atten2", "cross_atten1")


def _patch_signature(pinfo) -> tuple:
    out = []
    for s in (4, 8, 16):
        e = pinfo[s] if s in pinfo else pinfo[float(s)]
        out.append(tuple(tuple(int(v) for v in torch.as_tensor(e[k]).reshape(-1).tolist()) for k in ("pad_size", "patch_size", "index_wo_pad")))
    return tuple(out) + (tuple(int(v) for v in torch.as_tensor(pinfo["zone_num"]).reshape(-1).tolist()),)


class _CapturedTrainStep:
    """Forward and backward of the training step as two HIP graphs sharing one memory pool, for a fixed batch geometry.
    The eager tape is ~3 500 launches plus ~1 400 small re-layout kernels per step and host-bound (178 ms per step of 16 crops);
    replayed, the same work is ~50 ms.  Inputs are copied into static buffers, the parameters are read in place (torch
    optimizers update them in place), the random positional-encoding windows come from a device buffer, the running
    statistics are the module's own buffers."""

    def __init__(self, model, input_data, names, params):
        from .autograd_hip import Tape
        from .train_model import TrainNet
        dev = params[0].device
        add = input_data["additional"]
        self.inp = {"rgb": input_data["rgb"].to(dev, torch.float32).contiguous().clone(),
                    "additional": {"hist_data": add["hist_data"].to(dev, torch.float32).contiguous().clone(),
                                   "mask": add["mask"].to(dev).contiguous().clone(), "rect_data": add.get("rect_data"),
                                   "patch_info": add["patch_info"]}}
        self.offs = torch.zeros(3, 2, dtype=torch.int32, device=dev)
        offs_dev = {n: self.offs[i] for i, n in enumerate(_OFFSET_NAMES)}
        sd = {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}
        self.net = TrainNet(sd, model.layer_names, dev, n_bins=model.num_classes, min_val=model.min_val, max_val=model.max_val,
                            stem_act=model.stem_act, change_embedding=model.change_embedding, share_buffers=True, dtype=model.compute_dtype,
                            no_skip_inside=model.no_skip_inside, norm=model.norm, base_resolution=model.base_resolution)
        self.names = list(names)
        # one real execution before the capture (sets kernel attributes, builds the index maps); its update of the running
        # statistics is undone, capture itself only records
        saved = {k: v.clone() for k, v in self.net.buf.items()}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            t = Tape(dev, model.compute_dtype)
            pred, _, (B, h, w) = self.net.forward(t, self.inp, offs_dev)
            pred.g = torch.zeros(B * h * w, 1, dtype=torch.float32, device=dev)
            t.backward()
            self.net.zero_grad()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for k, v in saved.items():
            self.net.buf[k].copy_(v)
        self.gf, self.gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gf):
            self.tape = Tape(dev, model.compute_dtype)
            self.pred, self.edges, (B, h, w) = self.net.forward(self.tape, self.inp, offs_dev)
        self.shape = (B, 1, h, w)
        self.gpred = torch.zeros(B * h * w, 1, dtype=torch.float32, device=dev)
        with torch.cuda.graph(self.gb, pool=self.gf.pool()):
            self.pred.g = self.gpred
            self.tape.backward()
            g = self.net.grads()
            self.grads = [g.get(n) for n in self.names]
        self.ptrs = tuple(p.data_ptr() for p in params)
        self.generation = 0              # the captured activations belong to the LAST forward only

    def forward(self, input_data, pos_offsets):
        self.generation += 1
        add = input_data["additional"]
        self.inp["rgb"].copy_(input_data["rgb"], non_blocking=True)
        self.inp["additional"]["hist_data"].copy_(add["hist_data"], non_blocking=True)
        self.inp["additional"]["mask"].copy_(add["mask"], non_blocking=True)
        self.offs.copy_(torch.tensor([pos_offsets.get(n, (0, 0)) for n in _OFFSET_NAMES], dtype=torch.int32), non_blocking=True)
        self.gf.replay()
        return self.edges.clone(), self.pred.t.reshape(self.shape).clone()

    def backward(self, g_pred):
        self.gpred.copy_(g_pred.reshape(-1, 1))
        self.gb.replay()
        # a contiguous gradient could be adopted by autograd as `.grad` without a copy and would then alias this static buffer
        return tuple(None if g is None else (g.clone() if g.is_contiguous() else g) for g in self.grads)


class _TrainStep(torch.autograd.Function):
    """`model(input_data)` in `model.train()` for callers that then run `loss.backward()` like the reference's train.py:
    forward = the training-mode forward on the HIP tape, backward = the tape's backward; the parameter gradients come back
    through autograd in the reference's layout, the running statistics are updated in the module's buffers.  With
    `model.train_graphs` (default) both halves are replayed as HIP graphs captured per batch geometry."""

    @staticmethod
    def forward(ctx, model, input_data, pos_offsets, names, *params):
        from .autograd_hip import Tape
        from .train_model import TrainNet
        dev = params[0].device
        ctx.names = names
        if model.train_graphs:
            add = input_data["additional"]
            key = (tuple(input_data["rgb"].shape), tuple(add["hist_data"].shape), _patch_signature(add["patch_info"]), model.compute_dtype)
            cap = model._train_captures.get(key)
            if cap is None or cap.ptrs != tuple(p.data_ptr() for p in params):
                model._train_captures.clear()                     # one geometry at a time: a capture pins ~25 GB of activations
                cap = model._train_captures[key] = _CapturedTrainStep(model, input_data, names, params)
            ctx.cap = cap
            edges, pred = cap.forward(input_data, pos_offsets)
            ctx.generation = cap.generation
            ctx.mark_non_differentiable(edges)
            return edges, pred
        ctx.cap = None
        sd = {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}
        net = TrainNet(sd, model.layer_names, dev, n_bins=model.num_classes, min_val=model.min_val, max_val=model.max_val,
                       stem_act=model.stem_act, change_embedding=model.change_embedding, share_buffers=True, dtype=model.compute_dtype,
                       no_skip_inside=model.no_skip_inside, norm=model.norm, base_resolution=model.base_resolution)
        tape = Tape(dev, model.compute_dtype)
        pred, edges, (B, h, w) = net.forward(tape, input_data, pos_offsets)
        ctx.net, ctx.tape, ctx.pred = net, tape, pred
        ctx.mark_non_differentiable(edges)
        return edges, pred.t.reshape(B, 1, h, w)

    @staticmethod
    def backward(ctx, _g_edges, g_pred):
        if ctx.cap is not None:
            if ctx.generation != ctx.cap.generation:
                raise RuntimeError("cfpnet_amd: backward of a training-mode forward whose activations were overwritten by a later forward "
                                   "of the same captured step (two forwards before one backward); set model.train_graphs = False for that pattern")
            return (None, None, None, None) + ctx.cap.backward(g_pred.to(torch.float32))
        ctx.pred.g = g_pred.reshape(-1, 1).to(torch.float32).contiguous()
        ctx.tape.backward()
        grads = ctx.net.grads()
        return (None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


def _forward_train(self, input_data: Dict, pos_offsets=None):
    if self.compute_dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise NotImplementedError(self.compute_dtype)
    named = [(n, p) for n, p in self.named_parameters()]
    if not named[0][1].is_cuda:
        raise RuntimeError("cfpnet_amd: the training step runs on the GPU -- call model.to('cuda') first (there is no CPU path)")
    if pos_offsets is None:
        pos_offsets = self.draw_pos_offsets(input_data["rgb"].shape[-2], input_data["rgb"].shape[-1])
    edges, pred = _TrainStep.apply(self, input_data, pos_offsets, [n for n, _ in named], *[p for _, p in named])
    self._version_counter += 1            # the next eval-mode engine must repack the (about to change) parameters
    return edges, pred                     # deltar.py:64-65: training returns (bin_edges, pred)


Deltar._forward_train = _forward_train


def make_model(args, dtype=None):
    """`src/utils/utils.py:7-11`.  `dtype` (not in the reference) picks the numerics of the HIP path: "f32x3" (default: float32 storage,
    split-precision matrix math -- inside the reference tolerance on every weight family), torch.float32 (bit-level parity mode),
    torch.float16 / torch.bfloat16 (16-bit storage: opt-in speed modes).  See `Deltar.__init__`."""
    if args.model_name == "deltar":
        kw = {} if dtype is None else {"dtype": dtype}
        return Deltar(n_bins=args.n_bins, min_val=args.min_depth, max_val=args.max_depth, norm=args.norm, args=args, **kw)
    raise NotImplementedError(args.model_name)
