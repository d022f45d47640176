"""ToF (L5 zone histogram) simulation on the GPU: host-side mirror of the reference's data-loader functions.

Reference interface: `src/utils/dataloader.py` -- `get_hist_parallel(rgb, dep, config)` (:83-134) and
`sample_point_from_hist_parallel(hist_data, mask, config)` (:65-80), called per sample by the data-loader workers
(`src/dataloader/nyu.py:154,179`).  Same names, argument meaning and return values here, but the arrays are device
tensors and the work is the HIP kernel behind `cfp_tof_hist_sim` / `cfp_tof_sample_points` (`csrc/tof_sim.hip`).
`TofSimulator.simulate` is the batched form (one launch for B depth maps) that a device-side input pipeline uses.

There is no CPU implementation in this package: without the HIP library the calls raise.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import hip

BIN_WIDTH = 0.04      # dataloader.py:93,104
AMBIENT_FLOOR = 20    # dataloader.py:109


def _cfg(config, name, default):
    return getattr(config, name, default)


def icdf_ppf_points(nsamp: int) -> torch.Tensor:
    """dataloader.py:70-71: `torch.Tensor(np.arange(delta, 1, (1 - 2 delta) / (S - 1)).tolist())`, float32."""
    delta = 1e-3
    return torch.Tensor(np.arange(delta, 1, (1 - 2 * delta) / (nsamp - 1)).tolist())


def zone_layout(config, height: int, width: int) -> Tuple[int, int, int, int]:
    """(zone_num, zone_px, sy0, sx0) of the centred zone grid before the random offset (dataloader.py:94-103)."""
    train = _cfg(config, "mode", "online_eval") == "train"
    zp = 64 if train else 56
    zn = int(_cfg(config, "train_zone_num", 8)) if train else 8
    return zn, zp, int((height - zp * zn) / 2), int((width - zp * zn) / 2)


class TofSimulator:
    """Batched `get_hist_parallel` + `sample_point_from_hist_parallel` on one device.

    Both branches of `sample_point_from_hist_parallel` are built.  With `config.sample_uniform` the two float32
    interpolation tables of `tensor_linspace` are evaluated by torch on the host, as the reference does, and kept on the
    device.  Without it (the argparse default, dataloader.py:69-73) the samples are `Normal(mu, sigma).icdf` at
    `zone_sample_num` ppf points; torch evaluates `erfinv(2 ppf - 1)` in float32 (the ppf tensor is float32) and only the
    product with the float64 sigma is promoted, so that float32 table is evaluated on the host the same way."""

    def __init__(self, config, device="cuda:0"):
        self.config = config
        self.device = torch.device(device)
        self.nsamp = int(_cfg(config, "zone_sample_num", 16))
        self.uniform = bool(_cfg(config, "sample_uniform", False))          # config.py:71: store_true, default False
        if self.uniform:
            self.w0 = torch.linspace(1, 0, steps=self.nsamp).to(self.device)
            self.w1 = torch.linspace(0, 1, steps=self.nsamp).to(self.device)
        else:
            ppf = icdf_ppf_points(self.nsamp)
            if ppf.numel() != self.nsamp:      # the reference's `fh[mask] = ...` assignment fails on the shape in that case
                raise ValueError(f"arange(1e-3, 1, 0.998/{self.nsamp - 1}) has {ppf.numel()} points, not zone_sample_num")
            self.w0 = torch.erfinv(2 * ppf - 1).to(self.device)
            self.w1 = None

    @property
    def sample_mode(self) -> int:
        return hip.TOF_SAMPLE_UNIFORM if self.uniform else hip.TOF_SAMPLE_ICDF

    def set_weights(self, w0, w1=None) -> None:
        """Install tables evaluated elsewhere (golden fixtures carry the generating host's): the two linspace tables in
        uniform mode, the erfinv table alone otherwise."""
        self.w0 = torch.as_tensor(np.asarray(w0), dtype=torch.float32).to(self.device).contiguous()
        self.w1 = None if w1 is None else torch.as_tensor(np.asarray(w1), dtype=torch.float32).to(self.device).contiguous()

    def max_distance(self) -> float:
        c = self.config
        if _cfg(c, "random_simu_max_d", False):                       # dataloader.py:86-89
            return float(np.random.uniform(low=c.simu_min_d, high=c.simu_max_d, size=1)[0])
        return float(_cfg(c, "simu_max_distance", 4.0))

    def simulate(self, depth: torch.Tensor, offsets: Optional[torch.Tensor] = None, max_distance: Optional[float] = None,
                 want_hist: bool = False, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """depth [B,H,W] or [B,1,H,W] f32 on the device -> dict(fh [B,Z,2] f64, rect_data [B,Z,4] f32, mask [B,Z] bool,
        hist_data [B,Z,S] f32 (the samples the model consumes), hist [B,Z,bins] i32 if want_hist).

        offsets: [B] int32 device tensor of per-sample grid offsets (the reference draws
        random.randint(-train_zone_random_offset, +train_zone_random_offset) per sample); None = 0."""
        if depth.dim() == 4:
            if depth.shape[1] != 1:
                raise ValueError("depth must have one channel")
            depth = depth[:, 0]
        if depth.dim() != 3 or depth.dtype != torch.float32 or not depth.is_cuda:
            raise ValueError("depth must be a float32 device tensor [B,H,W] or [B,1,H,W]")
        if depth.stride(2) != 1 or depth.stride(1) != depth.shape[2]:
            depth = depth.contiguous()
        B, H, W = depth.shape
        zn, zp, sy0, sx0 = zone_layout(self.config, H, W)
        Z = zn * zn
        md = self.max_distance() if max_distance is None else float(max_distance)
        bins = int(md / BIN_WIDTH)
        bound = int(_cfg(self.config, "train_zone_random_offset", 0)) if offsets is not None else 0
        if offsets is not None and (offsets.dtype != torch.int32 or not offsets.is_cuda or offsets.numel() != B):
            raise ValueError("offsets must be an int32 device tensor with one entry per image")
        dev = depth.device
        if out is None:
            out = {"fh": torch.empty(B, Z, 2, dtype=torch.float64, device=dev), "rect_data": torch.empty(B, Z, 4, dtype=torch.float32, device=dev),
                   "mask": torch.empty(B, Z, dtype=torch.bool, device=dev), "hist_data": torch.empty(B, Z, self.nsamp, dtype=torch.float32, device=dev)}
            if want_hist:
                out["hist"] = torch.empty(B, Z, bins, dtype=torch.int32, device=dev)
        hist = out.get("hist")
        hip.call("cfp_tof_hist_sim", depth.data_ptr(), depth.stride(0) if B > 1 else H * W, B, H, W, zn, zp, sy0, sx0,
                 offsets.data_ptr() if offsets is not None else None, bound, md, bins, BIN_WIDTH, AMBIENT_FLOOR,
                 self.w0.data_ptr(), hip.ptr(self.w1), self.nsamp, self.sample_mode, out["fh"].data_ptr(), out["rect_data"].data_ptr(),
                 out["mask"].data_ptr(), out["hist_data"].data_ptr(), hist.data_ptr() if hist is not None else None,
                 hip.current_stream())
        return out

    def sample_points(self, hist_data: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """[.., 2] f64 (mu, sigma) + [..] bool -> [.., S] f32 (dataloader.py:65-80, the branch `config.sample_uniform` selects)."""
        fh = hist_data.to(device=self.device, dtype=torch.float64).contiguous()
        mk = mask.to(device=self.device, dtype=torch.bool).contiguous()
        nz = mk.numel()
        pts = torch.empty(*mk.shape, self.nsamp, dtype=torch.float32, device=self.device)
        hip.call("cfp_tof_sample_points", fh.data_ptr(), mk.data_ptr(), self.w0.data_ptr(), hip.ptr(self.w1), nz, self.nsamp,
                 self.sample_mode, pts.data_ptr(), hip.current_stream())
        return pts


_sims: Dict[Tuple, TofSimulator] = {}


def _sim_for(config, device) -> TofSimulator:
    key = (id(config), str(device))
    s = _sims.get(key)
    if s is None or s.config is not config:
        s = _sims[key] = TofSimulator(config, device)
    return s


def get_hist_parallel(rgb: torch.Tensor, dep: torch.Tensor, config):
    """Drop-in for `get_hist_parallel(rgb, dep, config)` (dataloader.py:83): rgb [3,H,W] (only its size is used), dep
    [1,H,W] f32 on the device -> (fh [Z,2] f64, fr [Z,4] f32, mask [Z] bool), all on the device.  The random grid
    offset is drawn on the host with `random.randint`, like the reference (:98-100)."""
    import random
    sim = _sim_for(config, dep.device)
    offs = None
    tro = int(_cfg(config, "train_zone_random_offset", 0))
    if tro > 0:
        offs = torch.tensor([random.randint(-tro, tro)], dtype=torch.int32).to(dep.device)
    if tuple(rgb.shape[1:]) != tuple(dep.shape[1:]):
        raise ValueError("rgb and depth sizes differ")
    r = sim.simulate(dep, offsets=offs)
    return r["fh"][0], r["rect_data"][0], r["mask"][0]


def sample_point_from_hist_parallel(hist_data: torch.Tensor, mask: torch.Tensor, config) -> torch.Tensor:
    """Drop-in for `sample_point_from_hist_parallel(hist_data, mask, config)` (dataloader.py:65)."""
    return _sim_for(config, hist_data.device).sample_points(hist_data, mask)
