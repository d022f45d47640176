"""Evaluation-side data path: samples -> model inputs, with the ToF simulation on the GPU.

Reference: `src/dataloader/nyu.py:62-197` (`DataLoadPreprocess.__getitem__`, online_eval branch) and `:266-310`
(`ToTensor`: HWC [0,1] -> CHW, ImageNet normalisation).  File decoding (PIL) stays on the host as in the reference;
everything after it -- `get_hist_parallel`, `sample_point_from_hist_parallel` -- runs as one HIP launch per batch
(`cfpnet_amd.tof`), and `patch_info_from_rect_data` is host integer arithmetic on the zone grid, which is known
without reading anything back from the device.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, List, Optional

import numpy as np
import torch

from . import synthetic
from .geometry import centered_zone_rects, collate_patch_info, patch_info_from_rect_data
from .tof import TofSimulator, zone_layout

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


class NYUEvalFiles:
    """The test split of `filenames_file_eval` (json, key 'test'): rgb_XXXXX.jpg + sync_depth_XXXXX.png under
    `data_path_eval` (nyu.py:72-76,101-107).  Yields (image [3,H,W] f32 normalised, depth [1,H,W] f32 metres, name)."""

    def __init__(self, args):
        with open(args.filenames_file_eval, "r") as f:
            self.samples = json.load(f)["test"]
        self.root = args.data_path_eval

    def __len__(self):
        return len(self.samples)

    def __iter__(self):
        from PIL import Image
        for s in self.samples:
            path = os.path.join(self.root, "/".join(s["filename"].split("/")[1:]))
            num = path.split("/")[-1].split(".")[0]
            base = "/".join(path.split("/")[:-1])
            rgb = np.asarray(Image.open(os.path.join(base, f"rgb_{num}.jpg")).convert("RGB"), dtype=np.float32) / 255.0
            dep = np.asarray(Image.open(os.path.join(base, f"sync_depth_{num}.png")), dtype=np.float32) / 1000.0
            img = (rgb - IMAGENET_MEAN) / IMAGENET_STD
            yield torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))), torch.from_numpy(dep)[None], s["filename"]


class NYUTrainFiles:
    """The train split of `filenames_file` (json, key 'train') under `data_path`, opened the way the reference's loader does
    (nyu.py:93-99,105-118): `rgb_XXXXX.jpg` + `sync_depth_XXXXX.png`, both cropped to (16, 12, 640-16, 480-12) "to avoid blank
    boundaries due to pixel registration".  Pixels stay what the files hold -- uint8 RGB and 16-bit millimetres: rotation,
    random crop, flip, jitter, /255, /1000 and normalisation happen on the device (`cfpnet_amd.augment`).

    `epoch_batches(bs)` yields (rgb [bs,456,608,3] uint8, depth [bs,456,608] int16 storage of uint16, names) in a fresh random
    order per epoch (`torch.randperm`, i.e. DataLoader(shuffle=True) semantics, incomplete last batch dropped), decoded by
    `num_threads` worker threads one batch ahead of the consumer."""

    CROP = (16, 12, 640 - 16, 480 - 12)

    def __init__(self, args, rank: int = 0, world: int = 1):
        with open(args.filenames_file, "r") as f:
            self.samples = json.load(f)["train"]
        self.root = args.data_path
        self.threads = max(1, int(getattr(args, "num_threads", 4) or 4))
        self.rank, self.world = rank, world

    def __len__(self):
        return len(self.samples)

    def paths(self, i: int):
        path = os.path.join(self.root, "/".join(self.samples[i]["filename"].split("/")[1:]))
        num = path.split("/")[-1].split(".")[0]
        base = "/".join(path.split("/")[:-1])
        return os.path.join(base, f"rgb_{num}.jpg"), os.path.join(base, f"sync_depth_{num}.png")

    def load(self, i: int):
        from PIL import Image
        rgb_path, depth_path = self.paths(i)
        with Image.open(rgb_path) as im:
            rgb = np.asarray(im.convert("RGB").crop(self.CROP), dtype=np.uint8)
        with Image.open(depth_path) as dm:
            dep = np.asarray(dm.crop(self.CROP)).astype(np.uint16)
        return rgb, dep

    def _batch(self, idx):
        from concurrent.futures import ThreadPoolExecutor
        if not hasattr(self, "_pool"):
            self._pool = ThreadPoolExecutor(self.threads)
        items = list(self._pool.map(self.load, idx))
        rgb = torch.from_numpy(np.stack([a for a, _ in items]))
        dep = torch.from_numpy(np.stack([b for _, b in items]).view(np.int16))
        return rgb, dep, [self.samples[i]["filename"] for i in idx]

    def epoch_batches(self, bs: int, generator: Optional[torch.Generator] = None) -> Iterator[tuple]:
        """This rank's share of one epoch: the global permutation is cut into global batches of bs * world samples and every
        rank takes its slice of each (the reference's DataParallel splits each batch over the GPUs the same way)."""
        import threading, queue
        if generator is None and self.world > 1:
            # every rank must cut the SAME permutation: an unseeded per-process randperm would duplicate some samples across
            # ranks and skip others in every epoch
            raise ValueError("epoch_batches: with world > 1 pass a torch.Generator seeded identically on every rank (seed + epoch)")
        order = torch.randperm(len(self.samples), generator=generator).tolist()
        gb = bs * self.world
        chunks = [order[k * gb + self.rank * bs:k * gb + (self.rank + 1) * bs] for k in range(len(order) // gb)]
        q: "queue.Queue" = queue.Queue(maxsize=2)

        def producer():
            try:
                for idx in chunks:
                    q.put(self._batch(idx))
                q.put(None)
            except BaseException as e:           # surfaces in the consumer
                q.put(e)
        threading.Thread(target=producer, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


class SyntheticEvalSamples:
    """Seeded stand-in for a dataset that is not on the box: random RGB, a planes-and-boxes depth map with holes."""

    def __init__(self, n: int, height: int = 480, width: int = 640, seed: int = synthetic.SEED):
        self.n, self.h, self.w, self.seed = n, height, width, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            rng = np.random.default_rng(self.seed + i)
            rgb = rng.random((3, self.h, self.w), dtype=np.float32)
            img = (rgb - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]
            dep = synthetic.make_depth(self.h, self.w, seed=self.seed + 7919 * (i + 1), holes=0.1 * (i % 3))
            yield torch.from_numpy(img), torch.from_numpy(dep)[None], f"synthetic/{i:05d}"


def batches(samples, batch_size: int) -> Iterator[tuple]:
    imgs, deps, names = [], [], []
    for img, dep, name in samples:
        imgs.append(img); deps.append(dep); names.append(name)
        if len(imgs) == batch_size:
            yield torch.stack(imgs), torch.stack(deps), names
            imgs, deps, names = [], [], []
    if imgs:
        yield torch.stack(imgs), torch.stack(deps), names


class EvalInputBuilder:
    """image/depth batch -> the `input_data` dict of `Deltar.forward` (`evaluate_all.py:55-65`), ToF branch on the GPU."""

    def __init__(self, args, device="cuda:0"):
        self.args = args
        self.device = torch.device(device)
        cfg = _as_mode(args, "online_eval")
        self.sim = TofSimulator(cfg, self.device)
        self.cfg = cfg
        self._pinfo: Dict = {}

    def patch_info(self, B: int, H: int, W: int):
        key = (B, H, W)
        if key not in self._pinfo:
            zn, zp, sy0, sx0 = zone_layout(self.cfg, H, W)
            rects = centered_zone_rects(H, W, zn, zp)
            pi = collate_patch_info([patch_info_from_rect_data(rects, (H, W))] * B)
            info = {s: {k: torch.from_numpy(v) for k, v in pi[s].items()} for s in (4, 8, 16)}
            info["zone_num"] = torch.from_numpy(pi["zone_num"])
            self._pinfo[key] = info
        return self._pinfo[key]

    def __call__(self, image: torch.Tensor, depth: torch.Tensor) -> Dict:
        dev = self.device
        img = image.to(dev, non_blocking=True)
        dep = depth.to(dev, dtype=torch.float32, non_blocking=True)
        B, _, H, W = img.shape
        sim = self.sim.simulate(dep)
        return {"rgb": img, "additional": {"hist_data": sim["hist_data"], "rect_data": sim["rect_data"], "mask": sim["mask"],
                                           "patch_info": self.patch_info(B, H, W)}}, dep


def _as_mode(args, mode):
    import copy
    c = copy.copy(args)
    c.mode = mode
    return c
