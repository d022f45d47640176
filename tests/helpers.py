"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np

from cfpnet_amd import spec, synthetic, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DECODER_CASES = ["eval480_b1", "eval480_b2_drop", "train416_b1", "overhang_b1", "baseline_b1", "noskip_stale_b1"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"decoder_{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def case_inputs(meta):
    """Regenerate the exact parameters / inputs the golden generator fed the reference."""
    sd = weights.make_torch_state_dict(spec.model_manifest(meta["layer_names"]))
    inp = synthetic.make_inputs(meta["B"], meta["H"], meta["W"], meta["zone_num"], meta["zone_px"], seed=meta["seed"],
                                drop_hist=meta["drop_hist"], rect_shift=tuple(meta["rect_shift"]))
    feats = synthetic.make_img_features(meta["B"], meta["H"], meta["W"], seed=meta["seed"] + 1)
    d = meta["pos_draws"]
    offs = {"cross_atten3": (d[0], d[1]), "cross_atten2": (d[2], d[3]), "cross_atten1": (d[4], d[5])} if d else None
    return sd, inp, feats, offs


def rel_l1(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30))


def calibrate_bn(sd, layers, seed=901, B=2, H=480, W=640, zn=8, zpx=56):
    """Give a state dict the BatchNorm running statistics TRAINING would leave in it: one train-mode forward of the CPU oracle on
    a seeded calibration batch with momentum 1, so every running_mean / running_var becomes that layer's batch statistic (far from the
    (0, 1) of an untrained module).  Makes the reference's own initialisation family (weights.make_tensor_kaiming) a numerically
    meaningful network in eval mode.  Returns the calibrated copy."""
    import torch
    import torch.nn.functional as F
    from oracle import cfpnet_oracle as O
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()}
    inp = synthetic.make_inputs(B, H, W, zn, zpx, seed=seed, drop_hist=0.2)
    real = F.batch_norm

    def bn_momentum_one(x, rm, rv, w=None, b=None, training=False, momentum=0.1, eps=1e-5):
        return real(x, rm, rv, w, b, training, 1.0 if training else momentum, eps)

    old = O.BN_TRAIN
    O.BN_TRAIN, F.batch_norm = True, bn_momentum_one
    try:
        with torch.no_grad():
            O.forward(sd, inp, layer_names=layers)
    finally:
        O.BN_TRAIN, F.batch_norm = old, real
    return sd
