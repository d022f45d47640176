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


from oracle.calibrate import calibrate_bn  # noqa: E402,F401  (moved to oracle/calibrate.py: bench.py's parity leg uses it too)
