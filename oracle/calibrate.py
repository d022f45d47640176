"""Test infrastructure (like everything under oracle/): BatchNorm running statistics for synthetic weight families, computed with the
CPU oracle.  Imported by tests/, bench.py's parity leg and tools/ only -- never by cfpnet_amd/."""
from cfpnet_amd import synthetic


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
