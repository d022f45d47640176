"""Pin the CPU oracle against outputs of the reference itself (fixtures from oracle/gen_golden.py).

Everything after the RGB encoder -- ToF histogram encoder, decoder with all fusion layer types,
depth head, bin maths -- must reproduce the reference's numbers on identical parameters/inputs."""
import numpy as np
import pytest
import torch

from helpers import DECODER_CASES, case_inputs, load_case, rel_l1
from oracle import cfpnet_oracle as O


@pytest.mark.parametrize("name", DECODER_CASES)
def test_decoder_head_matches_reference(name):
    z, meta = load_case(name)
    sd, inp, feats, offs = case_inputs(meta)
    taps = {}
    torch.set_num_threads(8)
    edges, pred, prob = O.forward(sd, inp, layer_names=meta["layer_names"], img_features=feats,
                                  change_embedding=meta["change_embedding"], no_skip_inside=meta["no_skip_inside"],
                                  pos_offsets=offs, taps=taps)
    mine = pred.numpy() if meta["full_pred"] else pred[:, :, ::4, ::4].numpy()
    assert mine.shape == z["pred"].shape
    assert rel_l1(mine, z["pred"]) < 2e-5          # fp32 re-association only
    assert np.abs(edges.numpy() - z["bin_edges"]).max() < 1e-5
    assert np.abs(prob[:, :, ::16, ::16].numpy() - z["prob_slice"]).max() < 1e-4
    assert rel_l1(taps["unet_out"][:, ::8, ::8, ::8].numpy(), z["unet_slice"]) < 2e-5
    for f in ("cross_atten1", "cross_atten2", "cross_atten3"):
        assert rel_l1(taps[f][:, ::4, ::4, ::4].numpy(), z[f + "_slice"]) < 2e-5
    # per-stage statistics (mean, |mean|, rms) recorded from forward hooks on the reference
    for k, ref in meta["tap_stats"].items():
        t = taps[k].double()
        got = [float(t.mean()), float(t.abs().mean()), float((t * t).mean().sqrt())]
        for g, r in zip(got[1:], ref[1:]):
            assert abs(g - r) <= 1e-4 * abs(r) + 1e-6, (k, got, ref)


def test_output_contract():
    """deltar.py:53-67: edges start at min_val, end at max_val, are increasing; pred lies inside."""
    z, meta = load_case("eval480_b1")
    e = z["bin_edges"]
    assert e.shape == (1, 257)
    assert abs(e[0, 0] - 1e-3) < 1e-7 and abs(e[0, -1] - 10.0) < 1e-4
    assert (np.diff(e, axis=1) > 0).all()
    assert z["pred"].shape == (1, 1, 240, 320) and z["pred"].min() > 1e-3 and z["pred"].max() < 10
    assert np.allclose(z["prob_slice"].sum(1), 1.0, atol=1e-4)


def test_misc_known_answers(golden_dir):
    import json
    import os
    m = json.load(open(os.path.join(golden_dir, "misc.json")))
    rng = np.random.default_rng(11)
    pred = torch.from_numpy(rng.uniform(0.3, 9.0, (2, 1, 26, 34)).astype(np.float32))
    gt = torch.from_numpy(rng.uniform(0.0, 9.0, (2, 1, 52, 68)).astype(np.float32))
    loss = O.silog_loss(pred, gt, mask=gt > 1.0, interpolate=True)
    assert abs(float(loss) - m["silog"]["loss"]) < 1e-5
    loss2 = O.silog_loss(pred, gt[:, :, ::2, ::2].clamp(min=0.1), mask=None, interpolate=False)
    assert abs(float(loss2) - m["silog"]["loss_nomask"]) < 1e-5
    g = rng.uniform(0.5, 9.0, 5000).astype(np.float32)
    p = (g * rng.uniform(0.7, 1.4, 5000)).astype(np.float32)
    for k, v in O.compute_errors(g, p).items():
        assert abs(float(v) - m["compute_errors"]["values"][k]) < 1e-6, k


def test_encoder_shapes():
    """encoder.py:71-79 / decoder.py:67: five taps with 16/40/56/136/232 channels at 1/2..1/32."""
    from cfpnet_amd import spec, weights
    sd = weights.make_torch_state_dict(spec.encoder_manifest())
    x = torch.zeros(1, 3, 64, 96)
    taps = O.encoder(sd, x)
    assert [t.shape[1] for t in taps] == [16, 40, 56, 136, 232]
    assert [tuple(t.shape[2:]) for t in taps] == [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    # odd sizes exercise the symmetric branch of TF-SAME padding
    taps = O.encoder(sd, torch.zeros(1, 3, 50, 70))
    assert [tuple(t.shape[2:]) for t in taps] == [(25, 35), (13, 18), (7, 9), (4, 5), (2, 3)]


def test_oracle_training_step_matches_the_reference_in_train_mode():
    """BN_TRAIN + autograd of the oracle against the reference's own modules in `.train()` + SILogLoss + loss.backward()
    (oracle/gen_golden_train.py; encoder bypassed with stand-in features): loss, prediction, the gradient statistics of all
    413 live parameters after the encoder and a dozen full gradient tensors."""
    import json
    import os
    from helpers import GOLDEN
    from cfpnet_amd import spec, synthetic, weights
    z = np.load(os.path.join(GOLDEN, "train_step.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    layers = spec.COMBINE1_LAYERS
    sd0 = weights.make_torch_state_dict(spec.model_manifest(layers))
    sd = {}
    for k, v in sd0.items():
        v = v.detach().clone()
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
        sd[k] = v
    inp = synthetic.make_inputs(meta["B"], meta["H"], meta["W"], meta["zn"], meta["zpx"], seed=meta["seed"], drop_hist=meta["drop"])
    feats = synthetic.make_img_features(meta["B"], meta["H"], meta["W"], seed=meta["seed"] + 1)
    target = torch.from_numpy(np.stack([synthetic.make_depth(meta["H"], meta["W"], seed=meta["seed"] + 5 + i, holes=0.1) for i in range(meta["B"])]))[:, None]
    d = [int(x) for x in z["draws"]]
    offs = {"cross_atten3": (d[0], d[1]), "cross_atten2": (d[2], d[3]), "cross_atten1": (d[4], d[5])}
    O.BN_TRAIN = True
    try:
        edges, pred, prob = O.forward(sd, inp, layer_names=layers, pos_offsets=offs, img_features=feats, grad=True)
        loss = O.silog_loss(torch.clip(pred, 1e-3), target, target > 1e-3, interpolate=True)
        loss.backward()
    finally:
        O.BN_TRAIN = False
    assert abs(float(loss.detach()) - float(z["loss"])) < 1e-5 * float(z["loss"])
    assert rel_l1(pred.detach()[:, :, ::4, ::4].numpy(), z["pred_slice"]) < 1e-5
    live = {k for k, v in sd.items() if getattr(v, "grad", None) is not None and float(v.grad.abs().max()) > 0}
    assert live == set(meta["names"])                                     # the same parameters get a gradient (the dead ones do not)
    gmax = float(z["grad_stats"][:, 3].max())
    for name, (mean, amean, rms, amax) in zip(meta["names"], z["grad_stats"]):
        g = sd[name].grad.double()
        assert abs(float((g * g).mean().sqrt()) - rms) <= 2e-2 * rms + 1e-6 * gmax, name       # f32 autograd on two builds of the same graph
    for name in meta["full"]:
        want = torch.from_numpy(z["grad." + name])
        got = sd[name].grad
        assert float((got - want).abs().max()) <= 2e-2 * max(float(want.abs().max()), 1e-5 * gmax), name
    assert np.allclose(sd["decoder.up1._net.1.running_mean"][:8].numpy(), meta["running"]["up1"], rtol=1e-4, atol=1e-6)
    assert np.allclose(sd["hist_encoder.hist_extractor1.pointnet_encoder.bn1.running_var"][:8].numpy(), meta["running"]["hist"], rtol=1e-4, atol=1e-6)
