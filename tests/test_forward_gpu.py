"""Whole-path parity on the GPU: HIP engine vs (a) goldens captured from the reference itself and
(b) the CPU oracle on the same seeded inputs.  Tolerance from BASELINE.json's north_star:
relative L1 <= 1e-3 on the predicted depth map (f32 parity mode); bf16 error is reported and
bounded separately."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import DECODER_CASES, calibrate_bn, case_inputs, load_case, rel_l1  # noqa: E402
from cfpnet_amd import spec, synthetic, weights  # noqa: E402
from cfpnet_amd.engine import Engine  # noqa: E402
from oracle import cfpnet_oracle as O  # noqa: E402

TOL_F32 = 1e-3      # north_star: "within 1e-3 relative L1 on the predicted depth map"
TOL_BF16 = 1e-2     # bf16 storage end to end: 1.5 x the 6.4e-3 measured on the benched batch (profiles/r2_precision_budget.md)
TOL_F16 = 1e-3      # fp16 storage end to end: THE north-star gate; measured 0.80e-3 on the benched batch, 0.63e-3 on the B=2 case


def make_engine(meta, sd, dtype):
    return Engine(sd, layer_names=meta["layer_names"], change_embedding=meta["change_embedding"],
                  no_skip_inside=meta["no_skip_inside"], dtype=dtype)


@pytest.mark.parametrize("name", DECODER_CASES)
def test_decoder_head_vs_reference_golden_f32(name):
    z, meta = load_case(name)
    sd, inp, feats, offs = case_inputs(meta)
    eng = make_engine(meta, sd, torch.float32)
    taps = {}
    edges, pred, prob = eng.forward(inp, img_features=feats, pos_offsets=offs, taps=taps)
    torch.cuda.synchronize()
    mine = pred.cpu().numpy() if meta["full_pred"] else pred[:, :, ::4, ::4].cpu().numpy()
    r = rel_l1(mine, z["pred"])
    print(f"{name}: pred relL1 vs reference = {r:.3e}")
    assert r < TOL_F32
    assert np.abs(edges.cpu().numpy() - z["bin_edges"]).max() < 1e-4
    assert np.abs(prob[:, :, ::16, ::16].float().cpu().numpy() - z["prob_slice"]).max() < 5e-3
    assert rel_l1(taps["unet_out"][:, ::8, ::8, ::8].numpy(), z["unet_slice"]) < TOL_F32
    for f in ("cross_atten1", "cross_atten2", "cross_atten3"):
        assert rel_l1(taps[f][:, ::4, ::4, ::4].numpy(), z[f + "_slice"]) < TOL_F32, f


def _full_case(B, H, W, zn, zpx, seed, drop):
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    inp = synthetic.make_inputs(B, H, W, zn, zpx, seed=seed, drop_hist=drop)
    return layers, sd, inp


@pytest.mark.parametrize("case", ["no_zone_valid", "one_image_without_zones"])
def test_batches_without_a_single_valid_tof_zone_vs_oracle(case):
    """The reference's edge of the ToF path: `mask` all False (every histogram dropped: fusion.py:120-131 then propagates nothing and the
    depth comes from the RGB path alone) for the whole batch, and for one image of a batch whose other image has all 64 zones -- the
    batch-reduced zone geometry is shared, the validity is per image.  Default numerics (f32x3), every image inside the 1e-3 gate."""
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers))
    if case == "no_zone_valid":
        inp = synthetic.make_inputs(2, 480, 640, 8, 56, seed=5, drop_hist=1.0)
        assert int(inp["additional"]["mask"].sum()) == 0
    else:
        inp = synthetic.make_inputs(2, 480, 640, 8, 56, seed=6)
        inp["additional"]["mask"][0] = False
        inp["additional"]["hist_data"][0] = 0
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers)
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
    e1, p1, pr1 = eng.forward(inp)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(p1).all()) and bool(torch.isfinite(pr1).all())
    for b in range(2):
        r = rel_l1(p1[b].cpu().numpy(), p0[b].numpy())
        print(f"  {case} image {b}: pred relL1 vs oracle = {r:.3e}")
        assert r < 1e-3, (case, b, r)
    assert torch.allclose(e1.cpu(), e0, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,H,W,zn,zpx,drop", [(1, 480, 640, 8, 56, 0.0), (2, 480, 640, 8, 56, 0.34), (1, 416, 544, 6, 64, 0.2)])
def test_full_model_vs_oracle_f32(B, H, W, zn, zpx, drop):
    layers, sd, inp = _full_case(B, H, W, zn, zpx, 7 + B, drop)
    offs = {"cross_atten3": (2, 3), "cross_atten2": (4, 7), "cross_atten1": (9, 11)} if H < 480 else None
    otaps, taps = {}, {}
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers, pos_offsets=offs, taps=otaps)
    eng = Engine(sd, layer_names=layers, dtype=torch.float32)
    e1, p1, pr1 = eng.forward(inp, pos_offsets=offs, taps=taps)
    torch.cuda.synchronize()
    for k in ("enc0", "enc1", "enc2", "enc3", "enc4", "hist2", "up1", "cross_atten3", "cross_atten1", "unet_out"):
        r = rel_l1(taps[k].numpy().reshape(otaps[k].shape), otaps[k].numpy())
        print(f"  {k}: relL1 {r:.3e}")
        assert r < TOL_F32, k
    r = rel_l1(p1.cpu().numpy(), p0.numpy())
    print(f"full model f32 B={B} {H}x{W}: pred relL1 vs oracle = {r:.3e}")
    assert r < TOL_F32
    assert torch.allclose(e1.cpu(), e0, rtol=1e-4, atol=1e-4)
    assert (pr1.float().cpu() - pr0).abs().max() < 5e-3
    assert p1.shape == (B, 1, H // 2, W // 2) and pr1.shape == (B, 256, H // 2, W // 2)


def test_config5_640x960_with_16x16_zones():
    """BASELINE.json configs[4]: 640x960 input, 16x16 ToF zones (40 px), fp16.  The reference cannot run it (its positional
    tables are hard-coded for 480x640); the same construction with tables sized for 640x960
    (`base_resolution`: windows of 7 / 10 / 14 tokens, 256 zones of 10x10 / 5x5 / 3x3 tokens, bilinear zone resampling
    at 2.5 tokens per zone) is checked HIP <-> CPU oracle: f32 parity mode to the north-star gate, fp16 storage bounded."""
    base = (640, 960)
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers, base_resolution=base))
    inp = synthetic.make_inputs(2, 640, 960, 16, 40, seed=31, drop_hist=0.2, image_hw=base)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    otaps, taps = {}, {}
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers, base_resolution=base, taps=otaps)
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, base_resolution=base)
    e1, p1, pr1 = eng.forward(inp, taps=taps)
    torch.cuda.synchronize()
    for k in ("enc4", "hist2", "cross_atten3", "cross_atten2", "cross_atten1", "unet_out"):
        r = rel_l1(taps[k].numpy().reshape(otaps[k].shape), otaps[k].numpy())
        print(f"  {k}: relL1 {r:.3e}")
        assert r < TOL_F32, k
    r = rel_l1(p1.cpu().numpy(), p0.numpy())
    print(f"config5 f32: pred relL1 vs oracle = {r:.3e}")
    assert r < TOL_F32 and p1.shape == (2, 1, 320, 480) and pr1.shape == (2, 256, 320, 480)
    assert torch.allclose(e1.cpu(), e0, rtol=1e-4, atol=1e-4)
    del eng
    e16 = Engine(sd, layer_names=layers, dtype=torch.float16, base_resolution=base)
    _, p2, _ = e16.forward(inp)
    e16.capture(inp)
    _, p3, _ = e16.replay()
    torch.cuda.synchronize()
    r16 = rel_l1(p2.cpu().numpy(), p0.numpy())
    print(f"config5 fp16: pred relL1 vs oracle = {r16:.3e}")
    assert r16 < TOL_F16 and torch.equal(p2, p3)
    del e16
    ex3 = Engine(sd, layer_names=layers, base_resolution=base)      # the default mode (float32 storage, f16x3 matrix math) at this shape
    assert ex3.x3
    _, p4, pr4 = ex3.forward(inp)
    ex3.capture(inp)
    _, p5, _ = ex3.replay()
    torch.cuda.synchronize()
    per_image = [rel_l1(p4[b].cpu().numpy(), p0[b].numpy()) for b in range(2)]
    print(f"config5 f32x3: pred relL1 vs oracle per image = {per_image}")
    assert max(per_image) < TOL_F32 and torch.equal(p4, p5) and pr4.dtype == torch.float32
    del ex3
    with pytest.raises(Exception):                       # the 480x640 tables cannot hold a 640x960 map: loud, like the reference
        Engine(weights.make_torch_state_dict(spec.model_manifest(layers)), layer_names=layers, dtype=torch.float32).forward(inp)


@pytest.mark.parametrize("dtype,bound", [(torch.bfloat16, TOL_BF16), (torch.float16, TOL_F16)])
def test_full_model_16bit_error_is_bounded(dtype, bound):
    """16-bit storage (f32 accumulation): the error is storage rounding accumulated over ~90 layers.  fp16 (10 mantissa
    bits) sits at the north-star gate of 1e-3 relative L1; bf16 (7 bits) is 8x coarser."""
    layers, sd, inp = _full_case(2, 480, 640, 8, 56, 21, 0.2)
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers)
    eng = Engine(sd, layer_names=layers, dtype=dtype)
    e1, p1, pr1 = eng.forward(inp)
    torch.cuda.synchronize()
    r = rel_l1(p1.cpu().numpy(), p0.numpy())
    abs_rel = float(np.mean(np.abs(p0.numpy() - p1.cpu().numpy()) / p0.numpy()))
    print(f"{dtype} full model: pred relL1 {r:.3e}, abs_rel {abs_rel:.3e}")
    assert r < bound
    assert pr1.dtype == dtype and abs(float(pr1.float().sum(1).mean()) - 1.0) < 2e-2
    # no prob requested -> identical pred
    e2, p2, pr2 = eng.forward(inp, return_prob=False)
    assert pr2 is None and torch.equal(p1, p2)


_B8 = {}


def _b8_case():
    """BASELINE.json configs[1] exactly as bench.py runs it: batch 8, 480x640, 8x8 zones of 56 px, the bench's own seed;
    the CPU oracle's forward of the 8 maps is computed once for the three dtype cases (8 x ~0.4 s on the GPU box)."""
    if not _B8:
        layers = spec.COMBINE1_LAYERS
        sd = weights.make_torch_state_dict(spec.model_manifest(layers))
        inp = synthetic.make_inputs(8, 480, 640, 8, 56, seed=synthetic.SEED)
        torch.set_num_threads(max(torch.get_num_threads(), 8))
        e0, p0, pr0 = O.forward(sd, inp, layer_names=layers)
        _B8.update(layers=layers, sd=sd, inp=inp, e0=e0, p0=p0, pr0=pr0[:, :, ::8, ::8].clone())
    return _B8


# measured on MI355X (profiles/r2_precision_taps.txt): f32 2e-6, fp16 and bf16 see TOL_* above
@pytest.mark.parametrize("dtype,bound", [(torch.float32, TOL_F32), (torch.float16, TOL_F16), (torch.bfloat16, TOL_BF16)])
def test_config1_batch8_as_benched_vs_oracle(dtype, bound):
    """The benched configuration (BASELINE.json configs[1]: batch 8, 480x640) in every storage mode against the CPU oracle,
    through the path bench.py times: the captured HIP graph (tile plans depend on M, so batch 8 picks other kernel variants
    than the batch-1/2 cases above), eager and replayed results bit-identical, every image of the batch checked."""
    c = _b8_case()
    eng = Engine(c["sd"], layer_names=c["layers"], dtype=dtype)
    dinp = synthetic.to_device(c["inp"], "cuda:0")
    e1, p1, pr1 = eng.forward(dinp)
    p1, e1, pr1 = p1.clone(), e1.clone(), pr1[:, :, ::8, ::8].float().cpu()
    eng.capture(dinp)
    e2, p2, pr2 = eng.replay()
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(e1, e2)
    per_image = [rel_l1(p2[b].cpu().numpy(), c["p0"][b].numpy()) for b in range(8)]
    r = rel_l1(p2.cpu().numpy(), c["p0"].numpy())
    abs_rel = float(np.mean(np.abs(c["p0"].numpy() - p2.cpu().numpy()) / c["p0"].numpy()))
    print(f"configs[1] B=8 {dtype}: pred relL1 {r:.3e} (per image max {max(per_image):.3e}), abs_rel {abs_rel:.3e}")
    assert r < bound and max(per_image) < 1.5 * bound
    assert torch.allclose(e2.cpu(), c["e0"], rtol=2e-3 if dtype != torch.float32 else 1e-4, atol=1e-3 if dtype != torch.float32 else 1e-4)
    assert (pr1 - c["pr0"]).abs().max() < (5e-3 if dtype == torch.float32 else 6e-2)
    assert p2.shape == (8, 1, 240, 320)


# measured on MI355X, round 3 (tools/precision_family.py, gpurun_out/r3j): uniform 0.68-0.85e-3; kaiming 0.72-1.08e-3 -- the reference's
# own initialisation sits AT the gate: that network amplifies ONE fp16 rounding of its input image alone to 1.9e-4 (uniform family:
# 0.5e-4), and an inference engine with 16-bit storage makes ~150 of them; the bound below is the gate plus that measured spread,
# and the bench line says `gate_met` per family.  kaiming_peaked (a confident head on the same network) is not a gate at all: rounding
# only the INPUT IMAGE to fp16, everything else float32, already moves its prediction by 2.4e-3 (test below).


@pytest.mark.parametrize("family,bound", [("uniform", TOL_F16), ("kaiming", 1.2e-3)], ids=["uniform_at_the_1e-3_gate", "kaiming_bounded_at_1.2e-3_OUTSIDE_the_gate"])
def test_fp16_speed_mode_error_bound_on_three_seeds_of_two_weight_families(family, bound):
    """fp16 storage is an OPT-IN SPEED MODE, not the compliant one (that is f32x3: next test, all four families at 1e-3).  What this test
    asserts is exactly its parametrisation: configs[1] (batch 8, 480x640) through the captured graph, three input seeds, every image --
    uniform family inside the north-star gate (<= 1e-3); the reference's own initialisation (kaiming-normal fan_out, trunc_normal(0.2)
    positional tables, deltar.py:23-32, fusion.py:22-23, BatchNorm statistics calibrated) bounded at 1.2e-3, i.e. OUTSIDE the gate by up
    to 20 % (measured 0.72-1.08e-3).  The confident-head and trained-like families are not asserted for fp16 at all (1.1e-2 / 8e-5
    measured; `test_ill_conditioned_network_is_reported_not_gated`, bench line `f32x3.families.*.f16`)."""
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=family)
    if family != "uniform":
        sd = calibrate_bn(sd, layers)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    eng = Engine(sd, layer_names=layers, dtype=torch.float16)
    worst = 0.0
    for si, seed in enumerate((synthetic.SEED, 4242, 97)):
        inp = synthetic.make_inputs(8, 480, 640, 8, 56, seed=seed, drop_hist=(0.0, 0.34, 0.1)[si])
        _, p0, _ = O.forward(sd, inp, layer_names=layers)
        dinp = synthetic.to_device(inp, "cuda:0")
        if si == 0:
            eng.capture(dinp)
        _, p1, _ = eng.replay(dinp)
        torch.cuda.synchronize()
        per_image = [rel_l1(p1[b].cpu().numpy(), p0[b].numpy()) for b in range(8)]
        print(f"fp16 {family} seed {seed}: rel-L1 {rel_l1(p1.cpu().numpy(), p0.numpy()):.3e}, worst image {max(per_image):.3e}, "
              f"pred std {float(p0.std()):.3f}")
        worst = max(worst, max(per_image))
        assert max(per_image) <= bound, (family, seed, per_image)
    print(f"fp16 {family}: worst image of 24 = {worst:.3e}")


_TRAINED = {}


def trained_like(layers):
    """One trained-like state dict per test session (300 real optimisation steps on the GPU, weights.trained_like_state_dict)."""
    if "sd" not in _TRAINED:
        sd = weights.trained_like_state_dict(layers, steps=300)
        print(f"trained-like family: SILog after 300 steps {sd.pop('__loss__'):.4f}")
        _TRAINED["sd"] = sd
    return _TRAINED["sd"]


@pytest.mark.parametrize("family", ["uniform", "kaiming", "kaiming_peaked", "trained"])
def test_f32x3_meets_the_gate_on_every_image_of_every_weight_family(family):
    """THE parity gate of the default mode (float32 storage, f16x3 matrix math: Engine(dtype=float32, x3=True), what `Deltar` and
    `make_model` build unless told otherwise): configs[1] -- batch 8, 480x640, 8x8 zones -- through the captured graph, two input seeds
    (with and without dropped zones) x four weight families, relative L1 of the depth map against the CPU oracle <= 1e-3 (north_star)
    on EVERY image, no family-specific bound:
      uniform         the key-addressed family of the golden fixtures;
      kaiming         the reference's own initialisation (deltar.py:23-32, fusion.py:22-23) with calibrated BatchNorm statistics;
      kaiming_peaked  the same with a confident head (conv_out x 6) -- the family on which every 16-bit STORAGE mode is outside the gate
                      (one fp16 rounding of the input image alone moves it by 2.4e-3, test below);
      trained         300 real optimisation steps from the reference's initialisation (weights.trained_like_state_dict): the closest
                      offline stand-in for the authors' checkpoint."""
    layers = spec.COMBINE1_LAYERS
    if family == "trained":
        sd = trained_like(layers)
    else:
        sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=family)
        if family != "uniform":
            sd = calibrate_bn(sd, layers)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
    worst = 0.0
    for si, seed in enumerate((synthetic.SEED, 4242)):
        inp = synthetic.make_inputs(8, 480, 640, 8, 56, seed=seed, drop_hist=(0.0, 0.34)[si])
        _, p0, pr0 = O.forward(sd, inp, layer_names=layers)
        dinp = synthetic.to_device(inp, "cuda:0")
        if si == 0:
            eng.capture(dinp)
        _, p1, _ = eng.replay(dinp)
        torch.cuda.synchronize()
        per_image = [rel_l1(p1[b].cpu().numpy(), p0[b].numpy()) for b in range(8)]
        print(f"f32x3 {family} seed {seed}: rel-L1 {rel_l1(p1.cpu().numpy(), p0.numpy()):.3e}, worst image {max(per_image):.3e}, "
              f"pred std {float(p0.std()):.3f}, mean max-prob {float(pr0.max(1)[0].mean()):.3f}")
        worst = max(worst, max(per_image))
        assert max(per_image) <= TOL_F32, (family, seed, per_image)
    print(f"f32x3 {family}: worst image of 16 = {worst:.3e}")


@pytest.mark.parametrize("family", ["uniform", "kaiming_peaked"])
def test_f32x3_single_image_plans_meet_the_gate(family):
    """A SINGLE image takes plans of its own in the default mode (round 5: 32 x 64 GEMM tiles up to 4 800 rows, real K splits for the long-K
    low-resolution convs, the head conv through the 64-channel chunk tile, one-wave workgroups in the fused tails, attention outputs split over
    lanes, LayerNorm inside the split-K reduce) that no batch-8 test reaches: 480x640 and a 416x544 crop with a positional window, eager and
    through the captured graph, against the CPU oracle, inside the 1e-3 gate."""
    layers = spec.COMBINE1_LAYERS
    sd = weights.make_torch_state_dict(spec.model_manifest(layers), family=family)
    if family != "uniform":
        sd = calibrate_bn(sd, layers)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
    for (H, W, zn, zpx, drop, offs) in ((480, 640, 8, 56, 0.2, None), (416, 544, 6, 64, 0.0, {"cross_atten3": (2, 3), "cross_atten2": (4, 7), "cross_atten1": (9, 11)})):
        inp = synthetic.make_inputs(1, H, W, zn, zpx, seed=77, drop_hist=drop)
        _, p0, pr0 = O.forward(sd, inp, layer_names=layers, pos_offsets=offs)
        dinp = synthetic.to_device(inp, "cuda:0")
        _, p1, pr1 = eng.forward(dinp, pos_offsets=offs)
        torch.cuda.synchronize()
        r = rel_l1(p1.cpu().numpy(), p0.numpy())
        print(f"f32x3 single image {family} {H}x{W}: eager rel-L1 {r:.3e}")
        assert r <= TOL_F32 and bool(torch.isfinite(pr1).all()), (family, H, W, r)
        eng.capture(dinp, pos_offsets=offs)
        _, p2, _ = eng.replay(dinp)
        torch.cuda.synchronize()
        assert torch.equal(p1, p2), "the captured graph must reproduce the eager forward bit for bit"


def test_ill_conditioned_network_is_reported_not_gated():
    """The reference's initialisation with a CONFIDENT head (conv_out x 6: a peaked 256-way softmax like a trained model's): the float32
    engine with nothing but its INPUT IMAGE rounded once to fp16 already differs from itself by > 1e-3 -- no 16-bit storage format can
    meet the north-star gate on this network, whatever the kernels do.  What is asserted is that the fp16 engine's error stays within a
    small multiple of that single-rounding sensitivity (measured: 1.2e-2 against 2.4e-3, x5; ~150 tensors are rounded on the way)."""
    layers = spec.COMBINE1_LAYERS
    sd = calibrate_bn(weights.make_torch_state_dict(spec.model_manifest(layers), family="kaiming_peaked"), layers)
    inp = synthetic.to_device(synthetic.make_inputs(4, 480, 640, 8, 56, seed=97), "cuda:0")
    e32 = Engine(sd, layer_names=layers, dtype=torch.float32)
    p32 = e32.forward(inp)[1].clone()
    pert = {"rgb": inp["rgb"].to(torch.float16).float(), "additional": inp["additional"]}
    sens = rel_l1(e32.forward(pert)[1].cpu().numpy(), p32.cpu().numpy())
    del e32
    p16 = Engine(sd, layer_names=layers, dtype=torch.float16).forward(inp)[1]
    err = rel_l1(p16.cpu().numpy(), p32.cpu().numpy())
    print(f"kaiming_peaked: input-rounding sensitivity {sens:.3e}, fp16 engine {err:.3e} = {err / sens:.1f} x")
    assert sens > 1e-3                      # the network itself is outside the gate for a single fp16 rounding
    assert err < 12 * sens


@pytest.mark.parametrize("env", [{"CFP_MBCONV_FUSED": "1"}, {"CFP_WEIGHTS2": "1"}, {"CFP_HEAD_FUSED": "0"}, {"CFP_HEAD_HILO": "10"}],
                         ids=["mbconv_fused", "two_term_weights", "separate_head_kernels", "wout_hilo"])
def test_optional_kernel_paths_keep_parity(env, monkeypatch):
    """The paths that are measured-and-not-default (fused expand->depthwise kernel, two-term pointwise weights, hi+lo conv_out
    weights) and the round-1 head kernels stay correct: fp16 engine against the CPU oracle at the gate, on the B=2 case."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    layers, sd, inp = _full_case(2, 480, 640, 8, 56, 21, 0.2)
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers)
    eng = Engine(sd, layer_names=layers, dtype=torch.float16)
    e1, p1, pr1 = eng.forward(inp)
    torch.cuda.synchronize()
    r = rel_l1(p1.cpu().numpy(), p0.numpy())
    print(f"{env}: fp16 pred relL1 {r:.3e}")
    assert r < TOL_F16


def test_optional_two_source_chunk_kernel_keeps_parity_in_the_default_mode(monkeypatch):
    """CFP_UP_FUSED_X3=1234 (measured slower, off by default): every decoder stage's first conv through cfp_upsample_cat_conv3x3 in the f16x3
    mode -- no resize launch, no materialised upsampled tensor -- stays inside the gate against the CPU oracle and within 1e-5 of the default
    plan's depth map."""
    layers, sd, inp = _full_case(2, 480, 640, 8, 56, 21, 0.2)
    e0, p0, pr0 = O.forward(sd, inp, layer_names=layers)
    ref = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True).forward(inp)[1].clone()
    monkeypatch.setenv("CFP_UP_FUSED_X3", "1234")
    eng = Engine(sd, layer_names=layers, dtype=torch.float32, x3=True)
    assert "decoder.up4.a.wcat" in eng.P
    p1 = eng.forward(inp)[1]
    torch.cuda.synchronize()
    r = rel_l1(p1.cpu().numpy(), p0.numpy())
    print(f"two-source chunk kernel at up1-4: pred relL1 vs oracle {r:.3e}, vs the default plan {rel_l1(p1.cpu().numpy(), ref.cpu().numpy()):.3e}")
    assert r < TOL_F32 and rel_l1(p1.cpu().numpy(), ref.cpu().numpy()) < 1e-5


def test_forward_is_deterministic_and_batch_independent():
    """Size-independent properties: same input -> identical bits; a sample's result does not depend
    on what else is in the batch (all-valid zones, so the batch-reduced geometry is shared) beyond
    f32 re-association (the reduction split counts scale with the batch)."""
    layers, sd, inp = _full_case(2, 480, 640, 8, 56, 33, 0.0)
    eng = Engine(sd, layer_names=layers, dtype=torch.float32)
    _, p1, _ = eng.forward(inp, return_prob=False)
    _, p2, _ = eng.forward(inp, return_prob=False)
    assert torch.equal(p1, p2)
    one = {"rgb": inp["rgb"][1:2], "additional": {
        "hist_data": inp["additional"]["hist_data"][1:2], "rect_data": inp["additional"]["rect_data"][1:2],
        "mask": inp["additional"]["mask"][1:2],
        "patch_info": {**{s: {k: v[1:2] for k, v in inp["additional"]["patch_info"][s].items()} for s in (4, 8, 16)},
                       "zone_num": inp["additional"]["patch_info"]["zone_num"][1:2]}}}
    _, p3, _ = eng.forward(one, return_prob=False)
    assert rel_l1(p3[0].cpu().numpy(), p1[1].cpu().numpy()) < 1e-5


def test_deltar_module_boundary():
    """make_model(args) -> forward(input_data) -> (edges, pred, prob, None); state_dict round trip."""
    from cfpnet_amd import config
    from cfpnet_amd.deltar import make_model
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = config.parse_args(["@" + os.path.join(root, "configs", "cfpnet_combine1.txt")])
    model = make_model(args).eval()
    sd = model.state_dict()
    assert set(sd) == {k for k, _, _ in spec.model_manifest(spec.COMBINE1_LAYERS)}
    inp = synthetic.to_device(synthetic.make_inputs(1), "cuda:0")
    out = model(inp)
    assert len(out) == 4 and out[3] is None
    edges, pred, prob, _ = out
    assert edges.shape == (1, 257) and pred.shape == (1, 1, 240, 320) and prob.shape == (1, 256, 240, 320)
    assert pred.is_cuda and float(pred.min()) > 1e-3 and float(pred.max()) < 10
    assert prob.dtype == torch.float32 and pred.dtype == torch.float32 and edges.dtype == torch.float32     # the reference's types (deltar.py:64-67)
    model.train()
    with pytest.raises(RuntimeError, match="no CPU path"):      # parameters still on the host: the training step refuses, it does not fall back
        model(inp)
    ones = [p for p in model.get_1x_lr_params()]
    tens = [p for p in model.get_10x_lr_params()]
    assert len(ones) + len(tens) == len(list(model.parameters()))


def _boundary_model():
    from cfpnet_amd import config
    from cfpnet_amd.deltar import make_model
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = config.parse_args(["@" + os.path.join(root, "configs", "cfpnet_combine1.txt")])
    return make_model(args).eval().to("cuda:0")


def test_default_boundary_mode_is_the_compliant_one():
    """`make_model(args)` with no dtype builds the f32x3 engine: float32 storage, split-precision matrix math, float32 `prob` straight
    from the kernels (no cast at the boundary), results equal to the eager Engine in that mode and inside the gate against the oracle."""
    model = _boundary_model()
    assert model.x3 and model.compute_dtype == torch.float32
    inp = synthetic.make_inputs(2, seed=11, drop_hist=0.2)
    dinp = synthetic.to_device(inp, "cuda:0")
    edges, pred, prob, _ = model(dinp)
    eng = model.engine("cuda:0")
    assert eng.x3 and eng.dtype == torch.float32 and prob.dtype == torch.float32
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    _, p0, pr0 = O.forward(sd, inp, layer_names=spec.COMBINE1_LAYERS)
    per_image = [rel_l1(pred[b].cpu().numpy(), p0[b].numpy()) for b in range(2)]
    print(f"default boundary mode vs oracle: {per_image}")
    assert max(per_image) <= TOL_F32
    assert float((prob.cpu() - pr0).abs().max()) < 1e-4


def test_eval_forward_replays_graphs_with_the_reference_semantics():
    """`model(input_data)` in eval mode is a graph replay: (a) other tensors of the same geometry give that input's result (private-copy
    graph), (b) the same tensors again -- contents changed IN PLACE -- give the new contents' result (the adopted-input graph reads the
    caller's tensors), (c) a result stays valid while the next forward runs (output ring of two), (d) all equal the eager engine."""
    model = _boundary_model()
    model.eval_static_outputs = True            # the opt-in ring semantics (the default clones: next test)
    ref = _boundary_model()
    ref.eval_graphs = False
    a = synthetic.to_device(synthetic.make_inputs(2, seed=1), "cuda:0")
    b = synthetic.to_device(synthetic.make_inputs(2, seed=2, drop_hist=0.3), "cuda:0")
    want_a, want_b = ref(a)[1].clone(), ref(b)[1].clone()
    pa = model(a)[1]
    pb = model(b)[1]
    torch.cuda.synchronize()
    assert torch.equal(pa, want_a) and torch.equal(pb, want_b)          # (a) + (c): pa still holds a's result after b's forward
    pb2 = model(b)[1]                                                     # second consecutive call with b's tensors: adopted-input graph
    pb3 = model(b)[1]
    torch.cuda.synchronize()
    assert torch.equal(pb2, want_b) and torch.equal(pb3, want_b)
    b["rgb"].copy_(a["rgb"]); b["additional"]["hist_data"].copy_(a["additional"]["hist_data"]); b["additional"]["mask"].copy_(a["additional"]["mask"])
    pb4 = model(b)[1]                                                     # (b): same tensors, new contents, no re-capture
    torch.cuda.synchronize()
    assert torch.equal(pb4, want_a)
    assert len(model._eval_caps) == 1


def test_eval_forward_returns_fresh_tensors_by_default_like_the_reference():
    """`preds.append(model(x)[1])` over a dataset must keep every result (the reference allocates its outputs per call,
    deltar.py:64-67).  By default the graph's static outputs are cloned at the boundary; five results held across five more forwards."""
    model = _boundary_model()
    ref = _boundary_model()
    ref.eval_graphs = False
    xs = [synthetic.to_device(synthetic.make_inputs(1, seed=30 + i, drop_hist=0.1 * i), "cuda:0") for i in range(5)]
    held = [model(x) for x in xs]
    torch.cuda.synchronize()
    ptrs = {o[i].data_ptr() for o in held for i in range(3)}
    assert len(ptrs) == 15                                        # no two results share storage
    for x, (e, p, pr, _) in zip(xs, held):
        we, wp, wpr, _ = ref(x)
        assert torch.equal(p, wp) and torch.equal(e, we) and torch.equal(pr, wpr)


def test_check_finite_turns_the_f32x3_range_limit_into_a_clean_error():
    """inputs far outside the two-half range of the f16x3 split give a non-finite map; with `model.check_finite` the boundary
    raises instead of returning it, and the float32 mode of the same module still serves the input."""
    model = _boundary_model()
    model.check_finite = True
    inp = synthetic.to_device(synthetic.make_inputs(1, seed=3), "cuda:0")
    model(inp)                                                           # ordinary input: fine
    big = {"rgb": inp["rgb"] * 1e7, "additional": inp["additional"]}
    with pytest.raises(FloatingPointError, match="two IEEE halves"):
        model(big)


def test_eval_graph_serves_every_random_positional_window_from_one_capture():
    """below the table size (416x544 crops) the reference draws a new positional window per forward (fusion.py:87-91).  The
    window reaches the captured graph through a device buffer, so the cache holds ONE entry and every forward equals the eager engine
    run with the same window."""
    model = _boundary_model()
    H, W = 416, 544
    inp = synthetic.to_device(synthetic.make_inputs(2, H, W, 6, 64, seed=8, drop_hist=0.2), "cuda:0")
    eng = model.engine("cuda:0")
    seen = set()
    for i in range(5):
        torch.manual_seed(100 + i)
        offs = model.draw_pos_offsets(H, W)
        seen.add(tuple(sorted(offs.items())))
        torch.manual_seed(100 + i)
        _, pred, _, _ = model(inp)                               # draws the same window itself
        _, want, _ = eng.forward(inp, pos_offsets=offs)
        torch.cuda.synchronize()
        assert torch.equal(pred, want), (i, offs)
    assert len(seen) > 1 and len(model._eval_caps) == 1


def test_eval_forward_launches_no_torch_kernel_and_costs_what_the_engine_costs():
    """The call the reference times (evaluate_time.py:73-82) is `model(input_data)`.  Called again with the same device tensors it must be
    host logic + one graph launch: no ATen operator runs (torch.profiler sees none), and its latency at batch 8 is within 5 % of a bare
    `Engine.replay()` of the same mode."""
    import time
    from torch.profiler import profile, ProfilerActivity
    model = _boundary_model()
    model.eval_static_outputs = True            # the latency loop's opt-in: results are the ring's own buffers (no clone kernels)
    inp = synthetic.to_device(synthetic.make_inputs(8), "cuda:0")
    with torch.no_grad():
        for _ in range(6):
            out = model(inp)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(3):
                model(inp)
        torch.cuda.synchronize()
        ops_seen = sorted({e.name for e in prof.events() if e.name.startswith("aten::")})
        print("ATen operators in a steady-state eval forward:", ops_seen)
        assert ops_seen == [], ops_seen

        def timed(fn, n=60):
            for _ in range(10):
                fn()
            ts = []
            for _ in range(n):
                torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            return sum(ts[1:-2]) / (n - 3) * 1e3
        t_model = timed(lambda: model(inp))
        eng = Engine({k: v.detach().cpu() for k, v in model.state_dict().items()}, layer_names=model.layer_names, dtype=torch.float32, x3=True,
                     change_embedding=model.change_embedding, no_skip_inside=model.no_skip_inside)
        eng.capture(inp)
        t_eng = timed(lambda: eng.replay())
    print(f"latency at batch 8: model(input_data) {t_model:.3f} ms, Engine.replay {t_eng:.3f} ms")
    assert t_model <= 1.05 * t_eng + 0.02


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)])
def test_batch_lanes_match_single_stream(dtype, tol):
    """forward_lanes (sub-batches on concurrent streams, whole-batch zone geometry) == forward, eager and captured."""
    layers, sd, inp = _full_case(4, 256, 320, 3, 64, 21, 0.25)
    eng = Engine(sd, layer_names=layers, dtype=dtype)
    dinp = synthetic.to_device(inp, "cuda:0")
    e0, p0, pr0 = eng.forward(dinp)
    torch.cuda.synchronize()
    for lanes in (2, 4):
        e1, p1, pr1 = eng.forward_lanes(dinp, lanes)
        torch.cuda.synchronize()
        assert rel_l1(p1.cpu().numpy(), p0.cpu().numpy()) < tol, lanes
        assert np.abs(e1.cpu().numpy() - e0.cpu().numpy()).max() < 1e-4 + tol
        assert rel_l1(pr1.float().cpu().numpy(), pr0.float().cpu().numpy()) < 5 * tol + 1e-4
    eng.capture(dinp, lanes=2)
    e2, p2, pr2 = eng.replay()
    torch.cuda.synchronize()
    assert rel_l1(p2.cpu().numpy(), p0.cpu().numpy()) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-6)])
def test_batches_in_flight_match_forward(dtype, tol):
    """capture(inflight=n): n whole batches in flight on concurrently scheduled streams.  Every result equals the plain
    forward of the input that was fed to that slot (different inputs per step, slots reused twice)."""
    layers, sd, _ = _full_case(2, 256, 320, 3, 64, 21, 0.0)
    eng = Engine(sd, layer_names=layers, dtype=dtype)
    inps = [synthetic.to_device(synthetic.make_inputs(2, 256, 320, 3, 64, seed=40 + i, drop_hist=0.2 * (i % 2)), "cuda:0") for i in range(7)]
    want = []
    eng.plan_mode(True)                             # the plan the in-flight slots are captured with (it picks other tiles: summation order)
    try:
        for x in inps:
            e, p, pr = eng.forward(x)
            want.append((e.clone(), p.clone(), pr.clone()))
        torch.cuda.synchronize()
    finally:
        eng.plan_mode(False)
    eng.capture(inps[0], inflight=3)
    assert 1 <= len(eng._slots) <= 3
    got = []
    for x in inps:
        (e, p, pr), ev = eng.replay_async(x)
        got.append((e, p, pr, ev))
        if len(got) >= len(eng._slots):            # consume the oldest before its slot is reused
            e0, p0, pr0, ev0 = got[len(got) - len(eng._slots)]
            ev0.synchronize()
            got[len(got) - len(eng._slots)] = (e0.clone(), p0.clone(), pr0.clone(), None)
    torch.cuda.synchronize()
    for i, (w, g) in enumerate(zip(want, got)):
        assert rel_l1(g[1].cpu().numpy(), w[1].cpu().numpy()) <= tol, i
        assert torch.equal(g[0], w[0]) or rel_l1(g[0].cpu().numpy(), w[0].cpu().numpy()) <= tol
        assert rel_l1(g[2].float().cpu().numpy(), w[2].float().cpu().numpy()) <= 10 * tol + 1e-7
    # the blocking call still works in this mode
    e, p, pr = eng.replay(inps[3])
    torch.cuda.synchronize()
    assert rel_l1(p.cpu().numpy(), want[3][1].cpu().numpy()) <= tol


@pytest.mark.parametrize("mode", ["f16", "f32x3"])
def test_in_flight_stress_is_bit_exact(mode):
    """(f32x3: the default mode's kernels -- f16x3 GEMMs with the fragment-pipelined loop, halo / chunk 3x3, fused tails, Toeplitz
    depthwise, fused bin head; round 4 found a hand-over race in its fused tails with exactly this kind of test.)
    Regression guard for the stage hand-over race round 2 found in the fused head (one wave's 32 pixels wrong about once per 640
    forwards, ONLY with several captured forwards in flight and DIFFERENT inputs per slot; profiles/r2_inflight_race.txt): 300 rounds
    of 8 different inputs through 4 in-flight slots = 2 400 forwards, every result bit-compared with the eager forward of the
    same input.  The kernels that stage through LDS-DMA (gen-2 GEMM, LoFTR / LKPM tails, fused head) all run in it."""
    layers, sd, _ = _full_case(2, 256, 320, 3, 64, 21, 0.0)
    eng = Engine(sd, layer_names=layers, dtype=torch.float16) if mode == "f16" else Engine(sd, layer_names=layers)
    inps = [synthetic.to_device(synthetic.make_inputs(2, 256, 320, 3, 64, seed=40 + i, drop_hist=0.2 * (i % 2)), "cuda:0") for i in range(8)]
    eng.plan_mode(True)                             # the kernel plan the in-flight slots are captured with (tile choices differ: summation order)
    try:
        want = [tuple(t.clone() for t in eng.forward(x)) for x in inps]
        torch.cuda.synchronize()
    finally:
        eng.plan_mode(False)
    eng.capture(inps[0], inflight=4)
    n = len(eng._slots)
    bad = []
    for r in range(300):
        got = []
        for i, x in enumerate(inps):
            (e, p, pr), ev = eng.replay_async(x)
            got.append((p, pr, ev))
            if len(got) >= n:                      # consume the oldest before its slot comes round again
                j = len(got) - n
                got[j][2].synchronize()
                got[j] = (got[j][0].clone(), got[j][1].clone(), None)
        torch.cuda.synchronize()
        for i, (p, pr, _) in enumerate(got):
            if not (torch.equal(p, want[i][1]) and torch.equal(pr, want[i][2])):
                bad.append((r, i))
    assert not bad, f"{len(bad)} of 2400 in-flight forwards differ from the eager result: {bad[:8]}"
