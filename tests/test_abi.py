"""The C-ABI shared library builds, loads and exports every symbol include/cfpnet_hip.h declares
(no kernel is launched: this runs without a GPU), and argument validation reports errors through
return codes + cfp_last_error() instead of faulting."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from cfpnet_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return hip.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cfpnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from cfpnet_amd import hip
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cfpnet_hip.h but not exported"
    assert set(names) == set(hip.SIGNATURES), set(names) ^ set(hip.SIGNATURES)
    assert lib.cfp_version() >= 100


def test_validation_errors_are_reported_not_faulted(lib):
    from cfpnet_amd import hip
    # null pointers / bad dtype / bad shapes must come back as negative codes with a message
    rc = lib.cfp_conv2d_nhwc(0, 8, 0, 0, 0, 0, 0, 0, 8, 1, 4, 4, 8, 8, 3, 3, 1, 1, 1, 4, 4, 0, hip.BF16, 0, 0, 0)
    assert rc == -1 and "null" in hip.last_error()
    rc = lib.cfp_layernorm(16, 24, 16, 16, 1e-5, 0, 0, 16, 24, 4, 24, hip.BF16, 0)    # 24/8 = 3 lanes: not a power of two
    assert rc == -2 and "power of two" in hip.last_error()
    rc = lib.cfp_attn_kv_reduce(16, 32, 16, 32, 16, 16, 0, 1, 1, 16, 1, 16, 0, 1, 0, 16, 0, 16.0, 4, 5, hip.F32, 0)
    assert rc == -2 and "head dim" in hip.last_error()
    rc = lib.cfp_dwconv_large_nhwc(16, 32, 16, 16, 16, 16, 32, 1, 8, 8, 32, 8, 0, hip.F32, 0)   # even kernel
    assert rc == -2
    with pytest.raises(RuntimeError, match="cfp_bin_softmax failed"):
        hip.call("cfp_bin_softmax", 16, 256, 16, 0, 16, 1, 64, 100, hip.F32, 0)
    assert lib.cfp_attn_kv_ws_floats(0, 1, 1, 1, 1, 1, 8) == 0
    assert lib.cfp_conv2d_variant(614400, 128) == 3 and lib.cfp_conv2d_variant(100, 16) == 0
    assert lib.cfp_conv2d_ws_bytes(614400, 128, 1152, hip.BF16) == 0 and lib.cfp_conv2d_ws_bytes(240, 128, 4608, hip.BF16) > 0


def test_product_path_has_no_cpu_fallback():
    """Engine refuses to run without a GPU (and never imports the oracle)."""
    import torch
    import cfpnet_amd.engine as E
    src = open(E.__file__).read() + open(os.path.join(ROOT, "cfpnet_amd", "deltar.py")).read() + \
        open(os.path.join(ROOT, "cfpnet_amd", "ops.py")).read()
    assert "oracle" not in src.replace("CPU oracle", "")
    if not torch.cuda.is_available():
        from cfpnet_amd import spec, weights
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            E.Engine(weights.make_torch_state_dict(spec.model_manifest()), layer_names=spec.COMBINE1_LAYERS)


def test_deltar_container_on_cpu():
    """State-dict contract and parameter groups need no GPU."""
    import torch
    from cfpnet_amd import config, spec
    from cfpnet_amd.deltar import make_model
    args = config.parse_args(["@" + os.path.join(ROOT, "configs", "cfpnet_combine1.txt")])
    m = make_model(args)
    sd = m.state_dict()
    man = spec.model_manifest(spec.COMBINE1_LAYERS)
    assert list(sd.keys()) == [k for k, _, _ in man] or set(sd) == {k for k, _, _ in man}
    for k, shape, _ in man:
        assert tuple(sd[k].shape) == tuple(shape), k
    # checkpoint round trip incl. the DataParallel 'module.' prefix (model_io.py:47-52)
    from cfpnet_amd.model_io import load_checkpoint, save_checkpoint
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "w", "ck.pt")
        wrapped = torch.nn.DataParallel(m) if False else None
        torch.save({"model": {"module." + k: v for k, v in sd.items()}, "epoch": 3}, os.path.join(d, "ck.pt"))
        m2 = make_model(args)
        with torch.no_grad():
            for p in m2.parameters():
                p.zero_()
        m2, _, ep = load_checkpoint(os.path.join(d, "ck.pt"), m2)
        assert ep == 3 and all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), sd.values()))
        save_checkpoint(m, torch.optim.AdamW(m.parameters()), 1, f)
        assert os.path.exists(f)
    n1 = sum(p.numel() for p in m.get_1x_lr_params())
    n10 = sum(p.numel() for p in m.get_10x_lr_params())
    assert n1 == 12_461_982 and n1 + n10 == spec.param_count(man)     # hist_encoder_10x: hist params at full lr
    with pytest.raises(NotImplementedError):
        args.attention_layer = ["hist2image", "bogus"]
        make_model(args)


def test_float32_depthwise_slot_count_is_the_same_with_and_without_the_input_extent():
    """Round 5 bug: `cfp_dwconv3x3_strips` (asked before the launch, without H / W) and the launch's own plan used slightly different byte
    counts in their time model, chose different run lengths at batch 4 and the kernel wrote more `partial` slots than the caller had
    allocated (a GPU memory fault in tools/probes/x3_b4_check.py).  The plan now depends on the output extent only; this sweep pins it for
    every encoder shape x batch 1..32 x both paddings of a SAME-padded input, and for odd sizes."""
    from cfpnet_amd import hip
    lib = hip.load()
    n = 0
    for (H, W, C, s) in [(60, 80, 224, 2), (30, 40, 448, 1), (30, 40, 672, 1), (30, 40, 816, 1), (30, 40, 816, 2), (15, 20, 1392, 1),
                         (52, 68, 224, 2), (26, 34, 672, 1), (13, 17, 1392, 1), (80, 120, 224, 2), (40, 60, 816, 1), (20, 30, 1392, 1),
                         (7, 5, 16, 1), (33, 130, 48, 2), (17, 3, 40, 1), (1, 1, 8, 1)]:
        Ho, Wo = -(-H // s), -(-W // s)
        for B in list(range(1, 33)) + [48, 64, 128, 256]:
            want = lib.cfp_dwconv3x3_strips(B, Ho, Wo, C, s, hip.F32)
            for (h, w) in ((H, W), (Ho * s, Wo * s), ((Ho - 1) * s + 1, (Wo - 1) * s + 1)):
                for ld in (C, C + 12):
                    got = lib.cfp_dwconv3x3_launch_slots(B, h, w, Ho, Wo, C, s, ld, ld, hip.F32)
                    assert got == want and got > 0, (B, H, W, C, s, h, w, ld, got, want)
                    n += 1
    assert n > 3000


def test_f16x3_plan_picks_the_occupancy_builds_only_for_in_flight_launches():
    """The 64x64 / 128x32 / 64x128 f16x3 tiles exist a second time under a register budget (variant ids 34-36: one more resident workgroup per
    CU; faster with several batches in flight, slower alone).  The plan must hand them out under the in-flight hint only, never for split-K
    or per-image split plans, and every other variant must be unchanged by the hint's occupancy rule (cfp_debug_set key 32 switches it off)."""
    import ctypes
    from cfpnet_amd import hip
    lib = hip.load()

    def plan(M, N, K, rpb=0, B=1):
        v, s = ctypes.c_int(0), ctypes.c_int(0)
        lib.cfp_conv2d_plan(M, N, K, 1, 1, hip.F32X3, rpb, B, ctypes.byref(v), ctypes.byref(s))
        return v.value, s.value

    shapes = [(9600, 816, 136), (2400, 1392, 232), (9600, 448, 112), (153600, 96, 32), (38400, 64, 128), (153600, 40, 160), (9600, 512, 128),
              (240, 256, 128), (2400, 232, 1392), (300, 304, 1824), (614400, 128, 64)]
    try:
        alone = [plan(*s) for s in shapes]
        assert not any(400 + 34 <= v <= 400 + 36 for v, _ in alone), alone
        lib.cfp_debug_set(17, 1)                       # plan as for several batches in flight
        flight = [plan(*s) for s in shapes]
        lib.cfp_debug_set(32, 0)
        flight_plain = [plan(*s) for s in shapes]
        lib.cfp_debug_set(32, 1)
        occ = {13: 34, 16: 35, 15: 36}
        assert any(400 + 34 <= v <= 400 + 36 for v, _ in flight), flight
        for (v1, s1), (v0, s0) in zip(flight, flight_plain):
            assert s1 == s0 and (v1 == v0 or (s0 <= 1 and v1 - 400 == occ.get(v0 - 400))), (flight, flight_plain)
    finally:
        lib.cfp_debug_set(17, 0)
        lib.cfp_debug_set(32, 1)
