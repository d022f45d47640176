"""Bit-exact integer geometry against known answers captured from the
reference (`src/utils/dataloader.py:13-40,65-80`)."""
import json
import os

import numpy as np
import pytest

from cfpnet_amd import geometry as G


@pytest.fixture(scope="module")
def kats(golden_dir):
    return json.load(open(os.path.join(golden_dir, "geometry.json")))


def test_patch_info_known_answers(kats):
    n = 0
    for name, case in kats.items():
        if name == "sample_points":
            continue
        pi = G.patch_info_from_rect_data(np.array(case["rects"], dtype=np.float32))
        assert pi["zone_num"] == case["zone_num"], name
        for s in (4, 8, 16):
            for k in ("pad_size", "patch_size", "index_wo_pad"):
                assert pi[s][k].dtype == np.int32
                assert pi[s][k].tolist() == case[str(s)][k], (name, s, k)
        n += 1
    assert n >= 8


def test_survey_appendix_a_values():
    pi = G.patch_info_from_rect_data(G.centered_zone_rects(480, 640, 8, 56))
    assert pi[4]["index_wo_pad"].tolist() == [4, 24, 116, 136] and pi[4]["patch_size"].tolist() == [14, 14]
    assert pi[8]["index_wo_pad"].tolist() == [2, 12, 58, 68] and pi[8]["patch_size"].tolist() == [7, 7]
    assert pi[16]["index_wo_pad"].tolist() == [1, 6, 29, 34] and pi[16]["patch_size"].tolist() == [4, 4]
    r = G.centered_zone_rects(480, 640, 8, 56)
    assert r[0].tolist() == [16, 96, 72, 152] and r[-1].tolist() == [408, 488, 464, 544]
    pi = G.patch_info_from_rect_data(G.centered_zone_rects(416, 544, 6, 64))
    assert pi[4]["index_wo_pad"].tolist() == [4, 20, 100, 116] and pi[16]["index_wo_pad"].tolist() == [1, 5, 25, 29]


def test_sample_points_bit_exact(kats):
    c = kats["sample_points"]
    out = G.sample_points_from_hist(np.array(c["mu_sigma"], np.float32), np.array(c["mask"]))
    got = [np.float32(v).tobytes().hex() for v in out.reshape(-1)]
    assert got == c["out_hex"]
    assert (out[~np.array(c["mask"])] == 0).all()


def test_fusion_geometry_eval_and_interp():
    infos = [G.patch_info_from_rect_data(G.centered_zone_rects(480, 640, 8, 56))] * 3
    pi = G.collate_patch_info(infos)
    g16 = G.FusionGeometry.from_patch_info(pi, 640 / 40)     # float key like fusion.py:41
    assert (g16.zone_num, g16.p1, g16.p2, g16.tzh, g16.tzw) == (8, 4, 4, 28, 28)
    assert g16.interpolate and (g16.grid_h, g16.grid_w) == (32, 32)      # 28 -> 32 -> 28
    assert g16.clipped(30, 40) == (1, 29, 6, 34)
    g8 = G.FusionGeometry.from_patch_info(pi, 8)
    assert not g8.interpolate and g8.clipped(60, 80) == (2, 58, 12, 68)
    g4 = G.FusionGeometry.from_patch_info(pi, 4.0)
    assert not g4.interpolate and (g4.p1, g4.tzh) == (14, 112)


def test_fusion_geometry_overhang():
    rects = G.centered_zone_rects(480, 640, 8, 56) + np.array([-40, -110, -40, -110], np.float32)
    g = G.FusionGeometry.from_patch_info(G.collate_patch_info([G.patch_info_from_rect_data(rects)]), 4)
    assert (g.pad_h, g.pad_w) == (6, 4) and (g.sy_wo, g.sx_wo) == (-6, -3)
    y0, y1, x0, x1 = g.clipped(120, 160)
    assert (y0, x0) == (0, 0) and y1 == g.ey_wo and x1 == g.ex_wo


def test_window_helpers():
    assert G.lsa_padding(30, 40, 6) == (0, 2) and G.lsa_padding(60, 80, 9) == (3, 1) and G.lsa_padding(120, 160, 12) == (0, 8)
    assert G.gsa_keys(30, 40, 6) == (5, 6) and G.gsa_keys(60, 80, 9) == (6, 8) and G.gsa_keys(120, 160, 12) == (10, 13)
