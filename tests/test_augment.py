"""NYU training augmentation: oracle vs the reference's own methods (CPU), HIP kernel vs oracle and
goldens (GPU).  Tolerance 2e-6 absolute on [0, 1] pixels (device powf vs libm powf), depth exact."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment_oracle as AO

from helpers import GOLDEN

Z = np.load(os.path.join(GOLDEN, "augment.npz"))
H, W = 416, 544


def _source(seed, H0=456, W0=608):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (H0, W0, 3), dtype=np.uint8), rng.integers(0, 10000, (H0, W0), dtype=np.uint16)


def _params(case):
    p = Z[f"c{case}.params"]
    return int(p[0]), int(p[1]), bool(p[2]), bool(p[3]), float(p[4]), float(p[5]), p[6:9]


@pytest.mark.parametrize("case", range(6))
def test_oracle_matches_reference_methods(case):
    rgb, dmm = _source(100 + int(Z["seeds"][case]))
    img, dep = AO.augment(rgb, dmm, *_params(case), H, W, normalize=False)
    assert np.array_equal(img.transpose(1, 2, 0)[::8, ::8], Z[f"c{case}.img"])          # same numpy ops: bit-equal
    assert np.array_equal(dep[0, ::8, ::8], Z[f"c{case}.dep"])


def test_draws_follow_the_reference_order():
    from cfpnet_amd import augment
    for case in range(6):
        s = int(Z["seeds"][case])
        random.seed(s); np.random.seed(s)
        x0, y0, flip, do_aug, gamma, brightness, colors = augment.draw_params(456, 608, H, W)
        want = _params(case)
        assert (x0, y0, flip, do_aug) == want[:4] and abs(gamma - want[4]) < 1e-15 and abs(brightness - want[5]) < 1e-15
        assert np.allclose(colors, want[6], rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_kernel_matches_oracle_and_reference():
    from cfpnet_amd import augment
    srcs = [_source(100 + int(Z["seeds"][c])) for c in range(6)]
    rgb = torch.from_numpy(np.stack([s[0] for s in srcs])).cuda()
    dmm = torch.from_numpy(np.stack([s[1] for s in srcs]).view(np.int16)).cuda()
    params = [_params(c) for c in range(6)]
    img, dep = augment.augment(rgb, dmm, params, H, W)
    torch.cuda.synchronize()
    for c in range(6):
        oi, od = AO.augment(srcs[c][0], srcs[c][1], *params[c], H, W)
        assert np.array_equal(dep[c].cpu().numpy(), od)
        assert float(np.abs(img[c].cpu().numpy() - oi).max()) <= 2e-6 / 0.224 + 1e-6          # 2e-6 on the [0,1] pixel, then / std
        raw = img[c].cpu().numpy() * AO.STD[:, None, None] + AO.MEAN[:, None, None]                # back to [0,1] for the reference's un-normalised golden
        assert float(np.abs(raw.transpose(1, 2, 0)[::8, ::8] - Z[f"c{c}.img"]).max()) <= 3e-6


# ---- random rotation (nyu.py:121-124): Pillow's Image.rotate ---------------------------------------------------------------------
ZR = np.load(os.path.join(GOLDEN, "augment_rotate.npz"))


@pytest.mark.parametrize("case", range(int(ZR["n"])))
def test_rotation_oracle_equals_pillow_golden(case):
    """The restatement of Pillow's rotate (bilinear RGB / nearest 16-bit) against outputs Pillow itself produced
    (oracle/gen_golden_augment_rotate.py): byte for byte."""
    angle = float(ZR[f"c{case}.angle"])
    rgb, dep = ZR[f"c{case}.rgb"], ZR[f"c{case}.dep"]
    if angle % 360.0 == 0.0:
        assert np.array_equal(ZR[f"c{case}.rgb_rot"], rgb) and np.array_equal(ZR[f"c{case}.dep_rot"], dep)      # Pillow: a copy
        return
    assert np.array_equal(AO.rotate_rgb_bilinear(rgb, angle), ZR[f"c{case}.rgb_rot"])
    assert np.array_equal(AO.rotate_u16_nearest(dep, angle), ZR[f"c{case}.dep_rot"])


def test_rotation_oracle_equals_pillow_live():
    """Same check against the installed Pillow at the loader's real size (456x608 after the Kinect-border crop)."""
    Image = pytest.importorskip("PIL.Image")
    rgb, dep = _source(321)
    for seed in (1, 2):
        random.seed(seed)
        angle = (random.random() - 0.5) * 2 * 2.5
        assert np.array_equal(AO.rotate_rgb_bilinear(rgb, angle), np.array(Image.fromarray(rgb, "RGB").rotate(angle, resample=Image.BILINEAR)))
        assert np.array_equal(AO.rotate_u16_nearest(dep, angle), np.array(Image.fromarray(dep).rotate(angle, resample=Image.NEAREST)))


def test_rotation_matrix_and_draw_follow_the_reference():
    from cfpnet_amd import augment
    random.seed(5)
    want = (random.random() - 0.5) * 2 * 2.5
    random.seed(5)
    assert augment.draw_rotation(2.5) == want
    assert augment.rotate_matrix(want, 608, 456) == AO.rotate_matrix(want, 608, 456)
    assert augment.rotate_matrix(0.0, 608, 456) == [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    with pytest.raises(NotImplementedError):
        augment.rotate_matrix(90.0, 608, 456)


@pytest.mark.gpu
def test_rotation_kernel_is_byte_exact():
    from cfpnet_amd import augment
    # goldens produced by Pillow
    for case in range(int(ZR["n"])):
        rgb = torch.from_numpy(ZR[f"c{case}.rgb"][None]).cuda()
        dep = torch.from_numpy(ZR[f"c{case}.dep"].view(np.int16)[None]).cuda()
        r, d = augment.rotate(rgb, dep, [float(ZR[f"c{case}.angle"])])
        torch.cuda.synchronize()
        assert np.array_equal(r[0].cpu().numpy(), ZR[f"c{case}.rgb_rot"]), case
        assert np.array_equal(d[0].cpu().numpy().view(np.uint16), ZR[f"c{case}.dep_rot"]), case
    # full size, a batch with a different angle per sample, against the oracle
    srcs = [_source(500 + i) for i in range(4)]
    angles = [-2.4, 0.37, 1.999, -0.001]
    rgb = torch.from_numpy(np.stack([s[0] for s in srcs])).cuda()
    dep = torch.from_numpy(np.stack([s[1] for s in srcs]).view(np.int16)).cuda()
    r, d = augment.rotate(rgb, dep, angles)
    torch.cuda.synchronize()
    for i, a in enumerate(angles):
        assert np.array_equal(r[i].cpu().numpy(), AO.rotate_rgb_bilinear(srcs[i][0], a))
        assert np.array_equal(d[i].cpu().numpy().view(np.uint16), AO.rotate_u16_nearest(srcs[i][1], a))
