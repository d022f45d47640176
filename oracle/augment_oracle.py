"""CPU oracle for the NYU training augmentation -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of `DataLoadPreprocess.random_crop` / `train_preprocess` / `augment_image` and `ToTensor` + `Normalize`
(`/root/reference/src/dataloader/nyu.py:128-136,204-245,266-285`) for GIVEN random draws.  Only `tests/` may import it.

Parity status: crop / flip / gamma / brightness / colour / clip PINNED against the reference's own methods
(`oracle/gen_golden_augment.py` imports `src.dataloader.nyu` behind import-only stubs for torchvision / h5py / matplotlib
and replays seeded draws; fixture `tests/golden/augment.npz`).  The final `transforms.Normalize` is torchvision (absent
here): restated as its documented arithmetic `(x - mean) / std` in float32 -- that one line is UNPINNED.
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def augment(rgb_u8, depth_mm, x0, y0, flip, do_aug, gamma, brightness, colors, H, W, normalize=True):
    """rgb_u8 [H0,W0,3] uint8, depth_mm [H0,W0] uint16 -> (image [3,H,W] f32, depth [1,H,W] f32)."""
    image = np.array(rgb_u8, dtype=np.float32) / 255.                       # nyu.py:128
    depth = (np.array(depth_mm, dtype=np.float32) / 1000.0)[:, :, None]     # nyu.py:129-130
    image = image[y0:y0 + H, x0:x0 + W, :]                                  # random_crop, nyu.py:204-213
    depth = depth[y0:y0 + H, x0:x0 + W, :]
    if flip:                                                                # nyu.py:217-220
        image = image[:, ::-1, :].copy()
        depth = depth[:, ::-1, :].copy()
    if do_aug:                                                              # augment_image, nyu.py:229-245
        image_aug = image ** gamma
        image_aug = image_aug * brightness
        white = np.ones((image.shape[0], image.shape[1]))
        color_image = np.stack([white * colors[i] for i in range(3)], axis=2)
        image_aug *= color_image
        image = np.clip(image_aug, 0, 1)
    img = image.transpose(2, 0, 1)                                          # ToTensor.to_tensor, nyu.py:296-297
    if normalize:
        img = (img - MEAN[:, None, None]) / STD[:, None, None]              # transforms.Normalize (restated)
    return np.ascontiguousarray(img, dtype=np.float32), np.ascontiguousarray(depth.transpose(2, 0, 1), dtype=np.float32)
