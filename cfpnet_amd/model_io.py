"""Checkpoint I/O with the reference's file format (`/root/reference/src/utils/model_io.py`):
`save_weights` writes a bare state_dict, `save_checkpoint` writes {"model", "optimizer", "epoch"},
`load_checkpoint` accepts either and strips DataParallel's "module." prefix; loads are strict."""
import os

import torch


def _mkdir_for(fpath):
    folder = os.path.dirname(fpath)
    if folder and not os.path.isdir(folder):
        os.makedirs(folder)


def save_weights(model, fpath):
    _mkdir_for(fpath)
    torch.save(model.state_dict(), fpath)


def load_weights(model, fpath):
    model.load_state_dict(torch.load(fpath, map_location="cpu"))
    return model


def save_checkpoint(model, optimizer, epoch, fpath):
    _mkdir_for(fpath)
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch}, fpath)


def load_checkpoint(fpath, model, optimizer=None):
    ckpt = torch.load(fpath, map_location="cpu")
    if optimizer is None:
        optimizer = ckpt.get("optimizer", None)
    else:
        optimizer.load_state_dict(ckpt["optimizer"])
    epoch = ckpt.get("epoch", None)
    sd = ckpt["model"] if "model" in ckpt else ckpt
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model, optimizer, epoch
