"""Checkpoints in the reference's layout (examples/train.py:153-164 writes them,
examples/evaluate.py:60-64 reads them): one ``torch.save`` dictionary with the
keys ``epoch``, ``model_state_dict``, ``optimizer_state_dict``,
``scheduler_state_dict``, ``train_loss``, ``test_loss``.  The model here has the
reference's parameter names, so files written by either side load on the other;
the reference's reader loads the optimizer and scheduler entries unconditionally
(examples/evaluate.py:63-64), so a file written WITHOUT them (a warning is issued)
loads only through ``load_checkpoint`` here.
Loading uses ``weights_only=True``: nothing stored in the file is executed.
"""
import os
import warnings

import torch

KEYS = ("epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "train_loss",
        "test_loss")


def save_checkpoint(checkpoint_dir, epoch, model, optimizer=None, scheduler=None, train_loss=None,
                    test_loss=None):
    """Writes ``<checkpoint_dir>/state-<epoch>.pth`` and returns its path."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = os.path.join(checkpoint_dir, f"state-{epoch}.pth")
    if optimizer is None or scheduler is None:
        warnings.warn("checkpoint without optimizer / scheduler state: the reference's "
                      "evaluate.py cannot read it (it loads both unconditionally)")
    torch.save({
        "epoch": epoch,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else None,
        "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
        "train_loss": train_loss,
        "test_loss": test_loss,
    }, path)
    return path


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location=None):
    """Restores model (strictly) and, when given and present in the file, optimizer and
    scheduler; returns the stored dictionary without the state dictionaries."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    if "model_state_dict" not in ckpt:
        raise KeyError(f"{path}: no 'model_state_dict' entry (keys: {sorted(ckpt)})")
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    if optimizer is not None and ckpt.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    if scheduler is not None and ckpt.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(ckpt["scheduler_state_dict"])
    return {k: v for k, v in ckpt.items() if not k.endswith("_state_dict")}
