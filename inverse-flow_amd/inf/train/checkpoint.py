"""Checkpoints in the reference's layout (inf/train/experiment.py:475-502 `Experiment.save` / `load`):

    {'summary': ..., 'model_state_dict': ..., 'optimizer_state_dict': ..., 'scheduler_state_dict': ..., 'config': ...}

written with torch.save.  The layers of this package keep the reference's parameter and buffer names (`weight_fwd` for
the inverse-flow layers, `conv.weight` for PaddedConv2d, `net.N.*` for the couplings, ...), so a checkpoint of either
side loads into the other with strict=True."""
import torch

KEYS = ("summary", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "config")


def save_checkpoint(path, model, optimizer, scheduler, summary=None, config=None):
    checkpoint = {"summary": summary if summary is not None else {},
                  "model_state_dict": model.state_dict(),
                  "optimizer_state_dict": optimizer.state_dict(),
                  "scheduler_state_dict": scheduler.state_dict(),
                  "config": config if config is not None else {}}
    torch.save(checkpoint, path)
    return checkpoint


def load_checkpoint(path, model, optimizer=None, scheduler=None, map_location=None):
    """Restores what experiment.py:495-502 restores; returns (summary, config)."""
    checkpoint = torch.load(path, map_location=map_location, weights_only=False)
    missing = [k for k in KEYS if k not in checkpoint]
    if missing:
        raise KeyError("checkpoint {} lacks {}".format(path, missing))
    model.load_state_dict(checkpoint["model_state_dict"])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
    return checkpoint["summary"], checkpoint["config"]
