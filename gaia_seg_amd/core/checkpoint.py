"""Checkpoint I/O in the reference's format: a torch pickle ``{'meta', 'state_dict', 'optimizer'}``
whose state_dict holds the MAX-size parameters as contiguous OIHW tensors under the reference's
module names (tools/train_supernet.py:197-202; gaiaseg/apis/train.py:172-175;
gaiaseg/models/backbones/dynamic_resnet.py:343-345).  The HWIO physical
layout of this implementation never leaks into a file: `optimizer` holds the SGD momentum per
parameter NAME in the same logical layout (ParamArena.state_dict)."""
import os

import torch


def state_dict_oihw(model):
    return {k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def save_checkpoint(model, filename, optimizer=None, meta=None):
    ck = dict(meta=dict(meta or {}), state_dict=state_dict_oihw(model))
    if optimizer is not None:
        ck["optimizer"] = _to_cpu(optimizer.state_dict())
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(ck, filename)


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    ck = torch.load(filename, map_location=map_location)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    own = model.state_dict()
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                if own[k].shape != v.shape:
                    raise RuntimeError("size mismatch for %s: checkpoint %s vs model %s"
                                       % (k, tuple(v.shape), tuple(own[k].shape)))
                own[k].copy_(v)  # in place: keeps arena views / HWIO storage
    if strict and (missing or unexpected):
        raise RuntimeError("missing keys %s, unexpected keys %s" % (missing, unexpected))
    if logger is not None and (missing or unexpected):
        logger.warning("missing keys: %s; unexpected keys: %s" % (missing, unexpected))
    return ck if isinstance(ck, dict) else dict(state_dict=sd)
