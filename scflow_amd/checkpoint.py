"""Checkpoint ingestion.

* ``convert_mmflow_state_dict`` restates tools/mmflow_ckpt_converter.py:30-35 of the
  reference: an mmflow RAFT checkpoint has ONE feature encoder under ``encoder.*``; SCFlow
  keeps it under two names (``real_encoder.*`` / ``render_encoder.*``, one shared module when
  ``seperate_encoder=False``), every other key is kept.
* ``load_checkpoint`` accepts a ``state_dict`` or the mmcv file layout ``{'state_dict': ...,
  'meta': ..., 'optimizer': ...}`` (train.py / test.py use mmcv's ``load_checkpoint``), strips
  an optional ``module.`` prefix (DDP), loads into a HIP refiner and re-packs its kernel-layout
  weights (``HipModule.load_state_dict`` drops the packed caches).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping, Optional, Union

import torch

__all__ = ['convert_mmflow_state_dict', 'load_checkpoint']


def convert_mmflow_state_dict(state_dict: Mapping[str, torch.Tensor]) -> 'OrderedDict[str, torch.Tensor]':
    out: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith('encoder'):
            out[k.replace('encoder', 'real_encoder')] = v
            out[k.replace('encoder', 'render_encoder')] = v
        else:
            out[k] = v
    return out


def load_checkpoint(model: torch.nn.Module, ckpt: Union[str, Mapping], strict: Optional[bool] = None,
                    from_mmflow: bool = False):
    """returns ``(missing, unexpected, mismatched)`` key lists.

    ``strict`` defaults to True for an SCFlow checkpoint and to False with ``from_mmflow`` (an
    mmflow RAFT checkpoint -- the ``init_cfg`` of configs/refine_models/scflow.py -- has no pose
    head / delta-flow / mask encoders, and its ``decoder.mask_pred.predict_layer`` is the
    576-channel convex-up-sampling head under the key SCFlow uses for its 1-channel mask head).
    Non-strict loading behaves like mmcv's ``load_checkpoint`` in the reference (train.py /
    test.py): entries whose shape differs from the model's are SKIPPED and reported, instead of
    raising as ``torch.nn.Module.load_state_dict`` does even with ``strict=False``."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location='cpu')
    sd = ckpt['state_dict'] if isinstance(ckpt, Mapping) and 'state_dict' in ckpt else ckpt
    sd = OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in sd.items())
    if from_mmflow:
        sd = convert_mmflow_state_dict(sd)
    if strict is None:
        strict = not from_mmflow
    mismatched = []
    if not strict:
        own = model.state_dict()
        keep = OrderedDict()
        for k, v in sd.items():
            if k in own and tuple(own[k].shape) != tuple(v.shape):
                mismatched.append((k, tuple(v.shape), tuple(own[k].shape)))
            else:
                keep[k] = v
        sd = keep
    res = model.load_state_dict(sd, strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys), mismatched
