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
from typing import Dict, Mapping, Union

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


def load_checkpoint(model: torch.nn.Module, ckpt: Union[str, Mapping], strict: bool = True,
                    from_mmflow: bool = False):
    """returns the (missing, unexpected) key lists of ``load_state_dict``."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location='cpu')
    sd = ckpt['state_dict'] if isinstance(ckpt, Mapping) and 'state_dict' in ckpt else ckpt
    sd = OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in sd.items())
    if from_mmflow:
        sd = convert_mmflow_state_dict(sd)
    return model.load_state_dict(sd, strict=strict)
