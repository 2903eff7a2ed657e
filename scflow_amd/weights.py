"""Deterministic synthetic weights for the SCFlow refiner architecture.

No checkpoint can be downloaded here, so benchmarks and parity tests run on
seeded random weights of the reference architecture (key layout of the
reference ``state_dict``).  Values depend only on
``(seed, key name, shape)`` -- never on construction order -- so the very same
tensors can be loaded into the reference model (tests/golden/make_golden.py),
the CPU oracle and the HIP refiner.

Scales keep activations O(1) through the recurrent loop and make the pose
head emit small non-identity updates (the reference zero-initialises it,
head/pose_head.py:187-198, which would leave the pose path untested).
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import torch

__all__ = ['fill_state_dict']


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def fill_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0,
                    num_class: int = 21, shared_encoder: bool = True) -> Dict[str, torch.Tensor]:
    """``shapes``: key -> shape (e.g. ``{k: v.shape for k, v in sd.items()}``).
    ``shared_encoder=False``: ``real_encoder.*`` gets its own values (``seperate_encoder=True``,
    base_refiner.py:33-35) instead of aliasing ``render_encoder.*``."""
    out: Dict[str, torch.Tensor] = {}
    for key in sorted(shapes):
        shape = tuple(int(s) for s in shapes[key])
        g = _gen(seed, key.replace('real_encoder.', 'render_encoder.') if shared_encoder else key)
        leaf = key.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            out[key] = torch.zeros(shape, dtype=torch.long)
            continue
        rnd = torch.randn(shape, generator=g, dtype=torch.float32) if shape else torch.zeros(())
        is_norm = any(t in key for t in ('.bn', '.gn', 'downsample.1', '.in1', '.in2')) \
            or key.split('.')[-2].startswith(('bn', 'gn'))
        if leaf == 'running_mean':
            val = 0.1 * rnd
        elif leaf == 'running_var':
            val = 1.0 + 0.2 * rnd.abs()
        elif is_norm and leaf == 'weight':
            val = 1.0 + 0.1 * rnd
        elif is_norm and leaf == 'bias':
            val = 0.1 * rnd
        elif 'rotation_pred' in key:
            if leaf == 'weight':
                val = 0.02 * rnd / (shape[1] ** 0.5)
            else:
                ident = torch.tensor([1., 0., 0., 0., 1., 0.]).repeat(num_class)
                val = ident[:shape[0]] + 0.01 * rnd
        elif 'translation_pred' in key:
            val = (0.05 * rnd / (shape[1] ** 0.5)) if leaf == 'weight' else 0.02 * rnd
        elif leaf == 'weight' and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            val = rnd * (1.6 / fan_in) ** 0.5
        elif leaf == 'bias':
            val = 0.05 * rnd
        else:
            val = rnd
        out[key] = val.contiguous()
    return out
