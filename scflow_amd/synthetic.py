"""Seeded synthetic inputs of the shapes the reference's test pipeline feeds
``SCFlowRefiner.get_pose``: there is no dataset and no
renderer in this build, so the rendered RGB / depth pair is synthesised.

Images are U[0,1) (the reference pipeline normalises to [0,1],
configs/refine_datasets/ycbv_real.py:12-13,58); K is a 600-px pin-hole centred
on the crop; the object is a disc of radius 0.31*H at ~800 mm whose depth
varies smoothly so that rotation updates produce a non-trivial flow field.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

__all__ = ['make_inputs']


def _rot_xyz(ax: float, ay: float, az: float) -> torch.Tensor:
    cx, sx, cy, sy, cz, sz = (math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay),
                              math.cos(az), math.sin(az))
    rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
    ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
    rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
    return (rz @ ry @ rx).float()


def make_inputs(n: int, height: int = 256, width: int = 256, seed: int = 0,
                num_class: int = 21, varied: bool = True) -> Dict[str, torch.Tensor]:
    """returns CPU tensors: render_images, real_images (n,3,H,W); ref_rotation
    (n,3,3); ref_translation (n,3); depth (n,H,W); internel_k (n,3,3);
    label (n,) int64."""
    g = torch.Generator(device='cpu')
    g.manual_seed(1234567 + seed)
    real = torch.rand((n, 3, height, width), generator=g)
    rend = torch.rand((n, 3, height, width), generator=g)
    k = torch.tensor([[600., 0., width / 2.], [0., 600., height / 2.], [0., 0., 1.]])
    ks = k[None].repeat(n, 1, 1)
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32),
                            torch.arange(width, dtype=torch.float32), indexing='ij')
    rots, trans, depths = [], [], []
    ang = torch.rand((n, 3), generator=g) * 0.6 - 0.3
    off = torch.rand((n, 3), generator=g) * 2 - 1
    for i in range(n):
        if varied:
            rots.append(_rot_xyz(*[float(a) for a in ang[i]]))
            t = torch.tensor([20. * float(off[i, 0]), 20. * float(off[i, 1]),
                              800. + 60. * float(off[i, 2])])
        else:
            rots.append(torch.eye(3))
            t = torch.tensor([0., 0., 800.])
        trans.append(t)
        cx = width / 2. + (6. * float(off[i, 0]) if varied else 0.)
        cy = height / 2. + (6. * float(off[i, 1]) if varied else 0.)
        rad = 0.31 * height
        rr = ((xs - cx) ** 2 + (ys - cy) ** 2) / (rad * rad)
        bump = 60. * torch.sqrt(torch.clamp(1. - rr, min=0.)) if varied else 0.
        d = torch.where(rr < 1., float(t[2]) - bump, torch.zeros(()))
        depths.append(d.float())
    label = torch.randint(0, num_class, (n,), generator=g)
    return dict(render_images=rend, real_images=real, ref_rotation=torch.stack(rots),
                ref_translation=torch.stack(trans), depth=torch.stack(depths),
                internel_k=ks, label=label)
