"""SE(3) exponential map (counterpart of the reference's ``se3.exp``, se3.py:22-27).

Only the artificial-light model uses it (``SUCRe.compute_l_z`` with ``light_model=True``, sucre.py:54-61); the
matching path uses ``sfm.Pose`` instead.  Kept differentiable: the twist is embedded in a 4x4 generator and pushed
through ``torch.matrix_exp`` exactly like the reference, so values and gradients agree with it.
"""
from __future__ import annotations

import torch
from torch import Tensor


def hat(twist: Tensor) -> Tensor:
    """(6,) twist ``(w1, w2, w3, p1, p2, p3)`` -> 4x4 se(3) generator ``[[skew(w), p], [0, 0]]``."""
    if twist.shape != (6,):
        raise ValueError(f'twist must have shape (6,), got {tuple(twist.shape)}')
    w, p = twist[:3], twist[3:]
    o = twist.new_zeros(())
    rows = [torch.stack([o, -w[2], w[1], p[0]]),
            torch.stack([w[2], o, -w[0], p[1]]),
            torch.stack([-w[1], w[0], o, p[2]]),
            torch.stack([o, o, o, o])]
    return torch.stack(rows)


def exp(pose: Tensor) -> tuple[Tensor, Tensor]:
    """Twist (6,) -> rotation (3,3) and translation (3,1) of ``matrix_exp(hat(pose))``."""
    T = torch.matrix_exp(hat(pose))
    return T[:3, :3], T[:3, 3:4]
