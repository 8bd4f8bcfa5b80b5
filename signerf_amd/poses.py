"""Camera-pose generators -- host-side mirror of the reference's helpers (row a3 of SURVEY.md §8(a)).

Same names, arguments and RNG consumption as /root/reference/signerf/utils/poses_generation.py:22-73 (circle_poses)
and :76-134 (random_sphere_poses), so callers (interface.py:62-71, 828-838) can switch imports.  A handful of 4x4
matrices per sheet: host work, not a kernel.  Checked bit-for-bit against tests/golden/poses.npz.
"""

from __future__ import annotations

import math
from typing import List, Tuple

import torch


def safe_normalize(x: torch.Tensor, eps: float = 1e-20) -> torch.Tensor:
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


def _orient(poses: torch.Tensor, target: List[float], device) -> torch.Tensor:
    size = poses.shape[0]
    tgt = torch.tensor([float(t) for t in target], dtype=torch.float32, device=device)
    world_up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float32, device=device).repeat(size, 1)
    z = safe_normalize(poses[:, :3, 3] - tgt)
    x = safe_normalize(torch.cross(world_up, z, dim=-1))
    y = safe_normalize(torch.cross(z, x, dim=-1))
    poses[:, :3, 0], poses[:, :3, 1], poses[:, :3, 2] = x, y, z
    return poses


def _place(poses: torch.Tensor, radius: float, thetas: torch.Tensor, phis: torch.Tensor, position: List[float]) -> None:
    poses[:, 0, 3] = radius * torch.sin(thetas) * torch.cos(phis) + position[0]
    poses[:, 1, 3] = radius * torch.sin(thetas) * torch.sin(phis) + position[1]
    poses[:, 2, 3] = radius * torch.cos(thetas) + position[2]


def circle_poses(size: int, device: torch.device, radius: float, theta: float, phi: Tuple[float, float],
                 position: List[float], target: List[float]) -> torch.Tensor:
    """[size,4,4] c2w on a circle of elevation ``theta`` (degrees), azimuth linspace(phi[0], phi[1], size), looking at ``target``."""
    poses = torch.eye(4, dtype=torch.float, device=device).repeat(size, 1, 1)
    phis = torch.linspace(math.radians(phi[0]), math.radians(phi[1]), size, device=device)
    th = torch.tensor([math.radians(theta)], dtype=torch.float32, device=device)
    _place(poses, radius, th, phis, position)
    return _orient(poses, target, device)


def random_sphere_poses(size: int, device: torch.device, radius: float, theta: Tuple[float, float], phi: Tuple[float, float],
                        position: List[float], target: List[float]) -> torch.Tensor:
    """[size,4,4] c2w uniformly distributed over the sphere patch theta x phi (degrees); torch global RNG."""
    t0, t1 = math.radians(theta[0]), math.radians(theta[1])
    p0, p1 = math.radians(phi[0]), math.radians(phi[1])
    poses = torch.eye(4, dtype=torch.float, device=device).repeat(size, 1, 1)
    lo, hi = (1 - math.cos(t0)) * 0.5, (1 - math.cos(t1)) * 0.5
    thetas = torch.acos(1 - 2 * (torch.rand(size, device=device) * (hi - lo) + lo))
    phis = torch.rand(size, device=device) * (p1 - p0) + p0
    _place(poses, radius, thetas, phis, position)
    return _orient(poses, target, device)
