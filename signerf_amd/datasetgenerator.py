"""``render_camera`` -- render + mask + condition for one camera, entirely on the GPU.

Mirrors ``DatasetGenerator.render_camera`` of the reference
(/root/reference/signerf/datasetgenerator/datasetgenerator.py:677-820) for ``masking_mode="aabb"`` (the default, :56):
same arguments, same 3-tuple ``(rgb [H,W,3], mask [H,W,1] bool, condition [H,W,1] fp32)``, same early-exit shapes.  The
reference moves the mask to the CPU for ``cv2.dilate`` (:776-778) and synchronises on ``torch.sum(visible_mask) > 1e-6``
(:770); here the slab test, the mask, the elliptical dilation, the masked depth min/max and the condition image are three
small kernels behind ``sn_aabb_mask_condition`` and nothing leaves the device.

``masking_mode="shape"`` needs the OpenGL mesh rasteriser (signerf/renderer, out of scope) and raises.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib


@dataclass
class DatasetGeneratorConfig:
    """The fields of the reference's DatasetGeneratorConfig (datasetgenerator.py:32-81) that shape render_camera."""

    masking_mode: str = "aabb"
    aabb_min: List[float] = field(default_factory=lambda: [-0.1, -0.1, -0.1])
    aabb_max: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.1])
    mask_dialation: Optional[Tuple[int, int]] = (50, 50)
    additional_depth_radius: float = 0.1
    inverse_mask: bool = False
    manual_depth: Optional[Tuple[float, float]] = None


def aabb_mask_and_condition(depth: Tensor, rays_o: Tensor, rays_d: Tensor, aabb: Tensor, mask_dialation: Optional[Tuple[int, int]] = (50, 50),
                            inverse_mask: bool = False, manual_depth: Optional[Tuple[float, float]] = None,
                            additional_depth_radius: float = 0.1, with_condition: bool = True):
    """datasetgenerator.py:758-818 on the GPU.  depth [H,W,1], rays_o/rays_d [H,W,3], aabb [2,3] ->
    (mask [H,W,1] bool, condition [H,W,1] fp32 | None).  If nothing is visible both are all-zero, as in the reference."""
    lib = _lib.load()
    H, W = depth.shape[0], depth.shape[1]
    dev = depth.device
    f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    o, d, z = f32(rays_o), f32(rays_d), f32(depth)
    opts = _lib.SnMaskOpts()
    opts.inverse_mask = int(bool(inverse_mask))
    if mask_dialation is not None:
        # cv2.getStructuringElement takes (width, height)
        opts.dilate_w, opts.dilate_h = int(mask_dialation[0]), int(mask_dialation[1])
    opts.has_manual_depth = int(manual_depth is not None)
    if manual_depth is not None:
        opts.manual_min, opts.manual_max = float(manual_depth[0]), float(manual_depth[1])
    opts.additional_depth_radius = float(additional_depth_radius)
    box = (C.c_float * 6)(*aabb.detach().to("cpu", torch.float32).reshape(-1).tolist())
    with torch.cuda.device(dev):
        mask = torch.empty((H, W, 1), dtype=torch.uint8, device=dev)
        cond = torch.empty((H, W, 1), dtype=torch.float32, device=dev) if with_condition else None
        ws = torch.empty(lib.sn_mask_workspace_bytes(H, W), dtype=torch.uint8, device=dev)
        _lib.check(lib.sn_aabb_mask_condition(_lib.ptr(o), _lib.ptr(d), _lib.ptr(z), H, W, box, C.byref(opts), _lib.ptr(mask),
                                              _lib.ptr(cond), ws.data_ptr(), ws.numel(), _lib.current_stream()),
                   None, "sn_aabb_mask_condition")
    return mask.bool(), cond


def render_camera(config: DatasetGeneratorConfig, graph, camera, with_mask: bool = True, with_condition: bool = True):
    """One camera: rgb, mask, condition (datasetgenerator.py:677-820, aabb mode)."""
    camera_ray_bundle = camera.generate_rays(camera_indices=0, aabb_box=graph.render_aabb)
    graph.eval()
    outputs = graph.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
    graph.train()
    if outputs is None:
        raise RuntimeError("Render thread did not return any outputs")
    rgb, depth = outputs["rgb"], outputs["depth"]
    if not with_mask:
        return rgb, None, None, None  # the reference's 4-tuple early exit (:708)
    if config.masking_mode != "aabb":
        raise NotImplementedError("masking_mode='shape' needs the OpenGL mesh rasteriser (signerf/renderer), which is out of scope")
    aabb = torch.tensor([config.aabb_min, config.aabb_max], dtype=torch.float32)
    mask, cond = aabb_mask_and_condition(depth, camera_ray_bundle.origins, camera_ray_bundle.directions, aabb, config.mask_dialation,
                                         config.inverse_mask, config.manual_depth, config.additional_depth_radius, with_condition)
    if not with_condition:
        return rgb, mask, None, None
    return rgb, mask, cond
