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

    downscale_factor: int = 2
    width: Optional[int] = None
    height: Optional[int] = None
    masking_mode: str = "aabb"
    aabb_min: List[float] = field(default_factory=lambda: [-0.1, -0.1, -0.1])
    aabb_max: List[float] = field(default_factory=lambda: [0.1, 0.1, 0.1])
    rows: int = 2
    cols: int = 3
    mask_dialation: Optional[Tuple[int, int]] = (50, 50)
    additional_depth_radius: float = 0.1
    border_width_between_images: int = 0
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


# ----------------------------------------------------------------------------------------------------------------------
# Reference-sheet composition (second half): datasetgenerator.py:470-593 and :597-674 around the diffuser call.
# The diffuser itself is a remote HTTP service in the reference (signerf/diffuser) and stays a callable the caller passes in.
# ----------------------------------------------------------------------------------------------------------------------
def sheet_geometry(config: DatasetGeneratorConfig, scaled_image_width: int, scaled_image_height: int) -> Tuple[int, int]:
    """(height, width) of the reference sheet: rows x cols cells plus borders, padded up to a multiple of 8 (:497-502)."""
    import math

    w = int(config.cols * scaled_image_width) + int((config.cols - 1) * config.border_width_between_images)
    h = int(config.rows * scaled_image_height) + int((config.rows - 1) * config.border_width_between_images)
    return int(math.ceil(h / 8) * 8), int(math.ceil(w / 8) * 8)


def cell_window(config: DatasetGeneratorConfig, index: int, scaled_image_width: int, scaled_image_height: int) -> Tuple[int, int, int, int]:
    """(row0, row1, col0, col1) of cell `index` in row-major order (:521-534)."""
    row, col = index // config.cols, index % config.cols
    r0 = row * scaled_image_height + row * config.border_width_between_images
    c0 = col * scaled_image_width + col * config.border_width_between_images
    return r0, r0 + scaled_image_height, c0, c0 + scaled_image_width


def compose_reference_sheet(config: DatasetGeneratorConfig, views, scaled_image_width: int, scaled_image_height: int):
    """datasetgenerator.py:504-553: down-scale every reference view and paste it into its cell.

    views: sequence of (render [H,W,3], mask [H,W,1] bool/uint8, condition [H,W,1]) on the GPU, one per reference camera
    (rows*cols - 1 of them: the last cell is left for the view being generated).  Each down-scale writes straight into the
    sheet (one kernel per image, no temporaries).  -> (image_sheet [SH,SW,3] (ones), mask_sheet [SH,SW,1] (zeros),
    condition_sheet [SH,SW,1] (zeros), references: list of dicts with the reference's keys)."""
    from .ops import resize_bilinear

    if len(views) != config.rows * config.cols - 1:
        raise ValueError(f"Camera count {len(views)} is not equal to (rows * cols) - 1 = {config.rows * config.cols - 1}")
    dev = views[0][0].device
    sh, sw = sheet_geometry(config, scaled_image_width, scaled_image_height)
    with torch.cuda.device(dev):
        image_sheet = torch.ones((sh, sw, 3), dtype=torch.float32, device=dev)
        mask_sheet = torch.zeros((sh, sw, 1), dtype=torch.float32, device=dev)
        condition_sheet = torch.zeros((sh, sw, 1), dtype=torch.float32, device=dev)
    references = []
    for i, (render, mask, condition) in enumerate(views):
        r0, r1, c0, c1 = cell_window(config, i, scaled_image_width, scaled_image_height)
        render_scaled = resize_bilinear(render, scaled_image_height, scaled_image_width, out=image_sheet[r0:r1, c0:c1, :])
        mask_scaled = resize_bilinear(mask, scaled_image_height, scaled_image_width, threshold=True, out=mask_sheet[r0:r1, c0:c1, :])
        condition_scaled = resize_bilinear(condition, scaled_image_height, scaled_image_width, out=condition_sheet[r0:r1, c0:c1, :])
        references.append({"render": render, "mask": mask, "condition": condition, "render_scaled": render_scaled,
                           "mask_scaled": mask_scaled > 0.5, "condition_scaled": condition_scaled})
    return image_sheet, mask_sheet, condition_sheet, references


def split_reference_sheet(config: DatasetGeneratorConfig, edited_sheet: Tensor, image_sheet: Tensor, mask_sheet: Tensor, references,
                          scaled_image_width: int, scaled_image_height: int, height: int, width: int) -> Tensor:
    """datasetgenerator.py:559-593: keep the edit inside the mask, cut every cell out and up-scale it to (height, width).
    Fills references[i]["edited" / "edited_scaled"]; returns the blended sheet."""
    from .ops import resize_bilinear

    edited_sheet = edited_sheet.to(image_sheet.device)
    m3 = mask_sheet.repeat(1, 1, 3)
    edited_sheet = edited_sheet * m3 + image_sheet * (1 - m3)
    for i, ref in enumerate(references):
        r0, r1, c0, c1 = cell_window(config, i, scaled_image_width, scaled_image_height)
        edited_scaled = edited_sheet[r0:r1, c0:c1, :]
        ref["edited"] = resize_bilinear(edited_scaled, height, width)
        ref["edited_scaled"] = edited_scaled
    return edited_sheet


def generate_reference_sheet(config: DatasetGeneratorConfig, graph, cameras, scaled_image_width: int, scaled_image_height: int, diffuse):
    """``DatasetGenerator.generate_reference_sheet`` (:470-593).  cameras: the rows*cols-1 reference cameras (indexable);
    diffuse(image, image, mask, condition) -> edited sheet [SH,SW,3] stands for ``self.diffuser.diffuse`` (:559).
    -> (image_sheet, mask_sheet, condition_sheet, edited_sheet, references) as the reference returns them."""
    # the reference cameras are independent: consecutive ones go to alternating streams (sheet.FrameStreams: the head of one frame fills
    # the wave slots the tail of the previous one leaves idle); the sheet is composed on the caller's stream after the join
    from .sheet import FrameStreams

    fs = FrameStreams(getattr(graph, "device", None))
    views = []
    for i in range(len(cameras)):
        with fs.frame(i):
            views.append(tuple(fs.keep(t) for t in render_camera(config, graph, cameras[i])))
    fs.join()
    image_sheet, mask_sheet, condition_sheet, references = compose_reference_sheet(config, views, scaled_image_width, scaled_image_height)
    edited = diffuse(image_sheet, image_sheet, mask_sheet, condition_sheet)
    H, W = views[0][0].shape[0], views[0][0].shape[1]
    edited = split_reference_sheet(config, edited, image_sheet, mask_sheet, references, scaled_image_width, scaled_image_height,
                                   config.height or H, config.width or W)
    return image_sheet, mask_sheet, condition_sheet, edited, references


def generate_with_reference_sheet(config: DatasetGeneratorConfig, graph, camera, original: Optional[Tensor], scaled_image_width: int,
                                  scaled_image_height: int, image_reference_sheet: Tensor, condition_reference_sheet: Tensor, diffuse):
    """``DatasetGenerator.generate_with_reference_sheet`` (:597-674): render one view, paste it into the LAST cell of the (edited)
    reference sheet, diffuse, cut the cell out, blend by the mask and up-scale.  `original`: the loaded original image
    [H,W,3] replacing the render (:628-630; file I/O stays with the caller), or None.  The two sheets are modified in place,
    as in the reference.  -> dict with the reference's keys."""
    from .ops import resize_bilinear

    render, mask, condition = render_camera(config, graph, camera)
    if original is not None:
        render = original.to(render.device)
    last = config.rows * config.cols - 1
    r0, r1, c0, c1 = cell_window(config, last, scaled_image_width, scaled_image_height)
    render_scaled = resize_bilinear(render, scaled_image_height, scaled_image_width)
    image_reference_sheet[r0:r1, c0:c1, :] = render_scaled
    mask_reference_sheet = torch.zeros_like(condition_reference_sheet)
    mask_scaled = resize_bilinear(mask, scaled_image_height, scaled_image_width, threshold=True, out=mask_reference_sheet[r0:r1, c0:c1, :]) > 0.5
    condition_scaled = resize_bilinear(condition, scaled_image_height, scaled_image_width, out=condition_reference_sheet[r0:r1, c0:c1, :])
    edited_sheet = diffuse(image_reference_sheet, image_reference_sheet, mask_reference_sheet, condition_reference_sheet)
    edited_scaled = edited_sheet[r0:r1, c0:c1, :].to(render.device)
    edited_scaled = edited_scaled * mask_scaled + render_scaled * (~mask_scaled)
    H, W = render.shape[0], render.shape[1]
    edited = resize_bilinear(edited_scaled, config.height or H, config.width or W)
    return {"render": render, "mask": mask, "condition": condition, "edited": edited, "render_scaled": render_scaled,
            "mask_scaled": mask_scaled, "condition_scaled": condition_scaled, "edited_scaled": edited_scaled}
