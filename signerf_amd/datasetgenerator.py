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
import datetime
import time
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib


@dataclass
class DatasetGeneratorConfig:
    """The fields of the reference's DatasetGeneratorConfig (datasetgenerator.py:32-81) except the two sub-configs of components that are
    out of scope (``renderer``: the OpenGL mesh rasteriser; ``diffuser``: the HTTP client -- ``DatasetGenerator`` takes a callable)."""

    path: Path = field(default_factory=lambda: Path("./generations"))
    dataset_name: str = field(default_factory=lambda: "experiment-" + datetime.datetime.now().strftime("%Y%m%d-%H%M%S"))
    downscale_factor: int = 2
    fx: Optional[float] = None
    fy: Optional[float] = None
    cx: Optional[float] = None
    cy: Optional[float] = None
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
    combine_shape_with_depth: bool = False   # shape mode only (mesh rasteriser, out of scope); carried for signature parity


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
    """One camera: rgb, mask, condition (datasetgenerator.py:677-820, aabb mode).  `camera`: a 0-dim ``Cameras`` of this package or
    any object with nerfstudio's camera accessors (adopted: ray generation always runs in the HIP kernel)."""
    camera = _adopt(camera)
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


def generate_reference_sheet(config: DatasetGeneratorConfig, graph, cameras, scaled_image_width: int, scaled_image_height: int, diffuse,
                             render_camera_fn=None):
    """``DatasetGenerator.generate_reference_sheet`` (:470-593).  cameras: the rows*cols-1 reference cameras (indexable);
    diffuse(image, image, mask, condition) -> edited sheet [SH,SW,3] stands for ``self.diffuser.diffuse`` (:559).
    -> (image_sheet, mask_sheet, condition_sheet, edited_sheet, references) as the reference returns them.
    render_camera_fn(config, graph, camera) replaces ``render_camera`` (``DatasetGenerator`` hands out pre-computed views)."""
    if len(cameras) != config.rows * config.cols - 1:
        raise ValueError(f"Camera count {len(cameras)} is not equal to (rows * cols) - 1 = {config.rows * config.cols - 1}")
    # the reference cameras are independent: consecutive ones go to alternating streams (sheet.FrameStreams: the head of one frame fills
    # the wave slots the tail of the previous one leaves idle); the sheet is composed on the caller's stream after the join
    from .sheet import FrameStreams

    render = render_camera_fn or render_camera
    views = []
    with FrameStreams(getattr(graph, "device", None)) as fs:
        for i in range(len(cameras)):
            with fs.frame(i):
                views.append(tuple(fs.keep(t) for t in render(config, graph, cameras[i])))
    image_sheet, mask_sheet, condition_sheet, references = compose_reference_sheet(config, views, scaled_image_width, scaled_image_height)
    edited = diffuse(image_sheet, image_sheet, mask_sheet, condition_sheet)
    H, W = views[0][0].shape[0], views[0][0].shape[1]
    edited = split_reference_sheet(config, edited, image_sheet, mask_sheet, references, scaled_image_width, scaled_image_height,
                                   config.height or H, config.width or W)
    return image_sheet, mask_sheet, condition_sheet, edited, references


def generate_with_reference_sheet(config: DatasetGeneratorConfig, graph, camera, original: Optional[Tensor], scaled_image_width: int,
                                  scaled_image_height: int, image_reference_sheet: Tensor, condition_reference_sheet: Tensor, diffuse,
                                  render_camera_fn=None):
    """``DatasetGenerator.generate_with_reference_sheet`` (:597-674): render one view, paste it into the LAST cell of the (edited)
    reference sheet, diffuse, cut the cell out, blend by the mask and up-scale.  `original`: the loaded original image
    [H,W,3] replacing the render (:628-630; file I/O stays with the caller), or None.  The two sheets are modified in place,
    as in the reference.  -> dict with the reference's keys."""
    from .ops import resize_bilinear

    render, mask, condition = (render_camera_fn or render_camera)(config, graph, camera)
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


# ----------------------------------------------------------------------------------------------------------------------
# The generator loop itself (BASELINE.json configs[4]; SURVEY §8(d) "Config 5", §8(e) "Ordering constraint"):
# ``DatasetGenerator.generate_dataset`` of the reference (datasetgenerator.py:185-393).
# ----------------------------------------------------------------------------------------------------------------------
def identity_diffuse(original_image: Tensor, rendered_image: Tensor, mask_image: Tensor, condition_image: Tensor) -> Tensor:
    """What ``Diffuser.diffuse`` returns when the Stable-Diffusion server cannot be reached
    (/root/reference/signerf/diffuser/diffuser.py:182-185): its first argument -- the same tensor object, not a copy."""
    return original_image


def _adopt(camera):
    """Any camera object with nerfstudio's public accessors -> this package's ``Cameras`` (``original_dataset.cameras`` are nerfstudio
    objects in a real run, datasetgenerator.py:274-275): its rays then come from the HIP kernel and its view can be pre-computed."""
    from .cameras import Cameras

    return Cameras.from_cameras(camera)


def _camera_key(camera) -> bytes:
    """Identity of a 0-dim camera (pose, intrinsics, lens), read from the host mirror: no device sync for this package's cameras."""
    return _adopt(camera)._host.contiguous().numpy().tobytes()


class DatasetGenerator:
    """``DatasetGenerator`` of the reference for ``masking_mode="aabb"``: same constructor arguments, attributes and methods
    (``init_directory``, ``generate_dataset``, ``save_generated_images``, ``generate_reference_sheet``,
    ``generate_with_reference_sheet``, ``render_camera``), the same files on disk.

    What is different, and why (SURVEY §8(e) "Ordering constraint to preserve"): the reference interleaves NeRF renders with diffusion
    calls in one sequential loop (:517-519, :331-338).  The renders depend only on the read-only field and their own camera, so
    ``generate_dataset`` computes ALL of them first -- render + mask + condition of the reference views, the generated views and (when
    merging) the original views, sharded camera i -> rank i mod N over the process group, two frames in flight per GPU, tiles gathered
    to rank 0 -- and only then runs the serial part on rank 0 alone: compose the sheet, diffuse, split, then per view paste into the
    LAST cell of the edited sheet (mutated in place, :643-646), diffuse, cut, blend, save.  The sequence of diffuser calls and every
    tensor handed to them are the reference's.

    ``diffuse(original_image, rendered_image, mask_image, condition_image) -> edited`` stands for ``self.diffuser.diffuse`` (the HTTP
    client is out of scope); the default is what that method returns when the server is unreachable (``identity_diffuse``).
    ``group``: the torch.distributed process group to shard over (default group when initialised, else one process).
    """

    def __init__(self, config: DatasetGeneratorConfig, original_transform_matrix: Optional[Tensor] = None, original_scale_factor: float = 1.0,
                 transform_poses_to_original_space: Optional[Callable[[Tensor], Tensor]] = None, device="cuda",
                 diffuse: Optional[Callable[[Tensor, Tensor, Tensor, Tensor], Tensor]] = None, group=None, write_images: bool = True,
                 save_workers: Optional[int] = None, precompute: bool = True, profile: bool = False,
                 png_compress_level: Optional[int] = None, finish_sync: bool = True, serial_stage_timeout_s: float = 24 * 3600.0,
                 precompute_budget_mb: Optional[int] = None, png_encoder: str = "native") -> None:
        self.config = config
        self.png_encoder = png_encoder   # "native": dataset_io.encode_png (same pixels, parallel on the pool); "pil": the reference's Image.save
        self.finish_sync, self.serial_stage_timeout_s = finish_sync, serial_stage_timeout_s
        self.precompute_budget_mb, self.precompute_skipped = precompute_budget_mb, 0
        self.png_compress_level = png_compress_level   # None: PIL's default, the files the reference writes (dataset_io.GeneratedDataset)
        self.device = device
        self.original_transform_matrix = original_transform_matrix if original_transform_matrix is not None else torch.eye(4)[:3]
        self.original_scale_factor = original_scale_factor
        self.transform_poses_to_original_space = transform_poses_to_original_space
        self.path, self.dataset_name = config.path, config.dataset_name
        self.fx, self.fy, self.cx, self.cy = config.fx, config.fy, config.cx, config.cy
        self.width, self.height, self.downscale_factor = config.width, config.height, config.downscale_factor
        self.masking_mode = config.masking_mode
        self.aabb = torch.tensor([config.aabb_min, config.aabb_max], dtype=torch.float32)
        self.inverse_mask, self.combine_shape_with_depth = config.inverse_mask, config.combine_shape_with_depth
        self.rows, self.cols = config.rows, config.cols
        self.border_width_between_images = config.border_width_between_images
        self.mask_dialation, self.additional_depth_radius, self.manual_depth = config.mask_dialation, config.additional_depth_radius, config.manual_depth
        self.diffuse = diffuse or identity_diffuse
        self.group = group
        if save_workers is None:  # PNG encoding is the slowest stage of the loop: spread it over the host's cores (dataset_io.encode_png releases the GIL)
            import os

            save_workers = max(1, min(32, (os.cpu_count() or 8) // 2))
            try:   # a container's CPU quota, not the host's CPU count, is what the threads get (cgroup v2)
                with open("/sys/fs/cgroup/cpu.max") as f:
                    q, per = f.read().split()[:2]
                if q != "max":
                    save_workers = max(1, min(save_workers, int(float(q) / float(per) + 0.5)))
            except (OSError, ValueError):
                pass
        self.write_images, self.save_workers = write_images, save_workers
        self.precompute, self.profile = precompute, profile
        self.is_synthetic = False
        self.dataset: Optional[Any] = None
        self.dataset_path = self.transforms_path = None
        self._views: Dict[bytes, Tuple[Tensor, int]] = {}
        self.timings: Dict[str, float] = {}

    # -- process group ---------------------------------------------------------------------------------------------------
    def _dist(self) -> Tuple[int, int]:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _tick(self, name: str, t0: float) -> float:
        if self.profile and torch.cuda.is_available():
            torch.cuda.synchronize()
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (now - t0)
        return now

    # -- datasetgenerator.py:146-182 ---------------------------------------------------------------------------------------
    def init_directory(self) -> None:
        from .dataset_io import GeneratedDataset

        self.dataset = GeneratedDataset(self.config.path, self.dataset_name, self.downscale_factor, write_images=self.write_images,
                                        save_workers=self.save_workers, png_compress_level=self.png_compress_level, png_encoder=self.png_encoder)
        self.dataset.init_directory()
        self.dataset_path, self.transforms_path = self.dataset.dataset_path, self.dataset.transforms_path
        for key, d in self.dataset.dirs.items():  # images_path, masks_scaled_path, ... as attributes, like the reference's
            setattr(self, f"{key}_path", d)
        import dataclasses

        import yaml

        # the reference dumps its config OBJECT (a yaml python/object tag of its own class); here the same fields as a plain mapping
        plain = {k: (str(v) if isinstance(v, Path) else (list(v) if isinstance(v, tuple) else v)) for k, v in dataclasses.asdict(self.config).items()}
        (self.dataset_path / "config.yml").write_text(yaml.safe_dump(plain), "utf8")

    # -- the pre-computed render stage -------------------------------------------------------------------------------------
    def precompute_views(self, graph, cameras: List) -> None:
        """render + mask + condition of every camera, sharded over the ranks (``sheet.render_views``: camera i -> rank i mod N, two
        frames in flight per GPU), [n,H,W,5] tiles gathered to rank 0, which keeps them for ``render_camera``."""
        from . import sheet

        t0 = time.perf_counter()
        rank, world = self._dist()
        tiles = sheet.render_views(graph, cameras, self.config, group=self.group, dst=0 if world > 1 else None,
                                   render_camera_fn=lambda cfg, g, cam: render_camera(cfg, g, cam))
        if tiles is not None:
            for i, cam in enumerate(cameras):
                self._views[_camera_key(cam)] = (tiles, i)
        self._tick("render_s", t0)

    def render_camera(self, graph, camera, with_mask: bool = True, with_condition: bool = True, combine_shape_with_depth: bool = False):
        """datasetgenerator.py:677-820 (aabb mode): a pre-computed view when there is one, else rendered now."""
        hit = self._views.get(_camera_key(camera)) if (with_mask and with_condition) else None
        if hit is None:
            return render_camera(self.config, graph, camera, with_mask, with_condition)
        tiles, i = hit
        return tiles[i, :, :, 0:3].contiguous(), tiles[i, :, :, 3:4] > 0.5, tiles[i, :, :, 4:5].contiguous()

    def _render_hook(self, config, graph, camera):
        return self.render_camera(graph, camera, combine_shape_with_depth=self.combine_shape_with_depth)

    # -- datasetgenerator.py:470-593, :597-674 -----------------------------------------------------------------------------
    def generate_reference_sheet(self, graph, cameras, scaled_image_width: int, scaled_image_height: int):
        return generate_reference_sheet(self.config, graph, cameras, scaled_image_width, scaled_image_height, self.diffuse,
                                        render_camera_fn=self._render_hook)

    def generate_with_reference_sheet(self, graph, camera, filename, scaled_image_width: int, scaled_image_height: int,
                                      image_reference_sheet: Tensor, condition_reference_sheet: Tensor) -> Dict[str, Tensor]:
        original = None
        if filename is not None:  # :628-630
            from PIL import Image

            from .dataset_io import image_to_tensor

            original = image_to_tensor(Image.open(filename))
        return generate_with_reference_sheet(self.config, graph, camera, original, scaled_image_width, scaled_image_height,
                                             image_reference_sheet, condition_reference_sheet, self.diffuse, render_camera_fn=self._render_hook)

    # -- datasetgenerator.py:398-468 -----------------------------------------------------------------------------------------
    def save_generated_images(self, idx: int, images: Dict[str, Tensor], camera, current_transforms: Dict[str, Any],
                              is_original: bool = False) -> Dict[str, Any]:
        return self.dataset.save_generated_images(idx, images, camera, current_transforms, is_original)

    # -- datasetgenerator.py:185-393 -----------------------------------------------------------------------------------------
    def generate_dataset(self, graph, reference_camera_to_worlds: Tensor, original_dataset=None,
                         synthetic_camera_to_worlds: Optional[Tensor] = None, merge_with_original_dataset: bool = False) -> None:
        from .cameras import Cameras
        from .ops import resize_bilinear

        if original_dataset is None and synthetic_camera_to_worlds is None:
            raise ValueError("Either original dataset or camera_to_worlds must be given")
        if merge_with_original_dataset and (original_dataset is None or synthetic_camera_to_worlds is None):
            raise ValueError("Original dataset and camera_to_worlds must be given to merge with original dataset")
        rank, world = self._dist()
        self.timings = {}
        if synthetic_camera_to_worlds is not None:
            self.is_synthetic = True
        scaled_image_width = int(self.width // self.downscale_factor)
        scaled_image_height = int(self.height // self.downscale_factor)

        reference_cameras = Cameras(reference_camera_to_worlds, self.fx, self.fy, self.cx, self.cy, self.width, self.height).to(self.device)
        cameras, original_filenames, original_cameras = None, None, None
        if original_dataset is not None:
            # nerfstudio `Cameras` in a real run (per-camera intrinsics, OPENCV distortion, camera type): adopted once, so that their rays
            # come from the HIP kernel and their views shard like the synthetic ones (one read-back of the intrinsics for the batch)
            original_cameras = Cameras.from_cameras(original_dataset.cameras)
            cameras = original_cameras
            original_filenames = original_dataset._dataparser_outputs.image_filenames  # pylint: disable=protected-access
        if synthetic_camera_to_worlds is not None:
            cameras = Cameras(synthetic_camera_to_worlds, self.fx, self.fy, self.cx, self.cy, self.width, self.height)
            original_filenames = [None] * synthetic_camera_to_worlds.shape[0]
        cameras = cameras.to(self.device)

        sync = self._sync_group(world)   # created while every rank is here: the serial stage below may take hours
        # From here on EVERY exit path of EVERY rank goes through `_finish`: the idle ranks wait there with a 24 h timeout, so a
        # rank 0 that raised in init_directory() or in stage 1 -- before the old try block began -- would have parked them.  (r06: and a
        # rank 0 that fails before stage 1 tells the others BEFORE they enter that stage's collectives, see init_error below.)
        try:
            # rank 0 prepares the output directory BEFORE the expensive stage (a missing dependency or an unwritable path fails now).  With
            # the pre-compute stage ahead, a failure here must not leave the other ranks alone in that stage's collectives (they would sit in
            # the budget all-reduce and the gathers until the RCCL / gloo timeout): the error is held, every rank learns of it
            # through the budget reduce below (-1), nobody enters stage 1, rank 0 re-raises and all meet in `_finish`.
            init_error = None
            if rank == 0:
                try:
                    self.init_directory()
                except BaseException as e:   # noqa: BLE001  (re-raised below, after the ranks have agreed)
                    if not (self.precompute and world > 1):
                        raise
                    init_error = e

            # Stage 1 (every rank): all NeRF renders, one gather per image size (an original dataset may hold several).  Views beyond the
            # memory budget are rendered inside the serial loop instead (rank 0 alone): correct, just not sharded.
            self._views = {}
            if self.precompute:
                todo = [reference_cameras[i] for i in range(len(reference_cameras))] + [cameras[i] for i in range(len(cameras))]
                if merge_with_original_dataset:
                    merged = original_cameras.to(graph.device)
                    todo += [merged[i] for i in range(len(merged))]
                groups: Dict[Tuple[int, int], List] = {}
                for c in todo:
                    groups.setdefault((int(c._host[0, 16]), int(c._host[0, 17])), []).append(c)
                # the number of views per gather must be the SAME on every rank or the per-group collectives desynchronise: the ranks
                # agree on the smallest budget (total_memory // 4 per device: equal on a homogeneous node, not assumed)
                budget = self._agreed_budget_bytes(world, failed=init_error is not None)
                if budget < 0:
                    if init_error is not None:
                        raise init_error
                    raise RuntimeError("generate_dataset: rank 0 could not prepare the dataset directory; nothing was rendered on this rank")
                for (w, h), cams in groups.items():
                    per_view = 3 * 5 * 4 * w * h   # the [n,H,W,5] fp32 tiles + the gather buffer + the reorder copy at their peak
                    fit = int(min(len(cams), budget // per_view))
                    if fit < len(cams):
                        self.precompute_skipped += len(cams) - fit
                    if fit > 0:
                        self.precompute_views(graph, cams[:fit])
                        budget -= fit * per_view
            if rank != 0:  # the serial stage belongs to the rank that talks to the diffuser and the disk
                return     # (through the finally below: `_finish`)

            # Stage 2 (rank 0): the reference's sequence
            transforms = self.dataset.new_transforms(self.original_transform_matrix, self.original_scale_factor, self.is_synthetic,
                                                     merge_with_original_dataset)
            t0 = time.perf_counter()
            image_sheet, mask_sheet, condition_sheet, edited_sheet, references = self.generate_reference_sheet(
                graph, reference_cameras, scaled_image_width, scaled_image_height)
            t0 = self._tick("sheet_s", t0)
            refs = self.dataset.dirs["references"]
            self.dataset.save_image(image_sheet, refs / "image_reference_sheet.png")
            self.dataset.save_image(mask_sheet, refs / "mask_reference_sheet.png")
            self.dataset.save_image(condition_sheet, refs / "condition_reference_sheet.png")
            self.dataset.save_image(edited_sheet, refs / "edited_reference_sheet.png")
            edited_image_idx = 0
            transforms["reference_indices"] = []
            for i, camera in enumerate(reference_cameras):
                transforms = self.save_generated_images(edited_image_idx, references[i], camera, transforms)
                transforms["reference_indices"].append(edited_image_idx)
                edited_image_idx += 1
            self.dataset.write_transforms(transforms)
            t0 = self._tick("save_s", t0)

            transforms["generated_indices"] = []
            for i, camera in enumerate(cameras):
                filename = original_filenames[i]
                images = self.generate_with_reference_sheet(graph, camera, filename, scaled_image_width, scaled_image_height, edited_sheet,
                                                            condition_sheet)
                t0 = self._tick("views_s", t0)
                transforms = self.save_generated_images(edited_image_idx, images, camera, transforms, filename is not None)
                transforms["generated_indices"].append(edited_image_idx)
                edited_image_idx += 1
                t0 = self._tick("save_s", t0)
            self.dataset.write_transforms(transforms)
            t0 = self._tick("save_s", t0)

            if merge_with_original_dataset:  # :344-388
                transforms["original_indices"] = []
                merged = original_cameras
                for idx in range(len(merged)):
                    image = original_dataset.get_image_float32(idx).to(graph.device)
                    camera = merged[idx].to(graph.device)
                    render, mask, condition = self.render_camera(graph, camera, combine_shape_with_depth=self.combine_shape_with_depth)
                    mask = ~mask  # the original views do not contain the object
                    images = {"render": render, "mask": mask, "condition": condition, "edited": image,
                              "render_scaled": resize_bilinear(render, scaled_image_height, scaled_image_width),
                              "mask_scaled": resize_bilinear(mask, scaled_image_height, scaled_image_width, threshold=True) > 0.5,
                              "condition_scaled": resize_bilinear(condition, scaled_image_height, scaled_image_width),
                              "edited_scaled": resize_bilinear(image, scaled_image_height, scaled_image_width)}
                    t0 = self._tick("views_s", t0)
                    transforms = self.save_generated_images(edited_image_idx, images, camera, transforms, True)
                    transforms["original_indices"].append(edited_image_idx)
                    edited_image_idx += 1
                    t0 = self._tick("save_s", t0)
                self.dataset.write_transforms(transforms)
                self._tick("save_s", t0)
            self.edited_reference_sheet, self.condition_reference_sheet = edited_sheet, condition_sheet  # (as left by the last view, :643-646)
        finally:  # also when a diffuser call or a write raised: no worker threads or queued host copies left behind
            self._views = {}
            try:
                if rank == 0 and self.dataset is not None:
                    self.dataset.close()
            finally:
                # ... and the idle ranks are released either way: they wait in `_finish` with a 24 h timeout, and a rank 0 that
                # leaves through an exception without meeting them would park them there (outside torchrun nothing kills them)
                self._finish(world, sync)

    # -- end-of-dataset synchronisation --------------------------------------------------------------------------------------------
    def _sync_group(self, world: int):
        """The group the ranks meet in once the dataset is on disk.  Ranks other than 0 are idle for the whole serial stage -- the
        sheet diffusion, one diffuser call per view (seconds each with a real Stable-Diffusion server, datasetgenerator.py:331-338),
        PNG encoding, the merge loop -- which can exceed the default collective timeout (RCCL 10 min, gloo 30 min): a barrier on the
        render group would have the watchdog abort the job before the dataset is written.  So the meeting point is a
        dedicated gloo group (host-side, no GPU watchdog) with ``serial_stage_timeout_s`` (default 24 h), created here while all
        ranks are together; ``finish_sync=False`` skips the meeting altogether (the other ranks return after the gather)."""
        if world <= 1 or not self.finish_sync:
            return None
        import datetime as _dt

        import torch.distributed as dist

        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
        return dist.new_group(ranks=ranks, backend="gloo", timeout=_dt.timedelta(seconds=float(self.serial_stage_timeout_s)),
                              use_local_synchronization=self.group is not None)

    def _agreed_budget_bytes(self, world: int, failed: bool = False) -> int:
        """``_precompute_budget_bytes`` reduced to the minimum over the ranks (one tiny collective per dataset).  ``failed``: this rank
        cannot go on (rank 0's ``init_directory`` raised) -- it contributes -1, so that EVERY rank sees a negative budget and skips stage 1."""
        budget = -1 if failed else self._precompute_budget_bytes()
        if world > 1:
            import torch.distributed as dist

            on_host = dist.get_backend(self.group) == "gloo"
            t = torch.tensor([budget], dtype=torch.int64, device="cpu" if on_host else torch.device(self.device))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            budget = int(t.item())
        return budget

    def _precompute_budget_bytes(self) -> int:
        if self.precompute_budget_mb is not None:
            return int(self.precompute_budget_mb) << 20
        dev = torch.device(self.device)
        if dev.type == "cuda" and torch.cuda.is_available():
            return int(torch.cuda.get_device_properties(dev).total_memory // 4)   # a quarter of the 288 GB part: ~1 800 views at 800 x 800
        return 8 << 30

    def _finish(self, world: int, sync=None) -> None:
        if world > 1 and sync is not None:  # every rank returns once the dataset is on disk
            import torch.distributed as dist

            dist.barrier(group=sync)
            dist.destroy_process_group(sync)
