"""``Cameras`` / ``RayBundle`` / ``SceneBox`` -- the slice of nerfstudio's camera surface that SIGNeRF touches,
backed by the HIP ray-generation kernel.

What the reference calls ("Camera/ray side"):
  Cameras(c2w, fx, fy, cx, cy, width, height)           datasetgenerator.py:267-268,281-283; interface.py:828-838
  .to(device), len(), iteration -> 0-dim cameras         datasetgenerator.py:314,331
  .generate_rays(camera_indices=0, aabb_box=...)         datasetgenerator.py:691
  bundle.origins / bundle.directions  [H,W,3]            datasetgenerator.py:759-760
  .fx/.fy/.cx/.cy/.width/.height (.item()), .camera_to_worlds   datasetgenerator.py:449-461; renderer.py:162-168
"""

from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Dict, Optional, Union

import torch
from torch import Tensor

from . import _lib


@dataclass
class SceneBox:
    """Axis-aligned box; ``aabb`` is [2,3] (min point, max point)."""

    aabb: Tensor


@dataclass
class OrientedBox:
    """nerfstudio's ``OrientedBox`` (the viewer's crop box): rotation ``R`` [3,3], translation ``T`` [3], side lengths ``S`` [3]."""

    R: Tensor
    T: Tensor
    S: Tensor


@dataclass
class RayBundle:
    """A bundle of rays.  For a full-image bundle every tensor is [H,W,C]."""

    origins: Tensor
    directions: Tensor
    pixel_area: Tensor
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    def _map(self, fn) -> "RayBundle":
        return RayBundle(
            origins=fn(self.origins),
            directions=fn(self.directions),
            pixel_area=fn(self.pixel_area),
            camera_indices=None if self.camera_indices is None else fn(self.camera_indices),
            nears=None if self.nears is None else fn(self.nears),
            fars=None if self.fars is None else fn(self.fars),
            metadata={k: fn(v) for k, v in self.metadata.items()},
            times=None if self.times is None else fn(self.times),
        )

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()._map(lambda t: t[start_idx:end_idx])

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))


@dataclass
class Frustums:
    """nerfstudio's ``Frustums`` (``cameras.rays``): the geometry of ray samples.  The Field methods of this package read
    ``get_positions()`` and ``directions`` -- of this class or of nerfstudio's own, which has the same two members."""

    origins: Tensor      # [..., 3]
    directions: Tensor   # [..., 3]
    starts: Tensor       # [..., 1]
    ends: Tensor         # [..., 1]
    pixel_area: Optional[Tensor] = None

    def get_positions(self) -> Tensor:
        """Sample centres o + d (start + end) / 2."""
        return self.origins + self.directions * ((self.starts + self.ends) / 2)

    @property
    def shape(self):
        return self.origins.shape[:-1]


@dataclass
class RaySamples:
    """nerfstudio's ``RaySamples`` as far as a Field reads it: ``frustums`` (+ ``camera_indices``, ``deltas``, unused in eval)."""

    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None

    @property
    def shape(self):
        return self.frustums.shape


def _as_column(x: Union[float, int, Tensor], batch: int, dtype) -> Tensor:
    t = torch.as_tensor(x, dtype=dtype)
    if t.ndim == 0:
        t = t.expand(batch)
    t = t.reshape(batch, -1)
    assert t.shape[1] == 1, "per-camera scalar expected"
    return t.clone()


class CameraType(enum.Enum):
    """nerfstudio's ``CameraType`` values (``nerfstudio.cameras.cameras``, [NS-RECALL] H): what a dataparser stores in
    ``Cameras.camera_type``.  The HIP ray generation implements PERSPECTIVE, FISHEYE and EQUIRECTANGULAR (the three the viewer can preview,
    signerf/interface/viewer.py:307-319); the others are rejected loudly."""

    PERSPECTIVE = 1
    FISHEYE = 2
    EQUIRECTANGULAR = 3
    OMNIDIRECTIONALSTEREO_L = 4
    OMNIDIRECTIONALSTEREO_R = 5
    VR180_L = 6
    VR180_R = 7
    ORTHOPHOTO = 8
    FISHEYE624 = 9


_SUPPORTED_TYPES = (CameraType.PERSPECTIVE.value, CameraType.FISHEYE.value, CameraType.EQUIRECTANGULAR.value)
# columns of the host mirror
_H_FX, _H_FY, _H_CX, _H_CY, _H_W, _H_H, _H_TYPE, _H_HASDIST, _H_DIST, _H_COLS = 12, 13, 14, 15, 16, 17, 18, 19, 20, 26


def _camera_type_column(camera_type, batch: int) -> Tensor:
    """int | CameraType (ours or nerfstudio's: anything with ``.value``) | list of those | tensor  ->  int64 [B,1]."""
    if hasattr(camera_type, "value") and not isinstance(camera_type, Tensor):
        camera_type = int(camera_type.value)
    if isinstance(camera_type, (list, tuple)):
        camera_type = torch.tensor([int(getattr(c, "value", c)) for c in camera_type], dtype=torch.int64)
    return _as_column(camera_type, batch, torch.int64)


class Cameras:
    """nerfstudio's ``Cameras`` surface as SIGNeRF uses it, backed by the HIP ray generation.

    Constructor arguments are nerfstudio 1.0.2's, in its order [NS-RECALL, H]:
    ``(camera_to_worlds, fx, fy, cx, cy, width=None, height=None, distortion_params=None, camera_type=PERSPECTIVE, times=None,
    metadata=None)``.  The reference builds its reference / synthetic views positionally without distortion
    (datasetgenerator.py:267-268,281-283) and takes the generated views' cameras from the original dataset by default
    (``cameras = original_dataset.cameras``, :274-275) -- nerfstudio ``Cameras`` with per-camera intrinsics, OPENCV
    ``distortion_params`` [k1 k2 k3 k4 p1 p2] and a ``camera_type``; ``Cameras.from_cameras`` adopts such an object.
    """

    def __init__(self, camera_to_worlds: Tensor, fx, fy, cx, cy, width=None, height=None, distortion_params: Optional[Tensor] = None,
                 camera_type=CameraType.PERSPECTIVE, times: Optional[Tensor] = None, metadata: Optional[Dict] = None,
                 _host: Optional[Tensor] = None):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        self._zero_dim = c2w.ndim == 2
        if self._zero_dim:
            c2w = c2w[None]
        assert c2w.ndim == 3 and c2w.shape[-2:] in ((3, 4), (4, 4)), "camera_to_worlds must be [B,3,4]"
        self.camera_to_worlds_batched = c2w[:, :3, :4].contiguous()
        b = c2w.shape[0]
        dev = c2w.device
        self._fx = _as_column(fx, b, torch.float32).to(dev)
        self._fy = _as_column(fy, b, torch.float32).to(dev)
        self._cx = _as_column(cx, b, torch.float32).to(dev)
        self._cy = _as_column(cy, b, torch.float32).to(dev)
        if width is None:
            width = (self._cx * 2).to(torch.int64)
        if height is None:
            height = (self._cy * 2).to(torch.int64)
        self._width = _as_column(width, b, torch.int64).to(dev)
        self._height = _as_column(height, b, torch.int64).to(dev)
        self._camera_type = _camera_type_column(camera_type, b).to(dev)
        self._distortion = None
        if distortion_params is not None:
            d = torch.as_tensor(distortion_params, dtype=torch.float32)
            if d.shape[-1] != 6:
                raise ValueError("distortion_params must hold 6 values per camera: k1 k2 k3 k4 p1 p2")
            self._distortion = (d.expand(b, 6) if d.ndim == 1 else d.reshape(b, 6)).clone().to(dev)
        self._times = None if times is None else torch.as_tensor(times, dtype=torch.float32).reshape(b, 1).to(dev)
        self._metadata = metadata
        # host mirror [B,26] = c2w(12), fx, fy, cx, cy, width, height, camera_type, has_distortion, distortion(6): lets generate_rays
        # launch without a device sync.  Built once from the constructor's arguments; indexing and .to() hand their slice of it on
        # (`_host`) instead of reading device tensors back -- cameras[i] in the sheet loops must not block the stream the previous
        # camera renders on.
        if _host is not None:
            self._host = _host
        else:
            dist = self._distortion if self._distortion is not None else torch.zeros((b, 6), dtype=torch.float32, device=dev)
            has = torch.full((b, 1), 0.0 if self._distortion is None else 1.0, dtype=torch.float32, device=dev)
            self._host = torch.cat([self.camera_to_worlds_batched.reshape(b, 12), self._fx, self._fy, self._cx, self._cy,
                                    self._width.to(torch.float32), self._height.to(torch.float32),
                                    self._camera_type.to(torch.float32), has, dist], dim=1).detach().cpu()

    @classmethod
    def from_cameras(cls, other) -> "Cameras":
        """Adopts ANY camera object that exposes nerfstudio's public accessors (``camera_to_worlds, fx, fy, cx, cy, width, height``
        and optionally ``distortion_params, camera_type, times, metadata``) -- e.g. ``original_dataset.cameras``
        (datasetgenerator.py:274-275) -- so that its rays come from the HIP kernel and its views can be sharded.  One read-back of
        the intrinsics (the host mirror), for the whole batch."""
        if isinstance(other, cls):
            return other
        missing = [k for k in ("camera_to_worlds", "fx", "fy", "cx", "cy", "width", "height") if not hasattr(other, k)]
        if missing:
            raise TypeError(f"{type(other).__name__} is not a camera object: it has no {', '.join(missing)}")
        ctype = getattr(other, "camera_type", None)
        return cls(other.camera_to_worlds, other.fx, other.fy, other.cx, other.cy, other.width, other.height,
                   distortion_params=getattr(other, "distortion_params", None),
                   camera_type=CameraType.PERSPECTIVE if ctype is None else ctype,
                   times=getattr(other, "times", None), metadata=getattr(other, "metadata", None))

    # -- nerfstudio-shaped accessors: a 0-dim camera exposes [1] tensors, a batch [B,1] -----------------
    def _view(self, t: Optional[Tensor]) -> Optional[Tensor]:
        return t if t is None else (t[0] if self._zero_dim else t)

    @property
    def camera_to_worlds(self) -> Tensor:
        return self._view(self.camera_to_worlds_batched)

    fx = property(lambda self: self._view(self._fx))
    fy = property(lambda self: self._view(self._fy))
    cx = property(lambda self: self._view(self._cx))
    cy = property(lambda self: self._view(self._cy))
    width = property(lambda self: self._view(self._width))
    height = property(lambda self: self._view(self._height))
    image_width = width
    image_height = height
    camera_type = property(lambda self: self._view(self._camera_type))
    distortion_params = property(lambda self: self._view(self._distortion))
    times = property(lambda self: self._view(self._times))
    metadata = property(lambda self: self._metadata)

    @property
    def device(self):
        return self.camera_to_worlds_batched.device

    @property
    def shape(self):
        return () if self._zero_dim else (self.camera_to_worlds_batched.shape[0],)

    @property
    def size(self) -> int:
        """Number of cameras (``original_dataset.cameras.size``, datasetgenerator.py:350)."""
        return self.camera_to_worlds_batched.shape[0]

    def __len__(self) -> int:
        if self._zero_dim:
            raise TypeError("len() of a 0-dim Cameras")
        return self.camera_to_worlds_batched.shape[0]

    def _select(self, idx, hidx, zero_dim: bool) -> "Cameras":
        md = self._metadata
        if md is not None:
            md = {k: (v[idx] if isinstance(v, Tensor) and v.ndim > 0 and v.shape[0] == self.size else v) for k, v in md.items()}
        cam = Cameras(self.camera_to_worlds_batched[idx], self._fx[idx], self._fy[idx], self._cx[idx], self._cy[idx],
                      self._width[idx], self._height[idx], None if self._distortion is None else self._distortion[idx],
                      self._camera_type[idx], None if self._times is None else self._times[idx], md, _host=self._host[hidx])
        cam._zero_dim = zero_dim
        return cam

    def __getitem__(self, idx) -> "Cameras":
        if isinstance(idx, int):
            sl = slice(idx, idx + 1) if idx != -1 else slice(idx, None)
            return self._select(sl, sl, True)
        return self._select(idx, idx.cpu() if isinstance(idx, Tensor) else idx, False)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def to(self, device) -> "Cameras":
        md = self._metadata
        if md is not None:
            md = {k: (v.to(device) if isinstance(v, Tensor) else v) for k, v in md.items()}
        cam = Cameras(self.camera_to_worlds_batched.to(device), self._fx, self._fy, self._cx, self._cy, self._width, self._height,
                      self._distortion, self._camera_type, self._times, md, _host=self._host)
        cam._zero_dim = self._zero_dim
        return cam

    def rescale_output_resolution(self, scaling_factor: float) -> None:
        """nerfstudio's in-place rescale (the viewer's low-resolution frames): intrinsics x factor, size floored [NS-RECALL, H]."""
        s = float(scaling_factor)
        self._fx, self._fy, self._cx, self._cy = self._fx * s, self._fy * s, self._cx * s, self._cy * s
        self._width = (self._width * s).to(torch.int64)
        self._height = (self._height * s).to(torch.int64)
        host = self._host.clone()
        host[:, _H_FX:_H_CY + 1] = host[:, _H_FX:_H_CY + 1] * s
        # the SAME arithmetic as the device tensors above -- nerfstudio's `(width * scaling_factor).to(torch.int64)`: an int64 tensor times a
        # Python float is an fp32 product, truncated -- so that the bundle generate_rays builds from the mirror has the camera's own
        # width / height (floor(double(W) * s) differs for viewer-style factors, e.g. W = 800, s = 0.0725: 57 vs 58)
        host[:, _H_W] = (host[:, _H_W].to(torch.int64) * s).to(torch.int64).float()
        host[:, _H_H] = (host[:, _H_H].to(torch.int64) * s).to(torch.int64).float()
        self._host = host

    # -- row a5 ------------------------------------------------------------------------------------------
    def generate_rays(self, camera_indices: int = 0, coords: Optional[Tensor] = None, camera_opt_to_camera: Optional[Tensor] = None,
                      distortion_params_delta: Optional[Tensor] = None, keep_shape: Optional[bool] = None, disable_distortion: bool = False,
                      aabb_box: Optional[SceneBox] = None, obb_box: Optional[OrientedBox] = None) -> RayBundle:
        """Ray bundle of one camera, generated on the GPU.  The signature is nerfstudio 1.0.2's [NS-RECALL, H]; an
        argument this implementation cannot honour raises instead of being ignored.

        ``camera_indices`` selects the camera of a batch (a 0-dim camera accepts only 0).  ``coords`` [..., 2] are image coordinates
        (y, x); None = every pixel centre, and the bundle is [H, W].  ``keep_shape=False`` flattens the bundle.  The camera's
        ``distortion_params`` (+ ``distortion_params_delta``) are un-distorted as nerfstudio does unless ``disable_distortion``.  With
        ``aabb_box`` the bundle carries nears / fars from nerfstudio's clamped slab test, and the model's collider is then skipped;
        ``obb_box`` (the viewer's crop, used when no ``aabb_box`` is given) does the same in the box's frame (``intersect_obb``).
        """
        if isinstance(camera_indices, Tensor) and camera_indices.numel() == 1:
            camera_indices = int(camera_indices.item())
        if not isinstance(camera_indices, int):
            raise NotImplementedError("only integer camera_indices are supported (the SIGNeRF call site passes 0)")
        if camera_opt_to_camera is not None:
            raise NotImplementedError("camera_opt_to_camera (training-time pose refinement) is not part of the render path")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SignerfHipError("Cameras.generate_rays needs the cameras on the GPU: call .to('cuda') first")
        i = camera_indices
        if not -self.size <= i < self.size:
            raise IndexError(f"camera index {i} out of range for {self.size} camera(s)")
        host = self._host[i].tolist()
        H, W = int(host[_H_H]), int(host[_H_W])
        ctype = int(host[_H_TYPE])
        if ctype not in _SUPPORTED_TYPES:
            name = CameraType(ctype).name if ctype in [t.value for t in CameraType] else str(ctype)
            raise NotImplementedError(f"camera_type {name} is not supported by the HIP ray generation (PERSPECTIVE, FISHEYE and EQUIRECTANGULAR are)")
        desc = _lib.SnCameraDesc()
        desc.c2w[:] = host[:12]
        desc.fx, desc.fy, desc.cx, desc.cy = host[_H_FX], host[_H_FY], host[_H_CX], host[_H_CY]
        desc.height, desc.width, desc.camera_type = H, W, ctype
        dist = [0.0] * 6
        if not disable_distortion:
            if host[_H_HASDIST]:
                dist = host[_H_DIST:_H_COLS]
            if distortion_params_delta is not None:
                delta = torch.as_tensor(distortion_params_delta, dtype=torch.float32).detach().cpu().reshape(-1)
                if delta.numel() != 6:
                    raise NotImplementedError("distortion_params_delta must hold the 6 values of the selected camera")
                dist = (torch.tensor(dist, dtype=torch.float32) + delta).tolist()
        desc.distortion[:] = dist
        desc.has_distortion = int(any(v != 0.0 for v in dist))  # all-zero parameters: the Newton steps are exactly 0
        lib = _lib.load()
        with torch.cuda.device(dev):
            if coords is not None:
                if coords.shape[-1] != 2:
                    raise ValueError("coords must be [..., 2] image coordinates (y, x)")
                shape = tuple(coords.shape[:-1])
                cflat = coords.to(device=dev, dtype=torch.float32).reshape(-1, 2).contiguous()
                n = cflat.shape[0]
                if n == 0:   # (an empty tensor has a NULL data pointer, which the C ABI reads as "coords == NULL: the full image")
                    z = lambda c, dt=torch.float32: torch.empty((*shape, c), dtype=dt, device=dev)  # noqa: E731
                    box = aabb_box is not None or obb_box is not None
                    return RayBundle(origins=z(3), directions=z(3), pixel_area=z(1), camera_indices=z(1, torch.int64), nears=z(1) if box else None,
                                     fars=z(1) if box else None, metadata={"directions_norm": z(1)}, times=None if self._times is None else z(1))
            else:
                shape, cflat, n = (H, W), None, H * W
            if keep_shape is False:
                shape = (n,)
            new = lambda c: torch.empty((*shape, c), dtype=torch.float32, device=dev)  # noqa: E731
            origins, directions, pixel_area, dnorm = new(3), new(3), new(1), new(1)
            nears = fars = None
            aabb_arr = None
            if aabb_box is not None:
                nears, fars = new(1), new(1)
                aabb_arr = (C.c_float * 6)(*aabb_box.aabb.detach().to("cpu", torch.float32).reshape(-1).tolist())
            st = lib.sn_generate_rays_camera(C.byref(desc), _lib.ptr(cflat), n, _lib.ptr(origins), _lib.ptr(directions),
                                             _lib.ptr(pixel_area), _lib.ptr(dnorm), aabb_arr, _lib.ptr(nears), _lib.ptr(fars),
                                             _lib.current_stream())
            _lib.check(st, None, "sn_generate_rays_camera")
            if aabb_box is None and obb_box is not None:
                pose = torch.eye(4, dtype=torch.float64)
                pose[:3, :3] = obb_box.R.detach().to("cpu", torch.float64)
                pose[:3, 3] = obb_box.T.detach().to("cpu", torch.float64).reshape(3)
                w2b = torch.linalg.inv(pose)[:3].to(torch.float32).reshape(-1).tolist()
                size = obb_box.S.detach().to("cpu", torch.float32).reshape(3).tolist()
                nears, fars = new(1), new(1)
                st = lib.sn_intersect_obb(_lib.ptr(origins), _lib.ptr(directions), n, (C.c_float * 12)(*w2b), (C.c_float * 3)(*size),
                                          _lib.ptr(nears), _lib.ptr(fars), _lib.current_stream())
                _lib.check(st, None, "sn_intersect_obb")
        cam_idx = torch.full((*shape, 1), i % self.size, dtype=torch.int64, device=dev)
        metadata = {"directions_norm": dnorm}
        if self._metadata is not None:  # per-camera metadata rides along, broadcast over the rays [NS-RECALL, M]
            for k, v in self._metadata.items():
                if isinstance(v, Tensor) and v.ndim > 0 and v.shape[0] == self.size:
                    metadata[k] = v[i].to(dev).reshape(*([1] * len(shape)), -1).expand(*shape, -1)
        times = None if self._times is None else self._times[i].reshape(*([1] * len(shape)), 1).expand(*shape, 1)
        return RayBundle(origins=origins, directions=directions, pixel_area=pixel_area, camera_indices=cam_idx,
                         nears=nears, fars=fars, metadata=metadata, times=times)
