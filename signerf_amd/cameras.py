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
        )

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()._map(lambda t: t[start_idx:end_idx])

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))


def _as_column(x: Union[float, int, Tensor], batch: int, dtype) -> Tensor:
    t = torch.as_tensor(x, dtype=dtype)
    if t.ndim == 0:
        t = t.expand(batch)
    t = t.reshape(batch, -1)
    assert t.shape[1] == 1, "per-camera scalar expected"
    return t.clone()


class Cameras:
    """Pin-hole cameras (no distortion parameters -- the SIGNeRF constructor passes none)."""

    def __init__(self, camera_to_worlds: Tensor, fx, fy, cx, cy, width=None, height=None, _host: Optional[Tensor] = None):
        c2w = torch.as_tensor(camera_to_worlds, dtype=torch.float32)
        self._zero_dim = c2w.ndim == 2
        if self._zero_dim:
            c2w = c2w[None]
        assert c2w.ndim == 3 and c2w.shape[-2:] in ((3, 4), (4, 4)), "camera_to_worlds must be [B,3,4]"
        self.camera_to_worlds_batched = c2w[:, :3, :4].contiguous()
        b = c2w.shape[0]
        self._fx = _as_column(fx, b, torch.float32).to(c2w.device)
        self._fy = _as_column(fy, b, torch.float32).to(c2w.device)
        self._cx = _as_column(cx, b, torch.float32).to(c2w.device)
        self._cy = _as_column(cy, b, torch.float32).to(c2w.device)
        if width is None:
            width = (self._cx * 2).to(torch.int64)
        if height is None:
            height = (self._cy * 2).to(torch.int64)
        self._width = _as_column(width, b, torch.int64).to(c2w.device)
        self._height = _as_column(height, b, torch.int64).to(c2w.device)
        # host mirror [B,18] = c2w(12), fx, fy, cx, cy, width, height: lets generate_rays launch without a device sync.  Built once
        # from the constructor's arguments; indexing and .to() hand their slice of it on (`_host`) instead of reading device
        # tensors back -- cameras[i] in the sheet loops must not block the stream the previous camera renders on.
        if _host is not None:
            self._host = _host
        else:
            self._host = torch.cat([self.camera_to_worlds_batched.reshape(b, 12), self._fx, self._fy, self._cx, self._cy,
                                    self._width.to(torch.float32), self._height.to(torch.float32)], dim=1).detach().cpu()

    # -- nerfstudio-shaped accessors: a 0-dim camera exposes [1] tensors, a batch [B,1] -----------------
    def _view(self, t: Tensor) -> Tensor:
        return t[0] if self._zero_dim else t

    @property
    def camera_to_worlds(self) -> Tensor:
        return self._view(self.camera_to_worlds_batched)

    fx = property(lambda self: self._view(self._fx))
    fy = property(lambda self: self._view(self._fy))
    cx = property(lambda self: self._view(self._cx))
    cy = property(lambda self: self._view(self._cy))
    width = property(lambda self: self._view(self._width))
    height = property(lambda self: self._view(self._height))

    @property
    def device(self):
        return self.camera_to_worlds_batched.device

    @property
    def shape(self):
        return () if self._zero_dim else (self.camera_to_worlds_batched.shape[0],)

    def __len__(self) -> int:
        if self._zero_dim:
            raise TypeError("len() of a 0-dim Cameras")
        return self.camera_to_worlds_batched.shape[0]

    def __getitem__(self, idx) -> "Cameras":
        if isinstance(idx, int):
            sl = slice(idx, idx + 1) if idx != -1 else slice(idx, None)
            cam = Cameras(self.camera_to_worlds_batched[sl], self._fx[sl], self._fy[sl], self._cx[sl], self._cy[sl],
                          self._width[sl], self._height[sl], _host=self._host[sl])
            cam._zero_dim = True
            return cam
        hidx = idx.cpu() if isinstance(idx, Tensor) else idx
        return Cameras(self.camera_to_worlds_batched[idx], self._fx[idx], self._fy[idx], self._cx[idx], self._cy[idx],
                       self._width[idx], self._height[idx], _host=self._host[hidx])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def to(self, device) -> "Cameras":
        cam = Cameras(self.camera_to_worlds_batched.to(device), self._fx, self._fy, self._cx, self._cy, self._width, self._height,
                      _host=self._host)
        cam._zero_dim = self._zero_dim
        return cam

    # -- row a5 ------------------------------------------------------------------------------------------
    def generate_rays(self, camera_indices: int = 0, aabb_box: Optional[SceneBox] = None, obb_box: Optional[OrientedBox] = None,
                      **_unused) -> RayBundle:
        """Full-image ray bundle of one camera, generated on the GPU.

        ``camera_indices`` selects the camera of a batch (a 0-dim camera accepts only 0).  With ``aabb_box`` the
        bundle carries nears/fars from nerfstudio's clamped slab test, and the model's collider is then skipped; ``obb_box`` (the
        viewer's crop, used when no ``aabb_box`` is given) does the same in the box's frame (nerfstudio's ``intersect_obb``).
        """
        if not isinstance(camera_indices, int):
            raise NotImplementedError("only integer camera_indices are supported (the SIGNeRF call site passes 0)")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SignerfHipError("Cameras.generate_rays needs the cameras on the GPU: call .to('cuda') first")
        i = camera_indices
        host = self._host[i].tolist()
        H, W = int(host[17]), int(host[16])
        lib = _lib.load()
        c2w_arr = (C.c_float * 12)(*host[:12])
        with torch.cuda.device(dev):
            origins = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            directions = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            pixel_area = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
            dnorm = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
            nears = fars = None
            aabb_arr = None
            if aabb_box is not None:
                nears = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
                fars = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
                aabb_arr = (C.c_float * 6)(*aabb_box.aabb.detach().to("cpu", torch.float32).reshape(-1).tolist())
            st = lib.sn_generate_rays(c2w_arr, host[12], host[13], host[14], host[15], H, W, _lib.ptr(origins), _lib.ptr(directions),
                                      _lib.ptr(pixel_area), _lib.ptr(dnorm), aabb_arr, _lib.ptr(nears), _lib.ptr(fars),
                                      _lib.current_stream())
            _lib.check(st, None, "sn_generate_rays")
            if aabb_box is None and obb_box is not None:
                pose = torch.eye(4, dtype=torch.float64)
                pose[:3, :3] = obb_box.R.detach().to("cpu", torch.float64)
                pose[:3, 3] = obb_box.T.detach().to("cpu", torch.float64).reshape(3)
                w2b = torch.linalg.inv(pose)[:3].to(torch.float32).reshape(-1).tolist()
                size = obb_box.S.detach().to("cpu", torch.float32).reshape(3).tolist()
                nears = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
                fars = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
                st = lib.sn_intersect_obb(_lib.ptr(origins), _lib.ptr(directions), H * W, (C.c_float * 12)(*w2b), (C.c_float * 3)(*size),
                                          _lib.ptr(nears), _lib.ptr(fars), _lib.current_stream())
                _lib.check(st, None, "sn_intersect_obb")
        cam_idx = torch.full((H, W, 1), i, dtype=torch.int64, device=dev)
        return RayBundle(origins=origins, directions=directions, pixel_area=pixel_area, camera_indices=cam_idx,
                         nears=nears, fars=fars, metadata={"directions_norm": dnorm})
