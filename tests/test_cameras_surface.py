"""CPU-side checks of the ``Cameras`` surface (nerfstudio's constructor and accessors, SURVEY §8(b) "Camera/ray side") and known-answer tests of
the oracle's un-distortion restatement.  No GPU: ``generate_rays`` itself is covered by tests/test_gpu_cameras.py."""
import math

import pytest
import torch

from oracle import nerfacto as onf
from signerf_amd import Cameras, CameraType, _lib, scene


def _batch(dist=None, ctype=CameraType.PERSPECTIVE, **kw):
    c2w = scene.benchmark_cameras(8)[:, :3]
    return Cameras(c2w, 50.0, 51.0, 24.0, 16.0, 48, 32, distortion_params=dist, camera_type=ctype, **kw)


def test_constructor_takes_nerfstudios_arguments():
    dist = torch.arange(48, dtype=torch.float32).reshape(8, 6) * 1e-3
    cams = _batch(dist=dist, ctype=[CameraType.PERSPECTIVE] * 7 + [CameraType.FISHEYE], times=torch.linspace(0, 1, 8),
                  metadata={"id": torch.arange(8)[:, None], "note": "kept"})
    assert cams.size == 8 and len(cams) == 8 and cams.shape == (8,)
    assert cams.distortion_params.shape == (8, 6) and cams.camera_type.shape == (8, 1) and cams.times.shape == (8, 1)
    cam = cams[7]
    assert cam.shape == () and cam.size == 1
    assert cam.camera_to_worlds.shape == (3, 4) and cam.fx.shape == (1,) and cam.width.dtype == torch.int64
    assert int(cam.camera_type) == CameraType.FISHEYE.value and torch.equal(cam.distortion_params, dist[7])
    assert float(cam.times) == 1.0 and int(cam.metadata["id"]) == 7 and cam.metadata["note"] == "kept"
    assert cam.fx.item() == 50.0 and cam.height.item() == 32        # datasetgenerator.py:449-461 reads them with .item()
    # the host mirror carries everything generate_rays needs (no device read-back at render time)
    assert cam._host.shape == (1, 26) and cam._host[0, 18] == 2 and cam._host[0, 19] == 1 and torch.equal(cam._host[0, 20:], dist[7])
    sub = cams[torch.tensor([1, 3])]
    assert len(sub) == 2 and torch.equal(sub.distortion_params, dist[[1, 3]])
    # a single lens for the whole batch, and the reference's positional call (no distortion) stay valid
    assert _batch(dist=dist[2]).distortion_params.shape == (8, 6)
    assert _batch().distortion_params is None and int(_batch().camera_type[0]) == 1
    with pytest.raises(ValueError):
        _batch(dist=torch.zeros(8, 4))


class _Foreign:
    """What `original_dataset.cameras` (datasetgenerator.py:274-275) looks like from outside: attribute tensors, a foreign enum."""

    class _Type:
        value = 1

    def __init__(self, zero_dim=False):
        c2w = scene.benchmark_cameras(4)[:, :3]
        col = lambda v, dt: torch.full((4, 1), v, dtype=dt)  # noqa: E731
        self.camera_to_worlds = c2w[0] if zero_dim else c2w
        pick = (lambda t: t[0]) if zero_dim else (lambda t: t)
        self.fx, self.fy, self.cx, self.cy = (pick(col(v, torch.float32)) for v in (60.0, 61.0, 32.0, 24.0))
        self.width, self.height = pick(col(64, torch.int64)), pick(col(48, torch.int64))
        self.distortion_params = pick(torch.full((4, 6), 0.01))
        self.camera_type = pick(col(1, torch.int64))


def test_from_cameras_adopts_foreign_objects():
    cams = Cameras.from_cameras(_Foreign())
    assert isinstance(cams, Cameras) and len(cams) == 4 and cams.distortion_params.shape == (4, 6)
    assert Cameras.from_cameras(cams) is cams
    one = Cameras.from_cameras(_Foreign(zero_dim=True))
    assert one.shape == () and one.width.item() == 64 and one._host.shape == (1, 26)
    with pytest.raises(TypeError, match="not a camera object"):
        Cameras.from_cameras(object())


def test_generate_rays_has_no_cpu_path_and_no_silent_arguments():
    cam = _batch()[0]
    with pytest.raises(_lib.SignerfHipError):
        cam.generate_rays(camera_indices=0)
    with pytest.raises(TypeError):
        cam.generate_rays(camera_indices=0, some_future_argument=True)


def test_rescale_output_resolution():
    cams = _batch()
    cams.rescale_output_resolution(0.5)
    assert cams.fx[0].item() == 25.0 and cams.cy[0].item() == 8.0 and cams.width[0].item() == 24 and cams.height[0].item() == 16
    assert cams._host[0, 12] == 25.0 and cams._host[0, 16] == 24 and cams._host[0, 17] == 16


def test_rescale_keeps_the_host_mirror_equal_to_the_camera_size():
    """the size generate_rays uses (the host mirror) must be the camera's own width / height for EVERY factor -- nerfstudio's
    `(width * s).to(int64)` is an fp32 product, truncated (W = 800, s = 0.0725 -> 58; floor(double(800) * 0.0725) is 57).  Sweep of the
    viewer-style factors k / W and a grid of plain ones."""
    bad = []
    for W, H in ((800, 800), (1920, 1080), (1297, 840)):
        factors = [k / W for k in range(1, W, 7)] + [i / 997.0 for i in range(1, 997, 5)] + [0.0725, 0.1, 0.3, 1.0 / 3.0, 0.7]
        for s in factors:
            cams = Cameras(torch.eye(4)[None, :3], 500.0, 500.0, W / 2, H / 2, W, H)
            cams.rescale_output_resolution(s)
            want = ((torch.tensor([W], dtype=torch.int64) * s).to(torch.int64).item(), (torch.tensor([H], dtype=torch.int64) * s).to(torch.int64).item())
            got_dev = (int(cams.width[0].item()), int(cams.height[0].item()))
            got_host = (int(cams._host[0, 16]), int(cams._host[0, 17]))
            if not (got_dev == want == got_host):
                bad.append((W, H, s, want, got_dev, got_host))
    assert not bad, bad[:5]
    cams = Cameras(torch.eye(4)[None, :3], 500.0, 500.0, 400.0, 400.0, 800, 800)
    cams.rescale_output_resolution(0.0725)
    assert int(cams.width[0].item()) == int(cams._host[0, 16]) == 58


# ---- known-answer tests of the un-distortion restatement (the oracle is unpinned: these anchor it analytically) ---------------------
def _distort(p, k):
    """The forward OPENCV model the Newton iteration inverts (double precision)."""
    x, y = p[..., 0].double(), p[..., 1].double()
    k1, k2, k3, k4, p1, p2 = (float(v) for v in k)
    r = x * x + y * y
    d = 1 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    return torch.stack([d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x), d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y)], -1)


def test_undistort_inverts_the_forward_model():
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(2000, 2, generator=g) - 0.5) * 1.2
    for k in ([0.05, -0.02, 0.0, 0.0, 0.001, -0.002], [-0.2, 0.05, -0.01, 0.001, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.01, 0.02]):
        und = onf.radial_and_tangential_undistort(pts, torch.tensor(k))
        assert float((_distort(und, k) - pts.double()).abs().max()) <= 5e-7   # fp32 Newton, converged
    # zero parameters: exactly the identity (what lets the kernel skip the step for them)
    assert torch.equal(onf.radial_and_tangential_undistort(pts, torch.zeros(6)), pts)


def test_undistort_pure_radial_closed_form():
    """k1 only, on the x axis: x_d = x (1 + k1 x^2) -- solve the cubic in double and compare."""
    k1 = 0.1
    xd = torch.linspace(0.05, 0.8, 16)
    und = onf.radial_and_tangential_undistort(torch.stack([xd, torch.zeros_like(xd)], -1), torch.tensor([k1, 0, 0, 0, 0, 0.0]))
    for a, b in zip(und[:, 0].tolist(), xd.tolist()):
        x = b
        for _ in range(60):
            x = x - (x * (1 + k1 * x * x) - b) / (1 + 3 * k1 * x * x)
        assert abs(a - x) <= 2e-7
    assert float(und[:, 1].abs().max()) == 0.0


def test_fisheye_direction_closed_form():
    """Equidistant fisheye: the image-plane radius IS the angle from the optical axis."""
    eye = torch.eye(4)[:3]
    coords = torch.tensor([[10.5, 30.5], [10.5, 10.5]])   # (y, x): on the axis row, 20 px and 0 px from the principal point
    out = onf.generate_rays(eye, 40.0, 40.0, 10.5, 10.5, 21, 41, camera_type=onf.CAMERA_FISHEYE, coords=coords)
    d = out["directions"][0]
    theta = 20.0 / 40.0
    assert torch.allclose(d, torch.tensor([math.sin(theta), 0.0, -math.cos(theta)]), atol=1e-6)


def test_equirectangular_direction_closed_form():
    """Image centre looks down -z; a quarter of the width to the right looks down +x; the top row looks up (+y)."""
    eye = torch.eye(4)[:3]
    H, W = 32, 64
    coords = torch.tensor([[H / 2, W / 2], [H / 2, 3 * W / 4], [0.0, W / 2]])
    d = onf.generate_rays(eye, float(H), float(H), W / 2, H / 2, H, W, camera_type=onf.CAMERA_EQUIRECTANGULAR, coords=coords)["directions"]
    assert torch.allclose(d[0], torch.tensor([0.0, 0.0, -1.0]), atol=1e-6)
    assert torch.allclose(d[1], torch.tensor([1.0, 0.0, 0.0]), atol=1e-6)
    assert torch.allclose(d[2], torch.tensor([0.0, 1.0, 0.0]), atol=1e-6)
