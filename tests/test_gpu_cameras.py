"""Row a5 for the cameras the reference takes from the ORIGINAL dataset (its default: `cameras = original_dataset.cameras`,
/root/reference/signerf/datasetgenerator/datasetgenerator.py:274-275, :331, :356-358; signerf_trainer.py:222): per-camera
intrinsics, OPENCV distortion parameters, camera types, explicit coords -- the HIP ray generation (``sn_generate_rays_camera``) against the oracle's
restatement of nerfstudio's ``_generate_rays_from_coords`` + ``radial_and_tangential_undistort`` ([NS-RECALL], oracle/nerfacto.py)."""
import pytest
import torch

from helpers import make_model, oracle_config, rmse, small_config
from oracle import nerfacto as onf
from signerf_amd import Cameras, CameraType, SceneBox, scene

pytestmark = pytest.mark.gpu

# k1 k2 k3 k4 p1 p2: a phone-like lens, a strong barrel, tangential only, a lens whose Jacobian degenerates inside the frame
DISTORTIONS = [
    [0.05, -0.02, 0.0, 0.0, 0.001, -0.002],
    [-0.28, 0.09, -0.01, 0.002, 0.0, 0.0],
    [0.0, 0.0, 0.0, 0.0, 0.01, 0.02],
    [-1.5, 0.3, 0.0, 0.0, 0.05, 0.0],
]


def _cam(gpu, H, W, dist=None, ctype=CameraType.PERSPECTIVE, cam=1):
    c2w = scene.benchmark_cameras(8)
    fx, fy, cx, cy = 0.9 * W, 0.95 * W, W / 2 + 0.25, H / 2 - 0.5
    cams = Cameras(c2w[:, :3], fx, fy, cx, cy, W, H, distortion_params=None if dist is None else torch.tensor(dist), camera_type=ctype).to(gpu)
    return cams[cam], (c2w[cam, :3], fx, fy, cx, cy, H, W)


def test_zero_distortion_is_bit_identical_to_pinhole(gpu):
    plain, _ = _cam(gpu, 48, 64)
    zeros, _ = _cam(gpu, 48, 64, dist=[0.0] * 6)
    a, b = plain.generate_rays(0), zeros.generate_rays(0)
    for k in ("origins", "directions", "pixel_area"):
        assert torch.equal(getattr(a, k), getattr(b, k))
    # disable_distortion=True on a distorted camera is the pin-hole bundle too
    dist, _ = _cam(gpu, 48, 64, dist=DISTORTIONS[0])
    c = dist.generate_rays(0, disable_distortion=True)
    assert torch.equal(a.directions, c.directions)
    d = dist.generate_rays(0)
    assert not torch.equal(a.directions, d.directions)


@pytest.mark.parametrize("dist", DISTORTIONS)
@pytest.mark.parametrize("H,W", [(40, 56), (33, 17)])
def test_distorted_perspective_matches_oracle(gpu, dist, H, W):
    cam, args = _cam(gpu, H, W, dist=dist)
    b = cam.generate_rays(camera_indices=0)
    ref = onf.generate_rays(*args, distortion_params=torch.tensor(dist))
    d = b.directions.cpu()
    # strict un-fused IEEE fp32 in the oracle's operand order: the image-plane points are the oracle's to the bit, what remains is the
    # rotation / normalisation tolerance of the pin-hole test
    assert float((d - ref["directions"]).abs().max()) <= 2e-7
    assert torch.equal(b.origins.cpu(), ref["origins"])
    rel = ((b.pixel_area.cpu() - ref["pixel_area"]).abs() / ref["pixel_area"].clamp_min(1e-12))
    assert float(rel.max()) <= 2e-3
    n_ref = ref["directions_norm"]
    assert float(((b.metadata["directions_norm"].cpu() - n_ref).abs() / n_ref).max()) <= 2e-7   # 1-2 ulp (the degenerate lens has norms ~60)


def test_undistorted_image_plane_points(gpu):
    """The un-distorted (u, v) themselves, through a camera with the identity pose: d = (u, v, -1) / n, so u = -d_x / d_z.  The kernel runs the
    Newton iteration in un-fused IEEE fp32 in the oracle's operand order (a numpy fp32 emulation of that order is bit-identical to the oracle);
    what separates the two here is one normalisation and one division: <= 4 ulp."""
    H, W = 24, 40
    eye = torch.eye(4)[:3]
    fx, fy, cx, cy = 30.0, 31.0, 20.5, 11.25
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pts = torch.stack([(xs + 0.5 - cx) / fx, -(ys + 0.5 - cy) / fy], -1).float()
    for dist in DISTORTIONS[:3]:
        cam = Cameras(eye, fx, fy, cx, cy, W, H, distortion_params=torch.tensor(dist)).to(gpu)
        d = cam.generate_rays(0).directions.cpu()
        uv = torch.stack([-d[..., 0] / d[..., 2], -d[..., 1] / d[..., 2]], -1)
        want = onf.radial_and_tangential_undistort(pts, torch.tensor(dist))
        assert float(((uv - want).abs() / want.abs().clamp_min(1e-3)).max()) <= 5e-7


@pytest.mark.parametrize("dist", [None, DISTORTIONS[0]])
def test_fisheye_matches_oracle(gpu, dist):
    cam, args = _cam(gpu, 36, 52, dist=dist, ctype=CameraType.FISHEYE)
    b = cam.generate_rays(0)
    ref = onf.generate_rays(*args, distortion_params=None if dist is None else torch.tensor(dist), camera_type=onf.CAMERA_FISHEYE)
    assert float((b.directions.cpu() - ref["directions"]).abs().max()) <= 1e-6   # sin / cos of the device library vs torch's
    assert float(((b.pixel_area.cpu() - ref["pixel_area"]).abs() / ref["pixel_area"]).max()) <= 5e-3


def test_equirectangular_matches_oracle(gpu):
    """The viewer's third preview camera type (signerf/interface/viewer.py:307-319): fx = fy = H = W / 2, cx = W / 2, cy = H / 2."""
    H, W = 32, 64
    c2w = scene.benchmark_cameras(8)
    cam = Cameras(c2w[:, :3], float(H), float(H), W / 2, H / 2, W, H, camera_type=CameraType.EQUIRECTANGULAR,
                  distortion_params=torch.tensor(DISTORTIONS[0])).to(gpu)[4]        # (lens parameters are ignored for this type)
    b = cam.generate_rays(0)
    ref = onf.generate_rays(c2w[4, :3], float(H), float(H), W / 2, H / 2, H, W, camera_type=onf.CAMERA_EQUIRECTANGULAR)
    assert float((b.directions.cpu() - ref["directions"]).abs().max()) <= 1e-6
    d = b.directions.cpu()
    assert float((d.norm(dim=-1) - 1).abs().max()) <= 1e-6
    # the full sphere: every hemisphere of the camera frame is seen
    local = d @ c2w[4, :3, :3]
    assert float(local[..., 2].max()) > 0.9 and float(local[..., 2].min()) < -0.9 and float(local[..., 1].max()) > 0.9


def test_explicit_coords_and_keep_shape(gpu):
    cam, args = _cam(gpu, 40, 56, dist=DISTORTIONS[0])
    g = torch.Generator().manual_seed(0)
    coords = torch.rand(7, 5, 2, generator=g) * torch.tensor([40.0, 56.0])
    b = cam.generate_rays(0, coords=coords.to(gpu))
    ref = onf.generate_rays(*args, distortion_params=torch.tensor(DISTORTIONS[0]), coords=coords)
    assert b.directions.shape == (7, 5, 3) and b.camera_indices.shape == (7, 5, 1)
    assert float((b.directions.cpu() - ref["directions"]).abs().max()) <= 2e-7
    flat = cam.generate_rays(0, keep_shape=False)
    full = cam.generate_rays(0)
    assert flat.directions.shape == (40 * 56, 3) and torch.equal(flat.directions, full.directions.reshape(-1, 3))
    # the full-image bundle IS coords = pixel centres
    ys, xs = torch.meshgrid(torch.arange(40), torch.arange(56), indexing="ij")
    centres = torch.stack([ys, xs], -1).float() + 0.5
    assert torch.equal(cam.generate_rays(0, coords=centres.to(gpu)).directions, full.directions)


def test_distortion_delta_and_per_camera_parameters(gpu):
    c2w = scene.benchmark_cameras(8)
    W, H = 48, 32
    dist = torch.zeros(8, 6)
    dist[3] = torch.tensor(DISTORTIONS[0])
    fx = torch.linspace(40, 60, 8)
    cams = Cameras(c2w[:, :3], fx[:, None], fx[:, None] * 1.1, 24.0, 16.0, W, H, distortion_params=dist).to(gpu)
    b3 = cams.generate_rays(camera_indices=3)
    ref3 = onf.generate_rays(c2w[3, :3], float(fx[3]), float(fx[3] * 1.1), 24.0, 16.0, H, W, distortion_params=dist[3])
    assert float((b3.directions.cpu() - ref3["directions"]).abs().max()) <= 2e-7
    assert int(b3.camera_indices.min()) == 3 and int(b3.camera_indices.max()) == 3
    # camera 2 has zero parameters; a delta turns it into camera 3's lens
    b2 = cams.generate_rays(camera_indices=2, distortion_params_delta=dist[3])
    ref2 = onf.generate_rays(c2w[2, :3], float(fx[2]), float(fx[2] * 1.1), 24.0, 16.0, H, W, distortion_params=dist[3])
    assert float((b2.directions.cpu() - ref2["directions"]).abs().max()) <= 2e-7


def test_unsupported_requests_raise(gpu):
    cam, _ = _cam(gpu, 16, 16)
    with pytest.raises(TypeError):
        cam.generate_rays(camera_indices=0, not_an_argument=1)          # r03 swallowed unknown keywords
    with pytest.raises(NotImplementedError):
        cam.generate_rays(camera_indices=0, camera_opt_to_camera=torch.eye(4)[:3])
    ortho, _ = _cam(gpu, 16, 16, ctype=CameraType.ORTHOPHOTO)
    with pytest.raises(NotImplementedError, match="ORTHOPHOTO"):
        ortho.generate_rays(0)
    with pytest.raises(IndexError):
        cam.generate_rays(camera_indices=1)


class ForeignCameras:
    """Stand-in for a nerfstudio ``Cameras`` batch as the dataparser builds it: plain attribute tensors, nothing of this package."""

    def __init__(self, c2w, fx, fy, cx, cy, w, h, dist, ctype):
        b = c2w.shape[0]
        col = lambda v, dt: torch.as_tensor(v, dtype=dt).expand(b).reshape(b, 1).clone()  # noqa: E731
        self.camera_to_worlds = c2w
        self.fx, self.fy, self.cx, self.cy = col(fx, torch.float32), col(fy, torch.float32), col(cx, torch.float32), col(cy, torch.float32)
        self.width, self.height = col(w, torch.int64), col(h, torch.int64)
        self.distortion_params = dist
        self.camera_type = col(ctype, torch.int64)
        self.times = None
        self.metadata = None


def test_render_through_an_adopted_foreign_camera(gpu):
    """`cameras = original_dataset.cameras` (:274-275): a foreign, distorted camera batch adopted by ``Cameras.from_cameras`` renders through the
    two reference calls and matches the oracle on the oracle's own rays."""
    cfg = small_config()
    model, sd = make_model(cfg, gpu)
    c2w = scene.benchmark_cameras(8)[:, :3]
    W, H = 40, 32
    dist = torch.tensor(DISTORTIONS[0]).expand(8, 6).clone()
    foreign = ForeignCameras(c2w.to(gpu), 44.0, 45.0, 20.0, 16.0, W, H, dist.to(gpu), 1)
    cams = Cameras.from_cameras(foreign)
    assert cams.device.type == "cuda" and len(cams) == 8 and cams.distortion_params.shape == (8, 6)
    box = SceneBox(aabb=torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]]))
    bundle = cams[5].generate_rays(camera_indices=0, aabb_box=box)
    model.eval()
    out = model.get_outputs_for_camera_ray_bundle(bundle)
    ref_b = onf.generate_rays(c2w[5], 44.0, 45.0, 20.0, 16.0, H, W, distortion_params=dist[5])
    assert float((bundle.directions.cpu() - ref_b["directions"]).abs().max()) <= 2e-7
    tmin, tmax = onf.intersect_aabb_ns(ref_b["origins"].reshape(-1, 3), ref_b["directions"].reshape(-1, 3), box.aabb.flatten())
    ref = onf.get_outputs_for_camera_ray_bundle(sd, oracle_config(cfg), ref_b["origins"], ref_b["directions"],
                                                nears=tmin.reshape(H, W, 1), fars=tmax.reshape(H, W, 1))
    assert rmse(out["rgb"], ref["rgb"]) <= 1e-3 and rmse(out["depth"], ref["depth"]) <= 1e-3   # north_star's gate


def test_degenerate_camera_matrix_takes_nerfstudios_normalisation_floor(gpu):
    """nerfstudio's normalize_with_norm floors the norm at camera_utils._EPS = 4 eps(float64) = 8.88e-16 (r05; r01-r04 restated 1e-20 on both
    sides): a rotation scaled by 1e-16 makes every direction shorter than the floor, so the bundle holds d / _EPS (length ~0.1), not unit
    vectors, and `directions_norm` is the floor itself.  HIP against the oracle; tools/make_nerfstudio_fixture.py puts the same camera
    into the fixture, which decides the constant by data."""
    H, W = 24, 40
    c2w = scene.benchmark_cameras(8)[1, :3].clone()
    c2w[:3, :3] *= 1e-16
    fx, fy, cx, cy = 0.9 * W, 0.95 * W, W / 2 + 0.25, H / 2 - 0.5
    b = Cameras(c2w[None], fx, fy, cx, cy, W, H).to(gpu)[0].generate_rays(camera_indices=0)
    ref = onf.generate_rays(c2w, fx, fy, cx, cy, H, W)
    assert float((b.directions.cpu() - ref["directions"]).abs().max()) <= 2e-7 * float(ref["directions"].abs().max())
    assert torch.equal(b.metadata["directions_norm"].cpu(), ref["directions_norm"])
    assert abs(float(ref["directions_norm"].max()) - onf.NORMALIZE_EPS) < 1e-22
    assert 0.05 < float(ref["directions"].norm(dim=-1).mean()) < 0.5
