"""signerf_amd -- MI355X (gfx950) implementation of SIGNeRF's reference-sheet render path.

Hot path: ``Cameras.generate_rays`` -> ``NerfactoModel.get_outputs_for_camera_ray_bundle`` (hand-written HIP kernels behind
the C ABI in include/signerf_hip.h), plus the in-tree helpers either side of it.  See DESIGN.md.
"""

from .cameras import Cameras, CameraType, Frustums, OrientedBox, RayBundle, RaySamples, SceneBox  # noqa: F401
from .config import NerfactoModelConfig, SIGNeRFModelConfig  # noqa: F401
from .intersection import intersect_with_aabb  # noqa: F401
from .nerfacto import FieldHeadNames, NerfactoModel, SIGNeRFModel  # noqa: F401
from .poses import circle_poses, random_sphere_poses  # noqa: F401
