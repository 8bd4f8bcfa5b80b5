"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference's own modules.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Writes small .npz files holding inputs and the reference's outputs -- data, not
source.  The reference Python never travels to the GPU box; the fixtures do.

Covered:
  intersect_with_aabb   signerf/utils/intersection.py:5-56
  circle_poses          signerf/utils/poses_generation.py:22-73   (GUI defaults, interface.py:62-71)
  random_sphere_poses   signerf/utils/poses_generation.py:76-134
  tensor_to_image       signerf/utils/image_tensor_converter.py:7-33
  load_previous_experiment_cameras  signerf/utils/load_previous_experiment_cameras.py:12-54
"""

import importlib.util
import json
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference/signerf/utils"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(f"ref_{name}", os.path.join(REF, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present; fixtures can only be regenerated in the build container")
    inter = _load("intersection")
    poses = _load("poses_generation")
    conv = _load("image_tensor_converter")
    prev = _load("load_previous_experiment_cameras")

    # ---- intersect_with_aabb -------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    H, W = 12, 10
    aabb = torch.tensor([[-0.1, -0.1, -0.1], [0.1, 0.1, 0.1]], dtype=torch.float32)  # datasetgenerator.py:58-61
    o = (torch.rand(H, W, 3, generator=g) - 0.5) * 1.2
    d = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)
    # special rows: axis-parallel rays (zero components), rays from inside the box, rays pointing away
    d[0, :, :] = torch.tensor([1.0, 0.0, 0.0])
    d[1, :, :] = torch.tensor([0.0, -1.0, 0.0])
    d[2, :, :] = torch.tensor([0.0, 0.0, 1.0])
    o[3, :, :] = (torch.rand(W, 3, generator=g) - 0.5) * 0.15  # inside the box
    o[4, :, :] = torch.tensor([0.5, 0.0, 0.0])
    d[4, :, :] = torch.nn.functional.normalize(-o[4] + 0.05 * torch.randn(W, 3, generator=g), dim=-1)  # towards box
    d[5, :, :] = -d[4, :, :]
    o[5, :, :] = o[4, :, :]  # away from box
    nears, fars = inter.intersect_with_aabb(o, d, aabb)
    aabb2 = torch.tensor([[-0.3, -0.2, 0.0], [0.1, 0.4, 0.25]], dtype=torch.float32)
    nears2, fars2 = inter.intersect_with_aabb(o, d, aabb2)
    np.savez(os.path.join(OUT, "intersect_with_aabb.npz"), origins=o.numpy(), directions=d.numpy(),
             aabb=aabb.numpy(), nears=nears.numpy(), fars=fars.numpy(),
             aabb2=aabb2.numpy(), nears2=nears2.numpy(), fars2=fars2.numpy())

    # ---- circle_poses (benchmark cameras) -------------------------------------------------
    out = {}
    for size in (5, 8):
        out[f"circle_{size}"] = poses.circle_poses(size, torch.device("cpu"), 0.5, 90.0, (0.0, 300.0),
                                                   [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]).numpy()
    out["circle_6_offset"] = poses.circle_poses(6, torch.device("cpu"), 1.25, 60.0, (30.0, 210.0),
                                                [0.1, -0.2, 0.3], [0.0, 0.05, -0.1]).numpy()
    torch.manual_seed(1)
    out["sphere_seed1_9"] = poses.random_sphere_poses(9, torch.device("cpu"), 0.5, (30.0, 120.0), (0.0, 360.0),
                                                      [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]).numpy()
    torch.manual_seed(7)
    out["sphere_seed7_50"] = poses.random_sphere_poses(50, torch.device("cpu"), 0.8, (10.0, 90.0), (-45.0, 200.0),
                                                       [0.2, 0.0, -0.1], [0.0, 0.1, 0.0]).numpy()
    np.savez(os.path.join(OUT, "poses.npz"), **out)

    # ---- tensor_to_image truncation --------------------------------------------------------
    vals = torch.tensor([0.0, 0.5, 254.9 / 255.0, 1.0, 0.999, 1.0 / 255.0, 0.00392, 127.5 / 255.0, 0.25, 0.75, 1e-7, 0.9961],
                        dtype=torch.float32)
    rgb = vals.reshape(2, 2, 3)
    gray = vals.reshape(4, 3, 1)
    np.savez(os.path.join(OUT, "tensor_to_image.npz"), rgb_in=rgb.numpy(), rgb_out=np.array(conv.tensor_to_image(rgb)),
             gray_in=gray.numpy(), gray_out=np.array(conv.tensor_to_image(gray)),
             back=conv.image_to_tensor(conv.tensor_to_image(rgb)).numpy())

    # ---- transforms.json round trip --------------------------------------------------------
    c2w = poses.circle_poses(4, torch.device("cpu"), 0.5, 90.0, (0.0, 300.0), [0, 0, 0], [0, 0, 0])
    frames = [{"file_path": f"images/{i:05d}.png", "scene_transform_matrix": c2w[i].tolist(),
               "transform_matrix": c2w[i].tolist()} for i in range(4)]
    transforms = {"reference_indices": [0, 2], "generated_indices": [1, 3], "is_synthetic": True,
                  "is_combined": False, "frames": frames}
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "transforms.json")
        with open(p, "w") as f:
            json.dump(transforms, f)
        ref_c2w, syn_c2w, combined = prev.load_previous_experiment_cameras(p)
    with open(os.path.join(OUT, "transforms_roundtrip.json"), "w") as f:
        json.dump({"transforms": transforms, "reference_c2w": ref_c2w.tolist(), "synthetic_c2w": syn_c2w.tolist(),
                   "is_combined": bool(combined)}, f)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
