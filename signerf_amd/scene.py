"""Synthetic random-weight scenes for tests, ``smoke()`` and ``bench.py``.

There are no datasets or checkpoints in this environment; the benchmark scene is a nerfacto field with random
weights chosen so that the image is non-trivial:
  torch.manual_seed(seed); hash tables ~ U(-1,1) (not x1e-3); nn.Linear default init, with the base-MLP weights
  scaled x2 and the colour-head weights x3 (default init alone gives a flat grey image, rgb std 0.018); density
  bias +4 so that sigma = average_init_density * exp(h0) is O(1); appearance table ~ N(0,1) with 50 rows (its mean
  is used).  Resulting 800x800x64 frame: rgb std ~0.15, median-depth std ~0.06; accumulation saturates at 1 because
  the last lindisp bin spans 32..1000 units.
Seeds 0 / 1 / 2 for the main field / proposal net 0 / proposal net 1.
"""

from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor

from .config import NerfactoModelConfig, SIGNeRFModelConfig
from .poses import circle_poses


def synthetic_state_dict(config: NerfactoModelConfig, seed: int = 0, density_bias: float = 4.0, base_gain: float = 2.0,
                         head_gain: float = 3.0) -> Dict[str, Tensor]:
    """CPU fp32 parameters under nerfstudio's torch-path state-dict names."""
    sd: Dict[str, Tensor] = {}

    def hash_mlp(prefix: str, levels: int, log2_t: int, hidden: int, out: int, s: int, gain: float = 1.0):
        g = torch.Generator().manual_seed(s)
        sd[f"{prefix}.encoder.hash_table"] = torch.rand(((2**log2_t) * levels, 2), generator=g) * 2 - 1
        torch.manual_seed(s)
        l0, l1 = torch.nn.Linear(levels * 2, hidden), torch.nn.Linear(hidden, out)
        sd[f"{prefix}.mlp.layers.0.weight"], sd[f"{prefix}.mlp.layers.0.bias"] = l0.weight.detach().clone() * gain, l0.bias.detach().clone()
        sd[f"{prefix}.mlp.layers.1.weight"], sd[f"{prefix}.mlp.layers.1.bias"] = l1.weight.detach().clone() * gain, l1.bias.detach().clone()
        sd[f"{prefix}.mlp.layers.1.bias"][0] += density_bias

    hash_mlp("field.mlp_base", config.num_levels, config.log2_hashmap_size, config.hidden_dim, 16, seed, base_gain)
    torch.manual_seed(seed + 100)
    cin = 16 + 15 + config.appearance_embed_dim
    dims = [cin, config.hidden_dim_color, config.hidden_dim_color, 3]
    for i in range(3):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        sd[f"field.mlp_head.layers.{i}.weight"] = lin.weight.detach().clone() * head_gain
        sd[f"field.mlp_head.layers.{i}.bias"] = lin.bias.detach().clone()
    g = torch.Generator().manual_seed(seed + 200)
    sd["field.embedding_appearance.embedding.weight"] = torch.randn((config.num_train_data, config.appearance_embed_dim), generator=g)
    for i in range(config.num_proposal_iterations):
        a = config.proposal_net_args_list[min(i, len(config.proposal_net_args_list) - 1)]
        hash_mlp(f"proposal_networks.{i}.mlp_base", a["num_levels"], a["log2_hashmap_size"], a["hidden_dim"], 1, seed + 1 + i, base_gain)
    if config.predict_normals:  # row a16: pred-normal MLP 27 -> 64 -> 64 -> 64 and PredNormalsFieldHead's Linear(64, 3)
        torch.manual_seed(seed + 300)
        dims = [12 + 15, 64, 64, 64]
        for i in range(3):
            lin = torch.nn.Linear(dims[i], dims[i + 1])
            sd[f"field.mlp_pred_normals.layers.{i}.weight"] = lin.weight.detach().clone() * head_gain
            sd[f"field.mlp_pred_normals.layers.{i}.bias"] = lin.bias.detach().clone()
        lin = torch.nn.Linear(64, 3)
        sd["field.field_head_pred_normals.net.weight"] = lin.weight.detach().clone() * head_gain
        sd["field.field_head_pred_normals.net.bias"] = lin.bias.detach().clone()
    return sd


def benchmark_config(samples: int = 64) -> SIGNeRFModelConfig:
    """BASELINE.json configs[1]: no proposal nets, `samples` uniform-in-s samples, hash grid L=16."""
    return SIGNeRFModelConfig(num_proposal_iterations=0, num_nerf_samples_per_ray=samples)


def proposal_config() -> SIGNeRFModelConfig:
    """BASELINE.json configs[3]: nerfacto defaults (256 + 96 proposal samples, 48 final)."""
    return SIGNeRFModelConfig()


def benchmark_cameras(n: int = 8) -> Tensor:
    """The GUI-default reference cameras (interface.py:62-71): circle_poses(n, radius 0.5, theta 90, phi (0,300))."""
    return circle_poses(n, torch.device("cpu"), 0.5, 90.0, (0.0, 300.0), [0.0, 0.0, 0.0], [0.0, 0.0, 0.0])
