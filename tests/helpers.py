"""Shared test helpers: package config -> oracle config, scene construction, error metrics."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import nerfacto as onf  # noqa: E402
from signerf_amd import scene  # noqa: E402
from signerf_amd.config import NerfactoModelConfig  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def oracle_config(cfg: NerfactoModelConfig) -> onf.NerfactoConfig:
    props = tuple(
        onf.HashMLPConfig(a["num_levels"], a.get("base_res", 16), a["max_res"], a["log2_hashmap_size"], 2, a["hidden_dim"], 2, 1)
        for a in cfg.proposal_net_args_list[: cfg.num_proposal_iterations]
    )
    return onf.NerfactoConfig(
        near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        num_proposal_samples_per_ray=tuple(cfg.num_proposal_samples_per_ray),
        num_nerf_samples_per_ray=cfg.num_nerf_samples_per_ray,
        num_proposal_iterations=cfg.num_proposal_iterations,
        eval_num_rays_per_chunk=cfg.eval_num_rays_per_chunk,
        average_init_density=cfg.average_init_density,
        appearance_embed_dim=cfg.appearance_embed_dim,
        hidden_dim_color=cfg.hidden_dim_color,
        sh_remap="torch" if cfg.implementation == "torch" else "tcnn",
        main=onf.HashMLPConfig(cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size, cfg.features_per_level,
                               cfg.hidden_dim, 2, 16),
        proposals=props,
    )


def small_config(**kw):
    """A config with small hash tables so CPU-side tests stay fast; architecture unchanged."""
    base = dict(log2_hashmap_size=14,
                proposal_net_args_list=[
                    {"hidden_dim": 16, "log2_hashmap_size": 12, "num_levels": 5, "max_res": 128, "use_linear": False},
                    {"hidden_dim": 16, "log2_hashmap_size": 12, "num_levels": 5, "max_res": 256, "use_linear": False}])
    base.update(kw)
    from signerf_amd.config import SIGNeRFModelConfig

    return SIGNeRFModelConfig(**base)


def make_model(cfg, device, seed=0, **scene_kw):
    """HIP-backed model with the synthetic scene loaded; returns (model, cpu state dict)."""
    sd = scene.synthetic_state_dict(cfg, seed=seed, **scene_kw)
    model = cfg.setup()
    model.load_state_dict(sd, strict=False)
    model.field.embedding_appearance.embedding.weight.data.copy_(sd["field.embedding_appearance.embedding.weight"])
    model = model.to(device)
    model.eval()
    return model, sd


def rmse(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(torch.sqrt(torch.mean((a.double().cpu() - b.double().cpu()) ** 2)))
