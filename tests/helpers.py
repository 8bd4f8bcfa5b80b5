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


def use_granted_cpus():
    """torch threads = the CPUs the container is GRANTED (cgroup quota), not the host's core count: a GPU box of the pool shows 256 logical
    CPUs and grants 16, and 128 throttled threads run the oracle 2.4x slower than 16 (r06: the GPU suite 404 -> 166 s).  Returns the count."""
    import conftest  # tests/conftest.py (the same rule, applied to every test session)

    q = conftest._granted_cpus()
    if q is not None and q >= 1 and torch.get_num_threads() > int(q + 0.5):
        torch.set_num_threads(max(1, int(q + 0.5)))
    return torch.get_num_threads()


def oracle_config(cfg: NerfactoModelConfig, scene_aabb=None) -> onf.NerfactoConfig:
    grid = "torch" if cfg.implementation == "torch" else "tcnn"
    props = tuple(
        onf.HashMLPConfig(a["num_levels"], a.get("base_res", 16), a["max_res"], a["log2_hashmap_size"], 2, a["hidden_dim"], 2, 1, grid)
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
        mlp_precision="fp16" if getattr(cfg, "precision", "") == "fp16" else "fp32",
        background_color=cfg.background_color,
        proposal_initial_sampler=cfg.proposal_initial_sampler,
        disable_scene_contraction=cfg.disable_scene_contraction,
        scene_aabb=tuple(tuple(float(v) for v in row) for row in scene_aabb) if scene_aabb is not None else ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)),
        main=onf.HashMLPConfig(cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size, cfg.features_per_level,
                               cfg.hidden_dim, 2, 16, grid),
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


def ulp_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in units in the last place between two fp32 tensors (monotone integer mapping of the bit patterns; inf for NaN)."""
    def key(t):
        i = t.detach().cpu().contiguous().view(torch.int32).to(torch.int64)
        return torch.where(i < 0, -(i & 0x7FFFFFFF), i)

    d = (key(a.float()) - key(b.float())).abs().double()
    bad = torch.isnan(a.cpu()) | torch.isnan(b.cpu())
    return torch.where(bad, torch.full_like(d, float("inf")), d)


def depth_error_report(got: torch.Tensor, want: torch.Tensor) -> dict:
    """Scale-free error of a depth-like quantity (nerfacto's eval depths span [0, far_plane = 1000] ray-distance units, so an absolute
    RMSE weighs a pixel at depth 300 by 1e5 x a pixel at depth 0.3): absolute RMSE (north_star's gate) beside the relative error
    |got - want| / |want| and the distance in fp32 ulps."""
    g, w = got.detach().double().cpu().reshape(-1), want.detach().double().cpu().reshape(-1)
    ok = torch.isfinite(g) & torch.isfinite(w)
    g, w = g[ok], w[ok]
    rel = (g - w).abs() / w.abs().clamp_min(1e-30)
    ulp = ulp_distance(got.reshape(-1)[ok], want.reshape(-1)[ok])
    q = lambda t, p: float(torch.quantile(t, p)) if t.numel() else 0.0  # noqa: E731
    return {"n": int(g.numel()), "abs_rmse": float(torch.sqrt(torch.mean((g - w) ** 2))), "abs_max": float((g - w).abs().max()),
            "rel_rmse": float(torch.sqrt(torch.mean(rel**2))), "rel_p50": q(rel, 0.5), "rel_p99": q(rel, 0.99), "rel_max": float(rel.max()),
            "ulp_p50": q(ulp, 0.5), "ulp_p99": q(ulp, 0.99), "ulp_max": float(ulp.max()), "bitwise_equal": int((ulp == 0).sum()),
            "depth_min": float(w.min()), "depth_p50": q(w, 0.5), "depth_max": float(w.max())}


def fmt_report(name: str, r: dict) -> str:
    return (f"{name}: abs rmse {r['abs_rmse']:.2e} (max {r['abs_max']:.2e}) | rel rmse {r['rel_rmse']:.2e} p50 {r['rel_p50']:.1e} p99 {r['rel_p99']:.1e} "
            f"max {r['rel_max']:.1e} | ulp p50 {r['ulp_p50']:.0f} p99 {r['ulp_p99']:.0f} max {r['ulp_max']:.0f} | bitwise equal "
            f"{r['bitwise_equal']}/{r['n']} | values in [{r['depth_min']:.3g}, {r['depth_max']:.3g}], median {r['depth_p50']:.3g}")


# ---- tiny-cuda-nn layout: synthetic checkpoints in the flat-vector layout -------------------------------
def synthetic_tcnn_checkpoint(cfg, seed=0, base_gain=2.0, head_gain=3.0, num_images=50):
    """A random "ns-train nerfacto" style state dict: one flat fp32 vector per tiny-cuda-nn module (network matrices, then grid
    rows), sized from the oracle's level table; bias-free nets, so opacity comes from cfg.average_init_density."""
    from oracle import tcnn_layout as tl

    g = torch.Generator().manual_seed(seed)

    def net(in_dim, width, layers, out_dim, gain):
        parts = []
        for r, c in tl.mlp_shapes(in_dim, width, layers, out_dim):
            parts.append((torch.randn(r, c, generator=g) * (gain / c**0.5)).reshape(-1))
        return torch.cat(parts)

    def grid(levels, base, mx, log2_t):
        meta = tl.grid_meta(levels, base, mx, log2_t)
        return (torch.rand(meta.n_rows * 2, generator=g) * 2 - 1)

    sd = {}
    sd["field.mlp_base.tcnn_encoding.params"] = torch.cat([net(2 * cfg.num_levels, cfg.hidden_dim, 2, 16, base_gain),
                                                           grid(cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size)])
    sd["field.mlp_head.tcnn_encoding.params"] = net(16 + 15 + cfg.appearance_embed_dim, cfg.hidden_dim_color, 3, 3, head_gain)
    for i in range(cfg.num_proposal_iterations):
        a = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
        sd[f"proposal_networks.{i}.mlp_base.tcnn_encoding.params"] = torch.cat(
            [net(2 * a["num_levels"], a["hidden_dim"], 2, 1, base_gain), grid(a["num_levels"], a.get("base_res", 16), a["max_res"], a["log2_hashmap_size"])])
    sd["field.embedding_appearance.embedding.weight"] = torch.randn(num_images, cfg.appearance_embed_dim, generator=g)
    if cfg.predict_normals:  # plain Network 27 -> 64 -> 64 -> 64 + the (torch) PredNormalsFieldHead
        sd["field.mlp_pred_normals.tcnn_encoding.params"] = net(12 + 15, 64, 3, 64, head_gain)
        sd["field.field_head_pred_normals.net.weight"] = torch.randn(3, 64, generator=g) * (head_gain / 8.0)
        sd["field.field_head_pred_normals.net.bias"] = torch.randn(3, generator=g) * 0.1
    return sd


def oracle_params_from_tcnn(sd, cfg):
    """The oracle's own unpacking of such a checkpoint (oracle/tcnn_layout.py) -- independent of signerf_amd/tcnn_import.py."""
    from oracle import tcnn_layout as tl

    out = {"field.embedding_appearance.embedding.weight": sd["field.embedding_appearance.embedding.weight"]}

    def stack(prefix, levels, base, mx, log2_t, width, out_dim):
        flat = sd[prefix + ".tcnn_encoding.params"]
        n_net = tl.mlp_n_params(2 * levels, width, 2, out_dim)
        for k, v in tl.mlp_unpack(flat[:n_net], 2 * levels, width, 2, out_dim, pad_value=0.0).items():
            out[f"{prefix}.mlp.{k}"] = v
        meta = tl.grid_meta(levels, base, mx, log2_t)
        out[f"{prefix}.encoder.tcnn_grid"] = flat[n_net:].reshape(meta.n_rows, 2)

    stack("field.mlp_base", cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size, cfg.hidden_dim, 16)
    for k, v in tl.mlp_unpack(sd["field.mlp_head.tcnn_encoding.params"], 16 + 15 + cfg.appearance_embed_dim, cfg.hidden_dim_color, 3, 3,
                              pad_value=1.0).items():
        out[f"field.mlp_head.{k}"] = v
    for i in range(cfg.num_proposal_iterations):
        a = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
        stack(f"proposal_networks.{i}.mlp_base", a["num_levels"], a.get("base_res", 16), a["max_res"], a["log2_hashmap_size"], a["hidden_dim"], 1)
    if "field.mlp_pred_normals.tcnn_encoding.params" in sd:
        # the oracle keeps the library's own input order (oracle/tcnn_layout.frequency_encoding), the importer permutes to this
        # package's: two independent routes to the same function
        for k, v in tl.mlp_unpack(sd["field.mlp_pred_normals.tcnn_encoding.params"], 12 + 15, 64, 3, 64, pad_value=1.0).items():
            out[f"field.mlp_pred_normals.{k}"] = v
        out["field.field_head_pred_normals.net.weight"] = sd["field.field_head_pred_normals.net.weight"]
        out["field.field_head_pred_normals.net.bias"] = sd["field.field_head_pred_normals.net.bias"]
    return out
