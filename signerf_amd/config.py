"""Model configuration, field-for-field the subset of nerfstudio 1.0.2's ``NerfactoModelConfig`` that shapes an
eval-mode render, plus the SIGNeRF overrides.

Reference:
  /root/reference/signerf/signerf.py:15-25          SIGNeRFModelConfig(NerfactoModelConfig)
  /root/reference/signerf/signerf_config.py:31-36   eval_num_rays_per_chunk=1<<15, predict_normals=True, average_init_density=0.01
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple, Type


@dataclass
class InstantiateConfig:
    """nerfstudio's config idiom: ``config.setup(**kwargs)`` instantiates ``config._target(config, **kwargs)``."""

    _target: Type = field(default=None, repr=False)

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


def _default_proposal_args() -> List[Dict]:
    return [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ]


@dataclass
class NerfactoModelConfig(InstantiateConfig):
    """nerfacto defaults (A0)."""

    _target: Type = field(default_factory=lambda: _nerfacto_model(), repr=False)
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: str = "last_sample"
    hidden_dim: int = 64
    hidden_dim_color: int = 64
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    num_proposal_iterations: int = 2
    proposal_net_args_list: List[Dict] = field(default_factory=_default_proposal_args)
    proposal_initial_sampler: str = "piecewise"
    use_average_appearance_embedding: bool = True
    appearance_embed_dim: int = 32
    camera_optimizer_mode: str = "SO3xR3"
    """nerfstudio's ``camera_optimizer.mode`` [NS]: "off" leaves the model without the ``camera_opt`` param group.  Training-time
    only -- the eval render never applies pose adjustments."""
    predict_normals: bool = False
    compute_normals: str = "lazy"
    """When ``predict_normals`` is set, the outputs hold "normals" and "pred_normals" (row a16).  "lazy" (default): they are
    rendered -- by a separate kernel -- the first time one of the two keys is read from the returned dict, so
    ``DatasetGenerator.render_camera``, which reads only rgb and depth (datasetgenerator.py:700-701), never pays for them;
    "always": rendered with every call, as nerfstudio does; "never": the keys are absent."""
    normals_bin_reuse_max_mb: int = 1024
    """Behind the proposal sampler a LAZY normals launch re-uses the final sample bins of the colour render instead of running the
    proposal kernel again (1920x1080: 16.7 -> 9.2 ms for the normals).  The price is memory: the render's whole workspace -- ~0.45 GB at
    1920x1080, ~0.14 GB at 800x800 -- stays alive for as long as the returned outputs dict does (until the normals have been read),
    multiplied by the number of outputs dicts a viewer or dataset loop holds on to.  Renders whose workspace is larger than this many MB
    let it go when the call returns and their lazy normals pay the proposal kernel again; 0 = never keep.  ("always" mode frees the
    workspace at the end of the call whatever this says.)"""
    disable_scene_contraction: bool = False
    average_init_density: float = 1.0
    eval_num_rays_per_chunk: int = 4096
    implementation: str = "torch"
    """Which nerfstudio semantics to reproduce: "torch" (the CPU-runnable fallback; the parity target of the render path) or
    "tcnn" (tiny-cuda-nn: its hash-grid indexing, bias-free MLPs and SH input convention, SURVEY.md A7/A8/A13 and §8(f) row 2 --
    for checkpoints trained with `ns-train nerfacto`; UNPINNED, see signerf_amd/tcnn_import.py)."""
    num_train_data: int = 50
    """Rows of the appearance embedding table (the new dataset's size; signerf_pipeline.py:110-111 drops the
    trained table, so eval uses the mean of a freshly initialised one)."""
    dense_levels: int = 0
    """Memory budget of the derived gather buffers the HIP library builds beside the uploaded tables (``SnFieldDesc.dense_levels``):
    0 = default (the 11 coarsest levels of the main grid get a de-hashed copy: 1.26 GB per model for nerfacto's grid, + 0.47 GB of x-paired
    tables with proposal nets), -1 = none (~14 % slower renders, no extra memory), 1..12 = that many.  ``ops.debug_layout(model)``
    reports the bytes a handle holds."""
    dense_copy_cap_mb: int = 0
    """Per-level size cap (MB) of those copies; 0 = default (600)."""
    precision: str = "fp16x2"
    """MFMA arithmetic of the tiny MLPs.  "fp16x2" (default): every fp32 operand is carried as an fp16 hi+lo pair and each
    product group is three fp16 MFMAs with fp32 accumulation -- measured error equals the exact path's (2.5e-6 relative on
    density, 3e-7 on colours) at 1.8x its speed; operands beyond +-65504 saturate.  "fp32": exact fp32 MFMA.
    "fp16" (opt-in, ``implementation="tcnn"`` only): single fp16 operands with fp16 activations between the layers, the arithmetic a
    tiny-cuda-nn checkpoint was trained in (FullyFusedMLP) -- ~1e-3 from the fp32-grade render, 1.3x faster; never the default."""


@dataclass
class SIGNeRFModelConfig(NerfactoModelConfig):
    """signerf.py:15-25 + the values signerf_config.py:31-36 sets on it."""

    _target: Type = field(default_factory=lambda: _signerf_model(), repr=False)
    eval_num_rays_per_chunk: int = 1 << 15
    predict_normals: bool = True
    average_init_density: float = 0.01
    use_lpips: bool = True
    use_l1: bool = True
    patch_size: int = 32
    lpips_loss_mult: float = 1.0


def _nerfacto_model():
    from .nerfacto import NerfactoModel

    return NerfactoModel


def _signerf_model():
    from .nerfacto import SIGNeRFModel

    return SIGNeRFModel
