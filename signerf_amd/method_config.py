"""Plugin registration: the counterpart of /root/reference/signerf/signerf_config.py and of the entry point
``[project.entry-points.'nerfstudio.method_configs'] signerf = 'signerf.signerf_config:signerf_method'``
(/root/reference/pyproject.toml:44-46).

nerfstudio discovers methods through that entry-point group; each entry resolves to a ``MethodSpecification(config, description)``
whose ``config.pipeline.model`` is the model config it instantiates.  nerfstudio is not installable here, so
``MethodSpecification`` below is a field-for-field stand-in of ``nerfstudio.plugins.types.MethodSpecification`` and the trainer /
pipeline / data-manager / generator configs -- outside the render path -- are named, not rebuilt: a maintainer with nerfstudio
installed swaps ONE import in the reference's signerf_config.py and keeps everything else.

What this module pins down is what the render path owns of that registration:
  * the model config and the values signerf_config.py:31-36 sets on it,
  * the optimizer group names signerf_config.py:47-60 attaches to ``Model.get_param_groups()``.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict

from .config import SIGNeRFModelConfig


@dataclass
class MethodSpecification:
    """``nerfstudio.plugins.types.MethodSpecification``: the object an entry point of group ``nerfstudio.method_configs`` yields."""

    config: Any
    description: str = "No description provided"


@dataclass
class AdamOptimizerConfig:
    lr: float = 1e-2
    eps: float = 1e-15


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_final: float = 1e-4
    max_steps: int = 200000


@dataclass
class PipelineStub:
    """Carries the model config where nerfstudio's ``VanillaPipelineConfig.model`` sits; the other members of the reference's
    SIGNeRFPipelineConfig (datamanager, dataset_generator) are the reference's own and stay there."""

    model: SIGNeRFModelConfig = field(default_factory=SIGNeRFModelConfig)


@dataclass
class TrainerStub:
    method_name: str = "signerf"
    pipeline: PipelineStub = field(default_factory=PipelineStub)
    optimizers: Dict[str, Dict[str, Any]] = field(default_factory=dict)


signerf_method = MethodSpecification(
    config=TrainerStub(
        method_name="signerf",
        pipeline=PipelineStub(
            # signerf_config.py:31-36
            model=SIGNeRFModelConfig(eval_num_rays_per_chunk=1 << 15, predict_normals=True, use_lpips=True, average_init_density=0.01),
        ),
        # signerf_config.py:47-60 -- the keys must be the names Model.get_param_groups() returns
        optimizers={
            "proposal_networks": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                                  "scheduler": ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)},
            "fields": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15),
                       "scheduler": ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)},
            "camera_opt": {"optimizer": AdamOptimizerConfig(lr=1e-15, eps=1e-15),
                           "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=5000)},
        },
    ),
    description="SIGNeRF method (high quality) -- eval render on MI355X through libsignerf_hip.so",
)
