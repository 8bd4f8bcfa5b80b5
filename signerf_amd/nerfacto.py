"""``NerfactoModel`` / ``SIGNeRFModel`` -- nerfstudio-``Model``-shaped objects whose eval render runs in the HIP library.

The drop-in boundary.  What ``DatasetGenerator.render_camera`` touches
(/root/reference/signerf/datasetgenerator/datasetgenerator.py:691-701):
    graph.render_aabb, graph.eval(), graph.get_outputs_for_camera_ray_bundle(bundle) -> {"rgb","depth",...},
    graph.train(), graph.device
and what ``SIGNeRFPipeline`` touches (/root/reference/signerf/signerf_pipeline.py:93-132,151):
    model.load_state_dict(state, strict=False) with keys under ``field.``, ``proposal_networks.`` (the appearance
    embedding and camera-optimizer keys are deleted by the pipeline before the call), get_training_callbacks,
    param groups "proposal_networks" / "fields" / "camera_opt" (signerf_config.py:47-60).

The modules below hold the parameters under nerfstudio's torch-path state-dict names; their arithmetic lives in
``csrc/`` and is reached through the C ABI -- there is no PyTorch forward and no CPU fallback.
"""

from __future__ import annotations

import re

import ctypes as C
import threading
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib
from .cameras import Frustums, RaySamples, RayBundle, SceneBox  # noqa: F401
from .config import NerfactoModelConfig, SIGNeRFModelConfig

PRECISIONS = {"fp32": 0, "fp16x2": 1, "fp16": 2}   # "fp16": opt-in, implementation="tcnn" only (config.py)
# RGBRenderer's background in EVAL mode [NS]: "last_sample" (nerfacto's default, what SIGNeRF runs), a named constant colour, or "random" --
# a training device: combine_rgb returns the composited colour without a background, i.e. black
BACKGROUNDS = {"last_sample": None, "black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0), "random": (0.0, 0.0, 0.0)}
# NerfactoModelConfig.proposal_initial_sampler [NS] -> SnRenderOpts.spacing_mode: UniformLinDispPiecewiseSampler (the default) or UniformSampler
INITIAL_SAMPLERS = {"piecewise": 0, "uniform": 1}


class _RWLock:
    """Many render calls, or one weight upload.  The C ABI orders uploads against renders ON THE DEVICE, but the host-side sequence
    sn_upload_weights ... sn_finalize_weights leaves the handle un-finalized in between, so a render CALL of another thread (the viewer
    rendering on the shared model while the generator thread reloads weights) must not start inside it.  Writers have preference: a
    waiting upload blocks NEW readers, so a viewer that renders back to back cannot starve it."""

    def __init__(self):
        self._cond = threading.Condition()
        self._readers = 0
        self._writer = False
        self._writers_waiting = 0

    def acquire_read(self):
        with self._cond:
            while self._writer or self._writers_waiting:
                self._cond.wait()
            self._readers += 1

    def release_read(self):
        with self._cond:
            self._readers -= 1
            if self._readers == 0:
                self._cond.notify_all()

    def acquire_write(self):
        with self._cond:
            self._writers_waiting += 1
            try:
                while self._writer or self._readers:
                    self._cond.wait()
            finally:
                self._writers_waiting -= 1
            self._writer = True

    def release_write(self):
        with self._cond:
            self._writer = False
            self._cond.notify_all()


# ------------------------------------------------------------------------------------------------------
# parameter containers (state-dict layout of nerfstudio's "torch" implementation)
# ------------------------------------------------------------------------------------------------------
class HashEncoding(nn.Module):
    """Parameters of nerfstudio's HashEncoding (torch path): ``hash_table`` [L*T, F]."""

    def __init__(self, num_levels: int, min_res: int, max_res: int, log2_hashmap_size: int, features_per_level: int = 2,
                 hash_init_scale: float = 0.001, implementation: str = "torch"):
        super().__init__()
        self.num_levels, self.min_res, self.max_res = num_levels, min_res, max_res
        self.log2_hashmap_size, self.features_per_level = log2_hashmap_size, features_per_level
        # "tcnn": tiny-cuda-nn grid semantics.  The table keeps the uniform [L * 2^log2_T, F] shape; a level
        # that tiny-cuda-nn indexes densely uses the first next_multiple(res^3, 8) rows of its slot (signerf_amd/tcnn_import.py).
        self.implementation = implementation
        table = torch.rand(size=((2**log2_hashmap_size) * num_levels, features_per_level)) * 2 - 1
        self.hash_table = nn.Parameter(table * hash_init_scale)

    @property
    def grid_mode(self) -> int:
        return 0 if self.implementation == "torch" else 1

    def scalings(self) -> Tensor:
        """torch: floor(min_res * growth**level), evaluated exactly as HashEncoding.__init__ does (numpy float64 growth factor
        raised to an int64 torch tensor -> fp32).  tcnn: the library's grid_scale(), exp2f(level * log2f(growth)) * base - 1."""
        growth = np.exp((np.log(self.max_res) - np.log(self.min_res)) / (self.num_levels - 1)) if self.num_levels > 1 else 1
        if self.implementation != "torch":
            log2_g = np.log2(np.float32(growth), dtype=np.float32)
            lv = np.arange(self.num_levels, dtype=np.float32)
            return torch.from_numpy(np.exp2(lv * log2_g, dtype=np.float32) * np.float32(self.min_res) - np.float32(1.0))
        levels = torch.arange(self.num_levels)
        return torch.floor(self.min_res * growth**levels).to(torch.float32)


class MLP(nn.Module):
    """Parameters of nerfstudio's torch MLP: ``layers`` = nn.Linear list (A8)."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: int):
        super().__init__()
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [out_dim]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])


class MLPWithHashEncoding(nn.Module):
    def __init__(self, num_levels, min_res, max_res, log2_hashmap_size, features_per_level, num_layers, layer_width, out_dim,
                 implementation: str = "torch"):
        super().__init__()
        self.encoder = HashEncoding(num_levels, min_res, max_res, log2_hashmap_size, features_per_level, implementation=implementation)
        self.mlp = MLP(num_levels * features_per_level, num_layers, layer_width, out_dim)


class Embedding(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)


class PredNormalsFieldHead(nn.Module):
    """Parameters of nerfstudio's PredNormalsFieldHead: ``net`` = Linear(in_dim, 3) (followed by tanh and L2 normalisation)."""

    def __init__(self, in_dim: int):
        super().__init__()
        self.net = nn.Linear(in_dim, 3)


class CameraOptimizer(nn.Module):
    """Parameters of nerfstudio's ``CameraOptimizer`` [NS]: ``pose_adjustment`` [num_cameras, 6] (so3 + translation), zeros at
    init.  It adjusts TRAINING rays only (``apply_to_raybundle`` is a no-op in eval mode), so the render path never reads it; it
    exists so that the "camera_opt" param group of signerf_config.py:56-59 and the ``camera_optimizer.*`` state-dict keys
    (deleted by signerf_pipeline.py:110-131 before loading, kept by plain nerfacto checkpoints) have their counterpart."""

    def __init__(self, num_cameras: int, mode: str = "SO3xR3"):
        super().__init__()
        self.mode = mode
        if mode != "off":
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6)))

    def get_param_groups(self, param_groups: dict) -> None:
        params = list(self.parameters())
        if self.mode != "off":
            assert len(params) > 0
            param_groups["camera_opt"] = params


try:  # inside a nerfstudio installation the Field outputs are keyed by ITS enum (members of different Enum classes never compare equal)
    from nerfstudio.field_components.field_heads import FieldHeadNames  # type: ignore  # noqa: F401
except Exception:  # pragma: no cover  (nerfstudio is not installed in the build container)
    import enum

    class FieldHeadNames(enum.Enum):
        """nerfstudio's ``FieldHeadNames`` values [NS-RECALL, H] -- the keys of ``Field.forward`` / ``get_outputs``."""

        RGB = "rgb"
        SH = "sh"
        DENSITY = "density"
        NORMALS = "normals"
        PRED_NORMALS = "pred_normals"
        UNCERTAINTY = "uncertainty"
        BACKGROUND_RGB = "background_rgb"
        TRANSIENT_RGB = "transient_rgb"
        TRANSIENT_DENSITY = "transient_density"
        SEMANTICS = "semantics"
        SDF = "sdf"
        ALPHA = "alpha"
        GRADIENT = "gradient"


class _HipField(nn.Module):
    """nerfstudio's ``Field`` call surface [NS-RECALL, H] as thin bindings over the stage entry point ``sn_field_forward_geo`` (the LITERAL
    torch-path arithmetic, eval mode).  signerf/signerf.py:27 subclasses NerfactoModel and so inherits ``model.field`` /
    ``model.proposal_networks[i]`` with these methods; nerfstudio's samplers (``density_fns``) and export tools call them.

        density_fn(positions [...,3])                 -> density [...,1]
        get_density(ray_samples)                      -> (density [...,1], base_mlp_out [...,15] | None)
        get_outputs(ray_samples, density_embedding)   -> {FieldHeadNames.RGB: [...,3]}                      (main field)
        forward(ray_samples, compute_normals=False)   -> {FieldHeadNames.DENSITY: ..., FieldHeadNames.RGB: ...}

    The parameters live in the owning model's HIP handle, so a Field evaluates through its owner (set by ``populate_modules``); there
    is no PyTorch forward and no CPU path."""

    _which = -1  # -1: the main field; i >= 0: proposal network i

    def _bind(self, owner, which: int) -> None:
        import weakref

        object.__setattr__(self, "_owner_ref", weakref.ref(owner))   # not a sub-module: no reference cycle, not in the state dict
        object.__setattr__(self, "_which", which)

    def _owner(self):
        owner = getattr(self, "_owner_ref", lambda: None)()
        if owner is None:
            raise _lib.SignerfHipError("this Field is not attached to a model: its parameters are evaluated through the owning "
                                       "NerfactoModel's HIP handle (there is no PyTorch forward)")
        return owner

    @torch.no_grad()
    def _evaluate(self, positions: Tensor, directions: Optional[Tensor], want_geo: bool):
        from . import ops

        owner = self._owner()
        if positions.shape[-1] != 3:
            raise ValueError("positions must be [..., 3]")
        shape = positions.shape[:-1]
        dev = owner.device
        pos = positions.reshape(-1, 3).to(device=dev, dtype=torch.float32)
        d = None if directions is None else directions.reshape(-1, 3).to(device=dev, dtype=torch.float32)
        if pos.shape[0] == 0:
            z = lambda c: torch.empty((*shape, c), dtype=torch.float32, device=dev)  # noqa: E731
            return z(1), (None if d is None or self._which >= 0 else z(3)), (z(15) if want_geo else None)
        if want_geo:
            density, rgb, geo = ops.field_forward(owner, pos, d, self._which, return_geo=True)
        else:
            (density, rgb), geo = ops.field_forward(owner, pos, d, self._which), None
        return (density.view(*shape, 1), None if rgb is None else rgb.view(*shape, 3), None if geo is None else geo.view(*shape, 15))

    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None) -> Tensor:
        """``Field.density_fn``: density at world positions (the scene contraction / normalisation is part of the field)."""
        return self._evaluate(positions, None, False)[0]

    def get_density(self, ray_samples):
        positions = ray_samples.frustums.get_positions()
        density, _, geo = self._evaluate(positions, None, self._which < 0)
        return density, geo

    def forward(self, ray_samples, compute_normals: bool = False) -> Dict:
        if compute_normals:
            raise NotImplementedError("per-sample normals are not exposed: the composited 'normals' / 'pred_normals' outputs of the model "
                                      "are (NerfactoModel.get_outputs_for_camera_ray_bundle, csrc/sn_normals.h)")
        fr = ray_samples.frustums
        density, rgb, _ = self._evaluate(fr.get_positions(), fr.directions if self._which < 0 else None, False)
        out = {FieldHeadNames.DENSITY: density}
        if rgb is not None:
            out[FieldHeadNames.RGB] = rgb
        return out


class NerfactoField(_HipField):
    """Parameters of nerfstudio's NerfactoField (A6-A9, A13-A15) + its eval-mode call surface (``_HipField``)."""

    def __init__(self, config: NerfactoModelConfig, num_images: int):
        super().__init__()
        self.geo_feat_dim = 15
        self.mlp_base = MLPWithHashEncoding(config.num_levels, config.base_res, config.max_res, config.log2_hashmap_size,
                                            config.features_per_level, 2, config.hidden_dim, 1 + self.geo_feat_dim,
                                            implementation=config.implementation)
        self.embedding_appearance = Embedding(num_images, config.appearance_embed_dim)
        self.mlp_head = MLP(16 + self.geo_feat_dim + config.appearance_embed_dim, 3, config.hidden_dim_color, 3)
        if config.predict_normals:  # row a16: NeRFEncoding(2 frequencies) of the position (12) + geo features -> 64 -> 64 -> 64 -> head
            self.mlp_pred_normals = MLP(12 + self.geo_feat_dim, 3, 64, 64)
            self.field_head_pred_normals = PredNormalsFieldHead(64)

    def get_outputs(self, ray_samples, density_embedding: Optional[Tensor] = None) -> Dict:
        """``NerfactoField.get_outputs``: {RGB: [...,3]} for the samples' positions and directions, eval mode (mean appearance embedding, A14).
        ``density_embedding`` is a function of the positions; the kernel evaluates the whole field from them in one pass, so the
        argument is only checked for its shape (``Field.forward`` always passes the value ``get_density`` returned for the same samples)."""
        fr = ray_samples.frustums
        if density_embedding is not None and tuple(density_embedding.shape) != (*fr.directions.shape[:-1], self.geo_feat_dim):
            raise ValueError(f"density_embedding must be [..., {self.geo_feat_dim}] for these samples")
        _, rgb, _ = self._evaluate(fr.get_positions(), fr.directions, False)
        return {FieldHeadNames.RGB: rgb}


class HashMLPDensityField(_HipField):
    """Parameters of a proposal network (row a9) + ``density_fn`` / ``get_density`` (``_HipField``; ``get_density`` returns
    ``(density, None)`` as nerfstudio's does)."""

    def __init__(self, hidden_dim=16, log2_hashmap_size=17, num_levels=5, max_res=128, base_res=16, features_per_level=2,
                 use_linear=False, implementation="torch", **_):
        super().__init__()
        if use_linear:
            raise NotImplementedError("use_linear proposal nets (a Linear layer instead of the hash grid + MLP) are not supported: nerfacto's and "
                                      "SIGNeRF's proposal_net_args_list set use_linear=False")
        self.mlp_base = MLPWithHashEncoding(num_levels, base_res, max_res, log2_hashmap_size, features_per_level, 2, hidden_dim, 1,
                                            implementation=implementation)


class LazyOutputs(dict):
    """The outputs dict of a render whose "normals" / "pred_normals" entries (row a16) are rendered on first use: reading either
    key, or enumerating the dict (keys / items / values / iteration / len / ``in``), launches the normals kernel once."""

    _PENDING = ("normals", "pred_normals")

    def __init__(self, base: Dict[str, Tensor], producer):
        super().__init__(base)
        self._producer = producer

    def _materialise(self):
        if self._producer is not None:
            producer, self._producer = self._producer, None
            super().update(producer())

    def __missing__(self, key):
        if key in self._PENDING and self._producer is not None:
            self._materialise()
            return super().__getitem__(key)
        raise KeyError(key)

    def get(self, key, default=None):
        if key in self._PENDING:
            self._materialise()
        return super().get(key, default)

    def __contains__(self, key):
        return super().__contains__(key) or (key in self._PENDING and self._producer is not None)

    def keys(self):
        self._materialise()
        return super().keys()

    def items(self):
        self._materialise()
        return super().items()

    def values(self):
        self._materialise()
        return super().values()

    def __iter__(self):
        self._materialise()
        return super().__iter__()

    def __len__(self):
        self._materialise()
        return super().__len__()

    def copy(self):
        self._materialise()
        return dict(self)


# ------------------------------------------------------------------------------------------------------
# the model
# ------------------------------------------------------------------------------------------------------
def _desc_of(enc: HashEncoding, hidden_dim: int, out_dim: int) -> _lib.SnHashMlpDesc:
    d = _lib.SnHashMlpDesc()
    d.num_levels = enc.num_levels
    d.features_per_level = enc.features_per_level
    d.log2_hashmap_size = enc.log2_hashmap_size
    d.hidden_dim = hidden_dim
    d.num_layers = 2
    d.out_dim = out_dim
    d.grid_mode = enc.grid_mode
    sc = enc.scalings().tolist()
    for i in range(_lib.SN_MAX_LEVELS):
        d.scalings[i] = sc[i] if i < len(sc) else 0.0
    return d


class NerfactoModel(nn.Module):
    """Eval-mode nerfacto render on MI355X behind nerfstudio's ``Model`` surface."""

    config: NerfactoModelConfig

    def __init__(self, config: NerfactoModelConfig, scene_box: Optional[SceneBox] = None, num_train_data: Optional[int] = None,
                 **kwargs):
        super().__init__()
        self.config = config
        self.scene_box = scene_box
        self.render_aabb: Optional[SceneBox] = None  # datasetgenerator.py:691 reads this
        self.num_train_data = num_train_data if num_train_data is not None else config.num_train_data
        self.kwargs = kwargs
        self.device_indicator_param = nn.Parameter(torch.empty(0))
        self._handle = C.c_void_p(None)
        self._handle_device = None
        self._handle_half_grid = False   # whether the handle holds the grid's fp16 storage (SnFieldDesc.half_grid)
        self._weights_dirty = True
        self._engine_generation = 0      # bumped whenever the handle's weights are (re)written
        self._weights_lock = threading.Lock()
        self._engine_rw = _RWLock()
        self._grid_cache: Dict = {}
        self._fallback_warned = False
        self.populate_modules()

    # -- module set-up (names = nerfstudio's; signerf.py:32-39 overrides this and calls super) -------------
    def populate_modules(self):
        cfg = self.config
        # the kernels implement the values SIGNeRF runs with (nerfacto's defaults); anything else must not render silently wrong
        if cfg.background_color not in BACKGROUNDS:
            raise NotImplementedError(f"background_color={cfg.background_color!r}: one of {sorted(BACKGROUNDS)} expected")
        if cfg.precision not in PRECISIONS:
            raise NotImplementedError(f"precision={cfg.precision!r}: one of {sorted(PRECISIONS)} expected")
        if cfg.precision == "fp16" and cfg.implementation != "tcnn":
            raise NotImplementedError('precision="fp16" (single fp16 operands, fp16 activations between layers) is the arithmetic of '
                                      'tiny-cuda-nn checkpoints: it needs implementation="tcnn"; the torch-path parity target runs "fp16x2" / "fp32"')
        if cfg.proposal_initial_sampler not in INITIAL_SAMPLERS:
            raise NotImplementedError(f"proposal_initial_sampler={cfg.proposal_initial_sampler!r}: one of {sorted(INITIAL_SAMPLERS)} expected")
        self.field = NerfactoField(cfg, self.num_train_data)
        self.proposal_networks = nn.ModuleList()
        for i in range(cfg.num_proposal_iterations):
            args = cfg.proposal_net_args_list[min(i, len(cfg.proposal_net_args_list) - 1)]
            self.proposal_networks.append(HashMLPDensityField(**args, implementation=cfg.implementation))
        self.field._bind(self, -1)
        for i, net in enumerate(self.proposal_networks):
            net._bind(self, i)
        self.density_fns = [net.density_fn for net in self.proposal_networks]   # nerfstudio's NerfactoModel attribute [NS-RECALL, H]
        self.camera_optimizer = CameraOptimizer(self.num_train_data, cfg.camera_optimizer_mode)

    @property
    def device(self):
        return self.device_indicator_param.device

    @property
    def _has_pred_normals(self) -> bool:
        """``predict_normals=True`` (signerf_config.py:33) gives the field its pred-normal MLP, in both implementations (a tiny-cuda-nn
        checkpoint's flat vector is unpacked by tcnn_import)."""
        return self.config.predict_normals

    # -- plugin surface the pipeline / trainer expect ---------------------------------------------------------
    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """nerfstudio's NerfactoModel.get_param_groups [NS]: the three groups signerf_config.py:47-60 attaches optimizers to."""
        groups = {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}
        self.camera_optimizer.get_param_groups(param_groups=groups)
        return groups

    def get_training_callbacks(self, training_callback_attributes=None) -> List:
        return []

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> dict:
        raise NotImplementedError("training losses are outside the render path (SURVEY.md §2 row 8)")

    # nerfstudio's MLPWithHashEncoding may register its torch-path pair as `model = nn.Sequential(encoder, mlp)` [NS-RECALL, M] -- a checkpoint then
    # holds `<...>.mlp_base.model.0.hash_table` / `<...>.mlp_base.model.1.layers.N.{weight,bias}` instead of (or beside) the `encoder.` / `mlp.` names
    # this package uses.  Both spellings load: an alias is renamed when its canonical key is absent and dropped when it is present (same tensor).
    _KEY_ALIASES = ((re.compile(r"^(.*\.mlp_base)\.model\.0\.(hash_table)$"), r"\1.encoder.\2"),
                    (re.compile(r"^(.*\.mlp_base)\.model\.1\.(layers\.\d+\.(?:weight|bias))$"), r"\1.mlp.\2"))

    @classmethod
    def _canonical_keys(cls, state_dict):
        out = {}
        for k, v in state_dict.items():
            for pat, rep in cls._KEY_ALIASES:
                if pat.match(k):
                    canon = pat.sub(rep, k)
                    if canon not in state_dict:
                        out[canon] = v
                    break
            else:
                out[k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        if any(k.endswith(".params") for k in state_dict):  # a tiny-cuda-nn checkpoint (flat parameter vectors)
            from .tcnn_import import convert_tcnn_state_dict

            state_dict = convert_tcnn_state_dict(state_dict, self.config)
        state_dict = self._canonical_keys(state_dict)
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._weights_dirty = True
        # strict=False is how the reference loads (signerf_pipeline.py:131): a FIELD parameter that is not in the checkpoint keeps its random
        # initialisation without a word from torch, and the render is noise.  The keys the reference strips on purpose are not reported: the
        # appearance table, the camera optimiser -- and the proposal nets when ALL of them are gone, which is SIGNeRFPipeline's default load
        # (load_model_with_proposal_weights=False deletes every `proposal*` key, :126-129; ADVICE r04: that normal flow must not warn).  A
        # checkpoint that holds SOME proposal keys and misses others is reported like a missing field parameter.
        prop_keys = [k for k in self.state_dict() if k.startswith("proposal_networks.")]
        prop_missing = [k for k in out.missing_keys if k.startswith("proposal_networks.")]
        stripped = len(prop_missing) == len(prop_keys)
        lost = [k for k in out.missing_keys if k.startswith(("field.mlp_base.", "field.mlp_head.")) or (not stripped and k.startswith("proposal_networks."))]
        if lost:
            mods = sorted({".".join(k.split(".")[:3 if k.startswith("proposal") else 2]) for k in lost})
            warnings.warn(f"load_state_dict: {len(lost)} render parameters are not in the state dict and keep their random initialisation "
                          f"({', '.join(mods)}; e.g. {lost[0]!r}) -- renders through these modules are meaningless until they are loaded",
                          RuntimeWarning, stacklevel=2)
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._weights_dirty = True
        return out

    def mark_weights_dirty(self):
        """Call after mutating parameters in place (e.g. an optimizer step) so the next render re-uploads them."""
        self._weights_dirty = True

    # -- HIP engine ---------------------------------------------------------------------------------------------
    def _field_desc(self) -> _lib.SnFieldDesc:
        cfg = self.config
        d = _lib.SnFieldDesc()
        d.main_field = _desc_of(self.field.mlp_base.encoder, cfg.hidden_dim, 1 + self.field.geo_feat_dim)
        d.geo_feat_dim = self.field.geo_feat_dim
        d.hidden_dim_color = cfg.hidden_dim_color
        d.appearance_embed_dim = cfg.appearance_embed_dim
        d.sh_levels = 4
        d.sh_remap = 0 if cfg.implementation == "torch" else 1
        d.num_proposals = len(self.proposal_networks)
        for i, net in enumerate(self.proposal_networks):
            d.proposals[i] = _desc_of(net.mlp_base.encoder, net.mlp_base.mlp.layers[0].out_features, 1)
        d.average_init_density = cfg.average_init_density
        d.histogram_padding = 0.01
        # nerfacto.py (nerfstudio): disable_scene_contraction -> the fields normalise positions with the model's scene box instead
        # (SceneBox.get_normalized_positions); nerfstudio's dataparsers hand out [-1, 1]^3 x scene_scale, the default here
        d.disable_scene_contraction = 1 if cfg.disable_scene_contraction else 0
        aabb = self.scene_box.aabb if self.scene_box is not None else torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])
        for k, v in enumerate(aabb.detach().to(torch.float32).cpu().reshape(6).tolist()):
            d.aabb[k] = v
        d.dense_levels = int(getattr(cfg, "dense_levels", 0))
        d.dense_copy_cap_mb = int(getattr(cfg, "dense_copy_cap_mb", 0))
        # the single-fp16 mode keeps the tiny-cuda-nn grid in fp16 storage as well (half the gather bytes; include/signerf_hip.h half_grid)
        d.half_grid = 1 if (cfg.precision == "fp16" and cfg.implementation == "tcnn") else 0
        return d

    @property
    def effective_precision(self) -> str:
        """What ``config.precision`` resolves to for the uploaded parameters ("fp16x2" falls back to "fp32" for a handle whose split-
        precision MLPs cannot be range-conditioned; sn_effective_precision)."""
        self._engine_rw.acquire_read()
        try:
            if not self._handle or self._weights_dirty:
                return self.config.precision
            eff = _lib.load().sn_effective_precision(self._handle, PRECISIONS[self.config.precision], 0)
        finally:
            self._engine_rw.release_read()
        return {0: "fp32", 1: "fp16x2", 2: "fp16"}.get(eff, self.config.precision)

    def _ensure_engine(self):
        lib = _lib.load()
        if self.device.type != "cuda":
            raise _lib.SignerfHipError("NerfactoModel renders on the GPU only: move it with .to('cuda') (no CPU fallback)")
        with self._weights_lock:
            # precision="fp16" set on a model whose handle was created without the grid's fp16 storage: start over with it (once)
            want_half = self.config.precision == "fp16" and self.config.implementation == "tcnn"
            stale_half = bool(self._handle) and want_half and not self._handle_half_grid
            if (self._handle and self._handle_device != self.device) or not self._handle or self._weights_dirty or stale_half:
                # the handle is replaced / its weights rewritten: no render CALL of another thread may hold or take the handle meanwhile
                self._engine_rw.acquire_write()
                try:
                    if self._handle and (self._handle_device != self.device or stale_half):
                        # model.to("cuda:N") after the first render: the handle and all its buffers live on the old GPU -- start over
                        lib.sn_destroy(self._handle)
                        self._handle = C.c_void_p(None)
                        self._weights_dirty = True
                        self._grid_cache.clear()
                    if not self._handle:
                        desc = self._field_desc()
                        with torch.cuda.device(self.device):
                            _lib.check(lib.sn_create(C.byref(desc), C.byref(self._handle)), None, "sn_create")
                        self._handle_device = self.device
                        self._handle_half_grid = bool(desc.half_grid)
                    if self._weights_dirty:
                        self._upload(lib)
                        self._weights_dirty = False
                        self._engine_generation += 1   # (a kept render state of the old weights no longer matches: _render_normals)
                finally:
                    self._engine_rw.release_write()
        if not self._fallback_warned and self.effective_precision != self.config.precision:
            self._fallback_warned = True
            warnings.warn(f"signerf_amd: precision={self.config.precision!r} cannot hold fp32 grade for these parameters; "
                          f"rendering with {self.effective_precision!r}", RuntimeWarning, stacklevel=3)
        return lib

    def _upload(self, lib):
        stream = _lib.current_stream()

        def up(name: str, t: Tensor):
            t = t.detach().to(torch.float32).contiguous()
            buf = t if t.is_cuda else t.cpu()
            _lib.check(lib.sn_upload_weights(self._handle, name.encode(), buf.data_ptr(), buf.numel() * 4, stream),
                       self._handle, f"sn_upload_weights({name})")

        with torch.cuda.device(self.device):
            sd = self.state_dict()
            for k, v in sd.items():
                if (k.endswith("hash_table") or ".mlp.layers." in k or k.startswith("field.mlp_head.layers.")
                        or (self._has_pred_normals and k.startswith(("field.mlp_pred_normals.layers.", "field.field_head_pred_normals.net.")))):
                    up(k, v)
            if self.config.appearance_embed_dim > 0:
                if self.config.use_average_appearance_embedding:
                    app = self.field.embedding_appearance.mean(0)
                else:
                    app = torch.zeros(self.config.appearance_embed_dim, device=self.device)
                up("field.embedding_appearance.mean", app)
            _lib.check(lib.sn_finalize_weights(self._handle, stream), self._handle, "sn_finalize_weights")
            # "fp16x2" is a request: the library conditions the split-precision MLPs into fp16's range from the uploaded parameters and
            # falls back to the exact-fp32 MFMA path for a handle it cannot condition (include/signerf_hip.h, sn_effective_precision)
            self._fallback_warned = False

    def __del__(self):
        try:
            if self._handle:
                _lib.load().sn_destroy(self._handle)
        except Exception:
            pass

    # sampler grids, computed with the same torch ops nerfstudio uses (bit-identical to the reference's)
    def _grids(self, n_levels: int):
        cfg = self.config
        counts = list(cfg.num_proposal_samples_per_ray[:n_levels]) + [cfg.num_nerf_samples_per_ray]
        key = (tuple(counts), str(self.device))
        if key not in self._grid_cache:
            bins0 = torch.linspace(0.0, 1.0, counts[0] + 1)
            us = []
            for m in counts[1:]:
                nb = m + 1
                u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb) + 1.0 / (2 * nb)
                us.append(u.to(self.device))
            self._grid_cache[key] = (bins0.to(self.device), us)
        return self._grid_cache[key]

    def _opts(self, H: int, W: int, lib, single_chunk: bool = False):
        cfg = self.config
        n_levels = cfg.num_proposal_iterations
        o = _lib.SnRenderOpts()
        o.num_proposal_iterations = n_levels
        for i in range(n_levels):
            o.num_proposal_samples[i] = cfg.num_proposal_samples_per_ray[i]
        o.num_nerf_samples = cfg.num_nerf_samples_per_ray
        o.near_plane = cfg.near_plane if self.training else 0.0  # NearFarCollider: eval near = 0 (A3)
        o.far_plane = cfg.far_plane
        # the chunk size only shapes the expected-depth clip bounds (A17): per eval_num_rays_per_chunk rays for a camera bundle
        # (Model.get_outputs_for_camera_ray_bundle's loop), over the whole bundle for Model.get_outputs / forward
        o.chunk_rays = max(H * W, 1) if single_chunk else cfg.eval_num_rays_per_chunk
        o.precision = PRECISIONS[cfg.precision]
        bg = BACKGROUNDS[cfg.background_color]
        o.background_mode = 0 if bg is None else 1
        o.spacing_mode = INITIAL_SAMPLERS[cfg.proposal_initial_sampler]
        stats = getattr(self, "_march_stats", None)   # diagnostics (ops.render_with_march_stats): 3 uint64 counters the kernels add to
        o.march_stats = stats.data_ptr() if stats is not None else None
        for c in range(3):
            o.background_rgb[c] = 0.0 if bg is None else bg[c]
        bins0, us = self._grids(n_levels)
        o.initial_spacing_bins = bins0.data_ptr()
        for i, u in enumerate(us):
            o.pdf_u[i] = u.data_ptr()
        need = lib.sn_workspace_bytes(self._handle, H, W, C.byref(o))
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=self.device)
        o.workspace = ws.data_ptr()
        o.workspace_bytes = ws.numel()
        return o, (ws, bins0, us)

    # -- rows a6-a17 ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """Whole-image render: dict of [H,W,C] fp32 tensors on the bundle's device ("rgb", "accumulation", "depth"
        (median), "expected_depth", "prop_depth_i").  Rays are visited in the reference's row-major order; the
        reference's 32 768-ray chunking only survives in the expected-depth clip bounds (A17)."""
        H, W = camera_ray_bundle.origins.shape[:2]
        return self._with_normals(camera_ray_bundle, H, W, (H, W))

    @torch.no_grad()
    def get_outputs_for_camera(self, camera, obb_box=None) -> Dict[str, Tensor]:
        """nerfstudio's ``Model.get_outputs_for_camera`` [NS], the call the viewer's render thread makes on the shared model
        (/root/reference/signerf/interface/viewer.py:334-336): rays of camera 0, then the whole-image
        render -- the same kernels at the viewer's resolution.  ``obb_box`` (the viewer's crop box) bounds the rays through
        nerfstudio's ``intersect_obb`` (``Cameras.generate_rays(obb_box=...)``)."""
        return self.get_outputs_for_camera_ray_bundle(camera.generate_rays(camera_indices=0, aabb_box=self.render_aabb, obb_box=obb_box))

    @torch.no_grad()
    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """Flat bundle [R,...] -> dict of [R,C]."""
        R = len(ray_bundle)
        return self._with_normals(ray_bundle.flatten(), 1, R, (R,), single_chunk=True)

    def forward(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        return self.get_outputs(ray_bundle)

    def _with_normals(self, b: RayBundle, H: int, W: int, shape, single_chunk: bool = False) -> Dict[str, Tensor]:
        mode = self.config.compute_normals if self.config.predict_normals else "never"
        if mode not in ("lazy", "always", "never"):
            raise ValueError(f"compute_normals must be 'lazy', 'always' or 'never', got {mode!r}")
        # Behind the proposal sampler the normals kernel marches the SAME final bins as the colour render: the render's workspace (which
        # holds them) stays alive with the outputs, and the normals launch reads it instead of running the proposal kernel a second time
        # (SnRenderOpts.reuse_final_bins; 17.5 -> 10 ms at 1920x1080).  The price: that workspace is released when the outputs dict
        # is, not when this call returns.
        keep_bins = mode != "never" and self.config.num_proposal_iterations > 0 and H * W > 0
        # lazy mode: only below config.normals_bin_reuse_max_mb (the workspace then lives as long as the outputs dict does)
        cap = None if mode == "always" else int(self.config.normals_bin_reuse_max_mb) << 20
        raw, state = self._render_ex(b, H, W, single_chunk, keep_state=keep_bins, keep_state_max_bytes=cap)
        out = {k: v.view(*shape, v.shape[-1]) for k, v in raw.items()}
        if mode == "never":
            return out

        held = [state]   # released with the first (only) normals launch, not with the outputs dict

        def producer():
            with torch.no_grad():
                st, held[0] = held[0], None
                return {k: v.view(*shape, v.shape[-1]) for k, v in self._render_normals(b, H, W, st).items()}

        if mode == "always":
            out.update(producer())
            return out
        return LazyOutputs(out, producer)

    def _render_normals(self, b: RayBundle, H: int, W: int, state=None) -> Dict[str, Tensor]:
        """Row a16: "normals" (analytic) and "pred_normals", [H*W,3] each, by the separate normals kernel (csrc/sn_normals.h).
        ``state`` = what ``_render_ex(..., keep_state=True)`` kept of the colour render of the SAME bundle: its options, its workspace
        (the final sample bins are still in it) and an event behind its kernels -- the proposal sampler is then not run again."""
        lib = self._ensure_engine()
        dev = self.device
        if H * W == 0:
            z = torch.empty((0, 3), dtype=torch.float32, device=dev)
            return {"normals": z, "pred_normals": z.clone()} if self._has_pred_normals else {"normals": z}
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        reuse = state is not None and state["generation"] == self._engine_generation
        # the very tensors the colour render marched (the library's workspace stamp compares their addresses: a second fp32 / contiguous
        # copy of a bundle that needed one would be "another bundle")
        origins, directions, nears, fars = state["rays"] if reuse else (f32(b.origins), f32(b.directions), f32(b.nears), f32(b.fars))
        with torch.cuda.device(dev):
            normals = torch.empty((H * W, 3), dtype=torch.float32, device=dev)
            pred = torch.empty((H * W, 3), dtype=torch.float32, device=dev) if self._has_pred_normals else None
            self._engine_rw.acquire_read()
            try:
                st = _lib.SN_ERR_STATE
                if reuse:
                    o, keep = state["opts"], state["keep"]
                    o.reuse_final_bins = 1
                    o.march_stats = None
                    cur = torch.cuda.current_stream(dev)
                    cur.wait_event(state["event"])            # (a viewer thread may read the normals on another stream)
                    keep[0].record_stream(cur)
                    st = lib.sn_render_normals(self._handle, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(nears), _lib.ptr(fars), H, W,
                                               C.byref(o), _lib.ptr(normals), _lib.ptr(pred), _lib.current_stream())
                    if st == _lib.SN_ERR_STATE:
                        # the library found that the workspace no longer holds this frame's bins (weights re-uploaded in between, the
                        # memory handed to another call): not an error for the caller -- the proposal kernel runs again
                        msg = lib.sn_last_error(self._handle)
                        warnings.warn("lazy normals: the kept sample bins could not be re-used (%s); running the proposal sampler again"
                                      % (msg.decode() if msg else ""), RuntimeWarning)
                        reuse = False
                if not reuse:
                    o, keep = self._opts(H, W, lib)   # (sn_workspace_bytes reads the handle: inside the read lock)
                    o.march_stats = None              # (the colour render of this frame has counted the proposal levels already)
                    st = lib.sn_render_normals(self._handle, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(nears), _lib.ptr(fars), H, W,
                                               C.byref(o), _lib.ptr(normals), _lib.ptr(pred), _lib.current_stream())
                _lib.check(st, self._handle, "sn_render_normals")
            finally:
                self._engine_rw.release_read()
        return {"normals": normals, "pred_normals": pred} if pred is not None else {"normals": normals}

    def _render(self, b: RayBundle, H: int, W: int, single_chunk: bool = False) -> Dict[str, Tensor]:
        return self._render_ex(b, H, W, single_chunk)[0]

    def _render_ex(self, b: RayBundle, H: int, W: int, single_chunk: bool = False, keep_state: bool = False, keep_state_max_bytes=None):
        """-> (outputs, state).  state (``keep_state``) = {"opts", "keep" (workspace + grids), "rays", "event", "generation"} for a normals
        launch on the same bundle that re-uses this render's final sample bins (``_render_normals``); None otherwise -- also when the
        workspace is larger than ``keep_state_max_bytes`` (config.normals_bin_reuse_max_mb)."""
        lib = self._ensure_engine()
        if H * W == 0:  # an empty bundle renders to empty outputs, as the reference's chunk loop does
            z = lambda c: torch.empty((0, c), dtype=torch.float32, device=self.device)  # noqa: E731
            out = {"rgb": z(3), "accumulation": z(1), "depth": z(1), "expected_depth": z(1)}
            out.update({f"prop_depth_{i}": z(1) for i in range(self.config.num_proposal_iterations)})
            return out, None
        dev = self.device
        f32 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        origins, directions, nears, fars = f32(b.origins), f32(b.directions), f32(b.nears), f32(b.fars)
        n = H * W
        with torch.cuda.device(dev):
            new = lambda c: torch.empty((n, c), dtype=torch.float32, device=dev)  # noqa: E731
            rgb, depth, acc, exp = new(3), new(1), new(1), new(1)
            props = [new(1) for _ in range(self.config.num_proposal_iterations)]
            pp = [_lib.ptr(p) for p in props] + [None] * (2 - len(props))
            self._engine_rw.acquire_read()
            try:
                o, keep = self._opts(H, W, lib, single_chunk)   # (sn_workspace_bytes reads the handle: inside the read lock)
                st = lib.sn_render_rays(self._handle, _lib.ptr(origins), _lib.ptr(directions), _lib.ptr(nears), _lib.ptr(fars), H, W,
                                        C.byref(o), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(exp), pp[0], pp[1],
                                        _lib.current_stream())
                _lib.check(st, self._handle, "sn_render_rays")
                state = None
                if keep_state and (keep_state_max_bytes is None or keep[0].numel() <= keep_state_max_bytes):
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    state = {"opts": o, "keep": keep, "rays": (origins, directions, nears, fars), "event": ev, "generation": self._engine_generation}
            finally:
                self._engine_rw.release_read()
            # (the workspace and grids in `keep` are consumed by work already enqueued on this stream; the caching allocator is
            # stream-ordered, so letting them go when this frame returns is safe)
        out = {"rgb": rgb, "accumulation": acc, "depth": depth, "expected_depth": exp}
        for i, p in enumerate(props):
            out[f"prop_depth_{i}"] = p
        return out, state


class SIGNeRFModel(NerfactoModel):
    """signerf.py:27-39: same render path; only training losses differ (out of scope here)."""

    config: SIGNeRFModelConfig
