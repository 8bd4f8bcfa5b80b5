"""Import of tiny-cuda-nn ("implementation='tcnn'") nerfacto checkpoints -- SURVEY.md §8(f) row 2.

Real SIGNeRF runs start from an `ns-train nerfacto` checkpoint (/root/reference/README.md:146,170;
signerf/signerf_pipeline.py:93-132 loads it with strict=False), whose fields are tiny-cuda-nn modules: one flat fp32 parameter
vector per module instead of `hash_table` / `layers.i.weight` tensors.  This module unpacks those vectors into the state-dict
layout of signerf_amd.nerfacto (the torch-path names), for a model built with ``implementation="tcnn"`` so that the kernels
evaluate the grid with tiny-cuda-nn's indexing (csrc/sn_device.h, sn_hash_corners_tcnn).

**UNPINNED.**  tinycudann is not installed and no checkpoint fixture exists, so nothing here has been checked against the
real library.  The layout facts are restated independently in oracle/tcnn_layout.py (ASSUMPTIONS there); the tests check this
importer + the kernels against that restatement on synthetic parameter vectors.  What a real checkpoint would have to confirm:
parameter order (network, then grid), row-major (out, in) matrices, the padding values (0 after a grid, 1 for a plain network),
`n_hidden_layers = num_layers - 1`, the state-dict key names (any key ending in ``.params`` is accepted, matched by module
prefix), and for the pred-normal MLP the output order and frequencies of the library's Frequency encoding.  The library computes in fp16 (parameters cast to half, fp16 blends and fp16-accumulated MMA); this path evaluates the
same parameters in fp32-grade arithmetic, so renders agree to fp16 noise, not bit for bit.  ``quantize_fp16=True`` rounds the
parameters through fp16 the way the library's inference copy is.
"""

from __future__ import annotations

import re
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import Tensor


def _next_multiple(v: int, m: int) -> int:
    return ((v + m - 1) // m) * m


def grid_level_table(num_levels: int, base_res: int, max_res: int, log2_hashmap_size: int) -> Tuple[List[int], List[int]]:
    """(resolution per level, row offsets [L+1]) of a tiny-cuda-nn HashGrid configured as nerfstudio configures it."""
    growth = np.exp((np.log(max_res) - np.log(base_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    log2_g = np.log2(np.float32(growth), dtype=np.float32)
    res, offs = [], [0]
    for level in range(num_levels):
        scale = np.float32(np.exp2(np.float32(level) * log2_g, dtype=np.float32)) * np.float32(base_res) - np.float32(1.0)
        r = int(np.ceil(scale)) + 1
        n = min(_next_multiple(min(r**3, (2**32 - 1) // 2), 8), 1 << log2_hashmap_size)
        res.append(r)
        offs.append(offs[-1] + n)
    return res, offs


def _mlp_shapes(in_dim: int, width: int, num_layers: int, out_dim: int) -> List[Tuple[int, int]]:
    shapes = [(width, _next_multiple(in_dim, 16))]
    shapes += [(width, width)] * (num_layers - 2)
    shapes.append((_next_multiple(out_dim, 16), width))
    return shapes


def _unpack_mlp(flat: Tensor, prefix: str, in_dim: int, width: int, num_layers: int, out_dim: int, pad_value: float, out: Dict[str, Tensor]) -> int:
    """Writes {prefix}.layers.i.weight / .bias; returns the number of parameters consumed."""
    o = 0
    shapes = _mlp_shapes(in_dim, width, num_layers, out_dim)
    for i, (r, c) in enumerate(shapes):
        w = flat[o : o + r * c].reshape(r, c)
        o += r * c
        rows = out_dim if i == len(shapes) - 1 else r
        if i == 0:
            out[f"{prefix}.layers.{i}.weight"] = w[:rows, :in_dim].clone()
            # padded input columns see a constant (0 after a grid encoding, 1 for a plain network): they are a bias
            out[f"{prefix}.layers.{i}.bias"] = (w[:rows, in_dim:] * pad_value).sum(dim=1)
        else:
            out[f"{prefix}.layers.{i}.weight"] = w[:rows].clone()
            out[f"{prefix}.layers.{i}.bias"] = torch.zeros(rows, dtype=flat.dtype)
    return o


def _unpack_grid(flat: Tensor, num_levels: int, base_res: int, max_res: int, log2_hashmap_size: int, features: int) -> Tensor:
    """Back-to-back level rows -> the uniform [L * 2^log2_T, F] table (a short level fills the start of its slot)."""
    _, offs = grid_level_table(num_levels, base_res, max_res, log2_hashmap_size)
    if flat.numel() != offs[-1] * features:
        raise ValueError(f"hash-grid vector has {flat.numel()} values, the level table needs {offs[-1] * features}")
    rows = flat.reshape(offs[-1], features)
    T = 1 << log2_hashmap_size
    table = torch.zeros((num_levels * T, features), dtype=flat.dtype)
    for level in range(num_levels):
        n = offs[level + 1] - offs[level]
        table[level * T : level * T + n] = rows[offs[level] : offs[level + 1]]
    return table


def _hits(sd: Dict[str, Tensor], module: str) -> List[str]:
    return [k for k in sd if re.fullmatch(re.escape(module) + r"\.[A-Za-z_]+\.params", k)]


def _find(sd: Dict[str, Tensor], module: str) -> Tensor:
    hits = _hits(sd, module)
    if len(hits) != 1:
        raise KeyError(f"expected exactly one '<{module}>.<attr>.params' vector in the checkpoint, found {hits}")
    return sd[hits[0]].detach().to("cpu", torch.float32).reshape(-1)


def convert_tcnn_state_dict(sd: Dict[str, Tensor], config, quantize_fp16: bool = False) -> Dict[str, Tensor]:
    """tiny-cuda-nn flat vectors -> the state-dict layout of signerf_amd.nerfacto (torch-path names).  Keys that are not tcnn
    parameter vectors pass through unchanged (appearance embedding, camera optimiser, ...)."""
    if config.implementation == "torch":
        raise ValueError('a tiny-cuda-nn checkpoint needs a model built with implementation="tcnn" (its grid indexing differs)')
    out: Dict[str, Tensor] = {k: v for k, v in sd.items() if not k.endswith(".params")}

    def stack(prefix: str, levels: int, base: int, mx: int, log2_t: int, feats: int, width: int, out_dim: int):
        # Partial checkpoints are normal: the reference strips every `proposal*` key before `load_state_dict(strict=False)` when it
        # retrains (signerf_pipeline.py:126-131,141-144).  A module whose vectors are absent is simply not converted -- its keys
        # then show up as missing keys (or raise under strict=True), exactly as for a torch-path checkpoint.
        if not (_hits(sd, prefix) or _hits(sd, f"{prefix}.mlp") or _hits(sd, f"{prefix}.encoder")):
            return
        if _hits(sd, prefix):  # one fused NetworkWithInputEncoding vector: network first, then the grid
            flat = _find(sd, prefix)
            n_net = _unpack_mlp(flat, f"{prefix}.mlp", levels * feats, width, 2, out_dim, 0.0, out)
            grid = flat[n_net:]
        else:                  # separate Encoding and Network modules (<prefix>.encoder.<attr>.params, <prefix>.mlp.<attr>.params)
            net = _find(sd, f"{prefix}.mlp")
            used = _unpack_mlp(net, f"{prefix}.mlp", levels * feats, width, 2, out_dim, 1.0, out)  # a plain Network pads with ones
            if used != net.numel():
                raise ValueError(f"{prefix}.mlp vector has {net.numel()} values, expected {used}")
            grid = _find(sd, f"{prefix}.encoder")
        out[f"{prefix}.encoder.hash_table"] = _unpack_grid(grid, levels, base, mx, log2_t, feats)

    stack("field.mlp_base", config.num_levels, config.base_res, config.max_res, config.log2_hashmap_size, config.features_per_level,
          config.hidden_dim, 16)
    if _hits(sd, "field.mlp_head"):
        head = _find(sd, "field.mlp_head")
        in_dim = 16 + 15 + config.appearance_embed_dim
        used = _unpack_mlp(head, "field.mlp_head", in_dim, config.hidden_dim_color, 3, 3, 1.0, out)
        if used != head.numel():
            raise ValueError(f"colour head vector has {head.numel()} values, expected {used}")
    if _hits(sd, "field.mlp_pred_normals"):
        # predict_normals=True (signerf_config.py:33): MLP [Frequency encoding of the position (12) | geo features (15)] -> 64 -> 64 ->
        # 64, a plain tiny-cuda-nn Network (inputs padded to 32 with ones).  The library's Frequency encoding orders its outputs
        # dimension-major -- (x: sin f0, cos f0, sin f1, cos f1), (y: ...), (z: ...) -- while this package (and nerfstudio's torch
        # NeRFEncoding) orders them [sin(a, k) a-major | cos(a, k) a-major]; the first layer's columns are permuted accordingly.
        # (Its frequencies are pi 2^k, not 2 pi 2^k: the kernel handles that, SnNormalsParams::pe_rev_scale.)
        pn = _find(sd, "field.mlp_pred_normals")
        used = _unpack_mlp(pn, "field.mlp_pred_normals", 12 + 15, 64, 3, 64, 1.0, out)
        if used != pn.numel():
            raise ValueError(f"pred-normal MLP vector has {pn.numel()} values, expected {used}")
        w0 = out["field.mlp_pred_normals.layers.0.weight"]
        perm = [a * 4 + k * 2 + 0 for a in range(3) for k in range(2)] + [a * 4 + k * 2 + 1 for a in range(3) for k in range(2)]
        out["field.mlp_pred_normals.layers.0.weight"] = torch.cat([w0[:, perm], w0[:, 12:]], dim=1).contiguous()
    for i in range(config.num_proposal_iterations):
        a = config.proposal_net_args_list[min(i, len(config.proposal_net_args_list) - 1)]
        stack(f"proposal_networks.{i}.mlp_base", a.get("num_levels", 5), a.get("base_res", 16), a.get("max_res", 128),
              a.get("log2_hashmap_size", 17), a.get("features_per_level", 2), a.get("hidden_dim", 16), 1)
    if quantize_fp16:
        for k in list(out):
            if k.endswith(("hash_table", ".weight", ".bias")) and ("mlp" in k or "hash_table" in k):
                out[k] = out[k].to(torch.float16).to(torch.float32)
    return out
