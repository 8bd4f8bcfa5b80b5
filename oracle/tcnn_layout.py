"""CPU restatement of tiny-cuda-nn's multiresolution hash grid and FullyFusedMLP PARAMETER LAYOUT.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  **PARITY UNPINNED**: tinycudann (the CUDA-only third-party dependency
nerfstudio 1.0.2 delegates to when ``implementation="tcnn"``) is absent from /root/reference and from this
image, there is no checkpoint fixture, and nothing here can be checked against it.  The functions restate its published
algorithm (Mueller et al., "Instant Neural Graphics Primitives", 2022, and the library's encodings/grid.h, networks/
fully_fused_mlp, network_with_input_encoding.h as of v1.6) from the description in SURVEY.md Appendix A7/A8 and from memory of
that source; every assumption that a real checkpoint could falsify is listed in ASSUMPTIONS below and in DESIGN.md.

What differs from the torch fallback restated in oracle/nerfacto.py (which is the parity target of the render path):
  * per-level scale  s_l = exp2f(l * log2f(g)) * base - 1  (fp32, not floored),  position  x = fmaf(s_l, q, 0.5),
    corners floor(x) and floor(x) + 1, weight of the "+1" corner = x - floor(x);
  * level l has  n_l = min(next_multiple((ceil(s_l) + 1)^3, 8), 2^log2_T)  rows; levels are stored back to back;
  * a level whose full grid fits (res^3 <= n_l) is indexed densely, x + y*res + z*res^2, else with the same xor hash as the
    torch path (uint32 wrap); both followed by  % n_l  (dense grids wrap at the far faces -- the library documents this);
  * MLPs have no bias; inputs are padded to a multiple of 16 (grid encodings pad with 0, plain networks pad with 1 -- the first
    padded column of a plain network therefore acts as a bias); outputs are padded to a multiple of 16 rows;
  * one flat fp32 vector per module: [network matrices, row-major (out, in), first to last | grid rows, feature-minor].
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import Tensor

ASSUMPTIONS = (
    "flat vector order = network parameters first, then encoding parameters (NetworkWithInputEncoding::set_params_impl)",
    "weight matrices are row-major (rows = outputs, columns = padded inputs), first layer .. output layer",
    "grid encodings pad their output with zeros, plain Network inputs are padded with ones",
    "n_hidden_layers = num_layers - 1 (nerfstudio's tcnn network config), so a 2-layer MLP is [W x in], [out_pad x W]",
    "hash primes (1, 2654435761, 805459861), uint32 arithmetic, index % level size",
    "state-dict key of every tcnn module ends in '.tcnn_encoding.params'",
    "Frequency encoding: output j of n_frequencies F is sin(pi 2^((j / 2) % F) x_(j / 2F) + (j % 2) pi / 2) -- dimension-major, then "
    "frequency, then (sin, cos); nerfstudio's NeRFEncoding(implementation='tcnn') returns it as is",
    "field.mlp_pred_normals of a tcnn nerfacto = plain Network 27 -> 64 -> 64 -> 64 (inputs padded to 32 with ones), followed by the "
    "torch PredNormalsFieldHead Linear(64, 3)",
)

HASH_PRIMES = (1, 2654435761, 805459861)


def next_multiple(v: int, m: int) -> int:
    return ((v + m - 1) // m) * m


@dataclass
class GridMeta:
    num_levels: int
    features_per_level: int
    log2_hashmap_size: int
    scales: List[float]       # fp32 values
    resolutions: List[int]
    offsets: List[int]        # rows; offsets[num_levels] = total rows
    dense: List[bool]

    @property
    def n_rows(self) -> int:
        return self.offsets[self.num_levels]

    @property
    def n_params(self) -> int:
        return self.n_rows * self.features_per_level


def grid_meta(num_levels: int, base_res: int, max_res: int, log2_hashmap_size: int, features_per_level: int = 2) -> GridMeta:
    """Level table of a HashGrid encoding configured the way nerfstudio configures it: per_level_scale =
    exp((ln max_res - ln base_res) / (L - 1)) computed in float64 numpy, handed to the library as a float."""
    growth = np.exp((np.log(max_res) - np.log(base_res)) / (num_levels - 1)) if num_levels > 1 else 1.0
    log2_g = np.log2(np.float32(growth), dtype=np.float32)  # std::log2(float)
    scales, res, offs, dense = [], [], [0], []
    T = 1 << log2_hashmap_size
    for level in range(num_levels):
        s = np.float32(np.exp2(np.float32(level) * log2_g, dtype=np.float32)) * np.float32(base_res) - np.float32(1.0)
        r = int(np.ceil(s)) + 1
        n = min(next_multiple(min(r**3, (2**32 - 1) // 2), 8), T)
        scales.append(float(s))
        res.append(r)
        dense.append(r**3 <= n)
        offs.append(offs[-1] + n)
    return GridMeta(num_levels, features_per_level, log2_hashmap_size, scales, res, offs, dense)


def grid_rows(meta: GridMeta, level: int, coords: Tensor) -> Tensor:
    """coords [...,3] int64 (non-negative) -> row inside the level (grid_index of the library)."""
    n = meta.offsets[level + 1] - meta.offsets[level]
    r = meta.resolutions[level]
    c = coords.to(torch.int64)
    if meta.dense[level]:
        idx = (c[..., 0] + c[..., 1] * r + c[..., 2] * r * r) & 0xFFFFFFFF
    else:
        p = [(c[..., d] * HASH_PRIMES[d]) & 0xFFFFFFFF for d in range(3)]  # uint32 wrap of each product
        idx = p[0] ^ p[1] ^ p[2]
    return idx % n


def grid_encode(q: Tensor, grid_params: Tensor, meta: GridMeta) -> Tensor:
    """q [P,3] in [0,1) fp32, grid_params [n_rows, F] fp32 -> [P, L*F] level-major, evaluated in fp32 (the library blends in
    fp16; see the module docstring)."""
    outs = []
    for level in range(meta.num_levels):
        s = torch.tensor(meta.scales[level], dtype=torch.float32)
        pos = (q.double() * s.double() + 0.5).float()  # fmaf(scale, q, 0.5): one rounding (the fp64 product of two fp32 is exact)
        fl = torch.floor(pos)
        w = pos - fl
        base = fl.to(torch.int64)
        acc = torch.zeros(q.shape[0], meta.features_per_level, dtype=torch.float32)
        table = grid_params[meta.offsets[level] : meta.offsets[level + 1]]
        for corner in range(8):
            weight = torch.ones(q.shape[0], dtype=torch.float32)
            c = base.clone()
            for d in range(3):
                if corner & (1 << d):
                    weight = weight * w[:, d]
                    c[:, d] += 1
                else:
                    weight = weight * (1 - w[:, d])
            acc = acc + weight[:, None] * table[grid_rows(meta, level, c)]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


# ----------------------------------------------------------------------------------------------------------------------
# FullyFusedMLP parameter vector
# ----------------------------------------------------------------------------------------------------------------------
def mlp_shapes(in_dim: int, width: int, num_layers: int, out_dim: int) -> List[Tuple[int, int]]:
    """(rows, cols) of the matrices of a network with `num_layers` linear layers (nerfstudio's counting)."""
    in_pad, out_pad = next_multiple(in_dim, 16), next_multiple(out_dim, 16)
    shapes = [(width, in_pad)]
    for _ in range(num_layers - 2):
        shapes.append((width, width))
    shapes.append((out_pad, width))
    return shapes


def mlp_n_params(in_dim: int, width: int, num_layers: int, out_dim: int) -> int:
    return sum(r * c for r, c in mlp_shapes(in_dim, width, num_layers, out_dim))


def mlp_unpack(flat: Tensor, in_dim: int, width: int, num_layers: int, out_dim: int, pad_value: float) -> Dict[str, Tensor]:
    """Flat network vector -> torch-style {layers.i.weight [out,in], layers.i.bias [out]}.  pad_value is what the padded input
    columns see (0 after a grid encoding, 1 for a plain network): their weights fold into the first layer's bias."""
    out: Dict[str, Tensor] = {}
    o = 0
    shapes = mlp_shapes(in_dim, width, num_layers, out_dim)
    for i, (r, c) in enumerate(shapes):
        w = flat[o : o + r * c].reshape(r, c)
        o += r * c
        rows = out_dim if i == len(shapes) - 1 else r
        if i == 0:
            out[f"layers.{i}.weight"] = w[:rows, :in_dim].clone()
            out[f"layers.{i}.bias"] = (w[:rows, in_dim:] * pad_value).sum(dim=1)
        else:
            out[f"layers.{i}.weight"] = w[:rows].clone()
            out[f"layers.{i}.bias"] = torch.zeros(rows, dtype=flat.dtype)
    assert o == mlp_n_params(in_dim, width, num_layers, out_dim)
    return out


def frequency_encoding(x: Tensor, n_frequencies: int) -> Tensor:
    """tiny-cuda-nn's Frequency encoding (encodings/frequency.h): x [P, D] -> [P, D * 2F]; output j reads input dimension j // 2F at
    frequency pi * 2^((j // 2) % F), even j = sin, odd j = the same shifted by pi / 2."""
    P, D = x.shape
    outs = []
    for i in range(D):
        for f in range(n_frequencies):
            arg = x[:, i].double() * (2.0**f) * np.pi
            outs += [torch.sin(arg), torch.sin(arg + np.pi / 2.0)]
    return torch.stack(outs, dim=-1).to(torch.float32)
