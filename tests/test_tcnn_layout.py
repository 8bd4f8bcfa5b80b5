"""SURVEY §8(f) row 2, CPU side: known answers of the oracle's tiny-cuda-nn layout restatement (oracle/tcnn_layout.py, UNPINNED
against the real library) and agreement of the product importer (signerf_amd/tcnn_import.py) with it."""
import pytest
import torch

from helpers import oracle_params_from_tcnn, small_config, synthetic_tcnn_checkpoint
from oracle import tcnn_layout as tl
from signerf_amd.tcnn_import import convert_tcnn_state_dict, grid_level_table


def test_level_table_of_the_nerfacto_grids():
    m = tl.grid_meta(16, 16, 2048, 19)
    assert m.scales[0] == 15.0 and m.resolutions[:6] == [16, 23, 31, 43, 59, 81]
    assert m.dense == [True] * 5 + [False] * 11            # 81^3 > 2^19
    assert m.offsets[:3] == [0, 4096, 4096 + 12168]        # 23^3 = 12167 -> 12168 rows
    assert all(m.offsets[i + 1] - m.offsets[i] == 1 << 19 for i in range(5, 16))
    p0, p1 = tl.grid_meta(5, 16, 128, 17), tl.grid_meta(5, 16, 256, 17)
    assert p0.resolutions == [16, 27, 46, 77, 128] and p0.dense == [True, True, True, False, False]
    assert p1.resolutions == [16, 32, 64, 128, 256] and p1.dense == [True, True, False, False, False]
    res, offs = grid_level_table(16, 16, 2048, 19)         # the importer's own table agrees
    assert res == m.resolutions and offs == m.offsets


def test_grid_rows_dense_hash_and_wrap():
    m = tl.grid_meta(16, 16, 2048, 19)
    c = torch.tensor([[1, 2, 3], [15, 15, 15], [16, 15, 15], [16, 16, 16]])
    # dense level 0 (res 16, 4096 rows): x + 16 y + 256 z, wrapping modulo the level size at the far faces
    assert tl.grid_rows(m, 0, c).tolist() == [1 + 32 + 768, 4095, (16 + 240 + 3840) % 4096, (16 + 256 + 4096) % 4096]
    # hashed level 8: uint32 products, xor, modulo 2^19
    x, y, z = 100, 200, 300
    want = (x ^ ((y * 2654435761) & 0xFFFFFFFF) ^ ((z * 805459861) & 0xFFFFFFFF)) % (1 << 19)
    assert tl.grid_rows(m, 8, torch.tensor([[x, y, z]])).tolist() == [want]


def test_grid_encode_on_a_hand_built_dense_level():
    """One level, 2 features: the value at a grid vertex is returned exactly, and half-way along x is the mean of two vertices."""
    m = tl.grid_meta(2, 4, 8, 10)  # level 0: scale 3, res 4, 64 rows, dense
    assert m.dense[0] and m.resolutions[0] == 4 and m.scales[0] == 3.0
    g = torch.Generator().manual_seed(0)
    params = torch.rand(m.n_rows, 2, generator=g)
    # pos = 3 q + 0.5: q = 0.5 / 3 -> pos 1.0 exactly -> vertex (1,1,1); q_x = 1/3 -> pos_x 1.5
    q = torch.tensor([[0.5 / 3, 0.5 / 3, 0.5 / 3], [1.0 / 3, 0.5 / 3, 0.5 / 3]], dtype=torch.float64).float()
    out = tl.grid_encode(q, params, m)[:, :2]
    v = lambda x, y, z: params[x + 4 * y + 16 * z]  # noqa: E731
    assert torch.allclose(out[0], v(1, 1, 1), atol=1e-6)
    assert torch.allclose(out[1], 0.5 * (v(1, 1, 1) + v(2, 1, 1)), atol=1e-6)


def test_importer_agrees_with_the_oracle_unpacking():
    cfg = small_config(implementation="tcnn")
    sd = synthetic_tcnn_checkpoint(cfg, seed=4)
    conv = convert_tcnn_state_dict(sd, cfg)
    ref = oracle_params_from_tcnn(sd, cfg)
    for k, v in ref.items():
        if k.endswith("tcnn_grid") or k == "field.mlp_pred_normals.layers.0.weight":   # (columns permuted: checked in its own test below)
            continue
        assert torch.equal(conv[k], v), k
    assert float(conv["field.mlp_head.layers.0.bias"].abs().max()) > 0       # the ones-padded input column became a bias
    assert float(conv["field.mlp_base.mlp.layers.0.bias"].abs().max()) == 0  # zero padding after a grid: no bias
    # grid rows: level l of the flat layout sits at the start of slot l of the uniform table
    for prefix, levels, base, mx, log2_t in [("field.mlp_base", cfg.num_levels, cfg.base_res, cfg.max_res, cfg.log2_hashmap_size),
                                             ("proposal_networks.1.mlp_base", 5, 16, 256, 12)]:
        m = tl.grid_meta(levels, base, mx, log2_t)
        table, flat, T = conv[f"{prefix}.encoder.hash_table"], ref[f"{prefix}.encoder.tcnn_grid"], 1 << log2_t
        for level in range(levels):
            n = m.offsets[level + 1] - m.offsets[level]
            assert torch.equal(table[level * T : level * T + n], flat[m.offsets[level] : m.offsets[level + 1]])
            assert float(table[level * T + n : (level + 1) * T].abs().sum()) == 0
    q = convert_tcnn_state_dict(sd, cfg, quantize_fp16=True)["field.mlp_base.mlp.layers.0.weight"]
    assert torch.equal(q, q.half().float())
    with pytest.raises(ValueError):
        bad = dict(sd)
        bad["field.mlp_base.tcnn_encoding.params"] = bad["field.mlp_base.tcnn_encoding.params"][:-2]
        convert_tcnn_state_dict(bad, cfg)
    with pytest.raises(ValueError):
        convert_tcnn_state_dict(sd, small_config())


def test_importer_accepts_separate_encoding_and_network_vectors():
    """Some nerfstudio versions keep the grid and the MLP of a field as two tiny-cuda-nn modules instead of one fused one."""
    cfg = small_config(implementation="tcnn", num_proposal_iterations=0)
    sd = synthetic_tcnn_checkpoint(cfg, seed=9)
    fused = convert_tcnn_state_dict(sd, cfg)
    flat = sd.pop("field.mlp_base.tcnn_encoding.params")
    n_net = tl.mlp_n_params(2 * cfg.num_levels, cfg.hidden_dim, 2, 16)
    sd["field.mlp_base.mlp.tcnn_encoding.params"] = flat[:n_net]
    sd["field.mlp_base.encoder.tcnn_encoding.params"] = flat[n_net:]
    split = convert_tcnn_state_dict(sd, cfg)
    assert set(split) == set(fused)
    for k in fused:
        assert torch.equal(split[k], fused[k]), k  # 32 grid features need no input padding, so the two forms coincide


def test_pred_normal_mlp_is_imported_with_permuted_encoding_columns():
    """predict_normals=True checkpoints: the flat 27 -> 64 -> 64 -> 64 Network is unpacked, its first-layer columns re-ordered from the
    library's Frequency-encoding order (dimension-major: sin f0, cos f0, sin f1, cos f1) to this package's [sines | cosines]."""
    cfg = small_config(implementation="tcnn")
    assert cfg.predict_normals
    sd = synthetic_tcnn_checkpoint(cfg, seed=2)
    conv = convert_tcnn_state_dict(sd, cfg)
    ref = oracle_params_from_tcnn(sd, cfg)
    w_lib, w_pkg = ref["field.mlp_pred_normals.layers.0.weight"], conv["field.mlp_pred_normals.layers.0.weight"]
    assert w_pkg.shape == (64, 27) and torch.equal(w_pkg[:, 12:], w_lib[:, 12:])
    for a in range(3):
        for k in range(2):
            assert torch.equal(w_pkg[:, a * 2 + k], w_lib[:, a * 4 + k * 2])           # sin(axis a, frequency k)
            assert torch.equal(w_pkg[:, 6 + a * 2 + k], w_lib[:, a * 4 + k * 2 + 1])   # cos
    for k in ("layers.0.bias", "layers.1.weight", "layers.2.weight"):
        assert torch.equal(conv[f"field.mlp_pred_normals.{k}"], ref[f"field.mlp_pred_normals.{k}"])
    # the encoding itself: library order, frequencies pi 2^k
    x = torch.tensor([[0.25, 0.5, 0.125]])
    e = tl.frequency_encoding(x, 2)[0]
    assert e.shape == (12,) and torch.allclose(e[:4], torch.tensor([2**-0.5, 2**-0.5, 1.0, 0.0]), atol=1e-6)   # sin, cos of pi/4; of pi/2


def test_partial_checkpoint_without_proposal_vectors_loads():
    """the reference strips every `proposal*` key before load_state_dict(strict=False) when it retrains
    (signerf_pipeline.py:126-131,141-144) -- for a tiny-cuda-nn checkpoint that is the normal case and must not raise."""
    cfg = small_config(implementation="tcnn")
    sd = synthetic_tcnn_checkpoint(cfg, seed=1)
    stripped = {k: v for k, v in sd.items() if "proposal" not in k}
    assert len(stripped) < len(sd)
    model = cfg.setup()
    before = model.proposal_networks[0].mlp_base.encoder.hash_table.detach().clone()
    res = model.load_state_dict(stripped, strict=False)
    assert not res.unexpected_keys
    assert any(k.startswith("proposal_networks.0.") for k in res.missing_keys) and any(k.startswith("proposal_networks.1.") for k in res.missing_keys)
    assert not any(k.startswith("field.mlp_base") or k.startswith("field.mlp_head") for k in res.missing_keys)
    assert torch.equal(model.proposal_networks[0].mlp_base.encoder.hash_table, before)        # untouched
    full = convert_tcnn_state_dict(sd, cfg)
    assert torch.equal(model.field.mlp_base.encoder.hash_table, full["field.mlp_base.encoder.hash_table"])
    with pytest.raises(RuntimeError):                                                          # strict=True reports them as missing
        cfg.setup().load_state_dict(stripped, strict=True)
    # the same for a checkpoint that lacks only the colour head
    no_head = {k: v for k, v in sd.items() if "mlp_head" not in k}
    res = cfg.setup().load_state_dict(no_head, strict=False)
    assert any(k.startswith("field.mlp_head") for k in res.missing_keys)
