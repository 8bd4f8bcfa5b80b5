"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol that
include/signerf_hip.h declares, and the ctypes mirrors of its structs have the C layout.  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from helpers import ROOT
from signerf_amd import _lib

HEADER = os.path.join(ROOT, "include", "signerf_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol(built_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sn_[a-z_0-9]+)", out))
    assert set(_declared_functions()) <= exported
    lib = _lib.load()
    for name in _declared_functions():
        assert getattr(lib, name) is not None


def test_library_contains_gfx950_code_object(built_lib):
    blob = open(built_lib, "rb").read()
    assert b"gfx950" in blob and b"sn_render_main_kernel" in blob


def test_ctypes_struct_layout_matches_c(tmp_path):
    prog = r"""
#include <stdio.h>
#include <stddef.h>
#include "signerf_hip.h"
int main(void) {
  printf("%zu %zu %zu\n", sizeof(SnHashMlpDesc), sizeof(SnFieldDesc), sizeof(SnRenderOpts));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(SnHashMlpDesc, scalings), offsetof(SnFieldDesc, proposals),
         offsetof(SnFieldDesc, average_init_density), offsetof(SnFieldDesc, num_proposals), offsetof(SnFieldDesc, disable_scene_contraction),
         offsetof(SnFieldDesc, aabb));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", offsetof(SnRenderOpts, num_nerf_samples), offsetof(SnRenderOpts, chunk_rays),
         offsetof(SnRenderOpts, workspace), offsetof(SnRenderOpts, initial_spacing_bins), offsetof(SnRenderOpts, pdf_u),
         offsetof(SnRenderOpts, background_mode), offsetof(SnRenderOpts, background_rgb), offsetof(SnRenderOpts, spacing_mode));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(SnDebugDump), offsetof(SnDebugDump, prop_q), offsetof(SnDebugDump, pdf_index),
         sizeof(SnDebugLayout), offsetof(SnDebugLayout, dense_bytes), offsetof(SnDebugLayout, pair_bytes));
  printf("%zu %zu %zu %zu\n", offsetof(SnFieldDesc, dense_levels), offsetof(SnFieldDesc, dense_copy_cap_mb), offsetof(SnDebugLayout, table_bytes),
         offsetof(SnDebugLayout, handle_bytes));
  printf("%zu %zu %zu %zu\n", sizeof(SnCameraDesc), offsetof(SnCameraDesc, height), offsetof(SnCameraDesc, camera_type),
         offsetof(SnCameraDesc, distortion));
  /* r06: the versioned structs begin with struct_size; their last fields */
  printf("%zu %zu %zu %zu\n", offsetof(SnFieldDesc, struct_size), offsetof(SnRenderOpts, struct_size), offsetof(SnMaskOpts, struct_size),
         offsetof(SnDebugLayout, struct_size));
  printf("%zu %zu %zu %zu %zu %zu %zu\n", offsetof(SnFieldDesc, main_field), offsetof(SnFieldDesc, half_grid), offsetof(SnRenderOpts, march_stats),
         offsetof(SnRenderOpts, reuse_final_bins), sizeof(SnMaskOpts), offsetof(SnMaskOpts, manual_min), offsetof(SnMaskOpts, additional_depth_radius));
  printf("%d\n", SN_ABI_VERSION);
  return 0;
}
"""
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_lib.SnHashMlpDesc), C.sizeof(_lib.SnFieldDesc), C.sizeof(_lib.SnRenderOpts),
            _lib.SnHashMlpDesc.scalings.offset, _lib.SnFieldDesc.proposals.offset, _lib.SnFieldDesc.average_init_density.offset,
            _lib.SnFieldDesc.num_proposals.offset, _lib.SnFieldDesc.disable_scene_contraction.offset, _lib.SnFieldDesc.aabb.offset,
            _lib.SnRenderOpts.num_nerf_samples.offset, _lib.SnRenderOpts.chunk_rays.offset, _lib.SnRenderOpts.workspace.offset,
            _lib.SnRenderOpts.initial_spacing_bins.offset, _lib.SnRenderOpts.pdf_u.offset,
            _lib.SnRenderOpts.background_mode.offset, _lib.SnRenderOpts.background_rgb.offset, _lib.SnRenderOpts.spacing_mode.offset,
            C.sizeof(_lib.SnDebugDump), _lib.SnDebugDump.prop_q.offset, _lib.SnDebugDump.pdf_index.offset,
            C.sizeof(_lib.SnDebugLayout), _lib.SnDebugLayout.dense_bytes.offset, _lib.SnDebugLayout.pair_bytes.offset,
            _lib.SnFieldDesc.dense_levels.offset, _lib.SnFieldDesc.dense_copy_cap_mb.offset, _lib.SnDebugLayout.table_bytes.offset,
            _lib.SnDebugLayout.handle_bytes.offset,
            C.sizeof(_lib.SnCameraDesc), _lib.SnCameraDesc.height.offset, _lib.SnCameraDesc.camera_type.offset,
            _lib.SnCameraDesc.distortion.offset,
            0, 0, 0, 0,
            _lib.SnFieldDesc.main_field.offset, _lib.SnFieldDesc.half_grid.offset, _lib.SnRenderOpts.march_stats.offset,
            _lib.SnRenderOpts.reuse_final_bins.offset, C.sizeof(_lib.SnMaskOpts), _lib.SnMaskOpts.manual_min.offset,
            _lib.SnMaskOpts.additional_depth_radius.offset,
            _lib.SN_ABI_VERSION]
    assert got == want
    for cls in (_lib.SnFieldDesc, _lib.SnRenderOpts, _lib.SnMaskOpts, _lib.SnDebugLayout):
        assert cls.struct_size.offset == 0 and cls().struct_size == C.sizeof(cls)   # set at construction


def test_abi_version_and_struct_size_handshake(built_lib):
    """include/signerf_hip.h "ABI evolution": the library states its version; a versioned struct is read up to the size its CALLER declares
    -- never behind it -- and a size the library does not know (0 = unset, larger = a newer caller) is refused with both sizes in the text.
    sn_create checks the descriptor before it touches a device, so all of this runs without a GPU."""
    lib = _lib.load()
    assert lib.sn_abi_version() == _lib.SN_ABI_VERSION == 6
    h = C.c_void_p(None)

    def create(d):
        h.value = None
        st = lib.sn_create(C.byref(d), C.byref(h))
        msg = (lib.sn_last_error(None) or b"").decode()
        if st == 0:
            lib.sn_destroy(h)
        return st, msg

    def good():
        from helpers import small_config
        m = small_config().setup()
        return m._field_desc()

    d = good()
    full = C.sizeof(_lib.SnFieldDesc)
    assert d.struct_size == full
    ok_status = create(d)[0]
    assert ok_status in (0, 2)                 # created (GPU box) or "no HIP device" (here): the descriptor itself passed
    for bad, word in ((0, "was not set"), (8, "knows sizes"), (full + 4, "newer header")):
        d = good()
        d.struct_size = bad
        st, msg = create(d)
        assert st == 1 and "struct_size" in msg and word in msg and str(full) in msg, (bad, msg)
    # A caller compiled against a SHORTER layout (the struct without the r04 / r05 appendices: dense_levels, dense_copy_cap_mb, half_grid):
    # what sits in the memory behind the size it declares is never looked at.  Shown with a field the library validates: inside the
    # declared size a bad value is refused, behind it the same bytes are ignored (and take their zero default).
    short = _lib.SnFieldDesc.dense_levels.offset
    d = good()
    d.disable_scene_contraction = 7             # inside every accepted size: refused
    assert create(d)[0] == 1 and "disable_scene_contraction" in create(d)[1]
    d = good()
    d.struct_size = short
    d.dense_levels, d.dense_copy_cap_mb, d.half_grid = -7, -7, 1234567     # behind the declared size: garbage, never read
    assert create(d)[0] == ok_status
    d.struct_size = short - 4                   # below the smallest accepted layout
    st, msg = create(d)
    assert st == 1 and "knows sizes" in msg and str(short) in msg


def test_plain_c_client_of_the_abi(built_lib, tmp_path):
    """tests/c/cabi_client.c: a C99 program that includes the header, dlopens the library and walks the ABI handshake (version, unset /
    too-new struct_size, unsupported architecture, a valid descriptor) -- the boundary holds no C++ or torch type, and is usable from
    the language a foreign binding would be written in."""
    import torch

    exe = tmp_path / "cabi_client"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "cabi_client.c"), "-o", str(exe), "-ldl"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(exe), built_lib], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict(ln.split(" ", 1) for ln in r.stdout.splitlines() if " " in ln)
    assert out["result"] == "ok" and out["abi_version"] == "%d header %d" % (_lib.SN_ABI_VERSION, _lib.SN_ABI_VERSION)
    sizes = out["sizeof"].split()
    assert int(sizes[1]) == C.sizeof(_lib.SnFieldDesc) and int(sizes[3]) == C.sizeof(_lib.SnRenderOpts)


def test_every_entry_point_survives_null_arguments(built_lib):
    """No entry point dereferences a NULL handle or pointer: SN_ERR_INVALID (size queries: 0, sn_effective_precision: -1, sn_destroy(NULL): ok),
    the text in sn_last_error.  Run in a child process -- the failure mode would be a segfault."""
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scripts", "null_arguments.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    got = dict(ln.split() for ln in r.stdout.splitlines() if len(ln.split()) == 2)
    declared = set(_declared_functions()) - {"sn_create", "sn_abi_version"}
    assert declared <= set(got) | {"sn_create"} and got["sn_create_null"] == "1", sorted(declared - set(got))
    special = {"sn_destroy": "0", "sn_last_error": "True", "sn_workspace_bytes": "0", "sn_mask_workspace_bytes": "0", "sn_effective_precision": "-1"}
    for name, st in got.items():
        assert st == special.get(name, "1"), (name, st)


def test_error_path_without_gpu(built_lib):
    """sn_create validates the architecture before touching the device, and reports through sn_last_error."""
    lib = _lib.load()
    d = _lib.SnFieldDesc()
    d.main_field.num_levels = 7  # unsupported
    h = C.c_void_p(None)
    st = lib.sn_create(C.byref(d), C.byref(h))
    assert st == 1 and not h
    assert b"num_levels" in lib.sn_last_error(None)
    with pytest.raises(_lib.SignerfHipError):
        _lib.check(st, None, "sn_create")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under signerf_amd/, bench.py's GPU leg or the C sources may use it."""
    pkg = os.path.join(ROOT, "signerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                # the reference is cited in comments / docstrings only -- never opened, imported or put on sys.path
                assert not re.search(r"(open|import_module|sys\.path\S*)\s*\(?[^\n]*/root/reference", text), f


def test_model_refuses_cpu_render():
    import torch
    from helpers import small_config
    from signerf_amd.cameras import RayBundle

    m = small_config(num_proposal_iterations=0).setup()
    b = RayBundle(torch.zeros(2, 2, 3), torch.ones(2, 2, 3), torch.ones(2, 2, 1))
    with pytest.raises(_lib.SignerfHipError):
        m.get_outputs_for_camera_ray_bundle(b)


def test_isa_has_no_packed_result_into_swizzle_pairs():
    """gfx950 hazard found on hardware (DESIGN.md "Hazards"): a half of a packed-fp32 VALU result read 1-2 instructions later by a
    swizzling consumer (permlane32_swap, v_fma_mix, v_pk_* with op_sel) can see the stale value when another kernel's MFMA waves
    share the SIMD.  The sources keep such pairs apart by hand; this scan of the compiled ISA keeps it that way."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_hazard_scan.py")
    r = subprocess.run([sys.executable, tool, "--window", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_static_instruction_counts_of_the_fused_kernels(built_lib):
    """bench.py's issue roofs are built from static counts of the loaded library (tools/kernel_counts.py): the sample loop of K1 holds the
    120 MFMAs of the split-precision MLPs, its gathers and no packed-fp32 VALU; the proposal kernel has one marching loop per net with
    the 6 MFMAs of the one-tile layer."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_counts

    k1 = kernel_counts.loop_counts("sn_render_main_kernelILi0ELi1ELi0ELi11E")       # (a prefix: trailing defaulted arguments are matched)
    assert k1["mfma"] == 120 and k1["gather"] == 84 and k1["packed_f32"] == 0 and 1300 < k1["valu"] < 1500
    # the hash phase keeps its gathers in flight in groups of 4 levels (SN_HASH_GROUP).  Same instruction COUNTS do not guarantee it: r05's
    # first march_stats (one atomicAdd in the exit branch of this loop) left every count above unchanged and cost 10 % per launch, because
    # hipcc then waited for each level's 4 loads before issuing the next (vmcnt never above 4).  Static guard for that class of regression:
    assert k1["vmcnt_max"] >= 16 and k1["vmcnt_mean"] >= 6.0, (k1["vmcnt_max"], k1["vmcnt_mean"])
    k1b = kernel_counts.loop_counts("sn_render_main_kernelILi1ELi1ELi0ELi11E")      # behind the proposal sampler (bins mode, x-paired hashed levels)
    assert k1b["mfma"] == 120 and k1b["gather"] == 64 and k1b["vmcnt_max"] >= 12, k1b
    k2 = kernel_counts.mfma_loops("sn_proposal_kernelILi0ELi5ELi4E")
    assert [c["mfma"] for c in k2] == [6, 6] and all(c["gather"] == 20 and 220 < c["valu"] < 340 for c in k2)
    assert all(c.get("packed_f32", 0) == 0 for c in k2)   # r03 re-measured a packed blend in K2: slower (profiles/r03_pk_blend_ab.txt)


def test_lazy_outputs_dict_semantics():
    """signerf_amd.nerfacto.LazyOutputs: the normals entries are produced once, on first use, and only then."""
    from signerf_amd.nerfacto import LazyOutputs

    calls = []

    def producer():
        calls.append(1)
        return {"normals": "n", "pred_normals": "p"}

    d = LazyOutputs({"rgb": 1, "depth": 2}, producer)
    assert d["rgb"] == 1 and d.get("depth") == 2 and d.get("missing", 7) == 7 and calls == []
    assert "normals" in d and "pred_normals" in d and "other" not in d and calls == []
    with pytest.raises(KeyError):
        d["other"]
    assert d["pred_normals"] == "p" and calls == [1]
    assert d["normals"] == "n" and calls == [1]
    assert set(d) == {"rgb", "depth", "normals", "pred_normals"} and len(d) == 4 and calls == [1]
    e = LazyOutputs({"rgb": 1}, producer)
    assert dict(e.items()) == {"rgb": 1, "normals": "n", "pred_normals": "p"} and calls == [1, 1]
    assert LazyOutputs({"rgb": 1}, producer).copy() == {"rgb": 1, "normals": "n", "pred_normals": "p"}


def test_unbuilt_config_values_are_rejected_at_model_set_up():
    """Values of nerfstudio's config that the kernels do not implement raise instead of rendering something else."""
    from signerf_amd import SIGNeRFModelConfig

    for kw in ({"background_color": "#ff0000"}, {"proposal_initial_sampler": "lindisp"}):
        with pytest.raises(NotImplementedError):
            SIGNeRFModelConfig(**kw).setup()
    for kw in ({"proposal_initial_sampler": "uniform"}, {"disable_scene_contraction": True}):   # built in r03 (tests/test_gpu_options.py)
        SIGNeRFModelConfig(log2_hashmap_size=12, **kw).setup()
    SIGNeRFModelConfig(log2_hashmap_size=12).setup()   # the defaults build (CPU-side module set-up only)
    for bg in ("last_sample", "black", "white", "random"):   # RGBRenderer's named backgrounds (r03)
        SIGNeRFModelConfig(log2_hashmap_size=12, background_color=bg).setup()


def test_method_registration_entry_point():
    """The plugin registration of /root/reference/pyproject.toml:44-46 + signerf_config.py:16-62: the entry-point group exists, resolves
    to a MethodSpecification-shaped object, its model config carries the reference's overrides and its optimizer groups are exactly
    the groups the model hands out."""
    import importlib

    import tomli

    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as f:
        eps = tomli.load(f)["project"]["entry-points"]["nerfstudio.method_configs"]
    assert eps, "no entry point in group nerfstudio.method_configs"
    mod, attr = next(iter(eps.values())).split(":")
    spec = getattr(importlib.import_module(mod), attr)
    assert spec.config.method_name == "signerf" and "SIGNeRF" in spec.description
    m = spec.config.pipeline.model
    assert (m.eval_num_rays_per_chunk, m.predict_normals, m.use_lpips, m.average_init_density) == (1 << 15, True, True, 0.01)
    model = type(m)(log2_hashmap_size=12).setup()          # CPU-side module set-up only
    assert set(spec.config.optimizers) == set(model.get_param_groups()) == {"proposal_networks", "fields", "camera_opt"}
    assert spec.config.optimizers["camera_opt"]["optimizer"].lr == 1e-15
