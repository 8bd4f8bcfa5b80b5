#!/usr/bin/env python3
"""Static instruction counts of the sample loop of a kernel in libsignerf_hip.so -- the constants `bench.py`'s issue roofs are
built from ("from constants the kernel fixes").

The gfx950 code object is extracted from the library that is actually loaded (llvm-objdump --offloading), the kernel is
disassembled, its sample loop is taken to be the backward branch with the largest span, and the instructions between the branch
target and the branch are counted by issue class.  Every instruction of the loop body is issued once per wave-step (the few
exec-masked regions inside it -- the contraction branch, NaN restore -- are issued whatever the mask), so these are the per-wave-step
issue counts; rocprofv3's SQ_INSTS_VALU / SQ_INSTS_MFMA / SQ_INSTS_VMEM_RD per wave-step (profiles/) agree with them.

    python tools/kernel_counts.py [mangled-kernel-name-substring]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "signerf_amd", "libsignerf_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _code_object(lib: str, workdir: str) -> str:
    dst = os.path.join(workdir, os.path.basename(lib))
    if os.path.lexists(dst):
        os.remove(dst)
    os.symlink(lib, dst)
    subprocess.run([OBJDUMP, "--offloading", os.path.basename(lib)], cwd=workdir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(workdir):
        if "amdgcn" in f and "gfx950" in f:
            return os.path.join(workdir, f)
    raise RuntimeError("no gfx950 code object in " + lib)


def classify(op: str) -> str:
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith(("buffer_load", "global_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "buffer_atomic", "global_atomic")):
        return "vmem_store"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def _disassemble(kernel_substr: str, lib: str):
    with tempfile.TemporaryDirectory() as wd:
        co = _code_object(lib, wd)
        syms = subprocess.run([OBJDUMP, "-t", co], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in syms.splitlines() if kernel_substr in ln and " F " in ln and ".text" in ln]
        names = [n for n in names if not n.endswith(".kd")]
        if len(names) > 1:
            # a prefix of the template argument list was given: take the instantiation whose REMAINING arguments are all 0 / false
            # (the production one; DUMP and experiment flags are trailing bools), so that adding a defaulted template parameter does
            # not silently break the callers
            def rest_is_default(n):
                rest = n.split(kernel_substr, 1)[1]
                args = rest.split("EvT", 1)[0] if "EvT" in rest else rest.split("Ev", 1)[0]
                return re.fullmatch(r"(L[bi]0E)*", args) is not None

            names = [n for n in names if rest_is_default(n)] or names
        if len(names) != 1:
            raise RuntimeError(f"{kernel_substr!r} matches {len(names)} kernels: {names[:4]}")
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f"--disassemble-symbols={names[0]}", co], capture_output=True, text=True,
                             check=True).stdout
    insts = []  # (address, opcode, text)
    for ln in dis.splitlines():
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), ln))
    return names[0], insts


def mfma_loops(kernel_substr: str, lib: str = LIB) -> list:
    """The innermost loops of a kernel that contain matrix-core instructions (the proposal kernel has two: the marching loops of its
    two nets), each as {"valu", "mfma", "gather", ...}.  Static counts: every instruction between the loop head and its backward branch,
    paths the workload does not take included."""
    name, insts = _disassemble(kernel_substr, lib)
    base = insts[0][0]
    spans = []
    for addr, op, ln in insts:
        if op.startswith(("s_branch", "s_cbranch")):
            m = re.search(r"\+0x([0-9a-fA-F]+)>", ln)
            if m:
                tgt = base + int(m.group(1), 16)
                if tgt < addr and any(o.startswith("v_mfma") for a, o, _ in insts if tgt <= a <= addr):
                    spans.append((tgt, addr))
    spans.sort(key=lambda s: s[1] - s[0])
    picked = []
    for s in spans:
        if not any(s[0] <= q[0] and q[1] <= s[1] for q in picked):
            picked.append(s)
    out = []
    for lo, hi in sorted(picked):
        c = {"kernel": name, "loop_bytes": hi - lo}
        opcodes = {}
        for addr, op, ln in insts:
            if lo <= addr <= hi:
                k = classify(op)
                c[k] = c.get(k, 0) + 1
                if k in ("valu", "mfma"):
                    base_op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                    opcodes[base_op] = opcodes.get(base_op, 0) + 1
                if op.startswith(("buffer_load_dwordx2", "buffer_load_dwordx4")):
                    c["gather"] = c.get("gather", 0) + 1
                if op.startswith("v_pk_") and op.endswith("_f32"):
                    c["packed_f32"] = c.get("packed_f32", 0) + 1
        c["opcodes"] = dict(sorted(opcodes.items(), key=lambda kv: -kv[1]))
        out.append(c)
    return out


def loop_counts(kernel_substr: str, lib: str = LIB) -> dict:
    """{"kernel": mangled name, "valu", "mfma", "vmem_load", "lds", "salu", ..., "packed_f32": n, "loop_bytes": n}"""
    with tempfile.TemporaryDirectory() as wd:
        co = _code_object(lib, wd)
        syms = subprocess.run([OBJDUMP, "-t", co], capture_output=True, text=True, check=True).stdout
        names = [ln.split()[-1] for ln in syms.splitlines() if kernel_substr in ln and " F " in ln and ".text" in ln]
        names = [n for n in names if not n.endswith(".kd")]
        if len(names) > 1:
            # a prefix of the template argument list was given: take the instantiation whose REMAINING arguments are all 0 / false
            # (the production one; DUMP and experiment flags are trailing bools), so that adding a defaulted template parameter does
            # not silently break the callers
            def rest_is_default(n):
                rest = n.split(kernel_substr, 1)[1]
                args = rest.split("EvT", 1)[0] if "EvT" in rest else rest.split("Ev", 1)[0]
                return re.fullmatch(r"(L[bi]0E)*", args) is not None

            names = [n for n in names if rest_is_default(n)] or names
        if len(names) != 1:
            raise RuntimeError(f"{kernel_substr!r} matches {len(names)} kernels: {names[:4]}")
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f"--disassemble-symbols={names[0]}", co], capture_output=True, text=True,
                             check=True).stdout
    insts = []  # (address, opcode, text)
    for ln in dis.splitlines():
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), ln))
    base = insts[0][0]
    spans = []
    for addr, op, ln in insts:
        if op.startswith(("s_branch", "s_cbranch")):
            m = re.search(r"\+0x([0-9a-fA-F]+)>", ln)
            if m:
                tgt = base + int(m.group(1), 16)
                if tgt < addr:
                    spans.append((tgt, addr))
    if not spans:
        raise RuntimeError("no backward branch in " + names[0])
    # The sample loop: the back-edges of the loop that holds the matrix-core instructions (a loop may close through several branches
    # a few instructions apart); NOT simply the largest backward branch -- an enclosing branch of the epilogue can span the whole kernel.
    with_mfma = [sp for sp in spans if any(o.startswith("v_mfma") for a, o, _ in insts if sp[0] <= a <= sp[1])]
    if with_mfma:
        inner = min(with_mfma, key=lambda sp: sp[1] - sp[0])
        same = [sp for sp in with_mfma if abs(sp[1] - inner[1]) <= 0x40 and inner[0] - sp[0] <= 0x200]
        best = max(same, key=lambda sp: sp[1] - sp[0])
    else:
        best = max(spans, key=lambda sp: sp[1] - sp[0])
    out = {"kernel": names[0], "loop_bytes": best[1] - best[0]}
    vmcnt = []
    opcodes = {}
    for addr, op, ln in insts:
        if best[0] <= addr <= best[1]:
            c = classify(op)
            out[c] = out.get(c, 0) + 1
            if c in ("valu", "mfma"):   # per-opcode histogram of what goes through the vector issue port (encoding suffixes dropped)
                base_op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
                opcodes[base_op] = opcodes.get(base_op, 0) + 1
            m = re.search(r"vmcnt\((\d+)\)", ln) if op == "s_waitcnt" else None
            if m:
                vmcnt.append(int(m.group(1)))
            if op.startswith("v_pk_") and op.endswith("_f32"):
                out["packed_f32"] = out.get("packed_f32", 0) + 1
            if op.startswith(("buffer_load_dwordx2", "buffer_load_dwordx4")):  # the hash-grid gathers (spill reloads are scratch_load)
                out["gather"] = out.get("gather", 0) + 1
    for k in ("valu", "mfma", "vmem_load", "lds", "packed_f32", "gather"):
        out.setdefault(k, 0)
    # how deep the loop lets its vector loads run ahead: the largest N of an `s_waitcnt vmcnt(N)` = loads still in flight when the first
    # result is consumed, and the mean over all such waits.  r05: one global atomic in a rarely taken branch of K1's loop made hipcc
    # schedule the hash phase load -> wait -> blend level by level (max 4 instead of 20 in flight): identical instruction counts, +10 % time
    out["opcodes"] = dict(sorted(opcodes.items(), key=lambda kv: -kv[1]))
    out["vmcnt_max"] = max(vmcnt) if vmcnt else 0
    out["vmcnt_mean"] = sum(vmcnt) / len(vmcnt) if vmcnt else 0.0
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = args[0] if args else "sn_render_main_kernelILi0ELi1ELi0ELi11ELb0E"
    res = loop_counts(pat)
    ops = res.pop("opcodes")
    print(res)
    if "--opcodes" in sys.argv:
        for k, v in ops.items():
            print(f"{v:5d} {k}")
