"""Build libodise_hip.so (hipcc, gfx950) in-tree.

`python -m odise_amd.build` compiles every source under odise_amd/csrc into
odise_amd/lib/libodise_hip.so.  Objects are cached under odise_amd/lib/obj and rebuilt only when the
source (or a header) is newer.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only
build container; the resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libodise_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _split_usage(stderr: str):
    """-> ({kernel: {"vgprs", "agprs", "scratch", "vgpr_spill", "sgpr_spill", "occupancy"}}, the rest of stderr)."""
    usage, rest, cur = {}, [], None
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill",
            "Occupancy [waves/SIMD]": "occupancy"}
    lines = stderr.splitlines()
    i = 0
    while i < len(lines):
        ln = lines[i]
        if "[-Rpass-analysis=kernel-resource-usage]" in ln:
            m = re.search(r"remark:\s+(.*?):\s*(\S+)\s+\[-Rpass", ln)
            if m:
                k, v = m.group(1).strip(), m.group(2)
                if k == "Function Name":
                    cur = usage.setdefault(v, {})
                elif cur is not None and k in keys and v.lstrip("-").isdigit():
                    cur[keys[k]] = int(v)
            # the remark is followed by a source excerpt and a caret line
            while i + 1 < len(lines) and (re.match(r"\s+\d+ \|", lines[i + 1]) or re.match(r"\s+\|", lines[i + 1])):
                i += 1
        else:
            rest.append(ln)
        i += 1
    return usage, "\n".join(rest)


def resource_usage(objdir: str = OBJDIR) -> dict:
    """{kernel (mangled): resource dict} of every kernel of the last build (the <obj>.usage files)."""
    out = {}
    if os.path.isdir(objdir):
        for f in sorted(os.listdir(objdir)):
            if f.endswith(".usage"):
                with open(os.path.join(objdir, f)) as fh:
                    out.update(json.load(fh))
    return out


def build(verbose: bool = True, force: bool = False, tools: bool = False, m32: bool = False) -> str:
    """tools=True builds the measurement variant (libodise_hip_tools.so, -DODISE_TOOLS: timing ablations and the ODISE_GEMM_FLAGS /
    ODISE_NO_GN_FUSION environment switches compiled in); tools/ scripts select it with ODISE_HIP_LIB.  The product library has none.
    m32=True builds the A/B variant libodise_hip_m32.so (-DODISE_MFMA32: every main loop of csrc/gemm.hip on v_mfma_f32_32x32x16_f16, the
    MFMA shape of rounds 1-4, instead of 16x16x32) for same-box comparisons of whole steps (tools/final_evidence.sh); never loaded by default."""
    objdir = OBJDIR + ("_tools" if tools else "_m32" if m32 else "")
    lib_path = os.path.join(LIBDIR, "libodise_hip_tools.so") if tools else os.path.join(LIBDIR, "libodise_hip_m32.so") if m32 else LIB
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hm = _headers_mtime()
    # -Rpass-analysis=kernel-resource-usage: the compiler reports every kernel's VGPRs / scratch / spills; kept next to the object (<obj>.usage,
    # read by resource_usage() and tests/test_build_resources.py: a main-loop kernel that starts spilling is a silent 10 % regression of the step)
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Rpass-analysis=kernel-resource-usage",
             "-Wno-unused-result", "-x", "hip"] + (["-DODISE_TOOLS=1"] if tools else []) + (["-DODISE_MFMA32=1"] if m32 else [])
    # per-source extras.  attn.hip: MFMA results are consumed by VALU code every tile (softmax, rescale), so keep them in VGPRs -
    # the default AGPR form costs a v_accvgpr_read/write per element (128 VALU slots per tile) and a wave of occupancy.
    extra = {"attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(op)
        if force or not os.path.exists(op) or not os.path.exists(op + ".usage") or os.path.getmtime(op) < max(os.path.getmtime(sp), hm):
            jobs.append((sp, op))

    def _compile(job):
        sp, op = job
        cmd = [hipcc, *flags, *extra.get(os.path.basename(sp), []), "-c", sp, "-o", op]
        if verbose:
            print("[odise_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        usage, rest = _split_usage(r.stderr)
        with open(op + ".usage", "w") as f:
            json.dump(usage, f, indent=0, sort_keys=True)
        if verbose and rest.strip():
            print(rest, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_compile, jobs))
    if jobs or not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path, *objs]
        if verbose:
            print("[odise_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, tools="--tools" in sys.argv, m32="--m32" in sys.argv))
