"""Register / scratch / occupancy table of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled for
gfx950; no GPU needed): `python tools/kernel_resources.py odise_amd/csrc/gemm.hip [more.hip ...]`.  Spilled registers in a hot kernel are
HBM traffic and issue slots the roofline does not account for."""
import re
import subprocess
import sys

import os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def table(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-x", "hip", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", *extra]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            m2 = re.search(r"Function Name: (\S+)", line)
            if m2:
                cur = {"name": m2.group(1)}
                rows.append(cur)
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().replace("odise::", "").replace("(odise::GemmArgs)", "")
    except OSError:
        return n


if __name__ == "__main__":
    for src in sys.argv[1:]:
        print(f"# {src}")
        print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'spillV':>6s} {'scratch':>7s} {'occ':>4s}")
        for r in table(src):
            print(f"{demangle(r['name'])[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} {r.get('ScratchSize [bytes/lane]', '?'):>7s} "
                  f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
