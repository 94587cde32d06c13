"""Mutation fuzzing of the HOST half of the JPEG decoder (marker parsing, Huffman / progressive entropy decoding) under
AddressSanitizer + UndefinedBehaviorSanitizer.  The device kernels are cut out of odise_amd/csrc/jpeg.hip (a host-only object with
kernels would need their fat binary), the rest is compiled as it is; seed files come from Pillow's encoder (baseline, optimised tables,
restart intervals, progressive, grey, EXIF).  Every mutated stream is handed over in an exact-size heap block, so one byte read past the
input is a report.

    python tools/jpeg_fuzz/run.py [iterations per seed file, default 4000]      # round 1: 480 000 streams, no report
"""
import io
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402
from tests.test_oracle_jpeg import _jpeg, _picture  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HIPCC = "/opt/rocm/bin/hipcc"
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]


def host_half(src: str) -> str:
    a, b = src.index("// ---- device side"), src.index("static void fill_info")
    c = src.index("void jpeg_release(odise_hip_ctx* ctx)")
    d = src.index("}  // namespace odise", c)
    e, f = src.index('extern "C" int odise_hip_jpeg_info'), src.index("namespace odise {\nstruct JpegDims")
    out = src[:a] + src[b:c] + src[d:e] + src[e:f]
    return out.replace('#include "common.h"', f'#include "{ROOT}/odise_amd/csrc/common.h"')


def main():
    iters = sys.argv[1] if len(sys.argv) > 1 else "4000"
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(ROOT, "odise_amd", "csrc", "jpeg.hip")) as f:
            open(os.path.join(tmp, "jpeg_host_only.cpp"), "w").write(host_half(f.read()))
        open(os.path.join(tmp, "stubs.cpp"), "w").write(
            '#include <stdarg.h>\n#include <stdio.h>\nnamespace odise {\nstatic thread_local char g_err[1024];\n'
            'void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap); }\n}\n')
        subprocess.check_call([HIPCC, *SAN, "--cuda-host-only", "-x", "hip", "-c", os.path.join(tmp, "jpeg_host_only.cpp"), "-o", os.path.join(tmp, "jpeg.o")])
        subprocess.check_call([CLANG, *SAN, os.path.join(HERE, "fuzz_main.cpp"), os.path.join(tmp, "stubs.cpp"), os.path.join(tmp, "jpeg.o"),
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", os.path.join(tmp, "fuzz")])
        cases = {"base420": dict(quality=75, subsampling=2), "base444opt": dict(quality=90, subsampling=0, optimize=True),
                 "base422rst": dict(quality=60, subsampling=1, restart_marker_blocks=3), "prog420": dict(quality=75, subsampling=2, progressive=True),
                 "prog444": dict(quality=85, subsampling=0, progressive=True), "prog422rst": dict(quality=50, subsampling=1, progressive=True, restart_marker_rows=1)}
        seeds = []
        for k, kw in cases.items():
            seeds.append(os.path.join(tmp, k + ".jpg"))
            open(seeds[-1], "wb").write(_jpeg(_picture(41, 53, len(k)), **kw))
        ex = Image.Exif()
        ex[0x0112] = 6
        buf = io.BytesIO()
        Image.fromarray(_picture(33, 20, 2)).save(buf, "JPEG", quality=80, exif=ex.tobytes())
        seeds.append(os.path.join(tmp, "exif.jpg"))
        open(seeds[-1], "wb").write(buf.getvalue())
        seeds.append(os.path.join(tmp, "greyprog.jpg"))
        open(seeds[-1], "wb").write(_jpeg(_picture(33, 20, 2), mode="L", quality=80, progressive=True))
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", FUZZ_ITERS=iters)
        sys.exit(subprocess.call([os.path.join(tmp, "fuzz"), *seeds], env=env))


if __name__ == "__main__":
    main()
