"""Build libuwir_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m underwater_image_restoration_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libuwir_hip.so")
SOURCES = ["elementwise.hip", "dwconv.hip", "gemm.hip", "gemm_wide.hip", "gemm_wide288.hip", "gate.hip", "ffn.hip", "attention.hip", "fft.hip", "tokens.hip", "unet.hip", "linattn.hip", "metrics.hip",
           "teacher.hip", "enhance.hip", "enhance_cv.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified file); without it every softmax /
# epilogue touch of an accumulator costs a v_accvgpr_read/write pair.
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "uwir.h")]

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers + [os.path.abspath(__file__)]):
            # attention: fmaxf on MFMA outputs would otherwise get a canonicalising v_max per element
            extra = ["-fno-honor-nans"] if src == "attention.hip" else []
            cmd = [HIPCC] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
