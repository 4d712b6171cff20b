"""Build libvaranneal_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m varanneal_amd._build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvaranneal_amd.so")
SOURCES = ["va_capi.hip", "va_kernels.hip", "va_nnet.hip"]
HEADERS = ["va_core.h", "va_tile2.h", "va_tile3.h", "va_tile4.h", "va_eval3.h", "va_eval4.h", "va_eval_flat.h", "va_epilogue.h", "va_device.h", "va_nnet.h", "va_nnet_kernels.h", os.path.join("..", "..", "include", "varanneal_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-sched-strategy=iterative-maxocc: measured on the whole library against the default scheduler
# (profiles/r02_ab_experiments.txt): C3 evaluation 9.67 -> 9.18 us, 4096 seeds 362 -> 353 us, everything else equal
SCHED = ["-mllvm", "-amdgpu-sched-strategy=iterative-maxocc"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-ldl"] + SCHED


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_stamps(verbose=True):
    """Diagnostic library with in-kernel wall-clock stamps (tools/timeline.py); never the product."""
    out = os.path.join(HERE, "libvaranneal_amd_stamps.so")
    cmd = [HIPCC] + FLAGS + ["-DVA_STAMPS", "-o", out] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    cmd = [HIPCC] + FLAGS + ["-o", OUT] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        build_stamps()
    else:
        build(force="--force" in sys.argv)
