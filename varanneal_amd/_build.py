"""Build libvaranneal_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m varanneal_amd._build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvaranneal_amd.so")
SOURCES = ["va_capi.hip", "va_kernels.hip", "va_eval5.hip", "va_lbfgsb.hip", "va_nnet.hip"]
HEADERS = ["va_core.h", "va_tile2.h", "va_tile3.h", "va_tile4.h", "va_tile5.h", "va_eval3.h", "va_eval4.h", "va_eval5.h", "va_eval_flat.h", "va_epilogue.h", "va_persist.h", "va_persist_geo.h", "va_measure.h", "va_device.h", "va_nnet.h", "va_nnet_kernels.h", os.path.join("..", "..", "include", "varanneal_amd.h")]
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


def _compile_link(out, defines, verbose):
    """every translation unit to an object of its own, side by side (the evaluation kernels take minutes each),
    then one link"""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build", os.path.basename(out).replace(".so", ""))
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")]

    def stale(obj, dep, cmd):
        """object missing, built by another command line, or older than any file its unit includes"""
        try:
            with open(dep) as fh:
                txt = fh.read()
            if txt.split("\n", 1)[0] != "# " + " ".join(cmd):
                return True
            t = os.path.getmtime(obj)
            files = txt.split(":", 1)[1].replace("\\\n", " ").split()
            return any(os.path.getmtime(f) > t for f in files)
        except (OSError, IndexError):
            return True

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        dep = obj + ".d"
        cmd = [HIPCC] + cflags + defines + ["-c", "-o", obj, os.path.join(CSRC, src)]
        if not stale(obj, dep, cmd):
            return obj
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd + ["-MD", "-MF", dep + ".tmp"])
        with open(dep + ".tmp") as fh:
            body = fh.read()
        with open(dep, "w") as fh:
            fh.write("# " + " ".join(cmd) + "\n" + body)
        os.remove(dep + ".tmp")
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_stamps(verbose=True):
    """Diagnostic library with in-kernel wall-clock stamps (tools/timeline.py); never the product."""
    return _compile_link(os.path.join(HERE, "libvaranneal_amd_stamps.so"), ["-DVA_STAMPS"], verbose)


def build_variant(tag, defines, verbose=True):
    """Diagnostic library libvaranneal_amd_<tag>.so with extra -D switches (ablations for profiles/); never the
    product.  Select it with VARANNEAL_AMD_LIB=<path> (tools/ab.sh)."""
    return _compile_link(os.path.join(HERE, "libvaranneal_amd_%s.so" % tag), list(defines), verbose)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    return _compile_link(OUT, [], verbose)


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    elif "--stamps" in sys.argv:
        build_stamps()
    else:
        build(force="--force" in sys.argv)
