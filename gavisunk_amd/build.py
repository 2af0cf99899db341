"""Builds libgavisunk.so (HIP kernels + C-ABI + native host parsers) in-tree for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgavisunk.so")
LIB_TUNING = os.path.join(HERE, "libgavisunk_tuning.so")  # -DGS_TUNING: experiment switches for tools/ (never shipped, never loaded by default)
SOURCES = ["gs_device.hip", "gs_filter.hip", "gs_pair.hip", "gs_chain.hip", "gs_diag.hip", "gs_format.hip", "gs_sunkdef.hip", "gs_ingest.hip", "gs_bgzf.hip", "gs_host.cpp", "gs_inflate.cpp"]
LIB_SYNTH = os.path.join(HERE, "libgssynth.so")  # bench / test support: synthetic inputs generated on the GPU (gs_synth.hip)


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libgavisunk.so cannot be built (there is no CPU fallback)")


def stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("gs_synth")] + [os.path.join(ROOT, "include", "gavisunk.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _deps_mtime() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "gavisunk.h")]
    return max(os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False, tuning: bool = False, variant: str = "", defines=()) -> str:
    """tuning=True builds libgavisunk_tuning.so with -DGS_TUNING: the phase-off / pipeline-depth switches the timing
    experiments under tools/ use (GAVISUNK_LIB=.../libgavisunk_tuning.so selects it).  The product library has none.
    variant="x" + defines: libgavisunk_x.so compiled with those -D switches (kernel A/B experiments, tools/kbench.sh).
    Every source is compiled to an object of its own (in parallel, only when it or a header changed), then linked."""
    lib = LIB_TUNING if tuning else LIB
    tag = "tuning" if tuning else "product"
    if variant:
        lib = os.path.join(HERE, f"libgavisunk_{variant}.so")
        tag = "variant_" + variant
        force = True
    if not force and not stale(lib):
        return lib
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", CSRC] + (
        ["-DGS_TUNING"] if tuning else []) + list(defines)
    hdr_t = _deps_mtime()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((s, subprocess.Popen(cmd)))
    failed = [s for s, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs + ["-lz", "-pthread", "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return lib


def build_synth(force: bool = False, verbose: bool = False) -> str:
    src = os.path.join(CSRC, "gs_synth.hip")
    if not force and os.path.exists(LIB_SYNTH) and os.path.getmtime(LIB_SYNTH) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "gs_synth.h"))):
        return LIB_SYNTH
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-I", CSRC, "-o", LIB_SYNTH, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_SYNTH


if __name__ == "__main__":
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        print(build(verbose=True, variant=name, defines=[x for x in sys.argv if x.startswith("-D")]))
    else:
        build_synth(force="--force" in sys.argv, verbose=True)
        print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv))
