"""Builds libmgx.so (HIP kernels + C ABI) for gfx950 in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmgx.so")
SOURCES = ["mgx_common.cpp", "mgx_tables.cpp", "sortdedup_pack.cpp", "sortdedup_route.cpp", "pairhmm_pack_batch.cpp", "mgx_pairhmm.hip", "mgx_sortdedup.hip", "mgx_smithwaterman.hip", "mgx_bgzf.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-ffp-contract=off",
         # the reference runs with MXCSR.FTZ set (IntelPairHmm.cc:230), which flushes fp64 results too
         "-Xarch_device", "-fdenormal-fp-math=preserve-sign",
         "-Wall", "-Wno-unused-function"]


SYNTH_LIB = os.path.join(HERE, "libmgx_synth.so")     # threaded workload generators (bench / test support, not the product)
SYNTH_SRC = os.path.join(CSRC, "synth", "synth_gen.cpp")


def build_synth(force=False, verbose=True):
    if not force and os.path.exists(SYNTH_LIB) and os.path.getmtime(SYNTH_LIB) >= os.path.getmtime(SYNTH_SRC):
        return SYNTH_LIB
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-o", SYNTH_LIB, SYNTH_SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return SYNTH_LIB


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    build_synth(force, verbose)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = [os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))] + [
        os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest(deps):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        objs.append(o)
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
