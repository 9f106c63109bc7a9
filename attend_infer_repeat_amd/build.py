"""Builds libair_hip.so (the gfx950 kernels behind include/air_hip.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the resulting
attend_infer_repeat_amd/lib/libair_hip.so travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libair_hip.so")
SOURCES = ["st_kernels.hip", "gemm_kernels.hip", "pointwise_kernels.hip", "loss_kernels.hip", "engine_kernels.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode()); h.update(f.read())
    return h.hexdigest()


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared object.  Returns the library path."""
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, "air_common.h"), os.path.join(PKG, "..", "include", "air_hip.h")]
    stamp = os.path.join(LIBDIR, "libair_hip.sha256")
    digest = _digest([os.path.abspath(d) for d in deps])
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    objs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s).replace(".hip", ".o"))
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o,
               "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
