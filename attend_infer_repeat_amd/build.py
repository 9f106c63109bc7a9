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
SOURCES = ["st_kernels.hip", "canvas_kernels.hip", "gemm_kernels.hip", "pointwise_kernels.hip", "loss_kernels.hip", "engine_kernels.hip",
           "comm_rccl.hip", "comm_ipc.hip", "mlp_chain_kernels.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths):
    """Content hash of the build inputs.  File NAMES enter relative to the package (the repo lives under different roots in the
    build container and on the GPU box: an absolute path in the hash would force a rebuild -- by every rank at once -- there)."""
    h = hashlib.sha256()
    for p in sorted(paths, key=os.path.basename):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode()); h.update(f.read())
    return h.hexdigest()


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return sources() + headers + [os.path.join(PKG, "..", "include", "air_hip.h")]


def source_digest():
    """Digest of the build inputs as they are on disk now (what a fresh build would embed as air_build_digest())."""
    return _digest([os.path.abspath(d) for d in _deps()])


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared object.  Returns the library path.
    Safe to call from several processes at once (one rank per GPU): an exclusive file lock serialises them, objects are
    compiled in a private temporary directory and the library is moved into place atomically."""
    import fcntl
    import shutil
    import tempfile
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libair_hip.sha256")
    digest = _digest([os.path.abspath(d) for d in _deps()])

    def fresh():
        return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest

    if not force and fresh():
        return LIB
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():          # another process built it while this one waited for the lock
                return LIB
            tmp = tempfile.mkdtemp(prefix="air_build_", dir=LIBDIR)
            try:
                objs = []
                for s in sources():
                    o = os.path.join(tmp, os.path.basename(s).replace(".hip", ".o"))
                    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o,
                           "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", f'-DAIR_BUILD_DIGEST="{digest}"']
                    if verbose:
                        print(" ".join(cmd), file=sys.stderr)
                    subprocess.check_call(cmd)
                    objs.append(o)
                out = os.path.join(tmp, "libair_hip.so")
                cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                subprocess.check_call(cmd)
                os.replace(out, LIB)
                with open(stamp + ".tmp", "w") as f:
                    f.write(digest)
                os.replace(stamp + ".tmp", stamp)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
