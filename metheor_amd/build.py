"""In-tree build of the gfx950 engine: hipcc -> metheor_amd/libmetheor_hip.so (+ the `metheor` CLI
when its sources exist).  No JIT cache, no pip install: the built files travel with the tree."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmetheor_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["mth_api.hip", "mth_pdr_lpmd.hip"]
HEADERS = ["mth_common.h", "mth_ctx.h", os.path.join("..", "..", "include", "metheor_hip.h")]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine is HIP-only (no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall",
                   "-Wno-unused-result", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
