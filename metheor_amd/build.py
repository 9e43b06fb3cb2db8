"""In-tree build of the gfx950 engine: hipcc -> metheor_amd/libmetheor_hip.so (+ the `metheor` CLI
when its sources exist).  No JIT cache, no pip install: the built files travel with the tree."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmetheor_hip.so")
ARCH = "gfx950"
# MTH_EXTRA_HIPFLAGS: experiment builds (-DMTH_TILE_TRACE ...); the recorded BUILD_INFO carries the flags
# -ffp-contract=off: every float this engine emits must equal the reference's UNFUSED f32 expressions (Rust never contracts
# a*b+c); hipcc's default is contract=fast
HIP_FLAGS = os.environ.get("MTH_EXTRA_HIPFLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off", "-Wno-unused-result"]

HIP_SOURCES = ["mth_api.hip", "mth_pdr_lpmd.hip", "mth_pdr_wide.hip", "mth_quartet.hip", "mth_scan.hip", "mth_sites.hip", "mth_mhl_tile.hip", "mth_fdrp.hip", "mth_fdrp_wtile.hip", "mth_pairs.hip", "mth_decode.hip", "mth_sort.hip", "mth_fileorder.hip", "mth_inflate.hip", "mth_rccl.hip", "mth_tag.hip"]
HOST_LIB = os.path.join(HERE, "libmetheor_host.so")
HOST_SOURCES = [os.path.join("host", "bam_reader.cpp"), os.path.join("host", "host_api.cpp"),
                os.path.join("host", "parallel_decode.cpp"), os.path.join("host", "synth_bam.cpp"), os.path.join("host", "bai_index.cpp"), os.path.join("host", "sam_text.cpp")]
HOST_HEADERS = [os.path.join("host", "bam_reader.h"), os.path.join("host", "parallel_decode.h"), os.path.join("host", "bai_internal.h"), os.path.join("host", "sam_text.h"), os.path.join("..", "..", "include", "metheor_host.h")]
HEADERS = ["mth_common.h", "mth_ctx.h", "mth_scan.h", "mth_tile_dev.h", "mth_wave_tile.h", os.path.join("..", "..", "include", "metheor_hip.h")]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine is HIP-only (no CPU fallback)")


BUILD_INFO = os.path.join(HERE, "BUILD_INFO.json")


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def source_files():
    """every file the three artefacts are built from (relative to the package directory)"""
    out = [os.path.join("csrc", s) for s in HIP_SOURCES + HEADERS + HOST_SOURCES + HOST_HEADERS] + [os.path.join("csrc", "host", "cli_main.cpp")]
    return sorted(set(os.path.normpath(p) for p in out))


def source_hashes():
    return {p: _sha(os.path.join(HERE, p)) for p in source_files()}


def _write_build_info(rebuilt):
    """metheor_amd/BUILD_INFO.json: what the in-tree binaries were built from (travels with them; tests/test_build_info.py
    compares the hashes with the tree, so a stale .so is a test failure, not a silent mismatch)"""
    try:
        ver = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        ver = "?"
    info = {"built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "rebuilt_this_call": bool(rebuilt), "arch": ARCH, "hipcc": ver,
            "hip_flags": HIP_FLAGS, "artefacts": {os.path.basename(p): _sha(p) for p in (LIB, HOST_LIB, CLI) if os.path.exists(p)},
            "sources": source_hashes()}
    old = None
    if os.path.exists(BUILD_INFO):
        try:
            old = json.load(open(BUILD_INFO))
        except Exception:
            old = None
    if old and not rebuilt and old.get("sources") == info["sources"] and old.get("artefacts") == info["artefacts"]:
        return
    with open(BUILD_INFO, "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)


def build_host(force=False, verbose=False):
    """g++ -> metheor_amd/libmetheor_host.so (BGZF/BAM reader + XM decode; needs zlib, no GPU)"""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HOST_HEADERS]
    if force or _stale(HOST_LIB, deps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-shared", "-pthread", "-o", HOST_LIB] + srcs + ["-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


CLI = os.path.join(HERE, "metheor")


def build_cli(force=False, verbose=False):
    """g++ -> metheor_amd/metheor (the drop-in command line; links the two C-ABI libraries)"""
    src = os.path.join(CSRC, "host", "cli_main.cpp")
    deps = [src, LIB, HOST_LIB, os.path.join(HERE, "..", "include", "metheor_hip.h"),
            os.path.join(HERE, "..", "include", "metheor_host.h")]
    if force or _stale(CLI, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib")
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-o", CLI, src, "-L" + HERE, "-lmetheor_hip",
               "-lmetheor_host", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + rocm_lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


def build(force=False, verbose=False):
    # staleness: mtimes as make would, plus the recorded source hashes (a checkout can rewind mtimes)
    if not force and os.path.exists(BUILD_INFO):
        try:
            if json.load(open(BUILD_INFO)).get("sources") != source_hashes():
                force = True
        except Exception:
            force = True
    before = {p: (os.path.getmtime(p) if os.path.exists(p) else 0) for p in (LIB, HOST_LIB, CLI)}
    lib = _build_libs(force, verbose)
    build_cli(force, verbose)
    rebuilt = any((os.path.getmtime(p) if os.path.exists(p) else 0) != before[p] for p in before)
    _write_build_info(rebuilt)
    return lib


def _build_libs(force=False, verbose=False):
    build_host(force, verbose)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            # -ffp-contract=off: every float this engine emits must equal the reference's UNFUSED f32
            # expressions (Rust never contracts a*b+c); hipcc's default is contract=fast
            cmd = [hipcc, "--offload-arch=" + ARCH] + HIP_FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
