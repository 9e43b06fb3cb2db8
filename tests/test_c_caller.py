"""A native caller of the C ABI written in C: tests/c_caller/caller.c is compiled with gcc -std=c99
-pedantic -Werror against include/metheor_hip.h (CPU: the header is C-clean and the program links), and on the MI355X it
runs test1.bam's SoA through mth_pdr_lpmd_accumulate / mth_pdr_fetch / mth_lpmd_global and checks the reference's golden
rows (pdr.rs:226-237, lpmd.rs:219)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import bamio, pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_array(ctype, name, a):
    return "static const %s %s[] = {%s};\n" % (ctype, name, ", ".join(str(int(x)) for x in a) or "0")


def build_caller(golden_dir, out_dir):
    soa = pyoracle.Reads.decode(bamio.read_bam(os.path.join(golden_dir, "test1.bam"))).soa()
    n, nc = len(soa["start"]), len(soa["cpg_pos"])
    h = "#define T1_N_READS %d\n#define T1_N_CPGS %d\n" % (n, nc)
    h += _c_array("int32_t", "t1_start", soa["start"]) + _c_array("int32_t", "t1_end", soa["end"])
    h += _c_array("uint8_t", "t1_mapq", soa["mapq"]) + _c_array("uint8_t", "t1_fwd", soa["fwd"])
    h += _c_array("uint32_t", "t1_off", soa["cpg_off"]) + _c_array("uint32_t", "t1_pos", soa["cpg_pos"].astype(np.uint32))
    h += _c_array("uint8_t", "t1_rel", soa["cpg_rel"])
    with open(os.path.join(out_dir, "test1_soa.h"), "w") as f:
        f.write(h)
    import metheor_amd
    metheor_amd.lib()                                      # built (and loadable) before linking against it
    libdir = os.path.join(ROOT, "metheor_amd")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib")
    exe = os.path.join(out_dir, "c_caller")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), "-I", out_dir,
           os.path.join(ROOT, "tests", "c_caller", "caller.c"), "-o", exe, "-L" + libdir, "-lmetheor_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link," + rocm_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_clean_and_caller_links(golden_dir, tmp_path):
    exe = build_caller(golden_dir, str(tmp_path))
    assert os.path.exists(exe)
    # the host header too
    src = tmp_path / "hosthdr.c"
    src.write_text('#include "metheor_host.h"\n#include "metheor_hip.h"\nint main(void) { return MTH_ABI_VERSION == 1 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "h.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_caller_golden_rows(golden_dir, tmp_path):
    exe = build_caller(golden_dir, str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c caller ok: 4 sites, lpmd 0.5" in r.stdout
