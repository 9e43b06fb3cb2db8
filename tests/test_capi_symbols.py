"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/metheor_hip.h
declares, and refuses to run without a gfx950 device (no CPU fallback)."""
import os
import re

import pytest

import metheor_amd
from metheor_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "metheor_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mth_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    L = metheor_amd.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libmetheor_hip.so does not export " + s
    assert sorted(capi.SYMBOLS) == syms, "capi.SYMBOLS out of sync with include/metheor_hip.h"


def test_abi_version_and_strerror():
    L = metheor_amd.lib()
    assert L.mth_abi_version() == 1
    assert L.mth_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.mth_strerror(-3)
    assert L.mth_timing_num_kernels() >= 4


def test_lpmd_from_counts_matches_reference_expression():
    import numpy as np
    L = metheor_amd.lib()
    assert L.mth_lpmd_from_counts(48, 48) == 0.5                       # lpmd.rs:219
    assert np.isnan(L.mth_lpmd_from_counts(0, 0))                      # lpmd.rs:268
    # lpmd.rs:11-12: i32 counters wrap in a release build
    c, d = 2 ** 31 + 5, 7
    wc = np.int32(np.uint32(c & 0xffffffff)); wd = np.int32(d)
    want = np.float32(wd) / np.float32(np.int32(np.uint32((int(wc) + int(wd)) & 0xffffffff)))
    assert np.float32(L.mth_lpmd_from_counts(c, d)) == want


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(metheor_amd.MthError) as e:
        metheor_amd.Engine(0)
    assert e.value.status == -3
