"""The k -> (i, j) step of the deep FDRP / qFDRP finalize (metheor_amd/csrc/mth_fdrp.hip finalize_deep, mth_fileorder.hip k_fo_fdrp;
reference loop order fdrp.rs:128-141): an f32 estimate of the row followed by exact integer steps.  This restates the device arithmetic
in numpy float32 and checks it against the closed form over EVERY pair index at the depths where the estimate alone goes wrong (round 5:
with one conditional step each way, pair 71 993 996 of 12 000 stored reads landed in row 11 997 instead of 11 996)."""
import numpy as np
import pytest


def device_rows(n_s, k):
    two_n = 2 * n_s
    bq = np.float32(two_n - 1)
    x = np.maximum((bq * bq - np.float32(8.0) * k.astype(np.float32)).astype(np.float32), np.float32(0.0))
    pi = ((bq - np.sqrt(x).astype(np.float32)) * np.float32(0.5)).astype(np.int64)
    pi = np.clip(pi, 0, n_s - 2)
    est = pi.copy()
    off = lambda r: (r * (two_n - r - 1)) >> 1
    while True:                                    # while (k < off) --pi
        m = k < off(pi)
        if not m.any():
            break
        pi = np.where(m, pi - 1, pi)
    while True:                                    # while (pi + 2 < nS && k >= off(pi + 1)) ++pi
        m = (pi + 2 < n_s) & (k >= off(pi + 1))
        if not m.any():
            break
        pi = np.where(m, pi + 1, pi)
    return est, pi, k - off(pi) + pi + 1


@pytest.mark.parametrize("n_s", [2, 3, 64, 257, 2049, 12000, 16384])
def test_every_pair_index(n_s):
    two_n = 2 * n_s
    r = np.arange(n_s - 1, dtype=np.int64)
    offs = (r * (two_n - r - 1)) >> 1
    total = n_s * (n_s - 1) // 2
    worst = 0
    starts = range(0, total, 16_000_000) if n_s <= 12000 else [0, total - 48_000_000, total - 32_000_000, total - 16_000_000]   # 16384: both ends
    for k0 in starts:
        k = np.arange(k0, min(total, k0 + 16_000_000), dtype=np.int64)
        est, pi, pj = device_rows(n_s, k)
        true_i = np.searchsorted(offs, k, side="right") - 1
        assert np.array_equal(pi, true_i)
        assert np.array_equal(pj, k - offs[true_i] + true_i + 1)
        assert (pj > pi).all() and (pj < n_s).all()
        worst = max(worst, int(np.abs(est - true_i).max()))
    if n_s >= 12000:
        assert worst >= 2          # the reason the steps are loops
