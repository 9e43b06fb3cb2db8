"""Where the reference panics on the measured path, the oracle says so instead of inventing a value (CPU; the engine's side is
tests/test_gpu_fileorder.py::test_reference_window_panic_is_reproduced)."""
import numpy as np
import pytest

from oracle import bamio, pyoracle


def _rec(flag, xm, n=210):
    return bamio.Records([("ctg0", 5000)], [0], [1000], [flag], [40], [[(n << 4) | 0]], [bytes(xm)])


def test_fdrp_window_index_panic():
    """fdrp.rs:70-72: reverse strand, calls on query bases 0 and 202 -> sites 999 and 1201; for site 1201 the read passes both window
    tests (start_relative_pos = 0, end_relative_pos = 209) and the call at 999 indexes new_read[-1]"""
    xm = bytearray(b"." * 210); xm[0] = ord("z"); xm[202] = ord("Z")
    rd = pyoracle.Reads.decode(_rec(16, xm))
    for f in (rd.fdrp, rd.qfdrp):
        with pytest.raises(pyoracle.ReferencePanic):
            f(min_depth=1)
        assert len(f(min_depth=1, min_qual=41).tid) == 0          # fdrp.rs:205: skipped before add_read


@pytest.mark.parametrize("flag,second", [(0, 202), (16, 201), (16, 203)])
def test_fdrp_neighbouring_cases_do_not_panic(flag, second):
    """forward strand (first call at start, index 0), or the second site one base nearer / further (then a window test returns first)"""
    xm = bytearray(b"." * 210); xm[0] = ord("z"); xm[second] = ord("Z")
    rd = pyoracle.Reads.decode(_rec(flag, xm))
    rd.fdrp(min_depth=1), rd.qfdrp(min_depth=1)              # no ReferencePanic
