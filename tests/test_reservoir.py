"""The FDRP / qFDRP reservoir (fdrp.rs:81-94, qfdrp.rs:81-94) as a RESERVOIR.

The reference draws j from an OS-seeded thread_rng (fdrp.rs:90): its sampling branch is not reproducible, so oracle and device
share the counter-based draw orc_sample_j(seed, tid, pos, total).  Bit-equality between the two says nothing about whether
that draw makes Algorithm R a uniform reservoir.  Here: the draw is uniform on 1..=n (chi-square), and Algorithm R driven by
it keeps every one of a site's n reads with probability D / n whatever its arrival rank, slots are replaced uniformly, and the
stored set has min(n, D) members.  (The device side of the statement is tests/test_gpu_fdrp.py::test_reservoir_* .)"""
import numpy as np
from scipy import stats

from oracle import pyoracle


def sample_j_np(seed, tid, pos, total):
    """orc_sample_j vectorised (splitmix64 over seed, tid, pos, total) -- checked against the C function below"""
    u = np.uint64
    with np.errstate(over="ignore"):
        z = u(seed) ^ ((np.asarray(tid, np.uint64) & u(0xffffffff)) << u(32) | (np.asarray(pos, np.uint64) & u(0xffffffff)))
        z = z + u(0x9e3779b97f4a7c15) * (np.asarray(total, np.uint64) & u(0xffffffff))
        z = (z ^ (z >> u(30))) * u(0xbf58476d1ce4e5b9)
        z = (z ^ (z >> u(27))) * u(0x94d049bb133111eb)
        z = z ^ (z >> u(31))
    return (z % np.asarray(total, np.uint64)).astype(np.int64) + 1


def test_numpy_restatement_equals_the_oracle_function():
    rng = np.random.default_rng(1)
    for _ in range(2000):
        seed, tid, pos, tot = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 30)), int(rng.integers(0, 2 ** 31 - 1)), int(rng.integers(1, 100000))
        assert pyoracle.sample_j(seed, tid, pos, tot) == int(sample_j_np(seed, tid, pos, tot))


def test_draw_is_uniform_on_1_to_n():
    # for every n the reference can ask for at depths 41 ... 1000: over many sites the draw hits each of 1..n equally often
    pos = np.arange(400_000, dtype=np.int64) * 7 + 11
    for n in (41, 50, 64, 97, 256, 1000):
        for seed in (0, 9, 123456789):
            j = sample_j_np(seed, 3, pos, n)
            assert j.min() >= 1 and j.max() <= n
            cnt = np.bincount(j, minlength=n + 1)[1:]
            chi2, p = stats.chisquare(cnt)
            assert p > 1e-4, (n, seed, chi2, p)
    # and along ONE site's sequence of totals (what a deep site sees): successive draws are not correlated with the total
    tot = np.arange(41, 200_041, dtype=np.int64)
    j = sample_j_np(5, 0, 12345, tot)
    frac = (j - 0.5) / tot                              # uniform on (0, 1) if j is uniform on 1..total
    assert stats.kstest(frac, "uniform").pvalue > 1e-4


def test_algorithm_r_with_this_draw_is_a_uniform_reservoir():
    """fdrp.rs:81-94 literally, for 20 000 sites of depth n = 50 ... 80 with D = 40: inclusion frequency of the k-th arriving read
    is D / n for every k (chi-square over the arrival ranks), every slot is the target of a replacement equally often, and the
    stored set has D members"""
    D = 40
    for n, seed in ((50, 7), (64, 8), (80, 9)):
        sites = np.arange(20_000, dtype=np.int64) * 13 + 1000
        slot = np.tile(np.arange(D), (len(sites), 1))                # slot[s][q] = arrival rank of the read stored in slot q
        repl = np.zeros(D, np.int64)
        for k in range(D, n):                                        # the (k+1)-th read arrives: num_total_read = k + 1
            j = sample_j_np(seed, 0, sites, k + 1)
            hit = j <= D
            slot[hit, j[hit] - 1] = k
            repl += np.bincount(j[hit] - 1, minlength=D)
        assert (np.sort(slot, axis=1)[:, 1:] != np.sort(slot, axis=1)[:, :-1]).all()      # D distinct reads per site
        incl = np.bincount(slot.ravel(), minlength=n)
        expect = len(sites) * D / n
        chi2 = ((incl - expect) ** 2 / (expect * (1 - D / n))).sum()      # binomial variance per rank
        assert stats.chi2.sf(chi2, n - 1) > 1e-4, (n, chi2)
        assert stats.chisquare(repl).pvalue > 1e-4
