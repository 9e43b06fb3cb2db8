"""metheor_amd/synth_device.py (the torch generator bench.py uses for the WGBS-depth legs and the strong-scaling genome) on the CPU
device: the arrays must be a valid batch of the layout include/metheor_hip.h describes, with the distributions of SURVEY 8(d)."""
import numpy as np
import torch

from metheor_amd import synth_device as sd


def test_generated_contig_is_a_valid_sorted_batch():
    gen = torch.Generator(device="cpu")
    gen.manual_seed(7)
    t, info = sd.make_contig_tensors(3, 4_000_000, 250_000, 0.0091, gen, "cpu")
    st = t["read_start"].numpy().astype(np.int64)
    en = t["read_end"].numpy().astype(np.int64)
    off = t["cpg_off"].numpy().astype(np.int64)
    pos = t["cpg_pos"].numpy().view(np.uint32)
    rel = t["cpg_rel"].numpy().astype(np.int64)
    n = np.diff(off)
    assert info["n_reads"] == len(st) == 250_000 and info["n_calls"] == len(pos) == off[-1] and off[0] == 0
    assert (np.diff(st) >= 0).all() and (en - st == 149).all() and st.min() >= 0 and en.max() < 4_000_000
    assert 1.2 < n.mean() < 1.5                                             # 150 bp x 0.0091 sites per bp
    ro = np.repeat(np.arange(len(st)), n)
    p = (pos & 0x7fffffff).astype(np.int64)
    d = p - st[ro]
    assert d.min() >= -1 and d.max() <= 149                                 # a call sits in [start - 1, end]
    rev = rel - d                                                           # relpos = pos - start (+ 1 on the reverse strand)
    assert set(np.unique(rev).tolist()) <= {0, 1} and rel.min() >= 0 and rel.max() <= 149
    same = np.diff(ro) == 0
    assert (np.diff(p)[same] >= 2).all()                                    # ascending inside a read, CpGs never adjacent
    assert 0.4 < rev.mean() < 0.6
    mq = t["read_mapq"].numpy()
    assert 0.03 < (mq < 10).mean() < 0.07 and set(np.unique(mq[mq >= 10]).tolist()) == {42}
    meth = (pos >> 31).mean()
    assert 0.6 < meth < 0.72                                                # 0.3 x 0.1 + 0.7 x 0.9 = 0.66
    # same seed, same arrays
    gen.manual_seed(7)
    t2, _ = sd.make_contig_tensors(3, 4_000_000, 250_000, 0.0091, gen, "cpu")
    assert all(torch.equal(t[k], t2[k]) for k in ("read_start", "read_mapq", "cpg_off", "cpg_pos", "cpg_rel"))


def test_slice_region_owns_every_read_once_and_keeps_the_halo():
    gen = torch.Generator(device="cpu")
    gen.manual_seed(11)
    t, info = sd.make_contig_tensors(0, 1_000_000, 60_000, 0.02, gen, "cpu")
    cuts = [0, 123_456, 500_000, 500_100, 1_000_000]
    owned = 0
    for beg, end in zip(cuts[:-1], cuts[1:]):
        sl, n_own = sd.slice_region(t, beg, end)
        owned += n_own
        s = sl["read_start"].numpy().astype(np.int64)
        assert ((s >= beg - 150) & (s <= end)).all() and ((s >= beg) & (s < end)).sum() == n_own
        o = sl["cpg_off"].numpy().astype(np.int64)
        assert o[0] == 0 and o[-1] == len(sl["cpg_pos"]) == len(sl["cpg_rel"]) and len(o) == len(s) + 1
        # the slice's calls are the original reads' calls
        i0 = int(np.searchsorted(t["read_start"].numpy(), beg - 150, side="left"))
        oo = t["cpg_off"].numpy().astype(np.int64)
        assert torch.equal(sl["cpg_pos"], t["cpg_pos"][oo[i0]:oo[i0] + o[-1]])
    assert owned == info["n_reads"]


def test_hotspots_have_the_exact_depth():
    gen_windows = 40
    # (Batch needs a device: look at the tensors through make_contig_tensors with the same starts)
    g = torch.Generator(device="cpu"); g.manual_seed(50)
    stride = 1000 + 2 * 150 + 404
    per = int(50 * 1000 / 150)
    starts = (torch.arange(gen_windows, dtype=torch.int64)[:, None] * stride + 300 + torch.randint(0, 850, (gen_windows, per), generator=g, dtype=torch.int64)).reshape(-1)
    starts, _ = torch.sort(starts)
    t, info = sd.make_contig_tensors(0, gen_windows * stride + 1000, len(starts), 0.08, g, "cpu", starts=starts)
    assert info["n_reads"] == gen_windows * per
    w = (t["read_start"].numpy().astype(np.int64) // stride)
    assert (np.bincount(w, minlength=gen_windows) == per).all()
