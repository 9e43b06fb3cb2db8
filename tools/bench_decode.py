#!/usr/bin/env python3
"""Device-side BAM record + XM decode throughput (NOT the BASELINE metric).
python tools/bench_decode.py [--reads N]: synthetic Bismark-style BAM (config-2 shape) -> BGZF inflate and record
walk on the host (untimed here) -> mth_decode_records, timed with the stream (a) already resident in HBM,
(b) handed over as pageable host memory (PCIe inclusive); the host decoder on the same file for context."""
import argparse, gzip, json, os, struct, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import torch, metheor_amd
    from metheor_amd import hostapi
    d = tempfile.mkdtemp(dir="/tmp")
    path = os.path.join(d, "syn.bam")
    from metheor_amd import synth
    L = int(58_617_616 * a.reads / 10_000_000)
    hostapi.write_synthetic_bam(path, synth.make_contig(0, L, a.reads, 0.02, np.random.default_rng(7)), seed=7)
    t0 = time.perf_counter()
    f = hostapi.BamFile(path)
    host = f.decode()
    t_host = time.perf_counter() - t0
    raw = gzip.decompress(open(path, "rb").read())
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o); o += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o); o += 8 + l_name
    body = np.frombuffer(raw, np.uint8)[o:].copy()
    mv = memoryview(body)
    offs = [0]
    n = len(body)
    while offs[-1] < n:
        offs.append(offs[-1] + 4 + int.from_bytes(mv[offs[-1]:offs[-1] + 4], "little"))
    offs = np.array(offs, np.uint64)
    n_rec = len(offs) - 1
    eng = metheor_amd.Engine(0)
    # parity on the full file: device decode == host decode
    eng.decode_records(body, offs)
    dev = eng.decoded_fetch()
    for k in ("tid", "start", "end", "mapq", "fwd", "cpg_off", "cpg_pos", "cpg_rel"):
        assert (dev[k] == host[k]).all(), k
    d_body, d_offs = torch.from_numpy(body).cuda(), torch.from_numpy(offs.view(np.int64)).cuda()
    out = {"workload": "synthetic Bismark BAM, %d x 150-bp reads" % n_rec, "record_bytes": int(n), "bytes_per_record": round(n / n_rec, 1),
           "cpg_calls": int(len(dev["cpg_pos"])), "parity_vs_host_decoder": "identical SoA"}
    for name, (b, of) in (("resident", (d_body, d_offs)), ("from_host_pageable", (body, offs))):
        for _ in range(2):
            eng.decode_records(b, of)
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.decode_records(b, of)
        eng.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {"ms": round(dt * 1e3, 3), "M_reads_per_s": round(n_rec / dt / 1e6, 1), "GB_per_s_record_bytes": round(n / dt / 1e9, 1)}
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(3):
        eng.decode_records(d_body, d_offs)
    tm = eng.timing()
    out["k_decode_ms_per_launch (count pass and fill pass)"] = round(tm["k_decode"][0], 4)
    # ---- the whole device load path: BGZF inflate + per-block record walk + decode (file bytes in pageable host memory) ----
    from tests.test_gpu_inflate import block_table
    fb, coff, csize, isize, hbytes, raw2 = block_table(path)
    fbn = np.frombuffer(fb, np.uint8)
    eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
    dev2 = eng.decoded_fetch()
    for k in ("tid", "start", "end", "mapq", "fwd", "cpg_off", "cpg_pos", "cpg_rel"):
        assert (dev2[k] == host[k]).all(), k
    for _ in range(2):
        eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
    eng.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(3):
        eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
    tm = eng.timing()
    out["bgzf_file_to_soa_on_device"] = {"compressed_bytes": len(fb), "blocks": int(len(coff)), "ms": round(dt * 1e3, 2),
                                          "M_reads_per_s": round(n_rec / dt / 1e6, 1), "compressed_GB_per_s": round(len(fb) / dt / 1e9, 2),
                                          "k_inflate_ms": round(tm["k_inflate"][0], 3), "k_decode_ms_per_launch": round(tm["k_decode"][0], 3),
                                          "k_inflate_GB_per_s_inflated": round(n / (tm["k_inflate"][0] * 1e-3) / 1e9, 1),
                                          "parity_vs_host_decoder": "identical SoA"}
    out["host_decoder"] = {"s": round(t_host, 3), "M_reads_per_s": round(n_rec / t_host / 1e6, 2), "threads": os.environ.get("METHEOR_THREADS", "default"),
                           "note": "whole libmetheor_host path incl. BGZF inflate"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
