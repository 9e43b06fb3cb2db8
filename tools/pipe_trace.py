"""timeline of the pipelined PDR + LPMD steps on config 2.
  run (under rocprofv3 --kernel-trace --output-format csv -d DIR):   python tools/pipe_trace.py run [steps]
  analyse the trace:                                                  python tools/pipe_trace.py show DIR
Also prints the host's enqueue time per step (loop without a sync) next to the synchronised time."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps):
    import metheor_amd
    from metheor_amd import batches, synth
    c = synth.chr19_10m()
    eng = metheor_amd.Engine(0)
    bt = batches.device_batch(c, device="cuda:0")
    p = metheor_amd.PdrLpmdParams()
    for _ in range(300):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print("steps %d: enqueue %.4f ms/step, with sync %.4f ms/step" % (steps, (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
    eng.close()


def show(d):
    import csv
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    ev = []
    for r in rows:
        n = r["Kernel_Name"]
        k = "idx" if "k_build_index" in n else "tile" if "k_pdr_lpmd_tile" in n else "gather" if "k_gather" in n else None
        if k:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
    ev.sort()
    ev = ev[len(ev) // 2:]                      # steady state: the second half
    t0 = ev[0][0]
    print("last 18 launches (us from the first shown; queue; kernel; start; end; duration):")
    for s, e, k, q in ev[-18:]:
        print("  q%-3s %-6s %9.1f %9.1f %7.1f" % (q, k, (s - ev[-18][0]) / 1e3, (e - ev[-18][0]) / 1e3, (e - s) / 1e3))
    tiles = [x for x in ev if x[2] == "tile"]
    per = (tiles[-1][0] - tiles[0][0]) / 1e3 / (len(tiles) - 1)
    print("tile launches %d, period %.2f us, mean duration: idx %.1f tile %.1f gather %.1f" % (
        len(tiles), per, *[sum(e - s for s, e, k, q in ev if k == kk) / 1e3 / max(1, sum(1 for x in ev if x[2] == kk)) for kk in ("idx", "tile", "gather")]))
    # fraction of the time 0 / 1 / 2 tile kernels are in flight
    pts = sorted([(s, 1) for s, e, k, q in tiles] + [(e, -1) for s, e, k, q in tiles])
    cur, last, acc = 0, pts[0][0], {}
    for t, dlt in pts:
        acc[cur] = acc.get(cur, 0) + (t - last)
        cur += dlt; last = t
    tot = sum(acc.values())
    print("tile kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(acc.items())})


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
    else:
        show(sys.argv[2])
