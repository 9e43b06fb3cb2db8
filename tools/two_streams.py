"""fused PDR+LPMD steps of config 2 on ONE context against the same number of steps spread over 2 / 3 / 4 contexts (a stream and work
buffers each, one host thread each) of the same GPU: how much of a step is the machine filling and draining at kernel boundaries?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth, batches
c = synth.chr19_10m()
n = len(c["read_start"])
bt = batches.device_batch(c, device="cuda:0")
p = metheor_amd.PdrLpmdParams()
K = 1200
for nctx in (1, 2, 3, 4):
    engs = []
    for _ in range(nctx):
        st = torch.cuda.Stream()
        engs.append((metheor_amd.Engine(0, stream=st.cuda_stream), st))
    def run(e, k):
        for _ in range(k):
            e.reset(); e.pdr_lpmd_accumulate(bt, p)
        e.sync()
    best = None
    for rep in range(3):
        th = [threading.Thread(target=run, args=(e, K // nctx)) for e, _ in engs]
        t0 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    steps = (K // nctx) * nctx
    print("%d context(s): %d steps in %.4f s = %.4f ms per step = %.1f G reads/s" % (nctx, steps, best, best / steps * 1e3, n * steps / best / 1e9), flush=True)
    for e, _ in engs:
        assert e.pdr_count() == 726028
        e.close()
