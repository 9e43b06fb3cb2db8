"""host-to-device copy rates on this box: pageable vs page-locked source, and parallel pread() of a /dev/shm file into page-locked pieces"""
import os, time, threading, torch, numpy as np
n = 1 << 30
dev = torch.device("cuda:0")
dst = torch.empty(n, dtype=torch.uint8, device=dev)
src = torch.empty(n, dtype=torch.uint8).random_(0, 255)
for name, s in (("pageable", src), ("page-locked", src.pin_memory())):
    for _ in range(2):
        dst.copy_(s, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        dst.copy_(s, non_blocking=True)
    torch.cuda.synchronize()
    print("%-12s H2D %.1f GB/s" % (name, 4 * n / (time.perf_counter() - t0) / 1e9))
path = "/dev/shm/h2d_probe.bin"
src.numpy().tofile(path)
piece = 16 << 20
for nthr in (1, 2, 4, 8, 16):
    bufs = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2 * nthr)]
    views = [b.numpy() for b in bufs]
    fd = os.open(path, os.O_RDONLY)
    streams = [torch.cuda.Stream() for _ in range(nthr)]
    def work(k):
        evs = [None, None]
        with torch.cuda.stream(streams[k]):
            for j, off in enumerate(range(k * piece, n, nthr * piece)):
                b = 2 * k + (j & 1)
                if evs[j & 1] is not None:
                    evs[j & 1].synchronize()
                got = os.preadv(fd, [views[b]], off)
                dst[off:off + got].copy_(bufs[b][:got], non_blocking=True)
                e = torch.cuda.Event(); e.record(streams[k]); evs[j & 1] = e
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    print("pread -> page-locked pieces (16 MiB x 2 per thread), %2d threads: %.1f GB/s" % (nthr, n / (time.perf_counter() - t0) / 1e9))
    os.close(fd)
os.remove(path)
