"""the ingredient of bench.py's e2e.floor on its own: python tools/h2d_floor_probe.py [file] (default: a 1.8-GB scratch file in /dev/shm)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
path = sys.argv[1] if len(sys.argv) > 1 else None
made = False
if path is None:
    path = "/dev/shm/h2d_floor_probe.bin"
    np.random.default_rng(1).integers(0, 256, size=1_786_000_000, dtype=np.uint8).tofile(path)
    made = True
try:
    for _ in range(2):
        print(json.dumps(bench.pageable_copy_probe(path)), flush=True)
finally:
    if made:
        os.remove(path)
