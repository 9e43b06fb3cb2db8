#!/bin/bash
# fresh process per variant, file written once into /dev/shm (1.78 GB of pseudo-random bytes)
python - <<PY
import numpy as np
np.random.default_rng(1).integers(0, 255, size=1_784_741_484, dtype=np.uint8).tofile("/dev/shm/h2d_probe.bin")
PY
for rep in 1 2; do
tools/h2d_probe /dev/shm/h2d_probe.bin 0
for cfg in "4 1024" "4 2048" "4 4096" "6 1024" "6 2048" "8 1024" "8 2048" "8 512"; do tools/h2d_probe /dev/shm/h2d_probe.bin 1 $cfg; done
H2D_NONCOHERENT=1 tools/h2d_probe /dev/shm/h2d_probe.bin 1 6 2048
done
rm -f /dev/shm/h2d_probe.bin
