hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_probe tools/h2d_probe.cpp -lpthread 2>&1 | grep -i error
python - <<PY
import numpy as np
np.random.default_rng(1).integers(0, 255, size=1_784_741_484, dtype=np.uint8).tofile("/dev/shm/h2d_probe.bin")
PY
for rep in 1 2; do
/tmp/h2d_probe /dev/shm/h2d_probe.bin 0
/tmp/h2d_probe /dev/shm/h2d_probe.bin 5
for t in 2 4 8; do /tmp/h2d_probe /dev/shm/h2d_probe.bin 6 $t 65536; done
/tmp/h2d_probe /dev/shm/h2d_probe.bin 6 4 16384
done
rm -f /dev/shm/h2d_probe.bin
