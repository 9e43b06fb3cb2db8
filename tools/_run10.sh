mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pdr_lpmd.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t10.log
out=gpurun_out/ab10.log; : > $out
for rnd in 1 2 3; do
  echo "== prev" >> $out; METHEOR_HIP_LIB=$PWD/ab/libprev.so python tools/time_tile.py 200 2>&1 | tail -1 >> $out
  echo "== tree" >> $out; python tools/time_tile.py 200 2>&1 | tail -1 >> $out
done
cat gpurun_out/t10.log; cat $out
