mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pdr_lpmd.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_mhl.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t9.log
out=gpurun_out/ab9.log; : > $out
for rnd in 1 2 3; do
  echo "== prev" >> $out; METHEOR_HIP_LIB=$PWD/ab/libprev.so python tools/time_tile.py 200 2>&1 | tail -1 >> $out
  echo "== tree" >> $out; python tools/time_tile.py 200 2>&1 | tail -1 >> $out
done
python bench.py --no-cpu-baseline --no-e2e > gpurun_out/bench9.json 2> gpurun_out/bench9.err
cat gpurun_out/t9.log; cat $out; cut -c1-200 gpurun_out/bench9.json
