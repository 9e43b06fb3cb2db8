mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fdrp.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t15.log
out=gpurun_out/ab15.log; : > $out
for rnd in 1 2; do
  echo "== prev" >> $out; METHEOR_HIP_LIB=$PWD/ab/libprev.so python tools/time_sparse.py 2>&1 | grep fdrp | cut -c1-300 >> $out
  echo "== tree" >> $out; python tools/time_sparse.py 2>&1 | grep fdrp | cut -c1-300 >> $out
  echo "== prev c2" >> $out; METHEOR_HIP_LIB=$PWD/ab/libprev.so python tools/time_fdrp.py 10 2>&1 | tail -1 >> $out
  echo "== tree c2" >> $out; python tools/time_fdrp.py 10 2>&1 | tail -1 >> $out
done
cat gpurun_out/t15.log; cat $out
