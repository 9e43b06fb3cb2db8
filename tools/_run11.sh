mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t11.log
out=gpurun_out/ab11.log; : > $out
for rnd in 1 2 3; do
  echo "== prev" >> $out; METHEOR_HIP_LIB=$PWD/ab/libprev.so python tools/time_tile.py 200 2>&1 | tail -1 >> $out
  echo "== tree" >> $out; python tools/time_tile.py 200 2>&1 | tail -1 >> $out
done
cat gpurun_out/t11.log; cat $out
