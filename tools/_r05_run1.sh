mkdir -p gpurun_out/r05a
{
for l in m0 m1 m2 m2o8 m0 m1 m2; do w=5; [ $l = m2o8 ] && w=8; echo "$l $(MTH_RUNS_WGS_PER_CU=$w METHEOR_HIP_LIB=$PWD/abx/lib$l.so timeout 300 python tools/time_tile.py 200 2>&1 | tail -1)"; done
} > gpurun_out/r05a/time3.log 2>&1
cat gpurun_out/r05a/time3.log
