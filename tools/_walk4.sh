mkdir -p gpurun_out
for w in 0 16 32; do echo "WALK4=$w"; METHEOR_FDRP_WALK4=$w timeout 300 python tools/time_sparse.py 2>/dev/null | grep fdrp; done > gpurun_out/walk4_sparse2.log 2>&1
for w in 0 16 32; do echo "WALK4=$w 20x"; METHEOR_FDRP_WALK4=$w timeout 300 python tools/time_sparse.py --reads 32000000 2>/dev/null | grep fdrp; done >> gpurun_out/walk4_sparse2.log 2>&1
cat gpurun_out/walk4_sparse2.log
timeout 900 python -m pytest tests/test_gpu_fdrp.py -x -q 2>&1 | tail -4 > gpurun_out/t_walk4.log
cat gpurun_out/t_walk4.log
