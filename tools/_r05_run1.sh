mkdir -p gpurun_out/r05b
timeout 1500 python -m pytest tests/test_gpu_fdrp.py -x -q -m gpu 2>&1 | tail -4
for l in tree tree; do
timeout 300 python bench.py --legs fdrp_pairs 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['fdrp_pairs']
print('$l config4 pass_ms', j['pass_ms'], j['kernels_ms'])"
done
