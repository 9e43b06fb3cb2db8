mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t_final.log
cat gpurun_out/t_final.log
timeout 600 python tools/bench_wgbs.py > gpurun_out/wgbs_v8.jsonl 2> gpurun_out/wgbs_v8.err; tail -1 gpurun_out/wgbs_v8.jsonl
timeout 600 python tools/bench_measures.py > gpurun_out/measures_v7.jsonl 2> gpurun_out/measures_v7.err; grep -c . gpurun_out/measures_v7.jsonl
