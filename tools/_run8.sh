bash tools/profile_round.sh r02_v5 59c075e > gpurun_out/profile_round_v5.log 2>&1
timeout 300 python tools/bench_measures.py > gpurun_out/r02_v5_measures.jsonl 2> gpurun_out/measures_e.err
timeout 500 python tools/bench_wgbs.py > gpurun_out/r02_v5_wgbs_200M.jsonl 2> gpurun_out/wgbs_e.err
tail -15 gpurun_out/profile_round_v5.log
