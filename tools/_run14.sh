mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t14.log
bash tools/profile_round.sh r02_v6 873802c > gpurun_out/profile_round_v6.log 2>&1
timeout 300 python tools/bench_measures.py > gpurun_out/r02_v6_measures.jsonl 2> gpurun_out/measures_g.err
tail -2 gpurun_out/smoke.log; cat gpurun_out/t14.log; tail -12 gpurun_out/profile_round_v6.log | cut -c1-300
