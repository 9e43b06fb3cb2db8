mkdir -p gpurun_out/r05b
python -m pytest tests/test_c_caller.py -q -m gpu 2>&1 | tail -3
python bench.py --legs all7,fdrp_pairs > gpurun_out/r05b/legs.json 2> gpurun_out/r05b/legs.err; tail -3 gpurun_out/r05b/legs.err
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r05b/legs.json") if l.startswith("{")][-1])
a=j["all7"]
print("per pass", a["per_pass_ms_one_sync_each"], "seven", a["seven_measures_ms"])
print("prepared", a["prepared_batches"])
print("kernels", json.dumps(a["kernels_ms_per_pass"]))
print("roofline", json.dumps(a["roofline_per_pass"]))
print("fdrp_pairs", json.dumps(j["fdrp_pairs"])[:1500])
PY
