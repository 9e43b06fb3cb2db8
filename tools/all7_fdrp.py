"""FDRP + qFDRP pass of `bench.py --legs all7` only: prints the pass's ms and kernels (tuning aid)"""
import json, subprocess, sys, os
r = subprocess.run([sys.executable, "bench.py", "--legs", "all7"], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.loads(r.stdout.strip().splitlines()[-1])
a = d.get("all7", d)
print(json.dumps({"seven": a["seven_measures_ms"], "per_pass": a["per_pass_ms_one_sync_each"], "fdrp_kernels": a["kernels_ms_per_pass"]["fdrp+qfdrp"], "mhl_kernels": a["kernels_ms_per_pass"]["mhl"]}))
