"""fill the @@PLACEHOLDERS@@ of DESIGN.md from a bench line: python tools/fill_design.py profiles/r05_v6_bench.json"""
import json, re, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = j["roofline"]; ak = r.get("all_kernels_ms", {})
a7 = j.get("all7", {}); pb = a7.get("prepared_batches", {}); e = j.get("e2e", {})
rep = {
    "IDX2": "%.4f" % ak.get("k_build_index", float("nan")), "TILE": "%.4f" % r["kernel_ms"], "GATHER": "%.4f" % ak.get("k_gather", float("nan")),
    "FRAC2": "%.3f" % r["frac"], "FRACL3": "%.3f" % r.get("frac_l3_resident", float("nan")),
    "WIDE": "%.4f" % j.get("roofline_wgbs", {}).get("kernel_ms", float("nan")), "WIDEFRAC": "%.3f" % j.get("roofline_wgbs", {}).get("frac", float("nan")),
    "VALUE": "%.1f" % (j["value"] / 1e3 if j["value"] > 1e4 else j["value"]), "STEP": "%.4f" % j["ms_per_step"],
    "TRAFFIC": "%.1f" % ((r.get("traffic") or 0) / 1e6), "WSF": "%.3f" % (r.get("whole_step_frac") or float("nan")),
    "ALL7": "%.2f" % a7.get("seven_measures_ms", float("nan")), "ALL7P": "%.2f" % pb.get("seven_measures_ms", float("nan")), "PREP": "%.2f" % pb.get("prepare_ms_once", float("nan")),
    "FDRP4": "%.2f" % j.get("fdrp_pairs", {}).get("pass_ms", float("nan")),
    "E2E": "%.1f" % e.get("M_reads_per_s_median", float("nan")), "E2EL": "%.1f" % e.get("large", {}).get("M_reads_per_s_median", float("nan")),
    "FLOOR": "%.1f M reads/s (copy %.1f GB/s)" % (e.get("floor", {}).get("floor_M_reads_per_s", float("nan")), e.get("floor", {}).get("pageable_copy_GBps", float("nan"))),
    "CPU": "%.1f" % j.get("cpu_baseline", {}).get("value", float("nan")),
}
s = open("DESIGN.md").read()
for k, v in rep.items():
    s = s.replace("@@%s@@" % k, v)
left = re.findall(r"@@[A-Z0-9]+@@", s)
open("DESIGN.md", "w").write(s)
print("filled", len(rep), "left", left)
