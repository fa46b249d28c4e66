"""Print the headline numbers of a bench.py JSON line (helper for GPU-box runs)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fm = d.get("fused_mask", {}).get("roofline", {})
print("value", d["value"], "ms/step", d["ms_per_step"], "k_match frac", d["roofline"]["frac"],
      "| fused ms", fm.get("avg_launch_ms"), "frac", fm.get("frac"),
      "| cpu", d.get("cpu_baseline", {}).get("value"), "mism", d.get("cpu_baseline", {}).get("parity_mismatches_vs_gpu"))
