"""Print the few figures of a bench.py JSON line (stdin) that kernel experiments compare."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(" ".join(sys.argv[1:]), d["config"]["workload"].split(":")[0], d["value"], "Mrays/s  trace", r["trace_ms_per_step"],
      "shade", r["shade_ms_per_step"], "ms/step", d["ms_per_step"])
