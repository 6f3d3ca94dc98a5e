"""Print the headline fields of a bench.py JSON line read from stdin."""
import json, sys
d = json.loads(sys.stdin.read())
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"], 1), "q/s", round(d["ms_per_step"], 2), "ms/step",
      {k: round(v, 2) for k, v in d["stage_ms_per_step_rank0"].items()})
