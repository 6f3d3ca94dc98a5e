"""Print the headline fields of a bench.py run read from stdin (its LAST line is the compact JSON line; a `[bench-detail] ` line precedes it)."""
import json, sys
lines = [l for l in sys.stdin.read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"], 1), "q/s", round(d["ms_per_step"], 2), "ms/step",
      {k: round(v, 2) for k, v in d["stage_ms_per_step_rank0"].items()})
