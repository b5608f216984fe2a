#!/usr/bin/env python3
"""One bench.py JSON line on stdin -> workload, frames/s, ms per step and the per-kernel launch times."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
kr = d.get("kernel_rooflines", {})
print(d["config"]["workload"], d["value"], d["ms_per_step"],
      {k: round(v["ms"], 3) for k, v in kr.items() if isinstance(v, dict) and "ms" in v})
