import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["stage_ms_per_step"].items()})
