import json,sys
d=json.load(open(sys.argv[1]));print(sys.argv[1], d["value"],d["stage_ms_per_step"][sys.argv[2]])
