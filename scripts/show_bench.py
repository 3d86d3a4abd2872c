import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[1], "%.3e"%d["value"],round(d["ms_per_step"],3),{k:round(v,3) for k,v in d["kernel_ms"].items()},round(d["host_issue_ms_per_step"],3))
