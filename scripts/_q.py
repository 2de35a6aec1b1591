import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], d["ms_per_step"], d["stages_ms_per_step"]["matching"], {k:v["ms_per_step"] for k,v in d["kernels"].items() if k.startswith("match")})
