import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("map_state"), {k:v for k,v in d["roofline"]["kernels_ms_per_step"].items() if v>0.15})
    except Exception as e: print(f, "FAILED", e)
