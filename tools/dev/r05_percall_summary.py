import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pc=d["per_call"]; print(sys.argv[1], "cold/warm", pc["cold"]["overhead_frac"], pc["warm"]["overhead_frac"], "ragged", pc["ragged_cfg3"]["cold"]["overhead_frac"], pc["ragged_cfg3"]["warm"]["overhead_frac"])
for nm in ("cold","warm"):
    for c in pc[nm]["calls"]:
        big={k:v for k,v in c["phases_ms"].items() if v>12}
        if big: print("  ", nm, c["L"], big)
