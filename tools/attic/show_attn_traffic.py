import json
for f in ("gpurun_out/r06x_pmc_traffic.json", "gpurun_out/r06_noxcd_pmc_traffic.json"):
    t = json.load(open(f))["kernels"]
    for k in t:
        if "attn_h" in k or "dq_reduce" in k: print(f.split("/")[-1], k, t[k])
