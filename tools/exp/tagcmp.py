import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernels"]
tags=["conv1x1_fwd~hbm","conv1x1_fwd","conv1x1_dgrad~hbm","conv1x1_dgrad","conv1x1_dgrad_add_x6~hbm","conv1x1_dgrad_add_x6","conv_s2_dgrad","conv_s2_fwd"]
print(sys.argv[1], d["ms_per_step"], " ".join(f'{t.split("_",1)[1]}={k[t]["avg_us"]:.0f}' for t in tags if t in k))
