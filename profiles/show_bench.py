import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d.get("hbm_kernels",{}).items(): print("fp32", k, v)
c5=d.get("c5_bf16")
if c5:
    print(c5["value"], c5["ms_per_step"], c5["roofline"]["frac"])
    for k,v in c5.get("hbm_kernels",{}).items(): print("bf16", k, v)
print(d.get("c2_64cube_b2"))
