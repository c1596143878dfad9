import json,sys
d=json.load(open(sys.argv[1]))
r=d.pop("roofline"); pk=r.pop("per_kernel")
print(d["value"], d["ms_per_step"], d.get("cpu_baseline"), d.get("speedup_vs_cpu_baseline")); print(r)
for k,v in pk.items(): print(k, v)
