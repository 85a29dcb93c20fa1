import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["repeats"]["value_min"], d["repeats"]["value_max"], "loop us", d["roofline"]["avg_us_rocprof_dispatch"], d["roofline"]["frac"])
print({k.split("(")[0][-40:]:(v["avg_us"],v["calls"]) for k,v in list(d["kernels"].items())[:7]})
print("single", d["single_batch"]["ms_per_batch"], "parity", d["parity"]["max_abs_joints_vs_exact_fp32_engine_all_requests"], d["parity"].get("max_abs_joints_vs_oracle"))
print("mix", d["length_mix"]["value"], d["length_mix"]["max_abs_joints_vs_oracle"])
