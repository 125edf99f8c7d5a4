import json, sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
print({k:(v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("error")) for k,v in d["other_configs"].items()})
print("acq", d["acquisition_step"].get("ms_per_step"), d["parity"])
print(d.get("phases_ms_per_step"))
