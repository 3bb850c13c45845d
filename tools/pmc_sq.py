"""Per-kernel SQ instruction / wait counters from a rocprofv3 --pmc pass
(`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`).

    python tools/pmc_sq.py DIR out.json
"""
import csv, glob, json, sys
from pmc_traffic import base_name

tot, dur, cnt = {}, {}, {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = base_name(r["Kernel_Name"])
        tot.setdefault(k, {})
        tot[k][r["Counter_Name"]] = tot[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d = r.get("Dispatch_Id")
        if (k, d) not in seen:
            seen.add((k, d))
            cnt[k] = cnt.get(k, 0) + 1
            if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                dur[k] = dur.get(k, 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {}
for k, c in tot.items():
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    e = dict(launches=cnt.get(k, 0), **{n: int(v) for n, v in c.items()})
    if wc:
        e["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 3)
        e["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
        e["active_inst_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    out[k] = e
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    print(f"{k:26s} valu {e.get('SQ_INSTS_VALU',0)/1e6:9.1f}M salu {e.get('SQ_INSTS_SALU',0)/1e6:9.1f}M lds {e.get('SQ_INSTS_LDS',0)/1e6:8.1f}M "
          f"vmem {e.get('SQ_INSTS_VMEM',0)/1e6:8.1f}M  wait_any {e.get('wait_any_frac')} wait_inst {e.get('wait_inst_frac')} active {e.get('active_inst_frac')}")
