"""Summarise rocprofv3 csv outputs (kernel stats / pmc) under a directory. Development tool."""
import collections, csv, glob, sys
d = sys.argv[1]
for f in glob.glob(f"{d}/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"].replace("void fcg::", "")[:48].ljust(48), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:10.1f} us', f'{float(r["Percentage"]):6.2f} %')
for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void fcg::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        extra = ""
        if "SQ_THREAD_CYCLES_VALU" in v and "SQ_ACTIVE_INST_VALU" in v and v["SQ_ACTIVE_INST_VALU"]:
            extra = f' lane_util={v["SQ_THREAD_CYCLES_VALU"]/(64*v["SQ_ACTIVE_INST_VALU"]):.2f}'
        if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
            extra += f' wait_frac={v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"]:.2f}'
        print(k[:40].ljust(40), {a: f"{b:.3g}" for a, b in v.items()}, extra)
