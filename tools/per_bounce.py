"""Development tool: per-dispatch kernel durations of one pass, in launch order, from a rocprofv3 --kernel-trace csv
directory (kernel_trace.csv): how k_extend / k_shadow / k_shade time moves with the bounce. usage: per_bounce.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void fcg::", "").split("(")[0]))
rows.sort()
seq = {}
for s, e, n in rows:
    key = n[:34]
    seq.setdefault(key, []).append((e - s) / 1e3)
for k, v in seq.items():
    if len(v) >= 5 and max(v) > 200:
        tail = v[-11:] if len(v) > 11 else v
        print(k.ljust(36), len(v), "last launches (us):", " ".join(f"{x:8.0f}" for x in tail))
