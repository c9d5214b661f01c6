"""Development tool: mean duration of the FULL-SIZE launches of each traversal / shade kernel in a rocprofv3 --kernel-trace
directory of `python3 bench.py` (the stats file's average also counts the small launches of the parity gate and the f64
leg). A launch counts as full-size when it lasts at least half as long as the kernel's 90th-percentile launch (not its longest:
one straggler would move the bar)."""
import csv, glob, sys
rows = {}
for f in glob.glob(f"{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void fcg::", "").split("(")[0]
        rows.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    ref = sorted(v)[min(len(v) - 1, int(0.9 * len(v)))]
    big = [x for x in v if x >= 0.5 * ref]
    if sum(v) > 5e3:
        print(f"{n[:44]:44s} launches {len(v):5d} total {sum(v) / 1e3:9.1f} ms | full-size launches {len(big):5d} mean {sum(big) / len(big):9.1f} us")
