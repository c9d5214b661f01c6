"""Development tool: registers, scratch, LDS and occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage output
(tools/build_variant.sh writes it to build/var/resources_NAME.txt). usage: kernel_resources.py FILE [SUBSTR...]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []
for m in re.finditer(r"remark: [^\n]*Function Name: (\S+)(.*?)LDS Size \[bytes/block\]: (\d+)", txt, flags=re.S):
    body = m.group(2)
    g = lambda k: re.search(k + r": (\d+)", body).group(1)
    rows.append((m.group(1), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), m.group(3)))
if not rows: sys.exit("no kernel-resource-usage remarks in " + sys.argv[1])
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
seen = set()
for r, n in zip(rows, names):
    n = n.split("(")[0].replace("void fcg::", "")
    if n in seen or (sys.argv[2:] and not any(k in n for k in sys.argv[2:])): continue
    seen.add(n)
    print(f"{n:50s} VGPRs {r[1]:>4s} AGPRs {r[2]:>3s} scratch {r[3]:>4s} occupancy {r[4]} LDS {r[5]}")
