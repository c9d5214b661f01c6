"""Development tool: VALU / VMEM / LDS instruction counts of the node-step loop (the innermost loop that holds the
v_cvt_f32_ubyte decodes) of kernels in a gfx950 assembly file (hipcc -save-temps). usage: isa_loop_count.py FILE.s SUBSTR..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
names = re.findall(r'^(_ZN3fcg\S+):', s, flags=re.M)
for name in names:
    if not any(k in name for k in sys.argv[2:]):
        continue
    a = s.index('\n' + name + ':'); b = s.index('.Lfunc_end', a)
    L = [l.strip() for l in s[a:b].splitlines()]
    labels = {}
    for k, l in enumerate(L):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: labels[m.group(1)] = k
    idx = [i for i, l in enumerate(L) if 'v_cvt_f32_ubyte' in l]
    if not idx: continue
    first, last = idx[0], idx[-1]
    best = None
    for j, l in enumerate(L):
        m = re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels:
            t = labels[m.group(1)]
            if t <= first and j >= last and (best is None or (j - t) < (best[1] - best[0])): best = (t, j)
    if best is None:
        print(name[:60], "no enclosing loop found"); continue
    region = L[best[0]:best[1] + 1]
    ins = [l for l in region if l and not l.startswith(('.', ';')) and not re.match(r'^\S+:', l)]
    valu = [l for l in ins if l.startswith('v_')]
    c = Counter(l.split()[0] for l in valu)
    print(name[8:70], '| loop instr', len(ins), 'VALU', len(valu), 'SALU', sum(l.startswith('s_') for l in ins), 'VMEM', sum(l.startswith(('global_', 'buffer_', 'flat_')) for l in ins), 'LDS', sum(l.startswith('ds_') for l in ins))
    print('    ', c.most_common(14))
