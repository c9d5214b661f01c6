"""Reads a FCGPU_TIMELINE dump (development aid): per-wave start/end of the last k_extend launch."""
import sys
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 24).astype(np.int64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start, end, rays, rounds = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2], t[:, 3]   # microseconds
span = end.max()
work = rays > 0
print(f"waves {len(t)} working {work.sum()}  span {span:.1f} us  rays {rays.sum()}")
print(f"working waves: start p50 {np.median(start[work]):.1f} max {start[work].max():.1f}; end p5 {np.percentile(end[work],5):.1f} "
      f"p50 {np.median(end[work]):.1f} p95 {np.percentile(end[work],95):.1f} max {end[work].max():.1f}; life mean {(end-start)[work].mean():.1f}")
print(f"idle waves: start p50 {np.median(start[~work]) if (~work).any() else 0:.1f}")
print(f"rays/wave mean {rays[work].mean():.0f} min {rays[work].min()} max {rays[work].max()}  rounds/wave mean {rounds[work].mean():.0f} max {rounds.max()}")
bins = np.linspace(0, span, 21)
occ = [((start[work] <= b) & (end[work] > b)).sum() for b in bins[:-1]]
print("resident working waves at 20 sample times:", occ)
nodes, nmax, prims, pmax = t[:, 4], t[:, 5], t[:, 6], t[:, 7]
life = (end - start)
w = work
print(f"per wave: node steps sum {nodes[w].mean():.0f} lane-max {nmax[w].mean():.0f}; prim tests sum {prims[w].mean():.0f} lane-max {pmax[w].mean():.0f}")
print(f"us per lane-max node step {(life[w] / np.maximum(nmax[w], 1)).mean():.3f}; lane efficiency nodes {(nodes[w] / (64.0 * np.maximum(nmax[w], 1))).mean():.2f}")
late = w & (end > np.percentile(end[w], 90))
early = w & (end < np.percentile(end[w], 10))
for name, m in (("latest 10%", late), ("earliest 10%", early)):
    print(f"{name}: rays {rays[m].mean():.0f} rounds {rounds[m].mean():.1f} nodes-sum {nodes[m].mean():.0f} nodes-lane-max {nmax[m].mean():.0f} life {life[m].mean():.1f}")
ph = t[:, 8:23].astype(float)
tot = ph[w][:, [0, 1, 4]].sum(1)
print("shader-clock cycles per wave: descent %.0f (%.0f%%)  leaf %.0f (%.0f%%)  finish+fetch %.0f (%.0f%%)" % (
    ph[w, 0].mean(), 100 * (ph[w, 0] / tot).mean(), ph[w, 1].mean(), 100 * (ph[w, 1] / tot).mean(), ph[w, 4].mean(), 100 * (ph[w, 4] / tot).mean()))
print("per wave: descent trips %.0f (%.0f cycles each)  leaf phases %.0f (%.0f cycles each)  finish phases %.0f (%.0f cycles each)" % (
    ph[w, 2].mean(), (ph[w, 0] / np.maximum(ph[w, 2], 1)).mean(), ph[w, 3].mean(), (ph[w, 1] / np.maximum(ph[w, 3], 1)).mean(),
    ph[w, 5].mean(), (ph[w, 4] / np.maximum(ph[w, 5], 1)).mean()))
print("cycles total / life us -> clock MHz: %.0f" % (tot / life[w]).mean())
print("lane efficiency: descent %.3f (node steps / 64 x trips)  leaf %.3f  finish %.3f (finished lanes / 64 x finish phases)" % (
    (nodes[w].sum() / (64.0 * ph[w, 2].sum())), ph[w, 6].sum() / (64.0 * np.maximum(ph[w, 3].sum(), 1)), ph[w, 7].sum() / (64.0 * np.maximum(ph[w, 5].sum(), 1))))
print("finish phase parts, cycles each: hit record %.0f  work_take %.0f  ray load + set-up %.0f" % tuple((ph[w, k] / np.maximum(ph[w, 5], 1)).mean() for k in (8, 9, 10)))
print("node steps that found nothing: %.3f of all steps; emptied only by the shrunken t_max: %.3f" % (
    ph[w, 11].sum() / nodes[w].sum(), ph[w, 12].sum() / nodes[w].sum()))
print("lanes per descent trip: descending %.1f  waiting with a leaf %.1f  finished / empty %.1f" % (
    nodes[w].sum() / ph[w, 2].sum(), ph[w, 13].sum() / ph[w, 2].sum(), ph[w, 14].sum() / ph[w, 2].sum()))
