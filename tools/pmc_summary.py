"""Reduces the rocprofv3 PMC passes of tools/pmc_passes.sh (OUTDIR/{sq,ta_tcp,tcc_a,...}/**/counter_collection.csv) to the
per-kernel numbers bench.py prints beside its own timing: VALU busy fraction, lane utilisation, TA busy fraction, fabric
bytes per ray. usage: pmc_summary.py OUTDIR SOURCE_LABEL BATCH_PATHS > ray-tracing_amd/pmc_summary.json
Formulas: kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; VALU busy = SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs
/ kernel cycles; lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); TA busy = TA_TA_BUSY_sum / 256 CUs
/ kernel cycles; fabric bytes = TCC_EA0_RDREQ * 128 (all requests are 128-byte on this part: RDREQ_32B is 0) +
TCC_EA0_WRREQ_64B * 64 + (TCC_EA0_WRREQ - WRREQ_64B) * 32."""
import collections, csv, glob, importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

out, source, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
# one rocprofv3 run per counter set: keep the sets apart (GRBM_GUI_ACTIVE is collected in two of them)
passes = {}
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    name = f[len(out):].strip("/").split("/")[0]
    agg = passes.setdefault(name, collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void fcg::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["_dispatches_" + r["Counter_Name"]] += 1.0
rays = json.load(open(f"{out}/rays.json")) if glob.glob(f"{out}/rays.json") else {}
# the instantiations the bench workload runs: no profiling, no spill, no textures, 4-wide hierarchy; the fp32 shadow rays run
# k_shadow_dense (22 LDS stack entries; its SPILL instantiation on the room, whose worst case is 28)
names = {"k_extend<float, false, false, false, false>": ("k_extend", "closest_rays"), "k_shadow_dense<true>": ("k_shadow", "shadow_rays"),
         "k_shade<float, 0, false>": ("k_shade_lambert", "closest_rays")}
kern = {}
for k, (short, unit) in names.items():
    e = {}
    sq, ta, tcc = passes.get("sq", {}).get(k), passes.get("ta_tcp", {}).get(k), passes.get("tcc_a", {}).get(k)
    if sq and sq.get("GRBM_GUI_ACTIVE"):
        cyc = sq["GRBM_GUI_ACTIVE"] / 8.0
        e["valu_busy"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
        e["lane_util"] = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
        e["wait_frac"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        e["launches"] = int(sq["_dispatches_GRBM_GUI_ACTIVE"])
    if ta and ta.get("GRBM_GUI_ACTIVE"):
        e["ta_busy"] = ta["TA_TA_BUSY_sum"] / 256.0 / (ta["GRBM_GUI_ACTIVE"] / 8.0)
    if tcc:
        b = tcc["TCC_EA0_RDREQ_sum"] * 128.0 + tcc["TCC_EA0_WRREQ_64B_sum"] * 64.0 + (tcc["TCC_EA0_WRREQ_sum"] - tcc["TCC_EA0_WRREQ_64B_sum"]) * 32.0
        e["fabric_bytes"] = b
        if rays.get(unit):
            e["fabric_bytes_per_ray"] = b / rays[unit]
    if e:
        kern[short] = e
missing = [short for short, _ in names.values() if short not in kern]
if missing:
    sys.exit(f"pmc_summary: no counters for {missing}: kernel names in the passes: {sorted(passes.get('sq', {}))}")
# the counters describe the kernels of THIS tree: bench.py refuses to print them beside the timing of another build
build = importlib.import_module("ray-tracing_amd.buildinfo").kernel_source_hash()
print(json.dumps({"source": source, "build": build, "batch_paths": batch, "rays": rays, "kernels": kern}, indent=1))
