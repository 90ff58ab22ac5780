#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (short name) and counter, and derive the per-launch
figures bench.py / DESIGN.md quote for the dominant kernel.
usage: pmc_sum.py DIR [DIR...]   (prints JSON: {"counters": {kernel: {counter: {"sum", "dispatches"}}}, derived...})"""
import csv, glob, json, os, re, sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if not k.startswith("k_"):
                continue
            o = out.setdefault(k, {}).setdefault(row["Counter_Name"], {"sum": 0.0, "ids": set()})
            o["sum"] += float(row["Counter_Value"])
            o["ids"].add(row["Dispatch_Id"])
res = {"command": "rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --warmup 0 (one run per counter group, tools/pmc.sh)",
       "counters": {k: {c: {"sum": v["sum"], "dispatches": len(v["ids"])} for c, v in sorted(cs.items())} for k, cs in sorted(out.items())}}


def derive(kname, tag):
    c = res["counters"].get(kname)
    if not c:
        return
    g = lambda n: c[n]["sum"] if n in c else None
    n = lambda name: max(c[name]["dispatches"], 1)
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        fetch = g("FETCH_SIZE") * 1024.0 / n("FETCH_SIZE")      # rocprofv3 reports KB
        write = g("WRITE_SIZE") * 1024.0 / n("WRITE_SIZE")
        res[tag + "_fetch_bytes_per_launch_raw"] = fetch
        res[tag + "_fetch_bytes_per_launch_x2"] = 2 * fetch       # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
        res[tag + "_write_bytes_per_launch"] = write
        res[tag + "_hbm_bytes_per_launch"] = 2 * fetch + write
    if g("TCC_HIT_sum") is not None:
        res[tag + "_l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
    if g("GRBM_GUI_ACTIVE") is not None and g("SQ_ACTIVE_INST_VALU") is not None:
        cyc = g("GRBM_GUI_ACTIVE") / 8.0                          # summed over the 8 XCDs
        res[tag + "_gpu_cycles"] = cyc
        res[tag + "_valu_busy_frac"] = g("SQ_ACTIVE_INST_VALU") * 4.0 / 1024.0 / cyc   # 4-cycle issue slots over 1024 SIMDs
        res[tag + "_salu_busy_frac"] = g("SQ_ACTIVE_INST_SCA") * 4.0 / 1024.0 / cyc
        if g("TA_TA_BUSY_sum") is not None:
            res[tag + "_ta_busy_frac"] = g("TA_TA_BUSY_sum") / 256.0 / cyc
    if g("SQ_THREAD_CYCLES_VALU") is not None:
        res[tag + "_live_lanes_per_valu_inst"] = g("SQ_THREAD_CYCLES_VALU") / max(g("SQ_INSTS_VALU"), 1)
    if g("SQ_WAIT_INST_ANY") is not None:
        res[tag + "_wait_inst_frac_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)


derive("k_extend<false, false>", "k_extend")
derive("k_connect<false>", "k_connect")
derive("k_shade<false>", "k_shade")
print(json.dumps(res, indent=1))
