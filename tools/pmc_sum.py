#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (short name) and counter, and derive the per-launch
figures bench.py / DESIGN.md quote for the traversal and shading kernels.
usage: pmc_sum.py --bench-line FILE DIR [DIR...]
  FILE: stdout of one of the profiled bench.py runs (its JSON line names the batch shape the counters belong to)
prints JSON: {"shape", "counters": {kernel: {counter: {"sum", "dispatches"}}}, derived per-kernel figures}"""
import csv, glob, json, os, re, sys

args = sys.argv[1:]
shape = None
if args and args[0] == "--bench-line":
    for line in open(args[1]):
        if line.startswith("{"):
            shape = json.loads(line)["roofline"].get("shape")
    args = args[2:]
out, dur = {}, {}
for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if not k.startswith("k_"):
                continue
            o = out.setdefault(k, {}).setdefault(row["Counter_Name"], {"sum": 0.0, "ids": set()})
            o["sum"] += float(row["Counter_Value"])
            o["ids"].add(row["Dispatch_Id"])
            dur.setdefault(k, {})[(d, row["Dispatch_Id"])] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
res = {"command": "rocprofv3 --pmc <group> -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --no-unshared-pass "
                  "(one run per counter group, tools/pmc.sh; kernels run serialised under the profiler)",
       "shape": shape,
       "counters": {k: {c: {"sum": v["sum"], "dispatches": len(v["ids"])} for c, v in sorted(cs.items())} for k, cs in sorted(out.items())}}


def derive(kname, tag):
    c = res["counters"].get(kname)
    if not c:
        return
    g = lambda n: c[n]["sum"] if n in c else None
    n = lambda name: max(c[name]["dispatches"], 1)
    secs = sum(dur[kname].values()) / max(len(dur[kname]), 1)      # mean launch duration over every profiled dispatch of the kernel
    res[tag + "_launch_seconds"] = secs
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        fetch = g("FETCH_SIZE") * 1024.0 / n("FETCH_SIZE")      # rocprofv3 reports KB
        write = g("WRITE_SIZE") * 1024.0 / n("WRITE_SIZE")
        res[tag + "_fetch_bytes_per_launch_raw"] = fetch
        res[tag + "_fetch_bytes_per_launch_x2"] = 2 * fetch       # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md)
        res[tag + "_write_bytes_per_launch"] = write
        res[tag + "_hbm_bytes_per_launch"] = 2 * fetch + write    # fabric-side (L2-miss) bytes: Infinity-Cache hits included, an upper bound on HBM
        res[tag + "_fabric_frac_of_hbm_peak"] = (2 * fetch + write) / secs / 8e12
    if g("TCC_HIT_sum") is not None:
        res[tag + "_l2_hit_rate"] = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
    if g("TCC_REQ_sum") is not None:
        res[tag + "_l2_requests_per_launch"] = g("TCC_REQ_sum") / n("TCC_REQ_sum")
        res[tag + "_l2_frac_of_peak"] = g("TCC_REQ_sum") / n("TCC_REQ_sum") * 128.0 / secs / 34.5e12
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
    cand = {"VALU issue": res.get(tag + "_valu_busy_frac"), "texture-address / L1 (TA busy)": res.get(tag + "_ta_busy_frac"),
            "L2 bandwidth": res.get(tag + "_l2_frac_of_peak"), "fabric / HBM bandwidth": res.get(tag + "_fabric_frac_of_hbm_peak")}
    cand = {k: v for k, v in cand.items() if v is not None}
    if cand:
        top = max(cand, key=cand.get)
        res[tag + "_limiter"] = top + " (" + ", ".join(f"{k} {v:.2f}" for k, v in sorted(cand.items(), key=lambda kv: -kv[1])) + \
            (f"; {res[tag + '_live_lanes_per_valu_inst']:.1f} of 64 lanes live per VALU instruction" if tag + "_live_lanes_per_valu_inst" in res else "") + ")"


derive("k_extend<false, false, 0>", "k_extend")
derive("k_connect<false>", "k_connect")
derive("k_shade<false, false>", "k_shade")
derive("k_bin", "k_bin")
print(json.dumps(res, indent=1))
