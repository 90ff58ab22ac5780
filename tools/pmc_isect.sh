#!/bin/bash
# SQ instruction counters of the traversal microbench, one rocprofv3 pass per library build: tools/pmc_isect.sh lib...
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so); out=$GRAFT_REPO_ROOT/gpurun_out/pmci/$tag; mkdir -p $out
  RT_LIB=$GRAFT_REPO_ROOT/$lib timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_BRANCH -d $out -o vm --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_intersect.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  python3 - $out $tag <<'PY'
import csv, glob, os, re, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        if not k.startswith("k_intersect"): continue
        a = acc.setdefault(k, {})
        a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        a.setdefault("_ids", set()).add(row["Dispatch_Id"])
        a.setdefault("_ns", {})[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for k, a in sorted(acc.items()):
    n = len(a["_ids"]); ms = sum(a["_ns"].values()) * 1e-6
    print(sys.argv[2], k, "launches", n, "ms %.1f" % ms, " ".join("%s %.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(a.items()) if not c.startswith("_")),
          "valu-cycles/SIMD/elapsed %.2f" % (a["SQ_INSTS_VALU"] * 4 / 1024 / (ms * 1e-3 * 2.4e9)))
PY
done
