#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel (short name) and counter.
usage: pmc_sum.py DIR [DIR...]   (prints a JSON object {kernel: {counter: sum, "_launches": n}})"""
import csv, glob, json, os, re, sys
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if not k.startswith("k_"):
                continue
            o = out.setdefault(k, {})
            o[row["Counter_Name"]] = o.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"], row["Counter_Name"])
            if (k, row["Dispatch_Id"]) not in seen:
                seen.add((k, row["Dispatch_Id"]))
for k in out:
    out[k] = {c: out[k][c] for c in sorted(out[k])}
print(json.dumps(out, indent=1))
