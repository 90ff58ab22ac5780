#!/usr/bin/env python3
"""CPU side of SURVEY.md 8(d), the flags question: the oracle (the reference's pointer-BVH path tracer restated in C) timed on this
box's host cores as bench.py's cpu_baseline times it (-O2 -ffp-contract=off, the build whose arithmetic the parity tests pin), and
again compiled the way upstream ships (/fp:fast /arch:AVX2 ~ -O3 -march=native -ffast-math), with 12 threads (upstream's threadNum)
and with this box's share.  The fast build is compiled here, on the box, into a temp dir; it is a number to put beside the GPU's,
not a checker (fast-math changes results).
    python tools/bench_cpu_flags.py [spp]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "oracle"))
import orc
from raytracer_amd import scenes
s = scenes.interior("bathroom", 1.0, 1920, 1080, seed=1)
o = orc.OracleScene(s)
spp, threads = int(sys.argv[2]), int(sys.argv[3])
t = time.perf_counter(); _, st = o.render(0, spp, seed=1, first_sample=0, threads=threads); dt = time.perf_counter() - t
print(json.dumps({"mrays": (st["closest_rays"] + st["any_rays"]) / dt / 1e6, "s_per_spp": dt / spp}))
"""


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    share = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    tmp = tempfile.mkdtemp()
    fast = os.path.join(tmp, "liboracle_fast.so")
    subprocess.run(["gcc", "-std=c11", "-O3", "-march=native", "-ffast-math", "-fPIC", "-shared", "-w",
                    os.path.join(ROOT, "oracle", "rt_oracle.c"), "-o", fast, "-lm", "-lpthread"], check=True)
    for label, so in (("-O2 -ffp-contract=off", ""), ("-O3 -march=native -ffast-math", fast)):
        for threads in sorted({12, share}):
            env = dict(os.environ, ORC_SO=so)
            r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(spp), str(threads)], env=env, capture_output=True, text=True, check=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(f"{label:32s} threads {threads:3d}: {d['mrays']:.2f} Mrays/s, {d['s_per_spp']:.2f} s/spp", flush=True)


if __name__ == "__main__":
    main()
