"""Throughput of the other integrators on the bench scene (BASELINE config 5): light tracing, instant radiosity, PSSMLT.
Usage: python tools/bench_modes.py [--detail 1.0]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--detail", type=float, default=1.0)
a = ap.parse_args()
s = scenes.interior("bathroom", a.detail, 1920, 1080)
sc = api.Scene(s, device_build=True)


def run(name, mode, spp, vpls=0, filt=0):
    rt = api.RayTracer(sc)
    if filt:
        rt.setFilter(filt)
    if vpls:
        rt.traceVPLs(vpls, 1)
    rt.render(mode, max(1, spp // 4), 1, 0); rt.sync(); rt.clear(); rt.counters(reset=True)
    t = time.perf_counter()
    rt.render(mode, spp, 1, 100); rt.sync()
    dt = time.perf_counter() - t
    c = rt.counters()
    rays = c["closest_rays"] + c["any_rays"]
    print(f"{name:34s} {spp:3d} spp  {dt*1e3/spp:9.2f} ms/spp  {rays/dt/1e6:9.1f} Mrays/s  ({rays/spp/1e6:.1f} M rays/spp)", flush=True)
    rt.close()


run("path tracing (type 0)", api.PATH, 32)
run("light tracing (type 1)", api.LIGHT, 32)
run("instant radiosity, 50 VPLs (type 4)", api.RADIOSITY, 2, vpls=50)
run("instant radiosity, 1024 VPLs", api.RADIOSITY, 1, vpls=1024)
run("PSSMLT (type 2), 100k proposals/spp", api.MLT, 64)
run("path tracing, Gaussian film filter", api.PATH, 32, filt=api.FILTER_GAUSSIAN)
run("path tracing, Mitchell film filter", api.PATH, 32, filt=api.FILTER_MITCHELL)
run("light tracing, Mitchell film filter", api.LIGHT, 32, filt=api.FILTER_MITCHELL)
sc.close()
