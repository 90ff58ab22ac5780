# full-size (bathroom-class 1080p) same-seed comparison of the other render modes with the CPU oracle
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import orc
from raytracer_amd import api, scenes
s = scenes.bathroom(1.0, 1920, 1080)
o = orc.OracleScene(s)
sc = api.Scene(s, device_build=True)
def cmp(name, want, got, st, cnt):
    fw, fg = np.isfinite(want).all(axis=2), np.isfinite(got).all(axis=2)
    print(name, "nonfinite px oracle", (~fw).sum(), "gpu", (~fg).sum(), "xor", (fw ^ fg).sum())
    both = fw & fg
    a, b = got[both].astype(np.float64), want[both].astype(np.float64)
    close = (np.abs(a - b) <= 1e-3 * np.maximum(np.abs(a), np.abs(b)) + 1e-4).all(axis=1)
    print("  close frac", close.mean(), "not close", (~close).sum())
    ok = both & (np.abs(want).max(axis=2) < 1e3) & (np.abs(got).max(axis=2) < 1e3)
    d, w = got[ok].astype(np.float64) - want[ok].astype(np.float64), want[ok].astype(np.float64)
    print("  global rel", (d * d).mean() / (w * w).mean(), "sum ratio", got[ok].astype(np.float64).sum() / w.sum())
    if st: print("  rays", cnt["closest_rays"], st["closest_rays"], cnt["any_rays"], st["any_rays"], flush=True)
t = time.time(); want, st = o.render(1, 1, seed=3, threads=16); print("oracle light s", time.time() - t, flush=True)
rt = api.RayTracer(sc); rt.render(api.LIGHT, 1, 3, 0); got, _ = rt.film(); cnt = rt.counters(); rt.close()
cmp("LIGHT", want, got, st, cnt)
n = o.vpl_trace(2, seed=9)
t = time.time(); want, st = o.render(4, 1, seed=9, threads=16); print("oracle radiosity s", time.time() - t, "vpls", n, flush=True)
rt = api.RayTracer(sc); n2 = rt.traceVPLs(2, 9); rt.render(api.RADIOSITY, 1, 9, 0); got, _ = rt.film(); cnt = rt.counters(); rt.close()
print("vpls gpu", n2)
cmp("RADIOSITY", want, got, st, cnt)
