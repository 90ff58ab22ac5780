import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import orc
from raytracer_amd import api, scenes
s = scenes.bathroom(1.0, 1920, 1080)
t = time.time(); want, st = orc.OracleScene(s).render(0, 1, seed=3, threads=16); print("oracle s", time.time() - t, flush=True)
sc = api.Scene(s, device_build=True); rt = api.RayTracer(sc); rt.render(api.PATH, 1, 3, 0); got, n = rt.film(); cnt = rt.counters()
fw, fg = np.isfinite(want).all(axis=2), np.isfinite(got).all(axis=2)
print("nonfinite px oracle", (~fw).sum(), "gpu", (~fg).sum(), "xor", (fw ^ fg).sum())
both = fw & fg
a, b = got[both].astype(np.float64), want[both].astype(np.float64)
close = np.abs(a - b) <= 1e-3 * np.maximum(np.abs(a), np.abs(b)) + 1e-4
print("pixels", both.sum(), "close frac", close.all(axis=1).mean(), "not close px", (~close.all(axis=1)).sum())
print("rays", cnt["closest_rays"], st["closest_rays"], cnt["any_rays"], st["any_rays"])
lim = 1e3
ok = both & (want.max(axis=2) < lim) & (got.max(axis=2) < lim)
d = got[ok].astype(np.float64) - want[ok].astype(np.float64)
print("relMSE on px < 1e3:", (d * d).mean() / (want[ok].astype(np.float64) ** 2).mean(), "px", ok.sum())
ys, xs = np.nonzero(fw ^ fg)
for y, x in zip(ys, xs): print("px", y, x, "oracle", want[y, x], "gpu", got[y, x])
nc = np.zeros(both.shape, bool); nc[both] = ~close.all(axis=1)
ys, xs = np.nonzero(nc)
for y, x in list(zip(ys, xs))[:25]: print("nc", y, x, "oracle", want[y, x], "gpu", got[y, x])
mx = np.maximum(want[nc].max(axis=1), got[nc].max(axis=1)); print("not-close magnitudes: <10:", (mx < 10).sum(), "<1e3:", (mx < 1e3).sum(), "all", len(mx))
