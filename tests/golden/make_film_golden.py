"""Converged reference film for the statistical image gate (relMSE < 1e-3 at 4096 spp).
The reference's Renderer.h cannot be built in this image (needs <Windows.h>/OIDN), so this film comes from the
CPU oracle (oracle/rt_oracle.c), with an RNG seed different from the one the GPU test uses.
    python tests/golden/make_film_golden.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402
from raytracer_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
for name, s in (("film_cornell64_4096", scenes.cornell(64, 64)), ("film_cornellspec64_4096", scenes.cornell(64, 64, with_specular=True))):
    o = orc.OracleScene(s)
    t = time.time()
    film, st = o.render(mode=0, spp=4096, seed=987654321, threads=8)
    print(name, "%.1fs" % (time.time() - t), st, film.mean())
    np.save(os.path.join(HERE, name + ".npy"), (film / 4096).astype(np.float32))
