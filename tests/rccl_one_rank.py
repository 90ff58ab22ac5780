"""Helper of tests/test_two_ranks_gpu.py::test_rccl_all_reduce_on_the_bound_film (run as its own process, so that RCCL's first
initialisation happens in a fresh process before any other GPU call): one rank, backend "nccl" (= RCCL on ROCm), the film
all-reduce of raytracer_amd.dist on a film tensor bound with rt_film_bind."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")

import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from raytracer_amd import api, scenes
from raytracer_amd import dist as rdist

s = scenes.interior("bathroom", 0.1, 192, 108)
sc = api.Scene(s)
rt = api.RayTracer(sc)
film = torch.zeros(108 * 192 * 3, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
spp = rdist.render_distributed(rt, api.PATH, 6, 5, film)          # renders [0, 6) and runs all_reduce(film) + all_reduce(spp) over RCCL
torch.cuda.synchronize()
got = film.cpu().numpy()
ref = api.RayTracer(sc); ref.render(api.PATH, 6, 5, 0); want, n = ref.film(); ref.close()
assert spp == 6 and rt.getSPP() == 6 and n == 6
assert np.array_equal(got.view(np.uint32), want.reshape(-1).view(np.uint32)), "a one-rank RCCL all-reduce changed the film"
t = torch.ones(1 << 20, device="cuda:0"); dist.all_reduce(t); assert float(t.sum()) == float(1 << 20)
print("rccl one-rank ok: backend", dist.get_backend(), "film sum", float(got.sum()))
rt.close(); sc.close()
dist.destroy_process_group()
