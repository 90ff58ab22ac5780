"""GPU, two ranks on the one GPU of the test box, gloo in place of RCCL: raytracer_amd.dist.render_distributed
partitions the global sample indices, each rank renders its range into its own film, and the all-reduced film must be exactly
the sum of the two ranges' films rendered by one process (a + b in fp32 is the same either way round).  bench.py's N>1 control
flow is rehearsed the same way (RT_BENCH_REHEARSE=1)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from raytracer_amd import api, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, TOTAL, SEED = 96, 54, 5, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene():
    return scenes.interior("bathroom", 0.1, W, H)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from raytracer_amd import dist as rdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert api.lib().rt_device_set(0) == 0
    sc = api.Scene(_scene()); rt = api.RayTracer(sc)
    film = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    spp = rdist.render_distributed(rt, api.PATH, TOTAL, SEED, film)
    if rank == 0:
        out.put((film.cpu().numpy().copy(), spp, rt.getSPP()))
    dist.barrier()
    rt.close(); sc.close()
    dist.destroy_process_group()


def test_two_ranks_sum_to_the_films_of_their_sample_ranges():
    from raytracer_amd import dist as rdist
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    film, spp, film_spp = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert spp == TOTAL and film_spp == TOTAL
    sc = api.Scene(_scene())
    parts = []
    for r in range(2):
        first, count = rdist.sample_range(TOTAL, r, 2)
        rt = api.RayTracer(sc); rt.render(api.PATH, count, SEED, first); parts.append(rt.film()[0].reshape(-1)); rt.close()
    sc.close()
    assert np.array_equal(film.view(np.uint32), (parts[0] + parts[1]).view(np.uint32))


@pytest.mark.parametrize("scaling,steps,per_rank,total", [("weak", 2, 2, 4), ("strong", 3, 2, 3)])
def test_bench_line_with_two_ranks(scaling, steps, per_rank, total):
    """bench.py --gpus 2 end to end (two ranks on the test box's one GPU, gloo instead of RCCL): weak scaling renders `steps` spp
    per rank, strong scaling splits `steps` spp over the ranks (3 -> 2 + 1); ms_per_step is per GLOBAL spp either way."""
    env = dict(os.environ, RT_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1",
           "--repeats", "3", "--detail", "0.1", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == steps and b["scaling"] == scaling and b["value"] > 0
    assert b["config"]["spp_per_rank"] == per_rank and b["config"]["spp_global"] == total and b["repeats"]["n"] == 3
    assert "cpu_baseline" not in b                           # N=1 only
    # both ranks' rays are in the total: value x the region's wall time = rays per pixel-sample x pixels x GLOBAL spp
    rays = b["value"] * 1e6 * b["ms_per_step"] * 1e-3 * total
    want = b["config"]["rays_per_pixel_sample"] * 1920 * 1080 * total
    assert 0.75 < rays / want < 1.33, (rays, want)          # median region vs the first region's count
    assert 4.0 < b["config"]["rays_per_pixel_sample"] < 7.5


def test_rccl_all_reduce_on_the_bound_film():
    """backend "nccl" is RCCL on ROCm: initialise it once on the test box (one rank), reduce a film that the renderer wrote
    through rt_film_bind, and check that the film and the SPP count come back unchanged -- so that the driver's multi-GPU run
    is not the first time RCCL meets this code."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py"), str(_free_port())], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl one-rank ok: backend nccl" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_bench_line_contract_one_gpu():
    """The one JSON line of `python bench.py` at N=1 carries every field the driver and the judge read (a reduced scene and
    one CPU sample keep it to seconds; the numbers themselves are not judged here)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "3", "--detail", "0.1", "--cpu-spp", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["higher_is_better"] is True and b["vs_baseline"] is None
    assert b["unit"] == "Mrays/s" and b["dtype"] == "f32" and b["data"] == "synthetic" and "workload" in b["config"]
    rf = b["roofline"]
    assert rf["bound"] == "issue" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # counter-derived figures are only printed for the batch shape they were measured on: this reduced scene has none
    assert rf["traffic"] is None and rf["ceilings"] is None
    assert rf["avg_launch_ms_unshared"] > 0 and rf["achieved_unshared"] > 0 and b["unshared"]["extend_launches"] == 6
    rp = b["repeats"]
    assert rp["n"] == 3 and rp["value_min"] <= b["value"] <= rp["value_max"] and rp["ms_per_step_min"] <= b["ms_per_step"] <= rp["ms_per_step_max"]
    assert b["config"]["film_nonfinite_values"] == 0
    cb = b["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mrays/s" and "spp" in cb["sample"]
    assert b["value"] > 20 * cb["value"]
