#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and ms per spp, bathroom-class scene, 1920x1080 path tracing (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--width 3840 --height 2160] [--scaling strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one sample per pixel of the whole frame (2 073 600 paths at 1080p, <= 6 bounces each, NEE shadow rays).
Inputs (BVH, triangles, materials, film, path state) are resident in HBM before the timed region.  The timed region is
exactly K steps between barrier + synchronize on both sides; it is repeated R times (fresh sample indices each time) and the
line reports the MEDIAN repeat, with min / max beside it.  With N > 1 each rank renders its own slice of the sample-index
space on a full scene replica and the accumulation films are summed with one RCCL all-reduce over xGMI inside the timed
region: K spp per rank with --scaling weak (the default), K spp in all, split by dist.sample_range, with --scaling strong.
`ms_per_step` is always ms per GLOBAL sample per pixel.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
NODE_BYTES, TRI_BYTES = 80, 64   # the walk's quantised 8-wide node (round 2: 64-B 4-wide nodes; round 1: 128-B float nodes) and the triangle record
# k_extend per ray besides node/triangle fetches: ray origin + direction (32) + flags / throughput (16) + hit write (16) + queue read/write (8)
# + the gate box of the reported hit (32)
EXTEND_FIXED_BYTES = 104
PMC_FILE = os.path.join("profiles", "r03_pmc.json")   # written by tools/pmc.sh + tools/pmc_sum.py from rocprofv3 --pmc runs of this script


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps each; the line reports their median (0: as many as make >= 2.5 s of timed work, at least 9)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)   # BASELINE config 4: --width 3840 --height 2160
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--detail", type=float, default=1.0, help="scene tessellation (1.0 = ~1.1 M triangles)")
    ap.add_argument("--scene", default="bathroom")
    ap.add_argument("--conductor", action="store_true", help="SURVEY 8(d)'s conductor variant of the scene (reported separately)")
    ap.add_argument("--dielectric", action="store_true", help="rough-dielectric variant (upstream's estimator yields inf / NaN pixels)")
    ap.add_argument("--glazed", action="store_true", help="PlasticBSDF on the smooth curved fixtures too (upstream's 0/0 at their silhouettes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-unshared-pass", action="store_true", help="skip the one-stream pass that measures unshared kernel durations")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` block (the other BASELINE configs, N = 1 only)")
    ap.add_argument("--cpu-spp", type=int, default=8)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs WORLD_SIZE={a.gpus} (launch with torch.distributed.run); got {world}")

    # the host driver only supports dmabuf IPC: RCCL's buffer registration across ranks needs this before the HIP runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    from raytracer_amd import api, scenes
    from raytracer_amd import dist as rdist

    # RT_BENCH_REHEARSE=1: the N>1 control flow on a one-GPU box (every rank on GPU 0, gloo instead of RCCL); its
    # numbers mean nothing, tests/test_two_ranks_gpu.py uses it to check that the line comes out and adds up
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    if api.lib().rt_device_set(local_rank) != 0:
        raise SystemExit("rt_device_set failed: " + api.lib().rt_last_error().decode())

    W, H = a.width, a.height
    sdata = scenes.interior(a.scene, a.detail, W, H, seed=1, conductor=a.conductor, dielectric=a.dielectric, glazed=a.glazed)
    scene = api.Scene(sdata)
    info = scene.info()
    rt = api.RayTracer(scene)
    film = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)  # film memory owned by torch so RCCL can reduce it
    torch.cuda.synchronize()
    rt.bind_film(film.data_ptr())

    K, Wm, R = a.steps, a.warmup, a.repeats
    seed = 1
    # this rank's share of one timed region: K spp (weak), or its slice of the K global spp (strong)
    if a.scaling == "strong":
        lo, k_rank = rdist.sample_range(K, rank, world)
        k_global = K
    else:
        lo, k_rank, k_global = rank * K, K, K * world
    # disjoint global sample indices: region j of the run covers [j * k_global, (j + 1) * k_global)
    def first_of(region):
        return region * k_global + lo

    # warm-up renders use the timed render's batch shape, so that every k_extend launch of the process has the same size and
    # the rocprofv3 per-kernel average matches the HIP-event average reported below
    warm_calls = (Wm + K - 1) // K if Wm > 0 else 0
    region = 0
    for _ in range(warm_calls):
        if k_rank:
            rt.render(api.PATH, k_rank, seed, first_of(region))
        rt.sync()   # (also keeps rt_render from merging consecutive warm-up calls into one bigger batch)
        region += 1
    if world > 1:
        dist.all_reduce(film)  # warm-up of the film reduction too: RCCL sets up its channels / buffers on the first call of a size
        torch.cuda.synchronize()
    rt.clear()
    rt.counters(reset=True)
    rt.profile_enable(True)
    rt.profile(reset=True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    samples = []   # per repeat: (dt max over ranks, rays of all ranks, closest rays of this rank)
    spp_in_film = 0
    closest_total, any_total = 0, 0
    rep = 0
    while True:
        # repeats: as asked, or (0) until the timed regions add up to 2.5 s and there are at least 9 of them; every rank sees the
        # same max-over-ranks times, so all leave the loop together
        if (R > 0 and rep >= R) or (R <= 0 and rep >= 9 and sum(t for t, _ in samples) >= 2.5) or rep >= 200:
            break
        rep += 1
        barrier()
        t0 = time.perf_counter()
        if k_rank:
            rt.render(api.PATH, k_rank, seed, first_of(region))
        rt.sync()
        if world > 1:
            dist.all_reduce(film)  # RCCL sum of the accumulation films (W*H*3 fp32), once per render
        barrier()
        dt = time.perf_counter() - t0
        region += 1
        cnt = rt.counters(reset=True)
        closest_total += cnt["closest_rays"]; any_total += cnt["any_rays"]
        rays = cnt["closest_rays"] + cnt["any_rays"]
        tt = torch.tensor([dt, float(rays)], dtype=torch.float64, device=dev)
        if world > 1:
            tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            samples.append((float(tmax[0]), float(tsum[1])))
        else:
            samples.append((dt, float(rays)))
        # after an all-reduce every rank's film holds the sum of all ranks' films, and the next region adds to that on every
        # rank: the film is only a throughput vehicle here, its statistics below are taken from the first region alone
        if spp_in_film == 0:
            spp_in_film = k_global
            finite = torch.isfinite(film)
            film_mean = float(film[finite].double().mean()) / max(k_global, 1)
            film_median = float(film[finite].median()) / max(k_global, 1)
            film_nan = int((~finite).sum())
    prof = rt.profile(reset=True)
    rt.profile_enable(False)

    R = len(samples)
    rates = sorted(r / t / 1e6 for t, r in samples)
    times = sorted(t for t, _ in samples)
    med_rate, med_dt = statistics.median(rates), statistics.median(times)

    # which GPU each rank ran on (PCI bus id), so that the record shows N distinct devices behind an N-GPU line
    my_dev = f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_properties(local_rank).name} pci {pci_bus_id(local_rank)}"
    devices = [my_dev]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, my_dev)

    out = None
    if rank == 0:
        # one-stream pass: the same K spp with every kernel running alone (no second batch sharing the GPU)
        unshared = None
        if not a.no_unshared_pass and k_rank:
            rt1 = api.RayTracer(scene)
            rt1.set_streams(1)
            rt1.render(api.PATH, k_rank, seed, first_of(region)); rt1.sync()   # warm-up of the tagged kernel
            rt1.profile_enable(True); rt1.profile(reset=True); rt1.counters(reset=True)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rt1.render(api.PATH, k_rank, seed, first_of(region + 1)); rt1.sync()
            dt1 = time.perf_counter() - t1
            p1 = rt1.profile(reset=True); c1 = rt1.counters(reset=True)
            rt1.close()
            unshared = {"wall_ms": dt1 * 1e3, "mrays_per_s": (c1["closest_rays"] + c1["any_rays"]) / dt1 / 1e6,
                        "kernel_ms": {k: v[0] for k, v in p1.items()}, "extend_launches": p1["extend"][1],
                        "extend_avg_launch_ms": p1["extend"][0] / max(p1["extend"][1], 1), "closest_rays": c1["closest_rays"]}
        # algorithmic bytes per closest-hit ray on this workload: counted (untimed) pass over the same samples
        scene.counters_enable(True)
        rt2 = api.RayTracer(scene)
        rt2.render(api.PATH, min(4, max(k_rank, 1)), seed, first_of(region - 1))
        c2 = rt2.counters(reset=True)
        scene.counters_enable(False)
        rt2.close()
        nodes_per_ray = c2["nodes_visited"] / max(c2["closest_rays"], 1)
        tris_per_ray = c2["tris_tested"] / max(c2["closest_rays"], 1)
        any_nodes_per_ray = c2["any_nodes_visited"] / max(c2["any_rays"], 1)
        any_tris_per_ray = c2["any_tris_tested"] / max(c2["any_rays"], 1)
        bytes_per_ray = EXTEND_FIXED_BYTES + nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES
        ext_ms, ext_n = prof["extend"]
        con_ms, con_n = prof["connect"]
        # dominant kernel = k_extend; over all R timed regions: closest_total rays in ext_n launches taking ext_ms
        rays_per_launch = closest_total / max(ext_n, 1)
        bytes_per_launch = rays_per_launch * bytes_per_ray
        avg_launch_ms = ext_ms / max(ext_n, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        achieved_unshared = None
        if unshared and unshared["extend_avg_launch_ms"] > 0:
            achieved_unshared = (unshared["closest_rays"] / max(unshared["extend_launches"], 1)) * bytes_per_ray / (unshared["extend_avg_launch_ms"] * 1e-3) / 1e9
        shape = {"scene": sdata.name, "triangles": info["n_tris"], "width": W, "height": H, "spp_per_call": k_rank,
                 "variant": [v for v in ("conductor", "dielectric", "glazed") if getattr(a, v)]}
        traffic, ceilings = pmc_ceilings(shape)
        out = {
            "metric": "Mrays/s (closest + any-hit rays issued / wall time of render); ms_per_step = ms per global spp, bathroom-class 1080p",
            "value": med_rate,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": Wm,
            "ms_per_step": med_dt * 1e3 / max(k_global, 1),
            "higher_is_better": True,
            "scaling": a.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "repeats": {"n": R, "value_min": rates[0], "value_max": rates[-1], "ms_per_step_min": times[0] * 1e3 / max(k_global, 1),
                        "ms_per_step_max": times[-1] * 1e3 / max(k_global, 1), "timed_seconds_total": sum(times),
                        "note": "value / ms_per_step are the median of `n` timed regions of exactly `steps` steps each"},
            "config": {
                "workload": f"{sdata.name} procedural stand-in{' (' + ', '.join(shape['variant']) + ' variant)' if shape['variant'] else ''}, {info['n_tris']} triangles, "
                            f"{W}x{H}, path tracing (RT_MODE_PATH), 1 spp per step, {k_rank} spp per rank and timed region, {k_global} spp per region over all ranks; "
                            f"materials: {material_mix(a)}",
                "devices": devices,
                "triangles": info["n_tris"], "width": W, "height": H, "spp_per_rank": k_rank, "spp_global": k_global,
                "parallelism": f"sample-partition x{world}" + (", RCCL film all-reduce" if world > 1 else "") + ", 2 streams per GPU"
                               + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearse else ""),
                "rays_per_pixel_sample": samples[0][1] / (W * H * max(k_global, 1)),
                "film_mean": film_mean, "film_median": film_median, "film_nonfinite_values": film_nan,
            },
            "roofline": {
                # what limits the dominant kernel: VALU instruction issue first, latency second, neither HBM nor MFMA.  By the PMC counters of
                # this command (profiles/) a launch issues 1.29e9 VALU wave-instructions at 4.4 SIMD cycles each against a measured mean issue
                # cost of ~3.9 (tools/ubench_valu.hip), with 38 of 64 lanes live per instruction
                "bound": "issue",
                "bound_is": "VALU instruction issue, neither HBM nor MFMA: cycles per ray = sum over step kinds of (lane-steps per ray / lanes served per wave step) x "
                            "instructions per step x issue cost reproduces the kernel's time (8-wide node step 265 VALU + 82 SALU instructions, triangle step 135 + 39; "
                            "issue cost 4.0-4.3 cycles per wave-instruction, 2.6 for add / mul / logic ops, measured); the issue slots are ~89 % full, the rest is latency "
                            "that 5 waves per SIMD do not hide; fabric traffic is ~0.26 of the HBM peak",
                "kernel": "k_extend<false,false,0> (closest-hit walk over the 8-wide quantised tree)",
                # the SURVEY 8(d) figure: ALGORITHMIC bytes (every 80-B node and 64-B triangle fetch counted, cache hits included)
                # per launch / average launch duration by HIP events over the timed regions, against the HBM peak
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_is": "algorithmic bytes, cache-served: the ~0.3 GB the walk touches sits in L2 / Infinity Cache, so this is NOT an HBM utilisation "
                           "and the unshared figure can exceed 1",
                "achieved_unshared": achieved_unshared,
                "frac_unshared": None if achieved_unshared is None else achieved_unshared / HBM_PEAK_GBS,
                # fabric-side bytes per launch (L2 misses: Infinity-Cache hits included) from a PMC run of this batch shape, else null
                "traffic": traffic,
                "traffic_source": None if traffic is None else f"constant read from {PMC_FILE} (rocprofv3 --pmc passes of this command on this batch shape, tools/pmc.sh), not measured by this run",
                "traffic_is": "2 x FETCH_SIZE + WRITE_SIZE per k_extend launch (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B on gfx950); "
                              "L2-miss traffic on the fabric, Infinity-Cache hits included - an upper bound on HBM bytes",
                "ceilings": ceilings,
                "shape": shape,
                "bytes_per_ray": bytes_per_ray,
                "nodes_per_ray": nodes_per_ray,
                "tris_per_ray": tris_per_ray,
                "rays_per_launch": rays_per_launch,
                "bytes_per_launch": bytes_per_launch,
                "lane_occupancy": c2["lane_busy"] / max(c2["lane_slots"], 1),
                "launches": ext_n,
                "avg_launch_ms": avg_launch_ms,
                "avg_launch_ms_unshared": None if unshared is None else unshared["extend_avg_launch_ms"],
                "note": "The render runs its batches on two streams so that launch tails overlap: avg_launch_ms / kernel_ms are durations with the other "
                        "batch's kernels sharing the GPU (they add up to more than the wall time).  avg_launch_ms_unshared is the same kernel "
                        "(k_extend<false,false,1>) in a one-stream pass of the same batch shape.",
            },
            "kernel_ms": {k: v[0] for k, v in prof.items()},
            "unshared": unshared,
            "connect": {"nodes_per_ray": any_nodes_per_ray, "tris_per_ray": any_tris_per_ray,
                        "lane_occupancy": c2["any_lane_busy"] / max(c2["any_lane_slots"], 1),
                        "rays": any_total, "ms": con_ms, "launches": con_n},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sdata, a.cpu_spp)
    rt.close()
    if rank == 0 and world == 1 and not a.no_workloads and W == 1920 and H == 1080 and a.scene == "bathroom" and a.detail == 1.0 and not shape["variant"]:
        out["workloads"] = other_workloads(api, scenes, torch, scene, sdata)
    scene.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def pci_bus_id(index):
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, index) == 0:
            return buf.value.decode()
    except Exception:
        pass
    return "?"


def material_mix(a):
    mix = "diffuse / Oren-Nayar / glass / mirror"
    if a.glazed:
        mix += " + plastic (fixtures, blinds, shelves)"
    if a.dielectric:
        mix += " + rough dielectric"
    if a.conductor:
        mix += " + conductor"
    if not (a.glazed and a.dielectric):
        mix += "; plastic and the rough dielectric are opt-in (--glazed --dielectric: upstream's estimators for them yield NaN / inf samples, DESIGN.md section 6)"
    return mix


def other_workloads(api, scenes, torch, bath_scene, bath_data):
    """The other BASELINE configs in the same driver-run line (N = 1): each the median of 3 short timed regions after one warm-up
    region, with its own film's non-finite count.  Rates are (closest + any-hit rays issued) / wall time, like `value`."""
    import numpy as np
    res = {}

    def measure(name, scene, sdata, mode, spp, regions=3, vpls=0, note=None):
        rt = api.RayTracer(scene)
        if vpls:
            rt.traceVPLs(vpls, 1)
        rt.render(mode, spp, 1, 0); rt.sync()                      # warm-up region of the timed shape
        rt.clear(); rt.counters(reset=True)
        rs, ts, first = [], [], spp
        for _ in range(regions):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rt.render(mode, spp, 1, first); rt.sync()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            c = rt.counters(reset=True)
            rs.append((c["closest_rays"] + c["any_rays"]) / dt / 1e6); ts.append(dt * 1e3 / spp)
            first += spp
        film, n = rt.film()
        rt.close()
        res[name] = {"mrays_per_s": statistics.median(rs), "ms_per_spp": statistics.median(ts), "mrays_per_s_min": min(rs), "mrays_per_s_max": max(rs),
                     "spp_per_region": spp, "regions": regions, "width": sdata.width, "height": sdata.height, "triangles": sdata.ntris,
                     "film_nonfinite_values": int((~np.isfinite(film)).sum()), "film_spp": int(n)}
        if note:
            res[name]["note"] = note

    measure("light_tracing_1080p", bath_scene, bath_data, api.LIGHT, 16, note="BASELINE config 5: RT_MODE_LIGHT (Renderer.h:327), atomic splats")
    measure("pssmlt_1080p", bath_scene, bath_data, api.MLT, 32, note="BASELINE config 5: RT_MODE_MLT, one chain of 100 000 proposals per spp (Renderer.h:357)")
    measure("instant_radiosity_50vpl_1080p", bath_scene, bath_data, api.RADIOSITY, 1, vpls=50,
            note="BASELINE config 5: RT_MODE_RADIOSITY after rt_vpl_trace(50) (the count upstream's commented-out call names, Renderer.h:68)")
    for label, kw, note in (("bathroom_3840x2160", dict(w=3840, h=2160), "BASELINE config 4's frame on one GPU"),
                            ("bathroom_full_mix_1080p", dict(glazed=True, dielectric=True), "SURVEY 8(d)'s full material mix: plastic + rough dielectric (upstream's NaN / inf samples included)"),
                            ("living_room_1080p", dict(name="living-room"), "BASELINE config 3")):
        sd = scenes.interior(kw.get("name", "bathroom"), 1.0, kw.get("w", 1920), kw.get("h", 1080), seed=1, glazed=kw.get("glazed", False), dielectric=kw.get("dielectric", False))
        sc = api.Scene(sd)
        measure(label, sc, sd, api.PATH, 8 if "w" in kw else 20, note=note + "; path tracing")
        sc.close()
    return res


def pmc_ceilings(shape):
    """Counter-derived ceilings of k_extend from the committed PMC summary -- only when that summary was taken on this very
    batch shape (tools/pmc_sum.py records the shape); otherwise (None, None): a number measured on another shape is
    not this run's."""
    path = os.path.join(ROOT, PMC_FILE)
    try:
        pj = json.load(open(path))
    except Exception:
        return None, None
    if pj.get("shape") != shape:
        return None, None
    g = lambda k: pj.get("k_extend_" + k)
    t = g("launch_seconds")
    l2 = None
    if g("l2_requests_per_launch") and t:
        l2_gbs = g("l2_requests_per_launch") * 128.0 / t / 1e9
        l2 = {"achieved": l2_gbs, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": l2_gbs / L2_PEAK_GBS,
              "is": "TCC_REQ x 128 B per launch / launch time under the profiler (kernels serialised): an upper bound, requests of 64 B count as 128",
              "hit_rate": g("l2_hit_rate")}
    valu = None
    if g("valu_busy_frac") is not None and g("live_lanes_per_valu_inst") is not None:
        valu = {"busy": g("valu_busy_frac"), "live_lanes_per_inst": g("live_lanes_per_valu_inst"),
                "lane_utilisation": min(g("valu_busy_frac"), 1.0) * g("live_lanes_per_valu_inst") / 64.0,
                "is": "share of the chip's VALU lane-cycles doing work: VALU busy (issue slots in use) x live lanes per instruction / 64"}
    ceilings = {"source": PMC_FILE, "shape": pj.get("shape"), "l2_bandwidth": l2, "valu": valu, "ta_busy": g("ta_busy_frac"),
                "wait_inst_frac_of_wave_cycles": g("wait_inst_frac_of_wave_cycles"),
                "fabric_gbs": None if not (g("hbm_bytes_per_launch") and t) else g("hbm_bytes_per_launch") / t / 1e9,
                "limiter": pj.get("k_extend_limiter")}
    return g("hbm_bytes_per_launch"), ceilings


def cpu_baseline(sdata, spp):
    """The CPU oracle (oracle/rt_oracle.c: the reference's pointer-BVH recursive traversal + path tracer, restated)
    on the host cores of this box, same scene and resolution, bounded to `spp` samples per pixel."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    o = orc.OracleScene(sdata)
    t = time.perf_counter()
    _, st = o.render(0, spp, seed=1, first_sample=0, threads=threads)
    dt = time.perf_counter() - t
    rays = st["closest_rays"] + st["any_rays"]
    o.close()
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{spp} spp of the same {sdata.width}x{sdata.height} frame ({rays} rays, {dt:.1f} s), gcc -O2 -ffp-contract=off",
            "ms_per_spp": dt * 1e3 / spp}


if __name__ == "__main__":
    main()
