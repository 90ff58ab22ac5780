"""CPU, world_size 2 over gloo: the N>1 path of bench.py / raytracer_amd.dist -- sample-index partition per rank and one
sum all-reduce of the accumulation films.  Each rank's film here is a deterministic stand-in (the
renderer itself needs a GPU); what is under test is the partition arithmetic and the reduction protocol."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total_spp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = rdist.sample_range(total_spp, rank, world)
    # stand-in film: sample k contributes (k+1) to every pixel, so the reduced film must be sum_{k<total}(k+1)
    film = torch.zeros(4 * 3 * 3, dtype=torch.float32)
    for k in range(first, first + count):
        film += float(k + 1)
    spp = rdist.allreduce_film(film, count)
    if rank == 0:
        out.put((film.numpy().copy(), spp))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_partition_covers_every_index_once():
    for total in (0, 1, 7, 16, 4096):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c = rdist.sample_range(total, r, world)
                seen += list(range(f, f + c))
            assert seen == list(range(total))
            counts = [rdist.sample_range(total, r, world)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def test_two_rank_film_allreduce_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    total = 11
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, out)) for r in range(2)]
    for p in procs:
        p.start()
    film, spp = out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert spp == total
    assert np.allclose(film, sum(k + 1 for k in range(total)))
