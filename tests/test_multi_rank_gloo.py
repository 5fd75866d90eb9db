"""N > 1 path on CPU: two ranks over gloo shard the image rows, render their shard (here with the
CPU oracle standing in for the HIP call, which needs a GPU) and gather on the host of rank 0.
Covers shard_rows / gather_shards / the barrier + max-over-ranks timing pattern of bench.py."""
import os
import socket
import sys

import numpy as np
import pytest

import _util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    import time

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _util as u

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    spt = u.load_pkg()
    sc = spt.load_scene(os.path.join(u.SCENES, "t_materials.json"))
    r = spt.PathTracer(max_depth=4, sampler=spt.SAMPLER_RANDOM, spp=2, seed=5)
    w, h, strip = 48, 40, 8
    dist.barrier()
    t0 = time.perf_counter()
    shard, _ = u.oracle_render(sc, r, w, h, camera="main", shard_index=rank, shard_count=world, strip_rows=strip, threads=2)
    full = spt.gather_shards(shard, h, w, rank, world, strip, dist)
    # the collective-free assembly bench.py uses: one shared-memory film, every rank writes its own rows
    name = [None]
    film = None
    if rank == 0:
        film = spt.SharedFilm(h, w, create=True)
        film.film[:] = -1.0
        name[0] = film.name
    dist.broadcast_object_list(name, src=0)
    dist.barrier()                      # the fill above is done before anyone writes
    if rank != 0:
        film = spt.SharedFilm(h, w, name=name[0])
    film.write_shard(shard, rank, world, strip)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        ref, _ = u.oracle_render(sc, r, w, h, camera="main", threads=2)
        np.savez(out_path, full=full, ref=ref, dt=dt.numpy(), shared=np.array(film.film))
    else:
        assert full is None
    dist.barrier()
    film.close()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_is_bit_identical(tmp_path):
    import torch.multiprocessing as mp

    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    assert z["full"].shape == (40, 48, 3)
    assert np.array_equal(z["full"].view(np.uint32), z["ref"].view(np.uint32))
    assert np.array_equal(z["shared"].view(np.uint32), z["ref"].view(np.uint32))
    assert z["dt"][0] > 0


def test_shard_rows_partition_every_layout():
    spt = _util.load_pkg()
    for h, world, strip in ((37, 3, 4), (1024, 8, 16), (5, 8, 16), (64, 2, 1)):
        seen = np.concatenate([spt.shard_rows(h, r, world, strip) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(h))


def test_bench_launches_its_own_ranks_when_typed_plainly():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) must start the two ranks itself: here, without a
    GPU, each rank gets as far as the device check and says so - the parent did not exit with a launch-me-differently error."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(_util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if p.returncode == 0:      # a box with GPUs: rank 0 printed the line
        assert '"n_gpus": 2' in p.stdout and '"scaling": "strong"' in p.stdout
    else:
        assert p.stderr.count("no GPU visible; the HIP path has no CPU fallback") >= 1, p.stderr[-2000:]
        assert "WORLD_SIZE=" not in p.stderr
