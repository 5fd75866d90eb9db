"""The N > 1 layout on real kernels: two processes (one per rank, as bench.py runs them) share the one GPU of the
box, each renders its interleaved row strips with the HIP path straight into one page-locked shared-memory film
(SharedFilm: no collective, no gather), and the assembled image equals a single-process render bit for bit - also
with a wide box filter, where every rank traces the halo rows of its strips itself."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

import _util

pytestmark = pytest.mark.gpu


def _rank(rank, world, name, scene_name, camera, w, h, spp, radius, strip):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _util as u
    spt = u.load_pkg()
    film = spt.SharedFilm(h, w, name=name)
    film.pin()
    sc = spt.load_scene(os.path.join(u.SCENES, scene_name))
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=spp, seed=77, filter_radius=radius)
    r.render_shard(sc, spt.OutputConfig(w, h, None, camera), shard_index=rank, shard_count=world, strip_rows=strip, film=film.film)
    sc.close()
    film.close()


@pytest.mark.parametrize("scene_name,camera,radius", [("t_materials.json", "main", 0.5), ("cfg2_cube.json", None, 1.5), ("t_bezier.json", "main", 0.5)])
def test_two_ranks_assemble_one_film_in_shared_memory(scene_name, camera, radius):
    spt = _util.load_pkg()
    w, h, spp, strip, world = 96, 72, 6, 8, 2
    film = spt.SharedFilm(h, w, create=True)
    film.film[:] = np.float32(-1.0)
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank, args=(k, world, film.name, scene_name, camera, w, h, spp, radius, strip)) for k in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    # a rank that is still alive hangs (GPU hang, deadlock): kill exactly that child - it would otherwise keep the GPU and
    # its page-locked mapping of the film for the rest of the lease - and fail with the ranks' exit states
    hung = [k for k, p in enumerate(procs) if p.is_alive()]
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join()
    try:
        assert not hung, "rank(s) %s did not finish within 240 s and were killed" % hung
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        sc = spt.load_scene(os.path.join(_util.SCENES, scene_name))
        r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=spp, seed=77, filter_radius=radius)
        whole = r.render_shard(sc, spt.OutputConfig(w, h, None, camera))
        got = np.array(film.film)
        nan = np.isnan(whole)
        assert np.array_equal(nan, np.isnan(got))
        assert np.array_equal(got.view(np.uint32)[~nan], whole.view(np.uint32)[~nan])
    finally:
        film.close()
