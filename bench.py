#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on MI355X.

Workload (BASELINE.json configs[1]): the reference's scenes/test_scene_01.json
(re-authored as scenes_amd/cfg2_cube.json: 12-triangle Lambert cube, one directional
light) with scenes/pt.json (max_depth 8, recurrence sampler, box filter 0.5) at
1024x1024, 256 spp on one GPU.  A "step" is one full render of that image.

N > 1 (weak scaling): the image stays 1024x1024 and spp becomes 256*N; image rows are
sharded over the ranks in interleaved 16-row strips, so every rank traces the same
268.4 M camera samples per step with the same scene coverage.  There is no data-path
collective; the shards are gathered on the host of rank 0 inside the timed region.

Launch: `python bench.py` (N=1) or, for N>1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
 --master-port P bench.py --gpus N --steps K --warmup W`.
PyTorch is used only for the barrier / synchronize / gather plumbing.
"""
import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    name = "simple_path_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "simple-path-tracer_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# record sizes of the queue layouts in simple-path-tracer_amd/csrc/hip/kernels.h (bytes)
S_PATH, S_HIT, S_SHADOW, S_RAD = 72, 20, 48, 12
S_PATH0 = 16  # compact bounce-0 record: direction + slot
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(spt, scene, renderer, width, height, budget_s=15.0):
    """Oracle (CPU restatement of the reference loop) timed on this box's host cores, on a
    bounded sample of the SAME workload: the full 1024x1024 frame at a reduced spp."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _util

    _util.ensure_cpu_build()
    cores = len(os.sched_getaffinity(0))
    threads = 2 * cores  # reference layout: num_cpus * 2 threads over row bands (pt.rs:243)
    probe = spt.PathTracer(renderer.max_depth, renderer.sampler, 1, 0, 0, renderer.filter_radius, renderer.seed)
    t0 = time.perf_counter()
    _util.oracle_render(scene, probe, width, height, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-3)
    spp = int(max(1, min(renderer.spp, budget_s / dt)))
    run = spt.PathTracer(renderer.max_depth, renderer.sampler, spp, 0, 0, renderer.filter_radius, renderer.seed)
    t0 = time.perf_counter()
    _, st = _util.oracle_render(scene, run, width, height, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(st.samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "threads": threads,
        "kind": "port",
        "sample": "%dx%d @ %d spp of the same scene/renderer (%.1f s, %d samples)" % (width, height, spp, dt, st.samples),
    }, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per GPU-share")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
    ap.add_argument("--renderer", default=os.path.join(ROOT, "scenes_amd", "pt.json"))
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernel classes with HIP events")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend of the control plane (N > 1): a barrier, one max-reduce of a double and one "
                         "name broadcast.  The data path has no collective (every rank writes its rows into the shared film), so "
                         "nothing needs RCCL; pass cpu:gloo,cuda:nccl to run the barriers over RCCL / xGMI instead")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (use with --backend gloo)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."); stdout carries exactly one JSON
        # line, so file descriptor 1 points at stderr while the process group comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
            dist.barrier(device_ids=[local_rank]) if "nccl" in args.backend else dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    spt = load_pkg()
    scene = spt.load_scene(args.scene)
    renderer = spt.load_renderer(args.renderer, seed=1)
    renderer.spp = args.spp * world
    cfg = spt.OutputConfig(args.width, args.height)
    strip_rows = 16
    # scene upload (excluded from the timed region: inputs resident in HBM)
    scene.device_scene(local_rank)

    def barrier():
        torch.cuda.synchronize()          # this rank's GPU work is done ...
        if dist is not None:
            if "nccl" in args.backend:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()            # ... and so is everybody else's
        torch.cuda.synchronize()

    film = None
    if dist is not None:
        name = [None]
        if rank == 0:
            film = spt.SharedFilm(args.height, args.width, create=True)
            name[0] = film.name
        dist.broadcast_object_list(name, src=0)
        if rank != 0:
            film = spt.SharedFilm(args.height, args.width, name=name[0])
        film.pin()

    kernel_ms = np.zeros(spt.N_KERNELS)
    kernel_launches = np.zeros(spt.N_KERNELS, dtype=np.int64)
    stats_last = None

    def step(profiled):
        nonlocal stats_last
        # N > 1: every rank's strips are DMA-ed straight into the node's shared-memory film (no collective, no host-side
        # scatter); N = 1: the pinned shard buffer is the image
        shard = renderer.render_shard(scene, cfg, device=local_rank, shard_index=rank, shard_count=world,
                                      strip_rows=strip_rows, samples_per_pass=args.samples_per_pass,
                                      profile=profiled, reuse_output=True, film=film.film if film is not None else None)
        st = renderer.last_stats
        if profiled:
            for k in range(spt.N_KERNELS):
                kernel_ms[k] += st.kernel_ms[k]
                kernel_launches[k] += st.kernel_launches[k]
        stats_last = st
        return shard

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel HIP-event timing on the render stream: separate, untimed steps right after the timed
    # region (an event pair around every launch costs ~7 % of a step, so it stays out of `value`)
    if not args.no_profile:
        for _ in range(args.profile_steps):
            step(True)
        barrier()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        samples_per_step = args.width * args.height * renderer.spp  # all ranks together
        ms_per_step = dt / args.steps * 1e3
        value = samples_per_step * args.steps / dt / 1e6
        st = stats_last                                              # rank 0's shard, one step
        n_launch = np.maximum(kernel_launches, 1)
        n_prof = max(args.profile_steps, 1)
        smp, seg_c, seg_s = st.samples, st.segments_closest, st.segments_shadow
        ext, hits0, verts = seg_c - smp, st.primary_hits, st.path_vertices
        n_pix = len(spt.shard_rows(args.height, 0, world, strip_rows)) * args.width
        passes = max(int(kernel_launches[0]) // n_prof, 1)
        RNAME = {"shade_first": "k_shade"}                           # kernel symbol behind a class name
        # ALGORITHMIC bytes each kernel class moves per step (DESIGN.md "Kernels and rooflines"):
        # queue records written/read once + radiance-slot / film read-modify-writes; scene geometry of
        # this workload (< 2 KB) is LDS-resident and counted as 0 ("count it once").
        # fused bounces (LDS-resident scenes): shade traces its own shadow / extension rays, so the shadow and
        # path-to-extend records do not exist; it reads the vertices, writes the next bounce's vertices and
        # read-modify-writes one radiance slot per (unoccluded) shadow ray
        fused = int(kernel_launches[2]) == 0 and int(kernel_launches[3]) == 0
        seg_s0, verts1 = st.shadow_first, st.vertices_second        # bounce 0: shadow rays issued, vertices kept for bounce 1
        later = verts - hits0                                        # path vertices of bounces >= 1
        alg = {
            # un-chunked: hit records + a zeroed slot per hit + film read / write; chunked (sample chunks per tile): every
            # sample of a pixel inside the screen-space bound zeroes / fills its slot, the film is left to k_resolve
            "primary": (hits0 * (S_PATH0 + S_HIT) + st.live_samples * S_RAD) if st.live_samples else
                       hits0 * (S_PATH0 + S_HIT + S_RAD) + passes * n_pix * 2 * S_RAD,
            # bounce-0 shade launches (one per pass): read the compact hit records, (fused) write the kept bounce-1
            # vertices and read-modify-write a radiance slot per shadow ray, (un-fused) write shadow + path records
            "shade_first": (hits0 * (S_PATH0 + S_HIT) + verts1 * (S_PATH + S_HIT) + seg_s0 * 2 * S_RAD) if fused else
                           hits0 * (S_PATH0 + S_HIT) + seg_s0 * S_SHADOW + min(ext, hits0) * S_PATH,
            "shade": (later * (S_PATH + S_HIT) + (later - verts1) * (S_PATH + S_HIT) + (seg_s - seg_s0) * 2 * S_RAD) if fused else
                     later * (S_PATH + S_HIT) + (seg_s - seg_s0) * S_SHADOW + max(ext - hits0, 0) * S_PATH,
            "shadow": seg_s * (S_SHADOW + 2 * S_RAD),
            "extend": ext * S_PATH + (verts - hits0) * (S_PATH + S_HIT),
            "resolve": (st.live_samples if st.live_samples else hits0) * S_RAD + passes * n_pix * 2 * S_RAD,
        }
        kern = {}
        for k in (0, 6, 1, 2, 3, 4):
            name = spt.KERNEL_NAMES[k]
            if kernel_launches[k] == 0:
                continue
            total_ms = float(kernel_ms[k]) / n_prof                 # per step
            launches = int(kernel_launches[k]) // n_prof
            gbs = alg[name] / (total_ms * 1e-3) / 1e9 if total_ms > 0 else 0.0
            kern[name] = {"launches_per_step": launches, "avg_launch_ms": round(total_ms / launches, 4),
                          "ms_per_step": round(total_ms, 3), "alg_MB_per_launch": round(alg[name] / launches / 1e6, 3),
                          "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        dom_name = max(kern, key=lambda n: kern[n]["ms_per_step"]) if kern else "primary"
        dom = kern.get(dom_name, {"GBps": 0.0, "frac": 0.0, "avg_launch_ms": None, "alg_MB_per_launch": None})
        pipeline_bytes = sum(alg[n] for n in kern)                     # only the kernel classes that ran (fused: no shadow / extend)
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), collected in a
        # separate rocprofv3 --pmc run of this same command and committed under profiles/ (tools/pmc_traffic.py)
        traffic, traffic_src = None, os.path.join(ROOT, "profiles", "r01_traffic_bench.json")
        if os.path.exists(traffic_src) and world == 1 and args.samples_per_pass == 0 and args.spp == 256:
            tk = json.load(open(traffic_src))["kernels"].get("k_" + dom_name)   # "k_shade_first": bounce-0 instances
            if tk and tk["launches"] == dom.get("launches_per_step"):   # (per-template-instance entries: see tools/pmc_traffic.py)
                traffic = round(tk["hbm_bytes_per_launch"] / 1e6, 3)
        roofline = {
            "bound": "hbm", "kernel": RNAME.get(dom_name, "k_" + dom_name) + (" (bounce 0)" if dom_name == "shade_first" else ""), "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_unit": "MB per launch (PMC, profiles/r01_traffic_bench.json)",
            "avg_launch_ms": dom["avg_launch_ms"], "alg_MB_per_launch": dom["alg_MB_per_launch"],
            "kernels": kern,
            "pipeline": {"alg_bytes_per_sample": round(pipeline_bytes / max(smp, 1), 2),
                         "GBps_at_value": round(pipeline_bytes / max(smp, 1) * value * 1e6 / world / 1e9, 1)},
            "note": "HIP-event kernel times from %d untimed steps after the timed region; the path is ALU/latency-bound "
                    "(tiny scene, misses never touch HBM), so the HBM fraction is low by design - see DESIGN.md" % n_prof,
        }
        out = {
            "metric": "Msamples/sec (whole node) at 1024x1024/256spp; per-pixel mean L1 vs CPU ref",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "scenes_amd/cfg2_cube.json (= reference scenes/test_scene_01.json) + pt.json, "
                                   "%dx%d @ %d spp (%d spp per GPU-share), max_depth %d, recurrence sampler"
                                   % (args.width, args.height, renderer.spp, args.spp, renderer.max_depth),
                       "width": args.width, "height": args.height, "spp": renderer.spp, "seed": 1,
                       "sharding": "interleaved %d-row strips over %d rank(s); each rank writes its rows into one shared-memory film "
                                   "(no collective)" % (strip_rows, world),
                       "segments_per_sample": round((seg_c + seg_s) / max(smp, 1), 4),
                       "primary_hit_fraction": round(hits0 / max(smp, 1), 4)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, ost = cpu_baseline(spt, scene, spt.load_renderer(args.renderer, seed=1), args.width, args.height)
            out["cpu_baseline"] = base
            out["config"]["speedup_vs_cpu_baseline"] = round(value / base["value"], 1)
        print(json.dumps(out))
    if dist is not None:
        barrier()
        if rank == 0 and film is not None:   # the assembled image: every pixel was written by exactly one rank
            assert np.isfinite(film.film).all() and float(film.film.max()) > 0.0
        barrier()
        film.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
