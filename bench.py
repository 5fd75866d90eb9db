#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on MI355X.

Workload (BASELINE.json metric / configs[1]): the reference's scenes/test_scene_01.json (scenes_amd/cfg2_cube.json:
12-triangle Lambert cube, one directional light) with scenes/pt.json (max_depth 8, recurrence sampler, box filter
0.5) at 1024x1024, 256 spp.  A "step" is one full render of that image.

N > 1 is STRONG scaling of that same run: the image rows are dealt to the ranks in interleaved 16-row strips, every
rank traces its 1/N of the 268.4 M camera samples with a full scene replica and DMAs its rows straight into one
shared-memory film (no data-path collective).  `value` = 268.4 M samples x steps / max-over-ranks wall time.
Secondary entries of the same line (their own short timed regions, outside `value`): the weak-scaling figure
(spp = 256 N, the same work per GPU as at N = 1) and BASELINE configs[2] (4096x4096 @ 1024 spp over N ranks); at
N = 1 also `other_configs` (configs[3] and one GPU's share of configs[4]) and `parity` against the CPU oracle.

Launch: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 the parent process only spawns
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
BEFORE anything touches the GPU and relays its output; launching it under torch.distributed.run directly (RANK /
WORLD_SIZE set) works too.  PyTorch is only the barrier / synchronize / max-reduce plumbing.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_pkg():
    import importlib.util
    name = "simple_path_tracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "simple-path-tracer_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# record sizes of the queue layouts in simple-path-tracer_amd/csrc/hip/kernels.h (bytes)
S_PATH, S_HIT, S_SHADOW, S_RAD = 72, 24, 48, 12   # S_HIT: (t, v, w, prim) + (instance, source index of the path record)
S_RAY = 32           # what a traversal reads of a path / shadow record: (origin, t_min) (direction, pdf | t_max)
S_PATH0 = 16  # compact bounce-0 record: direction + slot
# geometry records (SURVEY 8d / DESIGN.md section 3): node, triangle positions, triangle attributes, instance
# (instance: what a walker reads per instance visit - M^-1, ids, the BLAS root box and ref: 6 x 16 B, the streaming walker's
#  96-byte record and the same six loads in trace.h's to_object + mesh root; the ABI's 192-byte spt_instance is never fetched whole)
S_NODE, S_TRI, S_ATTR, S_INST = 64, 48, 144, 96
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Vector-ALU issue, measured on this chip (tools/issue_rate_bench*.hip -> profiles/r03_issue_rate.md): a wave64 instruction costs a
# SIMD 2 cycles (VOP1 / VOP2 with <= 2 VGPR sources), 4 (most VOP3: three sources, compares, selects, integer multiplies, the
# division helpers, packed f32) or 8 (transcendental unit); 1024 SIMDs at the ~2.1 GHz the chip holds under this load.
N_SIMD = 1024
VALU_CLOCK_GHZ = 2.1
VALU_PEAK_VOP2_G = 950.0   # G wave-instr/s, all-VOP2 stream, measured (the guide's 2 cycles per wave64 instruction)


def profile_json(*names):
    """First of the committed profile summaries that exists (newest round first)."""
    for n in names:
        p = os.path.join(ROOT, "profiles", n)
        if os.path.exists(p):
            return n, json.load(open(p))
    return None, None


def valu_block(kernel_class, kernel_symbol_regex, avg_launch_ms, launches_per_step, world, default_pass_layout):
    """The instruction-issue side of the roofline for one kernel class: wave-instructions per launch from the committed PMC pass of
    this same command, the kernel's own static instruction mix priced with the measured issue costs, and the live launch time."""
    import re
    vf, vj = profile_json("r03_valu_bench.json")
    mf, mj = profile_json("r03_isa_mix.json")
    if not vj or not avg_launch_ms or world != 1 or not default_pass_layout:
        return None
    e = vj["kernels"].get("k_" + kernel_class)
    if not e or "SQ_INSTS_VALU_per_launch" not in e or (launches_per_step and e["launches"] % launches_per_step):
        return None
    instr = e["SQ_INSTS_VALU_per_launch"]
    clock = VALU_CLOCK_GHZ
    if e.get("GRBM_GUI_ACTIVE_per_launch") and e.get("SQ_BUSY_CYCLES_per_launch"):
        pass   # (the PMC pass's own duration is not the live one; the clock constant above is the measured figure)
    avg_cost, mix_kernel, sat = None, None, None
    if mj:
        for name, m in mj["kernels"].items():
            if re.search(kernel_symbol_regex, name):
                avg_cost, mix_kernel, sat = m["avg_issue_cycles_per_wave_instr"], name, m.get("saturation_rate_G_wave_instr_per_s")
                break
    achieved = instr / (avg_launch_ms * 1e-3) / 1e9
    out = {"kernel": mix_kernel or "k_" + kernel_class, "unit": "G wave-instr/s", "wave_instr_per_launch": round(instr),
           "lane_utilisation": round(e.get("lane_utilisation", 0.0), 4), "achieved": round(achieved, 1),
           "peak_vop2_stream": VALU_PEAK_VOP2_G, "frac_of_vop2_stream": round(achieved / VALU_PEAK_VOP2_G, 4),
           "source": "SQ_INSTS_VALU of profiles/%s (rocprofv3 --pmc pass of this command) / the live HIP-event launch time" % vf}
    if avg_cost:
        peak = sat if sat else N_SIMD * clock / avg_cost
        out.update({"avg_issue_cycles_per_wave_instr": avg_cost, "peak": round(peak, 1), "frac": round(achieved / peak, 4),
                    "peak_note": "the rate at which THIS kernel's static instruction mix (profiles/%s: 2- / 4- / 8-cycle classes) saturates the vector "
                                 "ALU, from the measured chip-wide rates of the classes (950 / 560 / 300 G wave-instr/s, profiles/r03_issue_rate.md)" % mf})
    if e.get("valu_busy_fraction"):
        out["valu_busy_pmc"] = round(e["valu_busy_fraction"], 4)   # 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): an upper bound (quad-cycle granularity)
    return out


def counting_variant(kernel_name):
    """True for the SPT_RENDER_COUNT_VISITS instantiations (kCount = true), which the timed runs never launch."""
    import re
    m = re.match(r"(k_\w+)<([^>]*)>", kernel_name)
    if not m:
        return False
    args = [a.strip() for a in m.group(2).split(",")]
    pos = {"k_primary": 2, "k_extend": 1, "k_shadow": 1, "k_primary_stream": 1}.get(m.group(1), 0 if m.group(1).endswith(("_stream", "_dyn")) else None)
    return pos is not None and pos < len(args) and args[pos] == "true"


def oracle_util():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _util
    _util.ensure_cpu_build()
    return _util


def cpu_baseline(spt, scene, renderer, width, height, budget_s=15.0):
    """Oracle (CPU restatement of the reference loop, reference-faithful mode: exact six-division slab test, the reference's
    visit order) timed on this box's host cores, on a bounded sample of the SAME workload: the full frame at the largest spp
    that fits the budget (the full 256 spp on a many-core host).  Returns the line, the film and the renderer used."""
    _util = oracle_util()
    cores = len(os.sched_getaffinity(0))
    threads = 2 * cores  # reference layout: num_cpus * 2 threads over row bands (pt.rs:243)
    probe = spt.PathTracer(renderer.max_depth, renderer.sampler, 1, 0, 0, renderer.filter_radius, renderer.seed)
    t0 = time.perf_counter()
    _util.oracle_render(scene, probe, width, height, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-3)
    spp = int(max(1, min(renderer.spp, budget_s / dt)))
    run = spt.PathTracer(renderer.max_depth, renderer.sampler, spp, 0, 0, renderer.filter_radius, renderer.seed)
    t0 = time.perf_counter()
    film, st = _util.oracle_render(scene, run, width, height, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(st.samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "threads": threads,
        "kind": "port",
        "sample": "%dx%d @ %d spp of the same scene/renderer (%.1f s, %d samples)" % (width, height, spp, dt, st.samples),
    }, film, run


def film_parity(spt, got, ref):
    """The metric's parity figure: per-pixel mean L1 on linear float radiance, plus what bit-exactness adds."""
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    ok = ~(nan_g | nan_r)
    u8g, u8r = spt.film_to_rgb8(got).astype(np.int16), spt.film_to_rgb8(ref).astype(np.int16)
    return {"mean_L1": float(np.abs(got - ref)[ok].mean()) if ok.any() else None,
            "words_differ": int((got.view(np.uint32) != ref.view(np.uint32))[ok].sum()), "words": int(got.size),
            "nan_equal": bool(np.array_equal(nan_g, nan_r)), "nan_words": int(nan_r.sum()),
            "u8_pixels_differ_gt1": int((np.abs(u8g - u8r) > 1).any(axis=2).sum())}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` as typed: this process never initialises the GPU; it starts the N ranks as a child
    (torch.distributed.run) and passes their stdout (one JSON line from rank 0) and exit code on."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel of the image (split over the ranks)")
    ap.add_argument("--scene", default=os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
    ap.add_argument("--renderer", default=os.path.join(ROOT, "scenes_amd", "pt.json"))
    ap.add_argument("--samples-per-pass", type=int, default=0)
    ap.add_argument("--sync-steps", action="store_true", help="timed steps wait for their film one by one (no copy / compute overlap between steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline and parity)")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernel classes with HIP events")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary entries (weak scaling, configs[2], other_configs)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--backend", default="gloo",
                    help="torch.distributed backend of the control plane (N > 1): two barriers and one max-reduce of a double per "
                         "timed region, one name broadcast.  The data path has no collective (every rank writes its rows into the "
                         "shared film), so nothing needs RCCL; pass cpu:gloo,cuda:nccl to run the barriers over RCCL / xGMI instead")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses device 0 (use with --backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))     # nothing above imported torch or opened the HIP runtime

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
    # interleave granularity of the row strips: finer strips even out the ranks' shares of the object (8 ranks, slowest / mean
    # rank on one GPU: 0.634 / 0.550 ms with 16-row strips, 0.602 / 0.571 ms with 4-row strips; tools/strip_rows_sweep.py)
    strip_rows = 16 if world < 8 else 4
    scene.device_scene(local_rank)   # scene upload: outside every timed region (inputs resident in HBM)

    def barrier():
        torch.cuda.synchronize()          # this rank's GPU work is done ...
        if dist is not None:
            if "nccl" in args.backend:
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()            # ... and so is everybody else's
        torch.cuda.synchronize()

    films = {}

    def shared_film(h, w):
        """One (h, w, 3) f32 film in POSIX shared memory per image size, page-locked by every rank (N > 1)."""
        if dist is None:
            return None
        if (h, w) not in films:
            name = [None]
            f = None
            if rank == 0:
                f = spt.SharedFilm(h, w, create=True)
                name[0] = f.name
            dist.broadcast_object_list(name, src=0)
            if rank != 0:
                f = spt.SharedFilm(h, w, name=name[0])
            f.pin()
            films[(h, w)] = f
        return films[(h, w)]

    def timed(renderer, sc, cfg, steps, warmup, profile_steps=0):
        """`steps` renders of this rank's shard of `cfg`, bracketed by barrier + synchronize on both sides, max over ranks.
        Returns (seconds, last stats, per-class kernel ms, per-class launches) - the kernel times come from separate,
        untimed steps right after the timed region (an event pair around every launch costs ~7 % of a step)."""
        film = shared_film(cfg.height, cfg.width)
        kernel_ms = np.zeros(spt.N_KERNELS)
        kernel_launches = np.zeros(spt.N_KERNELS, dtype=np.int64)

        def step(profiled, wait=True):
            # N > 1: this rank's strips are DMA-ed straight into the node's shared-memory film (no collective, no host-side
            # scatter); N = 1: the pinned shard buffer is the image
            renderer.render_shard(sc, cfg, device=local_rank, shard_index=rank, shard_count=world, strip_rows=strip_rows,
                                  samples_per_pass=args.samples_per_pass, profile=profiled, reuse_output=True,
                                  film=film.film if film is not None else None, wait=wait)
            st = renderer.last_stats
            if profiled:
                for k in range(spt.N_KERNELS):
                    kernel_ms[k] += st.kernel_ms[k]
                    kernel_launches[k] += st.kernel_launches[k]
            return st

        st = None
        for _ in range(warmup):
            st = step(False)      # (synchronous: these also deliver the counters of the bench line, every step is the same work)
        barrier()
        t0 = time.perf_counter()
        # the timed steps are queued back to back (SPT_RENDER_ASYNC): step k's film leaves over PCIe on the copy stream while
        # step k + 1's kernels run; every film has arrived in host memory when wait() returns, before the closing barrier
        for _ in range(steps):
            step(False, wait=args.sync_steps)
        renderer.wait(sc, device=local_rank)
        barrier()
        dt = time.perf_counter() - t0
        if st is None:
            st = step(False)      # --warmup 0: the counters from one untimed step
        for _ in range(profile_steps):
            step(True)
        if profile_steps:
            barrier()
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, st, kernel_ms, kernel_launches

    # ---- the metric's run: 1024 x 1024 @ 256 spp over `world` ranks (strong scaling) ----------------------------------
    renderer = spt.load_renderer(args.renderer, seed=1)
    renderer.spp = args.spp
    cfg = spt.OutputConfig(args.width, args.height)
    n_prof = 0 if args.no_profile else max(args.profile_steps, 1)
    dt, stats_last, kernel_ms, kernel_launches = timed(renderer, scene, cfg, args.steps, args.warmup, n_prof)

    # ---- secondary entries: their own timed regions, never part of `value` -------------------------------------------
    secondary = {}
    if not args.no_secondary:
        if world > 1:
            weak = spt.load_renderer(args.renderer, seed=1)
            weak.spp = args.spp * world
            k = max(3, min(args.steps, 40))
            wdt, _, _, _ = timed(weak, scene, cfg, k, 1)
            secondary["weak_scaling"] = {
                "workload": "%dx%d @ %d spp (= %d spp per GPU-share: the N = 1 work on every GPU)" % (args.width, args.height, weak.spp, args.spp),
                "value": round(args.width * args.height * weak.spp * k / wdt / 1e6, 2), "unit": "Msamples/s", "steps": k,
                "ms_per_step": round(wdt / k * 1e3, 3), "scaling": "weak"}
        big = spt.load_renderer(args.renderer, seed=1)
        big.spp = 1024
        bcfg = spt.OutputConfig(4096, 4096)
        k = 3 if world > 1 else 2
        bdt, _, _, _ = timed(big, scene, bcfg, k, 1)
        secondary["configs2_cube_4096x4096_1024spp"] = {
            "workload": "BASELINE configs[2]: the same scene at 4096x4096 @ 1024 spp, rows dealt to %d rank(s)" % world,
            "value": round(4096 * 4096 * 1024 * k / bdt / 1e6, 2), "unit": "Msamples/s", "steps": k, "ms_per_step": round(bdt / k * 1e3, 2)}

    if rank == 0:
        samples_per_step = args.width * args.height * renderer.spp  # all ranks together
        ms_per_step = dt / args.steps * 1e3
        value = samples_per_step * args.steps / dt / 1e6
        st = stats_last                                              # rank 0's shard, one step
        n_launch = np.maximum(kernel_launches, 1)
        smp, seg_c, seg_s = st.samples, st.segments_closest, st.segments_shadow
        ext, hits0, verts = seg_c - smp, st.primary_hits, st.path_vertices
        n_pix = len(spt.shard_rows(args.height, 0, world, strip_rows)) * args.width
        passes = max(int(kernel_launches[0]) // max(n_prof, 1), 1)
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
        live = st.live_samples
        alg = {
            # chunked (sample chunks per tile): a hit writes its compact record, a black sample writes nothing; one mask byte per 8
            # samples of a live pixel says which samples own a radiance slot.  un-chunked: hit records + film read / write
            "primary": (hits0 * (S_PATH0 + S_HIT) + live // 8) if live else
                       hits0 * (S_PATH0 + S_HIT) + passes * n_pix * 2 * S_RAD,
            # bounce-0 shade launches (one per pass): read the compact hit records, WRITE the sample's radiance slot once (the first
            # contributions are summed in registers), (fused) write the kept bounce-1 vertices, (un-fused) shadow + path records
            "shade_first": (hits0 * (S_PATH0 + S_HIT + S_RAD) + verts1 * (S_PATH + S_HIT)) if fused else
                           hits0 * (S_PATH0 + S_HIT + S_RAD) + seg_s0 * S_SHADOW + min(ext, hits0) * S_PATH,
            "shade": (later * (S_PATH + S_HIT) + (later - verts1) * (S_PATH + S_HIT) + (seg_s - seg_s0) * 2 * S_RAD) if fused else
                     later * (S_PATH + S_HIT) + (seg_s - seg_s0) * S_SHADOW + max(ext - hits0, 0) * S_PATH,
            "shadow": seg_s * (S_SHADOW + 2 * S_RAD),
            # the extend stage reads the ray of a path record and leaves a hit + the record's index (no copy of the record)
            "extend": ext * S_RAY + (verts - hits0) * S_HIT,
            # chunked passes (k_resolve_bits): every slot of a live pixel is read (marked or not), its mask byte per 8 samples, the 4-byte
            # first_slot word of every pixel and the film read + write of the live pixels; plus k_finish's film read + image write
            "resolve": (live * S_RAD + live // 8 + passes * (n_pix * 4 + (live // max(renderer.spp, 1)) * 2 * S_RAD) + n_pix * 2 * S_RAD) if live else
                       hits0 * S_RAD + passes * n_pix * 2 * S_RAD,
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
        traffic, traffic_file = None, None
        for cand in ("r03_traffic_bench.json", "r02_traffic_bench.json", "r01_traffic_bench.json"):
            if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                traffic_file = cand
                break
        if traffic_file and world == 1 and args.samples_per_pass == 0 and args.spp == 256:
            tk = json.load(open(os.path.join(ROOT, "profiles", traffic_file)))["kernels"].get("k_" + dom_name)   # "k_shade_first": bounce-0 instances
            # (the PMC run renders the step a few times: its launch count is a multiple of the step's when the pass layout is the same)
            if tk and dom.get("launches_per_step") and tk["launches"] % dom["launches_per_step"] == 0:
                traffic = round(tk["hbm_bytes_per_launch"] / 1e6, 3)
        roofline = {
            "bound": "hbm", "kernel": RNAME.get(dom_name, "k_" + dom_name) + (" (bounce 0)" if dom_name == "shade_first" else ""), "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_unit": "MB per launch (PMC, profiles/%s)" % traffic_file,
            "avg_launch_ms": dom["avg_launch_ms"], "alg_MB_per_launch": dom["alg_MB_per_launch"],
            # `bound` keeps the vocabulary of the bench contract (hbm | mfma: achieved / peak / frac above are the HBM figures of the
            # dominant kernel); what actually limits the kernel is named here and priced in `valu`
            "limited_by": "valu-issue",
            "limiter": "VALU instruction issue, not HBM: the scene is LDS-resident and a missing ray touches no memory at all (DESIGN.md section 6)",
            "valu": valu_block(dom_name, {"primary": r"k_primary<true, true, false, true>", "shade_first": r"k_shade<0, true, true"}.get(dom_name, "k_" + dom_name),
                               dom.get("avg_launch_ms"), dom.get("launches_per_step"), world, args.samples_per_pass == 0 and args.spp == 256),
            "kernels": kern,
            "pipeline": {"alg_bytes_per_sample": round(pipeline_bytes / max(smp, 1), 2),
                         "GBps_at_value": round(pipeline_bytes / max(smp, 1) * value * 1e6 / world / 1e9, 1)},
            "note": "HIP-event kernel times from %d untimed steps after the timed region (rank 0's shard)" % n_prof,
        }
        out = {
            "metric": "Msamples/sec (whole node) at 1024x1024/256spp; per-pixel mean L1 vs CPU ref",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "scenes_amd/cfg2_cube.json (= reference scenes/test_scene_01.json) + pt.json, "
                                   "%dx%d @ %d spp, max_depth %d, recurrence sampler; the SAME image for every N (%d row strips of %d "
                                   "dealt round-robin to %d rank(s))" % (args.width, args.height, renderer.spp, renderer.max_depth,
                                                                      (args.height + strip_rows - 1) // strip_rows, strip_rows, world),
                       "width": args.width, "height": args.height, "spp": renderer.spp, "seed": 1,
                       "samples_per_step": samples_per_step,
                       "sharding": "interleaved %d-row strips over %d rank(s); each rank writes its rows into one shared-memory film "
                                   "(no collective)" % (strip_rows, world),
                       "per_rank": {"samples": int(smp), "live_samples": int(st.live_samples),
                                    "note": "live = camera samples of pixels inside the scene's screen-space bound; the others are provably "
                                            "black, move no bytes and run no sample loop (bit-identical film; SPT_NO_PIXEL_CULL=1 for the A/B)"},
                       "segments_per_sample": round((seg_c + seg_s) / max(smp, 1), 4),
                       "primary_hit_fraction": round(hits0 / max(smp, 1), 4),
                       "what_bounds_large_N": "a rank's share is 1/N of ~4.1 ms of kernels; the per-step fixed cost (counter memset, ~10 "
                                              "launches of ~5 us, strided D2H of the rank's rows, one stream sync) does not shrink with N"},
            "roofline": roofline,
        }
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            ref_renderer = spt.load_renderer(args.renderer, seed=1)
            ref_renderer.spp = args.spp
            base, ref_film, ran = cpu_baseline(spt, scene, ref_renderer, args.width, args.height)
            out["cpu_baseline"] = base
            out["config"]["speedup_vs_cpu_baseline"] = round(value / base["value"], 1)
            # parity of the metric: GPU film vs the oracle film cpu_baseline() just rendered (same spp), and vs the oracle
            # in the configuration the kernels reproduce bit for bit (no box culling, order-independent tie rule)
            _util = oracle_util()
            gpu_film = ran.render_shard(scene, cfg, device=local_rank).copy()
            exhaustive, _ = _util.oracle_render(scene, ran, args.width, args.height, flags=_util.ORACLE_EXHAUSTIVE,
                                                threads=2 * len(os.sched_getaffinity(0)))
            out["parity"] = {"spp": ran.spp, "tolerance_mean_L1": 1e-3,
                             "vs_reference_walk": film_parity(spt, gpu_film, ref_film),
                             "vs_exhaustive_oracle": film_parity(spt, gpu_film, exhaustive),
                             "note": "reference_walk = the oracle as bvh.rs / bbox.rs are written (the timed cpu_baseline); exhaustive = "
                                     "every triangle tested, (t, instance, prim) tie rule - the answer the library's padded trees reproduce"}
        if world == 1 and not args.no_secondary:
            try:
                out["other_configs"] = other_configs(spt, local_rank)
            except Exception as e:   # the headline line must survive a failure of the extras
                out["other_configs"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        barrier()
        f = films.get((args.height, args.width))
        if rank == 0 and f is not None:   # the assembled image: every pixel was written by exactly one rank
            assert np.isfinite(f.film).all() and float(f.film.max()) > 0.0
        barrier()
        for f in films.values():
            f.close()
        dist.destroy_process_group()


def other_configs(spt, device, only=""):
    """BASELINE configs[3] (cfg4, 1024x1024 @ 512 spp) and one GPU's share of configs[4] (cfg5, shard 0 of 8 of 2048x2048 @ 512
    spp): Msamples/s, per-kernel-class ms, and the dominant kernel's algorithmic bytes.  Scenes are generated from their
    seeds (scenes_amd/make_scenes.py).  Outside the timed `value`."""
    sys.path.insert(0, os.path.join(ROOT, "scenes_amd"))
    import make_scenes
    gen = make_scenes.make_full()
    res = {}
    for key, scene_f, rend_f, w, h, shard_count, label in (
            ("cfg4_materials_env", "cfg4_materials_env.json", "pt_random512.json", 1024, 1024, 1,
             "BASELINE configs[3]: GGX conductor + rough / smooth glass + 1024x512 EXR env MIS, 1024x1024 @ 512 spp, whole image"),
            ("cfg5_blob_medium_shard0of8", "cfg5_blob_medium.json", "pt_recurrence512.json", 2048, 2048, 8,
             "BASELINE configs[4]: 998 k-triangle mesh + homogeneous medium, 2048x2048 @ 512 spp, ONE GPU's share (shard 0 of 8)")):
        if only and not key.startswith(only):
            continue
        sc = spt.load_scene(os.path.join(gen, scene_f))
        r = spt.load_renderer(os.path.join(gen, rend_f), seed=1)
        cfg = spt.OutputConfig(w, h, None, "main")
        kw = dict(device=device, shard_index=0, shard_count=shard_count, strip_rows=16, reuse_output=True)
        r.render_shard(sc, cfg, **kw)                                  # warm-up (workspace allocation)
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            r.render_shard(sc, cfg, **kw)
            best = min(best, time.perf_counter() - t0)
        st = r.last_stats
        smp, seg_c, seg_s = st.samples, st.segments_closest, st.segments_shadow
        ext, hits0, verts = seg_c - smp, st.primary_hits, st.path_vertices
        r.render_shard(sc, cfg, profile=True, **kw)                    # per-class HIP-event times (one stream)
        pst = r.last_stats
        kms = {spt.KERNEL_NAMES[k]: float(pst.kernel_ms[k]) for k in range(spt.N_KERNELS) if pst.kernel_launches[k]}
        # visit counters of the traversal kernels (separate counting instantiations, SPT_RENDER_COUNT_VISITS)
        r.render_shard(sc, cfg, count_visits=True, **kw)
        vis = r.last_stats
        geo = vis.node_visits * S_NODE + vis.tri_tests * S_TRI + vis.instance_visits * S_INST
        queues = {
            "primary": hits0 * (S_PATH0 + S_HIT) + (st.live_samples // 8 if st.live_samples else 0),
            "shade_first": hits0 * (S_PATH0 + S_HIT + S_ATTR + S_RAD) + st.shadow_first * S_SHADOW + min(ext, hits0) * S_PATH,
            "shade": (verts - hits0) * (S_PATH + S_HIT + S_ATTR) + (seg_s - st.shadow_first) * S_SHADOW + max(ext - hits0, 0) * S_PATH,
            "shadow": seg_s * (S_RAY + 16 + 2 * S_RAD),
            "extend": ext * S_RAY + (verts - hits0) * S_HIT,
        }
        dom = max((k for k in kms if k in queues), key=lambda k: kms[k])
        entry = {
            "workload": label, "samples": int(smp), "ms": round(best * 1e3, 2), "Msamples_per_s": round(smp / best / 1e6, 1),
            "Mrays_per_s": round((seg_c + seg_s) / best / 1e6, 1), "segments_per_sample": round((seg_c + seg_s) / smp, 3),
            "kernel_ms": {k: round(v, 2) for k, v in kms.items()}, "dominant_kernel": dom,
        }
        traversal = {"primary", "shadow", "extend"}
        if vis.node_visits:
            segs = seg_c + seg_s
            entry["visits"] = {"node_records_per_segment": round(vis.node_visits / segs, 2), "triangles_per_segment": round(vis.tri_tests / segs, 2),
                               "instances_per_segment": round(vis.instance_visits / segs, 2), "geometry_bytes_per_segment": round(geo / segs, 1)}
            rays = {"primary": smp, "shadow": seg_s, "extend": ext}
            geo_cls = {}
            for ci, cname in enumerate(("primary", "shadow", "extend")):
                cv = vis.class_visits[ci]
                geo_cls[cname] = cv[0] * S_NODE + cv[1] * S_TRI + cv[2] * S_INST
                entry["visits"][cname] = {"rays": int(rays[cname]), "nodes_per_ray": round(cv[0] / max(rays[cname], 1), 2),
                                          "triangles_per_ray": round(cv[1] / max(rays[cname], 1), 2), "instances_per_ray": round(cv[2] / max(rays[cname], 1), 2)}
            alg = {k: queues[k] + (geo_cls[k] if k in traversal else 0) for k in queues}
        else:
            entry["visits"] = None      # LDS-resident geometry: read once per workgroup, not per visit
            alg = queues
        # HBM side from the committed counters of this same workload (tools/pmc_cfg.sh: FETCH_SIZE x 2 + WRITE_SIZE per dispatch of
        # the dominant kernel's instances) x this run's launches / this run's class time; the visit-counter figure is kept beside
        # it as what it is: the rate at which L2 / Infinity Cache serve the walk
        cls_kernels = {"extend": ("k_extend_stream", "k_extend_dyn", "k_extend<"), "shadow": ("k_shadow",), "primary": ("k_primary",),
                       "shade": ("k_shade<1, false", "k_shade<2, false", "k_shade<3, false", "k_shade<0, false"),
                       "shade_first": ("k_shade<1, true", "k_shade<2, true", "k_shade<3, true", "k_shade<0, true")}[dom]
        tag = "cfg4" if key.startswith("cfg4") else "cfg5"
        pf, pj = profile_json("r03_pmc_%s/pmc_summary.json" % tag)
        hbm = None
        if pj:
            byt, disp = 0.0, 0
            for kname, t in pj.items():
                if kname.startswith(cls_kernels) and not counting_variant(kname) and "FETCH_SIZE" in t and t.get("dispatches"):
                    byt += (2.0 * t["FETCH_SIZE"] + t.get("WRITE_SIZE", 0.0)) * 1024.0
                    disp += t["dispatches"]
            launches = int(pst.kernel_launches[list(spt.KERNEL_NAMES).index(dom)]) if dom in list(spt.KERNEL_NAMES) else 0
            if disp and launches:
                hbm = byt / disp * launches
        fetch_gbs = alg[dom] / (kms[dom] * 1e-3) / 1e9
        if hbm is not None:
            gbs = hbm / (kms[dom] * 1e-3) / 1e9
            lanes = [t["derived"]["valu_lane_utilisation"] for kname, t in pj.items()
                     if kname.startswith(cls_kernels) and not counting_variant(kname) and "derived" in t and "valu_lane_utilisation" in t["derived"]]
            lane_txt = ("%.2f" % (sum(lanes) / len(lanes))) if lanes else "n/a"
            limited = {"shade": "memory system: the stage streams every vertex's hit, path and shadow records (~220 B) and samples the environment map; its "
                                "L2-miss traffic is the fraction below (Infinity-Cache hits included), VALU lane utilisation %s with class-binned queues "
                                "(DESIGN.md section 6)" % lane_txt,
                       "shade_first": "memory system + VALU issue (lane utilisation %s)" % lane_txt}.get(
                dom, "divergent instruction issue at %s VALU lane utilisation (5 waves per SIMD; at 4 the kernel waited for its ~9 dependent node / leaf / instance "
                     "records per ray, served by L1 - hit rate 0.92 - / L2 / Infinity Cache; DESIGN.md section 6)" % lane_txt)
            entry["roofline"] = {"kernel": "k_" + dom, "bound": "hbm", "limited_by": limited,
                                 "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "hbm_MB": round(hbm / 1e6, 1), "ms": round(kms[dom], 2),
                                 "source": "FETCH_SIZE x 2 + WRITE_SIZE per dispatch (profiles/%s) x %d launches of this run" % (pf, launches),
                                 "cache_fetch": {"alg_MB": round(alg[dom] / 1e6, 1), "GBps": round(fetch_gbs, 1),
                                                 "note": "queue records + visit counters x record sizes (node %d, triangle %d, instance %d B): served by L2 / "
                                                         "Infinity Cache, NOT an HBM rate and not a roofline fraction" % (S_NODE, S_TRI, S_INST)}}
        else:
            entry["roofline"] = {"kernel": "k_" + dom, "bound": "hbm", "alg_MB": round(alg[dom] / 1e6, 1), "ms": round(kms[dom], 2),
                                 "achieved": round(fetch_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                                 "note": "no committed counter pass for this workload: `achieved` is the L2 / Infinity-Cache fetch rate (queue records + "
                                         "visit counters x record sizes), not HBM traffic, so no fraction of the HBM roof is claimed"}
        res[key] = entry
        sc.close()
    return res


if __name__ == "__main__":
    main()
