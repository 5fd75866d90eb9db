"""The C-ABI libraries load and export every symbol their headers declare; without a GPU the
product path fails loudly (no CPU fallback behind include/spt_abi.h)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
HIP_SO = os.path.join(spt.LIB_DIR, "libspt_hip.so")


def declared_functions(header):
    text = open(os.path.join(_util.ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spt_[a-z0-9_]+)\s*\(", text)) - {"spt_status"})


def test_host_library_exports_its_header():
    names = declared_functions("spt_host.h")
    assert len(names) >= 11
    lib = spt.host_lib()
    for n in names:
        assert hasattr(lib, n), n


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built (run __graft_entry__.build())")
def test_hip_library_exports_its_header_and_has_gfx950_code():
    names = declared_functions("spt_abi.h")
    assert {"spt_scene_create", "spt_scene_destroy", "spt_render", "spt_shard_rows", "spt_trace_closest", "spt_trace_any",
            "spt_last_error", "spt_abi_version", "spt_device_count"} <= set(names)
    lib = spt.hip_lib()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.spt_abi_version() == spt.SPT_ABI_VERSION
    blob = open(HIP_SO, "rb").read()
    assert b"gfx950" in blob and b"k_primary" in blob and b"k_shade" in blob
    # the Bezier-patch build of the same source, opened by libspt_hip.so for scenes with patches: same exports
    bez = C.CDLL(os.path.join(os.path.dirname(HIP_SO), "libspt_hip_bez.so"))
    for n in names:
        assert hasattr(bez, n), n
    bez.spt_abi_version.restype = C.c_uint32
    assert bez.spt_abi_version() == spt.SPT_ABI_VERSION
    blob = open(os.path.join(os.path.dirname(HIP_SO), "libspt_hip_bez.so"), "rb").read()
    # (the patch routine is inlined, so there is no symbol to look for: the forwarding code is what tells the two apart)
    assert b"gfx950" in blob and b"k_primary" in blob and b"libspt_hip_bez.so could not be loaded" not in blob
    assert b"libspt_hip_bez.so could not be loaded" in open(HIP_SO, "rb").read()


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built")
def test_struct_sizes_match_the_header():
    # sizes the kernels rely on for their 16-byte vector loads
    assert C.sizeof(spt.BvhNode) == 32 and C.sizeof(spt.TriPos) == 48 and C.sizeof(spt.TriAttr) == 144
    assert C.sizeof(spt.Instance) == 192 and C.sizeof(spt.Material) == 64 and C.sizeof(spt.Surface) == 32
    assert C.sizeof(spt.Light) == 64 and C.sizeof(spt.Medium) == 32 and C.sizeof(spt.Sphere) == 16
    assert spt.HIT_DTYPE.itemsize == 20 and spt.RAY_DTYPE.itemsize == 32


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built")
def test_no_gpu_means_loud_failure_not_fallback():
    n = spt.device_count()
    if n > 0:
        pytest.skip("a GPU is visible here")
    sc = spt.load_scene(os.path.join(_util.SCENES, "cfg2_cube.json"))
    with pytest.raises(spt.SptError) as e:
        sc.device_scene(0)
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)      # SPT_ERR_NO_DEVICE
    with pytest.raises(spt.SptError):
        spt.PathTracer(spp=1).render(sc, spt.OutputConfig(8, 8))


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built")
def test_invalid_descriptors_are_rejected_before_touching_a_device():
    lib = spt.hip_lib()
    h = C.c_void_p()
    assert lib.spt_scene_create(None, 0, C.byref(h)) == 1
    d = spt.SceneDesc()
    d.abi_version = 99
    assert lib.spt_scene_create(C.byref(d), 0, C.byref(h)) == 1 and b"abi_version" in lib.spt_last_error()
    sc = spt.load_scene(os.path.join(_util.SCENES, "cfg2_cube.json"))
    good = sc.desc
    bad = spt.SceneDesc.from_buffer_copy(good)
    bad.n_tris = good.n_tris + 5           # BLAS leaves would index past the arrays? no: mesh range check
    bad.tri_pos = None
    assert lib.spt_scene_create(C.byref(bad), 0, C.byref(h)) == 1
    # a scene with Bezier instances is validated by libspt_hip_bez.so (opened by spt_scene_create): same answers
    bz = spt.load_scene(os.path.join(_util.SCENES, "t_bezier.json"))
    bad = spt.SceneDesc.from_buffer_copy(bz.desc)
    bad.n_bezier_patches = 1               # the instances name patches 0 .. 2
    assert lib.spt_scene_create(C.byref(bad), 0, C.byref(h)) == 1 and b"Bezier patch index" in lib.spt_last_error()
    bad = spt.SceneDesc.from_buffer_copy(bz.desc)
    bad.abi_version = 7
    assert lib.spt_scene_create(C.byref(bad), 0, C.byref(h)) == 1 and b"abi_version" in lib.spt_last_error()
    if spt.device_count() == 0:
        assert lib.spt_scene_create(C.byref(bz.desc), 0, C.byref(h)) == 2 and b"no CPU fallback" in lib.spt_last_error()
    rows = C.c_uint32()
    p = spt.PathTracer(spp=2).params(10, 37, 1, 3, 4)
    assert lib.spt_shard_rows(C.byref(p), C.byref(rows)) == 0
    assert rows.value == len(spt.shard_rows(37, 1, 3, 4))


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built")
def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """A caller whose struct mirrors belong to another header version must fail before any call uses them (a stale CLI
    built against a shorter spt_render_stats once had its stack smashed by spt_render)."""
    monkeypatch.setattr(spt, "_hip_lib", None)
    monkeypatch.setattr(spt, "SPT_ABI_VERSION", spt.SPT_ABI_VERSION + 1)
    with pytest.raises(spt.SptError) as e:
        spt.hip_lib()
    assert "ABI version" in str(e.value)
    monkeypatch.undo()
    assert spt.hip_lib().spt_abi_version() == spt.SPT_ABI_VERSION
    # header, python mirror and the struct tail added with v9 agree
    hdr = open(os.path.join(_util.ROOT, "include", "spt_abi.h")).read()
    assert "#define SPT_ABI_VERSION %d" % spt.SPT_ABI_VERSION in hdr
    assert spt.RenderStats.class_visits.offset + 72 == C.sizeof(spt.RenderStats)
    assert spt.RenderParams.stats_size.offset + 4 == C.sizeof(spt.RenderParams)
    # the CLI checks the version of the library it found before doing anything else
    src = open(os.path.join(_util.PKG_DIR, "csrc", "cli", "main.cpp")).read()
    assert "spt_abi_version() != SPT_ABI_VERSION" in src and "params.stats_size" in src


@pytest.mark.gpu
def test_render_never_writes_past_the_callers_stats_struct():
    """spt_render_params::stats_size (ABI v9): a caller compiled against a shorter spt_render_stats gets only that many bytes."""
    lib = spt.hip_lib()
    sc = spt.load_scene(os.path.join(_util.SCENES, "cfg2_cube.json"))
    ds = sc.device_scene(0)
    cam = sc.get_camera(None)
    r = spt.PathTracer(spp=2)
    p = r.params(32, 32)
    out = np.zeros((32, 32, 3), dtype=np.float32)
    full = C.sizeof(spt.RenderStats)
    buf = (C.c_uint8 * (full + 64))()
    stats = C.cast(buf, C.POINTER(spt.RenderStats))
    for short in (full, spt.RenderStats.primary_hits.offset, 24):   # today's struct, the round-1 prefix, three counters only
        C.memset(buf, 0xAB, len(buf))
        p.stats_size = short
        assert lib.spt_render(ds._h, C.byref(cam), C.byref(p), out.ctypes.data, stats) == 0
        assert stats.contents.samples == 32 * 32 * 2
        assert bytes(buf[short:]) == b"\xab" * (len(buf) - short), "spt_render wrote past stats_size = %d" % short
    p.stats_size = 0
    assert lib.spt_render(ds._h, C.byref(cam), C.byref(p), out.ctypes.data, stats) == 1 and b"stats_size" in lib.spt_last_error()
    assert lib.spt_render(ds._h, C.byref(cam), C.byref(p), out.ctypes.data, None) == 0   # no stats: nothing to size


def test_product_sources_never_reference_the_oracle():
    """A product path that routes through the oracle voids every parity claim."""
    bad = []
    for root, _, files in os.walk(os.path.join(_util.ROOT, "simple-path-tracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"liboracle|oracle/|oracle_[a-z]|oracle\.h|import\s+oracle|_util", text):
                    bad.append(os.path.join(root, f))
    assert not bad, bad
    nm = subprocess.run(["nm", "-D", os.path.join(spt.LIB_DIR, "libspt_host.so")], capture_output=True, text=True).stdout
    assert "oracle_" not in nm


@pytest.mark.gpu
def test_async_render_delivers_the_same_film_after_wait():
    """SPT_RENDER_ASYNC (ABI v11): frames queued back to back, each film copied out next to the following frame's kernels;
    after spt_render_wait the buffer holds the last frame, bit-identical to a synchronous render; stats are refused."""
    sc = spt.load_scene(os.path.join(_util.SCENES, "cfg2_cube.json"))
    bz = spt.load_scene(os.path.join(_util.SCENES, "t_bezier.json"))          # forwarded to libspt_hip_bez.so
    for scene, cam in ((sc, None), (bz, "main")):
        cfg = spt.OutputConfig(96, 80, None, cam)
        want = {seed: spt.PathTracer(max_depth=6, spp=8, seed=seed).render_shard(scene, cfg).copy() for seed in (3, 4)}
        r = spt.PathTracer(max_depth=6, spp=8, seed=3)
        buf = r.render_shard(scene, cfg, reuse_output=True)                  # synchronous: allocates the pinned buffer
        for seed in (3, 4, 3, 4):
            r.seed = seed
            out = r.render_shard(scene, cfg, reuse_output=True, wait=False)
            assert out.ctypes.data == buf.ctypes.data
        r.wait(scene)
        assert np.array_equal(out.view(np.uint32), want[4].view(np.uint32))
        # a synchronous render after asynchronous ones: ordered behind their copies
        r.seed = 3
        r.render_shard(scene, cfg, reuse_output=True, wait=False)
        r.seed = 4
        r.render_shard(scene, cfg, reuse_output=True, wait=False)
        r.seed = 3
        got = r.render_shard(scene, cfg, reuse_output=True)
        assert np.array_equal(got.view(np.uint32), want[3].view(np.uint32))
        # different shard shapes in flight one after the other (strided copies into a full-image film)
        film = np.zeros((80, 96, 3), dtype=np.float32)
        for k in range(3):
            r.render_shard(scene, cfg, shard_index=k, shard_count=3, strip_rows=16, film=film, wait=False)
        r.wait(scene)
        assert np.array_equal(film.view(np.uint32), want[3].view(np.uint32))
        # more shards than strips: the shard without rows queues nothing and returns an empty film, asynchronous or not
        # (fuzz seed 9023 of round 3: the binding refused it as "needs a buffer that outlives the call")
        empty = r.render_shard(scene, cfg, shard_index=5, shard_count=6, strip_rows=16, reuse_output=True, wait=False)
        r.wait(scene)
        assert empty.shape == (0, 96, 3)
    lib = spt.hip_lib()
    p = spt.PathTracer(max_depth=2, spp=1).params(16, 16, 0, 1, 16, 0, spt.RENDER_ASYNC)
    out = np.zeros((16, 16, 3), dtype=np.float32)
    st = spt.RenderStats()
    cam = sc.get_camera(None)
    assert lib.spt_render(sc.device_scene(0)._h, C.byref(cam), C.byref(p), out.ctypes.data, C.byref(st)) == 1
    assert b"ASYNC" in lib.spt_last_error()


@pytest.mark.skipif(not os.path.exists(HIP_SO), reason="libspt_hip.so not built")
def test_pndf_trees_are_checked_for_depth_and_sharing():
    """the device's P-NDF walks hold SPT_PNDF_STACK = 32 pending entries and used to drop subtrees silently
    beyond that; a descriptor whose trees share a node made the validation walk exponential.  Both are refused now."""
    lib = spt.hip_lib()
    h = C.c_void_p()
    sc = spt.load_scene(os.path.join(_util.SCENES, "t_pndf.json"))
    good = sc.desc
    nodes = sc.array("pndf_nodes")
    node_t = nodes.dtype
    assert good.n_pndf_nodes == len(nodes) and len(nodes) > 8
    inner = [i for i in range(len(nodes)) if nodes["lc"][i] != 0xffffffff]
    # (1) two parents share a child
    shared = nodes.copy()
    a, b = inner[0], inner[1]
    shared["rc"][a] = shared["rc"][b] if shared["rc"][b] > a else shared["lc"][b]
    bad = spt.SceneDesc.from_buffer_copy(good)
    bad.pndf_nodes = C.cast(shared.ctypes.data, type(bad.pndf_nodes))
    rc = lib.spt_scene_create(C.byref(bad), 0, C.byref(h))
    assert rc == 1 and b"reachable twice" in lib.spt_last_error(), lib.spt_last_error()
    # (2) a chain of 40 inner nodes (each with a leaf as its left child) appended behind the table, hung below a leaf-turned-inner
    import numpy as np
    n0 = len(nodes)
    leaf = next(i for i in range(n0) if nodes["lc"][i] == 0xffffffff)
    chain = np.zeros(2 * 40 + 1, dtype=node_t)
    chain[:] = nodes[leaf]                       # boxes / ranges of a valid leaf
    # node `leaf` -> (n0, n0 + 1); inner n0 + 1 -> (n0 + 2, n0 + 3); ...
    deep = np.concatenate([nodes, chain])
    cur = leaf
    for k in range(40):
        deep["lc"][cur] = n0 + 2 * k
        deep["rc"][cur] = n0 + 2 * k + 1
        cur = n0 + 2 * k + 1
    bad = spt.SceneDesc.from_buffer_copy(good)
    bad.pndf_nodes = C.cast(deep.ctypes.data, type(bad.pndf_nodes))
    bad.n_pndf_nodes = len(deep)
    rc = lib.spt_scene_create(C.byref(bad), 0, C.byref(h))
    assert rc == 4 and b"deeper than the walks' stack" in lib.spt_last_error(), lib.spt_last_error()
    # the untouched descriptor passes validation (and then fails only for want of a device, here)
    rc = lib.spt_scene_create(C.byref(good), 0, C.byref(h))
    assert rc in (0, 2)
    if rc == 0:
        lib.spt_scene_destroy(h)
