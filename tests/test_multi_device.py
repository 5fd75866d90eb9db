"""One call, N devices, one film (spt_host_multi_*, include/spt_host.h): the host-side fan-out that stands where the
thread fan-out of PathTracer::render stands (reference src/renderer/pt.rs:243-287, UnsafeFilm src/core/film.rs:101-116).

CPU half: the device entry points are stand-ins (ctypes callbacks) that render a shard with the CPU oracle - the fan-out,
the shard parameters every worker receives, the strided in-place assembly of the film, the stats layout and the error
paths are the library's own code.  GPU half: the real libspt_hip.so functions, two workers on the box's one device, film
equal to the single-device film bit for bit; and the `spt --devices 0,0` command line."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import _util

spt = _util.load_pkg()

CREATE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p))
DESTROY = C.CFUNCTYPE(None, C.c_void_p)
RENDER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(spt.Camera), C.POINTER(spt.RenderParams), C.c_void_p, C.c_void_p)
LAST_ERROR = C.CFUNCTYPE(C.c_void_p)


class StubDevices:
    """Stand-in for libspt_hip.so: `render` computes the shard with the CPU oracle and writes it where spt_render would."""

    def __init__(self, scene, fail_render_shard=None, fail_create_device=None):
        self.scene = scene
        self.calls = []
        self.created, self.destroyed = [], []
        self.lock = threading.Lock()
        self.fail_render_shard, self.fail_create_device = fail_render_shard, fail_create_device
        self._err = C.create_string_buffer(b"stub: induced failure")
        self._create, self._destroy = CREATE(self.create), DESTROY(self.destroy)
        self._render, self._last = RENDER(self.render), LAST_ERROR(lambda: C.addressof(self._err))
        addr = lambda f: C.cast(f, C.c_void_p).value
        self.api = spt.DeviceApi(addr(self._create), addr(self._destroy), addr(self._render), addr(self._last), None, None)

    def create(self, desc, device, out):
        if device == self.fail_create_device:
            return 2
        with self.lock:
            self.created.append(device)
            out[0] = 1000 + len(self.created)      # an opaque, non-null handle
        return 0

    def destroy(self, handle):
        with self.lock:
            self.destroyed.append(handle)

    def render(self, handle, cam, params, out, stats):
        p = params.contents
        with self.lock:
            self.calls.append(dict(handle=handle, shard_index=p.shard_index, shard_count=p.shard_count, strip_rows=p.strip_rows,
                                   stride=p.out_strip_stride, stats_size=p.stats_size, thread=threading.get_ident(), out=out))
        if p.shard_index == self.fail_render_shard:
            return 3
        rows = spt.shard_rows(p.height, p.shard_index, p.shard_count, p.strip_rows)
        if len(rows) == 0:
            return 0
        lib = _util.oracle_lib()
        shard = np.zeros((len(rows), p.width, 3), dtype=np.float32)
        st = _util.OracleStats()
        desc = self.scene.desc
        assert lib.oracle_render(C.byref(desc), cam, params, 0, 2, shard.ctypes.data, C.byref(st)) == 0
        # spt_render's copy-out: strip s of the shard starts s * out_strip_stride bytes behind rgb_mean_out
        strip_bytes = p.strip_rows * p.width * 12
        flat = shard.reshape(-1)
        for s in range((len(rows) + p.strip_rows - 1) // p.strip_rows):
            n = min(p.strip_rows, len(rows) - s * p.strip_rows) * p.width * 3
            C.memmove(out + s * p.out_strip_stride, flat[s * strip_bytes // 4:].ctypes.data, n * 4)
        if stats:
            rs = spt.RenderStats()
            rs.samples = len(rows) * p.width * p.spp
            rs.gpu_ms = 1.0 + p.shard_index
            C.memmove(stats, C.byref(rs), min(p.stats_size, C.sizeof(rs)))
        return 0


def _scene():
    return spt.load_scene(os.path.join(_util.SCENES, "t_materials.json"))


@pytest.mark.parametrize("n,strip", [(1, 0), (2, 8), (3, 4), (5, 0)])
def test_fan_out_assembles_the_single_device_film(n, strip):
    sc = _scene()
    stub = StubDevices(sc)
    r = spt.PathTracer(max_depth=4, sampler=spt.SAMPLER_RANDOM, spp=2, seed=5)
    w, h = 40, 44
    md = spt.MultiDevice(sc, list(range(n)), api=stub.api)
    film = np.full((h, w, 3), -1.0, dtype=np.float32)
    got = md.render(r, spt.OutputConfig(w, h, used_camera_name="main"), strip_rows=strip, film=film)
    want, _ = _util.oracle_render(sc, r, w, h, camera="main")
    assert got is film and np.array_equal(got.view(np.uint32), want.view(np.uint32))     # every pixel written, exactly once, by its owner
    calls = sorted(stub.calls, key=lambda c: c["shard_index"])
    assert [c["shard_index"] for c in calls] == list(range(n)) and all(c["shard_count"] == n for c in calls)
    used_strip = calls[0]["strip_rows"]
    assert (used_strip == strip) if strip else (1 <= used_strip <= 16 and h >= used_strip * n * 8 or used_strip == 1)
    assert all(c["stride"] == n * used_strip * w * 12 for c in calls)
    assert all(c["out"] == film.ctypes.data + c["shard_index"] * used_strip * w * 12 for c in calls if c["shard_index"] * used_strip < h)
    assert len({c["thread"] for c in calls}) == n and len({c["handle"] for c in calls}) == n     # one worker thread and one replica per device
    assert [s.samples for s in md.last_stats] == [len(spt.shard_rows(h, k, n, used_strip)) * w * 2 for k in range(n)]
    assert sum(s.samples for s in md.last_stats) == w * h * 2
    # a second frame reuses the workers (persistent threads: the same thread ids)
    threads = {c["shard_index"]: c["thread"] for c in calls}
    stub.calls.clear()
    md.render(r, spt.OutputConfig(w, h, used_camera_name="main"), strip_rows=strip, film=film)
    assert {c["shard_index"]: c["thread"] for c in stub.calls} == threads
    md.close()
    assert sorted(stub.destroyed) == sorted(1001 + k for k in range(n))


def test_more_devices_than_strips_and_one_row_images():
    sc = _scene()
    stub = StubDevices(sc)
    r = spt.PathTracer(max_depth=2, sampler=spt.SAMPLER_RANDOM, spp=1, seed=1)
    md = spt.MultiDevice(sc, [0, 1, 2, 3], api=stub.api)
    for w, h, strip in ((9, 1, 16), (5, 3, 2), (7, 2, 0)):
        got = md.render(r, spt.OutputConfig(w, h, used_camera_name="main"), strip_rows=strip)
        want, _ = _util.oracle_render(sc, r, w, h, camera="main")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (w, h, strip)
    md.close()


def test_errors_come_back_with_the_device_that_failed():
    sc = _scene()
    r = spt.PathTracer(max_depth=2, sampler=spt.SAMPLER_RANDOM, spp=1, seed=1)
    stub = StubDevices(sc, fail_render_shard=1)
    md = spt.MultiDevice(sc, [4, 5, 6], api=stub.api)
    with pytest.raises(spt.SptError) as e:
        md.render(r, spt.OutputConfig(16, 16, used_camera_name="main"))
    assert e.value.status == 3 and "device 5" in str(e.value) and "shard 1 of 3" in str(e.value) and "induced" in str(e.value)
    stub.fail_render_shard = None        # the workers survive a failed frame
    got = md.render(r, spt.OutputConfig(16, 16, used_camera_name="main"))
    assert np.isfinite(got).all()
    md.close()
    bad = StubDevices(sc, fail_create_device=1)
    with pytest.raises(spt.SptError) as e:
        spt.MultiDevice(sc, [0, 1, 2], api=bad.api)
    assert e.value.status == 2 and "device 1" in str(e.value)
    assert len(bad.destroyed) == 2 and sorted(bad.created) == [0, 2]          # the replicas that did come up are released
    with pytest.raises(spt.SptError):
        spt.MultiDevice(sc, [], api=bad.api)
    # an asynchronous frame cannot be a complete film on return: refused
    stub2 = StubDevices(sc)
    md = spt.MultiDevice(sc, [0], api=stub2.api)
    p = r.params(8, 8, flags=spt.RENDER_ASYNC)
    cam = sc.get_camera("main")
    film = np.zeros((8, 8, 3), np.float32)
    rc = spt.host_lib().spt_host_multi_render(md._h, C.byref(cam), C.byref(p), 0, film.ctypes.data, None)
    assert rc == 1 and b"ASYNC" in spt.host_lib().spt_host_last_error()
    md.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,camera,devices,strip", [("cfg2_cube.json", None, [0, 0], 0), ("t_materials.json", "main", [0, 0, 0], 4),
                                                             ("t_medium.json", None, [0, 0], 16)])
def test_gpu_two_workers_on_one_device_equal_the_single_device_film(scene_name, camera, devices, strip):
    sc = spt.load_scene(os.path.join(_util.SCENES, scene_name))
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=9)
    cfg = spt.OutputConfig(200, 152, used_camera_name=camera)
    single = r.render_shard(sc, cfg).copy()
    md = spt.MultiDevice(sc, devices)
    for _ in range(2):
        multi = md.render(r, cfg, strip_rows=strip)
        assert np.array_equal(multi.view(np.uint32), single.view(np.uint32))
    assert sum(s.samples for s in md.last_stats) == 200 * 152 * 8 and all(s.gpu_ms > 0 for s in md.last_stats)
    want, _ = _util.oracle_render(sc, r, 200, 152, camera=camera, flags=_util.device_oracle_flags())
    assert np.array_equal(multi.view(np.uint32), want.view(np.uint32))
    md.close()


@pytest.mark.gpu
def test_gpu_cli_devices_flag_writes_the_same_image(tmp_path):
    exe = os.path.join(_util.PKG_DIR, "lib", "spt")
    base = [exe, "-s", os.path.join(_util.SCENES, "cfg2_cube.json"), "-r", os.path.join(_util.SCENES, "pt.json"), "-w", "256", "-h", "192", "--spp", "16"]
    for name, extra in (("one.png", []), ("two.png", ["--devices", "0,0"]), ("three.png", ["--devices", "0,0,0", "--strip-rows", "4"])):
        res = subprocess.run(base + ["-o", str(tmp_path / name)] + extra, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        assert "Finished" in res.stderr
    one = spt.read_png(str(tmp_path / "one.png"))
    assert np.array_equal(one, spt.read_png(str(tmp_path / "two.png"))) and np.array_equal(one, spt.read_png(str(tmp_path / "three.png")))
    assert one[..., :3].max() == 255
    res = subprocess.run(base + ["-o", str(tmp_path / "x.png"), "--gpus", "99"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and "--gpus 99" in res.stderr
