"""simple-path-tracer hot path, MI355X-native — Python binding over the C ABIs.

Mirrors the reference's host-side surface for the path
(reference src/main.rs:43-66, src/loader/mod.rs:9-31, src/renderer/mod.rs:9-38):

    scene    = load_scene("scene.json")          # loader::load_scene
    renderer = load_renderer("pt.json")          # loader::load_renderer -> PathTracer
    film     = renderer.render(scene, OutputConfig(width, height, output, camera))

`render` runs the hand-written HIP kernels of libspt_hip.so.  There is no CPU
fallback: if the library or a gfx950 device is missing the call raises.
PyTorch is not involved in this module at all (ctypes + numpy only).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.environ.get("SPT_LIB_DIR") or os.path.join(_HERE, "lib")   # (SPT_LIB_DIR: A/B runs against another build, tools/ only)
REPO_ROOT = os.path.dirname(_HERE)

SPT_ABI_VERSION = 13
SPT_LEAF_FLAG = 0x80000000

STATUS_NAMES = {
    0: "SPT_OK", 1: "SPT_ERR_INVALID_ARG", 2: "SPT_ERR_NO_DEVICE", 3: "SPT_ERR_HIP", 4: "SPT_ERR_UNSUPPORTED",
    5: "SPT_ERR_OUT_OF_MEMORY", 100: "SPT_HOST_ERR_IO", 101: "SPT_HOST_ERR_PARSE", 102: "SPT_HOST_ERR_SCHEMA",
    103: "SPT_HOST_ERR_UNSUPPORTED",
}


class SptError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, str(status)), message))
        self.status = status
        self.message = message


# ---- ctypes mirrors of include/spt_abi.h ---------------------------------------------

class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("a", C.c_uint32), ("bmax", C.c_float * 3), ("b", C.c_uint32)]


class TriPos(C.Structure):
    _fields_ = [("p0", C.c_float * 3), ("pad0", C.c_float), ("p1", C.c_float * 3), ("pad1", C.c_float),
                ("p2", C.c_float * 3), ("pad2", C.c_float)]


class TriAttr(C.Structure):
    _fields_ = [("n", (C.c_float * 3) * 3), ("t", (C.c_float * 3) * 3), ("b", (C.c_float * 3) * 3),
                ("uv", (C.c_float * 2) * 3), ("pad", C.c_float * 3)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float)]


class BezierPatch(C.Structure):
    _fields_ = [("cp", ((C.c_float * 4) * 4) * 4)]


class Mesh(C.Structure):
    _fields_ = [("root", C.c_uint32), ("node_count", C.c_uint32), ("tri_first", C.c_uint32), ("tri_count", C.c_uint32)]


class Instance(C.Structure):
    _fields_ = [("inv", C.c_float * 12), ("fwd", C.c_float * 12), ("nrm", C.c_float * 9),
                ("prim_type", C.c_uint32), ("prim_id", C.c_uint32), ("surface", C.c_uint32), ("light", C.c_int32),
                ("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("pad", C.c_float * 5)]


class Material(C.Structure):
    _fields_ = [("bxdf", C.c_uint32), ("c0", C.c_float * 3), ("c1", C.c_float * 3), ("ax", C.c_float),
                ("ay", C.c_float), ("ior", C.c_float), ("c2", C.c_float * 3), ("fresnel", C.c_uint32),
                ("substrate", C.c_uint32), ("recipe", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("image", C.c_uint32),
                ("value", C.c_float * 3), ("mode", C.c_int32), ("wrap", C.c_int32), ("tiling", C.c_float * 3),
                ("offset", C.c_float * 3), ("pad", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("first_level", C.c_uint32), ("n_levels", C.c_uint32)]


class ImageLevel(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("first_texel", C.c_uint32), ("pad", C.c_uint32)]


class MaterialRecipe(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex", C.c_uint32 * 4), ("rough_chan", C.c_uint32), ("metal_chan", C.c_uint32),
                ("ior", C.c_float)]


class Surface(C.Structure):
    _fields_ = [("material", C.c_uint32), ("flags", C.c_uint32), ("inside_medium", C.c_int32),
                ("emissive", C.c_float * 3), ("normal_map", C.c_uint32), ("emissive_map", C.c_uint32)]


class Medium(C.Structure):
    _fields_ = [("sigma_t", C.c_float * 3), ("sigma_s", C.c_float * 3), ("g", C.c_float), ("pad", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("strength", C.c_float * 3),
                ("cos_inner", C.c_float), ("cos_outer", C.c_float), ("instance", C.c_uint32), ("power", C.c_float),
                ("pad", C.c_float * 2)]


class AliasTable(C.Structure):
    _fields_ = [("n", C.c_uint32), ("props", C.POINTER(C.c_float)), ("u", C.POINTER(C.c_float)),
                ("k", C.POINTER(C.c_uint32))]


class Env(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texels", C.POINTER(C.c_float)),
                ("scale", C.c_float * 3), ("alias", AliasTable)]


class PndfTerm(C.Structure):
    _fields_ = [("u", C.c_float * 2), ("s", C.c_float * 2), ("jacobian", C.c_float * 4), ("mat_a", C.c_float * 4),
                ("mat_s", C.c_float * 4), ("mat_mu", C.c_float * 4)]


class PndfNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 4), ("bmax", C.c_float * 4), ("start", C.c_uint32), ("end", C.c_uint32),
                ("lc", C.c_uint32), ("rc", C.c_uint32)]


class Pndf(C.Structure):
    _fields_ = [("first_term", C.c_uint32), ("n_terms", C.c_uint32), ("s_block_count", C.c_uint32), ("first_root", C.c_uint32),
                ("uv_root", C.c_uint32), ("uv_first_ref", C.c_uint32), ("sigma_r", C.c_float), ("sigma_hx", C.c_float),
                ("sigma_hy", C.c_float), ("tiling", C.c_float * 2), ("offset", C.c_float * 2), ("pad", C.c_uint32 * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("aggregate", C.c_uint32),
        ("n_tlas_nodes", C.c_uint32), ("tlas_nodes", C.POINTER(BvhNode)),
        ("n_instances", C.c_uint32), ("instances", C.POINTER(Instance)),
        ("n_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh)),
        ("n_blas_nodes", C.c_uint32), ("blas_nodes", C.POINTER(BvhNode)),
        ("n_tris", C.c_uint32), ("tri_pos", C.POINTER(TriPos)), ("tri_attr", C.POINTER(TriAttr)),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(Sphere)),
        ("n_surfaces", C.c_uint32), ("surfaces", C.POINTER(Surface)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("n_mediums", C.c_uint32), ("mediums", C.POINTER(Medium)),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
        ("light_sampler", C.c_uint32), ("env_light_index", C.c_int32),
        ("light_alias", AliasTable), ("env", Env),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("n_images", C.c_uint32), ("images", C.POINTER(Image)),
        ("n_image_levels", C.c_uint32), ("image_levels", C.POINTER(ImageLevel)),
        ("n_texels", C.c_uint32), ("texels", C.POINTER(C.c_uint32)),
        ("n_material_recipes", C.c_uint32), ("material_recipes", C.POINTER(MaterialRecipe)),
        ("n_bezier_patches", C.c_uint32), ("bezier_patches", C.POINTER(BezierPatch)),
        ("n_pndfs", C.c_uint32), ("pndfs", C.POINTER(Pndf)),
        ("n_pndf_terms", C.c_uint32), ("pndf_terms", C.POINTER(PndfTerm)),
        ("n_pndf_nodes", C.c_uint32), ("pndf_nodes", C.POINTER(PndfNode)),
        ("n_pndf_refs", C.c_uint32), ("pndf_refs", C.POINTER(C.c_uint32)),
        ("n_pndf_roots", C.c_uint32), ("pndf_roots", C.POINTER(C.c_uint32)),
    ]


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("half_cot_half_fov", C.c_float)]


SAMPLER_RANDOM, SAMPLER_JITTERED, SAMPLER_RECURRENCE = 0, 1, 2
RENDER_PROFILE = 1
RENDER_BOX_RADIUS = 2
RENDER_COUNT_VISITS = 4
RENDER_ASYNC = 8
RENDER_DEBUG_NORMAL = 16   # the reference's cargo feature `debug_normal` (Cargo.toml:34-36, pt.rs:113-118)
N_KERNELS = 7
KERNEL_NAMES = ("primary", "shade", "shadow", "extend", "resolve", "other", "shade_first")


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("sampler", C.c_uint32), ("division_x", C.c_uint32), ("division_y", C.c_uint32),
                ("seed", C.c_uint64), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32),
                ("strip_rows", C.c_uint32), ("samples_per_pass", C.c_uint32), ("flags", C.c_uint32),
                ("out_strip_stride", C.c_uint64), ("filter_radius", C.c_float), ("stats_size", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("segments_closest", C.c_uint64), ("segments_shadow", C.c_uint64),
                ("gpu_ms", C.c_double), ("kernel_ms", C.c_double * N_KERNELS),
                ("kernel_launches", C.c_uint32 * N_KERNELS), ("primary_hits", C.c_uint64),
                ("path_vertices", C.c_uint64), ("shadow_first", C.c_uint64), ("vertices_second", C.c_uint64), ("live_samples", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("instance_visits", C.c_uint64), ("node_bytes", C.c_uint64),
                ("class_visits", (C.c_uint64 * 3) * 3)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("instance", "<i4"), ("prim", "<i4"), ("v", "<f4"), ("w", "<f4")])
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("t_min", "<f4"), ("d", "<f4", 3), ("t_max", "<f4")])


# ---- library loading -----------------------------------------------------------------

def _load(name: str) -> C.CDLL:
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise SptError(2 if "hip" in name else 100,
                       "%s is not built (run `make` or `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host_lib: Optional[C.CDLL] = None
_hip_lib: Optional[C.CDLL] = None


def host_lib() -> C.CDLL:
    global _host_lib
    if _host_lib is None:
        lib = _load("libspt_host.so")
        lib.spt_host_last_error.restype = C.c_char_p
        lib.spt_host_scene_set_bezier_newton.argtypes = [C.c_void_p, C.c_int32]
        lib.spt_host_multi_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]
        lib.spt_host_multi_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.spt_host_multi_device_count.argtypes = [C.c_void_p]
        lib.spt_host_multi_device_count.restype = C.c_uint32
        lib.spt_host_multi_destroy.argtypes = [C.c_void_p]
        lib.spt_host_multi_destroy.restype = None
        lib.spt_host_load_scene.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        lib.spt_host_scene_desc.argtypes = [C.c_void_p]
        lib.spt_host_scene_desc.restype = C.POINTER(SceneDesc)
        lib.spt_host_scene_camera.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Camera)]
        lib.spt_host_scene_free.argtypes = [C.c_void_p]
        lib.spt_host_scene_free.restype = None
        lib.spt_host_load_renderer.argtypes = [C.c_char_p, C.POINTER(RenderParams), C.POINTER(C.c_float)]
        lib.spt_host_film_to_rgb8.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        lib.spt_host_film_to_rgb8.restype = None
        lib.spt_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.spt_host_write_image.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.spt_host_write_jpeg.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32]
        lib.spt_host_read_exr.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                          C.POINTER(C.POINTER(C.c_float))]
        lib.spt_host_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.spt_host_catmull_clark.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]
        lib.spt_host_free.argtypes = [C.c_void_p]
        lib.spt_host_free.restype = None
        _host_lib = lib
    return _host_lib


def hip_lib() -> C.CDLL:
    """The product path.  Raises if libspt_hip.so is missing — never falls back."""
    global _hip_lib
    if _hip_lib is None:
        lib = _load("libspt_hip.so")
        lib.spt_last_error.restype = C.c_char_p
        lib.spt_abi_version.restype = C.c_uint32
        got = lib.spt_abi_version()
        if got != SPT_ABI_VERSION:   # the ctypes mirrors above would mis-size every struct: fail before any call uses them
            raise SptError(1, "libspt_hip.so exports ABI version %d, this binding mirrors version %d (rebuild: `make`)" % (got, SPT_ABI_VERSION))
        lib.spt_device_count.argtypes = [C.POINTER(C.c_int32)]
        lib.spt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int32, C.POINTER(C.c_void_p)]
        lib.spt_scene_destroy.argtypes = [C.c_void_p]
        lib.spt_scene_destroy.restype = None
        lib.spt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p,
                                   C.POINTER(RenderStats)]
        lib.spt_render_wait.argtypes = [C.c_void_p]
        lib.spt_shard_rows.argtypes = [C.POINTER(RenderParams), C.POINTER(C.c_uint32)]
        lib.spt_trace_closest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.spt_trace_any.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.spt_alloc_pinned.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        lib.spt_free_pinned.argtypes = [C.c_void_p]
        lib.spt_free_pinned.restype = None
        lib.spt_pin_host.argtypes = [C.c_void_p, C.c_uint64]
        lib.spt_unpin_host.argtypes = [C.c_void_p]
        lib.spt_unpin_host.restype = None
        lib.spt_debug_detmath.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.spt_debug_bxdf.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Material), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _hip_lib = lib
    return _hip_lib


def _check_host(status: int) -> None:
    if status != 0:
        raise SptError(status, host_lib().spt_host_last_error().decode("utf-8", "replace"))


def _check_hip(status: int) -> None:
    if status != 0:
        raise SptError(status, hip_lib().spt_last_error().decode("utf-8", "replace"))


# ---- host side: Scene / renderer loading -----------------------------------------------

class Scene:
    """Flattened scene (reference `Scene`, src/core/scene.rs:9-14) owned by libspt_host."""

    def __init__(self, path: str):
        self._h = C.c_void_p()
        _check_host(host_lib().spt_host_load_scene(os.fspath(path).encode(), C.byref(self._h)))
        self.path = os.fspath(path)
        self._device_scenes = {}

    @property
    def desc(self) -> SceneDesc:
        return host_lib().spt_host_scene_desc(self._h).contents

    def get_camera(self, name=None) -> Camera:
        """Scene::get_camera (src/core/scene.rs): by name, the first one for None; a Camera instance is passed through
        (callers that place their own camera, e.g. tests)."""
        if isinstance(name, Camera):
            return name
        cam = Camera()
        _check_host(host_lib().spt_host_scene_camera(self._h, name.encode() if name else None, C.byref(cam)))
        return cam

    def array(self, field: str) -> np.ndarray:
        """Copy of one desc array as a structured / float numpy array (for tests)."""
        d = self.desc
        table = {
            "tlas_nodes": (d.tlas_nodes, d.n_tlas_nodes, BvhNode), "blas_nodes": (d.blas_nodes, d.n_blas_nodes, BvhNode),
            "instances": (d.instances, d.n_instances, Instance), "meshes": (d.meshes, d.n_meshes, Mesh),
            "tri_pos": (d.tri_pos, d.n_tris, TriPos), "tri_attr": (d.tri_attr, d.n_tris, TriAttr),
            "spheres": (d.spheres, d.n_spheres, Sphere), "surfaces": (d.surfaces, d.n_surfaces, Surface),
            "materials": (d.materials, d.n_materials, Material), "mediums": (d.mediums, d.n_mediums, Medium),
            "lights": (d.lights, d.n_lights, Light),
            "textures": (d.textures, d.n_textures, Texture), "images": (d.images, d.n_images, Image),
            "image_levels": (d.image_levels, d.n_image_levels, ImageLevel), "texels": (d.texels, d.n_texels, C.c_uint32),
            "material_recipes": (d.material_recipes, d.n_material_recipes, MaterialRecipe),
            "bezier_patches": (d.bezier_patches, d.n_bezier_patches, BezierPatch),
            "pndfs": (d.pndfs, d.n_pndfs, Pndf), "pndf_terms": (d.pndf_terms, d.n_pndf_terms, PndfTerm),
            "pndf_nodes": (d.pndf_nodes, d.n_pndf_nodes, PndfNode), "pndf_refs": (d.pndf_refs, d.n_pndf_refs, C.c_uint32),
            "pndf_roots": (d.pndf_roots, d.n_pndf_roots, C.c_uint32),
        }
        ptr, n, ty = table[field]
        if n == 0:
            return np.zeros((0,), dtype=np.dtype(ty))
        buf = C.string_at(ptr, n * C.sizeof(ty))
        return np.frombuffer(buf, dtype=np.dtype(ty)).copy()

    def device_scene(self, device: int = 0) -> "DeviceScene":
        if device not in self._device_scenes:
            self._device_scenes[device] = DeviceScene(self, device)
        return self._device_scenes[device]

    def close(self) -> None:
        for ds in self._device_scenes.values():
            ds.close()
        self._device_scenes.clear()
        if self._h:
            host_lib().spt_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene(path: str, bezier_newton: Optional[bool] = None) -> Scene:
    """loader::load_scene (src/loader/mod.rs:20-31).  bezier_newton: True = the reference built with `--features bezier_ni`
    (patches intersected by Newton's iteration), False = Bezier clipping, None = the process default (clipping unless the
    environment variable SPT_BEZIER_NI is set).  Per scene: two scenes of one process may differ."""
    sc = Scene(path)
    if bezier_newton is not None:
        _check_host(host_lib().spt_host_scene_set_bezier_newton(sc._h, 1 if bezier_newton else 0))
    return sc


class DeviceScene:
    """HBM-resident copy of a Scene (spt_scene_create)."""

    def __init__(self, scene: Scene, device: int = 0):
        self._h = C.c_void_p()
        self.scene = scene
        self.device = device
        self._pinned = {}
        desc = scene.desc
        _check_hip(hip_lib().spt_scene_create(C.byref(desc), device, C.byref(self._h)))

    def film_buffer(self, rows: int, width: int) -> np.ndarray:
        """Reusable page-locked (rows, width, 3) f32 output buffer (spt_alloc_pinned)."""
        key = (rows, width)
        if key not in self._pinned:
            nbytes = max(rows * width * 3 * 4, 4)
            ptr = C.c_void_p()
            _check_hip(hip_lib().spt_alloc_pinned(nbytes, C.byref(ptr)))
            buf = (C.c_float * (rows * width * 3)).from_address(ptr.value)
            self._pinned[key] = (ptr, np.ctypeslib.as_array(buf).reshape(rows, width, 3))
        return self._pinned[key][1]

    def trace_closest(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        _check_hip(hip_lib().spt_trace_closest(self._h, rays.shape[0], rays.ctypes.data, hits.ctypes.data))
        return hits

    def trace_any(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        occ = np.zeros(rays.shape[0], dtype=np.uint8)
        _check_hip(hip_lib().spt_trace_any(self._h, rays.shape[0], rays.ctypes.data, occ.ctypes.data))
        return occ

    def close(self) -> None:
        # an asynchronous frame may still be copying into one of the pinned film buffers: the scene goes first (its destroy
        # drains the render and the copy stream), the buffers after it
        if self._h:
            hip_lib().spt_render_wait(self._h)
            hip_lib().spt_scene_destroy(self._h)
            self._h = C.c_void_p()
        for ptr, _ in self._pinned.values():
            hip_lib().spt_free_pinned(ptr)
        self._pinned.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class OutputConfig:
    """reference src/renderer/mod.rs:9-14"""
    width: int = 512
    height: int = 512
    output_filename: Optional[str] = None
    used_camera_name: Optional[str] = None


def shard_rows(height: int, shard_index: int, shard_count: int, strip_rows: int) -> np.ndarray:
    """Image rows (top = 0) rendered by one shard: interleaved strips of `strip_rows` rows."""
    j = np.arange(height)
    return j[(j // strip_rows) % shard_count == shard_index]


class PathTracer:
    """reference PathTracer{max_depth, pixel_sampler, filter} (src/renderer/pt.rs:24-37)."""

    def __init__(self, max_depth: int = 8, sampler: int = SAMPLER_RECURRENCE, spp: int = 256,
                 division_x: int = 0, division_y: int = 0, filter_radius: float = 0.5, seed: int = 1, debug_normal: bool = False):
        self.max_depth = max_depth
        self.sampler = sampler
        self.spp = spp
        self.division_x = division_x
        self.division_y = division_y
        self.filter_radius = filter_radius
        self.seed = seed
        self.debug_normal = debug_normal   # a build of the reference with `--features debug_normal`: colour = normal * 0.5 + 0.5
        self.last_stats: Optional[RenderStats] = None

    def params(self, width: int, height: int, shard_index: int = 0, shard_count: int = 1, strip_rows: int = 16,
               samples_per_pass: int = 0, flags: int = 0) -> RenderParams:
        p = RenderParams()
        p.width, p.height, p.spp, p.max_depth = width, height, self.spp, self.max_depth
        p.sampler, p.division_x, p.division_y = self.sampler, self.division_x, self.division_y
        p.seed = self.seed
        p.shard_index, p.shard_count, p.strip_rows = shard_index, shard_count, strip_rows
        p.samples_per_pass, p.flags = samples_per_pass, flags | (RENDER_DEBUG_NORMAL if self.debug_normal else 0)
        p.stats_size = C.sizeof(RenderStats)
        if self.filter_radius != 0.5:
            # BoxFilter of any radius (src/filter/boxf.rs): Film::filter_pixel sums the UNWEIGHTED colours of the
            # (2 ceil(radius - 0.5) + 1)^2 pixels around a pixel and divides by the number of those samples whose
            # offset lies within the radius (reference quirk Q1, src/core/film.rs:82-91); 0.5 is the plain mean
            p.flags |= RENDER_BOX_RADIUS
            p.filter_radius = self.filter_radius
        return p

    def render_shard(self, scene: Scene, config: OutputConfig, device: int = 0, shard_index: int = 0,
                     shard_count: int = 1, strip_rows: int = 16, samples_per_pass: int = 0,
                     profile: bool = False, reuse_output: bool = False, film: Optional[np.ndarray] = None,
                     count_visits: bool = False, wait: bool = True) -> np.ndarray:
        """Mean radiance of this shard's rows, shape (rows, width, 3) f32, via the HIP path.
        reuse_output=True returns a page-locked buffer owned by the device scene that the next call
        with the same shape overwrites (no per-call allocation, DMA-speed copy-out).
        count_visits=True runs the counting instantiations of the traversal kernels (last_stats.node_visits, ...).
        wait=False (SPT_RENDER_ASYNC): returns once the work is queued; the returned buffer is valid after `self.wait(scene)`
        or after a later wait=True call on the scene; no stats (last_stats keeps the previous synchronous call's)."""
        ds = scene.device_scene(device)
        cam = scene.get_camera(config.used_camera_name)
        p = self.params(config.width, config.height, shard_index, shard_count, strip_rows, samples_per_pass,
                        (RENDER_PROFILE if profile else 0) | (RENDER_COUNT_VISITS if count_visits else 0) | (0 if wait else RENDER_ASYNC))
        rows = C.c_uint32()
        _check_hip(hip_lib().spt_shard_rows(C.byref(p), C.byref(rows)))
        if film is not None:
            # write this shard's strips in place into a full-image (height, width, 3) f32 film (e.g. SharedFilm.film)
            assert film.shape == (config.height, config.width, 3) and film.dtype == np.float32 and film.flags["C_CONTIGUOUS"]
            row_bytes = config.width * 12
            p.out_strip_stride = shard_count * strip_rows * row_bytes
            stats = RenderStats()
            first = film.ctypes.data + shard_index * strip_rows * row_bytes
            _check_hip(hip_lib().spt_render(ds._h, C.byref(cam), C.byref(p), first if rows.value else film.ctypes.data, C.byref(stats) if wait else None))
            if wait:
                self.last_stats = stats
            return film
        if reuse_output and rows.value:
            out = ds.film_buffer(rows.value, config.width)
        else:
            if not wait and rows.value:
                # a fresh pageable array would be the target of a copy that is still queued when this call returns, with
                # nothing keeping it alive until wait(): the asynchronous form needs a buffer that outlives the call
                # (a shard without rows - more shards than strips - copies nothing: the empty array below is fine)
                raise SptError(1, "render_shard(wait=False) needs reuse_output=True (the scene's pinned buffer) or a caller-owned film=")
            out = np.zeros((rows.value, config.width, 3), dtype=np.float32)
        stats = RenderStats()
        _check_hip(hip_lib().spt_render(ds._h, C.byref(cam), C.byref(p), out.ctypes.data, C.byref(stats) if wait else None))
        if wait:
            self.last_stats = stats
        return out

    def wait(self, scene: Scene, device: int = 0) -> None:
        """spt_render_wait: every render_shard(..., wait=False) queued on the scene has delivered its film."""
        _check_hip(hip_lib().spt_render_wait(scene.device_scene(device)._h))

    def render(self, scene: Scene, config: OutputConfig, device: int = 0) -> np.ndarray:
        """RendererT::render (src/renderer/pt.rs:237-296): full image on one GPU; writes the PNG
        when config.output_filename is set and returns the float film (H, W, 3)."""
        film = self.render_shard(scene, config, device)
        if config.output_filename:
            write_image(config.output_filename, film)
        return film


class DeviceApi(C.Structure):
    """spt_device_api (include/spt_host.h): the device entry points the multi-device fan-out drives."""
    _fields_ = [("scene_create", C.c_void_p), ("scene_destroy", C.c_void_p), ("render", C.c_void_p), ("last_error", C.c_void_p),
                ("pin_host", C.c_void_p), ("unpin_host", C.c_void_p)]


def hip_device_api() -> DeviceApi:
    """The table filled with libspt_hip.so's own functions."""
    lib = hip_lib()
    addr = lambda f: C.cast(f, C.c_void_p).value
    return DeviceApi(addr(lib.spt_scene_create), addr(lib.spt_scene_destroy), addr(lib.spt_render), addr(lib.spt_last_error),
                     addr(lib.spt_pin_host), addr(lib.spt_unpin_host))


class MultiDevice:
    """One call, N devices, one film (spt_host_multi_*): a scene replica and a worker thread per device, interleaved row strips,
    every device's rows DMA-ed straight into the caller's film.  The reference's counterpart is the thread fan-out of
    PathTracer::render (src/renderer/pt.rs:243-287) into one UnsafeFilm (src/core/film.rs:101-116).
    `api` defaults to libspt_hip.so's functions; tests pass stand-ins."""

    def __init__(self, scene: Scene, devices, api: Optional[DeviceApi] = None):
        self.scene = scene
        self.devices = [int(d) for d in devices]
        self._api = api if api is not None else hip_device_api()
        self._h = C.c_void_p()
        desc = scene.desc
        devs = (C.c_int32 * len(self.devices))(*self.devices)
        _check_host(host_lib().spt_host_multi_create(C.byref(desc), C.byref(self._api), len(self.devices), devs, C.byref(self._h)))

    def render(self, renderer: "PathTracer", config: OutputConfig, strip_rows: int = 0, film: Optional[np.ndarray] = None,
               samples_per_pass: int = 0) -> np.ndarray:
        """RendererT::render over all devices: the (H, W, 3) f32 film (a caller-owned `film` is filled in place)."""
        cam = self.scene.get_camera(config.used_camera_name)
        p = renderer.params(config.width, config.height, 0, 1, 16, samples_per_pass)
        if film is None:
            film = np.zeros((config.height, config.width, 3), dtype=np.float32)
        assert film.shape == (config.height, config.width, 3) and film.dtype == np.float32 and film.flags["C_CONTIGUOUS"]
        stats = (RenderStats * len(self.devices))()
        _check_host(host_lib().spt_host_multi_render(self._h, C.byref(cam), C.byref(p), strip_rows, film.ctypes.data, C.byref(stats)))
        self.last_stats = list(stats)
        if config.output_filename:
            write_image(config.output_filename, film)
        return film

    def close(self) -> None:
        if self._h:
            host_lib().spt_host_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_camera(eye, forward, up, fov_degrees: float) -> Camera:
    """PerspectiveCamera::new (src/camera/perspective.rs:15-27) from f32 vectors."""
    f32 = np.float32
    e, f, u = (np.asarray(v, dtype=f32) for v in (eye, forward, up))
    f = f / f32(np.sqrt(f32(np.dot(f, f))))
    r = np.cross(f, u).astype(f32)
    r = r / f32(np.sqrt(f32(np.dot(r, r))))
    u = np.cross(r, f).astype(f32)
    cam = Camera()
    for k in range(3):
        cam.eye[k], cam.forward[k], cam.up[k], cam.right[k] = float(e[k]), float(f[k]), float(u[k]), float(r[k])
    cam.half_cot_half_fov = float(f32(0.5) / f32(np.tan(f32(np.radians(fov_degrees)) * f32(0.5))))
    return cam


def load_renderer(path: str, seed: int = 1) -> PathTracer:
    """loader::load_renderer (src/loader/json.rs:19-51)."""
    p = RenderParams()
    radius = C.c_float()
    _check_host(host_lib().spt_host_load_renderer(os.fspath(path).encode(), C.byref(p), C.byref(radius)))
    return PathTracer(p.max_depth, p.sampler, p.spp, p.division_x, p.division_y, radius.value, seed)


def film_to_rgb8(film: np.ndarray) -> np.ndarray:
    """color_to_rgb (src/core/film.rs:94-99): truncating, no gamma."""
    film = np.ascontiguousarray(film, dtype=np.float32)
    out = np.zeros(film.shape, dtype=np.uint8)
    host_lib().spt_host_film_to_rgb8(film.ctypes.data, film.size // 3, out.ctypes.data)
    return out


def write_png(path: str, film: np.ndarray) -> None:
    rgb8 = film_to_rgb8(film)
    h, w = rgb8.shape[:2]
    _check_host(host_lib().spt_host_write_png(os.fspath(path).encode(), rgb8.ctypes.data, w, h))


def write_image(path: str, film: np.ndarray) -> None:
    """`image.save(path)` (src/renderer/pt.rs:292-294): png or jpg / jpeg by extension."""
    rgb8 = film_to_rgb8(film)
    h, w = rgb8.shape[:2]
    _check_host(host_lib().spt_host_write_image(os.fspath(path).encode(), rgb8.ctypes.data, w, h))


def write_jpeg(path: str, rgb8: np.ndarray, quality: int = 75) -> None:
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[:2]
    _check_host(host_lib().spt_host_write_jpeg(os.fspath(path).encode(), rgb8.ctypes.data, w, h, quality))


def read_png(path: str) -> np.ndarray:
    """(h, w, 4) uint8 RGBA as `image::open` + `get_pixel` present an image texture file (PNG or JPEG)."""
    w, h = C.c_uint32(), C.c_uint32()
    ptr = C.POINTER(C.c_uint32)()
    _check_host(host_lib().spt_host_read_png(os.fspath(path).encode(), C.byref(w), C.byref(h), C.byref(ptr)))
    arr = np.ctypeslib.as_array(ptr, shape=(h.value, w.value)).copy()
    host_lib().spt_host_free(ptr)
    return arr.view(np.uint8).reshape(h.value, w.value, 4)


def read_exr(path: str) -> np.ndarray:
    w, h = C.c_uint32(), C.c_uint32()
    ptr = C.POINTER(C.c_float)()
    _check_host(host_lib().spt_host_read_exr(os.fspath(path).encode(), C.byref(w), C.byref(h), C.byref(ptr)))
    arr = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    host_lib().spt_host_free(ptr)
    return arr


def catmull_clark_patches(ply_path: str, fas_times: int = 4) -> np.ndarray:
    """CatmullClark::load (src/primitive/catmull.rs:93-101): (n, 4, 4, 3) Bezier control points of the subdivision surface."""
    n = C.c_uint32()
    ptr = C.POINTER(C.c_float)()
    _check_host(host_lib().spt_host_catmull_clark(os.fspath(ply_path).encode(), fas_times, C.byref(n), C.byref(ptr)))
    arr = np.ctypeslib.as_array(ptr, shape=(max(n.value, 1), 4, 4, 3))[: n.value].copy()
    host_lib().spt_host_free(ptr)
    return arr


def write_exr(path: str, rgb: np.ndarray) -> None:
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    _check_host(host_lib().spt_host_write_exr(os.fspath(path).encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))


def gather_shards(shard: np.ndarray, height: int, width: int, rank: int, world: int, strip_rows: int, dist=None):
    """Host-side gather of the per-rank row shards into the full (H, W, 3) film on rank 0.

    Shards are disjoint sets of image rows (interleaved strips), so this is a concatenation, not a
    reduction; `dist` is torch.distributed (its CPU/gloo path is used) or None for a single rank.
    Returns the film on rank 0 and None elsewhere.  (The reference has one address space and no
    counterpart; cf. UnsafeFilm, src/core/film.rs:101-116.)"""
    rows_all = [shard_rows(height, r, world, strip_rows) for r in range(world)]
    assert shard.shape == (len(rows_all[rank]), width, 3), (shard.shape, len(rows_all[rank]))
    if world == 1 or dist is None:
        full = np.zeros((height, width, 3), dtype=np.float32)
        full[rows_all[0]] = shard
        return full
    import torch
    max_rows = max(len(r) for r in rows_all)          # gather needs equal shapes: pad short shards
    padded = np.zeros((max_rows, width, 3), dtype=np.float32)
    padded[: shard.shape[0]] = shard
    t = torch.from_numpy(padded)
    if rank == 0:
        bufs = [torch.empty((max_rows, width, 3), dtype=torch.float32) for _ in range(world)]
        dist.gather(t, bufs, dst=0)
        full = np.zeros((height, width, 3), dtype=np.float32)
        for r in range(world):
            full[rows_all[r]] = bufs[r].numpy()[: len(rows_all[r])]
        return full
    dist.gather(t, None, dst=0)
    return None


class SharedFilm:
    """The full (H, W, 3) f32 film in POSIX shared memory, one mapping per rank of a node.

    Every rank writes the image rows of its own shard (disjoint interleaved strips), so assembling the
    image needs no collective and no copy through a socket: it is the multi-process counterpart of the
    reference's UnsafeFilm (src/core/film.rs:101-116), where all render threads write disjoint pixels of one
    film.  Rank 0 creates the segment and passes `name` to the others (e.g. dist.broadcast_object_list)."""

    def __init__(self, height: int, width: int, name: Optional[str] = None, create: bool = False):
        from multiprocessing import shared_memory
        self.height, self.width = height, width
        nbytes = height * width * 3 * 4
        if create:
            self._shm = shared_memory.SharedMemory(create=True, size=nbytes, name=name)
        else:
            self._shm = shared_memory.SharedMemory(name=name)
            # the creating rank owns the segment: keep this process' resource tracker from unlinking it
            try:
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self._owner = create
        self._pinned = False
        self.name = self._shm.name
        self.film = np.ndarray((height, width, 3), dtype=np.float32, buffer=self._shm.buf)

    def pin(self) -> None:
        """Page-lock this process' mapping (spt_pin_host) so that render_shard(film=self.film) copies out at DMA speed."""
        if not self._pinned:
            _check_hip(hip_lib().spt_pin_host(self.film.ctypes.data, self.film.nbytes))
            self._pinned = True

    def write_shard(self, shard: np.ndarray, rank: int, world: int, strip_rows: int) -> None:
        rows = shard_rows(self.height, rank, world, strip_rows)
        assert shard.shape == (len(rows), self.width, 3), (shard.shape, len(rows))
        # rows of one strip are contiguous in both arrays: one memcpy per strip
        k = 0
        while k < len(rows):
            e = k
            while e + 1 < len(rows) and rows[e + 1] == rows[e] + 1:
                e += 1
            self.film[rows[k]:rows[e] + 1] = shard[k:e + 1]
            k = e + 1

    def close(self) -> None:
        if self._pinned:
            hip_lib().spt_unpin_host(self.film.ctypes.data)
            self._pinned = False
        self.film = None
        self._shm.close()
        if self._owner:
            self._shm.unlink()


def device_detmath(fn: int, a: np.ndarray, b: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    """Test seam (spt_debug_detmath): include/spt_detmath.h evaluated on the GPU."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.zeros_like(a)
    _check_hip(hip_lib().spt_debug_detmath(device, fn, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data))
    return out


def device_bxdf_sample(mt: "Material", wo: np.ndarray, rng_state: np.ndarray, scene: Optional[Scene] = None, device: int = 0):
    """Test seam (spt_debug_bxdf op 0): Bxdf::sample of one constant material record on the GPU for n (wo, RNG state) pairs.
    Returns (wi (n, 3), f (n, 3), pdf (n,), dir (n,) 0 reflect / 1 transmit)."""
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    st = np.ascontiguousarray(rng_state, dtype=np.uint64)
    n = wo.shape[0]
    assert st.shape == (n,)
    wi, f, pdf, dr = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    h = scene.device_scene(device)._h if scene is not None else None
    _check_hip(hip_lib().spt_debug_bxdf(h, device, C.byref(mt), 0, n, wo.ctypes.data, None, st.ctypes.data, wi.ctypes.data, f.ctypes.data,
                                        pdf.ctypes.data, dr.ctypes.data))
    return wi, f, pdf, dr


def device_bxdf_eval(mt: "Material", wo: np.ndarray, wi: np.ndarray, scene: Optional[Scene] = None, device: int = 0):
    """Test seam (spt_debug_bxdf op 1): Bxdf::bxdf and Bxdf::pdf on the GPU for n (wo, wi) pairs -> (f (n, 3), pdf (n,))."""
    wo = np.ascontiguousarray(wo, dtype=np.float32).reshape(-1, 3)
    wi = np.ascontiguousarray(wi, dtype=np.float32).reshape(-1, 3)
    n = wo.shape[0]
    assert wi.shape == (n, 3)
    f, pdf = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    h = scene.device_scene(device)._h if scene is not None else None
    _check_hip(hip_lib().spt_debug_bxdf(h, device, C.byref(mt), 1, n, wo.ctypes.data, wi.ctypes.data, None, None, f.ctypes.data, pdf.ctypes.data, None))
    return f, pdf


def device_count() -> int:
    n = C.c_int32()
    _check_hip(hip_lib().spt_device_count(C.byref(n)))
    return n.value
