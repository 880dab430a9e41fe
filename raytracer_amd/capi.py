"""ctypes binding of include/raytracer_amd.h (the C-ABI of libraytracer_amd.so).

The classes mirror the reference's host API for the hot path: ``Context`` takes the place of the
JobSystem (reference src/JobSystem.h:17-19), ``Context.upload_scene`` of ``Scene`` (src/Scene.h:24-34),
``PhotonMap`` of the ``kdnode*`` returned by createPhotonMap / createCausticPhotonMap
(src/PhotonMap.h:28-31) and ``Context.render`` of renderScene's job loop (src/Raytrace.cpp:322-336).
"""
import ctypes as C
import os

import numpy as np

PM_MAP_GLOBAL, PM_MAP_CAUSTIC = 0, 1
PM_FLAG_NO_GLOBAL, PM_FLAG_NO_CAUSTIC, PM_FLAG_NO_DIRECT, PM_FLAG_COUNT_TESTS = 1, 2, 4, 8
PM_K = 63
PM_ROW_BLOCK = 16
PM_EMIT_STATIC_MAX_ATTEMPTS = 131072

# pm_prim / pm_photon as numpy structured dtypes (layouts identical to the C structs)
prim_dtype = np.dtype([
    ("kind", "<i4"), ("pad0", "<i4"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("radius", "<f8"),
    ("dx", "<f8"), ("dy", "<f8"), ("dz", "<f8"), ("min_bound", "<f8"), ("max_bound", "<f8"),
    ("red", "<f4"), ("green", "<f4"), ("blue", "<f4"), ("pad1", "<f4"),
    ("absorb_chance", "<f8"), ("diffuse_chance", "<f8"), ("specular_chance", "<f8"),
    ("transmission_chance", "<f8"), ("refraction", "<f8"), ("specular_coeff", "<f8")])
photon_dtype = np.dtype([
    ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("power", "u1", (4,)),
    ("dx", "<f4"), ("dy", "<f4"), ("dz", "<f4"), ("bounce", "i1"), ("pad", "u1", (3,))])
assert prim_dtype.itemsize == 144 and photon_dtype.itemsize == 32


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("cam", C.c_double * 3),
                ("light_color", C.c_double * 3), ("seed", C.c_uint64), ("flags", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("closest_hit_queries", C.c_uint64), ("shadow_queries", C.c_uint64), ("gathers", C.c_uint64),
                ("absorb_points", C.c_uint64), ("complex_pixels", C.c_uint64),
                ("ms_primary", C.c_float), ("ms_secondary", C.c_float), ("ms_shadow", C.c_float),
                ("ms_gather_global", C.c_float), ("ms_gather_caustic", C.c_float), ("ms_resolve", C.c_float),
                ("ms_total", C.c_float), ("ms_gather", C.c_float), ("ms_group_global", C.c_float), ("ms_group_caustic", C.c_float),
                ("grouped_global", C.c_uint32), ("grouped_caustic", C.c_uint32),
                ("tests_primary_sphere", C.c_uint64), ("tests_primary_plane", C.c_uint64),
                ("tests_shadow_sphere", C.c_uint64), ("tests_shadow_plane", C.c_uint64),
                ("left_global", C.c_uint32), ("left_caustic", C.c_uint32), ("early_points", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PMError(RuntimeError):
    pass


EXPORTS = ["pm_ctx_create", "pm_ctx_destroy", "pm_ctx_set_option", "pm_ctx_set_stream", "pm_ctx_synchronize", "pm_last_error", "pm_scene_upload",
           "pm_emit_range", "pm_emit_range_static", "pm_emit_shard", "pm_map_build", "pm_map_build_shards", "pm_map_create", "pm_map_size", "pm_map_times", "pm_map_download", "pm_map_destroy",
           "pm_knn_query", "pm_knn_profile", "pm_knn_gather", "pm_primary_hits", "pm_intersect_rays", "pm_render",
           "pm_downsample", "pm_render_image"]

_lib = None


def library_path():
    # PM_LIB selects another build of the same library (A/B experiments); default is the in-tree build
    return os.environ.get("PM_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libraytracer_amd.so")


def load_library():
    """Loads libraytracer_amd.so (built in-tree by raytracer_amd/csrc/Makefile).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own copy of the HIP runtime (torch/lib/libamdhip64.so) and this library is linked against
    # ROCm's: whichever is loaded first serves the process, and a second runtime initialised beside it sees no GPU
    # ("No HIP GPUs are available").  Where torch is installed it therefore goes first (the loader then binds this library to
    # torch's copy by soname); without torch the library runs on ROCm's runtime alone.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise PMError(f"{path} is missing: build it with `make -C raytracer_amd/csrc` (hipcc, gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    vp, u64, i64, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32
    sig = {
        "pm_ctx_create": [C.c_int, C.POINTER(vp)], "pm_ctx_destroy": [vp], "pm_ctx_set_stream": [vp, vp], "pm_ctx_synchronize": [vp],
        "pm_ctx_set_option": [vp, C.c_char_p, C.c_double],
        "pm_scene_upload": [vp, vp, i32],
        "pm_emit_range": [vp, i32, u64, C.POINTER(C.c_double), u64, u64, vp, u64, C.POINTER(u64)],
        "pm_emit_range_static": [vp, i32, u64, C.POINTER(C.c_double), u64, u64, vp, u64, C.POINTER(u64)],
        "pm_map_build": [vp, vp, i64, i32, C.POINTER(vp)],
        "pm_map_create": [vp, i32, i64, u64, C.POINTER(C.c_double), C.POINTER(vp), C.POINTER(u64)],
        "pm_emit_shard": [vp, i32, u64, C.POINTER(C.c_double), u64, u64, vp, u64],
        "pm_map_build_shards": [vp, vp, i32, u64, i64, C.POINTER(vp)],
        "pm_map_times": [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)],
        "pm_map_size": [vp, C.POINTER(i64)], "pm_map_download": [vp, vp], "pm_map_destroy": [vp],
        "pm_knn_query": [vp, vp, i64, vp, vp, vp, vp],
        "pm_knn_profile": [vp, vp, i64, vp, vp, vp, vp, vp],
        "pm_knn_gather": [vp, vp, i32, i64, vp, vp, vp, i32],
        "pm_primary_hits": [vp, C.POINTER(RenderParams), i32, i32, vp, vp, vp],
        "pm_render": [vp, vp, vp, C.POINTER(RenderParams), vp, i32, C.POINTER(Stats)],
        "pm_intersect_rays": [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp],
        "pm_downsample": [vp, vp, i32, i32, i32, i32, i32, vp, i32],
        "pm_render_image": [vp, vp, vp, C.POINTER(RenderParams), i32, i32, vp, i32, C.POINTER(Stats)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.pm_last_error.argtypes = []
    lib.pm_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise PMError(f"libraytracer_amd error {rc}: {load_library().pm_last_error().decode()}")


def _vec3(v):
    return (C.c_double * 3)(*[float(t) for t in v])


def default_scene():
    """The 7-object Cornell box of the reference (src/Image.cpp:20-28) as a pm_prim array."""
    s = np.zeros(7, dtype=prim_dtype)

    def mat(p, col, d, sp, t, a):
        p["red"] = np.float32((col >> 16) & 0xFF) / np.float32(255.0)
        p["green"] = np.float32((col >> 8) & 0xFF) / np.float32(255.0)
        p["blue"] = np.float32(col & 0xFF) / np.float32(255.0)
        p["diffuse_chance"], p["specular_chance"], p["transmission_chance"], p["absorb_chance"] = d, sp, t, a

    planes = [((0, -5, 0), 0xFFEEEEEE), ((0, 5, 0), 0xFDEEEEEE),  # src/Image.cpp:22: 9-digit literal truncates
              ((5, 0, 0), 0xFFFF3333), ((-5, 0, 0), 0xFF3333FF), ((0, 0, 10), 0xFFEEEEEE)]
    for i, (pos, col) in enumerate(planes):
        s[i]["kind"] = 1
        s[i]["x"], s[i]["y"], s[i]["z"] = pos
        s[i]["min_bound"], s[i]["max_bound"] = -5, 5
        mat(s[i], col, 0.4, 0.0, 0.0, 1.0)
    for i, (pos, d, sp, t, a, refr) in enumerate([((2, -3.5, 3), 0.0, 0.1, 0.9, 0.0, 2.5), ((-2, -3.5, 5), 0.0, 1.0, 0.0, 0.0, 0.0)]):
        p = s[5 + i]
        p["kind"] = 0
        p["x"], p["y"], p["z"] = pos
        p["radius"] = 1.5
        mat(p, 0xFFFFFFFF, d, sp, t, a)
        p["refraction"] = refr
    return s


class PhotonMap:
    """Opaque photon map handle (stands in for the reference's ``kdnode*`` tree root)."""

    def __init__(self, ctx, handle, kind):
        self.ctx, self.handle, self.kind = ctx, handle, kind

    def __len__(self):
        n = C.c_int64()
        _check(self.ctx.lib.pm_map_size(self.handle, C.byref(n)))
        return n.value

    def times(self):
        """(ms of emission, ms of index build) on the device; waits for the map."""
        a, b = C.c_float(), C.c_float()
        _check(self.ctx.lib.pm_map_times(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def download(self):
        out = np.zeros(len(self), dtype=photon_dtype)
        _check(self.ctx.lib.pm_map_download(self.handle, out.ctypes.data))
        return out

    def destroy(self):  # deleteTree, src/PhotonMap.cpp:504-513
        if self.handle:
            _check(self.ctx.lib.pm_map_destroy(self.handle))
            self.handle = None


class Context:
    """One GPU worker context: ``Context(device)`` replaces ``scheduler::startWorkers(cores)``."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.pm_ctx_create(int(device), C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            _check(self.lib.pm_ctx_destroy(self.handle))
            self.handle = None

    def synchronize(self):
        _check(self.lib.pm_ctx_synchronize(self.handle))

    def set_option(self, name, value):
        """Context switches outside the reference's interface: "debug" (0 / 1), "sp_cap_limit" (tests)."""
        _check(self.lib.pm_ctx_set_option(self.handle, name.encode(), C.c_double(float(value))))

    def set_stream(self, stream_ptr):
        _check(self.lib.pm_ctx_set_stream(self.handle, C.c_void_p(stream_ptr)))

    def upload_scene(self, prims):
        prims = np.ascontiguousarray(prims, dtype=prim_dtype)
        _check(self.lib.pm_scene_upload(self.handle, prims.ctypes.data, len(prims)))

    def emit_range(self, kind, seed, light_color, attempt_begin, attempt_count, d_out_ptr, cap):
        """Emission attempts [begin, begin+count) compacted in attempt order into DEVICE memory."""
        cnt = C.c_uint64()
        _check(self.lib.pm_emit_range(self.handle, kind, seed, _vec3(light_color), attempt_begin, attempt_count,
                                      C.c_void_p(d_out_ptr), cap, C.byref(cnt)))
        return cnt.value

    def emit_range_static(self, kind, seed, light_color, attempt_begin, attempt_count, d_out_ptr, cap):
        """emit_range on the library's static scratch array: allocates nothing, at most PM_EMIT_STATIC_MAX_ATTEMPTS attempts."""
        cnt = C.c_uint64()
        _check(self.lib.pm_emit_range_static(self.handle, kind, seed, _vec3(light_color), attempt_begin, attempt_count,
                                             C.c_void_p(d_out_ptr), cap, C.byref(cnt)))
        return cnt.value

    def emit_shard(self, kind, seed, light_color, attempt_begin, attempt_count, d_shard_ptr, cap):
        """pm_emit_range without a wait, into a shard (32-byte header + cap records) in DEVICE memory."""
        _check(self.lib.pm_emit_shard(self.handle, kind, seed, _vec3(light_color), attempt_begin, attempt_count, C.c_void_p(d_shard_ptr), cap))

    def build_map_shards(self, d_shards_ptr, world, shard_stride, kind, n):
        """createKDTree on the first n records, in rank order, of the all-gathered shards (device memory).  No wait."""
        h = C.c_void_p()
        _check(self.lib.pm_map_build_shards(self.handle, C.c_void_p(d_shards_ptr), world, shard_stride, int(n), C.byref(h)))
        return PhotonMap(self, h, kind)

    def build_map(self, records, kind, n=None, on_device=False):
        """createKDTree on n records: a numpy photon array, or a device pointer with on_device=True."""
        h = C.c_void_p()
        if on_device:
            _check(self.lib.pm_map_build(self.handle, C.c_void_p(records), int(n), 1, C.byref(h)))
        else:
            records = np.ascontiguousarray(records, dtype=photon_dtype)
            _check(self.lib.pm_map_build(self.handle, records.ctypes.data, len(records), 0, C.byref(h)))
        return PhotonMap(self, h, kind)

    def create_map(self, kind, n, seed, light_color=(0.6, 0.6, 0.6), wait=True):
        """createPhotonMap / createCausticPhotonMap (src/PhotonMap.h:28-29).  wait=False: emission and build are left on
        one of the library's build streams and the call returns at once; the first use of the map waits for them."""
        h, used = C.c_void_p(), C.c_uint64()
        _check(self.lib.pm_map_create(self.handle, kind, int(n), int(seed), _vec3(light_color), C.byref(h), C.byref(used) if wait else None))
        m = PhotonMap(self, h, kind)
        m.attempts_issued = used.value if wait else None
        return m

    def knn_query(self, pmap, points):
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        m = len(points)
        idx = np.empty((m, PM_K), dtype=np.int32)
        d2 = np.empty((m, PM_K), dtype=np.float64)
        found = np.empty(m, dtype=np.int32)
        _check(self.lib.pm_knn_query(self.handle, pmap.handle, m, points.ctypes.data, idx.ctypes.data, d2.ctypes.data, found.ctypes.data))
        return found, idx, d2

    def knn_profile(self, pmap, points):
        """knn_query plus per-query work counters [nodes, leaves, sort-merges, insertions]."""
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        m = len(points)
        idx = np.empty((m, PM_K), dtype=np.int32); d2 = np.empty((m, PM_K)); found = np.empty(m, dtype=np.int32)
        work = np.empty((m, 4), dtype=np.uint32)
        _check(self.lib.pm_knn_profile(self.handle, pmap.handle, m, points.ctypes.data, idx.ctypes.data, d2.ctypes.data,
                                       found.ctypes.data, work.ctypes.data))
        return found, idx, d2, work

    def knn_gather(self, pmap, points, normals, kind=None):
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        out = np.empty_like(points)
        _check(self.lib.pm_knn_gather(self.handle, pmap.handle, pmap.kind if kind is None else kind, len(points),
                                      points.ctypes.data, normals.ctypes.data, out.ctypes.data, 0))
        return out

    def knn_gather_device(self, pmap, m, d_points, d_normals, d_radiance, kind=None):
        _check(self.lib.pm_knn_gather(self.handle, pmap.handle, pmap.kind if kind is None else kind, int(m),
                                      C.c_void_p(d_points), C.c_void_p(d_normals), C.c_void_p(d_radiance), 1))

    @staticmethod
    def params(width, height, seed, cam=(0.0, 0.0, -12.0), light_color=(0.6, 0.6, 0.6), flags=0, rank=0, world=1):
        rp = RenderParams()
        rp.width, rp.height, rp.seed, rp.flags, rp.rank, rp.world = width, height, seed, flags, rank, world
        rp.cam = _vec3(cam)
        rp.light_color = _vec3(light_color)
        return rp

    def primary_hits(self, rp, y0=0, y1=None, want_geometry=True):
        y1 = rp.height if y1 is None else y1
        n = (y1 - y0) * rp.width
        ids = np.empty(n, dtype=np.int32)
        pts = np.empty((n, 3), dtype=np.float64) if want_geometry else None
        nrm = np.empty((n, 3), dtype=np.float64) if want_geometry else None
        _check(self.lib.pm_primary_hits(self.handle, C.byref(rp), y0, y1, ids.ctypes.data,
                                        pts.ctypes.data if want_geometry else None, nrm.ctypes.data if want_geometry else None))
        return ids.reshape(y1 - y0, rp.width), pts, nrm

    def intersect_rays(self, src, direction, dt=None, prim=None, exclude=None, res=None, nrm=None):
        """SceneObject::intersect (prim[i] >= 0) / Scene::intersect (prim None or < 0) for n rays.  Returns (hit, res, nrm)."""
        src = np.ascontiguousarray(src, dtype=np.float64).reshape(-1, 3)
        direction = np.ascontiguousarray(direction, dtype=np.float64).reshape(-1, 3)
        n = len(src)
        dt = None if dt is None else np.ascontiguousarray(dt, dtype=np.float64)
        prim = None if prim is None else np.ascontiguousarray(prim, dtype=np.int32)
        exclude = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.int32)
        hit = np.empty(n, dtype=np.int32)
        res = np.zeros((n, 3)) if res is None else np.ascontiguousarray(res, dtype=np.float64)
        nrm = np.zeros((n, 3)) if nrm is None else np.ascontiguousarray(nrm, dtype=np.float64)
        ptr = lambda a: None if a is None else a.ctypes.data
        _check(self.lib.pm_intersect_rays(self.handle, n, src.ctypes.data, direction.ctypes.data, ptr(dt), ptr(prim), ptr(exclude),
                                          hit.ctypes.data, res.ctypes.data, nrm.ctypes.data))
        return hit, res, nrm

    def downsample(self, pane, sample_ratio, flip=True):
        """Host pane [(h*s), (w*s)] uint32 -> host image [h, w] (src/Image.cpp:36-57)."""
        pane = np.ascontiguousarray(pane, dtype=np.uint32)
        h, w = pane.shape[0] // sample_ratio, pane.shape[1] // sample_ratio
        out = np.empty((h, w), dtype=np.uint32)
        _check(self.lib.pm_downsample(self.handle, pane.ctypes.data, 0, w, h, sample_ratio, 1 if flip else 0, out.ctypes.data, 0))
        return out

    def downsample_device(self, d_pane, w, h, sample_ratio, d_out, flip=True):
        _check(self.lib.pm_downsample(self.handle, C.c_void_p(d_pane), 1, w, h, sample_ratio, 1 if flip else 0, C.c_void_p(d_out), 1))

    def render_image(self, gmap, cmap, rp, sample_ratio, flip=True):
        """pm_render into a device pane + pm_downsample: returns (image [h/s, w/s] uint32, stats)."""
        st = Stats()
        out = np.empty((rp.height // sample_ratio, rp.width // sample_ratio), dtype=np.uint32)
        _check(self.lib.pm_render_image(self.handle, gmap.handle if gmap is not None else None, cmap.handle if cmap is not None else None,
                                        C.byref(rp), sample_ratio, 1 if flip else 0, out.ctypes.data, 0, C.byref(st)))
        return out, st

    def render(self, gmap, cmap, rp, pane=None, d_pane=None):
        """Traces this rank's row blocks.  Returns (pane, stats); with d_pane (device pointer) pane is None."""
        st = Stats()
        gh = gmap.handle if gmap is not None else None
        ch = cmap.handle if cmap is not None else None
        if d_pane is not None:
            _check(self.lib.pm_render(self.handle, gh, ch, C.byref(rp), C.c_void_p(d_pane), 1, C.byref(st)))
            return None, st
        if pane is None:
            pane = np.zeros((rp.height, rp.width), dtype=np.uint32)
        _check(self.lib.pm_render(self.handle, gh, ch, C.byref(rp), pane.ctypes.data, 0, C.byref(st)))
        return pane, st
