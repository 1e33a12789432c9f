"""ctypes view of oracle/_build/librl_oracle.so -- the CPU restatement used as the parity checker.
Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("RL_ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "librl_oracle.so")  # (RL_ORACLE_SO: the clang++ build, test_golden.py)
SO_CLANG = os.path.join(ROOT, "oracle", "_build", "librl_oracle_clang.so")


class RlVector3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class RlObjectDesc(C.Structure):
    _fields_ = [("surface_kind", C.c_uint32), ("material_kind", C.c_uint32), ("v0", RlVector3), ("v1", RlVector3),
                ("f0", C.c_float), ("f1", C.c_float), ("f2", C.c_float), ("f3", C.c_float),
                ("m0", C.c_float), ("m1", C.c_float), ("m2", C.c_float)]


class RlCameraDesc(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("phi0", "phi1", "alpha0", "alpha1", "dist0", "dist1", "fov_over_pi",
                                          "focal_factor", "depth_of_field", "chromatic_abberation")]


PHOTON_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("probability", "<f4"), ("wavelength", "<f4")])
OBJECT_DTYPE = np.dtype([("surface_kind", "<u4"), ("material_kind", "<u4"), ("v0", "<f4", 3), ("v1", "<f4", 3),
                         ("f", "<f4", 4), ("m", "<f4", 3)])
assert OBJECT_DTYPE.itemsize == C.sizeof(RlObjectDesc) == 60

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        build()
    L = C.CDLL(SO)
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    L.oracle_scene_create.restype = vp
    L.oracle_scene_create.argtypes = [vp, u32, vp]
    L.oracle_scene_destroy.argtypes = [vp]
    L.oracle_demo_scene_desc.restype = u32
    L.oracle_demo_scene_desc.argtypes = [C.c_int, vp, u32, vp]
    L.oracle_render.argtypes = [vp, u32, u32, u64, u32, u64, u64, vp, vp]
    L.oracle_render_mt.restype = C.c_double
    L.oracle_render_mt.argtypes = [vp, u32, u32, u64, u32, u64, u64, vp, vp, u32]
    L.oracle_plot.argtypes = [vp, u32, u32, vp, u64]
    L.oracle_accumulate.argtypes = [vp, vp, vp, u64]
    L.oracle_tonemap.argtypes = [vp, u32, u32, vp, vp, vp]
    L.oracle_intersect_object.restype = C.c_int
    L.oracle_intersect_object.argtypes = [vp, u32, vp, vp, vp]
    L.oracle_scene_intersect.restype = C.c_int
    L.oracle_scene_intersect.argtypes = [vp, vp, vp, vp]
    L.oracle_tristimulus.argtypes = [f32, vp]
    L.oracle_sf10_ior.restype = f32
    L.oracle_sf10_ior.argtypes = [f32]
    L.oracle_black_body.restype = f32
    L.oracle_black_body.argtypes = [f32, f32, f32, vp]
    L.oracle_srgb.argtypes = [vp, vp]
    L.oracle_camera.argtypes = [vp, f32, vp]
    L.oracle_material_bounce.restype = C.c_int
    L.oracle_material_bounce.argtypes = [u32, f32, f32, f32, vp, vp, u64, u32, u64, u32, vp]
    L.oracle_philox.argtypes = [vp, vp, vp]
    L.oracle_rng_blocks.argtypes = [C.c_uint64, C.c_uint32, vp, vp, vp, C.c_uint64]
    L.oracle_rng_block.argtypes = [u64, u32, u64, u32, vp]
    L.oracle_math_f32.argtypes = [C.c_int, vp, vp, u64]
    L.oracle_powf.argtypes = [vp, f32, vp, u64]
    L.oracle_exp_f64.argtypes = [vp, vp, u64]
    _lib = L
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def demo_scene_desc(seeds=0):
    """(objects ndarray[OBJECT_DTYPE], RlCameraDesc) of the oracle's own restatement of app.rs:166-363."""
    L = lib()
    cam = RlCameraDesc()
    n = L.oracle_demo_scene_desc(seeds, None, 0, C.byref(cam))
    objs = np.zeros(n, dtype=OBJECT_DTYPE)
    L.oracle_demo_scene_desc(seeds, ptr(objs), n, C.byref(cam))
    return objs, cam


class Scene:
    def __init__(self, objs, cam):
        self.objs = np.ascontiguousarray(objs)
        self.cam = cam
        self.h = lib().oracle_scene_create(ptr(self.objs), len(self.objs), C.byref(cam))
        assert self.h

    def __del__(self):
        try:
            lib().oracle_scene_destroy(self.h)
        except Exception:
            pass

    def render(self, w, h, seed, stream, first, n, threads=1):
        photons = np.zeros(n, dtype=PHOTON_DTYPE)
        segs = C.c_uint64(0)
        if threads <= 1:
            lib().oracle_render(self.h, w, h, seed, stream, first, n, ptr(photons), C.byref(segs))
        else:
            lib().oracle_render_mt(self.h, w, h, seed, stream, first, n, ptr(photons), C.byref(segs), threads)
        return photons, segs.value

    def intersect_object(self, index, origin, direction):
        o = np.asarray(origin, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        out = np.zeros(10, dtype=np.float32)
        hit = lib().oracle_intersect_object(self.h, index, ptr(o), ptr(d), ptr(out))
        return out if hit else None

    def intersect(self, origin, direction):
        o = np.asarray(origin, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        out = np.zeros(10, dtype=np.float32)
        idx = lib().oracle_scene_intersect(self.h, ptr(o), ptr(d), ptr(out))
        return idx, out


def plot(w, h, photons, buffer=None):
    if buffer is None:
        buffer = np.zeros((h * w, 3), dtype=np.float32)
    photons = np.ascontiguousarray(photons)
    lib().oracle_plot(ptr(buffer), w, h, ptr(photons), len(photons))
    return buffer


def accumulate(acc, comp, px):
    lib().oracle_accumulate(ptr(acc), ptr(comp), ptr(np.ascontiguousarray(px)), len(acc))


def tonemap(xyz, w, h):
    rgb = np.zeros((h * w, 3), dtype=np.uint8)
    srgb = np.zeros((h * w, 3), dtype=np.float32)
    mx = C.c_float(0)
    lib().oracle_tonemap(ptr(np.ascontiguousarray(xyz)), w, h, ptr(rgb), ptr(srgb), C.byref(mx))
    return rgb, srgb, mx.value


def math_f32(fn, x):
    names = {"sin": 0, "cos": 1, "tan": 2, "exp": 3, "log": 4, "acos": 5, "closed01": 6, "halfopen01": 7, "sin_d": 8, "cos_d": 9, "exp_d": 10, "acos_d": 11, "sf10": 12}
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x)
    lib().oracle_math_f32(names[fn], ptr(x), ptr(y), x.size)
    return y
