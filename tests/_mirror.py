"""ctypes view of tests/host_mirror (the kernel's per-path header compiled with g++).  Test-only."""
import ctypes as C
import os
import subprocess

import numpy as np

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "host_mirror", "_build", "librl_mirror.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-s", "-C", os.path.join(HERE, "host_mirror")], check=True)
        L = C.CDLL(SO)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.mirror_scene_create.restype = vp
        L.mirror_scene_create.argtypes = [vp, u32, vp]
        L.mirror_scene_destroy.argtypes = [vp]
        L.mirror_builtin_desc.restype = u32
        L.mirror_builtin_desc.argtypes = [C.c_int, C.c_int, vp, u32, vp]
        L.mirror_render.argtypes = [vp, u32, u32, u64, u32, u64, u64, vp, vp]
        L.mirror_plot.argtypes = [vp, u32, u32, vp, u64]
        L.mirror_prism_fast_check.argtypes = [vp, u64, u64, vp]
        L.mirror_prism_fast_check_paths.argtypes = [vp, u32, u32, u64, u32, u64, u64, vp]
        L.mirror_prism_cylinders.restype = u32
        L.mirror_prism_cylinders.argtypes = [vp, vp, u32]
        L.mirror_cull_counts.argtypes = [vp, u32, u32, u64, u32, u64, u64, vp]
        L.mirror_prism_pairs.restype = u64
        L.mirror_prism_pairs.argtypes = [vp, u64, u64, vp, vp, vp, u64]
        _lib = L
    return _lib


def builtin_desc(which, param=0):
    cam = O.RlCameraDesc()
    n = lib().mirror_builtin_desc(which, param, None, 0, C.byref(cam))
    objs = np.zeros(n, dtype=O.OBJECT_DTYPE)
    lib().mirror_builtin_desc(which, param, O.ptr(objs), n, C.byref(cam))
    return objs, cam


class Scene:
    def __init__(self, objs, cam):
        self.objs = np.ascontiguousarray(objs)
        self.h = lib().mirror_scene_create(O.ptr(self.objs), len(self.objs), C.byref(cam))
        assert self.h

    def __del__(self):
        try:
            lib().mirror_scene_destroy(self.h)
        except Exception:
            pass

    def render(self, w, h, seed, stream, first, n):
        photons = np.zeros(n, dtype=O.PHOTON_DTYPE)
        segs = C.c_uint64(0)
        lib().mirror_render(self.h, w, h, seed, stream, first, n, O.ptr(photons), C.byref(segs))
        return photons, segs.value


def prism_fast_check(scene, trials, seed):
    """rl_hex_prism_fast against rl_hex_prism on random and adversarial (prism, ray) pairs of `scene`:
    {pairs, decided hits, decided misses, undecided, decided-but-different (must be 0), tree hits}."""
    counts = np.zeros(6, dtype=np.uint64)
    lib().mirror_prism_fast_check(scene.h, trials, seed, O.ptr(counts))
    return dict(zip(("pairs", "hits", "misses", "undecided", "wrong", "tree_hits"), (int(c) for c in counts)))


def prism_fast_check_paths(scene, w, h, seed, stream, first, n):
    """The same comparison on the (prism, ray) pairs that paths [first, first + n) produce."""
    counts = np.zeros(6, dtype=np.uint64)
    lib().mirror_prism_fast_check_paths(scene.h, w, h, seed, stream, first, n, O.ptr(counts))
    return dict(zip(("pairs", "hits", "misses", "undecided", "wrong", "tree_hits"), (int(c) for c in counts)))


def prism_pairs(scene, trials, seed):
    """The adversarial (prism, ray) pairs of prism_fast_check as data: prism numbers, rays (n x 6), the tree's {t bits, half-space}."""
    prisms = np.zeros(trials, dtype=np.uint32)
    rays = np.zeros((trials, 6), dtype=np.float32)
    tree = np.zeros((trials, 2), dtype=np.uint32)
    n = lib().mirror_prism_pairs(scene.h, trials, seed, O.ptr(prisms), O.ptr(rays), O.ptr(tree), trials)
    return prisms[:n], rays[:n], tree[:n]


def prism_cylinders(scene, cap=4096):
    """The prisms' second bound as (n, 8) floats {point on the axis, radius, unit axis, 0}; empty when the scene does not use it."""
    out = np.zeros((cap, 8), dtype=np.float32)
    n = lib().mirror_prism_cylinders(scene.h, O.ptr(out), cap)
    return out[:n]


def plot(w, h, photons):
    buffer = np.zeros((h * w, 3), dtype=np.float32)
    photons = np.ascontiguousarray(photons)
    lib().mirror_plot(O.ptr(buffer), w, h, O.ptr(photons), len(photons))
    return buffer


def small_ordered(scene):
    """RlFlatScene::small_ordered of `scene`: the kernel may take `t < best.t` for scene.rs:51's tie rule among the small primitives."""
    L = lib()
    L.mirror_small_ordered.restype = C.c_int
    L.mirror_small_ordered.argtypes = [C.c_void_p]
    return bool(L.mirror_small_ordered(scene.h))


def small_axis_z(scene):
    """RlFlatScene::small_axis_z of `scene`: every paraboloid's, plane's and circle's normal lies along z."""
    L = lib()
    L.mirror_small_axis_z.restype = C.c_int
    L.mirror_small_axis_z.argtypes = [C.c_void_p]
    return bool(L.mirror_small_axis_z(scene.h))


def axis_z_check(seed, n):
    """rl_paraboloid_t<AXIS_Z> / rl_plane_t<AXIS_Z> against the general forms: (cases, hits compared, differences, cases with a zero n.d / n.o)."""
    L = lib()
    L.mirror_axis_z_check.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    c = np.zeros(4, np.uint64)
    L.mirror_axis_z_check(seed, n, O.ptr(c))
    return tuple(int(x) for x in c)


def parab_check(seed, n):
    """The paraboloid's one-division form against the reference's selection: (cases, admitted, hits, disagreements, rejected tiny numerators)."""
    L = lib()
    L.mirror_parab_check.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    c = np.zeros(5, np.uint64)
    L.mirror_parab_check(seed, n, O.ptr(c))
    return tuple(int(x) for x in c)


def cull_counts(scene, w, h, seed, stream, first, n):
    """What the kernel's sphere pass does with `scene`'s cull table on the segments of paths [first, first + n), counted on
    the host: per segment the (group, ray), (cluster, ray) and (member, ray) pairs that pass, and the table's shape."""
    c = np.zeros(8, dtype=np.uint64)
    lib().mirror_cull_counts(scene.h, w, h, seed, stream, first, n, O.ptr(c))
    seg = float(c[0])
    return {"segments": int(c[0]), "groups": int(c[1]), "group_pairs": float(c[2]) / seg, "cluster_pairs": float(c[3]) / seg,
            "member_pairs": float(c[4]) / seg, "clusters": int(c[5]), "clusters_per_group": int(c[6]), "members_per_cluster": int(c[7])}
