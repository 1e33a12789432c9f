"""robigo_luculenta_amd -- MI355X-native hot path of the spectral path tracer robigo-luculenta.

A thin ctypes mirror of the reference's unit structs over the C ABI (include/robigo_luculenta.h).
Class and method names follow the Rust sources: TraceUnit.render / PlotUnit.plot, clear /
GatherUnit.accumulate, save / TonemapUnit.tonemap / TaskScheduler.get_new_task.  All arithmetic
runs in hand-written gfx950 kernels; numpy is used only to hold downloaded buffers."""
import ctypes as C

import numpy as np

from ._lib import (RlAppConfig, RlAppStats, RlCameraDesc, RlError, RlMappedPhoton, RlObjectDesc, RlSceneDesc, RlTask, RlVector3, check, lib,
                   RL_TASK_MAX_UNITS)

PHOTON_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("probability", "<f4"), ("wavelength", "<f4")])
OBJECT_DTYPE = np.dtype([("surface_kind", "<u4"), ("material_kind", "<u4"), ("v0", "<f4", 3), ("v1", "<f4", 3),
                         ("f", "<f4", 4), ("m", "<f4", 3)])
NUMBER_OF_PHOTONS = 1024 * 512  # trace_unit.rs:67

SCENE_DEMO, SCENE_GLASS_STRESS = 0, 1
FETCH_LDS, FETCH_GLOBAL = 0, 1
TASK_SLEEP, TASK_TRACE, TASK_PLOT, TASK_GATHER, TASK_TONEMAP = range(5)


def device_pci_bus_id(device=0):
    """PCI bus id of a visible device (rl_device_pci_bus_id)."""
    buf = C.create_string_buffer(64)
    check(lib.rl_device_pci_bus_id(device, buf, 64))
    return buf.value.decode()


def device_count():
    return lib.rl_device_count()


def version():
    return lib.rl_version().decode()


def build_id():
    return lib.rl_build_id().decode()


def builtin_scene_desc(which=SCENE_DEMO, param=0):
    """Object array (OBJECT_DTYPE) and camera of a built-in scene (app.rs:166-363)."""
    n = C.c_uint32(0)
    cam = RlCameraDesc()
    lib.rl_scene_builtin_desc(which, param, None, 0, C.byref(n), C.byref(cam))
    if n.value == 0:
        raise RlError(-1, "unknown built-in scene %r" % (which,))
    objs = np.zeros(n.value, dtype=OBJECT_DTYPE)
    check(lib.rl_scene_builtin_desc(which, param, objs.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(cam)))
    return objs, cam


def save_scene_desc(path, objects, camera):
    """Writes a scene description file (RLSC v1, see include/robigo_luculenta.h)."""
    objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    desc = RlSceneDesc(len(objects), objects.ctypes.data_as(C.c_void_p), RlCameraDesc.from_buffer_copy(bytes(camera)))
    check(lib.rl_scene_desc_save(path.encode(), C.byref(desc)))


def load_scene_desc(path):
    n = C.c_uint32(0)
    cam = RlCameraDesc()
    rc = lib.rl_scene_desc_load(path.encode(), None, 0, C.byref(n), C.byref(cam))
    if rc not in (0, -1) or (rc == -1 and n.value == 0):
        check(rc)
    objs = np.zeros(n.value, dtype=OBJECT_DTYPE)
    check(lib.rl_scene_desc_load(path.encode(), objs.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(cam)))
    return objs, cam


class _Handle:
    _destroy = None

    def __init__(self):
        self._h = C.c_void_p()

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            type(self)._destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene(_Handle):
    """scene.rs:23-35; immutable after creation."""
    _destroy = lib.rl_scene_destroy

    def __init__(self, objects, camera, device=0):
        super().__init__()
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        camera = RlCameraDesc.from_buffer_copy(bytes(camera))  # accept any 40-byte camera record
        desc = RlSceneDesc(len(self.objects), self.objects.ctypes.data_as(C.c_void_p), camera)
        check(lib.rl_scene_create(C.byref(desc), device, C.byref(self._h)))
        self.device = device

    @classmethod
    def builtin(cls, which=SCENE_DEMO, param=0, device=0):
        objs, cam = builtin_scene_desc(which, param)
        return cls(objs, cam, device)


class TraceUnit(_Handle):
    """trace_unit.rs:51-168."""
    _destroy = lib.rl_trace_unit_destroy

    def __init__(self, id, width, height, n_photons=NUMBER_OF_PHOTONS, device=0):
        super().__init__()
        check(lib.rl_trace_unit_create(device, id, width, height, n_photons, C.byref(self._h)))
        self.id, self.width, self.height, self.n_photons, self.device = id, width, height, n_photons, device

    def set_fetch(self, fetch):
        check(lib.rl_trace_unit_set_fetch(self._h, fetch))

    def render(self, scene, seed=1, stream=0, first_path_index=0):
        check(lib.rl_trace_unit_render(self._h, scene.handle, seed, stream, first_path_index))

    def render_begin(self, scene, seed=1, stream=0, first_path_index=0):
        """First half of render(): the call is appended to the device's open launch; render_end() waits for it."""
        check(lib.rl_trace_unit_render_begin(self._h, scene.handle, seed, stream, first_path_index))

    def render_fused_begin(self, scene, plot_unit, n_paths, seed=1, stream=0, first_path_index=0):
        check(lib.rl_trace_unit_render_fused_begin(self._h, scene.handle, plot_unit.handle, seed, stream, first_path_index, n_paths))

    def render_end(self):
        check(lib.rl_trace_unit_render_end(self._h))

    def render_async(self, scene, seed=1, stream=0, first_path_index=0):
        check(lib.rl_trace_unit_render_async(self._h, scene.handle, seed, stream, first_path_index))

    def render_fused(self, scene, plot_unit, n_paths, seed=1, stream=0, first_path_index=0):
        check(lib.rl_trace_unit_render_fused(self._h, scene.handle, plot_unit.handle, seed, stream, first_path_index,
                                             n_paths))

    def render_fused_sync(self, scene, plot_unit, n_paths, seed=1, stream=0, first_path_index=0):
        """Blocking fused render; the calls of several threads share open launches (rl_trace_unit_render_fused_sync)."""
        check(lib.rl_trace_unit_render_fused_sync(self._h, scene.handle, plot_unit.handle, seed, stream, first_path_index, n_paths))

    def sync(self):
        check(lib.rl_trace_unit_sync(self._h))

    @property
    def mapped_photons(self):
        out = np.zeros(self.n_photons, dtype=PHOTON_DTYPE)
        check(lib.rl_trace_unit_photons(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self):
        """(paths, segments, kernel_ms) accumulated since creation."""
        p, s, ms = C.c_uint64(0), C.c_uint64(0), C.c_double(0)
        check(lib.rl_trace_unit_stats(self._h, C.byref(p), C.byref(s), C.byref(ms)))
        return p.value, s.value, ms.value


class PlotUnit(_Handle):
    """plot_unit.rs:23-102."""
    _destroy = lib.rl_plot_unit_destroy

    def __init__(self, id, width, height, device=0, external_xyz=None):
        super().__init__()
        check(lib.rl_plot_unit_create(device, id, width, height, C.c_void_p(external_xyz or 0), C.byref(self._h)))
        self.id, self.width, self.height, self.device = id, width, height, device

    def plot(self, trace_units):
        arr = (C.c_void_p * len(trace_units))(*[t.handle for t in trace_units])
        check(lib.rl_plot_unit_plot(self._h, arr, len(trace_units)))

    def clear(self):
        check(lib.rl_plot_unit_clear(self._h))

    def sync(self):
        check(lib.rl_plot_unit_sync(self._h))

    def reduce(self, comm, root=0):
        """The GatherUnit-time exchange: sum of every rank's buffer onto `root` (ncclReduce on the unit's stream)."""
        check(lib.rl_plot_unit_reduce(self._h, comm.handle, root))

    def exchange_stats(self):
        """(exchanges so far, device milliseconds they took): rl_plot_unit_exchange_stats."""
        n, ms = C.c_uint64(0), C.c_double(0.0)
        check(lib.rl_plot_unit_exchange_stats(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def add(self, other):
        """self += other (same device)."""
        check(lib.rl_plot_unit_add(self._h, other.handle))

    def device_buffer(self):
        p = C.c_void_p()
        check(lib.rl_plot_unit_device_buffer(self._h, C.byref(p)))
        return p.value

    @property
    def tristimulus_buffer(self):
        out = np.zeros((self.height * self.width, 3), dtype=np.float32)
        check(lib.rl_plot_unit_download(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def upload(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(self.height * self.width, 3)
        check(lib.rl_plot_unit_upload(self._h, xyz.ctypes.data_as(C.c_void_p)))


class GatherUnit(_Handle):
    """gather_unit.rs:24-92 (resume is explicit: load())."""
    _destroy = lib.rl_gather_unit_destroy

    def __init__(self, width, height, device=0):
        super().__init__()
        check(lib.rl_gather_unit_create(device, width, height, C.byref(self._h)))
        self.width, self.height, self.device = width, height, device

    def accumulate(self, plot_unit):
        """accumulate(&plot.tristimulus_buffer) then plot.clear() (app.rs:143-148)."""
        check(lib.rl_gather_unit_accumulate(self._h, plot_unit.handle))

    def allreduce(self, plot_unit, comm):
        """Task::Gather across ranks: reduce onto rank 0, which accumulates; the others clear (self may be None there)."""
        check(lib.rl_gather_unit_allreduce(self._h, plot_unit.handle, comm.handle))

    def sync(self):
        check(lib.rl_gather_unit_sync(self._h))

    def save(self, path="buffer.raw"):
        check(lib.rl_gather_unit_save(self._h, path.encode()))

    def load(self, path="buffer.raw"):
        check(lib.rl_gather_unit_load(self._h, path.encode()))

    def _download(self):
        t = np.zeros((self.height * self.width, 3), dtype=np.float32)
        c = np.zeros_like(t)
        check(lib.rl_gather_unit_download(self._h, t.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)))
        return t, c

    @property
    def tristimulus_buffer(self):
        return self._download()[0]

    @property
    def compensation_buffer(self):
        return self._download()[1]


def gather_allreduce(gather, plot_unit, comm):
    """rl_gather_unit_allreduce: Task::Gather across the ranks of `comm` (gather may be None on ranks other than 0)."""
    check(lib.rl_gather_unit_allreduce(gather.handle if gather is not None else None, plot_unit.handle, comm.handle))


class Comm(_Handle):
    """One rank of an RCCL communicator (rl_comm_*)."""
    _destroy = lib.rl_comm_destroy

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        check(lib.rl_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, world, rank, device=0):
        super().__init__()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(lib.rl_comm_init_rank(buf, world, rank, device, C.byref(self._h)))
        self.world, self.rank, self.device = world, rank, device

    def info(self):
        """{rank, world (as RCCL counts the communicator), rccl_version, library}: rl_comm_info."""
        rank, world, version = C.c_int(0), C.c_int(0), C.c_int(0)
        path = C.create_string_buffer(512)
        check(lib.rl_comm_info(self._h, C.byref(rank), C.byref(world), C.byref(version), path, 512))
        return {"rank": rank.value, "world": world.value, "rccl_version": version.value, "library": path.value.decode()}

    @classmethod
    def init_all(cls, devices):
        """One process, one rank per distinct device."""
        arr = (C.c_int * len(devices))(*devices)
        out = (C.c_void_p * len(devices))()
        check(lib.rl_comm_init_all(arr, len(devices), out))
        comms = []
        for i, d in enumerate(devices):
            c = cls.__new__(cls)
            _Handle.__init__(c)
            c._h = C.c_void_p(out[i])
            c.world, c.rank, c.device = len(devices), i, d
            comms.append(c)
        return comms


class TonemapUnit(_Handle):
    """tonemap_unit.rs:21-100."""
    _destroy = lib.rl_tonemap_unit_destroy

    def __init__(self, width, height, device=0):
        super().__init__()
        check(lib.rl_tonemap_unit_create(device, width, height, C.byref(self._h)))
        self.width, self.height, self.device = width, height, device

    def tonemap(self, gather_unit):
        check(lib.rl_tonemap_unit_tonemap(self._h, gather_unit.handle))

    @property
    def rgb_buffer(self):
        out = np.zeros((self.height * self.width, 3), dtype=np.uint8)
        check(lib.rl_tonemap_unit_rgb(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def srgb_float(self):
        """(clamped float sRGB before quantisation, max_intensity of find_exposure)."""
        out = np.zeros((self.height * self.width, 3), dtype=np.float32)
        mx = C.c_float(0)
        check(lib.rl_tonemap_unit_srgb_float(self._h, out.ctypes.data_as(C.c_void_p), C.byref(mx)))
        return out, mx.value


class Task:
    """enum Task (task_scheduler.rs:26-41) by unit ids."""

    def __init__(self, kind=TASK_SLEEP, unit=0, units=()):
        self.kind, self.unit, self.units = kind, unit, list(units)

    def _to_c(self):
        t = RlTask()
        t.kind, t.unit, t.n_units = self.kind, self.unit, len(self.units)
        for i, u in enumerate(self.units):
            t.units[i] = u
        return t

    @classmethod
    def _from_c(cls, t):
        return cls(t.kind, t.unit, [t.units[i] for i in range(t.n_units)])

    def __repr__(self):
        names = ["Sleep", "Trace", "Plot", "Gather", "Tonemap"]
        if self.kind == TASK_TRACE:
            return "Trace(%d)" % self.unit
        if self.kind == TASK_PLOT:
            return "Plot(%d, %r)" % (self.unit, self.units)
        if self.kind == TASK_GATHER:
            return "Gather(%r)" % (self.units,)
        return names[self.kind]

    def __eq__(self, other):
        return (self.kind, self.unit, self.units) == (other.kind, other.unit, other.units)


class TaskScheduler(_Handle):
    """task_scheduler.rs:48-325."""
    _destroy = lib.rl_scheduler_destroy

    def __init__(self, concurrency, tonemap_interval_ms=30000):
        super().__init__()
        check(lib.rl_scheduler_create(concurrency, tonemap_interval_ms, C.byref(self._h)))

    def get_new_task(self, completed_task, now_ms=0):
        c, n = completed_task._to_c(), RlTask()
        check(lib.rl_scheduler_get_new_task(self._h, C.byref(c), now_ms, C.byref(n)))
        return Task._from_c(n)

    def performance(self):
        m, s = C.c_float(0), C.c_float(0)
        check(lib.rl_scheduler_performance(self._h, C.byref(m), C.byref(s)))
        return m.value, s.value


def app_run(width, height, max_batches, concurrency=1, device=0, photons_per_batch=NUMBER_OF_PHOTONS, seed=1, stream=0,
            scene=SCENE_DEMO, scene_param=0, tonemap_interval_ms=30000, fused=False, output_ppm=None, checkpoint=None,
            resume=False, verbose=False, sleep_us=0, first_batch=0, devices=None, blocking_trace=False, threads=0):
    """App::new + worker loops (app.rs:54-111) until `max_batches` trace tasks are done, on one GPU or, with
    `devices` = a list of device indices (repeats allowed), on one rank per entry with the plot buffers summed
    onto rank 0 at every gather.  `concurrency` sizes the scheduler's pools, `threads` (0 = concurrency) the host worker pool.
    Returns (rgb image as (H, W, 3) uint8, stats dict)."""
    dev_arr = (C.c_int * len(devices))(*devices) if devices else None
    cfg = RlAppConfig(width, height, device, concurrency, photons_per_batch, seed, stream, scene, scene_param, max_batches,
                      tonemap_interval_ms, int(fused), output_ppm.encode() if output_ppm else None,
                      checkpoint.encode() if checkpoint else None, int(resume), int(verbose), sleep_us, first_batch,
                      len(devices) if devices else 0, int(blocking_trace), dev_arr, int(threads))
    stats = RlAppStats()
    rgb = np.zeros((height, width, 3), dtype=np.uint8)
    check(lib.rl_app_run(C.byref(cfg), C.byref(stats), rgb.ctypes.data_as(C.c_void_p)))
    out = {name: getattr(stats, name) for name, _ in RlAppStats._fields_ if name != "tasks"}
    out["tasks"] = dict(zip(["sleep", "trace", "plot", "gather", "tonemap"], list(stats.tasks)))
    return rgb, out


def batch_histogram(device=0):
    """{k: open launches that carried k blocking render calls} since the library was loaded (waits for running ones)."""
    out = (C.c_uint64 * 257)()
    check(lib.rl_debug_batch_histogram(device, out))
    return {k: int(out[k]) for k in range(1, 257) if out[k]}


def variant_launches():
    """rl_debug_variant_launches: launches per instantiation of the trace kernel since the library was loaded;
    index = 8 * whole scene staged in LDS + 4 * fused + 2 * open launch + 1 * prisms with a second bound; 16 + the low three
    bits for the variants that stage the scene's tables only."""
    out = (C.c_uint64 * 24)()
    check(lib.rl_debug_variant_launches(out))
    return list(out)


def math_probe(fn, x, device=0):
    """Evaluates csrc/rl_math.h function `fn` on the GPU (diagnostics for the parity tests)."""
    names = {"sin": 0, "cos": 1, "tan": 2, "exp": 3, "log": 4, "acos": 5, "sf10": 6, "sqrt": 7, "div": 8, "gamma": 9, "roulette": 10, "normalise": 11, "sin_d": 12, "cos_d": 13, "exp_d": 14, "acos_d": 15, "sqrt_short": 16, "recip_short": 17, "div200_short": 18}
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x)
    check(lib.rl_debug_math_probe(device, names[fn], x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size))
    return y


def math_sweep(fn, lo_bits, hi_bits, both_signs=False, device=0):
    """rl_debug_math_sweep: the short form `fn` ("sqrt_short", "recip_short", "div200_short") against the compiler's IEEE expansion
    on the device for every float with bits in [lo_bits, hi_bits) (and its negative): (mismatches, compared, example bits)."""
    counts = (C.c_uint64 * 2)()
    example = C.c_uint32(0)
    check(lib.rl_debug_math_sweep(device, {"sqrt_short": 16, "recip_short": 17, "div200_short": 18}[fn], int(lo_bits), int(hi_bits),
                                  1 if both_signs else 0, counts, C.byref(example)))
    return int(counts[0]), int(counts[1]), int(example.value)


def app_rank_plan(devices=None, device=0):
    """rl_debug_app_rank_plan: (rank_device, leader, comm_rank, communicator size) of rl_app_run for `devices` (no GPU needed)."""
    n = len(devices) if devices else 0
    m = max(n, 1)
    dev = (C.c_int * n)(*devices) if n else None
    rd, ld, cr, size = (C.c_int * m)(), (C.c_int * m)(), (C.c_int * m)(), C.c_uint32(0)
    check(lib.rl_debug_app_rank_plan(device, dev, n, rd, ld, cr, C.byref(size)))
    return list(rd), list(ld), list(cr), int(size.value)


def prism_probe(scene, prism, rays):
    """rl_debug_prism_probe: the prism shortcut (rl_hex_prism_fast) and the Compound tree, both on the GPU, for `rays`
    (n x 6: origin, direction) against prism number `prism` of the scene's flattened order.  Returns an (n, 5) uint32
    array: status (0 miss, 1 hit, 2 undecided), shortcut {t bits, half-space}, tree {t bits or 0xffffffff, half-space}."""
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((len(rays), 5), dtype=np.uint32)
    check(lib.rl_debug_prism_probe(scene.handle, int(prism), rays.ctypes.data_as(C.c_void_p), len(rays), out.ctypes.data_as(C.c_void_p)))
    return out


def prism_count(scene):
    n = C.c_uint32(0)
    check(lib.rl_debug_prism_count(scene.handle, C.byref(n)))
    return n.value
