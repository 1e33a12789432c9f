"""Loads librobigo_luculenta.so (the C ABI of include/robigo_luculenta.h) with ctypes.

There is no Python or CPU implementation of the hot path behind this module: if the HIP library is
missing the import fails, and if no GPU is visible every compute call raises RlError."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RL_LIBRARY: a diagnostic build of the same library (e.g. `make -C csrc stats`); never a different implementation.
LIB_PATH = os.environ.get("RL_LIBRARY") or os.path.join(HERE, "librobigo_luculenta.so")

RL_TASK_MAX_UNITS = 256
RL_COMM_ID_BYTES = 128


class RlError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("robigo_luculenta error %d: %s" % (code, message))
        self.code = code


class RlVector3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class RlMappedPhoton(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("probability", C.c_float), ("wavelength", C.c_float)]


class RlObjectDesc(C.Structure):
    _fields_ = [("surface_kind", C.c_uint32), ("material_kind", C.c_uint32), ("v0", RlVector3), ("v1", RlVector3),
                ("f0", C.c_float), ("f1", C.c_float), ("f2", C.c_float), ("f3", C.c_float),
                ("m0", C.c_float), ("m1", C.c_float), ("m2", C.c_float)]


class RlCameraDesc(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("phi0", "phi1", "alpha0", "alpha1", "dist0", "dist1", "fov_over_pi",
                                          "focal_factor", "depth_of_field", "chromatic_abberation")]


class RlSceneDesc(C.Structure):
    _fields_ = [("n_objects", C.c_uint32), ("objects", C.c_void_p), ("camera", RlCameraDesc)]


class RlTask(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("unit", C.c_uint32), ("n_units", C.c_uint32),
                ("units", C.c_uint32 * RL_TASK_MAX_UNITS)]


class RlAppConfig(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("device", C.c_int), ("concurrency", C.c_uint32),
                ("photons_per_batch", C.c_uint32), ("seed", C.c_uint64), ("stream", C.c_uint32),
                ("builtin_scene", C.c_int), ("builtin_param", C.c_int), ("max_batches", C.c_uint64),
                ("tonemap_interval_ms", C.c_int64), ("fused", C.c_int), ("output_ppm", C.c_char_p),
                ("checkpoint", C.c_char_p), ("resume", C.c_int), ("verbose", C.c_int), ("sleep_us", C.c_uint32),
                ("first_batch", C.c_uint64), ("n_devices", C.c_uint32), ("blocking_trace", C.c_int), ("devices", C.POINTER(C.c_int)), ("threads", C.c_uint32)]


class RlAppStats(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("paths", C.c_uint64), ("segments", C.c_uint64), ("tasks", C.c_uint64 * 5),
                ("seconds", C.c_double), ("kernel_ms", C.c_double), ("batches_per_sec_mean", C.c_float),
                ("batches_per_sec_stddev", C.c_float), ("tonemaps", C.c_uint32), ("next_batch", C.c_uint64)]


# name -> (restype, argtypes); every symbol include/robigo_luculenta.h declares.
_vp, _u8p = C.c_void_p, C.c_void_p
_i, _u32, _u64, _i64, _f = C.c_int, C.c_uint32, C.c_uint64, C.c_int64, C.c_float
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "rl_last_error": (C.c_char_p, []),
    "rl_device_count": (_i, []),
    "rl_device_pci_bus_id": (_i, [_i, C.c_char_p, _u32]),
    "rl_version": (C.c_char_p, []),
    "rl_build_id": (C.c_char_p, []),
    "rl_scene_builtin_desc": (_i, [_i, _i, _vp, _u32, C.POINTER(_u32), C.POINTER(RlCameraDesc)]),
    "rl_scene_desc_save": (_i, [C.c_char_p, C.POINTER(RlSceneDesc)]),
    "rl_scene_desc_load": (_i, [C.c_char_p, _vp, _u32, C.POINTER(_u32), C.POINTER(RlCameraDesc)]),
    "rl_scene_create": (_i, [C.POINTER(RlSceneDesc), _i, _pp]),
    "rl_scene_destroy": (_i, [_vp]),
    "rl_trace_unit_create": (_i, [_i, _u32, _u32, _u32, _u32, _pp]),
    "rl_trace_unit_destroy": (_i, [_vp]),
    "rl_trace_unit_set_fetch": (_i, [_vp, _i]),
    "rl_trace_unit_render": (_i, [_vp, _vp, _u64, _u32, _u64]),
    "rl_trace_unit_render_begin": (_i, [_vp, _vp, _u64, _u32, _u64]),
    "rl_trace_unit_render_end": (_i, [_vp]),
    "rl_trace_unit_render_async": (_i, [_vp, _vp, _u64, _u32, _u64]),
    "rl_trace_unit_render_fused": (_i, [_vp, _vp, _vp, _u64, _u32, _u64, _u64]),
    "rl_trace_unit_render_fused_sync": (_i, [_vp, _vp, _vp, _u64, _u32, _u64, _u64]),
    "rl_trace_unit_render_fused_begin": (_i, [_vp, _vp, _vp, _u64, _u32, _u64, _u64]),
    "rl_trace_unit_sync": (_i, [_vp]),
    "rl_trace_unit_photons": (_i, [_vp, _vp]),
    "rl_trace_unit_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(C.c_double)]),
    "rl_plot_unit_create": (_i, [_i, _u32, _u32, _u32, _vp, _pp]),
    "rl_plot_unit_destroy": (_i, [_vp]),
    "rl_plot_unit_plot": (_i, [_vp, _pp, _u32]),
    "rl_plot_unit_clear": (_i, [_vp]),
    "rl_plot_unit_sync": (_i, [_vp]),
    "rl_plot_unit_reduce": (_i, [_vp, _vp, _i]),
    "rl_plot_unit_exchange_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "rl_plot_unit_add": (_i, [_vp, _vp]),
    "rl_gather_unit_allreduce": (_i, [_vp, _vp, _vp]),
    "rl_comm_unique_id": (_i, [_vp]),
    "rl_comm_init_rank": (_i, [_vp, _i, _i, _i, _pp]),
    "rl_comm_init_all": (_i, [C.POINTER(_i), _i, _pp]),
    "rl_comm_destroy": (_i, [_vp]),
    "rl_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "rl_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _u32]),
    "rl_comm_group_start": (_i, []),
    "rl_comm_group_end": (_i, []),
    "rl_plot_unit_device_buffer": (_i, [_vp, _pp]),
    "rl_plot_unit_download": (_i, [_vp, _vp]),
    "rl_plot_unit_upload": (_i, [_vp, _vp]),
    "rl_gather_unit_sync": (_i, [_vp]),
    "rl_gather_unit_create": (_i, [_i, _u32, _u32, _pp]),
    "rl_gather_unit_destroy": (_i, [_vp]),
    "rl_gather_unit_accumulate": (_i, [_vp, _vp]),
    "rl_gather_unit_save": (_i, [_vp, C.c_char_p]),
    "rl_gather_unit_load": (_i, [_vp, C.c_char_p]),
    "rl_gather_unit_download": (_i, [_vp, _vp, _vp]),
    "rl_tonemap_unit_create": (_i, [_i, _u32, _u32, _pp]),
    "rl_tonemap_unit_destroy": (_i, [_vp]),
    "rl_tonemap_unit_tonemap": (_i, [_vp, _vp]),
    "rl_tonemap_unit_rgb": (_i, [_vp, _vp]),
    "rl_tonemap_unit_srgb_float": (_i, [_vp, _vp, C.POINTER(_f)]),
    "rl_scheduler_create": (_i, [_u32, _i64, _pp]),
    "rl_scheduler_destroy": (_i, [_vp]),
    "rl_scheduler_get_new_task": (_i, [_vp, C.POINTER(RlTask), _i64, C.POINTER(RlTask)]),
    "rl_scheduler_performance": (_i, [_vp, C.POINTER(_f), C.POINTER(_f)]),
    "rl_app_run": (_i, [C.POINTER(RlAppConfig), C.POINTER(RlAppStats), _vp]),
}
# include/robigo_luculenta_debug.h: diagnostics, not part of the drop-in boundary
DEBUG_SIGNATURES = {
    "rl_debug_math_probe": (_i, [_i, _i, _vp, _vp, _u32]),
    "rl_debug_math_sweep": (_i, [_i, _i, _u32, _u32, _i, _vp, _vp]),
    "rl_debug_app_rank_plan": (_i, [_i, _vp, _u32, _vp, _vp, _vp, C.POINTER(_u32)]),
    "rl_debug_batch_histogram": (_i, [_i, _vp]),
    "rl_debug_variant_launches": (_i, [_vp]),
    "rl_debug_prism_probe": (_i, [_vp, _u32, _vp, _u32, _vp]),
    "rl_debug_prism_count": (_i, [_vp, C.POINTER(_u32)]),
}

if not os.path.exists(LIB_PATH):
    # Nothing is built at import time and there is no CPU implementation to fall back to.
    raise ImportError("robigo_luculenta_amd: %s is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C robigo_luculenta_amd/csrc` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def check(rc):
    if rc != 0:
        raise RlError(rc, lib.rl_last_error().decode("utf-8", "replace"))
