"""The C-ABI library: loads without a GPU, exports every symbol include/robigo_luculenta.h declares,
refuses compute without a device (no CPU fallback), and its host-only entry points work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import robigo_luculenta_amd as R  # a missing HIP library is a failure, never a skip
from robigo_luculenta_amd import _lib  # noqa: E402


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "robigo_luculenta.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(_lib.lib, s), "library does not export %s" % s
        assert s in _lib.SIGNATURES, "python binding lacks %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_rust_binding_declares_every_entry_point():
    """bindings/rust/ffi.rs (what a maintainer adds as src/ffi.rs) must not drift from the header."""
    rust = open(os.path.join(ROOT, "bindings", "rust", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (rl_[a-z0-9_]+)\s*\(", rust))
    assert declared == set(declared_symbols())


def test_pod_layouts_are_frozen():
    assert C.sizeof(_lib.RlVector3) == 12         # gather_unit.rs:75 transmutes to [u8; 12]
    assert C.sizeof(_lib.RlMappedPhoton) == 16    # trace_unit.rs:23-37
    assert C.sizeof(_lib.RlObjectDesc) == 60 and R.OBJECT_DTYPE.itemsize == 60
    assert C.sizeof(_lib.RlCameraDesc) == 40
    assert C.sizeof(_lib.RlTask) == 12 + 4 * 64


def test_builtin_scene_desc_is_host_only_and_matches_golden():
    objs, cam = R.builtin_scene_desc(R.SCENE_DEMO)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "demo_scene_desc.npz"))
    assert objs.tobytes() == gold["objects"].tobytes()
    assert bytes(cam) == gold["camera"].tobytes()
    n = C.c_uint32(0)
    small = np.zeros(10, dtype=R.OBJECT_DTYPE)
    rc = _lib.lib.rl_scene_builtin_desc(0, 0, small.ctypes.data_as(C.c_void_p), 10, C.byref(n), None)
    assert rc == -1 and n.value == 339 and b"too small" in _lib.lib.rl_last_error()
    assert _lib.lib.rl_scene_builtin_desc(99, 0, None, 0, None, None) == -1


@pytest.mark.skipif(R.device_count() > 0, reason="checks the no-GPU behaviour")
def test_compute_entry_points_fail_loudly_without_a_device():
    objs, cam = R.builtin_scene_desc()
    with pytest.raises(R.RlError) as e:
        R.Scene(objs, cam)
    assert e.value.code == -2  # RL_E_NO_DEVICE: there is no CPU implementation to fall back to
    for ctor in (lambda: R.TraceUnit(0, 64, 36), lambda: R.PlotUnit(0, 64, 36), lambda: R.GatherUnit(64, 36),
                 lambda: R.TonemapUnit(64, 36)):
        with pytest.raises(R.RlError):
            ctor()


def test_app_reports_errors_with_a_message():
    """rl_app_run runs its tasks on worker threads; the message of whatever failed must reach the caller's
    rl_last_error() (per thread), and argument errors must carry one too."""
    with pytest.raises(R.RlError) as e:
        R.app_run(0, 36, 1)
    assert "zero" in str(e.value)
    if R.device_count() == 0:
        with pytest.raises(R.RlError) as e:
            R.app_run(64, 36, 1)
        assert "device" in str(e.value).lower()


def test_invalid_arguments_return_codes_not_crashes():
    h = C.c_void_p()
    assert _lib.lib.rl_scheduler_create(0, 30000, C.byref(h)) == -1
    assert _lib.lib.rl_scheduler_create(100, 30000, C.byref(h)) == -1   # 300 trace units > RL_TASK_MAX_UNITS
    assert _lib.lib.rl_trace_unit_render(None, None, 1, 0, 0) == -1
    assert _lib.lib.rl_gather_unit_accumulate(None, None) == -1
    assert _lib.lib.rl_scene_destroy(None) == 0 and _lib.lib.rl_trace_unit_destroy(None) == 0
    bad = np.zeros(1, dtype=R.OBJECT_DTYPE)
    bad["surface_kind"] = 9
    desc = _lib.RlSceneDesc(1, bad.ctypes.data_as(C.c_void_p), R.builtin_scene_desc()[1])
    assert _lib.lib.rl_scene_create(C.byref(desc), 0, C.byref(h)) == -1
    assert b"surface" in _lib.lib.rl_last_error()


def test_scene_description_file_round_trip(tmp_path):
    objs, cam = R.builtin_scene_desc(R.SCENE_GLASS_STRESS)
    path = str(tmp_path / "scene.rlsc")
    R.save_scene_desc(path, objs, cam)
    assert os.path.getsize(path) == 12 + 40 + 60 * len(objs)
    objs2, cam2 = R.load_scene_desc(path)
    assert objs2.tobytes() == objs.tobytes() and bytes(cam2) == bytes(cam)
    open(path, "r+b").write(b"XXXX")
    with pytest.raises(R.RlError):
        R.load_scene_desc(path)
    with pytest.raises(R.RlError):
        R.load_scene_desc(str(tmp_path / "missing.rlsc"))
