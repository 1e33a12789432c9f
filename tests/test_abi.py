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


# ---- the Rust binding against the header, type by type ---------------------------------------------------

C_SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "float": "f32",
             "double": "f64", "char": "c_char", "void": "c_void"}
RUST_ALIASES = {"c_float": "f32", "c_double": "f64"}
RUST_SIZES = {"c_int": (4, 4), "u32": (4, 4), "u64": (8, 8), "i64": (8, 8), "u8": (1, 1), "f32": (4, 4), "f64": (8, 8), "c_char": (1, 1)}


def c_type_to_rust(ctype):
    """'const RlScene*' -> '*const RlScene', 'RlTraceUnit* const*' -> '*const *mut RlTraceUnit', 'uint32_t' -> 'u32'."""
    tokens = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", ctype)
    base = [t for t in tokens if t not in ("const", "*", "struct")][0]
    rust = C_SCALARS.get(base, base)
    # walk the declarator left to right: each '*' is const when followed by 'const' (pointer itself) --
    # the pointee's constness is the 'const' that precedes the '*' (or the base type's for the first level)
    rest = tokens[tokens.index(base) + 1:]
    pointee_const = "const" in tokens[:tokens.index(base)]
    i = 0
    while i < len(rest):
        if rest[i] == "*":
            rust = ("*const " if pointee_const else "*mut ") + rust
            pointee_const = i + 1 < len(rest) and rest[i + 1] == "const"
        i += 1
    return rust


def split_decl(decl):
    """'uint8_t id[RL_COMM_ID_BYTES]' -> ('uint8_t*', 'id') (array parameter = pointer); 'const char* path' -> ('const char*', 'path')."""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?$", decl)
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    return (ctype + "*" if arr else ctype), name, arr


def header_text():
    text = open(os.path.join(ROOT, "include", "robigo_luculenta.h")).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_prototypes():
    protos = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(rl_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header_text(), flags=re.M):
        args = args.strip()
        params = [] if args in ("", "void") else [c_type_to_rust(split_decl(a)[0]) for a in args.split(",")]
        protos[name] = (c_type_to_rust(ret), params)
    return protos


def c_structs():
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", header_text(), flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first, *more = [d.strip() for d in stmt.split(",")]
            ctype, fname, arr = split_decl(first)
            base = ctype[:-1] if arr else ctype
            for nm, ar in [(fname, arr)] + [(split_decl(base + " " + d)[1], split_decl(base + " " + d)[2]) for d in more]:
                rt = c_type_to_rust(base)
                fields.append((nm, "[%s; %s]" % (rt, ar[1:-1].strip()) if ar else rt))
        structs[name] = fields
    return structs


def rust_text():
    text = open(os.path.join(ROOT, "bindings", "rust", "ffi.rs")).read()
    return re.sub(r"//[^\n]*", "", text)


def norm_rust(t):
    t = re.sub(r"\s+", " ", t.strip())
    for a, b in RUST_ALIASES.items():
        t = re.sub(r"\b%s\b" % a, b, t)
    return t


def rust_prototypes():
    protos = {}
    for name, args, ret in re.findall(r"pub fn (rl_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", rust_text()):
        params = [norm_rust(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        protos[name] = (norm_rust(ret) if ret else "()", params)
    return protos


def rust_structs():
    structs = {}
    for attrs, name, body in re.findall(r"((?:#\[[^\]]*\]\s*)*)pub struct (\w+)\s*\{(.*?)\}", rust_text(), flags=re.S):
        assert "repr(C)" in attrs, "%s is not #[repr(C)]" % name
        structs[name] = [(f.split(":", 1)[0].replace("pub", "").strip(), norm_rust(f.split(":", 1)[1]))
                         for f in re.split(r",(?![^\[]*\])", body) if f.strip()]
    return structs


def rust_layout(t, structs, consts):
    """(size, alignment) of a Rust type under #[repr(C)] on x86-64 / any LP64 target."""
    if t.startswith("*"):
        return 8, 8
    m = re.match(r"\[(.+); (\w+)\]", t)
    if m:
        size, align = rust_layout(m.group(1), structs, consts)
        n = int(m.group(2)) if m.group(2).isdigit() else consts[m.group(2)]
        return size * n, align
    if t in RUST_SIZES:
        return RUST_SIZES[t]
    offset, biggest = 0, 1
    for _, ft in structs[t]:
        size, align = rust_layout(ft, structs, consts)
        offset = (offset + align - 1) // align * align + size
        biggest = max(biggest, align)
    return (offset + biggest - 1) // biggest * biggest, biggest


def test_rust_binding_matches_the_header_type_by_type():
    """bindings/rust/ffi.rs (what a maintainer adds as src/ffi.rs) against include/robigo_luculenta.h:
    same functions, same arity, every argument and return type equal under the C -> Rust map (a u32 swapped for a
    u64, a *const for a *mut, a missing argument all fail)."""
    c, r = c_prototypes(), rust_prototypes()
    assert sorted(c) == declared_symbols() and len(c) >= 50
    assert sorted(r) == sorted(c)
    for name in sorted(c):
        assert r[name] == c[name], "%s: header %r, ffi.rs %r" % (name, c[name], r[name])


def test_rust_structs_match_the_header_and_the_ctypes_mirror():
    c, r = c_structs(), rust_structs()
    rust_src = rust_text()
    consts = {k: int(v) for k, v in re.findall(r"pub const (\w+): usize = (\d+);", rust_src)}
    c_consts = {k: int(v) for k, v in re.findall(r"#define (RL_[A-Z_]+) (\d+)", header_text())}
    for k, v in consts.items():
        assert c_consts[k] == v, k
    assert sorted(r) == sorted(c) and len(c) >= 8
    for name in sorted(c):
        assert r[name] == c[name], "%s: header %r, ffi.rs %r" % (name, c[name], r[name])
        mirror = getattr(_lib, name)
        assert [f[0] for f in mirror._fields_] == [f[0] for f in c[name]], name
        assert rust_layout(name, r, consts)[0] == C.sizeof(mirror), name
    # the map itself: a deliberately wrong declaration must be caught
    assert c_type_to_rust("uint32_t") != c_type_to_rust("uint64_t")
    assert c_type_to_rust("RlTraceUnit* const*") == "*const *mut RlTraceUnit"
    assert c_type_to_rust("const RlScene*") == "*const RlScene"


def test_pod_layouts_are_frozen():
    assert C.sizeof(_lib.RlVector3) == 12         # gather_unit.rs:75 transmutes to [u8; 12]
    assert C.sizeof(_lib.RlMappedPhoton) == 16    # trace_unit.rs:23-37
    assert C.sizeof(_lib.RlObjectDesc) == 60 and R.OBJECT_DTYPE.itemsize == 60
    assert C.sizeof(_lib.RlCameraDesc) == 40
    assert C.sizeof(_lib.RlTask) == 12 + 4 * 256


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
    assert b"RL_TASK_MAX_UNITS" in _lib.lib.rl_last_error()
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


def test_diagnostics_live_in_their_own_header_outside_the_boundary():
    """rl_debug_* are exported for tests and tools but are not part of the drop-in boundary: declared in
    include/robigo_luculenta_debug.h only, absent from the product header and from the Rust binding."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "robigo_luculenta_debug.h")).read(), flags=re.S)
    debug = sorted(set(re.findall(r"\b(rl_[a-z0-9_]+)\s*\(", text)))
    assert debug and all(s.startswith("rl_debug_") for s in debug)
    assert sorted(_lib.DEBUG_SIGNATURES) == debug
    for s in debug:
        assert hasattr(_lib.lib, s), "library does not export %s" % s
    assert not [s for s in declared_symbols() if s.startswith("rl_debug_")]
    assert "rl_debug_" not in open(os.path.join(ROOT, "bindings", "rust", "ffi.rs")).read()
