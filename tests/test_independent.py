"""The oracle (and, on a GPU, the HIP path) against tests/golden/independent_paths.npz: MappedPhoton records
computed by tools/independent_paths.py, a vectorised numpy-f32 restatement of the reference written from the
Rust sources only (its own scene construction, geometry with full Intersection records and the recursive
Compound, materials, camera, render_ray loop, its own Philox).  It shares with the build only the definition of
the random numbers and the libm outputs (rl_math.h evaluated element-wise) -- see the script's header.
Bit-for-bit on every photon and on the segment count.  This is not a pin by the reference itself (unbuildable,
unseedable): it removes the single-author risk on the glue of trace_unit.rs:81-168."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "independent_paths.npz")


def cases():
    g = np.load(FIXTURE)
    i = 0
    while "case%d_params" % i in g.files:
        w, h, seed, stream, first, n, which = (int(v) for v in g["case%d_params" % i])
        want = np.zeros(n, dtype=O.PHOTON_DTYPE)
        for k in ("x", "y", "probability", "wavelength"):
            want[k] = g["case%d_%s" % (i, k)]
        yield which, (w, h, seed, stream, first, n), want, int(g["case%d_segments" % i][0])
        i += 1


def scene_desc(R_or_mirror, which):
    """(objects, camera) of fixture scene 0 demo, 1 glass stress, 2 demo with 158 seeds -- from the PRODUCT's own
    generators, which the independent script does not use."""
    return R_or_mirror.builtin_scene_desc(*((0, 0), (1, 0), (0, 158))[which])


def test_the_builds_draws_against_an_rng_header_free_philox_over_a_million_tuples():
    """csrc/rl_rng.h is shared by the product and the oracle, so Random123's four published vectors were the
    only guard on it.  tools/independent_paths.py has its own numpy Philox (checked against the same published 10-round
    vectors at import of this test); the oracle's words -- rl_rng_block through rl_rng.h, RL_PHILOX_ROUNDS rounds -- must
    equal it for 2^20 random (seed, stream, path, block) tuples, path indices beyond 2^32 included."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import independent_paths as ip
    ip._philox_kat()
    rng = np.random.default_rng(2024)
    n_tuples = 0
    for _ in range(16):
        seed = int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(0, 2))
        stream = int(rng.integers(0, 1 << 32))
        n = 1 << 16
        path = rng.integers(0, 1 << 63, n, dtype=np.uint64) >> rng.integers(0, 40, n).astype(np.uint64)
        block = rng.integers(0, 64, n).astype(np.uint32)
        block[::97] = rng.integers(0, 1 << 32, len(block[::97])).astype(np.uint32)
        got = np.zeros((n, 4), dtype=np.uint32)
        O.lib().oracle_rng_blocks(seed, stream, O.ptr(path), O.ptr(block), O.ptr(got), n)
        want = ip.philox4x32_10(path & np.uint64(0xffffffff), path >> np.uint64(32), block.astype(np.uint64),
                                np.full(n, stream, dtype=np.uint64), seed & 0xffffffff, (seed >> 32) & 0xffffffff,
                                rounds=ip.PHILOX_ROUNDS)
        assert np.array_equal(got, np.stack(want, axis=1))
        n_tuples += n
    assert n_tuples == 1 << 20


def test_oracle_matches_the_independent_restatement():
    import robigo_luculenta_amd as R   # host-only generators: no GPU needed
    total, scenes = 0, set()
    for which, (w, h, seed, stream, first, n), want, segs in cases():
        objs, cam = scene_desc(R, which)
        scene = O.Scene(objs.view(O.OBJECT_DTYPE), O.RlCameraDesc.from_buffer_copy(bytes(cam)))
        got, got_segs = scene.render(w, h, seed, stream, first, n, threads=4)
        assert got.tobytes() == want.tobytes(), which
        assert got_segs == segs
        total += n
        scenes.add(which)
    assert total >= 28000 and scenes == {0, 1, 2}


def test_fixture_is_reproducible_from_the_script():
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "independent_paths.py"), "--check"], check=True, timeout=600)


@pytest.mark.gpu
def test_gpu_matches_the_independent_restatement():
    import robigo_luculenta_amd as R
    for fetch in (R.FETCH_LDS, R.FETCH_GLOBAL):
        for which, (w, h, seed, stream, first, n), want, segs in cases():
            scene = R.Scene(*scene_desc(R, which))
            unit = R.TraceUnit(0, w, h, n_photons=n)
            unit.set_fetch(fetch)
            unit.render(scene, seed=seed, stream=stream, first_path_index=first)
            assert unit.mapped_photons.tobytes() == want.tobytes()
            paths, segments, _ = unit.stats()
            assert (paths, segments) == (n, segs)
