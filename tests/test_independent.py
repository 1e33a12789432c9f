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
