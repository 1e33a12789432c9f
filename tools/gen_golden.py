#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/rl_oracle.cpp).

The reference cannot be run here (Rust, no toolchain) and is unseedable anyway (monte_carlo.rs:22),
so these fixtures pin the BUILD's oracle against regressions and give the GPU tests inputs/outputs
that do not need the oracle at hand; the oracle itself is pinned by tests/test_oracle_kat.py.
Re-run after an intended change of the oracle, the RNG slot map or rl_math.h."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
objs, cam = O.demo_scene_desc()
scene = O.Scene(objs, cam)

# 1. per-photon records of the demo scene for three (seed, stream, first) triples
cases = [(1, 0, 0), (2, 3, 5_000_000_000), (7, 1, 123456)]
photons = [scene.render(1280, 720, s, st, f, 4096)[0] for (s, st, f) in cases]
segments = [scene.render(1280, 720, s, st, f, 4096)[1] for (s, st, f) in cases]
np.savez_compressed(os.path.join(out, "demo_photons.npz"), cases=np.array(cases, dtype=np.uint64),
                    photons=np.stack(photons), segments=np.array(segments, dtype=np.uint64))

# 2. a small image: 64x36, 2^20 paths -> XYZ (oracle plot), Kahan gather of two halves, tonemap
W, H, N = 64, 36, 1 << 20
ph, _ = scene.render(W, H, 1, 0, 0, N, threads=8)
acc = np.zeros((W * H, 3), np.float32)
comp = np.zeros_like(acc)
O.accumulate(acc, comp, O.plot(W, H, ph[: N // 2]))
O.accumulate(acc, comp, O.plot(W, H, ph[N // 2:]))
rgb, srgb, mx = O.tonemap(acc, W, H)
np.savez_compressed(os.path.join(out, "demo_image_64x36.npz"), width=W, height=H, n_paths=N, seed=1, stream=0,
                    xyz=acc, compensation=comp, srgb=srgb, rgb=rgb, max_intensity=np.float32(mx))

# 3. scene description of the demo scene (what rl_scene_builtin_desc must reproduce)
np.savez_compressed(os.path.join(out, "demo_scene_desc.npz"), objects=objs, camera=np.frombuffer(bytes(cam), dtype=np.float32))
print("golden fixtures written to", out)
