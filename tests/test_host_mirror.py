"""The kernel's per-path header (csrc/rl_core.h: flat records, sphere clusters, prism culls) compiled
with g++ must be bit-identical to the literal oracle -- this is the no-GPU half of the parity proof;
tests/test_gpu_parity.py repeats it with the hipcc build on the device."""
import ctypes as C

import numpy as np
import pytest

import _mirror as M
import _oracle as O


def test_builtin_scene_desc_matches_oracle_restatement():
    """The product's App::set_up_scene (rl_scene.cpp) against the oracle's own (rl_oracle.cpp)."""
    for seeds in (0, 158, 7):
        oo, oc = O.demo_scene_desc(seeds)
        mo, mc = M.builtin_desc(0, seeds)
        assert oo.tobytes() == mo.tobytes()
        assert bytes(oc) == bytes(mc)


def test_glass_stress_scene_inventory():
    objs, cam = M.builtin_desc(1)
    assert len(objs) == 7 + 66
    assert list(np.bincount(objs["surface_kind"], minlength=5)) == [1, 1, 2, 3, 66]
    assert (objs[7:]["material_kind"] == 4).all()
    # three rings of 11 x 2 prisms (radius 10 / 17 / 24; the second variant of each pair stands at
    # 1.2 x the radius, app.rs:291-296): six distinct distances from the axis, 11 prisms each
    r = np.hypot(objs[7:]["v1"][:, 0].astype(np.float64), objs[7:]["v1"][:, 1])
    groups = np.unique(np.round(r, 2), return_counts=True)
    assert len(groups[0]) == 6 and (groups[1] == 11).all()


@pytest.mark.parametrize("which,param,n", [(0, 0, 120000), (1, 0, 60000), (0, 158, 60000), (0, 3, 60000)])
def test_core_bit_exact_vs_oracle(which, param, n):
    objs, cam = M.builtin_desc(which, param)
    so, sm = O.Scene(objs, cam), M.Scene(objs, cam)
    want, segs = so.render(1280, 720, 11, 2, 10_000_000_000, n, threads=8)
    got, segs2 = sm.render(1280, 720, 11, 2, 10_000_000_000, n)
    assert segs == segs2
    assert got.tobytes() == want.tobytes()


def test_core_bit_exact_on_custom_scene_without_clusters():
    """Fewer than 4*K spheres -> everything on the direct list; mixed primitives; ties between objects."""
    objs, cam = M.builtin_desc(0)
    pick = np.r_[0:7, 7:27, 207:215, 317:321]
    sub = objs[pick].copy()
    so, sm = O.Scene(sub, cam), M.Scene(sub, cam)
    want, _ = so.render(640, 360, 5, 0, 0, 80000, threads=8)
    got, _ = sm.render(640, 360, 5, 0, 0, 80000)
    assert got.tobytes() == want.tobytes()
    # duplicate objects: exact distance ties must resolve to the FIRST object (scene.rs:51)
    dup = np.concatenate([sub, sub[7:27]])
    dup[len(sub):]["material_kind"] = 3  # the duplicates are mirrors: a wrong tie-break changes the path
    so, sm = O.Scene(dup, cam), M.Scene(dup, cam)
    want, _ = so.render(640, 360, 5, 0, 0, 40000, threads=8)
    got, _ = sm.render(640, 360, 5, 0, 0, 40000)
    assert got.tobytes() == want.tobytes()


def test_duplicate_spheres_in_clusters_tie_to_first_object():
    objs, cam = M.builtin_desc(0)
    dup = np.concatenate([objs, objs[7:107]])
    dup[len(objs):]["material_kind"] = 3
    so, sm = O.Scene(dup, cam), M.Scene(dup, cam)
    want, _ = so.render(640, 360, 6, 0, 0, 40000, threads=8)
    got, _ = sm.render(640, 360, 6, 0, 0, 40000)
    assert got.tobytes() == want.tobytes()


def test_plot_weights_and_cie_lookup_bit_exact():
    objs, cam = M.builtin_desc(0)
    ph, _ = O.Scene(objs, cam).render(96, 54, 3, 0, 0, 200000, threads=8)
    assert M.plot(96, 54, ph).tobytes() == O.plot(96, 54, ph).tobytes()
    # edge photons: corners, out-of-gamut wavelengths
    edge = np.zeros(6, dtype=O.PHOTON_DTYPE)
    edge["x"] = [-1, 1, -1, 1, 0.999999, 0]
    edge["y"] = [-0.5625, 0.5625, 0.5625, -0.5625, 0.1, 0]
    edge["probability"] = 1.0
    edge["wavelength"] = [380, 780, 379.0, 781.0, 555, 374.9]
    assert M.plot(96, 54, edge).tobytes() == O.plot(96, 54, edge).tobytes()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_scenes_bit_exact(seed):
    """Culls must stay conservative on arbitrary geometry: overlapping, nested and huge spheres, randomly
    oriented prisms, glass everywhere (un-normalised directions, material.rs:246)."""
    from _random_scene import random_scene
    objs, cam = random_scene(seed, n_spheres=40 + 30 * seed)
    so, sm = O.Scene(objs, cam), M.Scene(objs, cam)
    want, segs = so.render(320, 180, seed, 0, 0, 40000, threads=8)
    got, segs2 = sm.render(320, 180, seed, 0, 0, 40000)
    assert segs == segs2 and got.tobytes() == want.tobytes()
    assert (want["probability"] > 0).any()


def test_degenerate_scenes_bit_exact():
    """The empty scene (scene.rs:43-60 returns None for every ray) and scenes of a single surface kind."""
    from _random_scene import random_scene
    full, cam = random_scene(3, n_spheres=40, n_prisms=3, n_planes=2, n_circles=2, n_parabs=2)
    subsets = [full[:0], full[:1].copy()] + [full[full["surface_kind"] == k].copy() for k in range(5)]
    for objs in subsets:
        want, segs = O.Scene(objs, cam).render(160, 90, 5, 2, 1000, 1 << 11, threads=4)
        got = M.Scene(objs, cam).render(160, 90, 5, 2, 1000, 1 << 11)
        got = got[0] if isinstance(got, tuple) else got
        assert got.tobytes() == want.tobytes()
        if len(objs) == 0:
            assert segs == 1 << 11 and not want["probability"].any()


def test_cull_table_is_conservative_by_construction():
    """The kernel's two-level cull table (rl_scene.cpp; not in the reference): every sphere lies inside its cluster's
    bounding sphere, every cluster / prism bound inside its group's -- so a ray that misses a bound cannot hit
    anything below it, whatever the parity sweeps happen to sample.  Checked in f64 on the built-in and on random
    scenes; the padding entries can never be reached (radius^2 = -inf)."""
    import ctypes as C
    from _random_scene import random_scene
    L = M.lib()
    L.mirror_bounds.restype = C.c_uint32
    L.mirror_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mirror_group_bounds.restype = C.c_uint32
    L.mirror_group_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    scenes = [M.builtin_desc(0, 0), M.builtin_desc(1, 0), M.builtin_desc(0, 158)]
    scenes += [random_scene(seed, n_spheres=90 + 37 * seed, n_prisms=3 + 2 * seed) for seed in range(4)]
    checked = 0
    group_sizes = []
    for objs, cam in scenes:
        sc = M.Scene(objs, cam)
        b = np.zeros((2048, 4), np.float32)
        nc = C.c_uint32(0)
        nb = L.mirror_bounds(sc.h, O.ptr(b), len(b), C.byref(nc))
        b = b[:nb].astype(np.float64)
        g = np.zeros((1024, 4), np.float32)
        sizes = (C.c_uint32 * 3)()
        ng = L.mirror_group_bounds(sc.h, O.ptr(g), len(g), sizes)
        g = g[:ng].astype(np.float64)
        gc, gp, ncg = sizes[0], sizes[1], sizes[2]          # clusters per group, prisms per group, cluster groups
        assert nb == gc * ncg + gp * (ng - ncg) and nc.value == gc * ncg
        group_sizes.append((gc, gp))
        for k in range(ng):
            first = gc * k if k < ncg else gc * ncg + gp * (k - ncg)
            members = b[first: first + (gc if k < ncg else gp)]
            real = members[np.isfinite(members[:, 3]) & (members[:, 3] > 0)]
            assert len(real) >= 1                                  # a group is never all padding
            if not np.isfinite(g[k, 3]):
                continue                                           # unbounded group: always reached
            reach = np.sqrt(((real[:, :3] - g[k, :3]) ** 2).sum(1)) + np.sqrt(real[:, 3])
            assert (reach <= np.sqrt(g[k, 3]) * (1 + 1e-6)).all(), (k, reach, g[k])
            checked += len(real)
        # level 1 over the spheres: every clustered sphere inside its cluster bound (5 % + 0.05 inflation)
        spheres = objs[objs["surface_kind"] == 0]
        centres, radii = spheres["v0"].astype(np.float64), np.abs(spheres["f"][:, 0].astype(np.float64))
        if nc.value:
            clusters = b[: nc.value]
            clusters = clusters[np.isfinite(clusters[:, 3]) & (clusters[:, 3] > 0)]
            d = np.sqrt(((centres[:, None, :] - clusters[None, :, :3]) ** 2).sum(-1)) + radii[:, None]
            inside_some = (d <= np.sqrt(clusters[None, :, 3])).any(1)
            big = radii > 4 * np.median(radii)                     # the direct list
            assert inside_some[~big].all()
    assert checked > 150
    # every scene gets one of the plans the kernel has an unrolled member loop for
    assert all(gc in (3, 4) and gp == 3 for gc, gp in group_sizes)


def test_third_level_of_the_cull_table_is_conservative_and_only_for_large_scenes():
    """Round 6: a scene with 112 or more cluster groups (several thousand spheres) gets a third level -- one SUPER bound per 8
    consecutive cluster groups, the groups padded with never-reached dummies to a multiple of 8 -- and every real group bound lies
    inside its super's; smaller scenes (everything that can be staged whole in LDS) keep the two-level table."""
    import ctypes as C
    from _random_scene import random_scene
    L = M.lib()
    L.mirror_group_bounds.restype = C.c_uint32
    L.mirror_group_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mirror_super_bounds.restype = C.c_uint32
    L.mirror_super_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    small = [M.builtin_desc(0, 0), M.builtin_desc(1, 0), M.builtin_desc(0, 158), random_scene(3, n_spheres=400, n_prisms=5)]
    small.append(M.builtin_desc(0, 600))    # 46 groups: two levels (the planner prices the third level: rl_scene.cpp, plan_cost)
    large = [M.builtin_desc(0, 1500), M.builtin_desc(0, 2500), random_scene(31, n_spheres=5000, n_prisms=12, n_planes=2, n_circles=3, n_parabs=1)]
    checked = 0
    for which, (objs, cam) in enumerate(small + large):
        sc = M.Scene(objs, cam)
        g = np.zeros((4096, 4), np.float32)
        sizes = (C.c_uint32 * 3)()
        ng = L.mirror_group_bounds(sc.h, O.ptr(g), len(g), sizes)
        ncg = sizes[2]
        g = g[:ncg].astype(np.float64)
        sup = np.zeros((1024, 4), np.float32)
        sg = C.c_uint32(0)
        ns = L.mirror_super_bounds(sc.h, O.ptr(sup), len(sup), C.byref(sg))
        if which < len(small):
            assert ns == 0
            continue
        assert ns >= 14 and sg.value == 8 and ncg == ns * sg.value
        sup = sup[:ns].astype(np.float64)
        for s in range(ns):
            members = g[sg.value * s: sg.value * (s + 1)]
            real = members[np.isfinite(members[:, 3]) & (members[:, 3] > 0)]
            assert len(real) >= 1                                  # a super is never all padding
            reach = np.sqrt(((real[:, :3] - sup[s, :3]) ** 2).sum(1)) + np.sqrt(real[:, 3])
            assert (reach <= np.sqrt(sup[s, 3]) * (1 + 1e-6)).all(), (s, reach, sup[s])
            checked += len(real)
        # (and the host mirror, which scans the flattened scene's spheres linearly, still renders the oracle's photons)
        want, segs = O.Scene(objs, cam).render(160, 90, 7, 1, 0, 1 << 9)
        got = M.Scene(objs, cam).render(160, 90, 7, 1, 0, 1 << 9)
        got = got[0] if isinstance(got, tuple) else got
        assert got.tobytes() == want.tobytes()
    assert checked >= 3 * 112


def test_normals_along_z_take_one_product_and_change_no_result():
    """Round 6: the built-in room's paraboloids, plane and circles all have normals along z (app.rs:179-231); the kernel's straight-line
    block for them takes dot(normal, v) as normal.z * v.z (rl_paraboloid_t<AXIS_Z>, rl_plane_t<AXIS_Z>) when rl_flatten_scene says so.
    The two forms can differ in the SIGN of a zero dot product only, and that never changes what the scan does with the result:
    4 M adversarial cases (exact zeros of either sign in the ray, origins at the primitive's height, rays along the axis)."""
    from _random_scene import random_scene
    assert M.small_axis_z(M.Scene(*M.builtin_desc(0, 0))) and M.small_axis_z(M.Scene(*M.builtin_desc(1, 0))) and M.small_axis_z(M.Scene(*M.builtin_desc(0, 158)))
    objs, cam = random_scene(5, n_spheres=30, n_prisms=2, n_planes=2, n_circles=2, n_parabs=2)
    assert not M.small_axis_z(M.Scene(objs, cam))                       # random normals
    objs, cam = M.builtin_desc(0, 0)
    tilted = objs.copy()
    k = int(np.nonzero(tilted["surface_kind"] == 1)[0][0])             # the ceiling plane, tilted by a hair
    tilted["v0"][k] = (1.0e-3, 0.0, -1.0)
    assert not M.small_axis_z(M.Scene(tilted, cam))
    cases, hits, differences, zeros = M.axis_z_check(2026, 2_000_000)
    assert cases == 4_000_000 and differences == 0 and hits > 400_000 and zeros > 300_000, (cases, hits, differences, zeros)


def test_paraboloid_one_division_form_agrees_wherever_its_condition_admits_it():
    """rl_paraboloid_t on the device divides ONE numerator, chosen by sign, where `a < 0 and (disc < 0 or max(|b|, sqrt|disc|) >= 2^-90)`
    (round 6: one compare instead of a range check per numerator; the argument is in rl_core.h).  3 M coefficient triples over the whole
    exponent range, a third of them adversarial (tiny b and c, rays that start on the surface, zero discriminants): wherever the
    condition holds, the reference's two-quotient selection (geometry.rs:316-341) gives the same answer bit for bit."""
    cases, admitted, hits, disagreements, rejected_tiny = M.parab_check(6, 3_000_000)
    assert cases == 3_000_000 and disagreements == 0, (cases, admitted, hits, disagreements)
    assert admitted > 2_000_000 and hits > 500_000 and rejected_tiny > 1_000, (admitted, hits, rejected_tiny)


def test_the_cull_table_is_planned_per_scene():
    """rl_flatten_scene builds the table for each cluster size the kernel has an unrolled member loop for x 3 / 4 clusters per
    group and keeps the plan its cost estimate likes best (rl_scene.cpp: plan_cost over the rays of sample paths).  The
    choices below are the ones that measured fastest on MI355X; what a plan changes is how many bounds
    a ray is tested against -- never a result (the bit-exact tests in this file run on whatever plan was chosen)."""
    demo = M.cull_counts(M.Scene(*M.builtin_desc(0, 0)), 1920, 1080, 42, 0, 0, 3000)
    replicated = M.cull_counts(M.Scene(*M.builtin_desc(0, 158)), 1920, 1080, 42, 0, 0, 3000)
    assert demo["members_per_cluster"] == 14 and demo["clusters_per_group"] in (3, 4)
    assert replicated["members_per_cluster"] == 10 and replicated["clusters_per_group"] == 4
    # the quantities the plan is about: a ray of the built-in scene reaches ~1.7 groups and ~1.6 clusters, 0.55 members pass
    assert 1.4 < demo["group_pairs"] < 2.0 and 1.4 < demo["cluster_pairs"] < 1.8 and 0.4 < demo["member_pairs"] < 0.7
    assert replicated["cluster_pairs"] < 2.6
    # a scene too small for clusters has no table at all
    objs, cam = M.builtin_desc(0, 0)
    few = M.cull_counts(M.Scene(objs[objs["surface_kind"] != 0][:30], cam), 640, 360, 1, 0, 0, 100)
    assert few["clusters"] == 0 and few["group_pairs"] == 0


@pytest.mark.parametrize("layout", ["same", "line", "zero_radius", "huge_spread", "infinite", "random"])
def test_planned_table_on_degenerate_sphere_layouts(layout):
    """The planner (median cuts, k-means, local search, sample paths, cost estimate) gets sphere sets it cannot do anything
    sensible with -- all at one point, on a line, of radius zero, spread over seven decades, one of infinite radius -- at the
    sizes where clusters start (40 spheres) and where groups fill unevenly: it must terminate, and the paths through the
    table it built must equal the oracle's linear scan bit for bit."""
    objs0, cam = M.builtin_desc(0, 0)
    proto = objs0[objs0["surface_kind"] == 0][:1]
    rest = objs0[objs0["surface_kind"] != 0]
    rng = np.random.default_rng(7)
    for n in (40, 41, 57, 141):
        o = np.repeat(proto, n)
        o["v0"] = rng.normal(0, 8, (n, 3)).astype(np.float32)
        o["v0"][:, 1] = np.abs(o["v0"][:, 1])
        o["f"][:, 0] = rng.uniform(0.1, 1.0, n).astype(np.float32)
        if layout == "same":
            o["v0"] = np.array([1.0, 2.0, 3.0], np.float32)
        elif layout == "line":
            o["v0"] = np.stack([np.linspace(-20, 20, n), np.ones(n), np.ones(n)], 1).astype(np.float32)
        elif layout == "zero_radius":
            o["f"][:, 0] = 0.0
        elif layout == "huge_spread":
            o["v0"] = (rng.normal(0, 1, (n, 3)) * np.exp(rng.uniform(-5, 12, (n, 1)))).astype(np.float32)
            o["f"][:, 0] = np.exp(rng.uniform(-8, 3, n)).astype(np.float32)
        elif layout == "infinite":
            o["f"][0, 0] = np.inf
        scene = np.concatenate([rest, o])
        so, sm = O.Scene(scene, cam), M.Scene(scene, cam)
        want, segs = so.render(320, 180, 3, 0, 0, 6000, threads=8)
        got, segs2 = sm.render(320, 180, 3, 0, 0, 6000)
        assert segs == segs2 and got.tobytes() == want.tobytes(), (layout, n)
        assert M.cull_counts(sm, 320, 180, 3, 0, 0, 200)["members_per_cluster"] in (10, 14)


# ---- the prism shortcut (rl_hex_prism_fast) against the Compound tree it stands in for (geometry.rs:380-407) ----------

@pytest.mark.parametrize("which", [0, 1])
def test_prism_shortcut_never_contradicts_the_compound_tree(which):
    """Whatever rl_hex_prism_fast decides must be the tree's answer bit for bit (distance and half-space); what it
    does not decide goes to the tree anyway.  Adversarial pairs: rays that start on a face (origin = hit + dir * 1e-5,
    trace_unit.rs:114, un-normalised directions as glass leaves them), rays aimed at edges and vertices, rays nearly
    parallel to faces -- with a reciprocal that is off by one ulp either way, as v_rcp_f32 may be (RL_TEST_RCP_NOISE)."""
    objs, cam = M.builtin_desc(which)
    sc = M.Scene(objs, cam)
    r = M.prism_fast_check(sc, 3_000_000, 20260928 + which)
    assert r["pairs"] > 2_000_000 and r["wrong"] == 0, r
    assert r["hits"] > 0.3 * r["pairs"] and r["misses"] > 0.3 * r["pairs"], r   # it does decide both ways
    assert r["undecided"] < 0.2 * r["pairs"], r                               # (adversarial mix: ~13 %)
    p = M.prism_fast_check_paths(sc, 1280, 720, 3, 1, 5_000_000_000, 150_000)
    assert p["pairs"] > 300_000 and p["wrong"] == 0, p
    assert p["undecided"] < 0.003 * p["pairs"], p                             # real paths: ~0.1 %
    assert abs(p["hits"] + p["undecided"] - p["tree_hits"]) <= p["undecided"], p


def test_prism_shortcut_on_random_prisms():
    """Randomly oriented, sized and placed prisms (tests/_random_scene.py), incl. far from the origin."""
    import _random_scene as RS
    for seed in (1, 2, 3, 4):
        objs, cam = RS.random_scene(seed, n_spheres=40, n_prisms=12)
        if seed == 4:   # far away from the origin: the margins scale with the coordinates
            objs = objs.copy()
            objs["v1"][objs["surface_kind"] == 4] += np.float32(900.0)
        sc = M.Scene(objs, cam)
        r = M.prism_fast_check(sc, 1_000_000, seed)
        assert r["pairs"] > 500_000 and r["wrong"] == 0, (seed, r)
        p = M.prism_fast_check_paths(sc, 640, 360, seed, 0, 0, 30_000)
        assert p["wrong"] == 0, (seed, p)


def test_prism_cylinders_hold_every_hit_and_are_used_only_where_they_pay():
    """The second conservative bound of a prism (rl_scene.cpp: prism_cylinder) -- a cylinder around its axis -- is built
    only for scenes with >= 40 prisms (the glass-stress scene's 66; not the built-in scene's 22), and every point where the
    reference's Compound tree reports a hit lies inside it: culling by it cannot change Scene::intersect's result."""
    objs, cam = M.builtin_desc(0)
    assert len(M.prism_cylinders(M.Scene(objs, cam))) == 0
    objs, cam = M.builtin_desc(1)
    sc = M.Scene(objs, cam)
    cyl = M.prism_cylinders(sc)
    assert len(cyl) == 66 and np.isfinite(cyl).all()
    assert np.allclose(np.linalg.norm(cyl[:, 4:7], axis=1), 1.0, atol=1e-6) and (cyl[:, 3] > 0.5).all() and (cyl[:, 3] < 2.5).all()
    prisms, rays, tree = M.prism_pairs(sc, 600_000, 4242)
    hit = tree[:, 0] != 0xffffffff
    t = tree[hit, 0].view(np.float32).astype(np.float64)
    p = rays[hit, :3].astype(np.float64) + rays[hit, 3:].astype(np.float64) * t[:, None]
    c, r, a = cyl[prisms[hit], :3].astype(np.float64), cyl[prisms[hit], 3].astype(np.float64), cyl[prisms[hit], 4:7].astype(np.float64)
    d = p - c
    off_axis = np.sqrt(np.maximum((d * d).sum(1) - (d * a).sum(1) ** 2, 0.0))
    assert hit.sum() > 150_000 and (off_axis <= r / 1.04).all(), (off_axis / r).max()   # inside, with the inflation to spare


def test_scan_order_may_replace_the_tie_rule_only_where_the_objects_are_ordered():
    """scene.rs:51 keeps the first object among equal distances.  The kernel's scan of the small primitives (paraboloids, then
    planes and circles) decides ties by `t < best.t` alone where that order IS the objects' order (RlFlatScene::small_ordered,
    computed by rl_flatten_scene and handed to the kernel as RlSceneLayout::small_ordered), and by the general rule
    elsewhere.  The built-in scenes are ordered (app.rs:172-199 lists the paraboloids before the sky lights and the ceiling);
    the random scenes of the parity tests are not (their planes and circles come first), so both forms of the kernel's loop
    run under the bit-exact GPU tests -- and a scene with a plane in front of a paraboloid must not be called ordered."""
    import _random_scene as RS
    assert M.small_ordered(M.Scene(*M.builtin_desc(0, 0))) and M.small_ordered(M.Scene(*M.builtin_desc(1, 0)))
    objs, cam = RS.random_scene(3)
    assert not M.small_ordered(M.Scene(objs, cam))
    demo, cam = M.builtin_desc(0, 0)
    kinds = demo["surface_kind"]
    parab, plane = int(np.flatnonzero(kinds == 3)[0]), int(np.flatnonzero(kinds == 1)[0])
    swapped = demo.copy()
    swapped[[parab, plane]] = swapped[[plane, parab]]          # the ceiling now precedes the floor paraboloid
    assert not M.small_ordered(M.Scene(swapped, cam))
    only_spheres = demo[kinds == 0]
    assert M.small_ordered(M.Scene(only_spheres.copy(), cam))   # nothing to order
