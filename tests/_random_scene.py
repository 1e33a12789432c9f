"""Random scenes for the parity tests: arbitrary mixes of every surface and material kind, including
overlapping / nested / huge spheres and randomly oriented prisms, so the conservative culls (sphere
clusters, prism bounds) are exercised far away from the demo scene's regular layout."""
import numpy as np

import _oracle as O


def random_scene(seed, n_spheres=120, n_prisms=6, n_planes=2, n_circles=2, n_parabs=1):
    rng = np.random.default_rng(seed)
    n = 1 + n_spheres + n_prisms + n_planes + n_circles + n_parabs
    objs = np.zeros(n, dtype=O.OBJECT_DTYPE)
    k = 0
    # a light so paths can end with a contribution
    objs[k]["surface_kind"], objs[k]["material_kind"] = 0, 0
    objs[k]["v0"], objs[k]["f"][0], objs[k]["m"] = (0, 0, 8), 4.0, (6000.0, 1.0, 0.0)
    k += 1
    for _ in range(n_spheres):
        objs[k]["surface_kind"] = 0
        objs[k]["v0"] = rng.normal(0, 12, 3)
        objs[k]["f"][0] = float(np.exp(rng.uniform(np.log(0.2), np.log(6.0)))) if rng.random() < 0.9 else 15.0
        k += 1
    for _ in range(n_prisms):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        objs[k]["surface_kind"] = 4
        objs[k]["v0"], objs[k]["v1"] = axis, rng.normal(0, 15, 3)
        objs[k]["f"] = (rng.uniform(2, 5), rng.uniform(0.2, 1.5), rng.uniform(0, 6.28), rng.uniform(3, 14))
        k += 1
    for _ in range(n_planes):
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        objs[k]["surface_kind"] = 1
        objs[k]["v0"], objs[k]["v1"] = nrm, nrm * rng.uniform(25, 50)
        k += 1
    for _ in range(n_circles):
        nrm = rng.normal(size=3)
        nrm /= np.linalg.norm(nrm)
        objs[k]["surface_kind"] = 2
        objs[k]["v0"], objs[k]["v1"], objs[k]["f"][0] = nrm, rng.normal(0, 20, 3), rng.uniform(3, 12)
        k += 1
    for _ in range(n_parabs):
        objs[k]["surface_kind"] = 3
        objs[k]["v0"], objs[k]["v1"], objs[k]["f"][0] = (0, 0, -1.0), (0, 0, -20.0), 30.0
        k += 1
    # materials for everything but the light: all six kinds, a few more emitters
    kinds = rng.choice([0, 1, 2, 3, 4, 5], size=n - 1, p=[0.05, 0.2, 0.25, 0.15, 0.15, 0.2])
    for i in range(1, n):
        mk = int(kinds[i - 1])
        if objs[i]["surface_kind"] == 4 and mk == 5:
            mk = 4  # soap bubbles read the sphere tangent; on other surfaces it is the zero vector (still legal)
        objs[i]["material_kind"] = mk
        objs[i]["m"] = {0: (rng.uniform(3000, 9000), rng.uniform(0.3, 1.0), 0), 1: (rng.uniform(0.3, 0.95), 0, 0),
                        2: (rng.uniform(0.5, 0.95), rng.uniform(400, 750), rng.uniform(20, 80)),
                        3: (rng.uniform(0.0, 0.5), 0, 0), 4: (0, 0, 0), 5: (0, 0, 0)}[mk]
    # the far planes glow, so paths that leave the cloud of objects end on a light
    for i in range(1, n):
        if objs[i]["surface_kind"] == 1:
            objs[i]["material_kind"], objs[i]["m"] = 0, (rng.uniform(4000, 8000), 0.5, 0.0)
    cam = O.demo_scene_desc()[1]
    return objs, cam
