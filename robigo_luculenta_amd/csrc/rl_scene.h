// rl_scene.h -- the scene as flat POD records, the layout the trace kernel scans.
//
// The reference keeps `Vec<Object>` of boxed trait objects and dispatches dynamically per primitive
// (object.rs:20-31, scene.rs:39-60).  Here a scene is five tightly packed arrays of 16-byte records,
// grouped by primitive type so each scan loop is branch-free and every record is one wave-uniform
// 16-byte fetch (LDS broadcast or scalar load):
//
//   spheres   1 x RlF4 each : {centre.xyz, radius^2}                         geometry.rs:186-200
//   planes    2 x RlF4 each : {normal.xyz, radius^2 or -1}, {offset.xyz, obj} geometry.rs:35-51,130-150
//   parabs    3 x RlF4 each : {offset.xyz, obj}, {normal.xyz, 0}, {focal_point.xyz, 0}   :269-295
//   prisms   17 x RlF4 each : 8 half-spaces x ({normal.xyz, 0}, {offset.xyz, obj})       :409-515
//                             + {bounding-sphere centre.xyz, radius^2} (not in the reference: a
//                             conservative cull, see rl_prism_bound_pass); the odd stride also keeps
//                             per-lane prism fetches off a single LDS bank row
//   objects   2 x RlF4 each : {surface_kind | material_kind << 8, group index, 0, 0} as bits,
//                             {m0, m1, m2, 0}  (black body: m0 = kelvins, m1 = normalisation factor)
//
// The sphere array is padded with never-hit dummies (radius^2 = -inf) to a multiple of 4 plus one
// extra group of 4, so the kernel can unroll by 4 and prefetch one group ahead without a bounds check.
//
// Spheres are scanned first with the reference's strict `<`; the other groups are merged with the
// lexicographic (distance, object index) rule, which is exactly scene.rs:51's "first object wins".
// sphere_obj maps a sphere's slot back to its object index.
#pragma once
#include <vector>

#include "../../include/robigo_luculenta.h"
#include "rl_math.h"

struct alignas(16) RlF4 {
    float x, y, z, w;
};

#define RL_PRISM_STRIDE 17 // records per hexagonal prism: 16 half-space records + 1 bound

// Everything the per-path code needs to read; pointers are device or host memory depending on
// who built the view.
struct RlSceneView {
    const RlF4* spheres;
    const RlF4* planes;
    const RlF4* parabs;
    const RlF4* prisms;
    const RlF4* objects;
    const uint32_t* sphere_obj;
    const RlF4* cie; // RL_CIE_SAMPLES rows {X, Y, Z, 0}
    uint32_t n_spheres, n_planes, n_parabs, n_prisms, n_objects;
    uint32_t n_spheres_padded; // multiple of 4; records [n_spheres, n_spheres_padded + 4) are dummies
    RlCameraDesc camera;
    float screen_distance; // 1 / tan(field_of_view / 2), camera.rs:56 (constant per scene)
};

// Host-side flattened scene (built once by rl_scene_create).
struct RlFlatScene {
    std::vector<RlF4> spheres, planes, parabs, prisms, objects;
    std::vector<uint32_t> sphere_obj;
    uint32_t n_spheres;        // real spheres; `spheres` also holds the dummy padding
    uint32_t n_spheres_padded; // see RlSceneView
    RlCameraDesc camera;
    float screen_distance;
    // Total bytes of the primitive arrays (what RL_FETCH_LDS stages per workgroup).
    size_t staged_bytes() const {
        return (spheres.size() + planes.size() + parabs.size() + prisms.size() + objects.size()) * sizeof(RlF4) +
               sphere_obj.size() * sizeof(uint32_t);
    }
};

// Flattens a description; returns 0 or RL_E_INVALID (message in *err).
int rl_flatten_scene(const RlSceneDesc* desc, RlFlatScene* out, const char** err);

// Built-in generators (host).  Return the object count; fill `out` when it is large enough.
uint32_t rl_builtin_scene(int which, int param, std::vector<RlObjectDesc>* out, RlCameraDesc* camera);

RL_HD uint32_t rl_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RL_HD float rl_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
