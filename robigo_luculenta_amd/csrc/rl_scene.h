// rl_scene.h -- the scene as flat POD records, the layout the trace kernel scans.
//
// The reference keeps `Vec<Object>` of boxed trait objects and dispatches dynamically per primitive
// (object.rs:20-31, scene.rs:39-60).  Here a scene is five tightly packed arrays of 16-byte records,
// grouped by primitive type so each scan loop is branch-free and every record is one wave-uniform
// 16-byte fetch (LDS broadcast or scalar load):
//
//   spheres   1 x RlF4 each : {centre.xyz, radius^2}                         geometry.rs:186-200
//             "direct" spheres first (tested by every ray), then clusters of cluster_k spatially
//             close spheres, each preceded by a {bounding-sphere centre, radius^2} record (on the DEVICE a clustered
//             sphere's record is {centre.xyz, |centre|^2 - 1.001 radius^2}, see RlSceneView::sphere_r2)
//   planes    2 x RlF4 each : {normal.xyz, radius^2 or -1}, {offset.xyz, obj} geometry.rs:35-51,130-150
//   parabs    3 x RlF4 each : {offset.xyz, obj}, {normal.xyz, 0}, {focal_point.xyz, 0}   :269-295
//   prisms   17 x RlF4 each : 8 half-spaces x ({normal.xyz, 0}, {offset.xyz, obj})       :409-515
//                             (the fourth components of normal records 0 and 1 hold max |offset|_1 and max |normal|_1:
//                             the scale of rl_hex_prism_fast's error bound)
//                             + {bounding-sphere centre.xyz, radius^2} (not in the reference: a
//                             conservative cull, see rl_bound_pass); the odd stride also keeps
//                             per-lane prism fetches off a single LDS bank row
//   objects   1 x RlF4 each : {m0, m1, m2, bits: surface_kind | material_kind << 3 | group index << 6}
//                             (black body: m0 = kelvins, m1 = normalisation factor; one 16-byte record per object since round 5 --
//                             the table is read once per bounce and is a third of what a scene stages in LDS)
//
// The direct sphere list is padded with never-hit dummies (radius^2 = -inf) to a multiple of 4 plus
// one extra group of 4, so the kernel can unroll by 4 and prefetch one group ahead without a bounds
// check; short clusters are padded with the same dummies.
//
// Sphere clusters are NOT in the reference (its scan is flat, scene.rs:46).  They are a conservative
// cull: a ray that misses a cluster's (inflated) bounding sphere cannot hit any member, so skipping
// the members leaves Scene::intersect's result unchanged -- the parity tests check exactly that.
//
// All candidates are merged with the lexicographic (distance, object index) rule, which is
// scene.rs:51's strict `<` over objects in scan order ("first object wins").  sphere_obj maps a
// record position in `spheres` back to its object index.
#pragma once
#include <vector>

#include "../../include/robigo_luculenta.h"
#include "rl_math.h"

struct alignas(16) RlF4 {
    float x, y, z, w;
};

#define RL_PRISM_STRIDE 17 // records per hexagonal prism: 16 half-space records + 1 bound
// Spheres per cluster: chosen per scene (RlFlatScene::cluster_k, rl_scene.cpp) among RL_CLUSTER_K_CHOICES.  A cluster
// takes cluster_k + 1 records: the bound in front.  -DRL_CLUSTER_K=n forces one size (A/B builds; at most
// RL_CLUSTER_K_MAX: the kernel keeps one bit per member in a 32-bit mask).
#define RL_CLUSTER_K_MAX 31
#define RL_CLUSTER_K_CHOICES {10, 14} // (the kernel's member loop is unrolled for exactly these)
// Bounds per second-level group of the cull table.  Clusters: 3 or 4, chosen per scene together with the cluster size
// (RlFlatScene::group_gc; -DRL_GROUP_GC=n forces one).  Prisms: threes (the glass-stress scene loses 2.5 % with fours).
#ifndef RL_GROUP_GP
#define RL_GROUP_GP 3
#endif

// Everything the per-path code needs to read; pointers are device or host memory depending on
// who built the view.
struct RlSceneView {
    const RlF4* spheres;
    const RlF4* planes;
    const RlF4* parabs;
    const RlF4* prisms;
    const RlF4* objects;
    const uint32_t* sphere_obj;
    // Device only: radius^2 per sphere record.  On the device a CLUSTERED sphere's record is {centre, |c|^2 - R^2} -- the form
    // the cull test reads (rl_kernels.hip.h) -- and its exact radius^2 lives here; the host-side view keeps
    // {centre, radius^2} everywhere and leaves this null.
    const float* sphere_r2;
    const RlF4* cie; // RL_CIE_SAMPLES rows {X, Y, Z, 0}
    uint32_t n_planes, n_parabs, n_prisms, n_objects;
    uint32_t n_direct;         // direct spheres: records [0, n_direct)
    uint32_t n_direct_padded;  // multiple of 4; records [n_direct, n_direct_padded + 4) are dummies
    uint32_t cluster_base;     // first cluster record (= n_direct_padded + 4)
    uint32_t n_clusters;       // each cluster_k + 1 records: bound, then cluster_k spheres
    uint32_t cluster_k;
    // 3 records: the 10 floats of RlCameraDesc, then screen_distance = 1 / tan(field_of_view / 2)
    // (camera.rs:56, constant per scene).  Read from memory where a path starts instead of being held
    // in a dozen scalar registers across the whole persistent loop.
    const RlF4* camera_rec;
    // Device only: the blob every record array above lives in.  On the device an object's "group" bits hold the blob index of the
    // record its hit is completed from -- the sphere's record, the plane's / circle's normal, the paraboloid's first, the prism's
    // first -- so that rl_finish_hit reads it with one load whatever the surface is (rl_api.hip: rl_scene_create writes them).
    const RlF4* records;
};

// Host-side flattened scene (built once by rl_scene_create).
struct RlFlatScene {
    std::vector<RlF4> spheres, planes, parabs, prisms, objects;
    // Device-only cull table, every bound as {centre.xyz, |centre|^2 - radius^2}: the form the kernel's expanded cull
    // test consumes (rl_kernels.hip.h).  Two levels: the cluster bounds, then the prism bounds, each list padded to
    // a multiple of the group size (group_gc clusters, RL_GROUP_GP prisms) with never-reached dummies and ordered so that the entries of a group are
    // spatial neighbours (clusters and prisms are stored in that order too); then one GROUP bound -- the bounding
    // sphere of the bounds it covers -- per group of clusters, then per group of prisms.  A ray is tested
    // against the group bounds in a wave-uniform loop and only the (group, ray) pairs that pass go on to the
    // group's members.  Not in the reference; conservative like the bounds themselves.
    std::vector<RlF4> cull_bounds;
    uint32_t n_cluster_groups, n_prism_groups; // cull_bounds = [group_gc * n_cluster_groups][GP * n_prism_groups][groups][groups][slack]
    uint32_t group_gc = 3;                     // clusters per group (3 or 4), chosen with cluster_k: rl_scene.cpp, plan_cost
    // Third level (round 6), for scenes with many sphere clusters: one SUPER bound per super_g consecutive cluster groups, behind the
    // group bounds in cull_bounds: [level 1][cluster groups][prism groups][cluster supers][slack].  A ray then tests the super bounds
    // with wave-uniform records instead of every group bound; the (super, ray) pairs that pass are compacted and a round tests the
    // super's groups, one pair per lane (rl_kernels.hip.h: ring T).  0 supers: the table has two levels, as every scene small enough
    // to be staged whole in LDS has.  The cluster groups are padded with never-reached dummies to a multiple of super_g.
    uint32_t n_cluster_supers = 0, super_g = 0;
    std::vector<RlF4> prism_cyl;               // 2 records per prism {point on the axis, radius}, {unit axis, 0}; empty unless
    bool prism_cylinders = false;              // ... the scene has enough prisms for the second bound to pay (rl_scene.cpp)
    std::vector<float> sphere_cull_w;          // per record of `spheres`: |c|^2 - R^2 of a clustered sphere (else +inf), see rl_flatten_scene
    float cull_cmax2; // max |centre|^2 over cull_bounds (scales the cull's rounding slack)
    // The paraboloids' objects all precede the planes' and circles', and each list is in object order: the kernel's scan of the
    // small primitives (paraboloids, then planes) then visits objects in ascending order and `t < best.t` alone is scene.rs:51's
    // "the first object wins a tie" (RlSceneLayout::small_ordered; the built-in scenes).  Other scenes take the general compare.
    bool small_ordered = false;
    // Every paraboloid's, plane's and circle's normal is along z (x and y components zero): the kernel's straight-line block for the
    // built-in room then takes the dot products with a normal as one product (rl_paraboloid_t<AXIS_Z>); RlSceneLayout::small_ordered bit 1.
    bool small_axis_z = false;
    std::vector<uint32_t> sphere_obj;
    uint32_t n_direct, n_direct_padded, cluster_base, n_clusters, cluster_k; // see RlSceneView
    RlCameraDesc camera;
    float screen_distance;
    std::vector<RlF4> camera_rec; // see RlSceneView
    // Total bytes of the primitive arrays (what RL_FETCH_LDS stages per workgroup).
    size_t staged_bytes() const {
        return (spheres.size() + planes.size() + parabs.size() + prisms.size() + objects.size() + cull_bounds.size() + camera_rec.size() + prism_cyl.size()) * sizeof(RlF4) +
               sphere_obj.size() * (sizeof(uint32_t) + sizeof(float));
    }
};

// Flattens a description; returns 0 or RL_E_INVALID (message in *err).
int rl_flatten_scene(const RlSceneDesc* desc, RlFlatScene* out, const char** err);

// Built-in generators (host).  Return the object count; fill `out` when it is large enough.
uint32_t rl_builtin_scene(int which, int param, std::vector<RlObjectDesc>* out, RlCameraDesc* camera);

RL_HD uint32_t rl_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RL_HD float rl_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// The fourth component of an object record: what it is and where its primitive's records are (the sphere's position in
// `spheres`, the plane's / paraboloid's / prism's number in its list).
RL_HD uint32_t rl_object_bits(uint32_t surface_kind, uint32_t material_kind, uint32_t group_index) {
    return (surface_kind & 7u) | ((material_kind & 7u) << 3) | (group_index << 6);
}
static_assert(RL_SURFACE_HEX_PRISM <= 7 && RL_MATERIAL_SOAP_BUBBLE <= 7, "rl_object_bits keeps three bits for the surface kind and three for the material kind");
static_assert(RL_GROUP_GP == 3 || RL_GROUP_GP == 4, "rl_scan_wave's unrolled ring-S round tests a group's first three children unconditionally and a fourth if there is one");
RL_HD uint32_t rl_object_surface(uint32_t bits) { return bits & 7u; }
RL_HD uint32_t rl_object_material(uint32_t bits) { return (bits >> 3) & 7u; }
RL_HD uint32_t rl_object_group(uint32_t bits) { return bits >> 6; }
