// rl_scene.cpp -- host side of the scene: the built-in generators and the flattening of an
// RlSceneDesc into the 16-byte records of rl_scene.h.  Pure host code (no device needed).
#include "rl_scene.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

#include "rl_core.h"

namespace {

const float PI = RL_PI_F;

RlVector3 V(float x, float y, float z) {
    RlVector3 v;
    v.x = x; v.y = y; v.z = z;
    return v;
}
RlVector3 V(RlF3 f) { return V(f.x, f.y, f.z); }
RlF3 F(const RlVector3& v) { return rl_f3(v.x, v.y, v.z); }
RlF4 F4(RlF3 v, float w) {
    RlF4 r;
    r.x = v.x; r.y = v.y; r.z = v.z; r.w = w;
    return r;
}

RlObjectDesc make_object(uint32_t surface, RlVector3 v0, RlVector3 v1, float f0, float f1, float f2, float f3,
                         uint32_t material, float m0, float m1, float m2) {
    RlObjectDesc o;
    std::memset(&o, 0, sizeof o);
    o.surface_kind = surface;
    o.material_kind = material;
    o.v0 = v0; o.v1 = v1;
    o.f0 = f0; o.f1 = f1; o.f2 = f2; o.f3 = f3;
    o.m0 = m0; o.m1 = m1; o.m2 = m2;
    return o;
}
RlObjectDesc sphere(RlF3 c, float r, uint32_t mat, float m0 = 0, float m1 = 0, float m2 = 0) {
    return make_object(RL_SURFACE_SPHERE, V(c), V(0, 0, 0), r, 0, 0, 0, mat, m0, m1, m2);
}

// Paraboloid::new's derived fields (geometry.rs:286-295).
struct ParabFields {
    RlF3 offset, normal, focal_point;
};
ParabFields paraboloid_fields(RlF3 normal, RlF3 offset, float focal_distance) {
    ParabFields p;
    p.normal = normal;
    p.offset = rl_sub(offset, rl_mul(normal, focal_distance));
    p.focal_point = rl_mul(normal, focal_distance * 2.0f);
    return p;
}

// The seven fixed objects of the demo scene: sun, floor, two walls, two sky lights, ceiling
// (app.rs:171-231).
void push_fixed_objects(std::vector<RlObjectDesc>& out, float sun_radius) {
    const float r2 = sun_radius * sun_radius; // sun_radius.powi(2)
    const float sky_height = 30.0f;
    const RlVector3 down = V(0, 0, -1.0f), up = V(0, 0, 1.0f);
    out.push_back(sphere(rl_f3(0, 0, 0), sun_radius, RL_MATERIAL_BLACK_BODY, 6504.0f, 1.0f));
    out.push_back(make_object(RL_SURFACE_PARABOLOID, down, V(0, 0, -sun_radius), r2, 0, 0, 0, RL_MATERIAL_DIFFUSE_GREY, 0.8f, 0, 0));
    out.push_back(make_object(RL_SURFACE_PARABOLOID, up, V(1.0f, 0, -r2), r2, 0, 0, 0, RL_MATERIAL_DIFFUSE_COLOURED, 0.9f, 550.0f, 40.0f));
    out.push_back(make_object(RL_SURFACE_PARABOLOID, up, V(-1.0f, 0, -r2), r2, 0, 0, 0, RL_MATERIAL_DIFFUSE_COLOURED, 0.9f, 660.0f, 60.0f));
    out.push_back(make_object(RL_SURFACE_CIRCLE, down, V(-sun_radius, 0, sky_height), 5.0f, 0, 0, 0, RL_MATERIAL_BLACK_BODY, 7600.0f, 0.6f, 0));
    out.push_back(make_object(RL_SURFACE_CIRCLE, down, V(-sun_radius * 0.5f, sun_radius * 2.0f + 15.0f, sky_height), 15.0f, 0,
                              0, 0, RL_MATERIAL_BLACK_BODY, 5000.0f, 0.6f, 0));
    out.push_back(make_object(RL_SURFACE_PLANE, down, V(0, 0, sky_height * 2.0f), 0, 0, 0, 0, RL_MATERIAL_DIFFUSE_COLOURED, 0.5f, 470.0f, 25.0f));
}

// A ring of `count` x 2 hexagonal SF10 prisms standing on the floor paraboloid (app.rs:287-325).
void push_prism_ring(std::vector<RlObjectDesc>& out, float sun_radius, int count, float prism_radius) {
    const ParabFields floor = paraboloid_fields(rl_f3(0, 0, -1.0f), rl_f3(0, 0, -sun_radius), sun_radius * sun_radius);
    const float prism_angle = PI * 2.0f / (float)count;
    const float prism_height = 8.0f;
    const float variants[2][4] = {{0.0f, 1.0f, 0.0f, 1.0f}, {0.5f * prism_angle, 1.2f, PI * 0.5f, 1.5f}};
    for (int i = 0; i < count; ++i) {
        for (int v = 0; v < 2; ++v) {
            const float ofs = variants[v][0], radius = variants[v][1], phi_ofs = variants[v][2], h = variants[v][3];
            const float phi = (float)i * prism_angle + ofs;
            RlF3 position = rl_f3(rl_cosf_d(phi) * prism_radius * radius, rl_sinf_d(phi) * prism_radius * radius, 0.0f);
            RlF3 normal = rl_f3(0, 0, -1.0f);
            // Shoot straight down at the floor to find where the prism stands and how it leans.
            const float t = rl_paraboloid_t(floor.offset, floor.normal, floor.focal_point, position, normal);
            if (!(t < 0.0f)) {
                const RlF3 pos = rl_add(position, rl_mul(normal, t));
                const RlF3 local_pos = rl_sub(pos, floor.offset);
                const RlF3 plane_pr = rl_sub(local_pos, rl_mul(floor.normal, rl_dot(local_pos, floor.normal)));
                const RlF3 surface_normal = rl_normalise(rl_sub(floor.focal_point, plane_pr));
                normal = rl_neg(surface_normal);
                position = rl_add(pos, rl_mul(rl_mul(normal, 2.0f), h));
            }
            out.push_back(make_object(RL_SURFACE_HEX_PRISM, V(normal), V(position), 3.0f, 1.0f, phi + phi_ofs, prism_height * h,
                                      RL_MATERIAL_SF10_GLASS, 0, 0, 0));
        }
    }
}

RlCameraDesc demo_camera() { // app.rs:327-357
    RlCameraDesc c;
    c.phi0 = 1.0f; c.phi1 = 0.01f;
    c.alpha0 = 0.3f; c.alpha1 = -0.01f;
    c.dist0 = 50.0f; c.dist1 = -0.5f;
    c.fov_over_pi = 0.35f;
    c.focal_factor = 0.9f;
    c.depth_of_field = 2.0f;
    c.chromatic_abberation = 0.012f;
    return c;
}

// App::set_up_scene (app.rs:166-325) with `seeds` sunflower seeds per spiral.
void demo_scene(int seeds, std::vector<RlObjectDesc>& out) {
    const float sun_radius = 5.0f;
    push_fixed_objects(out, sun_radius);
    const double golden_ratio = 1.6180339887498948482045868343656381177203091798057628; // constants.rs:17
    const float gamma = PI * 2.0f * (1.0f - 1.0f / (float)golden_ratio);
    const float seed_size = 0.8f, seed_scale = 1.5f;
    const float fs = sun_radius / seed_scale + 1.0f;
    const int first_seed = (int)(fs * fs + 0.5f);
    for (int i = first_seed; i < first_seed + seeds; ++i) { // app.rs:239-253
        const float phi = (float)i * gamma;
        const float r = sqrtf((float)i) * seed_scale;
        const RlF3 c = rl_add(rl_f3(rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * -0.5f), rl_f3(0, 0, 0));
        out.push_back(sphere(c, seed_size, RL_MATERIAL_DIFFUSE_COLOURED, 0.9f,
                             (float)(i - first_seed) / (float)seeds * 130.0f + 600.0f, 60.0f));
    }
    for (int i = first_seed; i < first_seed + seeds; ++i) { // app.rs:256-268
        const float fi = (float)i + 0.5f;
        const float phi = fi * gamma;
        const float r = sqrtf(fi) * seed_scale;
        const RlF3 c = rl_add(rl_f3(rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * -0.25f), rl_f3(0, 0, 0));
        out.push_back(sphere(c, seed_size * 0.5f, RL_MATERIAL_GLOSSY_MIRROR, 0.1f));
    }
    for (int i = first_seed / 2; i < first_seed + seeds; ++i) { // app.rs:271-284
        const float phi = (float)(-i) * gamma;
        const float root = sqrtf((float)i);
        const float r = root * seed_scale * 1.5f;
        const RlF3 c = rl_add(rl_f3(rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * 1.5f + sun_radius * 2.0f), rl_f3(0, 0, 0));
        out.push_back(sphere(c, seed_size * (0.5f + root * 0.2f), RL_MATERIAL_SOAP_BUBBLE));
    }
    push_prism_ring(out, sun_radius, 11, 17.0f);
}

// BASELINE config 3: the seven fixed objects plus three rings of SF10 prisms at radius 10/17/24
// built by the recipe of app.rs:287-325 (66 prisms, 528 half-spaces).  Not in the reference.
void glass_stress_scene(std::vector<RlObjectDesc>& out) {
    const float sun_radius = 5.0f;
    push_fixed_objects(out, sun_radius);
    push_prism_ring(out, sun_radius, 11, 10.0f);
    push_prism_ring(out, sun_radius, 11, 17.0f);
    push_prism_ring(out, sun_radius, 11, 24.0f);
}

// new_infinite_prism (geometry.rs:421-450): appends 3 half-spaces as {normal,0},{offset,obj}.
void push_infinite_prism(std::vector<RlF4>& recs, RlF3 axis, RlF3 offset, float edge_length, float angle, float objbits) {
    const float radius = sqrtf(3.0f) / 6.0f * edge_length;
    const float a[3] = {angle, angle + PI * 2.0f / 3.0f, angle + PI * 4.0f / 3.0f};
    for (int k = 0; k < 3; ++k) {
        const RlF3 p = rl_rotate_towards(rl_f3(rl_cosf_d(a[k]), rl_sinf_d(a[k]), 0.0f), axis);
        recs.push_back(F4(p, 0.0f));
        recs.push_back(F4(rl_add(rl_mul(p, radius), offset), objbits));
    }
}

// Conservative bounding sphere of the convex polytope cut out by a prism's 8 half-spaces
// (the 16 records at `pr`): vertices = triple-plane intersections inside all other planes, in f64.
// The radius is inflated by 5 % + 1e-3 so float rounding in the hit position or in the cull test can
// never reject a real hit; an unbounded or degenerate polytope gets an infinite radius (never culled).
// The vertices of the convex polytope cut out by a prism's 8 half-spaces (f64).
void prism_vertices(const RlF4* pr, std::vector<double>& vx, std::vector<double>& vy, std::vector<double>& vz) {
    double n[8][3], d[8];
    for (int k = 0; k < 8; ++k) {
        n[k][0] = pr[2 * k].x; n[k][1] = pr[2 * k].y; n[k][2] = pr[2 * k].z;
        d[k] = n[k][0] * pr[2 * k + 1].x + n[k][1] * pr[2 * k + 1].y + n[k][2] * pr[2 * k + 1].z; // n . offset
    }
    for (int a = 0; a < 8; ++a)
        for (int b = a + 1; b < 8; ++b)
            for (int c = b + 1; c < 8; ++c) {
                const double* A = n[a]; const double* B = n[b]; const double* Cc = n[c];
                const double bxc[3] = {B[1] * Cc[2] - B[2] * Cc[1], B[2] * Cc[0] - B[0] * Cc[2], B[0] * Cc[1] - B[1] * Cc[0]};
                const double cxa[3] = {Cc[1] * A[2] - Cc[2] * A[1], Cc[2] * A[0] - Cc[0] * A[2], Cc[0] * A[1] - Cc[1] * A[0]};
                const double axb[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};
                const double det = A[0] * bxc[0] + A[1] * bxc[1] + A[2] * bxc[2];
                if (std::fabs(det) < 1e-9) continue;
                double p[3];
                for (int i = 0; i < 3; ++i) p[i] = (d[a] * bxc[i] + d[b] * cxa[i] + d[c] * axb[i]) / det;
                bool inside = true;
                for (int k = 0; k < 8 && inside; ++k)
                    inside = (n[k][0] * p[0] + n[k][1] * p[1] + n[k][2] * p[2] - d[k]) <= 1e-6;
                if (inside) {
                    vx.push_back(p[0]); vy.push_back(p[1]); vz.push_back(p[2]);
                }
            }
}

RlF4 prism_bound(const RlF4* pr) {
    std::vector<double> vx, vy, vz;
    prism_vertices(pr, vx, vy, vz);
    bool unbounded = false;
    RlF4 r;
    r.x = r.y = r.z = 0.0f;
    r.w = std::numeric_limits<float>::infinity();
    if (vx.size() < 4) return r; // empty or degenerate: never cull
    double c[3] = {0, 0, 0};
    for (size_t i = 0; i < vx.size(); ++i) { c[0] += vx[i]; c[1] += vy[i]; c[2] += vz[i]; }
    for (int i = 0; i < 3; ++i) c[i] /= (double)vx.size();
    // The polytope is bounded iff every direction is blocked; with the reference's constructors it
    // always is.  Guard anyway: a vertex farther than 1e6 means "treat as unbounded".
    double r2 = 0;
    for (size_t i = 0; i < vx.size(); ++i) {
        const double dx = vx[i] - c[0], dy = vy[i] - c[1], dz = vz[i] - c[2];
        r2 = std::max(r2, dx * dx + dy * dy + dz * dz);
    }
    if (!(r2 < 1e12)) unbounded = true;
    if (unbounded) return r;
    const double radius = std::sqrt(r2) * 1.05 + 1e-3;
    r.x = (float)c[0]; r.y = (float)c[1]; r.z = (float)c[2];
    r.w = (float)(radius * radius);
    return r;
}

// A second conservative bound for a prism, used by the kernel when a scene holds so many prisms that the (prism, ray)
// pairs that pass the bounding spheres exceed one round per wave iteration (RlFlatScene::prism_cylinders): the prisms
// of the reference are sticks (height 8-12, cross-section radius < 1.8), whose bounding spheres are 3-4 times wider
// than they are.  out[0] = {a point on the axis, radius}, out[1] = {unit axis, 0}: the polytope's vertices lie within
// `radius` (inflated by 5 % + 1e-3) of the line, hence every hit does.  Degenerate / unbounded: radius = +inf (always passes).
void prism_cylinder(const RlF4* pr, RlF4 out[2]) {
    out[0] = RlF4{0.0f, 0.0f, 0.0f, std::numeric_limits<float>::infinity()};
    out[1] = RlF4{0.0f, 0.0f, 1.0f, 0.0f};
    std::vector<double> vx, vy, vz;
    prism_vertices(pr, vx, vy, vz);
    double a[3] = {pr[14].x, pr[14].y, pr[14].z}; // the far cap's normal = the axis (geometry.rs:455-468)
    const double al = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (vx.size() < 4 || !(al > 1e-12)) return;
    for (double& v : a) v /= al;
    double c[3] = {0, 0, 0};
    for (size_t i = 0; i < vx.size(); ++i) { c[0] += vx[i]; c[1] += vy[i]; c[2] += vz[i]; }
    for (double& v : c) v /= (double)vx.size();
    double r2 = 0;
    for (size_t i = 0; i < vx.size(); ++i) {
        const double d[3] = {vx[i] - c[0], vy[i] - c[1], vz[i] - c[2]};
        const double along = d[0] * a[0] + d[1] * a[1] + d[2] * a[2];
        r2 = std::max(r2, d[0] * d[0] + d[1] * d[1] + d[2] * d[2] - along * along);
    }
    if (!(r2 < 1e12)) return;
    out[0] = RlF4{(float)c[0], (float)c[1], (float)c[2], (float)(std::sqrt(std::max(r2, 0.0)) * 1.05 + 1e-3)};
    out[1] = RlF4{(float)a[0], (float)a[1], (float)a[2], 0.0f};
}

struct SphereIn {
    RlF4 rec;      // {centre, radius^2}
    uint32_t obj;
    double radius;
};

// Splits sphere indices into spatially compact groups of at most K by recursive median
// cuts along the longest axis of the centres.
void split_clusters(const std::vector<SphereIn>& sph, std::vector<uint32_t> idx, std::vector<std::vector<uint32_t>>& out, size_t K) {
    if (idx.size() <= K) {
        if (!idx.empty()) out.push_back(idx);
        return;
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i : idx) {
        const double c[3] = {sph[i].rec.x, sph[i].rec.y, sph[i].rec.z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a]);
            hi[a] = std::max(hi[a], c[a]);
        }
    }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
    auto coord = [&](uint32_t i) { return axis == 0 ? sph[i].rec.x : axis == 1 ? sph[i].rec.y : sph[i].rec.z; };
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return coord(a) < coord(b) || (coord(a) == coord(b) && a < b); });
    size_t left = ((idx.size() / 2 + K - 1) / K) * K; // full leaves on the left
    if (left >= idx.size()) left = idx.size() - K;
    split_clusters(sph, std::vector<uint32_t>(idx.begin(), idx.begin() + left), out, K);
    split_clusters(sph, std::vector<uint32_t>(idx.begin() + left, idx.end()), out, K);
}

// Balanced k-means refinement of the kd-split: keeps the number of clusters, caps every cluster at
// K members, and re-assigns spheres to the nearest centroid with room (most decided first).
// On the demo scene this cuts the clusters a ray reaches from ~2.5 to ~1.6.
void refine_clusters(const std::vector<SphereIn>& sph, std::vector<std::vector<uint32_t>>& clusters, size_t K) {
    const size_t k = clusters.size();
    if (k < 2) return;
    std::vector<uint32_t> all;
    for (const auto& c : clusters) all.insert(all.end(), c.begin(), c.end());
    std::sort(all.begin(), all.end());
    for (int iteration = 0; iteration < 24; ++iteration) {
        std::vector<double> cx(k, 0), cy(k, 0), cz(k, 0);
        for (size_t j = 0; j < k; ++j) {
            for (uint32_t i : clusters[j]) {
                cx[j] += sph[i].rec.x; cy[j] += sph[i].rec.y; cz[j] += sph[i].rec.z;
            }
            const double n = (double)std::max<size_t>(1, clusters[j].size());
            cx[j] /= n; cy[j] /= n; cz[j] /= n;
        }
        auto dist2 = [&](uint32_t i, size_t j) {
            const double dx = sph[i].rec.x - cx[j], dy = sph[i].rec.y - cy[j], dz = sph[i].rec.z - cz[j];
            return dx * dx + dy * dy + dz * dz;
        };
        // order: spheres whose best centroid is much closer than their second best go first
        std::vector<std::pair<double, uint32_t>> order;
        for (uint32_t i : all) {
            double best = 1e300, second = 1e300;
            for (size_t j = 0; j < k; ++j) {
                const double d = dist2(i, j);
                if (d < best) { second = best; best = d; }
                else if (d < second) second = d;
            }
            order.push_back({std::sqrt(best) - std::sqrt(second), i});
        }
        std::sort(order.begin(), order.end());
        std::vector<std::vector<uint32_t>> next(k);
        for (const auto& entry : order) {
            const uint32_t i = entry.second;
            size_t pick = k;
            double pick_d = 1e300;
            for (size_t j = 0; j < k; ++j) {
                if (next[j].size() >= K) continue;
                const double d = dist2(i, j);
                if (d < pick_d) { pick_d = d; pick = j; }
            }
            next[pick].push_back(i); // k * K >= number of spheres, so a slot always exists
        }
        bool same = true;
        for (size_t j = 0; j < k && same; ++j) {
            std::sort(next[j].begin(), next[j].end());
            std::vector<uint32_t> old = clusters[j];
            std::sort(old.begin(), old.end());
            same = old == next[j];
        }
        clusters = next;
        if (same) break;
    }
    std::vector<std::vector<uint32_t>> kept;
    for (auto& c : clusters)
        if (!c.empty()) kept.push_back(c);
    clusters = kept;
}

// ---- local search on a partition of spheres into bins of bounded size ------------------------------------------------
// Both levels of the cull table are the same problem: put spheres {centre, radius} -- the scene's spheres into clusters,
// the cluster / prism bounds into groups -- into bins of at most `capacity` so that a ray reaches few bins.  A random
// line that crosses the scene meets a convex body with a probability proportional to its surface area, so the
// expected number of bins a ray reaches is proportional to the SUM OF THE BINS' SQUARED BOUNDING RADII: that is what this
// minimises, starting from the median-cut + capacity-limited k-means partition, by moving one item to a neighbouring
// bin with room or exchanging two items between neighbouring bins while that lowers the sum.  (The start alone is at
// the mercy of how the item count divides: on the built-in scene its sum varies by 40 % between cluster sizes 10 .. 16.)
// Deterministic; changes which bounds a ray is tested against, never a result.
struct Ball {
    double c[3], r;
};
// Squared radius of (nearly) the smallest ball around `members`: "step towards the farthest member", as cluster_bound.
double bin_radius2(const std::vector<Ball>& items, const std::vector<uint32_t>& members) {
    if (members.empty()) return 0.0;
    double c[3] = {0, 0, 0};
    for (uint32_t i : members)
        for (int a = 0; a < 3; ++a) c[a] += items[i].c[a];
    for (int a = 0; a < 3; ++a) c[a] /= (double)members.size();
    auto reach = [&](uint32_t i) {
        const double dx = items[i].c[0] - c[0], dy = items[i].c[1] - c[1], dz = items[i].c[2] - c[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz) + items[i].r;
    };
    double radius = 0;
    for (int it = 0; it < 16; ++it) {
        uint32_t far = members[0];
        double far_reach = -1.0;
        for (uint32_t i : members) {
            const double d = reach(i);
            if (d > far_reach) { far_reach = d; far = i; }
        }
        radius = far_reach;
        const double step = 0.5 / (it + 2.0);
        for (int a = 0; a < 3; ++a) c[a] += (items[far].c[a] - c[a]) * step;
    }
    double last = 0;
    for (uint32_t i : members) last = std::max(last, reach(i));
    radius = std::min(radius, last);
    return radius * radius;
}
void improve_partition(const std::vector<Ball>& items, std::vector<std::vector<uint32_t>>& bins, size_t capacity) {
    const size_t k = bins.size();
    if (k < 2) return;
    for (const Ball& b : items)
        if (!(b.r < 1e15)) return; // an unbounded item: nothing to minimise
    const size_t NEIGHBOURS = 6, CANDIDATES = 3, PARTNERS = 4;
    std::vector<double> r2(k);
    std::vector<std::array<double, 3>> centre(k);
    auto refresh = [&](size_t j) {
        r2[j] = bin_radius2(items, bins[j]);
        std::array<double, 3> c = {0, 0, 0};
        for (uint32_t i : bins[j])
            for (int a = 0; a < 3; ++a) c[a] += items[i].c[a];
        for (int a = 0; a < 3; ++a) c[a] /= (double)std::max<size_t>(1, bins[j].size());
        centre[j] = c;
    };
    for (size_t j = 0; j < k; ++j) refresh(j);
    // the members of a bin that reach farthest from its centroid, at most `count`: the ones worth handing over
    auto outermost = [&](size_t j, size_t count) {
        std::vector<std::pair<double, uint32_t>> by_reach;
        for (uint32_t i : bins[j]) {
            double d2 = 0;
            for (int a = 0; a < 3; ++a) d2 += (items[i].c[a] - centre[j][a]) * (items[i].c[a] - centre[j][a]);
            by_reach.push_back({-(std::sqrt(d2) + items[i].r), i});
        }
        std::sort(by_reach.begin(), by_reach.end());
        std::vector<uint32_t> out;
        for (size_t n = 0; n < by_reach.size() && n < count; ++n) out.push_back(by_reach[n].second);
        return out;
    };
    auto without = [](std::vector<uint32_t> v, uint32_t i) {
        v.erase(std::find(v.begin(), v.end(), i));
        return v;
    };
    // (every candidate looks at all k bins for its neighbours: a pass is O(k^2 log k).  Scenes far beyond the ones this was
    // tuned on get fewer passes -- planning time, never a result, depends on it)
    const int passes = k <= 256 ? 8 : k <= 1024 ? 3 : 1;
    for (int pass = 0; pass < passes; ++pass) {
        bool improved = false;
        for (size_t a = 0; a < k; ++a) {
            if (bins[a].size() < 2) continue;
            for (uint32_t i : outermost(a, CANDIDATES)) {
                // the bins nearest to item i
                std::vector<std::pair<double, size_t>> near;
                for (size_t b = 0; b < k; ++b) {
                    if (b == a || bins[b].empty()) continue;
                    double d2 = 0;
                    for (int x = 0; x < 3; ++x) d2 += (items[i].c[x] - centre[b][x]) * (items[i].c[x] - centre[b][x]);
                    near.push_back({d2, b});
                }
                std::sort(near.begin(), near.end());
                double best_gain = 1e-9 * (r2[a] + 1.0);
                size_t best_b = k;
                std::vector<uint32_t> best_a_members, best_b_members;
                const std::vector<uint32_t> a_less = without(bins[a], i);
                for (size_t n = 0; n < near.size() && n < NEIGHBOURS; ++n) {
                    const size_t b = near[n].second;
                    const double before = r2[a] + r2[b];
                    if (bins[b].size() < capacity) {
                        std::vector<uint32_t> b_more = bins[b];
                        b_more.push_back(i);
                        const double gain = before - (bin_radius2(items, a_less) + bin_radius2(items, b_more));
                        if (gain > best_gain) { best_gain = gain; best_b = b; best_a_members = a_less; best_b_members = b_more; }
                    }
                    for (uint32_t j : outermost(b, PARTNERS)) {
                        std::vector<uint32_t> a_new = a_less, b_new = without(bins[b], j);
                        a_new.push_back(j);
                        b_new.push_back(i);
                        const double gain = before - (bin_radius2(items, a_new) + bin_radius2(items, b_new));
                        if (gain > best_gain) { best_gain = gain; best_b = b; best_a_members = a_new; best_b_members = b_new; }
                    }
                }
                if (best_b == k) continue;
                bins[a] = best_a_members;
                bins[best_b] = best_b_members;
                refresh(a);
                refresh(best_b);
                improved = true;
                break; // bin a changed: its candidates are stale
            }
        }
        if (!improved) break;
    }
}

// Bounding sphere of a cluster: centre by a few "move towards the farthest member" steps, radius =
// max(|centre - c_i| + r_i), inflated by 5 % + 0.05 so that neither float rounding in the cull test
// nor the reference's treatment of slightly un-normalised directions (material.rs:246, which its
// sphere test ignores, geometry.rs:207) can make the cull reject a sphere the reference would hit.
RlF4 cluster_bound(const std::vector<SphereIn>& sph, const std::vector<uint32_t>& members) {
    double c[3] = {0, 0, 0};
    for (uint32_t i : members) {
        c[0] += sph[i].rec.x; c[1] += sph[i].rec.y; c[2] += sph[i].rec.z;
    }
    for (int a = 0; a < 3; ++a) c[a] /= (double)members.size();
    auto reach = [&](uint32_t i) {
        const double dx = sph[i].rec.x - c[0], dy = sph[i].rec.y - c[1], dz = sph[i].rec.z - c[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz) + sph[i].radius;
    };
    for (int it = 0; it < 64; ++it) {
        uint32_t far = members[0];
        for (uint32_t i : members)
            if (reach(i) > reach(far)) far = i;
        const double step = 0.5 / (it + 2.0);
        c[0] += (sph[far].rec.x - c[0]) * step; c[1] += (sph[far].rec.y - c[1]) * step; c[2] += (sph[far].rec.z - c[2]) * step;
    }
    double radius = 0;
    for (uint32_t i : members) radius = std::max(radius, reach(i));
    radius = radius * 1.05 + 0.05;
    RlF4 b;
    b.x = (float)c[0]; b.y = (float)c[1]; b.z = (float)c[2];
    b.w = (float)(radius * radius);
    if (!(radius < 1e15)) b.w = std::numeric_limits<float>::infinity();
    return b;
}

// ---- second level of the cull table: groups of G neighbouring bounds (G = RlFlatScene::group_gc clusters or RL_GROUP_GP prisms) ------------------------------

double bound_radius(const RlF4& b) { return b.w > 0.0f ? std::sqrt((double)b.w) : 0.0; } // {c, R^2}; R^2 = +inf -> inf

// Partitions bounds (given as {centre, radius^2}) into groups of at most G: recursive median cuts along the
// longest axis into leaves of exactly G (the last may be short), then a few rounds of capacity-limited
// re-assignment to the nearest group centre (the same scheme as the sphere clusters above).
void split_groups(const std::vector<RlF4>& b, std::vector<uint32_t> idx, std::vector<std::vector<uint32_t>>& out, size_t G) {
    if (idx.size() <= G) {
        if (!idx.empty()) out.push_back(idx);
        return;
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i : idx) {
        const double c[3] = {b[i].x, b[i].y, b[i].z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a]);
            hi[a] = std::max(hi[a], c[a]);
        }
    }
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
    auto coord = [&](uint32_t i) { return axis == 0 ? b[i].x : axis == 1 ? b[i].y : b[i].z; };
    std::sort(idx.begin(), idx.end(), [&](uint32_t p, uint32_t q) { return coord(p) < coord(q) || (coord(p) == coord(q) && p < q); });
    size_t left = ((idx.size() / 2 + G - 1) / G) * G;
    if (left >= idx.size()) left = idx.size() - G;
    split_groups(b, std::vector<uint32_t>(idx.begin(), idx.begin() + left), out, G);
    split_groups(b, std::vector<uint32_t>(idx.begin() + left, idx.end()), out, G);
}

void refine_groups(const std::vector<RlF4>& b, std::vector<std::vector<uint32_t>>& groups, size_t G) {
    const size_t k = groups.size();
    if (k < 2) return;
    std::vector<uint32_t> all;
    for (const auto& g : groups) all.insert(all.end(), g.begin(), g.end());
    std::sort(all.begin(), all.end());
    for (int iteration = 0; iteration < 24; ++iteration) {
        std::vector<double> cx(k, 0), cy(k, 0), cz(k, 0);
        for (size_t j = 0; j < k; ++j) {
            for (uint32_t i : groups[j]) {
                cx[j] += b[i].x; cy[j] += b[i].y; cz[j] += b[i].z;
            }
            const double n = (double)std::max<size_t>(1, groups[j].size());
            cx[j] /= n; cy[j] /= n; cz[j] /= n;
        }
        auto dist2 = [&](uint32_t i, size_t j) {
            const double dx = b[i].x - cx[j], dy = b[i].y - cy[j], dz = b[i].z - cz[j];
            return dx * dx + dy * dy + dz * dz;
        };
        std::vector<std::pair<double, uint32_t>> order;
        for (uint32_t i : all) {
            double best = 1e300, second = 1e300;
            for (size_t j = 0; j < k; ++j) {
                const double d = dist2(i, j);
                if (d < best) { second = best; best = d; }
                else if (d < second) second = d;
            }
            order.push_back({std::sqrt(best) - std::sqrt(second), i});
        }
        std::sort(order.begin(), order.end());
        std::vector<std::vector<uint32_t>> next(k);
        for (const auto& entry : order) {
            const uint32_t i = entry.second;
            size_t pick = k;
            double pick_d = 1e300;
            for (size_t j = 0; j < k; ++j) {
                if (next[j].size() >= G) continue;
                const double d = dist2(i, j);
                if (d < pick_d) { pick_d = d; pick = j; }
            }
            next[pick].push_back(i);
        }
        bool same = true;
        for (size_t j = 0; j < k && same; ++j) {
            std::sort(next[j].begin(), next[j].end());
            std::vector<uint32_t> old = groups[j];
            std::sort(old.begin(), old.end());
            same = old == next[j];
        }
        groups = next;
        if (same) break;
    }
    std::vector<std::vector<uint32_t>> kept;
    for (auto& g : groups)
        if (!g.empty()) kept.push_back(g);
    groups = kept;
}

// Bounding sphere {centre, radius^2} of a group of bounding spheres, inflated by 0.1 % + 1e-3 (the members carry
// their own 5 % already; this only has to cover the rounding of this computation and of the float conversion).
RlF4 group_bound(const std::vector<RlF4>& b, const std::vector<uint32_t>& members) {
    RlF4 r;
    r.x = r.y = r.z = 0.0f;
    r.w = std::numeric_limits<float>::infinity();
    for (uint32_t i : members)
        if (!(b[i].w < 1e30f)) return r; // an unbounded member: the group is always reached
    double c[3] = {0, 0, 0};
    for (uint32_t i : members) {
        c[0] += b[i].x; c[1] += b[i].y; c[2] += b[i].z;
    }
    for (int a = 0; a < 3; ++a) c[a] /= (double)members.size();
    auto reach = [&](uint32_t i) {
        const double dx = b[i].x - c[0], dy = b[i].y - c[1], dz = b[i].z - c[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz) + bound_radius(b[i]);
    };
    for (int it = 0; it < 64; ++it) {
        uint32_t far = members[0];
        for (uint32_t i : members)
            if (reach(i) > reach(far)) far = i;
        const double step = 0.5 / (it + 2.0);
        c[0] += (b[far].x - c[0]) * step; c[1] += (b[far].y - c[1]) * step; c[2] += (b[far].z - c[2]) * step;
    }
    r.x = (float)c[0]; r.y = (float)c[1]; r.z = (float)c[2];
    c[0] = r.x; c[1] = r.y; c[2] = r.z; // measure from the centre as it will be stored
    double radius = 0;
    for (uint32_t i : members) radius = std::max(radius, reach(i));
    radius = radius * 1.001 + 1e-3;
    r.w = (float)(radius * radius);
    if (!(radius < 1e15)) r.w = std::numeric_limits<float>::infinity();
    return r;
}

// Orders `bounds` so that each group's members are consecutive; returns the new order (old indices) and, per
// group, its members as positions in the new order.
std::vector<uint32_t> order_by_groups(const std::vector<RlF4>& bounds, std::vector<std::vector<uint32_t>>* groups_out, size_t G) {
    std::vector<uint32_t> idx(bounds.size());
    std::iota(idx.begin(), idx.end(), 0u);
    std::vector<std::vector<uint32_t>> groups;
    split_groups(bounds, idx, groups, G);
    refine_groups(bounds, groups, G);
    {
        std::vector<Ball> balls;
        for (const RlF4& b : bounds) balls.push_back(Ball{{b.x, b.y, b.z}, bound_radius(b)});
        improve_partition(balls, groups, G);
    }
    std::vector<uint32_t> order;
    for (std::vector<uint32_t>& g : groups) {
        std::sort(g.begin(), g.end());
        order.insert(order.end(), g.begin(), g.end());
    }
    *groups_out = groups;
    return order;
}

// ---- choosing the cluster size and the group size of a scene --------------------------------------------------------
// How well a median-cut partition fits depends on how the scene's sphere count and layout divide (built-in scene: the
// clusters a ray reaches are 1.6 at 14 per cluster, 2.3 at 12, 1.7 at 10), and so does the trade between fewer, wider
// groups and more, tighter ones.  So rl_flatten_scene builds the table for every size in RL_CLUSTER_K_CHOICES x {3, 4}
// clusters per group and keeps the one that costs the kernel least, estimated from the rays of a few hundred sample paths
// with the kernel's measured cost per step (plan_cost).  Which table is chosen never changes a result.
#ifndef RL_SUPER_MIN_GROUPS
#define RL_SUPER_MIN_GROUPS 112 // cluster groups from which the cull table gets its third level (tools/spill_ab.py with RL_SUPER_MIN / RL_SUPER_G:
                                // 88 groups lose 3 % with it, 128-136 gain 0-10 %, 504 gain 76 %; 8 per super is at or near the best of 4 / 8 / 12 / 16 everywhere)
#endif
#ifndef RL_SUPER_G_DEFAULT
#define RL_SUPER_G_DEFAULT 8   // cluster groups per super
#endif
// From how many cluster groups on a table gets its third level, and how many groups a super bound covers (environment: measurement runs).
void super_params(uint32_t* min_groups, uint32_t* sg) {
    *min_groups = RL_SUPER_MIN_GROUPS;
    *sg = RL_SUPER_G_DEFAULT;
    if (const char* e = std::getenv("RL_SUPER_MIN")) *min_groups = (uint32_t)std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RL_SUPER_G")) *sg = (uint32_t)std::min(64, std::max(2, std::atoi(e)));
}
struct ClusterPlan {
    uint32_t k, group_gc;
    std::vector<std::vector<uint32_t>> clusters; // in table order, short groups filled with empty clusters
    std::vector<RlF4> cluster_bounds, group_bounds; // {centre, radius^2}; an empty cluster's bound has radius^2 = -inf
    double cost;
};

// Second level for one candidate: clusters whose bounds are neighbours become consecutive, group_gc per group; a short group is
// filled up with never-reached dummy clusters so that a cluster's number is also its position in the cull table.
ClusterPlan plan_clusters(const std::vector<SphereIn>& sph, const std::vector<std::vector<uint32_t>>& clusters, uint32_t k, uint32_t group_gc) {
    ClusterPlan plan;
    plan.k = k;
    plan.group_gc = group_gc;
    plan.cost = 0;
    std::vector<RlF4> bounds;
    for (const std::vector<uint32_t>& members : clusters) bounds.push_back(cluster_bound(sph, members));
    std::vector<std::vector<uint32_t>> groups;
    order_by_groups(bounds, &groups, group_gc);
    const RlF4 never = RlF4{0.0f, 0.0f, 0.0f, -std::numeric_limits<float>::infinity()};
    for (const std::vector<uint32_t>& g : groups) {
        std::vector<RlF4> members;
        std::vector<uint32_t> all(g.size());
        for (uint32_t k2 : g) {
            plan.clusters.push_back(clusters[k2]);
            plan.cluster_bounds.push_back(bounds[k2]);
            members.push_back(bounds[k2]);
        }
        std::iota(all.begin(), all.end(), 0u);
        plan.group_bounds.push_back(group_bound(members, all));
        for (size_t pad = g.size(); pad < group_gc; ++pad) {
            plan.clusters.push_back(std::vector<uint32_t>());
            plan.cluster_bounds.push_back(never);
        }
    }
    return plan;
}

// Sample rays for plan_cost: the segments of a few hundred paths of THIS scene, traced here with the per-path header the
// kernel is compiled from (every sphere on the direct list, no table yet), {origin.xyz, far}, {direction.xyz, 0} with
// far = the ray parameter of the nearest plane / circle / paraboloid hit -- what the kernel's cull knows when it starts on
// the spheres.  Rays made up from the geometry alone (surface points, uniform directions) rank the plans of the
// 513-object scene wrongly: where the paths go depends on the materials.  Planning input only: nothing of this reaches
// a result, every photon comes from the trace kernel.
std::vector<RlF4> sample_path_rays(const RlFlatScene& fs, const std::vector<SphereIn>& sph) {
    std::vector<RlF4> spheres, objects = fs.objects;
    std::vector<uint32_t> sphere_obj;
    for (const SphereIn& si : sph) {
        objects[si.obj].w = rl_u2f(rl_object_bits(rl_object_surface(rl_f2u(objects[si.obj].w)), rl_object_material(rl_f2u(objects[si.obj].w)), (uint32_t)spheres.size()));
        spheres.push_back(si.rec);
        sphere_obj.push_back(si.obj);
    }
    RlSceneView sv;
    std::memset(&sv, 0, sizeof sv);
    sv.spheres = spheres.data();
    sv.sphere_obj = sphere_obj.data();
    sv.objects = objects.data();
    sv.planes = fs.planes.data();
    sv.parabs = fs.parabs.data();
    sv.prisms = fs.prisms.data();
    sv.n_direct = sv.n_direct_padded = sv.cluster_base = (uint32_t)spheres.size();
    sv.n_planes = (uint32_t)(fs.planes.size() / 2);
    sv.n_parabs = (uint32_t)(fs.parabs.size() / 3);
    sv.n_prisms = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
    sv.n_objects = (uint32_t)objects.size();
    sv.camera_rec = fs.camera_rec.data();
    RlSceneView flat_things = sv; // planes, circles, paraboloids only
    flat_things.n_direct = 0;
    flat_things.n_prisms = 0;
    std::vector<RlF4> rays;
    const uint64_t seed = 0x706c616e6e696e67ull;
    for (uint64_t path = 0; path < 384 && rays.size() < 2 * 2048; ++path) {
        RlPath p;
        rl_begin_path(sv, 16.0f / 9.0f, seed, 0u, path, &p);
        for (int bounce = 0; bounce < 64; ++bounce) {
            const RlHit nearest_flat = rl_scan(flat_things, p.origin, p.direction);
            rays.push_back(RlF4{p.origin.x, p.origin.y, p.origin.z, nearest_flat.t * 1.0002f});
            rays.push_back(RlF4{p.direction.x, p.direction.y, p.direction.z, 0.0f});
            const RlHit hit = rl_scan(sv, p.origin, p.direction);
            float value = 0.0f;
            uint32_t emitter = 0;
            if (rl_bounce(sv, seed, 0u, path, &p, hit, &value, &emitter) != RL_PATH_CONTINUES) break;
        }
    }
    return rays;
}

// Estimated time of the kernel's sphere pass per ray with this plan, in picoseconds of a whole MI355X: every ray tests
// every group bound (wave-uniform); a (group, ray) pair that passes costs its share of a ring-S round (9 cross-lane
// fetches, then `group_gc` tests + compactions), a (cluster, ray) pair its share of a ring-A round (the fetches, then
// a test per member and two or three push steps).  The coefficients are a least-squares fit of the measured throughput
// of the built-in and the 513-object scene over cluster sizes 8 .. 16 x groups of 3 / 4 (20 builds, residual 0.14 ps
// against a spread of 0.65 ps) -- in instructions per 64 rays: 28 per group, 100 + 40 per
// cluster beyond three for a group pair, 12 + 19 per member for a cluster pair.  The exact sphere tests that follow
// depend on the spheres, not on the plan.
double plan_cost(const ClusterPlan& plan, const std::vector<RlF4>& rays) {
    // the segment [0, far] of the ray comes within the bound (the kernel's test without its rounding slack)
    auto reaches = [](const RlF4& b, const RlF4& o, const RlF4& d) {
        if (!(b.w > 0.0f)) return false;
        const double cx = (double)b.x - o.x, cy = (double)b.y - o.y, cz = (double)b.z - o.z;
        const double d2 = (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z;
        const double along = std::min(std::max(0.0, (cx * d.x + cy * d.y + cz * d.z) / d2), (double)o.w);
        const double px = cx - along * d.x, py = cy - along * d.y, pz = cz - along * d.z;
        return px * px + py * py + pz * pz <= (double)b.w;
    };
    const double n_rays = (double)(rays.size() / 2);
    // Round 6: a plan with enough groups gets the third level (rl_flatten_scene, the same rule and the same partition): its rays then
    // test the SUPER bounds wave-uniformly and a super's groups only where the super is reached -- a ring-T round per 64
    // (super, ray) pairs, in the units below 2.3 + 0.9 per group: calibrated on two measurements (tools/spill_ab.py with RL_SUPER_MIN:
    // 88 groups in 11 supers cost 3 % more than the 88 wave-uniform tests, 136 groups in 17 supers 10 % less).  Without this the estimate charged every ray for
    // every group of a large scene, and scenes of 7-8 k objects got clusters of 14 where 10 run 7-9 % faster (tools/spill_ab.py, RL_PLAN).
    uint32_t min_groups, sg;
    super_params(&min_groups, &sg);
    const size_t n_groups = plan.group_bounds.size();
    std::vector<std::vector<uint32_t>> supers;
    std::vector<RlF4> super_bounds;
    if (n_groups >= min_groups && n_groups > sg) {
        order_by_groups(plan.group_bounds, &supers, sg);
        for (const std::vector<uint32_t>& members : supers) super_bounds.push_back(group_bound(plan.group_bounds, members));
    } else {
        supers.assign(1, std::vector<uint32_t>()); // one pseudo-super that holds every group and is always "reached"
        for (uint32_t g = 0; g < n_groups; ++g) supers[0].push_back(g);
    }
    double super_pairs = 0, group_pairs = 0, cluster_pairs = 0;
    for (size_t r = 0; r + 1 < rays.size(); r += 2) {
        for (size_t s = 0; s < supers.size(); ++s) {
            if (!super_bounds.empty()) {
                if (!reaches(super_bounds[s], rays[r], rays[r + 1])) continue;
                super_pairs += 1;
            }
            for (uint32_t g : supers[s]) {
                if (!reaches(plan.group_bounds[g], rays[r], rays[r + 1])) continue;
                group_pairs += 1;
                for (size_t k = plan.group_gc * g; k < plan.group_gc * (g + 1); ++k)
                    if (reaches(plan.cluster_bounds[k], rays[r], rays[r + 1])) cluster_pairs += 1;
            }
        }
    }
    const double top = super_bounds.empty() ? 0.59 * (double)n_groups
                                            : 0.59 * (double)super_bounds.size() + super_pairs / n_rays * (2.3 + 0.9 * (double)sg);
    const double cost = top + group_pairs / n_rays * (2.13 + 0.83 * ((double)plan.group_gc - 3.0)) +
                        cluster_pairs / n_rays * (0.25 + 0.40 * plan.k);
#ifdef RL_PLAN_DEBUG
    std::fprintf(stderr, "plan: %u per cluster, %u per group: %zu groups, %.2f group pairs and %.2f cluster pairs per sample ray, cost %.2f\n", plan.k,
                 plan.group_gc, plan.group_bounds.size(), group_pairs / n_rays, cluster_pairs / n_rays, cost);
#endif
    return cost;
}

} // namespace

uint32_t rl_builtin_scene(int which, int param, std::vector<RlObjectDesc>* out, RlCameraDesc* camera) {
    std::vector<RlObjectDesc> objs;
    if (which == RL_SCENE_DEMO) demo_scene(param > 0 ? param : 100, objs);
    else if (which == RL_SCENE_GLASS_STRESS) glass_stress_scene(objs);
    else return 0;
    if (camera) *camera = demo_camera();
    if (out) *out = objs;
    return (uint32_t)objs.size();
}

static void set_group(RlF4& object, uint32_t group_index) {
    const uint32_t bits = rl_f2u(object.w);
    object.w = rl_u2f(rl_object_bits(rl_object_surface(bits), rl_object_material(bits), group_index));
}

int rl_flatten_scene(const RlSceneDesc* desc, RlFlatScene* out, const char** err) {
    *err = "";
    if (!desc || (!desc->objects && desc->n_objects)) {
        *err = "null scene description";
        return RL_E_INVALID;
    }
    if (desc->n_objects > (1u << 24)) { // object indices travel in 24-bit fields (merge keys, the emitter queue's tag)
        *err = "more than 2^24 objects";
        return RL_E_INVALID;
    }
    RlFlatScene& fs = *out;
    fs = RlFlatScene();
    std::vector<SphereIn> sph_in;
    fs.camera = desc->camera;
    const float fov = PI * desc->camera.fov_over_pi;
    fs.screen_distance = 1.0f / rl_tanf(fov * 0.5f); // camera.rs:56
    fs.camera_rec.assign(3, RlF4{0.0f, 0.0f, 0.0f, 0.0f});
    std::memcpy(fs.camera_rec.data(), &fs.camera, sizeof(RlCameraDesc));
    reinterpret_cast<float*>(fs.camera_rec.data())[10] = fs.screen_distance;
    for (uint32_t i = 0; i < desc->n_objects; ++i) {
        const RlObjectDesc& o = desc->objects[i];
        const float objbits = rl_u2f(i);
        uint32_t group_index = 0;
        switch (o.surface_kind) {
        case RL_SURFACE_SPHERE: {
            SphereIn si;
            si.rec = F4(F(o.v0), o.f0 * o.f0); // geometry.rs:195-200
            si.obj = i;
            si.radius = std::fabs((double)o.f0);
            sph_in.push_back(si);
            group_index = 0; // patched below, once the record's final position is known
            break;
        }
        case RL_SURFACE_PLANE:
        case RL_SURFACE_CIRCLE:
            group_index = (uint32_t)(fs.planes.size() / 2);
            fs.planes.push_back(F4(F(o.v0), o.surface_kind == RL_SURFACE_CIRCLE ? o.f0 * o.f0 : -1.0f));
            fs.planes.push_back(F4(F(o.v1), objbits));
            break;
        case RL_SURFACE_PARABOLOID: {
            group_index = (uint32_t)(fs.parabs.size() / 3);
            const ParabFields p = paraboloid_fields(F(o.v0), F(o.v1), o.f0);
            fs.parabs.push_back(F4(p.offset, objbits));
            fs.parabs.push_back(F4(p.normal, 0.0f));
            fs.parabs.push_back(F4(p.focal_point, 0.0f));
            break;
        }
        case RL_SURFACE_HEX_PRISM: { // geometry.rs:493-515
            group_index = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
            const RlF3 axis = F(o.v0), offset = F(o.v1);
            const float edge_length = o.f0, bevel_size = o.f1, angle = o.f2, height = o.f3;
            push_infinite_prism(fs.prisms, axis, offset, edge_length * 2.0f - bevel_size * 3.0f, angle + PI, objbits);
            push_infinite_prism(fs.prisms, axis, offset, edge_length, angle, objbits);
            fs.prisms.push_back(F4(rl_neg(axis), 0.0f)); // new_thick_plane, geometry.rs:455-468
            fs.prisms.push_back(F4(offset, objbits));
            fs.prisms.push_back(F4(axis, 0.0f));
            fs.prisms.push_back(F4(rl_add(offset, rl_mul(axis, height)), objbits));
            fs.prisms.push_back(prism_bound(&fs.prisms[fs.prisms.size() - 16]));
            { // scale constants of rl_hex_prism_fast (rl_core.h), in the unused fourth components of two normal records
                RlF4* pr = &fs.prisms[fs.prisms.size() - RL_PRISM_STRIDE];
                float off1 = 0.0f, n1 = 1.0f;
                for (int k = 0; k < 8; ++k) {
                    n1 = std::max(n1, std::fabs(pr[2 * k].x) + std::fabs(pr[2 * k].y) + std::fabs(pr[2 * k].z));
                    off1 = std::max(off1, std::fabs(pr[2 * k + 1].x) + std::fabs(pr[2 * k + 1].y) + std::fabs(pr[2 * k + 1].z));
                }
                pr[0].w = off1 * 1.000001f;
                pr[2].w = n1 * 1.000001f;
            }
            break;
        }
        default:
            *err = "unknown surface kind";
            return RL_E_INVALID;
        }
        RlF4 b;
        b.x = o.m0; b.y = o.m1; b.z = o.m2;
        b.w = rl_u2f(rl_object_bits(o.surface_kind, o.material_kind, group_index));
        if (o.material_kind == RL_MATERIAL_BLACK_BODY) b.y = rl_black_body_normalisation(o.m0, o.m1);
        else if (o.material_kind > RL_MATERIAL_SOAP_BUBBLE) {
            *err = "unknown material kind";
            return RL_E_INVALID;
        }
        fs.objects.push_back(b);
    }
    // ---- spheres: direct list + clusters ----
    // Direct: spheres much larger than the typical one (a bound around them would cull nothing), and
    // everything when the scene has too few spheres for clusters to pay.
    std::vector<uint32_t> direct, clustered;
    {
        std::vector<double> radii;
        for (const SphereIn& si : sph_in) radii.push_back(si.radius);
        double median = 0;
        if (!radii.empty()) {
            std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
            median = radii[radii.size() / 2];
        }
        const bool use_clusters = sph_in.size() >= 40;
        // ... but only a handful: every ray tests every direct sphere.  A scene whose radii are spread widely (the built-in
        // generator with 1,500 seeds: 1,213 of 4,511 spheres are more than four medians wide) used to put a quarter of its
        // spheres here -- 72 % of the kernel's time (round 4, tools/kernel_stats.py: 1.8 Grays/s, 6.8 since) -- so beyond the
        // RL_DIRECT_MAX largest the large ones are clustered like the rest.
        const size_t RL_DIRECT_MAX = 8;
        std::vector<std::pair<double, uint32_t>> large;
        for (uint32_t k = 0; k < sph_in.size(); ++k)
            if (use_clusters && std::isfinite(sph_in[k].radius) && sph_in[k].radius > 4.0 * median) large.push_back({-sph_in[k].radius, k});
        std::sort(large.begin(), large.end());
        std::vector<char> is_direct(sph_in.size(), 0);
        for (size_t n = 0; n < large.size() && n < RL_DIRECT_MAX; ++n) is_direct[large[n].second] = 1;
        for (uint32_t k = 0; k < sph_in.size(); ++k) {
            if (!use_clusters || is_direct[k] || !std::isfinite(sph_in[k].radius)) direct.push_back(k);
            else clustered.push_back(k);
        }
    }
    RlF4 dummy; // can never be hit: c = |co|^2 - (-inf) = +inf, q = -inf < 0
    dummy.x = dummy.y = dummy.z = 0.0f;
    dummy.w = -std::numeric_limits<float>::infinity();
    auto place = [&](uint32_t k) {
        const uint32_t pos = (uint32_t)fs.spheres.size();
        fs.spheres.push_back(sph_in[k].rec);
        fs.sphere_obj.push_back(sph_in[k].obj);
        set_group(fs.objects[sph_in[k].obj], pos); // group index = record position
    };
    for (uint32_t k : direct) place(k);
    fs.n_direct = (uint32_t)direct.size();
    fs.n_direct_padded = (fs.n_direct + 3u) & ~3u;
    fs.spheres.resize(fs.n_direct_padded + 4, dummy);
    fs.sphere_obj.resize(fs.spheres.size(), RL_HIT_NONE);
    fs.cluster_base = (uint32_t)fs.spheres.size();
    // Cluster size and group size: the plan with the lowest estimated cost per ray (plan_clusters, plan_cost).
    ClusterPlan plan;
    plan.k = 10;
    plan.group_gc = 3;
    if (!clustered.empty()) {
#ifdef RL_CLUSTER_K
        static_assert(RL_CLUSTER_K >= 2 && RL_CLUSTER_K <= RL_CLUSTER_K_MAX, "cluster size out of range");
        const std::vector<uint32_t> sizes = {RL_CLUSTER_K};
#else
        const std::vector<uint32_t> sizes = RL_CLUSTER_K_CHOICES;
#endif
        const std::vector<RlF4> rays = sample_path_rays(fs, sph_in);
        bool first = true;
        for (uint32_t k : sizes) {
            std::vector<std::vector<uint32_t>> clusters;
            split_clusters(sph_in, clustered, clusters, k);
            refine_clusters(sph_in, clusters, k);
            {
                std::vector<Ball> balls;
                for (const SphereIn& si : sph_in) balls.push_back(Ball{{si.rec.x, si.rec.y, si.rec.z}, si.radius});
                improve_partition(balls, clusters, k);
            }
#ifdef RL_GROUP_GC
            static_assert(RL_GROUP_GC == 3 || RL_GROUP_GC == 4, "the kernel's unrolled ring-S rounds test three bounds of a group unconditionally and a fourth if there is one");
            for (uint32_t g : {(uint32_t)RL_GROUP_GC}) {
#else
            for (uint32_t g : {3u, 4u}) {
#endif
                ClusterPlan candidate = plan_clusters(sph_in, clusters, k, g);
                candidate.cost = plan_cost(candidate, rays);
                // (RL_PLAN="k,g": measurement runs force one plan -- tools/plan_ab.sh -- to check the cost model against the kernel as it is now)
                if (const char* forced = std::getenv("RL_PLAN")) {
                    unsigned fk = 0, fg = 0;
                    if (std::sscanf(forced, "%u,%u", &fk, &fg) == 2 && fk == k && fg == g) candidate.cost = -1.0;
                }
                if (std::getenv("RL_PLAN_VERBOSE")) std::fprintf(stderr, "rl: plan k=%u g=%u: %zu clusters, estimated cost %.4f\n", k, g, candidate.clusters.size(), candidate.cost);
                if (first || candidate.cost < plan.cost * 0.995) plan = candidate; // (a later, larger plan has to win by more than the estimate's noise)
                first = false;
            }
        }
    }
    fs.cluster_k = plan.k;
    fs.group_gc = plan.group_gc;
    // Third level: with many cluster groups (a scene of thousands of spheres: every ray tests every group bound, 28 instructions each)
    // the groups are ordered so that super_g consecutive ones are spatial neighbours -- the same partition scheme one level up --
    // and get a bound of their own.  From RL_SUPER_MIN_GROUPS on (measured): such a scene is far too large to be staged whole in LDS, and
    // only the instantiations that stage the tables or nothing carry the third level's code.  (RL_SUPER_MIN / RL_SUPER_G:
    // measurement runs.)  Which table is chosen never changes a result.
    {
        uint32_t min_groups, sg;
        super_params(&min_groups, &sg);
        const size_t n_groups = plan.clusters.size() / plan.group_gc;
        if (n_groups >= min_groups && n_groups > sg) {
            std::vector<std::vector<uint32_t>> supers;
            order_by_groups(plan.group_bounds, &supers, sg);
            std::vector<std::vector<uint32_t>> ordered;
            for (const std::vector<uint32_t>& members : supers) {
                for (uint32_t g : members)
                    for (uint32_t c = 0; c < plan.group_gc; ++c) ordered.push_back(plan.clusters[(size_t)plan.group_gc * g + c]);
                for (size_t pad = members.size(); pad < sg; ++pad)
                    for (uint32_t c = 0; c < plan.group_gc; ++c) ordered.push_back(std::vector<uint32_t>()); // a dummy group of dummy clusters
            }
            plan.clusters = ordered;
            fs.n_cluster_supers = (uint32_t)supers.size();
            fs.super_g = sg;
        }
    }
    const RlF4 never = dummy; // as a bound {c, R^2 = -inf}: fails every cull test, host and device
    std::vector<std::vector<uint32_t>>& clusters = plan.clusters;
    for (std::vector<uint32_t>& members : clusters) {
        std::sort(members.begin(), members.end()); // ascending object order inside a cluster
        fs.spheres.push_back(members.empty() ? never : cluster_bound(sph_in, members));
        fs.sphere_obj.push_back(RL_HIT_NONE);
        for (uint32_t k : members) place(k);
        for (size_t pad = members.size(); pad < fs.cluster_k; ++pad) {
            fs.spheres.push_back(dummy);
            fs.sphere_obj.push_back(RL_HIT_NONE);
        }
    }
    fs.n_clusters = (uint32_t)clusters.size();
    fs.n_cluster_groups = fs.n_clusters / fs.group_gc;
    // The prisms likewise: reorder the records (scan order is irrelevant: ties are broken by the object index each
    // record carries), fill short groups with dummy prisms, point the prism objects at their new position.
    uint32_t n_prisms = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
    if (n_prisms != 0) {
        std::vector<RlF4> bounds;
        for (uint32_t i = 0; i < n_prisms; ++i) bounds.push_back(fs.prisms[RL_PRISM_STRIDE * i + 16]);
        std::vector<std::vector<uint32_t>> groups;
        order_by_groups(bounds, &groups, RL_GROUP_GP);
        std::vector<RlF4> sorted;
        for (const std::vector<uint32_t>& g : groups) {
            for (uint32_t k : g) {
                const RlF4* pr = &fs.prisms[RL_PRISM_STRIDE * k];
                set_group(fs.objects[rl_f2u(pr[1].w)], (uint32_t)(sorted.size() / RL_PRISM_STRIDE)); // group index = position
                sorted.insert(sorted.end(), pr, pr + RL_PRISM_STRIDE);
            }
            for (size_t pad = g.size(); pad < RL_GROUP_GP; ++pad) {
                sorted.resize(sorted.size() + RL_PRISM_STRIDE - 1, RlF4{0.0f, 0.0f, 0.0f, 0.0f});
                sorted.push_back(never);
            }
        }
        fs.prisms = sorted;
        n_prisms = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
    }
    fs.n_prism_groups = n_prisms / RL_GROUP_GP;
    // second bound per prism (in the prisms' final order); worth its 26 instructions per bound test only when the pairs that
    // pass the spheres fill more than one round per iteration: from ~40 prisms on (the glass-stress scene: 66, 2.2 pairs per
    // ray; the built-in scene: 22, 0.6)
    uint32_t real_prisms = 0;
    fs.prism_cyl.clear();
    for (uint32_t i = 0; i < n_prisms; ++i) {
        RlF4 cyl[2];
        prism_cylinder(&fs.prisms[RL_PRISM_STRIDE * i], cyl);
        fs.prism_cyl.push_back(cyl[0]);
        fs.prism_cyl.push_back(cyl[1]);
        if (std::isfinite(fs.prisms[RL_PRISM_STRIDE * i + 16].w) && fs.prisms[RL_PRISM_STRIDE * i + 16].w > 0.0f) real_prisms += 1;
    }
    fs.prism_cylinders = real_prisms >= 40u;
    if (!fs.prism_cylinders) fs.prism_cyl.clear();
    // Cull table for the kernel, {c, |c|^2 - R^2} per bound: clusters, prisms, then one group bound per group_gc
    // clusters and per RL_GROUP_GP prisms.
    fs.cull_cmax2 = 0.0f;
    auto add_bound = [&](const RlF4& b) {
        const double c2 = (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z;
        RlF4 r = b;
        r.w = (float)(c2 - (double)b.w); // -inf radius^2 (dummy) -> +inf: never reached; +inf (unbounded) -> -inf: always
        fs.cull_bounds.push_back(r);
        if (std::isfinite(b.w)) fs.cull_cmax2 = std::max(fs.cull_cmax2, (float)c2 * 1.0001f);
    };
    std::vector<RlF4> level1;
    for (uint32_t k = 0; k < fs.n_clusters; ++k) level1.push_back(fs.spheres[fs.cluster_base + (fs.cluster_k + 1u) * k]);
    for (uint32_t i = 0; i < n_prisms; ++i) level1.push_back(fs.prisms[RL_PRISM_STRIDE * i + 16]);
    for (const RlF4& b : level1) add_bound(b);
    const size_t n_cluster_level1 = (size_t)fs.group_gc * fs.n_cluster_groups;
    std::vector<RlF4> level2; // the group bounds as {centre, radius^2}
    for (size_t first = 0; first < level1.size();) {
        const size_t G = first < n_cluster_level1 ? fs.group_gc : RL_GROUP_GP;
        std::vector<uint32_t> members;
        for (size_t j = first; j < first + G; ++j)
            if (!(level1[j].w == never.w)) members.push_back((uint32_t)j);
        level2.push_back(members.empty() ? never : group_bound(level1, members));
        add_bound(level2.back());
        first += G;
    }
    for (uint32_t s = 0; s < fs.n_cluster_supers; ++s) { // third level: one bound per super_g cluster groups
        std::vector<uint32_t> members;
        for (uint32_t g = fs.super_g * s; g < fs.super_g * (s + 1u); ++g)
            if (!(level2[g].w == never.w)) members.push_back(g);
        add_bound(members.empty() ? never : group_bound(level2, members));
    }
    fs.cull_bounds.push_back(dummy); // slack for the kernel's prefetch
    fs.cull_bounds.push_back(dummy);
    {   // RlFlatScene::small_ordered (both lists were appended object by object: checked all the same)
        uint32_t last_parab = 0, first_plane = 0xffffffffu;
        for (size_t i = 0; i < fs.parabs.size(); i += 3) last_parab = std::max(last_parab, rl_f2u(fs.parabs[i].w));
        for (size_t i = 1; i < fs.planes.size(); i += 2) first_plane = std::min(first_plane, rl_f2u(fs.planes[i].w));
        bool sorted = fs.parabs.empty() || fs.planes.empty() || last_parab < first_plane;
        for (size_t i = 3; i < fs.parabs.size(); i += 3) sorted = sorted && rl_f2u(fs.parabs[i - 3].w) < rl_f2u(fs.parabs[i].w);
        for (size_t i = 3; i < fs.planes.size(); i += 2) sorted = sorted && rl_f2u(fs.planes[i - 2].w) < rl_f2u(fs.planes[i].w);
        fs.small_ordered = sorted;
        bool axis_z = !(fs.parabs.empty() && fs.planes.empty());
        for (size_t i = 1; i < fs.parabs.size(); i += 3) axis_z = axis_z && fs.parabs[i].x == 0.0f && fs.parabs[i].y == 0.0f;
        for (size_t i = 0; i < fs.planes.size(); i += 2) axis_z = axis_z && fs.planes[i].x == 0.0f && fs.planes[i].y == 0.0f;
        fs.small_axis_z = axis_z;
    }
    // The clustered spheres in the cull's form: the fourth component a cluster-member record carries on the DEVICE
    // instead of radius^2 (rl_api.hip swaps it in; the exact radius^2 goes to a separate float array that only the exact
    // tail reads).  The cluster-member rounds of the kernel use it as a conservative pre-test -- the reference arithmetic
    // (geometry.rs:204-240) runs afterwards, for the pairs that pass -- so the radius only has to cover the difference
    // between the reference's float discriminant and the geometric one (the cull's own slack term does,
    // rl_kernels.hip.h) plus a relative 1e-3.  Bound records, dummies and direct spheres get +inf: never reached.
    fs.sphere_cull_w.assign(fs.spheres.size(), std::numeric_limits<float>::infinity());
    for (size_t pos = fs.cluster_base; pos < fs.spheres.size(); ++pos) {
        const RlF4& sp = fs.spheres[pos];
        if (fs.sphere_obj[pos] == RL_HIT_NONE || !std::isfinite(sp.w) || !(sp.w >= 0.0f)) continue;
        const double c2 = (double)sp.x * sp.x + (double)sp.y * sp.y + (double)sp.z * sp.z;
        fs.sphere_cull_w[pos] = (float)(c2 - ((double)sp.w * 1.001 + 1.0e-6));
        fs.cull_cmax2 = std::max(fs.cull_cmax2, (float)c2 * 1.0001f);
    }
    return RL_OK;
}
