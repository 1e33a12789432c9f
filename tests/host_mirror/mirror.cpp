// mirror.cpp -- TEST INFRASTRUCTURE: compiles the kernel's per-path code (csrc/rl_core.h) with g++
// so the flattened device arithmetic can be compared bit-for-bit with the oracle on a machine
// without a GPU.  Never loaded by the product; the product's compute entry points run only the
// hipcc build of the same header.
#include <cstring>
#include <vector>

#include "../../robigo_luculenta_amd/csrc/rl_cie1931.h"
#include "../../robigo_luculenta_amd/csrc/rl_core.h"
#include "../../robigo_luculenta_amd/csrc/rl_scene.h"

struct MirrorScene {
    RlFlatScene flat;
    RlSceneView view;
};

extern "C" {

void* mirror_scene_create(const RlObjectDesc* objs, uint32_t n, const RlCameraDesc* cam) {
    RlSceneDesc d;
    d.n_objects = n;
    d.objects = objs;
    d.camera = *cam;
    MirrorScene* m = new MirrorScene();
    const char* err;
    if (rl_flatten_scene(&d, &m->flat, &err) != 0) {
        delete m;
        return nullptr;
    }
    RlSceneView& v = m->view;
    v.spheres = m->flat.spheres.data();
    v.planes = m->flat.planes.data();
    v.parabs = m->flat.parabs.data();
    v.prisms = m->flat.prisms.data();
    v.objects = m->flat.objects.data();
    v.sphere_obj = m->flat.sphere_obj.data();
    v.sphere_r2 = nullptr; // host view: every sphere record is {centre, radius^2}
    v.cie = (const RlF4*)RL_CIE1931_XYZ0;
    v.n_direct = m->flat.n_direct;
    v.n_direct_padded = m->flat.n_direct_padded;
    v.cluster_base = m->flat.cluster_base;
    v.n_clusters = m->flat.n_clusters;
    v.cluster_k = m->flat.cluster_k;
    v.n_planes = (uint32_t)(m->flat.planes.size() / 2);
    v.n_parabs = (uint32_t)(m->flat.parabs.size() / 3);
    v.n_prisms = (uint32_t)(m->flat.prisms.size() / RL_PRISM_STRIDE);
    v.n_objects = (uint32_t)m->flat.objects.size();
    v.camera_rec = m->flat.camera_rec.data();
    return m;
}
void mirror_scene_destroy(void* s) { delete (MirrorScene*)s; }

uint32_t mirror_builtin_desc(int which, int param, RlObjectDesc* out, uint32_t cap, RlCameraDesc* cam) {
    std::vector<RlObjectDesc> v;
    uint32_t n = rl_builtin_scene(which, param, &v, cam);
    if (out && cap >= n) memcpy(out, v.data(), n * sizeof(RlObjectDesc));
    return n;
}

void mirror_render(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first, uint64_t n,
                   RlMappedPhoton* photons, uint64_t* segments) {
    const RlSceneView& sv = ((MirrorScene*)scene)->view;
    const float aspect = (float)w / (float)h;
    uint64_t segs = 0;
    for (uint64_t i = 0; i < n; ++i) {
        RlPath p;
        rl_begin_path(sv, aspect, seed, stream, first + i, &p);
        float value = 0.0f;
        for (;;) {
            const RlHit hit = rl_scan(sv, p.origin, p.direction);
            segs += 1;
            uint32_t emitter = 0;
            const int status = rl_bounce(sv, seed, stream, first + i, &p, hit, &value, &emitter);
            if (status == RL_PATH_ENDED_ON_EMITTER) value = rl_emission(sv, p.intensity, p.wavelength, emitter);
            if (status != RL_PATH_CONTINUES) break;
        }
        photons[i].x = p.sx;
        photons[i].y = p.sy;
        photons[i].probability = value;
        photons[i].wavelength = p.wavelength;
    }
    if (segments) *segments = segs;
}

// plot_unit.rs:87-95 through the kernel's lookup + weights, sequential adds.
void mirror_plot(RlVector3* buffer, uint32_t w, uint32_t h, const RlMappedPhoton* photons, uint64_t n) {
    const float aspect = (float)w / (float)h;
    const RlF4* cie = (const RlF4*)RL_CIE1931_XYZ0;
    for (uint64_t i = 0; i < n; ++i) {
        const RlF3 t = rl_tristimulus(cie, photons[i].wavelength);
        const RlF3 c = rl_mul(t, photons[i].probability);
        const RlSplat s = rl_splat_weights(w, h, aspect, photons[i].x, photons[i].y);
        for (int k = 0; k < 4; ++k) {
            RlVector3& b = buffer[s.idx[k]];
            b.x = b.x + c.x * s.w[k];
            b.y = b.y + c.y * s.w[k];
            b.z = b.z + c.z * s.w[k];
        }
    }
}
}

// Dumps the rays (origin, direction) of every segment of paths [first, first+n): analysis helper for
// sizing the kernel's culling structures.  Returns the number of rays written (<= cap).
extern "C" uint64_t mirror_dump_rays(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first,
                                     uint64_t n, float* rays6, uint64_t cap) {
    const RlSceneView& sv = ((MirrorScene*)scene)->view;
    const float aspect = (float)w / (float)h;
    uint64_t count = 0;
    for (uint64_t i = 0; i < n; ++i) {
        RlPath p;
        rl_begin_path(sv, aspect, seed, stream, first + i, &p);
        float value = 0.0f;
        for (;;) {
            if (count < cap) {
                float* r = rays6 + 6 * count;
                r[0] = p.origin.x; r[1] = p.origin.y; r[2] = p.origin.z;
                r[3] = p.direction.x; r[4] = p.direction.y; r[5] = p.direction.z;
                count++;
            }
            const RlHit hit = rl_scan(sv, p.origin, p.direction);
            uint32_t emitter = 0;
            if (rl_bounce(sv, seed, stream, first + i, &p, hit, &value, &emitter) != RL_PATH_CONTINUES) break;
        }
    }
    return count;
}

// The kernel's conservative bounds {centre, radius^2}: sphere clusters first, then prisms (analysis helper).
extern "C" uint32_t mirror_bounds(void* scene, float* out4, uint32_t cap, uint32_t* n_clusters) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    uint32_t n = 0;
    auto put = [&](const RlF4& b) {
        if (n < cap) {
            out4[4 * n] = b.x; out4[4 * n + 1] = b.y; out4[4 * n + 2] = b.z; out4[4 * n + 3] = b.w;
        }
        n++;
    };
    for (uint32_t k = 0; k < fs.n_clusters; ++k) put(fs.spheres[fs.cluster_base + (fs.cluster_k + 1u) * k]);
    if (n_clusters) *n_clusters = fs.n_clusters;
    for (size_t i = 0; i < fs.prisms.size() / RL_PRISM_STRIDE; ++i) put(fs.prisms[RL_PRISM_STRIDE * i + 16]);
    return n;
}

// The second level of the cull table: one bound {centre, radius^2} per group of consecutive level-1 bounds -- RlFlatScene::group_gc
// clusters or RL_GROUP_GP prisms -- (reconstructed from the {c, |c|^2 - R^2} records the kernel reads).  Returns the
// number of groups; sizes[0..2] = {clusters per group, prisms per group, number of cluster groups}.
extern "C" uint32_t mirror_group_bounds(void* scene, float* out4, uint32_t cap, uint32_t* sizes) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    const uint32_t n_level1 = fs.group_gc * fs.n_cluster_groups + RL_GROUP_GP * fs.n_prism_groups;
    const uint32_t n_groups = fs.n_cluster_groups + fs.n_prism_groups;
    if (sizes) {
        sizes[0] = fs.group_gc;
        sizes[1] = RL_GROUP_GP;
        sizes[2] = fs.n_cluster_groups;
    }
    for (uint32_t g = 0; g < n_groups && g < cap; ++g) {
        const RlF4 r = fs.cull_bounds[n_level1 + g];
        const double c2 = (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z;
        out4[4 * g] = r.x; out4[4 * g + 1] = r.y; out4[4 * g + 2] = r.z;
        out4[4 * g + 3] = (float)(c2 - (double)r.w);
    }
    return n_groups;
}

// The third level (round 6): one bound {centre, radius^2} per RlFlatScene::super_g consecutive cluster groups, behind the group
// bounds in the table.  Returns the number of supers (0: a two-level table); *super_g_out = groups per super.
extern "C" uint32_t mirror_super_bounds(void* scene, float* out4, uint32_t cap, uint32_t* super_g_out) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    const uint32_t first = fs.group_gc * fs.n_cluster_groups + RL_GROUP_GP * fs.n_prism_groups + fs.n_cluster_groups + fs.n_prism_groups;
    if (super_g_out) *super_g_out = fs.super_g;
    for (uint32_t s = 0; s < fs.n_cluster_supers && s < cap; ++s) {
        const RlF4 r = fs.cull_bounds[first + s];
        const double c2 = (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z;
        out4[4 * s] = r.x; out4[4 * s + 1] = r.y; out4[4 * s + 2] = r.z;
        out4[4 * s + 3] = (float)(c2 - (double)r.w);
    }
    return fs.n_cluster_supers;
}

// rl_paraboloid_t<AXIS_Z> / rl_plane_t<AXIS_Z> against the general forms for normals along z (round 6): random and adversarial
// rays -- direction or origin components that are exact zeros of either sign, origins at the primitive's own height, rays along
// the axis -- and normals (+-0, +-0, +-1).  Equivalent = the scan's use of the result is the same: both "no hit" (t < 0, NaN, or not
// below the scan's initial 1e12) or the same float.  counts[0] = cases, [1] = hits compared bit for bit, [2] = differences (must be 0),
// [3] = cases in which n.d or n.o is a zero (where the two forms may differ in that zero's sign).
extern "C" void mirror_axis_z_check(uint64_t seed, uint64_t n, uint64_t* counts) {
    for (int i = 0; i < 4; ++i) counts[i] = 0;
    uint64_t s = seed;
    auto next = [&]() {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto unit = [&]() { return (float)(next() >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
    auto comp = [&](float scale) { // a component: mostly random, sometimes an exact zero of either sign or a tiny number
        const uint64_t k = next() % 16;
        if (k == 0) return 0.0f;
        if (k == 1) return -0.0f;
        if (k == 2) return 1.0e-30f * unit();
        return scale * unit();
    };
    auto used = [](float t) { return !(t < 0.0f) && t < 1.0e12f; }; // what rl_scan_wave does with a paraboloid's t
    for (uint64_t i = 0; i < n; ++i) {
        const float nz = (next() & 1) ? 1.0f : -1.0f;
        const RlF3 normal = rl_f3((next() & 1) ? 0.0f : -0.0f, (next() & 1) ? 0.0f : -0.0f, nz);
        const RlF3 offset = rl_f3(comp(30.0f), comp(30.0f), comp(30.0f));
        const RlF3 focal = rl_f3(comp(30.0f), comp(30.0f), comp(30.0f));
        RlF3 o = rl_f3(comp(60.0f), comp(60.0f), comp(60.0f));
        if (next() % 8 == 0) o.z = offset.z; // at the primitive's own height: n.o is a zero
        RlF3 d = rl_f3(comp(1.0f), comp(1.0f), comp(1.0f));
        if (next() % 16 == 0) d = rl_f3((next() & 1) ? 0.0f : -0.0f, (next() & 1) ? 0.0f : -0.0f, (next() & 1) ? 1.0f : -1.0f); // along the axis
        counts[0] += 2;
        const float lo_z = o.z - offset.z;
        if (d.z == 0.0f || lo_z == 0.0f) counts[3] += 1;
        {
            const float a = rl_paraboloid_t<false>(offset, normal, focal, o, d), b = rl_paraboloid_t<true>(offset, normal, focal, o, d);
            if (used(a) != used(b)) counts[2] += 1;
            else if (used(a)) {
                counts[1] += 1;
                if (rl_f2u(a) != rl_f2u(b)) counts[2] += 1;
            }
        }
        {
            float dn_a, dn_b;
            const float a = rl_plane_t<false>(normal, offset, o, d, &dn_a), b = rl_plane_t<true>(normal, offset, o, d, &dn_b);
            const bool hit_a = a > 0.0f, hit_b = b > 0.0f; // what the plane / circle code does with it
            if (hit_a != hit_b) counts[2] += 1;
            else if (hit_a) {
                counts[1] += 1;
                if (rl_f2u(a) != rl_f2u(b)) counts[2] += 1;
            }
        }
    }
}

// The paraboloid's one-division form (rl_paraboloid_t, device) against the reference's two-quotient selection (rl_paraboloid_roots) for
// coefficients over the whole exponent range, under the round-6 condition "a < 0 and (disc < 0 or max(|b|, sqrt|disc|) >= 2^-90)":
// wherever the condition holds the two must agree (no hit, or the same float).  The device evaluates the identical expressions (its
// short square root is the IEEE one, tests/test_gpu_parity.py).  counts[0] = cases, [1] = cases the condition admits, [2] = of those:
// hits, [3] = disagreements (must be 0), [4] = cases with a numerator below 2^-100 that the condition rejects (the reason it exists).
extern "C" void mirror_parab_check(uint64_t seed, uint64_t n, uint64_t* counts) {
    for (int i = 0; i < 5; ++i) counts[i] = 0;
    uint64_t s = seed;
    auto next = [&]() {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    auto wide = [&](int lo, int hi) { // a float with a random sign, a random 24-bit mantissa and an exponent in [lo, hi]
        const float m = 1.0f + (float)(next() >> 41) * (1.0f / 8388608.0f);
        const int e = lo + (int)(next() % (uint64_t)(hi - lo + 1));
        const float v = std::ldexp(m, e);
        return (next() & 1) ? v : -v;
    };
    for (uint64_t i = 0; i < n; ++i) {
        float a = -std::fabs(wide(-30, 4)), b, c;
        const uint64_t k = next() % 8;
        if (k == 0) { b = wide(-149, -80); c = wide(-149, -60) * ((next() & 1) ? 1.0f : 0.0f); }       // tiny b, tiny or zero c
        else if (k == 1) { b = wide(-20, 20); c = 0.0f; }                                               // a ray that starts on the paraboloid
        else if (k == 2) { b = wide(-20, 20); c = b * b / (4.0f * a); }                                 // a (nearly) zero discriminant
        else if (k == 3) { b = 0.0f; c = wide(-149, 20); }
        else { b = wide(-120, 20); c = wide(-120, 20); }
        counts[0] += 1;
        const float disc = b * b - 4.0f * a * c;
        const float sq = std::sqrt(std::fabs(disc));
        const float np = -b + sq, nq = -b - sq;
        const bool admitted = (a < 0.0f) && ((disc < 0.0f) || (std::fmax(std::fabs(b), sq) >= 8.0779356694631609e-28f));
        if (!admitted) {
            if ((np != 0.0f && std::fabs(np) < 7.9e-31f) || (nq != 0.0f && std::fabs(nq) < 7.9e-31f)) counts[4] += 1;
            continue;
        }
        counts[1] += 1;
        const float pick = np < 0.0f ? np : nq;
        const float t_fast = 0.5f * pick / a;
        const bool hit_fast = !(disc < 0.0f) && (pick < 0.0f);
        const float t_ref = rl_paraboloid_roots(a, b, c);
        const bool hit_ref = !(t_ref < 0.0f);
        if (hit_fast != hit_ref) counts[3] += 1;
        else if (hit_ref) {
            counts[2] += 1;
            if (rl_f2u(t_fast) != rl_f2u(t_ref)) counts[3] += 1;
        }
    }
}

// ---- rl_hex_prism_fast against the tree it replaces -----------------------------------------------------------------
// Random and adversarial (prism, ray) pairs over the prisms of `scene`: rays from anywhere, rays that start on a face
// (as after a refraction: origin = surface point + direction * 1e-5), rays aimed at edges and vertices, rays nearly
// parallel to faces.  counts[0] = pairs, [1] = decided hits, [2] = decided misses, [3] = undecided, [4] = decided but
// different from the tree (must be 0), [5] = tree hits.
static inline uint64_t mix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static inline float unit01(uint64_t& s) { return (float)(mix64(s) >> 40) * (1.0f / 16777216.0f); }
static inline float sym(uint64_t& s) { return unit01(s) * 2.0f - 1.0f; }

// One adversarial (prism, ray) pair; false when the draw has to be repeated.
static bool gen_prism_pair(const RlFlatScene& fs, uint64_t& s, uint32_t* prism_out, RlF3* o_out, RlF3* d_out) {
    const uint32_t n_prisms = (uint32_t)(fs.prisms.size() / RL_PRISM_STRIDE);
    const uint32_t prism = (uint32_t)(mix64(s) % n_prisms);
    const RlF4* pr = &fs.prisms[RL_PRISM_STRIDE * prism];
    const RlF4 bound = pr[16];
    if (!(bound.w > 0.0f) || !(bound.w < 1e30f)) return false; // a padding prism
    const float R = std::sqrt(bound.w);
    const RlF3 c = rl_xyz(bound);
    auto rnd_dir = [&]() {
        for (;;) {
            RlF3 v = rl_f3(sym(s), sym(s), sym(s));
            const float m = rl_dot(v, v);
            if (m > 0.01f && m <= 1.0f) return rl_normalise(v);
        }
    };
    // a point on the polytope's surface: a random ray from outside the bound, walked with the tree
    auto surface_point = [&](RlF3* p_out) {
        for (int tries = 0; tries < 64; ++tries) {
            const RlF3 o = rl_add(c, rl_mul(rnd_dir(), R * 2.0f));
            const RlF3 target = rl_add(c, rl_mul(rl_f3(sym(s), sym(s), sym(s)), R * 0.3f));
            const RlF3 d = rl_normalise(rl_sub(target, o));
            const RlCand h = rl_hex_prism(pr, o, d);
            if (h.t > 0.0f) {
                *p_out = rl_add(o, rl_mul(d, h.t));
                return true;
            }
        }
        return false;
    };
    RlF3 o, d;
    const uint32_t kind = (uint32_t)(mix64(s) % 6);
    if (kind == 0) { // anywhere -> anywhere
        o = rl_add(c, rl_mul(rl_f3(sym(s), sym(s), sym(s)), R * 3.0f));
        d = rnd_dir();
    } else if (kind == 1) { // towards the prism
        o = rl_add(c, rl_mul(rnd_dir(), R * (1.0f + 4.0f * unit01(s))));
        d = rl_normalise(rl_sub(rl_add(c, rl_mul(rl_f3(sym(s), sym(s), sym(s)), R * 0.5f)), o));
    } else if (kind == 2 || kind == 3) { // from a face, as after a bounce: origin = surface point + dir * 1e-5 (trace_unit.rs:114)
        RlF3 p;
        if (!surface_point(&p)) return false;
        d = rnd_dir();
        if (kind == 3) d = rl_mul(d, 0.9f + 0.2f * unit01(s)); // glass leaves directions un-normalised
        o = rl_add(p, rl_mul(d, 0.00001f));
    } else if (kind == 4) { // aimed at an edge or a vertex: a surface point pushed onto the nearest other plane(s)
        RlF3 p;
        if (!surface_point(&p)) return false;
        for (int pass = 0; pass < 2; ++pass) {
            int best = -1;
            float bd = 1e30f;
            for (int k = 0; k < 8; ++k) {
                const float dist = std::fabs(rl_dot(rl_sub(p, rl_xyz(pr[2 * k + 1])), rl_xyz(pr[2 * k])));
                if (dist > 1e-4f && dist < bd) { bd = dist; best = k; }
            }
            if (best < 0) break;
            const float sd = rl_dot(rl_sub(p, rl_xyz(pr[2 * best + 1])), rl_xyz(pr[2 * best]));
            p = rl_sub(p, rl_mul(rl_xyz(pr[2 * best]), sd * (1.0f + 1e-6f * sym(s))));
            if (mix64(s) & 1) break; // an edge; otherwise go on to a vertex
        }
        o = rl_add(c, rl_mul(rnd_dir(), R * (1.5f + 3.0f * unit01(s))));
        d = rl_normalise(rl_sub(rl_add(p, rl_mul(rl_f3(sym(s), sym(s), sym(s)), 1e-5f * unit01(s) * unit01(s))), o));
    } else { // nearly parallel to a face
        const RlF3 n = rl_xyz(pr[2 * (mix64(s) % 8)]);
        RlF3 t = rl_normalise(rl_cross(n, rnd_dir()));
        d = rl_normalise(rl_add(t, rl_mul(n, 1e-4f * sym(s) * unit01(s))));
        o = rl_add(c, rl_mul(rl_f3(sym(s), sym(s), sym(s)), R * 1.5f));
    }
    *prism_out = prism;
    *o_out = o;
    *d_out = d;
    return true;
}

extern "C" void mirror_prism_fast_check(void* scene, uint64_t trials, uint64_t seed, uint64_t* counts) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    for (int i = 0; i < 6; ++i) counts[i] = 0;
    if (fs.prisms.empty()) return;
    uint64_t s = seed;
    for (uint64_t it = 0; it < trials; ++it) {
        uint32_t prism;
        RlF3 o, d;
        if (!gen_prism_pair(fs, s, &prism, &o, &d)) continue;
        const RlF4* pr = &fs.prisms[RL_PRISM_STRIDE * prism];
        const RlCand want = rl_hex_prism(pr, o, d);
        RlCand got;
        const int status = rl_hex_prism_fast(pr, o, d, &got);
        counts[0] += 1;
        if (want.t >= 0.0f) counts[5] += 1;
        if (status == RL_PRISM_UNSURE) {
            counts[3] += 1;
        } else if (status == RL_PRISM_HIT) {
            counts[1] += 1;
            if (!(want.t >= 0.0f) || rl_f2u(want.t) != rl_f2u(got.t) || want.k != got.k) counts[4] += 1;
        } else {
            counts[2] += 1;
            if (want.t >= 0.0f) counts[4] += 1;
        }
    }
}

// The same pairs as data, for the device-side comparison (rl_debug_prism_probe): prism numbers, rays {o, d} and the g++
// tree's answer {t bits or 0xffffffff, half-space}.  Returns the number of pairs written (<= cap).
extern "C" uint64_t mirror_prism_pairs(void* scene, uint64_t trials, uint64_t seed, uint32_t* prisms, float* rays6, uint32_t* tree2, uint64_t cap) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    if (fs.prisms.empty()) return 0;
    uint64_t s = seed, n = 0;
    for (uint64_t it = 0; it < trials && n < cap; ++it) {
        uint32_t prism;
        RlF3 o, d;
        if (!gen_prism_pair(fs, s, &prism, &o, &d)) continue;
        const RlCand want = rl_hex_prism(&fs.prisms[RL_PRISM_STRIDE * prism], o, d);
        prisms[n] = prism;
        float* r = rays6 + 6 * n;
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
        tree2[2 * n] = want.t >= 0.0f ? rl_f2u(want.t) : 0xffffffffu;
        tree2[2 * n + 1] = want.k;
        n += 1;
    }
    return n;
}

// The same comparison on the (prism, ray) pairs a render actually produces: every segment of paths [first, first + n)
// against every prism whose bound it passes.
extern "C" void mirror_prism_fast_check_paths(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first,
                                              uint64_t n, uint64_t* counts) {
    const RlSceneView& sv = ((MirrorScene*)scene)->view;
    const float aspect = (float)w / (float)h;
    for (int i = 0; i < 6; ++i) counts[i] = 0;
    for (uint64_t i = 0; i < n; ++i) {
        RlPath p;
        rl_begin_path(sv, aspect, seed, stream, first + i, &p);
        float value = 0.0f;
        for (;;) {
            for (uint32_t k = 0; k < sv.n_prisms; ++k) {
                const RlF4* pr = sv.prisms + RL_PRISM_STRIDE * k;
                if (!rl_bound_pass(pr[16], p.origin, p.direction)) continue;
                const RlCand want = rl_hex_prism(pr, p.origin, p.direction);
                RlCand got;
                const int status = rl_hex_prism_fast(pr, p.origin, p.direction, &got);
                counts[0] += 1;
                if (want.t >= 0.0f) counts[5] += 1;
                if (status == RL_PRISM_UNSURE) counts[3] += 1;
                else if (status == RL_PRISM_HIT) {
                    counts[1] += 1;
                    if (!(want.t >= 0.0f) || rl_f2u(want.t) != rl_f2u(got.t) || want.k != got.k) counts[4] += 1;
                } else {
                    counts[2] += 1;
                    if (want.t >= 0.0f) counts[4] += 1;
                }
            }
            const RlHit hit = rl_scan(sv, p.origin, p.direction);
            uint32_t emitter = 0;
            if (rl_bounce(sv, seed, stream, first + i, &p, hit, &value, &emitter) != RL_PATH_CONTINUES) break;
        }
    }
}

// The prisms' second bound (RlFlatScene::prism_cyl): 8 floats per prism {point.xyz, radius, axis.xyz, 0}; returns the
// number of prisms, 0 when the scene does not use it (fewer than 40 prisms).
extern "C" uint32_t mirror_prism_cylinders(void* scene, float* out8, uint32_t cap) {
    const RlFlatScene& fs = ((MirrorScene*)scene)->flat;
    if (!fs.prism_cylinders) return 0;
    const uint32_t n = (uint32_t)(fs.prism_cyl.size() / 2);
    for (uint32_t i = 0; i < n && i < cap; ++i) {
        const RlF4 c = fs.prism_cyl[2 * i], a = fs.prism_cyl[2 * i + 1];
        float* o = out8 + 8 * i;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w; o[4] = a.x; o[5] = a.y; o[6] = a.z; o[7] = 0.0f;
    }
    return n;
}

// ---- the cull table's work per ray segment, counted on the host ----------------------------------------------------
// Traces paths [first, first + n) and counts, per segment, what the kernel's sphere pass would do with this scene's table:
// counts[0] = segments, [1] = cluster groups, [2] = (group, ray) pairs that pass, [3] = (cluster, ray) pairs that pass,
// [4] = (member, ray) pairs that pass, [5] = clusters, [6] = clusters per group, [7] = members per cluster.  Bounds are
// tested as the kernel does (reach the bound ahead of the origin and not beyond `far`, the nearest plane / circle /
// paraboloid hit), without its rounding slack: a planning figure for tools and tests, not a parity statement.
static bool mirror_reach(RlF4 b, RlF3 o, RlF3 dir, double far_t) {
    const double cox = (double)b.x - o.x, coy = (double)b.y - o.y, coz = (double)b.z - o.z;
    const double d2 = (double)dir.x * dir.x + (double)dir.y * dir.y + (double)dir.z * dir.z;
    const double dd = (dir.x * cox + dir.y * coy + dir.z * coz) / d2; // ray parameter of the point nearest to the centre
    double x = dd < 0.0 ? 0.0 : dd;
    if (x > far_t) x = far_t;
    const double px = cox - x * dir.x, py = coy - x * dir.y, pz = coz - x * dir.z;
    return px * px + py * py + pz * pz <= (double)b.w;
}
// RlFlatScene::small_ordered: may the kernel decide ties among the small primitives by scan order alone?
extern "C" int mirror_small_ordered(void* scene) { return ((MirrorScene*)scene)->flat.small_ordered ? 1 : 0; }
// RlFlatScene::small_axis_z: every paraboloid's, plane's and circle's normal lies along z (the kernel then takes n.v as n.z v.z)
extern "C" int mirror_small_axis_z(void* scene) { return ((MirrorScene*)scene)->flat.small_axis_z ? 1 : 0; }

extern "C" void mirror_cull_counts(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first, uint64_t n,
                                   uint64_t* counts) {
    const MirrorScene* ms = (MirrorScene*)scene;
    const RlSceneView& sv = ms->view;
    const RlFlatScene& fs = ms->flat;
    const float aspect = (float)w / (float)h;
    for (int i = 0; i < 8; ++i) counts[i] = 0;
    counts[1] = fs.n_cluster_groups;
    counts[5] = fs.n_clusters;
    counts[6] = fs.group_gc;
    counts[7] = fs.cluster_k;
    const uint32_t n_level1 = fs.group_gc * fs.n_cluster_groups + RL_GROUP_GP * fs.n_prism_groups;
    for (uint64_t i = 0; i < n; ++i) {
        RlPath p;
        rl_begin_path(sv, aspect, seed, stream, first + i, &p);
        float value = 0.0f;
        for (;;) {
            counts[0] += 1;
            // far bound: planes, circles, paraboloids only (what the kernel knows before its sphere pass)
            RlSceneView small = sv;
            small.n_direct = 0; small.n_clusters = 0; small.n_prisms = 0;
            const RlHit near = rl_scan(small, p.origin, p.direction);
            const double far_t = (double)near.t * 1.0002;
            for (uint32_t g = 0; g < fs.n_cluster_groups; ++g) {
                const RlF4 r = fs.cull_bounds[n_level1 + g];
                const double c2 = (double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z;
                RlF4 gb = r;
                gb.w = (float)(c2 - (double)r.w);
                if (!mirror_reach(gb, p.origin, p.direction, far_t)) continue;
                counts[2] += 1;
                for (uint32_t k = fs.group_gc * g; k < fs.group_gc * (g + 1); ++k) {
                    const uint32_t base = sv.cluster_base + (fs.cluster_k + 1u) * k;
                    if (!mirror_reach(sv.spheres[base], p.origin, p.direction, far_t)) continue;
                    counts[3] += 1;
                    for (uint32_t j = 1; j <= fs.cluster_k; ++j) {
                        RlF4 s = sv.spheres[base + j];
                        s.w *= 1.001f;
                        if (s.w > 0.0f && mirror_reach(s, p.origin, p.direction, far_t)) counts[4] += 1;
                    }
                }
            }
            const RlHit hit = rl_scan(sv, p.origin, p.direction);
            uint32_t emitter = 0;
            if (rl_bounce(sv, seed, stream, first + i, &p, hit, &value, &emitter) != RL_PATH_CONTINUES) break;
        }
    }
}
