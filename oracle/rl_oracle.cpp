// rl_oracle.cpp -- CPU restatement of robigo-luculenta's per-ray hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the parity oracle for the gfx950 kernels.  It follows the reference's Rust sources
// function by function, in the reference's operation order, with the reference's structure
// (trait objects -> virtual classes, Compound<T1,T2> -> recursive template) so that it checks the
// flattened, restructured device code against an independently written statement of the algorithm.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
// (robigo_luculenta_amd/, include/) never does.
//
// PARITY UNPINNED: the reference's own tests assert no numeric value (main.rs:69-74 has no
// assertions) and the Rust crate cannot be built here (no rustc/cargo), so this oracle is pinned by
// hand-derived known-answer tests (tests/test_oracle_kat.py, numpy f32/f64 restatements of the
// formulas) rather than by reference outputs.  Two things are the build's own definition because
// the reference leaves them undefined: the random stream (rand 0.3.11's OS-seeded thread RNG,
// monte_carlo.rs:22 -> Philox slots, csrc/rl_rng.h) and the libm (-> csrc/rl_math.h).
//
// Shared with the product on purpose: rl_math.h (transcendentals), rl_rng.h (draws), rl_cie1931.h
// (table data), include/robigo_luculenta.h (POD layouts).  Everything else is written here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../include/robigo_luculenta.h"
#include "../robigo_luculenta_amd/csrc/rl_cie1931.h"
#include "../robigo_luculenta_amd/csrc/rl_math.h"
#include "../robigo_luculenta_amd/csrc/rl_rng.h"

namespace {

const float PI = RL_PI_F; // std::f32::consts::PI

// constants.rs:17-25
const double GOLDEN_RATIO = 1.6180339887498948482045868343656381177203091798057628;
const double PLANCKS_CONSTANT = 6.62606957e-34;
const double BOLTZMANNS_CONSTANT = 1.3806488e-23;
const double SPEED_OF_LIGHT = 299792458.0;
const double WIENS_CONSTANT = 2.897772126e-3;

// ---- quaternion.rs / vector3.rs --------------------------------------------------------------

struct Quaternion {
    float x, y, z, w;
    // quaternion.rs:34-41
    static Quaternion rotation(float x, float y, float z, float angle) {
        return Quaternion{rl_sinf(angle * 0.5f) * x, rl_sinf(angle * 0.5f) * y, rl_sinf(angle * 0.5f) * z,
                          rl_cosf(angle * 0.5f)};
    }
    // quaternion.rs:43-45
    Quaternion conjugate() const { return Quaternion{-x, -y, -z, w}; }
};

// quaternion.rs:100-111
Quaternion operator*(Quaternion a, Quaternion b) {
    return Quaternion{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                      a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

struct Vector3 {
    float x, y, z;
    static Vector3 zero() { return Vector3{0.0f, 0.0f, 0.0f}; }
};
Vector3 operator+(Vector3 a, Vector3 b) { return Vector3{a.x + b.x, a.y + b.y, a.z + b.z}; }
Vector3 operator-(Vector3 a, Vector3 b) { return Vector3{a.x - b.x, a.y - b.y, a.z - b.z}; }
Vector3 operator-(Vector3 a) { return Vector3{-a.x, -a.y, -a.z}; }
Vector3 operator*(Vector3 a, float f) { return Vector3{a.x * f, a.y * f, a.z * f}; }

// vector3.rs:27-37
Vector3 cross(Vector3 a, Vector3 b) {
    return Vector3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
float dot(Vector3 a, Vector3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
float magnitude_squared(Vector3 v) { return dot(v, v); }
float magnitude(Vector3 v) { return sqrtf(magnitude_squared(v)); }
// vector3.rs:56-67
Vector3 normalise(Vector3 v) {
    float m = magnitude(v);
    if (m == 0.0f) return v;
    return Vector3{v.x / m, v.y / m, v.z / m};
}
// vector3.rs:69-83
Vector3 rotate_towards(Vector3 self, Vector3 normal) {
    float d = normal.z;
    if (d > 0.9999f) return self;
    if (d < -0.9999f) return Vector3{self.x, self.y, -self.z};
    Vector3 up{0.0f, 0.0f, 1.0f};
    Vector3 a1 = normalise(cross(up, normal));
    Vector3 a2 = normalise(cross(a1, normal));
    return a1 * self.x + a2 * self.y + normal * self.z;
}
// vector3.rs:85-89
Vector3 rotate(Vector3 self, Quaternion q) {
    Quaternion p{self.x, self.y, self.z, 0.0f};
    Quaternion r = q * p * q.conjugate();
    return Vector3{r.x, r.y, r.z};
}
// vector3.rs:91-93
Vector3 reflect(Vector3 self, Vector3 normal) { return self - normal * 2.0f * dot(normal, self); }

// ---- ray.rs / intersection.rs ------------------------------------------------------------------

struct Ray {
    Vector3 origin, direction;
    float wavelength, probability;
};
struct Intersection {
    Vector3 position, normal, tangent;
    float distance;
};
struct OptIsect {
    bool some;
    Intersection i;
};
const OptIsect NONE{false, {}};
OptIsect some(const Intersection& i) { return OptIsect{true, i}; }

// ---- the random draws (monte_carlo.rs) over rl_rng.h's slots ------------------------------------

struct PathRng {
    uint64_t seed;
    uint32_t stream;
    uint64_t path;
    uint32_t block;
    RlRngBlock cur;
    void begin(uint64_t s, uint32_t st, uint64_t p) {
        seed = s;
        stream = st;
        path = p;
        block = 0;
        cur = rl_rng_block(seed, stream, path, 0);
    }
    void next_block() {
        block += 1;
        cur = rl_rng_block(seed, stream, path, block);
    }
    float get_unit(int slot) const { return rl_get_unit(cur.w[slot]); }           // monte_carlo.rs:25-28
    float get_bi_unit(int slot) const { return rl_get_bi_unit(cur.w[slot]); }     // monte_carlo.rs:31-33
    float get_longitude(int slot) const { return rl_get_longitude(cur.w[slot]); } // monte_carlo.rs:36-38
    float get_wavelength(int slot) const { return rl_get_wavelength(cur.w[slot]); } // monte_carlo.rs:41-43
    // monte_carlo.rs:47-58
    Vector3 get_hemisphere_vector() const {
        float phi = get_longitude(0);
        float rq = get_unit(1);
        float r = sqrtf(rq);
        return Vector3{rl_cosf(phi) * r, rl_sinf(phi) * r, sqrtf(1.0f - rq)};
    }
};

// ---- geometry.rs --------------------------------------------------------------------------------

struct Surface {
    virtual ~Surface() {}
    virtual OptIsect intersect(const Ray& ray) const = 0;
};

// geometry.rs:55-71
struct PlaneHit {
    bool some;
    Vector3 pos;
    float t, d;
};
PlaneHit intersect_plane(const Vector3& normal, const Vector3& offset, const Ray& ray) {
    Vector3 origin = ray.origin - offset;
    float d = dot(normal, ray.direction);
    if (d == 0.0f) return PlaneHit{false, {}, 0, 0};
    float t = -dot(normal, origin) / d;
    if (t <= 0.0f) return PlaneHit{false, {}, 0, 0};
    return PlaneHit{true, ray.origin + ray.direction * t, t, d};
}

// geometry.rs:35-87
struct Plane : Surface {
    Vector3 normal, offset;
    Plane(Vector3 n, Vector3 o) : normal(n), offset(o) {}
    OptIsect intersect(const Ray& ray) const override {
        PlaneHit h = intersect_plane(normal, offset, ray);
        if (!h.some) return NONE;
        return some(Intersection{h.pos, h.d < 0.0f ? normal : -normal, Vector3::zero(), h.t});
    }
};

// geometry.rs:90-128
struct SpacePartitioning {
    Vector3 normal, offset;
    SpacePartitioning(Vector3 n, Vector3 o) : normal(n), offset(o) {}
    OptIsect intersect(const Ray& ray) const {
        PlaneHit h = intersect_plane(normal, offset, ray);
        if (!h.some) return NONE;
        return some(Intersection{h.pos, normal, Vector3::zero(), h.t});
    }
    bool lies_inside(Vector3 p) const { return dot(p - offset, normal) < 0.0f; }
};

// geometry.rs:130-184
struct Circle : Surface {
    Vector3 normal, position;
    float radius_squared;
    Circle(Vector3 n, Vector3 p, float r) : normal(n), position(p), radius_squared(r * r) {}
    OptIsect intersect(const Ray& ray) const override {
        PlaneHit h = intersect_plane(normal, position, ray);
        if (!h.some) return NONE;
        if (!(magnitude_squared(h.pos - position) <= radius_squared)) return NONE;
        return some(Intersection{h.pos, h.d < 0.0f ? normal : -normal, Vector3::zero(), h.t});
    }
};

// geometry.rs:186-267
struct Sphere : Surface {
    Vector3 position;
    float radius_squared;
    Sphere(Vector3 p, float r) : position(p), radius_squared(r * r) {}
    bool get_intersections(const Ray& ray, float* t1, float* t2) const {
        float a = 1.0f;
        Vector3 centre_offset = position - ray.origin;
        float b = 2.0f * dot(ray.direction, centre_offset);
        float c = magnitude_squared(centre_offset) - radius_squared;
        float discriminant = b * b - 4.0f * a * c;
        if (discriminant < 0.0f) return false;
        float d = sqrtf(discriminant);
        *t1 = -0.5f * (-b + d) / a;
        *t2 = -0.5f * (-b - d) / a;
        return true;
    }
    OptIsect intersect(const Ray& ray) const override {
        float t1, t2;
        if (!get_intersections(ray, &t1, &t2)) return NONE;
        float t;
        if (t1 > 0.0f && t1 < t2) t = t1;
        else if (t2 > 0.0f && t2 < t1) t = t2;
        else return NONE;
        Vector3 position_ = ray.origin + ray.direction * t;
        Vector3 normal = normalise(position_ - position);
        Vector3 up{0.0f, 1.0f, 0.0f};
        Vector3 tangent = normalise(cross(up, normal));
        return some(Intersection{position_, normal, tangent, t});
    }
};

// geometry.rs:269-358
struct Paraboloid : Surface {
    Vector3 offset, normal, focal_point;
    Paraboloid(Vector3 n, Vector3 o, float focal_distance)
        : offset(o - n * focal_distance), normal(n), focal_point(n * (focal_distance * 2.0f)) {}
    OptIsect intersect(const Ray& ray) const override {
        Vector3 origin = ray.origin - offset;
        Vector3 focal_offset = origin - focal_point;
        float n_dot_d = dot(normal, ray.direction);
        float n_dot_o = dot(normal, origin);
        float d_dot_f = dot(ray.direction, focal_offset);
        float a = n_dot_d * n_dot_d - 1.0f;
        float b = 2.0f * n_dot_d * n_dot_o - 2.0f * d_dot_f;
        float c = n_dot_o * n_dot_o - magnitude_squared(focal_offset);
        float t;
        if (a == 0.0f) {
            float t1 = -c / b;
            if (t1 < 0.0f) return NONE;
            t = t1;
        } else {
            float d = b * b - 4.0f * a * c;
            if (d < 0.0f) return NONE;
            float sqrt_d = sqrtf(d);
            float p = 0.5f * (-b + sqrt_d) / a;
            float q = 0.5f * (-b - sqrt_d) / a;
            if (p > 0.0f && (p < q || q < 0.0f)) t = p;
            else if (q > 0.0f) t = q;
            else return NONE;
        }
        Vector3 pos = ray.origin + ray.direction * t;
        Vector3 local_pos = pos - offset;
        Vector3 plane_pr = local_pos - normal * dot(local_pos, normal);
        Vector3 n = normalise(focal_point - plane_pr);
        return some(Intersection{pos, n, Vector3::zero(), t});
    }
};

// geometry.rs:361-407
template <class T1, class T2>
struct Compound {
    T1 surface1;
    T2 surface2;
    Compound(T1 s1, T2 s2) : surface1(s1), surface2(s2) {}
    OptIsect intersect(const Ray& ray) const {
        OptIsect i1 = surface1.intersect(ray);
        OptIsect i2 = surface2.intersect(ray);
        if (i1.some && !surface2.lies_inside(i1.i.position)) i1 = NONE;
        if (i2.some && !surface1.lies_inside(i2.i.position)) i2 = NONE;
        if (i1.some && i2.some) {
            if (i1.i.distance < i2.i.distance) return i1;
            return i2;
        }
        return i1.some ? i1 : i2;
    }
    bool lies_inside(Vector3 p) const { return surface1.lies_inside(p) && surface2.lies_inside(p); }
};

// geometry.rs:409-416
typedef Compound<Compound<SpacePartitioning, SpacePartitioning>, SpacePartitioning> InfinitePrism;
typedef Compound<SpacePartitioning, SpacePartitioning> ThickPlane;
typedef Compound<InfinitePrism, ThickPlane> Prism;
typedef Compound<InfinitePrism, Prism> HexagonalPrism;

// geometry.rs:421-450
InfinitePrism new_infinite_prism(Vector3 axis, Vector3 offset, float edge_length, float angle) {
    float radius = sqrtf(3.0f) / 6.0f * edge_length;
    float a1 = angle;
    float a2 = angle + PI * 2.0f / 3.0f;
    float a3 = angle + PI * 4.0f / 3.0f;
    Vector3 p1{rl_cosf_d(a1), rl_sinf_d(a1), 0.0f};
    Vector3 p2{rl_cosf_d(a2), rl_sinf_d(a2), 0.0f};
    Vector3 p3{rl_cosf_d(a3), rl_sinf_d(a3), 0.0f};
    p1 = rotate_towards(p1, axis);
    p2 = rotate_towards(p2, axis);
    p3 = rotate_towards(p3, axis);
    SpacePartitioning sp1(p1, p1 * radius + offset);
    SpacePartitioning sp2(p2, p2 * radius + offset);
    SpacePartitioning sp3(p3, p3 * radius + offset);
    return InfinitePrism(Compound<SpacePartitioning, SpacePartitioning>(sp1, sp2), sp3);
}
// geometry.rs:455-468
ThickPlane new_thick_plane(Vector3 normal, Vector3 offset, float thickness) {
    SpacePartitioning sp1(-normal, offset);
    SpacePartitioning sp2(normal, offset + normal * thickness);
    return ThickPlane(sp1, sp2);
}
// geometry.rs:474-486
Prism new_prism(Vector3 axis, Vector3 offset, float edge_length, float angle, float height) {
    return Prism(new_infinite_prism(axis, offset, edge_length, angle), new_thick_plane(axis, offset, height));
}
// geometry.rs:493-515
HexagonalPrism new_hexagonal_prism(Vector3 axis, Vector3 offset, float edge_length, float bevel_size, float angle,
                                   float height) {
    InfinitePrism iprism = new_infinite_prism(axis, offset, edge_length * 2.0f - bevel_size * 3.0f, angle + PI);
    Prism prism = new_prism(axis, offset, edge_length, angle, height);
    return HexagonalPrism(iprism, prism);
}

struct HexPrismSurface : Surface {
    HexagonalPrism prism;
    explicit HexPrismSurface(HexagonalPrism p) : prism(p) {}
    OptIsect intersect(const Ray& ray) const override { return prism.intersect(ray); }
};

// ---- material.rs --------------------------------------------------------------------------------

struct Material {
    virtual ~Material() {}
    virtual Ray get_new_ray(const Ray& incoming_ray, const Intersection& intersection, const PathRng& rng) const = 0;
};
struct EmissiveMaterial {
    virtual ~EmissiveMaterial() {}
    virtual float get_intensity(float wavelength) const = 0;
};

// material.rs:38-58
Ray get_diffuse_ray(const Ray& incoming_ray, const Intersection& intersection, const PathRng& rng) {
    Vector3 hemi_vec = rng.get_hemisphere_vector();
    Vector3 normal = dot(incoming_ray.direction, intersection.normal) < 0.0f ? intersection.normal : -intersection.normal;
    Vector3 direction = rotate_towards(hemi_vec, normal);
    return Ray{intersection.position, direction, incoming_ray.wavelength, 1.0f};
}

// material.rs:61-74
double boltzmann(double wavelength, double temperature) {
    double h = PLANCKS_CONSTANT, k = BOLTZMANNS_CONSTANT, c = SPEED_OF_LIGHT;
    double f = c / (wavelength * 1.0e-9);
    return (2.0 * h * f * f * f) / (c * c * (rl_exp_d(h * f / (k * temperature)) - 1.0));
}

// material.rs:77-105
struct BlackBodyMaterial : EmissiveMaterial {
    float temperature, normalisation_factor;
    BlackBodyMaterial(float kelvins, float intensity)
        : temperature(kelvins),
          normalisation_factor(intensity / (float)boltzmann((WIENS_CONSTANT / (double)kelvins) * 1.0e9, (double)kelvins)) {}
    float get_intensity(float wavelength) const override {
        return (float)boltzmann((double)wavelength, (double)temperature) * normalisation_factor;
    }
};

// material.rs:109-130
struct DiffuseGreyMaterial : Material {
    float reflectance;
    explicit DiffuseGreyMaterial(float r) : reflectance(r) {}
    Ray get_new_ray(const Ray& in, const Intersection& is, const PathRng& rng) const override {
        Ray ray = get_diffuse_ray(in, is, rng);
        ray.probability = reflectance;
        return ray;
    }
};

// material.rs:134-168
struct DiffuseColouredMaterial : Material {
    float reflectance, wavelength, deviation;
    DiffuseColouredMaterial(float r, float w, float d) : reflectance(r), wavelength(w), deviation(d) {}
    Ray get_new_ray(const Ray& in, const Intersection& is, const PathRng& rng) const override {
        float p = (wavelength - in.wavelength) / deviation;
        float q = rl_expf(-0.5f * p * p);
        Ray ray = get_diffuse_ray(in, is, rng);
        ray.probability = reflectance * q;
        return ray;
    }
};

// material.rs:171-196
struct GlossyMirrorMaterial : Material {
    float glossiness;
    explicit GlossyMirrorMaterial(float g) : glossiness(g) {}
    Ray get_new_ray(const Ray& in, const Intersection& is, const PathRng& rng) const override {
        Ray ray = get_diffuse_ray(in, is, rng);
        Vector3 reflection = reflect(in.direction, is.normal);
        ray.direction = normalise(ray.direction * glossiness + reflection * (1.0f - glossiness));
        return ray;
    }
};

// material.rs:199-261
struct Sf10GlassMaterial : Material {
    static float get_index_of_refraction(float wavelength) {
        double w2 = (double)(wavelength * wavelength * 1.0e-6f);
        return (float)sqrt(1.0 + 1.737596950 * w2 / (w2 - 0.0131887070) + 0.313747346 * w2 / (w2 - 0.0623068142) +
                           1.898781010 * w2 / (w2 - 155.23629000));
    }
    Ray get_new_ray(const Ray& in, const Intersection& is, const PathRng&) const override {
        float cos_i = -dot(in.direction, is.normal);
        float ior = get_index_of_refraction(in.wavelength);
        Vector3 normal = is.normal;
        if (cos_i > 0.0f) {
            ior = 1.0f / ior;
        } else {
            normal = -normal;
            cos_i = -cos_i;
        }
        float sin_t_sqr = ior * ior * (1.0f - cos_i * cos_i);
        Vector3 dir;
        if (sin_t_sqr > 1.0f) {
            dir = reflect(in.direction, normal);
        } else {
            float cos_t = sqrtf(1.0f - sin_t_sqr);
            dir = in.direction * ior + normal * (ior * cos_i - cos_t);
        }
        return Ray{is.position, dir, in.wavelength, 1.0f};
    }
};

// material.rs:265-306
struct SoapBubbleMaterial : Material {
    static float clamp(float x) {
        if (x < -0.999f) return -0.999f;
        if (x > 0.999f) return 0.999f;
        return x;
    }
    Ray get_new_ray(const Ray& in, const Intersection& is, const PathRng& rng) const override {
        float cos_alpha = dot(in.direction, is.normal);
        Vector3 direction;
        if (rng.get_unit(0) - 0.3f > fabsf(cos_alpha)) direction = reflect(in.direction, is.normal);
        else direction = in.direction;
        float phase_shift = (in.wavelength - 380.0f) / 200.0f * PI;
        float cos_phi = clamp(dot(direction, is.normal));
        float cos_theta = clamp(dot(direction, is.tangent));
        float p = rl_cosf(phase_shift - rl_acosf(cos_phi) * 3.0f - rl_acosf(cos_theta) * 2.0f + PI * 0.5f);
        return Ray{is.position, direction, in.wavelength, p * 0.1f + 0.9f};
    }
};

// ---- camera.rs ----------------------------------------------------------------------------------

struct Camera {
    Vector3 position;
    float field_of_view, focal_distance, depth_of_field, chromatic_abberation;
    Quaternion orientation;

    // camera.rs:47-90
    Ray get_screen_ray(float x, float y, float chromatic_abberation_factor, float dof_angle, float dof_radius) const {
        float screen_distance = 1.0f / rl_tanf(field_of_view * 0.5f);
        float xs = x * chromatic_abberation_factor;
        float ys = y * chromatic_abberation_factor;
        Vector3 direction = normalise(Vector3{xs, screen_distance, -ys});
        Vector3 focus_point = direction * (focal_distance / direction.y);
        Vector3 lens_point{rl_cosf(dof_angle) * dof_radius, 0.0f, rl_sinf(dof_angle) * dof_radius};
        return Ray{position + rotate(lens_point, orientation), normalise(rotate(focus_point - lens_point, orientation)),
                   0.0f, 1.0f};
    }
    // camera.rs:94-108
    Ray get_ray(float x, float y, float wavelength, const PathRng& rng) const {
        float dof_angle = rng.get_longitude(0);
        float dof_radius = rng.get_unit(1) / depth_of_field;
        float d = (wavelength - 580.0f) / 200.0f;
        float chromatic_zoom = 1.0f + d * chromatic_abberation;
        Ray r = get_screen_ray(x, y, chromatic_zoom, dof_angle, dof_radius);
        r.wavelength = wavelength;
        return r;
    }
};

// app.rs:327-357 generalised to RlCameraDesc (the demo scene's numbers reproduce it exactly).
Camera make_camera(const RlCameraDesc& cd, float t) {
    float phi = PI * (cd.phi0 + cd.phi1 * t);
    float alpha = PI * (cd.alpha0 + cd.alpha1 * t);
    float distance = cd.dist0 + cd.dist1 * t;
    Vector3 position{rl_cosf(alpha) * rl_sinf(phi) * distance, rl_cosf(alpha) * rl_cosf(phi) * distance,
                     rl_sinf(alpha) * distance};
    Quaternion orientation = Quaternion::rotation(0.0f, 0.0f, -1.0f, phi + PI) * Quaternion::rotation(1.0f, 0.0f, 0.0f, -alpha);
    return Camera{position, PI * cd.fov_over_pi, distance * cd.focal_factor, cd.depth_of_field, cd.chromatic_abberation,
                  orientation};
}

// ---- object.rs / scene.rs -----------------------------------------------------------------------

struct Object {
    std::shared_ptr<Surface> surface;
    std::shared_ptr<Material> reflective;      // MaterialBox::Reflective
    std::shared_ptr<EmissiveMaterial> emissive; // MaterialBox::Emissive
};

struct Scene {
    std::vector<Object> objects;
    RlCameraDesc camera;
    // scene.rs:39-60
    const Object* intersect(const Ray& ray, Intersection* out) const {
        const Object* result = nullptr;
        float distance = 1.0e12f;
        for (const Object& obj : objects) {
            OptIsect isect = obj.surface->intersect(ray);
            if (isect.some && isect.i.distance < distance) {
                result = &obj;
                *out = isect.i;
                distance = isect.i.distance;
            }
        }
        return result;
    }
};

Vector3 v3(const RlVector3& v) { return Vector3{v.x, v.y, v.z}; }

// Builds the object graph from the POD description (the Rust constructors named in
// include/robigo_luculenta.h).
Scene* scene_from_desc(const RlObjectDesc* objs, uint32_t n, const RlCameraDesc* cam) {
    Scene* s = new Scene();
    s->camera = *cam;
    for (uint32_t i = 0; i < n; ++i) {
        const RlObjectDesc& o = objs[i];
        Object obj;
        switch (o.surface_kind) {
        case RL_SURFACE_SPHERE: obj.surface = std::make_shared<Sphere>(v3(o.v0), o.f0); break;
        case RL_SURFACE_PLANE: obj.surface = std::make_shared<Plane>(v3(o.v0), v3(o.v1)); break;
        case RL_SURFACE_CIRCLE: obj.surface = std::make_shared<Circle>(v3(o.v0), v3(o.v1), o.f0); break;
        case RL_SURFACE_PARABOLOID: obj.surface = std::make_shared<Paraboloid>(v3(o.v0), v3(o.v1), o.f0); break;
        case RL_SURFACE_HEX_PRISM:
            obj.surface = std::make_shared<HexPrismSurface>(new_hexagonal_prism(v3(o.v0), v3(o.v1), o.f0, o.f1, o.f2, o.f3));
            break;
        default: delete s; return nullptr;
        }
        switch (o.material_kind) {
        case RL_MATERIAL_BLACK_BODY: obj.emissive = std::make_shared<BlackBodyMaterial>(o.m0, o.m1); break;
        case RL_MATERIAL_DIFFUSE_GREY: obj.reflective = std::make_shared<DiffuseGreyMaterial>(o.m0); break;
        case RL_MATERIAL_DIFFUSE_COLOURED: obj.reflective = std::make_shared<DiffuseColouredMaterial>(o.m0, o.m1, o.m2); break;
        case RL_MATERIAL_GLOSSY_MIRROR: obj.reflective = std::make_shared<GlossyMirrorMaterial>(o.m0); break;
        case RL_MATERIAL_SF10_GLASS: obj.reflective = std::make_shared<Sf10GlassMaterial>(); break;
        case RL_MATERIAL_SOAP_BUBBLE: obj.reflective = std::make_shared<SoapBubbleMaterial>(); break;
        default: delete s; return nullptr;
        }
        s->objects.push_back(obj);
    }
    return s;
}

// ---- app.rs:166-325: the demo scene, as a description -------------------------------------------
// Independent restatement; tests compare it field by field with the product's
// rl_scene_builtin_desc(RL_SCENE_DEMO).

RlVector3 rv(Vector3 v) { return RlVector3{v.x, v.y, v.z}; }

RlObjectDesc od(uint32_t sk, Vector3 v0, Vector3 v1, float f0, float f1, float f2, float f3, uint32_t mk, float m0,
                float m1, float m2) {
    RlObjectDesc o;
    memset(&o, 0, sizeof o);
    o.surface_kind = sk;
    o.material_kind = mk;
    o.v0 = rv(v0);
    o.v1 = rv(v1);
    o.f0 = f0; o.f1 = f1; o.f2 = f2; o.f3 = f3;
    o.m0 = m0; o.m1 = m1; o.m2 = m2;
    return o;
}

std::vector<RlObjectDesc> demo_scene_desc(int seeds_param) {
    std::vector<RlObjectDesc> objects;
    const Vector3 Z = Vector3::zero();
    float sun_radius = 5.0f;                                                        // app.rs:172
    Vector3 sun_position = Vector3::zero();
    objects.push_back(od(RL_SURFACE_SPHERE, sun_position, Z, sun_radius, 0, 0, 0, RL_MATERIAL_BLACK_BODY, 6504.0f, 1.0f, 0));
    Vector3 floor_normal{0.0f, 0.0f, -1.0f};                                       // app.rs:180
    Vector3 floor_position{0.0f, 0.0f, -sun_radius};
    float sun_r2 = sun_radius * sun_radius; // powi(2)
    Paraboloid floor_paraboloid(floor_normal, floor_position, sun_r2);
    objects.push_back(od(RL_SURFACE_PARABOLOID, floor_normal, floor_position, sun_r2, 0, 0, 0, RL_MATERIAL_DIFFUSE_GREY, 0.8f, 0, 0));
    objects.push_back(od(RL_SURFACE_PARABOLOID, Vector3{0, 0, 1.0f}, Vector3{1.0f, 0.0f, -sun_r2}, sun_r2, 0, 0, 0,
                         RL_MATERIAL_DIFFUSE_COLOURED, 0.9f, 550.0f, 40.0f));        // app.rs:189-196
    objects.push_back(od(RL_SURFACE_PARABOLOID, Vector3{0, 0, 1.0f}, Vector3{-1.0f, 0.0f, -sun_r2}, sun_r2, 0, 0, 0,
                         RL_MATERIAL_DIFFUSE_COLOURED, 0.9f, 660.0f, 60.0f));        // app.rs:199-206
    float sky_height = 30.0f;                                                       // app.rs:209
    float sky1_radius = 5.0f;
    objects.push_back(od(RL_SURFACE_CIRCLE, floor_normal, Vector3{-sun_radius, 0.0f, sky_height}, sky1_radius, 0, 0, 0,
                         RL_MATERIAL_BLACK_BODY, 7600.0f, 0.6f, 0));
    float sky2_radius = 15.0f;                                                      // app.rs:217
    objects.push_back(od(RL_SURFACE_CIRCLE, floor_normal,
                         Vector3{-sun_radius * 0.5f, sun_radius * 2.0f + sky2_radius, sky_height}, sky2_radius, 0, 0, 0,
                         RL_MATERIAL_BLACK_BODY, 5000.0f, 0.6f, 0));
    objects.push_back(od(RL_SURFACE_PLANE, floor_normal, Vector3{0.0f, 0.0f, sky_height * 2.0f}, 0, 0, 0, 0,
                         RL_MATERIAL_DIFFUSE_COLOURED, 0.5f, 470.0f, 25.0f));        // app.rs:227-231

    float gamma = PI * 2.0f * (1.0f - 1.0f / (float)GOLDEN_RATIO);                  // app.rs:234
    float seed_size = 0.8f, seed_scale = 1.5f;
    float fs = sun_radius / seed_scale + 1.0f;
    long first_seed = (long)(fs * fs + 0.5f);                                       // app.rs:237
    long seeds = seeds_param > 0 ? seeds_param : 100;                              // app.rs:238
    for (long i = first_seed; i < first_seed + seeds; ++i) {                        // app.rs:239-253
        float phi = (float)i * gamma;
        float r = sqrtf((float)i) * seed_scale;
        Vector3 position = Vector3{rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * -0.5f} + sun_position;
        objects.push_back(od(RL_SURFACE_SPHERE, position, Z, seed_size, 0, 0, 0, RL_MATERIAL_DIFFUSE_COLOURED, 0.9f,
                             (float)(i - first_seed) / (float)seeds * 130.0f + 600.0f, 60.0f));
    }
    for (long i = first_seed; i < first_seed + seeds; ++i) {                        // app.rs:256-268
        float phi = ((float)i + 0.5f) * gamma;
        float r = sqrtf((float)i + 0.5f) * seed_scale;
        Vector3 position = Vector3{rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * -0.25f} + sun_position;
        objects.push_back(od(RL_SURFACE_SPHERE, position, Z, seed_size * 0.5f, 0, 0, 0, RL_MATERIAL_GLOSSY_MIRROR, 0.1f, 0, 0));
    }
    for (long i = first_seed / 2; i < first_seed + seeds; ++i) {                    // app.rs:271-284
        float phi = (float)(-i) * gamma;
        float r = sqrtf((float)i) * seed_scale * 1.5f;
        Vector3 position = Vector3{rl_cosf_d(phi) * r, rl_sinf_d(phi) * r, (r - sun_radius) * 1.5f + sun_radius * 2.0f} + sun_position;
        objects.push_back(od(RL_SURFACE_SPHERE, position, Z, seed_size * (0.5f + sqrtf((float)i) * 0.2f), 0, 0, 0,
                             RL_MATERIAL_SOAP_BUBBLE, 0, 0, 0));
    }
    long prisms = 11;                                                               // app.rs:287-325
    float prism_angle = PI * 2.0f / (float)prisms;
    float prism_radius = 17.0f, prism_height = 8.0f;
    for (long i = 0; i < prisms; ++i) {
        const float variants[2][4] = {{0.0f, 1.0f, 0.0f, 1.0f}, {0.5f * prism_angle, 1.2f, PI * 0.5f, 1.5f}};
        for (int v = 0; v < 2; ++v) {
            float ofs = variants[v][0], radius = variants[v][1], phi_ofs = variants[v][2], h = variants[v][3];
            float phi = (float)i * prism_angle + ofs;
            Vector3 position{rl_cosf_d(phi) * prism_radius * radius, rl_sinf_d(phi) * prism_radius * radius, 0.0f};
            Vector3 normal{0.0f, 0.0f, -1.0f};
            Ray ray{position, normal, 0.0f, 1.0f};
            OptIsect is = floor_paraboloid.intersect(ray);
            if (is.some) {
                normal = -is.i.normal;
                position = is.i.position + normal * 2.0f * h;
            }
            objects.push_back(od(RL_SURFACE_HEX_PRISM, normal, position, 3.0f, 1.0f, phi + phi_ofs, prism_height * h,
                                 RL_MATERIAL_SF10_GLASS, 0, 0, 0));
        }
    }
    return objects;
}

// app.rs:327-357's constants.
RlCameraDesc demo_camera_desc() {
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

// ---- trace_unit.rs ------------------------------------------------------------------------------

// trace_unit.rs:81-132.  *segments counts Scene::intersect calls.
float render_ray(const Scene& scene, Ray ray, PathRng& rng, uint64_t* segments) {
    float continue_chance = 1.0f;
    float intensity = 1.0f;
    for (;;) {
        Intersection intersection;
        *segments += 1;
        const Object* object = scene.intersect(ray, &intersection);
        if (!object) return 0.0f;
        rng.next_block(); // bounce b uses block 2 + b
        if (object->emissive) return intensity * object->emissive->get_intensity(ray.wavelength);
        ray = object->reflective->get_new_ray(ray, intersection, rng);
        intensity = intensity * ray.probability;
        ray.origin = ray.origin + ray.direction * 0.00001f;
        continue_chance = continue_chance * 0.96f;
        if (rng.get_unit(2) * 0.85f > continue_chance * (1.0f - rl_expf(intensity * -20.0f))) break;
    }
    return 0.0f;
}

// trace_unit.rs:136-148
float render_camera_ray(const Scene& scene, float x, float y, float wavelength, PathRng& rng, uint64_t* segments) {
    float t = rng.get_unit(3);
    Camera camera = make_camera(scene.camera, t);
    rng.next_block(); // block 1: depth of field
    Ray ray = camera.get_ray(x, y, wavelength, rng);
    return render_ray(scene, ray, rng, segments);
}

// trace_unit.rs:151-168 for photons [first, first + n).
void render(const Scene& scene, uint32_t width, uint32_t height, uint64_t seed, uint32_t stream, uint64_t first,
            uint64_t n, RlMappedPhoton* photons, uint64_t* segments) {
    float aspect_ratio = (float)width / (float)height;
    uint64_t segs = 0;
    for (uint64_t i = 0; i < n; ++i) {
        PathRng rng;
        rng.begin(seed, stream, first + i);
        float wavelength = rng.get_wavelength(0);
        float x = rng.get_bi_unit(1);
        float y = rng.get_bi_unit(2) / aspect_ratio;
        photons[i].wavelength = wavelength;
        photons[i].x = x;
        photons[i].y = y;
        photons[i].probability = render_camera_ray(scene, x, y, wavelength, rng, &segs);
    }
    if (segments) *segments = segs;
}

// ---- cie1931.rs:20-48 ---------------------------------------------------------------------------

Vector3 get_tristimulus(float wavelength) {
    const float* T = RL_CIE1931_XYZ0;
    float indexf = (wavelength - 380.0f) / 5.0f;
    long index = (long)floorf(indexf);
    float remainder = indexf - (float)index;
    if (index < -1 || index > 80) return Vector3::zero();
    if (index == -1) return Vector3{T[0] * remainder, T[1] * remainder, T[2] * remainder};
    if (index == 80) return Vector3{T[320] * (1.0f - remainder), T[321] * (1.0f - remainder), T[322] * (1.0f - remainder)};
    long i = index;
    return Vector3{T[4 * i] * (1.0f - remainder) + T[4 * i + 4] * remainder,
                   T[4 * i + 1] * (1.0f - remainder) + T[4 * i + 5] * remainder,
                   T[4 * i + 2] * (1.0f - remainder) + T[4 * i + 6] * remainder};
}

// ---- plot_unit.rs:56-95 -------------------------------------------------------------------------

void plot_pixel(RlVector3* buffer, uint32_t image_width, uint32_t image_height, float aspect_ratio, float x, float y,
                Vector3 cie) {
    long w = (long)image_width, h = (long)image_height;
    float px = (x * 0.5f + 0.5f) * ((float)w - 1.0f);
    float py = (y * aspect_ratio * 0.5f + 0.5f) * ((float)h - 1.0f);
    long px1 = std::max(0L, std::min(w - 1, (long)floorf(px)));
    long px2 = std::max(0L, std::min(w - 1, (long)ceilf(px)));
    long py1 = std::max(0L, std::min(h - 1, (long)floorf(py)));
    long py2 = std::max(0L, std::min(h - 1, (long)ceilf(py)));
    float cx = px - (float)px1;
    float cy = py - (float)py1;
    float c11 = (1.0f - cx) * (1.0f - cy);
    float c12 = (1.0f - cx) * cy;
    float c21 = cx * (1.0f - cy);
    float c22 = cx * cy;
    auto add = [&](long idx, float c) {
        Vector3 b{buffer[idx].x, buffer[idx].y, buffer[idx].z};
        b = b + cie * c;
        buffer[idx] = RlVector3{b.x, b.y, b.z};
    };
    add(py1 * w + px1, c11);
    add(py1 * w + px2, c21);
    add(py2 * w + px1, c12);
    add(py2 * w + px2, c22);
}

void plot(RlVector3* buffer, uint32_t w, uint32_t h, const RlMappedPhoton* photons, uint64_t n) {
    float aspect_ratio = (float)w / (float)h;
    for (uint64_t i = 0; i < n; ++i) {
        Vector3 cie = get_tristimulus(photons[i].wavelength);
        plot_pixel(buffer, w, h, aspect_ratio, photons[i].x, photons[i].y, cie * photons[i].probability);
    }
}

// ---- gather_unit.rs:49-64 -----------------------------------------------------------------------

void accumulate(RlVector3* acc, RlVector3* comp, const RlVector3* px, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        Vector3 a{acc[i].x, acc[i].y, acc[i].z}, c{comp[i].x, comp[i].y, comp[i].z}, p{px[i].x, px[i].y, px[i].z};
        Vector3 extra = p - c;
        Vector3 sum = a + extra;
        c = (sum - a) - extra;
        a = sum;
        acc[i] = RlVector3{a.x, a.y, a.z};
        comp[i] = RlVector3{c.x, c.y, c.z};
    }
}

// ---- srgb.rs:20-41, tonemap_unit.rs:55-100 ------------------------------------------------------

float gamma_correct(float f) {
    if (f <= 0.0031308f) return 12.92f * f;
    return 1.055f * rl_powf_full(f, 1.0f / 2.4f) - 0.055f;
}
Vector3 srgb_transform(Vector3 cie) {
    float r = 3.2406f * cie.x - 1.5372f * cie.y - 0.4986f * cie.z;
    float g = -0.9689f * cie.x + 1.8758f * cie.y + 0.0415f * cie.z;
    float b = 0.0557f * cie.x - 0.2040f * cie.y + 1.0570f * cie.z;
    return Vector3{gamma_correct(r), gamma_correct(g), gamma_correct(b)};
}
float clamp01(float x) {
    if (x < 0.0f) return 0.0f;
    if (1.0f < x) return 1.0f;
    return x;
}
float find_exposure(const RlVector3* tristimuli, uint32_t w, uint32_t h) {
    float n = (float)(w * h);
    uint64_t count = (uint64_t)w * h;
    float sum = 0.0f;
    for (uint64_t i = 0; i < count; ++i) sum = sum + tristimuli[i].y;
    float mean = sum / n;
    float sq = 0.0f;
    for (uint64_t i = 0; i < count; ++i) sq = sq + tristimuli[i].y * tristimuli[i].y;
    float sqr_mean = sq / n;
    float variance = sqr_mean - mean * mean;
    return mean + sqrtf(variance);
}
void tonemap(const RlVector3* tristimuli, uint32_t w, uint32_t h, uint8_t* rgb, float* srgb_float, float* max_out) {
    float max_intensity = find_exposure(tristimuli, w, h);
    if (max_out) *max_out = max_intensity;
    float ln_4 = rl_logf(4.0f);
    uint64_t count = (uint64_t)w * h;
    for (uint64_t i = 0; i < count; ++i) {
        Vector3 cie{rl_logf_full(tristimuli[i].x / max_intensity + 1.0f) / ln_4,
                    rl_logf_full(tristimuli[i].y / max_intensity + 1.0f) / ln_4,
                    rl_logf_full(tristimuli[i].z / max_intensity + 1.0f) / ln_4};
        Vector3 rgbv = srgb_transform(cie);
        float r = clamp01(rgbv.x), g = clamp01(rgbv.y), b = clamp01(rgbv.z);
        if (srgb_float) {
            srgb_float[3 * i] = r;
            srgb_float[3 * i + 1] = g;
            srgb_float[3 * i + 2] = b;
        }
        if (rgb) {
            rgb[3 * i] = rl_to_u8(r * 255.0f);
            rgb[3 * i + 1] = rl_to_u8(g * 255.0f);
            rgb[3 * i + 2] = rl_to_u8(b * 255.0f);
        }
    }
}

} // namespace

// ---- C entry points for tests / bench (ctypes) ----------------------------------------------------

extern "C" {

void* oracle_scene_create(const RlObjectDesc* objs, uint32_t n, const RlCameraDesc* cam) {
    return scene_from_desc(objs, n, cam);
}
void oracle_scene_destroy(void* s) { delete (Scene*)s; }

// App::set_up_scene (app.rs:166-363) as a description; returns the object count.
uint32_t oracle_demo_scene_desc(int seeds, RlObjectDesc* out, uint32_t cap, RlCameraDesc* cam) {
    std::vector<RlObjectDesc> v = demo_scene_desc(seeds);
    if (cam) *cam = demo_camera_desc();
    if (out && cap >= v.size()) memcpy(out, v.data(), v.size() * sizeof(RlObjectDesc));
    return (uint32_t)v.size();
}

// TraceUnit::render for paths [first, first+n) on one thread.
void oracle_render(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first, uint64_t n,
                   RlMappedPhoton* photons, uint64_t* segments) {
    render(*(Scene*)scene, w, h, seed, stream, first, n, photons, segments);
}

// The same on `threads` threads (contiguous slices); photons may be NULL to time tracing only.
// Returns wall seconds.
double oracle_render_mt(void* scene, uint32_t w, uint32_t h, uint64_t seed, uint32_t stream, uint64_t first, uint64_t n,
                        RlMappedPhoton* photons, uint64_t* segments, uint32_t threads) {
    if (threads == 0) threads = 1;
    std::vector<uint64_t> segs(threads, 0);
    std::vector<std::thread> pool;
    auto t0 = std::chrono::steady_clock::now();
    for (uint32_t k = 0; k < threads; ++k) {
        uint64_t lo = n * k / threads, hi = n * (k + 1) / threads;
        pool.emplace_back([=, &segs]() {
            std::vector<RlMappedPhoton> scratch;
            RlMappedPhoton* dst = photons ? photons + lo : nullptr;
            if (!dst) {
                scratch.resize(hi - lo);
                dst = scratch.data();
            }
            render(*(Scene*)scene, w, h, seed, stream, first + lo, hi - lo, dst, &segs[k]);
        });
    }
    for (auto& t : pool) t.join();
    auto t1 = std::chrono::steady_clock::now();
    uint64_t total = 0;
    for (uint64_t s : segs) total += s;
    if (segments) *segments = total;
    return std::chrono::duration<double>(t1 - t0).count();
}

void oracle_plot(RlVector3* buffer, uint32_t w, uint32_t h, const RlMappedPhoton* photons, uint64_t n) {
    plot(buffer, w, h, photons, n);
}
void oracle_accumulate(RlVector3* acc, RlVector3* comp, const RlVector3* px, uint64_t n) { accumulate(acc, comp, px, n); }
void oracle_tonemap(const RlVector3* tristimuli, uint32_t w, uint32_t h, uint8_t* rgb, float* srgb_float, float* max_out) {
    tonemap(tristimuli, w, h, rgb, srgb_float, max_out);
}

// Single intersection of one object of a scene, for known-answer tests.  Returns 1 on hit.
int oracle_intersect_object(void* scene, uint32_t index, const float* origin, const float* direction, float* out10) {
    Scene* s = (Scene*)scene;
    Ray ray{Vector3{origin[0], origin[1], origin[2]}, Vector3{direction[0], direction[1], direction[2]}, 0.0f, 1.0f};
    OptIsect r = s->objects[index].surface->intersect(ray);
    if (!r.some) return 0;
    const Intersection& i = r.i;
    float v[10] = {i.position.x, i.position.y, i.position.z, i.normal.x, i.normal.y, i.normal.z,
                   i.tangent.x, i.tangent.y, i.tangent.z, i.distance};
    memcpy(out10, v, sizeof v);
    return 1;
}
// Scene::intersect: returns the object index or -1.
int oracle_scene_intersect(void* scene, const float* origin, const float* direction, float* out10) {
    Scene* s = (Scene*)scene;
    Ray ray{Vector3{origin[0], origin[1], origin[2]}, Vector3{direction[0], direction[1], direction[2]}, 0.0f, 1.0f};
    Intersection i{};
    const Object* o = s->intersect(ray, &i);
    if (!o) return -1;
    float v[10] = {i.position.x, i.position.y, i.position.z, i.normal.x, i.normal.y, i.normal.z,
                   i.tangent.x, i.tangent.y, i.tangent.z, i.distance};
    memcpy(out10, v, sizeof v);
    return (int)(o - s->objects.data());
}

// Pure functions for known-answer tests.
void oracle_tristimulus(float wavelength, float* xyz) {
    Vector3 v = get_tristimulus(wavelength);
    xyz[0] = v.x; xyz[1] = v.y; xyz[2] = v.z;
}
float oracle_sf10_ior(float wavelength) { return Sf10GlassMaterial::get_index_of_refraction(wavelength); }
float oracle_black_body(float kelvins, float intensity, float wavelength, float* norm) {
    BlackBodyMaterial m(kelvins, intensity);
    if (norm) *norm = m.normalisation_factor;
    return m.get_intensity(wavelength);
}
void oracle_srgb(const float* xyz, float* rgb) {
    Vector3 v = srgb_transform(Vector3{xyz[0], xyz[1], xyz[2]});
    rgb[0] = v.x; rgb[1] = v.y; rgb[2] = v.z;
}
// Camera at time t: out = position(3), orientation xyzw(4), fov, focal, screen_distance.
void oracle_camera(const RlCameraDesc* cd, float t, float* out10) {
    Camera c = make_camera(*cd, t);
    float v[10] = {c.position.x, c.position.y, c.position.z, c.orientation.x, c.orientation.y, c.orientation.z,
                   c.orientation.w, c.field_of_view, c.focal_distance, 1.0f / rl_tanf(c.field_of_view * 0.5f)};
    memcpy(out10, v, sizeof v);
}
// Material bounce for a given (seed, stream, path, block): in = origin(3) dir(3) wavelength,
// isect = position(3) normal(3) tangent(3); out = origin(3) dir(3) probability.
int oracle_material_bounce(uint32_t kind, float m0, float m1, float m2, const float* in7, const float* isect9, uint64_t seed,
                           uint32_t stream, uint64_t path, uint32_t block, float* out7) {
    std::shared_ptr<Material> m;
    switch (kind) {
    case RL_MATERIAL_DIFFUSE_GREY: m = std::make_shared<DiffuseGreyMaterial>(m0); break;
    case RL_MATERIAL_DIFFUSE_COLOURED: m = std::make_shared<DiffuseColouredMaterial>(m0, m1, m2); break;
    case RL_MATERIAL_GLOSSY_MIRROR: m = std::make_shared<GlossyMirrorMaterial>(m0); break;
    case RL_MATERIAL_SF10_GLASS: m = std::make_shared<Sf10GlassMaterial>(); break;
    case RL_MATERIAL_SOAP_BUBBLE: m = std::make_shared<SoapBubbleMaterial>(); break;
    default: return -1;
    }
    PathRng rng;
    rng.begin(seed, stream, path);
    rng.block = block;
    rng.cur = rl_rng_block(seed, stream, path, block);
    Ray in{Vector3{in7[0], in7[1], in7[2]}, Vector3{in7[3], in7[4], in7[5]}, in7[6], 1.0f};
    Intersection is{Vector3{isect9[0], isect9[1], isect9[2]}, Vector3{isect9[3], isect9[4], isect9[5]},
                    Vector3{isect9[6], isect9[7], isect9[8]}, 0.0f};
    Ray r = m->get_new_ray(in, is, rng);
    float v[7] = {r.origin.x, r.origin.y, r.origin.z, r.direction.x, r.direction.y, r.direction.z, r.probability};
    memcpy(out7, v, sizeof v);
    return 0;
}

// rl_math.h / rl_rng.h pass-throughs so numpy can check them.
void oracle_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    RlRngBlock b = rl_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, b.w, 16);
}
// n blocks at once: out[4 i ..] = the words of (seed, stream, path[i], block[i])
void oracle_rng_blocks(uint64_t seed, uint32_t stream, const uint64_t* path, const uint32_t* block, uint32_t* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        RlRngBlock b = rl_rng_block(seed, stream, path[i], block[i]);
        memcpy(out + 4 * i, b.w, 16);
    }
}
void oracle_rng_block(uint64_t seed, uint32_t stream, uint64_t path, uint32_t block, uint32_t* out) {
    RlRngBlock b = rl_rng_block(seed, stream, path, block);
    memcpy(out, b.w, 16);
}
// fn: 0 sin 1 cos 2 tan 3 exp 4 log 5 acos 6 closed01(bits) 7 halfopen01(bits); 8 sin 9 cos 10 exp 11 acos in their f64-evaluated forms
void oracle_math_f32(int fn, const float* x, float* y, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: y[i] = rl_sinf(x[i]); break;
        case 1: y[i] = rl_cosf(x[i]); break;
        case 2: y[i] = rl_tanf(x[i]); break;
        case 3: y[i] = rl_expf(x[i]); break;
        case 4: y[i] = rl_logf(x[i]); break;
        case 5: y[i] = rl_acosf(x[i]); break;
        case 6: y[i] = rl_closed01(rl_bits_f(x[i])); break;
        case 7: y[i] = rl_halfopen01(rl_bits_f(x[i])); break;
        case 8: y[i] = rl_sinf_d(x[i]); break;
        case 9: y[i] = rl_cosf_d(x[i]); break;
        case 10: y[i] = rl_expf_d(x[i]); break;
        case 11: y[i] = rl_acosf_d(x[i]); break;
        case 12: y[i] = Sf10GlassMaterial::get_index_of_refraction(x[i]); break; // material.rs:203-213 (the oracle's own restatement)
        default: y[i] = 0.0f;
        }
    }
}
void oracle_powf(const float* x, float e, float* y, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) y[i] = rl_powf(x[i], e);
}
void oracle_exp_f64(const double* x, double* y, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) y[i] = rl_exp_d(x[i]);
}

} // extern "C"
