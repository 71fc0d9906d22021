// oracle.cpp — CPU restatement of the reference's ray-path state machine.  TEST INFRASTRUCTURE ONLY.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product (raytracing_amd/, librt_hip.so) never links, imports or calls it.
//
// Every function follows the reference op for op in IEEE binary32 (build with -ffp-contract=off,
// no fast-math; x86-64 SSE scalar arithmetic has no excess precision), in glam 0.29.3's association
// order (Cargo.lock:2640-2649; source not in the container — assumed semantics listed in
// Vec3 is three scalar f32, dot = (x*x' + y*y') + z*z', normalize = v * (1/sqrt(v.v)),
// Vec3/f32 is true division, Mat4*point = ((X*x + Y*y) + Z*z) + W).
//
// PINNING: the reference has no tests; the only reference-derived numbers are the four XY points of
// python_files/tmp.py:4-9, reproduced by tests/test_oracle.py to the printed 3 decimals.  Agreement
// with a real `cargo run` in the last ulp at the glam/rand boundary is therefore "parity unpinned"
// (DESIGN.md §Oracle).
//
// The image part of orc_render (per-path value, sample accumulation) has no reference counterpart
// (SURVEY §8 a14); it restates DESIGN.md's definition and is the only check for it.
#include "../include/rt_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }  // Vec3 * f32
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }  // f32 * Vec3
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }  // true division
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length_squared(V3 a) { return dot(a, a); }
inline V3 normalize(V3 a)
{
    float len = std::sqrt(dot(a, a));
    float rcp = 1.0f / len;  // f32::recip
    return a * rcp;
}

struct V4 {
    float x, y, z, w;
};
inline V4 operator*(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline V4 operator-(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline V4 mul4(V4 a, V4 b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

// glam Mat4: four column vectors.
struct M4 {
    V4 c0, c1, c2, c3;
};

inline V4 mul_vec4(const M4& m, V4 v)
{
    V4 r = m.c0 * v.x;
    r = r + m.c1 * v.y;
    r = r + m.c2 * v.z;
    r = r + m.c3 * v.w;
    return r;
}
inline M4 mul_mat4(const M4& a, const M4& b) { return M4{mul_vec4(a, b.c0), mul_vec4(a, b.c1), mul_vec4(a, b.c2), mul_vec4(a, b.c3)}; }
inline M4 from_translation(V3 t) { return M4{V4{1, 0, 0, 0}, V4{0, 1, 0, 0}, V4{0, 0, 1, 0}, V4{t.x, t.y, t.z, 1}}; }
inline M4 from_rotation_y(float angle)
{
    // glam resolves with the `libm` feature (Cargo.lock:2644-2648); libm-crate sinf/cosf are not
    // bit-identical to glibc for every argument, exact for angle == 0 — parity unpinned otherwise.
    float s = std::sin(angle), c = std::cos(angle);
    return M4{V4{c, 0, -s, 0}, V4{0, 1, 0, 0}, V4{s, 0, c, 0}, V4{0, 0, 0, 1}};
}
inline M4 transpose(const M4& m)
{
    return M4{V4{m.c0.x, m.c1.x, m.c2.x, m.c3.x}, V4{m.c0.y, m.c1.y, m.c2.y, m.c3.y}, V4{m.c0.z, m.c1.z, m.c2.z, m.c3.z},
              V4{m.c0.w, m.c1.w, m.c2.w, m.c3.w}};
}
// glam's cofactor inverse (scalar formulation; exact for translation-only matrices whatever the order).
M4 inverse(const M4& m)
{
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z, m03 = m.c0.w;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z, m13 = m.c1.w;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z, m23 = m.c2.w;
    float m30 = m.c3.x, m31 = m.c3.y, m32 = m.c3.z, m33 = m.c3.w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    V4 fac0{coef00, coef00, coef02, coef03}, fac1{coef04, coef04, coef06, coef07}, fac2{coef08, coef08, coef10, coef11};
    V4 fac3{coef12, coef12, coef14, coef15}, fac4{coef16, coef16, coef18, coef19}, fac5{coef20, coef20, coef22, coef23};
    V4 vec0{m10, m00, m00, m00}, vec1{m11, m01, m01, m01}, vec2{m12, m02, m02, m02}, vec3{m13, m03, m03, m03};
    V4 inv0 = (mul4(vec1, fac0) - mul4(vec2, fac1)) + mul4(vec3, fac2);
    V4 inv1 = (mul4(vec0, fac0) - mul4(vec2, fac3)) + mul4(vec3, fac4);
    V4 inv2 = (mul4(vec0, fac1) - mul4(vec1, fac3)) + mul4(vec3, fac5);
    V4 inv3 = (mul4(vec0, fac2) - mul4(vec1, fac4)) + mul4(vec2, fac5);
    V4 sa{1, -1, 1, -1}, sb{-1, 1, -1, 1};
    M4 inv{mul4(inv0, sa), mul4(inv1, sb), mul4(inv2, sa), mul4(inv3, sb)};
    V4 col0{inv.c0.x, inv.c1.x, inv.c2.x, inv.c3.x};
    V4 d0 = mul4(m.c0, col0);
    float det = ((d0.x + d0.y) + d0.z) + d0.w;
    float rcp = 1.0f / det;
    return M4{inv.c0 * rcp, inv.c1 * rcp, inv.c2 * rcp, inv.c3 * rcp};
}
inline V3 transform_point3(const M4& m, V3 p)
{
    V4 r = m.c0 * p.x;
    r = m.c1 * p.y + r;
    r = m.c2 * p.z + r;
    r = m.c3 + r;
    return v3(r.x, r.y, r.z);
}
inline V3 transform_vector3(const M4& m, V3 p)
{
    V4 r = m.c0 * p.x;
    r = m.c1 * p.y + r;
    r = m.c2 * p.z + r;
    return v3(r.x, r.y, r.z);
}

struct Ray {
    V3 origin, direction;
    float current_ior;
};
struct Hit {
    float t;
    V3 point, normal;
    bool front_face;
    uint32_t mat_kind;
    float mat_param;
};

// primitives/sphere.rs:12-74 — returns the FIRST element of the hit list (scene.rs:55 takes hits.first()).
inline bool sphere_first_hit(V3 center, float radius, const Ray& ray, float t_min, float t_max, Hit& out)
{
    V3 oc = ray.origin - center;
    float a = length_squared(ray.direction);
    float half_b = dot(oc, ray.direction);
    float c = length_squared(oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = std::sqrt(discriminant);
    float t1 = (-half_b - sqrtd) / a;
    float t = 0;
    bool have = false;
    if (t1 > t_min && t1 < t_max) {
        t = t1;
        have = true;
    } else if (discriminant > 1e-6f) {
        float t2 = (-half_b + sqrtd) / a;
        if (t2 > t_min && t2 < t_max) {
            t = t2;
            have = true;
        }
    }
    if (!have) return false;
    V3 point = ray.origin + t * ray.direction;
    V3 outward = (point - center) / radius;
    bool front = dot(ray.direction, outward) < 0.0f;
    out.t = t;
    out.point = point;
    out.normal = front ? outward : -outward;
    out.front_face = front;
    return true;
}

// primitives/plane.rs:12-54
inline bool plane_first_hit(V3 ppoint, V3 normal, const Ray& ray, float t_min, float t_max, Hit& out)
{
    float denom = dot(normal, ray.direction);
    if (std::fabs(denom) < 1e-6f) return false;
    float t = dot(ppoint - ray.origin, normal) / denom;
    if (t < t_min || t_max < t) return false;
    V3 point = ray.origin + t * ray.direction;
    bool front = dot(ray.direction, normal) < 0.0f;
    out.t = t;
    out.point = point;
    out.normal = front ? normal : -normal;
    out.front_face = front;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Generic `Hittable::intersect_all` (primitives/mod.rs:46-48) for the shapes that are not on the fast
// sphere/plane path: every hit in (t_min, t_max), in the order the reference pushes them.
// ---------------------------------------------------------------------------------------------------
struct Prim {
    virtual ~Prim() {}
    virtual void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const = 0;
};

struct SpherePrim : Prim {  // sphere.rs:12-74
    V3 center;
    float radius;
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        V3 oc = ray.origin - center;
        float a = length_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = length_squared(oc) - radius * radius;
        float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return;
        float sqrtd = std::sqrt(discriminant);
        auto push = [&](float t) {
            V3 point = ray.origin + t * ray.direction;
            V3 outward = (point - center) / radius;
            bool front = dot(ray.direction, outward) < 0.0f;
            Hit h{};
            h.t = t; h.point = point; h.normal = front ? outward : -outward; h.front_face = front;
            out.push_back(h);
        };
        float t1 = (-half_b - sqrtd) / a;
        if (t1 > t_min && t1 < t_max) push(t1);
        if (discriminant > 1e-6f) {
            float t2 = (-half_b + sqrtd) / a;
            if (t2 > t_min && t2 < t_max) push(t2);
        }
    }
};

struct PlanePrim : Prim {  // plane.rs:12-54
    V3 point, normal;
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        Hit h{};
        if (plane_first_hit(point, normal, ray, t_min, t_max, h)) out.push_back(h);
    }
};

struct BoxPrim : Prim {  // axis_aligned_box.rs:12-95
    V3 mn, mx;
    static float comp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
    V3 calculate_normal(V3 point) const
    {
        const float epsilon = 1e-4f;
        V3 a = point - mn, b = point - mx;
        if (std::fabs(a.x) < epsilon) return v3(-1, 0, 0);
        if (std::fabs(b.x) < epsilon) return v3(1, 0, 0);
        if (std::fabs(a.y) < epsilon) return v3(0, -1, 0);
        if (std::fabs(b.y) < epsilon) return v3(0, 1, 0);
        if (std::fabs(a.z) < epsilon) return v3(0, 0, -1);
        if (std::fabs(b.z) < epsilon) return v3(0, 0, 1);
        return v3(0, 0, 0);
    }
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        float tmin = t_min, tmax = t_max;
        for (int i = 0; i < 3; ++i) {
            float inv_d = 1.0f / comp(ray.direction, i);
            float t0 = (comp(mn, i) - comp(ray.origin, i)) * inv_d;
            float t1 = (comp(mx, i) - comp(ray.origin, i)) * inv_d;
            if (inv_d < 0.0f) std::swap(t0, t1);
            tmin = std::fmax(tmin, t0);  // f32::max: NaN-ignoring
            tmax = std::fmin(tmax, t1);
            if (tmax <= tmin) return;
        }
        Hit h1{}, h2{};
        h1.t = tmin;
        h1.point = ray.origin + tmin * ray.direction;
        h1.normal = calculate_normal(h1.point);
        h1.front_face = dot(ray.direction, h1.normal) < 0.0f;
        out.push_back(h1);
        h2.t = tmax;
        h2.point = ray.origin + tmax * ray.direction;
        h2.normal = -calculate_normal(h2.point);
        h2.front_face = dot(ray.direction, h2.normal) < 0.0f;
        out.push_back(h2);
    }
};

struct InfCylinderPrim : Prim {  // infinite_cylinder.rs:13-84
    V3 axis_point, axis_dir;
    float radius;
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        V3 oc = ray.origin - axis_point;
        float d_dot_v = dot(ray.direction, axis_dir);
        V3 d_perp = ray.direction - d_dot_v * axis_dir;
        float oc_dot_v = dot(oc, axis_dir);
        V3 oc_perp = oc - oc_dot_v * axis_dir;
        float a = length_squared(d_perp);
        float b = 2.0f * dot(oc_perp, d_perp);
        float c = length_squared(oc_perp) - radius * radius;
        if (std::fabs(a) < 1e-6f) return;
        float discriminant = b * b - 4.0f * a * c;
        if (discriminant < 0.0f) return;
        float sqrtd = std::sqrt(discriminant);
        float ts[2] = {(-b - sqrtd) / (2.0f * a), (-b + sqrtd) / (2.0f * a)};
        for (float t : ts) {
            if (t > t_min && t < t_max) {
                V3 point = ray.origin + t * ray.direction;
                V3 p_minus_a = point - axis_point;
                float projection = dot(p_minus_a, axis_dir);
                V3 point_on_axis = axis_point + projection * axis_dir;
                V3 outward = normalize(point - point_on_axis);
                bool front = dot(ray.direction, outward) < 0.0f;
                Hit h{};
                h.t = t; h.point = point; h.normal = front ? outward : -outward; h.front_face = front;
                out.push_back(h);
            }
        }
    }
};

struct InfConePrim : Prim {  // infinite_cone.rs:14-86
    V3 vertex, axis_dir;
    float cos_angle_sq;
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        V3 co = ray.origin - vertex;
        float d_dot_v = dot(ray.direction, axis_dir);
        float co_dot_v = dot(co, axis_dir);
        float a = d_dot_v * d_dot_v - cos_angle_sq;  // powi(2)
        float b = 2.0f * (d_dot_v * co_dot_v - dot(ray.direction, co) * cos_angle_sq);
        float c = co_dot_v * co_dot_v - length_squared(co) * cos_angle_sq;
        float discriminant = b * b - 4.0f * a * c;
        if (discriminant < 0.0f) return;
        float sqrtd = std::sqrt(discriminant);
        float ts[2] = {(-b - sqrtd) / (2.0f * a), (-b + sqrtd) / (2.0f * a)};
        for (float t : ts) {
            if (t > t_min && t < t_max) {
                V3 point = ray.origin + t * ray.direction;
                V3 pv = point - vertex;
                float m = dot(pv, axis_dir);
                V3 outward = normalize(m * axis_dir - pv * cos_angle_sq);
                bool front = dot(ray.direction, outward) < 0.0f;
                Hit h{};
                h.t = t; h.point = point; h.normal = front ? outward : -outward; h.front_face = front;
                out.push_back(h);
            }
        }
    }
};

// Triangle mesh — no reference counterpart; restates DESIGN.md §4.9.  intersect_all returns the single
// nearest triangle hit in (t_min, t_max) (ties: lowest triangle index), found here by brute force over every triangle;
// the GPU path finds the same hit through a BVH.  Moller-Trumbore in strict f32, plus the padded-box rule below.
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
struct MeshPrim : Prim {
    std::vector<V3> v0, e1, e2;
    // the hit point must lie in the triangle's padded box: min / max over v0, v0+e1, v0+e2, grown by 1e-5 (|.|max + extent) + 1e-6
    static bool box_holds(float a, float ea, float eb, float p)
    {
        float b = a + ea, c = a + eb;
        float lo = std::fmin(a, std::fmin(b, c)), hi = std::fmax(a, std::fmax(b, c));
        float mag = std::fmax(std::fabs(lo), std::fabs(hi));
        float pad = 1e-5f * (mag + (hi - lo)) + 1e-6f;
        return p >= lo - pad && p <= hi + pad;
    }
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        float best = std::numeric_limits<float>::infinity();
        long best_k = -1;
        for (size_t k = 0; k < v0.size(); ++k) {
            V3 pvec = cross(ray.direction, e2[k]);
            float det = dot(e1[k], pvec);
            if (std::fabs(det) < 1e-12f) continue;
            float inv = 1.0f / det;
            V3 tvec = ray.origin - v0[k];
            float u = dot(tvec, pvec) * inv;
            if (u < 0.0f || u > 1.0f) continue;
            V3 qvec = cross(tvec, e1[k]);
            float v = dot(ray.direction, qvec) * inv;
            if (v < 0.0f || u + v > 1.0f) continue;
            float t = dot(e2[k], qvec) * inv;
            if (!(t > t_min && t < t_max)) continue;
            V3 p = ray.origin + t * ray.direction;
            if (!(box_holds(v0[k].x, e1[k].x, e2[k].x, p.x) && box_holds(v0[k].y, e1[k].y, e2[k].y, p.y) && box_holds(v0[k].z, e1[k].z, e2[k].z, p.z))) continue;
            if (t < best) {
                best = t;
                best_k = (long)k;
            }
        }
        if (best_k < 0) return;
        V3 n = cross(e1[best_k], e2[best_k]);
        bool front = dot(ray.direction, n) < 0.0f;
        Hit h{};
        h.t = best;
        h.point = ray.origin + best * ray.direction;
        h.normal = front ? n : -n;
        h.front_face = front;
        out.push_back(h);
    }
};

struct CsgPrim : Prim {  // csg.rs:3-76
    std::unique_ptr<Prim> left, right;
    uint32_t op;  // RT_SHAPE_UNION / INTERSECTION / DIFFERENCE
    bool inside(bool l, bool r) const { return op == RT_SHAPE_UNION ? (l || r) : (op == RT_SHAPE_INTERSECTION ? (l && r) : (l && !r)); }
    void intersect_all(const Ray& ray, float t_min, float t_max, std::vector<Hit>& out) const override
    {
        std::vector<Hit> hl, hr;
        left->intersect_all(ray, t_min, t_max, hl);
        right->intersect_all(ray, t_min, t_max, hr);
        std::vector<Hit> all = hl;
        all.insert(all.end(), hr.begin(), hr.end());
        std::stable_sort(all.begin(), all.end(), [](const Hit& x, const Hit& y) { return x.t < y.t; });
        bool in_left = false, in_right = false;
        for (const Hit& hit : all) {
            bool on_left = false;
            for (const Hit& h : hl) on_left = on_left || (std::fabs(h.t - hit.t) < 1e-6f);
            bool was = inside(in_left, in_right);
            if (on_left) in_left = !in_left; else in_right = !in_right;
            bool is = inside(in_left, in_right);
            if (was != is) {
                Hit h = hit;
                if (op == RT_SHAPE_DIFFERENCE && !on_left) {
                    h.normal = -hit.normal;
                    h.front_face = !hit.front_face;
                }
                out.push_back(h);
            }
        }
    }
};

inline std::unique_ptr<Prim> mk_plane(V3 point, V3 normal)
{
    auto p = std::make_unique<PlanePrim>();
    p->point = point; p->normal = normal;
    return p;
}
inline std::unique_ptr<Prim> mk_csg(std::unique_ptr<Prim> l, std::unique_ptr<Prim> r, uint32_t op)
{
    auto c = std::make_unique<CsgPrim>();
    c->left = std::move(l); c->right = std::move(r); c->op = op;
    return c;
}
const float kRadsPerDeg = 3.14159265358979323846f / 180.0f;  // f32::to_radians multiplies by PI/180 (f32 constant)

// ShapeConfig::into_with (shape_config.rs:53-149), wedge.rs:10-78, lens.rs:11-76.
std::unique_ptr<Prim> build_shape(uint32_t shape, const float* sp, const RtShapeNode* nodes, uint64_t n_nodes, int depth)
{
    switch (shape) {
        case RT_SHAPE_SPHERE: {
            auto s = std::make_unique<SpherePrim>();
            s->center = v3(0, 0, 0); s->radius = sp[0];
            return s;
        }
        case RT_SHAPE_PLANE: return mk_plane(v3(0, 0, 0), v3(sp[0], sp[1], sp[2]));
        case RT_SHAPE_BOX: {
            V3 h = v3(sp[0], sp[1], sp[2]) / 2.0f;
            auto b = std::make_unique<BoxPrim>();
            b->mn = -h; b->mx = h;
            return b;
        }
        case RT_SHAPE_CYLINDER: {
            float half_height = sp[0] / 2.0f;
            auto body = std::make_unique<InfCylinderPrim>();
            body->axis_point = v3(0, 0, 0); body->axis_dir = v3(0, 1, 0); body->radius = sp[1];
            auto capped = mk_csg(std::move(body), mk_plane(v3(0, half_height, 0), v3(0, -1, 0)), RT_SHAPE_INTERSECTION);
            return mk_csg(std::move(capped), mk_plane(v3(0, -half_height, 0), v3(0, 1, 0)), RT_SHAPE_INTERSECTION);
        }
        case RT_SHAPE_CONE: {
            auto cone = std::make_unique<InfConePrim>();
            cone->vertex = v3(0, 0, 0);
            cone->axis_dir = normalize(v3(0, 1, 0));
            float cs = std::cos(sp[0] * kRadsPerDeg);
            cone->cos_angle_sq = cs * cs;
            return mk_csg(std::move(cone), mk_plane(v3(0, sp[1], 0), v3(0, -1, 0)), RT_SHAPE_INTERSECTION);
        }
        case RT_SHAPE_WEDGE: {
            float half_depth = sp[2] / 2.0f;
            float ang = sp[3] * kRadsPerDeg;
            float angle_cos = std::cos(ang), angle_sin = std::sin(ang);
            auto c1 = mk_csg(mk_plane(v3(0, 0, 0), v3(0, 1, 0)), mk_plane(v3(0, 0, 0), v3(1, 0, 0)), RT_SHAPE_INTERSECTION);
            auto c2 = mk_csg(std::move(c1), mk_plane(v3(0, 0, 0), v3(-angle_sin, angle_cos, 0.0f)), RT_SHAPE_INTERSECTION);
            auto c3 = mk_csg(std::move(c2), mk_plane(v3(0, 0, half_depth), v3(0, 0, -1)), RT_SHAPE_INTERSECTION);
            return mk_csg(std::move(c3), mk_plane(v3(0, 0, -half_depth), v3(0, 0, 1)), RT_SHAPE_INTERSECTION);
        }
        case RT_SHAPE_LENS: {
            float half_thickness = sp[0] / 2.0f, r1 = sp[2], r2 = sp[3];
            std::unique_ptr<Prim> s1, s2;
            if (std::isfinite(r1)) {
                auto s = std::make_unique<SpherePrim>();
                s->center = v3(0, 0, -half_thickness + r1); s->radius = std::fabs(r1);
                s1 = std::move(s);
            } else {
                s1 = mk_plane(v3(0, 0, -half_thickness), v3(0, 0, 1));
            }
            if (std::isfinite(r2)) {
                auto s = std::make_unique<SpherePrim>();
                s->center = v3(0, 0, half_thickness + r2); s->radius = std::fabs(r2);
                s2 = std::move(s);
            } else {
                s2 = mk_plane(v3(0, 0, half_thickness), v3(0, 0, -1));
            }
            auto cyl = std::make_unique<InfCylinderPrim>();
            cyl->axis_point = v3(0, 0, 0); cyl->axis_dir = v3(0, 0, 1); cyl->radius = sp[1] / 2.0f;
            return mk_csg(mk_csg(std::move(s1), std::move(s2), RT_SHAPE_INTERSECTION), std::move(cyl), RT_SHAPE_INTERSECTION);
        }
        case RT_SHAPE_TREE: {
            uint32_t root;
            std::memcpy(&root, &sp[0], 4);
            if (root >= n_nodes || depth > 32) return nullptr;
            const RtShapeNode& n = nodes[root];
            return build_shape(n.shape, n.shape_param, nodes, n_nodes, depth + 1);
        }
        case RT_SHAPE_UNION:
        case RT_SHAPE_INTERSECTION:
        case RT_SHAPE_DIFFERENCE: return nullptr;  // operators are reached through build_node
    }
    return nullptr;
}
std::unique_ptr<Prim> build_node(const RtShapeNode* nodes, uint64_t n_nodes, uint32_t idx, int depth)
{
    if (idx >= n_nodes || depth > 32) return nullptr;
    const RtShapeNode& n = nodes[idx];
    if (n.shape >= RT_SHAPE_UNION && n.shape <= RT_SHAPE_DIFFERENCE) {
        auto l = build_node(nodes, n_nodes, n.a, depth + 1), r = build_node(nodes, n_nodes, n.b, depth + 1);
        if (!l || !r) return nullptr;
        return mk_csg(std::move(l), std::move(r), n.shape);
    }
    if (n.shape == RT_SHAPE_TREE) return nullptr;
    return build_shape(n.shape, n.shape_param, nodes, n_nodes, depth + 1);
}

struct Object {  // Transform(Box<primitive>) — transform.rs:4-17, object_config.rs:16-30
    uint32_t shape;
    uint32_t mat_kind;
    float mat_param;
    float radius;
    V3 normal;
    M4 transform, inverse_transform, inverse_transpose;
    std::shared_ptr<Prim> prim;  // shapes other than Sphere / Plane
};

Object build_object(const RtObject& o, const RtShapeNode* nodes = nullptr, uint64_t n_nodes = 0, const RtMesh* meshes = nullptr,
                    uint64_t n_meshes = 0)
{
    Object r{};
    r.shape = o.shape;
    r.mat_kind = o.material_kind;
    r.mat_param = o.material_param;
    r.radius = o.shape_param[0];
    r.normal = v3(o.shape_param[0], o.shape_param[1], o.shape_param[2]);
    M4 translation = from_translation(v3(o.position[0], o.position[1], o.position[2]));
    M4 rotation = from_rotation_y(o.rotation_y_deg * kRadsPerDeg);  // f32::to_radians
    r.transform = mul_mat4(translation, rotation);
    r.inverse_transform = inverse(r.transform);
    r.inverse_transpose = transpose(r.inverse_transform);
    if (o.shape != RT_SHAPE_SPHERE && o.shape != RT_SHAPE_PLANE) {
        std::unique_ptr<Prim> p;
        if (o.shape == RT_SHAPE_TREE) {
            uint32_t root;
            std::memcpy(&root, &o.shape_param[0], 4);
            p = build_node(nodes, n_nodes, root, 0);
        } else if (o.shape == RT_SHAPE_MESH) {
            uint32_t mi;
            std::memcpy(&mi, &o.shape_param[0], 4);
            if (mi < n_meshes) {
                const RtMesh& m = meshes[mi];
                auto mp = std::make_unique<MeshPrim>();
                for (uint64_t k = 0; k < m.n_triangles; ++k) {
                    auto vert = [&](uint32_t idx) { return v3(m.vertices[idx * 3 + 0], m.vertices[idx * 3 + 1], m.vertices[idx * 3 + 2]); };
                    V3 a = vert(m.indices[k * 3 + 0]), b = vert(m.indices[k * 3 + 1]), c = vert(m.indices[k * 3 + 2]);
                    mp->v0.push_back(a);
                    mp->e1.push_back(b - a);
                    mp->e2.push_back(c - a);
                }
                p = std::move(mp);
            }
        } else {
            p = build_shape(o.shape, o.shape_param, nodes, n_nodes, 0);
        }
        r.prim = std::move(p);
    }
    return r;
}

// transform.rs:20-51 around the wrapped primitive (local centre / local plane point = origin:
// shape_config.rs:56-60, 69-73).
inline bool object_first_hit(const Object& ob, const Ray& ray, float t_min, float t_max, Hit& out)
{
    Ray local;
    local.origin = transform_point3(ob.inverse_transform, ray.origin);
    local.direction = transform_vector3(ob.inverse_transform, ray.direction);
    local.current_ior = ray.current_ior;
    bool ok;
    if (ob.shape == RT_SHAPE_SPHERE) {
        ok = sphere_first_hit(v3(0, 0, 0), ob.radius, local, t_min, t_max, out);
    } else if (ob.shape == RT_SHAPE_PLANE) {
        ok = plane_first_hit(v3(0, 0, 0), ob.normal, local, t_min, t_max, out);
    } else {
        if (!ob.prim) return false;
        std::vector<Hit> hits;
        ob.prim->intersect_all(local, t_min, t_max, hits);
        ok = !hits.empty();
        if (ok) out = hits.front();  // scene.rs:55 hits.first()
    }
    if (!ok) return false;
    out.point = transform_point3(ob.transform, out.point);
    out.normal = normalize(transform_vector3(ob.inverse_transpose, out.normal));
    out.mat_kind = ob.mat_kind;
    out.mat_param = ob.mat_param;
    return true;
}

// scene.rs:9-11
inline V3 reflect(V3 incident, V3 normal) { return incident - 2.0f * dot(incident, normal) * normal; }

// scene.rs:14-27 (the println! on total internal reflection, scene.rs:19, is dropped)
inline bool refract(V3 incident, V3 normal, float ior_ratio, V3& out)
{
    float cos_theta = std::fmin(dot(-incident, normal), 1.0f);
    float sin_theta_squared = 1.0f - cos_theta * cos_theta;
    if (ior_ratio * ior_ratio * sin_theta_squared > 1.0f) return false;
    V3 perp = ior_ratio * (incident + cos_theta * normal);
    V3 parallel = -std::sqrt(std::fabs(1.0f - length_squared(perp))) * normal;
    out = normalize(perp + parallel);
    return true;
}

// Counter-based replacement for rand::thread_rng().gen::<f32>() (scene.rs:88): Philox4x32-10 keyed by
// the seed, counter = (path id lo, path id hi, bounce, tag); rand 0.8 `Standard` f32 mapping
// (next_u32() >> 8) * 2^-24.
inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
inline void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
inline float uniform01(uint64_t seed, uint64_t path_id, uint32_t bounce)
{
    uint32_t c0 = (uint32_t)path_id, c1 = (uint32_t)(path_id >> 32), c2 = bounce, c3 = 0x52415953u;
    philox4x32_10(c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

struct Scene {
    std::vector<Object> objects;
};

// One path of Scene::simulate_rays (scene.rs:43-103).  Returns the number of bounce-loop iterations
// executed (the "ray segments" of the Mrays/s metric).
struct PathOut {
    float* points;            // [(max_bounces+1)*3] or null
    uint32_t* n_points;       // or null
    RtBounceRecord* records;  // [max_bounces] or null
};
uint32_t trace_path(const Scene& sc, Ray ray, RtSettings st, uint64_t seed, uint64_t path_id, const PathOut& po,
                    bool* escaped, V3* final_dir)
{
    uint32_t np = 0, segments = 0;
    auto push = [&](V3 p) {
        if (po.points) {
            po.points[np * 3 + 0] = p.x;
            po.points[np * 3 + 1] = p.y;
            po.points[np * 3 + 2] = p.z;
        }
        ++np;
    };
    push(ray.origin);
    *escaped = false;
    for (uint32_t bounce = 0; bounce < st.max_bounces; ++bounce) {
        ++segments;
        float t_closest = std::numeric_limits<float>::infinity();
        bool have = false;
        Hit closest{};
        int32_t closest_id = -1;
        for (size_t k = 0; k < sc.objects.size(); ++k) {
            Hit h;
            if (object_first_hit(sc.objects[k], ray, 0.001f, t_closest, h)) {
                if (h.t < t_closest) {
                    t_closest = h.t;
                    closest = h;
                    closest_id = (int32_t)k;
                    have = true;
                }
            }
        }
        RtBounceRecord rec;
        std::memset(&rec, 0, sizeof rec);
        if (have) {
            push(closest.point);
            if (closest.mat_kind == RT_MAT_MIRROR) {
                ray.direction = reflect(ray.direction, closest.normal);
            } else if (closest.mat_kind == RT_MAT_GLASS) {
                float n1 = ray.current_ior;
                float n2 = closest.front_face ? closest.mat_param : 1.0f;
                float ior_ratio = n1 / n2;
                V3 refr;
                if (refract(ray.direction, closest.normal, ior_ratio, refr)) {
                    ray.direction = refr;
                    ray.current_ior = n2;
                } else {
                    ray.direction = reflect(ray.direction, closest.normal);
                }
            } else {  // HalfMirror
                if (uniform01(seed, path_id, bounce) < closest.mat_param) ray.direction = reflect(ray.direction, closest.normal);
            }
            ray.origin = closest.point + ray.direction * 0.001f;
            if (po.records) {
                rec.obj_id = closest_id;
                rec.t = closest.t;
                rec.point[0] = closest.point.x; rec.point[1] = closest.point.y; rec.point[2] = closest.point.z;
                rec.normal[0] = closest.normal.x; rec.normal[1] = closest.normal.y; rec.normal[2] = closest.normal.z;
                rec.front_face = closest.front_face ? 1u : 0u;
                rec.dir_out[0] = ray.direction.x; rec.dir_out[1] = ray.direction.y; rec.dir_out[2] = ray.direction.z;
                rec.ior_out = ray.current_ior;
                po.records[bounce] = rec;
            }
        } else {
            V3 far = ray.origin + ray.direction * st.infinity_distance;
            push(far);
            if (po.records) {
                rec.obj_id = -1;
                rec.t = std::numeric_limits<float>::infinity();
                rec.point[0] = far.x; rec.point[1] = far.y; rec.point[2] = far.z;
                rec.ior_out = ray.current_ior;
                po.records[bounce] = rec;
            }
            *escaped = true;
            break;
        }
    }
    if (po.n_points) *po.n_points = np;
    *final_dir = ray.direction;
    return segments;
}

// scene_config.rs:60-108 for one cell (i = cell / count_v, j = cell % count_v).
Ray generate_ray(const RtRayGenerator& g, uint64_t cell)
{
    uint32_t i = (uint32_t)(cell / g.count_v), j = (uint32_t)(cell % g.count_v);
    V3 a = v3(g.a[0], g.a[1], g.a[2]);
    V3 u_step = v3(g.u[0], g.u[1], g.u[2]) / (float)g.count_u;
    V3 v_step = v3(g.v[0], g.v[1], g.v[2]) / (float)g.count_v;
    Ray r;
    r.current_ior = g.current_ior;
    if (g.kind == RT_GEN_PARALLEL_GRID) {
        V3 dir = normalize(v3(g.direction[0], g.direction[1], g.direction[2]));
        r.origin = (a + ((float)i * u_step)) + ((float)j * v_step);
        r.direction = dir;
    } else {
        V3 corner = v3(g.corner[0], g.corner[1], g.corner[2]);
        V3 target = (corner + ((float)i * u_step)) + ((float)j * v_step);
        r.origin = a;
        r.direction = normalize(target - a);
    }
    return r;
}

// DESIGN.md "path value": a path that escapes contributes sky(final direction); one that exhausts
// max_bounces contributes 0.  sky(d): w = 0.5*(d.y + 1); rgb = (1-w)*white + w*(0.5, 0.7, 1.0).
inline V3 sky(V3 d)
{
    float w = 0.5f * (d.y + 1.0f);
    float k = 1.0f - w;
    return v3(k + w * 0.5f, k + w * 0.7f, k + w);
}

}  // namespace

extern "C" {

struct orc_scene {
    Scene sc;
};

orc_scene* orc_scene_create_desc(const RtSceneDesc* d)
{
    orc_scene* s = new orc_scene;
    s->sc.objects.reserve(d->n_objects);
    for (uint64_t k = 0; k < d->n_objects; ++k)
        s->sc.objects.push_back(build_object(d->objects[k], d->nodes, d->n_nodes, d->meshes, d->n_meshes));
    return s;
}
orc_scene* orc_scene_create_ex(const RtObject* objects, uint64_t n, const RtShapeNode* nodes, uint64_t n_nodes)
{
    RtSceneDesc d{objects, n, nodes, n_nodes, nullptr, 0};
    return orc_scene_create_desc(&d);
}
orc_scene* orc_scene_create(const RtObject* objects, uint64_t n) { return orc_scene_create_ex(objects, n, nullptr, 0); }
void orc_scene_destroy(orc_scene* s) { delete s; }

// Mirrors rt_hip_simulate (include/rt_hip.h).  Returns total segments.
uint64_t orc_simulate(const orc_scene* s, const RtRay* rays, uint64_t n_rays, RtSettings st, uint64_t seed,
                      uint64_t ray_id_base, float* out_points, uint32_t* out_n_points, RtBounceRecord* out_records)
{
    uint64_t total = 0;
    const uint64_t stride = (uint64_t)st.max_bounces + 1;
    if (out_records) std::memset(out_records, 0, sizeof(RtBounceRecord) * n_rays * st.max_bounces);
    if (out_points) std::memset(out_points, 0, sizeof(float) * 3 * stride * n_rays);
    for (uint64_t r = 0; r < n_rays; ++r) {
        Ray ray;
        ray.origin = v3(rays[r].origin[0], rays[r].origin[1], rays[r].origin[2]);
        ray.direction = v3(rays[r].direction[0], rays[r].direction[1], rays[r].direction[2]);
        ray.current_ior = rays[r].current_ior;
        PathOut po{out_points ? out_points + r * stride * 3 : nullptr, out_n_points ? out_n_points + r : nullptr,
                   out_records ? out_records + r * st.max_bounces : nullptr};
        bool esc;
        V3 fd;
        total += trace_path(s->sc, ray, st, seed, ray_id_base + r, po, &esc, &fd);
    }
    return total;
}

// Expand a generator into rays, in the reference's order (scene_config.rs:74-107).
void orc_generate_rays(const RtRayGenerator* gen, uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, RtRay* out)
{
    for (uint64_t m = 0; m < cell_count; ++m) {
        Ray r = generate_ray(*gen, cell_begin + m * cell_step);
        out[m].origin[0] = r.origin.x; out[m].origin[1] = r.origin.y; out[m].origin[2] = r.origin.z;
        out[m].direction[0] = r.direction.x; out[m].direction[1] = r.direction.y; out[m].direction[2] = r.direction.z;
        out[m].current_ior = r.current_ior;
    }
}

static void render_range(const orc_scene* s, const RtRayGenerator* gen, uint32_t spp, RtSettings st, uint64_t seed,
                         uint64_t cell_begin, uint64_t cell_step, uint64_t m0, uint64_t m1, float* fb, int32_t* first_id,
                         float* first_t, uint32_t* seg_count, uint64_t* total)
{
    uint64_t tot = 0;
    std::vector<RtBounceRecord> recs(st.max_bounces ? st.max_bounces : 1);
    for (uint64_t m = m0; m < m1; ++m) {
        uint64_t cell = cell_begin + m * cell_step;
        Ray ray = generate_ray(*gen, cell);
        uint64_t accx = 0, accy = 0, accz = 0;  // fixed-point 2^-24 sums (order-independent)
        uint32_t segs = 0;
        for (uint32_t sidx = 0; sidx < spp; ++sidx) {
            bool esc;
            V3 fd;
            PathOut po{nullptr, nullptr, (sidx == 0) ? recs.data() : nullptr};
            if (sidx == 0) std::memset(recs.data(), 0, sizeof(RtBounceRecord) * recs.size());
            segs += trace_path(s->sc, ray, st, seed, cell * (uint64_t)spp + sidx, po, &esc, &fd);
            if (sidx == 0) {
                if (first_id) first_id[m] = st.max_bounces ? recs[0].obj_id : -1;
                if (first_t) first_t[m] = st.max_bounces ? recs[0].t : std::numeric_limits<float>::infinity();
            }
            V3 val = esc ? sky(fd) : v3(0, 0, 0);
            auto q = [](float v) { return v == v ? (uint64_t)std::llrintf(v * 16777216.0f) : (uint64_t)0; };  // NaN path value -> 0
            accx += q(val.x);
            accy += q(val.y);
            accz += q(val.z);
        }
        const double denom = (double)spp * 16777216.0;
        fb[m * 3 + 0] = (float)((double)accx / denom);
        fb[m * 3 + 1] = (float)((double)accy / denom);
        fb[m * 3 + 2] = (float)((double)accz / denom);
        if (seg_count) seg_count[m] = segs;
        tot += segs;
    }
    *total = tot;
}

// Mirrors rt_hip_render.  n_threads <= 1: the reference's single-threaded loop; otherwise cells are
// split into contiguous ranges over std::thread workers (results identical: cells are independent).
uint64_t orc_render(const orc_scene* s, const RtRayGenerator* gen, uint32_t spp, RtSettings st, uint64_t seed,
                    uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* fb, int32_t* first_id,
                    float* first_t, uint32_t* seg_count, uint32_t n_threads)
{
    if (n_threads <= 1) {
        uint64_t tot = 0;
        render_range(s, gen, spp, st, seed, cell_begin, cell_step, 0, cell_count, fb, first_id, first_t, seg_count, &tot);
        return tot;
    }
    std::vector<std::thread> th;
    std::vector<uint64_t> tots(n_threads, 0);
    // dynamic chunking so uneven cells do not idle threads
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = 64;
    for (uint32_t t = 0; t < n_threads; ++t) {
        th.emplace_back([&, t]() {
            for (;;) {
                uint64_t m0 = next.fetch_add(chunk);
                if (m0 >= cell_count) break;
                uint64_t m1 = m0 + chunk < cell_count ? m0 + chunk : cell_count;
                uint64_t part = 0;
                render_range(s, gen, spp, st, seed, cell_begin, cell_step, m0, m1, fb, first_id, first_t, seg_count, &part);
                tots[t] += part;
            }
        });
    }
    for (auto& x : th) x.join();
    uint64_t tot = 0;
    for (auto v : tots) tot += v;
    return tot;
}

// Unit-level entry points so tests can pin individual functions (SURVEY §4 KAT list).
int orc_sphere_first_hit(const float center[3], float radius, const RtRay* ray, float t_min, float t_max, float* t,
                         float point[3], float normal[3], uint32_t* front_face)
{
    Ray r{v3(ray->origin[0], ray->origin[1], ray->origin[2]), v3(ray->direction[0], ray->direction[1], ray->direction[2]),
          ray->current_ior};
    Hit h;
    if (!sphere_first_hit(v3(center[0], center[1], center[2]), radius, r, t_min, t_max, h)) return 0;
    *t = h.t;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    *front_face = h.front_face;
    return 1;
}
int orc_plane_first_hit(const float ppoint[3], const float pnormal[3], const RtRay* ray, float t_min, float t_max, float* t,
                        float point[3], float normal[3], uint32_t* front_face)
{
    Ray r{v3(ray->origin[0], ray->origin[1], ray->origin[2]), v3(ray->direction[0], ray->direction[1], ray->direction[2]),
          ray->current_ior};
    Hit h;
    if (!plane_first_hit(v3(ppoint[0], ppoint[1], ppoint[2]), v3(pnormal[0], pnormal[1], pnormal[2]), r, t_min, t_max, h)) return 0;
    *t = h.t;
    point[0] = h.point.x; point[1] = h.point.y; point[2] = h.point.z;
    normal[0] = h.normal.x; normal[1] = h.normal.y; normal[2] = h.normal.z;
    *front_face = h.front_face;
    return 1;
}
void orc_reflect(const float i[3], const float n[3], float out[3])
{
    V3 r = reflect(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_refract(const float i[3], const float n[3], float ior_ratio, float out[3])
{
    V3 r;
    if (!refract(v3(i[0], i[1], i[2]), v3(n[0], n[1], n[2]), ior_ratio, r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
float orc_uniform01(uint64_t seed, uint64_t path_id, uint32_t bounce) { return uniform01(seed, path_id, bounce); }
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    philox4x32_10(c0, c1, c2, c3, key[0], key[1]);
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// transform / inverse of T(position)*R_y(deg) as 16 column-major floats each (for the exactness test).
void orc_object_matrices(const RtObject* o, float transform[16], float inverse_transform[16])
{
    Object ob = build_object(*o);
    std::memcpy(transform, &ob.transform, 64);
    std::memcpy(inverse_transform, &ob.inverse_transform, 64);
}

}  // extern "C"
