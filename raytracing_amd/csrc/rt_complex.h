// rt_complex.h — per-lane evaluation of the objects that are not on the sphere / plane scan path:
// AxisAlignedBox (axis_aligned_box.rs:12-95), InfiniteCylinder (infinite_cylinder.rs:13-84), InfiniteCone
// (infinite_cone.rs:26-86), CSGObject (csg.rs:9-75) and the shapes assembled from them (shape_config.rs:74-149,
// wedge.rs:10-78, lens.rs:11-76), each inside a general Transform (transform.rs:20-51; T(position) * R_y(deg)).
// Rotated spheres and planes take this path too.
//
// Such objects are few, so every lane walks them one after the other after the sphere scan, with the reference's
// own intersect_all: leaf hit lists, stable sort by t, the in/out state machine, hits.first().  The reference hands
// every object the closest t found so far in list order as t_max (scene.rs:54), and a CSG tree's result can depend on
// it (a child hit beyond t_max is dropped before the 1e-6 left/right attribution of csg.rs:33 looks for it).  The
// scan finds the closest sphere / plane of the WHOLE list first, so cx_min_key works out that prefix minimum exactly:
// it is the running best when that best sits earlier in the list, and otherwise it only matters when a leaf surface
// lies within a hair of the running best, in which case the lane replays the list in order.
#pragma once
#include "rt_device.h"

namespace rt {
namespace {

struct CxHit {
    float t;
    F3 p, n;
    uint32_t ff;
};

// glam Mat4 (column-major) products in the reference's association order.
__device__ __forceinline__ F3 xf_point(const float* m, F3 v)
{
    return f3(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12], ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]);
}
__device__ __forceinline__ F3 xf_vector(const float* m, F3 v)
{
    return f3((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z, (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
// transform_vector3 of the transposed inverse (transform.rs:39-43)
__device__ __forceinline__ F3 xf_normal(const float* minv, F3 n)
{
    return f3((minv[0] * n.x + minv[1] * n.y) + minv[2] * n.z, (minv[4] * n.x + minv[5] * n.y) + minv[6] * n.z,
              (minv[8] * n.x + minv[9] * n.y) + minv[10] * n.z);
}

__device__ __forceinline__ float comp(F3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

__device__ __forceinline__ void cx_push_quadric(CxHit* out, int& n, float t, F3 o, F3 d, F3 outward)
{
    bool front = dot(d, outward) < 0.0f;
    out[n].t = t;
    out[n].p = o + t * d;
    out[n].n = front ? outward : -outward;
    out[n].ff = front ? 1u : 0u;
    ++n;
}

// One leaf's intersect_all(ray, 0.001, t_max): hits in the order the reference pushes them.
__device__ __noinline__ int cx_leaf(const CxNode nd, F3 o, F3 d, float t_max, CxHit* out)
{
    int n = 0;
    const F3 A = f3(nd.a.x, nd.a.y, nd.a.z), B = f3(nd.b.x, nd.b.y, nd.b.z);
    if (nd.kind == CX_SPHERE) {  // sphere.rs:12-74
        const float radius = nd.a.w;
        F3 oc = o - A;
        float a = dot(d, d);
        float half_b = dot(oc, d);
        float c = dot(oc, oc) - radius * radius;
        float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return 0;
        float sqrtd = __builtin_sqrtf(disc);
        float t1 = (-half_b - sqrtd) / a;
        if (t1 > kTMin && t1 < t_max) cx_push_quadric(out, n, t1, o, d, ((o + t1 * d) - A) / radius);
        if (disc > 1e-6f) {
            float t2 = (-half_b + sqrtd) / a;
            if (t2 > kTMin && t2 < t_max) cx_push_quadric(out, n, t2, o, d, ((o + t2 * d) - A) / radius);
        }
    } else if (nd.kind == CX_PLANE) {  // plane.rs:12-54
        float denom = dot(B, d);
        if (__builtin_fabsf(denom) < 1e-6f) return 0;
        float t = dot(A - o, B) / denom;
        if (t < kTMin || t_max < t) return 0;  // NaN passes, exactly like the reference's test
        bool front = dot(d, B) < 0.0f;
        out[0].t = t;
        out[0].p = o + t * d;
        out[0].n = front ? B : -B;
        out[0].ff = front ? 1u : 0u;
        n = 1;
    } else if (nd.kind == CX_BOX) {  // axis_aligned_box.rs:12-95
        float tmin = kTMin, tmax = t_max;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float inv_d = 1.0f / comp(d, i);
            float t0 = (comp(A, i) - comp(o, i)) * inv_d;
            float t1 = (comp(B, i) - comp(o, i)) * inv_d;
            if (inv_d < 0.0f) {
                float s = t0;
                t0 = t1;
                t1 = s;
            }
            tmin = __builtin_fmaxf(tmin, t0);
            tmax = __builtin_fminf(tmax, t1);
            if (tmax <= tmin) return 0;
        }
        const float ts[2] = {tmin, tmax};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            F3 p = o + ts[k] * d;
            F3 pa = p - A, pb = p - B;
            const float eps = 1e-4f;
            F3 nn = f3(0.f, 0.f, 0.f);
            if (__builtin_fabsf(pa.x) < eps) nn = f3(-1.f, 0.f, 0.f);
            else if (__builtin_fabsf(pb.x) < eps) nn = f3(1.f, 0.f, 0.f);
            else if (__builtin_fabsf(pa.y) < eps) nn = f3(0.f, -1.f, 0.f);
            else if (__builtin_fabsf(pb.y) < eps) nn = f3(0.f, 1.f, 0.f);
            else if (__builtin_fabsf(pa.z) < eps) nn = f3(0.f, 0.f, -1.f);
            else if (__builtin_fabsf(pb.z) < eps) nn = f3(0.f, 0.f, 1.f);
            if (k == 1) nn = -nn;  // exit normal points inward (axis_aligned_box.rs:56)
            out[k].t = ts[k];
            out[k].p = p;
            out[k].n = nn;
            out[k].ff = dot(d, nn) < 0.0f ? 1u : 0u;
        }
        n = 2;
    } else if (nd.kind == CX_INFCYL) {  // infinite_cylinder.rs:13-84 (A = axis_point, B = axis_dir, a.w = radius)
        const float radius = nd.a.w;
        F3 oc = o - A;
        float d_dot_v = dot(d, B);
        F3 d_perp = d - d_dot_v * B;
        float oc_dot_v = dot(oc, B);
        F3 oc_perp = oc - oc_dot_v * B;
        float a = dot(d_perp, d_perp);
        float b = 2.0f * dot(oc_perp, d_perp);
        float c = dot(oc_perp, oc_perp) - radius * radius;
        if (__builtin_fabsf(a) < 1e-6f) return 0;
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return 0;
        float sqrtd = __builtin_sqrtf(disc);
        const float ts[2] = {(-b - sqrtd) / (2.0f * a), (-b + sqrtd) / (2.0f * a)};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float t = ts[k];
            if (t > kTMin && t < t_max) {
                F3 p = o + t * d;
                F3 pma = p - A;
                float proj = dot(pma, B);
                F3 on_axis = A + proj * B;
                cx_push_quadric(out, n, t, o, d, normalize(p - on_axis));
            }
        }
    } else if (nd.kind == CX_INFCONE) {  // infinite_cone.rs:26-86 (A = vertex, B = axis_dir, a.w = cos^2)
        const float cs = nd.a.w;
        F3 co = o - A;
        float d_dot_v = dot(d, B);
        float co_dot_v = dot(co, B);
        float a = d_dot_v * d_dot_v - cs;
        float b = 2.0f * (d_dot_v * co_dot_v - dot(d, co) * cs);
        float c = co_dot_v * co_dot_v - dot(co, co) * cs;
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return 0;
        float sqrtd = __builtin_sqrtf(disc);
        const float ts[2] = {(-b - sqrtd) / (2.0f * a), (-b + sqrtd) / (2.0f * a)};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float t = ts[k];
            if (t > kTMin && t < t_max) {
                F3 p = o + t * d;
                F3 pv = p - A;
                float m = dot(pv, B);
                cx_push_quadric(out, n, t, o, d, normalize(m * B - pv * cs));
            }
        }
    }
    return n;
}

__device__ __forceinline__ bool cx_inside(uint32_t op, bool l, bool r)
{
    return op == CX_UNION ? (l || r) : (op == CX_INTERSECTION ? (l && r) : (l && !r));
}

// csg.rs:9-75 on two child lists; result in `out` (capacity checked on the host: nl + nr <= kCxMaxHits).
__device__ __noinline__ int cx_combine(uint32_t op, const CxHit* L, int nl, const CxHit* Rr, int nr, CxHit* out)
{
    CxHit all[kCxMaxHits];
    int na = 0;
    for (int i = 0; i < nl; ++i) all[na++] = L[i];
    for (int i = 0; i < nr; ++i) all[na++] = Rr[i];
    // stable insertion sort by t (sort_by(partial_cmp), csg.rs:23)
    for (int i = 1; i < na; ++i) {
        CxHit x = all[i];
        int j = i;
        while (j > 0 && all[j - 1].t > x.t) {
            all[j] = all[j - 1];
            --j;
        }
        all[j] = x;
    }
    bool in_left = false, in_right = false;
    int no = 0;
    for (int i = 0; i < na; ++i) {
        bool on_left = false;
        for (int k = 0; k < nl; ++k) on_left = on_left || (__builtin_fabsf(L[k].t - all[i].t) < 1e-6f);
        bool was = cx_inside(op, in_left, in_right);
        if (on_left) in_left = !in_left;
        else in_right = !in_right;
        bool is = cx_inside(op, in_left, in_right);
        if (was != is) {
            CxHit h = all[i];
            if (op == CX_DIFFERENCE && !on_left) {
                h.n = -h.n;
                h.ff = h.ff ? 0u : 1u;
            }
            out[no++] = h;
        }
    }
    return no;
}

__device__ __forceinline__ F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// One axis of a triangle's padded box: min / max over v0, v0+e1, v0+e2, grown by 1e-5 (|.|max + extent) + 1e-6.
__device__ __forceinline__ bool tri_box_holds(float a, float e1, float e2, float p)
{
    const float b = a + e1, c = a + e2;
    const float lo = __builtin_fminf(a, __builtin_fminf(b, c)), hi = __builtin_fmaxf(a, __builtin_fmaxf(b, c));
    const float mag = __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi));
    const float pad = 1e-5f * (mag + (hi - lo)) + 1e-6f;
    return p >= lo - pad && p <= hi + pad;
}

// Nearest triangle of a mesh in (0.001, +inf), ties to the lowest triangle index: Moller-Trumbore in
// strict f32, and the hit point o + t d must lie in the triangle's padded box.  That last rule is what makes the BVH
// exact: node boxes are unions of those triangle boxes (padded once more) and the slab test is widened past its own
// rounding, so a node is skipped only when none of its triangles can pass; the result equals the oracle's brute-force
// loop bit for bit.
// Widened slab test of one child box (see mesh_first_hit): entry distance, or +inf for a miss.  o_lo / o_hi are the
// origin moved 8 ulp up / down, which grows the box by what the rounding of o + t d can add on top of the box's own pad.
__device__ __forceinline__ float bvh_child_entry(float4 lo, float4 hi, F3 o_lo, F3 o_hi, F3 inv, float best)
{
    // fmin / fmax drop the NaN of 0 * inf (origin on a slab plane, zero direction component); NaN bounds (absent child)
    // leave tn = tf = NaN, which the final comparison rejects
    float t0 = (lo.x - o_lo.x) * inv.x, t1 = (hi.x - o_hi.x) * inv.x;
    float tn = __builtin_fminf(t0, t1), tf = __builtin_fmaxf(t0, t1);
    t0 = (lo.y - o_lo.y) * inv.y; t1 = (hi.y - o_hi.y) * inv.y;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    t0 = (lo.z - o_lo.z) * inv.z; t1 = (hi.z - o_hi.z) * inv.z;
    tn = __builtin_fmaxf(tn, __builtin_fminf(t0, t1)); tf = __builtin_fminf(tf, __builtin_fmaxf(t0, t1));
    tn = __builtin_fmaf(__builtin_fabsf(tn), -1e-6f, tn) - 1e-7f;  // a cull margin, not a reported value: fused is fine
    tf = __builtin_fmaf(__builtin_fabsf(tf), 1e-6f, tf) + 1e-7f;
    const bool hit = tn <= tf && tf >= 0.0f && tn <= best;
    return hit ? tn : __builtin_inff();
}

__device__ __forceinline__ void bvh_order(float& ta, uint32_t& ca, float& tb, uint32_t& cb)
{
    const bool swap = tb < ta;
    const float tl = swap ? tb : ta, th = swap ? ta : tb;
    const uint32_t cl = swap ? cb : ca, ch = swap ? ca : cb;
    ta = tl; tb = th; ca = cl; cb = ch;
}

// LDS copy of the first nodes of sc.bvh_nodes (the tree tops: the builder numbers nodes breadth first), as float4[8] per node.
typedef float V4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const V4f LdsF4;
__device__ __forceinline__ float4 lds_f4(LdsF4* q) { const V4f v = *q; return make_float4(v.x, v.y, v.z, v.w); }
typedef unsigned V2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) V2u LdsU2;
struct BvhLds {
    LdsF4* nodes;
    uint32_t count;
    LdsU2* stack;     // this lane's column of the block's short traversal stack: entry k at stack[k * kBvhLdsStackStride] (nullptr: none)
    uint32_t n_stack;  // its entries
};
#ifndef RT_BVH_LDS_STACK
#define RT_BVH_LDS_STACK 6  // entries of the traversal stack that live in LDS (the bottom ones; deeper entries go to private memory)
#endif
#ifndef RT_BVH_LDS_STACK_STREAM
#define RT_BVH_LDS_STACK_STREAM 3  // the streamed kernel's static LDS is 16 KB larger: a shorter one keeps two blocks per CU
#endif
constexpr unsigned kBvhLdsStack = RT_BVH_LDS_STACK;
constexpr unsigned kBvhLdsStackStream = RT_BVH_LDS_STACK_STREAM;
constexpr unsigned kBvhLdsStackStride = 512;  // threads of the block that owns the array

// mesh_first_hit is inlined into its two callers, cx_first_hit stays a call (measured on C3, 256 spp: both calls 110.5 ms, the
// traversal inlined 108.6 ms, both inlined 117.0 ms: the kernel runs at the 128-VGPR cap and what is live across a call spills)
#ifndef RT_MESH_INLINE
#define RT_MESH_INLINE __forceinline__
#endif
#ifndef RT_CXHIT_INLINE
#define RT_CXHIT_INLINE __noinline__
#endif
__device__ RT_MESH_INLINE bool mesh_first_hit(const DevScene& sc, BvhLds top, uint32_t root, F3 o, F3 d, float t_limit, CxHit& out)
{
    const float inf = __builtin_inff();
    // t_limit = t of the closest hit the scene scan has so far (transform.rs passes t_max through unchanged, and t is
    // the same number in local and world space): a triangle beyond it cannot win, one exactly at it still can (key tie)
    float best = t_limit;
    uint32_t best_id = 0xffffffffu, best_k = 0;
    const F3 inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const F3 ro = f3(4.8e-7f * __builtin_fabsf(o.x), 4.8e-7f * __builtin_fabsf(o.y), 4.8e-7f * __builtin_fabsf(o.z));
    const F3 o_lo = o + ro, o_hi = o - ro;
    // the tables are global memory: said so, the loads are global_load (vmcnt only) instead of flat_load, whose waits also stall on LDS
    typedef __attribute__((address_space(1))) const V4f GlbF4;
    GlbF4* const g_nodes = (GlbF4*)sc.bvh_nodes;  // 8 vectors per node: lo[4], hi[4]
    GlbF4* const g_tris = (GlbF4*)sc.tris;        // 3 per triangle: v0, e1, e2
    auto glb_f4 = [](GlbF4* q) { const V4f v = *q; return make_float4(v.x, v.y, v.z, v.w); };
    auto load_tri = [&](uint32_t k) { TriDev t; t.v0 = glb_f4(g_tris + 3u * k); t.e1 = glb_f4(g_tris + 3u * k + 1u); t.e2 = glb_f4(g_tris + 3u * k + 2u); return t; };
    // The traversal stack: the first kBvhLdsStack entries in the block's LDS (most pushes and pops happen there: a 4-wide tree over 10^5
    // triangles is nine levels deep), the rest in private memory.  All in private memory it was most of the 2.4 TB/s the C3 frame moved.
    constexpr uint32_t kDone = 0x7fffffffu;  // not a node index (node counts stay far below 2^31)
    uint32_t stack[kBvhStack];
    float stack_t[kBvhStack];
    int sp = 0;
    const int n_lds = top.stack ? (int)top.n_stack : 0;
    auto push = [&](uint32_t c, float t) {
        if (sp < kBvhStack) {
            if (sp < n_lds) top.stack[sp * kBvhLdsStackStride] = V2u{c, __float_as_uint(t)};
            else { stack[sp - n_lds] = c; stack_t[sp - n_lds] = t; }
            ++sp;
        }
    };
    // pop, skipping entries the best hit found since has put out of reach
    auto pop = [&](float best_now) {
        while (sp > 0) {
            --sp;
            uint32_t c;
            float t;
            if (sp < n_lds) { const V2u e = top.stack[sp * kBvhLdsStackStride]; c = e.x; t = __uint_as_float(e.y); }
            else { c = stack[sp - n_lds]; t = stack_t[sp - n_lds]; }
            if (t <= best_now) return c;
        }
        return kDone;
    };
    uint32_t cur = root;
    // "while-while": every lane first descends to its next leaf, then the wave tests leaves together, so the expensive
    // triangle code runs with as many lanes as possible
    while (cur != kDone) {
        while (!(cur & kBvhLeaf) && cur != kDone) {
            BvhNode n;
            if (cur < top.count) {  // every ray passes through these: LDS answers in a fraction of an L1 round trip
                LdsF4* q = top.nodes + cur * 8u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    n.lo[i] = lds_f4(q + i);
                    n.hi[i] = lds_f4(q + 4 + i);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    n.lo[i] = glb_f4(g_nodes + 8u * cur + i);
                    n.hi[i] = glb_f4(g_nodes + 8u * cur + 4 + i);
                }
            }
            float t0 = bvh_child_entry(n.lo[0], n.hi[0], o_lo, o_hi, inv, best), t1 = bvh_child_entry(n.lo[1], n.hi[1], o_lo, o_hi, inv, best);
            float t2 = bvh_child_entry(n.lo[2], n.hi[2], o_lo, o_hi, inv, best), t3 = bvh_child_entry(n.lo[3], n.hi[3], o_lo, o_hi, inv, best);
            uint32_t c0 = __float_as_uint(n.lo[0].w), c1 = __float_as_uint(n.lo[1].w), c2 = __float_as_uint(n.lo[2].w), c3 = __float_as_uint(n.lo[3].w);
            // sort the four (entry, child) pairs by entry distance (misses = +inf go last); order never changes the
            // result, it only lets the nearest hit cull the rest
            bvh_order(t0, c0, t1, c1);
            bvh_order(t2, c2, t3, c3);
            bvh_order(t0, c0, t2, c2);
            bvh_order(t1, c1, t3, c3);
            bvh_order(t1, c1, t2, c2);
            if (t3 < inf) push(c3, t3);
            if (t2 < inf) push(c2, t2);
            if (t1 < inf) push(c1, t1);
            cur = t0 < inf ? c0 : pop(best);
        }
        if (cur == kDone) break;
        const uint32_t first = (cur & 0x7fffffffu) >> 3, cnt = cur & 7u;
        for (uint32_t k = first; k < first + cnt; ++k) {
            const TriDev tr = load_tri(k);
            const F3 v0 = f3(tr.v0.x, tr.v0.y, tr.v0.z), e1 = f3(tr.e1.x, tr.e1.y, tr.e1.z), e2 = f3(tr.e2.x, tr.e2.y, tr.e2.z);
            const F3 pvec = cross(d, e2);
            const float det = dot(e1, pvec);
            if (__builtin_fabsf(det) < 1e-12f) continue;
            const float idet = 1.0f / det;
            const F3 tvec = o - v0;
            const float u = dot(tvec, pvec) * idet;
            if (u < 0.0f || u > 1.0f) continue;
            const F3 qvec = cross(tvec, e1);
            const float v = dot(d, qvec) * idet;
            if (v < 0.0f || u + v > 1.0f) continue;
            const float t = dot(e2, qvec) * idet;
            if (!(t > kTMin && t < inf)) continue;
            const F3 p = o + t * d;
            if (!(tri_box_holds(v0.x, e1.x, e2.x, p.x) && tri_box_holds(v0.y, e1.y, e2.y, p.y) && tri_box_holds(v0.z, e1.z, e2.z, p.z))) continue;
            const uint32_t id = __float_as_uint(tr.v0.w);
            if (t < best || (t == best && id < best_id)) {
                best = t;
                best_id = id;
                best_k = k;
            }
        }
        cur = pop(best);
    }
    if (best_id == 0xffffffffu) return false;
    const TriDev tr = load_tri(best_k);
    const F3 nrm = cross(f3(tr.e1.x, tr.e1.y, tr.e1.z), f3(tr.e2.x, tr.e2.y, tr.e2.z));
    const bool front = dot(d, nrm) < 0.0f;
    out.t = best;
    out.p = o + best * d;
    out.n = front ? nrm : -nrm;
    out.ff = front ? 1u : 0u;
    return true;
}

// hits.first() of Transform(object).intersect_all(ray, 0.001, t_max), still in the object's local space (t is the same
// number in both spaces, transform.rs:31).  `near` is set when a leaf surface lies in (w_lo, w_hi].
__device__ RT_CXHIT_INLINE bool cx_first_hit(const DevScene& sc, BvhLds top, const CxObject& ob, F3 o, F3 d, float t_max, float w_lo, float w_hi,
                                          bool& near, CxHit& first)
{
    const F3 lo = xf_point(ob.minv, o);
    const F3 ld = xf_vector(ob.minv, d);
    // a mesh returns its nearest triangle, which does not depend on t_max below that triangle (no CSG inside a mesh)
    if (ob.mesh_root != kNoMesh) return mesh_first_hit(sc, top, ob.mesh_root, lo, ld, t_max, first);
    CxHit st[kCxStack][kCxMaxHits];
    int cnt[kCxStack];
    int sp = 0;
    for (uint32_t k = 0; k < ob.n_nodes; ++k) {
        const CxNode nd = sc.cx_nodes[ob.first_node + k];
        if (nd.kind < CX_UNION) {
            cnt[sp] = cx_leaf(nd, lo, ld, t_max, st[sp]);
            for (int i = 0; i < cnt[sp]; ++i) near = near || (st[sp][i].t > w_lo && st[sp][i].t <= w_hi);
            ++sp;
        } else {
            CxHit tmp[kCxMaxHits];
            int n = cx_combine(nd.kind, st[sp - 2], cnt[sp - 2], st[sp - 1], cnt[sp - 1], tmp);
            sp -= 2;
            for (int i = 0; i < n; ++i) st[sp][i] = tmp[i];
            cnt[sp] = n;
            ++sp;
        }
    }
    if (cnt[0] == 0) return false;
    first = st[0][0];
    return true;
}

// scene.rs:50-62 replayed in list order for objects 0 .. obj_end-1: the closest t the reference holds when it reaches
// object obj_end.  O(objects) per lane; only runs for the rare ray described at cx_min_key.
__device__ __noinline__ float cx_prefix_closest(const DevScene& sc, BvhLds top, F3 o, F3 d, int32_t obj_end)
{
    float best = __builtin_inff();
    const float a = dot(d, d);
    for (int32_t j = 0; j < obj_end; ++j) {
        const int32_t c = sc.obj_cx[j];
        if (c >= 0) {
            CxHit h;
            bool near = false;
            if (cx_first_hit(sc, top, sc.cx_objs[c], o, d, best, 0.f, 0.f, near, h) && h.t < best) best = h.t;
        } else if (sc.obj_shape[j] == RT_SHAPE_SPHERE) {
            const float4 g = sc.obj_geom[j];
            float t;
            if (sphere_first_t(o, d, a, f3(g.x, g.y, g.z), g.w, t) && t < best) best = t;
        } else if (sc.obj_shape[j] == RT_SHAPE_BOX) {
            const float4 g = sc.obj_geom[j], hb = sc.obj_aux[j];
            float t;
            if (box_first_t(o, d, f3(g.x, g.y, g.z), f3(hb.x, hb.y, hb.z), t) && t < best) best = t;
        } else {
            const float4 g = sc.obj_geom[j], n4 = sc.obj_aux[j];
            const F3 n = f3(n4.x, n4.y, n4.z), lo = o - f3(g.x, g.y, g.z);
            const float denom = dot(n, d);
            if (__builtin_fabsf(denom) < 1e-6f) continue;
            const float t = dot(f3(0.0f, 0.0f, 0.0f) - lo, n) / denom;
            if (t >= kTMin && t < best) best = t;
        }
    }
    return best;
}

// Merge every complex object into the running closest-hit key; remember the winner's local hit record.
//
// The reference evaluates object K with t_max = tau_K, the closest t among objects 0 .. K-1 (scene.rs:54).  `key` holds
// the closest of ALL spheres and planes plus the complex objects before K, (t_best, j_best):
//   * j_best < K (or no hit yet): tau_K == t_best exactly, K is evaluated with it.
//   * j_best > K: tau_K >= t_best is not known.  K is evaluated with t_max = +inf.  A truncated and an untruncated
//     evaluation agree on every result hit below tau_K - D * 1e-6 (D = tree depth; hits are processed in t order and
//     the attribution of csg.rs:33 looks 1e-6 ahead per level), so a first hit at or below w_lo = t_best - kCxWindow
//     is the reference's.  Otherwise K can only win with a hit in (w_lo, t_best], which must be a leaf surface of K:
//     if there is none K loses in the reference too; if there is one, tau_K is computed by replaying the list
//     (cx_prefix_closest) and K is evaluated again with it.
constexpr float kCxWindow = 1e-4f;
__device__ __forceinline__ unsigned long long cx_min_key(const DevScene& sc, BvhLds top, F3 o, F3 d, unsigned long long key, CxHit& best, int& best_cx)
{
    const float inf = __builtin_inff();
    for (uint32_t k = 0; k < sc.n_cx; ++k) {
        const CxObject& ob = sc.cx_objs[k];
        CxHit h;
        const uint32_t tbits = (uint32_t)(key >> 32);  // kNoHit has all bits set: no limit yet
        const float t_best = tbits > 0x7f800000u ? inf : __uint_as_float(tbits);
        const int32_t j_best = (int32_t)(uint32_t)(key & 0xffffffffull);
        bool hit, near = false;
        if (key == kNoHit || j_best < ob.obj_id || ob.mesh_root != kNoMesh) {
            hit = cx_first_hit(sc, top, ob, o, d, t_best, 0.f, 0.f, near, h);
        } else {
            const float w_lo = __builtin_fmaf(t_best, -0x1p-21f, t_best) - kCxWindow;  // a window edge, not a reported value
            hit = cx_first_hit(sc, top, ob, o, d, inf, w_lo, t_best, near, h);
            if (!(hit && h.t <= w_lo)) {
                hit = false;
                if (near) hit = cx_first_hit(sc, top, ob, o, d, cx_prefix_closest(sc, top, o, d, ob.obj_id), 0.f, 0.f, near, h);
            }
        }
        if (hit && h.t < inf) {  // scene.rs:56; NaN / inf never win
            unsigned long long kk = pack_key(h.t, ob.obj_id);
            if (kk < key) {
                key = kk;
                best = h;
                best_cx = (int)k;
            }
        }
    }
    return key;
}

// World-space record of a complex winner (transform.rs:33-46).
__device__ __forceinline__ void cx_to_world(const CxObject& ob, const CxHit& h, F3& point, F3& normal)
{
    point = xf_point(ob.m, h.p);
    normal = normalize(xf_normal(ob.minv, h.n));
}

}  // namespace
}  // namespace rt
