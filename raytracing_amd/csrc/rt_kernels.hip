// rt_kernels.hip — gfx950 kernels for the ray-path hot loop (Scene::simulate_rays, scene.rs:39-106).
//
// One persistent kernel does everything between "a ray exists" and "its polyline / pixel is final":
// primary-ray generation (scene_config.rs:60-108), the closest-hit scan over the object list
// (scene.rs:50-62 + transform.rs:20-51 + sphere.rs:12-74 + plane.rs:12-54), the material switch
// (scene.rs:68-96 with reflect/refract scene.rs:9-27), re-origin / miss terminator (scene.rs:97-101) and
// the per-cell sample accumulation.
//
// Mapping to CDNA4 (measured on MI355X, tools/ubench/valu_rate.hip):
//   * one lane owns R independent rays ("slots"); a wave scans the scene for 64*R rays at once;
//   * the sphere list lives in LDS as float4 {p, K} and is read with wave-uniform (broadcast)
//     ds_read_b128 into VGPRs — VALU ops with an SGPR operand issue at half rate on gfx950, so the scene is
//     deliberately NOT kept in scalar registers;
//   * the per-(ray, sphere) work is a conservative 8-instruction FMA filter; the few pairs that survive it
//     are compacted through a per-wave LDS queue and confirmed 64 at a time with the reference's exact
//     operation sequence, then merged with a 64-bit LDS atomic min on (t bits, object index), which is
//     exactly the reference's "strict <, lowest index wins" scan order (scene.rs:56);
//   * lanes that finish a path start the next sample / fetch the next cell immediately (wave-aggregated
//     atomic on a work counter), so a wave never carries dead lanes through a scan except at the very end.
//
// Everything outside the filter is compiled without FP contraction (-ffp-contract=off) so that it rounds
// exactly like the reference's scalar f32 code; the filter uses explicit fmaf and is protected by the
// error bound derived in DESIGN.md ("Conservative filter").
#include "rt_internal.h"

#include <cstdlib>

namespace rt {

namespace {

constexpr float kTMin = 0.001f;       // scene.rs:54
constexpr float kReorigin = 0.001f;   // scene.rs:97
constexpr float kCullKappaU = 7.62939453125e-06f;  // 128 * 2^-24 (needs >= 94 u)
constexpr unsigned kQueueCap = 256;   // per-wave candidate ring (entries), power of two
constexpr unsigned long long kNoHit = ~0ull;

struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 normalize(F3 a)
{
    float len = __builtin_sqrtf(dot(a, a));
    float rcp = 1.0f / len;
    return a * rcp;
}

// scene.rs:9-11
__device__ __forceinline__ F3 reflect(F3 i, F3 n) { return i - 2.0f * dot(i, n) * n; }
// scene.rs:14-27
__device__ __forceinline__ bool refract(F3 i, F3 n, float ratio, F3& out)
{
    float cos_theta = __builtin_fminf(dot(-i, n), 1.0f);
    float sin2 = 1.0f - cos_theta * cos_theta;
    if (ratio * ratio * sin2 > 1.0f) return false;
    F3 perp = ratio * (i + cos_theta * n);
    F3 par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dot(perp, perp))) * n;
    out = normalize(perp + par);
    return true;
}

// Philox4x32-10, counter (path lo, path hi, bounce, tag), key = seed; rand 0.8 Standard f32 mapping.
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t path_id, uint32_t bounce)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)path_id, c1 = (uint32_t)(path_id >> 32), c2 = bounce, c3 = 0x52415953u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ unsigned mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// LDS traffic between lanes of one wave: order the DS instructions, nothing leaves the CU.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long pack_key(float t, int32_t obj) { return ((unsigned long long)__float_as_uint(t) << 32) | (uint32_t)obj; }

// First element of the reference's hit list for Transform(Sphere) with rotation 0, as a (t, found) pair:
// transform.rs:22-23 (local origin = o - position exactly, local direction = d), sphere.rs:13-66.
__device__ __forceinline__ bool sphere_first_t(F3 o, F3 d, float a, F3 pos, float radius, float& t_out)
{
    F3 lo = o - pos;
    float half_b = dot(lo, d);
    float c = dot(lo, lo) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float sqrtd = __builtin_sqrtf(disc);
    float t1 = (-half_b - sqrtd) / a;
    if (t1 > kTMin && t1 < __builtin_inff()) {
        t_out = t1;
        return true;
    }
    if (disc > 1e-6f) {
        float t2 = (-half_b + sqrtd) / a;
        if (t2 > kTMin && t2 < __builtin_inff()) {
            t_out = t2;
            return true;
        }
    }
    return false;
}

template <int R>
struct WaveLds {
    float stash[7][R * 64];            // o.xyz, d.xyz, a per slot
    unsigned long long best[R * 64];   // (t bits << 32 | obj id), kNoHit when empty
    unsigned queue[kQueueCap];         // (sphere index << 8) | slot
};

// Confirm up to 64 queued (slot, sphere) pairs with the exact operation sequence.
template <int R>
__device__ __forceinline__ void confirm_batch(WaveLds<R>& w, const DevScene& sc, unsigned head, unsigned count, unsigned lane)
{
    if (lane < count) {
        unsigned e = w.queue[(head + lane) & (kQueueCap - 1)];
        unsigned slot = e & 255u, j = e >> 8;
        if (j < sc.n_sph) {
            F3 o = f3(w.stash[0][slot], w.stash[1][slot], w.stash[2][slot]);
            F3 d = f3(w.stash[3][slot], w.stash[4][slot], w.stash[5][slot]);
            float a = w.stash[6][slot];
            float4 s = sc.sph_exact[j];
            float t;
            if (sphere_first_t(o, d, a, f3(s.x, s.y, s.z), s.w, t)) atomicMin(&w.best[slot], pack_key(t, sc.sph_obj[j]));
        }
    }
}

template <int R>
__device__ __forceinline__ void push_candidates(WaveLds<R>& w, const DevScene& sc, unsigned long long mask, unsigned slot_base,
                                                unsigned j, unsigned& head, unsigned& tail, unsigned lane)
{
    unsigned pos = tail + mbcnt64(mask);
    if ((mask >> lane) & 1ull) w.queue[pos & (kQueueCap - 1)] = (j << 8) | (slot_base + lane);
    tail += (unsigned)__builtin_popcountll(mask);
    if (tail - head >= 64u) {
        wave_lds_sync();
        confirm_batch<R>(w, sc, head, 64u, lane);
        head += 64u;
        wave_lds_sync();
    }
}

struct Slot {
    F3 o, d;
    float ior;
    unsigned long long accx, accy, accz;  // fixed-point (2^-24) sums of the unit's path values
    uint32_t m;           // work item (cell / ray)
    uint32_t sample;      // current sample index
    uint32_t sample_end;  // one past the unit's last sample
    uint32_t bounce;
    uint32_t segs;    // segments of the current cell
    uint32_t np;      // polyline vertices written
    bool live;        // carries a ray through the next scan
};

__device__ __forceinline__ void primary_ray(const TraceParams& p, uint32_t m, F3& o, F3& d, float& ior)
{
    if (p.rays) {
        const RtRay r = p.rays[m];
        o = f3(r.origin[0], r.origin[1], r.origin[2]);
        d = f3(r.direction[0], r.direction[1], r.direction[2]);
        ior = r.current_ior;
        return;
    }
    const DevGen& g = p.gen;
    uint64_t cell = p.cell_begin + (uint64_t)m * p.cell_step;
    uint32_t i = (uint32_t)(cell / g.count_v), j = (uint32_t)(cell % g.count_v);
    F3 us = f3(g.u_step[0], g.u_step[1], g.u_step[2]), vs = f3(g.v_step[0], g.v_step[1], g.v_step[2]);
    F3 a = f3(g.a[0], g.a[1], g.a[2]);
    ior = g.current_ior;
    if (g.kind == RT_GEN_PARALLEL_GRID) {
        o = (a + ((float)i * us)) + ((float)j * vs);
        d = f3(g.dir_n[0], g.dir_n[1], g.dir_n[2]);
    } else {
        F3 corner = f3(g.corner[0], g.corner[1], g.corner[2]);
        F3 target = (corner + ((float)i * us)) + ((float)j * vs);
        o = a;
        d = normalize(target - a);
    }
}

// DESIGN.md "path value" (no reference counterpart).
__device__ __forceinline__ F3 sky(F3 d)
{
    float w = 0.5f * (d.y + 1.0f);
    float k = 1.0f - w;
    return f3(k + w * 0.5f, k + w * 0.7f, k + w);
}

template <int R, int MODE>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel(const TraceParams p)
{
    __shared__ float4 tile[kTileSpheres];
    __shared__ WaveLds<R> wl[kBlockThreads / 64];

    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x >> 6;
    WaveLds<R>& w = wl[wave];
    const DevScene& sc = p.sc;
    const unsigned n_tiles = (sc.n_sph_pad + kTileSpheres - 1) / kTileSpheres;
    const bool multi = n_tiles > 1;

    if (!multi) {
        for (unsigned k = threadIdx.x; k < sc.n_sph_pad; k += kBlockThreads) tile[k] = sc.sph_cull[k];
        __syncthreads();
    }

    Slot s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r].live = false;
        s[r].m = 0xffffffffu;  // no cell
        s[r].sample = 0;
        s[r].bounce = 0;
        s[r].segs = 0;
        s[r].np = 0;
        s[r].accx = s[r].accy = s[r].accz = 0;
        s[r].sample_end = 0;
        s[r].o = f3(0, 0, 0);
        s[r].d = f3(0, 0, 0);
        s[r].ior = 1.0f;
    }
    unsigned long long my_segments = 0;
    bool exhausted = false;  // work counter ran past the last unit (wave-uniform)
    const unsigned long long total_units = (unsigned long long)p.cell_count * p.units_per_cell;
    bool wave_done = false;
    const uint32_t max_b = p.st.max_bounces;
    const uint64_t stride_pts = (uint64_t)max_b + 1;

    for (;;) {
        // ---- 1. refill: start the next sample, or fetch a new cell ----
        if (!wave_done) {
            bool any_live = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bool need_unit = !s[r].live && (s[r].m == 0xffffffffu);
                unsigned long long need = __ballot(need_unit);
                if (need && !exhausted) {
                    unsigned cnt = (unsigned)__builtin_popcountll(need);
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(p.next_unit, (unsigned long long)cnt);
                    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(base >> 32)) << 32) |
                           __builtin_amdgcn_readfirstlane((unsigned)base);
                    if (base + cnt >= total_units) exhausted = true;
                    if (need_unit) {
                        unsigned long long unit = base + mbcnt64(need);
                        if (unit < total_units) {
                            uint32_t m, k;
                            if (p.units_per_cell == 1) {
                                m = (uint32_t)unit;
                                k = 0;
                            } else {
                                m = (uint32_t)(unit / p.units_per_cell);
                                k = (uint32_t)(unit - (unsigned long long)m * p.units_per_cell);
                            }
                            s[r].m = m;
                            s[r].sample = k * p.unit_samples;
                            uint32_t e = s[r].sample + p.unit_samples;
                            s[r].sample_end = e < p.spp ? e : p.spp;
                            s[r].segs = 0;
                            s[r].accx = s[r].accy = s[r].accz = 0;
                        }
                    }
                }
                if (!s[r].live && s[r].m != 0xffffffffu) {
                    // start sample s[r].sample of cell s[r].m
                    primary_ray(p, s[r].m, s[r].o, s[r].d, s[r].ior);
                    s[r].bounce = 0;
                    s[r].np = 1;
                    s[r].live = true;
                    if (p.points) {
                        float* pt = p.points + (uint64_t)s[r].m * stride_pts * 3;
                        pt[0] = s[r].o.x; pt[1] = s[r].o.y; pt[2] = s[r].o.z;
                    }
                }
                any_live |= (__ballot(s[r].live) != 0ull);
            }
            if (!any_live) wave_done = true;
        }
        if (multi) {
            if (__syncthreads_and(wave_done ? 1 : 0)) break;
        } else if (wave_done) {
            break;
        }

        // ---- 2. per-segment constants ----
        float a_[R];
        F3 dh[R], pp[R];
        float nod[R], s0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const F3 o = s[r].o, d = s[r].d;
            a_[r] = dot(d, d);
            const unsigned slot = r * 64 + lane;
            w.stash[0][slot] = o.x; w.stash[1][slot] = o.y; w.stash[2][slot] = o.z;
            w.stash[3][slot] = d.x; w.stash[4][slot] = d.y; w.stash[5][slot] = d.z;
            w.stash[6][slot] = a_[r];
            w.best[slot] = kNoHit;
            float rs = 1.0f / __builtin_sqrtf(a_[r]);
            dh[r] = d * rs;
            nod[r] = -dot(o, dh[r]);
            float oo = dot(o, o);
            float margin = kCullKappaU * (oo + sc.c2max);
            s0[r] = s[r].live ? (oo - margin) : __builtin_inff();
            pp[r] = f3(-2.0f * o.x, -2.0f * o.y, -2.0f * o.z);
        }
        unsigned head = 0, tail = 0;
        wave_lds_sync();

        // ---- 3. scan the sphere list ----
        for (unsigned t_i = 0; t_i < n_tiles; ++t_i) {
            unsigned count = sc.n_sph_pad - t_i * kTileSpheres;
            if (count > (unsigned)kTileSpheres) count = kTileSpheres;
            if (multi) {
                __syncthreads();
                for (unsigned k = threadIdx.x; k < count; k += kBlockThreads) tile[k] = sc.sph_cull[t_i * kTileSpheres + k];
                __syncthreads();
                if (wave_done) continue;
            }
            const unsigned jbase = t_i * kTileSpheres;
            for (unsigned j = 0; j < count; j += kUnroll) {
                float4 sp[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) sp[u] = tile[j + u];
                unsigned long long mk[kUnroll][R];
                unsigned long long any = 0;
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        bool pass;
                        if (MODE == 0) {
                            float h = __builtin_fmaf(sp[u].x, dh[r].x, __builtin_fmaf(sp[u].y, dh[r].y, __builtin_fmaf(sp[u].z, dh[r].z, nod[r])));
                            float S = __builtin_fmaf(sp[u].x, pp[r].x, __builtin_fmaf(sp[u].y, pp[r].y, __builtin_fmaf(sp[u].z, pp[r].z, s0[r])));
                            float q = __builtin_fmaf(h, h, -S);
                            pass = q >= sp[u].w;
                        } else {
                            // exact discriminant as the filter (sphere.rs:13-18); padding/dead slots never pass
                            float4 se = sc.sph_exact[jbase + j + u];
                            F3 lo = s[r].o - f3(se.x, se.y, se.z);
                            float half_b = dot(lo, s[r].d);
                            float c = dot(lo, lo) - se.w * se.w;
                            float disc = half_b * half_b - a_[r] * c;
                            pass = s[r].live && (jbase + j + u < sc.n_sph) && !(disc < 0.0f);
                        }
                        mk[u][r] = __ballot(pass);
                        any |= mk[u][r];
                    }
                }
                if (any) {
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (mk[u][r]) push_candidates<R>(w, sc, mk[u][r], r * 64, jbase + j + u, head, tail, lane);
                        }
                    }
                }
            }
        }
        if (wave_done) continue;  // multi-tile mode: this wave only keeps the barriers company

        // ---- 4. drain the queue ----
        wave_lds_sync();
        if (tail != head) confirm_batch<R>(w, sc, head, tail - head, lane);
        wave_lds_sync();

        // ---- 5. planes, resolve, scatter ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!s[r].live) continue;
            const unsigned slot = r * 64 + lane;
            unsigned long long key = w.best[slot];
            const F3 o = s[r].o, d = s[r].d;
            for (unsigned k = 0; k < sc.n_pl; ++k) {
                // Transform(Plane) with rotation 0: transform.rs:22-23, plane.rs:14-27
                float4 pp4 = sc.pl_pos[k], pn4 = sc.pl_nrm[k];
                F3 n = f3(pn4.x, pn4.y, pn4.z);
                F3 lo = o - f3(pp4.x, pp4.y, pp4.z);
                float denom = dot(n, d);
                if (__builtin_fabsf(denom) < 1e-6f) continue;
                float t = dot(f3(0.0f, 0.0f, 0.0f) - lo, n) / denom;
                if (t >= kTMin && t < __builtin_inff()) {
                    unsigned long long kk = pack_key(t, sc.pl_obj[k]);
                    key = kk < key ? kk : key;
                }
            }
            ++s[r].segs;
            ++my_segments;
            const uint32_t b = s[r].bounce;
            const uint64_t path_id = p.rays ? (p.path_id_base + s[r].m)
                                            : ((p.cell_begin + (uint64_t)s[r].m * p.cell_step) * (uint64_t)p.spp + s[r].sample);
            bool path_done = false;
            F3 value = f3(0, 0, 0);
            RtBounceRecord rec;
            if (key != kNoHit) {
                const int32_t obj = (int32_t)(uint32_t)(key & 0xffffffffull);
                const float t = __uint_as_float((uint32_t)(key >> 32));
                const float4 g = sc.obj_geom[obj];
                const uint2 mat = sc.obj_mat[obj];
                const F3 pos = f3(g.x, g.y, g.z);
                const F3 lo = o - pos;
                const F3 pl = lo + t * d;  // local hit point (sphere.rs:28 / plane.rs:32)
                F3 nl;
                bool front;
                if (sc.obj_shape[obj] == RT_SHAPE_SPHERE) {
                    F3 outward = pl / g.w;  // (point - 0) / radius, sphere.rs:29
                    front = dot(d, outward) < 0.0f;
                    nl = front ? outward : -outward;
                } else {
                    float4 n4 = sc.obj_aux[obj];
                    F3 n = f3(n4.x, n4.y, n4.z);
                    front = dot(d, n) < 0.0f;
                    nl = front ? n : -n;
                }
                const F3 point = pl + pos;        // transform.rs:37 (M = T(position))
                const F3 normal = normalize(nl);  // transform.rs:39-43
                F3 dn = d;
                float ior = s[r].ior;
                if (mat.x == RT_MAT_MIRROR) {
                    dn = reflect(d, normal);
                } else if (mat.x == RT_MAT_GLASS) {
                    float n1 = ior;
                    float n2 = front ? __uint_as_float(mat.y) : 1.0f;
                    float ratio = n1 / n2;
                    F3 refr;
                    if (refract(d, normal, ratio, refr)) {
                        dn = refr;
                        ior = n2;
                    } else {
                        dn = reflect(d, normal);
                    }
                } else {
                    if (uniform01(p.seed, path_id, b) < __uint_as_float(mat.y)) dn = reflect(d, normal);
                }
                s[r].d = dn;
                s[r].ior = ior;
                s[r].o = point + dn * kReorigin;
                if (p.points) {
                    float* pt = p.points + ((uint64_t)s[r].m * stride_pts + s[r].np) * 3;
                    pt[0] = point.x; pt[1] = point.y; pt[2] = point.z;
                }
                ++s[r].np;
                if (p.records) {
                    rec.obj_id = obj; rec.t = t;
                    rec.point[0] = point.x; rec.point[1] = point.y; rec.point[2] = point.z;
                    rec.normal[0] = normal.x; rec.normal[1] = normal.y; rec.normal[2] = normal.z;
                    rec.front_face = front ? 1u : 0u;
                    rec.dir_out[0] = dn.x; rec.dir_out[1] = dn.y; rec.dir_out[2] = dn.z;
                    rec.ior_out = ior;
                    p.records[(uint64_t)s[r].m * max_b + b] = rec;
                }
                if (b == 0 && s[r].sample == 0) {
                    if (p.first_id) p.first_id[s[r].m] = obj;
                    if (p.first_t) p.first_t[s[r].m] = t;
                }
                s[r].bounce = b + 1;
                if (s[r].bounce == max_b) path_done = true;  // bounce budget exhausted: value stays 0
            } else {
                const F3 far = o + d * p.st.infinity_distance;  // scene.rs:99
                if (p.points) {
                    float* pt = p.points + ((uint64_t)s[r].m * stride_pts + s[r].np) * 3;
                    pt[0] = far.x; pt[1] = far.y; pt[2] = far.z;
                }
                ++s[r].np;
                if (p.records) {
                    rec.obj_id = -1; rec.t = __builtin_inff();
                    rec.point[0] = far.x; rec.point[1] = far.y; rec.point[2] = far.z;
                    rec.normal[0] = 0; rec.normal[1] = 0; rec.normal[2] = 0;
                    rec.front_face = 0;
                    rec.dir_out[0] = 0; rec.dir_out[1] = 0; rec.dir_out[2] = 0;
                    rec.ior_out = s[r].ior;
                    p.records[(uint64_t)s[r].m * max_b + b] = rec;
                }
                if (b == 0 && s[r].sample == 0) {
                    if (p.first_id) p.first_id[s[r].m] = -1;
                    if (p.first_t) p.first_t[s[r].m] = __builtin_inff();
                }
                value = sky(d);
                path_done = true;
            }
            if (path_done) {
                s[r].live = false;
                // fixed-point accumulation (DESIGN.md "path value"): order-independent, so units of one cell
                // may finish in any order on any lane / GPU and the sum is still bit-identical
                s[r].accx += (unsigned long long)__float2ll_rn(value.x * 16777216.0f);
                s[r].accy += (unsigned long long)__float2ll_rn(value.y * 16777216.0f);
                s[r].accz += (unsigned long long)__float2ll_rn(value.z * 16777216.0f);
                if (p.n_points) p.n_points[s[r].m] = s[r].np;
                ++s[r].sample;
                if (s[r].sample == s[r].sample_end) {
                    if (p.fb_q) {
                        unsigned long long* q = p.fb_q + (uint64_t)s[r].m * 3;
                        if (s[r].accx) atomicAdd(q + 0, s[r].accx);
                        if (s[r].accy) atomicAdd(q + 1, s[r].accy);
                        if (s[r].accz) atomicAdd(q + 2, s[r].accz);
                    }
                    if (p.seg_count) atomicAdd(p.seg_count + s[r].m, s[r].segs);
                    s[r].m = 0xffffffffu;
                }
            }
        }
    }

    // ---- total segment count: wave reduce, one atomic per wave ----
    if (p.total_segments) {
        unsigned long long v = my_segments;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(p.total_segments, v);
    }
}

}  // namespace

// fb = sum_q / (spp * 2^24), IEEE f64 division rounded once to f32
__global__ void finalize_kernel(const unsigned long long* __restrict__ q, float* __restrict__ fb, uint64_t n, double denom)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fb[i] = (float)((double)q[i] / denom);
}

template <class K>
static int blocks_per_cu(K kernel)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, kBlockThreads, 0) != hipSuccess || n < 1) n = 1;
    if (const char* e = getenv("RT_HIP_BLOCKS_PER_CU")) {  // tuning knob
        int v = atoi(e);
        if (v >= 1 && v <= 8) n = v;
    }
    return n;
}

hipError_t launch_trace(const TraceParams& p, int n_cu, hipStream_t stream)
{
    constexpr int R = 2;
    static const int occ0 = blocks_per_cu(trace_kernel<R, 0>);
    static const int occ1 = blocks_per_cu(trace_kernel<R, 1>);
    // persistent grid: exactly the resident capacity (every block pulls cells until the counter runs out),
    // never more blocks than the work can feed
    const uint64_t slots_per_block = (uint64_t)kBlockThreads * R;
    const uint64_t want = ((uint64_t)p.cell_count * p.units_per_cell + slots_per_block - 1) / slots_per_block;
    const uint64_t cap = (uint64_t)n_cu * (p.scan_mode == 0 ? occ0 : occ1);
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (blocks == 0) blocks = 1;
    if (p.scan_mode == 0)
        hipLaunchKernelGGL((trace_kernel<R, 0>), dim3(blocks), dim3(kBlockThreads), 0, stream, p);
    else
        hipLaunchKernelGGL((trace_kernel<R, 1>), dim3(blocks), dim3(kBlockThreads), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_finalize(const unsigned long long* fb_q, float* fb, uint64_t n_values, uint32_t spp, hipStream_t stream)
{
    const unsigned threads = 256;
    const unsigned blocks = (unsigned)((n_values + threads - 1) / threads);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(threads), 0, stream, fb_q, fb, n_values, (double)spp * 16777216.0);
    return hipGetLastError();
}

}  // namespace rt
