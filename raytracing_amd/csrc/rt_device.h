// rt_device.h — device-side helpers shared by the gfx950 kernels: strict-f32 vector math in the reference's
// operation order, scatter functions (scene.rs:9-27), the counter-based RNG, the per-wave candidate queue and
// the exact confirm stage.  Included by rt_kernels.hip and rt_pinhole.hip only.
#pragma once
#include "rt_internal.h"

namespace rt {
namespace {

constexpr float kTMin = 0.001f;       // scene.rs:54
constexpr float kReorigin = 0.001f;   // scene.rs:97
constexpr float kCullKappaU = 7.62939453125e-06f;  // 128 * 2^-24 (needs >= 94 u)
constexpr unsigned kQueueCap = 128;   // per-wave candidate ring (entries), power of two; at most 127 are ever outstanding
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
// 64-bit integer add to a location in LDS (ds_add_u64; wavefront scope is all the callers need).
__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v)
{
    typedef __attribute__((address_space(3))) unsigned long long LdsU64;
    __hip_atomic_fetch_add((LdsU64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
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

// First element of the reference's hit list for Transform(AxisAlignedBox(-half, half)) with rotation 0
// (transform.rs:22-23: local origin = o - position exactly, local direction = d; axis_aligned_box.rs:12-41): the slab
// test started from (t_min, +inf) and the entry hit at tmin.  A ray that starts inside "enters" at t_min itself, as in the
// reference.  (With the reference's finite t_max the entry is the same number or the box yields nothing because
// tmax <= tmin, i.e. it could not have won: DESIGN.md §4.2.)
__device__ __forceinline__ bool box_first_t(F3 o, F3 d, F3 pos, F3 half, float& t_out)
{
    const F3 lo = o - pos;
    float tmin = kTMin, tmax = __builtin_inff();
    const float lo_c[3] = {lo.x, lo.y, lo.z}, d_c[3] = {d.x, d.y, d.z}, h_c[3] = {half.x, half.y, half.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float inv_d = 1.0f / d_c[i];
        float t0 = (-h_c[i] - lo_c[i]) * inv_d;
        float t1 = (h_c[i] - lo_c[i]) * inv_d;
        if (inv_d < 0.0f) {
            const float s = t0;
            t0 = t1;
            t1 = s;
        }
        tmin = __builtin_fmaxf(tmin, t0);  // f32::max / f32::min: the non-NaN operand wins
        tmax = __builtin_fminf(tmax, t1);
        if (tmax <= tmin) return false;
    }
    t_out = tmin;
    return true;
}
// AxisAlignedBox::calculate_normal (axis_aligned_box.rs:70-95) at local point p: first face within 1e-4, else zero
__device__ __forceinline__ F3 box_face_normal(F3 p, F3 half)
{
    const float eps = 1e-4f;
    const F3 pa = p + half, pb = p - half;  // p - min, p - max with min = -half
    if (__builtin_fabsf(pa.x) < eps) return f3(-1.f, 0.f, 0.f);
    if (__builtin_fabsf(pb.x) < eps) return f3(1.f, 0.f, 0.f);
    if (__builtin_fabsf(pa.y) < eps) return f3(0.f, -1.f, 0.f);
    if (__builtin_fabsf(pb.y) < eps) return f3(0.f, 1.f, 0.f);
    if (__builtin_fabsf(pa.z) < eps) return f3(0.f, 0.f, -1.f);
    if (__builtin_fabsf(pb.z) < eps) return f3(0.f, 0.f, 1.f);
    return f3(0.f, 0.f, 0.f);
}

template <int R>
struct WaveLds {
    float stash[7][R * 64];            // o.xyz, d.xyz, a per slot
    unsigned long long best[R * 64];   // (t bits << 32 | obj id), kNoHit when empty
    unsigned queue[kQueueCap];         // (sphere index << 8) | slot
};

// Confirm up to 64 queued (slot, sphere) pairs with the exact operation sequence.  best[slot] collects (t bits << 32 | j):
// scan-path entries are in object order (sph_obj increases with j), so the smallest (t, j) is the smallest (t, object).
// sc.sph_exact may point into LDS here (the Projector kernel's grid mode keeps the table there).
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
            const float4 s = sc.sph_exact[j];
            float t;
            bool hit;
            if (sc.n_box) {
                const float4 hb = sc.sph_box[j];  // {half extents, 1} when entry j is the bounding sphere of a rotation-0 Box, else 0
                hit = hb.w != 0.0f ? (box_first_t(o, d, f3(s.x, s.y, s.z), f3(hb.x, hb.y, hb.z), t) && t < __builtin_inff())
                                   : sphere_first_t(o, d, a, f3(s.x, s.y, s.z), s.w, t);
            } else {
                hit = sphere_first_t(o, d, a, f3(s.x, s.y, s.z), s.w, t);
            }
            if (hit) atomicMin(&w.best[slot], pack_key(t, (int32_t)j));
        }
    }
}

// A ray whose direction is NaN in every component (what leaves a hit with a zero normal: a box entered from inside) fails the
// sphere test everywhere and "enters" EVERY rotation-0 box at t_min: f32::max / f32::min drop the NaN bounds of all three slabs
// (axis_aligned_box.rs:24-29).  All boxes tie at t_min, so the first one in object order wins: such a ray queues that single
// entry and takes no part in the scan (confirm_batch still runs the exact test on it).
__device__ __forceinline__ bool nan_ray(F3 d) { return d.x != d.x && d.y != d.y && d.z != d.z; }
template <int R>
__device__ __forceinline__ void push_candidates(WaveLds<R>& w, const DevScene& sc, unsigned long long mask, bool mine, unsigned slot_base,
                                                unsigned j, unsigned& head, unsigned& tail, unsigned lane);
template <int R>
__device__ __forceinline__ void push_nan_rays(WaveLds<R>& w, const DevScene& sc, bool is_nan, unsigned slot_base, unsigned& head, unsigned& tail,
                                              unsigned lane)
{
    const unsigned long long mask = __ballot(is_nan);
    if (mask && sc.n_box) push_candidates<R>(w, sc, mask, is_nan, slot_base, sc.first_box, head, tail, lane);
}

// best[slot] -> the (t, object) key the rest of the step works with; sj = the scan-path entry it came from (-1: none).
// need_obj = false (wave-uniform; no planes, no complex objects, no first-hit output): nothing reads the object index, it
// stays the entry index and the load is skipped.
__device__ __forceinline__ unsigned long long scan_key(const DevScene& sc, unsigned long long kb, bool need_obj, int& sj)
{
    sj = -1;
    if (kb == kNoHit) return kb;
    sj = (int)(uint32_t)kb;
    if (!need_obj) return kb;
    return (kb & 0xffffffff00000000ull) | (uint32_t)sc.sph_obj[sj];
}

template <int R>
__device__ __forceinline__ void push_candidates(WaveLds<R>& w, const DevScene& sc, unsigned long long mask, bool mine, unsigned slot_base,
                                                unsigned j, unsigned& head, unsigned& tail, unsigned lane)
{
    // `mine` is this lane's bit of `mask` (the compare result itself), so the branch below is a plain exec = mask
    unsigned pos = tail + mbcnt64(mask);
    if (mine) w.queue[pos & (kQueueCap - 1)] = (j << 8) | (slot_base + lane);
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

// Generator cell (i, j) of work item m: cell = i * count_v + j (scene_config.rs:74-75).  32-bit division whenever the
// cell index fits (a 64-bit one costs a few hundred instructions).
__device__ __forceinline__ void cell_indices(const TraceParams& p, uint32_t m, uint32_t& i, uint32_t& j)
{
    const uint64_t cell = p.cell_begin + (uint64_t)m * p.cell_step;
    const uint32_t cv = p.gen.count_v;
    if ((cell >> 32) == 0) {
        const uint32_t c32 = (uint32_t)cell;
        i = c32 / cv;
        j = c32 - i * cv;
    } else {
        i = (uint32_t)(cell / cv);
        j = (uint32_t)(cell % cv);
    }
}

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
    uint32_t i, j;
    cell_indices(p, m, i, j);
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


// ---------------------------------------------------------------------------------------------------
// The scan step: every live slot of the wave against `count` spheres of an LDS tile.
//   FLAVOR 0  general ray:   pass iff fma(h,h,-S) >= K,  h = p.dh + nod,  S = p.pp + s0        (8 VALU / test)
//   (the shared-origin filter |A' + i B' + j C'| >= L of Projector rays runs on per-wave cull lists: cull_group / scan_cull_list)
//   FLAVOR 2  exact filter:  pass iff the reference's own discriminant is not < 0 (sphere.rs:13-18)
// Survivors go to the wave's candidate queue and are confirmed with the exact sequence.
// ---------------------------------------------------------------------------------------------------
template <int R>
struct ScanConsts {
    F3 dh[R];      // unit direction
    float ci[R], cj[R], lt[R];  // shared-origin filter: cell indices i, j and a lower bound of |target - O|
    float nod[R];  // -(o . dh)
    F3 pp[R];      // -2 o
    float s0[R];   // |o|^2 - margin  (+inf for dead slots)
    F3 o[R], d[R]; // FLAVOR 2 only
    float a[R];
    bool live[R];
};

// One scan step: kUnroll spheres (already in registers) against every slot of the wave.
template <int R, int FLAVOR>
__device__ __forceinline__ void scan_step(const float4 (&sp)[kUnroll], unsigned jg, const ScanConsts<R>& c, WaveLds<R>& w,
                                          const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane)
{
    unsigned long long mk[kUnroll][R];
    bool ps[kUnroll][R];
    unsigned long long any = 0;
    static_assert(FLAVOR == 0 || FLAVOR == 2, "general filter or exact filter");
    {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bool pass;
                if (FLAVOR == 0) {
                    float h = __builtin_fmaf(sp[u].x, c.dh[r].x, __builtin_fmaf(sp[u].y, c.dh[r].y, __builtin_fmaf(sp[u].z, c.dh[r].z, c.nod[r])));
                    float S = __builtin_fmaf(sp[u].x, c.pp[r].x, __builtin_fmaf(sp[u].y, c.pp[r].y, __builtin_fmaf(sp[u].z, c.pp[r].z, c.s0[r])));
                    float q = __builtin_fmaf(h, h, -S);
                    // "not below", not "at or above": a ray with a NaN component passes.  The reference's box test reports a hit at
                    // t_min for such rays (f32::max / f32::min drop the NaN, axis_aligned_box.rs:24-29), its sphere test none; the
                    // confirm stage reproduces both.  Dead slots (s0 = +inf) and padding (K = +inf) still fail.
                    pass = !(q < sp[u].w);
                } else {
                    float4 se = sc.sph_exact[jg + u];
                    F3 lo = c.o[r] - f3(se.x, se.y, se.z);
                    float half_b = dot(lo, c.d[r]);
                    float cc = dot(lo, lo) - se.w * se.w;
                    float disc = half_b * half_b - c.a[r] * cc;
                    pass = c.live[r] && (jg + u < sc.n_sph) && !(disc < 0.0f);
                }
                ps[u][r] = pass;
                mk[u][r] = __ballot(pass);
                any |= mk[u][r];
            }
        }
    }
    if (any) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (mk[u][r]) push_candidates<R>(w, sc, mk[u][r], ps[u][r], r * 64, jg + u, head, tail, lane);
            }
        }
    }
}

// The scan over a whole sphere table.
//   PF = false: `tile` is the block's LDS table; every step reads kUnroll spheres with wave-uniform ds_read_b128.
//   PF = true : the table (count % 64 == 0) lives in global memory / L2 (scenes above kTileSpheres).  A wave-uniform
//               vector load would cost a full 1 KiB TA pass per 16 bytes, so instead each lane fetches ONE sphere
//               (64 spheres per coalesced load), the wave parks them in its own 2 x 1 KiB LDS staging buffer and then
//               broadcasts them exactly like the LDS mode; the next 64 are in flight while the current 64 are tested.
template <int R, int FLAVOR, bool PF>
__device__ __forceinline__ void scan_tile(const float4* tile, unsigned count, float4* stage, const ScanConsts<R>& c,
                                          WaveLds<R>& w, const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane)
{
    if constexpr (!PF) {
        for (unsigned j = 0; j < count; j += kUnroll) {
            float4 sp[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) sp[u] = tile[j + u];
            scan_step<R, FLAVOR>(sp, j, c, w, sc, head, tail, lane);
        }
    } else {
        if (count == 0) return;
        float4 nxt = tile[lane];
        unsigned buf = 0;
        for (unsigned base = 0; base < count; base += 64) {
            stage[buf * 64 + lane] = nxt;
            if (base + 64 < count) nxt = tile[base + 64 + lane];
            wave_lds_sync();
            for (unsigned j = 0; j < 64; j += kUnroll) {
                float4 sp[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) sp[u] = stage[buf * 64 + j + u];
                scan_step<R, FLAVOR>(sp, base + j, c, w, sc, head, tail, lane);
            }
            buf ^= 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wave-level cull of the shared-origin table.  The work units of one slot group are dealt to neighbouring
// lanes, so the 64 rays of a group cover a short run of generator cells, and they keep those cells for unit_samples steps.
// A sphere that no ray of the group's bounding box of cell indices can pass is of no interest in any of those steps; the few
// that remain are copied to a per-group LDS list once, and the primary scan of the following steps reads that list instead
// of the whole table.  The per-ray test on the list is the expression h = fma(i, B', fma(j, C', A')) against the ray's own
// threshold, so every ray sees exactly the candidates it would have seen in the full scan.
//
// Bound.  A ray passes when |h(i, j)| >= lt(i, j), h affine in the cell indices and lt = |v(i, j)| (1 - 2^-20), v = target - O
// affine too.  Both sides move across the box, nearly in step for a sphere close to the line of sight, so bounding them
// separately (max |h| against min lt: rounds 1 and 2 until here) keeps every sphere within ~10 sphere radii of an off-axis
// group.  Instead the cull bounds the difference: |v| is convex, so with x_c the box centre, L_c = |v(x_c)| and
// g = grad |v|(x_c) = (v_c . us, v_c . vs) / L_c,
//     |v(x)| >= L_c + g . (x - x_c)                      for every x,
//     +-h(x) - |v(x)| <= (+-h_c - L_c) + di |+-b - g_u| + dj |+-c - g_v|   over the box (half extents di, dj).
// If that is negative for both signs (by more than the rounding below) no ray of the group passes.  Rounding, u = 2^-24:
// the per-ray h and the centre value h_c are each within 2.01 u T of the real affine form, T = |a| + jmax |c| + imax |b|;
// the per-ray v is computed with five roundings per component (scene_config.rs:98-100), each below u m_k with m_k =
// |corner_k| + imax |us_k| + jmax |vs_k| + |O_k|, so lt(i, j) >= |v|(1 - 2^-19) - 5 u M with M = |m|; v_c, L_c, g are computed
// here with errors of the same size, which reach the bound through L_c (<= 8 u M) and through g times the half extents
// (W = di |us| + dj |vs| times 8 u (1 + M / L_c)); the dozen operations of the bound itself each round below u (T + M).
// Allowed: 12 u T + 32 u M + 2^-19 (L_c + W) + 8 u W (1 + M / L_c).  inf entries ("always") and NaNs survive every compare.
// ---------------------------------------------------------------------------------------------------
constexpr unsigned kCullCap = 40;  // list entries per slot group; a group that needs more falls back to the full table scan

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// The Projector as the cull needs it (wave-uniform): v(i, j) = (corner - O) + i us + j vs, and M (see above; it grows with the
// largest cell indices of the launch).
struct CullGeom {
    F3 c0, us, vs;
    float M, ul, vl;  // M as above; |us|, |vs| (all three rounded up on the host: make_dev_gen)
};
__device__ __forceinline__ CullGeom cull_geom(const DevGen& g)
{
    CullGeom q;
    q.c0 = f3(g.corner[0] - g.a[0], g.corner[1] - g.a[1], g.corner[2] - g.a[2]);
    q.us = f3(g.u_step[0], g.u_step[1], g.u_step[2]);
    q.vs = f3(g.v_step[0], g.v_step[1], g.v_step[2]);
    q.M = g.cull_M;
    q.ul = g.cull_ul;
    q.vl = g.cull_vl;
    return q;
}
// A member of the kernel's argument block (TraceParams is the first argument of every kernel here, so the block starts with
// it), read where it is needed rather than held in scalar registers for the whole kernel.  The optimiser would hoist such loads
// out of the main loop, and the hundred-odd scalars this kernel keeps live already spill; each newcomer pushed the grid's
// constants out of their registers (reloaded a hundred times per walk: 8 % of a C2 frame, measured twice).  The empty asm
// makes the address opaque, so the loads stay where they are written: two or three s_load per use.
#define RT_KERNARG_PTR(T, name, offset)                                                                                                    \
    const char __attribute__((address_space(4)))* name##_base = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(name##_base));                                                                                                  \
    const T __attribute__((address_space(4)))* name = reinterpret_cast<const T __attribute__((address_space(4)))*>(name##_base + (offset))
// one scalar or pointer member
template <typename T>
__device__ __forceinline__ T kernarg_value(size_t offset)
{
    RT_KERNARG_PTR(T, vp, offset);
    return *vp;
}
// (member by member: a whole-struct copy goes through private memory)
__device__ __forceinline__ DevGen kernarg_gen(size_t offset)
{
    RT_KERNARG_PTR(DevGen, gp, offset);
    DevGen g;
    g.kind = gp->kind;
    g.count_u = gp->count_u;
    g.count_v = gp->count_v;
    g.current_ior = gp->current_ior;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.a[k] = gp->a[k];
        g.corner[k] = gp->corner[k];
        g.u_step[k] = gp->u_step[k];
        g.v_step[k] = gp->v_step[k];
        g.dir_n[k] = gp->dir_n[k];
    }
    g.cull_M = gp->cull_M;
    g.cull_ul = gp->cull_ul;
    g.cull_vl = gp->cull_vl;
    return g;
}
__device__ __forceinline__ DevGrid kernarg_grid(size_t offset)
{
    RT_KERNARG_PTR(DevGrid, gp, offset);
    DevGrid g;
    g.cells = gp->cells;
    g.items = gp->items;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.lo[k] = gp->lo[k];
        g.hi[k] = gp->hi[k];
        g.dim[k] = gp->dim[k];
        g.centre[k] = gp->centre[k];
    }
    g.cell = gp->cell;
    g.inv_cell = gp->inv_cell;
    g.radius = gp->radius;
    g.origin_r2 = gp->origin_r2;
    g.slack_k = gp->slack_k;
    g.n_cells = gp->n_cells;
    g.n_items = gp->n_items;
    g.enabled = gp->enabled;
    return g;
}

__device__ __forceinline__ float sgpr(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// The bound for one box of cell indices (all members wave-uniform, held in scalar registers).
struct CullBox {
    float ic, jc, di, dj, imax, jmax, Lc, gu, gv, slop0;
};
__device__ __forceinline__ CullBox cull_box(const CullGeom& q, float imin, float imax, float jmin, float jmax)
{
    CullBox b;
    const float u = 0x1p-24f;
    b.imax = sgpr(imax);
    b.jmax = sgpr(jmax);
    // cell indices are integers below 2^24: sums, differences and halves are exact
    b.ic = sgpr(0.5f * (imin + imax));
    b.di = sgpr(0.5f * (imax - imin));
    b.jc = sgpr(0.5f * (jmin + jmax));
    b.dj = sgpr(0.5f * (jmax - jmin));
    const F3 vc = f3(__builtin_fmaf(b.ic, q.us.x, __builtin_fmaf(b.jc, q.vs.x, q.c0.x)), __builtin_fmaf(b.ic, q.us.y, __builtin_fmaf(b.jc, q.vs.y, q.c0.y)),
                     __builtin_fmaf(b.ic, q.us.z, __builtin_fmaf(b.jc, q.vs.z, q.c0.z)));
    b.Lc = sgpr(__builtin_sqrtf(dot(vc, vc)));
    b.gu = sgpr(dot(vc, q.us) / b.Lc);  // L_c = 0: NaN, and every sphere is kept
    b.gv = sgpr(dot(vc, q.vs) / b.Lc);
    const float W = b.di * q.ul + b.dj * q.vl;
    b.slop0 = sgpr(32.0f * u * q.M + 0x1p-19f * (b.Lc + W) + 8.0f * u * W * (1.0f + q.M / b.Lc));
    return b;
}
// May a ray of the box pass entry {a, b, c}?  ("not provably none": NaNs and inf entries pass)
__device__ __forceinline__ bool cull_pass(const CullBox& x, float a, float b, float c)
{
    const float hc = __builtin_fmaf(x.ic, b, __builtin_fmaf(x.jc, c, a));
    const float T = __builtin_fmaf(x.imax, __builtin_fabsf(b), __builtin_fmaf(x.jmax, __builtin_fabsf(c), __builtin_fabsf(a)));
    const float slop = __builtin_fmaf(T, 12.0f * 0x1p-24f, x.slop0);
    const float up = (hc - x.Lc) + __builtin_fmaf(x.di, __builtin_fabsf(b - x.gu), __builtin_fmaf(x.dj, __builtin_fabsf(c - x.gv), slop));
    const float um = (-hc - x.Lc) + __builtin_fmaf(x.di, __builtin_fabsf(b + x.gu), __builtin_fmaf(x.dj, __builtin_fabsf(c + x.gv), slop));
    return !(up < 0.0f && um < 0.0f);
}

// Tile lists (built per launch by pinhole_tiles_kernel): the same cull, done once for every run of `tile_cells` consecutive
// work items, so that a group looks at the hundred-odd spheres of its one or two tiles and not at the whole table.
// Record of tile t: tl[t * kTileRec] = number of spheres that passed, then the first kTileListCap of their indices, ascending.
constexpr unsigned kTileRec = 128;
constexpr unsigned kTileListCap = kTileRec - 1;

// table: {A', B', C', -} per sphere in global memory, n_pad entries.  Returns the number of surviving spheres;
// list[0 .. min(n, kCullCap)) = {A', B', C', sphere index bits}.  n > kCullCap: the list is incomplete, scan the table.
// Four 64-sphere rows per trip, their loads issued together: the loop is a chain of L2 round trips otherwise.
__device__ __forceinline__ unsigned cull_table(const float4* table, unsigned n_pad, const CullBox& x, float4* list, unsigned lane)
{
    unsigned n = 0;
    constexpr int kRows = 4;
    for (unsigned base = 0; base < n_pad; base += 64u * kRows) {
        float4 e[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const unsigned k = base + 64u * (unsigned)r + lane;
            e[r] = table[k < n_pad ? k : n_pad - 1u];
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const unsigned k = base + 64u * (unsigned)r + lane;
            const bool pass = k < n_pad && cull_pass(x, e[r].x, e[r].y, e[r].z);
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                const unsigned pos = n + mbcnt64(mask);
                if (pass && pos < kCullCap) list[pos] = make_float4(e[r].x, e[r].y, e[r].z, __uint_as_float(k));
                n += (unsigned)__builtin_popcountll(mask);
            }
        }
    }
    return n;
}

// The same over the lists of tiles t0 .. t1 (at most two).  false: a tile's list is incomplete, the caller scans the table.
// A sphere listed by both tiles is copied twice; its candidates are confirmed twice with the same key, which changes nothing.
__device__ __forceinline__ bool cull_tiles(const float4* table, const uint32_t* tl, unsigned t0, unsigned t1, const CullBox& x, float4* list, unsigned lane, unsigned& n_out)
{
    unsigned n = 0;
    for (unsigned t = t0; t <= t1; ++t) {
        const uint32_t* rec = tl + (size_t)t * kTileRec;
        const unsigned cnt = __builtin_amdgcn_readfirstlane(rec[0]);
        if (cnt > kTileListCap) return false;
        for (unsigned base = 0; base < cnt; base += 64u) {  // one row, two for a long list
            const unsigned k0 = base + lane < cnt ? rec[1u + base + lane] : 0u;
            const float4 e0 = table[k0];
            const bool p0 = base + lane < cnt && cull_pass(x, e0.x, e0.y, e0.z);
            const unsigned long long m0 = __ballot(p0);
            const unsigned pos0 = n + mbcnt64(m0);
            if (p0 && pos0 < kCullCap) list[pos0] = make_float4(e0.x, e0.y, e0.z, __uint_as_float(k0));
            n += (unsigned)__builtin_popcountll(m0);
        }
    }
    n_out = n;
    return true;
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// The cull of one slot group: ci / cj the lanes' cell indices, m their work items (for the tile lists; tl == nullptr: none).
__device__ __forceinline__ unsigned cull_group(const float4* table, unsigned n_pad, const uint32_t* tl, unsigned tile_shift, const CullGeom& q, float ci, float cj,
                                               uint32_t m, bool live, float4* list, unsigned lane)
{
    const float inf = __builtin_inff();
    const unsigned long long lmask = __ballot(live);
    if (!lmask) return 0;  // no live slot in this group
    float imin, imax, jmin, jmax;
    unsigned mmin, mmax;
    // A group takes its units in lane order (lane l of a batch holds unit base + l), so its live lanes are lanes 0 .. n - 1 and
    // their work items, hence their cells and the column index i, do not decrease with the lane: the extremes sit in the first and
    // the last live lane.  The row index j wraps where i changes; only then (and for any other lane pattern) the six wave
    // reductions are needed.  They were a third of the ~900 instructions a batch costs beyond its rays.
    const unsigned n_live = (unsigned)__builtin_popcountll(lmask);
    const bool dense = lmask == (n_live == 64u ? ~0ull : ((1ull << n_live) - 1ull));
    const unsigned last = n_live - 1u;
    const float i0 = sgpr(ci), i1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ci), last));
    if (dense && i0 == i1) {
        imin = imax = i0;
        jmin = sgpr(cj);
        jmax = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cj), last));
        mmin = __builtin_amdgcn_readfirstlane(m);
        mmax = __builtin_amdgcn_readlane(m, last);
        if (jmax < jmin) {  // a negative cell stride is not a thing (cell_step is unsigned), but cost nothing to allow
            const float t = jmin;
            jmin = jmax;
            jmax = t;
        }
    } else {
        imin = wave_min(live ? ci : inf);
        imax = wave_max(live ? ci : -inf);
        jmin = wave_min(live ? cj : inf);
        jmax = wave_max(live ? cj : -inf);
        mmin = wave_min_u32(live ? m : 0xffffffffu);
        mmax = wave_max_u32(live ? m : 0u);
    }
    const CullBox x = cull_box(q, imin, imax, jmin, jmax);
    if (tl) {
        const unsigned t0 = __builtin_amdgcn_readfirstlane(mmin) >> tile_shift;
        const unsigned t1 = __builtin_amdgcn_readfirstlane(mmax) >> tile_shift;
        unsigned n = 0;
        if (t1 - t0 <= 1u && cull_tiles(table, tl, t0, t1, x, list, lane, n)) return n;
    }
    return cull_table(table, n_pad, x, list, lane);
}

// The primary scan of slot group r over its cull list: same expression, same threshold as the full-table filter had
// ("not below": NaNs pass, as in the general filter).  Two entries per trip: the broadcast reads and the FMAs of a pair do
// not depend on each other.
template <int R>
__device__ __forceinline__ void scan_cull_list(const float4* list, unsigned n, int r, float ci, float cj, float lt, bool live, WaveLds<R>& w,
                                               const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane)
{
    for (unsigned k = 0; k < n; k += 2) {
        const float4 e0 = list[k], e1 = list[k + 1 < n ? k + 1 : k];
        const float h0 = __builtin_fmaf(ci, e0.y, __builtin_fmaf(cj, e0.z, e0.x)), h1 = __builtin_fmaf(ci, e1.y, __builtin_fmaf(cj, e1.z, e1.x));
        const bool p0 = live && !(__builtin_fabsf(h0) < lt), p1 = live && (k + 1 < n) && !(__builtin_fabsf(h1) < lt);
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        if (m0) push_candidates<R>(w, sc, m0, p0, (unsigned)r * 64u, __float_as_uint(e0.w), head, tail, lane);
        if (m1) push_candidates<R>(w, sc, m1, p1, (unsigned)r * 64u, __float_as_uint(e1.w), head, tail, lane);
    }
}

// Planes are few: exact test inline (Transform(Plane) with rotation 0: transform.rs:22-23, plane.rs:14-27).
__device__ __forceinline__ unsigned long long planes_min_key(const DevScene& sc, F3 o, F3 d, unsigned long long key)
{
    for (unsigned k = 0; k < sc.n_pl; ++k) {
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
    return key;
}

// Hit record of the winning object + material switch (scene.rs:63-97) for a ray (o, d, ior).
struct Shade {
    int32_t obj;
    float t;
    F3 point, normal;
    bool front;
    F3 dn;      // new direction
    float ior;  // new current_ior
    F3 o_next;  // scene.rs:97
};
// `pre` (optional): the winner came from the complex-object path, its world point / normal / front_face are given.
struct PreHit {
    bool valid;
    F3 point, normal;
    bool front;
};
__device__ __forceinline__ Shade shade_hit(const DevScene& sc, F3 o, F3 d, float ior, unsigned long long key, uint64_t seed,
                                           uint64_t path_id, uint32_t bounce, const PreHit& pre, int sj)
{
    Shade h;
    h.obj = (int32_t)(uint32_t)(key & 0xffffffffull);
    h.t = __uint_as_float((uint32_t)(key >> 32));
    uint2 mat;
    if (sj >= 0) {
        // the winner is scan-path entry sj: geometry and material by entry index (same values as the object-indexed arrays)
        mat = sc.sph_mat[sj];
        const float4 g = sc.sph_exact[sj];  // sphere: {p, r}; box: {p, bounding radius}
        const F3 pos = f3(g.x, g.y, g.z);
        const F3 lo = o - pos;
        const F3 pl = lo + h.t * d;  // local hit point (sphere.rs:28)
        F3 nl;
        float4 hb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sc.n_box) hb = sc.sph_box[sj];
        if (hb.w != 0.0f) {  // axis_aligned_box.rs:33-41: the entry normal is kept as computed, not turned against the ray
            nl = box_face_normal(pl, f3(hb.x, hb.y, hb.z));
            h.front = dot(d, nl) < 0.0f;
        } else {
            F3 outward = pl / g.w;  // (point - 0) / radius, sphere.rs:29
            h.front = dot(d, outward) < 0.0f;
            nl = h.front ? outward : -outward;
        }
        h.point = pl + pos;        // transform.rs:37 (M = T(position))
        h.normal = normalize(nl);  // transform.rs:39-43
    } else if (pre.valid) {
        mat = sc.obj_mat[h.obj];
        h.point = pre.point;
        h.normal = pre.normal;
        h.front = pre.front;
    } else {
        mat = sc.obj_mat[h.obj];
        const float4 g = sc.obj_geom[h.obj];
        const F3 pos = f3(g.x, g.y, g.z);
        const F3 lo = o - pos;
        const F3 pl = lo + h.t * d;  // local hit point (sphere.rs:28 / plane.rs:32)
        F3 nl;
        if (sc.obj_shape[h.obj] == RT_SHAPE_BOX) {  // axis_aligned_box.rs:33-41: the entry normal is kept as computed, not turned against the ray
            const float4 hb = sc.obj_aux[h.obj];
            nl = box_face_normal(pl, f3(hb.x, hb.y, hb.z));
            h.front = dot(d, nl) < 0.0f;
        } else if (sc.obj_shape[h.obj] == RT_SHAPE_SPHERE) {
            F3 outward = pl / g.w;  // (point - 0) / radius, sphere.rs:29
            h.front = dot(d, outward) < 0.0f;
            nl = h.front ? outward : -outward;
        } else {
            float4 n4 = sc.obj_aux[h.obj];
            F3 n = f3(n4.x, n4.y, n4.z);
            h.front = dot(d, n) < 0.0f;
            nl = h.front ? n : -n;
        }
        h.point = pl + pos;        // transform.rs:37 (M = T(position))
        h.normal = normalize(nl);  // transform.rs:39-43
    }
    h.dn = d;
    h.ior = ior;
    if (mat.x == RT_MAT_MIRROR) {
        h.dn = reflect(d, h.normal);
    } else if (mat.x == RT_MAT_GLASS) {
        float n1 = ior;
        float n2 = h.front ? __uint_as_float(mat.y) : 1.0f;
        float ratio = n1 / n2;
        F3 refr;
        if (refract(d, h.normal, ratio, refr)) {
            h.dn = refr;
            h.ior = n2;
        } else {
            h.dn = reflect(d, h.normal);
        }
    } else {
        if (uniform01(seed, path_id, bounce) < __uint_as_float(mat.y)) h.dn = reflect(d, h.normal);
    }
    h.o_next = h.point + h.dn * kReorigin;
    return h;
}

// ---------------------------------------------------------------------------------------------------
// Per-cell accumulators: fb_q[cell][4] = {sum x, sum y, sum z, segments} as 64-bit integers (32 bytes per cell).
// flush_cells adds every flushing lane's (x, y, z, segs) to its cell.  Lanes that sit next to each other in the wave and
// flush into the same cell (the sample blocks of one cell are dealt to neighbouring lanes; the continuing rays they spawn
// stay neighbours in the ring) are summed first, and the four words of a cell are then added by four neighbouring lanes
// of ONE wave instruction, so a cell's record travels as a single 32-byte atomic request instead of up to four scattered
// ones per path.  The runs are summed by LDS atomics (ds_add_u64 into one staged record per run: integer adds, so the order
// does not matter); the register version this replaces -- a six-step segmented reduction over nine shuffled dwords -- cost
// 1,600 VALU instructions per call, a tenth of a primary step.  Wave-uniform control flow; `stage` is >= 576 dwords of the
// wave's LDS that nothing else uses during the call (the ray stash and the best keys: dead between the confirm stage and the
// next step's set-up), 8-byte aligned.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void flush_cells(unsigned long long* fb_q, bool flush, uint32_t cell, unsigned long long x, unsigned long long y,
                                            unsigned long long z, unsigned long long segs, uint32_t* stage, unsigned lane)
{
    const unsigned long long fmask = __ballot(flush);
    if (!fmask) return;
    // runs of neighbouring flushing lanes with the same cell
    const uint32_t key = flush ? cell : 0xffffffffu;  // cell indices stay below 2^31
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool head = flush && (lane == 0 || prev != key);
    const unsigned long long hmask = __ballot(head);
    const unsigned n = (unsigned)__builtin_popcountll(hmask);
    const unsigned e = mbcnt64(hmask) + (head ? 1u : 0u) - 1u;  // this lane's run (flushing lanes only: a head lies at or before them)
    // one staged record per run: {cell} in stage[0..63], {x, y, z, segs} as 8 dwords from stage[64 + 8 e]
    unsigned long long* rec = reinterpret_cast<unsigned long long*>(stage + 64);
    for (unsigned k = lane; k < 4u * n; k += 64u) rec[k] = 0ull;
    if (head) stage[e] = cell;
    wave_lds_sync();
    if (flush) {
        lds_add_u64(rec + 4u * e + 0u, x);
        lds_add_u64(rec + 4u * e + 1u, y);
        lds_add_u64(rec + 4u * e + 2u, z);
        lds_add_u64(rec + 4u * e + 3u, segs);
    }
    wave_lds_sync();
    for (unsigned base = 0; base < n; base += 16) {
        const unsigned r = base + (lane >> 2), c = lane & 3u;
        if (r < n) {
            const unsigned long long v = rec[4u * r + c];
            if (v) atomicAdd(fb_q + (uint64_t)stage[r] * 4 + c, v);
        }
    }
    wave_lds_sync();
}

// q = rint(v * 2^24); a NaN value (a path the reference's own arithmetic turned into NaN, e.g. a ray that
// started inside a Box, axis_aligned_box.rs:70-95) contributes 0.
__device__ __forceinline__ unsigned long long quantize(float v)
{
    const float x = v * 16777216.0f;
    // path values are sky colours in [0, 1]: the 32-bit conversion (two instructions) serves them; anything else, NaN included,
    // takes the 64-bit one (a dozen).  Same rounding (to nearest even), same result.
    if (__builtin_fabsf(x) < 2147483520.0f) return (unsigned long long)(long long)__float2int_rn(x);
    return v == v ? (unsigned long long)__float2ll_rn(x) : 0ull;
}

}  // namespace
}  // namespace rt
