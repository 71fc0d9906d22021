// rt_grid.h — continuing rays of the Projector kernel walk a uniform grid over the scan-path spheres.
//
// A ray that left a surface has no shared origin to exploit, and scanning the whole table for it costs eight VALU
// instructions per sphere.  Instead every lane steps its ray through the grid's cells (3-D DDA) and looks only at the
// spheres registered there.  Nothing about the RESULT changes hands: a sphere whose reference discriminant (sphere.rs:13-18,
// computed here with the same operations on the same operands as sphere_first_t) is not negative goes to the wave's
// candidate queue, and confirm_batch runs the exact sequence for it as it does for the table scan; a box entry goes there
// unconditionally.  The walk only has to guarantee that the entry which ends up closest is seen, and it may stop early:
//   * registration (rt_host_grid.h) puts every entry into each cell within E of its surface, E >= the distance by which the
//     float discriminant can accept a ray outside the sphere + the error of the returned t + the walk's placement error;
//   * a sphere whose first root t1 is safely inside (t_min, inf) WILL be confirmed with a t within `slack` of the fast
//     estimate made here, so `bound` = estimate + slack is an upper bound of the winning t, and once the walk has covered
//     the ray up to bound + slack every entry that could still win has been seen.
// Control flow is wave-uniform (the loop runs while any lane is walking), so the candidate queue works as in the table scan.
#pragma once
#include <type_traits>

#include "rt_device.h"

#ifndef RT_WALK_ITEMS
#define RT_WALK_ITEMS 2  // measured with cells of 0.8 x the mean spacing: 1 / 2 / 3 / 4 entries per pass = 3.97 / 3.76 / - / 4.00 ms per C2 frame
#endif

namespace rt {

typedef __attribute__((address_space(3))) const uint32_t LdsU32;
typedef __attribute__((address_space(3))) const uint16_t LdsU16;
typedef float RtGV4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const RtGV4 LdsGV4;

__device__ __forceinline__ float4 grid_sphere(const float4* t, unsigned j) { return t[j]; }
__device__ __forceinline__ float4 grid_sphere(LdsGV4* t, unsigned j)
{
    const RtGV4 v = t[j];
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int kWalkItems = RT_WALK_ITEMS;  // entries of a cell tested per pass of the walk
static_assert((unsigned)kWalkItems <= kGridItemPad, "the walk reads kWalkItems entries at a time, used or not");

// Rays the grid may serve: the origin within the error budget (rt_host_grid.h: |o - centre| <= 2 * radius) and a finite
// direction.  A ray with a NaN component "hits" every rotation-0 box of the scene at t_min in the reference (f32::max / min
// drop the NaN in the slab test), wherever the box is: only the table scan reproduces that.
__device__ __forceinline__ bool grid_admits(const DevGrid& g, F3 o, F3 d)
{
    const F3 oc = o - f3(g.centre[0], g.centre[1], g.centre[2]);
    return dot(oc, oc) <= g.origin_r2 && dot(d, d) < __builtin_inff();
}

template <int R, bool ASK = true, typename CellPtr, typename ItemPtr, typename SphPtr>
__device__ __forceinline__ void grid_walk(const DevGrid& g, CellPtr cells, ItemPtr items, SphPtr sph, F3 o, F3 d, float a, bool live,
                                          unsigned slot_base, WaveLds<R>& w, const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane,
                                          unsigned long long& dbg)
{
    // Reads that may turn out unused (entries a lane does not have, the next cell's record of a lane that stays) are made by every lane
    // without a branch when the grid is in LDS.  With the grid in global memory the caller chooses: ASK = true reads per lane and only what
    // exists -- for rays that walk in from outside through empty cells (ray lists: the 10,000-sphere list 2.12 -> 1.51 ms); ASK = false
    // reads unconditionally as in LDS -- for the Projector kernel's continuing rays, which start on surfaces inside the cloud (C4 at
    // 128 spp: 36.9 against 37.7 ms).
    constexpr bool kUnconditional = !std::is_same<CellPtr, const uint32_t*>::value || !ASK;
    const float inf = __builtin_inff();
    const float rs = __builtin_amdgcn_rsqf(a);  // ~ 1 / |d|
    const F3 oc = o - f3(g.centre[0], g.centre[1], g.centre[2]);
    const float slack = g.slack_k * (__builtin_amdgcn_sqrtf(dot(oc, oc)) + g.radius) * rs;
    const float ia = __builtin_amdgcn_rcpf(a);
    const float rs1 = rs * 1.001f;  // |r| * rs1 >= the half chord of any sphere of radius r in t units

    // entry: slab test against the grid bounds (fmax / fmin drop NaNs; directions below 1e-30 count as parallel)
    bool active = live;
    float t0 = 0.0f, t1 = inf;
    float tdx = inf, tdy = inf, tdz = inf, ivx = 0.f, ivy = 0.f, ivz = 0.f;
    if (__builtin_fabsf(d.x) >= 1e-30f) {
        ivx = __builtin_amdgcn_rcpf(d.x);
        const float ta = (g.lo[0] - o.x) * ivx, tb = (g.hi[0] - o.x) * ivx;
        t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        tdx = g.cell * __builtin_fabsf(ivx);
    } else if (o.x < g.lo[0] || o.x > g.hi[0]) active = false;
    if (__builtin_fabsf(d.y) >= 1e-30f) {
        ivy = __builtin_amdgcn_rcpf(d.y);
        const float ta = (g.lo[1] - o.y) * ivy, tb = (g.hi[1] - o.y) * ivy;
        t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        tdy = g.cell * __builtin_fabsf(ivy);
    } else if (o.y < g.lo[1] || o.y > g.hi[1]) active = false;
    if (__builtin_fabsf(d.z) >= 1e-30f) {
        ivz = __builtin_amdgcn_rcpf(d.z);
        const float ta = (g.lo[2] - o.z) * ivz, tb = (g.hi[2] - o.z) * ivz;
        t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
        t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        tdz = g.cell * __builtin_fabsf(ivz);
    } else if (o.z < g.lo[2] || o.z > g.hi[2]) active = false;
    if (t0 > t1 + slack) active = false;  // misses the bounds by more than the slack: touches no registered box

    // start cell (clamped: the entry point may sit a rounding error outside) and the first boundary crossings
    const float ex = __builtin_fmaf(t0, d.x, o.x), ey = __builtin_fmaf(t0, d.y, o.y), ez = __builtin_fmaf(t0, d.z, o.z);
    const int dimx = (int)g.dim[0], dimy = (int)g.dim[1], dimz = (int)g.dim[2];
    // clamped as floats (fmax / fmin drop a NaN), so the conversion is defined for every lane, live or not
    const int cx = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf((ex - g.lo[0]) * g.inv_cell), 0.0f), (float)(dimx - 1));
    const int cy = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf((ey - g.lo[1]) * g.inv_cell), 0.0f), (float)(dimy - 1));
    const int cz = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf((ez - g.lo[2]) * g.inv_cell), 0.0f), (float)(dimz - 1));
    const int sx = d.x > 0.0f ? 1 : -1, sy = d.y > 0.0f ? 1 : -1, sz = d.z > 0.0f ? 1 : -1;
    float tmx = tdx < inf ? (g.lo[0] + (float)(cx + (sx > 0 ? 1 : 0)) * g.cell - o.x) * ivx : inf;
    float tmy = tdy < inf ? (g.lo[1] + (float)(cy + (sy > 0 ? 1 : 0)) * g.cell - o.y) * ivy : inf;
    float tmz = tdz < inf ? (g.lo[2] + (float)(cz + (sz > 0 ? 1 : 0)) * g.cell - o.z) * ivz : inf;

    // The walk keeps its cell as one linear index into an array with a shell of "outside" records around the grid's cells
    // (rt_host_grid.h): a step adds the stride of its axis, and a ray leaves the grid when the record it steps to says so -- no
    // coordinates, no range compares, no counters.  (Round 3 first packed the cells left on each axis into one word: two selects, a
    // mask, a compare and a subtraction per step, plus three literals the compiler re-materialised in every pass.)
    const unsigned rowx = (unsigned)dimx + 2u, rowy = (unsigned)dimy + 2u;
    unsigned idx = (((unsigned)cz + 1u) * rowy + (unsigned)cy + 1u) * rowx + (unsigned)cx + 1u;
    const int ix = sx, iy = sy * (int)rowx, iz = sz * (int)(rowx * rowy);
    float bound = inf;  // upper bound of the t that will win, from spheres already queued
    unsigned k = 0, cnt = 0, start = 0;
    {
        // The start cell's record, read here and not by a first pass that tests nothing (that pass was one in fourteen).
        // Every lane reads (the indices are clamped, so the address is valid whatever the lane holds) and inactive lanes drop the
        // result.  Written as `if (active) { rec = cells[..]; start = ..; cnt = ..; }` the block is IDENTICAL to the one at the end of
        // the loop below (k is 0 here too); ROCm 7.2's clang merges the two, the loop then has two entries, and the structuriser
        // nests it inside a divergent loop (lanes that fetch a record / lanes that walk), where the walk's ballots, mbcnt ranks and
        // 64-lane confirm batches run on a subset of the wave: candidates handed to absent lanes were never confirmed -- lost hits
        // on half of the fuzz cases.  profiles/r03_walk_isa_if_vs_select.txt has both listings; tools/check_walk_isa.py (run by
        // tests/test_walk_isa.py) reads the loop shape out of every build.  Rule: no two identical conditional blocks at the
        // entry and at the latch of a loop that contains wave-level operations.
#ifdef RT_GRID_ACTIVE_LOAD  // the form that lost hits (kept for the ISA comparison)
        if (active) {
            const unsigned rec = cells[idx];
            start = rec & 0xfffffu;
            cnt = rec >> 20;
        }
#else
        const unsigned rec = cells[idx];
        start = active ? rec & 0xfffffu : 0u;
        cnt = active ? rec >> 20 : 0u;
#endif
    }
#if RT_DBG_WALK == 4
    if (lane == 0) ++dbg;
#endif
    while (__ballot(active)) {
#if RT_DBG_WALK == 1
        if (lane == 0) ++dbg;
#endif
#if RT_DBG_WALK == 2
        if (active) ++dbg;
#endif
        // up to kWalkItems entries of the current cell per pass: their loads and discriminants are independent of each other
        bool pass[kWalkItems];
        unsigned jj[kWalkItems];
        const unsigned rem = active ? cnt - k : 0u;
        // A pass that uses its cell up (most do: a cell holds two entries on average) will step unless the walk ends, and where to
        // does not depend on the entries: the next cell's record is read together with them, not after them.
        const bool last = active && rem <= (unsigned)kWalkItems;
        const float texit = __builtin_fminf(tmx, __builtin_fminf(tmy, tmz));
        const bool stepx = tmx <= tmy && tmx <= tmz, stepy = !stepx && tmy <= tmz, stepz = !stepx && !stepy;
        const unsigned nidx = idx + (unsigned)(stepx ? ix : (stepy ? iy : iz));
        // (a walking lane sits in a cell of the grid, so nidx is at worst in the shell)
        unsigned rec_next = 0u;
        if constexpr (kUnconditional) rec_next = cells[last ? nidx : idx];  // every lane reads, no branch
        else if (last) rec_next = cells[nidx];                              // only the lanes that need it ask
#if RT_DBG_WALK == 6
        if (active && rem == 0u) ++dbg;  // lane-passes spent on a cell without entries (or on the start cell's record)
#endif
#if RT_DBG_WALK == 7
        if (lane == 0 && __ballot(rem != 0u) == 0ull) ++dbg;  // passes in which no lane tests anything
#endif
        // Straight-line: the kWalkItems item reads go out together, then the kWalkItems sphere reads, then the arithmetic -- two LDS
        // (or memory) round trips per pass instead of four (round 2 had each entry under its own branches: C2 -1.4 %, C4 -4 %).
        // Lanes without a u-th entry read whatever follows in the item list (it ends with kWalkItems spare entries) and drop the
        // result; a box entry's table row is read like a sphere's and its discriminant ignored.
        // With the grid in global memory the reads are per lane and only for entries that exist: a ray that walks in from outside through
        // empty cells would otherwise wait for gathers it drops (the 10,000-sphere ray list: 1.6 -> 2.15 ms with unconditional reads).
        unsigned it[kWalkItems];
        float4 s[kWalkItems];
        if constexpr (kUnconditional) {
#pragma unroll
            for (int u = 0; u < kWalkItems; ++u) it[u] = items[start + k + (unsigned)u];
#pragma unroll
            for (int u = 0; u < kWalkItems; ++u) {
                jj[u] = it[u] & 0x7fffu;
                s[u] = grid_sphere(sph, jj[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < kWalkItems; ++u) {
                it[u] = 0u;
                if ((unsigned)u < rem) it[u] = items[start + k + (unsigned)u];
            }
#pragma unroll
            for (int u = 0; u < kWalkItems; ++u) {
                jj[u] = it[u] & 0x7fffu;
                s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)u < rem) s[u] = grid_sphere(sph, jj[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < kWalkItems; ++u) {
#if RT_DBG_WALK == 3
            if ((unsigned)u < rem && !(it[u] & 0x8000u)) ++dbg;
#endif
            // sphere.rs:13-18, operation for operation as in sphere_first_t
            const F3 lo = o - f3(s[u].x, s[u].y, s[u].z);
            const float half_b = dot(lo, d);
            const float c = dot(lo, lo) - s[u].w * s[u].w;
            const float disc = half_b * half_b - a * c;
            const bool box = (it[u] & 0x8000u) != 0u;  // box: the exact slab test decides
            const bool sphere_hit = (unsigned)u < rem && !box && !(disc < 0.0f);
            pass[u] = sphere_hit || ((unsigned)u < rem && box);
            // tca = the parameter of closest approach.  If it is safely beyond t_min and the discriminant above the reference's 1e-6
            // (sphere.rs:52; an absolute number, so it has to be asked for: slack scales with the scene), sphere_first_t WILL report a
            // hit: t1 <= tca if t1 is valid, else t2 = tca + half chord <= tca + |r| / |d|, which lies beyond tca > t_min.
            const float tca = -half_b * ia;
            const float nb = __builtin_fminf(bound, tca + __builtin_fmaf(__builtin_fabsf(s[u].w), rs1, slack));
            bound = (sphere_hit && tca > kTMin + slack && disc > 1e-6f) ? nb : bound;
        }
        k += rem < (unsigned)kWalkItems ? rem : (unsigned)kWalkItems;
#pragma unroll
        for (int u = 0; u < kWalkItems; ++u) {
            const unsigned long long mask = __ballot(pass[u]);
            if (mask) push_candidates<R>(w, sc, mask, pass[u], slot_base, jj[u], head, tail, lane);
        }
        // once the cell is used up: stop, or step to the next cell
        if (last) {
            if (bound < texit - slack || rec_next == kGridOutside) {
                active = false;  // everything that can still win starts before texit - slack and has been seen / the ray leaves the grid
            } else {
#if RT_DBG_WALK == 5
                ++dbg;
#endif
                idx = nidx;
                tmx += stepx ? tdx : 0.0f;
                tmy += stepy ? tdy : 0.0f;
                tmz += stepz ? tdz : 0.0f;
                start = rec_next & 0xfffffu;
                cnt = rec_next >> 20;
                k = 0;
            }
        }
    }
}

}  // namespace rt
