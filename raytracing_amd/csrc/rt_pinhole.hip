// rt_pinhole.hip — the render loop specialised for RayGeneratorConfig::Projector (scene_config.rs:84-108): the bench kernel.
//
// Every primary ray leaves the same origin O towards target(i, j) = corner + i us + j vs, so "can this ray reach sphere p" is an
// affine form in the cell indices, |A' + i B' + j C'| >= |target - O| with {A, B, C} = (p - O) . {corner - O, us, vs} / sK
// (pinhole_table_kernel).  The kernel never runs that test over the whole table:
//   * pinhole_tiles_kernel culls the table once per launch for every tile of 64 work items (the spheres ANY ray of the tile's
//     cells can pass: the same expression bounded over the tile's box of cell indices);
//   * a wave's 64 lanes hold work units (a cell and a block of its samples) of neighbouring cells; when their cells change, the
//     wave culls its tiles' lists once more with its own tighter box into a <= 40-entry LDS list (cull_group), and every primary
//     step tests only that list (scan_cull_list) -- on C2 0.4 entries on average, and a group whose list is empty skips stash, scan
//     and confirm altogether.
// One ray per lane (R = 1).  A step of a wave is either a PRIMARY step (every lane starts the next sample of its unit) or a
// CONTINUING step: rays that hit something are scattered (scene.rs:68-97) and appended to the wave's ring in global memory, and
// whenever 64 have collected the wave pops them and walks the uniform grid over the spheres (rt_grid.h; in LDS for tables of up to
// 1,024 entries, else in global memory), falling back to the 8-instruction table filter only for rays the grid
// does not admit.  Every candidate either way is confirmed with the reference's exact operation sequence (confirm_batch), so the
// results are those of the reference's linear scan.  Work units come from 32 interleaved atomic counters in batches of 64;
// per-cell sums are integers (order-independent), combined per wave in LDS before the global atomics (flush_cells).
#include "rt_complex.h"
#include "rt_grid.h"

#include <cstdlib>

namespace rt {
namespace {

// In-kernel stopwatch for one region per build (-DRT_DBG_TIME=n, measurement builds only): lane 0 of every wave adds the
// s_memtime deltas of region n to a sum that the kernel returns in place of the segment total (tools/time_frame.py prints it).
#ifdef RT_DBG_TIME
#define RT_T0(n) if (RT_DBG_TIME == n) t_region = __builtin_readcyclecounter();
#define RT_T1(n) if (RT_DBG_TIME == n && lane == 0) my_dbg += __builtin_readcyclecounter() - t_region;
#else
#define RT_T0(n)
#define RT_T1(n)
#endif


// 512 threads: 8 waves share the LDS tables, 2 blocks per CU = 4 waves per SIMD at 128 VGPRs (none spilled with one ray per lane,
// RT_PIN_R = 1, and packed-f32 code generation off; 109 with two rays per lane).  Measured alternatives (two rays per lane): 384 threads / 3 waves per
// SIMD / 168 VGPRs and 256 threads / 2 waves per SIMD / 187 VGPRs without spills were both slower (C2 frame 9.9 ms and 7.7 ms
// against 6.0 ms): the grid walk lives on occupancy, not on registers.
#ifndef RT_PRIMARY_REUSE
#define RT_PRIMARY_REUSE 0
#endif
#ifndef RT_PIN_THREADS
#define RT_PIN_THREADS 512
#define RT_PIN_WAVES 4
#endif
constexpr int kPinThreads = RT_PIN_THREADS;

// One entry of the shared-origin table.  Every primary ray of a Projector points from O to
// T(i, j) = corner + i us + j vs, so with q = p - O the projection numerator q . (T - O) = A + i B + j C is affine in the
// cell indices: A = q . (corner - O), B = q . us, C = q . vs.  The entry is {A, B, C} / sK, and the test is
// |A' + i B' + j C'| >= |T - O|: two FMAs and a compare per ray-sphere pair.
__device__ __forceinline__ float4 pinhole_entry(const TraceParams& p, unsigned k, F3 O)
{
    const DevScene& sc = p.sc;
    if (k >= sc.n_sph) return make_float4(0.f, 0.f, 0.f, 0.f);  // padding: |0| >= L never passes
    const float4 e = sc.sph_exact[k];
    // q = fl(p - O) is exactly minus the reference's local origin fl(O - p) (transform.rs:22), so K' below is exact for
    // the operands every later test uses.
    const float qx = e.x - O.x, qy = e.y - O.y, qz = e.z - O.z;
    const double qq = (double)qx * qx + (double)qy * qy + (double)qz * qz;
    const double Kq = (qq - (double)e.w * e.w) - (double)p.prim_margin;
    const float4 always = make_float4(__builtin_inff(), 0.f, 0.f, 0.f);  // inf + i 0 + j 0 passes against every L
    if (!(Kq > 0.0)) return always;
    const double qn = __builtin_sqrt(qq);
    const double sK = (__builtin_sqrt(Kq) - 0x1p-24 * qn * (4.0 + 2.0 * (double)p.prim_rho)) * (1.0 - 0x1p-22);
    if (!(sK > 1e-6 * qn)) return always;
    const DevGen& g = p.gen;
    const double cx = (double)g.corner[0] - (double)O.x, cy = (double)g.corner[1] - (double)O.y, cz = (double)g.corner[2] - (double)O.z;
    const double A = qx * cx + qy * cy + qz * cz;
    const double B = (double)qx * g.u_step[0] + (double)qy * g.u_step[1] + (double)qz * g.u_step[2];
    const double C = (double)qx * g.v_step[0] + (double)qy * g.v_step[1] + (double)qz * g.v_step[2];
    const float a = (float)(A / sK), b = (float)(B / sK), c = (float)(C / sK);
    const float big = 1e30f;
    if (!(__builtin_fabsf(a) < big && __builtin_fabsf(b) < big && __builtin_fabsf(c) < big)) return always;  // no 0 * inf later
    return make_float4(a, b, c, 0.f);
}

// The shared-origin table of a launch: built once in global memory, read by the per-wave cull (rt_device.h cull_group).
__global__ void pinhole_table_kernel(const TraceParams p, float4* table)
{
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < p.sc.n_sph_pad) table[k] = pinhole_entry(p, k, f3(p.gen.a[0], p.gen.a[1], p.gen.a[2]));
}

// The tile lists of a launch (rt_device.h cull_tiles): one wave per tile culls the table against the tile's box of cells.
__global__ __launch_bounds__(256) void pinhole_tiles_kernel(const TraceParams p, const float4* table, uint32_t* tl)
{
    const unsigned lane = lane_id();
    const unsigned t = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned tile_cells = 1u << p.tile_shift;
    const unsigned n_tiles = (p.cell_count + tile_cells - 1u) >> p.tile_shift;
    if (t >= n_tiles) return;
    const uint32_t m0 = t << p.tile_shift;
    const uint32_t m1 = p.cell_count - m0 < tile_cells ? p.cell_count : m0 + tile_cells;
    const float inf = __builtin_inff();
    float imin = inf, imax = -inf, jmin = inf, jmax = -inf;
    for (uint32_t m = m0 + lane; m < m1; m += 64u) {
        uint32_t i, j;
        cell_indices(p, m, i, j);
        imin = __builtin_fminf(imin, (float)i);
        imax = __builtin_fmaxf(imax, (float)i);
        jmin = __builtin_fminf(jmin, (float)j);
        jmax = __builtin_fmaxf(jmax, (float)j);
    }
    const CullBox x = cull_box(cull_geom(p.gen), wave_min(imin), wave_max(imax), wave_min(jmin), wave_max(jmax));
    uint32_t* rec = tl + (size_t)t * kTileRec;
    const unsigned n_pad = p.sc.n_sph_pad;
    unsigned n = 0;
    for (unsigned base = 0; base < n_pad; base += 64u) {
        const unsigned k = base + lane;
        const float4 e = table[k < n_pad ? k : n_pad - 1u];
        const bool pass = k < n_pad && cull_pass(x, e.x, e.y, e.z);
        const unsigned long long mask = __ballot(pass);
        const unsigned pos = n + mbcnt64(mask);
        if (pass && pos < kTileListCap) rec[1u + pos] = k;
        n += (unsigned)__builtin_popcountll(mask);
    }
    if (lane == 0) rec[0] = n;
}

// Constants of the general filter (FLAVOR 0 of scan_step) for slot r.
template <int R>
__device__ __forceinline__ void general_consts(ScanConsts<R>& c, int r, F3 o, F3 d, float a, bool live, const DevScene& sc)
{
    const float rs = 1.0f / __builtin_sqrtf(a);
    c.dh[r] = d * rs;
    c.nod[r] = -dot(o, c.dh[r]);
    const float oo = dot(o, o);
    const float margin = kCullKappaU * (oo + sc.c2max);
    c.s0[r] = live ? (oo - margin) : __builtin_inff();
    c.pp[r] = f3(-2.0f * o.x, -2.0f * o.y, -2.0f * o.z);
    if (!live) {  // q = 0 * 0 - inf = -inf fails the filter, whatever the slot holds (NaNs pass it otherwise)
        c.dh[r] = f3(0.f, 0.f, 0.f);
        c.nod[r] = 0.0f;
        c.pp[r] = f3(0.f, 0.f, 0.f);
    }
}

// The table scan with the general filter.  from_global: the block's LDS holds the grid's tables instead of {p, K}, so this
// (rare) scan reads them from global memory with wave-uniform loads.
template <int R, bool STREAM>
__device__ __forceinline__ void general_scan(bool from_global, const float4* tileG, unsigned n_pad, float4* stage, const ScanConsts<R>& c, WaveLds<R>& w,
                                             const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane)
{
    if (STREAM) scan_tile<R, 0, true>(tileG, n_pad, stage, c, w, sc, head, tail, lane);
    else if (from_global) scan_tile<R, 0, false>(sc.sph_cull, n_pad, stage, c, w, sc, head, tail, lane);
    else scan_tile<R, 0, false>(tileG, n_pad, stage, c, w, sc, head, tail, lane);
}

// STREAM = false: the sphere tables (<= kTileSpheres spheres) live in LDS.  STREAM = true: both are streamed from global / L2.
template <int R, bool STREAM, bool CX>
__global__ __launch_bounds__(kPinThreads, RT_PIN_WAVES) void pinhole_kernel(const TraceParams p)
{
    extern __shared__ float4 tiles[];  // [n_sph_pad] general {p, K}, then 3 floats per sphere {A, B, C} / sK (LDS mode), then the BVH top
    __shared__ WaveLds<R> wl[kPinThreads / 64];
    __shared__ float4 stage_all[STREAM ? (kPinThreads / 64) * 128 : 1];
    __shared__ float4 cull_all[kPinThreads / 64][R][kCullCap];  // per wave and slot group: the spheres its current cells can reach
    // Per work unit: the primary ray's direction, its cell indices and the filter threshold.  Every sample of a unit starts with the
    // same ray (the reference's Projector has no jitter), so these are worked out when the unit is dealt and read back each step.
    // (rows 6 / 7, RT_PRIMARY_REUSE builds only: the scan's result for that ray, the 64-bit best key)
    __shared__ float unit_ray_all[kPinThreads / 64][R][RT_PRIMARY_REUSE ? 8 : 6][64];

    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x >> 6;
    WaveLds<R>& w = wl[wave];
    float4* stage = stage_all + (STREAM ? wave * 128 : 0);
    DevScene sc = p.sc;
    const unsigned n_pad = sc.n_sph_pad;
    const F3 O = f3(p.gen.a[0], p.gen.a[1], p.gen.a[2]);
    // LDS tables (not streamed): grid mode 1 keeps {p, r} of every sphere, the grid's cells and its item lists; otherwise the
    // general filter's {p, K}.  The shared-origin table always stays in global memory: only the cull reads it, once per unit.
    const unsigned gm = p.grid_mode;  // 0: no grid, 1: grid resident in LDS (LDS-table kernel), 2: grid read from global memory (streamed kernel)
    const DevGrid& grid = sc.grid;
    const unsigned grid_words = gm == 1 ? grid.n_cells + ((grid.n_items + 1u) >> 1) : 0u;
    uint32_t* grid_lds = reinterpret_cast<uint32_t*>(tiles + n_pad);
    const float4* tileG = STREAM ? sc.sph_cull : tiles;
    if (!STREAM) {
        const float4* src = gm == 1 ? sc.sph_exact : sc.sph_cull;
        for (unsigned k = threadIdx.x; k < n_pad; k += kPinThreads) tiles[k] = src[k];
        if (gm == 1) {
            const uint32_t* items32 = reinterpret_cast<const uint32_t*>(grid.items);  // padded to whole words by the host
            for (unsigned k = threadIdx.x; k < grid_words; k += kPinThreads) grid_lds[k] = k < grid.n_cells ? grid.cells[k] : items32[k - grid.n_cells];
        }
    }
    // triangle meshes: the top of the BVH (first nodes of the breadth-first array) sits behind the tables
    float4* bvh_top = tiles + (STREAM ? 0u : n_pad + ((grid_words + 3u) >> 2));
    if (CX) {
        const float4* src = reinterpret_cast<const float4*>(sc.bvh_nodes);
        for (unsigned k = threadIdx.x; k < p.bvh_lds_nodes * 8u; k += kPinThreads) bvh_top[k] = src[k];
    }
    if (!STREAM && gm == 1) sc.sph_exact = tiles;  // confirm and shading read {p, r} from the LDS copy
    // the object index of a scan-path hit is only looked up when something reads it
    const bool need_obj = CX || sc.n_pl != 0 || p.first_id != nullptr;
    // scenes with meshes: the bottom of every lane's traversal stack (rt_complex.h mesh_first_hit)
    constexpr unsigned kStack = CX ? (STREAM ? kBvhLdsStackStream : kBvhLdsStack) : 0u;
    __shared__ uint2 bvh_stack_all[kStack ? kStack * kBvhLdsStackStride : 1];
    static_assert(kPinThreads == (int)kBvhLdsStackStride, "one stack column per thread of the block");
    const BvhLds top{(LdsF4*)bvh_top, CX ? p.bvh_lds_nodes : 0u, kStack ? (LdsU2*)bvh_stack_all + threadIdx.x : nullptr, kStack};
    if (!STREAM || CX) __syncthreads();

    // per-wave ring of continuing rays (global memory, only this wave touches it)
    const unsigned gwave = blockIdx.x * (kPinThreads / 64) + wave;
    uint32_t* ring = p.secq + (size_t)gwave * kSecFields * kSecCap;
    unsigned sec_count = 0;  // wave-uniform

    // primary feeder slots: one work unit (cell, sample block) each
    uint32_t um[R], usample[R], uend[R], usegs[R];
    unsigned long long ax[R], ay[R], az[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        um[r] = 0xffffffffu;
        usample[r] = uend[r] = usegs[r] = 0;
        ax[r] = ay[r] = az[r] = 0;
    }
    unsigned long long my_segments = 0, my_dbg = 0;
    bool exhausted = false, prim_done = false;
    unsigned long long chunk_next = 0, chunk_end = 0;  // the wave's current batch of work units (wave-uniform)
    // cull lists (wave-uniform state): entries in the list, and whether the group's cells changed since it was built
    unsigned n_cull[R];
    bool cull_stale[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        n_cull[r] = 0;
        cull_stale[r] = true;
    }
    const unsigned long long total_units = (unsigned long long)p.cell_count * p.units_per_cell;
    const unsigned long long total_batches = (total_units + 63ull) / 64ull;
    unsigned ctr = gwave % kUnitCounters, ctr_done = 0;  // the work counter this wave draws from; counters it has seen run out
    const uint32_t max_b = p.st.max_bounces;

    for (;;) {
        const bool do_sec = sec_count >= 64u * R || (prim_done && sec_count > 0);
        if (!do_sec && prim_done) break;
        unsigned long long t_region = 0;
        (void)t_region;
        RT_T0(11)
        RT_T0(1)  // region 1: the whole step
        RT_T0(2)  // region 2: set-up of a primary step
        RT_T0(3)  // region 3: set-up of a continuing step (ring loads)

        ScanConsts<R> c;
        F3 ro[R], rd[R];
        float rior[R];
        uint32_t rm[R], rsample[R], rb[R];
        bool live[R], scan_live[R];
        bool fresh[R];  // primary step: this lane's unit is at its first sample, so its ray has not been scanned yet
        bool have_best = true;  // w.best holds this step's scan results (false: a primary step whose cull lists were empty)
#pragma unroll
        for (int r = 0; r < R; ++r) fresh[r] = false;
        unsigned head = 0, tail = 0;

        if (!do_sec) {
            // ================= primary step =================
            bool any_live = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                bool need_unit = (um[r] == 0xffffffffu);
                unsigned long long need = __ballot(need_unit);
                if (need == ~0ull && !exhausted) {
                    // Units come in batches of 64 from kUnitCounters interleaved work counters (batch b belongs to counter
                    // b % kUnitCounters): all waves of the chip adding to ONE address took, by that counter alone, the time of a
                    // third of a C2 frame.  A wave starts at its own counter and moves on when one runs out, so neighbouring
                    // batches are still in flight at the same time (cull lists, grid cells and accumulators stay cache-warm).
                    // A group takes a new batch only when all of its lanes are through: lanes whose unit was shorter (the last
                    // sample block of a cell when spp is not a multiple of the unit) wait for the others.  Serving them at once
                    // let the lanes of a group drift over cells of many batches, and a group whose cells are far apart rebuilds
                    // an overflowing cull list at every step (spp 64 in units of 6: 24 ms instead of 3.5 ms per C2 frame).
                    if (chunk_next == chunk_end) {
                        for (;;) {
                            unsigned long long v = 0;
                            if (lane == 0) v = atomicAdd(p.next_unit + (size_t)ctr * kUnitCounterStride, 1ull);
                            v = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)v);
                            const unsigned long long batch = v * kUnitCounters + ctr;
                            if (batch < total_batches) {
                                chunk_next = batch * 64ull;
                                chunk_end = chunk_next + 64ull < total_units ? chunk_next + 64ull : total_units;
                                break;
                            }
                            ctr = (ctr + 1u) % kUnitCounters;
                            if (++ctr_done == kUnitCounters) {
                                exhausted = true;
                                break;
                            }
                        }
                    }
                    const unsigned long long avail = chunk_end - chunk_next;
                    const unsigned cnt = (unsigned)__builtin_popcountll(need);
                    const unsigned take = avail < cnt ? (unsigned)avail : cnt;
                    const unsigned long long base = chunk_next;
                    chunk_next += take;
                    cull_stale[r] = true;
                    if (need_unit && mbcnt64(need) < take) {
                        unsigned long long unit = base + mbcnt64(need);
                        {
                            uint32_t m, k;
                            if ((total_units >> 32) == 0) {  // (a 64-bit division costs a few hundred instructions)
                                m = (uint32_t)unit / p.units_per_cell;
                                k = (uint32_t)unit - m * p.units_per_cell;
                            } else {
                                m = (uint32_t)(unit / p.units_per_cell);
                                k = (uint32_t)(unit - (unsigned long long)m * p.units_per_cell);
                            }
                            um[r] = m;
                            usample[r] = k * p.unit_samples;
                            uint32_t e = usample[r] + p.unit_samples;
                            uend[r] = e < p.spp ? e : p.spp;
                            usegs[r] = 0;
                            ax[r] = ay[r] = az[r] = 0;
                            F3 o_unused = O, dir;
                            float ior_unused = 0.f;
                            primary_ray(p, m, o_unused, dir, ior_unused);
                            uint32_t ci, cj;
                            cell_indices(p, m, ci, cj);
                            const F3 corner = f3(p.gen.corner[0], p.gen.corner[1], p.gen.corner[2]);
                            const F3 us = f3(p.gen.u_step[0], p.gen.u_step[1], p.gen.u_step[2]), vs = f3(p.gen.v_step[0], p.gen.v_step[1], p.gen.v_step[2]);
                            const F3 dv = ((corner + ((float)ci * us)) + ((float)cj * vs)) - O;  // the reference's target - origin (scene_config.rs:98-100)
                            float (*ur)[64] = unit_ray_all[wave][r];
                            ur[0][lane] = dir.x; ur[1][lane] = dir.y; ur[2][lane] = dir.z;
                            ur[3][lane] = (float)ci; ur[4][lane] = (float)cj;
                            ur[5][lane] = __builtin_sqrtf(dot(dv, dv)) * (1.0f - 0x1p-20f);  // a lower bound of |target - O|
                        }
                    }
                }
                live[r] = (um[r] != 0xffffffffu);
                any_live |= (__ballot(live[r]) != 0ull);
            }
            if (!any_live) {
                prim_done = true;
                continue;
            }
            if (true) { RT_T1(11) }  // region 11: unit assignment (work counter, division) of a primary step
            RT_T0(12)                // region 12: ray set-up of a primary step
#pragma unroll
            for (int r = 0; r < R; ++r) {
                rm[r] = um[r];
                rsample[r] = usample[r];
                rb[r] = 0;
                ro[r] = O;
                const float (*ur)[64] = unit_ray_all[wave][r];  // written by this lane when its unit was dealt
                rd[r] = live[r] ? f3(ur[0][lane], ur[1][lane], ur[2][lane]) : f3(0.f, 0.f, 0.f);
                rior[r] = p.gen.current_ior;
                // shared-origin filter constants: cell indices (exact in f32) and a lower bound of |target - O|
                c.ci[r] = live[r] ? ur[3][lane] : 0.0f;
                c.cj[r] = live[r] ? ur[4][lane] : 0.0f;
                c.lt[r] = live[r] ? ur[5][lane] : __builtin_inff();  // dead slots are masked by `live`
                fresh[r] = live[r] && (!RT_PRIMARY_REUSE || usegs[r] == 0u);  // usegs counts the unit's primary segments
            }
            if (true) { RT_T1(2) }
            if (true) { RT_T1(12) }
            RT_T0(4)  // region 4: cull lists and the primary scan over them
            // groups whose cells changed rebuild their list (once per unit_samples steps when the units are of equal length)
            bool listed = true, any_fresh = false;
#pragma unroll
            for (int r = 0; r < R; ++r) any_fresh |= __ballot(fresh[r]) != 0ull;
            // -DRT_PRIMARY_REUSE=1 (measured, not shipped: C2 -2 %, C4 -5 %): every sample of a unit starts with the same ray and
            // therefore finds the same closest scan-path entry, so only the unit's first sample would have to look for it and the
            // others could pick its key up below.  The shipped kernel searches for every sample's ray: `fresh` is `live`.
            if (any_fresh) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (cull_stale[r]) {
                    n_cull[r] = cull_group(p.prim_table, n_pad, p.tile_lists, p.tile_shift, cull_geom(kernarg_gen(offsetof(TraceParams, gen))), c.ci[r], c.cj[r], um[r], live[r],
                                           cull_all[wave][r], lane);
                    cull_stale[r] = false;
#if RT_DBG_WALK == 10
                    if (lane == 0) my_dbg += n_cull[r];
#endif
#if RT_DBG_WALK == 11
                    if (lane == 0) ++my_dbg;
#endif
                }
                listed &= n_cull[r] <= kCullCap;
            }
#if RT_DBG_WALK == 9
            if (lane == 0 && !listed) ++my_dbg;
#endif
            // A group with empty lists (most of C2's: 0.4 entries on average) has nothing to scan and nothing to confirm: its rays'
            // stash and best keys are not even written, and the step goes on with "no scan-path hit".
            bool scan = !listed || RT_PRIMARY_REUSE;
#pragma unroll
            for (int r = 0; r < R; ++r) scan |= n_cull[r] != 0u;
            have_best = scan;
            if (scan) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const unsigned slot = r * 64 + lane;
                    w.stash[0][slot] = ro[r].x; w.stash[1][slot] = ro[r].y; w.stash[2][slot] = ro[r].z;
                    w.stash[3][slot] = rd[r].x; w.stash[4][slot] = rd[r].y; w.stash[5][slot] = rd[r].z;
                    w.stash[6][slot] = dot(rd[r], rd[r]);
                    w.best[slot] = kNoHit;
                }
                wave_lds_sync();
                if (listed) {
#pragma unroll
                    for (int r = 0; r < R; ++r) scan_cull_list<R>(cull_all[wave][r], n_cull[r], r, c.ci[r], c.cj[r], c.lt[r], fresh[r], w, sc, head, tail, lane);
                } else {
                    // more spheres in reach than a list holds (the origin sits inside a crowd): these rays take the general filter
#pragma unroll
                    for (int r = 0; r < R; ++r) general_consts<R>(c, r, ro[r], rd[r], dot(rd[r], rd[r]), fresh[r], sc);
                    general_scan<R, STREAM>(gm == 1, tileG, n_pad, stage, c, w, sc, head, tail, lane);
                }
            }
            }
            RT_T1(4)
        } else {
            // ================= continuing-ray step =================
            const unsigned n_pop = sec_count < 64u * R ? sec_count : 64u * R;
            const unsigned base = sec_count - n_pop;
            sec_count = base;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned idx = r * 64 + lane;
                live[r] = idx < n_pop;
                const unsigned e = base + (live[r] ? idx : 0u);
                float v[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) v[k] = __uint_as_float(ring[k * kSecCap + e]);
                ro[r] = f3(v[0], v[1], v[2]);
                rd[r] = f3(v[3], v[4], v[5]);
                rior[r] = v[6];
                rm[r] = ring[7 * kSecCap + e];
                rsample[r] = ring[8 * kSecCap + e];
                rb[r] = ring[9 * kSecCap + e];
                const F3 o = ro[r], d = rd[r];
                const float a = dot(d, d);
                const unsigned slot = r * 64 + lane;
                w.stash[0][slot] = o.x; w.stash[1][slot] = o.y; w.stash[2][slot] = o.z;
                w.stash[3][slot] = d.x; w.stash[4][slot] = d.y; w.stash[5][slot] = d.z;
                w.stash[6][slot] = a;
                w.best[slot] = kNoHit;
                scan_live[r] = live[r] && !nan_ray(d);  // all-NaN directions are settled by push_nan_rays
            }
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < R; ++r) push_nan_rays<R>(w, sc, live[r] && !scan_live[r], r * 64u, head, tail, lane);
            if (true) { RT_T1(3) }
            RT_T0(5)  // region 5: grid walk / table scan of the continuing rays
            // the grid serves every ray that starts within its error budget; a wave with one that does not scans the table
            bool walk = gm != 0;
            const DevGrid gw = kernarg_grid(offsetof(TraceParams, sc) + offsetof(DevScene, grid));  // this step's copy of the grid's constants
            if (walk) {
#pragma unroll
                for (int r = 0; r < R; ++r) walk &= __ballot(scan_live[r] && !grid_admits(gw, ro[r], rd[r])) == 0ull;
            }
            if (walk) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float a = dot(rd[r], rd[r]);
                    if constexpr (STREAM)  // grid mode 2
                        grid_walk<R, false>(gw, gw.cells, gw.items, sc.sph_exact, ro[r], rd[r], a, scan_live[r], r * 64u, w, sc, head, tail, lane, my_dbg);
                    else  // grid mode 1
                        grid_walk<R>(gw, (LdsU32*)grid_lds, (LdsU16*)(grid_lds + gw.n_cells), (LdsGV4*)tiles, ro[r], rd[r], a, scan_live[r], r * 64u, w, sc,
                                     head, tail, lane, my_dbg);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) general_consts<R>(c, r, ro[r], rd[r], dot(rd[r], rd[r]), scan_live[r], sc);
                general_scan<R, STREAM>(gm == 1, tileG, n_pad, stage, c, w, sc, head, tail, lane);
            }
        }

        if (do_sec) { RT_T1(5) }
        // ---- drain the candidate queue, resolve ----
        RT_T0(6)  // region 6: last confirm batch
        if (have_best) {
            wave_lds_sync();
            if (tail != head) confirm_batch<R>(w, sc, head, tail - head, lane);
            wave_lds_sync();
        }
        if (RT_PRIMARY_REUSE && !do_sec) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float (*ur)[64] = unit_ray_all[wave][r];
                const unsigned slot = r * 64 + lane;
                if (fresh[r]) {
                    const unsigned long long k = w.best[slot];
                    ur[6][lane] = __uint_as_float((unsigned)k);
                    ur[7][lane] = __uint_as_float((unsigned)(k >> 32));
                } else if (live[r]) {
                    w.best[slot] = ((unsigned long long)__float_as_uint(ur[7][lane]) << 32) | __float_as_uint(ur[6][lane]);
                }
            }
        }
        RT_T1(6)
        RT_T0(7)  // regions 7 / 8: resolve, shade, accumulate, ring push of a primary / continuing step
        RT_T0(8)

#pragma unroll
        for (int r = 0; r < R; ++r) {
            RT_T0(13)  // regions 13 / 14 / 15: resolve + shade, per-cell sums, ring push (primary steps only)
            bool push = false, fl = false;
            uint32_t fcell = 0;
            unsigned long long fx = 0, fy = 0, fz = 0, fsegs = 0;
            Shade h;
            h.o_next = ro[r]; h.dn = rd[r]; h.ior = rior[r];
            if (live[r]) {
                const unsigned slot = r * 64 + lane;
                int sj;
                const unsigned long long kscan = scan_key(sc, have_best ? w.best[slot] : kNoHit, need_obj, sj);
                unsigned long long key = planes_min_key(sc, ro[r], rd[r], kscan);
                PreHit pre;
                pre.valid = false;
                if (CX) {
                    CxHit ch;
                    int ci = -1;
                    key = cx_min_key(sc, top, ro[r], rd[r], key, ch, ci);
                    if (ci >= 0) {
                        pre.valid = true;
                        pre.front = ch.ff != 0;
                        cx_to_world(sc.cx_objs[ci], ch, pre.point, pre.normal);
                    }
                }
                ++my_segments;
                const uint32_t b = rb[r];
                const uint64_t path_id = (p.cell_begin + (uint64_t)rm[r] * p.cell_step) * (uint64_t)p.spp + rsample[r];
                bool ended = false;
                F3 value = f3(0.f, 0.f, 0.f);
                if (key != kNoHit) {
                    if (b == 0 && rsample[r] == 0) {
                        if (p.first_id) p.first_id[rm[r]] = (int32_t)(uint32_t)key;
                        if (p.first_t) p.first_t[rm[r]] = __uint_as_float((uint32_t)(key >> 32));
                    }
                    if (b + 1 == max_b) {
                        ended = true;  // bounce budget exhausted: value 0, and nobody asks where the ray would have gone
                    } else {
                        h = shade_hit(sc, ro[r], rd[r], rior[r], key, p.seed, path_id, b, pre, key == kscan ? sj : -1);
                        push = true;
                    }
                } else {
                    if (b == 0 && rsample[r] == 0) {
                        if (p.first_id) p.first_id[rm[r]] = -1;
                        if (p.first_t) p.first_t[rm[r]] = __builtin_inff();
                    }
                    value = sky(rd[r]);
                    ended = true;
                }
                if (!do_sec) {
                    // the feeder owns the unit's register sums; a primary miss ends the path here
                    ++usegs[r];
                    if (ended) {
                        ax[r] += quantize(value.x);
                        ay[r] += quantize(value.y);
                        az[r] += quantize(value.z);
                    }
                    ++usample[r];
                    if (usample[r] == uend[r]) {
                        fl = true;
                        fcell = um[r];
                        fx = ax[r];
                        fy = ay[r];
                        fz = az[r];
                        fsegs = usegs[r];
                        um[r] = 0xffffffffu;
                    }
                } else if (ended) {
                    // a continuing path ends: it has run b segments after its primary one
                    fl = true;
                    fcell = rm[r];
                    fx = quantize(value.x);
                    fy = quantize(value.y);
                    fz = quantize(value.z);
                    fsegs = b;
                }
            }
            if (!do_sec) { RT_T1(13) }
            RT_T0(14)
            // per-cell sums: neighbouring lanes of one cell are combined first (rt_device.h flush_cells)
            flush_cells(p.fb_q, fl, fcell, fx, fy, fz, fsegs, reinterpret_cast<uint32_t*>(&w.stash[0][0]), lane);
            if (!do_sec) { RT_T1(14) }
            RT_T0(15)
            // append continuing rays to the ring (wave-uniform control flow)
            const unsigned long long pm = __ballot(push);
            if (pm) {
                const unsigned e = sec_count + mbcnt64(pm);
                if (push) {
                    ring[0 * kSecCap + e] = __float_as_uint(h.o_next.x);
                    ring[1 * kSecCap + e] = __float_as_uint(h.o_next.y);
                    ring[2 * kSecCap + e] = __float_as_uint(h.o_next.z);
                    ring[3 * kSecCap + e] = __float_as_uint(h.dn.x);
                    ring[4 * kSecCap + e] = __float_as_uint(h.dn.y);
                    ring[5 * kSecCap + e] = __float_as_uint(h.dn.z);
                    ring[6 * kSecCap + e] = __float_as_uint(h.ior);
                    ring[7 * kSecCap + e] = rm[r];
                    ring[8 * kSecCap + e] = rsample[r];
                    ring[9 * kSecCap + e] = rb[r] + 1;
                }
                sec_count += (unsigned)__builtin_popcountll(pm);
            }
            if (!do_sec) { RT_T1(15) }
        }
        if (!do_sec) { RT_T1(7) } else { RT_T1(8) }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        RT_T1(1)
    }

    if (p.total_segments) {
#if defined(RT_DBG_WALK) || defined(RT_DBG_TIME)
        unsigned long long v = my_dbg;
#else
        unsigned long long v = my_segments + 0 * my_dbg;
#endif
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(p.total_segments, v);
    }
}

#ifndef RT_PIN_R
#define RT_PIN_R 1
#endif
constexpr int kR = RT_PIN_R;

// The streamed kernel (tables and grid read from global memory) serves tables beyond an LDS tile, and smaller ones whose grid does not fit
// into LDS beside the table and the per-wave state (grid_mode 2: long item lists -- large radii, grids of boxes).
inline bool stream_mode(const TraceParams& p) { return p.sc.n_sph_pad > (uint32_t)kTileSpheres || p.grid_mode == 2; }
constexpr size_t kLdsPerBlock = 80 * 1024;  // two 512-thread blocks per CU share 160 KB
inline size_t grid_lds_bytes(const DevGrid& g) { return ((size_t)g.n_cells * 4 + ((size_t)g.n_items + 1) / 2 * 4 + 15) / 16 * 16; }
// 0: no grid; 1: LDS-table kernel with {p, r}, cells and items in LDS; 2: streamed kernel, grid read from global memory
inline uint32_t grid_mode_of(const TraceParams& p, size_t static_lds)
{
    if (!p.sc.grid.enabled || getenv("RT_HIP_NO_GRID")) return 0;
    if (stream_mode(p)) return 2;
    const size_t need = static_lds + (size_t)p.sc.n_sph_pad * sizeof(float4) + grid_lds_bytes(p.sc.grid) + 512;
    return need <= kLdsPerBlock && !getenv("RT_HIP_GRID_IN_MEMORY") ? 1 : 2;  // (the environment switch: tests of the fallback)
}
inline size_t table_lds(const TraceParams& p)
{
    if (stream_mode(p)) return 0;
    return (size_t)p.sc.n_sph_pad * sizeof(float4) + (p.grid_mode == 1 ? grid_lds_bytes(p.sc.grid) : 0);
}
inline size_t dyn_lds(const TraceParams& p) { return table_lds(p) + (size_t)p.bvh_lds_nodes * sizeof(BvhNode); }

// How many tree-top nodes fit beside the kernel's static LDS and the sphere tables without costing a block per CU.
template <bool STREAM>
uint32_t bvh_lds_fit(const TraceParams& p)
{
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, (const void*)pinhole_kernel<kR, STREAM, true>) != hipSuccess) return 0;
    const size_t used = fa.sharedSizeBytes + table_lds(p) + 512;
    const size_t room = used < kLdsPerBlock ? (kLdsPerBlock - used) / sizeof(BvhNode) : 0;
    uint32_t n = p.sc.n_bvh_nodes < kBvhLdsNodes ? p.sc.n_bvh_nodes : kBvhLdsNodes;
    return room < n ? (uint32_t)room : n;
}

template <bool STREAM, bool CX>
int pinhole_blocks_per_cu(size_t dyn_bytes)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pinhole_kernel<kR, STREAM, CX>, kPinThreads, dyn_bytes) != hipSuccess || n < 1) n = 1;
    if (const char* e = getenv("RT_HIP_BLOCKS_PER_CU")) {  // tuning knob
        int v = atoi(e);
        if (v >= 1 && v <= 8) n = v;
    }
    return n;
}

unsigned pinhole_grid(const TraceParams& p, int n_cu)
{
    const size_t dyn = dyn_lds(p);
    const uint64_t slots_per_block = (uint64_t)kPinThreads * kR;
    const uint64_t want = ((uint64_t)p.cell_count * p.units_per_cell + slots_per_block - 1) / slots_per_block;
    const bool cx = p.sc.n_cx > 0;
    const int occ = stream_mode(p) ? (cx ? pinhole_blocks_per_cu<true, true>(dyn) : pinhole_blocks_per_cu<true, false>(dyn))
                                   : (cx ? pinhole_blocks_per_cu<false, true>(dyn) : pinhole_blocks_per_cu<false, false>(dyn));
    uint64_t cap = (uint64_t)n_cu * occ;
    // the persistent kernel gives the same results with any number of blocks: a caller whose rings hold fewer waves gets fewer blocks
    if (p.max_blocks && cap > p.max_blocks) cap = p.max_blocks;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    return blocks ? blocks : 1;
}

}  // namespace

size_t pinhole_secq_bytes(const TraceParams& p, int n_cu)
{
    return (size_t)pinhole_grid(p, n_cu) * (kPinThreads / 64) * kSecFields * kSecCap * sizeof(uint32_t);
}

uint32_t pinhole_grid_mode(const TraceParams& p)
{
    if (stream_mode(p)) return grid_mode_of(p, 0);
    hipFuncAttributes fa{};
    const void* fn = p.sc.n_cx > 0 ? (const void*)pinhole_kernel<kR, false, true> : (const void*)pinhole_kernel<kR, false, false>;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return 0;
    return grid_mode_of(p, fa.sharedSizeBytes);
}

// Tiles pay once the table is longer than the couple of rows a tile's list costs.  A tile should span about 64 generator cells:
// 64 work items of a contiguous range, fewer (down to 8) when the launch takes every cell_step-th cell, as the ranks of a sharded
// frame do -- a tile's box, and with it its list, grows with the span.  At most 2^18 tiles (128 MB of lists) per launch.
uint32_t pinhole_tile_shift(const TraceParams& p)
{
    if (p.sc.n_sph_pad <= 256 || getenv("RT_HIP_NO_TILES")) return 0;
    uint32_t s = 6;
    for (uint64_t step = p.cell_step; step > 1 && s > 3; step >>= 1) --s;
    if (const char* e = getenv("RT_HIP_TILE_SHIFT")) {  // tuning knob
        const int v = atoi(e);
        if (v >= 1 && v <= 12) s = (uint32_t)v;
    }
    while ((((uint64_t)p.cell_count + (1ull << s) - 1) >> s) > (1ull << 18)) ++s;
    return s;
}
static size_t tile_list_bytes(const TraceParams& p)
{
    return p.tile_shift ? (size_t)(((uint64_t)p.cell_count + (1ull << p.tile_shift) - 1) >> p.tile_shift) * kTileRec * sizeof(uint32_t) : 0;
}

size_t pinhole_table_bytes(const TraceParams& p) { return (size_t)p.sc.n_sph_pad * sizeof(float4) + tile_list_bytes(p); }

template <bool CX>
static hipError_t launch_pinhole_t(const TraceParams& p_in, int n_cu, int device, hipStream_t stream)
{
    TraceParams p = p_in;
    p.bvh_lds_nodes = CX ? (stream_mode(p) ? bvh_lds_fit<true>(p) : bvh_lds_fit<false>(p)) : 0;
    const unsigned grid = pinhole_grid(p, n_cu);
    static bool attr_set_dev[kMaxDevices] = {};  // per instantiation and device (function attributes are per device); callers hold the library mutex
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    bool& attr_set = attr_set_dev[device];
    if (!attr_set) {
        const int max_dyn = (int)kLdsPerBlock;  // sphere table + grid + tree top, bounded by grid_mode_of / bvh_lds_fit
        hipError_t e = hipFuncSetAttribute((const void*)pinhole_kernel<kR, false, CX>, hipFuncAttributeMaxDynamicSharedMemorySize, max_dyn);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)pinhole_kernel<kR, true, CX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kBvhLdsNodes * sizeof(BvhNode)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    // the shared-origin table of this launch's generator: read by the per-wave cull only
    p.tile_lists = nullptr;
    if (p.sc.n_sph_pad) {
        hipLaunchKernelGGL(pinhole_table_kernel, dim3((p.sc.n_sph_pad + 255) / 256), dim3(256), 0, stream, p, p.prim_table);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        if (p.tile_shift && p.cell_count) {
            p.tile_lists = reinterpret_cast<uint32_t*>(p.prim_table + p.sc.n_sph_pad);
            const unsigned n_tiles = (unsigned)(((uint64_t)p.cell_count + (1ull << p.tile_shift) - 1) >> p.tile_shift);
            hipLaunchKernelGGL(pinhole_tiles_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, stream, p, p.prim_table, p.tile_lists);
            if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        }
    }
    if (stream_mode(p)) {
        hipLaunchKernelGGL((pinhole_kernel<kR, true, CX>), dim3(grid), dim3(kPinThreads), dyn_lds(p), stream, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((pinhole_kernel<kR, false, CX>), dim3(grid), dim3(kPinThreads), dyn_lds(p), stream, p);
    return hipGetLastError();
}

hipError_t launch_pinhole(const TraceParams& p, int n_cu, int device, hipStream_t stream)
{
    return p.sc.n_cx > 0 ? launch_pinhole_t<true>(p, n_cu, device, stream) : launch_pinhole_t<false>(p, n_cu, device, stream);
}

}  // namespace rt
