// rt_paths.hip — Scene::simulate_rays in its own call shape (scene.rs:39-106): explicit rays in, polylines out.
//
// paths_kernel is the Projector kernel's machinery (rt_pinhole.hip) without a generator: 512-thread blocks whose LDS holds the
// scan-path table {p, r} and the uniform grid, one ray per lane, a per-wave ring of continuing rays in global memory.  A step of
// a wave is either
//   * a FRESH step: a batch of 64 consecutive rays of the caller's list.  Ray lists come out of the reference's generators in
//     (i, j) order (scene_config.rs:74-75, 104-105), so the rays of a batch usually form a thin bundle.  The wave bounds the bundle
//     (a cone around it and a slab through it, both measured on the rays themselves: bundle_of) and copies the table entries the
//     bound can touch to a short LDS list (bundle_cull); every ray then runs the reference's own discriminant (sphere.rs:13-18)
//     on the list, exactly as the grid walk does per cell.  A bundle that is not thin, or whose list overflows, walks the grid.
//   * a CONTINUING step: up to 64 rays popped from the ring (rays that left a surface), grid walk as in the Projector kernel.
// Every candidate is confirmed by confirm_batch with the reference's exact sequence, so nothing about a result depends on
// which of the three ways (list, walk, table scan) found it.  Polyline vertices, vertex counts and the optional bounce
// records are written as the path goes (same layout as trace_kernel, rt_kernels.hip).
//
// A call of a few million rays is a few dozen steps per wave, so what a step WAITS for matters as much as what it issues:
// batches are drawn from the interleaved work counters two batches ahead and their rays are loaded one batch ahead, the
// object-index lookup is skipped when nothing reads it, and the ring's stores are only waited for when the ring is read.
// Measured on the 4.1 M-ray call (tools/bench_simulate.py): batches of 64 / 128 / 256 / 512 rays sharing one bound and list:
// 0.47 / 0.54 / 0.63 / 0.90 ms -- a heavy batch (every ray enters the cloud and bounces eight times) is the unit one wave works
// through alone at the end of the launch, and that tail grows with the batch; one work counter instead of 32: 0.69 ms; static
// dealing (wave g takes batches g, g + W, ...): 0.58 ms.
#include "rt_complex.h"
#include "rt_grid.h"

#include <cstdlib>

namespace rt {
namespace {

// In-kernel stopwatch for one region per build (-DRT_DBG_PTIME=n, measurement builds only): lane 0 of every wave adds the
// s_memtime deltas of region n to a sum that the kernel returns in place of the segment total.
#ifdef RT_DBG_PTIME
#define RT_P0(n) if (RT_DBG_PTIME == n) t_reg = __builtin_readcyclecounter();
#define RT_P1(n) if (RT_DBG_PTIME == n && lane == 0) dbg += __builtin_readcyclecounter() - t_reg;
#else
#define RT_P0(n)
#define RT_P1(n)
#endif

constexpr int kPathThreads = 512;
constexpr unsigned kBundleCap = 48;  // list entries of a fresh step; a bundle that needs more walks the grid
constexpr unsigned long long kNoBatch = ~0ull;

// Constants of the general filter (table scan) for one ray; a dead slot fails every test.
__device__ __forceinline__ void path_general_consts(ScanConsts<1>& c, F3 o, F3 d, float a, bool live, const DevScene& sc)
{
    const float rs = 1.0f / __builtin_sqrtf(a);
    c.dh[0] = d * rs;
    c.nod[0] = -dot(o, c.dh[0]);
    const float oo = dot(o, o);
    const float margin = kCullKappaU * (oo + sc.c2max);
    c.s0[0] = live ? (oo - margin) : __builtin_inff();
    c.pp[0] = f3(-2.0f * o.x, -2.0f * o.y, -2.0f * o.z);
    if (!live) {
        c.dh[0] = f3(0.f, 0.f, 0.f);
        c.nod[0] = 0.0f;
        c.pp[0] = f3(0.f, 0.f, 0.f);
    }
}

// The bundle of a fresh step (wave-uniform).  Two bounds, both measured on the rays themselves, so neither depends on how the
// caller ordered them:
//   cone: apex o0 (midway between the first and the last live ray's origin), unit axis ax (between their directions), sine /
//         cosine of a half angle that contains every ray's direction, rho = the largest distance of a ray origin from o0;
//   slab: a unit normal nrm (the plane through o0 that holds the first and the last ray, as well as it can), rho_n = the largest
//         |(o_i - o0) . nrm| and sig = the largest |d^_i . nrm|: a point o_i + s d^_i lies within rho_n + s sig of that plane.
// The rays of a generator column (scene_config.rs:98-105: one origin, targets along a line) are coplanar, which the cone alone
// does not see: it is as thick as the fan is long.  coherent == false: no usable bound.
struct Bundle {
    F3 o0, ax, nrm;
    float s, c, rho, rho2, rho_n, sig;
    bool coherent;
};

__device__ __forceinline__ float lane_value(float v, unsigned l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }

__device__ __forceinline__ F3 ray_o(const RtRay& r) { return f3(r.origin[0], r.origin[1], r.origin[2]); }
__device__ __forceinline__ F3 ray_d(const RtRay& r) { return f3(r.direction[0], r.direction[1], r.direction[2]); }

// The bundle of a batch, measured on its live rays (o, d, a = d . d per lane).  Rays with an all-NaN direction are not `live`
// here (push_nan_rays settles them); every live ray has been admitted by the grid (finite, starts within its error budget).
// Zero directions give NaN unit vectors, which every reduction drops: such rays hit nothing.
__device__ __forceinline__ Bundle bundle_of(F3 o, F3 d, float a, bool live)
{
    Bundle b;
    b.coherent = false;
    const unsigned long long lm = __ballot(live);
    if (!lm) return b;
    const unsigned first = (unsigned)__builtin_ctzll(lm), last = 63u - (unsigned)__builtin_clzll(lm);
    const F3 dh = d * __builtin_amdgcn_rsqf(a);  // unit direction to ~1 ulp
    const F3 oF = f3(lane_value(o.x, first), lane_value(o.y, first), lane_value(o.z, first));
    const F3 oL = f3(lane_value(o.x, last), lane_value(o.y, last), lane_value(o.z, last));
    const F3 dF = f3(lane_value(dh.x, first), lane_value(dh.y, first), lane_value(dh.z, first));
    const F3 dL = f3(lane_value(dh.x, last), lane_value(dh.y, last), lane_value(dh.z, last));
    b.o0 = 0.5f * (oF + oL);
    const F3 am = dF + dL;
    b.ax = am * __builtin_amdgcn_rsqf(dot(am, am));
    // the plane of the first and the last ray: spanned by the axis and by what separates the two (directions of a fan, origins of
    // a parallel sheet); whatever comes out, rho_n and sig below are measured against it
    const F3 sep = (dL - dF) + (oL - oF) * __builtin_amdgcn_rsqf(dot(oL - oF, oL - oF) + 1e-30f);
    const F3 nn = cross(b.ax, sep);
    const float n2 = dot(nn, nn);
    b.nrm = n2 > 1e-12f ? nn * __builtin_amdgcn_rsqf(n2) : f3(0.f, 0.f, 0.f);  // no separation (one ray, identical rays): no slab
    const F3 cr = cross(dh, b.ax), od = o - b.o0;
    const float sine = sgpr(wave_max(live ? __builtin_amdgcn_sqrtf(dot(cr, cr)) : 0.0f));
    const float cosmin = sgpr(wave_min(live ? dot(dh, b.ax) : 1.0f));
    b.rho2 = sgpr(wave_max(live ? dot(od, od) : 0.0f)) * 1.0001f;
    b.rho = __builtin_amdgcn_sqrtf(b.rho2) * 1.0001f;
    b.rho_n = sgpr(wave_max(live ? __builtin_fabsf(dot(od, b.nrm)) : 0.0f)) * 1.0001f + 1e-6f * b.rho;
    b.sig = sgpr(wave_max(live ? __builtin_fabsf(dot(dh, b.nrm)) : 0.0f)) * 1.0001f + 1e-6f;
    // directions within 14.5 degrees of the axis and all in its half space; a zero or non-finite axis fails the compares
    b.coherent = cosmin > 0.5f && sine <= 0.25f && dot(b.ax, b.ax) > 0.5f && dot(b.o0, b.o0) < __builtin_inff() && b.rho2 < __builtin_inff() &&
                 dot(b.nrm, b.nrm) < 2.0f;
    b.s = sine * 1.0001f + 1e-6f;  // rounding of the unit directions and of the products above: a few 1e-7
    b.c = __builtin_amdgcn_sqrtf(1.0f - b.s * b.s) * 0.9999f;
    return b;
}

// May a ray of the bundle reach entry {p, r}?
// Cone: the rays lie within rho of the cone (apex o0, axis ax, half angle theta), so an entry is out of reach when its distance
// from the cone exceeds r_eff + rho.  With q = p - o0, qa = q . ax, perp = the distance of p from the axis,
// perp cos(theta) - qa sin(theta) never exceeds the distance of p from the cone (it is that distance where the nearest cone
// point lies on the mantle, negative inside, and smaller than |q| where the apex is nearest).
// Slab: a ray point within r_eff of p has s <= |q| + rho + r_eff, and lies within rho_n + s sig of the plane (o0, nrm); so the
// entry is out of reach when |q . nrm| > r_eff + rho_n + (|q| + rho + r_eff) sig.
// r_eff covers the rays whose float discriminant is non-negative although they pass outside the sphere: |fl(disc) - disc| <=
// 25 u a L^2 with L = |o - p| <= |q| + rho, i.e. r_eff^2 = r^2 + 25 u L^2; the code allows 64 u * 3 (|q|^2 + rho^2)
// >= 64 u L^2.  The bounds' own arithmetic (a dozen roundings relative to |q| + r + rho) is covered by the 1e-4 relative slop.
// NaN passes.
__device__ __forceinline__ bool bundle_reach(const Bundle& b, F3 q, float qq, float reff)
{
    const float qa = dot(q, b.ax), qn = dot(q, b.nrm);
    // v_sqrt_f32 (1 ulp) serves a bound with 1e-4 of slop; the correctly rounded square root costs fifteen instructions more
    const float perp = __builtin_amdgcn_sqrtf(__builtin_fmaxf(qq - qa * qa, 0.0f));
    const float R = reff + b.rho;
    const float mag = (perp + __builtin_fabsf(qa)) + R;  // >= |q| + r_eff + rho
    const float f = perp * b.c - qa * b.s;
    const bool cone_out = f > R + 1e-4f * mag;
    const bool slab_out = __builtin_fabsf(qn) > (reff + b.rho_n) + mag * (b.sig + 1e-4f);
    return !(cone_out || slab_out);
}
constexpr float kDiscSlop = 64.0f * 3.0f * 0x1p-24f;  // r_eff^2 = r^2 + kDiscSlop (|q|^2 + rho^2), see above
__device__ __forceinline__ bool bundle_pass(const Bundle& b, float4 e)
{
    const F3 q = f3(e.x - b.o0.x, e.y - b.o0.y, e.z - b.o0.z);
    const float qq = dot(q, q);
    const float reff = __builtin_amdgcn_sqrtf(e.w * e.w + kDiscSlop * (qq + b.rho2)) * 1.00001f;
    return bundle_reach(b, q, qq, reff);
}
// The same for a cluster {c, R} of entries (rt_host_grid.h build_clusters: R >= |p_j - c| + r_j for every member j).  A ray point
// within r_eff_j of member j lies within r_eff_j + |p_j - c| of c, and r_eff_j <= r_j + sqrt(kDiscSlop (|q_j|^2 + rho^2)) with
// |q_j| <= |q_c| + R: the cluster is out of reach when the sphere {c, R + sqrt(kDiscSlop ((|q_c| + R)^2 + rho^2))} is.
__device__ __forceinline__ bool bundle_pass_cluster(const Bundle& b, float4 e)
{
    const F3 q = f3(e.x - b.o0.x, e.y - b.o0.y, e.z - b.o0.z);
    const float qq = dot(q, q);
    const float far = __builtin_amdgcn_sqrtf(qq) * 1.00001f + e.w;
    const float reff = (e.w + __builtin_amdgcn_sqrtf(kDiscSlop * (far * far + b.rho2))) * 1.0001f;
    return bundle_reach(b, q, qq, reff);
}

// The table entries (LDS copy of sph_exact: {p, r}, a box as its bounding sphere) the bundle can reach -> list / idx.
// Returns their number; more than kBundleCap: the list is incomplete.
__device__ __forceinline__ unsigned bundle_cull(const Bundle& b, LdsGV4* table, unsigned n_sph, const DevScene& sc, float4* list, uint32_t* idx, unsigned lane)
{
    unsigned n = 0;
    for (unsigned base = 0; base < n_sph; base += 128u) {  // two rows per trip: their LDS reads and square roots overlap
        const unsigned k0 = base + lane, k1 = base + 64u + lane;
        const float4 e0 = grid_sphere(table, k0 < n_sph ? k0 : 0u), e1 = grid_sphere(table, k1 < n_sph ? k1 : 0u);
        const bool p0 = k0 < n_sph && bundle_pass(b, e0), p1 = k1 < n_sph && bundle_pass(b, e1);
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        if (m0 | m1) {
            const unsigned pos0 = n + mbcnt64(m0), pos1 = n + (unsigned)__builtin_popcountll(m0) + mbcnt64(m1);
            if (p0 && pos0 < kBundleCap) {
                list[pos0] = e0;
                idx[pos0] = k0 | ((sc.n_box && sc.sph_box[k0].w != 0.0f) ? 0x8000u : 0u);
            }
            if (p1 && pos1 < kBundleCap) {
                list[pos1] = e1;
                idx[pos1] = k1 | ((sc.n_box && sc.sph_box[k1].w != 0.0f) ? 0x8000u : 0u);
            }
            n += (unsigned)__builtin_popcountll(m0) + (unsigned)__builtin_popcountll(m1);
        }
    }
    return n;
}

// The same list through the clusters: one row of cluster bounds, then the members of the clusters that pass, four clusters (of
// 16 entries) per row -- five rows or so for a thin bundle instead of the table's sixteen.  clu_bound / clu_items are LDS copies.
template <typename SphPtr>
__device__ __forceinline__ unsigned bundle_cull_clusters(const Bundle& b, SphPtr table, LdsGV4* clu_bound, LdsU16* clu_items, unsigned n_clu, const DevScene& sc,
                                                         float4* list, uint32_t* idx, unsigned lane)
{
    unsigned n = 0;
    const unsigned slot = lane >> 4, member = lane & 15u;
    for (unsigned base = 0; base < n_clu; base += 64u) {  // one row of cluster bounds per 64 clusters (1,024 entries)
        const unsigned c_mine = base + lane;
        unsigned long long m = __ballot(c_mine < n_clu && bundle_pass_cluster(b, grid_sphere(clu_bound, c_mine < n_clu ? c_mine : 0u)));
        while (m) {
            unsigned c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c[k] = m ? base + (unsigned)__builtin_ctzll(m) : 0xffffffffu;
                m &= m - 1ull;  // 0 stays 0
            }
            const unsigned mine = slot == 0 ? c[0] : slot == 1 ? c[1] : slot == 2 ? c[2] : c[3];
            // (read as whole words: tools/check_walk_isa.py finds the grid walk by its 16-bit loads, the only ones in these kernels)
            unsigned j = 0xffffu;
            if (mine != 0xffffffffu) j = (reinterpret_cast<LdsU32*>(clu_items)[(mine * kClusterSize + member) >> 1] >> ((member & 1u) * 16u)) & 0xffffu;
            const bool have = j != 0xffffu;
            const float4 e = grid_sphere(table, have ? j : 0u);
            const bool pass = have && bundle_pass(b, e);
            const unsigned long long pm = __ballot(pass);
            if (pm) {
                const unsigned pos = n + mbcnt64(pm);
                if (pass && pos < kBundleCap) {
                    list[pos] = e;
                    idx[pos] = j | ((sc.n_box && sc.sph_box[j].w != 0.0f) ? 0x8000u : 0u);
                }
                n += (unsigned)__builtin_popcountll(pm);
            }
        }
    }
    return n;
}

// Every live ray against the list: the reference's discriminant (sphere.rs:13-18, the operations of sphere_first_t on the same
// operands) decides which pairs go to the candidate queue; a box entry goes there unconditionally (the exact slab test decides).
__device__ __forceinline__ void scan_bundle_list(const float4* list, const uint32_t* idx, unsigned n, F3 o, F3 d, float a, bool live, WaveLds<1>& w,
                                                 const DevScene& sc, unsigned& head, unsigned& tail, unsigned lane)
{
    for (unsigned k = 0; k < n; k += 2) {
        const unsigned k1 = k + 1 < n ? k + 1 : k;
        const float4 s0 = list[k], s1 = list[k1];
        const uint32_t i0 = idx[k], i1 = idx[k1];
        bool p0, p1;
        {
            const F3 lo = o - f3(s0.x, s0.y, s0.z);
            const float half_b = dot(lo, d);
            const float c = dot(lo, lo) - s0.w * s0.w;
            const float disc = half_b * half_b - a * c;
            p0 = live && ((i0 & 0x8000u) || !(disc < 0.0f));
        }
        {
            const F3 lo = o - f3(s1.x, s1.y, s1.z);
            const float half_b = dot(lo, d);
            const float c = dot(lo, lo) - s1.w * s1.w;
            const float disc = half_b * half_b - a * c;
            p1 = live && k + 1 < n && ((i1 & 0x8000u) || !(disc < 0.0f));
        }
        const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
        if (m0) push_candidates<1>(w, sc, m0, p0, 0u, i0 & 0x7fffu, head, tail, lane);
        if (m1) push_candidates<1>(w, sc, m1, p1, 0u, i1 & 0x7fffu, head, tail, lane);
    }
}

// GM = 0: small tables (no grid): LDS holds {p, K}, every step scans it with the general filter.
// GM = 1: LDS holds {p, r}, the grid's cells and its item lists (bundle lists for fresh rays, grid walk for the others) and the
//         clusters of the bundle cull.
// GM = 2: tables beyond an LDS tile (more than 1,024 scan-path entries): nothing is staged; every ray, fresh or continuing, walks the grid
//         in global memory (cells, items and {p, r} come through L2, as in the streamed Projector kernel).
//
// RENDER = true (round 3): the same kernel driven by a ray generator instead of a ray list -- the fast path of renders the Projector
// kernel does not take (ParallelGrid, or a Projector whose target plane runs through its origin).  Work item m = cell * spp + sample of
// the launch; a batch is 64 consecutive items, i.e. the samples of a few neighbouring cells (scene_config.rs:74-75: consecutive cells
// lie along v), whose rays form the same kind of thin bundle as a slice of a ray list.  A path's value and segment count go to its
// cell's integer sums when it ends (flush_cells) instead of vertices to a polyline; results are those of every other
// render kernel, bit for bit (integer sums, ids by cell and sample).
template <int GM, bool RENDER>
__global__ __launch_bounds__(kPathThreads, 4) void paths_kernel(const TraceParams p)
{
    extern __shared__ float4 tiles[];
    __shared__ WaveLds<1> wl[kPathThreads / 64];
    __shared__ float4 list_all[kPathThreads / 64][GM != 0 ? kBundleCap : 1];
    __shared__ uint32_t idx_all[kPathThreads / 64][GM != 0 ? kBundleCap : 1];

    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x >> 6;
#ifdef RT_DBG_TIMELINE  // measurement builds: per-wave timestamps (100 MHz wall clock) into the records buffer, tools/experiments/paths_timeline.py
    unsigned long long tl[4] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0}, tl_steps = 0;
#define RT_TL(k) tl[k] = __builtin_amdgcn_s_memrealtime();
#else
#define RT_TL(k)
#endif
    WaveLds<1>& w = wl[wave];
    DevScene sc = p.sc;
    const unsigned n_pad = sc.n_sph_pad;
    const unsigned n_cells = sc.grid.n_cells;
    const unsigned grid_words = GM == 1 ? n_cells + ((sc.grid.n_items + 1u) >> 1) : 0u;
    uint32_t* grid_lds = reinterpret_cast<uint32_t*>(tiles + n_pad);
    if (GM != 2) {
        const float4* src = GM == 1 ? sc.sph_exact : sc.sph_cull;
        for (unsigned k = threadIdx.x; k < n_pad; k += kPathThreads) tiles[k] = src[k];
        if (GM == 1) {
            const uint32_t* items32 = reinterpret_cast<const uint32_t*>(sc.grid.items);  // padded to whole words by the host
            for (unsigned k = threadIdx.x; k < grid_words; k += kPathThreads) grid_lds[k] = k < n_cells ? sc.grid.cells[k] : items32[k - n_cells];
        }
    }
    // behind the grid: the clusters of the bundle cull, bounds first (16-byte aligned: the grid's words are padded to a multiple of 4)
    // (GM = 2: the clusters are all the dynamic LDS holds)
    const unsigned n_clu = GM != 0 ? sc.n_clu : 0u;
    float4* clu_lds = GM == 2 ? tiles : reinterpret_cast<float4*>(grid_lds + ((grid_words + 3u) & ~3u));
    uint32_t* clu_items_lds = reinterpret_cast<uint32_t*>(clu_lds + n_clu);
    if (GM != 0 && n_clu) {
        for (unsigned k = threadIdx.x; k < n_clu; k += kPathThreads) clu_lds[k] = sc.clu_bound[k];
        const uint32_t* src32 = reinterpret_cast<const uint32_t*>(sc.clu_items);
        for (unsigned k = threadIdx.x; k < n_clu * (kClusterSize / 2); k += kPathThreads) clu_items_lds[k] = src32[k];
    }
    if (GM == 1) sc.sph_exact = tiles;  // confirm and shading read {p, r} from the LDS copy
    // the object index of a scan-path hit is only looked up when something reads it (bounce records; planes to compare with)
#ifdef RT_DBG_TIMELINE
    RtBounceRecord* const records = nullptr;  // the buffer takes the timestamps instead
#else
    RtBounceRecord* const records = RENDER ? nullptr : p.records;
#endif
    const bool need_obj = records != nullptr || sc.n_pl != 0 || (RENDER && p.first_id != nullptr);
    __syncthreads();
    RT_TL(1)

    const unsigned gwave = blockIdx.x * (kPathThreads / 64) + wave;
    uint32_t* ring = p.secq + (size_t)gwave * kSecFields * kSecCap;
    unsigned sec_count = 0;  // wave-uniform
    bool ring_dirty = false;  // entries pushed since the ring was last read (their stores may still be in flight)
    unsigned long long my_segments = 0, dbg = 0;
    const unsigned long long total_rays = RENDER ? (unsigned long long)p.cell_count * p.spp : p.cell_count;  // work items
    const unsigned long long total_batches = (total_rays + 63ull) / 64ull;
    const uint32_t max_b = p.st.max_bounces;
    const uint64_t stride_pts = (uint64_t)max_b + 1;

    // Work: batch b (rays 64 b .. 64 b + 63) belongs to work counter b % kUnitCounters; a wave starts at its own counter and moves
    // on when one runs out, as in the Projector kernel.  `ticket` is a draw in flight (lane 0), `next_batch` the batch whose rays
    // are in `pre` (loaded while the batch before it was traced).
    unsigned ctr = gwave % kUnitCounters, ctr_done = 0;
    unsigned ticket = 0;
    auto draw = [&]() {
        if (lane == 0) ticket = atomicAdd(reinterpret_cast<unsigned*>(p.next_unit + (size_t)ctr * kUnitCounterStride), 1u);
    };
    // the drawn batch, or kNoBatch when every counter has run out; leaves the following draw in flight
    auto resolve = [&]() -> unsigned long long {
        if (ctr_done == kUnitCounters) return kNoBatch;
        unsigned long long b = (unsigned long long)__builtin_amdgcn_readfirstlane(ticket) * kUnitCounters + ctr;
        while (b >= total_batches) {  // this counter has run out: on to the next one (only at the end of the launch)
            ctr = (ctr + 1u) % kUnitCounters;
            if (++ctr_done == kUnitCounters) return kNoBatch;
            draw();
            b = (unsigned long long)__builtin_amdgcn_readfirstlane(ticket) * kUnitCounters + ctr;
        }
        draw();
        return b;
    };
    auto load_batch = [&](unsigned long long b) -> RtRay {
        if (RENDER) return RtRay{};  // the generator makes the rays: nothing to fetch
        const unsigned long long m = b * 64ull + lane;
        return p.rays[(b != kNoBatch && m < total_rays) ? m : 0ull];
    };
    draw();
    unsigned long long next_batch = resolve();
    RtRay pre = load_batch(next_batch);

    for (;;) {
        const bool fresh_left = next_batch != kNoBatch;
        const bool do_sec = sec_count >= 64u || (!fresh_left && sec_count > 0);
        if (!do_sec && !fresh_left) break;
        unsigned long long t_reg = 0;
        (void)t_reg;
        RT_P0(1)  // region 1: the whole step
        RT_P0(2)  // region 2: ray set-up (ray loads / ring pops)
        F3 ro, rd;
        float rior;
        uint32_t rm, rb, rsample = 0;  // rm: the ray's index in the list, or (RENDER) its cell of the launch
        bool live;
        if (!do_sec) {
            // ---- fresh step: batch `next_batch`, whose rays arrived while the previous batch was traced ----
#ifdef RT_DBG_TIMELINE
            tl_steps += 1ull;
#endif
            const unsigned long long m64 = next_batch * 64ull + lane;
            live = m64 < total_rays;
            rb = 0;
            if (RENDER) {
                // item -> (cell, sample); a 64-bit division costs a few hundred instructions, so only launches that need one pay it
                const unsigned long long item = live ? m64 : 0ull;
                if ((total_rays >> 32) == 0) {
                    rm = (uint32_t)item / p.spp;
                    rsample = (uint32_t)item - rm * p.spp;
                } else {
                    rm = (uint32_t)(item / p.spp);
                    rsample = (uint32_t)(item - (unsigned long long)rm * p.spp);
                }
                primary_ray(p, rm, ro, rd, rior);
            } else {
                rm = (uint32_t)m64;
                ro = ray_o(pre);
                rd = ray_d(pre);
                rior = pre.current_ior;
            }
            // The rays of this batch are taken out of `pre` HERE, before the next batch's loads are issued.  Left to itself the
            // compiler sinks these copies to the join behind the loads, where its wait for them (vmcnt counts in order) is a wait
            // for the loads just issued (seen in the ISA: s_waitcnt vmcnt(0) two instructions after the global_load of the next batch).
            asm volatile("" : "+v"(ro.x), "+v"(ro.y), "+v"(ro.z), "+v"(rd.x), "+v"(rd.y), "+v"(rd.z), "+v"(rior) : : "memory");
            next_batch = resolve();        // waits for a draw made a whole batch ago; makes the next one
            pre = load_batch(next_batch);  // in flight while this batch is traced
        } else {
            // ---- continuing step: up to 64 rays from the ring ----
#ifdef RT_DBG_TIMELINE
            tl_steps += 1ull << 16;
#endif
            const unsigned n_pop = sec_count < 64u ? sec_count : 64u;
            const unsigned base = sec_count - n_pop;
            sec_count = base;
            // the ring is global memory that only this wave touches: its stores have to have landed before they are read back
            if (ring_dirty) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            ring_dirty = false;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            live = lane < n_pop;
            const unsigned e = base + (live ? lane : 0u);
            ro = f3(__uint_as_float(ring[0 * kSecCap + e]), __uint_as_float(ring[1 * kSecCap + e]), __uint_as_float(ring[2 * kSecCap + e]));
            rd = f3(__uint_as_float(ring[3 * kSecCap + e]), __uint_as_float(ring[4 * kSecCap + e]), __uint_as_float(ring[5 * kSecCap + e]));
            rior = __uint_as_float(ring[6 * kSecCap + e]);
            rm = ring[7 * kSecCap + e];
            if (RENDER) rsample = ring[8 * kSecCap + e];
            rb = ring[9 * kSecCap + e];
        }
        RT_P1(2)
        RT_P0(3)  // region 3: scan (bound + list / walk / table) and confirm
        const float a = dot(rd, rd);
        w.stash[0][lane] = ro.x; w.stash[1][lane] = ro.y; w.stash[2][lane] = ro.z;
        w.stash[3][lane] = rd.x; w.stash[4][lane] = rd.y; w.stash[5][lane] = rd.z;
        w.stash[6][lane] = a;
        w.best[lane] = kNoHit;
        const bool scan_live = live && !nan_ray(rd);  // all-NaN directions are settled by push_nan_rays
        unsigned head = 0, tail = 0;
        wave_lds_sync();
        push_nan_rays<1>(w, sc, live && !scan_live, 0u, head, tail, lane);

        bool done = false;
        if (GM != 0) {
            // this step's copy of the grid's constants (rt_device.h RT_KERNARG_PTR: read where used, not held through the loop)
            const DevGrid grid = kernarg_grid(offsetof(TraceParams, sc) + offsetof(DevScene, grid));
            // the grid (and the bundle bound) serve finite rays that start within the grid's error budget
            const bool admitted = __ballot(scan_live && !grid_admits(grid, ro, rd)) == 0ull;
            if ((GM == 1 || n_clu != 0u) && admitted && !do_sec) {
                RT_P0(5)  // region 5: the bundle's bound
                const Bundle bd = bundle_of(ro, rd, a, scan_live);
                RT_P1(5)
#if RT_DBG_PATHS == 5
                if (lane == 0 && !bd.coherent) ++dbg;
#endif
                if (bd.coherent) {
                    RT_P0(6)  // region 6: the cull
                    unsigned n;
                    if (GM == 2) n = bundle_cull_clusters(bd, sc.sph_exact, (LdsGV4*)clu_lds, (LdsU16*)clu_items_lds, n_clu, sc, list_all[wave], idx_all[wave], lane);
                    else if (n_clu) n = bundle_cull_clusters(bd, (LdsGV4*)tiles, (LdsGV4*)clu_lds, (LdsU16*)clu_items_lds, n_clu, sc, list_all[wave], idx_all[wave], lane);
                    else n = bundle_cull(bd, (LdsGV4*)tiles, sc.n_sph, sc, list_all[wave], idx_all[wave], lane);
                    RT_P1(6)
#if RT_DBG_PATHS == 3
                    if (lane == 0) dbg += n;
#endif
#if RT_DBG_PATHS == 2
                    if (lane == 0 && n > kBundleCap) ++dbg;
#endif
                    if (n <= kBundleCap) {
                        wave_lds_sync();
                        RT_P0(7)  // region 7: the list scan
                        scan_bundle_list(list_all[wave], idx_all[wave], n, ro, rd, a, scan_live, w, sc, head, tail, lane);
                        RT_P1(7)
                        done = true;
                    }
                }
            }
            if (admitted && !done) {
#ifdef RT_DBG_TIMELINE
                if (!do_sec) tl_steps += 1ull << 32;
#endif
                RT_P0(8)  // region 8: the grid walk
                if (GM == 2) grid_walk<1>(grid, grid.cells, grid.items, sc.sph_exact, ro, rd, a, scan_live, 0u, w, sc, head, tail, lane, dbg);
                else grid_walk<1>(grid, (LdsU32*)grid_lds, (LdsU16*)(grid_lds + n_cells), (LdsGV4*)tiles, ro, rd, a, scan_live, 0u, w, sc, head, tail, lane, dbg);
                RT_P1(8)
                done = true;
            }
        }
        if (!done) {
            ScanConsts<1> c;
            path_general_consts(c, ro, rd, a, scan_live, sc);
            // GM != 0: LDS holds {p, r} or nothing, so this (rare) scan reads {p, K} from global memory with wave-uniform loads
            scan_tile<1, 0, false>(GM != 0 ? sc.sph_cull : tiles, n_pad, nullptr, c, w, sc, head, tail, lane);
        }

        wave_lds_sync();
        RT_P0(9)  // region 9: the last confirm batch
        if (tail != head) confirm_batch<1>(w, sc, head, tail - head, lane);
        wave_lds_sync();
        RT_P1(9)
        RT_P1(3)
        RT_P0(4)  // region 4: resolve, shade, outputs, ring push

        // ---- planes, resolve, scatter (scene.rs:63-101), outputs ----
        bool push = false;
        Shade h;
        h.o_next = ro; h.dn = rd; h.ior = rior;
        // RENDER: a path that ends adds its value and its segments to its cell
        bool fl = false;
        unsigned long long fx = 0, fy = 0, fz = 0;
        if (RENDER) {
            if (live) {
                int sj;
                const unsigned long long kscan = scan_key(sc, w.best[lane], need_obj, sj);
                const unsigned long long key = planes_min_key(sc, ro, rd, kscan);
                ++my_segments;
                const uint64_t path_id = (p.cell_begin + (uint64_t)rm * p.cell_step) * (uint64_t)p.spp + rsample;
                F3 value = f3(0.f, 0.f, 0.f);
                if (rb == 0 && rsample == 0) {  // first hit of the cell's first sample
                    if (p.first_id) p.first_id[rm] = key != kNoHit ? (int32_t)(uint32_t)key : -1;
                    if (p.first_t) p.first_t[rm] = key != kNoHit ? __uint_as_float((uint32_t)(key >> 32)) : __builtin_inff();
                }
                if (key != kNoHit) {
                    if (rb + 1 == max_b) {
                        fl = true;  // bounce budget exhausted: value 0
                    } else {
                        PreHit pre_hit;
                        pre_hit.valid = false;
                        h = shade_hit(sc, ro, rd, rior, key, kernarg_value<uint64_t>(offsetof(TraceParams, seed)), path_id, rb, pre_hit, key == kscan ? sj : -1);
                        push = true;
                    }
                } else {
                    value = sky(rd);
                    fl = true;
                }
                if (fl) {
                    fx = quantize(value.x);
                    fy = quantize(value.y);
                    fz = quantize(value.z);
                }
            }
            // (wave-uniform call; the stash and the best keys are dead from here to the next step's set-up)
            flush_cells(p.fb_q, fl, rm, fx, fy, fz, (unsigned long long)rb + 1ull, reinterpret_cast<uint32_t*>(&w.stash[0][0]), lane);
        } else
        if (live) {
            int sj;
            const unsigned long long kscan = scan_key(sc, w.best[lane], need_obj, sj);
            const unsigned long long key = planes_min_key(sc, ro, rd, kscan);
            ++my_segments;
            const uint64_t path_id = kernarg_value<uint64_t>(offsetof(TraceParams, path_id_base)) + rm;
            float* pt = kernarg_value<float*>(offsetof(TraceParams, points)) + ((uint64_t)rm * stride_pts + rb) * 3;  // vertices rb and rb + 1 of the polyline
            if (rb == 0) {  // scene.rs:45: the path starts with the ray's origin
                pt[0] = ro.x; pt[1] = ro.y; pt[2] = ro.z;
            }
            RtBounceRecord rec;
            bool path_done;
            if (key != kNoHit) {
                PreHit pre_hit;
                pre_hit.valid = false;
                h = shade_hit(sc, ro, rd, rior, key, kernarg_value<uint64_t>(offsetof(TraceParams, seed)), path_id, rb, pre_hit, key == kscan ? sj : -1);
                pt[3] = h.point.x; pt[4] = h.point.y; pt[5] = h.point.z;
                if (records) {
                    rec.obj_id = h.obj; rec.t = h.t;
                    rec.point[0] = h.point.x; rec.point[1] = h.point.y; rec.point[2] = h.point.z;
                    rec.normal[0] = h.normal.x; rec.normal[1] = h.normal.y; rec.normal[2] = h.normal.z;
                    rec.front_face = h.front ? 1u : 0u;
                    rec.dir_out[0] = h.dn.x; rec.dir_out[1] = h.dn.y; rec.dir_out[2] = h.dn.z;
                    rec.ior_out = h.ior;
                    records[(uint64_t)rm * max_b + rb] = rec;
                }
                path_done = rb + 1 == max_b;  // bounce budget exhausted: the path ends at this hit
                push = !path_done;
            } else {
                const F3 far = ro + rd * kernarg_value<float>(offsetof(TraceParams, st) + offsetof(RtSettings, infinity_distance));  // scene.rs:99
                pt[3] = far.x; pt[4] = far.y; pt[5] = far.z;
                if (records) {
                    rec.obj_id = -1; rec.t = __builtin_inff();
                    rec.point[0] = far.x; rec.point[1] = far.y; rec.point[2] = far.z;
                    rec.normal[0] = 0; rec.normal[1] = 0; rec.normal[2] = 0;
                    rec.front_face = 0;
                    rec.dir_out[0] = 0; rec.dir_out[1] = 0; rec.dir_out[2] = 0;
                    rec.ior_out = rior;
                    records[(uint64_t)rm * max_b + rb] = rec;
                }
                path_done = true;
            }
            if (path_done) kernarg_value<uint32_t*>(offsetof(TraceParams, n_points))[rm] = rb + 2;
        }
        // append continuing rays to the ring (wave-uniform control flow); at most 63 + 64 entries are ever queued
        const unsigned long long pm = __ballot(push);
        if (pm) {
            const unsigned e = sec_count + mbcnt64(pm);
            if (push && e < kSecCap) {  // (always true: see above; a store outside the wave's ring must not happen whatever else goes wrong)
                ring[0 * kSecCap + e] = __float_as_uint(h.o_next.x);
                ring[1 * kSecCap + e] = __float_as_uint(h.o_next.y);
                ring[2 * kSecCap + e] = __float_as_uint(h.o_next.z);
                ring[3 * kSecCap + e] = __float_as_uint(h.dn.x);
                ring[4 * kSecCap + e] = __float_as_uint(h.dn.y);
                ring[5 * kSecCap + e] = __float_as_uint(h.dn.z);
                ring[6 * kSecCap + e] = __float_as_uint(h.ior);
                ring[7 * kSecCap + e] = rm;
                if (RENDER) ring[8 * kSecCap + e] = rsample;
                ring[9 * kSecCap + e] = rb + 1;
            }
            sec_count += (unsigned)__builtin_popcountll(pm);
            ring_dirty = true;
        }
        RT_P1(4)
        RT_P1(1)
    }

#ifdef RT_DBG_TIMELINE
    RT_TL(2)
    {
        unsigned long long v = my_segments;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && p.records) {
            unsigned long long* out = reinterpret_cast<unsigned long long*>(p.records) + (size_t)gwave * 8;
            out[0] = tl[0]; out[1] = tl[1]; out[2] = tl[2]; out[3] = tl[2]; out[4] = tl[2];
            out[5] = v;
            out[6] = 1;
            out[7] = tl_steps;  // fresh steps | continuing steps << 16 | walked fresh steps << 32
        }
    }
#endif
    if (p.total_segments) {
#if defined(RT_DBG_PATHS) || defined(RT_DBG_PTIME)
        unsigned long long v = dbg;
#else
        unsigned long long v = my_segments + 0 * dbg;
#endif
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && v) atomicAdd(p.total_segments, v);
    }
}

constexpr size_t kPathLdsPerBlock = 80 * 1024;  // two 512-thread blocks per CU share 160 KB
inline size_t path_grid_bytes(const DevScene& sc)  // cells, items, cluster bounds, cluster items
{
    const DevGrid& g = sc.grid;
    return ((size_t)g.n_cells * 4 + ((size_t)g.n_items + 1) / 2 * 4 + 15) / 16 * 16 + (size_t)sc.n_clu * (sizeof(float4) + kClusterSize * sizeof(uint16_t));
}

template <int GM, bool RENDER>
int paths_blocks_per_cu(size_t dyn)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, paths_kernel<GM, RENDER>, kPathThreads, dyn) != hipSuccess || n < 1) n = 1;
    return n;
}

}  // namespace

// The explicit-ray kernel serves tables that fit an LDS tile and scenes without objects of the per-lane path (those, larger
// tables and the exact-filter mode stay on trace_kernel): ray lists with polylines out, and generator-driven renders (RENDER).
bool paths_kernel_serves(const TraceParams& p)
{
    if (p.scan_mode != 0 || p.sc.n_cx != 0 || getenv("RT_HIP_NO_PATHS")) return false;
    // tables beyond an LDS tile need the grid (mode 2); without one they stay on the general kernel's block-wide tiled scan
    if (p.sc.n_sph_pad > (uint32_t)kTileSpheres && (!p.sc.grid.enabled || getenv("RT_HIP_NO_GRID"))) return false;
    if (p.rays != nullptr) return p.points != nullptr && p.n_points != nullptr;
    return p.fb_q != nullptr && p.spp != 0;  // a render launch: cell_count * spp work items
}

static bool paths_render(const TraceParams& p) { return p.rays == nullptr; }

// 2: grid in global memory (tables beyond an LDS tile, and grids that do not fit into LDS beside the table and the per-wave state);
// 1: {p, r} + grid in LDS; 0: no grid (small table)
uint32_t paths_grid_mode(const TraceParams& p)
{
    if (!p.sc.grid.enabled || getenv("RT_HIP_NO_GRID")) return 0;
    if (p.sc.n_sph_pad > (uint32_t)kTileSpheres) return 2;
    hipFuncAttributes fa{};
    const void* fn = paths_render(p) ? (const void*)paths_kernel<1, true> : (const void*)paths_kernel<1, false>;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return 0;
    const size_t need = fa.sharedSizeBytes + (size_t)p.sc.n_sph_pad * sizeof(float4) + path_grid_bytes(p.sc) + 512;
    return need <= kPathLdsPerBlock && !getenv("RT_HIP_GRID_IN_MEMORY") ? 1u : 2u;  // long item lists: the grid stays in global memory
}

static size_t paths_dyn_lds(const TraceParams& p)
{
    if (p.grid_mode == 2) return (size_t)p.sc.n_clu * (sizeof(float4) + kClusterSize * sizeof(uint16_t));  // the clusters of the bundle cull
    return (size_t)p.sc.n_sph_pad * sizeof(float4) + (p.grid_mode == 1 ? path_grid_bytes(p.sc) : 0);
}

static unsigned paths_grid(const TraceParams& p, int n_cu)
{
    const size_t dyn = paths_dyn_lds(p);
    const uint64_t items = paths_render(p) ? (uint64_t)p.cell_count * p.spp : (uint64_t)p.cell_count;
    const uint64_t want = (items + kPathThreads - 1) / kPathThreads;
    const bool r = paths_render(p);
    const int occ = p.grid_mode == 2   ? (r ? paths_blocks_per_cu<2, true>(dyn) : paths_blocks_per_cu<2, false>(dyn))
                    : p.grid_mode == 1 ? (r ? paths_blocks_per_cu<1, true>(dyn) : paths_blocks_per_cu<1, false>(dyn))
                                       : (r ? paths_blocks_per_cu<0, true>(dyn) : paths_blocks_per_cu<0, false>(dyn));
    uint64_t cap = (uint64_t)n_cu * occ;
    if (p.max_blocks && cap > p.max_blocks) cap = p.max_blocks;  // a caller whose rings hold fewer waves (the fixed workspace)
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    return blocks ? blocks : 1;
}

size_t paths_secq_bytes(const TraceParams& p, int n_cu)
{
    return (size_t)paths_grid(p, n_cu) * (kPathThreads / 64) * kSecFields * kSecCap * sizeof(uint32_t);
}

hipError_t launch_paths(const TraceParams& p, int n_cu, int device, hipStream_t stream)
{
    static bool attr_set_dev[kMaxDevices] = {};  // function attributes are per device; callers hold the library mutex
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!attr_set_dev[device]) {
        const void* fns[6] = {(const void*)paths_kernel<0, false>, (const void*)paths_kernel<1, false>, (const void*)paths_kernel<2, false>,
                              (const void*)paths_kernel<0, true>,  (const void*)paths_kernel<1, true>,  (const void*)paths_kernel<2, true>};
        for (const void* fn : fns) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPathLdsPerBlock);
            if (e != hipSuccess) return e;
        }
        attr_set_dev[device] = true;
    }
    const unsigned grid = paths_grid(p, n_cu);
    const size_t dyn = paths_dyn_lds(p);
    if (paths_render(p)) {
        if (p.grid_mode == 2) hipLaunchKernelGGL((paths_kernel<2, true>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
        else if (p.grid_mode == 1) hipLaunchKernelGGL((paths_kernel<1, true>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
        else hipLaunchKernelGGL((paths_kernel<0, true>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
    } else {
        if (p.grid_mode == 2) hipLaunchKernelGGL((paths_kernel<2, false>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
        else if (p.grid_mode == 1) hipLaunchKernelGGL((paths_kernel<1, false>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
        else hipLaunchKernelGGL((paths_kernel<0, false>), dim3(grid), dim3(kPathThreads), dyn, stream, p);
    }
    return hipGetLastError();
}

}  // namespace rt
