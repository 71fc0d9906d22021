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
#include "rt_complex.h"
#include "rt_grid.h"

#include <cstdlib>

namespace rt {

namespace {

// STREAM = false: the sphere table (<= kTileSpheres) is staged in LDS once per block.
// STREAM = true : larger scenes; every wave streams the table from global memory / L2 (scan_tile<.., PF = true>).
// Waves never synchronise with each other after the table is staged.
// CX = the scene holds objects of the per-lane complex path (rt_complex.h); kept out of the other instantiations
// so they carry no private-memory frame.
template <int R, int MODE, bool STREAM, bool CX>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel(const TraceParams p)
{
    __shared__ float4 tile[STREAM ? 1 : kTileSpheres];
    __shared__ WaveLds<R> wl[kBlockThreads / 64];
    __shared__ float4 stage_all[STREAM ? (kBlockThreads / 64) * 128 : 1];

    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x >> 6;
    WaveLds<R>& w = wl[wave];
    float4* stage = stage_all + (STREAM ? wave * 128 : 0);
    const DevScene& sc = p.sc;

    if (!STREAM) {
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
    const unsigned long long total_batches = (total_units + 63ull) / 64ull;
    unsigned long long chunk_next = 0, chunk_end = 0;  // the wave's current batch of work units (wave-uniform)
    unsigned ctr = (blockIdx.x * (kBlockThreads / 64) + wave) % kUnitCounters, ctr_done = 0;
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
                    // batches of 64 units from kUnitCounters interleaved work counters (see rt_pinhole.hip: one counter for the
                    // whole chip serialises large launches); lanes a depleted batch cannot serve are served from the next one
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
                    if (need_unit && mbcnt64(need) < take) {
                        unsigned long long unit = base + mbcnt64(need);
                        {
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
        if (wave_done) break;

        // ---- 2. per-segment constants ----
        ScanConsts<R> c;
        bool scan_live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const F3 o = s[r].o, d = s[r].d;
            const float a = dot(d, d);
            const unsigned slot = r * 64 + lane;
            w.stash[0][slot] = o.x; w.stash[1][slot] = o.y; w.stash[2][slot] = o.z;
            w.stash[3][slot] = d.x; w.stash[4][slot] = d.y; w.stash[5][slot] = d.z;
            w.stash[6][slot] = a;
            w.best[slot] = kNoHit;
            float rs = 1.0f / __builtin_sqrtf(a);
            c.dh[r] = d * rs;
            c.nod[r] = -dot(o, c.dh[r]);
            float oo = dot(o, o);
            float margin = kCullKappaU * (oo + sc.c2max);
            scan_live[r] = s[r].live && !(MODE == 0 && nan_ray(d));  // all-NaN directions are settled by push_nan_rays
            c.s0[r] = scan_live[r] ? (oo - margin) : __builtin_inff();
            c.pp[r] = f3(-2.0f * o.x, -2.0f * o.y, -2.0f * o.z);
            if (!scan_live[r]) {  // q = -inf: fails the filter
                c.dh[r] = f3(0.f, 0.f, 0.f);
                c.nod[r] = 0.0f;
                c.pp[r] = f3(0.f, 0.f, 0.f);
            }
            c.o[r] = o; c.d[r] = d; c.a[r] = a; c.live[r] = s[r].live;
        }
        unsigned head = 0, tail = 0;
        wave_lds_sync();
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) push_nan_rays<R>(w, sc, s[r].live && !scan_live[r], r * 64u, head, tail, lane);
        }

        // ---- 3. scan the sphere list, or walk the grid over it when every ray of the wave may ----
        bool walk = MODE == 0 && p.grid_mode != 0;
        if (walk) {
#pragma unroll
            for (int r = 0; r < R; ++r) walk &= __ballot(scan_live[r] && !grid_admits(sc.grid, s[r].o, s[r].d)) == 0ull;
        }
        if (walk) {
            unsigned long long unused = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
                grid_walk<R>(sc.grid, sc.grid.cells, sc.grid.items, sc.sph_exact, s[r].o, s[r].d, c.a[r], scan_live[r], r * 64u, w, sc, head, tail, lane, unused);
        } else if (STREAM)
            scan_tile<R, (MODE == 0 ? 0 : 2), true>(sc.sph_cull, sc.n_sph_pad, stage, c, w, sc, head, tail, lane);
        else
            scan_tile<R, (MODE == 0 ? 0 : 2), false>(tile, sc.n_sph_pad, stage, c, w, sc, head, tail, lane);

        // ---- 4. drain the queue ----
        wave_lds_sync();
        if (tail != head) confirm_batch<R>(w, sc, head, tail - head, lane);
        wave_lds_sync();

        // ---- 5. planes, resolve, scatter ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!s[r].live) continue;
            const unsigned slot = r * 64 + lane;
            int sj;
            const unsigned long long kscan = scan_key(sc, w.best[slot], true, sj);
            const F3 o = s[r].o, d = s[r].d;
            unsigned long long key = planes_min_key(sc, o, d, kscan);
            PreHit pre;
            pre.valid = false;
            if (CX) {
                CxHit ch;
                int ci = -1;
                key = cx_min_key(sc, BvhLds{nullptr, 0u, nullptr, 0u}, o, d, key, ch, ci);  // no LDS tree top in this kernel
                if (ci >= 0) {
                    pre.valid = true;
                    pre.front = ch.ff != 0;
                    cx_to_world(sc.cx_objs[ci], ch, pre.point, pre.normal);
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
                const Shade sh = shade_hit(sc, o, d, s[r].ior, key, p.seed, path_id, b, pre, key == kscan ? sj : -1);
                const int32_t obj = sh.obj;
                const float t = sh.t;
                const F3 point = sh.point, normal = sh.normal, dn = sh.dn;
                const bool front = sh.front;
                const float ior = sh.ior;
                s[r].d = dn;
                s[r].ior = ior;
                s[r].o = sh.o_next;
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
                s[r].accx += quantize(value.x);
                s[r].accy += quantize(value.y);
                s[r].accz += quantize(value.z);
                if (p.n_points) p.n_points[s[r].m] = s[r].np;
                ++s[r].sample;
                if (s[r].sample == s[r].sample_end) {
                    if (p.fb_q) {  // {x, y, z, segments} per cell (rt_device.h flush_cells)
                        unsigned long long* q = p.fb_q + (uint64_t)s[r].m * 4;
                        if (s[r].accx) atomicAdd(q + 0, s[r].accx);
                        if (s[r].accy) atomicAdd(q + 1, s[r].accy);
                        if (s[r].accz) atomicAdd(q + 2, s[r].accz);
                        atomicAdd(q + 3, (unsigned long long)s[r].segs);
                    }
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

// fb = sum_q / (spp * 2^24), IEEE f64 division rounded once to f32; seg_count = the cell's fourth accumulator
__global__ void finalize_kernel(const unsigned long long* __restrict__ q, float* __restrict__ fb, uint32_t* __restrict__ seg_count, uint64_t n_cells,
                                double denom)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one lane per accumulator word
    if (i >= n_cells * 4) return;
    const uint64_t cell = i >> 2;
    const unsigned c = (unsigned)(i & 3);
    const unsigned long long v = q[i];
    if (c < 3) fb[cell * 3 + c] = (float)((double)v / denom);
    else if (seg_count) seg_count[cell] = (uint32_t)v;
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

template <int MODE, bool STREAM, bool CX>
static hipError_t launch_trace_t(const TraceParams& p, int n_cu, int device, hipStream_t stream)
{
    // Two rays per lane share the wave-uniform work of a table scan; with objects of the per-lane path (CSG trees, rotated shapes,
    // meshes) a lane's time goes into those, one ray after the other, and the calls that bring them are small: one ray per lane
    // spreads them over twice the waves (template-like scene, 1,369 rays: 1.83 -> see DESIGN.md §7).
    constexpr int R = CX ? 1 : 2;
    static int occ_dev[kMaxDevices] = {};  // per device (the query runs on the current device); callers hold the library mutex
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (!occ_dev[device]) occ_dev[device] = blocks_per_cu(trace_kernel<R, MODE, STREAM, CX>);
    const int occ = occ_dev[device];
    // persistent grid: exactly the resident capacity (every block pulls units until the counter runs out),
    // never more blocks than the work can feed
    const uint64_t slots_per_block = (uint64_t)kBlockThreads * R;
    const uint64_t want = ((uint64_t)p.cell_count * p.units_per_cell + slots_per_block - 1) / slots_per_block;
    const uint64_t cap = (uint64_t)n_cu * occ;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((trace_kernel<R, MODE, STREAM, CX>), dim3(blocks), dim3(kBlockThreads), 0, stream, p);
    return hipGetLastError();
}

// Scenes from this many scan-path entries on walk the grid (read from global memory in this kernel); below it the LDS-resident
// table scan is at least as fast.
constexpr uint32_t kTraceGridMinSpheres = 256;

hipError_t launch_trace(const TraceParams& p_in, int n_cu, int device, hipStream_t stream)
{
    TraceParams p = p_in;
    p.grid_mode = (p.sc.grid.enabled && p.sc.n_sph >= kTraceGridMinSpheres && !getenv("RT_HIP_NO_GRID")) ? 2u : 0u;
    const bool stream_table = p.sc.n_sph_pad > (uint32_t)kTileSpheres;
    const bool cx = p.sc.n_cx > 0;
    if (p.scan_mode == 0) {
        if (cx) return stream_table ? launch_trace_t<0, true, true>(p, n_cu, device, stream) : launch_trace_t<0, false, true>(p, n_cu, device, stream);
        return stream_table ? launch_trace_t<0, true, false>(p, n_cu, device, stream) : launch_trace_t<0, false, false>(p, n_cu, device, stream);
    }
    if (cx) return stream_table ? launch_trace_t<1, true, true>(p, n_cu, device, stream) : launch_trace_t<1, false, true>(p, n_cu, device, stream);
    return stream_table ? launch_trace_t<1, true, false>(p, n_cu, device, stream) : launch_trace_t<1, false, false>(p, n_cu, device, stream);
}

hipError_t launch_finalize(const unsigned long long* fb_q, float* fb, uint32_t* seg_count, uint64_t n_cells, uint32_t spp, hipStream_t stream)
{
    const unsigned threads = 256;
    const unsigned blocks = (unsigned)((n_cells * 4 + threads - 1) / threads);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(threads), 0, stream, fb_q, fb, seg_count, n_cells, (double)spp * 16777216.0);
    return hipGetLastError();
}

}  // namespace rt
