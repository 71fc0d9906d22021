// rt_internal.h — types shared by the C-ABI host code (rt_capi.cpp) and the gfx950 kernels (rt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"

namespace rt {

constexpr int kTileSpheres = 1024;  // largest sphere table staged in LDS (16 KiB of float4); larger scenes are streamed from L2
constexpr int kBlockThreads = 256;  // 4 waves
constexpr uint32_t kUnitSamples = 8;  // samples of one cell traced back to back by one lane: the general kernel's render work unit (the Projector path picks its own, rt_capi.cpp render_unit_samples)
constexpr int kMaxDevices = 64;     // per-device launch state (function attributes, occupancy) is kept in arrays of this size
constexpr int kUnroll = 4;          // spheres per scan step; sphere arrays are padded to a multiple

// ---- objects off the scan path (rt_complex.h): post-order node programs evaluated per lane ----
enum : uint32_t { CX_SPHERE = 0, CX_PLANE = 1, CX_BOX = 2, CX_INFCYL = 3, CX_INFCONE = 4, CX_UNION = 16, CX_INTERSECTION = 17, CX_DIFFERENCE = 18 };
#ifndef RT_CX_MAX_HITS
#define RT_CX_MAX_HITS 16  // (8 / 4 until round 3; 16 / 6 and 32 / 8 measured the same on the template-like scene and on C3: the lists live in private memory)
#define RT_CX_STACK 6
#endif
constexpr int kCxMaxHits = RT_CX_MAX_HITS;  // hits a node's list may hold (checked when the scene is built)
constexpr int kCxStack = RT_CX_STACK;       // list stack depth of the post-order evaluation (checked likewise)
struct CxNode {
    uint32_t kind, pad0, pad1, pad2;
    float4 a;  // sphere: centre, radius | plane: point | box: min | cylinder: axis point, radius | cone: vertex, cos^2
    float4 b;  // plane: normal | box: max | cylinder / cone: axis direction
};
struct CxObject {
    uint32_t first_node, n_nodes;
    int32_t obj_id;
    uint32_t mesh_root;  // root BvhNode of a triangle mesh object, kNoMesh otherwise
    float m[16];     // Transform.transform          (column-major, glam Mat4)
    float minv[16];  // Transform.inverse_transform
};

constexpr uint32_t kNoMesh = 0xffffffffu;
// Triangle meshes (extension): one BVH per mesh, nodes and triangles of all meshes in two flat arrays.
// A 4-wide node holds the padded boxes of its (up to) four children, so one 128-byte load decides all of them and the
// chain of dependent loads per ray is half as long as in a binary tree.  Child reference (w of the lo vector, as bits):
// bit 31 set = leaf, bits 3..30 = first triangle, bits 0..2 = triangle count (1..4); bit 31 clear = index of the child
// node.  A child that does not exist has NaN bounds, which fail every slab test.
struct BvhNode {
    float4 lo[4];  // lo[i].w = child reference
    float4 hi[4];
};
constexpr uint32_t kBvhLeaf = 0x80000000u;
constexpr int kBvhLeafTris = 4;
struct TriDev {
    float4 v0;  // w bits = triangle index in the caller's mesh (tie-break key)
    float4 e1;  // v1 - v0 (f32)
    float4 e2;  // v2 - v0 (f32)
};
constexpr int kBvhStack = 64;
constexpr uint32_t kBvhLdsNodes = 85;  // 1 + 4 + 16 + 64: the top four levels of a 4-wide tree, 10,880 bytes

// Uniform grid over the scan-path spheres / box bounding spheres: continuing rays of the Projector kernel
// walk it cell by cell instead of scanning the whole table.  Built on the host when the scene is created (rt_host_grid.h).
constexpr uint32_t kGridOutside = 0xffffffffu;  // record of the shell around the grid's cells: a ray that steps into it has left the grid
constexpr uint32_t kGridItemPad = 4;            // spare entries behind the item list (>= the entries a walk pass reads at once)
struct DevGrid {
    const uint32_t* cells;   // [(dim x + 2) * (dim y + 2) * (dim z + 2)] first item (bits 0..19) | item count (bits 20..31); x fastest; the outermost layer is kGridOutside
    const uint16_t* items;   // sphere index j (bits 0..14) | bit 15: entry j is a box (no discriminant pre-test); kGridItemPad spare entries at the end
    float lo[3], hi[3];      // bounds of the grid (contain every registered, inflated sphere box)
    float cell, inv_cell;    // cubic cells
    uint32_t dim[3];
    float centre[3];
    float radius;            // half diagonal of the bounds
    float origin_r2;         // rays whose origin lies farther than sqrt(origin_r2) from `centre` take the table scan instead
    float slack_k;           // t slack of a ray = slack_k * (|o - centre| + radius) / |d|
    uint32_t n_cells, n_items;  // array lengths, shell and spare entries included
    uint32_t enabled;        // 0: no grid for this scene (few spheres, non-finite data, oversized lists)
};

// Flattened scene as the kernels see it (all pointers are device memory).
struct DevScene {
    const float4* sph_cull;   // [n_sph_pad] {px, py, pz, K = |p|^2 - r^2}; padding has K = +inf
    const float4* sph_exact;  // [n_sph_pad] {px, py, pz, r}; for a rotation-0 Box: its centre and bounding radius
    const float4* sph_box;    // [n_sph_pad] {half extents, 1} for such a Box, zero for spheres
    const int32_t* sph_obj;   // [n_sph_pad] object-list index of sphere j (increasing in j)
    const uint2* sph_mat;     // [n_sph_pad] obj_mat of sphere j's object (shading a scan-path hit needs no object-indexed load)
    const float4* pl_pos;     // [n_pl] {px, py, pz, 0}
    const float4* pl_nrm;     // [n_pl] {nx, ny, nz, 0} (not normalised, as configured)
    const int32_t* pl_obj;    // [n_pl]
    const float4* obj_geom;   // [n_obj] sphere: {px,py,pz,r}; plane: {px,py,pz,0}
    const float4* obj_aux;    // [n_obj] plane: {nx,ny,nz,0}; scan-path box: {half extents, 0}
    const uint2* obj_mat;     // [n_obj] {material kind, bits of material param}
    const uint32_t* obj_shape;  // [n_obj]
    const int32_t* obj_cx;      // [n_obj] index into cx_objs for objects off the scan path, -1 for scan-path spheres / planes
    const CxNode* cx_nodes;     // node programs of the complex objects
    const CxObject* cx_objs;    // [n_cx]
    const BvhNode* bvh_nodes;   // all meshes
    const TriDev* tris;         // all meshes, BVH leaf order
    uint32_t n_cx;
    uint32_t n_bvh_nodes;  // entries of bvh_nodes
    uint32_t n_sph, n_sph_pad, n_pl, n_obj;
    uint32_t n_box;      // scan-path entries that are boxes (0: sph_box is never read)
    uint32_t first_box;  // the first of them (lowest object index): what a ray with an all-NaN direction hits (rt_device.h nan_ray)
    float c2max;  // >= max_j (|p_j|^2 + r_j^2), rounded up: magnitude bound used by the cull margin
    DevGrid grid;
    // clusters of scan-path entries for the explicit-ray kernel's bundle cull (rt_host_grid.h build_clusters); n_clu == 0: none
    const float4* clu_bound;    // [n_clu] {centre, radius}: radius >= |p_j - centre| + r_j for every member j
    const uint16_t* clu_items;  // [n_clu * 16] entry indices, 0xffff = none
    uint32_t n_clu;
};

// Host-precomputed generator constants (scene_config.rs:70-72, 96-97): the per-component divisions and the
// ParallelGrid normalisation are done once on the host with the same IEEE operations.
struct DevGen {
    uint32_t kind, count_u, count_v;
    float current_ior;
    float a[3], corner[3], u_step[3], v_step[3], dir_n[3];
    // for the Projector's cull (rt_device.h cull_group), rounded up: |m| with m_k = |corner_k| + |a_k| + count_u |u_step_k| +
    // count_v |v_step_k| (the largest magnitude the per-ray target - origin arithmetic can meet), |u_step| and |v_step|
    float cull_M, cull_ul, cull_vl;
};

struct TraceParams {
    DevScene sc;
    const RtRay* rays;  // explicit rays (simulate) or nullptr (generator mode)
    DevGen gen;
    uint32_t spp;
    uint32_t unit_samples;    // samples per work unit (render); 1 for simulate
    uint32_t units_per_cell;  // ceil(spp / unit_samples)
    uint32_t cell_count;  // number of work items m (cells or rays), < 2^32
    uint64_t cell_begin, cell_step;
    uint64_t path_id_base;
    RtSettings st;
    uint64_t seed;
    unsigned long long* fb_q;  // [cell_count][4] = {x, y, z fixed-point sums (2^-24), segment count}, zeroed by the launcher
    int32_t* first_id;
    float* first_t;
    uint32_t* seg_count;  // written by the finalize kernel from fb_q[cell][3]; the trace kernels only test it for null
    float* points;
    uint32_t* n_points;
    RtBounceRecord* records;
    unsigned long long* total_segments;  // accumulated with atomics; zeroed by the launcher
    unsigned long long* next_unit;       // work counters (kUnitCounterBytes, zeroed by the launcher); the general kernel uses the first
    int scan_mode;                       // 0 = conservative cull, 1 = exact discriminant filter
    // pinhole kernel only
    uint32_t* secq;     // per-wave rings of continuing rays: [grid waves][kSecFields][kSecCap] dwords
    float prim_margin;  // >= 64 u * max_j(|p_j - O|^2 + r_j^2): margin of the shared-origin filter
    float prim_rho;     // (3G + 7M) / d_min of the Projector: rounding of the affine numerator relative to |target - O|
    float4* prim_table; // {A, B, C, 0} / sK per sphere, rebuilt per launch; read by the per-wave cull
    uint32_t bvh_lds_nodes;  // pinhole kernel: the first this-many BvhNodes are copied into LDS (set by launch_pinhole)
    uint32_t grid_mode;      // pinhole kernel: 0 no grid, 1 grid in LDS, 2 grid in global memory (pinhole_grid_mode, set before sizing)
    uint32_t tile_shift;     // pinhole kernel: a cull tile is 2^tile_shift consecutive work items (pinhole_tile_shift); 0 = no tile lists
    uint32_t* tile_lists;    // set by launch_pinhole: behind the shared-origin table in the same buffer
    uint32_t max_blocks;     // pinhole kernel: launch at most this many blocks (0: as many as are resident); the fixed workspace's rings hold 512
};

// Work counters of the Projector kernel: batch b of 64 units belongs to counter b % kUnitCounters; one 128-byte line each.
constexpr unsigned kUnitCounters = 32;
constexpr unsigned kUnitCounterStride = 16;  // in 64-bit words
constexpr size_t kUnitCounterBytes = (size_t)kUnitCounters * kUnitCounterStride * sizeof(unsigned long long);  // 4,096

constexpr uint32_t kClusterMaxEntries = 16384;  // 1,024 clusters = 48 KB of LDS in the explicit-ray kernel's global-grid mode
constexpr uint32_t kClusterSize = 16;  // entries per cluster of the explicit-ray kernel's bundle cull (rt_host_grid.h build_clusters)

constexpr unsigned kSecCap = 256;    // entries per wave ring (a primary step adds <= 64 R to a residue < 64 R; R <= 2 rays per lane)
constexpr unsigned kSecFields = 10;  // o.xyz d.xyz ior cell sample bounce

// ---- fixed workspace of rt_hip_render_dev_noalloc (rt_scratch.hip): one static device array, carved into regions ----
constexpr size_t kNoAllocMaxCells = 1u << 19;       // cells per call: 32 B of accumulators each
constexpr size_t kNoAllocMaxSpheres = 1u << 14;     // scan-path spheres (streamed shared-origin table, 16 B each)
constexpr size_t kNoAllocHeaderBytes = 4096;        // work counters (kUnitCounterBytes)
constexpr size_t kNoAllocTableBytes = kNoAllocMaxSpheres * sizeof(float4);
constexpr size_t kNoAllocRingBytes = (size_t)512 * 8 * kSecFields * kSecCap * sizeof(uint32_t);  // 512 resident blocks x 8 waves (256 CUs x 2)
constexpr size_t kNoAllocFbBytes = kNoAllocMaxCells * 4 * sizeof(unsigned long long);
constexpr size_t kNoAllocScratchBytes = kNoAllocHeaderBytes + kNoAllocTableBytes + kNoAllocRingBytes + kNoAllocFbBytes;  // 58,986,496 B (56.3 MiB)
static_assert(kNoAllocHeaderBytes >= kUnitCounterBytes && kNoAllocHeaderBytes % 256 == 0 && kNoAllocTableBytes % 256 == 0 && kNoAllocRingBytes % 256 == 0, "regions stay 256-byte aligned");
// Device address of the calling thread's current device's copy of the array (hipGetSymbolAddress; no allocation).
hipError_t noalloc_scratch_address(void** out);

// Launches the trace kernel on `stream` of `device` (the current device; per-device launch state is keyed by it).
hipError_t launch_trace(const TraceParams& p, int n_cu, int device, hipStream_t stream);
// Shared-origin (Projector) specialisation. secq must hold pinhole_secq_bytes(), prim_table pinhole_table_bytes() (table + tile lists).
hipError_t launch_pinhole(const TraceParams& p, int n_cu, int device, hipStream_t stream);
size_t pinhole_secq_bytes(const TraceParams& p, int n_cu);
size_t pinhole_table_bytes(const TraceParams& p);
uint32_t pinhole_tile_shift(const TraceParams& p);  // cull tiles for this launch (TraceParams::tile_shift), 0 when they would not pay
uint32_t pinhole_grid_mode(const TraceParams& p);  // how this launch's continuing rays find their spheres (TraceParams::grid_mode)
// Explicit rays in, polylines out on the LDS-grid machinery (rt_paths.hip); secq must hold paths_secq_bytes(), grid_mode = paths_grid_mode().
bool paths_kernel_serves(const TraceParams& p);
uint32_t paths_grid_mode(const TraceParams& p);
size_t paths_secq_bytes(const TraceParams& p, int n_cu);
hipError_t launch_paths(const TraceParams& p, int n_cu, int device, hipStream_t stream);
hipError_t launch_finalize(const unsigned long long* fb_q, float* fb, uint32_t* seg_count, uint64_t n_cells, uint32_t spp, hipStream_t stream);

}  // namespace rt
