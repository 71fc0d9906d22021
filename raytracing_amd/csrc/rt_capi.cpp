// rt_capi.cpp — C-ABI entry points of librt_hip.so (include/rt_hip.h): context, scene flattening
// (the GPU-side counterpart of SceneConfig -> Scene, scene_config.rs:22-52 + object_config.rs:16-30),
// buffer management and kernel launches.  There is no CPU fallback: every entry point either runs the
// gfx950 kernels or fails with an error code.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "rt_host_bvh.h"
#include "rt_host_grid.h"
#include "rt_host_shapes.h"
#include "rt_internal.h"

namespace {

thread_local std::string g_err;
std::mutex g_mu;

struct Context {
    bool ready = false;
    int device = -1;
    int n_cu = 0;
    unsigned long long* counters = nullptr;  // ring of per-launch work counters
    unsigned next_counter = 0;
    unsigned long long* fb_q = nullptr;  // fixed-point framebuffer scratch (grow-only)
    size_t fb_q_bytes = 0;
    uint32_t* secq = nullptr;  // continuing-ray rings of the pinhole kernel (grow-only)
    size_t secq_bytes = 0;
    float4* prim_table = nullptr;  // streamed shared-origin table (grow-only)
    size_t prim_table_bytes = 0;
    // The scratch above is shared by every call on the context.  Calls are serialised on the host by g_mu; on the device a
    // call enqueued on another stream than its predecessor first waits for `last_done` (recorded behind every enqueue).
    unsigned char* noalloc_scratch = nullptr;  // this device's copy of rt_scratch.hip's static array (hipGetSymbolAddress at init)
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    // device staging of the host-buffer entry points (rays in, polylines out): grow-only up to kStageKeepBytes, so that a call at
    // the reference's own sizes (a few rays) costs no hipMalloc / hipFree (each hipFree waits for the device); guarded by g_stage_mu
    unsigned char* stage = nullptr;
    size_t stage_bytes = 0;
};
Context g_ctx;
std::mutex g_stage_mu;  // taken before g_mu, for the whole of a host-buffer call
constexpr size_t kStageKeepBytes = (size_t)256 << 20;  // larger calls allocate and free their own buffers
constexpr unsigned kCounterRing = 64;  // per-launch sets of work counters (kUnitCounterBytes each), used round-robin

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char* what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return RT_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return fail_hip(e_, #expr); \
    } while (0)

template <class T>
hipError_t upload(const std::vector<T>& v, const T** out)
{
    *out = nullptr;
    size_t n = v.size() ? v.size() : 1;
    T* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, n * sizeof(T));
    if (e != hipSuccess) return e;
    if (v.size()) {
        e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return e;
        }
    }
    *out = d;
    return hipSuccess;
}

}  // namespace

struct rt_scene {
    rt::DevScene dev{};
    std::vector<void*> allocations;
    std::vector<float4> host_spheres;  // {p, r} per sphere, for per-call constants that depend on the generator
};

namespace {

void free_scene(rt_scene* s)
{
    if (!s) return;
    for (void* p : s->allocations) (void)hipFree(p);
    delete s;
}

// scene_config.rs:70-72 / 96-97 and the ParallelGrid normalisation (scene_config.rs:72), in IEEE f32.
rt::DevGen make_dev_gen(const RtRayGenerator& g)
{
    rt::DevGen d{};
    d.kind = g.kind;
    d.count_u = g.count_u;
    d.count_v = g.count_v;
    d.current_ior = g.current_ior;
    const float cu = (float)g.count_u, cv = (float)g.count_v;
    for (int k = 0; k < 3; ++k) {
        d.a[k] = g.a[k];
        d.corner[k] = g.corner[k];
        d.u_step[k] = g.u[k] / cu;
        d.v_step[k] = g.v[k] / cv;
    }
    const float x = g.direction[0], y = g.direction[1], z = g.direction[2];
    const float len = std::sqrt((x * x + y * y) + z * z);
    const float rcp = 1.0f / len;
    d.dir_n[0] = x * rcp;
    d.dir_n[1] = y * rcp;
    d.dir_n[2] = z * rcp;
    double m2 = 0.0, ul2 = 0.0, vl2 = 0.0;
    for (int k = 0; k < 3; ++k) {
        const double m = (std::fabs((double)d.corner[k]) + std::fabs((double)d.a[k])) + ((double)cu * std::fabs((double)d.u_step[k]) + (double)cv * std::fabs((double)d.v_step[k]));
        m2 += m * m;
        ul2 += (double)d.u_step[k] * d.u_step[k];
        vl2 += (double)d.v_step[k] * d.v_step[k];
    }
    d.cull_M = (float)(std::sqrt(m2) * 1.001);  // NaN / inf data: the cull then keeps every sphere
    d.cull_ul = (float)(std::sqrt(ul2) * 1.001);
    d.cull_vl = (float)(std::sqrt(vl2) * 1.001);
    return d;
}

// HIP's current device is per host thread: every entry point that touches the device selects the context's device first
// (a caller may legally call from a thread other than the one that ran rt_hip_init).  Called with g_mu held.
int select_device()
{
    if (!g_ctx.ready) return fail(RT_ERR_NOT_INIT, "rt_hip_init() has not been called");
    HIP_TRY(hipSetDevice(g_ctx.device));
    return RT_OK;
}

int check_common(const rt_scene* scene, RtSettings st)
{
    if (int rc = select_device()) return rc;
    if (!scene) return fail(RT_ERR_INVALID, "scene is null");
    if (!(st.max_bounces < (1u << 20))) return fail(RT_ERR_INVALID, "max_bounces out of range");
    return RT_OK;
}

// Device-side ordering between consecutive calls that use different streams (the context's scratch is shared).
int order_after_previous(hipStream_t stream)
{
    if (g_ctx.last_valid && g_ctx.last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, g_ctx.last_done, 0));
    return RT_OK;
}
int mark_enqueued(hipStream_t stream)
{
    HIP_TRY(hipEventRecord(g_ctx.last_done, stream));
    g_ctx.last_stream = stream;
    g_ctx.last_valid = true;
    return RT_OK;
}

// cell_begin + (cell_count - 1) * cell_step < n_cells without overflow (cell_count >= 1, cell_step >= 1)
bool cell_range_ok(uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, uint64_t n_cells)
{
    return cell_begin < n_cells && (cell_count - 1) <= (n_cells - 1 - cell_begin) / cell_step;
}

// Argument checks shared by rt_hip_render and rt_hip_render_dev (everything but the buffers).
int check_render_args(const RtRayGenerator* gen, uint32_t spp, uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step)
{
    if (!gen) return fail(RT_ERR_INVALID, "gen must not be null");
    if (gen->kind > RT_GEN_PROJECTOR) return fail(RT_ERR_INVALID, "unknown generator kind");
    if (gen->count_u == 0 || gen->count_v == 0 || spp == 0) return fail(RT_ERR_INVALID, "count_u, count_v and spp must be > 0");
    if (cell_count == 0) return RT_OK;
    if (cell_count > 0x7fffffffull) return fail(RT_ERR_INVALID, "cell_count exceeds 2^31-1 per call");
    if (cell_step == 0) return fail(RT_ERR_INVALID, "cell_step must be > 0");
    if (!cell_range_ok(cell_begin, cell_count, cell_step, (uint64_t)gen->count_u * gen->count_v))
        return fail(RT_ERR_INVALID, "cell range exceeds count_u*count_v");
    return RT_OK;
}

int g_scan_mode = 0;

// Projector + single LDS tile + default filter: the shared-origin kernel; everything else: the general kernel.
bool use_pinhole(const rt::TraceParams& p)
{
    if (getenv("RT_HIP_NO_PINHOLE")) return false;  // tuning / A-B knob
    // cell indices must be exact in f32 (the shared-origin filter and its cull work on them as floats)
    return !p.rays && p.gen.kind == RT_GEN_PROJECTOR && p.scan_mode == 0 && p.fb_q != nullptr && p.gen.count_u <= (1u << 24) &&
           p.gen.count_v <= (1u << 24);
}

// Rounding budget of the shared-origin filter's affine numerator: rho = (3G + 7M) / d_min, with
// G = |corner - O| + |u| + |v|, M = |corner| + |u| + |v| + |O| and d_min = distance from O to the target plane (a lower
// bound of every |target - O|).  Not finite (plane through O, degenerate parallelogram) => the general kernel runs instead.
double pinhole_rho(const rt::DevGen& g)
{
    auto norm = [](const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    double O[3], c[3], co[3], u[3], v[3], n[3];
    for (int k = 0; k < 3; ++k) {
        O[k] = g.a[k];
        c[k] = g.corner[k];
        co[k] = c[k] - O[k];
        u[k] = (double)g.u_step[k] * g.count_u;
        v[k] = (double)g.v_step[k] * g.count_v;
    }
    n[0] = u[1] * v[2] - u[2] * v[1];
    n[1] = u[2] * v[0] - u[0] * v[2];
    n[2] = u[0] * v[1] - u[1] * v[0];
    const double nn = norm(n);
    const double dmin = std::fabs(co[0] * n[0] + co[1] * n[1] + co[2] * n[2]) / nn;
    const double G = norm(co) + norm(u) + norm(v), M = norm(c) + norm(u) + norm(v) + norm(O);
    return (3.0 * G + 7.0 * M) / dmin;  // NaN / inf when nn == 0 or dmin == 0
}

// Device scratch a launch uses beyond the caller's inputs and outputs.
//   fixed == false: the context's grow-only hipMalloc'ed buffers (every entry point but the no-alloc one);
//   fixed == true : regions of the static device array of rt_scratch.hip; the byte counts are limits, nothing is allocated.
struct Workspace {
    bool fixed;
    unsigned long long* next_unit;
    unsigned long long* fb_q;
    size_t fb_q_bytes;
    uint32_t* secq;
    size_t secq_bytes;
    float4* prim_table;
    size_t prim_table_bytes;
};

Workspace context_workspace()
{
    Workspace w{};
    w.fixed = false;
    w.next_unit = g_ctx.counters + (size_t)(g_ctx.next_counter++ % kCounterRing) * (rt::kUnitCounterBytes / sizeof(unsigned long long));
    w.fb_q = g_ctx.fb_q;
    w.fb_q_bytes = g_ctx.fb_q_bytes;
    w.secq = g_ctx.secq;
    w.secq_bytes = g_ctx.secq_bytes;
    w.prim_table = g_ctx.prim_table;
    w.prim_table_bytes = g_ctx.prim_table_bytes;
    return w;
}

Workspace fixed_workspace()
{
    unsigned char* base = g_ctx.noalloc_scratch;
    Workspace w{};
    w.fixed = true;
    w.next_unit = reinterpret_cast<unsigned long long*>(base);
    w.prim_table = reinterpret_cast<float4*>(base + rt::kNoAllocHeaderBytes);
    w.prim_table_bytes = rt::kNoAllocTableBytes;
    w.secq = reinterpret_cast<uint32_t*>(base + rt::kNoAllocHeaderBytes + rt::kNoAllocTableBytes);
    w.secq_bytes = rt::kNoAllocRingBytes;
    w.fb_q = reinterpret_cast<unsigned long long*>(base + rt::kNoAllocHeaderBytes + rt::kNoAllocTableBytes + rt::kNoAllocRingBytes);
    w.fb_q_bytes = rt::kNoAllocFbBytes;
    return w;
}

// Make sure a scratch buffer holds `need` bytes: grow the context's buffer, or refuse when the workspace is fixed.
template <class T>
int ensure_scratch(Workspace& w, T*& buf, size_t& have, T*& ctx_buf, size_t& ctx_have, size_t need, const char* what)
{
    if (need <= have) return RT_OK;
    if (w.fixed)
        return fail(RT_ERR_UNSUPPORTED, std::string("rt_hip_render_dev_noalloc: ") + what + " needs " + std::to_string(need) +
                                            " bytes, the fixed workspace holds " + std::to_string(have));
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(ctx_buf);
    ctx_buf = nullptr;
    ctx_have = 0;
    HIP_TRY(hipMalloc((void**)&ctx_buf, need));
    ctx_have = need;
    buf = ctx_buf;
    have = need;
    return RT_OK;
}

int enqueue(rt::TraceParams& p, unsigned long long* d_total, hipStream_t stream, const rt_scene* scene, Workspace& ws)
{
    p.next_unit = ws.next_unit;
    p.total_segments = d_total;
    HIP_TRY(hipMemsetAsync(p.next_unit, 0, rt::kUnitCounterBytes, stream));
    if (d_total) HIP_TRY(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream));
    bool pinhole = use_pinhole(p);
    if (pinhole) {
        const double rho = pinhole_rho(p.gen);
        if (rho < 1e4) p.prim_rho = (float)(rho * (1.0 + 1e-6));
        else pinhole = false;  // also NaN
    }
    if (pinhole) {
        // margin of the shared-origin filter: 64 u * max_j(|p_j - O|^2 + r_j^2), rounded up
        double c2 = 0.0;
        for (const float4& s : scene->host_spheres) {
            const double x = (double)s.x - p.gen.a[0], y = (double)s.y - p.gen.a[1], z = (double)s.z - p.gen.a[2];
            const double v = x * x + y * y + z * z + (double)s.w * s.w;
            if (v > c2) c2 = v;
        }
        p.prim_margin = (float)(c2 * 64.0 * 0x1p-24 * (1.0 + 1e-6));
        p.grid_mode = rt::pinhole_grid_mode(p);  // decides the LDS layout, hence the resident blocks the rings are sized for
        // the fixed workspace's rings hold 512 blocks of 8 waves: a device or build with more resident blocks launches 512
        p.max_blocks = ws.fixed ? (uint32_t)(rt::kNoAllocRingBytes / (8 * rt::kSecFields * rt::kSecCap * sizeof(uint32_t))) : 0u;
        p.tile_shift = ws.fixed ? 0 : rt::pinhole_tile_shift(p);  // the fixed workspace has no room for tile lists: its calls cull the whole table
        if (int rc = ensure_scratch(ws, ws.secq, ws.secq_bytes, g_ctx.secq, g_ctx.secq_bytes, rt::pinhole_secq_bytes(p, g_ctx.n_cu),
                                    "the continuing-ray rings"))
            return rc;
        p.secq = ws.secq;
        if (int rc = ensure_scratch(ws, ws.prim_table, ws.prim_table_bytes, g_ctx.prim_table, g_ctx.prim_table_bytes,
                                    rt::pinhole_table_bytes(p), "the shared-origin table"))
            return rc;
        p.prim_table = ws.prim_table;
        HIP_TRY(rt::launch_pinhole(p, g_ctx.n_cu, g_ctx.device, stream));
    } else if (rt::paths_kernel_serves(p)) {
        p.grid_mode = rt::paths_grid_mode(p);
        p.max_blocks = ws.fixed ? (uint32_t)(rt::kNoAllocRingBytes / (8 * rt::kSecFields * rt::kSecCap * sizeof(uint32_t))) : 0u;
        if (int rc = ensure_scratch(ws, ws.secq, ws.secq_bytes, g_ctx.secq, g_ctx.secq_bytes, rt::paths_secq_bytes(p, g_ctx.n_cu),
                                    "the continuing-ray rings"))
            return rc;
        p.secq = ws.secq;
        HIP_TRY(rt::launch_paths(p, g_ctx.n_cu, g_ctx.device, stream));
    } else {
        HIP_TRY(rt::launch_trace(p, g_ctx.n_cu, g_ctx.device, stream));
    }
    if (getenv("RT_HIP_TRACE_DISPATCH"))  // which kernel served the call and where its grid lives (tests of the dispatch rules read this)
        std::fprintf(stderr, "rt_hip dispatch: %s grid_mode %u entries %u grid %u\n",
                     pinhole ? "pinhole_kernel" : (rt::paths_kernel_serves(p) ? "paths_kernel" : "trace_kernel"), p.grid_mode, p.sc.n_sph, p.sc.grid.enabled);
    return RT_OK;
}

// Samples per work unit of a render call.  The result does not depend on it (integer sums; ids by cell and sample); the speed
// does.  A unit costs ~900 instructions to deal, cull for and flush, so longer is cheaper; but the 64 lanes of a group should
// share as few cells as the sample count allows (lanes on one cell start with the same ray, list the same spheres, and their
// continuing rays walk together), and the units of a cell should be of one length (a group waits for its longest unit).
// Measured on one GPU, C2 scene / C4 scene, frame time by unit length:
//   64 spp: 2 3.82, 4 3.43, 8 3.52, 16 4.33, 32 6.04 ms | C4 64 spp: 4 27.4, 8 33.1 ms | C4 256 spp: 8 76.6, 16 93.4 ms
//   C4 1,024 spp: 8 244, 16 234, 32 281 ms | 16 spp: 2 1.25, 4 1.33 ms | 12 spp: 2 1.10, 3 1.08, 4 1.20, 12 2.18 ms
//   256 spp: 4 10.5, 8 10.1 ms | 128 spp: 2 6.70, 4 5.75, 8 6.01 ms
// Rule: spp / 32 within [4, 16] (2 up to 16 spp), then the nearest exact divisor of spp up to twice that, if there is one.
uint32_t render_unit_samples(uint32_t spp)
{
    uint32_t target = spp >= 32 ? spp / 32 : (spp > 16 ? 4u : 2u);
    if (spp >= 32) target = target < 4 ? 4u : (target > 16 ? 16u : target);
    if (target > spp) target = spp ? spp : 1u;
    for (uint32_t u = target; u <= 2 * target && u <= 16; ++u)
        if (spp % u == 0) return u;
    return target;
}

// rt_hip_render_dev and rt_hip_render_dev_noalloc: everything after the argument checks (g_mu held, device selected).
int render_enqueue(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings, uint64_t seed, uint64_t cell_begin,
                   uint64_t cell_count, uint64_t cell_step, float* d_fb, int32_t* d_first_id, float* d_first_t, uint32_t* d_seg_count,
                   uint64_t* d_total_segments, hipStream_t hs, Workspace& ws)
{
    rt::TraceParams p{};
    p.sc = scene->dev;
    p.rays = nullptr;
    p.gen = make_dev_gen(*gen);
    p.spp = spp;
    p.unit_samples = p.gen.kind == RT_GEN_PARALLEL_GRID ? rt::kUnitSamples : render_unit_samples(spp);  // the general kernel's lanes are on their own
    if (const char* e = getenv("RT_HIP_UNIT_SAMPLES")) {  // tuning knob: the result does not depend on it (integer sums, ids by cell and sample)
        const int v = atoi(e);
        if (v >= 1 && v <= 4096) p.unit_samples = (uint32_t)v;
    }
    p.units_per_cell = (spp + p.unit_samples - 1) / p.unit_samples;
    p.cell_count = (uint32_t)cell_count;
    p.cell_begin = cell_begin;
    p.cell_step = cell_step;
    p.st = settings;
    p.seed = seed;
    // fixed-point accumulation scratch, shared by every call that uses the same workspace
    const size_t need = sizeof(unsigned long long) * 4 * cell_count;  // {x, y, z, segments} per cell
    if (int rc = ensure_scratch(ws, ws.fb_q, ws.fb_q_bytes, g_ctx.fb_q, g_ctx.fb_q_bytes, need, "the fixed-point framebuffer")) return rc;
    HIP_TRY(hipMemsetAsync(ws.fb_q, 0, need, hs));
    p.fb_q = ws.fb_q;
    p.first_id = d_first_id;
    p.first_t = d_first_t;
    p.seg_count = d_seg_count;
    p.scan_mode = g_scan_mode;
    if (int rc = enqueue(p, (unsigned long long*)d_total_segments, hs, scene, ws)) return rc;
    HIP_TRY(rt::launch_finalize(ws.fb_q, d_fb, d_seg_count, cell_count, spp, hs));
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_hip_last_error(void) { return g_err.c_str(); }

int rt_hip_init(int device_ordinal)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_ordinal < 0 || device_ordinal >= n || device_ordinal >= rt::kMaxDevices) return fail(RT_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RT_ERR_UNSUPPORTED, std::string("librt_hip.so is built for gfx950 only, device is ") + prop.gcnArchName);
    if (g_ctx.ready && g_ctx.device != device_ordinal) {
        (void)hipSetDevice(g_ctx.device);  // the old context's memory lives on the old device
        (void)hipDeviceSynchronize();
        (void)hipFree(g_ctx.counters);
        (void)hipFree(g_ctx.fb_q);
        (void)hipFree(g_ctx.secq);
        (void)hipFree(g_ctx.prim_table);
        (void)hipFree(g_ctx.stage);
        if (g_ctx.last_done) (void)hipEventDestroy(g_ctx.last_done);
        g_ctx = Context{};
        HIP_TRY(hipSetDevice(device_ordinal));
    }
    if (!g_ctx.ready) {
        // (a failure half way leaves nothing behind: the next rt_hip_init starts from an empty context)
        HIP_TRY(hipMalloc((void**)&g_ctx.counters, kCounterRing * rt::kUnitCounterBytes));
        hipError_t e = hipEventCreateWithFlags(&g_ctx.last_done, hipEventDisableTiming);
        // this device's copy of the no-alloc workspace (rt_scratch.hip); device_ordinal is current
        void* scratch = nullptr;
        if (e == hipSuccess) e = rt::noalloc_scratch_address(&scratch);
        if (e != hipSuccess) {
            (void)hipFree(g_ctx.counters);
            if (g_ctx.last_done) (void)hipEventDestroy(g_ctx.last_done);
            g_ctx = Context{};
            return fail_hip(e, "rt_hip_init");
        }
        g_ctx.noalloc_scratch = static_cast<unsigned char*>(scratch);
        g_ctx.device = device_ordinal;
        g_ctx.n_cu = prop.multiProcessorCount;
        g_ctx.ready = true;
    }
    return RT_OK;
}

void rt_hip_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.ready) {
        (void)hipSetDevice(g_ctx.device);
        (void)hipDeviceSynchronize();
        (void)hipFree(g_ctx.counters);
        (void)hipFree(g_ctx.fb_q);
        (void)hipFree(g_ctx.secq);
        (void)hipFree(g_ctx.prim_table);
        (void)hipFree(g_ctx.stage);
        if (g_ctx.last_done) (void)hipEventDestroy(g_ctx.last_done);
        g_ctx = Context{};
    }
}

int rt_hip_set_scan_mode(int mode)
{
    if (mode != 0 && mode != 1) return fail(RT_ERR_INVALID, "scan mode must be 0 or 1");
    g_scan_mode = mode;
    return RT_OK;
}

int rt_hip_scene_create(const RtObject* objects, uint64_t n_objects, rt_scene** out_scene)
{
    return rt_hip_scene_create_ex(objects, n_objects, nullptr, 0, out_scene);
}

int rt_hip_scene_create_ex(const RtObject* objects, uint64_t n_objects, const RtShapeNode* shape_nodes, uint64_t n_shape_nodes,
                           rt_scene** out_scene)
{
    RtSceneDesc d{objects, n_objects, shape_nodes, n_shape_nodes, nullptr, 0};
    return rt_hip_scene_create_desc(&d, out_scene);
}

int rt_hip_scene_create_desc(const RtSceneDesc* desc, rt_scene** out_scene)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!desc) return fail(RT_ERR_INVALID, "null argument");
    const RtObject* objects = desc->objects;
    const uint64_t n_objects = desc->n_objects;
    const RtShapeNode* shape_nodes = desc->nodes;
    const uint64_t n_shape_nodes = desc->n_nodes;
    if (!desc->meshes && desc->n_meshes) return fail(RT_ERR_INVALID, "null argument");
    for (uint64_t mi = 0; mi < desc->n_meshes; ++mi) {
        const RtMesh& m = desc->meshes[mi];
        if ((!m.vertices && m.n_vertices) || (!m.indices && m.n_triangles)) return fail(RT_ERR_INVALID, "mesh " + std::to_string(mi) + ": null array");
        if (m.n_triangles >= (1ull << 28)) return fail(RT_ERR_INVALID, "mesh " + std::to_string(mi) + ": too many triangles");
        for (uint64_t k = 0; k < m.n_triangles * 3; ++k)
            if (m.indices[k] >= m.n_vertices) return fail(RT_ERR_INVALID, "mesh " + std::to_string(mi) + ": vertex index out of range");
        for (uint64_t k = 0; k < m.n_vertices * 3; ++k)
            if (!std::isfinite(m.vertices[k])) return fail(RT_ERR_INVALID, "mesh " + std::to_string(mi) + ": non-finite vertex");
    }
    std::vector<rt::BvhNode> bvh_nodes;
    std::vector<rt::TriDev> tris;
    std::vector<uint32_t> mesh_roots(desc->n_meshes, rt::kNoMesh);
    if (int rc = select_device()) return rc;
    if (!out_scene || (!objects && n_objects) || (!shape_nodes && n_shape_nodes)) return fail(RT_ERR_INVALID, "null argument");
    if (n_objects >= (1ull << 23)) return fail(RT_ERR_INVALID, "too many objects (limit 2^23)");
    *out_scene = nullptr;

    std::vector<float4> sph_cull, sph_exact, sph_box, pl_pos, pl_nrm, obj_geom(n_objects), obj_aux(n_objects);
    std::vector<int32_t> sph_obj, pl_obj;
    std::vector<uint2> obj_mat(n_objects);
    std::vector<uint32_t> obj_shape(n_objects);
    std::vector<int32_t> obj_cx(n_objects, -1);
    std::vector<rt::CxNode> cx_nodes;
    std::vector<rt::CxObject> cx_objs;
    double c2max = 0.0;
    for (uint64_t k = 0; k < n_objects; ++k) {
        const RtObject& o = objects[k];
        if (o.material_kind > RT_MAT_HALF_MIRROR) return fail(RT_ERR_INVALID, "object " + std::to_string(k) + ": unknown material kind");
        uint32_t pbits;
        std::memcpy(&pbits, &o.material_param, 4);
        obj_mat[k] = make_uint2(o.material_kind, pbits);
        obj_shape[k] = o.shape;
        const float px = o.position[0], py = o.position[1], pz = o.position[2];
        // Spheres, planes and boxes with rotation 0 fold their Transform exactly and ride the scan path (a box
        // behind its bounding sphere); every other object is evaluated per lane with the reference's intersect_all (rt_complex.h).
        const bool finite_box = o.shape == RT_SHAPE_BOX && std::isfinite(o.shape_param[0]) && std::isfinite(o.shape_param[1]) &&
                                std::isfinite(o.shape_param[2]);
        const bool fast = (o.shape == RT_SHAPE_SPHERE || o.shape == RT_SHAPE_PLANE || finite_box) && o.rotation_y_deg == 0.0f;
        if (!fast) {
            rt::host::Program pr;
            std::string err;
            uint32_t mesh_root = rt::kNoMesh;
            if (o.shape == RT_SHAPE_MESH) {
                uint32_t mi;
                std::memcpy(&mi, &o.shape_param[0], 4);
                if (mi >= desc->n_meshes) return fail(RT_ERR_INVALID, "object " + std::to_string(k) + ": mesh index out of range");
                if (mesh_roots[mi] == rt::kNoMesh) mesh_roots[mi] = rt::host::build_mesh_bvh(desc->meshes[mi], bvh_nodes, tris);
                mesh_root = mesh_roots[mi];
                if (tris.size() >= (1ull << 28)) return fail(RT_ERR_INVALID, "more than 2^28 triangles in one scene");
            } else if (rt::host::emit_shape(o.shape, o.shape_param, shape_nodes, n_shape_nodes, pr, 0, 0, err) < 0)
                return fail(RT_ERR_INVALID, "object " + std::to_string(k) + ": " + err);
            if (pr.max_hits > rt::kCxMaxHits || pr.max_stack > rt::kCxStack)
                return fail(RT_ERR_UNSUPPORTED, "object " + std::to_string(k) + ": shape tree needs " + std::to_string(pr.max_hits) +
                                                    " hits / stack " + std::to_string(pr.max_stack) + " (limits " +
                                                    std::to_string(rt::kCxMaxHits) + " / " + std::to_string(rt::kCxStack) + ")");
            rt::CxObject co{};
            co.first_node = (uint32_t)cx_nodes.size();
            co.n_nodes = (uint32_t)pr.nodes.size();
            co.obj_id = (int32_t)k;
            co.mesh_root = mesh_root;
            rt::host::Mat4 m, mi;
            rt::host::object_matrices(o, m, mi);
            std::memcpy(co.m, m.c, sizeof co.m);
            std::memcpy(co.minv, mi.c, sizeof co.minv);
            cx_nodes.insert(cx_nodes.end(), pr.nodes.begin(), pr.nodes.end());
            obj_cx[k] = (int32_t)cx_objs.size();
            cx_objs.push_back(co);
            obj_geom[k] = make_float4(px, py, pz, 0);
            obj_aux[k] = make_float4(0, 0, 0, 0);
        } else if (o.shape == RT_SHAPE_SPHERE) {
            const float r = o.shape_param[0];
            const double pp = (double)px * px + (double)py * py + (double)pz * pz;
            const double rr = (double)r * r;
            sph_cull.push_back(make_float4(px, py, pz, (float)(pp - rr)));
            sph_exact.push_back(make_float4(px, py, pz, r));
            sph_box.push_back(make_float4(0, 0, 0, 0));
            sph_obj.push_back((int32_t)k);
            obj_geom[k] = make_float4(px, py, pz, r);
            obj_aux[k] = make_float4(0, 0, 0, 0);
            if (pp + rr > c2max) c2max = pp + rr;
        } else if (o.shape == RT_SHAPE_BOX) {
            // shape_config.rs:61-68: AxisAlignedBox(-size/2, size/2).  The filter sees the box's bounding sphere (rounded up); the
            // confirm stage runs the exact slab test.
            const float hx = o.shape_param[0] / 2.0f, hy = o.shape_param[1] / 2.0f, hz = o.shape_param[2] / 2.0f;
            const double rr = ((double)hx * hx + (double)hy * hy + (double)hz * hz) * (1.0 + 1e-6) + 1e-30;
            const double pp = (double)px * px + (double)py * py + (double)pz * pz;
            float rb = (float)std::sqrt(rr);
            if ((double)rb * rb < rr) rb = std::nextafter(rb, std::numeric_limits<float>::infinity());
            sph_cull.push_back(make_float4(px, py, pz, (float)(pp - (double)rb * rb)));
            sph_exact.push_back(make_float4(px, py, pz, rb));
            sph_box.push_back(make_float4(hx, hy, hz, 1.0f));
            sph_obj.push_back((int32_t)k);
            obj_geom[k] = make_float4(px, py, pz, 0);
            obj_aux[k] = make_float4(hx, hy, hz, 0);
            if (pp + (double)rb * rb > c2max) c2max = pp + (double)rb * rb;
        } else if (o.shape == RT_SHAPE_PLANE) {
            pl_pos.push_back(make_float4(px, py, pz, 0));
            pl_nrm.push_back(make_float4(o.shape_param[0], o.shape_param[1], o.shape_param[2], 0));
            pl_obj.push_back((int32_t)k);
            obj_geom[k] = make_float4(px, py, pz, 0);
            obj_aux[k] = make_float4(o.shape_param[0], o.shape_param[1], o.shape_param[2], 0);
        }
    }
    const uint32_t n_sph = (uint32_t)sph_cull.size();
    // LDS mode: pad to the scan step; streamed mode (above kTileSpheres): pad to whole 64-sphere wave loads
    const uint32_t pad_to = n_sph > (uint32_t)rt::kTileSpheres ? 64u : (uint32_t)rt::kUnroll;
    const uint32_t n_pad = (n_sph + pad_to - 1) / pad_to * pad_to;
    const float inf = std::numeric_limits<float>::infinity();
    // The arrays themselves end on a 64-entry boundary whatever n_pad is: a table of the LDS size class whose grid does not fit into
    // LDS runs on the streamed kernels (grid_mode 2), and their scans read whole 64-sphere rows.
    const uint32_t n_alloc = (n_pad + 63u) / 64u * 64u;
    for (uint32_t k = n_sph; k < n_alloc; ++k) {
        sph_cull.push_back(make_float4(0, 0, 0, inf));  // never passes the filter
        sph_exact.push_back(make_float4(0, 0, 0, 1));
        sph_box.push_back(make_float4(0, 0, 0, 0));
        sph_obj.push_back(-1);
    }
    std::vector<uint2> sph_mat(n_alloc, make_uint2(0, 0));
    uint32_t n_box = 0, first_box = 0;
    for (uint32_t k = 0; k < n_sph; ++k) {
        sph_mat[k] = obj_mat[sph_obj[k]];
        if (sph_box[k].w != 0.0f && n_box++ == 0) first_box = k;
    }
    float c2max_f = (float)c2max;
    if ((double)c2max_f < c2max) c2max_f = std::nextafter(c2max_f, inf);
    // uniform grid for the continuing rays of the Projector kernel (disabled for small or unsuitable scenes)
    rt::host::GridHost grid = rt::host::build_grid(sph_exact, sph_box, n_sph);
    if (grid.items.size() & 1) grid.items.push_back(0);  // the kernel copies the item list in whole words
    rt::host::ClusterHost clusters;
    if (grid.g.enabled && n_sph <= rt::kClusterMaxEntries) clusters = rt::host::build_clusters(sph_exact, n_sph);

    rt_scene* s = new rt_scene;
    rt::DevScene& d = s->dev;
    hipError_t e = hipSuccess;
    auto up = [&](auto& vec, auto** dst) {
        if (e != hipSuccess) return;
        e = upload(vec, dst);
        if (e == hipSuccess) s->allocations.push_back((void*)*dst);
    };
    up(sph_cull, &d.sph_cull);
    up(sph_exact, &d.sph_exact);
    up(sph_box, &d.sph_box);
    up(sph_obj, &d.sph_obj);
    up(sph_mat, &d.sph_mat);
    up(pl_pos, &d.pl_pos);
    up(pl_nrm, &d.pl_nrm);
    up(pl_obj, &d.pl_obj);
    up(obj_geom, &d.obj_geom);
    up(obj_aux, &d.obj_aux);
    up(obj_mat, &d.obj_mat);
    up(obj_shape, &d.obj_shape);
    up(obj_cx, &d.obj_cx);
    up(cx_nodes, &d.cx_nodes);
    up(cx_objs, &d.cx_objs);
    up(bvh_nodes, &d.bvh_nodes);
    up(tris, &d.tris);
    d.grid = grid.g;
    if (grid.g.enabled) {
        up(grid.cells, &d.grid.cells);
        up(grid.items, &d.grid.items);
    }
    d.n_clu = (uint32_t)clusters.bound.size();
    if (d.n_clu) {
        up(clusters.bound, &d.clu_bound);
        up(clusters.items, &d.clu_items);
    }
    if (e != hipSuccess) {
        free_scene(s);
        return fail_hip(e, "scene upload");
    }
    d.n_sph = n_sph;
    d.n_box = n_box;
    d.first_box = first_box;
    d.n_sph_pad = n_pad;
    d.n_pl = (uint32_t)pl_pos.size();
    d.n_obj = (uint32_t)n_objects;
    d.n_cx = (uint32_t)cx_objs.size();
    d.n_bvh_nodes = (uint32_t)bvh_nodes.size();
    d.c2max = c2max_f;
    s->host_spheres.assign(sph_exact.begin(), sph_exact.begin() + n_sph);
    *out_scene = s;
    return RT_OK;
}

void rt_hip_scene_destroy(rt_scene* scene)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.ready) (void)hipSetDevice(g_ctx.device);
    free_scene(scene);
}

int rt_hip_scene_counts(const rt_scene* scene, uint64_t* n_objects, uint64_t* n_spheres, uint64_t* n_planes)
{
    if (!scene) return fail(RT_ERR_INVALID, "scene is null");
    if (n_objects) *n_objects = scene->dev.n_obj;
    if (n_spheres) *n_spheres = scene->dev.n_sph;
    if (n_planes) *n_planes = scene->dev.n_pl;
    return RT_OK;
}

int rt_hip_simulate_dev(const rt_scene* scene, const RtRay* d_rays, uint64_t n_rays, RtSettings settings, uint64_t seed,
                        uint64_t ray_id_base, float* d_points, uint32_t* d_n_points, RtBounceRecord* d_records,
                        uint64_t* d_total_segments, void* stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = check_common(scene, settings)) return rc;
    if (n_rays == 0) return RT_OK;
    if (!d_rays || !d_points || !d_n_points) return fail(RT_ERR_INVALID, "rays / points / n_points must not be null");
    if (n_rays > 0x7fffffffull) return fail(RT_ERR_INVALID, "n_rays exceeds 2^31-1 per call");
    if (settings.max_bounces == 0) return fail(RT_ERR_INVALID, "max_bounces == 0 is handled by the host entry point only");
    rt::TraceParams p{};
    p.sc = scene->dev;
    p.rays = d_rays;
    p.spp = 1;
    p.unit_samples = 1;
    p.units_per_cell = 1;
    p.cell_count = (uint32_t)n_rays;
    p.cell_begin = 0;
    p.cell_step = 1;
    p.path_id_base = ray_id_base;
    p.st = settings;
    p.seed = seed;
    p.points = d_points;
    p.n_points = d_n_points;
    p.records = d_records;
    p.scan_mode = g_scan_mode;
    if (int rc = order_after_previous((hipStream_t)stream)) return rc;
    Workspace ws = context_workspace();
    if (int rc = enqueue(p, (unsigned long long*)d_total_segments, (hipStream_t)stream, scene, ws)) return rc;
    return mark_enqueued((hipStream_t)stream);
}

int rt_hip_simulate(const rt_scene* scene, const RtRay* rays, uint64_t n_rays, RtSettings settings, uint64_t seed,
                    uint64_t ray_id_base, float* out_points, uint32_t* out_n_points, RtBounceRecord* out_records)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = check_common(scene, settings)) return rc;
    }
    if (n_rays == 0) return RT_OK;
    if (!rays || !out_points || !out_n_points) return fail(RT_ERR_INVALID, "rays / out_points / out_n_points must not be null");
    if (settings.max_bounces == 0) {  // scene.rs:45-49: the path is just its origin
        for (uint64_t r = 0; r < n_rays; ++r) {
            std::memcpy(out_points + r * 3, rays[r].origin, 12);
            out_n_points[r] = 1;
        }
        return RT_OK;
    }
    const size_t pts_bytes = sizeof(float) * 3 * ((size_t)settings.max_bounces + 1) * n_rays;
    const size_t rec_bytes = sizeof(RtBounceRecord) * (size_t)settings.max_bounces * n_rays;
    RtRay* d_rays = nullptr;
    float* d_pts = nullptr;
    uint32_t* d_np = nullptr;
    RtBounceRecord* d_rec = nullptr;
    int rc = RT_OK;
    // one device block {rays | polylines | vertex counts | records}, each part 256-byte aligned: the context's staging buffer when
    // it is (or may become) large enough, an allocation of this call's own otherwise
    auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t rays_bytes = sizeof(RtRay) * n_rays, np_bytes = sizeof(uint32_t) * n_rays;
    const size_t need = pad(rays_bytes) + pad(pts_bytes) + pad(np_bytes) + (out_records ? pad(rec_bytes) : 0);
    std::lock_guard<std::mutex> stage_lk(g_stage_mu);
    unsigned char* block = nullptr;
    bool own_block = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (need <= kStageKeepBytes) {
            if (g_ctx.stage_bytes < need) {
                (void)hipFree(g_ctx.stage);
                g_ctx.stage = nullptr;
                g_ctx.stage_bytes = 0;
                const size_t want = need < ((size_t)1 << 20) ? ((size_t)1 << 20) : need + need / 2;
                HIP_TRY(hipMalloc((void**)&g_ctx.stage, want));
                g_ctx.stage_bytes = want;
            }
            block = g_ctx.stage;
        }
    }
    if (!block) {
        HIP_TRY(hipMalloc((void**)&block, need));
        own_block = true;
    }
    auto cleanup = [&]() {
        if (own_block) (void)hipFree(block);
    };
#define TRY_OR_CLEAN(expr)                       \
    do {                                         \
        hipError_t e_ = (expr);                  \
        if (e_ != hipSuccess) {                  \
            cleanup();                           \
            return fail_hip(e_, #expr);          \
        }                                        \
    } while (0)
    d_rays = reinterpret_cast<RtRay*>(block);
    d_pts = reinterpret_cast<float*>(block + pad(rays_bytes));
    d_np = reinterpret_cast<uint32_t*>(block + pad(rays_bytes) + pad(pts_bytes));
    if (out_records) d_rec = reinterpret_cast<RtBounceRecord*>(block + pad(rays_bytes) + pad(pts_bytes) + pad(np_bytes));
    TRY_OR_CLEAN(hipMemcpy(d_rays, rays, rays_bytes, hipMemcpyHostToDevice));
    // polylines (and records) beyond a path's last vertex read as zeros
    TRY_OR_CLEAN(hipMemsetAsync(d_pts, 0, need - pad(rays_bytes), nullptr));
    rc = rt_hip_simulate_dev(scene, d_rays, n_rays, settings, seed, ray_id_base, d_pts, d_np, d_rec, nullptr, nullptr);
    if (rc != RT_OK) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipDeviceSynchronize());
    TRY_OR_CLEAN(hipMemcpy(out_points, d_pts, pts_bytes, hipMemcpyDeviceToHost));
    TRY_OR_CLEAN(hipMemcpy(out_n_points, d_np, np_bytes, hipMemcpyDeviceToHost));
    if (d_rec) TRY_OR_CLEAN(hipMemcpy(out_records, d_rec, rec_bytes, hipMemcpyDeviceToHost));
    cleanup();
    return RT_OK;
}

int rt_hip_render_dev(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings, uint64_t seed,
                      uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* d_fb, int32_t* d_first_id,
                      float* d_first_t, uint32_t* d_seg_count, uint64_t* d_total_segments, void* stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = check_common(scene, settings)) return rc;
    if (!d_fb) return fail(RT_ERR_INVALID, "gen / fb must not be null");
    if (int rc = check_render_args(gen, spp, cell_begin, cell_count, cell_step)) return rc;
    if (cell_count == 0) return RT_OK;
    if (settings.max_bounces == 0) return fail(RT_ERR_INVALID, "max_bounces == 0 is handled by the host entry point only");
    hipStream_t hs = (hipStream_t)stream;
    if (int rc = order_after_previous(hs)) return rc;
    Workspace ws = context_workspace();
    if (int rc = render_enqueue(scene, gen, spp, settings, seed, cell_begin, cell_count, cell_step, d_fb, d_first_id, d_first_t, d_seg_count,
                                d_total_segments, hs, ws))
        return rc;
    return mark_enqueued(hs);
}

int rt_hip_render_dev_noalloc(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings, uint64_t seed,
                              uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* d_fb, int32_t* d_first_id,
                              float* d_first_t, uint32_t* d_seg_count, uint64_t* d_total_segments, void* stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (int rc = check_common(scene, settings)) return rc;
    if (!d_fb) return fail(RT_ERR_INVALID, "gen / fb must not be null");
    if (int rc = check_render_args(gen, spp, cell_begin, cell_count, cell_step)) return rc;
    if (cell_count == 0) return RT_OK;
    if (settings.max_bounces == 0) return fail(RT_ERR_INVALID, "max_bounces == 0 is handled by the host entry point only");
    if (cell_count > rt::kNoAllocMaxCells)
        return fail(RT_ERR_UNSUPPORTED, "rt_hip_render_dev_noalloc: cell_count " + std::to_string(cell_count) + " exceeds the fixed workspace's limit of " +
                                            std::to_string(rt::kNoAllocMaxCells) + " cells per call (split the cell range)");
    if (scene->dev.n_sph > rt::kNoAllocMaxSpheres)
        return fail(RT_ERR_UNSUPPORTED, "rt_hip_render_dev_noalloc: the scene's " + std::to_string(scene->dev.n_sph) + " scan-path spheres exceed the fixed workspace's limit of " +
                                            std::to_string(rt::kNoAllocMaxSpheres));
    // No hipMalloc / hipFree below this line: the workspace is the static device array resolved at rt_hip_init.  It is shared
    // by every call of this entry point on the device and keeps the last call's contents; what must start at zero (work
    // counter, fixed-point sums) is cleared on `stream` first.  Calls that overlap on one GPU race: the caller orders them.
    Workspace ws = fixed_workspace();
    return render_enqueue(scene, gen, spp, settings, seed, cell_begin, cell_count, cell_step, d_fb, d_first_id, d_first_t, d_seg_count,
                          d_total_segments, (hipStream_t)stream, ws);
}

int rt_hip_render(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings, uint64_t seed,
                  uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* out_fb, int32_t* out_first_id,
                  float* out_first_t, uint32_t* out_seg_count, uint64_t* out_total_segments)
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (int rc = check_common(scene, settings)) return rc;
    }
    if (!out_fb) return fail(RT_ERR_INVALID, "gen / out_fb must not be null");
    if (int rc = check_render_args(gen, spp, cell_begin, cell_count, cell_step)) return rc;
    if (cell_count == 0) {
        if (out_total_segments) *out_total_segments = 0;
        return RT_OK;
    }
    if (settings.max_bounces == 0) {  // no segment is ever traced: every path value is 0
        std::memset(out_fb, 0, sizeof(float) * 3 * cell_count);
        for (uint64_t m = 0; m < cell_count; ++m) {
            if (out_first_id) out_first_id[m] = -1;
            if (out_first_t) out_first_t[m] = std::numeric_limits<float>::infinity();
            if (out_seg_count) out_seg_count[m] = 0;
        }
        if (out_total_segments) *out_total_segments = 0;
        return RT_OK;
    }
    float* d_fb = nullptr;
    int32_t* d_id = nullptr;
    float* d_t = nullptr;
    uint32_t* d_seg = nullptr;
    uint64_t* d_tot = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_fb);
        (void)hipFree(d_id);
        (void)hipFree(d_t);
        (void)hipFree(d_seg);
        (void)hipFree(d_tot);
    };
    TRY_OR_CLEAN(hipMalloc((void**)&d_fb, sizeof(float) * 3 * cell_count));
    if (out_first_id) TRY_OR_CLEAN(hipMalloc((void**)&d_id, sizeof(int32_t) * cell_count));
    if (out_first_t) TRY_OR_CLEAN(hipMalloc((void**)&d_t, sizeof(float) * cell_count));
    if (out_seg_count) TRY_OR_CLEAN(hipMalloc((void**)&d_seg, sizeof(uint32_t) * cell_count));
    TRY_OR_CLEAN(hipMalloc((void**)&d_tot, sizeof(uint64_t)));
    int rc = rt_hip_render_dev(scene, gen, spp, settings, seed, cell_begin, cell_count, cell_step, d_fb, d_id, d_t, d_seg, d_tot, nullptr);
    if (rc != RT_OK) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipDeviceSynchronize());
    TRY_OR_CLEAN(hipMemcpy(out_fb, d_fb, sizeof(float) * 3 * cell_count, hipMemcpyDeviceToHost));
    if (d_id) TRY_OR_CLEAN(hipMemcpy(out_first_id, d_id, sizeof(int32_t) * cell_count, hipMemcpyDeviceToHost));
    if (d_t) TRY_OR_CLEAN(hipMemcpy(out_first_t, d_t, sizeof(float) * cell_count, hipMemcpyDeviceToHost));
    if (d_seg) TRY_OR_CLEAN(hipMemcpy(out_seg_count, d_seg, sizeof(uint32_t) * cell_count, hipMemcpyDeviceToHost));
    if (out_total_segments) TRY_OR_CLEAN(hipMemcpy(out_total_segments, d_tot, sizeof(uint64_t), hipMemcpyDeviceToHost));
    cleanup();
    return RT_OK;
}

}  // extern "C"
