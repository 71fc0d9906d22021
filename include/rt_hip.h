/* rt_hip.h — C ABI of the MI355X (gfx950) ray-path engine.
 *
 * Drop-in boundary for the reference's single hot call
 *     Scene::simulate_rays(&self, SimulationSettingsConfig) -> Vec<Vec<Vec3>>
 *     (crates/raytracing_core/src/scene.rs:39, called from crates/raytracing_cli/src/cli.rs:14)
 * The reference has no FFI of its own; `Scene.objects` is `Vec<Box<dyn Hittable>>` (scene.rs:29),
 * which cannot be introspected, so the seam sits one step earlier: the caller hands over the
 * plain-data object list that `impl Into<Scene> for SceneConfig` (scene_config.rs:22-113) would
 * have boxed, in the same order (explicit objects first, then every ObjectGrid with i outer,
 * j inner).  Index in that list is the `obj_id` reported by every entry point.
 *
 * Plain pointers and sizes only; no C++/torch types.  All structs are tightly packed POD made of
 * 32-bit fields, so a Rust `#[repr(C)]`, ctypes or cgo mirror is one-to-one.
 *
 * Conventions: return 0 on success, a negative RT_ERR_* otherwise; rt_hip_last_error() gives the
 * text for the calling thread's last failure.  "Host" entry points take caller-owned host buffers
 * and synchronise before returning.  "_dev" entry points take device pointers plus a hipStream_t
 * (as void*), enqueue work and return without synchronising.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---- */
#define RT_OK 0
#define RT_ERR_INVALID (-1)     /* bad argument (null pointer, zero count, unknown enum) */
#define RT_ERR_UNSUPPORTED (-2) /* valid in the reference, beyond a limit of the GPU path (a CSG tree needing > 16 hits per list or > 6 stacked lists) */
#define RT_ERR_HIP (-3)         /* a HIP runtime call failed; text in rt_hip_last_error() */
#define RT_ERR_NOT_INIT (-4)    /* rt_hip_init() has not been called on this process */

/* ---- Material (crates/raytracing_core/src/primitives/mod.rs:39-44) ---- */
#define RT_MAT_MIRROR 0u
#define RT_MAT_GLASS 1u       /* param = ior */
#define RT_MAT_HALF_MIRROR 2u /* param = reflectance */

/* ---- ShapeConfig arms on the GPU path (crates/raytracing_config/src/model/shape_config.rs:56-60, 69-73) ---- */
#define RT_SHAPE_SPHERE 0u /* shape_param[0] = radius; local centre is the origin */
#define RT_SHAPE_PLANE 1u  /* shape_param[0..2] = normal (NOT normalised, as in the reference); local point is the origin */
/* The remaining ShapeConfig arms (shape_config.rs:61-68, 74-149).  They, and any object with rotation_y_deg != 0, take the
 * per-lane "complex object" path of the kernels (exact intersect_all + CSG state machine, csg.rs:9-75). */
#define RT_SHAPE_BOX 2u      /* shape_param[0..2] = size                        -> AxisAlignedBox(-size/2, size/2) */
#define RT_SHAPE_CYLINDER 3u /* shape_param[0] = height, [1] = radius           -> (InfiniteCylinder & cap) & cap  */
#define RT_SHAPE_CONE 4u     /* shape_param[0] = angle_deg, [1] = height        -> InfiniteCone & cap              */
#define RT_SHAPE_WEDGE 5u    /* shape_param[0..2] = size, [3] = angle_deg       -> five planes intersected         */
#define RT_SHAPE_LENS 6u     /* thickness, diameter, r1, r2                     -> (surface & surface) & cylinder  */
#define RT_SHAPE_TREE 7u     /* Union/Intersection/Difference tree; the bits of shape_param[0] are the uint32 index of
                                the root RtShapeNode (rt_hip_scene_create_ex) */
#define RT_SHAPE_MESH 8u     /* triangle mesh — NO reference counterpart (BASELINE.json configs[2]); the bits of
                                shape_param[0] are the uint32 index into RtSceneDesc.meshes.  Semantics: DESIGN.md §4.9 */
/* operator codes, RtShapeNode only (shape_config.rs:39-50) */
#define RT_SHAPE_UNION 16u
#define RT_SHAPE_INTERSECTION 17u
#define RT_SHAPE_DIFFERENCE 18u

/* One ObjectConfig after generator expansion (object_config.rs:10-14, transform_config.rs:3-7).
 * The reference wraps every such object in Transform(T(position) * R_y(rotation_y_deg))
 * (object_config.rs:24-28). */
typedef struct RtObject {
    uint32_t shape;          /* RT_SHAPE_* */
    uint32_t material_kind;  /* RT_MAT_* */
    float material_param;    /* ior | reflectance | unused */
    float rotation_y_deg;    /* any value; 0 keeps spheres and planes on the fast scan path */
    float position[3];
    float shape_param[4];
} RtObject; /* 44 bytes */

/* One node of a Union/Intersection/Difference ShapeConfig tree (shape_config.rs:39-50): a leaf is any of
 * RT_SHAPE_SPHERE..RT_SHAPE_LENS with its shape_param (no transform: only the whole object is wrapped in a
 * Transform), an operator is RT_SHAPE_UNION/INTERSECTION/DIFFERENCE with child node indices a and b. */
typedef struct RtShapeNode {
    uint32_t shape;
    uint32_t a;
    uint32_t b;
    float shape_param[4];
} RtShapeNode; /* 28 bytes */

/* Indexed triangle mesh in the object's local space (extension, see RT_SHAPE_MESH). */
typedef struct RtMesh {
    const float* vertices;   /* [n_vertices][3] */
    uint64_t n_vertices;
    const uint32_t* indices; /* [n_triangles][3] */
    uint64_t n_triangles;
} RtMesh; /* 32 bytes */

/* Everything a scene is made of, for rt_hip_scene_create_desc. */
typedef struct RtSceneDesc {
    const RtObject* objects;
    uint64_t n_objects;
    const RtShapeNode* nodes; /* may be NULL when n_nodes == 0 */
    uint64_t n_nodes;
    const RtMesh* meshes;     /* may be NULL when n_meshes == 0 */
    uint64_t n_meshes;
} RtSceneDesc; /* 48 bytes */

/* Ray (scene.rs:110-115), 28 bytes. Direction is used as given (the reference normalises when it
 * builds rays from config: ray_config.rs:16, scene_config.rs:72,103). */
typedef struct RtRay {
    float origin[3];
    float direction[3];
    float current_ior;
} RtRay;

/* SimulationSettingsConfig (scene.rs:33-36). */
typedef struct RtSettings {
    float infinity_distance;
    uint32_t max_bounces;
} RtSettings;

/* RayGeneratorConfig (object_generator_config.rs:13-33; semantics scene_config.rs:60-108).
 * Ray order is i (u) outer, j (v) inner: cell = i * count_v + j. */
#define RT_GEN_PARALLEL_GRID 0u /* a = origin_corner, u = vec_u, v = vec_v, direction = direction */
#define RT_GEN_PROJECTOR 1u     /* a = origin, corner = target_corner, u = target_u, v = target_v */
typedef struct RtRayGenerator {
    uint32_t kind;
    uint32_t count_u;
    uint32_t count_v;
    float current_ior;
    float a[3];
    float corner[3];
    float u[3];
    float v[3];
    float direction[3];
} RtRayGenerator; /* 76 bytes */

/* Per-bounce record (HitRecord scene.rs:118-125 plus the scatter result scene.rs:68-97).
 * obj_id = index in the object list, or -1 for the miss terminator (then t = +inf and point is
 * the appended far point scene.rs:99; normal/dir_out = 0). */
typedef struct RtBounceRecord {
    int32_t obj_id;
    float t;
    float point[3];
    float normal[3];
    uint32_t front_face;
    float dir_out[3];
    float ior_out;
} RtBounceRecord; /* 52 bytes */

typedef struct rt_scene rt_scene; /* opaque: flattened scene resident in HBM on the init device */

/* ---- lifetime ---- */
/* Select the device, create the library context. Replaces nothing in the reference (it has no
 * device); must precede every other call. */
int rt_hip_init(int device_ordinal);
void rt_hip_shutdown(void);
const char* rt_hip_last_error(void);

/* Flatten and upload the object list: the GPU-side counterpart of building `Scene.objects`
 * (scene_config.rs:25-52 + object_config.rs:16-30 + transform.rs:11-17). */
int rt_hip_scene_create(const RtObject* objects, uint64_t n_objects, rt_scene** out_scene);
/* Same, with the node array that RT_SHAPE_TREE objects index (nodes may be NULL when n_nodes == 0). */
int rt_hip_scene_create_ex(const RtObject* objects, uint64_t n_objects, const RtShapeNode* nodes, uint64_t n_nodes,
                           rt_scene** out_scene);
/* Same, everything through one descriptor (adds triangle meshes; a BVH per mesh is built here on the host). */
int rt_hip_scene_create_desc(const RtSceneDesc* desc, rt_scene** out_scene);
void rt_hip_scene_destroy(rt_scene* scene);

/* ---- the hot call: Scene::simulate_rays (scene.rs:39-106) ----
 * For ray r (0 <= r < n_rays): out_points[(r*(max_bounces+1) + k)*3 .. +3] is polyline vertex k,
 * out_n_points[r] the vertex count (2 .. max_bounces+1) — exactly the inner Vec<Vec3> the
 * reference returns.  out_records (optional, may be NULL) holds [n_rays][max_bounces] records;
 * entries past the path's last bounce are zero-filled.  The HalfMirror draw (scene.rs:88) uses the
 * counter-based stream u(seed, ray_id_base + r, bounce). */
int rt_hip_simulate(const rt_scene* scene, const RtRay* rays, uint64_t n_rays, RtSettings settings,
                    uint64_t seed, uint64_t ray_id_base, float* out_points, uint32_t* out_n_points,
                    RtBounceRecord* out_records);

/* Same, device buffers, asynchronous on `stream` (a hipStream_t).  The library's scratch buffers are shared by all calls
 * on the context: calls are serialised on the host by a mutex, and a _dev call enqueued on a different stream than the
 * previous one first makes its stream wait (hipStreamWaitEvent) for the previous call's work, so _dev calls never overlap
 * on the device.  Any host thread may call (each entry point selects the init device itself). */
int rt_hip_simulate_dev(const rt_scene* scene, const RtRay* d_rays, uint64_t n_rays, RtSettings settings,
                        uint64_t seed, uint64_t ray_id_base, float* d_points, uint32_t* d_n_points,
                        RtBounceRecord* d_records, uint64_t* d_total_segments, void* stream);

/* ---- generator-driven render loop (no reference counterpart for the image part: SURVEY §8 a14) ----
 * Cells m = 0..cell_count-1 map to generator cells cell_begin + m*cell_step; each cell traces `spp`
 * paths (path id = cell*spp + s, which keys the HalfMirror stream) generated per scene_config.rs:60-108
 * and accumulates, in sample order, the path value defined in DESIGN.md ("sky" of the escaping
 * direction, 0 when the bounce budget runs out), divided by spp at the end.
 * Outputs (each optional except fb): fb[m*3..] RGB f32; first_id[m]/first_t[m] = object index and t of
 * the primary segment's closest hit (-1 / +inf on a miss); seg_count[m] = bounce-loop iterations
 * executed for the cell (the "rays" of the Mrays/s metric); *total_segments = their sum. */
int rt_hip_render(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings,
                  uint64_t seed, uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* out_fb,
                  int32_t* out_first_id, float* out_first_t, uint32_t* out_seg_count,
                  uint64_t* out_total_segments);

int rt_hip_render_dev(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings,
                      uint64_t seed, uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* d_fb,
                      int32_t* d_first_id, float* d_first_t, uint32_t* d_seg_count, uint64_t* d_total_segments,
                      void* stream);

/* Same as rt_hip_render_dev, but allocates nothing.  NO COUNTERPART IN THE REFERENCE (and none in its call sites): it exists
 * for callers that cannot call hipMalloc — code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation.  Its kernels are rt_hip_render_dev's; their scratch (work counter, streamed sphere table, continuing-ray
 * rings, fixed-point framebuffer) lives in one static device array of 58,986,496 bytes (56.3 MiB) that every GPU the
 * library is initialised on holds until the process ends.  Limits, beyond which the call fails with RT_ERR_UNSUPPORTED and
 * a message naming the limit: cell_count <= 524,288 per call (split the cell range over several calls), at most 16,384
 * scan-path spheres in the scene, at most 512 resident 512-thread blocks (256 CUs x 2).  It calls no hipMalloc / hipFree /
 * allocating library routine and does not synchronise.  All calls on a GPU share that GPU's array, which keeps the last
 * call's contents (the call clears what must start at zero on `stream`): two calls that overlap on one GPU race — issue
 * them on one stream, or order them yourself.  It does not interact with the other entry points' scratch. */
int rt_hip_render_dev_noalloc(const rt_scene* scene, const RtRayGenerator* gen, uint32_t spp, RtSettings settings,
                              uint64_t seed, uint64_t cell_begin, uint64_t cell_count, uint64_t cell_step, float* d_fb,
                              int32_t* d_first_id, float* d_first_t, uint32_t* d_seg_count, uint64_t* d_total_segments,
                              void* stream);

/* Number of objects / scan-path entries (rotation-0 spheres and boxes) / planes held by a scene (for harnesses and roofline
 * accounting). */
int rt_hip_scene_counts(const rt_scene* scene, uint64_t* n_objects, uint64_t* n_spheres, uint64_t* n_planes);

/* Tuning / diagnostics: 0 = conservative filters + exact confirm (default; the list scan is pruned by per-wave cull lists and a
 * uniform grid over the spheres where the scene is large enough), 1 = the reference's own discriminant as the filter over the
 * whole list (op-for-op, no cull, general kernel). Results are bit-identical; only speed differs. */
int rt_hip_set_scan_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
