"""Host-side mirror of the reference's runtime interface on top of the C ABI.

Names follow the reference: `Scene { objects, rays }` + `Scene.simulate_rays(settings)`
(crates/raytracing_core/src/scene.rs:28-39), `SimulationSettingsConfig` (scene.rs:33-36),
`Material` (primitives/mod.rs:39-44).  The objects are the plain-data `ObjectConfig`s (object_config.rs:10-14)
after generator expansion, because `Box<dyn Hittable>` cannot cross an FFI.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi
from ._lib import check, lib

_INITIALISED_DEVICE = None


def init(device=0):
    """rt_hip_init: select the GPU. Must be called before anything else touches the device."""
    global _INITIALISED_DEVICE
    check(lib().rt_hip_init(int(device)))
    _INITIALISED_DEVICE = int(device)


def shutdown():
    global _INITIALISED_DEVICE
    lib().rt_hip_shutdown()
    _INITIALISED_DEVICE = None


def set_scan_mode(mode):
    check(lib().rt_hip_set_scan_mode(int(mode)))


@dataclass
class SimulationSettingsConfig:
    """scene.rs:33-36"""
    infinity_distance: float
    max_bounces: int

    def as_c(self):
        return abi.RtSettings(float(self.infinity_distance), int(self.max_bounces))


def sphere(position, radius, material_kind, material_param=0.0, rotation_y_deg=0.0):
    o = np.zeros((), abi.OBJECT_DTYPE)
    o["shape"] = abi.RT_SHAPE_SPHERE
    o["material_kind"] = material_kind
    o["material_param"] = material_param
    o["rotation_y_deg"] = rotation_y_deg
    o["position"] = position
    o["shape_param"] = (radius, 0, 0, 0)
    return o


def plane(position, normal, material_kind, material_param=0.0, rotation_y_deg=0.0):
    o = np.zeros((), abi.OBJECT_DTYPE)
    o["shape"] = abi.RT_SHAPE_PLANE
    o["material_kind"] = material_kind
    o["material_param"] = material_param
    o["rotation_y_deg"] = rotation_y_deg
    o["position"] = position
    o["shape_param"] = (normal[0], normal[1], normal[2], 0)
    return o


def shape_object(shape, shape_param, position, material_kind, material_param=0.0, rotation_y_deg=0.0):
    """Any ShapeConfig arm (shape_config.rs:9-51) as one RtObject; shape_param as documented in include/rt_hip.h."""
    o = np.zeros((), abi.OBJECT_DTYPE)
    o["shape"] = shape
    o["material_kind"] = material_kind
    o["material_param"] = material_param
    o["rotation_y_deg"] = rotation_y_deg
    o["position"] = position
    sp = list(shape_param) + [0.0] * (4 - len(shape_param))
    o["shape_param"] = sp
    return o


def tree_object(root_index, position, material_kind, material_param=0.0, rotation_y_deg=0.0):
    """Union/Intersection/Difference tree: the root node index travels as the bits of shape_param[0]."""
    o = shape_object(abi.RT_SHAPE_TREE, (0.0,), position, material_kind, material_param, rotation_y_deg)
    sp = np.zeros(4, np.float32)
    sp.view(np.uint32)[0] = root_index
    o["shape_param"] = sp
    return o


def mesh_object(mesh_index, position, material_kind, material_param=0.0, rotation_y_deg=0.0):
    """Triangle-mesh object (extension): the mesh index travels as the bits of shape_param[0]."""
    o = tree_object(mesh_index, position, material_kind, material_param, rotation_y_deg)
    o["shape"] = abi.RT_SHAPE_MESH
    return o


def make_rays(origins, directions, current_ior=1.0):
    origins = np.asarray(origins, np.float32).reshape(-1, 3)
    rays = np.zeros(len(origins), abi.RAY_DTYPE)
    rays["origin"] = origins
    rays["direction"] = np.asarray(directions, np.float32).reshape(-1, 3)
    rays["current_ior"] = current_ior
    return rays


class Scene:
    """`Scene { objects, rays }` (scene.rs:28-31) with the object list resident in HBM."""

    def __init__(self, objects, rays=None, nodes=None, meshes=None):
        self.objects = np.ascontiguousarray(np.asarray(objects, dtype=abi.OBJECT_DTYPE).reshape(-1))
        self.rays = np.zeros(0, abi.RAY_DTYPE) if rays is None else np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        self.nodes = np.zeros(0, abi.NODE_DTYPE) if nodes is None else np.ascontiguousarray(nodes, dtype=abi.NODE_DTYPE)
        self.meshes = list(meshes) if meshes else []
        self._h = C.c_void_p()
        desc, keep = abi.scene_desc(self.objects, self.nodes, self.meshes)
        check(lib().rt_hip_scene_create_desc(C.byref(desc), C.byref(self._h)))
        del keep

    def close(self):
        if self._h:
            lib().rt_hip_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib().rt_hip_scene_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- the hot call ----
    def simulate_rays(self, setting, seed=0, ray_id_base=0):
        """scene.rs:39: returns one (k, 3) float32 array per ray — the reference's Vec<Vec<Vec3>>."""
        pts, npts, _ = self.simulate_rays_raw(setting, seed=seed, ray_id_base=ray_id_base, records=False)
        return [pts[i, : npts[i]].copy() for i in range(len(npts))]

    def simulate_rays_raw(self, setting, rays=None, seed=0, ray_id_base=0, records=True):
        rays = self.rays if rays is None else np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        n = len(rays)
        mb = int(setting.max_bounces)
        pts = np.zeros((n, mb + 1, 3), np.float32)
        npts = np.zeros(n, np.uint32)
        recs = np.zeros((n, mb), abi.RECORD_DTYPE) if records else None
        check(lib().rt_hip_simulate(self._h, abi.ptr(rays), n, setting.as_c(), seed, ray_id_base, abi.ptr(pts), abi.ptr(npts),
                                    abi.ptr(recs)))
        return pts, npts, recs

    def simulate_rays_dev(self, setting, d_rays, n_rays, seed, ray_id_base, d_points, d_n_points, d_records, d_total, stream):
        """Device-pointer variant of the hot call (ints from tensor.data_ptr(); d_records / d_total may be 0); asynchronous on `stream`."""
        check(lib().rt_hip_simulate_dev(self._h, d_rays, n_rays, setting.as_c(), seed, ray_id_base, d_points, d_n_points,
                                        d_records or None, d_total or None, stream))

    def render(self, gen, spp, setting, seed=0, cell_begin=0, cell_count=None, cell_step=1):
        """Generator-driven loop: per-cell framebuffer + first-hit ids/t + segment counts (host buffers)."""
        if cell_count is None:
            cell_count = gen.count_u * gen.count_v
        fb = np.zeros((cell_count, 3), np.float32)
        fid = np.zeros(cell_count, np.int32)
        ft = np.zeros(cell_count, np.float32)
        seg = np.zeros(cell_count, np.uint32)
        tot = C.c_uint64()
        check(lib().rt_hip_render(self._h, C.byref(gen), spp, setting.as_c(), seed, cell_begin, cell_count, cell_step,
                                  abi.ptr(fb), abi.ptr(fid), abi.ptr(ft), abi.ptr(seg), C.byref(tot)))
        return fb, fid, ft, seg, tot.value

    def render_dev(self, gen, spp, setting, seed, cell_begin, cell_count, cell_step, d_fb, d_first_id, d_first_t, d_seg_count,
                   d_total, stream):
        """Device-pointer variant (ints from tensor.data_ptr()); asynchronous on `stream`."""
        check(lib().rt_hip_render_dev(self._h, C.byref(gen), spp, setting.as_c(), seed, cell_begin, cell_count, cell_step,
                                      d_fb, d_first_id, d_first_t, d_seg_count, d_total, stream))


    def render_dev_noalloc(self, gen, spp, setting, seed, cell_begin, cell_count, cell_step, d_fb, d_first_id, d_first_t, d_seg_count,
                           d_total, stream):
        """rt_hip_render_dev_noalloc: same call on the library's fixed device workspace — no hipMalloc (include/rt_hip.h for the limits)."""
        check(lib().rt_hip_render_dev_noalloc(self._h, C.byref(gen), spp, setting.as_c(), seed, cell_begin, cell_count, cell_step,
                                              d_fb, d_first_id, d_first_t, d_seg_count, d_total, stream))


def projector(origin, target_corner, target_u, target_v, count_u, count_v, current_ior=1.0):
    """RayGeneratorConfig::Projector (object_generator_config.rs:23-31)."""
    g = abi.RtRayGenerator()
    g.kind = abi.RT_GEN_PROJECTOR
    g.count_u, g.count_v, g.current_ior = int(count_u), int(count_v), float(current_ior)
    g.a[:] = [float(x) for x in origin]
    g.corner[:] = [float(x) for x in target_corner]
    g.u[:] = [float(x) for x in target_u]
    g.v[:] = [float(x) for x in target_v]
    g.direction[:] = [0.0, 0.0, 0.0]
    return g


def parallel_grid(origin_corner, vec_u, vec_v, count_u, count_v, direction, current_ior=1.0):
    """RayGeneratorConfig::ParallelGrid (object_generator_config.rs:14-22)."""
    g = abi.RtRayGenerator()
    g.kind = abi.RT_GEN_PARALLEL_GRID
    g.count_u, g.count_v, g.current_ior = int(count_u), int(count_v), float(current_ior)
    g.a[:] = [float(x) for x in origin_corner]
    g.corner[:] = [0.0, 0.0, 0.0]
    g.u[:] = [float(x) for x in vec_u]
    g.v[:] = [float(x) for x in vec_v]
    g.direction[:] = [float(x) for x in direction]
    return g
