"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by raytracing_amd.
"""
import ctypes as C
import os

import numpy as np

from raytracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_scene_create_ex.restype = C.c_void_p
        L.orc_scene_create_ex.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_scene_create_desc.restype = C.c_void_p
        L.orc_scene_create_desc.argtypes = [C.POINTER(abi.RtSceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_simulate.restype = C.c_uint64
        L.orc_simulate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, abi.RtSettings, C.c_uint64, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_generate_rays.argtypes = [C.POINTER(abi.RtRayGenerator), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_render.restype = C.c_uint64
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.RtRayGenerator), C.c_uint32, abi.RtSettings, C.c_uint64,
                                 C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_uint32]
        f3 = C.POINTER(C.c_float)
        L.orc_sphere_first_hit.restype = C.c_int
        L.orc_sphere_first_hit.argtypes = [f3, C.c_float, C.c_void_p, C.c_float, C.c_float, f3, f3, f3, C.POINTER(C.c_uint32)]
        L.orc_plane_first_hit.restype = C.c_int
        L.orc_plane_first_hit.argtypes = [f3, f3, C.c_void_p, C.c_float, C.c_float, f3, f3, f3, C.POINTER(C.c_uint32)]
        L.orc_reflect.argtypes = [f3, f3, f3]
        L.orc_refract.restype = C.c_int
        L.orc_refract.argtypes = [f3, f3, C.c_float, f3]
        L.orc_uniform01.restype = C.c_float
        L.orc_uniform01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_object_matrices.argtypes = [C.c_void_p, f3, f3]
        u4 = C.POINTER(C.c_uint32)
        L.orc_philox4x32_10.argtypes = [u4, u4, u4]
        _LIB = L
    return _LIB


class OracleScene:
    def __init__(self, objects, nodes=None, meshes=None):
        self.objects = np.ascontiguousarray(objects, dtype=abi.OBJECT_DTYPE)
        self.nodes = np.zeros(0, abi.NODE_DTYPE) if nodes is None else np.ascontiguousarray(nodes, dtype=abi.NODE_DTYPE)
        desc, keep = abi.scene_desc(self.objects, self.nodes, list(meshes) if meshes else [])
        self._h = lib().orc_scene_create_desc(C.byref(desc))
        del keep

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_destroy(self._h)
            self._h = None

    def simulate(self, rays, infinity_distance, max_bounces, seed=0, ray_id_base=0, records=True):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        n = len(rays)
        pts = np.zeros((n, max_bounces + 1, 3), np.float32)
        npts = np.zeros(n, np.uint32)
        recs = np.zeros((n, max_bounces), abi.RECORD_DTYPE) if records else None
        st = abi.RtSettings(infinity_distance, max_bounces)
        total = lib().orc_simulate(self._h, abi.ptr(rays), n, st, seed, ray_id_base, abi.ptr(pts), abi.ptr(npts), abi.ptr(recs))
        return pts, npts, recs, int(total)

    def render(self, gen, spp, infinity_distance, max_bounces, seed=0, cell_begin=0, cell_count=None, cell_step=1, threads=1):
        if cell_count is None:
            cell_count = gen.count_u * gen.count_v
        fb = np.zeros((cell_count, 3), np.float32)
        fid = np.zeros(cell_count, np.int32)
        ft = np.zeros(cell_count, np.float32)
        seg = np.zeros(cell_count, np.uint32)
        st = abi.RtSettings(infinity_distance, max_bounces)
        total = lib().orc_render(self._h, C.byref(gen), spp, st, seed, cell_begin, cell_count, cell_step, abi.ptr(fb),
                                 abi.ptr(fid), abi.ptr(ft), abi.ptr(seg), threads)
        return fb, fid, ft, seg, int(total)


def generate_rays(gen, cell_begin=0, cell_count=None, cell_step=1):
    if cell_count is None:
        cell_count = gen.count_u * gen.count_v
    out = np.zeros(cell_count, abi.RAY_DTYPE)
    lib().orc_generate_rays(C.byref(gen), cell_begin, cell_count, cell_step, abi.ptr(out))
    return out


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def sphere_first_hit(center, radius, origin, direction, t_min=0.001, t_max=float("inf")):
    ray = np.zeros(1, abi.RAY_DTYPE)
    ray["origin"][0] = origin
    ray["direction"][0] = direction
    ray["current_ior"] = 1.0
    t = C.c_float()
    p, n = (C.c_float * 3)(), (C.c_float * 3)()
    ff = C.c_uint32()
    ok = lib().orc_sphere_first_hit(_f3(center), radius, abi.ptr(ray), t_min, t_max, C.byref(t), p, n, C.byref(ff))
    if not ok:
        return None
    return np.float32(t.value), np.array(p[:], np.float32), np.array(n[:], np.float32), bool(ff.value)


def plane_first_hit(ppoint, pnormal, origin, direction, t_min=0.001, t_max=float("inf")):
    ray = np.zeros(1, abi.RAY_DTYPE)
    ray["origin"][0] = origin
    ray["direction"][0] = direction
    ray["current_ior"] = 1.0
    t = C.c_float()
    p, n = (C.c_float * 3)(), (C.c_float * 3)()
    ff = C.c_uint32()
    ok = lib().orc_plane_first_hit(_f3(ppoint), _f3(pnormal), abi.ptr(ray), t_min, t_max, C.byref(t), p, n, C.byref(ff))
    if not ok:
        return None
    return np.float32(t.value), np.array(p[:], np.float32), np.array(n[:], np.float32), bool(ff.value)


def reflect(i, n):
    out = (C.c_float * 3)()
    lib().orc_reflect(_f3(i), _f3(n), out)
    return np.array(out[:], np.float32)


def refract(i, n, ior_ratio):
    out = (C.c_float * 3)()
    ok = lib().orc_refract(_f3(i), _f3(n), ior_ratio, out)
    return np.array(out[:], np.float32) if ok else None


def uniform01(seed, path_id, bounce):
    return float(lib().orc_uniform01(seed, path_id, bounce))


def philox4x32_10(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def object_matrices(obj):
    o = np.ascontiguousarray(obj, dtype=abi.OBJECT_DTYPE).reshape(1)
    m, mi = (C.c_float * 16)(), (C.c_float * 16)()
    lib().orc_object_matrices(abi.ptr(o), m, mi)
    return np.array(m[:], np.float32).reshape(4, 4).T, np.array(mi[:], np.float32).reshape(4, 4).T
