"""ctypes / numpy mirrors of include/rt_hip.h (one-to-one, tightly packed 32-bit fields)."""
import ctypes as C

import numpy as np

RT_MAT_MIRROR, RT_MAT_GLASS, RT_MAT_HALF_MIRROR = 0, 1, 2
RT_SHAPE_SPHERE, RT_SHAPE_PLANE = 0, 1
RT_SHAPE_BOX, RT_SHAPE_CYLINDER, RT_SHAPE_CONE, RT_SHAPE_WEDGE, RT_SHAPE_LENS, RT_SHAPE_TREE = 2, 3, 4, 5, 6, 7
RT_SHAPE_MESH = 8
RT_SHAPE_UNION, RT_SHAPE_INTERSECTION, RT_SHAPE_DIFFERENCE = 16, 17, 18
RT_GEN_PARALLEL_GRID, RT_GEN_PROJECTOR = 0, 1

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_HIP, RT_ERR_NOT_INIT = 0, -1, -2, -3, -4

# numpy structured dtypes (arrays of these are passed by pointer)
OBJECT_DTYPE = np.dtype(
    [
        ("shape", "<u4"),
        ("material_kind", "<u4"),
        ("material_param", "<f4"),
        ("rotation_y_deg", "<f4"),
        ("position", "<f4", (3,)),
        ("shape_param", "<f4", (4,)),
    ]
)
NODE_DTYPE = np.dtype([("shape", "<u4"), ("a", "<u4"), ("b", "<u4"), ("shape_param", "<f4", (4,))])
RAY_DTYPE = np.dtype([("origin", "<f4", (3,)), ("direction", "<f4", (3,)), ("current_ior", "<f4")])
RECORD_DTYPE = np.dtype(
    [
        ("obj_id", "<i4"),
        ("t", "<f4"),
        ("point", "<f4", (3,)),
        ("normal", "<f4", (3,)),
        ("front_face", "<u4"),
        ("dir_out", "<f4", (3,)),
        ("ior_out", "<f4"),
    ]
)
assert NODE_DTYPE.itemsize == 28
assert OBJECT_DTYPE.itemsize == 44 and RAY_DTYPE.itemsize == 28 and RECORD_DTYPE.itemsize == 52


class RtSettings(C.Structure):
    _fields_ = [("infinity_distance", C.c_float), ("max_bounces", C.c_uint32)]


class RtRayGenerator(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("count_u", C.c_uint32),
        ("count_v", C.c_uint32),
        ("current_ior", C.c_float),
        ("a", C.c_float * 3),
        ("corner", C.c_float * 3),
        ("u", C.c_float * 3),
        ("v", C.c_float * 3),
        ("direction", C.c_float * 3),
    ]


class RtMesh(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint64), ("indices", C.c_void_p), ("n_triangles", C.c_uint64)]


class RtSceneDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("n_objects", C.c_uint64), ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint64)]


def scene_desc(objects, nodes, meshes):
    """Build an RtSceneDesc from numpy arrays; `meshes` is a list of (vertices float32 [n,3], indices uint32 [m,3]).
    Returns (desc, keepalive) — keep `keepalive` referenced while the descriptor is in use."""
    keep = [objects, nodes]
    marr = (RtMesh * max(1, len(meshes)))()
    for k, (v, idx) in enumerate(meshes):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        keep += [v, idx]
        marr[k] = RtMesh(v.ctypes.data, len(v), idx.ctypes.data, len(idx))
    keep.append(marr)
    d = RtSceneDesc(objects.ctypes.data if len(objects) else None, len(objects), nodes.ctypes.data if len(nodes) else None, len(nodes),
                    C.cast(marr, C.c_void_p).value if len(meshes) else None, len(meshes))
    return d, keep


assert C.sizeof(RtRayGenerator) == 76 and C.sizeof(RtSettings) == 8 and C.sizeof(RtMesh) == 32 and C.sizeof(RtSceneDesc) == 48


def ptr(arr, ctype=C.c_void_p):
    """Pointer to a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(ctype)
