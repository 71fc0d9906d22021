"""Loader for librt_hip.so (the C-ABI library declared in include/rt_hip.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(_HERE, "librt_hip.so")  # RT_HIP_LIBRARY: explicit path (A/B builds)
_LIB = None

# every symbol include/rt_hip.h declares
EXPORTS = [
    "rt_hip_init",
    "rt_hip_shutdown",
    "rt_hip_last_error",
    "rt_hip_scene_create",
    "rt_hip_scene_create_ex",
    "rt_hip_scene_create_desc",
    "rt_hip_scene_destroy",
    "rt_hip_scene_counts",
    "rt_hip_simulate",
    "rt_hip_simulate_dev",
    "rt_hip_render",
    "rt_hip_render_dev",
    "rt_hip_render_dev_noalloc",
    "rt_hip_set_scan_mode",
]


class RtHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"rt_hip error {code}: {text}")
        self.code = code


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) and load it by
    the unversioned name, so if librt_hip.so has already pulled in /opt/rocm's copy, a later `import torch` loads a SECOND
    runtime, which then finds no GPU ("No HIP GPUs are available").  When torch is installed its copy is therefore loaded first
    (without importing torch); librt_hip.so's NEEDED libamdhip64.so.7 then resolves to it, and torch finds it already loaded."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch's runtime is already the process's runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """dlopen librt_hip.so and declare prototypes. Does not touch the GPU."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.rt_hip_init.argtypes = [C.c_int]
    L.rt_hip_init.restype = C.c_int
    L.rt_hip_shutdown.argtypes = []
    L.rt_hip_shutdown.restype = None
    L.rt_hip_last_error.argtypes = []
    L.rt_hip_last_error.restype = C.c_char_p
    L.rt_hip_scene_create.argtypes = [vp, u64, C.POINTER(vp)]
    L.rt_hip_scene_create.restype = C.c_int
    L.rt_hip_scene_create_ex.argtypes = [vp, u64, vp, u64, C.POINTER(vp)]
    L.rt_hip_scene_create_ex.restype = C.c_int
    L.rt_hip_scene_create_desc.argtypes = [C.POINTER(abi.RtSceneDesc), C.POINTER(vp)]
    L.rt_hip_scene_create_desc.restype = C.c_int
    L.rt_hip_scene_destroy.argtypes = [vp]
    L.rt_hip_scene_destroy.restype = None
    L.rt_hip_scene_counts.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.rt_hip_scene_counts.restype = C.c_int
    L.rt_hip_simulate.argtypes = [vp, vp, u64, abi.RtSettings, u64, u64, vp, vp, vp]
    L.rt_hip_simulate.restype = C.c_int
    L.rt_hip_simulate_dev.argtypes = [vp, vp, u64, abi.RtSettings, u64, u64, vp, vp, vp, vp, vp]
    L.rt_hip_simulate_dev.restype = C.c_int
    L.rt_hip_render.argtypes = [vp, C.POINTER(abi.RtRayGenerator), u32, abi.RtSettings, u64, u64, u64, u64, vp, vp, vp, vp,
                                C.POINTER(u64)]
    L.rt_hip_render.restype = C.c_int
    L.rt_hip_render_dev.argtypes = [vp, C.POINTER(abi.RtRayGenerator), u32, abi.RtSettings, u64, u64, u64, u64, vp, vp, vp,
                                    vp, vp, vp]
    L.rt_hip_render_dev.restype = C.c_int
    L.rt_hip_render_dev_noalloc.argtypes = L.rt_hip_render_dev.argtypes
    L.rt_hip_render_dev_noalloc.restype = C.c_int
    L.rt_hip_set_scan_mode.argtypes = [C.c_int]
    L.rt_hip_set_scan_mode.restype = C.c_int
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise RtHipError(rc, lib().rt_hip_last_error().decode("utf-8", "replace"))
