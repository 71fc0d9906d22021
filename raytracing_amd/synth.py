"""Synthetic benchmark scenes.  Host-side numpy; the same object list goes to the oracle and the GPU."""
import numpy as np

from . import abi
from .engine import projector

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)


def philox_u01(seed, n, stream):
    """n uniform f32 in [0,1): Philox4x32-10, key = seed, counter = (i, 0, stream, tag) — same generator as the kernels."""
    c0 = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c1 = np.zeros(n, np.uint64)
    c2 = np.full(n, stream, np.uint64)
    c3 = np.full(n, 0x53434E45, np.uint64)  # 'SCNE'
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return ((c0 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def sphere_scene(n_spheres, extent, seed=0xC0FFEE, r_min=0.5, r_max=2.0):
    """n spheres, centres uniform in [-ex,ex]x[-ey,ey]x[-ez,ez], radii uniform [r_min,r_max];
    materials by index mod 10: 6 Glass{1.5}, 3 Mirror, 1 HalfMirror{0.5}; rotation 0."""
    ex, ey, ez = extent
    u = [philox_u01(seed, n_spheres, s) for s in range(4)]
    objs = np.zeros(n_spheres, abi.OBJECT_DTYPE)
    objs["shape"] = abi.RT_SHAPE_SPHERE
    objs["position"][:, 0] = (u[0] * np.float32(2) - np.float32(1)) * np.float32(ex)
    objs["position"][:, 1] = (u[1] * np.float32(2) - np.float32(1)) * np.float32(ey)
    objs["position"][:, 2] = (u[2] * np.float32(2) - np.float32(1)) * np.float32(ez)
    objs["shape_param"][:, 0] = np.float32(r_min) + u[3] * np.float32(r_max - r_min)
    k = np.arange(n_spheres) % 10
    objs["material_kind"] = np.where(k < 6, abi.RT_MAT_GLASS, np.where(k < 9, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR))
    objs["material_param"] = np.where(k < 6, np.float32(1.5), np.where(k < 9, np.float32(0.0), np.float32(0.5)))
    return objs


def config_c2(count_u=1920, count_v=1080):
    """BASELINE.json configs[1]: 1,000 spheres, 1920x1080, 64 spp, depth 8."""
    objs = sphere_scene(1000, (50.0, 20.0, 50.0))
    gen = projector((0.0, 0.0, -120.0), (-64.0, -36.0, -60.0), (128.0, 0.0, 0.0), (0.0, 72.0, 0.0), count_u, count_v)
    return dict(objects=objs, gen=gen, spp=64, max_bounces=8, infinity_distance=50.0, seed=0xC0FFEE, name="C2")


def config_c4(count_u=3840, count_v=2160):
    """BASELINE.json configs[3]: 10,000 spheres, 3840x2160, 1024 spp, depth 8."""
    objs = sphere_scene(10000, (150.0, 60.0, 150.0))
    gen = projector((0.0, 0.0, -360.0), (-192.0, -108.0, -180.0), (384.0, 0.0, 0.0), (0.0, 216.0, 0.0), count_u, count_v)
    return dict(objects=objs, gen=gen, spp=1024, max_bounces=8, infinity_distance=50.0, seed=0xC0FFEE, name="C4")


def heightfield_mesh(nx, nz, extent=(60.0, 60.0), amplitude=6.0, seed=0xC0FFEE):
    """(vertices [n,3] f32, indices [m,3] u32): a bumpy nx x nz grid in the XZ plane, 2*nx*nz triangles."""
    ex, ez = extent
    xs = np.linspace(-ex, ex, nx + 1, dtype=np.float32)
    zs = np.linspace(-ez, ez, nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    noise = philox_u01(seed, (nx + 1) * (nz + 1), 7).reshape(nx + 1, nz + 1)
    Y = (np.float32(amplitude) * (np.sin(X * np.float32(0.21)) * np.cos(Z * np.float32(0.17)) + np.float32(0.3) * (noise - np.float32(0.5)))).astype(np.float32)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    a = (i * (nz + 1) + j).ravel()
    b, c, d = a + (nz + 1), a + 1, a + (nz + 1) + 1
    idx = np.concatenate([np.stack([a, b, c], 1), np.stack([c, b, d], 1)], 0).astype(np.uint32)
    return v, idx


def config_c3(count_u=1920, count_v=1080, nx=224, nz=224):
    """BASELINE.json configs[2]: synthetic ~100k-triangle mesh (BVH path), 1920x1080, 256 spp.  No reference counterpart."""
    from .engine import mesh_object

    v, idx = heightfield_mesh(nx, nz)
    objs = np.array([mesh_object(0, (0.0, -10.0, 0.0), abi.RT_MAT_MIRROR, 0.0, rotation_y_deg=15.0)] + list(sphere_scene(50, (40.0, 8.0, 40.0), seed=3)),
                    dtype=abi.OBJECT_DTYPE)
    objs["position"][1:, 1] += np.float32(6.0)
    gen = projector((0.0, 40.0, -120.0), (-64.0, -40.0, -40.0), (128.0, 0.0, 0.0), (0.0, 72.0, 0.0), count_u, count_v)
    return dict(objects=objs, meshes=[(v, idx)], gen=gen, spp=256, max_bounces=8, infinity_distance=50.0, seed=0xC0FFEE, name="C3")
