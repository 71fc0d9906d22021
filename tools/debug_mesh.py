"""Diagnostic: GPU BVH mesh path vs oracle brute force, one mesh at a time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raytracing_amd import abi, engine, synth
from raytracing_amd.engine import mesh_object
from raytracing_amd.config import SimulationSettingsConfig
from oracle import binding as ob
from helpers import random_rays

engine.init(0)
rng = np.random.default_rng(1)
n = 3000
c = rng.uniform(-15, 15, size=(n, 1, 3))
soup = ((c + rng.normal(size=(n, 3, 3)) * 3).astype(np.float32).reshape(-1, 3), np.arange(n * 3, dtype=np.uint32).reshape(-1, 3))
hf = synth.heightfield_mesh(40, 40, extent=(20.0, 20.0), amplitude=3.0)
dup = (np.array([[-5, -5, 0], [5, -5, 0], [0, 5, 0]] * 2, np.float32), np.array([[0, 1, 2], [3, 4, 5]], np.uint32))
one = (np.array([[-5, -5, 0], [5, -5, 0], [0, 5, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
rays = random_rays(8000, seed=17, extent=(28.0, 14.0, 28.0))
for name, mesh, rot, mat, nb in [("one", one, 0.0, abi.RT_MAT_MIRROR, 1), ("dup", dup, 0.0, abi.RT_MAT_MIRROR, 1), ("hf", hf, 0.0, abi.RT_MAT_MIRROR, 1),
                                 ("soup", soup, 0.0, abi.RT_MAT_MIRROR, 1), ("soup-rot", soup, 20.0, abi.RT_MAT_MIRROR, 1),
                                 ("soup-6", soup, 0.0, abi.RT_MAT_MIRROR, 6), ("soup-glass", soup, 0.0, abi.RT_MAT_GLASS, 6)]:
    objs = np.array([mesh_object(0, (0, 0, 0), mat, 1.4, rotation_y_deg=rot)], dtype=abi.OBJECT_DTYPE)
    sc = engine.Scene(objs, meshes=[mesh])
    st = SimulationSettingsConfig(50.0, nb)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=2)
    opts, onpts, orecs, _ = ob.OracleScene(objs, meshes=[mesh]).simulate(rays, 50.0, nb, seed=2)
    recs = recs.reshape(len(rays), -1); orecs = orecs.reshape(len(rays), -1)
    bad = [r for r in range(len(rays)) if npts[r] != onpts[r] or recs[r].tobytes() != orecs[r].tobytes()]
    print(name, "hits gpu", int((recs["obj_id"][:, 0] >= 0).sum()), "oracle", int((orecs["obj_id"][:, 0] >= 0).sum()), "bad rays", len(bad), flush=True)
    for r in bad[:3]:
        print("  ray", r, rays[r])
        for b in range(recs.shape[1]):
            if recs[r, b].tobytes() != orecs[r, b].tobytes():
                print("   bounce", b, "\n    gpu", recs[r, b], "\n    orc", orecs[r, b])
                break
