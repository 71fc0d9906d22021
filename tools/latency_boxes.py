"""Latency of simulate_rays on an ObjectGrid of rotation-0 boxes (scene_config.rs:25-49): n boxes x 1,369 rays, 10 bounces.
Round 1 walked every box per lane (rt_complex.h); round 2 puts them on the wave-parallel scan behind bounding spheres."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import abi, engine
from raytracing_amd.engine import SimulationSettingsConfig, shape_object, make_rays

engine.init(0)
rng = np.random.default_rng(1)
for n in (19, 100, 1000, 4000):
    side = int(np.ceil(np.sqrt(n)))
    objs = np.array([shape_object(abi.RT_SHAPE_BOX, (1.5, 2.0, 1.0), (3.0 * (k % side) - 1.5 * side, 0.0, 3.0 * (k // side)), abi.RT_MAT_GLASS, 1.5)
                     for k in range(n)], dtype=abi.OBJECT_DTYPE)
    o = np.stack([rng.uniform(-1.5 * side, 1.5 * side, 1369), rng.uniform(-1, 1, 1369), np.full(1369, -10.0)], 1).astype(np.float32)
    d = np.tile(np.array([0.05, 0.01, 1.0], np.float32), (1369, 1))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sc = engine.Scene(objs, make_rays(o, d))
    st = SimulationSettingsConfig(50.0, 10)
    sc.simulate_rays(st)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        sc.simulate_rays(st)
        ts.append(time.perf_counter() - t0)
    print(f"{n} boxes x 1369 rays x 10 bounces: median {np.median(ts) * 1e3:.3f} ms per simulate_rays (host buffers in and out)", flush=True)
