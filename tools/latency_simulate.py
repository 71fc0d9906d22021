"""Latency of the reference's own call at the reference's own sizes: rt_hip_simulate (host buffers in, polylines out) for the C1
scene (3 rays, 27 objects) and a template.toml-like scene (1,369 rays, 19 objects incl. CSG), median of 200 calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from raytracing_amd import engine
from raytracing_amd.config import SimulationConfig
from helpers import TEMPLATE_LIKE_TOML

engine.init(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name, text in [("C1 (simulation.toml)", open(os.path.join(root, "tests", "golden", "c1_scene.toml"), encoding="utf-8").read()),
                   ("template-like", TEMPLATE_LIKE_TOML)]:
    cfg = SimulationConfig.from_toml_str(text)
    t0 = time.perf_counter()
    scene = cfg.scene.into_scene()
    t_build = time.perf_counter() - t0
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        paths = scene.simulate_rays_raw(cfg.simulation_settings, records=False)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts[5:]) * 1e6
    print(f"{name}: {len(scene.rays)} rays, {len(scene.objects)} objects; scene build {t_build * 1e3:.2f} ms; "
          f"simulate_rays median {np.median(ts):.0f} us (min {ts.min():.0f}, p95 {np.percentile(ts, 95):.0f})", flush=True)
