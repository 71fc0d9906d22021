"""Small driver for rocprofv3: a few C2 render launches, nothing else (no torch)."""
import ctypes as C
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cfgname = sys.argv[5] if len(sys.argv) > 5 else "c2"
depth = int(sys.argv[6]) if len(sys.argv) > 6 else None
engine.init(0)
cfg = {"c2": synth.config_c2, "c3": synth.config_c3, "c4": synth.config_c4}[cfgname](w, h)
t0 = __import__("time").perf_counter()
sc = engine.Scene(cfg["objects"], meshes=cfg.get("meshes"))
print(f"scene build {(__import__('time').perf_counter() - t0)*1e3:.1f} ms", flush=True)
st = SimulationSettingsConfig(cfg["infinity_distance"], depth or cfg["max_bounces"])
import time
for r in range(reps):
    t0 = time.perf_counter()
    fb, fid, ft, seg, tot = sc.render(cfg["gen"], spp, st, seed=cfg["seed"])
    dt = time.perf_counter() - t0
    print(f"rep {r}: {tot} segments, mesh-hit pixels {(fid == 0).mean():.2f}, host wall {dt*1e3:.1f} ms (includes alloc+copies)", flush=True)
