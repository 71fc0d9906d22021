"""Frame time of the Projector render path on one GPU (HIP events around rt_hip_render_dev), for quick A/B runs.
usage: time_frame.py [c2|c3|c4] [width height spp depth reps cell_step]   (cell_step N: rank 0's share of a frame dealt round-robin to N ranks)   (RT_HIP_NO_GRID=1 etc. are read by the library)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raytracing_amd import engine, sharding, synth
from raytracing_amd.engine import SimulationSettingsConfig

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
make = {"c2": synth.config_c2, "c3": synth.config_c3, "c4": synth.config_c4}[name]
w = int(sys.argv[2]) if len(sys.argv) > 2 else (3840 if name == "c4" else 1920)
h = int(sys.argv[3]) if len(sys.argv) > 3 else (2160 if name == "c4" else 1080)
cfg = make(w, h)
spp = int(sys.argv[4]) if len(sys.argv) > 4 else cfg["spp"]
depth = int(sys.argv[5]) if len(sys.argv) > 5 else cfg["max_bounces"]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
step = int(sys.argv[7]) if len(sys.argv) > 7 else 1
engine.init(0)
sc = engine.Scene(cfg["objects"], meshes=cfg.get("meshes"))
st = SimulationSettingsConfig(cfg["infinity_distance"], depth)
n = (w * h + step - 1) // step
fb = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
chunk = 1 << 21


def frame():
    for b in range(0, n, chunk):
        c = min(chunk, n - b)
        sc.render_dev(cfg["gen"], spp, st, cfg["seed"], b * step, c, step, fb[b:].data_ptr(), 0, 0, 0, tot.data_ptr(), stream)


frame()
torch.cuda.synchronize()
t_end = __import__("time").perf_counter() + 0.5
while __import__("time").perf_counter() < t_end and name == "c2":
    frame()
    torch.cuda.synchronize()
ts = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    frame()
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(f"{name} {w}x{h} spp {spp} depth {depth} step {step}: median {np.median(ts):.3f} ms (min {min(ts):.3f}); last launch segments {int(tot.item())}; "
      f"checksum {sharding.frame_checksum(fb):#018x}", flush=True)
