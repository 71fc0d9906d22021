"""The simulate call shape on the 10,000-sphere C4 scene (tables beyond an LDS tile): the C4 projector's primary rays (3840x2160) as explicit rays.
usage: bench_simulate_c4.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
engine.init(0)
cfg = synth.config_c4(3840, 2160)
rays = bench.c2_primary_rays(cfg, 1)
n = len(rays)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
dev = torch.device("cuda", 0)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, -1)).to(dev)
d_pts = torch.empty((n, st.max_bounces + 1, 3), dtype=torch.float32, device=dev)
d_np = torch.empty(n, dtype=torch.int32, device=dev)
d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
ms = []
for k in range(steps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), 0, d_tot.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
segs = int(d_tot.item())
print(f"C4 scene: {n} rays, {segs} segments per call, {np.mean(ms[1:]):.2f} ms -> {segs / np.mean(ms[1:]) / 1e3:.0f} Mrays/s", flush=True)
