"""Throughput of the reference's own call shape — explicit rays in, polylines out (rt_hip_simulate_dev) — on the C2 scene.
Rays: the C2 projector's primary rays, `reps` copies (distinct ray ids, so HalfMirror draws differ).  Device-resident, HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raytracing_amd import abi, engine, synth
from raytracing_amd.engine import SimulationSettingsConfig

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # only the first `limit` rays (fixed cost of a call)
engine.init(0)
cfg = synth.config_c2(1920, 1080)
g = cfg["gen"]
# primary rays on the host (scene_config.rs:84-108), vectorised in f32
i, j = np.meshgrid(np.arange(g.count_u, dtype=np.float32), np.arange(g.count_v, dtype=np.float32), indexing="ij")
a, corner = np.array(g.a[:], np.float32), np.array(g.corner[:], np.float32)
us, vs = (np.array(g.u[:], np.float32) / np.float32(g.count_u)), (np.array(g.v[:], np.float32) / np.float32(g.count_v))
target = (corner + i[..., None] * us).astype(np.float32) + j[..., None] * vs
d = (target - a).astype(np.float32).reshape(-1, 3)
d /= np.sqrt((d * d).sum(1, keepdims=True)).astype(np.float32)
rays = np.zeros(len(d) * reps, abi.RAY_DTYPE)
rays["origin"] = a
rays["direction"] = np.tile(d, (reps, 1))
rays["current_ior"] = 1.0
if limit:
    rays = rays[:limit].copy()
n = len(rays)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
dev = torch.device("cuda", 0)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, -1)).to(dev)
d_pts = torch.empty((n, st.max_bounces + 1, 3), dtype=torch.float32, device=dev)
d_np = torch.empty(n, dtype=torch.int32, device=dev)
d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps + 1)]
for k in range(steps + 1):
    ev[k][0].record()
    sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), 0, d_tot.data_ptr(), stream)
    ev[k][1].record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev][1:]
segs = int(d_tot.item())
byts = n * 28 + int(d_np.sum().item()) * 12 + n * 4
print(f"{n} rays, {segs} segments per call, {np.mean(ms):.2f} ms (min {min(ms):.2f}) -> {segs / np.mean(ms) / 1e3:.0f} Mrays/s, "
      f"{n / np.mean(ms) / 1e3:.0f} Mpaths/s; algorithmic HBM bytes {byts / 1e6:.0f} MB -> {byts / np.mean(ms) / 1e6:.0f} GB/s", flush=True)
