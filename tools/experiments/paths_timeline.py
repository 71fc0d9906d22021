"""Per-wave timeline of paths_kernel (library built with -DRT_DBG_TIMELINE: every wave writes {entry, loop start, own work done, drain
start, end} (100 MHz wall clock), its segments, drain rounds + 1 and its block into the records buffer).
usage: RT_HIP_LIBRARY=build/ab/tl.so python tools/experiments/paths_timeline.py [copies]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engine.init(0)
cfg = synth.config_c2(1920, 1080)
rays = bench.c2_primary_rays(cfg, copies)
n = len(rays)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
dev = torch.device("cuda", 0)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, -1)).to(dev)
d_pts = torch.empty((n, st.max_bounces + 1, 3), dtype=torch.float32, device=dev)
d_np = torch.empty(n, dtype=torch.int32, device=dev)
d_rec = torch.zeros(4096 * 8 + 64, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), 0, 0, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), d_rec.data_ptr(), 0, s)
e1.record()
torch.cuda.synchronize()
t = d_rec.cpu().numpy()[: 4096 * 8].reshape(4096, 8)
t = t[t[:, 0] != 0]
x0 = t[:, 0].min()
us = lambda v: (v - x0) / 100.0
print(f"call {e0.elapsed_time(e1) * 1e3:.0f} us by events; {len(t)} waves; kernel span by wave clocks {us(t[:, 4].max()):.0f} us")
for name, col in (("entry", 0), ("loop start (tables staged)", 1), ("own work done", 2), ("drain start (block barrier passed)", 3), ("end", 4)):
    v = us(t[:, col])
    print(f"  {name:36s} min {v.min():7.1f}  p10 {np.percentile(v, 10):7.1f}  median {np.median(v):7.1f}  p90 {np.percentile(v, 90):7.1f}  max {v.max():7.1f} us")
own = (t[:, 2] - t[:, 1]) / 100.0
wait = (t[:, 3] - t[:, 2]) / 100.0
drain = (t[:, 4] - t[:, 3]) / 100.0
print(f"  per wave: own work mean {own.mean():.1f} us (p10 {np.percentile(own, 10):.1f}, p90 {np.percentile(own, 90):.1f}); barrier wait mean {wait.mean():.1f}; drain mean {drain.mean():.1f} max {drain.max():.1f}; "
      f"rounds mean {t[:, 6].mean() - 1:.1f} max {t[:, 6].max() - 1}; segments per wave mean {t[:, 5].mean():.0f} max {t[:, 5].max()}")
fresh, cont, walked = t[:, 7] & 0xffff, (t[:, 7] >> 16) & 0xffff, t[:, 7] >> 32
print(f"  steps per wave: fresh mean {fresh.mean():.1f} (min {fresh.min()}, max {fresh.max()}), of them walked (no bundle list) {walked.mean():.2f}; continuing mean {cont.mean():.1f} max {cont.max()}; segments mean {t[:, 5].mean():.0f}")
order = np.argsort(-t[:, 4])[:12]
print("  last waves to end: end us, own work us, fresh, walked, continuing, segments")
for k in order:
    print(f"    {us(t[k, 4]):7.1f} {own[k]:7.1f} {fresh[k]:4d} {walked[k]:4d} {cont[k]:4d} {t[k, 5]:6d}")
dur = (t[:, 2] - t[:, 1]) / 100.0
A = np.stack([fresh - walked, walked, cont, np.ones(len(t))], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print(f"  least squares: own work us = {coef[0]:.1f} x listed fresh + {coef[1]:.1f} x walked fresh + {coef[2]:.1f} x continuing + {coef[3]:.1f}")
