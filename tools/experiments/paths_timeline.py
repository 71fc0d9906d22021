"""Per-wave timeline of paths_kernel (library built with -DRT_DBG_PTIME=9: every wave writes {entry, loop start, end, steps, prep cycles,
super-batches, end of first prep} into the records buffer).  usage: RT_HIP_LIBRARY=build/ab/pt9.so python tools/experiments/paths_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig
engine.init(0)
cfg = synth.config_c2(1920, 1080)
rays = bench.c2_primary_rays(cfg, 2)
n = len(rays)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
dev = torch.device("cuda", 0)
d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, -1)).to(dev)
d_pts = torch.empty((n, st.max_bounces + 1, 3), dtype=torch.float32, device=dev)
d_np = torch.empty(n, dtype=torch.int32, device=dev)
d_rec = torch.zeros(4096 * 8 + 64, dtype=torch.int64, device=dev)
sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
sc.simulate_rays_dev(st, d_rays.data_ptr(), n, cfg["seed"], 0, d_pts.data_ptr(), d_np.data_ptr(), d_rec.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t = d_rec.cpu().numpy()[: 4096 * 8].reshape(4096, 8)
# clock domains: timestamps of different XCDs are offset by billions of cycles; group waves whose entry times lie within 10 M cycles
key = t[:, 0] // 2_000_000_000
for k in np.unique(key):
    x = t[key == k]
    x0 = x[:, 0].min()
    life = x[:, 2] - x[:, 1]
    print(f"domain {k}: {len(x)} waves; entry spread {x[:,0].max()-x0}; staging mean {np.mean(x[:,1]-x[:,0]):.0f}; first prep done at mean {np.mean(x[:,6]-x0):.0f} max {np.max(x[:,6]-x0)}; "
          f"end min {x[:,2].min()-x0} median {np.median(x[:,2]-x0):.0f} max {x[:,2].max()-x0}; loop mean {life.mean():.0f}; steps mean {x[:,3].mean():.1f}; "
          f"prep mean {x[:,4].mean():.0f} over {x[:,5].mean():.1f} super-batches")
