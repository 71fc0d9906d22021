"""Frame time of a ParallelGrid render (orthographic view of the C2 cloud, 1920x1080 cells) on the general kernel, and the same view through
the Projector kernel for scale.  usage: time_parallel_grid.py [spp reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig, parallel_grid

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
engine.init(0)
cfg = synth.config_c2(1920, 1080)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
gens = {"ParallelGrid": parallel_grid((-64.0, -36.0, -120.0), (128.0, 0, 0), (0, 72.0, 0), 1920, 1080, (0.0, 0.0, 1.0)), "Projector": cfg["gen"]}
n = 1920 * 1080
fb = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
tot = torch.zeros(1, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, gen in gens.items():
    ms = []
    for k in range(reps + 1):
        tot.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sc.render_dev(gen, spp, st, cfg["seed"], 0, n, 1, fb.data_ptr(), 0, 0, 0, tot.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms = ms[1:]
    print(f"{name}: {spp} spp, {int(tot.item())} segments, median {np.median(ms):.2f} ms -> {int(tot.item()) / np.median(ms) / 1e3:.0f} Mrays/s", flush=True)
