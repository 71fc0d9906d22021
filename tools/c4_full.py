"""One full C4 frame (BASELINE.json configs[3]: 10,000 spheres, 3840x2160, 1024 spp, depth 8) on one GPU, rendered as
`bands` consecutive cell ranges (what one rank of a tile-sharded job would run, one after the other).  Host wall per band."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig

bands = int(sys.argv[1]) if len(sys.argv) > 1 else 16
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
engine.init(0)
cfg = synth.config_c4(3840, 2160)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
n_cells = 3840 * 2160
per = (n_cells + bands - 1) // bands
tot_seg, tot_t = 0, 0.0
checksum = np.zeros(3, np.float64)
for b in range(bands):
    begin, count = b * per, min(per, n_cells - b * per)
    t0 = time.perf_counter()
    fb, fid, ft, seg, tot = sc.render(cfg["gen"], spp, st, seed=cfg["seed"], cell_begin=begin, cell_count=count)
    dt = time.perf_counter() - t0
    tot_seg += tot
    tot_t += dt
    checksum += fb.sum(0, dtype=np.float64)
    print(f"band {b}: {count} cells, {tot} segments, {dt:.3f} s -> {tot / dt / 1e6:.0f} Mrays/s", flush=True)
n_sph = sc.counts()[1]
print(f"C4 frame: {tot_seg} segments x {n_sph} spheres in {tot_t:.2f} s = {tot_seg / tot_t / 1e6:.0f} Mrays/s, "
      f"{tot_seg * n_sph * 17 / tot_t / 1e12:.1f} algorithmic TFLOP/s; fb checksum {checksum.tolist()}", flush=True)
