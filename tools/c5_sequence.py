"""C5 rehearsal (BASELINE.json configs[4]) on one GPU: writes an n-frame sequence of complete simulation.toml files (the C2
sphere cloud, every sphere moving a little per frame, explicit [[scene.objects]] — 1,000 tables per file), then renders it
with the native driver and with the Python driver and reports frames/s and Mrays/s, host side (TOML parse + flatten +
upload + PFM write) included.  usage: tools/c5_sequence.py OUT_DIR [n_frames] [spp] [width] [height]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytracing_amd import abi, synth

out = sys.argv[1]
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
w = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
h = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
fdir, ndir, pdir = (os.path.join(out, d) for d in ("frames", "native", "python"))
for d in (fdir, ndir, pdir):
    os.makedirs(d, exist_ok=True)
cfg = synth.config_c2(w, h)
g, objs = cfg["gen"], cfg["objects"]
mat = {abi.RT_MAT_GLASS: 'type = "Glass", ior = {p!r}', abi.RT_MAT_MIRROR: 'type = "Mirror"', abi.RT_MAT_HALF_MIRROR: 'type = "HalfMirror", reflectance = {p!r}'}
t0 = time.perf_counter()
for f in range(n_frames):
    lines = ["[simulation_settings]", f"infinity_distance = {cfg['infinity_distance']!r}", f"max_bounces = {cfg['max_bounces']}", "",
             "[[scene.ray_generators]]", 'type = "Projector"', f"origin = {list(g.a)!r}", f"target_corner = {list(g.corner)!r}",
             f"target_u = {list(g.u)!r}", f"target_v = {list(g.v)!r}", f"count_u = {g.count_u}", f"count_v = {g.count_v}", "current_ior = 1.0", ""]
    phase = 0.05 * f
    for k, o in enumerate(objs):
        p = o["position"].astype(np.float64) + np.array([np.sin(phase + k), 0.3 * np.cos(phase + 0.7 * k), np.cos(phase + k)]) * 0.5
        lines += ["[[scene.objects]]", f'shape = {{ type = "Sphere", radius = {float(o["shape_param"][0])!r} }}',
                  "material = { " + mat[int(o["material_kind"])].format(p=float(o["material_param"])) + " }",
                  f"transform = {{ position = [{float(np.float32(p[0]))!r}, {float(np.float32(p[1]))!r}, {float(np.float32(p[2]))!r}], rotation_y_deg = 0.0 }}"]
    with open(os.path.join(fdir, f"frame_{f:05d}.toml"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
print(f"wrote {n_frames} frame files ({os.path.getsize(os.path.join(fdir, 'frame_00000.toml')) / 1e3:.0f} kB each) in {time.perf_counter() - t0:.1f} s", flush=True)

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
binp = os.path.join(root, "raytracing_amd", "bin", "raytracing")
t0 = time.perf_counter()
r = subprocess.run([binp, "--render", str(spp), "--frames", fdir, "--out", ndir], capture_output=True, text=True, timeout=600)
dt = time.perf_counter() - t0
print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
segs = int(r.stdout.split("frames,")[1].split()[0]) if "frames," in r.stdout else 0
print(f"native: {n_frames} frames in {dt:.2f} s = {n_frames / dt:.1f} frames/s, {segs / dt / 1e6:.0f} Mrays/s (process start, hipInit and all host work included)", flush=True)

from raytracing_amd import engine, frames

engine.init(0)
t0 = time.perf_counter()
done = frames.render_frames(fdir, pdir, spp)
dt = time.perf_counter() - t0
print(f"python: {len(done)} frames in {dt:.2f} s = {len(done) / dt:.1f} frames/s, {sum(d[2] for d in done) / dt / 1e6:.0f} Mrays/s", flush=True)
same = all(open(os.path.join(ndir, n), "rb").read() == open(os.path.join(pdir, n), "rb").read() for n in sorted(os.listdir(ndir)))
print("native and python images identical:", same)
