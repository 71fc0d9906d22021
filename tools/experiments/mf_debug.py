"""Debug driver for the matrix-pipe filter experiment (tools/experiments/mfma_primary_filter.patch).

  mf_debug.py dump OUT.npz          one spp=1 / depth=2 frame of a reduced C2 with the library RT_HIP_LIBRARY names
  mf_debug.py cmp REF.npz [reps]    the same frame `reps` times, every difference against REF.npz listed
  mf_debug.py time [reps]           full C2 frames (1920x1080, 64 spp) timed with HIP events
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from raytracing_amd import engine, synth
from raytracing_amd.engine import SimulationSettingsConfig

mode = sys.argv[1]
engine.init(0)
if mode == "time":
    import torch

    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    cfg = synth.config_c2(1920, 1080)
    sc = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    n = 1920 * 1080
    fb = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
    tot = torch.zeros(1, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def frame():
        sc.render_dev(cfg["gen"], cfg["spp"], st, cfg["seed"], 0, n, 1, fb.data_ptr(), 0, 0, 0, tot.data_ptr(), stream)

    for _ in range(40):
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
    from raytracing_amd import sharding

    print(f"frame ms: median {np.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f}; segments {int(tot.item())}; "
          f"checksum {sharding.frame_checksum(fb):#018x}", flush=True)
    sys.exit(0)

W, H = 640, 432
cfg = synth.config_c2(W, H)
sc = engine.Scene(cfg["objects"])
st = SimulationSettingsConfig(cfg["infinity_distance"], 2)


def frame():
    return sc.render(cfg["gen"], 1, st, seed=cfg["seed"])


if mode == "dump":
    fb, fid, ft, seg, tot = frame()
    np.savez(sys.argv[2], fb=fb, fid=fid, ft=ft, seg=seg, tot=tot)
    print("dumped", sys.argv[2], "segments", tot, flush=True)
else:
    ref = np.load(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    cv = cfg["gen"].count_v if hasattr(cfg["gen"], "count_v") else H
    for rep in range(reps):
        fb, fid, ft, seg, tot = frame()
        bad = np.nonzero((fb != ref["fb"]).any(axis=1) | (fid != ref["fid"]) * (mode == "cap"))[0]
        # debug builds: fid = ring position | lane << 8 | r << 14 | wave << 15 of the continuing ray, ft = its direction x as read
        dpos = np.count_nonzero(fid != ref["fid"])
        dft = np.count_nonzero(ft.view(np.uint32) != ref["ft"].view(np.uint32))
        print(f"rep {rep}: fb diff {len(bad)}, seg diff {np.count_nonzero(seg != ref['seg'])}, pos diff {dpos}, dir.x diff {dft}, tot {tot} vs {ref['tot']}", flush=True)
        for m in bad[:24]:
            if mode == "cap":  # capture builds: fid = first-hit object, ft = the captured quantity of the primary segment
                print(f"   cell {m} ({m // cv},{m % cv}) fb {fb[m]} want {ref['fb'][m]} seg {seg[m]} want {ref['seg'][m]} fid {fid[m]} want {ref['fid'][m]} "
                      f"cap {ft[m]:.9g} want {ref['ft'][m]:.9g} ({ft.view(np.uint32)[m]:#x} / {ref['ft'].view(np.uint32)[m]:#x})", flush=True)
                continue
            f, rf = int(fid[m]) & 0xffffffff, int(ref["fid"][m]) & 0xffffffff
            print(f"   cell {m} ({m // cv},{m % cv}) fb {fb[m]} want {ref['fb'][m]} dir.x {ft[m]:.7g} want {ref['ft'][m]:.7g}; "
                  f"ring e {f & 255} lane {(f >> 8) & 63} r {(f >> 14) & 1} wave {f >> 15} (ref: e {rf & 255} lane {(rf >> 8) & 63} r {(rf >> 14) & 1} wave {rf >> 15})", flush=True)
