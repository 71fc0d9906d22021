#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-path hot loop (BASELINE.json metric).

A step = one pass of the hot path over one frame (or one batch of explicit rays) of synthetic input.  A "ray" = one
bounce-loop iteration (scene.rs:49), counted by the kernel.

  --gpus 1 (default)  workload C2 = BASELINE.json configs[1], the configuration the metric is quoted on: 1,000 spheres,
                      Projector 1920x1080 cells, 64 paths per cell, max_bounces 8, one launch per frame.
                      The line also carries `config.secondary`: the reference's own call shape (explicit rays in,
                      polylines out, rt_hip_simulate_dev) on the same scene.
  --gpus N > 1        (launched by torch.distributed.run, one rank per GPU) workload C4 = configs[3]: 10,000 spheres,
                      3840x2160 cells, 1,024 spp, the frame's cells dealt round-robin over the ranks
                      (sharding.cell_shard), ONE all_gather_into_tensor over RCCL per frame inside the timed region, fixed
                      total work ("scaling": "strong").  Rank 0 checks the gathered image against the stored single-GPU
                      checksum (tests/golden/frame_checksums.json) after the timed region.
  --workload c2|c3|c4|simulate   overrides the choice: `c4` on one GPU = the single-GPU point of the scaling curve; `c3` =
                      configs[2] (100k-triangle mesh, BVH path, 256 spp); `simulate` = Scene::simulate_rays' call shape
                      (scene.rs:39): 4,147,200 explicit rays on the C2 scene, polylines written to HBM.

roofline: EXECUTED work.  The numerator is the wave-level VALU instruction count of this workload's
frame, by instruction class, from the committed rocprofv3 PMC pass (profiles/pmc_constants.json); it is used only if that
pass was taken from the very sources this library was built from (SHA-256 over raytracing_amd/csrc, tools/csrc_hash.py)
and for this width / height / spp / depth — otherwise every PMC-derived field is null and `pmc_stale` says why.  The
ceiling weights that instruction mix with the issue cost of each class measured on this chip
(profiles/valu_class_cycles.json, tools/ubench/valu_rate.hip).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOP_PER_TEST = 17  # algorithmic reject-path cost of one ray-sphere test (sphere.rs:13-17)
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (v_fma_f32 every 2 cycles per SIMD: 256 FLOP/clk/CU) at 2.4 GHz
N_SIMD = 256 * 4
CLOCK_GHZ = 2.4
PEAK_VALU_GINST_FP32_RATE = N_SIMD * CLOCK_GHZ / 2.0  # 1,228.8: every instruction at the FMA rate (one per 2 cycles per SIMD)
PEAK_VALU_GINST_4CYCLE = N_SIMD * CLOCK_GHZ / 4.0     # 614.4: rounds 1-2 priced against this (one per 4 cycles)
CHECKSUMS = os.path.join(ROOT, "tests", "golden", "frame_checksums.json")
PMC = os.path.join(ROOT, "profiles", "pmc_constants.json")
RATES = os.path.join(ROOT, "profiles", "valu_class_cycles.json")
CLASSES = ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "INT64", "CVT", "OTHER")
COHERENCE_NOTE = ("the reference's Projector has no jitter (scene_config.rs:84-108), so a cell's samples start as identical primary "
                  "rays and diverge only at HalfMirrors; every sample's ray is traced (nothing is reused), but cull lists and grid "
                  "walks ride that coherence -- a jittered tracer would not see this rate")


def cpu_baseline(n_spheres):
    """Oracle (C++ restatement, kind 'port') on the host cores, on a bounded sample of the C2 workload."""
    from oracle import binding as ob
    from raytracing_amd import synth

    small = synth.config_c2(192, 108)
    osc = ob.OracleScene(small["objects"])
    spp = 128  # ~10-15 s of single-thread work
    t0 = time.perf_counter()
    *_, total = osc.render(small["gen"], spp, small["infinity_distance"], small["max_bounces"], seed=small["seed"], threads=1)
    dt1 = time.perf_counter() - t0
    threads = min(16, os.cpu_count() or 1)
    t0 = time.perf_counter()
    *_, total_mt = osc.render(small["gen"], spp * 2, small["infinity_distance"], small["max_bounces"], seed=small["seed"], threads=threads)
    dtn = time.perf_counter() - t0
    return {
        "value": total / dt1 / 1e6,
        "unit": "Mrays/s",
        "cores": 1,
        "kind": "port",
        "sample": f"C2 scene ({n_spheres} spheres), 192x108 cells x {spp} spp, depth 8: {total} segments in {dt1:.2f} s, 1 thread "
                  "(C++ restatement of the reference; the Rust binary cannot be built here)",
        "all_cores": {"value": total_mt / dtn / 1e6, "cores": threads, "sample": f"192x108 x {spp * 2} spp, {total_mt} segments in {dtn:.2f} s"},
    }


def checksum_key(name, w, h, spp, seed, depth):
    return f"{name}:{w}x{h}:spp{spp}:seed{seed}:depth{depth}"


def load_json(path):
    try:
        return json.load(open(path))
    except Exception:
        return {}


def pmc_entry(name, run_cfg):
    """(entry, None) if profiles/pmc_constants.json[name] was collected from these sources and this configuration, else ({}, reason)."""
    from csrc_hash import csrc_hash

    e = load_json(PMC).get(name)
    if not e:
        return {}, f"no entry '{name}' in profiles/pmc_constants.json"
    have = csrc_hash()
    if e.get("csrc_sha256") != have:
        return {}, f"profiles/pmc_constants.json[{name}] was collected from sources {str(e.get('csrc_sha256'))[:12]}, this tree is {have[:12]}: re-run tools/profile_bench.sh"
    if e.get("config") != run_cfg:
        return {}, f"profiles/pmc_constants.json[{name}] was collected for {e.get('config')}, this run is {run_cfg}"
    return e, None


def valu_roofline(pmc, stale, rates, render_ms, launches, share, segs_local, kernel_label):
    """The executed-work roofline of one workload.  render_ms: HIP-event time of one step's launches."""
    sec = render_ms * 1e-3
    insts = None if stale else pmc.get("valu_insts_per_frame")
    mix = None if stale else pmc.get("valu_class_per_frame")
    achieved = None if insts is None else insts * share / sec / 1e9
    ceiling = mean_cycles = None
    if mix and rates:
        total = sum(mix.get(c, 0.0) for c in CLASSES)
        if total > 0 and all(c in rates for c in CLASSES):
            mean_cycles = sum(mix.get(c, 0.0) * rates[c] for c in CLASSES) / total
            ceiling = N_SIMD * CLOCK_GHZ / mean_cycles
    traffic = None if stale else pmc.get("hbm_bytes_per_frame")
    clk = None if stale else pmc.get("clock_ghz")
    issue, lane = (None, None) if stale else (pmc.get("valu_issue_util"), pmc.get("valu_lane_util"))
    flops = None if stale else pmc.get("flops_fp32_per_frame")
    roof = {
        "bound": "valu_issue",
        "kernel": kernel_label,
        "achieved": achieved,
        "peak": ceiling,
        "unit": "Ginst/s",
        "frac": None if achieved is None or ceiling is None else achieved / ceiling,
        "traffic": None if traffic is None else traffic * share / launches,
        "kernel_ms": render_ms / launches,
        "launches_per_step": launches,
        "definition": "EXECUTED work: wave-level VALU instructions of this rank's share of the frame (SQ_INSTS_VALU of the committed "
                      "rocprofv3 pass of this kernel, workload and source hash, profiles/pmc_constants.json) / its kernel time (HIP "
                      "events on the launch stream).  peak = 1,024 SIMDs x 2.4 GHz / (mean issue cycles of the measured instruction mix: "
                      "SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F32, _INT32, _INT64, _CVT, rest = OTHER, each weighted with the cycles per "
                      "wave instruction measured on this chip by tools/ubench/valu_rate.hip, profiles/valu_class_cycles.json).  "
                      "frac_fp32_rate prices every instruction at the FP32 FMA rate (2 cycles, MI355X_MICROARCH.md: 1,228.8 Ginst/s), "
                      "frac_4cycle at the 4-cycle rate rounds 1-2 used (614.4).",
        "mean_issue_cycles_of_mix": mean_cycles,
        "frac_fp32_rate": None if achieved is None else achieved / PEAK_VALU_GINST_FP32_RATE,
        "frac_4cycle": None if achieved is None else achieved / PEAK_VALU_GINST_4CYCLE,
        "valu_class_mix": None if not mix else {c: mix.get(c, 0.0) / max(1.0, sum(mix.get(k, 0.0) for k in CLASSES)) for c in CLASSES},
        "valu_class_cycles": rates or None,
        "valu_insts_per_segment": None if insts is None or not segs_local else insts * share / segs_local,
        # SQ_INSTS_VALU_FLOPS_FP32 counts per WAVE instruction (1 per add / mul, 2 per fma: it equals ADD + MUL + 2 FMA of the mix
        # pass), so the operations executed are 64 x that with every lane counted, and x valu_lane_util for the lanes that were live
        "executed_fp32": None if flops is None else {"achieved": 64.0 * flops * share / sec / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                                     "frac": 64.0 * flops * share / sec / 1e12 / PEAK_FP32_TFLOPS,
                                                     "frac_live_lanes": None if lane is None else 64.0 * flops * share * lane / sec / 1e12 / PEAK_FP32_TFLOPS,
                                                     "note": "64 lanes x SQ_INSTS_VALU_FLOPS_FP32 (+ _TRANS) of the same pass: FP32 operations the "
                                                             "kernel really executed, over the packed-FMA vector peak (north_star's 50 % bar). "
                                                             "Low by construction: half of the instructions are compares, selects, moves and "
                                                             "integer work, and the reference's arithmetic allows no FMA contraction"},
        "frac_at_sustained_clock": None if not clk or achieved is None or ceiling is None else achieved / (ceiling * clk / CLOCK_GHZ),
        "sustained_clock_ghz": clk,
        "valu_issue_util": issue,
        "valu_lane_util": lane,
        "useful_lane_fraction": None if issue is None or lane is None else issue * lane,
        "lane_util_note": "valu_lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): lanes ENABLED in the exec mask, not lanes with "
                          "work.  The grid walk's pass is straight-line code since late round 3 (entries a lane does not have are computed and "
                          "dropped), so this reads higher than the share of lanes doing useful work; the walk's own occupancy (lanes still "
                          "walking per pass, -DRT_DBG_WALK counters) is 0.61 on C2 (DESIGN.md 4.2b, 4.8)",
        "pmc_source": None if stale else pmc.get("source"),
        "pmc_stale": stale,
        "hbm": None if traffic is None else {"achieved": traffic * share / sec / 1e9, "peak": 8000.0, "unit": "GB/s",
                                              "frac": traffic * share / sec / 1e9 / 8000.0},
    }
    return roof


def c2_primary_rays(cfg, copies):
    """The C2 projector's primary rays as explicit RtRays (scene_config.rs:84-108, vectorised in f32 on the host), `copies` times."""
    import numpy as np

    from raytracing_amd import abi

    g = cfg["gen"]
    i, j = np.meshgrid(np.arange(g.count_u, dtype=np.float32), np.arange(g.count_v, dtype=np.float32), indexing="ij")
    a, corner = np.array(g.a[:], np.float32), np.array(g.corner[:], np.float32)
    us, vs = (np.array(g.u[:], np.float32) / np.float32(g.count_u)), (np.array(g.v[:], np.float32) / np.float32(g.count_v))
    target = (corner + i[..., None] * us).astype(np.float32) + j[..., None] * vs
    d = (target - a).astype(np.float32).reshape(-1, 3)
    d /= np.sqrt((d * d).sum(1, keepdims=True)).astype(np.float32)
    rays = np.zeros(len(d) * copies, abi.RAY_DTYPE)
    rays["origin"] = a
    rays["direction"] = np.tile(d, (copies, 1))
    rays["current_ior"] = 1.0
    return rays


class SimulateJob:
    """rt_hip_simulate_dev over this rank's contiguous share of the explicit rays; everything resident in HBM."""

    def __init__(self, scene, cfg, st, dev, rank, world, copies=2):
        import numpy as np
        import torch

        rays = c2_primary_rays(cfg, copies)
        self.n_total = len(rays)
        lo, hi = self.n_total * rank // world, self.n_total * (rank + 1) // world
        self.n, self.base = hi - lo, lo
        self.scene, self.st, self.seed = scene, st, cfg["seed"]
        self.d_rays = torch.from_numpy(rays[lo:hi].view(np.uint8).reshape(self.n, -1).copy()).to(dev)
        self.d_pts = torch.empty((self.n, st.max_bounces + 1, 3), dtype=torch.float32, device=dev)
        self.d_np = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def call(self):
        self.scene.simulate_rays_dev(self.st, self.d_rays.data_ptr(), self.n, self.seed, self.base, self.d_pts.data_ptr(), self.d_np.data_ptr(), 0,
                                     self.d_tot.data_ptr(), self.stream)

    def segments(self):
        return int(self.d_tot.item())

    def algorithmic_bytes(self):
        """28 B per ray in; per path its polyline (12 B per vertex) and vertex count (4 B) out."""
        return self.n * 28 + int(self.d_np.sum().item()) * 12 + self.n * 4


def time_calls(torch, fn, steps, barrier):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        fn(k)
        ev[k][1].record()
    barrier()
    return time.perf_counter() - t0, ev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 100 for C2 and simulate, whose steps take 3 ms and 0.5 ms; 20 otherwise)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["auto", "c2", "c3", "c4", "simulate"], default="auto")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--print-checksum", action="store_true", help="also print the frame checksum key and value (to store it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import engine, sharding, synth
    from raytracing_amd.engine import SimulationSettingsConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, all-gather) even with one rank, to rehearse it on a 1-GPU box
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knob for a 1-GPU box: all ranks share device 0 and the collective runs over gloo on host copies.
    rehearsal = os.environ.get("RT_BENCH_SINGLE_DEVICE") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if (world > 1 and not rehearsal) else 0)
    engine.init(dev.index)

    name = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c4")
    if args.steps is None:
        args.steps = 100 if name in ("c2", "simulate") else 20  # a timed region of 0.3 s rather than 60 ms on a part whose clock moves
    scene_name = "c2" if name == "simulate" else name
    make = {"c2": synth.config_c2, "c3": synth.config_c3, "c4": synth.config_c4}[scene_name]
    w = args.width or (3840 if scene_name == "c4" else 1920)
    h = args.height or (2160 if scene_name == "c4" else 1080)
    cfg = make(w, h)
    spp = args.spp or cfg["spp"]
    scene = engine.Scene(cfg["objects"], meshes=cfg.get("meshes"))
    n_obj, n_sph, n_pl = scene.counts()
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    n_cells = w * h

    class GlooDist:  # rehearsal: the all-gather runs over gloo on host copies of the shards
        @staticmethod
        def all_gather_into_tensor(out, inp):
            ho = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu())
            out.copy_(ho)

    def render_cells(begin, count, step, fb_rows, total_slot):
        scene.render_dev(cfg["gen"], spp, st, cfg["seed"], begin, count, step, fb_rows.data_ptr(), 0, 0, 0, total_slot.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def prewarm(fn):
        # bring the GPU to its steady clocks before anything is counted (a cold first frame runs ~8 % slower); untimed, like the W
        # warmup steps (no collective: the number of pre-warm frames is time-based and may differ between ranks)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.5:
            fn()
            torch.cuda.synchronize()

    sim = None
    if name == "simulate":
        sim = SimulateJob(scene, cfg, st, dev, rank, world)
        prewarm(sim.call)
        for _ in range(args.warmup):
            sim.call()
        elapsed, ev = time_calls(torch, lambda k: sim.call(), args.steps, barrier)
        segs_local = sim.segments()
        launches, share = 1, sim.n / sim.n_total
    else:
        # inputs (scene, generator constants) and outputs resident in HBM before the timed region; C2's 2,073,600 cells are one launch
        frame = sharding.ShardedFrame(render_cells, n_cells, rank, world, GlooDist if rehearsal else dist, dev, max_cells_per_launch=1 << 21)
        prewarm(frame.render)
        for _ in range(args.warmup):
            frame.step()
        barrier()
        # kernel duration: HIP events on the launch stream around each frame's launches (the collective is outside the pair)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record()
            frame.render()
            ev[k][1].record()
            frame.gather()
        barrier()
        elapsed = time.perf_counter() - t0
        segs_local = frame.segments()
        launches, share = len(frame.launches), frame.count / n_cells
    render_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    s = torch.tensor([segs_local], dtype=torch.int64, device=dev)
    if use_dist:
        for x, op in ((t, dist.ReduceOp.MAX), (s, dist.ReduceOp.SUM)):
            if rehearsal:
                hx = x.cpu()
                dist.all_reduce(hx, op=op)
                x.copy_(hx)
            else:
                dist.all_reduce(x, op=op)
    elapsed_max, segs_all = float(t.item()), int(s.item())

    if rank == 0:
        if sim is None:
            # the assembled image against the stored single-GPU checksum (integer checksum: independent of the rank count)
            key = checksum_key(name, w, h, spp, cfg["seed"], cfg["max_bounces"])
            got = sharding.frame_checksum(frame.image())
            stored = load_json(CHECKSUMS)
            if key in stored:
                if int(stored[key], 16) != got:
                    raise SystemExit(f"bench.py: frame checksum {got:#018x} differs from the stored single-GPU value {stored[key]} for {key}")
                image_check = f"gathered image == stored single-GPU checksum {stored[key]}"
            else:
                image_check = f"no stored checksum for {key} (computed {got:#018x})"
            if args.print_checksum:
                print(f"checksum {key} {got:#018x}", file=sys.stderr, flush=True)
        else:
            image_check = "polylines: bit-exact against the oracle in tests/test_gpu_parity.py (test_simulate_*), not re-checked here"

        value = segs_all * args.steps / elapsed_max / 1e6
        run_cfg = {"width": w, "height": h, "spp": spp, "max_bounces": cfg["max_bounces"]} if sim is None else \
                  {"rays": sim.n_total, "max_bounces": cfg["max_bounces"]}
        pmc, stale = pmc_entry(name, run_cfg)
        rates = load_json(RATES).get("cycles")
        if sim is None:
            kernel = ("pinhole_kernel<1, %s, %s> (Projector render loop, one ray per lane)" % ("false" if n_sph <= 1024 else "true", "true" if name == "c3" else "false"))
        else:
            kernel = "paths_kernel<1> (explicit rays in, polylines out: the call shape of scene.rs:39; rt_paths.hip)"
        roof = valu_roofline(pmc, stale, rates, render_ms, launches, share, segs_local, kernel)
        if sim is None and name != "c3":
            algorithmic = segs_local * FLOP_PER_TEST * n_sph / (render_ms * 1e-3) / 1e12
            roof["algorithmic_equiv"] = {"flop_per_test": FLOP_PER_TEST, "achieved": algorithmic, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                         "frac": algorithmic / PEAK_FP32_TFLOPS,
                                         "note": "SURVEY.md §8d accounting: 17 FLOP for every (segment, sphere) pair the reference's linear scan "
                                                 "would test, over the FP32 vector peak; far above 1 because almost none of those tests is "
                                                 "executed (cull lists + uniform grid, results bit-identical to the linear scan) -- a statement "
                                                 "about the algorithm, not a roofline"}
        if sim is not None:
            ab = sim.algorithmic_bytes()
            roof["hbm_algorithmic"] = {"bytes_per_call": ab, "achieved": ab / (render_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                       "frac": ab / (render_ms * 1e-3) / 1e9 / 8000.0,
                                       "note": "28 B per ray in, 12 B per polyline vertex + 4 B per path out (SURVEY §8d)"}
        full = {"c2": "C2", "c3": "C3", "c4": "C4", "simulate": "simulate_rays on the C2 scene"}[name]
        if sim is None:
            workload = (f"{full}: {n_sph} spheres" + (f" + {len(cfg['meshes'][0][1])}-triangle mesh" if name == "c3" else "") +
                        f" (seed 0xC0FFEE), Projector {w}x{h} cells, {spp} spp, max_bounces {cfg['max_bounces']}; one frame per step" +
                        ("" if world == 1 else f", cells dealt round-robin over {world} GPUs, one all-gather per frame") + "; " + COHERENCE_NOTE)
        else:
            workload = (f"{full}: {n_sph} spheres, {sim.n_total} explicit rays (two copies of the 1920x1080 Projector's primary rays), "
                        f"max_bounces {cfg['max_bounces']}, polylines [n][{cfg['max_bounces'] + 1}][3] f32 + vertex counts written to HBM; one call per step")
        out = {
            "metric": "Mrays/sec (primary+bounce) at 1080p 64spp, 1/8 GPU; % of FP32 VALU peak",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": workload,
                **run_cfg,
                "segments_per_step": segs_all,
                "sharding": "single GPU" if world == 1 else ("cell = rank + m * world (sharding.cell_shard); framebuffer shards all-gathered over RCCL inside the timed region"
                                                             if sim is None else "contiguous ray ranges, no exchange"),
                "image_check": image_check,
                # "1/8 GPU" read both ways: value is per whole MI355X (one of a node's eight) at N=1; one XCD's share is value/8/N
                "mrays_per_s_per_gpu": value / world,
                "mrays_per_s_per_xcd": value / world / 8,
                # north_star's second number, two honest readings: FP32 operations really executed over the packed-FMA peak, and
                # VALU instruction issue against the FMA issue rate
                "pct_of_fp32_valu_peak": None if not roof.get("executed_fp32") else 100.0 * roof["executed_fp32"]["frac"],
                "pct_of_valu_issue_at_fp32_rate": None if roof["frac_fp32_rate"] is None else 100.0 * roof["frac_fp32_rate"],
            },
            "roofline": roof,
        }
        if world == 1 and name == "c2" and not args.no_secondary:
            # the reference's own call shape on the same scene, measured after the headline's timed region
            sj = SimulateJob(scene, cfg, st, dev, 0, 1)
            for _ in range(3):
                sj.call()
            n_calls = 10
            el, sev = time_calls(torch, lambda k: sj.call(), n_calls, barrier)
            sms = sum(a.elapsed_time(b) for a, b in sev) / n_calls
            ssegs = sj.segments()
            spmc, sstale = pmc_entry("simulate", {"rays": sj.n_total, "max_bounces": cfg["max_bounces"]})
            sroof = valu_roofline(spmc, sstale, rates, sms, 1, 1.0, ssegs, "paths_kernel<1> (explicit rays, polylines out)")
            out["config"]["secondary"] = {
                "workload": f"Scene::simulate_rays call shape (scene.rs:39, rt_hip_simulate_dev): {sj.n_total} explicit rays on the same scene, polylines out",
                "value": ssegs * n_calls / el / 1e6, "unit": "Mrays/s", "ms_per_call": el / n_calls * 1e3, "kernel_ms": sms,
                "segments_per_call": ssegs, "paths_per_s": sj.n_total * n_calls / el,
                "hbm_algorithmic_gbps": sj.algorithmic_bytes() / (sms * 1e-3) / 1e9,
                "roofline_frac": sroof["frac"], "roofline_frac_fp32_rate": sroof["frac_fp32_rate"], "pmc_stale": sstale,
            }
        if world == 1 and name == "c2" and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n_sph)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
