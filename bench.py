#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-path hot loop (BASELINE.json metric).

A step = one pass of the hot path over one frame of synthetic input.  A "ray" = one bounce-loop iteration
(scene.rs:49), counted by the kernel.

  --gpus 1 (default)  workload C2 = BASELINE.json configs[1], the configuration the metric is quoted on: 1,000 spheres,
                      Projector 1920x1080 cells, 64 paths per cell, max_bounces 8, one launch per frame.
  --gpus N > 1        (launched by torch.distributed.run, one rank per GPU) workload C4 = configs[3]: 10,000 spheres,
                      3840x2160 cells, 1,024 spp, the frame's cells dealt round-robin over the ranks
                      (sharding.cell_shard), ONE all_gather_into_tensor over RCCL per frame inside the timed region, fixed
                      total work ("scaling": "strong").  Rank 0 checks the gathered image against the stored single-GPU
                      checksum (tests/golden/frame_checksums.json) after the timed region.
  --workload c2|c4    overrides the choice (e.g. `--gpus 1 --workload c4` = the single-GPU reference of the scaling curve).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TEST = 17  # algorithmic reject-path cost of one ray-sphere test (sphere.rs:13-17)
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (packed FMA) at 2.4 GHz
# VALU issue peak: 256 CUs x 4 SIMDs, one wave-level VALU instruction per 4 cycles per SIMD, 2.4 GHz (MI355X_MICROARCH.md)
PEAK_VALU_GINST = 256 * 4 * 2.4 / 4.0
CHECKSUMS = os.path.join(ROOT, "tests", "golden", "frame_checksums.json")
PMC = os.path.join(ROOT, "profiles", "pmc_constants.json")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["auto", "c2", "c4"], default="auto")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    make = synth.config_c2 if name == "c2" else synth.config_c4
    w = args.width or (1920 if name == "c2" else 3840)
    h = args.height or (1080 if name == "c2" else 2160)
    cfg = make(w, h)
    spp = args.spp or cfg["spp"]
    scene = engine.Scene(cfg["objects"])
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

    # inputs (scene, generator constants) and outputs resident in HBM before the timed region; C2's 2,073,600 cells are one launch
    frame = sharding.ShardedFrame(render_cells, n_cells, rank, world, GlooDist if rehearsal else dist, dev, max_cells_per_launch=1 << 21)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # bring the GPU to its steady clocks before anything is counted (a cold first frame runs ~8 % slower); untimed, like the W
    # warmup steps (render only, no collective: the number of pre-warm frames is time-based and may differ between ranks)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        frame.render()
        torch.cuda.synchronize()
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
        # the assembled image against the stored single-GPU checksum (integer checksum: independent of the rank count)
        key = checksum_key(name, w, h, spp, cfg["seed"], cfg["max_bounces"])
        got = sharding.frame_checksum(frame.image())
        stored = json.load(open(CHECKSUMS)) if os.path.exists(CHECKSUMS) else {}
        if key in stored:
            if int(stored[key], 16) != got:
                raise SystemExit(f"bench.py: frame checksum {got:#018x} differs from the stored single-GPU value {stored[key]} for {key}")
            image_check = f"gathered image == stored single-GPU checksum {stored[key]}"
        else:
            image_check = f"no stored checksum for {key} (computed {got:#018x})"
        if args.print_checksum:
            print(f"checksum {key} {got:#018x}", file=sys.stderr, flush=True)

        value = segs_all * args.steps / elapsed_max / 1e6
        launches = len(frame.launches)
        kernel_ms = render_ms / launches  # average duration of one launch of the dominant kernel (+ the table, tile-list and finalize kernels and memsets beside it: 2 % on C2, 0.4 % on C4)
        algorithmic = segs_local * FLOP_PER_TEST * n_sph / (render_ms * 1e-3) / 1e12
        pmc = {}
        if os.path.exists(PMC):
            try:
                pmc = json.load(open(PMC)).get(name, {})
            except Exception:
                pmc = {}
        traffic = pmc.get("hbm_bytes_per_frame")
        clk = pmc.get("clock_ghz")
        share = frame.count / n_cells  # this rank's part of the frame (cells are dealt round-robin: a uniform sample of the image)
        insts = pmc.get("valu_insts_per_frame")
        achieved = None if insts is None else insts * share / (render_ms * 1e-3) / 1e9
        roof = {
            "bound": "valu_issue",
            "kernel": "pinhole_kernel<1, %s, false> (Projector render loop, one ray per lane)" % ("false" if n_sph <= 1024 else "true"),
            "achieved": achieved,
            "peak": PEAK_VALU_GINST,
            "unit": "Ginst/s",
            "frac": None if achieved is None else achieved / PEAK_VALU_GINST,
            "traffic": None if traffic is None else traffic * share / launches,
            "kernel_ms": kernel_ms,
            "launches_per_step": launches,
            "definition": "EXECUTED work: wave-level VALU instructions of this rank's share of the frame (SQ_INSTS_VALU of the committed "
                          "rocprofv3 pass of this kernel and workload, profiles/pmc_constants.json) / its kernel time (HIP events on the "
                          "launch stream), over the VALU issue peak of the chip: 1,024 SIMDs x one wave instruction per 4 cycles at 2.4 GHz. "
                          "The kernel no longer scans the sphere table: primary rays test a per-wave cull list, continuing rays walk a "
                          "uniform grid, so FLOP per sphere test is not the executed work any more (algorithmic_equiv keeps that figure).",
            "frac_at_sustained_clock": None if not clk or achieved is None else achieved / (PEAK_VALU_GINST * clk / 2.4),
            "sustained_clock_ghz": clk,
            "valu_issue_util": pmc.get("valu_issue_util"),
            "valu_lane_util": pmc.get("valu_lane_util"),
            "pmc_source": pmc.get("source"),
            "algorithmic_equiv": {"flop_per_test": FLOP_PER_TEST, "achieved": algorithmic, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                  "frac": algorithmic / PEAK_FP32_TFLOPS,
                                  "note": "SURVEY.md §8d accounting: 17 FLOP for every (segment, sphere) pair the reference's linear scan would "
                                          "test, over the FP32 vector peak; far above 1 because almost none of those tests is executed"},
            "hbm": None if traffic is None else {"achieved": traffic * share / (render_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                  "frac": traffic * share / (render_ms * 1e-3) / 1e9 / 8000.0},
        }
        full = "C2" if name == "c2" else "C4"
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
                "workload": f"{full}: {n_sph} spheres (seed 0xC0FFEE), Projector {w}x{h} cells, {spp} spp, max_bounces {cfg['max_bounces']}; "
                            f"one frame per step" + ("" if world == 1 else f", cells dealt round-robin over {world} GPUs, one all-gather per frame"),
                "segments_per_step": segs_all,
                "sharding": "single GPU" if world == 1 else "cell = rank + m * world (sharding.cell_shard); framebuffer shards all-gathered over RCCL inside the timed region",
                "image_check": image_check,
                # "1/8 GPU" read both ways: value is per whole MI355X (one of a node's eight) at N=1; one XCD's share is value/8/N
                "mrays_per_s_per_gpu": value / world,
                "mrays_per_s_per_xcd": value / world / 8,
                "pct_of_fp32_valu_peak": None if achieved is None else 100.0 * achieved / PEAK_VALU_GINST,
            },
            "roofline": roof,
        }
        if world == 1 and name == "c2" and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n_sph)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
