#!/usr/bin/env python3
"""bench.py — Mrays/s of the ray-path hot loop on the C2 workload (BASELINE.json configs[1]).

A step = one pass of the hot path over one frame of synthetic input: 1,000 spheres, Projector 1920x1080 cells,
64 paths per cell, max_bounces 8.  A "ray" = one bounce-loop iteration (scene.rs:49), counted by
the kernel.  N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders the same frame for
its own sample block (weak scaling) and the framebuffers are summed to rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_TEST = 17  # algorithmic reject-path cost of one ray-sphere test (sphere.rs:13-17)
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (packed FMA)


def cpu_baseline(cfg, n_spheres):
    """Oracle (C++ restatement, kind 'port') on the host cores, on a bounded sample of the same workload."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import engine, synth
    from raytracing_amd.engine import SimulationSettingsConfig

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RT_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, reduce) even with one rank, to rehearse it on a 1-GPU box
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
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    def reduce_fb(t):
        """Sum the ranks' framebuffers to rank 0.  Over RCCL the reduce is asynchronous (RCCL's own stream, ordered after the
        render that filled `t`), so it overlaps the next frame's render into the other buffer; returns the work handle."""
        if rehearsal:
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            return None
        return dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, async_op=True)

    def allreduce_scalar(t, op):
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)
    engine.init(dev.index)

    cfg = synth.config_c2(args.width, args.height)
    cfg["spp"] = args.spp
    scene = engine.Scene(cfg["objects"])
    n_obj, n_sph, n_pl = scene.counts()
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    n_cells = args.width * args.height
    # inputs (scene, generator constants) and outputs resident in HBM before the timed region
    fbs = [torch.empty((n_cells, 3), dtype=torch.float32, device=dev) for _ in range(2 if use_dist else 1)]  # double-buffered for the reduce
    pending = [None, None]
    first_id = torch.empty(n_cells, dtype=torch.int32, device=dev)
    first_t = torch.empty(n_cells, dtype=torch.float32, device=dev)
    seg = torch.empty(n_cells, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    seed = (cfg["seed"] + rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF  # rank = sample block

    frame = [0]

    def step(events=None):
        b = frame[0] % len(fbs)
        frame[0] += 1
        if pending[b] is not None:  # the reduce that last read this buffer must be done before it is overwritten
            pending[b].wait()
            pending[b] = None
        fb = fbs[b]
        if events:
            events[0].record()
        stream = torch.cuda.current_stream().cuda_stream
        scene.render_dev(cfg["gen"], cfg["spp"], st, seed, 0, n_cells, 1, fb.data_ptr(), first_id.data_ptr(), first_t.data_ptr(),
                         seg.data_ptr(), total.data_ptr(), stream)
        if events:
            events[1].record()
        if use_dist:
            pending[b] = reduce_fb(fb)

    def barrier():
        for b in range(len(pending)):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # bring the GPU to its steady clocks before anything is counted (a cold first frame runs ~8 % slower); untimed, like the W warmup steps
    # (render only, no collective: the number of pre-warm frames is time-based and may differ between ranks)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5:
        scene.render_dev(cfg["gen"], cfg["spp"], st, seed, 0, n_cells, 1, fbs[0].data_ptr(), first_id.data_ptr(), first_t.data_ptr(),
                         seg.data_ptr(), total.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    # kernel duration: HIP events on the launch stream around each render_dev (no collective inside the pair)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    segs_done = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t0
    segs_per_step = int(total.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    s = torch.tensor([segs_per_step], dtype=torch.int64, device=dev)
    if use_dist:
        allreduce_scalar(t, dist.ReduceOp.MAX)
        allreduce_scalar(s, dist.ReduceOp.SUM)
    elapsed_max = float(t.item())
    segs_all = int(s.item())

    if rank == 0:
        value = segs_all * args.steps / elapsed_max / 1e6
        achieved = segs_per_step * FLOP_PER_TEST * n_sph / (kernel_ms * 1e-3) / 1e12
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec (primary+bounce) at 1080p 64spp, 1/8 GPU; % of FP32 VALU peak",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: {n_sph} spheres (seed 0xC0FFEE), Projector {args.width}x{args.height} cells, {cfg['spp']} spp, "
                            f"max_bounces {cfg['max_bounces']}; one frame per GPU per step",
                "segments_per_step_per_gpu": segs_per_step,
                "sharding": "one sample block (frame) per rank; framebuffer reduce to rank 0 over RCCL, overlapped with the next frame" if world > 1 else "single GPU",
                # "1/8 GPU" read both ways: value is per whole MI355X (one of a node's eight) at N=1; one XCD's share is value/8/N
                "mrays_per_s_per_gpu": value / world,
                "mrays_per_s_per_xcd": value / world / 8,
                "pct_of_fp32_valu_peak": 100.0 * achieved / PEAK_FP32_TFLOPS,
            },
            "roofline": {
                "bound": "valu_fp32",
                "kernel": "pinhole_kernel (Projector render loop; general trace_kernel for other generators)",
                "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": traffic,
                "kernel_ms": kernel_ms,
                "flop_per_segment": FLOP_PER_TEST * n_sph,
                "note": "algorithmic 17 FLOP per ray-sphere test x spheres x segments / kernel time (HIP events on the launch stream); "
                        "the path is FP32-VALU-bound (no MFMA by design), HBM shown for completeness",
                "hbm": None if traffic is None else {"achieved": traffic / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                      "frac": traffic / (kernel_ms * 1e-3) / 1e9 / 8000.0},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, n_sph)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
