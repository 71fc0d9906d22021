"""Multi-GPU sharding of the generator-driven loop: one process per GPU, scene replicated.

Cells of one frame are dealt round-robin over ranks (`cell = rank + m * world`), which balances load without
any knowledge of the scene; the only exchange is one all-gather of equal-sized shards per frame.  The
fixed-point accumulation makes every cell's value independent of who computed it, so the assembled image is
bit-identical to the single-GPU image.
"""
import numpy as np


def cell_shard(rank, world, n_cells):
    """(cell_begin, cell_count, cell_step) of `rank`: cells rank, rank+world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    count = (n_cells - rank + world - 1) // world if n_cells > rank else 0
    return rank, count, world


def shard_capacity(world, n_cells):
    """Rows every rank allocates so the all-gather has equal-sized inputs."""
    return (n_cells + world - 1) // world


def interleave_shards(shards, n_cells):
    """[world, capacity, ...] gathered shards -> [n_cells, ...] in cell order (torch tensor or numpy array)."""
    world, cap = shards.shape[0], shards.shape[1]
    if hasattr(shards, "transpose") and not isinstance(shards, np.ndarray):  # torch
        full = shards.transpose(0, 1).reshape((cap * world,) + tuple(shards.shape[2:]))
    else:
        full = np.swapaxes(shards, 0, 1).reshape((cap * world,) + tuple(shards.shape[2:]))
    return full[:n_cells]


def gather_image(local, rank, world, n_cells, dist, dst=None):
    """All-gather equal-sized shards (torch tensors, padded to shard_capacity) and return the full image.

    `dist` is torch.distributed (backend nccl = RCCL on the GPU box, gloo in the CPU tests)."""
    import torch

    cap = shard_capacity(world, n_cells)
    if local.shape[0] != cap:
        pad = torch.zeros((cap - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return interleave_shards(out.reshape((world, cap) + tuple(local.shape[1:])), n_cells)


def render_sharded(scene, gen, spp, setting, seed, rank, world, dist, device, max_cells_per_launch=1 << 19):
    """One frame over `world` GPUs: this rank renders cells rank, rank+world, ... (in launches of at most
    `max_cells_per_launch` cells), then one all-gather assembles the image on every rank.
    Returns (fb [n_cells, 3] float32 torch tensor in cell order, this rank's ray-segment count)."""
    import torch

    n_cells = gen.count_u * gen.count_v
    begin, count, step = cell_shard(rank, world, n_cells)
    cap = shard_capacity(world, n_cells)
    fb = torch.zeros((cap, 3), dtype=torch.float32, device=device)
    total = torch.zeros(1, dtype=torch.int64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    segments, done = 0, 0
    while done < count:
        n = min(max_cells_per_launch, count - done)
        scene.render_dev(gen, spp, setting, seed, begin + done * step, n, step, fb[done:].data_ptr(), 0, 0, 0, total.data_ptr(), stream)
        segments += int(total.item())  # also orders the launches: the library's scratch is shared between calls
        done += n
    if world > 1:
        return gather_image(fb, rank, world, n_cells, dist), segments
    return fb[:n_cells], segments


def main(argv=None):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m raytracing_amd.sharding
           --workload c4 [--width 3840 --height 2160 --spp 1024] [--out frame.pfm]
    BASELINE.json configs[3]: the 10,000-sphere frame tile-sharded over N GPUs, framebuffer gathered over RCCL."""
    import argparse
    import os
    import time

    import torch
    import torch.distributed as dist

    from . import engine, synth
    from .engine import SimulationSettingsConfig
    from .image import write_pfm

    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["c2", "c4"], default="c4")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    engine.init(local)
    make = synth.config_c4 if a.workload == "c4" else synth.config_c2
    w, h = (a.width or (3840 if a.workload == "c4" else 1920)), (a.height or (2160 if a.workload == "c4" else 1080))
    cfg = make(w, h)
    spp = a.spp or (1024 if a.workload == "c4" else 64)
    scene = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fb, segs = render_sharded(scene, cfg["gen"], spp, st, cfg["seed"], rank, world, dist, dev)
    seg_t = torch.tensor([segs], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(seg_t)
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        total = int(seg_t.item())
        print(f"{a.workload} {w}x{h} x {spp} spp on {world} GPU(s): {total} segments in {dt:.3f} s = {total / dt / 1e6:.0f} Mrays/s; "
              f"fb checksum {fb.sum(0, dtype=torch.float64).tolist()}", flush=True)
        if a.out:
            write_pfm(a.out, fb.cpu().numpy(), w, h)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
