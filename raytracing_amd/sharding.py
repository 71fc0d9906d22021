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


def frame_checksum(fb):
    """Order-independent, position-sensitive 64-bit checksum of an [n, 3] float32 image (torch tensor, any device):
    sum over elements of (bit pattern * (2654435761 * index + 1)) mod 2^64.  Integer arithmetic, so the value does not
    depend on how the image was computed, sharded or summed — equal images give equal checksums on any rank count."""
    import torch

    bits = fb.contiguous().view(torch.int32).reshape(-1).to(torch.int64) & 0xFFFFFFFF
    idx = torch.arange(bits.numel(), dtype=torch.int64, device=bits.device)
    return int((bits * (idx * 2654435761 + 1)).sum().item()) & 0xFFFFFFFFFFFFFFFF  # int64 products / sums wrap mod 2^64


class ShardedFrame:
    """One frame of the generator-driven loop tile-sharded over `world` ranks (BASELINE.json configs[3]).

    `render_cells(cell_begin, cell_count, cell_step, fb_rows, total_slot)` enqueues the render of those generator cells into
    the [cell_count, 3] float32 tensor `fb_rows` and their segment count into the 1-element int64 tensor `total_slot`
    (on the GPU: engine.Scene.render_dev on the current stream; the CPU tests pass an oracle-backed stand-in).  `step()`
    renders this rank's cells rank, rank + world, ... in launches of at most `max_cells_per_launch` cells and then does the
    frame's ONE exchange, an all-gather of the equal-sized shards; nothing in it synchronises with the host."""

    def __init__(self, render_cells, n_cells, rank, world, dist, device, max_cells_per_launch=1 << 19):
        import torch

        self.render_cells, self.n_cells, self.rank, self.world, self.dist = render_cells, n_cells, rank, world, dist
        self.begin, self.count, self.step_cells = cell_shard(rank, world, n_cells)
        self.cap = shard_capacity(world, n_cells)
        self.launches = [(done, min(max_cells_per_launch, self.count - done)) for done in range(0, self.count, max_cells_per_launch)]
        self.local = torch.zeros((self.cap, 3), dtype=torch.float32, device=device)  # rows past `count` stay zero (padding)
        self.totals = torch.zeros(max(1, len(self.launches)), dtype=torch.int64, device=device)
        self.gathered = torch.empty((world * self.cap, 3), dtype=torch.float32, device=device) if world > 1 else None

    def render(self):
        for k, (done, n) in enumerate(self.launches):
            self.render_cells(self.begin + done * self.step_cells, n, self.step_cells, self.local[done:done + n], self.totals[k:k + 1])

    def gather(self):
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.gathered, self.local)

    def step(self):
        self.render()
        self.gather()

    def image(self):
        """[n_cells, 3] in cell order (a view of the gathered shards)."""
        if self.world == 1:
            return self.local[: self.n_cells]
        return interleave_shards(self.gathered.reshape(self.world, self.cap, 3), self.n_cells)

    def segments(self):
        """This rank's ray-segment count of the last frame (synchronises)."""
        return int(self.totals.sum().item())


def render_sharded(scene, gen, spp, setting, seed, rank, world, dist, device, max_cells_per_launch=1 << 19):
    """One frame over `world` GPUs: this rank renders cells rank, rank+world, ... (in launches of at most
    `max_cells_per_launch` cells), then one all-gather assembles the image on every rank.
    Returns (fb [n_cells, 3] float32 torch tensor in cell order, this rank's ray-segment count)."""
    import torch

    n_cells = gen.count_u * gen.count_v
    stream = torch.cuda.current_stream(device).cuda_stream

    def render_cells(begin, count, step, fb_rows, total_slot):
        scene.render_dev(gen, spp, setting, seed, begin, count, step, fb_rows.data_ptr(), 0, 0, 0, total_slot.data_ptr(), stream)

    frame = ShardedFrame(render_cells, n_cells, rank, world, dist, device, max_cells_per_launch)
    frame.render()
    if world > 1 and hasattr(dist, "all_gather_into_tensor"):
        frame.gather()
        return frame.image(), frame.segments()
    return gather_image(frame.local, rank, world, n_cells, dist) if world > 1 else frame.local[:n_cells], frame.segments()


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
