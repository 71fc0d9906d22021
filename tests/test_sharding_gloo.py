"""N > 1 path on CPU: two gloo ranks shard the cells round-robin, gather, and must reproduce the 1-rank image
bit for bit (the oracle stands in for the kernel here; the GPU version of this check is in test_gpu_parity.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from raytracing_amd import synth

    cfg = synth.config_c2(w, h)
    n = w * h
    begin, count, step = sharding.cell_shard(rank, world, n)
    fb, fid, ft, seg, tot = ob.OracleScene(cfg["objects"]).render(cfg["gen"], 3, cfg["infinity_distance"], cfg["max_bounces"], seed=11,
                                                                cell_begin=begin, cell_count=count, cell_step=step)
    full_fb = sharding.gather_image(torch.from_numpy(fb), rank, world, n, dist)
    full_id = sharding.gather_image(torch.from_numpy(fid), rank, world, n, dist)
    t = torch.tensor([tot], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        np.savez(out_path, fb=full_fb.numpy(), fid=full_id.numpy(), total=t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(16, 9), (15, 7)])  # even and odd cell counts (ragged last shard)
def test_two_rank_gather_equals_single_rank(tmp_path, w, h):
    from oracle import binding as ob
    from raytracing_amd import synth

    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    g = np.load(out)
    cfg = synth.config_c2(w, h)
    fb, fid, ft, seg, tot = ob.OracleScene(cfg["objects"]).render(cfg["gen"], 3, cfg["infinity_distance"], cfg["max_bounces"], seed=11)
    assert np.array_equal(g["fb"], fb) and np.array_equal(g["fid"], fid) and int(g["total"][0]) == tot


def test_cell_shard_partitions_exactly():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, c, s = sharding.cell_shard(r, world, n)
                seen += [b + m * s for m in range(c)]
                assert c <= sharding.shard_capacity(world, n)
            assert sorted(seen) == list(range(n))
    shards = np.arange(2 * 4 * 3).reshape(2, 4, 3)
    full = sharding.interleave_shards(shards, 7)
    assert np.array_equal(full[0], shards[0, 0]) and np.array_equal(full[1], shards[1, 0]) and np.array_equal(full[6], shards[0, 3])


def _frame_worker(rank, world, port, name, w, h, spp, out_path):
    """bench.py's N > 1 step (sharding.ShardedFrame: this rank's cells in launches, then ONE all_gather_into_tensor) over gloo,
    with the oracle standing in for rt_hip_render_dev."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from raytracing_amd import synth

    cfg = (synth.config_c2 if name == "c2" else synth.config_c4)(w, h)
    osc = ob.OracleScene(cfg["objects"])

    def render_cells(begin, count, step, fb_rows, total_slot):
        fb, _, _, _, tot = osc.render(cfg["gen"], spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], cell_begin=begin,
                                      cell_count=count, cell_step=step)
        fb_rows.copy_(torch.from_numpy(fb))
        total_slot.fill_(tot)

    frame = sharding.ShardedFrame(render_cells, w * h, rank, world, dist, torch.device("cpu"), max_cells_per_launch=50)
    assert len(frame.launches) >= 2  # several launches per rank, like C4's 2^19-cell launches
    frame.step()
    frame.step()  # a second frame into the same buffers
    t = torch.tensor([frame.segments()], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        np.savez(out_path, fb=frame.image().numpy(), total=t.numpy(), checksum=np.array([sharding.frame_checksum(frame.image())], np.uint64))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,w,h,spp,world", [("c2", 16, 9, 3, 2), ("c2", 15, 7, 3, 2), ("c4", 24, 16, 2, 3)])
def test_sharded_frame_step_matches_stored_checksum(tmp_path, name, w, h, spp, world):
    import json

    from oracle import binding as ob
    from raytracing_amd import synth

    out = str(tmp_path / "frame.npz")
    mp.spawn(_frame_worker, args=(world, _free_port(), name, w, h, spp, out), nprocs=world, join=True)
    g = np.load(out)
    cfg = (synth.config_c2 if name == "c2" else synth.config_c4)(w, h)
    fb, _, _, _, tot = ob.OracleScene(cfg["objects"]).render(cfg["gen"], spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"])
    assert np.array_equal(g["fb"], fb) and int(g["total"][0]) == tot
    stored = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "frame_checksums.json")))
    key = f"{name}:{w}x{h}:spp{spp}:seed{cfg['seed']}:depth{cfg['max_bounces']}"
    assert int(stored[key], 16) == int(g["checksum"][0]) == sharding.frame_checksum(torch.from_numpy(fb))
    # position-sensitive: swapping two pixels changes it
    fb2 = fb.copy()
    fb2[[0, 1]] = fb2[[1, 0]]
    if not np.array_equal(fb2, fb):
        assert sharding.frame_checksum(torch.from_numpy(fb2)) != int(g["checksum"][0])
