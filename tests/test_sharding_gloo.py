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
