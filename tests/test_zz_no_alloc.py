"""rt_hip_render_dev_noalloc (include/rt_hip.h): the render entry point that allocates nothing must produce exactly what
rt_hip_render_dev produces — called through the C ABI on device buffers, two calls in a row, the second with a smaller input
(the fixed workspace keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import abi, synth
from raytracing_amd._lib import RtHipError, lib
from raytracing_amd.engine import SimulationSettingsConfig, parallel_grid

from helpers import random_scene

pytestmark = pytest.mark.gpu


def _render(scene, which, gen, spp, st, seed, begin, count, step):
    """One call through the C ABI into freshly poisoned device buffers; returns host copies of every output."""
    import torch

    dev = torch.device("cuda", 0)
    fb = torch.full((count, 3), float("nan"), dtype=torch.float32, device=dev)
    fid = torch.full((count,), -7, dtype=torch.int32, device=dev)
    ft = torch.full((count,), float("nan"), dtype=torch.float32, device=dev)
    seg = torch.full((count,), 12345, dtype=torch.int32, device=dev)
    tot = torch.full((1,), -1, dtype=torch.int64, device=dev)
    fn = getattr(lib(), which)
    rc = fn(scene.handle, C.byref(gen), spp, st.as_c(), seed, begin, count, step, fb.data_ptr(), fid.data_ptr(), ft.data_ptr(),
            seg.data_ptr(), tot.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RtHipError(rc, lib().rt_hip_last_error().decode())
    torch.cuda.synchronize()
    return fb.cpu().numpy(), fid.cpu().numpy(), ft.cpu().numpy(), seg.cpu().numpy().astype(np.uint32), int(tot.item())


def _same(a, b):
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    assert a[4] == b[4]


def _buffers(count):
    import torch

    dev = torch.device("cuda", 0)
    return (torch.full((count, 3), float("nan"), dtype=torch.float32, device=dev), torch.full((count,), -7, dtype=torch.int32, device=dev),
            torch.full((count,), float("nan"), dtype=torch.float32, device=dev), torch.full((count,), 12345, dtype=torch.int32, device=dev),
            torch.full((1,), -1, dtype=torch.int64, device=dev))


def _call(scene, which, gen, spp, st, seed, begin, count, step, bufs):
    import torch

    fb, fid, ft, seg, tot = bufs
    rc = getattr(lib(), which)(scene.handle, C.byref(gen), spp, st.as_c(), seed, begin, count, step, fb.data_ptr(), fid.data_ptr(), ft.data_ptr(),
                               seg.data_ptr(), tot.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RtHipError(rc, lib().rt_hip_last_error().decode())


def _host(bufs):
    fb, fid, ft, seg, tot = bufs
    return fb.cpu().numpy(), fid.cpu().numpy(), ft.cpu().numpy(), seg.cpu().numpy().astype(np.uint32), int(tot.item())


def test_noalloc_entry_point_matches_render_dev(engine):
    import torch

    cfg = synth.config_c2(160, 90)
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    n = 160 * 90
    # first call: the whole grid, 5 spp (ragged sample block); second call: a smaller, strided range, other spp and seed
    calls = [(cfg["gen"], 5, cfg["seed"], 0, n, 1), (cfg["gen"], 9, 77, 3, 1000, 7), (cfg["gen"], 1, 5, n - 1, 1, 1)]
    # A FRESH context: nothing an earlier test made rt_hip_render_dev allocate (rings, accumulators, tables) is there to fall
    # back on, and the output buffers exist before the free-memory reading.  The no-alloc calls come first.
    lib().rt_hip_shutdown()
    engine.init(0)
    sc = engine.Scene(cfg["objects"])
    bufs = [_buffers(c[4]) for c in calls]
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for c, b in zip(calls, bufs):
        _call(sc, "rt_hip_render_dev_noalloc", *c[:2], st, *c[2:], b)
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    assert free_after >= free_before - (1 << 20), f"the no-alloc entry point took {free_before - free_after} bytes of device memory"
    got = [_host(b) for b in bufs]
    want = [_render(sc, "rt_hip_render_dev", *c[:2], st, *c[2:]) for c in calls]
    for g, w in zip(got, want):
        _same(g, w)
    assert got[0][4] > 5 * n  # continuing rays took part
    # the same again in the opposite order: results do not depend on what the workspace held before
    for c, w in reversed(list(zip(calls, want))):
        _same(_render(sc, "rt_hip_render_dev_noalloc", *c[:2], st, *c[2:]), w)


def test_noalloc_launches_fit_the_fixed_rings_whatever_the_occupancy(engine, monkeypatch):
    """The fixed workspace's rings hold 512 blocks; a build or device with more resident blocks per CU launches 512 (same result)."""
    cfg = synth.config_c2(256, 144)
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    sc = engine.Scene(cfg["objects"])
    args = (cfg["gen"], 8, st, cfg["seed"], 0, 256 * 144, 1)
    want = _render(sc, "rt_hip_render_dev", *args)
    monkeypatch.setenv("RT_HIP_BLOCKS_PER_CU", "4")  # 1,024 blocks on 256 CUs; read by the library at every launch
    _same(_render(sc, "rt_hip_render_dev_noalloc", *args), want)
    _same(_render(sc, "rt_hip_render_dev", *args), want)


def test_noalloc_other_kernels(engine):
    """General kernel (ParallelGrid), streamed sphere table (> 1,024 spheres) and the per-lane object path on the fixed workspace."""
    from raytracing_amd.engine import shape_object

    st = SimulationSettingsConfig(60.0, 6)
    objs = list(random_scene(1500, 2, seed=3, extent=(60.0, 30.0, 60.0)))
    objs.insert(7, shape_object(abi.RT_SHAPE_BOX, (8.0, 8.0, 8.0), (10, 5, -80), abi.RT_MAT_GLASS, 1.5, rotation_y_deg=30.0))
    sc = engine.Scene(np.array(objs, dtype=abi.OBJECT_DTYPE))
    from raytracing_amd.engine import projector

    gens = [projector((0.0, 0.0, -150.0), (-70.0, -40.0, -70.0), (140.0, 0.0, 0.0), (0.0, 80.0, 0.0), 64, 40),
            parallel_grid((-60, -30, -100), (120, 0, 0), (0, 60, 0), 50, 30, (0.02, 0.01, 1.0))]
    for gen in gens:
        n = gen.count_u * gen.count_v
        for spp, begin, count, step in [(3, 0, n, 1), (2, 5, 300, 3)]:
            _same(_render(sc, "rt_hip_render_dev_noalloc", gen, spp, st, 9, begin, count, step),
                  _render(sc, "rt_hip_render_dev", gen, spp, st, 9, begin, count, step))


def test_noalloc_limits_are_refused_with_a_clear_error(engine):
    cfg = synth.config_c2(1024, 600)  # 614,400 cells > 524,288
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    sc = engine.Scene(cfg["objects"])
    with pytest.raises(RtHipError) as e:
        _render(sc, "rt_hip_render_dev_noalloc", cfg["gen"], 1, st, 1, 0, 1024 * 600, 1)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "524288" in str(e.value)
    # the same frame in two calls is fine and equals the allocating entry point
    half = 1024 * 300
    a = _render(sc, "rt_hip_render_dev_noalloc", cfg["gen"], 1, st, 1, 0, half, 1)
    b = _render(sc, "rt_hip_render_dev_noalloc", cfg["gen"], 1, st, 1, half, half, 1)
    w = _render(sc, "rt_hip_render_dev", cfg["gen"], 1, st, 1, 0, 2 * half, 1)
    assert np.array_equal(np.concatenate([a[0], b[0]]), w[0]) and np.array_equal(np.concatenate([a[3], b[3]]), w[3])
    assert a[4] + b[4] == w[4]
    big = random_scene(17000, 0, seed=1, extent=(200.0, 100.0, 200.0))
    with pytest.raises(RtHipError) as e:
        _render(engine.Scene(big), "rt_hip_render_dev_noalloc", cfg["gen"], 1, st, 1, 0, 64, 1)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "16384" in str(e.value)
