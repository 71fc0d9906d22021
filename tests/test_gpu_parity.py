"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from oracle import binding as ob
from raytracing_amd import abi, synth
from raytracing_amd.engine import SimulationSettingsConfig, projector, parallel_grid, sphere, plane, make_rays

from helpers import assert_records_equal, random_rays, random_scene

pytestmark = pytest.mark.gpu


def _simulate_both(engine, objs, rays, inf_dist, max_b, seed=7, base=0):
    sc = engine.Scene(objs)
    st = SimulationSettingsConfig(inf_dist, max_b)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=seed, ray_id_base=base)
    osc = ob.OracleScene(objs)
    opts, onpts, orecs, _ = osc.simulate(rays, inf_dist, max_b, seed=seed, ray_id_base=base)
    return (pts, npts, recs), (opts, onpts, orecs)


def test_tmp_py_known_answer(engine):
    """python_files/tmp.py:4-9 — the only numeric data the reference holds."""
    objs = np.array([sphere((0, 0, 0), 5.0, abi.RT_MAT_GLASS, 1.5)], dtype=abi.OBJECT_DTYPE)
    rays = make_rays([(-10.0, 2.0, 0.0)], [(1.0, 0.0, 0.0)])
    sc = engine.Scene(objs, rays)
    paths = sc.simulate_rays(SimulationSettingsConfig(20.0, 3))
    want = np.array([(-10.000, 2.000), (-4.583, 2.000), (4.959, 0.640), (24.163, -4.948)])
    assert paths[0].shape == (4, 3)
    assert np.abs(paths[0][:, :2] - want).max() <= 5e-4
    assert np.all(paths[0][:, 2] == 0)


@pytest.mark.parametrize("n_spheres,n_planes,n_rays,max_b", [(1, 0, 64, 4), (37, 0, 1000, 8), (300, 3, 5000, 8), (1000, 2, 20000, 10),
                                                            (2500, 1, 3000, 6), (5, 4, 777, 12)])
def test_simulate_records_bit_exact(engine, n_spheres, n_planes, n_rays, max_b):
    objs = random_scene(n_spheres, n_planes, seed=n_spheres + 11)
    rays = random_rays(n_rays, seed=n_rays)
    (pts, npts, recs), (opts, onpts, orecs) = _simulate_both(engine, objs, rays, 50.0, max_b)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "records")
    assert np.array_equal(pts, opts)
    assert (recs["obj_id"] >= 0).sum() > 0  # the case really exercises hits


def test_scan_modes_agree(engine):
    """Conservative filter (mode 0) and exact-discriminant filter (mode 1) give identical results."""
    objs = random_scene(700, 2, seed=5)
    rays = random_rays(20000, seed=9)
    sc = engine.Scene(objs)
    st = SimulationSettingsConfig(40.0, 8)
    try:
        engine.set_scan_mode(1)
        a = sc.simulate_rays_raw(st, rays=rays, seed=3)
    finally:
        engine.set_scan_mode(0)
    b = sc.simulate_rays_raw(st, rays=rays, seed=3)
    assert np.array_equal(a[1], b[1])
    assert_records_equal(a[2], b[2], "mode1 vs mode0")


def test_render_parity_small_c2(engine):
    """Reduced C2 (same scene, 96x54 cells, 4 spp): framebuffer, first-hit ids/t, segment counts all bit-exact."""
    cfg = synth.config_c2(96, 54)
    sc = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    fb, fid, ft, seg, tot = sc.render(cfg["gen"], 4, st, seed=cfg["seed"])
    osc = ob.OracleScene(cfg["objects"])
    ofb, ofid, oft, oseg, otot = osc.render(cfg["gen"], 4, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], threads=8)
    assert np.array_equal(fid, ofid)
    assert np.array_equal(ft, oft)
    assert np.array_equal(seg, oseg)
    assert tot == otot == int(seg.sum())
    assert np.array_equal(fb, ofb)
    assert (fid >= 0).mean() > 0.05


def test_render_strided_cells_match_full(engine):
    """Sharding invariant (multi-GPU): cells begin+m*step computed alone equal the same cells of the full image."""
    cfg = synth.config_c2(64, 36)
    sc = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    full = sc.render(cfg["gen"], 3, st, seed=1)
    n = 64 * 36
    for rank, world in [(0, 2), (1, 2), (2, 4)]:
        cnt = (n - rank + world - 1) // world
        part = sc.render(cfg["gen"], 3, st, seed=1, cell_begin=rank, cell_count=cnt, cell_step=world)
        for a, b in zip(full[:4], part[:4]):
            assert np.array_equal(a[rank::world], b)


def test_parallel_grid_generator(engine):
    objs = random_scene(50, 1, seed=3)
    gen = parallel_grid((-20, -10, -30), (40, 0, 0), (0, 20, 0), 40, 20, (0.1, -0.05, 1.0))
    sc = engine.Scene(objs)
    st = SimulationSettingsConfig(30.0, 6)
    got = sc.render(gen, 2, st, seed=4)
    want = ob.OracleScene(objs).render(gen, 2, 30.0, 6, seed=4)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)
    assert got[4] == want[4]


def test_edge_cases(engine):
    st = SimulationSettingsConfig(10.0, 5)
    # empty scene: every ray is origin + far point
    sc = engine.Scene(np.zeros(0, abi.OBJECT_DTYPE))
    rays = random_rays(130, seed=1)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays)
    opts, onpts, orecs, _ = ob.OracleScene(np.zeros(0, abi.OBJECT_DTYPE)).simulate(rays, 10.0, 5)
    assert np.array_equal(npts, onpts) and np.all(npts == 2)
    assert np.array_equal(pts, opts)
    # no rays
    p2, n2, _ = sc.simulate_rays_raw(st, rays=np.zeros(0, abi.RAY_DTYPE))
    assert p2.shape == (0, 6, 3) and n2.shape == (0,)
    # max_bounces = 0: path is just the origin (scene.rs:45-49)
    p3, n3, _ = sc.simulate_rays_raw(SimulationSettingsConfig(10.0, 0), rays=rays, records=False)
    assert np.all(n3 == 1) and np.array_equal(p3[:, 0], rays["origin"])
    # ray starting inside a sphere, tangent ray, ray pointing away
    objs = np.array([sphere((0, 0, 0), 2.0, abi.RT_MAT_GLASS, 1.5), sphere((10, 0, 0), 1.0, abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE)
    rays = make_rays([(0.5, 0.1, 0), (-5, 2.0, 0), (-5, 0, 0), (0, 0, 0)], [(1, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0)])
    sc = engine.Scene(objs)
    got = sc.simulate_rays_raw(st, rays=rays)
    want = ob.OracleScene(objs).simulate(rays, 10.0, 5)
    assert np.array_equal(got[1], want[1])
    assert_records_equal(got[2], want[2], "edge")
    # identical overlapping spheres: tie on t goes to the lowest index (scene.rs:56 strict <)
    objs = np.array([sphere((0, 0, 5), 1.0, abi.RT_MAT_MIRROR), sphere((0, 0, 5), 1.0, abi.RT_MAT_GLASS, 1.5)], dtype=abi.OBJECT_DTYPE)
    rays = make_rays([(0, 0, 0)], [(0, 0, 1)])
    got = engine.Scene(objs).simulate_rays_raw(st, rays=rays)
    assert got[2]["obj_id"][0, 0] == 0


def test_unsupported_inputs_fail_loudly(engine):
    from raytracing_amd._lib import RtHipError

    with pytest.raises(RtHipError) as e:
        engine.Scene(np.array([sphere((0, 0, 0), 1.0, abi.RT_MAT_MIRROR, rotation_y_deg=30.0)], dtype=abi.OBJECT_DTYPE))
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    bad = sphere((0, 0, 0), 1.0, abi.RT_MAT_MIRROR)
    bad["shape"] = 7
    with pytest.raises(RtHipError):
        engine.Scene(np.array([bad], dtype=abi.OBJECT_DTYPE))


def test_half_mirror_stream_is_counter_based(engine):
    """Same (seed, ray id, bounce) -> same draw, whatever the batch composition (scene.rs:88 replacement)."""
    objs = np.array([sphere((0, 0, 5 + 3 * k), 1.0, abi.RT_MAT_HALF_MIRROR, 0.5) for k in range(6)], dtype=abi.OBJECT_DTYPE)
    rays = make_rays([(0.01 * k, 0.0, 0.0) for k in range(256)], [(0, 0, 1)] * 256)
    sc = engine.Scene(objs)
    st = SimulationSettingsConfig(10.0, 12)
    full = sc.simulate_rays_raw(st, rays=rays, seed=42)
    part = sc.simulate_rays_raw(st, rays=rays[100:164], seed=42, ray_id_base=100)
    assert_records_equal(full[2][100:164], part[2], "sub-batch")
    want = ob.OracleScene(objs).simulate(rays, 10.0, 12, seed=42)
    assert_records_equal(full[2], want[2], "halfmirror")
    other = sc.simulate_rays_raw(st, rays=rays, seed=43)
    assert not np.array_equal(other[1], full[1]) or not np.array_equal(other[2]["obj_id"], full[2]["obj_id"])
