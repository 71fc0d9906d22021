"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

from oracle import binding as ob
from raytracing_amd import abi, synth
from raytracing_amd.engine import SimulationSettingsConfig, projector, parallel_grid, sphere, plane, make_rays

from helpers import TEMPLATE_LIKE_TOML, assert_records_equal, random_rays, random_scene, same

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


def _sphere_union_nodes(n, param):
    """Post-order nodes of ((s0 U s1) U s2) U ... : n sphere leaves, then n - 1 unions; the root is the last node."""
    nodes = np.zeros(2 * n - 1, abi.NODE_DTYPE)
    for k in range(n):
        nodes[k]["shape"], nodes[k]["shape_param"] = abi.RT_SHAPE_SPHERE, param(k)
    for k in range(n - 1):
        nodes[n + k]["shape"], nodes[n + k]["a"], nodes[n + k]["b"] = abi.RT_SHAPE_UNION, (0 if k == 0 else n + k - 1), k + 1
    return nodes


def test_union_with_sixteen_hits_per_list(engine):
    """csg.rs:9-75 has no limit on the hits a list holds; the GPU path's per-lane lists hold 16 (8 until round 3): a union of eight
    concentric glass shells, rays through all of them (every list full), inside them and past them, against the oracle."""
    from raytracing_amd.engine import tree_object

    nodes = _sphere_union_nodes(8, lambda k: (1.0 + 0.5 * k, 0, 0, 0))
    objs = np.array([tree_object(len(nodes) - 1, (0.5, -0.25, 0.0), abi.RT_MAT_GLASS, 1.3, rotation_y_deg=20.0), sphere((9, 0, 0), 1.0, abi.RT_MAT_MIRROR)],
                    dtype=abi.OBJECT_DTYPE)
    rays = random_rays(4000, seed=21, extent=(7.0, 7.0, 7.0))
    # 200 of them from outside through (nearly) the common centre: all sixteen surfaces on one ray
    rng = np.random.default_rng(2)
    o = rng.standard_normal((200, 3)).astype(np.float32)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * np.float32(9.0) + np.array([0.5, -0.25, 0.0], np.float32)
    rays[:200] = make_rays(o, (np.array([0.5, -0.25, 0.0], np.float32) - o + 0.01 * rng.standard_normal((200, 3)).astype(np.float32)))
    st = SimulationSettingsConfig(40.0, 6)
    pts, npts, recs = engine.Scene(objs, nodes=nodes).simulate_rays_raw(st, rays=rays, seed=5)
    opts, onpts, orecs, _ = ob.OracleScene(objs, nodes).simulate(rays, 40.0, 6, seed=5)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "union of eight shells")
    assert same(pts, opts)
    assert (recs["obj_id"][:, 0] == 0).sum() > 600 and np.all(recs["obj_id"][:200, 0] == 0)  # the shells were hit


def test_unsupported_inputs_fail_loudly(engine):
    from raytracing_amd._lib import RtHipError

    bad = sphere((0, 0, 0), 1.0, abi.RT_MAT_MIRROR)
    bad["shape"] = 99
    with pytest.raises(RtHipError) as e:
        engine.Scene(np.array([bad], dtype=abi.OBJECT_DTYPE))
    assert e.value.code == abi.RT_ERR_INVALID
    bad["shape"] = abi.RT_SHAPE_TREE  # tree object without a node array
    with pytest.raises(RtHipError):
        engine.Scene(np.array([bad], dtype=abi.OBJECT_DTYPE))
    bad["shape"], bad["material_kind"] = abi.RT_SHAPE_SPHERE, 9
    with pytest.raises(RtHipError):
        engine.Scene(np.array([bad], dtype=abi.OBJECT_DTYPE))
    # a union whose hit list could exceed the per-lane capacity (16 hits) is refused, not truncated: nine nested spheres
    from raytracing_amd.engine import tree_object
    nodes = _sphere_union_nodes(9, lambda k: (1.0 + k, 0, 0, 0))
    with pytest.raises(RtHipError) as e:
        engine.Scene(np.array([tree_object(len(nodes) - 1, (0, 0, 0), abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE), nodes=nodes)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED
    # meshes: index out of range, vertex index out of range, non-finite vertex, mesh shape inside a tree
    from raytracing_amd.engine import mesh_object
    tri = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.uint32))
    mobj = np.array([mesh_object(1, (0, 0, 0), abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE)
    for objs_, meshes_ in [(mobj, [tri]), (mobj, None),
                           (np.array([mesh_object(0, (0, 0, 0), abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE), [(tri[0], np.array([[0, 1, 3]], np.uint32))]),
                           (np.array([mesh_object(0, (0, 0, 0), abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE), [(np.full((3, 3), np.inf, np.float32), tri[1])])]:
        with pytest.raises(RtHipError) as e:
            engine.Scene(objs_, meshes=meshes_)
        assert e.value.code == abi.RT_ERR_INVALID
    nodes = np.zeros(1, abi.NODE_DTYPE)
    nodes[0]["shape"] = abi.RT_SHAPE_MESH
    with pytest.raises(RtHipError):
        engine.Scene(np.array([tree_object(0, (0, 0, 0), abi.RT_MAT_MIRROR)], dtype=abi.OBJECT_DTYPE), nodes=nodes, meshes=[tri])


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


import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name,args,kw", [("c1_paths", (50.0, 10), {}), ("synth200_paths", (40.0, 8), dict(seed=5, ray_id_base=1000))])
def test_gpu_matches_golden_paths(engine, name, args, kw):
    """Committed fixtures (tests/golden/make_golden.py): the C1 scene of the reference's simulation.toml and a seeded scene."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    sc = engine.Scene(g["objects"])
    pts, npts, recs = sc.simulate_rays_raw(SimulationSettingsConfig(*args), rays=g["rays"], **kw)
    assert np.array_equal(npts, g["n_points"]) and np.array_equal(pts, g["points"])
    assert_records_equal(recs, g["records"], name)


def test_gpu_matches_golden_render(engine):
    g = np.load(os.path.join(GOLD, "c2_48x27_4spp.npz"))
    c2 = synth.config_c2(48, 27)
    fb, fid, ft, seg, tot = engine.Scene(c2["objects"]).render(c2["gen"], 4, SimulationSettingsConfig(c2["infinity_distance"], c2["max_bounces"]),
                                                               seed=c2["seed"])
    assert np.array_equal(fb, g["fb"]) and np.array_equal(fid, g["first_id"]) and np.array_equal(ft, g["first_t"])
    assert np.array_equal(seg, g["seg_count"]) and tot == int(g["total"])


def test_c1_toml_end_to_end(engine):
    """cli.rs:12-14 flow on the GPU: load TOML -> Scene -> simulate_rays -> polylines equal the oracle's."""
    from raytracing_amd.config import SimulationConfig

    cfg = SimulationConfig.load_from_path(os.path.join(GOLD, "c1_scene.toml"))
    scene = cfg.scene.into_scene()
    paths = scene.simulate_rays(cfg.simulation_settings)
    g = np.load(os.path.join(GOLD, "c1_paths.npz"))
    assert len(paths) == 3
    for k, p in enumerate(paths):
        assert np.array_equal(p, g["points"][k, : g["n_points"][k]])


def test_full_size_c2_properties(engine):
    """BASELINE.json configs[1] at full size (1920x1080, 64 spp): size-independent checks + sampled oracle comparison."""
    c2 = synth.config_c2()
    sc = engine.Scene(c2["objects"])
    st = SimulationSettingsConfig(c2["infinity_distance"], c2["max_bounces"])
    fb, fid, ft, seg, tot = sc.render(c2["gen"], c2["spp"], st, seed=c2["seed"])
    n = 1920 * 1080
    assert tot == int(seg.astype(np.int64).sum())  # checksum of checksums
    assert seg.min() >= 64 and seg.max() <= 64 * 8  # every path has 1..max_bounces segments
    assert np.all(seg[fid < 0] == 64)  # primary miss: exactly one segment per path
    assert np.all(np.isinf(ft[fid < 0])) and np.all(ft[fid >= 0] > 0.001)
    assert np.all((fb >= 0) & (fb <= 1.0))
    # idempotence: a second run is bit-identical although lanes pick up work in a different order
    fb2, fid2, ft2, seg2, tot2 = sc.render(c2["gen"], c2["spp"], st, seed=c2["seed"])
    assert tot2 == tot and np.array_equal(fb2, fb) and np.array_equal(seg2, seg) and np.array_equal(fid2, fid)
    # sampled cells against the oracle: first-hit id/t, segment count and pixel value bit-exact
    rng = np.random.default_rng(0)
    hit_cells = np.flatnonzero(fid >= 0)
    cells = np.concatenate([rng.choice(n, 300, replace=False), rng.choice(hit_cells, 300, replace=False)])
    osc = ob.OracleScene(c2["objects"])
    for c in cells[:: max(1, len(cells) // 120)]:
        ofb, ofid, oft, oseg, _ = osc.render(c2["gen"], c2["spp"], c2["infinity_distance"], c2["max_bounces"], seed=c2["seed"],
                                            cell_begin=int(c), cell_count=1)
        assert ofid[0] == fid[c] and oft[0] == ft[c] and oseg[0] == seg[c] and np.array_equal(ofb[0], fb[c]), int(c)
    # sharding: two interleaved halves reassemble the same image
    from raytracing_amd import sharding

    parts = []
    for rank in range(2):
        b, cnt, step = sharding.cell_shard(rank, 2, n)
        parts.append(sc.render(c2["gen"], c2["spp"], st, seed=c2["seed"], cell_begin=b, cell_count=cnt, cell_step=step)[0])
    assert np.array_equal(sharding.interleave_shards(np.stack(parts), n), fb)


def test_many_planes_and_far_hits(engine):
    """Far plane hits give bounce origins with large |o|: the filter's margin grows with |o|^2 and must stay conservative."""
    objs = list(random_scene(120, 0, seed=8))
    objs += [plane((0, -900, 0), (0, 1, 0), abi.RT_MAT_MIRROR), plane((0, 900, 0), (0.05, -1, 0.02), abi.RT_MAT_MIRROR),
             plane((4000, 0, 0), (-1, 0.01, 0), abi.RT_MAT_GLASS, 1.3)]
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    rays = random_rays(4000, seed=31)
    (pts, npts, recs), (opts, onpts, orecs) = _simulate_both(engine, objs, rays, 100.0, 10)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "far planes")
    assert np.array_equal(pts, opts)


def test_grazing_rays_stress_filter(engine):
    """Rays aimed at sphere limbs (|disc| tiny): the conservative filter must never drop a pair the exact test accepts."""
    rng = np.random.default_rng(5)
    objs = random_scene(400, 0, seed=21, extent=(60.0, 30.0, 60.0))
    pos, rad = objs["position"].astype(np.float64), objs["shape_param"][:, 0].astype(np.float64)
    n = 30000
    k = rng.integers(0, len(objs), n)
    o = rng.uniform(-1, 1, size=(n, 3)) * np.array([150.0, 80.0, 150.0])
    to_c = pos[k] - o
    dist = np.linalg.norm(to_c, axis=1, keepdims=True)
    u = np.cross(to_c, rng.normal(size=(n, 3)))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    # aim at a point on the limb, within +-1e-5 relative of the radius
    limb = pos[k] + u * (rad[k, None] * (1 + rng.uniform(-1e-5, 1e-5, size=(n, 1))))
    d = limb - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = make_rays(o.astype(np.float32), d.astype(np.float32))
    (pts, npts, recs), (opts, onpts, orecs) = _simulate_both(engine, objs, rays, 50.0, 4)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "grazing")


@pytest.mark.parametrize("n_spheres", [1025, 3000])
def test_render_parity_streamed_table(engine, n_spheres):
    """Scenes above the LDS tile size (1,024 spheres): both kernels stream the sphere table from L2 instead."""
    objs = random_scene(n_spheres, 2, seed=n_spheres, extent=(60.0, 30.0, 60.0))
    gen = projector((0.0, 0.0, -150.0), (-70.0, -40.0, -70.0), (140.0, 0.0, 0.0), (0.0, 80.0, 0.0), 48, 32)
    st = SimulationSettingsConfig(60.0, 6)
    sc = engine.Scene(objs)
    got = sc.render(gen, 3, st, seed=9)
    want = ob.OracleScene(objs).render(gen, 3, 60.0, 6, seed=9, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)
    assert got[4] == want[4]
    # general kernel (ParallelGrid) on the same scene
    gen2 = parallel_grid((-60, -30, -100), (120, 0, 0), (0, 60, 0), 40, 24, (0.02, 0.01, 1.0))
    got = sc.render(gen2, 2, st, seed=9)
    want = ob.OracleScene(objs).render(gen2, 2, 60.0, 6, seed=9, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)


def test_all_shapes_and_rotations(engine):
    """template.toml's object set (Box, Cylinder, Wedge, Lens, Plane) plus Cone, CSG trees, rotated Sphere / Plane and an
    ObjectGrid of plain spheres: every ShapeConfig arm (shape_config.rs:9-51) against the oracle, records bit-exact."""
    from raytracing_amd.config import SimulationConfig

    cfg = SimulationConfig.from_toml_str(TEMPLATE_LIKE_TOML)
    objs, nodes = cfg.scene.object_and_node_lists()
    rays = cfg.scene.ray_list()
    assert len(rays) == 24 * 32 + 20 * 30 + 1 and len(objs) == 19 and len(nodes) == 8
    st = cfg.simulation_settings
    sc = engine.Scene(objs, nodes=nodes)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=3)
    opts, onpts, orecs, _ = ob.OracleScene(objs, nodes).simulate(rays, st.infinity_distance, st.max_bounces, seed=3)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "all shapes")
    assert same(pts, opts)
    hit_objs = set(np.unique(recs["obj_id"][recs["obj_id"] >= 0]).tolist())
    # every complex object is exercised except the capped Cylinder (1): with the reference's in/out state machine a ray that
    # starts inside both cap half-spaces never registers as inside the intersection (csg.rs:28-29) — mirrored, not fixed
    assert {0, 2, 3, 4, 5, 6, 7, 8, 9}.issubset(hit_objs), hit_objs
    # render loop (both generators -> general kernel and pinhole kernel with the complex path)
    for gen in cfg.scene.generators():
        got = sc.render(gen, 3, st, seed=8)
        want = ob.OracleScene(objs, nodes).render(gen, 3, st.infinity_distance, st.max_bounces, seed=8, threads=4)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        assert got[4] == want[4]


def test_complex_objects_random_rays(engine):
    """Random rays (many starting inside shapes) through rotated boxes / cylinders / lenses among 300 spheres."""
    from raytracing_amd.engine import shape_object

    rng = np.random.default_rng(12)
    objs = list(random_scene(300, 1, seed=44))
    kinds = [(abi.RT_SHAPE_BOX, (3.0, 2.0, 4.0)), (abi.RT_SHAPE_CYLINDER, (6.0, 1.5)), (abi.RT_SHAPE_CONE, (25.0, 5.0)),
             (abi.RT_SHAPE_WEDGE, (5.0, 4.0, 3.0, 20.0)), (abi.RT_SHAPE_LENS, (1.5, 8.0, 12.0, -9.0)),
             (abi.RT_SHAPE_LENS, (1.0, 6.0, float("inf"), -7.0)), (abi.RT_SHAPE_SPHERE, (2.0,)), (abi.RT_SHAPE_PLANE, (0.3, 1.0, -0.2))]
    mats = [(abi.RT_MAT_GLASS, 1.5), (abi.RT_MAT_MIRROR, 0.0), (abi.RT_MAT_HALF_MIRROR, 0.5)]
    for k in range(16):
        shape, sp = kinds[k % len(kinds)]
        mk, mp = mats[k % 3]
        pos = (rng.uniform(-1, 1, 3) * np.array([18.0, 9.0, 18.0])).astype(np.float32)
        objs.insert(int(rng.integers(0, len(objs))), shape_object(shape, sp, pos, mk, mp, rotation_y_deg=float(rng.uniform(-180, 180))))
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    rays = random_rays(6000, seed=3)
    (pts, npts, recs), (opts, onpts, orecs) = _simulate_both(engine, objs, rays, 60.0, 8)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "complex random")
    assert same(pts, opts)


def test_csg_sees_the_closest_t_of_the_objects_before_it(engine):
    """scene.rs:54: a CSG tree is evaluated with t_max = closest t of the objects earlier in the list, which decides the 1e-6
    left/right attribution of csg.rs:33 when child surfaces nearly coincide there (tests/test_oracle.py::
    test_csg_result_depends_on_the_closest_t_so_far derives the expected winners by hand).  The kernels find the closest
    sphere / plane of the whole list first, so this is the case that needs the exact prefix minimum."""
    from test_oracle import _window_scene

    nodes, diff, p_at_5, p_at_tb, t_a, t_b = _window_scene()
    rays = make_rays([(0, 0, 0)] * 70, [(0, 0, 1)] * 70)
    st = SimulationSettingsConfig(100.0, 1)
    for objs, want_id, want_t in [([diff], 0, t_b), ([p_at_5, diff], 0, np.float32(5.0)), ([diff, p_at_5], 0, t_b),
                                  ([p_at_5, diff, p_at_tb], 2, t_b), ([diff, p_at_tb], 0, t_b),
                                  ([sphere((0, 0, 6), 1.0, abi.RT_MAT_MIRROR), diff, p_at_tb], 2, t_b)]:
        objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
        pts, npts, recs = engine.Scene(objs, nodes=nodes).simulate_rays_raw(st, rays=rays, seed=1)
        opts, onpts, orecs, _ = ob.OracleScene(objs, nodes).simulate(rays, 100.0, 1, seed=1)
        assert_records_equal(recs, orecs, "csg window")
        assert np.all(recs["obj_id"][:, 0] == want_id) and np.all(recs["t"][:, 0] == want_t)
        # the render loop's kernels carry the same per-lane path
        gen = projector((0.0, 0.0, 0.0), (-0.001, -0.001, 1.0), (0.002, 0.0, 0.0), (0.0, 0.002, 0.0), 1, 1)
        gen.corner[0], gen.corner[1] = 0.0, 0.0  # the single cell's ray is exactly +z
        got = engine.Scene(objs, nodes=nodes).render(gen, 2, st, seed=1)
        want = ob.OracleScene(objs, nodes).render(gen, 2, 100.0, 1, seed=1)
        assert got[1][0] == want[1][0] == want_id and got[2][0] == want[2][0] == want_t


def test_gpu_matches_golden_allshapes(engine):
    g = np.load(os.path.join(GOLD, "allshapes_paths.npz"))
    sc = engine.Scene(g["objects"], nodes=g["nodes"])
    pts, npts, recs = sc.simulate_rays_raw(SimulationSettingsConfig(50.0, 10), rays=g["rays"], seed=3)
    assert np.array_equal(npts, g["n_points"]) and same(pts, g["points"])
    assert_records_equal(recs, g["records"], "allshapes golden")


@pytest.mark.parametrize("spp,max_b,w,h", [(1, 1, 40, 30), (5, 2, 33, 17), (13, 8, 24, 24), (8, 3, 1, 1), (17, 12, 2, 70)])
def test_render_ragged_sample_blocks_and_depths(engine, spp, max_b, w, h):
    """spp not a multiple of the 8-sample work unit, depth 1 (no continuing rays), single-cell and skinny grids; planes and
    HalfMirrors in the scene; strided cell ranges.  Pinhole kernel and general kernel against the oracle."""
    objs = random_scene(150, 2, seed=spp * 7 + max_b, extent=(25.0, 15.0, 25.0))
    st = SimulationSettingsConfig(35.0, max_b)
    sc = engine.Scene(objs)
    osc = ob.OracleScene(objs)
    gens = [projector((0.0, 1.0, -60.0), (-30.0, -18.0, -25.0), (60.0, 0.0, 0.0), (0.0, 36.0, 0.0), w, h, 1.0),
            parallel_grid((-28, -16, -50), (56, 0, 0), (0, 32, 0), w, h, (0.03, -0.02, 1.0), 1.0)]
    n = w * h
    for gen in gens:
        for begin, step in ((0, 1), (n // 3, 2)):
            count = (n - begin + step - 1) // step
            got = sc.render(gen, spp, st, seed=77, cell_begin=begin, cell_count=count, cell_step=step)
            want = osc.render(gen, spp, 35.0, max_b, seed=77, cell_begin=begin, cell_count=count, cell_step=step, threads=4)
            for a, b in zip(got[:4], want[:4]):
                assert np.array_equal(a, b)
            assert got[4] == want[4]


def test_simulate_many_rays_throughput_path(engine):
    """2 x 10^5 explicit rays through the C2 scene: the general kernel with polyline output, sampled against the oracle."""
    c2 = synth.config_c2(500, 400)
    rays = ob.generate_rays(c2["gen"])
    sc = engine.Scene(c2["objects"])
    st = SimulationSettingsConfig(c2["infinity_distance"], c2["max_bounces"])
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=5)
    idx = np.random.default_rng(1).choice(len(rays), 3000, replace=False)
    opts, onpts, orecs, _ = ob.OracleScene(c2["objects"]).simulate(rays[idx], c2["infinity_distance"], c2["max_bounces"], seed=5)
    # ray ids differ (sub-batch), which only matters after a HalfMirror hit: compare rays that never met one
    mats = c2["objects"]["material_kind"]
    clean = np.array([not np.any(mats[orecs["obj_id"][k][orecs["obj_id"][k] >= 0]] == abi.RT_MAT_HALF_MIRROR) for k in range(len(idx))])
    assert clean.sum() > 2500
    assert np.array_equal(npts[idx][clean], onpts[clean]) and np.array_equal(pts[idx][clean], opts[clean])


def _soup(n_tris, seed, extent=15.0, size=3.0):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, size=(n_tris, 1, 3))
    v = (c + rng.normal(size=(n_tris, 3, 3)) * size).astype(np.float32).reshape(-1, 3)
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    return v, idx


def test_triangle_meshes_bvh_equals_brute_force(engine):
    """Extension (no reference counterpart): the BVH traversal must return exactly the brute-force nearest triangle
    (t bit-exact, ties to the lowest index) for a random triangle soup, a shared-vertex heightfield and an empty mesh,
    under rotated transforms, mixed with spheres and planes."""
    from raytracing_amd.engine import mesh_object

    soup = _soup(3000, 1)
    hf = synth.heightfield_mesh(40, 40, extent=(20.0, 20.0), amplitude=3.0)
    dup = (np.array([[-5, -5, 0], [5, -5, 0], [0, 5, 0]] * 2, np.float32), np.array([[0, 1, 2], [3, 4, 5]], np.uint32))  # coincident pair
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    objs = list(random_scene(80, 1, seed=5))
    objs.insert(3, mesh_object(0, (0, 0, 0), abi.RT_MAT_GLASS, 1.4, rotation_y_deg=20.0))
    objs.insert(10, mesh_object(1, (0, -12, 0), abi.RT_MAT_MIRROR, 0.0, rotation_y_deg=-35.0))
    objs.insert(20, mesh_object(2, (0, 0, 22), abi.RT_MAT_HALF_MIRROR, 0.5))
    objs.append(mesh_object(3, (0, 0, 0), abi.RT_MAT_MIRROR))
    objs.append(mesh_object(1, (30, 5, 0), abi.RT_MAT_GLASS, 1.2, rotation_y_deg=90.0))  # a second instance of mesh 1
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    meshes = [soup, hf, dup, empty]
    rays = random_rays(8000, seed=17, extent=(28.0, 14.0, 28.0))
    sc = engine.Scene(objs, meshes=meshes)
    st = SimulationSettingsConfig(50.0, 6)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=2)
    opts, onpts, orecs, _ = ob.OracleScene(objs, meshes=meshes).simulate(rays, 50.0, 6, seed=2)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, "meshes")
    assert same(pts, opts)
    # grazing rays: nearly in the plane of a flat sheet, where Moller-Trumbore is worst conditioned
    flat = synth.heightfield_mesh(24, 24, extent=(20.0, 20.0), amplitude=0.0)
    rng = np.random.default_rng(3)
    o = np.stack([rng.uniform(-30, 30, 4000), rng.uniform(-1e-3, 1e-3, 4000), rng.uniform(-30, 30, 4000)], 1)
    tgt = np.stack([rng.uniform(-20, 20, 4000), rng.uniform(-1e-4, 1e-4, 4000), rng.uniform(-20, 20, 4000)], 1)
    dirs = tgt - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    graze = make_rays(o, dirs)
    gobjs = np.array([mesh_object(0, (0, 0, 0), abi.RT_MAT_MIRROR, rotation_y_deg=33.0)], dtype=abi.OBJECT_DTYPE)
    gp, gn, gr = engine.Scene(gobjs, meshes=[flat]).simulate_rays_raw(SimulationSettingsConfig(50.0, 3), rays=graze, seed=1)
    wp, wn, wr, _ = ob.OracleScene(gobjs, meshes=[flat]).simulate(graze, 50.0, 3, seed=1)
    assert np.array_equal(gn, wn) and same(gp, wp)
    assert_records_equal(gr, wr, "grazing")
    assert (gn > 2).sum() > 500
    hit = set(np.unique(recs["obj_id"][recs["obj_id"] >= 0]).tolist())
    mesh_ids = set(np.flatnonzero(objs["shape"] == abi.RT_SHAPE_MESH).tolist())
    assert len(hit & mesh_ids) >= 3
    # render loop on the reduced C3 scene (pinhole kernel + mesh)
    c3 = synth.config_c3(48, 27, nx=32, nz=32)
    sc3 = engine.Scene(c3["objects"], meshes=c3["meshes"])
    st3 = SimulationSettingsConfig(c3["infinity_distance"], c3["max_bounces"])
    got = sc3.render(c3["gen"], 3, st3, seed=1)
    want = ob.OracleScene(c3["objects"], meshes=c3["meshes"]).render(c3["gen"], 3, c3["infinity_distance"], c3["max_bounces"], seed=1, threads=4)
    for a, b in zip(got[:4], want[:4]):
        assert same(a, b)
    assert got[4] == want[4] and (got[1] == 0).sum() > 50  # the mesh (object 0) is what most primary rays hit


def test_device_pointer_entry_points_match_host_ones(engine):
    """rt_hip_simulate_dev / rt_hip_render_dev (device buffers, caller's stream) against the host-buffer entry points."""
    import torch

    objs = random_scene(150, 2, seed=9)
    rays = random_rays(5000, seed=4)
    sc = engine.Scene(objs)
    st = SimulationSettingsConfig(40.0, 7)
    pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=5, ray_id_base=77)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        d_rays = torch.from_numpy(rays.view(np.uint8).reshape(len(rays), -1)).to(dev)
        d_pts = torch.zeros((len(rays), 8, 3), dtype=torch.float32, device=dev)
        d_np = torch.zeros(len(rays), dtype=torch.int32, device=dev)
        d_rec = torch.zeros((len(rays), 7, abi.RECORD_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        d_tot = torch.zeros(1, dtype=torch.int64, device=dev)
        sc.simulate_rays_dev(st, d_rays.data_ptr(), len(rays), 5, 77, d_pts.data_ptr(), d_np.data_ptr(), d_rec.data_ptr(), d_tot.data_ptr(),
                             side.cuda_stream)
    side.synchronize()
    assert np.array_equal(d_np.cpu().numpy().astype(np.uint32), npts)
    assert same(d_pts.cpu().numpy(), pts)
    assert_records_equal(d_rec.cpu().numpy().view(abi.RECORD_DTYPE).reshape(recs.shape), recs, "simulate_dev")
    assert int(d_tot.item()) == int((npts - 1).sum())  # one segment per bounce-loop iteration


def test_streamed_tables_with_meshes_and_csg(engine):
    """The four kernel variants that carry the per-lane object path — {LDS tables, streamed tables} x {pinhole, general} — on
    one scene: 1,100 spheres (above the LDS tile), two planes, a mesh (LDS tree top in the pinhole kernel), a rotated
    box and a lens, through Projector, ParallelGrid and explicit rays; then the same objects with 200 spheres."""
    from raytracing_amd.engine import mesh_object, shape_object

    hf = synth.heightfield_mesh(24, 24, extent=(50.0, 50.0), amplitude=4.0)
    for n_spheres in (1100, 200):
        objs = list(random_scene(n_spheres, 2, seed=21, extent=(60.0, 30.0, 60.0)))
        objs.insert(5, mesh_object(0, (0, -25, 0), abi.RT_MAT_MIRROR, rotation_y_deg=12.0))
        objs.insert(40, shape_object(abi.RT_SHAPE_BOX, (8.0, 8.0, 8.0), (10, 5, -80), abi.RT_MAT_GLASS, 1.5, rotation_y_deg=30.0))
        objs.append(shape_object(abi.RT_SHAPE_LENS, (2.0, 10.0, 12.0, 12.0), (-15, 0, -85), abi.RT_MAT_GLASS, 1.5))
        objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
        sc = engine.Scene(objs, meshes=[hf])
        osc = ob.OracleScene(objs, meshes=[hf])
        st = SimulationSettingsConfig(60.0, 6)
        gen = projector((0.0, 10.0, -150.0), (-70.0, -50.0, -70.0), (140.0, 0.0, 0.0), (0.0, 80.0, 0.0), 40, 28)
        got = sc.render(gen, 3, st, seed=9)
        want = osc.render(gen, 3, 60.0, 6, seed=9, threads=8)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        assert got[4] == want[4]
        first = set(np.unique(got[1]).tolist())
        assert 5 in first and (40 in first or len(objs) - 1 in first)  # the mesh and a CSG object are primary hits somewhere
        gen2 = parallel_grid((-60, -30, -120), (120, 0, 0), (0, 60, 0), 30, 20, (0.02, -0.1, 1.0))
        got = sc.render(gen2, 2, st, seed=9)
        want = osc.render(gen2, 2, 60.0, 6, seed=9, threads=8)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        rays = random_rays(3000, seed=8, extent=(70.0, 35.0, 70.0))
        pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=4)
        opts, onpts, orecs, _ = osc.simulate(rays, 60.0, 6, seed=4)
        assert np.array_equal(npts, onpts) and same(pts, opts)
        assert_records_equal(recs, orecs, f"mixed-{n_spheres}")


def test_render_sharded_assembles_the_single_gpu_image(engine):
    """sharding.render_sharded as each of 3 ranks would run it (several launches per rank), shards interleaved as the
    all-gather would: bit-identical to one full-frame render.  (The collective itself is covered by the gloo test.)"""
    import torch

    from raytracing_amd import sharding

    cfg = synth.config_c2(50, 31)
    sc = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    full = sc.render(cfg["gen"], 5, st, seed=3)
    n, world = 50 * 31, 3
    dev = torch.device("cuda", 0)
    shards, segs = [], 0
    for rank in range(world):
        fb, s_ = _render_rank(sc, cfg, st, rank, world, dev)
        shards.append(fb)
        segs += s_
    got = sharding.interleave_shards(torch.stack(shards), n).cpu().numpy()
    assert np.array_equal(got, full[0]) and segs == full[4]
    one, s1 = sharding.render_sharded(sc, cfg["gen"], 5, st, 3, 0, 1, None, dev, max_cells_per_launch=300)
    assert np.array_equal(one.cpu().numpy(), full[0]) and s1 == full[4]


def _render_rank(sc, cfg, st, rank, world, dev):
    """render_sharded's local part for one rank of `world` (no process group here): its padded shard and segment count."""
    import torch

    from raytracing_amd import sharding

    class NoGather:  # stands in for torch.distributed: hands the local shard back
        pass

    orig = sharding.gather_image
    sharding.gather_image = lambda local, r, w, n, dist: local
    try:
        return sharding.render_sharded(sc, cfg["gen"], 5, st, 3, rank, world, NoGather(), dev, max_cells_per_launch=200)
    finally:
        sharding.gather_image = orig


@pytest.mark.parametrize("case", ["centered", "offset", "skewed"])
def test_projector_rays_grazing_sphere_limbs(engine, case):
    """Stress for the shared-origin (affine) filter of the Projector kernel: every sphere is placed so that the ray of one
    particular cell passes within +-1e-5 r (some within +-1e-7 r) of its limb, at assorted distances, for projectors near
    the origin, far from it (large coordinates: more rounding in the target) and with a skewed target parallelogram.
    First-hit ids, t and the image must still equal the oracle's, i.e. the filter never dropped a pair the exact test accepts."""
    rng = np.random.default_rng({"centered": 1, "offset": 2, "skewed": 3}[case])
    cu, cv = 96, 64
    if case == "centered":
        O, corner, u, v = (0.0, 0.0, -120.0), (-64.0, -36.0, -60.0), (128.0, 0.0, 0.0), (0.0, 72.0, 0.0)
    elif case == "offset":
        O, corner, u, v = (900.0, -400.0, 2000.0), (880.0, -415.0, 2030.0), (40.0, 0.0, 3.0), (0.0, 30.0, 1.0)
    else:
        O, corner, u, v = (5.0, 3.0, -40.0), (-30.0, -20.0, 10.0), (50.0, 10.0, 25.0), (-8.0, 45.0, 5.0)
    gen = projector(O, corner, u, v, cu, cv)
    f32 = np.float32
    O32, c32 = np.array(O, f32), np.array(corner, f32)
    us, vs = (np.array(u, f32) / f32(cu)).astype(f32), (np.array(v, f32) / f32(cv)).astype(f32)
    n = 700
    ci, cj = rng.integers(0, cu, n), rng.integers(0, cv, n)
    target = ((c32 + ci[:, None].astype(f32) * us).astype(f32) + cj[:, None].astype(f32) * vs).astype(f32)
    d = (target - O32).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rad = rng.uniform(0.3, 3.0, n)
    dist = rng.uniform(1.2, 6.0, n) * np.linalg.norm(np.array(u)) + 5.0 * rad
    perp = np.cross(d, rng.normal(size=(n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True)
    eps = np.where(rng.random(n) < 0.3, 1e-7, 1e-5) * rng.uniform(-1, 1, n)
    centre = np.array(O) + d * dist[:, None] + perp * (rad * (1.0 + eps))[:, None]
    kinds = [abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR]
    objs = np.array([sphere(centre[k], rad[k], kinds[k % 3], [1.5, 0.0, 0.5][k % 3]) for k in range(n)], dtype=abi.OBJECT_DTYPE)
    st = SimulationSettingsConfig(80.0, 4)
    got = engine.Scene(objs).render(gen, 2, st, seed=6)
    want = ob.OracleScene(objs).render(gen, 2, 80.0, 4, seed=6, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert same(a, b)
    assert got[4] == want[4]
    assert (got[1] >= 0).mean() > 0.05  # the spheres are where the rays go


def test_projector_with_origin_in_the_target_plane_uses_the_general_kernel(engine):
    """rho = inf: the shared-origin filter has no rounding budget, the launch must take the general kernel
    and still equal the oracle.  Also a projector whose plane is merely very close to its origin (large rho, loose filter)."""
    objs = random_scene(120, 1, seed=33, extent=(10.0, 10.0, 10.0), r_min=0.3, r_max=1.0)
    st = SimulationSettingsConfig(30.0, 5)
    sc, osc = engine.Scene(objs), ob.OracleScene(objs)
    for O, corner in [((0.0, 0.0, 0.0), (-6.0, -6.0, 0.0)), ((0.0, 0.0, -0.004), (-6.0, -6.0, 0.0)), ((0.0, 0.0, -0.5), (-6.0, -6.0, 0.0))]:
        gen = projector(O, corner, (12.0, 0.0, 0.0), (0.0, 12.0, 0.0), 40, 30)
        got = sc.render(gen, 2, st, seed=2)
        want = osc.render(gen, 2, 30.0, 5, seed=2, threads=4)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        assert got[4] == want[4]


def test_projector_filter_differential_fuzz():
    """tools/fuzz_pinhole.py: 300 random projector / sphere-cloud configurations rendered with the shared-origin filter and
    with the reference's own discriminant as the filter (scan mode 1, general kernel) must be bit-identical."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_pinhole.py"), "300", "7000"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "300 cases, 0 mismatches" in r.stdout


def test_fuzz_explicit_ray_kernel_against_general_kernel():
    """tools/fuzz_paths.py: 300 of the same random configurations as ray lists (generator order, shuffled, parallel sheets) and as
    ParallelGrid renders on paths_kernel and on the general kernel (RT_HIP_NO_PATHS=1) must be byte-identical."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_paths.py"), "300", "9000"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "300 cases, 0 mismatching" in r.stdout


@pytest.mark.parametrize("in_memory", [False, True])
def test_fuzz_renders_against_the_oracle(in_memory):
    """tests/fuzz_oracle.py: 500 of the random projector / sphere-cloud configurations rendered on the GPU and by the CPU oracle, with
    the grid where the library puts it and forced into global memory (15,000 cases ran clean when this was written)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **({"RT_HIP_GRID_IN_MEMORY": "1"} if in_memory else {}))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_oracle.py"), "500", "11000" if in_memory else "12000"], capture_output=True, text=True,
                       timeout=280, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "500 cases, 0 mismatching" in r.stdout


@pytest.mark.parametrize("which", ["c4", "c3"])
def test_full_resolution_c4_c3_properties(engine, which):
    """BASELINE.json configs[3] (10,000 spheres, 3840x2160) and configs[2] (100,352-triangle mesh, 1920x1080) at their full
    resolution and scene size, 8 spp instead of 1,024 / 256 (the sample count only repeats the per-cell loop): size-independent
    properties, idempotence, sharded reassembly, and sampled cells against the oracle bit for bit."""
    from raytracing_amd import sharding

    cfg = synth.config_c4() if which == "c4" else synth.config_c3()
    spp = 8
    meshes = cfg.get("meshes")
    sc = engine.Scene(cfg["objects"], meshes=meshes)
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    g = cfg["gen"]
    n = g.count_u * g.count_v
    fb, fid, ft, seg, tot = sc.render(g, spp, st, seed=cfg["seed"])
    assert tot == int(seg.astype(np.int64).sum())
    assert seg.min() >= spp and seg.max() <= spp * cfg["max_bounces"]
    assert np.all(seg[fid < 0] == spp) and np.all(np.isinf(ft[fid < 0])) and np.all(ft[fid >= 0] > 0.001)
    assert np.all((fb >= 0) & (fb <= 1.0))
    assert 0.05 < (fid >= 0).mean() < 0.95
    fb2, fid2, ft2, seg2, tot2 = sc.render(g, spp, st, seed=cfg["seed"])
    assert tot2 == tot and np.array_equal(fb2, fb) and np.array_equal(seg2, seg) and np.array_equal(fid2, fid)
    parts = []
    for rank in range(3):
        b, cnt, step = sharding.cell_shard(rank, 3, n)
        parts.append(sc.render(g, spp, st, seed=cfg["seed"], cell_begin=b, cell_count=cnt, cell_step=step)[0])
    cap = sharding.shard_capacity(3, n)
    parts = [np.concatenate([q, np.zeros((cap - len(q), 3), np.float32)]) for q in parts]
    assert np.array_equal(sharding.interleave_shards(np.stack(parts), n), fb)
    rng = np.random.default_rng(1)
    cells = np.concatenate([rng.choice(n, 30, replace=False), rng.choice(np.flatnonzero(fid >= 0), 30, replace=False)])
    osc = ob.OracleScene(cfg["objects"], meshes=meshes)
    for c in cells:
        ofb, ofid, oft, oseg, _ = osc.render(g, spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], cell_begin=int(c), cell_count=1)
        assert ofid[0] == fid[c] and oft[0] == ft[c] and oseg[0] == seg[c] and np.array_equal(ofb[0], fb[c]), int(c)


@pytest.mark.parametrize("which", ["c4", "c3"])
def test_full_spp_cells_against_oracle(engine, which):
    """BASELINE.json configs[3] / configs[2] at their FULL sample counts (1,024 / 256 spp) on sampled cells, every output
    bit for bit against the oracle.  C4's cells are taken from the end of the grid, where path_id = cell * spp + s exceeds
    2^32, so the high counter word of the HalfMirror stream (rt_device.h uniform01, scene.rs:88) is non-zero; a third of
    the cells are chosen because their primary hit IS a HalfMirror sphere."""
    cfg = synth.config_c4() if which == "c4" else synth.config_c3()
    spp = cfg["spp"]
    assert spp == (1024 if which == "c4" else 256)
    meshes = cfg.get("meshes")
    sc = engine.Scene(cfg["objects"], meshes=meshes)
    osc = ob.OracleScene(cfg["objects"], meshes=meshes)
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    g = cfg["gen"]
    n = g.count_u * g.count_v
    # probe pass (1 spp) over a band of cells to learn what each cell's primary ray hits
    band = 400_000
    # C4: columns i = 2000 .. 2185 look into the sphere cloud and already have cell * 1024 >= 2^32 (cell >= 4,194,304)
    begin = 2000 * g.count_v if which == "c4" else n // 2 - band // 2
    _, fid, _, _, _ = sc.render(g, 1, st, seed=cfg["seed"], cell_begin=begin, cell_count=band)
    mats = cfg["objects"]["material_kind"]
    kind = np.where(fid >= 0, mats[np.clip(fid, 0, None)], 255)
    rng = np.random.default_rng(4)
    hm = np.flatnonzero(kind == abi.RT_MAT_HALF_MIRROR)
    other_hit = np.flatnonzero((kind != abi.RT_MAT_HALF_MIRROR) & (fid >= 0))
    assert len(hm) >= 10 and len(other_hit) >= 10
    cells = begin + np.unique(np.concatenate([rng.choice(hm, 10, replace=False), rng.choice(other_hit, 10, replace=False),
                                              rng.choice(band, 10, replace=False)]))
    if which == "c4":
        assert np.all((cells.astype(np.uint64) * np.uint64(spp)) >> np.uint64(32) >= 1)  # 64-bit path ids
    varied = 0
    for c in cells:
        c = int(c)
        got = sc.render(g, spp, st, seed=cfg["seed"], cell_begin=c, cell_count=1)
        want = osc.render(g, spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], cell_begin=c, cell_count=1, threads=4)
        assert got[1][0] == want[1][0] and got[2][0] == want[2][0], c          # first-hit id and t
        assert got[3][0] == want[3][0] and got[4] == want[4], c                # segment counts
        assert np.array_equal(got[0], want[0]), c                              # accumulated pixel
        if kind[c - begin] == abi.RT_MAT_HALF_MIRROR:
            # identical primary rays, so the samples differ only through the HalfMirror draws: with a working stream the
            # cell's segment count is neither "all reflected" nor "all passed"
            per_path = want[3][0] / spp
            varied += int(per_path != int(per_path))
    assert varied >= 5
    # several cells in one launch, strided (what a rank of the tile-sharded job runs), same sample count
    b0, step, cnt = int(cells[0]), 4099, 12
    if b0 + (cnt - 1) * step < n:
        got = sc.render(g, spp, st, seed=cfg["seed"], cell_begin=b0, cell_count=cnt, cell_step=step)
        want = osc.render(g, spp, cfg["infinity_distance"], cfg["max_bounces"], seed=cfg["seed"], cell_begin=b0, cell_count=cnt, cell_step=step, threads=8)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        assert got[4] == want[4]


def test_calls_from_another_host_thread_and_stream(engine):
    """Every entry point selects the init device itself (HIP's current device is per thread), and a _dev call on another
    stream than its predecessor waits for it: a worker thread rendering on its own stream, interleaved with the main thread's
    calls on the default stream, gets the same images as a single-threaded run."""
    import threading

    import torch

    cfg = synth.config_c2(120, 70)
    sc = engine.Scene(cfg["objects"])
    st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
    n = 120 * 70
    want = {seed: sc.render(cfg["gen"], 3, st, seed=seed) for seed in (1, 2)}
    dev = torch.device("cuda", 0)
    results, errors = {}, []

    def run(seed, reps):
        try:
            stream = torch.cuda.Stream(device=dev)
            fb = torch.zeros((n, 3), dtype=torch.float32, device=dev)
            seg = torch.zeros(n, dtype=torch.int32, device=dev)
            tot = torch.zeros(1, dtype=torch.int64, device=dev)
            for _ in range(reps):
                sc.render_dev(cfg["gen"], 3, st, seed, 0, n, 1, fb.data_ptr(), 0, 0, seg.data_ptr(), tot.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            results[seed] = (fb.cpu().numpy(), seg.cpu().numpy(), int(tot.item()))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=run, args=(2, 6))
    t.start()
    run(1, 6)
    t.join()
    assert not errors, errors
    for seed in (1, 2):
        fb, seg, tot = results[seed]
        assert np.array_equal(fb, want[seed][0]) and np.array_equal(seg.astype(np.uint32), want[seed][3]) and tot == want[seed][4]


def test_bench_sharded_path_and_stored_checksums(engine):
    """bench.py's frame step (sharding.ShardedFrame on rt_hip_render_dev) as ranks of a 3-GPU job would run it, assembled as
    the all-gather would, and the single-rank run, against the oracle-made checksums in tests/golden/frame_checksums.json."""
    import json

    import torch

    from raytracing_amd import sharding

    stored = json.load(open(os.path.join(GOLD, "frame_checksums.json")))
    dev = torch.device("cuda", 0)
    for name, w, h, spp in [("c2", 96, 54, 4), ("c4", 24, 16, 2)]:
        cfg = (synth.config_c2 if name == "c2" else synth.config_c4)(w, h)
        sc = engine.Scene(cfg["objects"])
        st = SimulationSettingsConfig(cfg["infinity_distance"], cfg["max_bounces"])
        stream = torch.cuda.current_stream().cuda_stream

        def render_cells(begin, count, step, fb_rows, total_slot):
            sc.render_dev(cfg["gen"], spp, st, cfg["seed"], begin, count, step, fb_rows.data_ptr(), 0, 0, 0, total_slot.data_ptr(), stream)

        key = f"{name}:{w}x{h}:spp{spp}:seed{cfg['seed']}:depth{cfg['max_bounces']}"
        one = sharding.ShardedFrame(render_cells, w * h, 0, 1, None, dev, max_cells_per_launch=1000)
        one.step()
        assert sharding.frame_checksum(one.image()) == int(stored[key], 16)
        shards, segs = [], 0
        for rank in range(3):
            f = sharding.ShardedFrame(render_cells, w * h, rank, 3, None, dev, max_cells_per_launch=700)
            f.render()
            shards.append(f.local.clone())
            segs += f.segments()
        img = sharding.interleave_shards(torch.stack(shards), w * h)
        assert sharding.frame_checksum(img) == int(stored[key], 16) and segs == one.segments()


@pytest.mark.parametrize("case,seed0", [(497, 200), (441, 200), (6, 0), (36, 0), (91, 0), (16, 0), (2, 0)])
def test_grid_and_cull_list_cases_against_the_oracle(engine, case, seed0):
    """Configurations of tools/fuzz_pinhole.py that once broke the Projector kernel's cull list / grid walk, replayed against the
    oracle: a sphere grazed within the reference's absolute 1e-6 discriminant threshold in a scene 0.2 units across (the walk's
    early stop took it for a hit: cases 497, 441 of seed 200), boxes entered from inside (NaN rays, cases 6, 36, 91), a crowd of
    spheres around the projector origin (more than a cull list holds, case 16), a mirror plane behind the cloud (continuing rays
    that start outside the grid's origin range, case 2)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import fuzz_pinhole as fz

    fz.seed0 = seed0
    gen, objs, st, scale, n_sph, cu, cv = fz.make_case(case)
    got = engine.Scene(objs).render(gen, 2, st, seed=case)
    want = ob.OracleScene(objs).render(gen, 2, st.infinity_distance, st.max_bounces, seed=case, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert same(a, b)
    assert got[4] == want[4]


@pytest.mark.parametrize("n_extra", [20, 400, 1200])
def test_nan_rays_from_box_interiors_hit_boxes_as_in_the_reference(engine, n_extra):
    """A ray that starts inside a rotation-0 box "enters" it at t_min with a zero normal (axis_aligned_box.rs:33-41, 70-95), so its
    scattered ray is NaN; the reference's slab test then drops the NaNs (f32::max / f32::min, axis_aligned_box.rs:24-29) and reports
    a hit at t_min on EVERY such box, wherever it is, until the bounce budget ends.  The conservative filters, the cull list and the
    grid walk must not lose those rays (they once did: a NaN fails every ">=").  Projector inside a glass box, no rotated box in
    front of the list to mask the effect; table scan (few objects), LDS-resident grid, streamed tables with the grid in memory."""
    from raytracing_amd.engine import shape_object

    rng = np.random.default_rng(31 + n_extra)
    objs = [shape_object(abi.RT_SHAPE_BOX, (6.0, 5.0, 7.0), (0.2, -0.1, 0.3), abi.RT_MAT_GLASS, 1.5)]
    kinds = [abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR]
    for k in range(n_extra):
        pos = rng.uniform(-40.0, 40.0, 3) + np.array([0.0, 0.0, 30.0])
        if k % 4 == 0:
            objs.append(shape_object(abi.RT_SHAPE_BOX, tuple(rng.uniform(0.5, 4.0, 3)), tuple(pos), kinds[k % 3], [1.5, 0.0, 0.5][k % 3]))
        else:
            objs.append(sphere(tuple(pos), float(rng.uniform(0.4, 2.5)), kinds[k % 3], [1.5, 0.0, 0.5][k % 3]))
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    st = SimulationSettingsConfig(80.0, 5)
    sc, osc = engine.Scene(objs), ob.OracleScene(objs)
    for O in ((0.0, 0.0, 0.0), (0.0, 0.0, -20.0)):  # inside box 0; outside, looking through it
        gen = projector(O, (-30.0, -22.0, 40.0), (60.0, 0.0, 0.0), (0.0, 44.0, 0.0), 48, 36)
        got = sc.render(gen, 3, st, seed=6)
        want = osc.render(gen, 3, 80.0, 5, seed=6, threads=8)
        for a, b in zip(got[:4], want[:4]):
            assert same(a, b)
        assert got[4] == want[4]
        if O == (0.0, 0.0, 0.0):
            assert got[4] == 48 * 36 * 3 * 5  # every path keeps hitting box 0 at t_min until the bounce budget ends


def test_object_grid_of_boxes_rides_the_scan_path(engine):
    """scene_config.rs:25-49 lets an ObjectGrid stamp any shape thousands of times.  Rotation-0 boxes (axis_aligned_box.rs:12-95
    inside a translation-only Transform) take the wave-parallel scan behind their bounding spheres; records, polylines and
    images must equal the oracle's, including rays that start inside a box (t = t_min, zero normal -> NaN, as the reference),
    rays along box faces / through edges, unit-cube and sliver boxes, mixed with spheres, planes, a rotated box (per-lane path)
    and enough objects to stream the tables (> 1,024)."""
    from raytracing_amd.config import SimulationConfig
    from raytracing_amd.engine import shape_object

    toml = """
[simulation_settings]
infinity_distance = 40.0
max_bounces = 6
[[scene.objects]]
shape = { type = "Plane", normal = [0.0, 1.0, 0.0] }
material = { type = "Mirror" }
transform = { position = [0.0, -12.0, 0.0], rotation_y_deg = 0.0 }
[[scene.objects]]
shape = { type = "Box", size = [3.0, 5.0, 2.0] }
material = { type = "Glass", ior = 1.4 }
transform = { position = [2.0, 6.0, -4.0], rotation_y_deg = 35.0 }
[[scene.object_generators]]
type = "ObjectGrid"
count_x = 12
count_z = 10
position_start = [-22.0, 0.0, -18.0]
step_x = [4.0, 0.3, 0.0]
step_z = [0.0, -0.2, 4.0]
template.shape = { type = "Box", size = [2.0, 3.0, 1.5] }
template.material = { type = "Glass", ior = 1.5 }
template.transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }
[[scene.object_generators]]
type = "ObjectGrid"
count_x = 6
count_z = 5
position_start = [-20.0, 7.0, -16.0]
step_x = [8.0, 0.0, 0.0]
step_z = [0.0, 0.0, 8.0]
template.shape = { type = "Box", size = [1.0, 1.0, 1.0] }
template.material = { type = "Mirror" }
template.transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }
[[scene.rays]]
origin = [0.0, 0.0, -30.0]
direction = [0.0, 0.0, 1.0]
"""
    cfg = SimulationConfig.from_toml_str(toml)
    objs, nodes = cfg.scene.object_and_node_lists()
    assert len(objs) == 2 + 120 + 30
    rng = np.random.default_rng(8)
    objs = list(objs) + list(random_scene(60, 0, seed=12, extent=(22.0, 8.0, 18.0)))
    objs.append(shape_object(abi.RT_SHAPE_BOX, (30.0, 0.02, 30.0), (0.0, -6.0, 0.0), abi.RT_MAT_HALF_MIRROR, 0.5))  # a sliver
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    rays = list(random_rays(5000, seed=6, extent=(24.0, 9.0, 20.0)))  # many start inside boxes
    pos = objs["position"][2:122]
    inside = make_rays(pos + rng.uniform(-0.4, 0.4, size=pos.shape).astype(np.float32), rng.normal(size=pos.shape).astype(np.float32))
    face = make_rays(pos + np.array([-5.0, 1.5, 0.0], np.float32), np.tile(np.array([1.0, 0.0, 0.0], np.float32), (len(pos), 1)))   # along the top face
    edge = make_rays(pos + np.array([-4.0, -4.0 + 1.5 - 1.0, 0.0], np.float32), np.tile(np.array([1.0, 1.0, 0.0], np.float32), (len(pos), 1)))  # towards an edge
    axis = make_rays(pos + np.array([0.0, 0.0, -6.0], np.float32), np.tile(np.array([0.0, 0.0, 1.0], np.float32), (len(pos), 1)))   # zero direction components
    rays = np.concatenate([np.array(rays, dtype=abi.RAY_DTYPE), inside, face, edge, axis])
    st = SimulationSettingsConfig(40.0, 6)
    for extra in (0, 1100):  # LDS-resident tables, then streamed tables
        o2 = objs if not extra else np.concatenate([objs, random_scene(extra, 0, seed=3, extent=(60.0, 30.0, 60.0))])
        sc, osc = engine.Scene(o2), ob.OracleScene(o2)
        pts, npts, recs = sc.simulate_rays_raw(st, rays=rays, seed=4)
        opts, onpts, orecs, _ = osc.simulate(rays, 40.0, 6, seed=4)
        assert np.array_equal(npts, onpts)
        assert_records_equal(recs, orecs, f"box grid +{extra}")
        assert same(pts, opts)
        hit = np.unique(recs["obj_id"][recs["obj_id"] >= 0])
        assert np.isin(np.arange(2, 122), hit).mean() > 0.9 and np.isnan(recs["normal"]).any()  # boxes are hit; inside starts give NaN normals
        for gen in (projector((0.0, 3.0, -60.0), (-30.0, -14.0, -25.0), (60.0, 0.0, 0.0), (0.0, 30.0, 0.0), 64, 40),
                    parallel_grid((-26, -10, -40), (52, 0, 0), (0, 22, 0), 48, 30, (0.02, -0.03, 1.0))):
            got = sc.render(gen, 3, st, seed=5)
            want = osc.render(gen, 3, 40.0, 6, seed=5, threads=8)
            for a, b in zip(got[:4], want[:4]):
                assert same(a, b)
            assert got[4] == want[4]
    try:  # exact-discriminant filter (scan mode 1): the bounding sphere's discriminant stands in for a box
        engine.set_scan_mode(1)
        a = engine.Scene(objs).simulate_rays_raw(st, rays=rays[:3000], seed=4)
    finally:
        engine.set_scan_mode(0)
    b = engine.Scene(objs).simulate_rays_raw(st, rays=rays[:3000], seed=4)
    assert np.array_equal(a[1], b[1])
    assert_records_equal(a[2], b[2], "box grid scan modes")


@pytest.mark.parametrize("n_spheres,spp,w,h,crowd", [(400, 8, 96, 70, 0), (400, 13, 64, 50, 0), (1500, 5, 70, 64, 0), (700, 9, 40, 200, 300)])
def test_cull_tile_lists_change_nothing(engine, n_spheres, spp, w, h, crowd):
    """The per-launch tile lists of the primary cull only narrow which spheres a group looks at: the same
    frame with RT_HIP_NO_TILES=1 (every group culls the whole table) is bit-identical, and both are the oracle's frame.
    Cases: whole and ragged sample blocks (lanes of a group drift apart, groups then span several tiles), strided cell
    ranges (a tile's cells are not neighbours), tall grids (a tile wraps from one column of cells to the next) and a crowd
    of spheres in front of the projector (tile lists overflow, those groups fall back to the table)."""
    objs = random_scene(n_spheres, 1, seed=n_spheres + spp, extent=(30.0, 18.0, 30.0))
    if crowd:
        rng = np.random.default_rng(5)
        extra = np.zeros(crowd, abi.OBJECT_DTYPE)
        extra["shape"] = abi.RT_SHAPE_SPHERE
        extra["position"] = (rng.normal(size=(crowd, 3)) * (2.0, 2.0, 0.5) + (0.0, 1.0, -40.0)).astype(np.float32)
        extra["shape_param"][:, 0] = rng.uniform(0.3, 1.2, crowd).astype(np.float32)
        extra["material_kind"] = abi.RT_MAT_MIRROR
        objs = np.concatenate([objs, extra])
    gen = projector((0.0, 1.0, -70.0), (-32.0, -19.0, -30.0), (64.0, 0.0, 0.0), (0.0, 38.0, 0.0), w, h, 1.0)
    st = SimulationSettingsConfig(40.0, 4)
    sc = engine.Scene(objs)
    osc = ob.OracleScene(objs)
    n = w * h
    for begin, step in ((0, 1), (n // 5, 3)):
        count = (n - begin + step - 1) // step
        got = sc.render(gen, spp, st, seed=31, cell_begin=begin, cell_count=count, cell_step=step)
        os.environ["RT_HIP_NO_TILES"] = "1"
        try:
            plain = sc.render(gen, spp, st, seed=31, cell_begin=begin, cell_count=count, cell_step=step)
        finally:
            del os.environ["RT_HIP_NO_TILES"]
        want = osc.render(gen, spp, 40.0, 4, seed=31, cell_begin=begin, cell_count=count, cell_step=step, threads=8)
        for a, b, c in zip(got[:4], plain[:4], want[:4]):
            assert np.array_equal(a, b) and np.array_equal(a, c)
        assert got[4] == plain[4] == want[4]


@pytest.mark.parametrize("spp", [13, 64])
def test_work_unit_length_changes_nothing(engine, spp):
    """The host picks the length of a work unit (samples of one cell traced by one lane) by sample count
    (rt_capi.cpp render_unit_samples); sums are integers and ids go by cell and sample, so any length gives the oracle's frame.
    RT_HIP_UNIT_SAMPLES overrides the choice: lengths that divide spp, that do not (ragged last blocks: the group waits), 1 and
    more than spp."""
    objs = random_scene(300, 1, seed=spp, extent=(30.0, 18.0, 30.0))
    gen = projector((0.0, 1.0, -70.0), (-32.0, -19.0, -30.0), (64.0, 0.0, 0.0), (0.0, 38.0, 0.0), 60, 44, 1.0)
    st = SimulationSettingsConfig(40.0, 5)
    sc = engine.Scene(objs)
    want = ob.OracleScene(objs).render(gen, spp, 40.0, 5, seed=3, threads=8)
    for unit in (None, 1, 3, 8, 16, 100):
        if unit is None:
            os.environ.pop("RT_HIP_UNIT_SAMPLES", None)
        else:
            os.environ["RT_HIP_UNIT_SAMPLES"] = str(unit)
        try:
            got = sc.render(gen, spp, st, seed=3)
        finally:
            os.environ.pop("RT_HIP_UNIT_SAMPLES", None)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b), unit
        assert got[4] == want[4], unit


# ---- the explicit-ray kernel (rt_paths.hip): coherent batches take the bundle cull (clusters, short lists), the rest walk the grid ----

def _fan_rays(origin, corner, u, v, nu, nv):
    """Rays from one origin to an nu x nv grid of targets, column by column (the order scene_config.rs:98-105 produces)."""
    i, j = np.meshgrid(np.arange(nu, dtype=np.float32), np.arange(nv, dtype=np.float32), indexing="ij")
    t = np.asarray(corner, np.float32) + i[..., None] * (np.asarray(u, np.float32) / np.float32(nu)) + j[..., None] * (np.asarray(v, np.float32) / np.float32(nv))
    d = (t - np.asarray(origin, np.float32)).reshape(-1, 3).astype(np.float32)
    return make_rays(np.tile(np.asarray(origin, np.float32), (len(d), 1)), d)


def _check_simulate(engine, objs, rays, inf_dist, max_b, what, monkeypatch=None):
    (pts, npts, recs), (opts, onpts, orecs) = _simulate_both(engine, objs, rays, inf_dist, max_b)
    assert np.array_equal(npts, onpts), what
    assert_records_equal(recs, orecs, what)
    assert same(pts, opts), what
    if monkeypatch is not None:  # the general kernel (trace_kernel) gives the same bytes
        monkeypatch.setenv("RT_HIP_NO_PATHS", "1")
        pts2, npts2, recs2 = engine.Scene(objs).simulate_rays_raw(SimulationSettingsConfig(inf_dist, max_b), rays=rays, seed=7)
        monkeypatch.delenv("RT_HIP_NO_PATHS")
        assert np.array_equal(npts2, npts) and same(pts2, pts), what
        assert_records_equal(recs2, recs, what)
    return npts


def test_explicit_ray_kernel_bundles_against_the_oracle(engine, monkeypatch):
    """Thin fans and parallel sheets through the C2 cloud (grid, clusters and bundle lists in LDS), column ends inside a batch
    (incoherent batches walk the grid), a fan along a row of 60 spheres (list overflow), rays that start inside spheres."""
    cfg = synth.config_c2(1920, 1080)
    objs = cfg["objects"]
    g = cfg["gen"]
    # a window of the C2 projector, 40 columns of 100 cells: batches of 64 straddle column ends
    fan = _fan_rays(g.a[:], (np.array(g.corner[:]) + 0.45 * np.array(g.u[:]) + 0.4 * np.array(g.v[:])), np.array(g.u[:]) * 0.05, np.array(g.v[:]) * 0.2, 40, 100)
    n = _check_simulate(engine, objs, fan, cfg["infinity_distance"], cfg["max_bounces"], "C2 fan", monkeypatch)
    assert (n > 2).sum() > 200  # the window looks at spheres
    # a parallel sheet (ParallelGrid-like), rows of 64
    oi, oj = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(48, dtype=np.float32), indexing="ij")
    o = np.stack([-30 + 0.9 * oi.ravel(), -15 + 0.6 * oj.ravel(), np.full(oi.size, -80.0, np.float32)], 1).astype(np.float32)
    sheet = make_rays(o, np.tile(np.array([[0.05, 0.02, 1.0]], np.float32), (len(o), 1)))
    _check_simulate(engine, objs, sheet, cfg["infinity_distance"], cfg["max_bounces"], "C2 sheet")
    # a row of 60 touching spheres seen end-on by a thin fan: more entries than a bundle list holds (the batch walks the grid),
    # beside 200 random spheres so that the grid and the clusters exist
    row = [sphere((0.0, 0.0, 2.0 * k), 1.0, (abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR)[k % 3], 1.4 if k % 3 == 0 else 0.5) for k in range(60)]
    objs2 = np.array(row + list(random_scene(200, 0, seed=9, extent=(30.0, 15.0, 60.0))), dtype=abi.OBJECT_DTYPE)
    fan2 = _fan_rays((0.1, 0.2, -40.0), (-1.0, -1.0, 0.0), (2.0, 0, 0), (0, 2.0, 0), 16, 64)
    _check_simulate(engine, objs2, fan2, 80.0, 10, "row of spheres", monkeypatch)
    # origins inside a sphere of the cloud, directions in a narrow cone
    inside = make_rays(np.tile(np.array([[0.0, 0.0, 0.3]], np.float32), (640, 1)),
                       np.array([0, 0, 1], np.float32) + 0.02 * np.random.default_rng(4).standard_normal((640, 3)).astype(np.float32))
    _check_simulate(engine, objs2, inside, 80.0, 10, "inside a sphere")


def test_explicit_ray_kernel_boxes_planes_and_small_tables(engine, monkeypatch):
    """Bundle lists with box entries (confirmed by the slab test), planes beside the scan path, and tables too small for a grid."""
    from raytracing_amd.engine import shape_object

    rng = np.random.default_rng(31)
    objs = list(random_scene(400, 2, seed=17, extent=(25.0, 12.0, 25.0)))
    for k in range(150):
        pos = (rng.uniform(-1, 1, 3) * np.array([25.0, 12.0, 25.0])).astype(np.float32)
        objs.insert(int(rng.integers(0, len(objs))), shape_object(abi.RT_SHAPE_BOX, tuple(rng.uniform(0.5, 3.0, 3)), pos, (abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR)[k % 2], 1.5))
    objs = np.array(objs, dtype=abi.OBJECT_DTYPE)
    fan = _fan_rays((0.0, 1.0, -70.0), (-20.0, -10.0, -30.0), (40.0, 0, 0), (0, 20.0, 0), 48, 64)
    _check_simulate(engine, objs, fan, 90.0, 8, "boxes and planes", monkeypatch)
    small = np.array(list(random_scene(40, 1, seed=5)), dtype=abi.OBJECT_DTYPE)  # no grid: every step scans the LDS table
    _check_simulate(engine, small, _fan_rays((0, 0, -40.0), (-15.0, -8.0, -10.0), (30.0, 0, 0), (0, 16.0, 0), 32, 64), 60.0, 8, "small table", monkeypatch)
    _check_simulate(engine, small, random_rays(3000, seed=8), 60.0, 8, "small table, random rays")
    # tables beyond an LDS tile: every ray walks the grid in global memory (GM = 2)
    large = np.array(list(random_scene(1800, 2, seed=41, extent=(40.0, 20.0, 40.0))), dtype=abi.OBJECT_DTYPE)
    _check_simulate(engine, large, _fan_rays((0.0, 2.0, -110.0), (-30.0, -15.0, -45.0), (60.0, 0, 0), (0, 30.0, 0), 40, 64), 120.0, 8, "large table, fan", monkeypatch)
    _check_simulate(engine, large, random_rays(3000, seed=12, extent=(45.0, 22.0, 45.0)), 120.0, 8, "large table, random rays")


@pytest.mark.parametrize("shape", ["sheet", "line", "lattice", "sheet_large"])
def test_thin_and_flat_grids_against_the_oracle(engine, monkeypatch, shape):
    """Grids one cell thick (sphere centres in a plane), one cell wide in two axes (centres on a line) and a lattice whose spheres
    sit on cell boundaries: the shell of "outside" records around the cells (rt_grid.h) is then most of the array.  Random rays,
    axis-parallel rays (no boundary crossings on two axes), rays inside the sheet's plane and rays that touch the grid's faces;
    ray lists on both kernels and a Projector render, LDS-resident and global-memory grids."""
    rng = np.random.default_rng({"sheet": 1, "line": 2, "lattice": 3, "sheet_large": 4}[shape])
    kinds = (abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR)
    if shape in ("sheet", "sheet_large"):
        side = 20 if shape == "sheet" else 40
        ctr = np.array([(2.5 * i + rng.uniform(-0.4, 0.4), 0.0, 2.5 * j + rng.uniform(-0.4, 0.4)) for i in range(side) for j in range(side)])
        ctr -= ctr.mean(0)
        rad = rng.uniform(0.5, 1.0, len(ctr))
    elif shape == "line":
        ctr = np.array([(0.0, 0.0, 2.2 * k) for k in range(300)]) - np.array([0.0, 0.0, 330.0])
        rad = rng.uniform(0.6, 1.0, len(ctr))
    else:
        ctr = np.array([(3.0 * i, 3.0 * j, 3.0 * k) for i in range(8) for j in range(8) for k in range(8)], dtype=np.float64) - 10.5
        rad = np.full(len(ctr), 1.2)
    objs = np.array([sphere(ctr[k], rad[k], kinds[k % 3], (1.5, 0.0, 0.5)[k % 3]) for k in range(len(ctr))], dtype=abi.OBJECT_DTYPE)
    ext = np.abs(ctr).max(0) + 3.0
    n = 1500
    o = (rng.uniform(-1.3, 1.3, (n, 3)) * ext).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    axis = np.zeros((600, 3), np.float32)  # axis-parallel rays, both senses, from outside and from between the spheres
    axis[np.arange(600), np.arange(600) % 3] = np.where(np.arange(600) % 2, 1.0, -1.0)
    ao = (rng.uniform(-1.1, 1.1, (600, 3)) * ext).astype(np.float32)
    inplane = rng.standard_normal((400, 3)).astype(np.float32)  # in the plane y = 0 of the sheet / through the line's axis
    inplane[:, 1] = 0.0
    io = (rng.uniform(-1.2, 1.2, (400, 3)) * ext).astype(np.float32)
    io[:, 1] = 0.0
    lo, hi = (ctr - rad[:, None]).min(0), (ctr + rad[:, None]).max(0)  # rays along the faces of the spheres' bounding box
    fo = np.array([(lo[0] if k % 2 else hi[0], rng.uniform(lo[1], hi[1]), lo[2] - 5.0) for k in range(100)], np.float32)
    fd = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (100, 1))
    rays = make_rays(np.concatenate([o, ao, io, fo]), np.concatenate([d, axis, inplane, fd]))
    inf_dist = float(4.0 * np.linalg.norm(ext))
    npts = _check_simulate(engine, objs, rays, inf_dist, 8, shape, monkeypatch)
    assert (npts > 2).sum() > 150, shape
    # the Projector kernel's continuing rays on the same grid
    st = SimulationSettingsConfig(inf_dist, 6)
    eye = np.array([0.3 * ext[0], 2.5 * ext[1] + 20.0, -1.6 * ext[2]])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, [0.0, 1.0, 0.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    dist = np.linalg.norm(eye)
    w_, h_ = 0.9 * dist, 0.6 * dist
    gen = projector(eye, eye + fwd * dist - 0.5 * w_ * right - 0.5 * h_ * up, w_ * right, h_ * up, 48, 40)
    got = engine.Scene(objs).render(gen, 3, st, seed=11)
    want = ob.OracleScene(objs).render(gen, 3, st.infinity_distance, st.max_bounces, seed=11, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert same(a, b), shape
    assert got[4] == want[4] and (got[1] >= 0).mean() > (0.002 if shape == "line" else 0.1), shape


@pytest.mark.parametrize("how", ["fat", "forced", "forced_odd"])
def test_grids_that_do_not_fit_into_lds_are_walked_in_memory(engine, monkeypatch, capfd, how):
    """A table of the LDS size class whose grid does not fit into LDS beside it (item lists of fat, overlapping spheres; or any scene with
    RT_HIP_GRID_IN_MEMORY=1) keeps the grid in global memory: the Projector render runs on the streamed kernel (whose scans read whole
    64-sphere rows: the sphere count of "forced_odd" is not a multiple of 64 or 4) and ray lists on paths_kernel<2>."""
    rng = np.random.default_rng(77)
    kinds = (abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR, abi.RT_MAT_HALF_MIRROR)
    if how == "fat":
        n = 1000  # 10 x 10 x 10 lattice, spacing 3, radii up to 3: a sphere is registered in a dozen cells on average
        ctr = np.array([(3.0 * i, 3.0 * j, 3.0 * k) for i in range(10) for j in range(10) for k in range(10)], np.float64) - 13.5 + rng.uniform(-0.3, 0.3, (1000, 3))
        rad = rng.uniform(0.3, 3.0, n)
        objs = np.array([sphere(ctr[k], rad[k], kinds[k % 3], (1.5, 0.0, 0.5)[k % 3]) for k in range(n)], dtype=abi.OBJECT_DTYPE)
    else:
        monkeypatch.setenv("RT_HIP_GRID_IN_MEMORY", "1")
        objs = np.array(list(random_scene(401 if how == "forced_odd" else 640, 2, seed=23, extent=(25.0, 12.0, 25.0))), dtype=abi.OBJECT_DTYPE)
    st = SimulationSettingsConfig(150.0, 7)
    gen = projector((1.0, 3.0, -70.0), (-24.0, -14.0, -20.0), (50.0, 0.0, 0.0), (0.0, 32.0, 0.0), 56, 40)
    monkeypatch.setenv("RT_HIP_TRACE_DISPATCH", "1")
    capfd.readouterr()
    got = engine.Scene(objs).render(gen, 5, st, seed=3)
    assert "pinhole_kernel grid_mode 2" in capfd.readouterr().err, how
    want = ob.OracleScene(objs).render(gen, 5, st.infinity_distance, st.max_bounces, seed=3, threads=8)
    for a, b in zip(got[:4], want[:4]):
        assert same(a, b), how
    assert got[4] == want[4] and (got[1] >= 0).mean() > 0.3, how
    fan = _fan_rays((0.0, 1.0, -70.0), (-20.0, -10.0, -30.0), (40.0, 0, 0), (0, 20.0, 0), 40, 64)
    npts = _check_simulate(engine, objs, fan, 150.0, 7, how, monkeypatch)
    err = capfd.readouterr().err
    assert "paths_kernel grid_mode 2" in err and "trace_kernel" in err, how  # (the second call of the check: the general kernel)
    assert (npts > 2).sum() > 500, how
    _check_simulate(engine, objs, random_rays(3000, seed=5, extent=(30.0, 15.0, 30.0)), 150.0, 7, how + ", random rays")


def test_generator_renders_on_the_explicit_ray_kernel(engine, monkeypatch):
    """ParallelGrid renders (and Projector renders the Projector kernel declines: RT_HIP_NO_PINHOLE) run on paths_kernel<GM, RENDER = true>:
    framebuffer, first-hit ids / t, per-cell segment counts and the segment total against the oracle, and the same bytes from the general
    kernel (RT_HIP_NO_PATHS); whole grids, strided and offset cell ranges, ragged sample counts, planes, boxes, small and grid-sized tables."""
    from raytracing_amd.engine import shape_object

    rng = np.random.default_rng(77)
    big = list(random_scene(500, 2, seed=23, extent=(25.0, 12.0, 25.0)))
    for k in range(60):
        pos = (rng.uniform(-1, 1, 3) * np.array([25.0, 12.0, 25.0])).astype(np.float32)
        big.insert(int(rng.integers(0, len(big))), shape_object(abi.RT_SHAPE_BOX, tuple(rng.uniform(0.5, 3.0, 3)), pos, (abi.RT_MAT_GLASS, abi.RT_MAT_MIRROR)[k % 2], 1.5))
    big = np.array(big, dtype=abi.OBJECT_DTYPE)
    small = np.array(list(random_scene(30, 1, seed=6)), dtype=abi.OBJECT_DTYPE)
    pg = parallel_grid((-26.0, -13.0, -60.0), (52.0, 0, 0), (0, 26.0, 0), 96, 50, (0.03, -0.02, 1.0))
    pj = projector((0.0, 1.0, -70.0), (-20.0, -10.0, -30.0), (40.0, 0, 0), (0, 20.0, 0), 80, 45)
    st = SimulationSettingsConfig(70.0, 7)
    large = np.array(list(random_scene(1500, 1, seed=43, extent=(25.0, 12.0, 25.0), r_min=0.3, r_max=1.0)), dtype=abi.OBJECT_DTYPE)  # grid in global memory
    cases = [(big, pg, 5, 0, None, 1), (big, pg, 70, 11, 400, 3), (big, pg, 1, 96 * 50 - 1, 1, 1), (small, pg, 3, 0, None, 1), (big, pj, 4, 0, None, 1),
             (large, pg, 3, 0, None, 1)]
    for objs, gen, spp, begin, count, step in cases:
        if gen is pj:
            monkeypatch.setenv("RT_HIP_NO_PINHOLE", "1")
        sc = engine.Scene(objs)
        got = sc.render(gen, spp, st, seed=9, cell_begin=begin, cell_count=count, cell_step=step)
        want = ob.OracleScene(objs).render(gen, spp, 70.0, 7, seed=9, cell_begin=begin, cell_count=count, cell_step=step)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b)
        assert got[4] == want[4]
        monkeypatch.setenv("RT_HIP_NO_PATHS", "1")
        other = sc.render(gen, spp, st, seed=9, cell_begin=begin, cell_count=count, cell_step=step)
        monkeypatch.delenv("RT_HIP_NO_PATHS")
        for a, b in zip(got[:4], other[:4]):
            assert np.array_equal(a, b)
        assert got[4] == other[4]
        monkeypatch.delenv("RT_HIP_NO_PINHOLE", raising=False)
    assert (got[1] >= 0).sum() > 100  # the views look at the scene
