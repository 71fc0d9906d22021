"""Manual stress (not collected by pytest: run `python -m pytest tests/stress_mesh_fuzz.py -m gpu -q`): random triangle soups,
degenerate slivers, huge and tiny coordinates, rays from inside / far outside / axis-aligned — BVH traversal on the GPU
against the oracle's brute-force loop, records bit for bit."""
import os
import sys

import numpy as np
import pytest

from oracle import binding as ob
from raytracing_amd import abi
from raytracing_amd.engine import SimulationSettingsConfig, make_rays, mesh_object

sys.path.insert(0, os.path.dirname(__file__))
from helpers import assert_records_equal, same  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_STRESS_SEEDS", "40"))))
def test_mesh_fuzz(engine, seed):
    rng = np.random.default_rng(seed)
    scale = 10.0 ** rng.uniform(-2, 3)
    n_tris = int(rng.integers(1, 6000))
    c = rng.uniform(-1, 1, size=(n_tris, 1, 3)) * scale
    size = scale * 10.0 ** rng.uniform(-2.5, -0.3)
    v = c + rng.normal(size=(n_tris, 3, 3)) * size
    if seed % 3 == 0:  # slivers: third vertex almost on the first edge
        v[::5, 2] = v[::5, 0] + (v[::5, 1] - v[::5, 0]) * rng.random((len(v[::5]), 1)) + rng.normal(size=(len(v[::5]), 3)) * size * 1e-6
    if seed % 4 == 0:  # an axis-aligned sheet: boxes of zero thickness
        v[1::7, :, 1] = v[1::7, :1, 1]
    offset = rng.normal(size=3) * scale * (100.0 if seed % 5 == 0 else 0.1)
    v = (v + offset).astype(np.float32).reshape(-1, 3)
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    objs = np.array([mesh_object(0, tuple(rng.normal(size=3) * scale * 0.1), [abi.RT_MAT_MIRROR, abi.RT_MAT_GLASS, abi.RT_MAT_HALF_MIRROR][seed % 3], 1.3,
                                 rotation_y_deg=float(rng.uniform(-180, 180)) if seed % 2 else 0.0)], dtype=abi.OBJECT_DTYPE)
    n = 6000
    o = offset + rng.uniform(-1.5, 1.5, size=(n, 3)) * scale
    d = rng.normal(size=(n, 3))
    d[::9] = np.eye(3)[rng.integers(0, 3, len(d[::9]))] * rng.choice([-1.0, 1.0], (len(d[::9]), 1))  # axis-aligned rays (zeros in d)
    tgt = v[rng.integers(0, len(v), n)].astype(np.float64)
    aim = rng.random(n) < 0.5
    d[aim] = tgt[aim] - o[aim]  # half of the rays aim at a vertex: edge and corner hits
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = make_rays(o.astype(np.float32), d.astype(np.float32))
    st = SimulationSettingsConfig(float(scale * 5.0), 5)
    pts, npts, recs = engine.Scene(objs, meshes=[(v, idx)]).simulate_rays_raw(st, rays=rays, seed=seed)
    opts, onpts, orecs, _ = ob.OracleScene(objs, meshes=[(v, idx)]).simulate(rays, float(scale * 5.0), 5, seed=seed)
    assert np.array_equal(npts, onpts)
    assert_records_equal(recs, orecs, f"mesh fuzz {seed}")
    assert same(pts, opts)
