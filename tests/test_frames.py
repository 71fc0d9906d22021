"""Image files and the multi-frame driver (extension, BASELINE.json configs[4]; no reference counterpart): PFM layout,
frame dealing across ranks, frame files written from an update callback, and — on the GPU — the native
`raytracing --render --frames` and the Python driver against the oracle's render of every frame."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracing_amd import frames, image
from raytracing_amd.config import SimulationConfig

sys.path.insert(0, os.path.dirname(__file__))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "raytracing_amd", "bin", "raytracing")

BASE = """
[simulation_settings]
infinity_distance = 40.0
max_bounces = 5

[scene]
[[scene.ray_generators]]
type = "Projector"
origin = [0.0, 2.0, -30.0]
target_corner = [-8.0, -4.0, -15.0]
target_u = [16.0, 0.0, 0.0]
target_v = [0.0, 9.0, 0.0]
count_u = 32
count_v = 18
current_ior = 1.0

[[scene.objects]]
transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }
shape = { type = "Sphere", radius = 3.0 }
material = { type = "Glass", ior = 1.5 }

[[scene.objects]]
transform = { position = [5.0, 1.0, 2.0], rotation_y_deg = 0.0 }
shape = { type = "Sphere", radius = 2.0 }
material = { type = "HalfMirror", reflectance = 0.5 }

[[scene.objects]]
transform = { position = [-4.0, 0.0, 3.0], rotation_y_deg = 10.0 }
shape = { type = "Box", size = [3.0, 3.0, 3.0] }
material = { type = "Mirror" }

[[scene.objects]]
transform = { position = [0.0, -4.0, 0.0], rotation_y_deg = 0.0 }
shape = { type = "Plane", normal = [0.0, 1.0, 0.0] }
material = { type = "Mirror" }
"""


def _update(f, doc):
    """per-frame scene update: the glass sphere orbits, the box spins"""
    a = 0.5 * f
    doc["scene"]["objects"][0]["transform"]["position"] = [4.0 * np.sin(a).item(), 0.0, 4.0 * np.cos(a).item()]
    doc["scene"]["objects"][2]["transform"]["rotation_y_deg"] = 10.0 + 25.0 * f


def test_pfm_layout_and_round_trip(tmp_path):
    cu, cv = 5, 3
    fb = np.arange(cu * cv * 3, dtype=np.float32).reshape(cu * cv, 3) * np.float32(0.1)
    p = tmp_path / "a.pfm"
    image.write_pfm(p, fb, cu, cv)
    raw = p.read_bytes()
    head = b"PF\n5 3\n-1.0\n"
    assert raw.startswith(head) and len(raw) == len(head) + cu * cv * 12
    body = np.frombuffer(raw[len(head):], "<f4").reshape(cv, cu, 3)
    for i in range(cu):
        for j in range(cv):
            assert np.array_equal(body[j, i], fb[i * cv + j])  # column = i, row from the bottom = j (cell = i*count_v + j)
    back, bu, bv = image.read_pfm(p)
    assert (bu, bv) == (cu, cv) and np.array_equal(back, fb)
    image.write_ppm(tmp_path / "a.ppm", np.clip(fb / 4.5, 0, 1), cu, cv)
    ppm = (tmp_path / "a.ppm").read_bytes()
    assert ppm.startswith(b"P6\n5 3\n255\n") and len(ppm) == 11 + cu * cv * 3
    assert ppm[11 + 0] == 0 or True  # top-left pixel is cell (i=0, j=cv-1)
    top_left = np.clip(fb[0 * cv + (cv - 1)] / 4.5, 0, 1) ** (1 / 2.2)
    assert list(ppm[11:14]) == np.rint(top_left * 255).astype(int).tolist()


def test_frames_are_dealt_round_robin():
    for n in (0, 1, 7, 16):
        for world in (1, 2, 3, 8):
            parts = [frames.frames_of_rank(n, r, world) for r in range(world)]
            assert sorted(f for p in parts for f in p) == list(range(n))
            assert all(f % world == r for r, p in enumerate(parts) for f in p)
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        frames.frames_of_rank(4, 2, 2)


def test_frame_files_from_update_parse_back(tmp_path):
    frames.frames_from_update(BASE, 3, _update, str(tmp_path))
    files = frames.frame_files(str(tmp_path))
    assert files == ["frame_00000.toml", "frame_00001.toml", "frame_00002.toml"]
    cfgs = [SimulationConfig.load_from_path(tmp_path / f) for f in files]
    objs = [c.scene.object_list() for c in cfgs]
    assert objs[0]["position"][0].tolist() == [0.0, 0.0, 4.0]
    assert objs[2]["rotation_y_deg"][2] == np.float32(60.0) and objs[1]["position"][0][0] == np.float32(4.0 * np.sin(0.5))
    assert all(len(c.scene.generators()) == 1 and c.simulation_settings.max_bounces == 5 for c in cfgs)
    # the native reader accepts what the writer produced, and flattens it to the same floats
    r = subprocess.run([BIN, "--dump", "--config", str(tmp_path / files[1])], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("object")]
    got = np.array([[float.fromhex(x) for x in ln[5:8]] for ln in lines], np.float32)
    assert np.array_equal(got, objs[1]["position"])


@pytest.mark.gpu
def test_frame_sequence_native_python_oracle(tmp_path):
    """4 frames, 2 'ranks' run one after the other: native and Python drivers write identical PFM bytes, the union of the
    ranks' files is the whole sequence, and every frame equals the oracle's render of that frame's config."""
    from oracle import binding as ob
    from raytracing_amd import engine

    engine.init(0)
    fdir, out_n, out_p = tmp_path / "frames", tmp_path / "native", tmp_path / "python"
    for d in (fdir, out_n, out_p):
        os.mkdir(d)
    frames.frames_from_update(BASE, 4, _update, str(fdir))
    spp, seed = 6, 11
    for rank in range(2):
        r = subprocess.run([BIN, "--render", str(spp), "--frames", str(fdir), "--out", str(out_n), "--seed", str(seed), "--rank", str(rank),
                            "--world", "2"], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        assert f"rank {rank}/2: 2 of 4 frames" in r.stdout
        done = frames.render_frames(str(fdir), str(out_p), spp, seed=seed, rank=rank, world=2)
        assert [d[0] for d in done] == [rank, rank + 2]
    names = [f"frame_{f:05d}.pfm" for f in range(4)]
    assert sorted(os.listdir(out_n)) == names and sorted(os.listdir(out_p)) == names
    prev = None
    for f, name in enumerate(names):
        raw = (out_n / name).read_bytes()
        assert raw == (out_p / name).read_bytes()
        fb, cu, cv = image.read_pfm(out_n / name)
        cfg = SimulationConfig.load_from_path(fdir / f"frame_{f:05d}.toml")
        objs, nodes = cfg.scene.object_and_node_lists()
        st = cfg.simulation_settings
        want = ob.OracleScene(objs, nodes).render(cfg.scene.generators()[0], spp, st.infinity_distance, st.max_bounces, seed=seed, threads=4)
        assert (cu, cv) == (32, 18) and np.array_equal(fb, want[0].reshape(-1, 3))
        assert prev is None or not np.array_equal(prev, fb)  # the scene really changes from frame to frame
        prev = fb
    # single image mode
    r = subprocess.run([BIN, "--render", str(spp), "--config", str(fdir / "frame_00002.toml"), "--out", str(tmp_path), "--seed", str(seed)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "frame.pfm").read_bytes() == (out_n / names[2]).read_bytes()


def test_obj_mesh_reader(tmp_path):
    from raytracing_amd import meshio, synth

    p = tmp_path / "quad.obj"
    p.write_text("# a quad and a triangle\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 2\nf 1 2 3 4\nf -1/1/1 1/2/2 2/3/3  # relative index\n")
    v, idx = meshio.load_obj(p)
    assert v.shape == (5, 3) and v.dtype == np.float32 and idx.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1]]
    hv, hi = synth.heightfield_mesh(6, 5)
    meshio.save_obj(tmp_path / "hf.obj", hv, hi)
    bv, bi = meshio.load_obj(tmp_path / "hf.obj")
    assert np.array_equal(bv, hv) and np.array_equal(bi, hi)  # repr() of an f32 widened to f64 round-trips exactly
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ValueError):
        meshio.load_obj(tmp_path / "bad.obj")
