"""The native `raytracing` binary (raytracing_amd/host): its TOML reader + SceneConfig flattening must agree bit for bit with
the Python mirror (CPU, --dump), and its CSV output with the golden polylines (GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracing_amd import abi
from raytracing_amd.config import SimulationConfig

sys.path.insert(0, os.path.dirname(__file__))
from helpers import TEMPLATE_LIKE_TOML  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "raytracing_amd", "bin", "raytracing")
GOLD = os.path.join(ROOT, "tests", "golden")


def _dump(toml_text, tmp_path):
    p = tmp_path / "simulation.toml"
    p.write_text(toml_text, encoding="utf-8")
    r = subprocess.run([BIN, "--dump"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    return r


def _parse_dump(out):
    objs, nodes, rays, settings = [], [], [], None
    for line in out.splitlines():
        f = line.split()
        if not f:
            continue
        if f[0] == "settings":
            settings = (np.float32(float.fromhex(f[1])), int(f[2]))
        elif f[0] == "object":
            objs.append(([int(f[1]), int(f[2])], [np.float32(float.fromhex(x)) for x in f[3:12]], int(f[12])))
        elif f[0] == "node":
            nodes.append(([int(x) for x in f[1:4]], [np.float32(float.fromhex(x)) for x in f[4:8]]))
        elif f[0] == "ray":
            rays.append([np.float32(float.fromhex(x)) for x in f[1:8]])
    return settings, objs, nodes, rays


@pytest.mark.parametrize("which", ["c1", "allshapes"])
def test_native_flatten_equals_python_mirror(tmp_path, which):
    text = open(os.path.join(GOLD, "c1_scene.toml"), encoding="utf-8").read() if which == "c1" else TEMPLATE_LIKE_TOML
    r = _dump(text, tmp_path)
    assert r.returncode == 0, r.stderr
    settings, objs, nodes, rays = _parse_dump(r.stdout)
    cfg = SimulationConfig.from_toml_str(text)
    pobjs, pnodes = cfg.scene.object_and_node_lists()
    prays = cfg.scene.ray_list()
    assert settings == (np.float32(cfg.simulation_settings.infinity_distance), cfg.simulation_settings.max_bounces)
    assert len(objs) == len(pobjs) and len(nodes) == len(pnodes) and len(rays) == len(prays)
    for (ints, floats, sp0), po in zip(objs, pobjs):
        assert ints == [int(po["shape"]), int(po["material_kind"])]
        want = [po["material_param"], po["rotation_y_deg"], *po["position"], *po["shape_param"]]
        if po["shape"] == abi.RT_SHAPE_TREE:  # shape_param[0] carries the root index as raw bits
            assert sp0 == int(po["shape_param"].view(np.uint32)[0])
            floats, want = floats[:5] + floats[6:], want[:5] + want[6:]
        assert all(np.float32(a) == np.float32(b) for a, b in zip(floats, want)), (floats, want)
    for (ints, floats), pn in zip(nodes, pnodes):
        assert ints == [int(pn["shape"]), int(pn["a"]), int(pn["b"])] and all(a == b for a, b in zip(floats, pn["shape_param"]))
    got = np.array(rays, np.float32)
    want = np.column_stack([prays["origin"], prays["direction"], prays["current_ior"]])
    assert np.array_equal(got, want)


def test_native_config_errors(tmp_path):
    """Both top-level tables are required (simulation_config.rs:8-11); errors end the process with a message, like `?` in cli.rs:12."""
    r = _dump("[simulation_settings]\ninfinity_distance = 1.0\nmax_bounces = 2\n[[rays]]\norigin=[0,0,0]\ndirection=[1,0,0]\n", tmp_path)
    assert r.returncode == 1 and "scene" in r.stderr
    r = _dump("[simulation_settings\n", tmp_path)
    assert r.returncode == 1 and "TOML parse error" in r.stderr
    r = subprocess.run([BIN, "--dump"], cwd=tmp_path / "..", capture_output=True, text=True, timeout=60) if False else None
    (tmp_path / "simulation.toml").unlink()
    r = subprocess.run([BIN, "--dump"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "No such file" in r.stderr


@pytest.mark.gpu
def test_native_cli_end_to_end(tmp_path):
    """`raytracing` with no arguments: ./simulation.toml -> ./dist/path_{i}.csv, equal to the golden polylines."""
    import shutil

    shutil.copy(os.path.join(GOLD, "c1_scene.toml"), tmp_path / "simulation.toml")
    r = subprocess.run([BIN], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "dist" in r.stderr  # the reference does not create ./dist either (cli.rs:18-19)
    os.mkdir(tmp_path / "dist")
    r = subprocess.run([BIN], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "光路 2 を './dist/path_2.csv' に出力しました。" in r.stdout
    g = np.load(os.path.join(GOLD, "c1_paths.npz"))
    from raytracing_amd.cli import format_f32

    for i in range(3):
        text = (tmp_path / "dist" / f"path_{i}.csv").read_text()
        want = "x,y,z\n" + "".join(",".join(format_f32(c) for c in p) + "\n" for p in g["points"][i, : g["n_points"][i]])
        assert text == want  # same bytes as the Python writer, i.e. Rust's f32::to_string() digits


def test_native_csv_number_format_matches_rust_display():
    """cli.rs:21-27 writes `f32::to_string()`: shortest round-trip digits, always positional.  The native writer and the
    Python mirror must print the same text at every magnitude (1234567936f32 is "1234568000", never "1234567936")."""
    from raytracing_amd.cli import format_f32

    rng = np.random.default_rng(7)
    vals = [0.0, -0.0, 1.0, -1.5, 0.001, 1e-7, 1.17549435e-38, 1e-45, 16777216.0, 16777218.0, 1234567936.0, -987654321.0, 1e9, 3.0e10,
            1e30, -1e30, 3.4028235e38, 24.163, -4.948, 0.1, 123456.789, float("inf"), float("-inf"), float("nan")]
    vals += list(rng.normal(size=200) * 10.0 ** rng.integers(-12, 38, 200))
    bits = np.array(vals, np.float32).view(np.uint32)
    bits = np.concatenate([bits, rng.integers(0, 2**32, 500, dtype=np.uint64).astype(np.uint32)])
    r = subprocess.run([BIN, "--format-f32"] + [f"{b:08x}" for b in bits], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = r.stdout.split("\n")[:-1]
    assert len(got) == len(bits)
    for b, text in zip(bits, got):
        x = np.array([b], np.uint32).view(np.float32)[0]
        assert text == format_f32(x), (hex(int(b)), text, format_f32(x))
    assert format_f32(np.float32(1234567936.0)) == "1234568000" and format_f32(np.float32(1e-7)) == "0.0000001"
    assert format_f32(np.float32(-0.0)) == "-0" and format_f32(np.float32(1e30)) == "1000000000000000000000000000000"
