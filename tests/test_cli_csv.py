"""CSV writer of the CLI mirror (cli.rs:17-30): Rust f32::to_string() formatting, file layout (CPU); end-to-end on GPU."""
import os

import numpy as np
import pytest

from raytracing_amd.cli import format_f32, write_path_csv

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_format_matches_rust_display():
    cases = [(5.0, "5"), (-15.0, "-15"), (0.0, "0"), (-0.0, "-0"), (0.1, "0.1"), (47.050392, "47.050392"), (1e-7, "0.0000001"),
             (1e20, "100000000000000000000"), (3.4028235e38, "340282350000000000000000000000000000000"), (1.5, "1.5"),
             (16777216.0, "16777216"), (0.33333334, "0.33333334"), (-9.38762, "-9.38762")]
    for v, want in cases:
        assert format_f32(np.float32(v)) == want, (v, format_f32(np.float32(v)))
    assert format_f32(np.float32("nan")) == "NaN" and format_f32(np.float32("inf")) == "inf" and format_f32(-np.float32("inf")) == "-inf"
    rng = np.random.default_rng(0)
    for x in rng.normal(size=2000).astype(np.float32) * np.float32(100):
        s = format_f32(x)
        assert np.float32(s) == x and "e" not in s  # round-trips, positional


def test_csv_layout(tmp_path):
    g = np.load(os.path.join(GOLD, "c1_paths.npz"))
    pts = g["points"][0, : g["n_points"][0]]
    f = tmp_path / "path_0.csv"
    write_path_csv(str(f), pts)
    lines = f.read_text().split("\n")
    assert lines[0] == "x,y,z" and lines[-1] == "" and len(lines) == len(pts) + 2
    assert lines[1] == "-15,0,0"
    back = np.array([[np.float32(c) for c in ln.split(",")] for ln in lines[1:-1]], np.float32)
    assert np.array_equal(back, pts)


@pytest.mark.gpu
def test_cli_end_to_end(engine, tmp_path, monkeypatch):
    """`raytracing` with no arguments in a directory holding simulation.toml and dist/ (cli.rs:12,18)."""
    import shutil

    from raytracing_amd import cli

    shutil.copy(os.path.join(GOLD, "c1_scene.toml"), tmp_path / "simulation.toml")
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):  # the reference does not create ./dist either
        cli.cli()
    os.mkdir(tmp_path / "dist")
    assert cli.cli() == 0
    g = np.load(os.path.join(GOLD, "c1_paths.npz"))
    for i in range(3):
        rows = (tmp_path / "dist" / f"path_{i}.csv").read_text().split("\n")[1:-1]
        back = np.array([[np.float32(c) for c in r.split(",")] for r in rows], np.float32)
        assert np.array_equal(back, g["points"][i, : g["n_points"][i]])
