"""`raytracing` CLI on the HIP path — mirror of crates/raytracing_cli/src/cli.rs:7-33.

Same invocation (no arguments): reads ./simulation.toml, simulates, writes ./dist/path_{i}.csv (header x,y,z, one row per
polyline vertex, f32 printed like Rust's `f32::to_string()`), prints the same progress lines.  The Bevy window
(cli.rs:15) is not part of the hot path and is omitted.  Like the reference, ./dist must already exist.

    python -m raytracing_amd.cli            # in a directory holding simulation.toml and dist/
"""
import os
import sys

import numpy as np


def format_f32(x):
    """Rust `f32::to_string()`: shortest digits that round-trip, positional (never exponent), no trailing '.0'."""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def write_path_csv(path, points):
    """cli.rs:19-28 with the csv crate's defaults (',' delimiter, '\\n' terminator, no quoting needed)."""
    with open(path, "w", newline="") as f:
        f.write("x,y,z\n")
        for p in points:
            f.write(f"{format_f32(p[0])},{format_f32(p[1])},{format_f32(p[2])}\n")


def cli(config_path="simulation.toml", out_dir="./dist", device=0, seed=0):
    from . import engine
    from .config import SimulationConfig

    print("設定ファイル simulation.toml を読み込んでいます...")
    cfg = SimulationConfig.load_from_path(config_path)
    engine.init(device)
    scene = cfg.scene.into_scene()
    results = scene.simulate_rays(cfg.simulation_settings, seed=seed)
    for i, result in enumerate(results):
        file_name = f"{out_dir}/path_{i}.csv"
        write_path_csv(file_name, result)  # raises if ./dist is missing, like csv::Writer::from_path (cli.rs:19)
        print(f"光路 {i} を '{file_name}' に出力しました。")
    return 0


if __name__ == "__main__":
    sys.exit(cli())
