"""Multi-frame driver (extension, BASELINE.json configs[4]; no reference counterpart).

A sequence is a directory of `*.toml` files, each a complete, unchanged-schema `simulation.toml`
(simulation_config.rs:7-11): frame f is the f-th file in name order, and "per-frame scene update" is whatever differs
between consecutive files.  `frames_from_update` writes such a directory from a base config and a Python callback.

Frames are independent, so N GPUs deal them round-robin (frame f on rank f mod N) and exchange nothing; each rank
writes `frame_%05d.pfm` for its own frames (raytracing_amd/image.py).  The host work of the next frame (TOML parse,
flatten) runs on a worker thread while the GPU renders the current one.  `raytracing --render SPP --frames DIR` is the
native equivalent and produces the same bytes.

    python -m torch.distributed.run --nproc-per-node 8 -m raytracing_amd.frames --frames DIR --out OUT --spp 64
"""
import argparse
import os
from concurrent.futures import ThreadPoolExecutor

from . import engine
from .config import SimulationConfig
from .image import write_pfm


def frame_files(frames_dir):
    return sorted(n for n in os.listdir(frames_dir) if n.endswith(".toml"))


def frames_of_rank(n_frames, rank, world):
    """Frame indices rendered by `rank` of `world`: rank, rank + world, ..."""
    if not (world >= 1 and 0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    return list(range(rank, n_frames, world))


def _load(path, generator):
    cfg = SimulationConfig.load_from_path(path)
    gens = cfg.scene.generators()
    if generator >= len(gens):
        raise ValueError(f"{path}: the config has {len(gens)} ray generators, generator {generator} asked")
    objs, nodes = cfg.scene.object_and_node_lists()
    return cfg.simulation_settings, objs, nodes, gens[generator]


def render_frames(frames_dir, out_dir, spp, seed=0, generator=0, rank=0, world=1, writer=write_pfm):
    """Render this rank's frames; returns [(frame index, output path, ray segments)]."""
    files = frame_files(frames_dir)
    mine = frames_of_rank(len(files), rank, world)
    done = []
    with ThreadPoolExecutor(max_workers=4) as pool:
        nxt = pool.submit(_load, os.path.join(frames_dir, files[mine[0]]), generator) if mine else None
        writing = []  # up to 3 images being reordered / written while the GPU renders on
        for k, f in enumerate(mine):
            settings, objs, nodes, gen = nxt.result()
            if k + 1 < len(mine):
                nxt = pool.submit(_load, os.path.join(frames_dir, files[mine[k + 1]]), generator)
            scene = engine.Scene(objs, nodes=nodes)
            fb, _, _, _, total = scene.render(gen, spp, settings, seed=seed)
            del scene
            while len(writing) >= 3:
                writing.pop(0).result()
            path = os.path.join(out_dir, f"frame_{f:05d}.pfm")
            writing.append(pool.submit(writer, path, fb, gen.count_u, gen.count_v))
            done.append((f, path, int(total)))
        for wr in writing:
            wr.result()
    return done


def frames_from_update(base_toml_text, n_frames, update, frames_dir):
    """Write frame_%05d.toml files: `update(frame, doc)` edits the parsed TOML document (a dict) of each frame in place."""
    import copy

    from .config import _toml
    from .toml_write import dumps

    base = _toml.loads(base_toml_text)
    for f in range(n_frames):
        doc = copy.deepcopy(base)
        update(f, doc)
        with open(os.path.join(frames_dir, f"frame_{f:05d}.toml"), "w", encoding="utf-8") as w:
            w.write(dumps(doc))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--frames", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--spp", type=int, required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--generator", type=int, default=0)
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    engine.init(int(os.environ.get("LOCAL_RANK", 0)))
    done = render_frames(a.frames, a.out, a.spp, a.seed, a.generator, rank, world)
    print(f"rank {rank}/{world}: {len(done)} of {len(frame_files(a.frames))} frames, {sum(d[2] for d in done)} ray segments")


if __name__ == "__main__":
    main()
