"""Regenerates the golden vectors in this directory from the CPU oracle (run here, in the build container).

The reference (Rust) cannot be built in this image, so these vectors freeze the oracle — which itself is pinned
only by python_files/tmp.py:4-9 of the reference (see oracle/oracle.cpp header).  Usage:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import binding as ob  # noqa: E402
from raytracing_amd import synth  # noqa: E402
from raytracing_amd.config import SimulationConfig  # noqa: E402

from helpers import TEMPLATE_LIKE_TOML, random_rays, random_scene  # noqa: E402


def main():
    # C1: the reference's own scene
    cfg = SimulationConfig.load_from_path(os.path.join(HERE, "c1_scene.toml"))
    objs, rays = cfg.scene.object_list(), cfg.scene.ray_list()
    st = cfg.simulation_settings
    pts, npts, recs, tot = ob.OracleScene(objs).simulate(rays, st.infinity_distance, st.max_bounces)
    np.savez(os.path.join(HERE, "c1_paths.npz"), objects=objs, rays=rays, points=pts, n_points=npts, records=recs, total=np.int64(tot))

    # seeded synthetic scene with every material and planes mixed into the list
    objs = random_scene(200, 2, seed=2024)
    rays = random_rays(512, seed=77)
    pts, npts, recs, tot = ob.OracleScene(objs).simulate(rays, 40.0, 8, seed=5, ray_id_base=1000)
    np.savez(os.path.join(HERE, "synth200_paths.npz"), objects=objs, rays=rays, points=pts, n_points=npts, records=recs,
             total=np.int64(tot))

    # reduced C2 render (48x27 cells, 4 spp)
    c2 = synth.config_c2(48, 27)
    fb, fid, ft, seg, tot = ob.OracleScene(c2["objects"]).render(c2["gen"], 4, c2["infinity_distance"], c2["max_bounces"], seed=c2["seed"])
    np.savez(os.path.join(HERE, "c2_48x27_4spp.npz"), fb=fb, first_id=fid, first_t=ft, seg_count=seg, total=np.int64(tot))
    # every ShapeConfig arm (template.toml's object set + Cone + CSG trees + rotated Sphere / Plane)
    cfg = SimulationConfig.from_toml_str(TEMPLATE_LIKE_TOML)
    objs, nodes = cfg.scene.object_and_node_lists()
    rays = cfg.scene.ray_list()
    st = cfg.simulation_settings
    pts, npts, recs, tot = ob.OracleScene(objs, nodes).simulate(rays, st.infinity_distance, st.max_bounces, seed=3)
    np.savez_compressed(os.path.join(HERE, "allshapes_paths.npz"), objects=objs, nodes=nodes, rays=rays, points=pts, n_points=npts,
                        records=recs, total=np.int64(tot))
    print("golden vectors written")


if __name__ == "__main__":
    main()
