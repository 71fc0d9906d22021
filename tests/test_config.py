"""Host-side mirror of crates/raytracing_config: schema, ordering, error behaviour (CPU only)."""
import os

import numpy as np
import pytest

from raytracing_amd import abi
from raytracing_amd.config import ConfigError, SimulationConfig

GOLD = os.path.join(os.path.dirname(__file__), "golden")

BASE = """
[simulation_settings]
infinity_distance = 5.5
max_bounces = 4
"""


def test_loads_reference_schema():
    cfg = SimulationConfig.load_from_path(os.path.join(GOLD, "c1_scene.toml"))
    assert cfg.simulation_settings.infinity_distance == 50.0 and cfg.simulation_settings.max_bounces == 10
    objs = cfg.scene.object_list()
    assert len(objs) == 27 and objs.dtype == abi.OBJECT_DTYPE
    assert np.all(objs["material_kind"] == abi.RT_MAT_GLASS)
    assert np.array_equal(objs["material_param"][:2], np.float32([1.2, 1.2])) and np.all(objs["material_param"][2:] == np.float32(1.5))
    assert len(cfg.scene.ray_list()) == 3 and len(cfg.scene.generators()) == 1


def test_explicit_rays_are_normalised_with_ior_one():
    cfg = SimulationConfig.from_toml_str(BASE + """
[[scene.rays]]
origin = [1.0, 2.0, 3.0]
direction = [0.0, 3.0, 4.0]
[[scene.objects]]
shape = { type = "Sphere", radius = 2.0 }
material = { type = "HalfMirror", reflectance = 0.25 }
transform = { position = [1.0, 0.0, 0.0], rotation_y_deg = 0.0 }
[[scene.objects]]
shape = { type = "Plane", normal = [0.0, 2.0, 0.0] }
material = { type = "Mirror" }
transform = { position = [0.0, -1.0, 0.0], rotation_y_deg = 0.0 }
""")
    rays = cfg.scene.ray_list()
    assert np.allclose(rays["direction"][0], [0, 0.6, 0.8]) and rays["current_ior"][0] == 1.0  # ray_config.rs:16-17
    objs = cfg.scene.object_list()
    assert objs["shape"].tolist() == [abi.RT_SHAPE_SPHERE, abi.RT_SHAPE_PLANE]
    assert objs["material_kind"].tolist() == [abi.RT_MAT_HALF_MIRROR, abi.RT_MAT_MIRROR]
    assert np.array_equal(objs["shape_param"][1][:3], np.float32([0, 2, 0]))  # normal kept un-normalised (shape_config.rs:69-73)


def test_missing_tables_fail_like_serde():
    """simulation_config.rs:8-11: neither field has #[serde(default)] — the reference's own scene.toml/template.toml
    (top-level [[rays]]/[[objects]]) therefore fail to load; so must we."""
    with pytest.raises(ConfigError, match="scene"):
        SimulationConfig.from_toml_str(BASE + "[[rays]]\norigin=[0,0,0]\ndirection=[1,0,0]\n")
    with pytest.raises(ConfigError, match="simulation_settings"):
        SimulationConfig.from_toml_str("[scene]\n")
    with pytest.raises(ConfigError, match="max_bounces"):
        SimulationConfig.from_toml_str("[simulation_settings]\ninfinity_distance = 1.0\n[scene]\n")
    with pytest.raises(ConfigError, match="unknown material"):
        SimulationConfig.from_toml_str(BASE + """
[[scene.objects]]
shape = { type = "Sphere", radius = 2.0 }
material = { type = "Wood" }
transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }
""").scene.object_list()
    with pytest.raises(ConfigError, match="TOML parse error"):
        SimulationConfig.from_toml_str("[simulation_settings\n")


def test_empty_scene_lists_default():
    cfg = SimulationConfig.from_toml_str(BASE + "[scene]\n")
    assert len(cfg.scene.object_list()) == 0 and len(cfg.scene.ray_list()) == 0


def test_every_shape_arm_maps_to_the_abi():
    """shape_config.rs:9-51: all ten ShapeConfig arms, including nested Union / Intersection / Difference trees."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import TEMPLATE_LIKE_TOML

    cfg = SimulationConfig.from_toml_str(TEMPLATE_LIKE_TOML)
    objs, nodes = cfg.scene.object_and_node_lists()
    assert objs["shape"][:10].tolist() == [abi.RT_SHAPE_BOX, abi.RT_SHAPE_CYLINDER, abi.RT_SHAPE_WEDGE, abi.RT_SHAPE_LENS, abi.RT_SHAPE_PLANE,
                                           abi.RT_SHAPE_CONE, abi.RT_SHAPE_TREE, abi.RT_SHAPE_TREE, abi.RT_SHAPE_SPHERE, abi.RT_SHAPE_PLANE]
    assert np.array_equal(objs["shape_param"][2], np.float32([10, 8, 4, 15]))  # Wedge: size + angle_deg
    assert np.array_equal(objs["shape_param"][3], np.float32([2, 18, 20, -20]))  # Lens: thickness, diameter, r1, r2
    assert objs["rotation_y_deg"][0] == np.float32(25.0) and objs["rotation_y_deg"][2] == np.float32(-90.0)
    # post-order node list: Difference(Sphere, Box) then Union(Sphere, Intersection(Cylinder, Sphere))
    assert nodes["shape"].tolist() == [abi.RT_SHAPE_SPHERE, abi.RT_SHAPE_BOX, abi.RT_SHAPE_DIFFERENCE, abi.RT_SHAPE_SPHERE, abi.RT_SHAPE_CYLINDER,
                                       abi.RT_SHAPE_SPHERE, abi.RT_SHAPE_INTERSECTION, abi.RT_SHAPE_UNION]
    assert (nodes["a"][2], nodes["b"][2]) == (0, 1) and (nodes["a"][7], nodes["b"][7]) == (3, 6) and (nodes["a"][6], nodes["b"][6]) == (4, 5)
    assert objs["shape_param"][6].view(np.uint32)[0] == 2 and objs["shape_param"][7].view(np.uint32)[0] == 7  # root indices
    with pytest.raises(ConfigError, match="unknown shape"):
        SimulationConfig.from_toml_str(BASE + """
[[scene.objects]]
shape = { type = "Torus", radius = 1.0 }
material = { type = "Mirror" }
transform = { position = [0.0, 0.0, 0.0], rotation_y_deg = 0.0 }
""").scene.object_list()
