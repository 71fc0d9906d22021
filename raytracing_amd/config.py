"""TOML schema loader: the host-side mirror of crates/raytracing_config (schema unchanged).

`SimulationConfig.load_from_path` (simulation_config.rs:14-19) -> `SceneConfig.into_scene_data()`
(impl Into<Scene> for SceneConfig, scene_config.rs:22-113).  Object order: explicit [[scene.objects]] first,
then every ObjectGrid with i (x) outer, j (z) inner; ray order: explicit [[scene.rays]] first, then every
generator with i (u) outer, j (v) inner.  All arithmetic is IEEE f32 in the reference's operation order.
"""
from dataclasses import dataclass, field

import numpy as np

try:  # Python >= 3.11
    import tomllib as _toml
except ImportError:  # pragma: no cover
    import tomli as _toml

from . import abi
from .engine import SimulationSettingsConfig

f32 = np.float32


class ConfigError(ValueError):
    """Raised where serde/toml deserialisation of the reference would fail."""


def _req(table, key, where):
    if not isinstance(table, dict) or key not in table:
        raise ConfigError(f"missing field `{key}` in {where}")
    return table[key]


def _vec3(v, where):
    if not isinstance(v, (list, tuple)) or len(v) != 3:
        raise ConfigError(f"{where}: expected an array of 3 numbers")
    return np.array([f32(x) for x in v], dtype=f32)


def _material(t, where):
    """MaterialConfig (material_config.rs:6-21), #[serde(tag = "type")]."""
    kind = _req(t, "type", where)
    if kind == "Mirror":
        return abi.RT_MAT_MIRROR, f32(0.0)
    if kind == "Glass":
        return abi.RT_MAT_GLASS, f32(_req(t, "ior", where))
    if kind == "HalfMirror":
        return abi.RT_MAT_HALF_MIRROR, f32(_req(t, "reflectance", where))
    raise ConfigError(f"{where}: unknown material type `{kind}`")


_OPS = {"Union": abi.RT_SHAPE_UNION, "Intersection": abi.RT_SHAPE_INTERSECTION, "Difference": abi.RT_SHAPE_DIFFERENCE}


def _leaf_shape(shape, where):
    """A non-operator ShapeConfig arm -> (RT_SHAPE_*, shape_param[4])  (shape_config.rs:11-38)."""
    kind = _req(shape, "type", where)
    if kind == "Sphere":
        return abi.RT_SHAPE_SPHERE, (f32(_req(shape, "radius", where)), 0, 0, 0)
    if kind == "Plane":
        n = _vec3(_req(shape, "normal", where), where + ".normal")
        return abi.RT_SHAPE_PLANE, (n[0], n[1], n[2], 0)
    if kind == "Box":
        sz = _vec3(_req(shape, "size", where), where + ".size")
        return abi.RT_SHAPE_BOX, (sz[0], sz[1], sz[2], 0)
    if kind == "Cylinder":
        return abi.RT_SHAPE_CYLINDER, (f32(_req(shape, "height", where)), f32(_req(shape, "radius", where)), 0, 0)
    if kind == "Cone":
        return abi.RT_SHAPE_CONE, (f32(_req(shape, "angle_deg", where)), f32(_req(shape, "height", where)), 0, 0)
    if kind == "Wedge":
        sz = _vec3(_req(shape, "size", where), where + ".size")
        return abi.RT_SHAPE_WEDGE, (sz[0], sz[1], sz[2], f32(_req(shape, "angle_deg", where)))
    if kind == "Lens":
        return abi.RT_SHAPE_LENS, (f32(_req(shape, "thickness", where)), f32(_req(shape, "diameter", where)),
                                   f32(_req(shape, "r1", where)), f32(_req(shape, "r2", where)))
    raise ConfigError(f"{where}: unknown shape type `{kind}`")


def _shape_node(shape, where, nodes):
    """Append the ShapeConfig subtree to `nodes` (RtShapeNode records) and return its index (shape_config.rs:39-50)."""
    kind = _req(shape, "type", where)
    rec = np.zeros((), abi.NODE_DTYPE)
    if kind in _OPS:
        a = _shape_node(_req(shape, "a", where), where + ".a", nodes)
        b = _shape_node(_req(shape, "b", where), where + ".b", nodes)
        rec["shape"], rec["a"], rec["b"] = _OPS[kind], a, b
    else:
        rec["shape"], sp = _leaf_shape(shape, where)
        rec["shape_param"] = sp
    nodes.append(rec)
    return len(nodes) - 1


def _object(t, where, position_override=None, nodes=None):
    """ObjectConfig -> one RtObject (object_config.rs:10-30, shape_config.rs:53-149)."""
    shape = _req(t, "shape", where)
    mat = _req(t, "material", where)
    tr = _req(t, "transform", where)
    o = np.zeros((), abi.OBJECT_DTYPE)
    kind = _req(shape, "type", where + ".shape")
    if kind in _OPS:
        if nodes is None:
            raise ConfigError(f"{where}: CSG tree needs a node list")
        root = _shape_node(shape, where + ".shape", nodes)
        o["shape"] = abi.RT_SHAPE_TREE
        sp = np.zeros(4, f32)
        sp.view(np.uint32)[0] = root
        o["shape_param"] = sp
    else:
        o["shape"], sp = _leaf_shape(shape, where + ".shape")
        o["shape_param"] = sp
    o["material_kind"], o["material_param"] = _material(mat, where + ".material")
    pos = _vec3(_req(tr, "position", where + ".transform"), where + ".transform.position")
    o["position"] = pos if position_override is None else position_override
    o["rotation_y_deg"] = f32(_req(tr, "rotation_y_deg", where + ".transform"))
    return o


def _normalize(v):
    """glam Vec3::normalize: v * (1 / sqrt((x*x + y*y) + z*z)) in f32."""
    d = f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])
    rcp = f32(1.0) / np.sqrt(f32(d))
    return (v * f32(rcp)).astype(f32)


@dataclass
class SceneConfig:
    """scene_config.rs:10-20 (every list defaults to empty)."""
    rays: list = field(default_factory=list)
    ray_generators: list = field(default_factory=list)
    object_generators: list = field(default_factory=list)
    objects: list = field(default_factory=list)

    def object_list(self):
        """scene_config.rs:25-52 -> numpy array of RtObject in `Scene.objects` order."""
        return self.object_and_node_lists()[0]

    def object_and_node_lists(self):
        """(objects, nodes): nodes hold the Union/Intersection/Difference trees that RT_SHAPE_TREE objects point into."""
        nodes = []
        out = [_object(t, f"scene.objects[{k}]", nodes=nodes) for k, t in enumerate(self.objects)]
        for g, gen in enumerate(self.object_generators):
            where = f"scene.object_generators[{g}]"
            kind = _req(gen, "type", where)
            if kind != "ObjectGrid":
                raise ConfigError(f"{where}: unknown object generator `{kind}`")
            start = _vec3(_req(gen, "position_start", where), where + ".position_start")
            sx = _vec3(_req(gen, "step_x", where), where + ".step_x")
            sz = _vec3(_req(gen, "step_z", where), where + ".step_z")
            template = _req(gen, "template", where)
            for i in range(int(_req(gen, "count_x", where))):
                for j in range(int(_req(gen, "count_z", where))):
                    pos = ((start + f32(i) * sx).astype(f32) + f32(j) * sz).astype(f32)  # scene_config.rs:44
                    out.append(_object(template, where + ".template", position_override=pos, nodes=nodes))
        objs = np.array(out, dtype=abi.OBJECT_DTYPE) if out else np.zeros(0, abi.OBJECT_DTYPE)
        nds = np.array(nodes, dtype=abi.NODE_DTYPE) if nodes else np.zeros(0, abi.NODE_DTYPE)
        return objs, nds

    def generators(self):
        """RayGeneratorConfig list as C-ABI RtRayGenerator structs (object_generator_config.rs:13-33)."""
        from .engine import parallel_grid, projector

        gens = []
        for g, gen in enumerate(self.ray_generators):
            where = f"scene.ray_generators[{g}]"
            kind = _req(gen, "type", where)
            if kind == "ParallelGrid":
                gens.append(parallel_grid(_req(gen, "origin_corner", where), _req(gen, "vec_u", where), _req(gen, "vec_v", where),
                                          _req(gen, "count_u", where), _req(gen, "count_v", where), _req(gen, "direction", where),
                                          _req(gen, "current_ior", where)))
            elif kind == "Projector":
                gens.append(projector(_req(gen, "origin", where), _req(gen, "target_corner", where), _req(gen, "target_u", where),
                                      _req(gen, "target_v", where), _req(gen, "count_u", where), _req(gen, "count_v", where),
                                      _req(gen, "current_ior", where)))
            else:
                raise ConfigError(f"{where}: unknown ray generator `{kind}`")
        return gens

    def ray_list(self):
        """scene_config.rs:55-110 -> numpy array of RtRay in `Scene.rays` order."""
        rays = []
        for k, t in enumerate(self.rays):  # ray_config.rs:12-19: direction normalised, current_ior = 1.0
            where = f"scene.rays[{k}]"
            o = _vec3(_req(t, "origin", where), where + ".origin")
            d = _normalize(_vec3(_req(t, "direction", where), where + ".direction"))
            rays.append((o, d, f32(1.0)))
        for gen in self.generators():
            a = np.array(gen.a[:], f32)
            u_step = (np.array(gen.u[:], f32) / f32(gen.count_u)).astype(f32)
            v_step = (np.array(gen.v[:], f32) / f32(gen.count_v)).astype(f32)
            if gen.kind == abi.RT_GEN_PARALLEL_GRID:
                d = _normalize(np.array(gen.direction[:], f32))
                for i in range(gen.count_u):
                    for j in range(gen.count_v):
                        o = ((a + f32(i) * u_step).astype(f32) + f32(j) * v_step).astype(f32)
                        rays.append((o, d, f32(gen.current_ior)))
            else:
                corner = np.array(gen.corner[:], f32)
                for i in range(gen.count_u):
                    for j in range(gen.count_v):
                        target = ((corner + f32(i) * u_step).astype(f32) + f32(j) * v_step).astype(f32)
                        rays.append((a, _normalize((target - a).astype(f32)), f32(gen.current_ior)))
        out = np.zeros(len(rays), abi.RAY_DTYPE)
        for k, (o, d, ior) in enumerate(rays):
            out[k]["origin"], out[k]["direction"], out[k]["current_ior"] = o, d, ior
        return out

    def into_scene(self):
        """`let scene: Scene = scene.into();` (cli.rs:13) — uploads the object list to the GPU."""
        from .engine import Scene

        objs, nodes = self.object_and_node_lists()
        return Scene(objs, self.ray_list(), nodes=nodes)


@dataclass
class SimulationConfig:
    """simulation_config.rs:7-11: both tables are required (no #[serde(default)])."""
    simulation_settings: SimulationSettingsConfig
    scene: SceneConfig

    @staticmethod
    def from_toml_str(text):
        try:
            doc = _toml.loads(text)
        except Exception as e:  # toml::from_str error
            raise ConfigError(f"TOML parse error: {e}") from e
        st = _req(doc, "simulation_settings", "document")
        sc = _req(doc, "scene", "document")
        mb = _req(st, "max_bounces", "simulation_settings")
        if not isinstance(mb, int) or mb < 0:
            raise ConfigError("simulation_settings.max_bounces must be a u32")
        settings = SimulationSettingsConfig(float(f32(_req(st, "infinity_distance", "simulation_settings"))), int(mb))
        scene = SceneConfig(rays=list(sc.get("rays", [])), ray_generators=list(sc.get("ray_generators", [])),
                            object_generators=list(sc.get("object_generators", [])), objects=list(sc.get("objects", [])))
        return SimulationConfig(settings, scene)

    @staticmethod
    def load_from_path(path):
        with open(path, "rb") as f:
            return SimulationConfig.from_toml_str(f.read().decode("utf-8"))
