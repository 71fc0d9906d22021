// scene_config.h — C++ mirror of crates/raytracing_config: SimulationConfig { simulation_settings, scene }
// (simulation_config.rs:7-19) and `impl Into<Scene> for SceneConfig` (scene_config.rs:22-113), producing the plain-data
// lists the C ABI takes (include/rt_hip.h).  All arithmetic is IEEE f32 in the reference's operation order
// (build with -ffp-contract=off).
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "toml_lite.h"

namespace rtcfg {

using toml_lite::Value;

struct ConfigError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct SimulationConfig {
    RtSettings settings{};
    std::vector<RtObject> objects;    // Scene.objects order (scene_config.rs:25-52)
    std::vector<RtShapeNode> nodes;   // Union / Intersection / Difference trees
    std::vector<RtRay> rays;          // Scene.rays order (scene_config.rs:55-110)
    std::vector<RtRayGenerator> generators;  // [[scene.ray_generators]] as given, for the render loop (rt_hip_render)
};

inline const Value& req(const Value& t, const std::string& key, const std::string& where)
{
    if (t.kind != Value::TABLE) throw ConfigError(where + ": expected a table");
    auto it = t.tab.find(key);
    if (it == t.tab.end()) throw ConfigError("missing field `" + key + "` in " + where);
    return *it->second;
}
inline float num(const Value& v, const std::string& where)
{
    if (!v.is_number()) throw ConfigError(where + ": expected a number");
    return (float)v.number();  // serde accepts an integer for an f32 field
}
inline uint32_t u32(const Value& v, const std::string& where)
{
    if (v.kind != Value::INT || v.i < 0 || v.i > 0xffffffffll) throw ConfigError(where + ": expected a u32");
    return (uint32_t)v.i;
}
inline void vec3(const Value& v, const std::string& where, float out[3])
{
    if (v.kind != Value::ARRAY || v.arr.size() != 3) throw ConfigError(where + ": expected an array of 3 numbers");
    for (int k = 0; k < 3; ++k) out[k] = num(*v.arr[k], where);
}
inline std::string str(const Value& v, const std::string& where)
{
    if (v.kind != Value::STRING) throw ConfigError(where + ": expected a string");
    return v.s;
}

// glam Vec3::normalize: v * (1 / sqrt((x*x + y*y) + z*z))
inline void normalize3(float v[3])
{
    const float d = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    const float rcp = 1.0f / std::sqrt(d);
    v[0] = v[0] * rcp;
    v[1] = v[1] * rcp;
    v[2] = v[2] * rcp;
}

inline void material(const Value& t, const std::string& where, uint32_t& kind, float& param)
{
    const std::string ty = str(req(t, "type", where), where + ".type");  // material_config.rs:6-21
    param = 0.0f;
    if (ty == "Mirror") kind = RT_MAT_MIRROR;
    else if (ty == "Glass") {
        kind = RT_MAT_GLASS;
        param = num(req(t, "ior", where), where + ".ior");
    } else if (ty == "HalfMirror") {
        kind = RT_MAT_HALF_MIRROR;
        param = num(req(t, "reflectance", where), where + ".reflectance");
    } else
        throw ConfigError(where + ": unknown material type `" + ty + "`");
}

inline bool op_code(const std::string& ty, uint32_t& op)
{
    if (ty == "Union") op = RT_SHAPE_UNION;
    else if (ty == "Intersection") op = RT_SHAPE_INTERSECTION;
    else if (ty == "Difference") op = RT_SHAPE_DIFFERENCE;
    else return false;
    return true;
}

// a non-operator ShapeConfig arm (shape_config.rs:11-38)
inline uint32_t leaf_shape(const Value& s, const std::string& where, float sp[4])
{
    const std::string ty = str(req(s, "type", where), where + ".type");
    sp[0] = sp[1] = sp[2] = sp[3] = 0.0f;
    if (ty == "Sphere") {
        sp[0] = num(req(s, "radius", where), where + ".radius");
        return RT_SHAPE_SPHERE;
    }
    if (ty == "Plane") {
        vec3(req(s, "normal", where), where + ".normal", sp);
        return RT_SHAPE_PLANE;
    }
    if (ty == "Box") {
        vec3(req(s, "size", where), where + ".size", sp);
        return RT_SHAPE_BOX;
    }
    if (ty == "Cylinder") {
        sp[0] = num(req(s, "height", where), where + ".height");
        sp[1] = num(req(s, "radius", where), where + ".radius");
        return RT_SHAPE_CYLINDER;
    }
    if (ty == "Cone") {
        sp[0] = num(req(s, "angle_deg", where), where + ".angle_deg");
        sp[1] = num(req(s, "height", where), where + ".height");
        return RT_SHAPE_CONE;
    }
    if (ty == "Wedge") {
        vec3(req(s, "size", where), where + ".size", sp);
        sp[3] = num(req(s, "angle_deg", where), where + ".angle_deg");
        return RT_SHAPE_WEDGE;
    }
    if (ty == "Lens") {
        sp[0] = num(req(s, "thickness", where), where + ".thickness");
        sp[1] = num(req(s, "diameter", where), where + ".diameter");
        sp[2] = num(req(s, "r1", where), where + ".r1");
        sp[3] = num(req(s, "r2", where), where + ".r2");
        return RT_SHAPE_LENS;
    }
    throw ConfigError(where + ": unknown shape type `" + ty + "`");
}

inline uint32_t shape_node(const Value& s, const std::string& where, std::vector<RtShapeNode>& nodes)
{
    RtShapeNode n{};
    uint32_t op;
    if (op_code(str(req(s, "type", where), where + ".type"), op)) {
        const uint32_t a = shape_node(req(s, "a", where), where + ".a", nodes);
        const uint32_t b = shape_node(req(s, "b", where), where + ".b", nodes);
        n.shape = op;
        n.a = a;
        n.b = b;
    } else {
        n.shape = leaf_shape(s, where, n.shape_param);
    }
    nodes.push_back(n);
    return (uint32_t)nodes.size() - 1;
}

// ObjectConfig (object_config.rs:10-30)
inline RtObject object(const Value& t, const std::string& where, std::vector<RtShapeNode>& nodes, const float* position_override)
{
    RtObject o{};
    const Value& shape = req(t, "shape", where);
    const Value& mat = req(t, "material", where);
    const Value& tr = req(t, "transform", where);
    uint32_t op;
    if (op_code(str(req(shape, "type", where + ".shape"), where + ".shape.type"), op)) {
        const uint32_t root = shape_node(shape, where + ".shape", nodes);
        o.shape = RT_SHAPE_TREE;
        std::memcpy(&o.shape_param[0], &root, 4);
    } else {
        o.shape = leaf_shape(shape, where + ".shape", o.shape_param);
    }
    material(mat, where + ".material", o.material_kind, o.material_param);
    vec3(req(tr, "position", where + ".transform"), where + ".transform.position", o.position);
    if (position_override) std::memcpy(o.position, position_override, 12);
    o.rotation_y_deg = num(req(tr, "rotation_y_deg", where + ".transform"), where + ".transform.rotation_y_deg");
    return o;
}

inline const toml_lite::Array& list_or_empty(const Value& scene, const std::string& key)
{
    static const toml_lite::Array empty;
    auto it = scene.tab.find(key);  // #[serde(default)] on every SceneConfig list (scene_config.rs:12-19)
    if (it == scene.tab.end()) return empty;
    if (it->second->kind != Value::ARRAY) throw ConfigError("scene." + key + ": expected an array");
    return it->second->arr;
}

// A generator expands on the host (the reference pushes every object / ray into a Vec, scene_config.rs:25-110); a count
// product beyond this is refused with an error instead of exhausting memory (the reference would abort in the allocator).
constexpr uint64_t kMaxGeneratorItems = 1ull << 27;
inline void check_expansion(uint32_t a, uint32_t b, const std::string& where)
{
    if ((uint64_t)a * b > kMaxGeneratorItems) throw ConfigError(where + ": generator expands to more than 2^27 items");
}

inline SimulationConfig from_toml(const std::string& text)
{
    std::shared_ptr<Value> doc = toml_lite::parse(text);
    SimulationConfig cfg;
    const Value& st = req(*doc, "simulation_settings", "document");  // no serde default: both tables are required
    const Value& sc = req(*doc, "scene", "document");
    cfg.settings.infinity_distance = num(req(st, "infinity_distance", "simulation_settings"), "simulation_settings.infinity_distance");
    cfg.settings.max_bounces = u32(req(st, "max_bounces", "simulation_settings"), "simulation_settings.max_bounces");

    // --- objects: explicit first, then ObjectGrid generators, i (x) outer, j (z) inner (scene_config.rs:25-52)
    size_t k = 0;
    for (auto& e : list_or_empty(sc, "objects")) cfg.objects.push_back(object(*e, "scene.objects[" + std::to_string(k++) + "]", cfg.nodes, nullptr));
    k = 0;
    for (auto& e : list_or_empty(sc, "object_generators")) {
        const std::string where = "scene.object_generators[" + std::to_string(k++) + "]";
        if (str(req(*e, "type", where), where + ".type") != "ObjectGrid") throw ConfigError(where + ": unknown object generator");
        float start[3], sx[3], sz[3];
        vec3(req(*e, "position_start", where), where + ".position_start", start);
        vec3(req(*e, "step_x", where), where + ".step_x", sx);
        vec3(req(*e, "step_z", where), where + ".step_z", sz);
        const uint32_t cx = u32(req(*e, "count_x", where), where + ".count_x"), cz = u32(req(*e, "count_z", where), where + ".count_z");
        const Value& tmpl = req(*e, "template", where);
        check_expansion(cx, cz, where);
        for (uint32_t i = 0; i < cx; ++i)
            for (uint32_t j = 0; j < cz; ++j) {
                float pos[3];
                for (int c = 0; c < 3; ++c) pos[c] = (start[c] + (float)i * sx[c]) + (float)j * sz[c];  // scene_config.rs:44
                cfg.objects.push_back(object(tmpl, where + ".template", cfg.nodes, pos));
            }
    }

    // --- rays: explicit first (normalised, ior 1.0: ray_config.rs:12-19), then generators, i (u) outer, j (v) inner
    k = 0;
    for (auto& e : list_or_empty(sc, "rays")) {
        const std::string where = "scene.rays[" + std::to_string(k++) + "]";
        RtRay r{};
        vec3(req(*e, "origin", where), where + ".origin", r.origin);
        vec3(req(*e, "direction", where), where + ".direction", r.direction);
        normalize3(r.direction);
        r.current_ior = 1.0f;
        cfg.rays.push_back(r);
    }
    k = 0;
    for (auto& e : list_or_empty(sc, "ray_generators")) {
        const std::string where = "scene.ray_generators[" + std::to_string(k++) + "]";
        const std::string ty = str(req(*e, "type", where), where + ".type");
        const uint32_t cu = u32(req(*e, "count_u", where), where + ".count_u"), cv = u32(req(*e, "count_v", where), where + ".count_v");
        const float ior = num(req(*e, "current_ior", where), where + ".current_ior");
        check_expansion(cu, cv, where);
        float a[3], u[3], v[3], us[3], vs[3];
        if (ty == "ParallelGrid") {  // scene_config.rs:60-83
            float dir[3];
            vec3(req(*e, "origin_corner", where), where + ".origin_corner", a);
            vec3(req(*e, "vec_u", where), where + ".vec_u", u);
            vec3(req(*e, "vec_v", where), where + ".vec_v", v);
            vec3(req(*e, "direction", where), where + ".direction", dir);
            for (int c = 0; c < 3; ++c) {
                us[c] = u[c] / (float)cu;
                vs[c] = v[c] / (float)cv;
            }
            RtRayGenerator g{};  // raw direction: the device normalises it once, like scene_config.rs:67
            g.kind = RT_GEN_PARALLEL_GRID;
            g.count_u = cu;
            g.count_v = cv;
            g.current_ior = ior;
            for (int c = 0; c < 3; ++c) {
                g.a[c] = a[c];
                g.u[c] = u[c];
                g.v[c] = v[c];
                g.direction[c] = dir[c];
            }
            cfg.generators.push_back(g);
            normalize3(dir);
            for (uint32_t i = 0; i < cu; ++i)
                for (uint32_t j = 0; j < cv; ++j) {
                    RtRay r{};
                    for (int c = 0; c < 3; ++c) {
                        r.origin[c] = (a[c] + (float)i * us[c]) + (float)j * vs[c];
                        r.direction[c] = dir[c];
                    }
                    r.current_ior = ior;
                    cfg.rays.push_back(r);
                }
        } else if (ty == "Projector") {  // scene_config.rs:84-108
            float corner[3];
            vec3(req(*e, "origin", where), where + ".origin", a);
            vec3(req(*e, "target_corner", where), where + ".target_corner", corner);
            vec3(req(*e, "target_u", where), where + ".target_u", u);
            vec3(req(*e, "target_v", where), where + ".target_v", v);
            for (int c = 0; c < 3; ++c) {
                us[c] = u[c] / (float)cu;
                vs[c] = v[c] / (float)cv;
            }
            RtRayGenerator g{};
            g.kind = RT_GEN_PROJECTOR;
            g.count_u = cu;
            g.count_v = cv;
            g.current_ior = ior;
            for (int c = 0; c < 3; ++c) {
                g.a[c] = a[c];
                g.corner[c] = corner[c];
                g.u[c] = u[c];
                g.v[c] = v[c];
            }
            cfg.generators.push_back(g);
            for (uint32_t i = 0; i < cu; ++i)
                for (uint32_t j = 0; j < cv; ++j) {
                    RtRay r{};
                    for (int c = 0; c < 3; ++c) {
                        const float target = (corner[c] + (float)i * us[c]) + (float)j * vs[c];
                        r.origin[c] = a[c];
                        r.direction[c] = target - a[c];
                    }
                    normalize3(r.direction);
                    r.current_ior = ior;
                    cfg.rays.push_back(r);
                }
        } else {
            throw ConfigError(where + ": unknown ray generator `" + ty + "`");
        }
    }
    return cfg;
}

}  // namespace rtcfg
