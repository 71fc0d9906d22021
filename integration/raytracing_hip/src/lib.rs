//! MI355X path for `raytracing_core::Scene::simulate_rays` (scene.rs:39-106) behind the C ABI of `include/rt_hip.h`.
//!
//! `Scene.objects` is `Vec<Box<dyn Hittable>>` (scene.rs:29) and cannot be read back, so the GPU path starts one step
//! earlier, from the `SceneConfig` (plain serde data), and flattens it in the order `impl Into<Scene> for SceneConfig`
//! uses (scene_config.rs:22-112): explicit objects, then every `ObjectGrid` with `i` outer / `j` inner.  Rays are taken
//! from the reference's own expansion, so their order and bits are the reference's.
//!
//! ```ignore
//! // crates/raytracing_cli/src/cli.rs, replacing lines 13-14
//! let results = raytracing_hip::simulate_rays_hip(&scene_cfg_objects, &scene.rays, simulation_settings.into(), 0)?;
//! ```
use glam::Vec3;
use raytracing_config::{
    material_config::MaterialConfig, object_config::ObjectConfig, object_generator_config::ObjectGeneratorConfig,
    object_generator_config::RayGeneratorConfig, scene_config::SceneConfig, shape_config::ShapeConfig,
};
use raytracing_core::{Ray, SimulationSettingsConfig};
use std::ffi::{c_char, c_void, CStr};

// ---- include/rt_hip.h ------------------------------------------------------------------------------------------------
pub const RT_OK: i32 = 0;
const RT_MAT_MIRROR: u32 = 0;
const RT_MAT_GLASS: u32 = 1;
const RT_MAT_HALF_MIRROR: u32 = 2;
const RT_SHAPE_SPHERE: u32 = 0;
const RT_SHAPE_PLANE: u32 = 1;
const RT_SHAPE_BOX: u32 = 2;
const RT_SHAPE_CYLINDER: u32 = 3;
const RT_SHAPE_CONE: u32 = 4;
const RT_SHAPE_WEDGE: u32 = 5;
const RT_SHAPE_LENS: u32 = 6;
const RT_SHAPE_TREE: u32 = 7;
const RT_SHAPE_UNION: u32 = 16;
const RT_SHAPE_INTERSECTION: u32 = 17;
const RT_SHAPE_DIFFERENCE: u32 = 18;
const RT_GEN_PARALLEL_GRID: u32 = 0;
const RT_GEN_PROJECTOR: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct RtObject {
    pub shape: u32,
    pub material_kind: u32,
    pub material_param: f32,
    pub rotation_y_deg: f32,
    pub position: [f32; 3],
    pub shape_param: [f32; 4],
} // 44 bytes

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct RtShapeNode {
    pub shape: u32,
    pub a: u32,
    pub b: u32,
    pub shape_param: [f32; 4],
} // 28 bytes

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct RtRay {
    pub origin: [f32; 3],
    pub direction: [f32; 3],
    pub current_ior: f32,
} // 28 bytes, scene.rs:110-115

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtSettings {
    pub infinity_distance: f32,
    pub max_bounces: u32,
} // scene.rs:33-36

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtRayGenerator {
    pub kind: u32,
    pub count_u: u32,
    pub count_v: u32,
    pub current_ior: f32,
    pub a: [f32; 3],
    pub corner: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub direction: [f32; 3],
} // 76 bytes

#[repr(C)]
pub struct RtSceneHandle {
    _private: [u8; 0],
}

unsafe extern "C" {
    fn rt_hip_init(device_ordinal: i32) -> i32;
    fn rt_hip_last_error() -> *const c_char;
    fn rt_hip_scene_create_ex(objects: *const RtObject, n_objects: u64, nodes: *const RtShapeNode, n_nodes: u64,
                              out_scene: *mut *mut RtSceneHandle) -> i32;
    fn rt_hip_scene_destroy(scene: *mut RtSceneHandle);
    fn rt_hip_simulate(scene: *const RtSceneHandle, rays: *const RtRay, n_rays: u64, settings: RtSettings, seed: u64,
                       ray_id_base: u64, out_points: *mut f32, out_n_points: *mut u32, out_records: *mut c_void) -> i32;
    fn rt_hip_render(scene: *const RtSceneHandle, generator: *const RtRayGenerator, spp: u32, settings: RtSettings, seed: u64,
                     cell_begin: u64, cell_count: u64, cell_step: u64, out_fb: *mut f32, out_first_id: *mut i32,
                     out_first_t: *mut f32, out_seg_count: *mut u32, out_total_segments: *mut u64) -> i32;
}

fn check(rc: i32) -> Result<(), String> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rt_hip_last_error()) }.to_string_lossy().into_owned();
    Err(format!("rt_hip error {rc}: {msg}"))
}

// ---- SceneConfig -> flat object list (scene_config.rs:25-52, object_config.rs:16-30, shape_config.rs:53-149) -----------
fn material(m: &MaterialConfig) -> (u32, f32) {
    match *m {
        MaterialConfig::Mirror => (RT_MAT_MIRROR, 0.0),
        MaterialConfig::Glass { ior } => (RT_MAT_GLASS, ior),
        MaterialConfig::HalfMirror { reflectance } => (RT_MAT_HALF_MIRROR, reflectance),
    }
}

/// Leaf arms: (shape code, shape_param).  `None` for the three operator arms.
fn leaf(shape: &ShapeConfig) -> Option<(u32, [f32; 4])> {
    Some(match *shape {
        ShapeConfig::Sphere { radius } => (RT_SHAPE_SPHERE, [radius, 0.0, 0.0, 0.0]),
        ShapeConfig::Plane { normal } => (RT_SHAPE_PLANE, [normal[0], normal[1], normal[2], 0.0]),
        ShapeConfig::Box { size } => (RT_SHAPE_BOX, [size[0], size[1], size[2], 0.0]),
        ShapeConfig::Cylinder { height, radius } => (RT_SHAPE_CYLINDER, [height, radius, 0.0, 0.0]),
        ShapeConfig::Cone { angle_deg, height } => (RT_SHAPE_CONE, [angle_deg, height, 0.0, 0.0]),
        ShapeConfig::Wedge { size, angle_deg } => (RT_SHAPE_WEDGE, [size[0], size[1], size[2], angle_deg]),
        ShapeConfig::Lens { thickness, diameter, r1, r2 } => (RT_SHAPE_LENS, [thickness, diameter, r1, r2]),
        ShapeConfig::Union { .. } | ShapeConfig::Intersection { .. } | ShapeConfig::Difference { .. } => return None,
    })
}

/// Appends the tree under `shape` to `nodes` (children before their parent) and returns the index of its root node.
fn push_tree(shape: &ShapeConfig, nodes: &mut Vec<RtShapeNode>) -> u32 {
    if let Some((code, shape_param)) = leaf(shape) {
        nodes.push(RtShapeNode { shape: code, a: 0, b: 0, shape_param });
        return (nodes.len() - 1) as u32;
    }
    let (code, a, b) = match shape {
        ShapeConfig::Union { a, b } => (RT_SHAPE_UNION, a, b),
        ShapeConfig::Intersection { a, b } => (RT_SHAPE_INTERSECTION, a, b),
        ShapeConfig::Difference { a, b } => (RT_SHAPE_DIFFERENCE, a, b),
        _ => unreachable!(),
    };
    let ia = push_tree(a, nodes);
    let ib = push_tree(b, nodes);
    nodes.push(RtShapeNode { shape: code, a: ia, b: ib, shape_param: [0.0; 4] });
    (nodes.len() - 1) as u32
}

fn object(cfg: &ObjectConfig, position: [f32; 3], nodes: &mut Vec<RtShapeNode>) -> RtObject {
    let (material_kind, material_param) = material(&cfg.material);
    let (shape, shape_param) = match leaf(&cfg.shape) {
        Some(l) => l,
        None => {
            let root = push_tree(&cfg.shape, nodes);
            (RT_SHAPE_TREE, [f32::from_bits(root), 0.0, 0.0, 0.0]) // the root index travels as the bits of shape_param[0]
        }
    };
    RtObject { shape, material_kind, material_param, rotation_y_deg: cfg.transform.rotation_y_deg, position, shape_param }
}

/// `Scene.objects` order: explicit `[[scene.objects]]`, then each `ObjectGrid`, `i` (x) outer, `j` (z) inner.
pub fn flatten_objects(scene: &SceneConfig) -> (Vec<RtObject>, Vec<RtShapeNode>) {
    let mut nodes = Vec::new();
    let mut objects: Vec<RtObject> = scene.objects.iter().map(|o| object(o, o.transform.position, &mut nodes)).collect();
    for generator in &scene.object_generators {
        match generator {
            ObjectGeneratorConfig::ObjectGrid { count_x, count_z, position_start, step_x, step_z, template } => {
                let (start, x_step, z_step) = (Vec3::from(*position_start), Vec3::from(*step_x), Vec3::from(*step_z));
                for i in 0..*count_x {
                    for j in 0..*count_z {
                        let pos = start + (i as f32 * x_step) + (j as f32 * z_step); // scene_config.rs:44, same operation order
                        objects.push(object(template, pos.to_array(), &mut nodes));
                    }
                }
            }
        }
    }
    (objects, nodes)
}

pub fn flatten_rays(rays: &[Ray]) -> Vec<RtRay> {
    rays.iter()
        .map(|r| RtRay { origin: r.origin.to_array(), direction: r.direction.to_array(), current_ior: r.current_ior })
        .collect()
}

pub fn generator(g: &RayGeneratorConfig) -> RtRayGenerator {
    match *g {
        RayGeneratorConfig::ParallelGrid { origin_corner, vec_u, vec_v, count_u, count_v, direction, current_ior } => RtRayGenerator {
            kind: RT_GEN_PARALLEL_GRID, count_u, count_v, current_ior, a: origin_corner, corner: [0.0; 3], u: vec_u, v: vec_v, direction,
        },
        RayGeneratorConfig::Projector { origin, target_corner, target_u, target_v, count_u, count_v, current_ior } => RtRayGenerator {
            kind: RT_GEN_PROJECTOR, count_u, count_v, current_ior, a: origin, corner: target_corner, u: target_u, v: target_v,
            direction: [0.0; 3],
        },
    }
}

// ---- the scene on the GPU ---------------------------------------------------------------------------------------------
pub struct HipScene {
    handle: *mut RtSceneHandle,
}

impl HipScene {
    /// Uploads the object list of `scene` (call before `scene.into()` consumes it).
    pub fn new(scene: &SceneConfig, device: i32) -> Result<Self, String> {
        let (objects, nodes) = flatten_objects(scene);
        let mut handle = std::ptr::null_mut();
        unsafe {
            check(rt_hip_init(device))?;
            check(rt_hip_scene_create_ex(objects.as_ptr(), objects.len() as u64,
                                         if nodes.is_empty() { std::ptr::null() } else { nodes.as_ptr() }, nodes.len() as u64, &mut handle))?;
        }
        Ok(HipScene { handle })
    }

    /// Drop-in for `Scene::simulate_rays` (scene.rs:39): same return shape, same ray order.  `seed` keys the HalfMirror
    /// stream u(seed, ray index, bounce) that replaces `thread_rng()` (scene.rs:88).
    pub fn simulate_rays(&self, rays: &[Ray], setting: SimulationSettingsConfig, seed: u64) -> Result<Vec<Vec<Vec3>>, String> {
        let flat = flatten_rays(rays);
        if setting.max_bounces == 0 || flat.is_empty() {
            return Ok(rays.iter().map(|r| vec![r.origin]).collect()); // scene.rs:44-45 with an empty bounce loop
        }
        let stride = setting.max_bounces as usize + 1;
        let mut pts = vec![0f32; flat.len() * stride * 3];
        let mut npts = vec![0u32; flat.len()];
        let st = RtSettings { infinity_distance: setting.infinity_distance, max_bounces: setting.max_bounces };
        unsafe {
            check(rt_hip_simulate(self.handle, flat.as_ptr(), flat.len() as u64, st, seed, 0, pts.as_mut_ptr(), npts.as_mut_ptr(),
                                  std::ptr::null_mut()))?;
        }
        Ok((0..flat.len())
            .map(|r| (0..npts[r] as usize).map(|k| { let b = (r * stride + k) * 3; Vec3::new(pts[b], pts[b + 1], pts[b + 2]) }).collect())
            .collect())
    }

    /// The per-cell x per-sample loop (no reference call site): RGB per generator cell, cell = i * count_v + j.
    pub fn render(&self, g: &RayGeneratorConfig, spp: u32, setting: SimulationSettingsConfig, seed: u64) -> Result<(Vec<f32>, u64), String> {
        let gen = generator(g);
        let cells = gen.count_u as u64 * gen.count_v as u64;
        let mut fb = vec![0f32; cells as usize * 3];
        let mut total = 0u64;
        let st = RtSettings { infinity_distance: setting.infinity_distance, max_bounces: setting.max_bounces };
        unsafe {
            check(rt_hip_render(self.handle, &gen, spp, st, seed, 0, cells, 1, fb.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(),
                                std::ptr::null_mut(), &mut total))?;
        }
        Ok((fb, total))
    }
}

impl Drop for HipScene {
    fn drop(&mut self) {
        unsafe { rt_hip_scene_destroy(self.handle) }
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn struct_sizes_match_the_header() {
        assert_eq!(std::mem::size_of::<RtObject>(), 44);
        assert_eq!(std::mem::size_of::<RtShapeNode>(), 28);
        assert_eq!(std::mem::size_of::<RtRay>(), 28);
        assert_eq!(std::mem::size_of::<RtRayGenerator>(), 76);
        assert_eq!(std::mem::size_of::<RtSettings>(), 8);
    }
}
