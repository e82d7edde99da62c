//! `repr(C)` mirrors of include/rt_core.h and a safe builder on top of them.
//!
//! The reference side implements `Flatten` for its shapes (one added trait method on
//! `Intersectable`) and calls `Scene::render_rgb8` where it calls
//! `scene::camera::render` today (src/main.rs:53).
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RT_SPHERE: i32 = 0;
pub const RT_PLANE: i32 = 1;
pub const RT_CUBE: i32 = 2;
pub const RT_CYLINDER: i32 = 3;
pub const RT_CONE: i32 = 4;
pub const RT_SOLID: i32 = 0;
pub const RT_STRIPE: i32 = 1;
pub const RT_GRADIENT: i32 = 2;
pub const RT_RINGS: i32 = 3;
pub const RT_CHECKERED: i32 = 4;
pub const RT_MODE_F64_PARITY: c_int = 0;
pub const RT_MODE_F32_FAST: c_int = 1;
/// The same mode under the name that says what it is: the f32-search variant (not faster on CDNA4).
pub const RT_MODE_F32_SEARCH: c_int = 1;
pub const RT_PIPELINE_MEGAKERNEL: c_int = 0;
pub const RT_PIPELINE_WAVEFRONT: c_int = 1;
pub const RT_PIPELINE_AUTO: c_int = 2;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtMaterial {
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern_kind: i32,
    pub reserved: i32,
    pub color_a: [f64; 3],
    pub color_b: [f64; 3],
    pub pattern_inv: [f64; 16],
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtGroup {
    pub inv: [f64; 16],
    pub parent: i32,
    pub reserved: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtPrimitive {
    pub kind: i32,
    pub material: i32,
    pub parent: i32,
    pub closed: i32,
    pub minimum: f64,
    pub maximum: f64,
    pub inv: [f64; 16],
}
#[repr(C)]
pub struct RtMesh {
    pub group: i32,
    pub material: i32,
    pub ntris: i64,
    pub p1: *const f64,
    pub p2: *const f64,
    pub p3: *const f64,
    pub n1: *const f64,
    pub n2: *const f64,
    pub n3: *const f64,
    pub smooth: *const u8,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCamera {
    pub hsize: i32,
    pub vsize: i32,
    pub half_width: f64,
    pub half_height: f64,
    pub pixel_size: f64,
    pub inv: [f64; 16],
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtStats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_reflect: u64,
    pub rays_refract: u64,
    pub depth0_skipped: u64,
    pub pred_queries: u64,
    pub nodes_visited: u64,
    pub tri_tests: u64,
    pub prim_tests: u64,
    pub inst_entries: u64,
    pub shade_calls: u64,
    pub launches: u64,
    pub kernel_ms: f64,
    pub total_ms: f64,
    pub max_nodes_per_query: u64,
    pub node_wave_steps: u64,
    pub tri_wave_steps: u64,
    pub fast_candidates: u64,
    pub fast_exact: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtSceneInfo {
    pub n_groups: i64,
    pub n_primitives: i64,
    pub n_meshes: i64,
    pub n_triangles: i64,
    pub n_materials: i64,
    pub n_lights: i64,
    pub n_bvh_nodes: i64,
    pub device_bytes: i64,
    pub node_bytes: i32,
    pub tri_bytes: i32,
    pub device: i32,
    pub bvh_builder: i32,
    pub commit_ms: f64,
}
pub const RT_TILE: i32 = 32;
pub const RT_NOALLOC_MAX_TILES: i64 = 8192;

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_create() -> *mut c_void;
    pub fn rt_scene_destroy(s: *mut c_void);
    pub fn rt_scene_add_material(s: *mut c_void, m: *const RtMaterial) -> c_int;
    pub fn rt_scene_add_group(s: *mut c_void, g: *const RtGroup) -> c_int;
    pub fn rt_scene_add_primitive(s: *mut c_void, p: *const RtPrimitive) -> c_int;
    pub fn rt_scene_add_mesh(s: *mut c_void, m: *const RtMesh) -> c_int;
    pub fn rt_scene_add_light(s: *mut c_void, intensity: *const f64, position: *const f64) -> c_int;
    pub fn rt_scene_commit(s: *mut c_void, device: c_int, bvh_builder: c_int) -> c_int;
    pub fn rt_render(
        s: *mut c_void,
        cam: *const RtCamera,
        depth: c_int,
        mode: c_int,
        pipeline: c_int,
        out_rgb_f64: *mut f64,
        out_rgb_u8: *mut u8,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_scene_get_info(s: *const c_void, info: *mut RtSceneInfo) -> c_int;
    /// Device buffers, asynchronous on `hip_stream` (a `hipStream_t`); renders the tiles
    /// `t % tile_step == tile_first` (include/rt_core.h).
    pub fn rt_render_device(
        s: *mut c_void,
        cam: *const RtCamera,
        depth: c_int,
        mode: c_int,
        pipeline: c_int,
        tile_first: c_int,
        tile_step: c_int,
        d_rgb_f64: *mut f64,
        d_rgb_u8: *mut u8,
        hip_stream: *mut c_void,
        want_stats: c_int,
    ) -> c_int;
    pub fn rt_render_stats(s: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_shard_tiles(hsize: c_int, vsize: c_int, tile_first: c_int, tile_step: c_int) -> i64;
    pub fn rt_detile_u8(
        d_tiles: *const u8,
        nshards: c_int,
        tiles_per_shard: i64,
        hsize: c_int,
        vsize: c_int,
        d_image: *mut u8,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn rt_detile_f64(
        d_tiles: *const f64,
        nshards: c_int,
        tiles_per_shard: i64,
        hsize: c_int,
        vsize: c_int,
        d_image: *mut f64,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn rt_scene_commit_multi(s: *mut c_void, device_ids: *const c_int, ndev: c_int, bvh_builder: c_int) -> c_int;
    pub fn rt_render_multi(
        s: *mut c_void,
        cam: *const RtCamera,
        depth: c_int,
        mode: c_int,
        pipeline: c_int,
        out_rgb_u8: *mut u8,
        d_rgb_u8: *mut u8,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_render_multi_f64(
        s: *mut c_void,
        cam: *const RtCamera,
        depth: c_int,
        mode: c_int,
        pipeline: c_int,
        out_rgb_f64: *mut f64,
        d_rgb_f64: *mut f64,
        out_rgb_u8: *mut u8,
        d_rgb_u8: *mut u8,
        stats: *mut RtStats,
    ) -> c_int;
    pub fn rt_scene_device_count(s: *const c_void) -> c_int;
    pub fn rt_render_device_noalloc(
        s: *mut c_void,
        cam: *const RtCamera,
        depth: c_int,
        mode: c_int,
        tile_first: c_int,
        tile_step: c_int,
        d_rgb_f64: *mut f64,
        d_rgb_u8: *mut u8,
        hip_stream: *mut c_void,
    ) -> c_int;
    pub fn rt_noalloc_max_tiles(depth: c_int) -> i64;
    /// `out`: 32 doubles per ray (layout in include/rt_core.h).
    pub fn rt_trace_rays(s: *mut c_void, n: i64, origins: *const f64, directions: *const f64, depth: c_int,
                         out: *mut f64) -> c_int;
}

/// Error text of the last failed call on this thread.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(rt_last_error()).to_string_lossy().into_owned() }
}

/// Owns an `rt_scene*`.  Leaves must be added in the depth-first order of `World.objects`:
/// the reference's stable sorts (group.rs:69, world.rs:29) resolve equal `t` by that order.
pub struct SceneBuilder {
    raw: *mut c_void,
}

impl SceneBuilder {
    pub fn new() -> Self {
        SceneBuilder { raw: unsafe { rt_scene_create() } }
    }
    fn index(&self, rc: c_int, what: &str) -> i32 {
        if rc < 0 {
            panic!("{what}: {}", last_error()); // the reference panics on every error path too
        }
        rc
    }
    pub fn material(&mut self, m: &RtMaterial) -> i32 {
        let rc = unsafe { rt_scene_add_material(self.raw, m) };
        self.index(rc, "rt_scene_add_material")
    }
    pub fn group(&mut self, inv: [f64; 16], parent: i32) -> i32 {
        let g = RtGroup { inv, parent, reserved: 0 };
        let rc = unsafe { rt_scene_add_group(self.raw, &g) };
        self.index(rc, "rt_scene_add_group")
    }
    pub fn primitive(&mut self, p: &RtPrimitive) -> i32 {
        let rc = unsafe { rt_scene_add_primitive(self.raw, p) };
        self.index(rc, "rt_scene_add_primitive")
    }
    /// `p1..p3`: xyz per triangle in face order; normals empty for flat triangles.
    pub fn mesh(&mut self, group: i32, material: i32, p1: &[f64], p2: &[f64], p3: &[f64],
                n1: &[f64], n2: &[f64], n3: &[f64], smooth: &[u8]) -> i32 {
        assert!(p1.len() % 3 == 0, "p1 must hold xyz per triangle");
        assert!(p2.len() == p1.len() && p3.len() == p1.len(), "p1, p2, p3 must have one length");
        let ntris = (p1.len() / 3) as i64;
        let has_n = !(n1.is_empty() && n2.is_empty() && n3.is_empty());
        if has_n {
            assert!(n1.len() == p1.len() && n2.len() == p1.len() && n3.len() == p1.len(),
                    "n1, n2, n3 must all be empty or as long as p1");
        }
        assert!(smooth.is_empty() || (has_n && smooth.len() as i64 == ntris),
                "smooth needs normals and one byte per triangle");
        let opt = |v: &[f64]| if v.is_empty() { std::ptr::null() } else { v.as_ptr() };
        let m = RtMesh {
            group, material, ntris,
            p1: p1.as_ptr(), p2: p2.as_ptr(), p3: p3.as_ptr(),
            n1: opt(n1), n2: opt(n2), n3: opt(n3),
            smooth: if smooth.is_empty() { std::ptr::null() } else { smooth.as_ptr() },
        };
        let rc = unsafe { rt_scene_add_mesh(self.raw, &m) };
        self.index(rc, "rt_scene_add_mesh")
    }
    pub fn light(&mut self, intensity: [f64; 3], position: [f64; 3]) -> i32 {
        let rc = unsafe { rt_scene_add_light(self.raw, intensity.as_ptr(), position.as_ptr()) };
        self.index(rc, "rt_scene_add_light")
    }
    /// The scene on several GPUs of the node (`thread_count` of camera.rs:83 becomes a list of
    /// devices); `Scene::render_rgb8` then deals the frame's tiles to all of them.
    pub fn commit_multi(self, devices: &[i32], bvh_builder: i32) -> Scene {
        let rc = unsafe { rt_scene_commit_multi(self.raw, devices.as_ptr(), devices.len() as c_int, bvh_builder) };
        if rc != 0 {
            panic!("rt_scene_commit_multi: {}", last_error());
        }
        let raw = self.raw;
        std::mem::forget(self);
        Scene { raw, multi: true }
    }
    /// Builds the BVHs and uploads the scene; `bvh_builder` 0 = host SAH, 1 = device LBVH.
    pub fn commit(self, device: i32, bvh_builder: i32) -> Scene {
        let rc = unsafe { rt_scene_commit(self.raw, device, bvh_builder) };
        if rc != 0 {
            panic!("rt_scene_commit: {}", last_error());
        }
        let raw = self.raw;
        std::mem::forget(self);
        Scene { raw, multi: false }
    }
}
impl Drop for SceneBuilder {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}

/// A committed, immutable scene resident on one GPU.
pub struct Scene {
    raw: *mut c_void,
    multi: bool,
}
impl Scene {
    pub fn info(&self) -> RtSceneInfo {
        let mut i = RtSceneInfo::default();
        if unsafe { rt_scene_get_info(self.raw, &mut i) } != 0 {
            panic!("rt_scene_get_info: {}", last_error());
        }
        i
    }
    /// Counters of one frame (the kernels run with counters enabled).
    pub fn render_rgb8_with_stats(&self, cam: &RtCamera, depth: i32) -> (Vec<u8>, RtStats) {
        let mut rgb = vec![0u8; cam.hsize as usize * cam.vsize as usize * 3];
        let mut st = RtStats::default();
        let rc = unsafe {
            if self.multi {
                rt_render_multi(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, rgb.as_mut_ptr(),
                                std::ptr::null_mut(), &mut st)
            } else {
                rt_render(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, std::ptr::null_mut(),
                          rgb.as_mut_ptr(), &mut st)
            }
        };
        if rc != 0 {
            panic!("render: {}", last_error());
        }
        (rgb, st)
    }
    /// The known-answer interface: 32 doubles per ray (include/rt_core.h).
    pub fn trace_rays(&self, origins: &[[f64; 3]], directions: &[[f64; 3]], depth: i32) -> Vec<[f64; 32]> {
        assert!(origins.len() == directions.len());
        let mut out = vec![[0f64; 32]; origins.len()];
        let rc = unsafe {
            rt_trace_rays(self.raw, origins.len() as i64, origins.as_ptr() as *const f64,
                          directions.as_ptr() as *const f64, depth, out.as_mut_ptr() as *mut f64)
        };
        if rc != 0 {
            panic!("rt_trace_rays: {}", last_error());
        }
        out
    }
    /// One frame as the reference's Canvas holds it (draw/canvas.rs:6-27): `hsize*vsize*3` f64
    /// colour components, row-major, unclamped.
    pub fn render_f64(&self, cam: &RtCamera, depth: i32) -> Vec<f64> {
        let mut rgb = vec![0f64; cam.hsize as usize * cam.vsize as usize * 3];
        let rc = unsafe {
            if self.multi {
                rt_render_multi_f64(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, rgb.as_mut_ptr(),
                                    std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(),
                                    std::ptr::null_mut())
            } else {
                rt_render(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, rgb.as_mut_ptr(),
                          std::ptr::null_mut(), std::ptr::null_mut())
            }
        };
        if rc != 0 {
            panic!("render: {}", last_error());
        }
        rgb
    }
    /// One frame, quantised as Color::clamp does (draw/color.rs:148-155): `hsize*vsize*3` bytes.
    pub fn render_rgb8(&self, cam: &RtCamera, depth: i32) -> Vec<u8> {
        let mut rgb = vec![0u8; cam.hsize as usize * cam.vsize as usize * 3];
        let rc = unsafe {
            if self.multi {
                rt_render_multi(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, rgb.as_mut_ptr(),
                                std::ptr::null_mut(), std::ptr::null_mut())
            } else {
                rt_render(self.raw, cam, depth, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT,
                          std::ptr::null_mut(), rgb.as_mut_ptr(), std::ptr::null_mut())
            }
        };
        if rc != 0 {
            panic!("render: {}", last_error());
        }
        rgb
    }
}
impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.raw) }
    }
}

/// What the reference implements for every shape: append yourself (and your children, in
/// order) to the builder.  `parent` is the index of the enclosing group or -1.
pub trait Flatten {
    fn flatten(&self, b: &mut SceneBuilder, parent: i32);
}
