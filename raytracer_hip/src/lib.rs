//! Safe wrapper over librthip's C ABI (include/rthip.h): the MI355X replacement for the CPU trace
//! loops of `raytracer_headless` (src/main.rs:34-57) and `raytracer_runtime` (src/main.rs:69-94).
//! Method names follow the wgpu renderer's (raytracer_gpu/src/renderer.rs:60,484,562,577).
//!
//! UNCOMPILED in the repository this file ships in (no Rust toolchain there): the declarations
//! mirror include/rthip.h one to one, and that header's functions are what is tested.
use raytracer::{Cube, Material, Object, Scalar, Scene, Sphere, Vec3};
use std::os::raw::c_char;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RthipObject {
    pub kind: i32,     // 0 sphere, 1 cube
    pub material: i32, // 0 Lambertain, 1 Metal, 2 Dielectric
    pub center: [f64; 3],
    pub extent: [f64; 3], // sphere: [radius, 0, 0]; cube: half extents
    pub albedo: [f64; 3],
    pub param: f64, // emission | fuzz | ir
}

#[repr(C)]
pub struct RthipCameraDesc {
    pub origin: [f64; 3],
    pub direction: [f64; 3],
    pub fov: f64,
    pub focus_dist: f64,
    pub defocus_angle: f64,
    pub width: i32,
    pub height: i32,
}

#[repr(C)]
#[derive(Default)]
pub struct RthipRenderParams {
    pub spp_begin: u32,
    pub spp_count: u32,
    pub max_depth: i32,
    pub seed: u32,
    pub sky_mode: i32,
    pub precision: i32,
    pub variant: i32,
    pub part_index: i32,
    pub part_count: i32,
    pub reserved: [i32; 3],
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct RthipStats {
    pub paths: u64,
    pub rays: u64,
    pub box_tests: u64,
    pub sphere_tests: u64,
    pub cube_tests: u64,
    pub kernel_ms: f64,
    pub n_nodes: i32,
    pub n_leaves: i32,
    pub n_prims: i32,
    pub tree_depth: i32,
    pub scene_in_lds: i32,
    pub n_gates: i32,
    pub launch_threads: i32,
    pub launch_grid: i32,
    pub launch_lds_bytes: i32,
    pub waves_per_cu: i32,
}

#[repr(C)]
pub struct RthipCtx {
    _private: [u8; 0],
}

unsafe extern "C" {
    fn rthip_create(device: i32, out: *mut *mut RthipCtx) -> i32;
    fn rthip_destroy(ctx: *mut RthipCtx);
    fn rthip_last_error(ctx: *const RthipCtx) -> *const c_char;
    fn rthip_set_option(ctx: *mut RthipCtx, name: *const c_char, value: f64) -> i32;
    fn rthip_scene_upload(ctx: *mut RthipCtx, objects: *const RthipObject, n: i64) -> i32;
    fn rthip_scene_refit(ctx: *mut RthipCtx, objects: *const RthipObject, n: i64) -> i32;
    fn rthip_camera_set(ctx: *mut RthipCtx, cam: *const RthipCameraDesc) -> i32;
    fn rthip_render_accumulate(ctx: *mut RthipCtx, p: *const RthipRenderParams) -> i32;
    fn rthip_render_accumulate_async(ctx: *mut RthipCtx, p: *const RthipRenderParams) -> i32;
    fn rthip_render_accumulate_noalloc(ctx: *mut RthipCtx, p: *const RthipRenderParams) -> i32;
    fn rthip_synchronize(ctx: *mut RthipCtx) -> i32;
    fn rthip_reset_accumulation(ctx: *mut RthipCtx) -> i32;
    fn rthip_read_rgb8(ctx: *mut RthipCtx, spp_total: u32, rgb: *mut u8) -> i32;
    fn rthip_read_xrgb32(ctx: *mut RthipCtx, spp_total: u32, xrgb: *mut u32) -> i32;
    fn rthip_read_accum(ctx: *mut RthipCtx, rgb: *mut f64) -> i32;
    fn rthip_get_stats(ctx: *mut RthipCtx, out: *mut RthipStats) -> i32;
    fn rthip_accum_download(ctx: *mut RthipCtx, words: *mut u64, n_words: u64) -> i32;
    fn rthip_accum_upload(ctx: *mut RthipCtx, words: *const u64, n_words: u64) -> i32;
    // multi-GPU: one context per GPU, one thread per context
    fn rthip_comm_unique_id(id: *mut u8) -> i32;
    fn rthip_comm_init(ctx: *mut RthipCtx, id: *const u8, rank: i32, world: i32) -> i32;
    fn rthip_merge_reduce(ctx: *mut RthipCtx, root: i32) -> i32;
    fn rthip_merge_wait(ctx: *mut RthipCtx, device_ptr: *const std::ffi::c_void) -> i32;
    fn rthip_merge_local(ctxs: *const *mut RthipCtx, n: i32, root: i32) -> i32;
}

fn material(m: &Material) -> (i32, [f64; 3], f64) {
    // src/materials.rs:4-8
    match *m {
        Material::Lambertain { albedo, emission } => (0, [albedo.x, albedo.y, albedo.z], emission),
        Material::Metal { albedo, fuzz } => (1, [albedo.x, albedo.y, albedo.z], fuzz),
        Material::Dielectric { ir } => (2, [0.0; 3], ir),
    }
}

/// public fields: src/shapes/sphere.rs:5-7
pub fn sphere(s: &Sphere) -> RthipObject {
    let (material, albedo, param) = material(&s.material);
    RthipObject { kind: 0, material, center: [s.center.x, s.center.y, s.center.z], extent: [s.radius, 0.0, 0.0], albedo, param }
}

/// public fields: src/shapes/cube.rs:5-7
pub fn cube(c: &Cube) -> RthipObject {
    let (material, albedo, param) = material(&c.material);
    RthipObject {
        kind: 1,
        material,
        center: [c.center.x, c.center.y, c.center.z],
        extent: [c.half_extend.x, c.half_extend.y, c.half_extend.z],
        albedo,
        param,
    }
}

/// Objects in insertion order (src/scene.rs:80-98): the position in the list is the object id.
pub fn objects(scene: &Scene) -> Vec<RthipObject> {
    scene
        .objects
        .iter()
        .map(|o| match o {
            Object::Sphere(s) => sphere(s),
            Object::Cube(c) => cube(c),
        })
        .collect()
}

fn check(ctx: *const RthipCtx, code: i32) -> Result<(), String> {
    if code == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(rthip_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(format!("rthip status {code}: {msg}"))
}

/// Drop-in for the CPU trace loop.  `!Sync`: one caller thread at a time, like the reference's `&mut` renderers.
pub struct HipRenderer {
    ctx: *mut RthipCtx,
    width: usize,
    height: usize,
    pub spp_done: u32,
    pub max_depth: i32,
    pub seed: u32,
}

impl HipRenderer {
    /// Replaces `BVH::new(scene)` (src/bvh.rs:17): uploads the scene to GPU `device`.
    pub fn new(scene: &Scene, device: i32, max_depth: i32, seed: u32) -> Result<Self, String> {
        let objs = objects(scene);
        let mut ctx = std::ptr::null_mut();
        check(std::ptr::null(), unsafe { rthip_create(device, &mut ctx) })?;
        let r = Self { ctx, width: 0, height: 0, spp_done: 0, max_depth, seed };
        check(ctx, unsafe { rthip_scene_upload(ctx, objs.as_ptr(), objs.len() as i64) })?; // on error `r` drops and frees the context
        Ok(r)
    }

    /// The same objects at new places: the tree on the GPU keeps its shape and has its boxes fitted again
    /// (rthip_scene_refit; the reference has no counterpart, its scenes are static).  Resets the accumulation.
    pub fn refit(&mut self, scene: &Scene) -> Result<(), String> {
        let objs = objects(scene);
        check(self.ctx, unsafe { rthip_scene_refit(self.ctx, objs.as_ptr(), objs.len() as i64) })?;
        self.reset_accumulation()
    }

    /// Tuning or build knob by name (see rthip_set_option in include/rthip.h).
    pub fn set_option(&mut self, name: &str, value: f64) -> Result<(), String> {
        let c = std::ffi::CString::new(name).map_err(|e| e.to_string())?;
        check(self.ctx, unsafe { rthip_set_option(self.ctx, c.as_ptr(), value) })
    }

    /// `Camera::new`'s arguments (the fields of Camera are private, src/camera.rs:5-11).  Resets the accumulation,
    /// as raytracer_runtime does on a camera change (src/main.rs:101-103).
    pub fn update_camera(
        &mut self, origin: Vec3, direction: Vec3, fov: Scalar, focus_dist: Scalar, defocus_angle: Scalar, width: usize, height: usize,
    ) -> Result<(), String> {
        let d = RthipCameraDesc {
            origin: [origin.x, origin.y, origin.z],
            direction: [direction.x, direction.y, direction.z],
            fov,
            focus_dist,
            defocus_angle,
            width: width as i32,
            height: height as i32,
        };
        check(self.ctx, unsafe { rthip_camera_set(self.ctx, &d) })?;
        self.width = width;
        self.height = height;
        self.reset_accumulation()
    }

    pub fn reset_accumulation(&mut self) -> Result<(), String> {
        self.spp_done = 0;
        check(self.ctx, unsafe { rthip_reset_accumulation(self.ctx) })
    }

    fn params(&self, spp_begin: u32, spp_count: u32, part: (i32, i32)) -> RthipRenderParams {
        RthipRenderParams {
            spp_begin,
            spp_count,
            max_depth: self.max_depth,
            seed: self.seed,
            part_index: part.0,
            part_count: part.1,
            ..Default::default()
        }
    }

    /// Adds `spp` samples per pixel: the body of raytracer_headless/src/main.rs:41-54 for the whole image.
    pub fn accumulate(&mut self, spp: u32) -> Result<(), String> {
        let p = self.params(self.spp_done, spp, (0, 1));
        check(self.ctx, unsafe { rthip_render_accumulate(self.ctx, &p) })?;
        self.spp_done += spp;
        Ok(())
    }

    /// Queues a frame without waiting (a viewer's loop); `synchronize` or a read-back waits for it.
    pub fn accumulate_async(&mut self, spp: u32) -> Result<(), String> {
        let p = self.params(self.spp_done, spp, (0, 1));
        check(self.ctx, unsafe { rthip_render_accumulate_async(self.ctx, &p) })?;
        self.spp_done += spp;
        Ok(())
    }

    /// The same, guaranteed to allocate no device memory (HIP graph capture, serving loops).
    pub fn accumulate_noalloc(&mut self, spp: u32) -> Result<(), String> {
        let p = self.params(self.spp_done, spp, (0, 1));
        check(self.ctx, unsafe { rthip_render_accumulate_noalloc(self.ctx, &p) })?;
        self.spp_done += spp;
        Ok(())
    }

    /// This GPU's share of a multi-GPU render: samples [begin, begin + count) of the 8x8 tiles t with
    /// t % part.1 == part.0.  Merge with `merge_reduce` (one process or thread per GPU) or `merge_local`.
    pub fn accumulate_share(&mut self, spp_begin: u32, spp_count: u32, part: (i32, i32)) -> Result<(), String> {
        let p = self.params(spp_begin, spp_count, part);
        check(self.ctx, unsafe { rthip_render_accumulate_async(self.ctx, &p) })
    }

    pub fn synchronize(&mut self) -> Result<(), String> {
        check(self.ctx, unsafe { rthip_synchronize(self.ctx) })
    }

    /// `(accum / spp * 255.0) as u8`, RGB8 row-major (raytracer_headless/src/main.rs:6-12,55,64-69).
    pub fn read_rgb8(&mut self) -> Result<Vec<u8>, String> {
        let mut v = vec![0u8; self.width * self.height * 3];
        check(self.ctx, unsafe { rthip_read_rgb8(self.ctx, self.spp_done.max(1), v.as_mut_ptr()) })?;
        Ok(v)
    }

    /// 0x00RRGGBB per pixel for minifb (raytracer_runtime/src/main.rs:7-17,154-161).
    pub fn read_xrgb32(&mut self) -> Result<Vec<u32>, String> {
        let mut v = vec![0u32; self.width * self.height];
        check(self.ctx, unsafe { rthip_read_xrgb32(self.ctx, self.spp_done.max(1), v.as_mut_ptr()) })?;
        Ok(v)
    }

    /// Sum over the samples of sqrt(colour), the `output` Vec of raytracer_headless/src/main.rs:34 before `/ samples`.
    pub fn read_accum(&mut self) -> Result<Vec<Vec3>, String> {
        let mut v = vec![0f64; self.width * self.height * 3];
        check(self.ctx, unsafe { rthip_read_accum(self.ctx, v.as_mut_ptr()) })?;
        Ok(v.chunks_exact(3).map(|c| Vec3::new(c[0], c[1], c[2])).collect())
    }

    pub fn stats(&mut self) -> Result<RthipStats, String> {
        let mut s = RthipStats::default();
        check(self.ctx, unsafe { rthip_get_stats(self.ctx, &mut s) })?;
        Ok(s)
    }

    /// Checkpoint of a long render: the integer accumulation words (tiles_x * tiles_y * 192 of them).
    pub fn download_words(&mut self) -> Result<Vec<u64>, String> {
        let n = self.width.div_ceil(8) * self.height.div_ceil(8) * 192;
        let mut v = vec![0u64; n];
        check(self.ctx, unsafe { rthip_accum_download(self.ctx, v.as_mut_ptr(), n as u64) })?;
        Ok(v)
    }

    pub fn upload_words(&mut self, words: &[u64], spp_done: u32) -> Result<(), String> {
        check(self.ctx, unsafe { rthip_accum_upload(self.ctx, words.as_ptr(), words.len() as u64) })?;
        self.spp_done = spp_done;
        Ok(())
    }

    /// 128 bytes that rank 0 makes and hands to the other ranks (threads or processes).
    pub fn comm_unique_id() -> Result<[u8; 128], String> {
        let mut id = [0u8; 128];
        check(std::ptr::null(), unsafe { rthip_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }

    /// Collective over all ranks (every rank must be in the call at once).
    pub fn comm_init(&mut self, id: &[u8; 128], rank: i32, world: i32) -> Result<(), String> {
        check(self.ctx, unsafe { rthip_comm_init(self.ctx, id.as_ptr(), rank, world) })
    }

    /// One ncclReduce (root >= 0) or ncclAllReduce (root = -1) of the accumulation words over xGMI.
    pub fn merge_reduce(&mut self, root: i32) -> Result<(), String> {
        check(self.ctx, unsafe { rthip_merge_reduce(self.ctx, root) })
    }

    pub fn merge_wait(&mut self) -> Result<(), String> {
        check(self.ctx, unsafe { rthip_merge_wait(self.ctx, std::ptr::null()) })
    }

    /// Sums the buffers of several renderers of this process into `renderers[root]`'s, without RCCL.
    pub fn merge_local(renderers: &mut [&mut HipRenderer], root: usize) -> Result<(), String> {
        let ctxs: Vec<*mut RthipCtx> = renderers.iter().map(|r| r.ctx).collect();
        check(ctxs[root], unsafe { rthip_merge_local(ctxs.as_ptr(), ctxs.len() as i32, root as i32) })
    }
}

impl Drop for HipRenderer {
    fn drop(&mut self) {
        unsafe { rthip_destroy(self.ctx) }
    }
}
