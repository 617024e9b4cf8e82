// art_renderer.rs -- UNVERIFIED sketch (no Rust toolchain in the build image): the safe wrapper a maintainer of the reference would put over
// art_sys.rs so that main.rs:15-66 keeps its shape.  Method names and panicking error behaviour are the reference's
// (VulkanTempleRayTracedRenderer, src/vk_renderer/renderer.rs:140, :346, :356, :371, :511-521); its verified twins are
// araytracingjourney_amd/host/art_renderer.hpp (C++, examples/host_mirror_demo.cpp) and araytracingjourney_amd/renderer.py (ctypes).
use crate::vk_renderer::art_sys::*;
use std::ffi::{CStr, CString};

fn check(code: i32) {
    // the reference panics on every failure (unwrap / expect); libart returns a code and keeps the message
    if code != ART_OK {
        panic!("libart error {}: {}", code, unsafe { CStr::from_ptr(art_last_error()) }.to_string_lossy());
    }
}

pub struct ArtRayTracedRenderer {
    ctx: *mut ArtContext, // !Send / !Sync like the reference's Rc<RefCell<..>> objects (renderer.rs:122-126): one host thread per context
    pub camera: ArtCamera,
    pub lights: Vec<ArtLight>,
}

impl ArtRayTracedRenderer {
    /// VulkanTempleRayTracedRenderer::new (renderer.rs:140): extent + the FrameData ring depth (renderer.rs:135 keeps 3)
    pub fn new(width: u32, height: u32, frames_in_flight: u32) -> Self {
        let cfg = ArtConfig { device: -1, width, height, morton_bits: 0, shard_rank: 0, shard_count: 1, flags: ART_FLAG_DYNAMIC_SCENE /* models move: art_scene_build also makes the version ring */, frames_in_flight, root_relief: 0 };
        let mut ctx = std::ptr::null_mut();
        check(unsafe { art_create(&cfg, &mut ctx) });
        let (pos, dir) = ([0.0f32; 3], [0.0f32, 0.0, 1.0]); // defaults of renderer.rs:222-231
        let mut camera: ArtCamera = unsafe { std::mem::zeroed() };
        check(unsafe { art_camera_from_params(pos.as_ptr(), dir.as_ptr(), width as f32 / height as f32, std::f32::consts::FRAC_PI_2, 0.1, 1000.0, &mut camera) });
        Self { ctx, camera, lights: Vec::new() }
    }
    /// add_model (renderer.rs:346 -> vk_model.rs:494-528): every primitive of the .glb through the C++ GltfModelReader
    pub fn add_model(&mut self, path: &str, model_matrix_3x4: &[f32; 12]) -> std::ops::Range<u32> {
        let (mut glb, mut first, mut n) = (std::ptr::null_mut(), 0u32, 0u32);
        let p = CString::new(path).unwrap();
        if unsafe { art_glb_open(p.as_ptr(), 1, 2, &mut glb) } != ART_OK {
            panic!("{}", unsafe { CStr::from_ptr(art_glb_last_error()) }.to_string_lossy());
        }
        check(unsafe { art_scene_add_glb(self.ctx, glb, model_matrix_3x4.as_ptr(), &mut first, &mut n) });
        unsafe { art_glb_close(glb) };
        first..first + n
    }
    /// VkModel::set_model_matrix (vk_model.rs:461-466): the model's primitives get a new object -> world matrix; the reference rebuilds its TLAS every frame for
    /// this (renderer.rs:637-651), libart refits on the device in front of the next frame
    pub fn set_model_matrix(&mut self, model: std::ops::Range<u32>, model_matrix_3x4: &[f32; 12]) {
        check(unsafe { art_scene_set_model_matrix(self.ctx, model.start, model.end - model.start, model_matrix_3x4.as_ptr()) });
    }
    /// update_model_status (vk_model.rs:334-345) decided that a model enters or leaves the Device state (renderer.rs:637-651 instances Device models only): a model
    /// that was part of the last build leaves / re-enters by the refit in front of the next frame; one that the build never saw needs the build
    pub fn set_model_resident(&mut self, model: std::ops::Range<u32>, resident: bool) {
        for id in model { check(unsafe { art_scene_set_primitive_enabled(self.ctx, id, resident as i32) }); }
        if unsafe { art_scene_needs_build(self.ctx) } == 1 { check(unsafe { art_scene_build(self.ctx) }); }
    }
    /// prepare_first_frame (renderer.rs:356): uploads + BLAS/TLAS builds
    pub fn prepare_first_frame(&mut self) { check(unsafe { art_scene_build(self.ctx) }); }
    /// render_frame (renderer.rs:371): camera.update_host_buffer, lights.update_host_and_device_buffer, lightning_layer.trace_rays
    pub fn render_frame(&mut self) {
        check(unsafe { art_set_camera(self.ctx, &self.camera) });
        check(unsafe { art_set_lights(self.ctx, self.lights.as_ptr(), self.lights.len() as u32) });
        check(unsafe { art_trace(self.ctx) }); // asynchronous, like a queue submit; wait_for_frame() is the fence (renderer.rs:451-466)
    }
    pub fn compute_ao(&mut self) { check(unsafe { art_trace_ao(self.ctx, 16, 0.2 * 1.457) }); } // ao_layer.compute_ao (renderer.rs:688)
    pub fn present(&mut self) { check(unsafe { art_present(self.ctx) }); }                        // tonemap_layer.present (renderer.rs:566-615)
    pub fn wait_for_frame(&mut self) { check(unsafe { art_sync(self.ctx) }); }
    pub fn resize(&mut self, width: u32, height: u32) { check(unsafe { art_resize(self.ctx, width, height) }); } // renderer.rs:523-564
    /// get_color_output_image (vk_rt_lightning_shadows.rs:161-183): the fp32 RGBA colour of the latest frame
    pub fn read_color(&mut self, width: u32, height: u32) -> Vec<f32> {
        let mut px = vec![0f32; (width * height * 4) as usize];
        check(unsafe { art_read_color(self.ctx, px.as_mut_ptr() as *mut _, px.len() * 4) });
        px
    }
}

impl Drop for ArtRayTracedRenderer {
    fn drop(&mut self) { unsafe { art_destroy(self.ctx) }; }
}
