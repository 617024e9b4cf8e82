// art_renderer.hpp -- C++ host-side mirror of the reference's renderer API over the libart C ABI (include/art.h).
//
// The reference is compiled Rust; its toolchain is absent from this image, so the host layer a Rust maintainer would write over
// the `extern "C"` block of INTEGRATION.md is provided in C++ with the same names, argument meaning and error behaviour:
//   art::Renderer          ~ VulkanTempleRayTracedRenderer (src/vk_renderer/renderer.rs:121-137: new, add_model, prepare_first_frame,
//                            render_frame, camera_mut, lights_mut)
//   art::Camera            ~ VkCamera (vk_camera.rs:128-193)
//   art::Lights, PointLight, SpotLight, DirectionalLight, AreaLight ~ lights.rs
//   art::GltfModelReader   ~ model_reader/gltf_model_reader.rs
// The reference panics on every error (unwrap/expect); here every failed C-ABI call throws art::Panic carrying art_last_error().
#pragma once
#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/art.h"

namespace art {

struct Panic : std::runtime_error { int32_t code; Panic(int32_t c, const std::string &m) : std::runtime_error(m), code(c) {} };
inline void check(int32_t r) { if (r != ART_OK) throw Panic(r, art_last_error()); }
inline void check_glb(int32_t r) { if (r != ART_OK) throw Panic(r, art_glb_last_error()); }

using Vector3 = std::array<float, 3>;
using Vector2 = std::array<float, 2>;
using Matrix3x4 = std::array<float, 12>; // row-major

// ---- lights.rs ------------------------------------------------------------------------------------------------------
struct PointLight {      // lights.rs:95-159
    Vector3 pos, color; float falloff_distance; bool casts_shadows;
    PointLight(Vector3 p, Vector3 c, float f, bool s) : pos(p), color(c), falloff_distance(f), casts_shadows(s) {}
    ArtLight get_light_shader_data() const { ArtLight l; check(art_light_point(pos.data(), color.data(), falloff_distance, casts_shadows, &l)); return l; }
};
struct SpotLight {       // lights.rs:161-243
    Vector3 pos, dir, color; float falloff_distance; Vector2 penumbra_umbra_angles; bool casts_shadows;
    SpotLight(Vector3 p, Vector3 d, Vector3 c, float f, Vector2 a, bool s) : pos(p), dir(d), color(c), falloff_distance(f), penumbra_umbra_angles(a), casts_shadows(s) {}
    ArtLight get_light_shader_data() const { ArtLight l; check(art_light_spot(pos.data(), dir.data(), color.data(), falloff_distance, penumbra_umbra_angles[0], penumbra_umbra_angles[1], casts_shadows, &l)); return l; }
};
struct DirectionalLight { // lights.rs:245-296
    Vector3 dir, color; bool casts_shadows;
    DirectionalLight(Vector3 d, Vector3 c, bool s) : dir(d), color(c), casts_shadows(s) {}
    ArtLight get_light_shader_data() const { ArtLight l; check(art_light_directional(dir.data(), color.data(), casts_shadows, &l)); return l; }
};
struct AreaLight {       // lights.rs:298-403
    Vector3 pos, pos2, pos3; bool invert_normal; Vector3 color; float falloff_distance; Vector2 penumbra_umbra_angles; bool casts_shadows;
    AreaLight(Vector3 p, Vector3 p2, Vector3 p3, bool inv, Vector3 c, float f, Vector2 a, bool s)
        : pos(p), pos2(p2), pos3(p3), invert_normal(inv), color(c), falloff_distance(f), penumbra_umbra_angles(a), casts_shadows(s) {}
    ArtLight get_light_shader_data() const { ArtLight l; check(art_light_area(pos.data(), pos2.data(), pos3.data(), invert_normal, color.data(), falloff_distance, penumbra_umbra_angles[0], penumbra_umbra_angles[1], casts_shadows, &l)); return l; }
};
class Lights {           // lights.rs:4-67
    std::vector<PointLight> point_; std::vector<SpotLight> spot_; std::vector<DirectionalLight> directional_; std::vector<AreaLight> area_;
public:
    std::vector<PointLight> &get_point_lights_mut() { return point_; }
    std::vector<SpotLight> &get_spot_lights_mut() { return spot_; }
    std::vector<DirectionalLight> &get_directional_lights_mut() { return directional_; }
    std::vector<AreaLight> &get_area_lights_mut() { return area_; }
    size_t get_lights_count() const { return point_.size() + spot_.size() + directional_.size() + area_.size(); }
    // order point, spot, directional, area; every light gets its own slot (the reference's lights.rs:29-46 reuses one slot per kind)
    std::vector<ArtLight> copy_lights_shader_data() const {
        std::vector<ArtLight> out;
        for (auto &l : point_) out.push_back(l.get_light_shader_data());
        for (auto &l : spot_) out.push_back(l.get_light_shader_data());
        for (auto &l : directional_) out.push_back(l.get_light_shader_data());
        for (auto &l : area_) out.push_back(l.get_light_shader_data());
        return out;
    }
};

// ---- vk_camera.rs ---------------------------------------------------------------------------------------------------
class Camera {
    Vector3 pos_, dir_; float aspect_, fovy_, znear_, zfar_; bool needs_update_ = true; ArtCamera block_{};
public:
    Camera(Vector3 pos, Vector3 dir, float aspect, float fovy, float znear, float zfar) : pos_(pos), dir_(dir), aspect_(aspect), fovy_(fovy), znear_(znear), zfar_(zfar) {}
    void set_pos(Vector3 p) { pos_ = p; needs_update_ = true; }
    void set_dir(Vector3 d) { dir_ = d; needs_update_ = true; }   // normalised when the block is built (vk_camera.rs:133-136)
    void set_aspect(float a) { aspect_ = a; needs_update_ = true; }
    void set_fovy(float f) { fovy_ = f; needs_update_ = true; }
    void set_znear(float z) { znear_ = z; needs_update_ = true; }
    void set_zfar(float z) { zfar_ = z; needs_update_ = true; }
    Vector3 pos() const { return pos_; }
    Vector3 dir() const { return dir_; }
    float aspect() const { return aspect_; }
    float fovy() const { return fovy_; }
    const ArtCamera &update_host_buffer() { // vk_camera.rs:104-126
        if (needs_update_) { check(art_camera_from_params(pos_.data(), dir_.data(), aspect_, fovy_, znear_, zfar_, &block_)); needs_update_ = false; }
        return block_;
    }
};

// ---- model_reader/gltf_model_reader.rs ---------------------------------------------------------------------------------
class GltfModelReader {
    ArtGlb *h_ = nullptr;
public:
    enum Coerce { NONE = 0, R8G8B8A8_UNORM = 1, B8G8R8A8_UNORM = 2, B8G8R8_UNORM = 3 };
    static GltfModelReader open(const std::string &file_path, bool normalize_vectors, Coerce coerce_image_to_format) {
        GltfModelReader r; check_glb(art_glb_open(file_path.c_str(), normalize_vectors, (int32_t)coerce_image_to_format, &r.h_)); return r;
    }
    GltfModelReader() = default;
    GltfModelReader(GltfModelReader &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    GltfModelReader(const GltfModelReader &) = delete;
    ~GltfModelReader() { if (h_) art_glb_close(h_); }
    ArtGlb *handle() const { return h_; }
    std::pair<Vector3, float> get_primitives_bounding_sphere() const { Vector3 c; float r; check_glb(art_glb_bounding_sphere(h_, c.data(), &r)); return {c, r}; }
};

// ---- renderer.rs ------------------------------------------------------------------------------------------------------
// ---- model_reader.rs:100-146 + vk_model.rs:280-345: bounding sphere and the residency state machine ----------------------
struct Sphere {
    Vector3 center{0, 0, 0}; float radius = 0;
    float get_distance_from_point(const Vector3 &p) const { // model_reader.rs:124-126
        float dx = center[0] - p[0], dy = center[1] - p[1], dz = center[2] - p[2];
        return std::sqrt(dx * dx + dy * dy + dz * dz) - radius;
    }
    Sphere transform(const Matrix3x4 &m) const {              // model_reader.rs:128-141
        float sc = 0;
        for (int k = 0; k < 3; k++) sc = std::fmax(sc, std::sqrt(m[k] * m[k] + m[4 + k] * m[4 + k] + m[8 + k] * m[8 + k]));
        Sphere o;
        for (int r = 0; r < 3; r++) o.center[r] = m[4 * r] * center[0] + m[4 * r + 1] * center[1] + m[4 * r + 2] * center[2] + m[4 * r + 3];
        o.radius = sc * radius;
        return o;
    }
};
enum class ModelState { Storage, Host, Device };
struct Model { // VkModel: only Device models are instanced in the acceleration structure (renderer.rs:640-651)
    std::vector<uint32_t> primitive_ids; Sphere model_bounding_sphere; ModelState state = ModelState::Host; bool needs_cb_submit = false, instanced = true;
    ArtContext *ctx = nullptr; Matrix3x4 model_matrix{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; Sphere object_sphere; // the reader's sphere, before any model matrix
    // VkModel::set_model_matrix (vk_model.rs:461-466): the instance's object -> world matrix and the sphere that goes with it.  Fixed against the reference, which
    // transforms the sphere it HOLDS (already transformed) by the new matrix (:463-465) and so compounds the matrices of a model that moves every frame; the same for
    // the one call main.rs makes.  The reference rebuilds its TLAS every frame for this (renderer.rs:637-651); libart refits in front of the next frame.
    void set_model_matrix(const Matrix3x4 &m) {
        model_matrix = m;
        model_bounding_sphere = object_sphere.transform(m);
        if (ctx && !primitive_ids.empty()) check(art_scene_set_model_matrix(ctx, primitive_ids.front(), (uint32_t)primitive_ids.size(), m.data())); // a model's ids are consecutive (art_scene_add_glb)
    }
    const Matrix3x4 &get_transform_model_matrix() const { return model_matrix; } // vk_model.rs:358-363
    void update_model_status(const Vector3 &camera_pos) { // vk_model.rs:334-345
        float d = model_bounding_sphere.get_distance_from_point(camera_pos);
        ModelState want = d <= 10.0f ? ModelState::Device : (d <= 20.0f ? ModelState::Host : ModelState::Storage);
        if ((want == ModelState::Device) != (state == ModelState::Device)) needs_cb_submit = true;
        state = want;
    }
    bool needs_command_buffer_submission() const { return needs_cb_submit; }
    void reset_command_buffer_submission_status() { needs_cb_submit = false; }
};

class Renderer {
    ArtContext *ctx_ = nullptr; uint32_t w_, h_; Camera camera_; Lights lights_; std::vector<Model> models_;
public:
    // VulkanTempleRayTracedRenderer::new (renderer.rs:140); camera defaults of renderer.rs:222-231
    Renderer(uint32_t width, uint32_t height, int device = -1, uint32_t frames_in_flight = 1)
        : w_(width), h_(height), camera_({0, 0, 0}, {0, 0, 1}, (float)width / (float)height, 1.57079632679f, 0.1f, 1000.0f) {
        ArtConfig cfg{}; cfg.device = device; cfg.width = width; cfg.height = height; cfg.frames_in_flight = frames_in_flight;
        cfg.flags = ART_FLAG_DYNAMIC_SCENE;   // this host moves its models (set_model_matrix) and switches them in and out by residency: the ring of structure versions is made by the build, not by the first moved frame
        check(art_create(&cfg, &ctx_));
    }
    Renderer(const Renderer &) = delete;
    ~Renderer() { if (ctx_) art_destroy(ctx_); }
    void add_model(const std::string &file_path, const Matrix3x4 &model_matrix) { // renderer.rs:346 -> vk_model.rs:494-528
        GltfModelReader r = GltfModelReader::open(file_path, true, GltfModelReader::B8G8R8A8_UNORM);
        uint32_t first = 0, n = 0;
        check_glb(art_scene_add_glb(ctx_, r.handle(), model_matrix.data(), &first, &n));
        Model m; m.ctx = ctx_; m.model_matrix = model_matrix;
        for (uint32_t i = 0; i < n; i++) m.primitive_ids.push_back(first + i);
        auto cs = r.get_primitives_bounding_sphere();                      // vk_model.rs:501, then set_model_matrix (:461-466)
        Sphere sp; sp.center = cs.first; sp.radius = cs.second;
        m.object_sphere = sp;
        m.model_bounding_sphere = sp.transform(model_matrix);
        models_.push_back(m);
    }
    std::vector<Model> &models_mut() { return models_; }
    // renderer.rs:637-651: residency by camera distance; rebuilds the acceleration structure over the Device models when the set changed
    bool update_models_status(bool build = true) {
        bool changed = false;
        for (Model &m : models_) {
            m.update_model_status(camera_.pos());
            m.reset_command_buffer_submission_status();
            bool dev = m.state == ModelState::Device;
            if (dev != m.instanced) { m.instanced = dev; for (uint32_t id : m.primitive_ids) check(art_scene_set_primitive_enabled(ctx_, id, dev ? 1 : 0)); changed = true; }
        }
        if (changed && build && art_scene_needs_build(ctx_) != 0) check(art_scene_build(ctx_)); // (part of the last build: in and out by the next frame's refit)
        return changed && build;
    }
    void add_primitive(const ArtVertex *v, uint32_t nv, const void *idx, uint32_t n_idx, uint32_t idx_bytes, const uint8_t *rgba8, uint32_t tw, uint32_t th, const Matrix3x4 &m) {
        check(art_scene_add_primitive(ctx_, v, nv, idx, n_idx, idx_bytes, rgba8, tw, th, m.data(), nullptr));
    }
    void prepare_first_frame() { update_models_status(false); check(art_scene_build(ctx_)); } // renderer.rs:356
    Camera &camera_mut() { return camera_; }                              // renderer.rs:515
    Lights &lights_mut() { return lights_; }                              // renderer.rs:519
    void render_frame(bool wait = true) {                                  // renderer.rs:371
        update_models_status();
        check(art_set_camera(ctx_, &camera_.update_host_buffer()));
        std::vector<ArtLight> ls = lights_.copy_lights_shader_data();
        check(art_set_lights(ctx_, ls.data(), (uint32_t)ls.size()));
        check(art_trace(ctx_));
        if (wait) check(art_sync(ctx_));
    }
    void compute_ao(uint32_t spp = 16, float radius = 0.2f * 1.457f) { check(art_trace_ao(ctx_, spp, radius)); } // ao_layer.compute_ao, renderer.rs:688
    void resize(uint32_t w, uint32_t h) { check(art_resize(ctx_, w, h)); w_ = w; h_ = h; camera_.set_aspect((float)w / (float)h); } // renderer.rs:523-564
    std::vector<float> color_output() { std::vector<float> o((size_t)w_ * h_ * 4); check(art_read_color(ctx_, o.data(), o.size() * 4)); return o; }
    std::vector<float> depth_output() { std::vector<float> o((size_t)w_ * h_); check(art_read_depth(ctx_, o.data(), o.size() * 4)); return o; }
    std::vector<float> normal_output() { std::vector<float> o((size_t)w_ * h_ * 4); check(art_read_normal(ctx_, o.data(), o.size() * 4)); return o; }
    std::vector<uint32_t> ao_output() { std::vector<uint32_t> o((size_t)w_ * h_); check(art_read_ao(ctx_, o.data(), o.size() * 4)); return o; }
    ArtStats stats() { ArtStats s; check(art_get_stats(ctx_, &s)); return s; }
    ArtContext *handle() const { return ctx_; }
};

} // namespace art
