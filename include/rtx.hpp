// rtx.hpp -- C++ host API over the C ABI of librtx_hip.so, mirroring the public surface of the
// reference crate `rtx` (src/lib.rs:1-5):
//
//     rtx::math::Vector3                                   src/math/vector.rs
//     rtx::Camera                                          src/raytracing/camera.rs
//     rtx::Config, rtx::Scene                              src/raytracing/scene.rs
//     rtx::object::{Object, Material, CustomShape,
//                   sphere::Sphere, plane::Plane, triangle::Triangle}
//                                                          src/raytracing/object.rs, object/*.rs
//
// Same names, argument meaning and error behaviour: the reference reports failures by panicking
// (scene.rs:168, object.rs:38,50); here a failed render throws rtx::Panic.  The reference is
// Rust and this image has no Rust toolchain, so this header is the compiled-language host side;
// INTEGRATION.md holds the Rust `extern "C"` shim a maintainer would add to the crate itself.
//
// Header-only; link with -lrtx_hip.  Nothing here computes pixels: Scene::render packs
// Scene.objects into RtxObject records and calls rtx_render.
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rtx_hip.h"

namespace rtx {

struct Panic : std::runtime_error {                                 // the reference panics; we throw
    int32_t status;
    Panic(int32_t st, const std::string &what) : std::runtime_error(what), status(st) {}
};

namespace math {

struct Vector3 {                                                    // math/vector.rs:12-20
    double x = 0.0, y = 0.0, z = 0.0;
    Vector3() = default;
    Vector3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}   // Vector3::new, vector.rs:67-69
    static Vector3 zeros() { return {0.0, 0.0, 0.0}; }              // vector.rs:81-83
    static Vector3 ones() { return {1.0, 1.0, 1.0}; }               // vector.rs:47-53
    static Vector3 x_axis() { return {1.0, 0.0, 0.0}; }             // Vector3::x(), vector.rs:55-57
    static Vector3 y_axis() { return {0.0, 1.0, 0.0}; }             // Vector3::y(), vector.rs:59-61
    static Vector3 z_axis() { return {0.0, 0.0, 1.0}; }             // Vector3::z(), vector.rs:63-65
    double dot(const Vector3 &o) const { return x * o.x + y * o.y + z * o.z; }          // vector.rs:85-87
    Vector3 operator-() const { return {-x, -y, -z}; }             // vector.rs:115-121
    Vector3 operator+(const Vector3 &o) const { return {x + o.x, y + o.y, z + o.z}; }   // vector/add.rs:16-24
    Vector3 operator-(const Vector3 &o) const { return {x - o.x, y - o.y, z - o.z}; }   // vector/sub.rs:16-24
    bool operator==(const Vector3 &o) const { return x == o.x && y == o.y && z == o.z; } // derive(PartialEq)
};

}  // namespace math

using math::Vector3;

namespace object {

// What the device can run for a shape.  The reference's trait has only distance()/normal()
// (object.rs:53-76), from which geometry cannot be recovered, so the drop-in adds one provided
// method: primitive().  A user-defined CustomShape that does not override it cannot be rendered
// (there is no CPU fallback) and Scene::render throws.
struct Primitive {
    uint32_t kind;                       // RTX_SPHERE / RTX_PLANE / RTX_TRIANGLE
    std::array<double, 9> geom;
};

struct CustomShape {                                                // object.rs:53-76
    virtual ~CustomShape() = default;
    virtual std::optional<Primitive> primitive() const { return std::nullopt; }
};

namespace sphere {
struct Sphere : CustomShape {                                       // object/sphere.rs:8-17
    Vector3 position;
    double radius;
    Sphere(Vector3 position_, double radius_) : position(position_), radius(radius_) {}
    std::optional<Primitive> primitive() const override
    {
        return Primitive{RTX_SPHERE, {position.x, position.y, position.z, radius, 0, 0, 0, 0, 0}};
    }
};
}  // namespace sphere

namespace plane {
struct Plane : CustomShape {                                        // object/plane.rs:8-17
    Vector3 position, normal;
    Plane(Vector3 position_, Vector3 normal_) : position(position_), normal(normal_) {}
    std::optional<Primitive> primitive() const override
    {
        return Primitive{RTX_PLANE, {position.x, position.y, position.z, normal.x, normal.y, normal.z, 0, 0, 0}};
    }
};
}  // namespace plane

namespace triangle {
struct Triangle : CustomShape {                                     // object/triangle.rs:8-17
    std::array<Vector3, 3> vertices;
    explicit Triangle(std::array<Vector3, 3> vertices_) : vertices(vertices_) {}
    std::optional<Primitive> primitive() const override
    {
        const auto &v = vertices;
        return Primitive{RTX_TRIANGLE, {v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z}};
    }
};
}  // namespace triangle

struct Material {                                                   // object.rs:78-86
    Vector3 base_color, emission_color;
    double roughness;
    Material(Vector3 base, Vector3 emission, double rough)          // Material::new, object.rs:92-94
        : base_color(base), emission_color(emission), roughness(rough) {}
    static Material colored(Vector3 color) { return {color, Vector3::zeros(), 1.0}; }       // object.rs:111-113
    static Material light(Vector3 light_color) { return {Vector3::zeros(), light_color, 1.0}; } // object.rs:130-132
    static Material mirror() { return {Vector3::ones(), Vector3::zeros(), 1.0}; }           // object.rs:133-135
};

struct Object {                                                     // object.rs:9-28
    std::shared_ptr<const CustomShape> shape;                       // Arc<Mutex<dyn CustomShape>>: immutable snapshot here
    Material material;
    template <class T>
    Object(T shape_, Material material_) : shape(std::make_shared<T>(std::move(shape_))), material(material_) {}
};

}  // namespace object

struct Config {                                                     // scene.rs:16-28; Default scene.rs:55-65
    std::size_t rays_per_pixel = 16;
    std::size_t max_bounces = 10;
    double focal_length = 10.0;
    double focal_offset = 1e-4;
    double non_focal_offset = 1e-1;
    uint64_t seed = 42;                                             // build-added
    uint32_t kernel = RTX_KERNEL_AUTO;                              // build-added
    Config with_rays_per_pixel(std::size_t v) const { Config c = *this; c.rays_per_pixel = v; return c; }   // scene.rs:39-41
    Config with_max_bounces(std::size_t v) const { Config c = *this; c.max_bounces = v; return c; }         // scene.rs:42-44
    Config with_focal_length(double v) const { Config c = *this; c.focal_length = v; return c; }            // scene.rs:45-47
    Config with_focal_offset(double v) const { Config c = *this; c.focal_offset = v; return c; }            // scene.rs:48-50
    Config with_non_focal_offset(double v) const { Config c = *this; c.non_focal_offset = v; return c; }    // scene.rs:51-53
    Config with_seed(uint64_t v) const { Config c = *this; c.seed = v; return c; }
    Config with_kernel(uint32_t v) const { Config c = *this; c.kernel = v; return c; }
};

class Camera {                                                      // camera.rs:7-15
public:
    double fov;                                                     // horizontal fov in radians (camera.rs:8)
    Vector3 position;
    Camera(Vector3 position_, Vector3 direction_, double fov_) : fov(fov_), position(position_), direction_(direction_)
    {                                                               // Camera::new, camera.rs:19-28
        derive(direction_);
    }
    Vector3 get_direction() const { return direction_; }            // camera.rs:30-32
    void set_direction(Vector3 direction)                           // camera.rs:35-40
    {
        derive(direction_);           // as the reference: matrices from the OLD direction, then store the new one
        direction_ = direction;
    }
    Vector3 to_cam_space(Vector3 v) const { return mul(c_.to_cam_space, v - position); }        // camera.rs:51-53
    Vector3 to_world_space(Vector3 v) const { return mul(c_.to_world_space, v) + position; }    // camera.rs:55-57
    Vector3 rotate_to_world_space(Vector3 v) const { return mul(c_.to_world_space, v); }        // camera.rs:65-67
    RtxCamera to_c() const
    {
        RtxCamera c = c_;
        c.fov = fov;
        c.position[0] = position.x; c.position[1] = position.y; c.position[2] = position.z;
        c.direction[0] = direction_.x; c.direction[1] = direction_.y; c.direction[2] = direction_.z;
        return c;
    }

private:
    Vector3 direction_;
    RtxCamera c_{};
    void derive(Vector3 d)
    {
        const double p[3] = {position.x, position.y, position.z}, dd[3] = {d.x, d.y, d.z};
        int32_t rc = rtx_camera_new(p, dd, fov, &c_);
        if (rc != RTX_OK) throw Panic(rc, rtx_last_error());
    }
    static Vector3 mul(const double m[9], Vector3 v)                // mat/mul.rs:42-50
    {
        return {v.x * m[0] + v.y * m[1] + v.z * m[2], v.x * m[3] + v.y * m[4] + v.z * m[5],
                v.x * m[6] + v.y * m[7] + v.z * m[8]};
    }
};

class Scene {                                                       // scene.rs:78-85
public:
    std::vector<object::Object> objects;
    Camera camera;
    Config config;

    Scene() : camera(Vector3(0, 0, 0), Vector3(1, 0, 0), 90.0) {}   // Default, scene.rs:86-94 (90f64, as the reference)
    Scene(Config config_, Camera camera_) : camera(std::move(camera_)), config(config_) {}   // Scene::new, scene.rs:112-118
    void add_object(object::Object o) { objects.push_back(std::move(o)); }                  // scene.rs:126-128

    // Scene::render, scene.rs:144-170: img[y][x].  `devices`: the GPUs the frame is partitioned over (interleaved row
    // bands, one gather on devices[0]; rtx_render_devices); empty = device 0.  The pixels do not depend on it.
    std::vector<std::vector<Vector3>> render(std::size_t width, std::size_t height, const std::vector<int32_t> &devices = {}) const
    {
        std::vector<double> flat(width * height * 3);
        std::vector<RtxObject> packed = pack();
        RtxScene sc = to_c(packed);
        int32_t rc = devices.empty() ? rtx_render(&sc, (uint32_t)width, (uint32_t)height, flat.data())
                                     : rtx_render_devices(&sc, (uint32_t)width, (uint32_t)height, devices.data(),
                                                          (uint32_t)devices.size(), flat.data());
        if (rc != RTX_OK) throw Panic(rc, rtx_last_error());
        std::vector<std::vector<Vector3>> img(height, std::vector<Vector3>(width));
        for (std::size_t y = 0; y < height; ++y)
            for (std::size_t x = 0; x < width; ++x) {
                const double *c = &flat[(y * width + x) * 3];
                img[y][x] = Vector3(c[0], c[1], c[2]);
            }
        return img;
    }

    // Scene::render_to_image, scene.rs:172-178: RGB8, row 0 = top of the image
    std::vector<uint8_t> render_to_image(std::size_t width, std::size_t height, const std::vector<int32_t> &devices = {}) const
    {
        std::vector<uint8_t> out(width * height * 3);
        std::vector<RtxObject> packed = pack();
        RtxScene sc = to_c(packed);
        int32_t rc = devices.empty() ? rtx_render_to_image(&sc, (uint32_t)width, (uint32_t)height, out.data())
                                     : rtx_render_to_image_devices(&sc, (uint32_t)width, (uint32_t)height, devices.data(),
                                                                   (uint32_t)devices.size(), out.data());
        if (rc != RTX_OK) throw Panic(rc, rtx_last_error());
        return out;
    }

    std::vector<RtxObject> pack() const
    {
        std::vector<RtxObject> packed(objects.size());
        for (std::size_t i = 0; i < objects.size(); ++i) {
            const auto prim = objects[i].shape->primitive();
            if (!prim)
                throw Panic(RTX_ERR_UNSUPPORTED, "object " + std::to_string(i) +
                                                     ": CustomShape without primitive() cannot run on the GPU (no CPU fallback)");
            RtxObject &o = packed[i];
            o.kind = prim->kind;
            o.reserved = 0;
            for (int k = 0; k < 9; ++k) o.geom[k] = prim->geom[k];
            const object::Material &m = objects[i].material;
            o.base_color[0] = m.base_color.x; o.base_color[1] = m.base_color.y; o.base_color[2] = m.base_color.z;
            o.emission_color[0] = m.emission_color.x; o.emission_color[1] = m.emission_color.y;
            o.emission_color[2] = m.emission_color.z;
            o.roughness = m.roughness;
        }
        return packed;
    }

    class Resident;
    // Row N4 of SURVEY 8f (the use-case of the crate's wgpu path, gpu_state/buffer.rs:69-115): the scene stays on
    // `device`; camera / config changes do not re-upload it.
    Resident upload(int device = 0) const;

private:
    RtxScene to_c(const std::vector<RtxObject> &packed) const
    {
        RtxScene sc{};
        sc.config.rays_per_pixel = config.rays_per_pixel;
        sc.config.max_bounces = config.max_bounces;
        sc.config.focal_length = config.focal_length;
        sc.config.focal_offset = config.focal_offset;
        sc.config.non_focal_offset = config.non_focal_offset;
        sc.config.seed = config.seed;
        sc.config.kernel = config.kernel;
        sc.camera = camera.to_c();
        sc.n_objects = packed.size();
        sc.objects = packed.empty() ? nullptr : packed.data();
        return sc;
    }
};

// A scene resident in the HBM of one GPU (rtx_scene_upload ... rtx_scene_free).  Output goes to DEVICE memory the
// caller owns (h*w*3 doubles for a full frame), on the HIP stream it names; a row band per call is what one rank of
// a multi-GPU job renders.
class Scene::Resident {
public:
    Resident(const Resident &) = delete;
    Resident &operator=(const Resident &) = delete;
    Resident(Resident &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~Resident() { if (h_) rtx_scene_free(h_); }

    void set_camera(const Camera &camera)                              // Camera::set_position / set_direction, camera.rs:35-40
    {
        const RtxCamera c = camera.to_c();
        check(rtx_scene_set_camera(h_, &c));
    }
    void set_config(const Config &config)
    {
        RtxConfig c{};
        c.rays_per_pixel = config.rays_per_pixel; c.max_bounces = config.max_bounces;
        c.focal_length = config.focal_length; c.focal_offset = config.focal_offset; c.non_focal_offset = config.non_focal_offset;
        c.seed = config.seed; c.kernel = config.kernel;
        check(rtx_scene_set_config(h_, &c));
    }
    void add_object(const object::Object &o)                           // Scene::add_object, scene.rs:126-128
    {
        Scene one;
        one.add_object(o);
        const std::vector<RtxObject> packed = one.pack();
        check(rtx_scene_append_objects(h_, packed.data(), packed.size()));
    }
    // rows row_begin, row_begin + row_stride, ... (n_rows of them) of the width x height image -> d_out_rgb[n_rows][width][3]
    RtxStats render_rows(std::size_t width, std::size_t height, std::size_t row_begin, std::size_t row_stride, std::size_t n_rows,
                         double *d_out_rgb, void *hip_stream = nullptr)
    {
        RtxStats st{};
        check(rtx_render_rows(h_, (uint32_t)width, (uint32_t)height, (uint32_t)row_begin, (uint32_t)row_stride, (uint32_t)n_rows,
                              d_out_rgb, hip_stream, &st));
        return st;
    }
    RtxStats render(std::size_t width, std::size_t height, double *d_out_rgb, void *hip_stream = nullptr)
    {
        return render_rows(width, height, 0, 1, height, d_out_rgb, hip_stream);
    }
    // what rank `part` of `n_parts` of a multi-GPU job renders: the blocks of `block_rows` rows part, part + n_parts, ...
    // (8 keeps the tree kernels' 8x8 ray tiles whole) -> d_out_rgb[band_rows(...)][width][3], rows in increasing image order
    static std::size_t band_rows(std::size_t height, std::size_t block_rows, std::size_t part, std::size_t n_parts)
    {
        return rtx_blocks_row_count((uint32_t)height, (uint32_t)block_rows, (uint32_t)part, (uint32_t)n_parts);
    }
    RtxStats render_blocks(std::size_t width, std::size_t height, std::size_t block_rows, std::size_t part, std::size_t n_parts,
                           double *d_out_rgb, void *hip_stream = nullptr)
    {
        RtxStats st{};
        check(rtx_render_blocks(h_, (uint32_t)width, (uint32_t)height, (uint32_t)block_rows, (uint32_t)part, (uint32_t)n_parts,
                                d_out_rgb, hip_stream, &st));
        return st;
    }
    void set_scratch_limit(std::uint64_t bytes) { check(rtx_scene_set_scratch_limit(h_, bytes)); }

private:
    friend class Scene;
    explicit Resident(RtxSceneHandle h) : h_(h) {}
    static void check(int32_t rc) { if (rc != RTX_OK) throw Panic(rc, rtx_last_error()); }
    RtxSceneHandle h_;
};

inline Scene::Resident Scene::upload(int device) const
{
    std::vector<RtxObject> packed = pack();
    RtxScene sc = to_c(packed);
    RtxSceneHandle h = nullptr;
    int32_t rc = rtx_scene_upload(&sc, device, &h);
    if (rc != RTX_OK) throw Panic(rc, rtx_last_error());
    return Resident(h);
}

}  // namespace rtx
