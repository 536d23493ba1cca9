// orbit_camera.cpp -- a resident scene rendered from several camera positions without re-upload,
// through the C++ host API (include/rtx.hpp).  The frames land in device memory; this example copies them back with
// the HIP runtime and writes them as raw f64.  Usage: orbit_camera W H SPP N_FRAMES out.f64
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "rtx.hpp"

using namespace rtx;
using rtx::object::Material;
using rtx::object::Object;
using rtx::object::sphere::Sphere;
using rtx::object::triangle::Triangle;

int main(int argc, char **argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s W H SPP N_FRAMES out.f64\n", argv[0]); return 2; }
    const std::size_t w = std::strtoul(argv[1], nullptr, 10), h = std::strtoul(argv[2], nullptr, 10);
    const std::size_t spp = std::strtoul(argv[3], nullptr, 10), frames = std::strtoul(argv[4], nullptr, 10);
    const double fov = std::acos(-1.0) / 2;
    try {
        Scene scene(Config().with_rays_per_pixel(spp), Camera(Vector3(0, 0, 0), Vector3(1, 0, 0), fov));
        scene.add_object(Object(Sphere(Vector3(6, 0, 8), 5), Material::light(Vector3(1, 1, 1))));
        scene.add_object(Object(Sphere(Vector3(6, -1.2, 0), 1), Material::colored(Vector3(0.8, 0.2, 0.2))));
        scene.add_object(Object(Sphere(Vector3(6, 1.2, 0), 1), Material(Vector3(0.9, 0.9, 0.9), Vector3::zeros(), 0.1)));
        scene.add_object(Object(Triangle({Vector3(8, -3, -1), Vector3(8, 3, -1), Vector3(8, 0, 2.5)}), Material::colored(Vector3(0.2, 0.6, 0.9))));
        Scene::Resident resident = scene.upload(0);

        double *d_frame = nullptr;
        const std::size_t n = w * h * 3;
        if (hipMalloc((void **)&d_frame, n * sizeof(double)) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 3; }
        std::vector<double> frame(n);
        std::FILE *f = std::fopen(argv[5], "wb");
        if (!f) return 3;
        for (std::size_t k = 0; k < frames; ++k) {
            // the camera slides along y and keeps looking at the spheres: position and direction as the tests recompute them
            const Vector3 pos(0.25 * (double)k, -1.5 + 0.75 * (double)k, 0.1 * (double)k);
            const Vector3 dir(6.0 - pos.x, 0.0 - pos.y, 0.5 - pos.z);
            resident.set_camera(Camera(pos, dir, fov));
            resident.render(w, h, d_frame);
            if (hipMemcpy(frame.data(), d_frame, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 4;
            std::fwrite(frame.data(), sizeof(double), n, f);
        }
        std::fclose(f);
        (void)hipFree(d_frame);
    } catch (const rtx::Panic &p) {
        std::fprintf(stderr, "rtx panic (status %d): %s\n", (int)p.status, p.what());
        return 1;
    }
    return 0;
}
