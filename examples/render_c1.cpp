// render_c1.cpp -- the 3-sphere scene of BASELINE.json configs[0] through the C++ host API
// (include/rtx.hpp), written the way a user of the reference crate writes it (cf. the doc
// example at src/raytracing/scene.rs:107-111).  Usage: render_c1 W H SPP out.f64 [out.rgb8]
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "rtx.hpp"

using namespace rtx;
using rtx::object::Material;
using rtx::object::Object;
using rtx::object::sphere::Sphere;

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s W H SPP out.f64 [out.rgb8]\n", argv[0]); return 2; }
    const std::size_t w = std::strtoul(argv[1], nullptr, 10), h = std::strtoul(argv[2], nullptr, 10);
    const std::size_t spp = std::strtoul(argv[3], nullptr, 10);
    try {
        Scene scene(Config().with_rays_per_pixel(spp), Camera(Vector3(0, 0, 0), Vector3(1, 0, 0), std::acos(-1.0) / 2));
        scene.add_object(Object(Sphere(Vector3(6, 0, 8), 5), Material::light(Vector3(1, 1, 1))));
        scene.add_object(Object(Sphere(Vector3(6, -1.2, 0), 1), Material::colored(Vector3(0.8, 0.2, 0.2))));
        scene.add_object(Object(Sphere(Vector3(6, 1.2, 0), 1), Material(Vector3(0.9, 0.9, 0.9), Vector3::zeros(), 0.1)));
        auto img = scene.render(w, h);
        std::FILE *f = std::fopen(argv[4], "wb");
        if (!f) return 3;
        for (auto &row : img)
            for (auto &px : row) { double c[3] = {px.x, px.y, px.z}; std::fwrite(c, sizeof(double), 3, f); }
        std::fclose(f);
        if (argc > 5) {
            auto q = scene.render_to_image(w, h);
            std::FILE *g = std::fopen(argv[5], "wb");
            if (!g) return 3;
            std::fwrite(q.data(), 1, q.size(), g);
            std::fclose(g);
        }
    } catch (const rtx::Panic &p) {
        std::fprintf(stderr, "rtx panic (status %d): %s\n", (int)p.status, p.what());
        return 1;
    }
    return 0;
}
