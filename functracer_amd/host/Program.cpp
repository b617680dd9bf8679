// Program.cpp — command-line driver with the flow of Program.fs:51-100: parse the scene,
// create the image plane + rays + shade + blend (one ft_render call on the GPU), write the PNG.
//   functracer <scene-file> [output.png]        (2 args: file output; otherwise PNG to stdout)
// Phase timings go to stderr like the reference's eprintfn calls (Program.fs:53-67).  Unlike the
// reference nothing but the PNG is ever written to stdout (its printfn calls at Program.fs:96-97
// corrupt the PNG stream the GUI reads).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_api.h"

static const fth_builder kHipBuilder = {
    [](void* c, int32_t k) { return ft_sg_primitive((ft_context*)c, k); },
    [](void* c, const double* v) { return ft_sg_triangle((ft_context*)c, v); },
    [](void* c, int32_t d, const double* t, int64_t n) { return ft_sg_bsp_mesh((ft_context*)c, d, t, n); },
    [](void* c, const ft_transform* ts, int32_t n, ft_node ch) { return ft_sg_transform((ft_context*)c, ts, n, ch); },
    [](void* c, const ft_material* m, ft_node ch) { return ft_sg_material((ft_context*)c, m, ch); },
    [](void* c, double a, ft_node ch) { return ft_sg_hue_shift((ft_context*)c, a, ch); },
    [](void* c, ft_node ch) { return ft_sg_ignore_light((ft_context*)c, ch); },
    [](void* c, const ft_node* ch, int32_t n) { return ft_sg_group((ft_context*)c, ch, n); },
    [](void* c, int32_t op, ft_node a, ft_node b) { return ft_sg_csg((ft_context*)c, op, a, b); },
    [](void* c, const double* ca, const double* cb, const double* ops, int32_t n, ft_node ch) { return ft_sg_texture_grid((ft_context*)c, ca, cb, ops, n, ch); },
    [](void* c, const uint8_t* px, int32_t w, int32_t h, const double* ops, int32_t n, ft_node ch) { return ft_sg_texture_image((ft_context*)c, px, w, h, ops, n, ch); },
    [](void* c) { return ft_scene_clear((ft_context*)c); },
    [](void* c, ft_node r) { return ft_scene_set_objects((ft_context*)c, r); },
    [](void* c, const double* d, const double* col) { return ft_scene_add_directional((ft_context*)c, d, col); },
    [](void* c, const double* d, int32_t s, double sc, const double* col) { return ft_scene_add_soft_directional((ft_context*)c, d, s, sc, col); },
    [](void* c, const double* p, const double* f, const double* col) { return ft_scene_add_positional((ft_context*)c, p, f, col); },
    [](void* c) { return ft_scene_commit((ft_context*)c); },
};

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: functracer <scene-file> [output.png]\n"); return 2; }
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count(); };
    std::fprintf(stderr, "Using input file: %s\n", argv[1]);
    char err[1024] = {0};
    fth_scene* scene = fth_parse_scene_file(argv[1], err, sizeof err);
    if (!scene) { std::fprintf(stderr, "%s\n", err); return 1; }                 // readScene: message + exit 1 (Program.fs:10-16)
    std::fprintf(stderr, "Parsed input %lims\n", ms());
    fth_options opt;
    fth_scene_options(scene, &opt);
    if (opt.corner) { std::fprintf(stderr, "samples corner (Image.fs:125-150) is not on the device path yet\n"); return 1; }

    int32_t dev = 0;
    if (const char* e = std::getenv("FT_DEVICE")) dev = std::atoi(e);
    ft_context* ctx = nullptr;
    int32_t rc = ft_create(&dev, 1, &ctx);
    if (rc != FT_OK) { std::fprintf(stderr, "ft_create failed (%d): no usable HIP device; this program has no CPU path\n", rc); return 1; }
    rc = fth_scene_lower(scene, &kHipBuilder, ctx);
    if (rc != FT_OK) { std::fprintf(stderr, "scene upload failed (%d): %s\n", rc, ft_last_error(ctx)); return 1; }
    std::fprintf(stderr, "Geometry created %lims\n", ms());

    uint64_t seed = 20260104ull;
    if (const char* e = std::getenv("FT_SEED")) seed = std::strtoull(e, nullptr, 10);
    std::vector<double> jitter(2 * (size_t)opt.samples);
    fth_jitter_pattern(seed, opt.samples, jitter.data());
    std::vector<double> rgb((size_t)opt.res_h * opt.res_v * 3);
    ft_stats st;
    rc = ft_render(ctx, &opt.camera, opt.res_h, opt.res_v, opt.samples, jitter.data(), 8, seed, nullptr, 0, rgb.data(), &st);
    if (rc != FT_OK) { std::fprintf(stderr, "ft_render failed (%d): %s\n", rc, ft_last_error(ctx)); return 1; }
    std::fprintf(stderr, "Shaded scene %lims (%.3f ms on the GPU, %.1f Mrays/s, %llu rays)\n", ms(), st.kernel_ms,
                 (double)st.rays_traced / (st.kernel_ms * 1e3), (unsigned long long)st.rays_traced);
    std::vector<uint8_t> rgba((size_t)opt.res_h * opt.res_v * 4);
    ft_quantise_rgba8(rgb.data(), (int64_t)opt.res_h * opt.res_v, rgba.data());
    std::fprintf(stderr, "Writing output %lims\n", ms());
    const char* out = argc >= 3 ? argv[2] : "/dev/stdout";
    if (argc >= 3) std::fprintf(stderr, "Using output file: %s\n", argv[2]); else std::fprintf(stderr, "Using standard output\n");
    rc = fth_write_png(out, rgba.data(), opt.res_h, opt.res_v);
    ft_destroy(ctx);
    fth_scene_free(scene);
    std::fprintf(stderr, "Elapsed Time: %lims\n", ms());
    return rc == FT_OK ? 0 : 1;
}
