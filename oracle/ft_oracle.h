/*
 * ft_oracle.h — C ABI of oracle/libft_oracle.so.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a CPU restatement (C++17, IEEE double, no FMA contraction)
 * of the reference's per-pixel render loop, written from the F# source of antonburger/FuncTracer.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (libfunctracer_hip.so) never links or calls it.
 *
 * The builder functions have the same shape as the ft_sg_ / ft_scene_ functions of
 * include/functracer_hip.h (so one scene description can be lowered into either library), but
 * the implementation behind them keeps the reference's structure: a tree of geometry closures
 * with nested per-level transforms (Scene.fs:67-104), all-hits sequences, stable sorts and the
 * literal recursion of Shading.getColourForRay (Shading.fs:131-139).
 *
 * Pinning: checked in tests/test_oracle_golden.py against every known-answer vector the
 * reference's own tests hold for this path (FuncTracer.Tests/Geometry/BoundingBox.fs:11-27,
 * Triangle.Tests.fs:12-54, Sphere.fs:18-30).  Everything else on the path has no reference
 * test: those parts are "parity unpinned" beyond the source text and are
 * pinned by hand-derived known answers in tests/golden/known_answers.json.
 */
#ifndef FT_ORACLE_H
#define FT_ORACLE_H
#include <stdint.h>
#include "../include/functracer_hip.h" /* shared plain structs: ft_transform, ft_material, ft_camera, ft_rect */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fto_context fto_context;

typedef struct fto_stats {
    uint64_t rays_primary, rays_shadow, rays_reflect, rays_traced;
    double wall_ms;
    int32_t threads;
    int32_t _pad;
} fto_stats;

int32_t fto_create(fto_context** out);
void    fto_destroy(fto_context* ctx);
const char* fto_last_error(const fto_context* ctx);

ft_node fto_sg_primitive(fto_context* ctx, int32_t kind);
ft_node fto_sg_triangle(fto_context* ctx, const double v[9]);
ft_node fto_sg_bsp_mesh(fto_context* ctx, int32_t depth, const double* tris, int64_t n_tris);
ft_node fto_sg_transform(fto_context* ctx, const ft_transform* ts, int32_t n, ft_node child);
ft_node fto_sg_material(fto_context* ctx, const ft_material* m, ft_node child);
ft_node fto_sg_hue_shift(fto_context* ctx, double angle, ft_node child);
ft_node fto_sg_ignore_light(fto_context* ctx, ft_node child);
ft_node fto_sg_group(fto_context* ctx, const ft_node* children, int32_t n);
ft_node fto_sg_csg(fto_context* ctx, int32_t op, ft_node a, ft_node b);
ft_node fto_sg_texture_grid(fto_context* ctx, const double colour_a[3], const double colour_b[3],
                            const double* uv_ops, int32_t n_uv_ops, ft_node child);
ft_node fto_sg_texture_image(fto_context* ctx, const uint8_t* rgb24, int32_t width, int32_t height,
                             const double* uv_ops, int32_t n_uv_ops, ft_node child);

int32_t fto_scene_clear(fto_context* ctx);
int32_t fto_scene_set_objects(fto_context* ctx, ft_node root);
int32_t fto_scene_add_directional(fto_context* ctx, const double dir[3], const double colour[3]);
int32_t fto_scene_add_soft_directional(fto_context* ctx, const double dir[3], int32_t samples,
                                       double scatter_rad, const double colour[3]);
int32_t fto_scene_add_positional(fto_context* ctx, const double pos[3], const double falloff[3],
                                 const double colour[3]);
int32_t fto_scene_commit(fto_context* ctx);

/* Program.fs:54-64 on the CPU.  threads <= 0 ⇒ all hardware threads; rays are handed out in
 * chunks of 1000 with output order preserved, as Shading.fs:143-146 does. */
int32_t fto_render(fto_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp,
                   const double* jitter_xy, int32_t max_depth, uint64_t seed,
                   const ft_rect* tiles, int32_t n_tiles, double* out_rgb, int32_t threads,
                   fto_stats* stats);

/* Scene.intersectScene (Scene.fs:118) for single rays (no slightOffset). */
int32_t fto_closest(fto_context* ctx, const double* origins, const double* dirs, int64_t n,
                    int32_t* hit, double* t, double* p, double* nrm, double* colour);
/* All hits of the scene geometry along each ray, in sequence order (unsorted):
 * out arrays hold at most cap hits per ray; counts[i] = number found (may exceed cap). */
int32_t fto_all_hits(fto_context* ctx, const double* origins, const double* dirs, int64_t n,
                     int32_t cap, int32_t* counts, double* t, double* p, double* nrm);
int32_t fto_blocked(fto_context* ctx, const double* origins, const double* dirs,
                    const double* max_dist, int64_t n, int32_t* blocked);
/* Shading.getColourForRay (Shading.fs:131-139) for single rays. */
int32_t fto_colour_for_ray(fto_context* ctx, const double* origins, const double* dirs, int64_t n,
                           int32_t max_depth, double* rgb);
/* ImagePlane.rayThroughPixel (Image.fs:83-89): ray for pixel (px,py) with a jitter offset. */
int32_t fto_ray_through_pixel(const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t px, int32_t py,
                              double jx, double jy, double o[3], double d[3]);
/* ImagePlane.create constants (Image.fs:67-81): out = {pixelWidth, pixelHeight, topLeftX, topLeftY, i[3], j[3], k[3]}. */
int32_t fto_image_plane(const ft_camera* cam, int32_t res_h, int32_t res_v, double out[13]);

/* Unit-level entry points for the reference's own golden vectors. */
int32_t fto_aabb_intersects(const double bmin[3], const double bmax[3], const double o[3], const double d[3]); /* BoundingBox.fs:32-58 */
/* Triangle.slice (Triangle.fs:24-41) with Plane(p0,n): writes triangles (9 doubles each), returns counts. */
int32_t fto_slice(const double p0[3], const double n[3], const double tri[9],
                  double* above, int32_t* n_above, double* below, int32_t* n_below);
/* BspMesh.compile statistics (BspMesh.fs:51-65, 78-86): out = {maxDepth, leaves, leafTriangles}. */
int32_t fto_bsp_stats(const double* tris, int64_t n_tris, int32_t depth, int64_t out[3]);
/* Math.quadratic (Math.fs:4-10): returns number of roots (0 or 2). */
int32_t fto_quadratic(double a, double b, double c, double roots[2]);
int32_t fto_quantise_rgba8(const double* rgb, int64_t n_pixels, uint8_t* out_rgba); /* Image.fs:36 */

#ifdef __cplusplus
}
#endif
#endif
