/*
 * host_api.h — C ABI of libfunctracer_host.so: the host side that, in the reference, is F#
 * (SceneParser.fs, PlyParser.fs, Program.fs).  It parses `.scene` text into the Scene.hpp mirror
 * of Scene.fs and lowers it through a table of builder entry points, so the same parsed scene can
 * be handed to libfunctracer_hip.so (the product) or, in tests, to the CPU oracle.
 */
#ifndef FUNCTRACER_HOST_API_H
#define FUNCTRACER_HOST_API_H
#include <stdint.h>
#include "../../include/functracer_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fth_scene fth_scene;

/* The builder half of include/functracer_hip.h as a table (ctx is passed through untouched). */
typedef struct fth_builder {
    ft_node (*sg_primitive)(void* ctx, int32_t kind);
    ft_node (*sg_triangle)(void* ctx, const double v[9]);
    ft_node (*sg_bsp_mesh)(void* ctx, int32_t depth, const double* tris, int64_t n_tris);
    ft_node (*sg_transform)(void* ctx, const ft_transform* ts, int32_t n, ft_node child);
    ft_node (*sg_material)(void* ctx, const ft_material* m, ft_node child);
    ft_node (*sg_hue_shift)(void* ctx, double angle, ft_node child);
    ft_node (*sg_ignore_light)(void* ctx, ft_node child);
    ft_node (*sg_group)(void* ctx, const ft_node* children, int32_t n);
    ft_node (*sg_csg)(void* ctx, int32_t op, ft_node a, ft_node b);
    ft_node (*sg_texture_grid)(void* ctx, const double ca[3], const double cb[3], const double* uv_ops, int32_t n_uv_ops, ft_node child);
    ft_node (*sg_texture_image)(void* ctx, const uint8_t* rgb24, int32_t width, int32_t height, const double* uv_ops, int32_t n_uv_ops, ft_node child);
    int32_t (*scene_clear)(void* ctx);
    int32_t (*scene_set_objects)(void* ctx, ft_node root);
    int32_t (*scene_add_directional)(void* ctx, const double dir[3], const double colour[3]);
    int32_t (*scene_add_soft_directional)(void* ctx, const double dir[3], int32_t samples, double scatter_rad, const double colour[3]);
    int32_t (*scene_add_positional)(void* ctx, const double pos[3], const double falloff[3], const double colour[3]);
    int32_t (*scene_commit)(void* ctx);
} fth_builder;

typedef struct fth_options {        /* Scene.SceneOptions (Scene.fs:56-65) */
    ft_camera camera;
    int32_t res_h, res_v;           /* Image.Resolution */
    int32_t samples;                /* JitteredSampling.strategy n */
    int32_t corner;                 /* CornerSampling.strategy */
} fth_options;

/* SceneParser.parse (SceneParser.fs:360-366).  On failure returns NULL and writes the message to err. */
fth_scene* fth_parse_scene(const char* text, const char* base_dir, char* err, int32_t err_len);
fth_scene* fth_parse_scene_file(const char* path, char* err, int32_t err_len);
void fth_scene_free(fth_scene* s);
int32_t fth_scene_options(const fth_scene* s, fth_options* out);
int32_t fth_scene_counts(const fth_scene* s, int32_t* n_top_level_objects, int32_t* n_lights);
/* Lower the parsed scene: scene_clear, nodes bottom-up, lights in file order, scene_set_objects,
 * scene_commit.  Returns the first negative status a builder call gives, else 0. */
int32_t fth_scene_lower(const fth_scene* s, const fth_builder* b, void* ctx);
/* Parsers.pcolour (SceneParser.fs:85-87): returns 0 and rgb, or -1. */
int32_t fth_parse_colour(const char* text, double rgb[3]);
/* PlyParser.parse: returns triangle count (>= 0) and, when out != NULL, up to cap triangles (9 doubles each). */
int64_t fth_parse_ply(const char* text, double* out, int64_t cap, char* err, int32_t err_len);

/* Image.Load<Rgb24> for local PNG / PPM / PGM files (Textures/Image.fs:21-26).  Returns width*height*3 (bytes needed) and the
 * size; fills out_rgb24 when cap is large enough.  Negative status + message on failure. */
int64_t fth_load_image(const char* path, int32_t* width, int32_t* height, uint8_t* out_rgb24, int64_t cap, char* err, int32_t err_len);

/* Jitter.pattern random Jitter.circle n (Jitter.fs:15-24) on a documented counter-based stream
 * (splitmix64 keyed by seed; uniform = 2*u - 1 with u = top 53 bits / 2^53) in place of the
 * reference's unseeded System.Random (Image.fs:101). */
int32_t fth_jitter_pattern(uint64_t seed, int32_t n, double* out_xy);

/* Image.write (Image.fs:35-44): RGBA8 PNG (stored deflate blocks); pixels row 0 = top. */
int32_t fth_write_png(const char* path, const uint8_t* rgba, int32_t width, int32_t height);

#ifdef __cplusplus
}
#endif
#endif
