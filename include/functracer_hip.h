/*
 * functracer_hip.h — C ABI of libfunctracer_hip.so, the MI355X (gfx950) replacement for
 * FuncTracer's per-pixel render loop.
 *
 * The reference (antonburger/FuncTracer, F#) has no FFI seam of its own.  The seam this
 * library fills is the module surface of Scene.fs + Image.fs as consumed by SceneParser.fs
 * and Program.fs:
 *
 *   - the ft_sg_* / ft_scene_* builder mirrors the constructors of Scene.Primitive
 *     (Scene.fs:8-18), Scene.SceneGraph / SceneFunction (Scene.fs:33-46), Transform
 *     (Transform.fs:25-38), Ray.Material (Ray.fs:4-10), Light (Light.fs:19-26) and
 *     Scene.Scene (Scene.fs:107-110);
 *   - ft_render replaces, in one call, what Program.fs:54-64 does on the CPU:
 *     ImagePlane.create (Image.fs:67-81), JitteredSampling.generateRays (Image.fs:100-110),
 *     Shading.shade (Shading.fs:141-147) and JitteredSampling.blendPixels (Image.fs:112-116);
 *   - ft_quantise_rgba8 is Image.write's toByte (Image.fs:36).
 *
 * Conventions: plain C, no exceptions cross the boundary.  Functions returning int32_t return
 * FT_OK (0) or a negative ft_status; functions returning ft_node return a non-negative handle
 * or a negative ft_status.  ft_last_error(ctx) gives a UTF-8 message owned by the context.
 * All arrays are caller-owned and only read for the duration of the call.  A context is not
 * re-entrant (one call at a time); different contexts may be used from different threads.
 * All floating point is IEEE double, as in the reference (F# float).
 *
 * There is NO CPU fallback: ft_create fails with FT_ERR_NO_DEVICE when no HIP device is
 * usable.  The CPU restatement under oracle/ is test infrastructure and is never linked here.
 */
#ifndef FUNCTRACER_HIP_H
#define FUNCTRACER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FT_ABI_VERSION 2

typedef struct ft_context ft_context;
typedef int32_t ft_node;

typedef enum ft_status {
    FT_OK = 0,
    FT_ERR_INVALID = -1,      /* bad argument / bad handle                                   */
    FT_ERR_NO_DEVICE = -2,    /* no usable HIP device (there is no CPU fallback)            */
    FT_ERR_HIP = -3,          /* a HIP runtime call failed; see ft_last_error               */
    FT_ERR_UNSUPPORTED = -4,  /* part of the Scene.fs surface not on the device path yet     */
    FT_ERR_STATE = -5,        /* call order violated (e.g. render before commit)            */
    FT_ERR_OVERFLOW = -6,     /* a per-ray CSG hit list exceeded its capacity (never silent) */
    FT_ERR_BUILD = -7         /* BSP build failed (degenerate edge, see Triangle.fs:8-10)    */
} ft_status;

/* Scene.Primitive without payload, in the order of Scene.fs:10-17. */
typedef enum ft_primitive_kind {
    FT_PRIM_CIRCLE = 0,         /* Cylinder.circle        Cylinder.fs:22    */
    FT_PRIM_SQUARE = 1,         /* Cube.square            Cube.fs:9-15      */
    FT_PRIM_CUBE = 2,           /* Cube.cube              Cube.fs:17-25     */
    FT_PRIM_SPHERE = 3,         /* Sphere.sphere          Sphere.fs:11-21   */
    FT_PRIM_PLANE = 4,          /* Plane.plane            Plane.fs:28-33    */
    FT_PRIM_CONE = 5,           /* Cone.cone              Cone.fs:7-27      */
    FT_PRIM_SOLID_CYLINDER = 6, /* Cylinder.solidCylinder Cylinder.fs:25-29 */
    FT_PRIM_CYLINDER = 7        /* Cylinder.cylinder      Cylinder.fs:8-20  */
} ft_primitive_kind;

/* Csg.union / intersect / subtract / exclude, Csg.fs:96-99, Scene.fs:36-39. */
typedef enum ft_csg_op { FT_CSG_UNION = 0, FT_CSG_INTERSECT = 1, FT_CSG_SUBTRACT = 2, FT_CSG_EXCLUDE = 3 } ft_csg_op;

/* One basic Transform.Transform (Transform.fs:25-29).  A list of them is a Composed
 * (Transform.fs:30, 41-45): first listed is applied first (Transform.fs:70-71). */
typedef enum ft_transform_kind { FT_TRANSLATE = 0, FT_SCALE = 1, FT_ROTATE = 2 } ft_transform_kind;
typedef struct ft_transform {
    int32_t kind;     /* ft_transform_kind                                                   */
    int32_t _pad;
    double v[3];      /* translation vector | scale factors | rotation axis (normalised by the
                         library exactly as Transform.rotate does, Transform.fs:37-38)       */
    double angle;     /* radians, FT_ROTATE only                                             */
} ft_transform;

/* Ray.Material (Ray.fs:4-10). */
typedef struct ft_material {
    double colour[3];
    double roughness;
    double reflectance;
    double shineyness;
    int32_t apply_lighting; /* bool */
    int32_t _pad;
} ft_material;

/* Image.Camera (Image.fs:10-17).  focus = depth of field (Image.fs:91-94); the reference draws its
 * direction jitter from an unseeded System.Random, here it comes from the seeded stream keyed by
 * ft_render's `seed`. */
typedef struct ft_camera {
    double o[3];
    double look_at[3];
    double up[3];
    double fov_y;        /* radians */
    double aspect_ratio;
    int32_t has_focus;
    int32_t _pad;
    double focal_length;
    double aperture_angular_size; /* radians */
} ft_camera;

typedef struct ft_rect { int32_t x0, y0, w, h; } ft_rect;

/* Counters and timings of one ft_render call. */
typedef struct ft_stats {
    uint64_t rays_primary;   /* W*H*spp over the rendered tiles                                */
    uint64_t rays_shadow;    /* shadow rays actually traced                                    */
    uint64_t rays_reflect;   /* reflection rays actually traced                                */
    uint64_t rays_traced;    /* rays the device traced: rays_primary - rays_primary_culled + rays_shadow + rays_reflect */
    double   rays_reference_equivalent; /* what the F# recursion would trace (Shading.fs:109-139):
                                L shadow rays per hit and L reflection rays per reflective hit  */
    uint64_t hits_primary;   /* primary rays that hit something                                */
    uint64_t csg_overflow;   /* rays whose CSG hit list overflowed (render then fails)         */
    double   kernel_ms;      /* HIP-event time over all kernels of the call, on the library's stream */
    double   wall_ms;        /* host wall time of the call incl. copies                        */
    double   trace_kernel_ms;/* HIP-event time of the closest-hit + shade/shadow kernels only  */
    uint64_t algorithmic_bytes; /* bytes the pipeline has to move for this frame by construction (roofline) */
    int32_t  n_launches;
    int32_t  n_chunks;
    uint64_t hits_total;     /* hits shaded over all bounces                                   */
    uint64_t algorithmic_bytes_closest; /* unused since ABI 2 (always 0)                       */
    uint64_t algorithmic_bytes_shade;   /* the k_bounce share (bounces >= 1)                   */
    uint64_t rays_tail;      /* unused since ABI 2 (always 0): kept so the layout of ABI 1 callers' struct prefix stays */
    uint64_t rays_primary_culled; /* primary rays (part of rays_primary) resolved as misses per 64-pixel block: the block's ray
                                   * bundle cannot reach any object, so they were never generated one by one          */
    uint64_t algorithmic_bytes_primary; /* the k_primary (fused bounce 0) share of algorithmic_bytes                          */
    uint64_t rays_shadow_primary;  /* the part of rays_shadow cast by hits of primary rays (traced inside k_primary)          */
    uint64_t rays_reflect_primary; /* the part of rays_reflect spawned by hits of primary rays                                */
} ft_stats;

/* ---- context ---------------------------------------------------------------------------- */
int32_t ft_abi_version(void);
/* device_ids: HIP device ordinals; n_devices must be >= 1 (0 ⇒ FT_ERR_NO_DEVICE: no CPU path).  With several
 * ordinals the scene is replicated on each device and every ft_render splits its region into 8-row bands dealt
 * round-robin over them, one host thread per device, no exchange between devices: every device copies its bands (whole rows of
 * the frame) straight into the caller's buffer. */
int32_t ft_create(const int32_t* device_ids, int32_t n_devices, ft_context** out);
void    ft_destroy(ft_context* ctx);
const char* ft_last_error(const ft_context* ctx);
/* Tunables: "chunk_samples" (samples in flight per launch; default 16 Mi, frames whose pixel blocks are classified use twice that), "csg_mesh_capacity" (hit-list entries a mesh may add under
 * CSG; default 32), "csg_auto_grow" (default 1: a blocking ft_render (and the ft_debug_* ray queries) whose hit lists overflow doubles that capacity, re-commits and renders the
 * frame again instead of returning FT_ERR_OVERFLOW; the error remains for lists that stop fitting in the LDS and for ft_render_enqueue), "timing" (HIP events recorded inside ft_render: 0 around the frame only, 1 = default: also around
 * k_primary and the k_bounce levels, 2 around every stage; each bracketed boundary costs about 6 us of stream time), "classify_pixels" (default 1: 64-pixel blocks whose ray bundle
 * cannot reach any object are finished before any ray is generated; the bundle is bounded from the jitter pattern handed to ft_render, whatever its range), "level_hint" (default 1: a frame launches as many levels of the reflection tree as the
 * previous frame of the same scene, size and samples had rays in, plus one, whose rays are followed to the end inside the launch; 0: always max_depth levels), "follow_below" (levels in which that
 * previous frame had no more rays than this are not worth a launch and are followed as well; -1 = default: two rays per SIMD of the device; 0: every level that had a ray), "mesh_unclipped_bvh" (non-default fast mode: ignore bspMesh depth, BVH over the original triangles; pixels may
 * differ from the reference-shaped clipped BSP in the last bits), "bvh_builder" (who builds the exact BVH of top-level-Leaf meshes at commit - 0: the host, a swept
 * surface-area split (the best tree, a slow build: 160 ms for 70 K triangles); 1: the device, a linear BVH (1 ms, traces ~9 % slower); 3: the device, a binned surface-area tree
 * over the Morton order (5.5 ms, traces like the host's or better); 2 = default: the host below 4096 triangles, the device's surface-area tree from there on),
 * "classify_ahead" / "resolve_aside" / "zero_fill_skip" (1 = default: what a stream of queued frames does that a single frame cannot - the next frame's k_classify on a second
 * stream, k_resolve on a third with the sample colours double-buffered, Colour.Zero not written again into blocks the last frame of the same signature left zero; 0 switches each off; k_resolve goes aside only in frames of one chunk), "mains" (2 = default, 1 .. 3: queued frames of one chunk take turns on that many main streams, so a frame's kernels are dispatched while its predecessor's drain
 * and two frames' reflection levels fill each other's idle stretches; "two_mains" = 0 / 1 is mains = 1 / 2), "primary_reserve" (0 = default: workgroup slots such a frame's k_primary leaves free for the small kernels queued beside it), "window_hint" (0 = default; 1: the chunks of a
 * classified frame are cut as wide as the last frame of the same scene, size and sample count left them room for, up to "window_cap" listed samples - fewer empty launches on sparse frames, measured no net gain), "wave_samples" (0 = default, 16: a bounce-0 wavefront takes up to that many jitter offsets of 64 / that many pixels of an
 * 8x8 block when the sample count has the power of two in it - a narrower bundle; 1, 2, 4, 8, 16; no pixel depends on it).  Scene-affecting options need a new ft_scene_commit. */
int32_t ft_set_option(ft_context* ctx, const char* key, int64_t value);

/* ---- scene graph builder (Scene.fs:8-53) ------------------------------------------------- */
ft_node ft_sg_primitive(ft_context* ctx, int32_t kind);                       /* Scene.fs:10-17  */
ft_node ft_sg_triangle(ft_context* ctx, const double v[9]);                   /* Scene.fs:18     */
/* BspMesh.bspMesh false depth triangles (BspMesh.fs:88-97); tris = n x 9 doubles (a,b,c). */
ft_node ft_sg_bsp_mesh(ft_context* ctx, int32_t depth, const double* tris, int64_t n_tris); /* Scene.fs:9 */
ft_node ft_sg_transform(ft_context* ctx, const ft_transform* ts, int32_t n, ft_node child); /* Scene.fs:42 */
ft_node ft_sg_material(ft_context* ctx, const ft_material* m, ft_node child);  /* Scene.fs:43     */
ft_node ft_sg_hue_shift(ft_context* ctx, double angle, ft_node child);         /* Scene.fs:45     */
ft_node ft_sg_ignore_light(ft_context* ctx, ft_node child);                    /* Scene.fs:46     */
ft_node ft_sg_group(ft_context* ctx, const ft_node* children, int32_t n);      /* Scene.fs:35     */
ft_node ft_sg_csg(ft_context* ctx, int32_t op, ft_node a, ft_node b);          /* Scene.fs:36-39  */
/* Scene.fs:44,47-53.  uv_ops: n x {kind, a, b}, outermost TextureFunction first: {0, sx, sy} = Scale, {1, radians, 0} =
 * Rotate (Textures/Texture.fs:13-22); at most 5 per texture. */
ft_node ft_sg_texture_grid(ft_context* ctx, const double colour_a[3], const double colour_b[3],
                           const double* uv_ops, int32_t n_uv_ops, ft_node child);
/* Texture.Image (Scene.fs:48; ImageTexture.image, Textures/Image.fs:20-36).  The F# closure owns the decoded pixels;
 * here the caller hands them over: rgb24 = image.SavePixelData() of an Image<Rgb24>, width*height*3 bytes, row 0 first
 * (copied).  Lookup is the reference's nearest texel at index y*(3*width)+3*x with x = floor(u*width), y = floor(v*height)
 * after Texture.repeat; an index past the last texel (where the reference raises) reads the last texel. */
ft_node ft_sg_texture_image(ft_context* ctx, const uint8_t* rgb24, int32_t width, int32_t height,
                            const double* uv_ops, int32_t n_uv_ops, ft_node child);

/* ---- scene (Scene.fs:107-110, Light.fs:19-26) -------------------------------------------- */
int32_t ft_scene_clear(ft_context* ctx);
int32_t ft_scene_set_objects(ft_context* ctx, ft_node root);
int32_t ft_scene_add_directional(ft_context* ctx, const double dir[3], const double colour[3]);
int32_t ft_scene_add_soft_directional(ft_context* ctx, const double dir[3], int32_t samples,
                                      double scatter_rad, const double colour[3]);
int32_t ft_scene_add_positional(ft_context* ctx, const double pos[3], const double falloff[3],
                                const double colour[3]);
/* Flatten the graph, build BSP trees (BspMesh.compile, BspMesh.fs:51-65) and upload to HBM.  The exact BVH that stands in for the
 * linear scan of a `bspMesh 0` (BspMesh.fs:95-97) is built by the host below 4096 triangles and on the device from there on ("bvh_builder" = 2, default). */
int32_t ft_scene_commit(ft_context* ctx);
/* Wall time of the last ft_scene_commit in ms: [0] flatten on the host (includes the host's BVH builds with "bvh_builder" = 0),
 * [1] BVH builds on the device, [2] uploads and the rest; [3] is not a time: the height of the tallest device-built tree. */
int32_t ft_get_commit_times(ft_context* ctx, double ms[4]);

/* ---- render (Program.fs:54-64) ----------------------------------------------------------- */
/* res_h x res_v is Image.Resolution (Image.fs:28).  jitter_xy = spp x 2 offsets, the ONE pattern
 * shared by every pixel (Image.fs:105); spp == 0 selects `samples corner` (CornerSampling, Image.fs:125-150:
 * one ray per pixel corner, jitter_xy ignored).  max_depth = the recursion limit (8 in Shading.fs:142).
 * seed keys the counter-based streams of soft lights.  tiles == NULL renders the whole frame;
 * otherwise only pixels inside the n_tiles rects are written.  out_rgb is res_v x res_h x 3
 * doubles, row 0 = top (Image.fs:39).  out_rgb may be NULL: the frame then stays in HBM until
 * ft_fetch_frame copies it out (same layout; only the last render's tile pixels are written). */
int32_t ft_render(ft_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp,
                  const double* jitter_xy, int32_t max_depth, uint64_t seed,
                  const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats);

int32_t ft_fetch_frame(ft_context* ctx, double* out_rgb);

/* The same frame as Image.write consumes it (Image.fs:35-44): RGBA8, one byte per channel = truncate(clamp01(c) * 255) (Image.fs:36,
 * Math.fs:12-16; NaN -> 0), alpha 255, res_v x res_h x 4 bytes, row 0 = top.  The quantisation runs on the device in the kernel
 * that averages the samples, so 4 instead of 24 bytes per pixel cross the PCIe link; bytes are identical to ft_quantise_rgba8 of
 * ft_render's frame.  After ft_render_rgba8 / ft_render_enqueue_rgba8 the frame in HBM is the RGBA8 one: fetch it with
 * ft_fetch_frame_rgba8 (ft_fetch_frame then returns FT_ERR_STATE, and vice versa). */
int32_t ft_render_rgba8(ft_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp,
                        const double* jitter_xy, int32_t max_depth, uint64_t seed,
                        const ft_rect* tiles, int32_t n_tiles, uint8_t* out_rgba, ft_stats* stats);
int32_t ft_fetch_frame_rgba8(ft_context* ctx, uint8_t* out_rgba);

/* Page-locked host memory (hipHostMalloc) for frames handed to ft_render / ft_fetch_frame*: the copy out of HBM is then one DMA
 * without the runtime's staging through its own pinned buffers.  Optional: any host pointer works. */
void* ft_host_alloc(size_t bytes);
void  ft_host_free(void* p);

/* A queued frame that also leaves the device: the copy into host_out (res_v x res_h x 3 doubles, or x 4 bytes with rgba8 != 0) is queued
 * behind the frame's last kernel and is complete when ft_render_wait returns (or when a later call retires the frame).  host_out should
 * come from ft_host_alloc (one DMA beside the next frame's tracing); one buffer per frame in flight (four at most: a buffer is the host's again once
 * a later call has retired its frame - queuing frame k + 4 retires frame k - or ft_render_wait has returned). */
int32_t ft_render_enqueue_into(ft_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                               int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, int32_t rgba8, void* host_out);

/* Pipelined rendering, for hosts that render frame after frame (an animation, a progressive preview):
 * ft_render_enqueue queues a frame exactly as ft_render(out_rgb = NULL) would and returns without waiting, so the host prepares
 * the next frame while this one runs; at most four frames are in flight (queuing a fifth first waits for the oldest: one or two are traced, the next
 * waits dispatched on another stream, the last is being classified).  On a context
 * over several devices every device queues its bands on its own stream; ft_render_wait waits for all of them and sums their statistics.
 * ft_render_wait blocks until everything queued has finished, reports the statistics of the LAST frame, and leaves in
 * ft_get_kernel_times the stage times and launch counts summed over all frames since the previous wait.  The frame buffer holds
 * the last frame (ft_fetch_frame).  The reference's own flow is synchronous (Program.fs:63-64): ft_render stays that way. */
int32_t ft_render_enqueue(ft_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                          int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles);
int32_t ft_render_enqueue_rgba8(ft_context* ctx, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                                int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles);
int32_t ft_render_wait(ft_context* ctx, ft_stats* stats);

/* Closest hit of single rays through the device path (Scene.intersectScene, Scene.fs:118, after
 * Shading.slightOffset is NOT applied): for tests.  Outputs per ray: t, p[3], n[3], material index
 * resolved colour[3]; hit[i] = 0 when the ray misses. */
int32_t ft_debug_closest(ft_context* ctx, const double* origins, const double* dirs, int64_t n,
                         int32_t* hit, double* t, double* p, double* nrm, double* colour);
/* lightIsBocked (Scene.fs:119-121) for single rays. */
int32_t ft_debug_blocked(ft_context* ctx, const double* origins, const double* dirs,
                         const double* max_dist, int64_t n, int32_t* blocked);

/* getColourForRay (Shading.fs:131-139) for single rays through the device path: slightOffset, closest hit, shadow queries, the
 * shaders of Program.fs:59 and up to max_depth reflection bounces, exactly as a frame's samples are shaded.  rgb = n x 3. */
int32_t ft_debug_colour(ft_context* ctx, const double* origins, const double* dirs, int64_t n, int32_t max_depth, double* rgb);

/* Host-logic test hooks (no device work): a context that can build, flatten and BSP-compile a scene
 * but whose render/debug calls fail with FT_ERR_NO_DEVICE; flattened-scene sizes
 * (out = leaves, program words, meshes, bsp nodes, bsp leaves, triangles, csg capacity, stack capacity, top-level items,
 * items with a bounding sphere, 1 if some item is unbounded, distinct face directions or -1);
 * and Triangle.slice (Triangle.fs:24-41) as the BSP builder implements it (9 doubles per triangle,
 * at most 2 triangles per side). */
int32_t ft_create_host_only(ft_context** out);
int32_t ft_debug_scene_info(ft_context* ctx, int64_t out[12]);
int32_t ft_debug_slice(const double p0[3], const double n[3], const double tri[9],
                       double above[18], int32_t* n_above, double below[18], int32_t* n_below);
/* HIP-event time per stage over the last ft_render: index 4 primary (bounce 0 fused: generate + closest + shade), 2 the later
 * bounces (one k_bounce per level; one bracket around them all, or with "timing" = 2 one per level), 3 resolve and 0 the rest (the
 * fill, classification) with "timing" = 2; otherwise 0 = everything that is not bracketed and 3 = 0.  Index 1 is unused. */
int32_t ft_get_kernel_times(ft_context* ctx, double ms[5], int32_t launches[5]);

/* The device ordinals behind a context, in the order given to ft_create (at most `capacity` written); returns how many there are. */
int32_t ft_debug_devices(ft_context* ctx, int32_t* ordinals, int32_t capacity);

/* Image.write's toByte (Image.fs:36): clamp to [0,1], *255, truncate; alpha = 255. */
int32_t ft_quantise_rgba8(const double* rgb, int64_t n_pixels, uint8_t* out_rgba);

#ifdef __cplusplus
}
#endif
#endif
