// ft_device.h — device-side buffer descriptors and the host-callable launch interface of
// ft_kernels.hip.  Included by the C-ABI layer (ft_capi.cpp); contains no HIP device code.
#ifndef FT_DEVICE_H
#define FT_DEVICE_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "ft_flat.h"

namespace ftk {

constexpr int kBlock = 256;        // 4 wavefronts of 64; waves never synchronise with each other
constexpr int kMaxBounce = 16;     // recursion limits above this are rejected by ft_render
constexpr int kWorkGroups = 64;    // independent work cursors per launch

struct DevScene {
    const double* leaves;          // n_leaves x 16 doubles (ftd::Leaf)
    const double* m2w;             // n_leaves x 12
    const ftd::Material* materials;
    const ftd::Light* lights;
    const ftd::Texture* textures;
    const uint32_t* program;
    const ftd::Mesh* meshes;
    const ftd::BspNode* nodes;
    const ftd::BspLeaf* bsp_leaves;
    const double* tris;            // 9 per triangle: v0, e1, e2
    const double* culls;           // 24 doubles per ftd::CullRecord
    const uint32_t* tri_orig;      // 1 per triangle
    const double* wide;            // 28 doubles per 4-wide BVH node (ft_flat.h)
    const int32_t* mesh_wide;      // per mesh: root of its 4-wide BVH or INT32_MIN
    const uint8_t* tex_pixels;     // Rgb24 rows of the image textures (ftd::Texture::pixel_base indexes into it)
    const float* cull_items;       // 8 floats per top-level item (centre, radius, row mask; bare meshes: + first coarse box, count, leaf)
    const float* coarse_boxes;     // 6 floats per box: model-space boxes that cover a mesh (k_classify)
    const uint32_t* item_pc;       // n_items + 1 program counters: where each top-level item starts (last: the OP_END word)
    const double* cull_rows;       // 3 per distinct parallel-sensitive direction
    int32_t n_leaves, n_lights, csg_cap, stack_cap;
    int32_t shadow_rays_per_hit;   // sum over lights of the shadow rays the reference casts per hit
    int32_t n_items, n_cull_rows;  // n_cull_rows < 0: pre-test disabled
    int32_t csg_rows, lane_fold;   // LDS rows per hit-list column and lanes folded together (see HitList): csg_cap <= csg_rows * lane_fold
    int32_t coherent_waves;        // 1 (default): bounce-0 wavefronts use the bundle paths (cone cull, packet traversal); 0: every wave is treated as incoherent (diagnostic)   // sum over lights of the shadow rays the reference casts per hit
};

// Ray wavefront buffer, struct-of-arrays so a wave's 64 records are 512 contiguous bytes per field.
// sizeof(RayRec) = 7*8 + 4 = 60 bytes; HitRec = 8 + 4 + 4 = 16 bytes (roofline).
struct RayBuf { double *ox, *oy, *oz, *dx, *dy, *dz, *w; uint32_t* slot; };
struct HitBuf { double* t; uint32_t* id0; uint32_t* id1; };
constexpr uint64_t kRayRecBytes = 60, kHitRecBytes = 16;
// Bytes a launch has to move per unit, by construction of the pipeline (roofline): a primary ray costs a 4-byte
// pixel id read and a 1-byte flag write (it is regenerated, never stored); a reflection ray a 60-byte record written once and
// read in k_closest (48 B: origin + direction) and twice in k_shade; a hit a 16-byte record + 4-byte list entry, written once,
// read once; an accumulator 24 B stored (bounce 0) or read-modify-written (later bounces); a pixel 24 B out.
constexpr uint64_t kPixelIdBytes = 4, kTouchedBytes = 1, kListBytes = 4, kAccBytes = 24;

// Length of the frame's active pixel list (k_classify / k_compact), in device memory: the later stages size themselves from it.
struct PixCount { uint32_t n_pix, pad[3]; };
// Per-chunk device counters (zeroed before every chunk).
struct ChunkCounters {
    uint32_t n_rays[kMaxBounce + 2];   // rays queued for bounce k
    uint32_t n_hits[kMaxBounce + 2];   // compacted lit/unlit hit count of bounce k
    // Work cursors: kWorkGroups independent counters per kernel and bounce, one 64-byte line each (see ft_kernels.hip).
    uint32_t work_trace[kMaxBounce + 2][kWorkGroups * 16];
    uint32_t work_shade[kMaxBounce + 2][kWorkGroups * 16];
};
// Per-render device statistics (zeroed before every render).
struct RenderCounters {
    unsigned long long rays_shadow, rays_reflect, hits_primary, csg_overflow;
    double ref_equiv;
    unsigned long long hits_total;      // hits shaded over all bounces
    unsigned long long tail_in, tail_rays, tail_hits;   // k_tail: rays handed over, reflection rays it spawned, hits it shaded
    unsigned long long pixels_culled;                   // k_classify: pixels whose every primary ray provably misses everything
};

struct Camera {                    // ImagePlane (Image.fs:55-63), computed on the host
    double o[3], k[3], i[3], j[3];
    double pw, ph, tlx, tly;
    double focal_length, tan_half_aperture;   // Image.Focus (Image.fs:9), tan (apetureAngularSize / 2) from the host
    int32_t res_h, res_v;
    int32_t has_focus, pad;
};

struct Launch {
    hipStream_t stream;
    int grid;                      // persistent grid size (workgroups)
    size_t lds_bytes;
    int variant;                   // kernel variant: bit 0 = Oren-Nayar / textures, bit 1 = soft lights, bit 2 = meshes compiled in
};

// Everything bounce 0 needs to regenerate a primary ray from its sample index i = s*n_pix + pixel.
struct Primary {
    Camera cam;
    const uint32_t* pixel_ids;     // pixel id (y*res_h + x) of list entry pix_base + pixel
    const double* jitter;          // spp x 2, the ONE pattern shared by every pixel (Image.fs:105)
    uint32_t pix_base, n_pix;
    int32_t spp;
    uint32_t stride;               // ids are y*stride + x: res_h for pixels, res_h + 1 for the corner grid of `samples corner`
    unsigned long long seed;       // keys the counter-based streams of soft shadows / depth of field
    double inv_n_pix, inv_stride;  // 1.0 / n_pix, 1.0 / stride (division-free index arithmetic, see div_by)
    const PixCount* counts;        // non-null: pixel_ids is the FRAME's active pixel list (k_classify), counts->n_pix its length, and this chunk
                                   // works on the window [pix_base, pix_base + n_pix) of it
};
// K2: closest hit of every ray of bounce k; compacts the indices of rays that hit into hit_list.  Bounce 0 also records, one
// byte per sample, whether the primary ray hit anything (`touched`): untouched samples are Colour.Zero and their accumulator is
// neither cleared nor read.
void launch_closest(const Launch& L, const DevScene& S, const Primary& gen, RayBuf rays, HitBuf hits, uint32_t* hit_list, uint8_t* touched, int bounce, uint32_t tail_threshold,
                    ChunkCounters* cc, RenderCounters* rc);
// K3: shading + shadow rays + accumulation for the compacted hits of bounce k; emits bounce k+1 rays.
// K1: pixel-block classification.  A block of 64 pixels (all its samples) whose ray bundle cannot reach any top-level item is
// finished on the spot (its output pixels are written as Colour.Zero); the others are compacted into the chunk's active
// pixel list, which is all the later stages see.
void launch_classify(const Launch& L, const DevScene& S, const Primary& gen_list, uint8_t* block_active, uint32_t* segment_count, uint32_t* active_ids,
                     uint32_t* active_pos, PixCount* counts, double* out, int whole, double jitter_extent, RenderCounters* rc);
constexpr uint32_t kClassifySegmentBlocks = 256;               // blocks per compaction segment (ft_kernels.hip: kSegmentBlocks)
// Tail of the bounce loop (k_tail): once a bounce has fewer than `threshold` rays the per-bounce stages stand down and this one
// launch follows every remaining path to its end inside registers.
void launch_tail(const Launch& L, const DevScene& S, const Primary& gen, RayBuf rays_even, RayBuf rays_odd, double* acc, uint32_t acc_stride,
                 int max_depth, uint32_t threshold, ChunkCounters* cc, RenderCounters* rc);
int occupancy_blocks_tail(size_t lds_bytes, int variant);
// Bounce 0 fused (k_primary): generate, closest hit, shadow queries, shaders and reflection spawn for the frame's primary rays.
void launch_primary(const Launch& L, const DevScene& S, const Primary& gen, RayBuf next, double* acc, uint8_t* touched, uint32_t acc_stride, int max_depth,
                    ChunkCounters* cc, RenderCounters* rc);
int occupancy_blocks_primary(size_t lds_bytes, int variant);
void launch_shade(const Launch& L, const DevScene& S, const Primary& gen, RayBuf rays, HitBuf hits, const uint32_t* hit_list, RayBuf next,
                  double* acc, uint32_t acc_stride, int bounce, int max_depth, ChunkCounters* cc, RenderCounters* rc);
// K4: mean over the spp samples of each pixel, in sample order (Image.fs:112-116).
// out_index == nullptr: pixel p is written at out_rgb + 3p (packed); else at out_rgb + 3*out_index[p] (in place in the frame).
void launch_blend(const Launch& L, const double* acc, const uint8_t* touched, uint32_t acc_stride, uint32_t n_pix, const PixCount* counts, uint32_t first, int32_t spp, const uint32_t* out_index, double* out_rgb);
// CornerSampling.blendPixels (Image.fs:134-144) for a w x h rect whose (w+1) x (h+1) corner colours are in acc (one sample each).
void launch_blend_corner(const Launch& L, const double* acc, const uint8_t* touched, uint32_t acc_stride, uint32_t w, uint32_t h, const uint32_t* out_index, double* out_rgb);
// Sum the per-wave statistic slots 1..n_slots into slot 0.  The caller sizes the slots from the device: n_cu x 8 blocks x 4 waves,
// the largest grid any launcher here uses (persistent grids are n_cu x clamp_blocks(occupancy) <= 8, the others are clamped to n_cu x 8).
void launch_reduce_stats(const Launch& L, RenderCounters* slots, uint32_t n_slots);
// Debug: closest hit / blocked for arbitrary rays (no slightOffset).
void launch_debug_closest(const Launch& L, const DevScene& S, const double* o, const double* d, uint32_t n,
                          int32_t* hit, double* t, double* p, double* nrm, double* colour, RenderCounters* rc);
void launch_debug_blocked(const Launch& L, const DevScene& S, const double* o, const double* d, const double* max_dist,
                          uint32_t n, int32_t* blocked, RenderCounters* rc);

} // namespace ftk
#endif
