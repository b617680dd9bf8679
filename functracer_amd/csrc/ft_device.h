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
    int32_t n_simd;                // SIMDs of the device (CUs x 4): how far few rays are spread (batch_lanes_for)
    int32_t coherent_waves;        // 1 (default): bounce-0 wavefronts use the bundle paths (cone cull, packet traversal); 0: every wave is treated as incoherent (diagnostic)   // sum over lights of the shadow rays the reference casts per hit
};

// Ray wavefront buffer, struct-of-arrays so a wave's 64 records are 512 contiguous bytes per field: 7 doubles + the sample slot =
// 60 bytes per reflection ray (roofline).  Hits never become records: a level's closest hits are shaded in the same kernel.
struct RayBuf { double *ox, *oy, *oz, *dx, *dy, *dz, *w; uint32_t* slot; };
constexpr uint64_t kRayRecBytes = 60;
// Bytes a launch has to move per unit, by construction of the pipeline (roofline): a primary ray costs a 4-byte pixel id
// read and its sample's 24-byte colour stored (it is generated, never stored); a reflection ray a 60-byte record written once and read
// once; a hit of a later bounce a 48-byte read-modify-write of its sample's colour; a pixel 24 B (4 B as RGBA8) out.
constexpr uint64_t kPixelIdBytes = 4, kAccBytes = 24;

// Length of the frame's active pixel list (k_classify), in device memory: the later stages size themselves from it.
struct PixCount { uint32_t n_pix, pad[3]; };
// Per-chunk device counters (zeroed before every chunk).
struct ChunkCounters {
    uint32_t n_rays[kMaxBounce + 2];   // rays queued for bounce k
    uint32_t pad[14];
    // Work cursors: kWorkGroups independent counters per bounce, one 64-byte line each (see ft_kernels.hip).
    uint32_t work_trace[kMaxBounce + 2][kWorkGroups * 16];
};
// Per-render device statistics.  Every wave adds what it counted to one of kStatStripes copies (its wave index mod 64), one
// no-return atomic per non-zero counter at the end of a launch; the host sums the stripes when it retires the frame.
struct RenderCounters {
    unsigned long long rays_shadow, rays_reflect, hits_primary, csg_overflow;
    double ref_equiv;
    unsigned long long hits_total;      // hits shaded over all bounces
    unsigned long long unused[3];
    unsigned long long pixels_culled;                   // k_classify: pixels whose every primary ray provably misses everything
    unsigned long long rays_shadow_primary, rays_reflect_primary;   // the k_primary share of rays_shadow / rays_reflect
    unsigned long long pad[4];                          // 128 bytes: one stripe per pair of cache lines
};
constexpr int kStatStripes = 64;
// Everything a frame's kernels count in, in one allocation so that ONE fill clears it: the chunk counters (cleared again before
// every further chunk), the statistic stripes, the length of the active pixel list and k_classify's ticket / error words.
struct FrameCounters {
    ChunkCounters cc;
    RenderCounters stats[kStatStripes];
    PixCount counts;
    uint32_t classify_ticket;      // k_classify: waves take their 64-block segment in ticket order, so a wave's predecessors are always running
    uint32_t classify_error;       // set when a bounded wait ran out (never observed; the host then fails the frame instead of hanging)
    uint32_t report_ticket;        // k_resolve / k_report: ticket stripes that are complete (the workgroup that completes the last one writes the frame's report)
    uint32_t pad[1];
    uint32_t report_stripe[64];    // workgroups of stripe blockIdx % 64 that have finished (8000 tickets on one word took 80 us)
};
// What the host reads of a frame's counters, in pinned host memory.  The last workgroup of the frame's last k_resolve writes it and
// then clears the FrameCounters for the next frame: a steady stream of frames needs no fill and no device-to-host copy in between
// (three dispatches less per frame, ~20 us of a 0.4 ms frame with the gaps around them).
struct FrameReport {               // 256 bytes: one wavefront writes it with one store instruction (word by word over PCIe a report of
    RenderCounters total;          // all 64 stripes took 200 us); the stripes summed in stripe order
    uint32_t n_rays[kMaxBounce + 2];   // the last chunk's rays per bounce
    uint32_t n_pix_active;         // PixCount::n_pix
    uint32_t classify_error;
    uint32_t pad[12];
};
static_assert(sizeof(FrameReport) == 256, "FrameReport is written as 64 words");

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
    const PixCount* counts;        // non-null: the chunk's list is the FRAME's active pixel list (k_classify), counts->n_pix its length, and this
                                   // chunk works on the window [pix_base, pix_base + n_pix) of it
    const uint32_t* block_map;     // with counts: block b of the active list is block block_map[b] of pixel_ids; null: pixel_ids is the list itself
    int32_t group_log2;            // samples are numbered in groups of 2^this per 64-pixel block (slot_at, ft_kernels.hip); 0: sample plane by sample plane
};
// Bounce 0 fused (k_primary): generate the primary rays of the chunk, closest hit, shadow queries, shaders, reflection spawn, and
// one colour per sample stored into acc (Colour.Zero for a miss).
void launch_primary(const Launch& L, const DevScene& S, const Primary& gen, RayBuf next, double* acc, uint32_t acc_stride, int max_depth, FrameCounters* fc);
int occupancy_blocks_primary(size_t lds_bytes, int* variant);   // may add bit 3 to *variant: the lean kernel built for five workgroups per CU
// One level of the reflection tree (bounce k >= 1) fused (k_bounce): closest hit, shadow queries, shaders and accumulation for every
// ray of rays (n = fc->cc.n_rays[bounce]); the level's reflection rays are compacted into `next`, or, with `follow`, followed to their
// end inside the launch (the last level the host launches).  A launch that finds no rays returns at once.
void launch_bounce(const Launch& L, const DevScene& S, const Primary& gen, RayBuf rays, RayBuf next, double* acc, uint32_t acc_stride, int bounce, int max_depth, bool follow, FrameCounters* fc);
int occupancy_blocks_bounce(size_t lds_bytes, int variant);
// Pixel-block classification + compaction in one kernel.  One LANE per 64-pixel block (an 8x8 tile of the pixel list): the block's
// ray bundle - all samples of its pixels - is bounded by a cone through its outermost jittered corners and tested against every
// top-level item (bare meshes also against their coarse boxes).  Blocks nothing can be hit from get block_pos = -1 (k_resolve writes
// their pixels as Colour.Zero; none of their rays is ever generated); the others are appended, in block order, to the frame's active
// pixel list (pos_block: the block of the pixel list behind each block of the active list), whose length lands in fc->counts.  `epoch` tags this frame's entries of wave_counts.
struct ClassifyOut { int32_t* block_pos; uint32_t* pos_block; uint32_t* wave_counts; };
void launch_classify(const Launch& L, const DevScene& S, const Primary& gen_list, const ClassifyOut& out, double jitter_extent, uint32_t epoch, FrameCounters* fc);
// The frame's pixels: mean over the spp samples of each pixel of the chunk's window, in sample order (Image.fs:112-116), written as
// FP64 RGB (out_rgb) and / or as Image.write's RGBA8 bytes (out_rgba, Image.fs:36); with `zero_culled` also Colour.Zero for every
// pixel of the blocks k_classify finished.  Pixel p of the list goes to out index pixel_ids[p] (whole frame) or p (tiles, packed).
struct ResolveArgs {
    const double* acc; uint32_t acc_stride; const PixCount* counts; uint32_t first, n_pix_host; int32_t spp;
    const uint32_t* pos_block;     // classified frames: block of the ORIGINAL pixel list behind block b of the active list; else null (identity)
    const int32_t* block_pos;      // non-null: this launch also clears the culled blocks (block_pos[b] < 0) of the n_blocks_total blocks
    uint32_t n_blocks_total;
    const uint32_t* pixel_ids;     // the original pixel list (whole frame: out index = pixel id); null: out index = list position
    double* out_rgb; uint8_t* out_rgba;
    uint32_t group_log2;           // the chunk's slot numbering (Primary::group_log2)
    FrameCounters* fc; FrameReport* report;   // report non-null: the frame's last launch (see FrameReport; fc is cleared behind it)
};
void launch_resolve(const Launch& L, const ResolveArgs& a);   // L.grid: at most this many workgroups (occupancy_blocks_resolve() per CU: one resident round)
int occupancy_blocks_resolve();
void launch_report(const Launch& L, FrameCounters* fc, FrameReport* report);   // the same hand-over as a launch of its own (frames that end in another kernel)
// CornerSampling.blendPixels (Image.fs:134-144) for a w x h rect whose (w+1) x (h+1) corner colours are in acc (one sample each).
void launch_resolve_corner(const Launch& L, const double* acc, uint32_t acc_stride, uint32_t w, uint32_t h, const uint32_t* out_index, double* out_rgb, uint8_t* out_rgba);
// Device-side BVH build (ft_bvh.hip): a linear BVH over triangles [first_global, first_global + n) of `tris`, written into the ranges
// the flattener reserved for the mesh: n - 1 BspNodes at node_base, (n - 1) + n BspLeafs at leaf_base, n sorted triangle records and
// list indices at tri_base, n - 1 4-wide nodes at wide_base, coarse_count (<= 64) float boxes at coarse.  *height = height of the
// binary tree in nodes, 0 when the mesh holds a non-finite coordinate (nothing usable was written).
struct LbvhTarget {
    double* tris; uint32_t first_global, n;
    ftd::BspNode* nodes; uint32_t node_base;
    ftd::BspLeaf* leaves; uint32_t leaf_base;
    uint32_t* tri_orig; uint32_t tri_base;
    double* wide; uint32_t wide_base;
    float* coarse; uint32_t coarse_count;
};
hipError_t build_lbvh(hipStream_t stream, const LbvhTarget& t, uint32_t* height, int kind = 1);   // kind 0: linear BVH, 1: binned surface-area tree over the Morton order

// Debug: closest hit / blocked for arbitrary rays (no slightOffset).
void launch_debug_closest(const Launch& L, const DevScene& S, const double* o, const double* d, uint32_t n,
                          int32_t* hit, double* t, double* p, double* nrm, double* colour, unsigned long long* overflow);
void launch_debug_blocked(const Launch& L, const DevScene& S, const double* o, const double* d, const double* max_dist,
                          uint32_t n, int32_t* blocked, unsigned long long* overflow);

} // namespace ftk
#endif
