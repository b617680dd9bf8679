// ft_capi.cpp — the C ABI of libfunctracer_hip.so (include/functracer_hip.h): context, scene
// builder, HBM residency of the flattened scene and the per-frame wavefront pipeline driver.
// Reference citations are relative to /root/reference/FuncTracer/.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/functracer_hip.h"
#include "ft_device.h"
#include "ft_scene.h"



struct DeviceBuf {
    void* p = nullptr; size_t bytes = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Stage indices of ft_get_kernel_times.
enum { kStageOther = 0, kStageClosest = 1, kStageShade = 2, kStageResolve = 3, kStagePrimary = 4, kStages = 5 };
// What a frame copies back when it retires: FrameCounters from `stats` to its end.
static_assert(sizeof(ftk::FrameCounters) % 16 == 0 && offsetof(ftk::RenderCounters, ref_equiv) == 32, "the hand-over at the end of a frame copies words and clears 16 bytes at a time");

// One host thread per extra device of a multi-device context, alive as long as the context: every frame hands each of them its share
// (round 2 created and joined a std::thread per device per frame - the same order of time as a device's share of a 4K frame).
struct DeviceWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, quit = false;
    void start() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<void()> fn = std::move(job);
                has_job = false;
                lk.unlock();
                fn();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<void()> fn) { std::lock_guard<std::mutex> lk(m); job = std::move(fn); has_job = true; done = false; cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this] { return done; }); }
    void stop() { { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); } if (th.joinable()) th.join(); }
};

constexpr int64_t kDeviceBvhMinTris = 4096;   // "bvh_builder" = 2: smaller meshes get the host's swept SAH tree (a few ms at most), larger ones the device's binned one
struct ft_context {
    static constexpr int kMains = 3;   // main streams at most: consecutive simple frames trace on different ones (option "mains" says how many are in use)
    static constexpr int kAcc = kMains;   // copies of the sample colours: one per frame between its k_primary and its k_resolve
    static constexpr int kSlots = kMains + 1;   // frames in flight: one per main stream + the one being classified ahead
    std::vector<ft_context*> peers;      // multi-device contexts: one more single-device context per extra GPU (scene replicated)
    std::vector<DeviceWorker*> workers;  // ... and one host thread per peer
    bool host_only = false;
    int device = -1;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipStream_t more_mains[kMains - 1] = {};   // further main streams: consecutive simple queued frames trace on different ones, so that a frame's kernels are dispatched while its predecessors' drain
    std::string err;

    fth::SceneGraph graph;
    fth::FlatScene flat;
    std::vector<float> cull_items_and_rows;   // what d_cull_items holds (the upload's source)
    bool committed = false;
    int bvh_builder = 2;            // who builds the exact BVH of top-level-Leaf meshes: 0 = the host (swept surface-area split: the best tree, 1.2 ms for 980
                                    // triangles but 160 ms for 69.6 K), 1 = the device's linear BVH (ft_bvh.hip: ~1 ms, traces ~9 % slower), 3 = the device's
                                    // binned surface-area tree over the Morton order, 2 = by size: the host's below kDeviceBvhMinTris triangles, 3's from there on
    double commit_ms[4] = {0, 0, 0, 0};   // last ft_scene_commit: flatten on the host, device BVH builds, uploads + the rest, BVH height (not a time)

    int64_t chunk_samples = 16ll << 20;   // measured: 8 Mi costs 10-25 % (more, smaller launches), 32 Mi slows the shading on many-light scenes
    int wave_samples_log2 = -1;     // bounce-0 wavefronts take 2^this samples of 64 / 2^this pixels when the sample count allows (option wave_samples); -1: the default, 16
    bool coherent_waves = true;     // diagnostic: 0 routes every wavefront through the incoherent paths
    int timing = 1;                 // HIP events: 0 around the frame only, 1 + around every tracing kernel (k_primary, the k_bounce levels), 2 around every stage
    bool classify_pixels = true;    // k_classify: pixel blocks that cannot see any item are finished before any ray is generated
    int64_t follow_below = -1;      // option "follow_below": a level of the reflection tree in which the previous frame had no more rays than this gets no launch of
                                    // its own: the last level launched follows them in registers.  -1: two rays per SIMD (2048 on 256 CUs: 8 x n_cu).  Measured at 1080p
                                    // (0 -> 10 000): hollow-sphere x1 0.881 -> 0.863 ms, sample-det x16 1.190 -> 1.164, sample-soft x4 0.905 -> 0.855; following
                                    // levels of 50 000 rays and more loses (hollow-sphere x1 0.976): a lane then drags its wave through every level
    bool level_hint = true;         // launch only as many k_bounce levels as the previous frame of the same signature had (+ 1); 0: always max_depth

    // scene in HBM
    DeviceBuf d_leaves, d_m2w, d_materials, d_lights, d_program, d_meshes, d_nodes, d_bleaves, d_tris, d_culls, d_tri_orig, d_textures, d_tex_pixels, d_cull_items, d_cull_rows, d_item_pc, d_wave_counts, d_wide, d_mesh_wide, d_coarse;
    // What k_classify writes and the frame's later kernels read exists once per frame slot, so that a queued frame's classification can
    // run (on `side`, behind an event) while the frame before it is still tracing: block_pos / pos_block and the frame's counters.
    DeviceBuf d_block_pos[kSlots], d_pos_block[kSlots], d_fc[kSlots];
    hipStream_t side = nullptr;     // the second stream: k_classify of frame N + 1 beside k_primary's tail / k_resolve of frame N (ft_render_enqueue)
    bool classify_ahead = true;     // option "classify_ahead": 0 keeps every kernel on the one stream
    bool classify_after_trace = false;   // option "classify_after_trace": the classification run ahead waits for the previous frame's tracing kernels
    // Kernel variants and resident workgroups per CU for the committed scene (they only change at commit): bit 0 FANCY, 1 SOFT, 2 MESH; the
    // primary's variant may carry bit 3 (the five-workgroup lean build).
    int variant = 0, variant_primary = 0, blocks_primary = 1, blocks_bounce = 1, blocks_resolve = 2;
    int resolve_blocks_cap = 0;     // option "resolve_blocks": k_resolve workgroups per CU (0: every resident one)
    hipEvent_t classified = nullptr;  // behind the latest k_classify on either stream: the next one waits for it (they share the ticket words of d_wave_counts)
    ftk::DevScene dev_scene{};
    // frame buffers in HBM
    DeviceBuf d_rays[2 * kMains], d_acc[kAcc], d_out, d_out8, d_pixels, d_jitter, d_dbg_in, d_dbg_out;
    // The sample colours exist twice: a queued frame's k_resolve runs on a stream of its own (`tail`), behind an event, while the next
    // chunk's / frame's k_primary already fills the other copy - the small kernel hides in the big one's ramp instead of standing between
    // two of them.  acc_free[i]: behind the last k_resolve that read copy i (the next k_primary into that copy waits for it).
    int acc_turn = 0;
    hipStream_t tail = nullptr;
    hipEvent_t acc_free[kAcc] = {};
    bool acc_busy[kAcc] = {};
    bool resolve_aside = true;      // option "resolve_aside": 0 keeps k_resolve on the main stream
    bool fc_clean[kSlots] = {};   // d_fc[slot] is all zero: the slot's previous frame cleared it behind its report (no fill needed)
    // Colour.Zero in the blocks k_classify finished: what the last frame written into d_out / d_out8 classified (scene, camera, size, pixel
    // list, jitter extent).  A frame of the same signature finds those pixels zero already and does not write them again.
    uint64_t zero_signature[2] = {0, 0};
    bool zero_fill_skip = true;     // option "zero_fill_skip"
    uint32_t classify_epoch = 0;    // tags the entries k_classify's waves publish in d_wave_counts (cleared only when it wraps or the buffer grows)
    int64_t ray_capacity = 0, acc_capacity = 0;
    // Per-frame host state.  Two slots, so that one frame can be queued while the previous one still runs (ft_render_enqueue).
    struct FrameSlot {
        std::vector<hipEvent_t> events; size_t events_used = 0;
        struct Span { hipEvent_t a, b; int kind; };
        std::vector<Span> spans;
        bool simple = false, alt = false;       // one chunk, k_resolve aside; alt: traced on main stream `main_ix` != 0
        int main_ix = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
        hipEvent_t traced = nullptr;            // (one of `events`, not owned) behind the frame's last tracing kernel, in front of its k_resolve: where the NEXT frame's k_classify may start
        ftk::FrameReport* h_report = nullptr;   // pinned: the frame's statistic stripes, k_classify's error word and the last chunk's rays per bounce,
        ftk::FrameReport* d_report = nullptr;   // written by the frame's last kernel through this device-side address of the same memory
        int levels_launched = 0, last_bounce = 0;
        uint64_t signature = 0;                 // what the frame rendered (scene, size, samples, depth, threshold): keys the staged-launch hint
        bool pending = false;
        uint64_t rays_primary = 0; int64_t n_pix_total = 0; int32_t spp = 0, n_launches = 0, n_chunks = 0, timing = 1, format = 0; bool classify = false;
        std::chrono::steady_clock::time_point wall0;
    };
    FrameSlot slots[kSlots];
    int slot_turn = 0;
    // Levels of the reflection tree worth launching: the host cannot know how deep the rays of a frame go without waiting, and a
    // k_bounce launch that finds no rays still costs a few microseconds.  It launches as many levels as the previous frame of the
    // same signature had rays in, plus one; the last one launched follows whatever it still spawns to the end inside the kernel,
    // so the frame is complete however deep it goes.  -1: no history, launch max_depth levels.
    int staged_hint = -1;
    uint64_t staged_signature = 0;
    int64_t active_hint = -1;        // active pixels of the last classified frame retired (and its signature): how wide the next frame's windows may be
    uint64_t active_signature = 0;
    int64_t window_cap = 64ll << 20; // option "window_cap": listed samples a hinted window may span
    int64_t primary_reserve = 0;     // option "primary_reserve": workgroup slots a simple frame's k_primary leaves free
    int ray_sets = 0;                // main streams whose pair of ray buffers holds ray_capacity records
    int mains = 2;                   // option "mains" (1 .. 3): main streams in use (measured: 2 is best - the headline 0.263 / 0.231 / 0.249 ms with 1 / 2 / 3, hollow-sphere x1 0.703 / 0.471 / 0.470); "two_mains" = 0 is mains = 1
    bool window_hint = false;        // option "window_hint": 1 widens a classified frame's windows by what the last frame of its signature left inactive (see render_single)
    uint64_t commit_serial = 0;
    bool csg_auto_grow = true;   // ft_render: double csg_mesh_capacity and render again when a hit list overflows
    bool accum_open = false;        // kernel times are being summed over pipelined frames (reset by the next enqueue after a wait)
    // pixel list of the last render, cached across calls with the same resolution and tiles
    std::vector<uint32_t> pixels;
    std::vector<double> jitter_on_device;   // what d_jitter holds
    std::vector<ft_rect> pixel_rects;
    bool pixels_corner = false, pixels_tiled = false;   // the list holds corner-sampling pixels / is made of whole 8x8 tiles
    int last_format = 0;            // 0: the last frame is FP64 RGB in d_out, 1: RGBA8 in d_out8
    DeviceBuf d_out_index;
    int64_t last_n_pix = 0;
    int32_t last_res_h = 0, last_res_v = 0;
    double k_ms[kStages] = {0, 0, 0, 0, 0};
    int32_t k_launches[kStages] = {0, 0, 0, 0, 0};
};

namespace {

#define FT_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
            return FT_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)

int32_t ensure(ft_context* c, DeviceBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return FT_OK;
    if (b.p) { FT_HIP(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    FT_HIP(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return FT_OK;
}
template <class T> int32_t upload(ft_context* c, DeviceBuf& b, const std::vector<T>& v) {
    int32_t rc = ensure(c, b, v.size() * sizeof(T));
    if (rc != FT_OK) return rc;
    if (!v.empty()) FT_HIP(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return FT_OK;
}
void release(DeviceBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }

bool need_device(ft_context* c) {
    if (!c) return false;
    if (c->host_only) { c->err = "host-only context: no HIP device bound (there is no CPU fallback for rendering)"; return false; }
    return true;
}

ftk::RayBuf ray_view(const DeviceBuf& b, int64_t cap) {
    double* d = b.as<double>();
    ftk::RayBuf r;
    r.ox = d; r.oy = d + cap; r.oz = d + 2 * cap; r.dx = d + 3 * cap; r.dy = d + 4 * cap; r.dz = d + 5 * cap; r.w = d + 6 * cap;
    r.slot = reinterpret_cast<uint32_t*>(d + 7 * cap);
    return r;
}

// Per-sample accumulators for every frame; the ray wavefront buffers only for scenes with reflective materials (bounce >= 1).
int32_t ensure_frame_buffers(ft_context* c, int64_t cap, bool reflective) {
    int32_t rc;
    if (cap > c->acc_capacity) { for (int k = 0; k < ft_context::kAcc; ++k) if ((rc = ensure(c, c->d_acc[k], (size_t)cap * 24)) != FT_OK) return rc; c->acc_capacity = cap; }
    if (!reflective || (cap <= c->ray_capacity && c->ray_sets >= c->mains)) return FT_OK;
    const int64_t want = std::max(cap, c->ray_capacity);
    for (int i = 0; i < 2 * c->mains; ++i) if ((rc = ensure(c, c->d_rays[i], (size_t)want * (7 * 8 + 4))) != FT_OK) return rc;   // a ping-pong pair per main stream in use
    c->ray_capacity = want; c->ray_sets = c->mains;
    return FT_OK;
}

hipEvent_t next_event(ft_context::FrameSlot& f) {
    if (f.events_used == f.events.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; f.events.push_back(e); }
    return f.events[f.events_used++];
}

// ImagePlane.create (Image.fs:48-53, 67-81), evaluated once per frame on the host.
ftk::Camera make_camera(const ft_camera& cam, int res_h, int res_v) {
    auto norm = [](double v[3]) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (!(l < 0.0000001)) { double s = 1.0 / l; v[0] = s * v[0]; v[1] = s * v[1]; v[2] = s * v[2]; } };
    ftk::Camera out{};
    double k[3] = {cam.look_at[0] - cam.o[0], cam.look_at[1] - cam.o[1], cam.look_at[2] - cam.o[2]};
    norm(k);
    const double* u = cam.up;
    double i[3] = {u[1] * k[2] - u[2] * k[1], k[0] * u[2] - k[2] * u[0], u[0] * k[1] - u[1] * k[0]};     // up .** k
    norm(i);
    double j[3] = {k[1] * i[2] - k[2] * i[1], i[0] * k[2] - i[2] * k[0], k[0] * i[1] - k[1] * i[0]};     // k .** i
    const double height = std::tan(cam.fov_y / 2.0) * 2.0;
    const double width = height * cam.aspect_ratio;
    const double pixel_height = height / (double)(res_h - 1);      // sic, Image.fs:71: resH
    const double pixel_width = width / (double)(res_v - 1);        // sic, Image.fs:72: resV
    for (int a = 0; a < 3; ++a) { out.o[a] = cam.o[a]; out.k[a] = k[a]; out.i[a] = i[a]; out.j[a] = j[a]; }
    out.pw = pixel_width; out.ph = pixel_height;
    out.tlx = -width / 2.0 + pixel_width / 2.0; out.tly = height / 2.0 - pixel_height / 2.0;
    out.res_h = res_h; out.res_v = res_v;
    out.has_focus = cam.has_focus ? 1 : 0; out.focal_length = cam.focal_length;
    out.tan_half_aperture = std::tan(cam.aperture_angular_size / 2.0);          // Jitter.fs:30
    return out;
}

// LDS per workgroup: the per-lane CSG hit lists (4 words per entry) and tree stacks.  When the lists alone would not fit, lanes are
// folded (ft_kernels.hip, HitList): fold live lanes share the columns of 64 lanes, so a column needs only ceil(capacity / fold) rows.
constexpr size_t kLdsPerWorkgroup = 160 * 1024;
int lane_fold_for(const fth::FlatScene& f) {
    for (int fold = 1; fold <= 16; fold *= 2) {
        const size_t rows = ((size_t)f.csg_capacity + (size_t)fold - 1) / (size_t)fold;
        if ((4 * rows + (size_t)f.stack_capacity) * ftk::kBlock * 4 <= kLdsPerWorkgroup) return fold;
    }
    return 0;
}
// Materials only the FANCY kernel variants shade: Oren-Nayar, textures, and a specular exponent that is not a small whole number
// (the lean variants raise to whole powers up to 64 by square-and-multiply and do not carry Math.Pow, ft_kernels.hip shade_lights).
bool needs_fancy(const ftd::Material& m) {
    const bool whole = m.shineyness <= 64.0 && m.shineyness == std::floor(m.shineyness);
    return m.roughness != 0.0 || m.texture >= 0 || (m.shineyness > 0.0 && !whole) || m.shineyness != m.shineyness;
}
size_t lds_bytes_for(const fth::FlatScene& f) {
    const int fold = std::max(1, lane_fold_for(f));
    return (4 * (((size_t)f.csg_capacity + (size_t)fold - 1) / (size_t)fold) + (size_t)f.stack_capacity) * ftk::kBlock * 4;
}

} // namespace

extern "C" {

int32_t ft_abi_version(void) { return FT_ABI_VERSION; }

static int32_t create_single(int32_t device_id, int count, ft_context** out) {
    if (device_id < 0 || device_id >= count) return FT_ERR_INVALID;
    ft_context* c = new ft_context();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipSetDevice(c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return FT_ERR_HIP; }
    int pr_low = 0, pr_high = 0;                                   // the side stream's few workgroups go first whenever slots come free
    if (hipDeviceGetStreamPriorityRange(&pr_low, &pr_high) != hipSuccess) pr_high = 0;
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, pr_high) != hipSuccess || hipStreamCreateWithPriority(&c->tail, hipStreamNonBlocking, pr_high) != hipSuccess) {
        if (c->side) (void)hipStreamDestroy(c->side);
        (void)hipStreamDestroy(c->stream); delete c; return FT_ERR_HIP;
    }
    for (hipStream_t& m : c->more_mains) if (hipStreamCreateWithFlags(&m, hipStreamNonBlocking) != hipSuccess) { m = nullptr; c->mains = 1; }   // (without them every frame takes the one main stream)
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return FT_OK;
}

int32_t ft_create(const int32_t* device_ids, int32_t n_devices, ft_context** out) {
    if (!out) return FT_ERR_INVALID;
    *out = nullptr;
    if (n_devices < 1 || !device_ids) return FT_ERR_NO_DEVICE;     // no CPU backend exists in this library
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return FT_ERR_NO_DEVICE;
    ft_context* c = nullptr;
    int32_t rc = create_single(device_ids[0], count, &c);
    if (rc != FT_OK) return rc;
    // More devices: the scene is replicated and every frame is split into 8-row bands dealt round-robin (no exchange
    // between devices; the bands meet in the caller's host buffer).  The same ordinal may be listed twice.
    for (int32_t k = 1; k < n_devices; ++k) {
        ft_context* p = nullptr;
        rc = create_single(device_ids[k], count, &p);
        if (rc != FT_OK) { ft_destroy(c); return rc; }
        c->peers.push_back(p);
        DeviceWorker* w = new DeviceWorker();
        w->start();
        c->workers.push_back(w);
    }
    *out = c;
    return FT_OK;
}

int32_t ft_create_host_only(ft_context** out) {
    if (!out) return FT_ERR_INVALID;
    ft_context* c = new ft_context();
    c->host_only = true;
    *out = c;
    return FT_OK;
}

void ft_destroy(ft_context* c) {
    if (!c) return;
    for (DeviceWorker* w : c->workers) { w->stop(); delete w; }
    c->workers.clear();
    for (ft_context* p : c->peers) ft_destroy(p);
    c->peers.clear();
    if (!c->host_only) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (hipStream_t m : c->more_mains) if (m) (void)hipStreamSynchronize(m);
        if (c->side) (void)hipStreamSynchronize(c->side);
        if (c->tail) (void)hipStreamSynchronize(c->tail);
        DeviceBuf* bufs[] = {&c->d_leaves, &c->d_m2w, &c->d_materials, &c->d_lights, &c->d_program, &c->d_meshes, &c->d_nodes, &c->d_bleaves, &c->d_tris, &c->d_culls, &c->d_tri_orig, &c->d_textures, &c->d_tex_pixels, &c->d_cull_items, &c->d_cull_rows, &c->d_item_pc, &c->d_block_pos[0], &c->d_block_pos[1], &c->d_block_pos[2], &c->d_block_pos[3], &c->d_pos_block[0], &c->d_pos_block[1], &c->d_pos_block[2], &c->d_pos_block[3], &c->d_wave_counts, &c->d_wide, &c->d_mesh_wide, &c->d_coarse, &c->d_out_index,
                             &c->d_rays[0], &c->d_rays[1], &c->d_rays[2], &c->d_rays[3], &c->d_rays[4], &c->d_rays[5], &c->d_acc[0], &c->d_acc[1], &c->d_acc[2], &c->d_out, &c->d_out8, &c->d_pixels, &c->d_jitter, &c->d_fc[0], &c->d_fc[1], &c->d_fc[2], &c->d_fc[3],
                             &c->d_dbg_in, &c->d_dbg_out};
        for (auto* b : bufs) release(*b);
        for (auto& f : c->slots) { f.traced = nullptr; if (f.h_report) { (void)hipHostFree(f.h_report); f.h_report = nullptr; f.d_report = nullptr; } for (auto e : f.events) (void)hipEventDestroy(e); f.events.clear(); }
        if (c->classified) (void)hipEventDestroy(c->classified);
        for (hipEvent_t& e : c->acc_free) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (c->side) (void)hipStreamDestroy(c->side);
        if (c->tail) (void)hipStreamDestroy(c->tail);
        for (hipStream_t m : c->more_mains) if (m) (void)hipStreamDestroy(m);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

const char* ft_last_error(const ft_context* c) { return c ? c->err.c_str() : "null context"; }

int32_t ft_set_option(ft_context* c, const char* key, int64_t value) {
    if (!c || !key) return FT_ERR_INVALID;
    if (!std::strcmp(key, "chunk_samples")) { if (value < 64) return FT_ERR_INVALID; c->chunk_samples = value; for (ft_context* p : c->peers) p->chunk_samples = value; return FT_OK; }
    if (!std::strcmp(key, "csg_mesh_capacity")) { if (value < 1 || value > 255) return FT_ERR_INVALID; c->graph.csg_mesh_capacity = (int32_t)value; c->committed = false; return FT_OK; }
    if (!std::strcmp(key, "coherent_waves")) { c->coherent_waves = value != 0; c->dev_scene.coherent_waves = value != 0 ? 1 : 0; for (ft_context* p : c->peers) { p->coherent_waves = value != 0; p->dev_scene.coherent_waves = c->dev_scene.coherent_waves; } return FT_OK; }
    if (!std::strcmp(key, "wave_samples")) {
        int l = 0; while ((1ll << l) < value) ++l;                 // 0: the default (16); 1, 2, 4, 8, 16: that many samples per wavefront (k_resolve's LDS tile holds 16)
        if (value < 0 || value > 16 || (value > 0 && (1ll << l) != value)) return FT_ERR_INVALID;
        if (value == 0) l = -1;
        c->wave_samples_log2 = l; for (ft_context* p : c->peers) p->wave_samples_log2 = l; return FT_OK;
    }
    if (!std::strcmp(key, "timing")) { if (value < 0 || value > 2) return FT_ERR_INVALID; c->timing = (int)value; for (ft_context* p : c->peers) p->timing = (int)value; return FT_OK; }
    if (!std::strcmp(key, "window_cap")) { if (value < 64 || value > (1ll << 30)) return FT_ERR_INVALID; c->window_cap = value; for (ft_context* p : c->peers) p->window_cap = value; return FT_OK; }
    if (!std::strcmp(key, "primary_reserve")) { if (value < 0 || value > 4096) return FT_ERR_INVALID; c->primary_reserve = value; for (ft_context* p : c->peers) p->primary_reserve = value; return FT_OK; }
    if (!std::strcmp(key, "two_mains") || !std::strcmp(key, "mains")) {
        const int m = key[0] == 't' ? (value != 0 ? 2 : 1) : (int)value;
        if (m < 1 || m > ft_context::kMains) return FT_ERR_INVALID;
        auto set = [&](ft_context* p) { p->mains = p->more_mains[0] && p->more_mains[1] ? m : 1; };
        set(c); for (ft_context* p : c->peers) set(p);
        return FT_OK;
    }
    if (!std::strcmp(key, "window_hint")) { c->window_hint = value != 0; for (ft_context* p : c->peers) p->window_hint = value != 0; return FT_OK; }
    if (!std::strcmp(key, "resolve_aside")) { c->resolve_aside = value != 0; for (ft_context* p : c->peers) p->resolve_aside = value != 0; return FT_OK; }
    if (!std::strcmp(key, "resolve_blocks")) { if (value < 0 || value > 8) return FT_ERR_INVALID; c->resolve_blocks_cap = (int)value; for (ft_context* p : c->peers) p->resolve_blocks_cap = (int)value; return FT_OK; }
    if (!std::strcmp(key, "classify_after_trace")) { c->classify_after_trace = value != 0; for (ft_context* p : c->peers) p->classify_after_trace = value != 0; return FT_OK; }
    if (!std::strcmp(key, "classify_ahead")) { c->classify_ahead = value != 0; for (ft_context* p : c->peers) p->classify_ahead = value != 0; return FT_OK; }
    if (!std::strcmp(key, "zero_fill_skip")) { c->zero_fill_skip = value != 0; c->zero_signature[0] = c->zero_signature[1] = 0; for (ft_context* p : c->peers) { p->zero_fill_skip = value != 0; p->zero_signature[0] = p->zero_signature[1] = 0; } return FT_OK; }
    if (!std::strcmp(key, "classify_pixels")) { c->classify_pixels = value != 0; for (ft_context* p : c->peers) p->classify_pixels = value != 0; return FT_OK; }
    if (!std::strcmp(key, "csg_auto_grow")) { c->csg_auto_grow = value != 0; return FT_OK; }
    if (!std::strcmp(key, "follow_below")) { if (value < -1) return FT_ERR_INVALID; c->follow_below = value; c->staged_hint = -1; for (ft_context* p : c->peers) { p->follow_below = value; p->staged_hint = -1; } return FT_OK; }
    if (!std::strcmp(key, "level_hint")) { c->level_hint = value != 0; for (ft_context* p : c->peers) p->level_hint = value != 0; return FT_OK; }
    if (!std::strcmp(key, "bvh_builder")) { if (value < 0 || value > 3) return FT_ERR_INVALID; c->bvh_builder = (int)value; c->committed = false; return FT_OK; }
    if (!std::strcmp(key, "mesh_unclipped_bvh")) { c->graph.mesh_unclipped_bvh = value != 0; c->committed = false; return FT_OK; }
    c->err = std::string("unknown option: ") + key;
    return FT_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------ builder
static ft_node add_node(ft_context* c, fth::GraphNode&& n) { c->graph.nodes.push_back(std::move(n)); c->committed = false; return (ft_node)c->graph.nodes.size() - 1; }

ft_node ft_sg_primitive(ft_context* c, int32_t kind) {
    if (!c || kind < 0 || kind > FT_PRIM_CYLINDER) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Prim; n.prim = kind; return add_node(c, std::move(n));
}
ft_node ft_sg_triangle(ft_context* c, const double v[9]) {
    if (!c || !v) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::TriangleP; std::memcpy(n.tri, v, sizeof n.tri); return add_node(c, std::move(n));
}
ft_node ft_sg_bsp_mesh(ft_context* c, int32_t depth, const double* tris, int64_t n_tris) {
    if (!c || n_tris < 0 || (n_tris > 0 && !tris) || depth < 0) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Mesh; n.depth = depth; n.tris.assign(tris, tris + 9 * n_tris); return add_node(c, std::move(n));
}
ft_node ft_sg_transform(ft_context* c, const ft_transform* ts, int32_t n_ts, ft_node child) {
    if (!c || !c->graph.valid(child) || !ts || n_ts < 1) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Transform;
    for (int i = 0; i < n_ts; ++i) { if (ts[i].kind < FT_TRANSLATE || ts[i].kind > FT_ROTATE) return FT_ERR_INVALID; n.xf.push_back(ts[i]); }
    n.children = {child};
    return add_node(c, std::move(n));
}
ft_node ft_sg_material(ft_context* c, const ft_material* m, ft_node child) {
    if (!c || !c->graph.valid(child) || !m) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::MaterialF; n.mat = *m; n.children = {child}; return add_node(c, std::move(n));
}
ft_node ft_sg_hue_shift(ft_context* c, double, ft_node child) {
    if (!c || !c->graph.valid(child)) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::HueShift; n.children = {child}; return add_node(c, std::move(n));
}
ft_node ft_sg_ignore_light(ft_context* c, ft_node child) {
    if (!c || !c->graph.valid(child)) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::IgnoreLight; n.children = {child}; return add_node(c, std::move(n));
}
ft_node ft_sg_group(ft_context* c, const ft_node* children, int32_t n_children) {
    if (!c || n_children < 0 || (n_children > 0 && !children)) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Group;
    for (int i = 0; i < n_children; ++i) { if (!c->graph.valid(children[i])) return FT_ERR_INVALID; n.children.push_back(children[i]); }
    return add_node(c, std::move(n));
}
ft_node ft_sg_csg(ft_context* c, int32_t op, ft_node a, ft_node b) {
    if (!c || !c->graph.valid(a) || !c->graph.valid(b) || op < FT_CSG_UNION || op > FT_CSG_EXCLUDE) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Csg; n.op = op; n.children = {a, b}; return add_node(c, std::move(n));
}
ft_node ft_sg_texture_grid(ft_context* c, const double ca[3], const double cb[3], const double* uv_ops, int32_t n_uv_ops, ft_node child) {
    if (!c || !c->graph.valid(child) || !ca || !cb || n_uv_ops < 0 || (n_uv_ops > 0 && !uv_ops)) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Texture;
    std::memcpy(n.ca, ca, sizeof n.ca); std::memcpy(n.cb, cb, sizeof n.cb);
    n.uv_ops.assign(uv_ops, uv_ops + 3 * n_uv_ops); n.children = {child};
    return add_node(c, std::move(n));
}

ft_node ft_sg_texture_image(ft_context* c, const uint8_t* rgb24, int32_t width, int32_t height, const double* uv_ops, int32_t n_uv_ops, ft_node child) {
    if (!c || !c->graph.valid(child) || !rgb24 || width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 28) || n_uv_ops < 0 || (n_uv_ops > 0 && !uv_ops)) return FT_ERR_INVALID;
    fth::GraphNode n; n.kind = fth::GraphNode::Texture;
    n.pixels.assign(rgb24, rgb24 + (size_t)width * height * 3); n.img_w = width; n.img_h = height;
    n.uv_ops.assign(uv_ops, uv_ops + 3 * n_uv_ops); n.children = {child};
    return add_node(c, std::move(n));
}

int32_t ft_scene_clear(ft_context* c) {
    if (!c) return FT_ERR_INVALID;
    c->graph.nodes.clear(); c->graph.lights.clear(); c->graph.root = -1; c->committed = false;
    return FT_OK;
}
int32_t ft_scene_set_objects(ft_context* c, ft_node root) {
    if (!c || !c->graph.valid(root)) return FT_ERR_INVALID;
    c->graph.root = root; c->committed = false;
    return FT_OK;
}
static void norm3(double v[3]) {                                    // Vector.normalise (CommonTypes.fs:63-67)
    double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(l < 0.0000001)) { double s = 1.0 / l; v[0] = s * v[0]; v[1] = s * v[1]; v[2] = s * v[2]; }
}
int32_t ft_scene_add_directional(ft_context* c, const double dir[3], const double colour[3]) {      // Light.directional (Light.fs:19-20)
    if (!c || !dir || !colour) return FT_ERR_INVALID;
    ftd::Light l{}; l.kind = ftd::LT_DIRECTIONAL;
    std::memcpy(l.v, dir, sizeof l.v); norm3(l.v); std::memcpy(l.colour, colour, sizeof l.colour);
    c->graph.lights.push_back(l); c->committed = false;
    return FT_OK;
}
int32_t ft_scene_add_soft_directional(ft_context* c, const double dir[3], int32_t samples, double scatter_rad, const double colour[3]) {  // Light.fs:22-23
    if (!c || !dir || !colour || samples < 1) return FT_ERR_INVALID;
    ftd::Light l{}; l.kind = ftd::LT_SOFT; l.samples = samples; l.scatter = scatter_rad;
    std::memcpy(l.v, dir, sizeof l.v); norm3(l.v); std::memcpy(l.colour, colour, sizeof l.colour);
    c->graph.lights.push_back(l); c->committed = false;            // rejected at commit until the seeded stream lands
    return FT_OK;
}
int32_t ft_scene_add_positional(ft_context* c, const double pos[3], const double falloff[3], const double colour[3]) {  // Light.fs:25-26
    if (!c || !pos || !falloff || !colour) return FT_ERR_INVALID;
    ftd::Light l{}; l.kind = ftd::LT_POINT;
    std::memcpy(l.v, pos, sizeof l.v); std::memcpy(l.falloff, falloff, sizeof l.falloff); std::memcpy(l.colour, colour, sizeof l.colour);
    c->graph.lights.push_back(l); c->committed = false;
    return FT_OK;
}

static int32_t upload_scene(ft_context* c);
static int32_t retire_pending(ft_context* c, ft_stats* stats);
static bool any_pending(const ft_context* c, bool on_second_main = false) { for (const auto& f : c->slots) if (f.pending && (!on_second_main || f.alt)) return true; return false; }

int32_t ft_scene_commit(ft_context* c) {
    if (!c) return FT_ERR_INVALID;
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
    for (double& v : c->commit_ms) v = 0.0;
    // A device context builds the exact BVH of top-level-Leaf meshes on the device ("bvh_builder" = 1; 2, the default: from 4096 triangles on): the flattener
    // reserves the ranges, upload_scene fills them.  A build the device refuses (a tree too deep for the traversal stacks) falls
    // back to the host's builder, once, for the whole scene.
    for (int attempt = 0; attempt < 2; ++attempt) {
        c->graph.device_bvh = !c->host_only && c->bvh_builder >= 1 && attempt == 0;
        c->graph.device_bvh_min_tris = c->bvh_builder == 2 ? kDeviceBvhMinTris : 0;   // 1: the device's linear BVH, 3: its surface-area tree, whatever the size
        auto t0 = clock::now();
        int32_t rc = c->graph.flatten(c->flat, c->err);
        c->commit_ms[0] += ms_since(t0);
        if (rc != FT_OK) return rc;
        if (c->host_only) { c->committed = true; return FT_OK; }
        t0 = clock::now();
        rc = upload_scene(c);
        for (ft_context* p : c->peers) {                            // replicate the flattened scene on every other device
            if (rc != FT_OK) break;
            p->flat = c->flat;
            if ((rc = upload_scene(p)) != FT_OK) c->err = p->err;
            c->commit_ms[1] += p->commit_ms[1];
        }
        c->commit_ms[2] += ms_since(t0) - c->commit_ms[1];
        if (rc == FT_ERR_BUILD && c->graph.device_bvh) continue;    // refused by the device builder: the host builds it
        return rc;
    }
    return FT_ERR_BUILD;
}

int32_t ft_get_commit_times(ft_context* c, double ms[4]) {
    if (!c || !ms) return FT_ERR_INVALID;
    for (int k = 0; k < 4; ++k) ms[k] = c->commit_ms[k];
    return FT_OK;
}

static int32_t upload_scene(ft_context* c) {
    int32_t rc;
    FT_HIP(c, hipSetDevice(c->device));
    // frames still queued trace the scene these uploads replace, and not all of them on the stream the uploads travel on (FrameSlot::alt)
    if (any_pending(c)) { if ((rc = retire_pending(c, nullptr)) != FT_OK) return rc; c->accum_open = false; }
    const fth::FlatScene& f = c->flat;
    if (lane_fold_for(f) == 0) { c->err = "scene needs more than 160 KiB of LDS per workgroup for CSG lists / BSP stacks even with 4 live lanes per wave"; return FT_ERR_UNSUPPORTED; }
    if ((rc = upload(c, c->d_leaves, f.leaves)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_m2w, f.m2w)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_materials, f.materials)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_lights, f.lights)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_textures, f.textures)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_tex_pixels, f.tex_pixels)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_program, f.program)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_meshes, f.meshes)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_nodes, f.nodes)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_bleaves, f.bsp_leaves)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_tris, f.tris)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_culls, f.culls)) != FT_OK) return rc;
    {   // behind the items' float records: a float image of every parallel-sensitive direction (x, y, z, its length rounded up), which lane k of a
        // coherent wave tests against the bundle's cone before any ray is tested against it exactly (rows_nearly_parallel, ft_kernels.hip)
        std::vector<float>& v = c->cull_items_and_rows;
        v = f.cull_items;
        v.resize(8 * (f.item_pc.size() - 1), 0.0f);
        for (size_t k = 0; k + 2 < f.cull_rows.size(); k += 3) {
            const double len = std::sqrt(f.cull_rows[k] * f.cull_rows[k] + f.cull_rows[k + 1] * f.cull_rows[k + 1] + f.cull_rows[k + 2] * f.cull_rows[k + 2]);
            float lf = (float)len; while ((double)lf < len) lf = std::nextafter(lf, std::numeric_limits<float>::infinity());
            v.push_back((float)f.cull_rows[k]); v.push_back((float)f.cull_rows[k + 1]); v.push_back((float)f.cull_rows[k + 2]); v.push_back(lf);
        }
        if ((rc = upload(c, c->d_cull_items, v)) != FT_OK) return rc;
    }
    if ((rc = upload(c, c->d_cull_rows, f.cull_rows)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_item_pc, f.item_pc)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_wide, f.wide)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_mesh_wide, f.mesh_wide)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_coarse, f.coarse_boxes)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_tri_orig, f.tri_orig)) != FT_OK) return rc;
    for (int k = 0; k < ft_context::kSlots; ++k) { if ((rc = ensure(c, c->d_fc[k], sizeof(ftk::FrameCounters))) != FT_OK) return rc; c->fc_clean[k] = false; }
    c->zero_signature[0] = c->zero_signature[1] = 0;
    FT_HIP(c, hipStreamSynchronize(c->stream));
    {   // the BVHs the flattener left to the device (ft_bvh.hip), straight into the ranges reserved in the arrays just uploaded
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t tallest = 0;
        for (const fth::FlatScene::BvhJob& j : f.bvh_jobs) {
            const ftk::LbvhTarget t{c->d_tris.as<double>(), j.first_global, j.n, c->d_nodes.as<ftd::BspNode>(), j.node_base, c->d_bleaves.as<ftd::BspLeaf>(), j.leaf_base,
                                    c->d_tri_orig.as<uint32_t>(), j.tri_base, c->d_wide.as<double>(), j.wide_base, c->d_coarse.as<float>() + 6 * (size_t)j.coarse_first, j.coarse_count};
            uint32_t height = 0;
            FT_HIP(c, ftk::build_lbvh(c->stream, t, &height, c->bvh_builder == 1 ? 0 : 1));
            // height 0: a non-finite coordinate; > 40: deeper than the packet walk's 64-entry stack allows (3 entries per 4-wide level)
            if (height == 0 || height > 40) { c->err = "device BVH build refused (non-finite vertex or a tree deeper than 40 levels): the host builder takes over"; return FT_ERR_BUILD; }
            tallest = std::max(tallest, height);
        }
        if (!f.bvh_jobs.empty()) {
            c->commit_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            c->commit_ms[3] = tallest;
            if ((int32_t)tallest + 1 > c->flat.stack_capacity) c->flat.stack_capacity = (int32_t)tallest + 1;   // per-lane node stacks of the incoherent walk (LDS)
            if (lane_fold_for(c->flat) == 0) { c->err = "scene needs more than 160 KiB of LDS per workgroup for CSG lists / BSP stacks even with 4 live lanes per wave"; return FT_ERR_UNSUPPORTED; }
        }
    }
    ftk::DevScene& S = c->dev_scene;
    S.leaves = c->d_leaves.as<double>(); S.m2w = c->d_m2w.as<double>();
    S.materials = c->d_materials.as<ftd::Material>(); S.lights = c->d_lights.as<ftd::Light>(); S.textures = c->d_textures.as<ftd::Texture>();
    S.program = c->d_program.as<uint32_t>(); S.meshes = c->d_meshes.as<ftd::Mesh>();
    S.nodes = c->d_nodes.as<ftd::BspNode>(); S.bsp_leaves = c->d_bleaves.as<ftd::BspLeaf>(); S.tris = c->d_tris.as<double>(); S.culls = c->d_culls.as<double>(); S.tri_orig = c->d_tri_orig.as<uint32_t>(); S.tex_pixels = c->d_tex_pixels.as<uint8_t>();
    S.coarse_boxes = c->d_coarse.as<float>();
    S.cull_items = c->d_cull_items.as<float>(); S.cull_rows = c->d_cull_rows.as<double>();
    S.wide = c->d_wide.as<double>(); S.mesh_wide = c->d_mesh_wide.as<int32_t>();
    S.item_pc = c->d_item_pc.as<uint32_t>();
    S.coherent_waves = c->coherent_waves ? 1 : 0;
    S.n_simd = c->n_cu * 4;
    S.n_items = (int32_t)f.item_pc.size() - 1; S.n_cull_rows = f.cull_bundle ? (int32_t)(f.cull_rows.size() / 3) : -1;
    S.n_leaves = (int32_t)f.leaves.size(); S.n_lights = (int32_t)f.lights.size();
    S.csg_cap = f.csg_capacity; S.stack_cap = f.stack_capacity;
    S.lane_fold = lane_fold_for(f); S.csg_rows = (f.csg_capacity + S.lane_fold - 1) / S.lane_fold;
    S.shadow_rays_per_hit = 0;
    for (auto& l : f.lights) S.shadow_rays_per_hit += (l.kind == ftd::LT_SOFT) ? l.samples : 1;   // Shading.fs:24-42
    c->variant = 0;
    for (auto& m : f.materials) if (needs_fancy(m)) c->variant |= 1;                               // FANCY
    for (auto& l : f.lights) if (l.kind == ftd::LT_SOFT) c->variant |= 2;                          // SOFT
    if (!f.meshes.empty()) c->variant |= 4;                                                        // MESH
    const size_t lds = lds_bytes_for(c->flat);
    c->variant_primary = c->variant;
    c->blocks_primary = ftk::occupancy_blocks_primary(lds, &c->variant_primary);
    c->blocks_bounce = ftk::occupancy_blocks_bounce(lds, c->variant);
    c->blocks_resolve = ftk::occupancy_blocks_resolve();
    c->committed = true;
    ++c->commit_serial; c->staged_hint = -1; c->active_hint = -1;
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ frames out of HBM
// The device keeps the last frame in FRAME layout (row 0 = top, Image.fs:39) whatever the tiles were: d_out as FP64 RGB or d_out8 as
// Image.write's RGBA8 bytes (Image.fs:36).  Fetching copies the rendered rects - whole rows as one copy, narrower rects as a 2D copy -
// straight into the caller's frame; nothing is gathered or scattered on the host.
static int32_t copy_frame_out(ft_context* c, void* out, int format, hipStream_t async);
static int32_t fetch_single(ft_context* c, void* out, int format) {
    if (c->last_n_pix <= 0) { c->err = "no frame rendered yet"; return FT_ERR_STATE; }
    if (format != c->last_format) { c->err = format == 1 ? "the last frame was rendered as FP64 RGB (ft_render): no RGBA8 frame to fetch" : "the last frame was rendered as RGBA8 (ft_render_rgba8): no FP64 frame to fetch"; return FT_ERR_STATE; }
    FT_HIP(c, hipSetDevice(c->device));
    FT_HIP(c, hipStreamSynchronize(c->stream));                     // frames queued with ft_render_enqueue may still be running (the streams are non-blocking)
    for (hipStream_t m : c->more_mains) if (m) FT_HIP(c, hipStreamSynchronize(m));
    FT_HIP(c, hipStreamSynchronize(c->tail));                       // ... their k_resolve on its own stream
    return copy_frame_out(c, out, format, nullptr);
}
// The rects of the context's pixel list out of d_out / d_out8 into the caller's frame: blocking copies, or (async != null) queued on that
// stream behind the frame's k_resolve - the caller's memory should then be page-locked (ft_host_alloc), or the runtime stages the copy.
static int32_t copy_frame_out(ft_context* c, void* out, int format, hipStream_t async) {
    auto copy1 = [&](void* d, const void* s_, size_t n) { return async ? hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToHost, async) : hipMemcpy(d, s_, n, hipMemcpyDeviceToHost); };
    auto copy2 = [&](void* d, size_t dp, const void* s_, size_t sp, size_t w, size_t h) { return async ? hipMemcpy2DAsync(d, dp, s_, sp, w, h, hipMemcpyDeviceToHost, async) : hipMemcpy2D(d, dp, s_, sp, w, h, hipMemcpyDeviceToHost); };
    const size_t px = format == 1 ? 4 : 24, pitch = (size_t)c->last_res_h * px;
    const char* src = static_cast<const char*>(format == 1 ? c->d_out8.p : c->d_out.p);
    char* dst = static_cast<char*>(out);
    // A device of a multi-device context holds every N-th 8-row band of the frame: whole rows, equally high, equally spaced.  Those go
    // out as ONE two-dimensional copy whose "rows" are the bands (band = 8 x pitch contiguous bytes, 8 N x pitch apart on both sides) -
    // 34 blocking copies of 737 KB per device at 4K otherwise.  A shorter last band follows on its own.
    size_t k0 = 0;
    {
        const auto& R = c->pixel_rects;
        size_t run = 0;
        if (R.size() >= 3 && R[0].x0 == 0 && R[0].w == c->last_res_h) {
            const int step = R[1].y0 - R[0].y0;
            run = 1;
            while (run < R.size() && R[run].x0 == 0 && R[run].w == R[0].w && R[run].h == R[0].h && R[run].y0 == R[0].y0 + (int)run * step) ++run;
            if (step > R[0].h && run >= 3) {
                const size_t off = (size_t)R[0].y0 * pitch;
                FT_HIP(c, copy2(dst + off, (size_t)step * pitch, src + off, (size_t)step * pitch, (size_t)R[0].h * pitch, run));
                k0 = run;
            }
        }
    }
    for (size_t k = k0; k < c->pixel_rects.size();) {
        const ft_rect r = c->pixel_rects[k];
        if (r.x0 == 0 && r.w == c->last_res_h) {                    // whole rows; vertically adjacent rects go out as one copy
            int rows = r.h;
            size_t k2 = k + 1;
            while (k2 < c->pixel_rects.size() && c->pixel_rects[k2].x0 == 0 && c->pixel_rects[k2].w == r.w && c->pixel_rects[k2].y0 == r.y0 + rows) { rows += c->pixel_rects[k2].h; ++k2; }
            FT_HIP(c, copy1(dst + (size_t)r.y0 * pitch, src + (size_t)r.y0 * pitch, (size_t)rows * pitch));
            k = k2;
        } else {
            const size_t off = (size_t)r.y0 * pitch + (size_t)r.x0 * px;
            FT_HIP(c, copy2(dst + off, pitch, src + off, pitch, (size_t)r.w * px, (size_t)r.h));
            ++k;
        }
    }
    return FT_OK;
}

// fn(d) for every listed device d of the context: device 0 on the calling thread, the others on their workers, all at once.
static void on_every_device(ft_context* c, const std::vector<bool>& take, const std::function<void(size_t)>& fn) {
    for (size_t d = 1; d < take.size(); ++d) if (take[d]) c->workers[d - 1]->post([&fn, d] { fn(d); });
    if (take[0]) fn(0);
    for (size_t d = 1; d < take.size(); ++d) if (take[d]) c->workers[d - 1]->wait();
}

static int32_t fetch_all(ft_context* c, void* out, int format) {
    if (!c || !out) return FT_ERR_INVALID;
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    std::vector<ft_context*> devs{c};
    devs.insert(devs.end(), c->peers.begin(), c->peers.end());
    std::vector<ft_context*> with;
    for (ft_context* d : devs) if (d->last_n_pix > 0) with.push_back(d);
    if (with.empty()) { c->err = "no frame rendered yet"; return FT_ERR_STATE; }
    if (with.size() == 1) { const int32_t rc = fetch_single(with[0], out, format); if (rc != FT_OK && with[0] != c) c->err = with[0]->err; return rc; }
    std::vector<int32_t> rcs(devs.size(), FT_OK);                   // every device copies its own bands into the caller's frame, all at once
    std::vector<bool> take(devs.size());
    for (size_t d = 0; d < devs.size(); ++d) take[d] = devs[d]->last_n_pix > 0;
    on_every_device(c, take, [&](size_t d) { rcs[d] = fetch_single(devs[d], out, format); });
    for (size_t d = 0; d < devs.size(); ++d) if (rcs[d] != FT_OK) { if (devs[d] != c) c->err = devs[d]->err; return rcs[d]; }
    return FT_OK;
}
int32_t ft_fetch_frame(ft_context* c, double* out_rgb) { return fetch_all(c, out_rgb, 0); }
int32_t ft_fetch_frame_rgba8(ft_context* c, uint8_t* out_rgba) { return fetch_all(c, out_rgba, 1); }

/* Page-locked host memory for frames (hipHostMalloc): a D2H copy into it is one DMA at link rate, without the runtime's staging. */
void* ft_host_alloc(size_t bytes) { void* p = nullptr; return (bytes && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) ? p : nullptr; }
void ft_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ------------------------------------------------------------------------------------------ render
struct RenderRequest {
    const ft_camera* cam; int32_t res_h, res_v, spp; const double* jitter_xy; int32_t max_depth; uint64_t seed;
    const ft_rect* tiles; int32_t n_tiles; int format;               // 0: FP64 RGB frame, 1: RGBA8 frame
};
static int32_t render_single(ft_context* c, const RenderRequest& q, void* out, ft_stats* stats, bool defer);
static int32_t retire_frame(ft_context* c, ft_context::FrameSlot& f, ft_stats* stats);
static int32_t retire_pending(ft_context* c, ft_stats* stats);
static int32_t render_frame(ft_context* c, const RenderRequest& q, void* out, ft_stats* stats, bool defer);

static int32_t with_growing_hit_lists(ft_context* c, const std::function<int32_t()>& run) {
    // Frames still queued by ft_render_enqueue are retired first, so that an overflow of one of THEM is reported as what it is
    // (queued frames are not rendered again) instead of being taken for this call's.
    if (!c->host_only) {
        std::vector<ft_context*> devs{c};
        devs.insert(devs.end(), c->peers.begin(), c->peers.end());
        for (ft_context* d : devs) {
            if (!any_pending(d)) continue;
            if (hipSetDevice(d->device) != hipSuccess) { c->err = "hipSetDevice failed"; return FT_ERR_NO_DEVICE; }
            const int32_t prc = retire_pending(d, nullptr);
            d->accum_open = false;
            if (prc != FT_OK) { if (d != c) c->err = d->err; return prc; }
        }
    }
    int32_t rc = run();
    while (rc == FT_ERR_OVERFLOW && c->csg_auto_grow && c->graph.csg_mesh_capacity < 255) {
        const int32_t before = c->graph.csg_mesh_capacity;
        const std::string why = c->err;
        c->graph.csg_mesh_capacity = std::min(255, before * 2);
        if (ft_scene_commit(c) != FT_OK) {                          // the longer lists do not fit: back to the scene as it was
            c->graph.csg_mesh_capacity = before;
            if (ft_scene_commit(c) == FT_OK) c->err = why;
            return FT_ERR_OVERFLOW;
        }
        rc = run();
    }
    return rc;
}

// The reference's hit lists are unbounded F# lists; the device's are sized at commit time.  A line that crosses a mesh under CSG
// more often than "csg_mesh_capacity" allows is detected (never truncated): the blocking call then doubles the capacity,
// re-commits the scene and renders the frame again, so the caller sees the reference's result without tuning anything.  The
// larger capacity stays for the following frames.  Only when the lists stop fitting is FT_ERR_OVERFLOW handed to the caller.
int32_t ft_render(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                  int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats) {
    if (!c) return FT_ERR_INVALID;
    const RenderRequest q{cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, 0};
    return with_growing_hit_lists(c, [&] { return render_frame(c, q, out_rgb, stats, false); });
}
int32_t ft_render_rgba8(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                        int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, uint8_t* out_rgba, ft_stats* stats) {
    if (!c) return FT_ERR_INVALID;
    const RenderRequest q{cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, 1};
    return with_growing_hit_lists(c, [&] { return render_frame(c, q, out_rgba, stats, false); });
}

static void add_stats(ft_stats* t, const ft_stats& s) {
    t->rays_primary += s.rays_primary; t->rays_shadow += s.rays_shadow; t->rays_reflect += s.rays_reflect; t->rays_traced += s.rays_traced;
    t->rays_reference_equivalent += s.rays_reference_equivalent; t->hits_primary += s.hits_primary; t->csg_overflow += s.csg_overflow;
    t->kernel_ms = std::max(t->kernel_ms, s.kernel_ms); t->trace_kernel_ms = std::max(t->trace_kernel_ms, s.trace_kernel_ms);
    t->algorithmic_bytes += s.algorithmic_bytes; t->hits_total += s.hits_total; t->algorithmic_bytes_closest += s.algorithmic_bytes_closest;
    t->algorithmic_bytes_shade += s.algorithmic_bytes_shade; t->algorithmic_bytes_primary += s.algorithmic_bytes_primary; t->n_launches += s.n_launches; t->n_chunks += s.n_chunks;
    t->rays_tail += s.rays_tail; t->rays_primary_culled += s.rays_primary_culled; t->rays_shadow_primary += s.rays_shadow_primary; t->rays_reflect_primary += s.rays_reflect_primary;
}

// Image-tile partition of a region over the devices of a context: 8-row bands of every requested rect, dealt round-robin.
static std::vector<std::vector<ft_rect>> band_shares(const RenderRequest& q, size_t n_devs) {
    std::vector<std::vector<ft_rect>> share(n_devs);
    const ft_rect whole_frame{0, 0, q.res_h, q.res_v};
    const ft_rect* src = q.tiles ? q.tiles : &whole_frame;
    const int n_src = q.tiles ? q.n_tiles : 1;
    size_t band = 0;
    for (int k = 0; k < n_src; ++k)
        for (int y = src[k].y0; y < src[k].y0 + src[k].h; y += 8, ++band)
            share[band % n_devs].push_back(ft_rect{src[k].x0, y, src[k].w, std::min(8, src[k].y0 + src[k].h - y)});
    return share;
}

static int32_t render_frame(ft_context* c, const RenderRequest& q, void* out, ft_stats* stats, bool defer) {
    if (c->peers.empty() || c->host_only) return render_single(c, q, out, stats, defer);
    if (!q.cam || q.res_h < 2 || q.res_v < 2 || (q.tiles && q.n_tiles < 1)) { c->err = "bad ft_render argument"; return FT_ERR_INVALID; }
    if (!c->committed) { c->err = "scene not committed (ft_scene_commit)"; return FT_ERR_STATE; }
    const auto wall0 = std::chrono::steady_clock::now();
    std::vector<ft_context*> devs{c};
    devs.insert(devs.end(), c->peers.begin(), c->peers.end());
    const std::vector<std::vector<ft_rect>> share = band_shares(q, devs.size());
    std::vector<int32_t> rcs(devs.size(), FT_OK);
    std::vector<ft_stats> sts(devs.size());
    // One host thread per device: each queues its bands' frame on its own stream, waits for it and copies its bands straight into the
    // caller's frame (whole rows: one contiguous copy per band).  No device waits for another; the bands meet in `out`.
    on_every_device(c, std::vector<bool>(devs.size(), true), [&](size_t d) {
        std::memset(&sts[d], 0, sizeof(ft_stats));
        if (share[d].empty()) { devs[d]->last_n_pix = 0; return; }
        RenderRequest qd = q;
        qd.tiles = share[d].data(); qd.n_tiles = (int32_t)share[d].size();
        rcs[d] = render_single(devs[d], qd, out, &sts[d], defer);
    });
    for (size_t d = 0; d < devs.size(); ++d) if (rcs[d] != FT_OK) { if (d) c->err = devs[d]->err; return rcs[d]; }
    if (stats && !defer) {
        std::memset(stats, 0, sizeof *stats);
        for (auto& s : sts) add_stats(stats, s);
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    return FT_OK;
}

static int32_t render_single(ft_context* c, const RenderRequest& q, void* out, ft_stats* stats, bool defer) {
    if (!c) return FT_ERR_INVALID;
    const ft_camera* cam = q.cam;
    const int32_t res_h = q.res_h, res_v = q.res_v, max_depth = q.max_depth, n_tiles = q.n_tiles;
    const ft_rect* tiles = q.tiles;
    const double* jitter_xy = q.jitter_xy;
    const uint64_t seed = q.seed;
    int32_t spp = q.spp;
    if (!cam || res_h < 2 || res_v < 2 || spp < 0 || (spp > 0 && !jitter_xy) || max_depth < 0 || (tiles && n_tiles < 1)) { c->err = "bad ft_render argument"; return FT_ERR_INVALID; }
    if (max_depth > ftk::kMaxBounce) { c->err = "max_depth above 16"; return FT_ERR_UNSUPPORTED; }
    if ((int64_t)res_h * res_v > (int64_t)0x7FFFFFFF) { c->err = "resolution too large"; return FT_ERR_INVALID; }
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    if (!c->committed) { c->err = "scene not committed (ft_scene_commit)"; return FT_ERR_STATE; }
    const bool corner = spp == 0;                                  // CornerSampling.strategy (Image.fs:125-150): one ray per pixel corner
    if (corner) { spp = 1; }
    const auto wall0 = std::chrono::steady_clock::now();
    FT_HIP(c, hipSetDevice(c->device));

    // Pixel list restricted to the tiles.  The reference enumerates pixels y-major, x (Image.fs:104); samples are
    // independent, so the device is free to walk them in any order: rects whose sides are multiples of 8 are
    // walked in 8x8 pixel blocks, which makes the 64 lanes of a wavefront a compact bundle of rays.
    const bool whole = tiles == nullptr;
    std::vector<ft_rect> rects;
    if (whole) rects.push_back(ft_rect{0, 0, res_h, res_v});
    else for (int k = 0; k < n_tiles; ++k) {
        ft_rect r = tiles[k];
        if (r.x0 < 0) { r.w += r.x0; r.x0 = 0; }
        if (r.y0 < 0) { r.h += r.y0; r.y0 = 0; }
        if (r.x0 + r.w > res_h) r.w = res_h - r.x0;
        if (r.y0 + r.h > res_v) r.h = res_v - r.y0;
        if (r.w > 0 && r.h > 0) rects.push_back(r);
    }
    const bool same_list = !corner && !c->pixels_corner && c->last_n_pix > 0 && c->last_res_h == res_h && c->last_res_v == res_v &&
                           c->pixel_rects.size() == rects.size() && (rects.empty() || std::memcmp(c->pixel_rects.data(), rects.data(), rects.size() * sizeof(ft_rect)) == 0);
    struct Job { uint32_t id_base, n_ids, w, h, out_base, n_out; };
    std::vector<Job> jobs;
    std::vector<uint32_t> corner_ids;
    if (corner) {
        // Each rect (split by rows so that its corner grid fits one chunk) is a job of (w+1) x (h+1) corner rays.
        std::vector<uint32_t>& px = c->pixels;
        px.clear();
        const uint32_t cs = (uint32_t)res_h + 1;
        for (const ft_rect& r : rects) {
            int64_t max_rows = c->chunk_samples / (r.w + 1) - 1;
            if (max_rows < 1) max_rows = 1;
            for (int y0 = r.y0; y0 < r.y0 + r.h; y0 += (int)max_rows) {
                const int h = (int)std::min<int64_t>(max_rows, r.y0 + r.h - y0);
                Job j{(uint32_t)corner_ids.size(), (uint32_t)((r.w + 1) * (h + 1)), (uint32_t)r.w, (uint32_t)h, (uint32_t)px.size(), (uint32_t)(r.w * h)};
                for (int y = y0; y <= y0 + h; ++y) for (int x = r.x0; x <= r.x0 + r.w; ++x) corner_ids.push_back((uint32_t)y * cs + (uint32_t)x);
                for (int y = y0; y < y0 + h; ++y) for (int x = r.x0; x < r.x0 + r.w; ++x) px.push_back((uint32_t)(y * res_h + x));
                jobs.push_back(j);
            }
        }
        c->pixel_rects = rects; c->pixels_corner = true; c->pixels_tiled = false; c->last_n_pix = 0;
    } else if (!same_list) {
        std::vector<uint32_t>& px = c->pixels;
        px.clear();
        bool tiled = true;
        for (const ft_rect& r : rects) {
            if (r.w % 8 == 0 && r.h % 8 == 0) {
                // inside a block the pixels run in Z order (first its top-left corner, last its bottom-right one, as k_classify expects):
                // the 4 or 16 consecutive pixels a wavefront takes under grouped numbering are a 2x2 or 4x4 square, not a strip
                for (int ty = 0; ty < r.h; ty += 8) for (int tx = 0; tx < r.w; tx += 8)
                    for (int k = 0; k < 64; ++k) {
                        const int ix = (k & 1) | ((k >> 1) & 2) | ((k >> 2) & 4), iy = ((k >> 1) & 1) | ((k >> 2) & 2) | ((k >> 3) & 4);
                        px.push_back((uint32_t)((r.y0 + ty + iy) * res_h + r.x0 + tx + ix));
                    }
            } else {
                tiled = false;
                for (int y = r.y0; y < r.y0 + r.h; ++y) for (int x = r.x0; x < r.x0 + r.w; ++x) px.push_back((uint32_t)(y * res_h + x));
            }
        }
        c->pixel_rects = rects; c->pixels_corner = false; c->pixels_tiled = tiled; c->last_n_pix = 0;
    }
    const std::vector<uint32_t>& pixels = c->pixels;
    const int64_t n_pix_total = (int64_t)pixels.size();
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_pix_total == 0) return FT_OK;

    int32_t rc;
    // k_classify bounds every sample of a pixel by a square of +-extent pixels around its centre.  The reference's offsets lie in the
    // unit disc (Jitter.fs:15-21) but the pattern is the caller's: the square follows the pattern, and a pattern with a non-finite
    // or absurd offset turns classification off instead of bounding nothing.
    double jitter_extent = 1.0;
    bool jitter_bounded = true;
    if (!corner) for (size_t k = 0; k < 2 * (size_t)spp; ++k) { const double v = jitter_xy[k]; if (!(std::fabs(v) <= 64.0)) jitter_bounded = false; else jitter_extent = std::max(jitter_extent, std::fabs(v)); }
    // k_classify applies to pinhole cameras over pixel lists made of 8x8 tiles and scenes in which every top-level item is bounded (with
    // a ground plane in view an exact plane test does find the sky blocks - 20 % of night-house - but the denser first chunk makes the
    // shading slower than the blocks save).  A classified frame's chunks are windows of its ACTIVE pixel list, usually a fraction
    // of the frame: they are twice as wide (measured at 1080p x 16 in round 1: bunny 0.58 -> 0.55 ms, hollow-sphere 6.1 -> 5.8, sample
    // 1.64 -> 1.50; the unclassified night-house loses 14 % at that width and keeps the narrow one).
    const bool classify = c->classify_pixels && jitter_bounded && !corner && c->pixels_tiled && !cam->has_focus && c->flat.cull_bundle && c->flat.item_pc.size() > 1 && !c->flat.unbounded;
    uint64_t signature = c->commit_serial * 0x9E3779B97F4A7C15ull;
    for (uint64_t v : {(uint64_t)res_h, (uint64_t)res_v, (uint64_t)spp, (uint64_t)max_depth, (uint64_t)n_pix_total, (uint64_t)c->chunk_samples, (uint64_t)(corner ? 1 : 0)})
        signature = (signature ^ v) * 0x100000001B3ull;
    // (round 3: five times as wide, not twice - 80 Mi listed samples.  A frame of one window is a SIMPLE frame below: its k_resolve goes aside and its
    //  k_primary to the other main stream.  A rank's eighth of 3840x2160x64 - 66 M listed samples, 5 M of them active - was two windows, the
    //  second one empty: 0.458 -> 0.417 ms per frame as one; its half 1.69 -> 1.62, its quarter and the whole frame unchanged, tools/rank_share_ab.py)
    // An unclassified frame without soft lights is worth one chunk of twice the width for the same reason (night-house-det 1080p x 16: two
    // chunks 2.60 ms, one - a simple, pipelined frame - 2.46); with soft lights the narrow chunks still win (night-house: 3.62 against 3.75).
    int64_t chunk_budget = classify ? 5 * c->chunk_samples : ((c->variant & 2) ? c->chunk_samples : 2 * c->chunk_samples);
    // The windows of a classified frame are cut from its LISTED pixels (the host does not know the active list's length when it queues
    // them), so a sparse frame is one window of work and a row of launches that find theirs empty (~20 us each: k_primary + k_resolve +
    // the counter fill; 3840x2160x64 of the bunny: 16 windows, 14 empty).  Option "window_hint" = 1: when the last frame of this
    // signature kept one pixel in `widen`, windows up to `widen` times as wide (at most "window_cap" listed samples) still hold no more
    // ACTIVE samples than the measured optimum.  OFF by default - measured (tools/window_hint_ab.py, tools/window_sweep.py): the empty
    // launches go (`other` 0.33 -> 0.23 ms on that frame) but k_primary over the SAME active samples runs 0 - 9 % slower behind wider
    // colour planes, varying from one allocation of the planes to the next (the stride between the planes is not it: padding it changed
    // nothing): a rank's quarter gains 4 %, a half loses 3 - 7 %, an eighth and the whole frame stay where they were.
    if (classify && c->window_hint && c->active_hint >= 0 && c->active_signature == signature) {
        const int64_t widen = std::max<int64_t>(1, std::min<int64_t>(16, n_pix_total / std::max<int64_t>(64, c->active_hint)));
        chunk_budget = std::max(chunk_budget, std::min(c->window_cap, chunk_budget * widen));
    }
    int64_t pix_per_chunk = std::max<int64_t>(1, std::min<int64_t>(n_pix_total, chunk_budget / spp));
    if (pix_per_chunk > 64) {
        // equal chunks rather than full ones and a remainder: a short last chunk is all latency (measured on night-house at
        // 1080p x 16: 25 M + 8 M samples 5.35 ms, 2 x 16.6 M 4.83 ms); 8x8 blocks (= wavefronts) stay whole
        const int64_t n_chunks = (n_pix_total + pix_per_chunk - 1) / pix_per_chunk;
        const int64_t even = ((n_pix_total + n_chunks - 1) / n_chunks + 63) / 64 * 64;
        pix_per_chunk -= pix_per_chunk % 64;
        if (even < pix_per_chunk) pix_per_chunk = even;
    }
    int64_t cap = pix_per_chunk * spp;
    if (corner) { cap = 1; for (auto& j : jobs) cap = std::max<int64_t>(cap, j.n_ids); }
    else for (int64_t p0 = 0; p0 < n_pix_total; p0 += pix_per_chunk) {
        const uint32_t n = (uint32_t)std::min<int64_t>(pix_per_chunk, n_pix_total - p0);
        jobs.push_back(Job{(uint32_t)p0, n, 0, 0, (uint32_t)p0, n});
    }
    if (cap > 0x7FFFFFFFll) { c->err = "chunk too large"; return FT_ERR_INVALID; }
    const int last_bounce = c->flat.any_reflective ? max_depth : 0;   // no reflective material ⇒ no reflection rays are ever spawned
    if ((rc = ensure_frame_buffers(c, cap, last_bounce > 0)) != FT_OK) return rc;
    const size_t frame_pixels = (size_t)res_h * (size_t)res_v;
    {
        DeviceBuf& ob = q.format == 1 ? c->d_out8 : c->d_out;
        const void* before = ob.p;
        if ((rc = ensure(c, ob, frame_pixels * (q.format == 1 ? 4 : 24))) != FT_OK) return rc;
        if (ob.p != before) c->zero_signature[q.format] = 0;       // a new allocation holds nothing yet
    }
    std::vector<double> jit;
    if (corner) jit = {-0.5, 0.5};                                 // Image.fs:131
    else jit.assign(jitter_xy, jitter_xy + 2 * (size_t)spp);
    // The uploads below travel on the first main stream: a frame still tracing on the second one (FrameSlot::alt), or one whose k_resolve is
    // still to run on the tail stream (it reads the pixel list), reads what they replace.
    if ((corner || !same_list || jit != c->jitter_on_device) && any_pending(c)) {
        int32_t prc = retire_pending(c, nullptr); if (prc != FT_OK) return prc;
    }
    if (corner) {
        if ((rc = upload(c, c->d_pixels, corner_ids)) != FT_OK) return rc;
        if ((rc = upload(c, c->d_out_index, pixels)) != FT_OK) return rc;
    } else if (!same_list) { if ((rc = upload(c, c->d_pixels, pixels)) != FT_OK) return rc; }
    bool jitter_uploaded = false;
    if (jit != c->jitter_on_device) {                              // frames usually reuse the pattern: skip the staged host-to-device copy
        c->jitter_on_device = jit;                                 // (the copy source outlives this call)
        if ((rc = upload(c, c->d_jitter, c->jitter_on_device)) != FT_OK) return rc;
        jitter_uploaded = true;
    }
    // A blocking call retires whatever is in flight first; a deferred one only the frame whose slot (host state, counters, classification
    // buffers) it is about to reuse.
    if (!defer) { int32_t prc = retire_pending(c, nullptr); if (prc != FT_OK) return prc; for (int k = 0; k < kStages; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; } c->accum_open = false; }
    const int turn = c->slot_turn;
    ft_context::FrameSlot& F = c->slots[turn];
    if (F.pending) { int32_t prc = retire_frame(c, F, nullptr); if (prc != FT_OK) return prc; }
    if (defer && !c->accum_open) { for (int k = 0; k < kStages; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; } c->accum_open = true; }
    bool uploads_queued = !same_list || corner || jitter_uploaded; // something this frame's k_classify reads is still on its way on the main stream
    ftk::ClassifyOut cls{};
    if (classify) {
        const size_t n_blocks = (size_t)n_pix_total / 64, n_waves = (n_blocks + 255) / 256;   // one word per k_classify workgroup
        if ((rc = ensure(c, c->d_block_pos[turn], n_blocks * 4)) != FT_OK) return rc;
        if ((rc = ensure(c, c->d_pos_block[turn], n_blocks * 4)) != FT_OK) return rc;
        if (c->d_wave_counts.bytes < n_waves * 4 || c->classify_epoch >= 0x3FFFFEu) {   // entries are tagged with the frame's epoch and never cleared in between
            if ((rc = ensure(c, c->d_wave_counts, std::max<size_t>(n_waves * 4, 4096) + 4096 * 4 + 2048 * 64)) != FT_OK) return rc;   // (+ room for the diagnostic build's stamps)
            FT_HIP(c, hipStreamSynchronize(c->side));              // (a classification of the other slot may still be publishing into the old words)
            FT_HIP(c, hipMemsetAsync(c->d_wave_counts.p, 0, c->d_wave_counts.bytes, c->stream));
            c->classify_epoch = 0;
            uploads_queued = true;
        }
        cls = ftk::ClassifyOut{c->d_block_pos[turn].as<int32_t>(), c->d_pos_block[turn].as<uint32_t>(), c->d_wave_counts.as<uint32_t>()};
    }
    auto* fc = c->d_fc[turn].as<ftk::FrameCounters>();
    // Which main stream.  Two consecutive k_primary launches on ONE stream are an in-order pair: the second is dispatched when the first has
    // drained, and a persistent grid drains slowly (its last batches run on a machine that is mostly idle).  A simple frame - one chunk,
    // k_resolve aside - shares nothing with its predecessor that events do not already order (sample colours: acc_free; counters and
    // classification: per slot; the frame buffer: the tail stream; ray buffers: a pair per main stream), so every other one goes to the second main stream and its
    // workgroups take the CUs as the predecessor's leave them.
    const ft_context::FrameSlot& prevF = c->slots[(turn + ft_context::kSlots - 1) % ft_context::kSlots];
    const bool simple = defer && c->resolve_aside && !corner && c->timing < 2 && jobs.size() == 1;
    if (!simple && any_pending(c, true)) { int32_t prc = retire_pending(c, nullptr); if (prc != FT_OK) return prc; }   // anything else keeps the one-stream order
    const int main_ix = (simple && c->mains > 1 && !uploads_queued && prevF.pending && prevF.simple) ? (prevF.main_ix + 1) % c->mains : 0;   // the next stream after its predecessor's
    const bool alt = main_ix != 0;
    const hipStream_t ms = alt ? c->more_mains[main_ix - 1] : c->stream;
    // chunk counters, statistic stripes, list length, tickets: cleared by the slot's previous frame's last kernel, or by a fill when there was none
    if (!c->fc_clean[turn]) { FT_HIP(c, hipMemsetAsync(fc, 0, sizeof(ftk::FrameCounters), ms)); uploads_queued = true; }
    c->fc_clean[turn] = false;                                     // until this frame's own hand-over is queued

    const ftk::Camera dcam = make_camera(*cam, res_h, res_v);
    const size_t lds = lds_bytes_for(c->flat);
    const int variant = c->variant;
    // Samples per bounce-0 wavefront (slot_at, ft_kernels.hip): 2^group_log2 samples of 64 / 2^group_log2 pixels when the sample count
    // has that power of two in it and the list is made of whole 8x8 blocks.  Narrow bundles pay most where a wave walks a BVH
    // (measured at 1080p x 16, 1 -> 16 samples per wave: bunny through BSP leaves 1.46 -> 1.29 ms, night-house 4.63 -> 4.45).
    int group_log2 = 0;
    {
        const int cap = c->wave_samples_log2 >= 0 ? c->wave_samples_log2 : 4;
        while (group_log2 < cap && !((spp >> group_log2) & 1)) ++group_log2;
        if (corner || !c->pixels_tiled) group_log2 = 0;
    }
    // Option "primary_reserve": workgroup slots a simple frame's k_primary leaves free for the k_resolve of the frame before and the k_classify
    // of the frame after, which otherwise get their slots from its tail.  Measured (tools/reserve_sweep.py): the headline is best at 0
    // (0.2505 ms; 64 free: 0.2545, 256: 0.2715), bunny-bsp12 too; moon x16 gains 3 % at 64.  Default 0.
    const int reserve = simple && c->primary_reserve > 0 ? (int)c->primary_reserve : 0;
    ftk::Launch Lp{ms, std::max(c->n_cu, c->n_cu * c->blocks_primary - reserve), lds, c->variant_primary};
    ftk::Launch Lb{ms, c->n_cu * c->blocks_bounce, lds, variant};
    ftk::Launch Lg{ms, c->n_cu * 8, 0, 0};
    const int resolve_per_cu = c->resolve_blocks_cap > 0 ? std::min(c->resolve_blocks_cap, c->blocks_resolve) : c->blocks_resolve;
    ftk::Launch Lr{ms, c->n_cu * resolve_per_cu, 0, 0};
    ftk::RayBuf rb[2] = {ray_view(c->d_rays[2 * main_ix], c->ray_capacity), ray_view(c->d_rays[2 * main_ix + 1], c->ray_capacity)};

    F.events_used = 0; F.spans.clear();
    auto& spans = F.spans;
    using Span = ft_context::FrameSlot::Span;
    // HIP events between stages.  An event between two dependent kernels costs about 6 us of stream time, so by default ("timing"
    // = 1) only the kernels that trace rays (k_primary, the k_bounce levels) are bracketed; 2 brackets every stage, 0 only the frame.
    // The frame's first event is recorded in front of its first launch on the main stream, behind the waits for other streams' events: on a
    // queued frame it doubles as the start of k_primary's bracket (an event record costs ~5 us of stream time; a frame of 0.27 ms had four
    // between two k_primary launches, now two).
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t boundary = nullptr;
    bool boundary_fresh = false;                                   // `boundary` was recorded right before the next launch
    auto open_frame = [&]() {
        if (ev0) return;
        ev0 = next_event(F);
        if (ev0) (void)hipEventRecord(ev0, ms);
        boundary = ev0; boundary_fresh = true;
    };
    const int timing = c->timing;
    auto timed = [&](int kind, auto&& fn) {
        const bool bracket = timing >= 2 || (timing == 1 && (kind == kStageClosest || kind == kStageShade || kind == kStagePrimary));
        open_frame();
        if (bracket && !boundary_fresh) { boundary = next_event(F); if (boundary) (void)hipEventRecord(boundary, ms); }
        fn();
        if (!bracket) { boundary_fresh = false; return; }
        hipEvent_t b = next_event(F);
        if (b) (void)hipEventRecord(b, ms);
        if (boundary && b) spans.push_back(Span{boundary, b, kind});
        boundary = b; boundary_fresh = true;
    };
    // What decides which blocks k_classify finishes: the scene, the camera, the frame's size and pixel list, the jitter pattern's extent.
    uint64_t zsig = signature;
    {
        auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t k = 0; k < n; ++k) zsig = (zsig ^ b[k]) * 0x100000001B3ull; };
        mix(&dcam, sizeof dcam); mix(&jitter_extent, sizeof jitter_extent);
        if (!rects.empty()) mix(rects.data(), rects.size() * sizeof(ft_rect));
        zsig |= 1ull;                                              // never 0: 0 means "nothing known about the buffer"
    }
    const bool zeros_in_place = classify && c->zero_fill_skip && c->zero_signature[q.format] == zsig;
    c->zero_signature[q.format] = classify ? zsig : 0;
    int n_chunks = 0, n_launches = 0, levels_launched = 0;
    // The whole frame is classified once; the chunks then take consecutive windows of the frame's ACTIVE pixel list, so a sparse
    // frame is one chunk of real work and launches that find their window empty return at once.
    if (classify) {
        const ftk::Primary all{dcam, c->d_pixels.as<uint32_t>(), c->d_jitter.as<double>(), 0u, (uint32_t)n_pix_total, spp, (uint32_t)res_h,
                               (unsigned long long)seed, 1.0 / (double)n_pix_total, 1.0 / (double)res_h, nullptr, nullptr};
        const uint32_t epoch = ++c->classify_epoch;
        // A queued frame's classification reads nothing the frames before it write (its slot's buffers were free once the slot's previous
        // frame was retired above): it goes to the side stream and the main stream waits for its event, so it runs beside the previous
        // frame's k_primary tail and k_resolve instead of behind them.  A blocking frame, or one whose inputs are still being uploaded on
        // the main stream, classifies in line.
        const bool ahead = defer && c->classify_ahead && !uploads_queued;
        hipStream_t cs = ahead ? c->side : ms;
        if (!c->classified) FT_HIP(c, hipEventCreateWithFlags(&c->classified, hipEventDisableTiming));
        else FT_HIP(c, hipStreamWaitEvent(cs, c->classified, 0));  // one classification at a time, whichever streams they are on
        if (ahead) {
            // beside the previous frame's k_resolve (a bandwidth-bound kernel that leaves registers free), not beside the head of its
            // k_primary: started as soon as it was queued, the classification took the first workgroup slots of a grid that fills the chip
            // (measured: k_primary 226 -> 242 us, the 24 us merely moved)
            const ft_context::FrameSlot& prev = c->slots[(turn + ft_context::kSlots - 1) % ft_context::kSlots];
            if (c->classify_after_trace && prev.pending && prev.traced) FT_HIP(c, hipStreamWaitEvent(c->side, prev.traced, 0));
            const ftk::Launch Ls{c->side, Lg.grid, 0, 0};
            ftk::launch_classify(Ls, c->dev_scene, all, cls, jitter_extent, epoch, fc);
            FT_HIP(c, hipEventRecord(c->classified, c->side));
            FT_HIP(c, hipStreamWaitEvent(ms, c->classified, 0));
            boundary_fresh = false;
        } else {
            timed(kStageOther, [&] { ftk::launch_classify(Lg, c->dev_scene, all, cls, jitter_extent, epoch, fc); });
            FT_HIP(c, hipEventRecord(c->classified, ms));
            boundary_fresh = false;
        }
        ++n_launches;
    }
    if (!F.h_report) {
        FT_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&F.h_report), sizeof(ftk::FrameReport), hipHostMallocDefault));
        FT_HIP(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&F.d_report), F.h_report, 0));
    }
    double* const out_rgb = q.format == 1 ? nullptr : c->d_out.as<double>();
    uint8_t* const out_rgba = q.format == 1 ? c->d_out8.as<uint8_t>() : nullptr;
    // queued frames: k_resolve on its own stream (blocking frames have nothing to hide it in).  Frames of one chunk only: a frame cut into many
    // windows (3840x2160x64: 16, most of them empty behind the classification) pays an event pair per window and gains nothing - the windows'
    // small launches already overlap on one stream (measured: 3.46 -> 3.63 ms with it, profiles/r03_z_overlap_by_scene.json)
    const bool aside = defer && c->resolve_aside && !corner && timing < 2 && jobs.size() == 1;
    for (const Job& job : jobs) {
        const uint32_t n_pix = job.n_ids;
        const uint32_t n_samples = n_pix * (uint32_t)spp;
        if (n_chunks > 0) timed(kStageOther, [&] { (void)hipMemsetAsync(&fc->cc, 0, sizeof(ftk::ChunkCounters), ms); });
        ++n_chunks;
        ftk::Primary gen{dcam, c->d_pixels.as<uint32_t>(), c->d_jitter.as<double>(), job.id_base, n_pix, spp,
                         (uint32_t)(corner ? res_h + 1 : res_h), (unsigned long long)seed,
                         1.0 / (double)n_pix, 1.0 / (double)(corner ? res_h + 1 : res_h), nullptr, nullptr};
        if (classify) { gen.counts = &fc->counts; gen.block_map = c->d_pos_block[turn].as<uint32_t>(); }   // pix_base = job.id_base: the window's start in the active list
        gen.group_log2 = (n_pix % 64u == 0u) ? group_log2 : 0;
        const int at = c->acc_turn;
        double* const acc = c->d_acc[at].as<double>();
        if (c->acc_busy[at]) { FT_HIP(c, hipStreamWaitEvent(ms, c->acc_free[at], 0)); c->acc_busy[at] = false; boundary_fresh = false; }   // a k_resolve on `tail` may still be reading this copy
        timed(kStagePrimary, [&] { ftk::launch_primary(Lp, c->dev_scene, gen, rb[1], acc, n_samples, max_depth, fc); });
        ++n_launches;
        // Bounces >= 1: one k_bounce per level of the reflection tree, as many as the previous frame of this signature had (+ 1).
        // With "timing" = 1 the whole region is one bracket (kind shade): a bracket per launch costs more than a small level does.
        const bool hinted = c->level_hint && c->staged_hint >= 0 && c->staged_signature == signature;
        const int n_levels = hinted ? std::min(last_bounce, c->staged_hint + 1) : last_bounce;
        levels_launched = n_levels;
        auto bounces = [&](auto&& stage) {
            for (int b = 1; b <= n_levels; ++b) {
                stage(kStageShade, [&] { ftk::launch_bounce(Lb, c->dev_scene, gen, rb[b & 1], rb[(b + 1) & 1], acc, n_samples, b, max_depth, b == n_levels && n_levels < last_bounce, fc); });
                ++n_launches;
            }
        };
        if (timing >= 2) bounces(timed);
        else if (n_levels >= 1) timed(kStageShade, [&] { bounces([](int, auto&& fn) { fn(); }); });
        if (&job == &jobs.back()) {                                // where the frame's tracing ends: the event that closed its last bracket, if that is still the stream's last entry
            if (boundary_fresh && boundary) F.traced = boundary;
            else { F.traced = next_event(F); if (!F.traced) { c->err = "hipEventCreate failed"; return FT_ERR_HIP; } FT_HIP(c, hipEventRecord(F.traced, ms)); }
        }
        if (!aside) for (int k = 0; k < ft_context::kAcc; ++k) if (c->acc_busy[k]) {   // a queued frame's k_resolve may still be writing the frame on `tail`: frames reach d_out in order
            FT_HIP(c, hipStreamWaitEvent(ms, c->acc_free[k], 0)); c->acc_busy[k] = false; boundary_fresh = false;
        }
        if (corner) timed(kStageResolve, [&] { ftk::launch_resolve_corner(Lg, acc, n_samples, job.w, job.h, c->d_out_index.as<uint32_t>() + job.out_base, out_rgb, out_rgba); });
        else {
            const bool last_job = &job == &jobs.back();            // the frame's last kernel hands the counters over (FrameReport)
            ftk::ResolveArgs ra{acc, n_samples, classify ? &fc->counts : nullptr, job.id_base, n_pix, spp,
                                classify ? c->d_pos_block[turn].as<uint32_t>() : nullptr, (classify && n_chunks == 1 && !zeros_in_place) ? c->d_block_pos[turn].as<int32_t>() : nullptr,
                                (uint32_t)(n_pix_total / 64), c->d_pixels.as<uint32_t>(), out_rgb, out_rgba, (uint32_t)gen.group_log2, fc, last_job ? F.d_report : nullptr};
            if (aside) {
                // behind the chunk's tracing kernels, on its own stream: the main stream goes straight on with the next chunk or frame
                hipEvent_t et = last_job ? F.traced : next_event(F);
                if (!et) { c->err = "hipEventCreate failed"; return FT_ERR_HIP; }
                if (!last_job) FT_HIP(c, hipEventRecord(et, ms));
                FT_HIP(c, hipStreamWaitEvent(c->tail, et, 0));
                ftk::Launch La = Lr; La.stream = c->tail;
                ftk::launch_resolve(La, ra);
                if (!c->acc_free[at]) FT_HIP(c, hipEventCreateWithFlags(&c->acc_free[at], hipEventDisableTiming));
                FT_HIP(c, hipEventRecord(c->acc_free[at], c->tail));
                c->acc_busy[at] = true;
                c->acc_turn = (c->acc_turn + 1) % ft_context::kAcc;
                boundary_fresh = false;
            } else timed(kStageResolve, [&] { ftk::launch_resolve(Lr, ra); });
            if (last_job) c->fc_clean[turn] = true;
        }
        ++n_launches;
    }
    if (!c->fc_clean[turn]) { ftk::launch_report(Lg, fc, F.d_report); c->fc_clean[turn] = true; }   // corner frames end in k_resolve_corner: the hand-over is a launch of its own
    c->last_n_pix = n_pix_total; c->last_res_h = res_h; c->last_res_v = res_v; c->last_format = q.format;
    if (defer && out) {                                            // ft_render_enqueue_into: the frame's way out is queued behind its last kernel
        const hipStream_t cs = aside ? c->tail : ms;
        int32_t crc = copy_frame_out(c, out, q.format, cs);
        if (crc != FT_OK) return crc;
        boundary_fresh = false;
    }
    open_frame();
    if (aside) { ev1 = next_event(F); if (ev1) (void)hipEventRecord(ev1, c->tail); }                  // the frame ends where its last k_resolve (and copy) does
    else if (boundary_fresh) ev1 = boundary;
    else { ev1 = next_event(F); if (ev1) (void)hipEventRecord(ev1, ms); }
    FT_HIP(c, hipGetLastError());
    F.signature = signature; F.levels_launched = levels_launched; F.last_bounce = last_bounce;
    F.simple = simple; F.alt = alt; F.main_ix = main_ix;
    F.done = ev1;                                                  // nothing follows the last kernel: its end is the frame's
    F.ev0 = ev0; F.ev1 = ev1; F.pending = true; F.wall0 = wall0; F.timing = timing;
    F.rays_primary = 0; for (auto& j : jobs) F.rays_primary += (uint64_t)j.n_ids * (uint64_t)spp;
    F.n_pix_total = n_pix_total; F.spp = spp; F.n_launches = n_launches; F.n_chunks = n_chunks; F.classify = classify; F.format = q.format;
    c->last_n_pix = n_pix_total; c->last_res_h = res_h; c->last_res_v = res_v; c->last_format = q.format;
    c->slot_turn = (c->slot_turn + 1) % ft_context::kSlots;
    if (defer) return FT_OK;                                       // ft_render_enqueue: the frame is retired by a later call
    int32_t rrc = retire_frame(c, F, stats);
    if (rrc != FT_OK) return rrc;
    if (out) { int32_t frc = fetch_single(c, out, q.format); if (frc != FT_OK) return frc; }   // out == NULL: the frame stays in HBM
    if (stats) stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return FT_OK;
}

// Wait for a queued frame, add its stage times to the context's sums and fill its statistics.
static int32_t retire_frame(ft_context* c, ft_context::FrameSlot& F, ft_stats* stats) {
    if (!F.pending) return FT_OK;
    F.pending = false;
    if (F.done) FT_HIP(c, hipEventSynchronize(F.done)); else FT_HIP(c, hipStreamSynchronize(c->stream));
    const ftk::RenderCounters hrc = F.h_report->total;              // the stripes, summed by the frame's last kernel
    const bool classify_failed = F.h_report->classify_error != 0;
    // How deep this frame's rays went in numbers worth a launch (more than "follow_below" rays; levels followed in registers count
    // theirs too): the next frame of the same signature launches that many levels + 1, and that last one follows what is left.
    int deepest = 0;
    const int64_t few = c->follow_below >= 0 ? c->follow_below : 8ll * c->n_cu;   // -1: two rays per SIMD
    while (deepest + 1 <= ftk::kMaxBounce && (int64_t)F.h_report->n_rays[deepest + 1] > few) ++deepest;
    c->staged_hint = deepest; c->staged_signature = F.signature;
    if (F.classify && !classify_failed) { c->active_hint = (int64_t)F.h_report->n_pix_active; c->active_signature = F.signature; }
    const int timing = F.timing; const int32_t spp = F.spp; const int64_t n_pix_total = F.n_pix_total; const bool classify = F.classify;
    hipEvent_t ev0 = F.ev0, ev1 = F.ev1;
    double bracketed = 0.0, traced = 0.0;
    for (auto& s : F.spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
        c->k_ms[s.kind] += ms; c->k_launches[s.kind]++; bracketed += ms;
        if (s.kind == kStageClosest || s.kind == kStageShade || s.kind == kStagePrimary) traced += ms;
    }
    float total = 0;
    if (ev0 && ev1) (void)hipEventElapsedTime(&total, ev0, ev1);
    if (timing < 2) c->k_ms[kStageOther] += std::max(0.0, (double)total - bracketed);   // everything that was not bracketed: the fill, k_classify, k_resolve
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rays_primary = F.rays_primary;
        stats->rays_shadow = hrc.rays_shadow; stats->rays_reflect = hrc.rays_reflect;
        // rays the device really traced: primaries of pixel blocks k_classify finished (Colour.Zero for the whole block, no ray generated)
        // are part of rays_primary and of the reference-equivalent count, not of rays_traced
        stats->rays_primary_culled = (uint64_t)hrc.pixels_culled * (uint64_t)spp;
        stats->rays_traced = stats->rays_primary - std::min<uint64_t>(stats->rays_primary, stats->rays_primary_culled) + stats->rays_shadow + stats->rays_reflect;
        stats->rays_reference_equivalent = (double)stats->rays_primary + hrc.ref_equiv;
        stats->hits_primary = hrc.hits_primary; stats->csg_overflow = hrc.csg_overflow;
        stats->rays_shadow_primary = hrc.rays_shadow_primary; stats->rays_reflect_primary = hrc.rays_reflect_primary;
        stats->kernel_ms = total; stats->trace_kernel_ms = traced;
        {   // Bytes the pipeline has to move by construction of its data layout (ft_device.h).  P generated primaries, Rp / R
            // reflection rays spawned by k_primary / in all, Hb hits shaded by the k_bounce levels.
            const uint64_t P = stats->rays_primary - std::min<uint64_t>(stats->rays_primary, stats->rays_primary_culled);
            const uint64_t RR = hrc.rays_reflect, Rp = hrc.rays_reflect_primary;
            const uint64_t Hb = hrc.hits_total - std::min(hrc.hits_total, hrc.hits_primary);     // hits shaded by k_bounce
            stats->hits_total = hrc.hits_total;
            stats->rays_tail = 0;
            stats->algorithmic_bytes_primary = P * (ftk::kPixelIdBytes + ftk::kAccBytes) + Rp * ftk::kRayRecBytes;
            stats->algorithmic_bytes_closest = 0;
            stats->algorithmic_bytes_shade = RR * ftk::kRayRecBytes + Hb * 2 * ftk::kAccBytes + (RR - std::min(RR, Rp)) * ftk::kRayRecBytes;   // k_bounce: rays in, colours read-modify-written, rays out
            const uint64_t out_px = F.format == 1 ? 4 : 24, blocks = (uint64_t)n_pix_total / 64;
            stats->algorithmic_bytes = stats->algorithmic_bytes_primary + stats->algorithmic_bytes_shade +
                                       P * ftk::kAccBytes + out_px * (uint64_t)n_pix_total + 4 * (uint64_t)n_pix_total +  // + k_resolve: samples in, pixels out, pixel ids
                                       (classify ? blocks * 16 : 0ull);                                                 // + k_classify: two ids in, two words out per block
        }
        stats->n_launches = F.n_launches; stats->n_chunks = F.n_chunks;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - F.wall0).count();
    }
    if (classify_failed) { c->err = "k_classify: a bounded wait ran out (device error)"; return FT_ERR_HIP; }
    if (hrc.csg_overflow) {
        c->err = "CSG hit list overflow on " + std::to_string(hrc.csg_overflow) + " rays: raise csg_mesh_capacity (ft_set_option)";
        return FT_ERR_OVERFLOW;
    }
    return FT_OK;
}

// Retire every queued frame, oldest first; `stats` receives the newest one's.
static int32_t retire_pending(ft_context* c, ft_stats* stats) {
    int32_t rc = FT_OK;
    int last = -1;
    for (int k = 0; k < ft_context::kSlots; ++k) if (c->slots[(c->slot_turn + k) % ft_context::kSlots].pending) last = k;
    for (int k = 0; k < ft_context::kSlots; ++k) {                  // oldest first; the statistics asked for are the newest frame's
        ft_context::FrameSlot& f = c->slots[(c->slot_turn + k) % ft_context::kSlots];
        if (f.pending) { int32_t r = retire_frame(c, f, k == last ? stats : nullptr); if (r != FT_OK) rc = r; }
    }
    return rc;
}

/* Pipelined rendering: queue the frame and return; see functracer_hip.h.  On a context over several devices every device queues
 * its bands of the frame on its own stream. */
static int32_t enqueue(ft_context* c, const RenderRequest& q) {
    if (!c) return FT_ERR_INVALID;
    return render_frame(c, q, nullptr, nullptr, true);
}
int32_t ft_render_enqueue(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                          int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles) {
    return enqueue(c, RenderRequest{cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, 0});
}
int32_t ft_render_enqueue_rgba8(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                                int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles) {
    return enqueue(c, RenderRequest{cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, 1});
}
/* A queued frame that also leaves the device: the copy into host_out (res_v x res_h x 3 doubles, or x 4 bytes with rgba8 != 0) is queued
 * behind the frame's last kernel and is complete when ft_render_wait returns (or when a later call retires the frame).  host_out should
 * come from ft_host_alloc: the copy is then one DMA beside the next frame's tracing - a stream of RGBA8 frames reaches the host at the
 * rate the device renders them. */
int32_t ft_render_enqueue_into(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                               int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, int32_t rgba8, void* host_out) {
    if (!c || !host_out) return FT_ERR_INVALID;
    return render_frame(c, RenderRequest{cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, rgba8 ? 1 : 0}, host_out, nullptr, true);
}
int32_t ft_render_wait(ft_context* c, ft_stats* stats) {
    if (!c) return FT_ERR_INVALID;
    if (c->host_only) return FT_ERR_NO_DEVICE;
    if (stats) std::memset(stats, 0, sizeof *stats);
    std::vector<ft_context*> devs{c};
    devs.insert(devs.end(), c->peers.begin(), c->peers.end());
    int32_t rc = FT_OK;
    for (ft_context* d : devs) {
        FT_HIP(c, hipSetDevice(d->device));
        ft_stats sd;
        std::memset(&sd, 0, sizeof sd);
        const int32_t r = retire_pending(d, &sd);
        d->accum_open = false;
        if (r != FT_OK && rc == FT_OK) { rc = r; if (d != c) c->err = d->err; }
        if (stats) { const double wall = std::max(stats->wall_ms, sd.wall_ms); add_stats(stats, sd); stats->wall_ms = wall; }
    }
    return rc;
}

int32_t ft_get_kernel_times(ft_context* c, double ms[5], int32_t launches[5]) {
    if (!c || !ms || !launches) return FT_ERR_INVALID;
    for (int k = 0; k < kStages; ++k) { ms[k] = c->k_ms[k]; launches[k] = c->k_launches[k]; }
    for (ft_context* p : c->peers) for (int k = 0; k < kStages; ++k) { ms[k] = std::max(ms[k], p->k_ms[k]); launches[k] = std::max(launches[k], p->k_launches[k]); }   // the slowest device's
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ debug / tests
static int32_t debug_closest(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t* hit, double* t, double* p, double* nrm, double* colour);
int32_t ft_debug_closest(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t* hit, double* t, double* p, double* nrm, double* colour) {
    if (!c) return FT_ERR_INVALID;
    return with_growing_hit_lists(c, [&] { return debug_closest(c, origins, dirs, n, hit, t, p, nrm, colour); });
}
static int32_t debug_closest(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t* hit, double* t, double* p, double* nrm, double* colour) {
    if (!c || !origins || !dirs || n < 0 || !hit || !t || !p || !nrm || !colour) return FT_ERR_INVALID;
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    if (!c->committed) { c->err = "scene not committed"; return FT_ERR_STATE; }
    if (n == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    int32_t rc;
    const size_t N = (size_t)n;
    if ((rc = ensure(c, c->d_dbg_in, N * 48)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_dbg_out, N * (4 + 8 + 72))) != FT_OK) return rc;
    double* din = c->d_dbg_in.as<double>();
    FT_HIP(c, hipMemcpyAsync(din, origins, N * 24, hipMemcpyHostToDevice, c->stream));
    FT_HIP(c, hipMemcpyAsync(din + 3 * N, dirs, N * 24, hipMemcpyHostToDevice, c->stream));
    c->fc_clean[0] = false;
    FT_HIP(c, hipMemsetAsync(c->d_fc[0].p, 0, sizeof(unsigned long long), c->stream));   // the overflow count of this query
    double* dt = c->d_dbg_out.as<double>();
    double* dp = dt + N; double* dn = dp + 3 * N; double* dc = dn + 3 * N;
    int32_t* dh = reinterpret_cast<int32_t*>(dc + 3 * N);
    const size_t lds = lds_bytes_for(c->flat);
    ftk::Launch L{c->stream, c->n_cu * 4, lds, 0};
    ftk::launch_debug_closest(L, c->dev_scene, din, din + 3 * N, (uint32_t)n, dh, dt, dp, dn, dc, c->d_fc[0].as<unsigned long long>());
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(t, dt, N * 8, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(p, dp, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(nrm, dn, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(colour, dc, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(hit, dh, N * 4, hipMemcpyDeviceToHost, c->stream));
    unsigned long long n_overflow = 0;
    FT_HIP(c, hipMemcpyAsync(&n_overflow, c->d_fc[0].p, sizeof n_overflow, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipStreamSynchronize(c->stream));
    if (n_overflow) { c->err = "CSG hit list overflow"; return FT_ERR_OVERFLOW; }
    return FT_OK;
}

static int32_t debug_blocked(ft_context* c, const double* origins, const double* dirs, const double* max_dist, int64_t n, int32_t* blocked);
int32_t ft_debug_blocked(ft_context* c, const double* origins, const double* dirs, const double* max_dist, int64_t n, int32_t* blocked) {
    if (!c) return FT_ERR_INVALID;
    return with_growing_hit_lists(c, [&] { return debug_blocked(c, origins, dirs, max_dist, n, blocked); });
}
static int32_t debug_blocked(ft_context* c, const double* origins, const double* dirs, const double* max_dist, int64_t n, int32_t* blocked) {
    if (!c || !origins || !dirs || !max_dist || n < 0 || !blocked) return FT_ERR_INVALID;
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    if (!c->committed) { c->err = "scene not committed"; return FT_ERR_STATE; }
    if (n == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    int32_t rc;
    const size_t N = (size_t)n;
    if ((rc = ensure(c, c->d_dbg_in, N * 56)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_dbg_out, N * 4)) != FT_OK) return rc;
    double* din = c->d_dbg_in.as<double>();
    FT_HIP(c, hipMemcpyAsync(din, origins, N * 24, hipMemcpyHostToDevice, c->stream));
    FT_HIP(c, hipMemcpyAsync(din + 3 * N, dirs, N * 24, hipMemcpyHostToDevice, c->stream));
    FT_HIP(c, hipMemcpyAsync(din + 6 * N, max_dist, N * 8, hipMemcpyHostToDevice, c->stream));
    c->fc_clean[0] = false;
    FT_HIP(c, hipMemsetAsync(c->d_fc[0].p, 0, sizeof(unsigned long long), c->stream));   // the overflow count of this query
    const size_t lds = lds_bytes_for(c->flat);
    ftk::Launch L{c->stream, c->n_cu * 4, lds, 0};
    ftk::launch_debug_blocked(L, c->dev_scene, din, din + 3 * N, din + 6 * N, (uint32_t)n, c->d_dbg_out.as<int32_t>(), c->d_fc[0].as<unsigned long long>());
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(blocked, c->d_dbg_out.p, N * 4, hipMemcpyDeviceToHost, c->stream));
    unsigned long long n_overflow = 0;
    FT_HIP(c, hipMemcpyAsync(&n_overflow, c->d_fc[0].p, sizeof n_overflow, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipStreamSynchronize(c->stream));
    if (n_overflow) { c->err = "CSG hit list overflow"; return FT_ERR_OVERFLOW; }
    return FT_OK;
}

// getColourForRay (Shading.fs:131-139) for explicit rays through the device path: the rays enter k_bounce as level 0 with weight 1
// and are followed to their end, so closest hit, shadow queries, shaders and up to max_depth reflection bounces run exactly as they
// do for a frame's samples.  Streams of soft lights are keyed with seed 0 and sample = ray index.
static int32_t debug_colour(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t max_depth, double* rgb);
int32_t ft_debug_colour(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t max_depth, double* rgb) {
    if (!c) return FT_ERR_INVALID;
    return with_growing_hit_lists(c, [&] { return debug_colour(c, origins, dirs, n, max_depth, rgb); });
}
static int32_t debug_colour(ft_context* c, const double* origins, const double* dirs, int64_t n, int32_t max_depth, double* rgb) {
    if (!c || !origins || !dirs || n < 0 || !rgb || max_depth < 0) return FT_ERR_INVALID;
    if (max_depth > ftk::kMaxBounce) { c->err = "max_depth above 16"; return FT_ERR_UNSUPPORTED; }
    if (n >= (1ll << 30)) return FT_ERR_INVALID;
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    if (!c->committed) { c->err = "scene not committed"; return FT_ERR_STATE; }
    if (n == 0) return FT_OK;
    FT_HIP(c, hipSetDevice(c->device));
    int32_t rc;
    if ((rc = ensure_frame_buffers(c, n, true)) != FT_OK) return rc;
    const size_t N = (size_t)n, cap = (size_t)c->ray_capacity;
    std::vector<double> soa(7 * N);
    std::vector<uint32_t> slot(N);
    for (size_t i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) { soa[(size_t)k * N + i] = origins[3 * i + k]; soa[(size_t)(3 + k) * N + i] = dirs[3 * i + k]; }
        soa[6 * N + i] = 1.0; slot[i] = (uint32_t)i;
    }
    auto* fc = c->d_fc[0].as<ftk::FrameCounters>();
    c->fc_clean[0] = false;
    FT_HIP(c, hipMemsetAsync(fc, 0, sizeof(ftk::FrameCounters), c->stream));
    const ftk::RayBuf rb0 = ray_view(c->d_rays[0], c->ray_capacity), rb1 = ray_view(c->d_rays[1], c->ray_capacity);
    for (int k = 0; k < 7; ++k) FT_HIP(c, hipMemcpyAsync(c->d_rays[0].as<double>() + (size_t)k * cap, soa.data() + (size_t)k * N, N * 8, hipMemcpyHostToDevice, c->stream));
    FT_HIP(c, hipMemcpyAsync(rb0.slot, slot.data(), N * 4, hipMemcpyHostToDevice, c->stream));
    const uint32_t n_rays = (uint32_t)n;
    FT_HIP(c, hipMemcpyAsync(&fc->cc.n_rays[0], &n_rays, 4, hipMemcpyHostToDevice, c->stream));
    FT_HIP(c, hipMemsetAsync(c->d_acc[0].p, 0, 3 * N * 8, c->stream));
    const size_t lds = lds_bytes_for(c->flat);
    ftk::Launch Lt{c->stream, c->n_cu * c->blocks_bounce, lds, c->variant};
    ftk::Primary gen{};
    gen.pixel_ids = nullptr; gen.pix_base = 0; gen.n_pix = n_rays; gen.spp = 1; gen.inv_n_pix = 1.0 / (double)n_rays; gen.seed = 0ull; gen.counts = nullptr; gen.block_map = nullptr;
    ftk::launch_bounce(Lt, c->dev_scene, gen, rb0, rb1, c->d_acc[0].as<double>(), n_rays, 0, max_depth, true, fc);   // level 0, followed to the end
    FT_HIP(c, hipGetLastError());
    std::vector<double> planes(3 * N);
    FT_HIP(c, hipMemcpyAsync(planes.data(), c->d_acc[0].p, 3 * N * 8, hipMemcpyDeviceToHost, c->stream));
    struct { ftk::RenderCounters stats[ftk::kStatStripes]; } tail;
    FT_HIP(c, hipMemcpyAsync(&tail, &fc->stats[0], sizeof tail, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < N; ++i) { rgb[3 * i] = planes[i]; rgb[3 * i + 1] = planes[N + i]; rgb[3 * i + 2] = planes[2 * N + i]; }
    unsigned long long ovf = 0;
    for (int k = 0; k < ftk::kStatStripes; ++k) ovf += tail.stats[k].csg_overflow;
    if (ovf) { c->err = "CSG hit list overflow"; return FT_ERR_OVERFLOW; }
    return FT_OK;
}

/* Diagnostic builds (-DFT_STAMPS): the s_memrealtime stamps k_classify's workgroups left behind (8 per workgroup). */
int32_t ft_debug_classify_stamps(ft_context* c, unsigned long long* out, int32_t n_groups) {
    if (!c || !out || n_groups < 1 || n_groups > 2048 || !c->d_wave_counts.p) return FT_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return FT_ERR_HIP;
    return hipMemcpy(out, c->d_wave_counts.as<uint32_t>() + 4096, (size_t)n_groups * 64, hipMemcpyDeviceToHost) == hipSuccess ? FT_OK : FT_ERR_HIP;
}

int32_t ft_debug_scene_info(ft_context* c, int64_t out[12]) {
    if (!c || !out) return FT_ERR_INVALID;
    if (!c->committed) { c->err = "scene not committed"; return FT_ERR_STATE; }
    const fth::FlatScene& f = c->flat;
    out[0] = (int64_t)f.leaves.size(); out[1] = (int64_t)f.program.size(); out[2] = (int64_t)f.meshes.size(); out[3] = (int64_t)f.nodes.size() - f.bvh_nodes;
    out[4] = (int64_t)f.bsp_leaves.size() - f.bvh_leaves; out[5] = (int64_t)(f.tris.size() / 9) - f.bvh_tris; out[6] = f.csg_capacity; out[7] = f.bsp_stack_capacity;   // BSP-only: excludes the device-side BVH
    int64_t bounded = 0; for (size_t k = 0; k + 1 < f.item_pc.size(); ++k) if (f.cull_items[8 * k + 3] < 1e30f) ++bounded;
    out[8] = (int64_t)f.item_pc.size() - 1; out[9] = bounded; out[10] = f.unbounded ? 1 : 0; out[11] = f.cull_bundle ? (int64_t)(f.cull_rows.size() / 3) : -1;
    return FT_OK;
}

int32_t ft_debug_slice(const double p0[3], const double n[3], const double tri[9], double above[18], int32_t* n_above, double below[18], int32_t* n_below) {
    if (!p0 || !n || !tri || !above || !below || !n_above || !n_below) return FT_ERR_INVALID;
    std::vector<double> a, b; std::string err;
    int32_t rc = fth::slice_triangle(p0, n, tri, a, b, err);
    if (rc != FT_OK) return rc;
    *n_above = (int32_t)(a.size() / 9); *n_below = (int32_t)(b.size() / 9);
    if (!a.empty()) std::memcpy(above, a.data(), a.size() * 8);
    if (!b.empty()) std::memcpy(below, b.data(), b.size() * 8);
    return FT_OK;
}

int32_t ft_debug_devices(ft_context* c, int32_t* ordinals, int32_t capacity) {   // the device ordinals behind a context, in order; returns how many
    if (!c || capacity < 0 || (capacity > 0 && !ordinals)) return FT_ERR_INVALID;
    if (c->host_only) return 0;
    int32_t n = 0;
    if (n < capacity) ordinals[n] = c->device;
    ++n;
    for (ft_context* p : c->peers) { if (n < capacity) ordinals[n] = p->device; ++n; }
    return n;
}

int32_t ft_quantise_rgba8(const double* rgb, int64_t n_pixels, uint8_t* out) {   // Image.fs:36, Math.fs:12-16
    if (!rgb || !out || n_pixels < 0) return FT_ERR_INVALID;
    for (int64_t i = 0; i < n_pixels; ++i) {
        for (int k = 0; k < 3; ++k) {
            double x = rgb[3 * i + k];
            if (x > 1.0) x = 1.0; else if (x < 0.0) x = 0.0;                     // NaN passes through the clamp unchanged
            x = x * 255.0;
            out[4 * i + k] = (x != x) ? 0 : (uint8_t)x;                          // truncation, not rounding
        }
        out[4 * i + 3] = 255;
    }
    return FT_OK;
}

} // extern "C"
