// ft_capi.cpp — the C ABI of libfunctracer_hip.so (include/functracer_hip.h): context, scene
// builder, HBM residency of the flattened scene and the per-frame wavefront pipeline driver.
// Reference citations are relative to /root/reference/FuncTracer/.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/functracer_hip.h"
#include "ft_device.h"
#include "ft_scene.h"

namespace ftk {
int occupancy_blocks_primary(size_t lds_bytes, int variant);
int occupancy_blocks_closest(size_t lds_bytes, int variant);
int occupancy_blocks_shade(size_t lds_bytes, int variant);
}

struct DeviceBuf {
    void* p = nullptr; size_t bytes = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct ft_context {
    std::vector<ft_context*> peers;      // multi-device contexts: one more single-device context per extra GPU (scene replicated)
    bool host_only = false;
    int device = -1;
    int n_cu = 0;
    uint32_t n_stat_slots = 0;      // one RenderCounters slot per wave of the largest persistent grid: n_cu x 8 blocks x 4 waves
    hipStream_t stream = nullptr;
    std::string err;

    fth::SceneGraph graph;
    fth::FlatScene flat;
    bool committed = false;

    int64_t chunk_samples = 16ll << 20;   // measured: 8 Mi costs 10-25 % (more, smaller launches), 32 Mi slows k_shade on many-light scenes
    bool coherent_waves = true;     // diagnostic: 0 routes every wavefront through the incoherent paths
    int timing = 1;                 // HIP events: 0 around the frame only, 1 + around every k_closest / k_shade, 2 around every stage
    bool classify_pixels = true;    // k_classify: pixel blocks that cannot see any item are finished before any ray is generated
    bool fused_primary = true;      // bounce 0 through k_primary (one kernel) instead of k_closest + k_shade
    int64_t tail_rays = 262144;      // a bounce that starts with fewer rays is finished by k_tail (0 = never)

    // scene in HBM
    DeviceBuf d_leaves, d_m2w, d_materials, d_lights, d_program, d_meshes, d_nodes, d_bleaves, d_tris, d_culls, d_tri_orig, d_textures, d_tex_pixels, d_cull_items, d_cull_rows, d_item_pc, d_active_ids, d_active_pos, d_wide, d_mesh_wide, d_coarse, d_block_flags;
    ftk::DevScene dev_scene{};
    // frame buffers in HBM
    DeviceBuf d_rays[2], d_hits, d_hit_list, d_touched, d_acc, d_out, d_pixels, d_jitter, d_cc, d_rc, d_dbg_in, d_dbg_out;
    int64_t ray_capacity = 0;
    // Per-frame host state.  Two slots, so that one frame can be queued while the previous one still runs (ft_render_enqueue).
    struct FrameSlot {
        std::vector<hipEvent_t> events; size_t events_used = 0;
        struct Span { hipEvent_t a, b; int kind; };
        std::vector<Span> spans;
        hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
        ftk::RenderCounters* h_rc = nullptr;    // pinned landing place of the frame's statistics
        bool pending = false;
        uint64_t rays_primary = 0; int64_t n_pix_total = 0; int32_t spp = 0, n_launches = 0, n_chunks = 0, timing = 1; bool classify = false;
        std::chrono::steady_clock::time_point wall0;
    };
    FrameSlot slots[2];
    int slot_turn = 0;
    bool csg_auto_grow = true;   // ft_render: double csg_mesh_capacity and render again when a hit list overflows
    bool accum_open = false;        // kernel times are being summed over pipelined frames (reset by the next enqueue after a wait)
    // pixel list of the last render, cached across calls with the same resolution and tiles
    std::vector<uint32_t> pixels;
    std::vector<double> jitter_on_device;   // what d_jitter holds
    std::vector<ft_rect> pixel_rects;
    bool pixels_whole = false, pixels_corner = false;
    DeviceBuf d_out_index;
    int64_t last_n_pix = 0;
    int32_t last_res_h = 0, last_res_v = 0;
    double k_ms[5] = {0, 0, 0, 0, 0};
    int32_t k_launches[5] = {0, 0, 0, 0, 0};
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

int32_t ensure_frame_buffers(ft_context* c, int64_t cap) {
    if (cap <= c->ray_capacity) return FT_OK;
    int32_t rc;
    for (int i = 0; i < 2; ++i) if ((rc = ensure(c, c->d_rays[i], (size_t)cap * (7 * 8 + 4))) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_hits, (size_t)cap * 16)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_hit_list, (size_t)cap * 4)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_touched, (size_t)cap)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_acc, (size_t)cap * 24)) != FT_OK) return rc;
    c->ray_capacity = cap;
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
    c->n_cu = prop.multiProcessorCount;
    c->n_stat_slots = (uint32_t)c->n_cu * 8u * (uint32_t)(ftk::kBlock / 64);   // every grid is n_cu x (at most 8) blocks (clamp_blocks, Lg)
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
    for (ft_context* p : c->peers) ft_destroy(p);
    c->peers.clear();
    if (!c->host_only) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        DeviceBuf* bufs[] = {&c->d_leaves, &c->d_m2w, &c->d_materials, &c->d_lights, &c->d_program, &c->d_meshes, &c->d_nodes, &c->d_bleaves, &c->d_tris, &c->d_culls, &c->d_tri_orig, &c->d_textures, &c->d_tex_pixels, &c->d_cull_items, &c->d_cull_rows, &c->d_item_pc, &c->d_active_ids, &c->d_active_pos, &c->d_wide, &c->d_mesh_wide, &c->d_coarse, &c->d_block_flags, &c->d_out_index,
                             &c->d_rays[0], &c->d_rays[1], &c->d_hits, &c->d_hit_list, &c->d_touched, &c->d_acc, &c->d_out, &c->d_pixels, &c->d_jitter, &c->d_cc, &c->d_rc,
                             &c->d_dbg_in, &c->d_dbg_out};
        for (auto* b : bufs) release(*b);
        for (auto& f : c->slots) { if (f.h_rc) { (void)hipHostFree(f.h_rc); f.h_rc = nullptr; } for (auto e : f.events) (void)hipEventDestroy(e); f.events.clear(); }
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
    if (!std::strcmp(key, "timing")) { if (value < 0 || value > 2) return FT_ERR_INVALID; c->timing = (int)value; for (ft_context* p : c->peers) p->timing = (int)value; return FT_OK; }
    if (!std::strcmp(key, "classify_pixels")) { c->classify_pixels = value != 0; for (ft_context* p : c->peers) p->classify_pixels = value != 0; return FT_OK; }
    if (!std::strcmp(key, "fused_primary")) { c->fused_primary = value != 0; for (ft_context* p : c->peers) p->fused_primary = value != 0; return FT_OK; }
    if (!std::strcmp(key, "csg_auto_grow")) { c->csg_auto_grow = value != 0; return FT_OK; }
    if (!std::strcmp(key, "tail_rays")) { if (value < 0 || value > 0x7FFFFFFF) return FT_ERR_INVALID; c->tail_rays = value; for (ft_context* p : c->peers) p->tail_rays = value; return FT_OK; }
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

int32_t ft_scene_commit(ft_context* c) {
    if (!c) return FT_ERR_INVALID;
    int32_t rc = c->graph.flatten(c->flat, c->err);
    if (rc != FT_OK) return rc;
    if (c->host_only) { c->committed = true; return FT_OK; }
    if ((rc = upload_scene(c)) != FT_OK) return rc;
    for (ft_context* p : c->peers) {                                // replicate the flattened scene on every other device
        p->flat = c->flat;
        if ((rc = upload_scene(p)) != FT_OK) { c->err = p->err; return rc; }
    }
    return FT_OK;
}

static int32_t upload_scene(ft_context* c) {
    int32_t rc;
    FT_HIP(c, hipSetDevice(c->device));
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
    if ((rc = upload(c, c->d_cull_items, f.cull_items)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_cull_rows, f.cull_rows)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_item_pc, f.item_pc)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_wide, f.wide)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_mesh_wide, f.mesh_wide)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_coarse, f.coarse_boxes)) != FT_OK) return rc;
    if ((rc = upload(c, c->d_tri_orig, f.tri_orig)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_cc, sizeof(ftk::ChunkCounters))) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_rc, sizeof(ftk::RenderCounters) * ((size_t)c->n_stat_slots + 2))) != FT_OK) return rc;   // + one slot's worth for the frame's PixCount
    FT_HIP(c, hipStreamSynchronize(c->stream));
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
    S.n_items = (int32_t)f.item_pc.size() - 1; S.n_cull_rows = f.cull_bundle ? (int32_t)(f.cull_rows.size() / 3) : -1;
    S.n_leaves = (int32_t)f.leaves.size(); S.n_lights = (int32_t)f.lights.size();
    S.csg_cap = f.csg_capacity; S.stack_cap = f.stack_capacity;
    S.lane_fold = lane_fold_for(f); S.csg_rows = (f.csg_capacity + S.lane_fold - 1) / S.lane_fold;
    S.shadow_rays_per_hit = 0;
    for (auto& l : f.lights) S.shadow_rays_per_hit += (l.kind == ftd::LT_SOFT) ? l.samples : 1;   // Shading.fs:24-42
    c->committed = true;
    return FT_OK;
}

// Copy the pixels of the last ft_render from HBM into the caller's frame (row 0 = top, Image.fs:39).
static int32_t fetch_single(ft_context* c, double* out_rgb);

int32_t ft_fetch_frame(ft_context* c, double* out_rgb) {
    if (!c || !out_rgb) return FT_ERR_INVALID;
    if (!need_device(c)) return FT_ERR_NO_DEVICE;
    bool any = false;
    int32_t rc = FT_OK;
    if (c->last_n_pix > 0) { rc = fetch_single(c, out_rgb); any = true; }
    for (ft_context* p : c->peers) {                                // every device copies its own bands into the caller's frame
        if (rc != FT_OK || p->last_n_pix <= 0) continue;
        rc = fetch_single(p, out_rgb); any = true;
        if (rc != FT_OK) c->err = p->err;
    }
    if (!any) { c->err = "no frame rendered yet"; return FT_ERR_STATE; }
    return rc;
}

static int32_t fetch_single(ft_context* c, double* out_rgb) {
    if (c->last_n_pix <= 0) { c->err = "no frame rendered yet"; return FT_ERR_STATE; }
    FT_HIP(c, hipSetDevice(c->device));
    FT_HIP(c, hipStreamSynchronize(c->stream));                     // frames queued with ft_render_enqueue may still be running (the stream is non-blocking)
    const int64_t n = c->last_n_pix;
    if (c->pixels_whole) {                                         // k_blend wrote the frame in place
        FT_HIP(c, hipMemcpy(out_rgb, c->d_out.p, (size_t)c->last_res_h * c->last_res_v * 24, hipMemcpyDeviceToHost));
    } else {
        std::vector<double> packed((size_t)n * 3);
        FT_HIP(c, hipMemcpy(packed.data(), c->d_out.p, packed.size() * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) std::memcpy(out_rgb + 3 * (size_t)c->pixels[(size_t)i], &packed[3 * (size_t)i], 24);
    }
    return FT_OK;
}

// ------------------------------------------------------------------------------------------ render
static int32_t render_single(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                             int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats, bool defer = false);
static int32_t retire_frame(ft_context* c, ft_context::FrameSlot& f, ft_stats* stats);
static int32_t retire_pending(ft_context* c, ft_stats* stats);

static int32_t render_frame(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                            int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats);

static int32_t with_growing_hit_lists(ft_context* c, const std::function<int32_t()>& run) {
    // Frames still queued by ft_render_enqueue are retired first, so that an overflow of one of THEM is reported as what it is
    // (queued frames are not rendered again) instead of being taken for this call's.
    if (!c->host_only && (c->slots[0].pending || c->slots[1].pending)) {
        if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return FT_ERR_NO_DEVICE; }
        const int32_t prc = retire_pending(c, nullptr);
        c->accum_open = false;
        if (prc != FT_OK) return prc;
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
    return with_growing_hit_lists(c, [&] { return render_frame(c, cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, out_rgb, stats); });
}

static int32_t render_frame(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                            int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats) {
    if (c->peers.empty() || c->host_only) return render_single(c, cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, out_rgb, stats);
    if (!cam || res_h < 2 || res_v < 2 || (tiles && n_tiles < 1)) { c->err = "bad ft_render argument"; return FT_ERR_INVALID; }
    if (!c->committed) { c->err = "scene not committed (ft_scene_commit)"; return FT_ERR_STATE; }
    // Image-tile partition over the devices: 8-row bands of every requested rect, dealt round-robin.
    const auto wall0 = std::chrono::steady_clock::now();
    std::vector<ft_context*> devs{c};
    devs.insert(devs.end(), c->peers.begin(), c->peers.end());
    std::vector<std::vector<ft_rect>> share(devs.size());
    std::vector<ft_rect> whole_frame{ft_rect{0, 0, res_h, res_v}};
    const ft_rect* src = tiles ? tiles : whole_frame.data();
    const int n_src = tiles ? n_tiles : 1;
    size_t band = 0;
    for (int k = 0; k < n_src; ++k)
        for (int y = src[k].y0; y < src[k].y0 + src[k].h; y += 8, ++band)
            share[band % devs.size()].push_back(ft_rect{src[k].x0, y, src[k].w, std::min(8, src[k].y0 + src[k].h - y)});
    std::vector<int32_t> rcs(devs.size(), FT_OK);
    std::vector<ft_stats> sts(devs.size());
    std::vector<std::thread> threads;
    for (size_t d = 0; d < devs.size(); ++d)
        threads.emplace_back([&, d] {
            std::memset(&sts[d], 0, sizeof(ft_stats));
            if (share[d].empty()) { devs[d]->last_n_pix = 0; return; }
            rcs[d] = render_single(devs[d], cam, res_h, res_v, spp, jitter_xy, max_depth, seed, share[d].data(), (int32_t)share[d].size(), out_rgb, &sts[d]);
        });
    for (auto& t : threads) t.join();
    for (size_t d = 0; d < devs.size(); ++d) if (rcs[d] != FT_OK) { if (d) c->err = devs[d]->err; return rcs[d]; }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (auto& s : sts) {
            stats->rays_primary += s.rays_primary; stats->rays_shadow += s.rays_shadow; stats->rays_reflect += s.rays_reflect; stats->rays_traced += s.rays_traced;
            stats->rays_reference_equivalent += s.rays_reference_equivalent; stats->hits_primary += s.hits_primary; stats->csg_overflow += s.csg_overflow;
            stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms); stats->trace_kernel_ms = std::max(stats->trace_kernel_ms, s.trace_kernel_ms);
            stats->algorithmic_bytes += s.algorithmic_bytes; stats->hits_total += s.hits_total; stats->algorithmic_bytes_closest += s.algorithmic_bytes_closest;
            stats->algorithmic_bytes_shade += s.algorithmic_bytes_shade; stats->n_launches += s.n_launches; stats->n_chunks += s.n_chunks;
        }
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    }
    return FT_OK;
}

static int32_t render_single(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                             int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, ft_stats* stats, bool defer) {
    if (!c) return FT_ERR_INVALID;
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
    const bool same_list = !corner && !c->pixels_corner && c->last_n_pix > 0 && c->last_res_h == res_h && c->last_res_v == res_v && c->pixels_whole == whole &&
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
        c->pixel_rects = rects; c->pixels_whole = whole; c->pixels_corner = true; c->last_n_pix = 0;
    } else if (!same_list) {
        std::vector<uint32_t>& px = c->pixels;
        px.clear();
        for (const ft_rect& r : rects) {
            if (r.w % 8 == 0 && r.h % 8 == 0) {
                for (int ty = 0; ty < r.h; ty += 8) for (int tx = 0; tx < r.w; tx += 8)
                    for (int iy = 0; iy < 8; ++iy) for (int ix = 0; ix < 8; ++ix) px.push_back((uint32_t)((r.y0 + ty + iy) * res_h + r.x0 + tx + ix));
            } else {
                for (int y = r.y0; y < r.y0 + r.h; ++y) for (int x = r.x0; x < r.x0 + r.w; ++x) px.push_back((uint32_t)(y * res_h + x));
            }
        }
        c->pixel_rects = rects; c->pixels_whole = whole; c->pixels_corner = false; c->last_n_pix = 0;
    }
    const std::vector<uint32_t>& pixels = c->pixels;
    const int64_t n_pix_total = (int64_t)pixels.size();
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n_pix_total == 0) return FT_OK;

    int32_t rc;
    // k_classify applies to pinhole cameras over whole 64-pixel blocks and scenes in which every top-level item is bounded (with a
    // ground plane in view an exact plane test does find the sky blocks - 20 % of night-house - but the denser first chunk makes
    // k_shade slower than the blocks save).  A classified frame's chunks are windows of its ACTIVE pixel list, usually a fraction
    // of the frame: they are twice as wide (measured at 1080p x 16: bunny 0.58 -> 0.55 ms, hollow-sphere 6.1 -> 5.8, sample 1.64 ->
    // 1.50; the unclassified night-house loses 14 % at that width and keeps the narrow one).
    // k_classify bounds every sample of a pixel by a square of +-extent pixels around its centre.  The reference's offsets lie in the
    // unit disc (Jitter.fs:15-21) but the pattern is the caller's: the square follows the pattern, and a pattern with a non-finite
    // or absurd offset turns classification off instead of bounding nothing.
    double jitter_extent = 1.0;
    bool jitter_bounded = true;
    if (!corner) for (size_t k = 0; k < 2 * (size_t)spp; ++k) { const double v = jitter_xy[k]; if (!(std::fabs(v) <= 64.0)) jitter_bounded = false; else jitter_extent = std::max(jitter_extent, std::fabs(v)); }
    const bool classifiable = c->classify_pixels && jitter_bounded && !corner && !cam->has_focus && c->flat.cull_bundle && c->flat.item_pc.size() > 1 && !c->flat.unbounded;
    const int64_t chunk_budget = classifiable ? 2 * c->chunk_samples : c->chunk_samples;
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
    if ((rc = ensure_frame_buffers(c, cap)) != FT_OK) return rc;
    if ((rc = ensure(c, c->d_out, (size_t)(whole ? (int64_t)res_h * res_v : n_pix_total) * 24)) != FT_OK) return rc;
    if (corner) {
        if ((rc = upload(c, c->d_pixels, corner_ids)) != FT_OK) return rc;
        if ((rc = upload(c, c->d_out_index, pixels)) != FT_OK) return rc;
    } else if (!same_list) { if ((rc = upload(c, c->d_pixels, pixels)) != FT_OK) return rc; }
    std::vector<double> jit;
    if (corner) jit = {-0.5, 0.5};                                 // Image.fs:131
    else jit.assign(jitter_xy, jitter_xy + 2 * (size_t)spp);
    if (jit != c->jitter_on_device) {                              // frames usually reuse the pattern: skip the staged host-to-device copy
        c->jitter_on_device = jit;                                 // (the copy source outlives this call)
        if ((rc = upload(c, c->d_jitter, c->jitter_on_device)) != FT_OK) return rc;
    }
    FT_HIP(c, hipMemsetAsync(c->d_rc.p, 0, sizeof(ftk::RenderCounters) * ((size_t)c->n_stat_slots + 2), c->stream));

    bool classify = classifiable && pix_per_chunk % 64 == 0;
    for (const Job& j : jobs) if (j.n_ids % 64u) classify = false;
    if (classify) {
        if ((rc = ensure(c, c->d_active_ids, (size_t)n_pix_total * 4)) != FT_OK) return rc;
        if ((rc = ensure(c, c->d_active_pos, (size_t)n_pix_total * 4)) != FT_OK) return rc;
        const size_t n_blocks = (size_t)n_pix_total / 64, n_seg = (n_blocks + ftk::kClassifySegmentBlocks - 1) / ftk::kClassifySegmentBlocks;
        if ((rc = ensure(c, c->d_block_flags, n_seg * 4 + n_blocks)) != FT_OK) return rc;             // segment counts, then one byte per block
        FT_HIP(c, hipMemsetAsync(c->d_block_flags.p, 0, n_seg * 4, c->stream));
    }
    const ftk::Camera dcam = make_camera(*cam, res_h, res_v);
    const size_t lds = lds_bytes_for(c->flat);
    int variant = 0;
    for (auto& m : c->flat.materials) if (m.roughness != 0.0 || m.texture >= 0) variant |= 1;   // FANCY
    for (auto& l : c->flat.lights) if (l.kind == ftd::LT_SOFT) variant |= 2;                      // SOFT
    if (!c->flat.meshes.empty()) variant |= 4;                                                     // MESH
    ftk::Launch Lc{c->stream, c->n_cu * ftk::occupancy_blocks_closest(lds, variant), lds, variant};
    ftk::Launch Ls{c->stream, c->n_cu * ftk::occupancy_blocks_shade(lds, variant), lds, variant};
    ftk::Launch Lt{c->stream, c->n_cu * ftk::occupancy_blocks_tail(lds, variant), lds, variant};
    ftk::Launch Lp{c->stream, c->n_cu * ftk::occupancy_blocks_primary(lds, variant), lds, variant};
    ftk::Launch Lg{c->stream, c->n_cu * 8, 0, 0};
    const int last_bounce = c->flat.any_reflective ? max_depth : 0;   // no reflective material ⇒ no reflection rays are ever spawned
    ftk::RayBuf rb[2] = {ray_view(c->d_rays[0], c->ray_capacity), ray_view(c->d_rays[1], c->ray_capacity)};
    ftk::HitBuf hb{c->d_hits.as<double>(), reinterpret_cast<uint32_t*>(c->d_hits.as<double>() + c->ray_capacity),
                   reinterpret_cast<uint32_t*>(c->d_hits.as<double>() + c->ray_capacity) + c->ray_capacity};
    auto* cc = c->d_cc.as<ftk::ChunkCounters>();
    auto* rcount = c->d_rc.as<ftk::RenderCounters>();

    // A blocking call retires whatever is in flight first; a deferred one only the frame whose slot it is about to reuse.
    if (!defer) { int32_t prc = retire_pending(c, nullptr); if (prc != FT_OK) return prc; for (int k = 0; k < 5; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; } c->accum_open = false; }
    ft_context::FrameSlot& F = c->slots[c->slot_turn];
    if (F.pending) { int32_t prc = retire_frame(c, F, nullptr); if (prc != FT_OK) return prc; }
    if (defer && !c->accum_open) { for (int k = 0; k < 5; ++k) { c->k_ms[k] = 0; c->k_launches[k] = 0; } c->accum_open = true; }
    F.events_used = 0; F.spans.clear();
    auto& spans = F.spans;
    using Span = ft_context::FrameSlot::Span;
    // HIP events between stages.  An event between two dependent kernels costs about 6 us of stream time (measured: 0.2 us
    // between k_classify and k_classify_finish, which have none between them), so by default ("timing" = 1) only the two
    // kernels that matter, k_closest and k_shade, are bracketed; 2 brackets every stage, 0 only the frame.
    hipEvent_t ev0 = next_event(F), ev1 = nullptr;
    if (ev0) (void)hipEventRecord(ev0, c->stream);
    hipEvent_t boundary = ev0;
    bool boundary_fresh = true;                                    // `boundary` was recorded right before the next launch
    const int timing = c->timing;
    auto timed = [&](int kind, auto&& fn) {
        const bool bracket = timing >= 2 || (timing == 1 && (kind == 1 || kind == 2 || kind == 4));
        if (bracket && !boundary_fresh) { boundary = next_event(F); if (boundary) (void)hipEventRecord(boundary, c->stream); }
        fn();
        if (!bracket) { boundary_fresh = false; return; }
        hipEvent_t b = next_event(F);
        if (b) (void)hipEventRecord(b, c->stream);
        if (boundary && b) spans.push_back(Span{boundary, b, kind});
        boundary = b; boundary_fresh = true;
    };
    int n_chunks = 0, n_launches = 0;
    // The whole frame is classified once; the chunks then take consecutive windows of the frame's ACTIVE pixel list, so a sparse
    // frame is one chunk of real work and launches that find their window empty return at once.
    ftk::PixCount* const frame_counts = reinterpret_cast<ftk::PixCount*>(rcount + c->n_stat_slots + 1);
    if (classify) {
        const ftk::Primary all{dcam, c->d_pixels.as<uint32_t>(), c->d_jitter.as<double>(), 0u, (uint32_t)n_pix_total, spp, (uint32_t)res_h,
                               (unsigned long long)seed, 1.0 / (double)n_pix_total, 1.0 / (double)res_h, nullptr};
        const size_t n_seg = ((size_t)n_pix_total / 64 + ftk::kClassifySegmentBlocks - 1) / ftk::kClassifySegmentBlocks;
        timed(0, [&] { ftk::launch_classify(Lg, c->dev_scene, all, c->d_block_flags.as<uint8_t>() + n_seg * 4, c->d_block_flags.as<uint32_t>(), c->d_active_ids.as<uint32_t>(),
                                            c->d_active_pos.as<uint32_t>(), frame_counts, c->d_out.as<double>(), whole ? 1 : 0, jitter_extent, rcount); });
        n_launches += 2;
    }
    for (const Job& job : jobs) {
        ++n_chunks;
        const uint32_t n_pix = job.n_ids;
        const uint32_t n_samples = n_pix * (uint32_t)spp;
        timed(0, [&] { (void)hipMemsetAsync(cc, 0, sizeof(ftk::ChunkCounters), c->stream); });
        ftk::Primary gen{dcam, c->d_pixels.as<uint32_t>(), c->d_jitter.as<double>(), job.id_base, n_pix, spp,
                         (uint32_t)(corner ? res_h + 1 : res_h), (unsigned long long)seed,
                         1.0 / (double)n_pix, 1.0 / (double)(corner ? res_h + 1 : res_h), nullptr};
        double* const chunk_out = whole ? c->d_out.as<double>() : c->d_out.as<double>() + 3 * (size_t)job.out_base;
        if (classify) { gen.pixel_ids = c->d_active_ids.as<uint32_t>(); gen.counts = frame_counts; }   // pix_base = job.id_base: the window's start
        for (int b = 0; b <= last_bounce; ++b) {
            if (b == 0 && c->fused_primary) {
                timed(4, [&] { ftk::launch_primary(Lp, c->dev_scene, gen, rb[1], c->d_acc.as<double>(), c->d_touched.as<uint8_t>(), n_samples, max_depth, cc, rcount); });
                ++n_launches;
                continue;
            }
            timed(1, [&] { ftk::launch_closest(Lc, c->dev_scene, gen, rb[b & 1], hb, c->d_hit_list.as<uint32_t>(), c->d_touched.as<uint8_t>(), b, (uint32_t)c->tail_rays, cc, rcount); });
            timed(2, [&] { ftk::launch_shade(Ls, c->dev_scene, gen, rb[b & 1], hb, c->d_hit_list.as<uint32_t>(), rb[(b + 1) & 1], c->d_acc.as<double>(), n_samples, b, max_depth, cc, rcount); });
            n_launches += 2;
        }
        if (last_bounce >= 1 && c->tail_rays > 0) {
            timed(2, [&] { ftk::launch_tail(Lt, c->dev_scene, gen, rb[0], rb[1], c->d_acc.as<double>(), n_samples, max_depth, (uint32_t)c->tail_rays, cc, rcount); });
            ++n_launches;
        }
        const uint32_t* out_index = !whole ? nullptr : (corner ? c->d_out_index.as<uint32_t>() : c->d_pixels.as<uint32_t>()) + job.out_base;
        double* out_ptr = chunk_out;
        if (classify) { out_index = (whole ? c->d_active_ids.as<uint32_t>() : c->d_active_pos.as<uint32_t>()) + job.id_base; out_ptr = c->d_out.as<double>(); }
        if (corner) timed(3, [&] { ftk::launch_blend_corner(Lg, c->d_acc.as<double>(), c->d_touched.as<uint8_t>(), n_samples, job.w, job.h, out_index, out_ptr); });
        else timed(3, [&] { ftk::launch_blend(Lg, c->d_acc.as<double>(), c->d_touched.as<uint8_t>(), n_samples, n_pix, classify ? frame_counts : nullptr, job.id_base, spp, out_index, out_ptr); });
        ++n_launches;
    }
    timed(0, [&] { ftk::launch_reduce_stats(Lg, rcount, c->n_stat_slots); });
    if (boundary_fresh) ev1 = boundary;
    else { ev1 = next_event(F); if (ev1) (void)hipEventRecord(ev1, c->stream); }
    FT_HIP(c, hipGetLastError());
    if (!F.h_rc) FT_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&F.h_rc), sizeof(ftk::RenderCounters), hipHostMallocDefault));
    FT_HIP(c, hipMemcpyAsync(F.h_rc, c->d_rc.p, sizeof(ftk::RenderCounters), hipMemcpyDeviceToHost, c->stream));   // rides the frame's one wait
    F.done = next_event(F);
    if (F.done) FT_HIP(c, hipEventRecord(F.done, c->stream));
    F.ev0 = ev0; F.ev1 = ev1; F.pending = true; F.wall0 = wall0; F.timing = timing;
    F.rays_primary = 0; for (auto& j : jobs) F.rays_primary += (uint64_t)j.n_ids * (uint64_t)spp;
    F.n_pix_total = n_pix_total; F.spp = spp; F.n_launches = n_launches; F.n_chunks = n_chunks; F.classify = classify;
    c->last_n_pix = n_pix_total; c->last_res_h = res_h; c->last_res_v = res_v;
    c->slot_turn ^= 1;
    if (defer) return FT_OK;                                       // ft_render_enqueue: the frame is retired by a later call
    int32_t rrc = retire_frame(c, F, stats);
    if (rrc != FT_OK) return rrc;
    if (out_rgb) { int32_t frc = fetch_single(c, out_rgb); if (frc != FT_OK) return frc; }   // out_rgb == NULL: the frame stays in HBM
    return FT_OK;
}

// Wait for a queued frame, add its stage times to the context's sums and fill its statistics.
static int32_t retire_frame(ft_context* c, ft_context::FrameSlot& F, ft_stats* stats) {
    if (!F.pending) return FT_OK;
    F.pending = false;
    if (F.done) FT_HIP(c, hipEventSynchronize(F.done)); else FT_HIP(c, hipStreamSynchronize(c->stream));
    const ftk::RenderCounters hrc = *F.h_rc;
    const int timing = F.timing; const int32_t spp = F.spp; const int64_t n_pix_total = F.n_pix_total; const bool classify = F.classify;
    hipEvent_t ev0 = F.ev0, ev1 = F.ev1;
    double k1 = 0.0, k2 = 0.0;
    for (auto& s : F.spans) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { c->k_ms[s.kind] += ms; c->k_launches[s.kind]++; if (s.kind == 1) k1 += ms; if (s.kind == 2 || s.kind == 4) k2 += ms; } }
    if (timing < 2) {                                              // index 0 = everything that was not bracketed (memsets, k_classify, k_blend, statistics)
        float total = 0; if (ev0 && ev1) (void)hipEventElapsedTime(&total, ev0, ev1);
        c->k_ms[0] += std::max(0.0, (double)total - k1 - k2);
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        float ms = 0;
        if (ev0 && ev1) (void)hipEventElapsedTime(&ms, ev0, ev1);
        stats->rays_primary = F.rays_primary;
        stats->rays_shadow = hrc.rays_shadow; stats->rays_reflect = hrc.rays_reflect;
        // rays the device really traced: primaries of pixel blocks k_classify finished (Colour.Zero for the whole block, no ray generated)
        // are part of rays_primary and of the reference-equivalent count, not of rays_traced
        stats->rays_primary_culled = (uint64_t)hrc.pixels_culled * (uint64_t)spp;
        stats->rays_traced = stats->rays_primary - std::min<uint64_t>(stats->rays_primary, stats->rays_primary_culled) + stats->rays_shadow + stats->rays_reflect;
        stats->rays_reference_equivalent = (double)stats->rays_primary + hrc.ref_equiv;
        stats->hits_primary = hrc.hits_primary; stats->csg_overflow = hrc.csg_overflow;
        stats->kernel_ms = ms; stats->trace_kernel_ms = k1 + k2;
        {   // bytes the pipeline has to move by construction (ft_device.h); P primary rays, R reflection rays, H hits, H0 primary hits
            // rays and hits that k_tail handled never became records: Ti rays were handed to it (written once, read once), Tr spawned and Th shaded inside it
            const uint64_t Pc = (uint64_t)hrc.pixels_culled * (uint64_t)spp, P = stats->rays_primary - std::min<uint64_t>(stats->rays_primary, Pc), Ti = hrc.tail_in, Tr = hrc.tail_rays, Th = hrc.tail_hits, RR = hrc.rays_reflect, HH = hrc.hits_total;
            const uint64_t R = RR - std::min(RR, Ti + Tr), Rw = RR - std::min(RR, Tr);
            const uint64_t H = HH - std::min(HH, Th), H0 = hrc.hits_primary, HL = H - std::min(H, H0);
            stats->hits_total = hrc.hits_total;
            stats->rays_tail = Ti + Tr;
            stats->algorithmic_bytes_closest = P * (ftk::kPixelIdBytes + ftk::kTouchedBytes) + R * 48 + H * (ftk::kHitRecBytes + ftk::kListBytes);
            stats->algorithmic_bytes_shade = H * (ftk::kHitRecBytes + ftk::kListBytes) + H0 * (2 * ftk::kPixelIdBytes + ftk::kAccBytes) +
                                             HL * (48 + ftk::kRayRecBytes + 2 * ftk::kAccBytes) + Rw * ftk::kRayRecBytes;
            stats->algorithmic_bytes = stats->algorithmic_bytes_closest + stats->algorithmic_bytes_shade +
                                       Ti * ftk::kRayRecBytes + Th * 2 * ftk::kAccBytes +                       // + k_tail
                                       P * ftk::kTouchedBytes + H0 * ftk::kAccBytes + 24ull * (uint64_t)n_pix_total +   // + k_blend (and the pixels k_classify wrote)
                                       (classify ? (uint64_t)n_pix_total * ftk::kPixelIdBytes + (P / (uint64_t)spp) * 2 * ftk::kPixelIdBytes : 0ull);   // + k_classify
        }
        stats->n_launches = F.n_launches; stats->n_chunks = F.n_chunks;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - F.wall0).count();
    }
    if (hrc.csg_overflow) {
        c->err = "CSG hit list overflow on " + std::to_string(hrc.csg_overflow) + " rays: raise csg_mesh_capacity (ft_set_option)";
        return FT_ERR_OVERFLOW;
    }
    return FT_OK;
}

// Retire every queued frame, oldest first; `stats` receives the newest one's.
static int32_t retire_pending(ft_context* c, ft_stats* stats) {
    ft_context::FrameSlot& older = c->slots[c->slot_turn];
    ft_context::FrameSlot& newer = c->slots[c->slot_turn ^ 1];
    int32_t rc = FT_OK;
    if (older.pending) { int32_t r = retire_frame(c, older, newer.pending ? nullptr : stats); if (r != FT_OK) rc = r; }
    if (newer.pending) { int32_t r = retire_frame(c, newer, stats); if (r != FT_OK) rc = r; }
    return rc;
}

/* Pipelined rendering (one device): queue the frame and return; see functracer_hip.h. */
int32_t ft_render_enqueue(ft_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                          int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles) {
    if (!c) return FT_ERR_INVALID;
    if (!c->peers.empty()) { c->err = "ft_render_enqueue works on a one-device context"; return FT_ERR_UNSUPPORTED; }
    return render_single(c, cam, res_h, res_v, spp, jitter_xy, max_depth, seed, tiles, n_tiles, nullptr, nullptr, true);
}
int32_t ft_render_wait(ft_context* c, ft_stats* stats) {
    if (!c) return FT_ERR_INVALID;
    if (c->host_only) return FT_ERR_NO_DEVICE;
    if (stats) std::memset(stats, 0, sizeof *stats);
    FT_HIP(c, hipSetDevice(c->device));
    int32_t rc = retire_pending(c, stats);
    c->accum_open = false;
    return rc;
}

int32_t ft_get_kernel_times(ft_context* c, double ms[5], int32_t launches[5]) {
    if (!c || !ms || !launches) return FT_ERR_INVALID;
    for (int k = 0; k < 5; ++k) { ms[k] = c->k_ms[k]; launches[k] = c->k_launches[k]; }
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
    FT_HIP(c, hipMemsetAsync(c->d_rc.p, 0, sizeof(ftk::RenderCounters), c->stream));
    double* dt = c->d_dbg_out.as<double>();
    double* dp = dt + N; double* dn = dp + 3 * N; double* dc = dn + 3 * N;
    int32_t* dh = reinterpret_cast<int32_t*>(dc + 3 * N);
    const size_t lds = lds_bytes_for(c->flat);
    ftk::Launch L{c->stream, c->n_cu * 4, lds, 0};
    ftk::launch_debug_closest(L, c->dev_scene, din, din + 3 * N, (uint32_t)n, dh, dt, dp, dn, dc, c->d_rc.as<ftk::RenderCounters>());
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(t, dt, N * 8, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(p, dp, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(nrm, dn, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(colour, dc, N * 24, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipMemcpyAsync(hit, dh, N * 4, hipMemcpyDeviceToHost, c->stream));
    ftk::RenderCounters hrc{};
    FT_HIP(c, hipMemcpyAsync(&hrc, c->d_rc.p, sizeof hrc, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipStreamSynchronize(c->stream));
    if (hrc.csg_overflow) { c->err = "CSG hit list overflow"; return FT_ERR_OVERFLOW; }
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
    FT_HIP(c, hipMemsetAsync(c->d_rc.p, 0, sizeof(ftk::RenderCounters), c->stream));
    const size_t lds = lds_bytes_for(c->flat);
    ftk::Launch L{c->stream, c->n_cu * 4, lds, 0};
    ftk::launch_debug_blocked(L, c->dev_scene, din, din + 3 * N, din + 6 * N, (uint32_t)n, c->d_dbg_out.as<int32_t>(), c->d_rc.as<ftk::RenderCounters>());
    FT_HIP(c, hipGetLastError());
    FT_HIP(c, hipMemcpyAsync(blocked, c->d_dbg_out.p, N * 4, hipMemcpyDeviceToHost, c->stream));
    ftk::RenderCounters hrc{};
    FT_HIP(c, hipMemcpyAsync(&hrc, c->d_rc.p, sizeof hrc, hipMemcpyDeviceToHost, c->stream));
    FT_HIP(c, hipStreamSynchronize(c->stream));
    if (hrc.csg_overflow) { c->err = "CSG hit list overflow"; return FT_ERR_OVERFLOW; }
    return FT_OK;
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
