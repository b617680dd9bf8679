// ft_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the FuncTracer render loop.
//
// Pipeline (DESIGN.md §"Kernels"); per frame:
//   k_classify  which 64-pixel blocks can see anything, and the frame's active pixel list (in block order), in one kernel
// per chunk of samples of the active list:
//   k_primary   bounce 0, fused: primary rays generated in registers (Image.fs:83-89, 100-110), closest hit (Scene.fs:112-118 over
//               the flattened Scene.intersect, Scene.fs:67-104), shadow rays + Phong + reflection spawn (Shading.fs:24-139)
//   k_bounce    one launch per level of the reflection tree (bounce k >= 1), fused the same way: the rays live in a ping-pong
//               wavefront buffer in HBM, the ones that terminate are dropped by wave-ballot / prefix-sum compaction when the next
//               level is spawned, so every lane of a level is live
//   k_resolve   per-pixel mean (Image.fs:112-116), Colour.Zero for the blocks k_classify finished, FP64 and / or RGBA8 (Image.fs:36)
// All tracing kernels are persistent grids whose waves pull 64-ray batches from 64 interleaved cursors.
//
// Execution model notes (wave64, CDNA4):
//   * one lane = one ray; the scene program, leaf records, matrices, materials, lights and
//     brute-force triangle lists are wave-uniform and are read with scalar loads (SGPR operands);
//   * FP64 throughout (the reference is F# float); no MFMA — this is branchy scalar geometry;
//   * per-lane CSG hit lists and BSP node stacks live in LDS, laid out [entry][lane] so that a
//     wave access is always bank-conflict free whatever entry each lane touches;
//   * waves are independent: no __syncthreads in any tracing kernel;
//   * coherent wavefronts (one 8x8 pixel block) walk mesh trees as a packet with a wave-uniform stack, and bound their
//     rays by a cone that is tested against the bounding spheres of all top-level items at once (one lane per item);
//   * launch arguments are read from the kernarg segment where they are used (scalar loads), not held in SGPRs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ft_device.h"

using namespace ftd;

namespace ftk {
namespace {

#define FT_DEV __device__ __forceinline__

constexpr double kEps = 0.0000001;
constexpr int kDone = INT32_MIN;

// Scene data is immutable for the lifetime of a launch.  Reading it through constant-address-space
// pointers lets the compiler use scalar loads (s_load_*, SGPR operands) for every wave-uniform
// access even though the kernels also store to global memory; with generic pointers it falls back to
// per-lane global_load of the same address.
#define FT_CONST __attribute__((address_space(4)))
typedef const FT_CONST double* cdp;
typedef const FT_CONST uint32_t* cup;
typedef const FT_CONST int32_t* cip;
template <class T> FT_DEV const FT_CONST T* to_const_as(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const FT_CONST T*)p;
#pragma clang diagnostic pop
}
// Launch arguments are read where they are used, through the kernarg segment (constant address space, scalar loads),
// instead of living in SGPRs for the whole kernel: with ~60 SGPRs of camera / buffer pointers held across the scene
// interpreter every kernel ran out of scalar registers and the compiler parked them in VGPR lanes (v_writelane /
// v_readlane - VALU instructions - by the hundred per batch).  `fresh()` returns the same pointer behind an opaque
// zero so that loads through it cannot be hoisted out of the batch loop and stay short-lived.
FT_DEV uint32_t opaque_zero() { uint32_t z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }
template <class T> FT_DEV const FT_CONST T* kernel_args() {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const FT_CONST T*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma clang diagnostic pop
}
template <class T> FT_DEV const FT_CONST T* fresh(const FT_CONST T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const FT_CONST T*)((const FT_CONST char*)p + opaque_zero());
#pragma clang diagnostic pop
}
typedef const FT_CONST Primary* PrimaryArg;

struct Scene {
    cdp leaves;       // 16 doubles per leaf (ftd::Leaf)
    cdp m2w;          // 12 per leaf
    cdp materials;    // 8 doubles per ftd::Material
    cdp lights;       // 12 doubles per ftd::Light
    cdp textures;     // 48 doubles per ftd::Texture
    cup program;
    cip meshes;       // 4 words per ftd::Mesh
    cdp nodes;        // 8 doubles per ftd::BspNode
    cup bsp_leaves;   // 2 words per ftd::BspLeaf
    cdp tris;         // 9 doubles per triangle
    cdp culls;        // 24 doubles per ftd::CullRecord
    cup tri_orig;     // 1 per triangle
    cdp wide;         // 28 doubles per 4-wide BVH node
    cip mesh_wide;    // 1 per mesh
    const uint8_t* tex_pixels;   // per-lane byte gathers: ordinary global loads
    const float* cull_items;     // lane k reads record k: ordinary global loads
    const float* coarse_boxes;   // lane k reads box k
    cdp cull_rows;
    cup item_pc;
    int32_t n_leaves, n_lights, csg_cap, stack_cap, n_items, n_cull_rows, csg_rows, lane_fold, n_simd;
};
static_assert(sizeof(Texture) == 384 && sizeof(CullRecord) == 192 && sizeof(Leaf) == 128 && sizeof(Material) == 64 && sizeof(Light) == 96 && sizeof(Mesh) == 16 && sizeof(BspNode) == 64 && sizeof(BspLeaf) == 8, "flat layout");
template <class DS> FT_DEV Scene scene_view(const DS& g) {
    Scene s;
    s.leaves = to_const_as(g.leaves); s.m2w = to_const_as(g.m2w);
    s.materials = to_const_as(reinterpret_cast<const double*>(g.materials)); s.lights = to_const_as(reinterpret_cast<const double*>(g.lights));
    s.textures = to_const_as(reinterpret_cast<const double*>(g.textures));
    s.program = to_const_as(g.program); s.meshes = to_const_as(reinterpret_cast<const int32_t*>(g.meshes));
    s.nodes = to_const_as(reinterpret_cast<const double*>(g.nodes)); s.bsp_leaves = to_const_as(reinterpret_cast<const uint32_t*>(g.bsp_leaves));
    s.tris = to_const_as(g.tris); s.culls = to_const_as(g.culls); s.tri_orig = to_const_as(g.tri_orig); s.wide = to_const_as(g.wide); s.mesh_wide = to_const_as(g.mesh_wide);
    s.tex_pixels = g.tex_pixels; s.cull_items = g.cull_items; s.coarse_boxes = g.coarse_boxes; s.cull_rows = to_const_as(g.cull_rows); s.item_pc = to_const_as(g.item_pc); s.n_items = g.n_items; s.n_cull_rows = g.n_cull_rows;
    s.n_leaves = g.n_leaves; s.n_lights = g.n_lights; s.csg_cap = g.csg_cap; s.stack_cap = g.stack_cap; s.csg_rows = g.csg_rows; s.lane_fold = g.lane_fold; s.n_simd = g.n_simd;
    return s;
}
struct MaterialV { double colour[3]; double roughness, reflectance, shineyness; uint32_t apply_lighting; int32_t texture; uint32_t hue_rot; };
FT_DEV MaterialV material_at(const Scene& S, uint32_t i) {
    cdp m = S.materials + 8ull * i;
    MaterialV v;
    v.colour[0] = m[0]; v.colour[1] = m[1]; v.colour[2] = m[2]; v.roughness = m[3]; v.reflectance = m[4]; v.shineyness = m[5];
    v.apply_lighting = reinterpret_cast<cup>(m + 6)[0]; v.texture = reinterpret_cast<cip>(m + 6)[1]; v.hue_rot = reinterpret_cast<cup>(m + 7)[0];
    return v;
}

struct Ray { double ox, oy, oz, dx, dy, dz; };
struct V3 { double x, y, z; };

FT_DEV double dot3(double ax, double ay, double az, double bx, double by, double bz) { return ax * bx + ay * by + az * bz; }
// Vector.normalise (CommonTypes.fs:63-67): v unchanged when |v| < eps, else (1 / |v|) * v.  1 / |v| comes from the hardware
// reciprocal square root refined by two Newton steps (full double precision, within an ulp or two of 1 / sqrt) in place of a
// square root followed by a division: 15 instructions instead of 35, and shading normalises seven vectors per light.
FT_DEV V3 normalise(V3 v) {
    const double l2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!(l2 >= kEps * kEps) || !(l2 < 1e300)) {                    // tiny, NaN or overflowing: the reference's own sequence
        const double l = sqrt(l2);
        if (l < kEps) return v;
        const double s = 1.0 / l;
        return {s * v.x, s * v.y, s * v.z};
    }
    double y = __builtin_amdgcn_rsq(l2);
    const double h = 0.5 * l2;
    y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-(h * y), y, 0.5), y);
    return {y * v.x, y * v.y, y * v.z};
}
FT_DEV double fs_max(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? b : a); }  // F# max on float = Math.Max
FT_DEV double fs_min(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a < b ? a : b); }

FT_DEV uint32_t lane_id() { return __lane_id(); }
// A wave-uniform double the compiler computed with vector instructions (a division, a conversion), moved to a scalar register
// pair: held across the batch loop in VGPRs such values - reciprocals of the launch's counts, the number of lights as a double -
// were what the register allocator spilled to scratch first.
FT_DEV double uniform_f64(double v) {
#ifdef FT_AB_NO_UNIFORM
    return v;
#endif
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
FT_DEV uint32_t lanes_below(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }

// ---------------------------------------------------------------------------------------------
// Per-lane hit list in LDS (only used under CSG nodes).  Entry e, word w of this lane lives at
// base[(e*4 + w) * kBlock]: consecutive lanes hit consecutive banks for every (e, w).
struct HitList {
    uint32_t* base;
    int len;
    int cap;
    int rows, stride;                                               // lane folding (below): LDS rows per column, lanes between a lane's columns
    unsigned long long marks_lo, marks_hi;
    bool overflow;

    // Lane folding: a scene whose lists need more LDS than a workgroup has (meshes under CSG) runs with 64 / fold live lanes per
    // wave, and live lane l also owns the columns of the idle lanes l + 64/fold, l + 2*64/fold, ...: entry e lives in row
    // e mod rows of column e / rows.  fold = 1 (the rule): rows = cap and the column never changes.
    FT_DEV void init(uint32_t* lds, int capacity, int rows_per_column, int fold) {
        base = lds + threadIdx.x; len = 0; cap = capacity; rows = rows_per_column; stride = 64 / fold; marks_lo = marks_hi = 0; overflow = false;
    }
    FT_DEV uint32_t* at(int e) const { int col = 0; while (e >= rows) { e -= rows; col += stride; } return base + (e * 4) * kBlock + col; }
    FT_DEV void store(int e, double t, uint32_t id0, uint32_t id1) {
        unsigned long long b = __double_as_longlong(t);
        uint32_t* p = at(e);
        p[0] = (uint32_t)b; p[kBlock] = (uint32_t)(b >> 32); p[2 * kBlock] = id0; p[3 * kBlock] = id1;
    }
    FT_DEV double t_of(int e) const {
        const uint32_t* p = at(e);
        unsigned long long b = (unsigned long long)p[0] | ((unsigned long long)p[kBlock] << 32);
        return __longlong_as_double(b);
    }
    FT_DEV uint32_t id0_of(int e) const { return at(e)[2 * kBlock]; }
    FT_DEV uint32_t id1_of(int e) const { return at(e)[3 * kBlock]; }
    FT_DEV void set_id0(int e, uint32_t v) { at(e)[2 * kBlock] = v; }
    FT_DEV void push(double t, uint32_t id0, uint32_t id1) {
        if (len < cap) { store(len, t, id0, id1); ++len; } else overflow = true;   // never silently dropped: reported via RenderCounters
    }
    FT_DEV void mark() { marks_hi = (marks_hi << 8) | (marks_lo >> 56); marks_lo = (marks_lo << 8) | (unsigned long long)(uint32_t)len; }
    FT_DEV int pop_mark() { int m = (int)(marks_lo & 0xFF); marks_lo = (marks_lo >> 8) | (marks_hi << 56); marks_hi >>= 8; return m; }
};

// Csg.constructedSolid (Csg.fs:74-94) on the two topmost segments of the lane's list.
// Rule tables (Csg.fs:19-55) as bit masks over IntersectionType
//   0 OutsideIntoA 1 OutsideIntoB 2 BIntoAB 3 AIntoAB 4 ABleaveA 5 ABleaveB 6 AIntoOutside 7 BIntoOutside
FT_DEV void csg_merge(HitList& L, uint32_t op) {
    const int seg_b = L.pop_mark();
    const int seg_a = L.pop_mark();
    const int end = L.len;
    if (!__any(end > seg_a)) return;                               // no lane has anything to merge
    uint32_t take, flip;
    switch (op) {                                                   // wave-uniform
        case 0: take = 0xC3u; flip = 0x00u; break;                 // union
        case 1: take = 0x3Cu; flip = 0x00u; break;                 // intersect
        case 2: take = 0x41u; flip = 0x28u; break;                 // subtract
        default: take = 0xC3u; flip = 0x3Cu; break;                 // exclude
    }
    for (int e = seg_b; e < end; ++e) L.set_id0(e, L.id0_of(e) | ID_SIDE_B);       // HitB tag (Csg.fs:77)
    for (int i = seg_a + 1; i < end; ++i) {                         // stable insertion sort by t (Seq.sortBy, Csg.fs:78-79)
        const double kt = L.t_of(i); const uint32_t k0 = L.id0_of(i), k1 = L.id1_of(i);
        int j = i;
        while (j > seg_a && kt < L.t_of(j - 1)) { L.store(j, L.t_of(j - 1), L.id0_of(j - 1), L.id1_of(j - 1)); --j; }
        if (j != i) L.store(j, kt, k0, k1);
    }
    bool in_a = false, in_b = false;
    int w = seg_a;
    for (int r = seg_a; r < end; ++r) {                             // iterate (Csg.fs:81-93)
        const double t = L.t_of(r); uint32_t id0 = L.id0_of(r); const uint32_t id1 = L.id1_of(r);
        const bool side_b = (id0 & ID_SIDE_B) != 0;
        // getIntersectionType (Csg.fs:59-72): index = sideB*4 + inA*2 + inB
        const uint32_t type = (0x53714620u >> (4 * ((side_b ? 4 : 0) + (in_a ? 2 : 0) + (in_b ? 1 : 0)))) & 0xF;
        if (side_b) in_b = !in_b; else in_a = !in_a;
        id0 &= ~ID_SIDE_B;
        if ((flip >> type) & 1u) id0 ^= ID_FLIP;                    // Flip: n <- -1.0 * n
        if (((take | flip) >> type) & 1u) { L.store(w, t, id0, id1); ++w; }
    }
    L.len = w;
}

// ---------------------------------------------------------------------------------------------
// Running result of a scene query.  ANY = lightIsBocked (Scene.fs:119-121), else closest (Scene.fs:112-116):
// hits arrive in the reference's sequence order, so "strictly smaller t wins" reproduces
// stable-sort-then-head, t = 0 (and -0) is kept, negatives are skipped.
template <bool ANY>
struct Query {
    double best_t; uint32_t id0, id1;
    double max_dist; bool blocked;
    bool active;
    FT_DEV void hit(double t, uint32_t i0, uint32_t i1, bool lit) {
        if (!active) return;
        if (ANY) { if (t >= 0.0 && t < max_dist && lit) blocked = true; }
        else { if (t >= 0.0 && t < best_t) { best_t = t; id0 = i0; id1 = i1; } }
    }
};

// ---------------------------------------------------------------------------------------------
// Leaf intersection.  `emit(t, sub, tri)` is called once per hit in the reference's order.

// Plane.intersect with the canonical plane (Plane.fs:9-20, 28-33) given num = (p0-o).n, den = d.n.
FT_DEV bool plane_t(double num, double den, double& t) {
    if (fabs(den) < kEps) { t = 0.0; return num < kEps; }          // parallel ray: hit at the origin iff on/above
    t = num / den;
    return true;
}

// Möller–Trumbore (Triangle.fs:43-66) against v0,e1,e2.  The u / v range tests are decided without the
// division whenever the outcome cannot depend on its rounding; the surviving lanes run the
// reference's exact sequence.
FT_DEV bool tri_hit(cdp T, const Ray& r, double& t_out) {
    const double v0x = T[0], v0y = T[1], v0z = T[2], e1x = T[3], e1y = T[4], e1z = T[5], e2x = T[6], e2y = T[7], e2z = T[8];
    const double hx = r.dy * e2z - r.dz * e2y, hy = e2x * r.dz - e2z * r.dx, hz = r.dx * e2y - r.dy * e2x;   // ray.d .** edge2
    const double a = e1x * hx + e1y * hy + e1z * hz;
    if (a > -kEps && a < kEps) return false;
    const double sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    const double sh = sx * hx + sy * hy + sz * hz;
    const double abs_a = fabs(a);
    // u = (1/a)*sh: u < 0 iff sh and a have strictly opposite signs; |sh| > |a|(1+4e-15) implies fl(u) > 1.
    if ((sh < 0.0 && a > 0.0) || (sh > 0.0 && a < 0.0) || fabs(sh) > abs_a * 1.000000000000004) return false;
    const double qx = sy * e1z - sz * e1y, qy = e1x * sz - e1z * sx, qz = sx * e1y - sy * e1x;               // s .** edge1
    const double dq = r.dx * qx + r.dy * qy + r.dz * qz;
    if ((dq < 0.0 && a > 0.0) || (dq > 0.0 && a < 0.0) || fabs(dq) > abs_a * 1.000000000000004) return false;
    const double f = 1.0 / a;
    const double u = f * sh;
    if (u < 0.0 || u > 1.0) return false;
    const double v = f * dq;
    if (v < 0.0 || u + v > 1.0) return false;
    const double t = f * (e2x * qx + e2y * qy + e2z * qz);
    if (t > kEps) { t_out = t; return true; }
    return false;
}

// tri_hit for the packet walk: the same products and the same comparisons, with wave-uniform exits in place of eight per-lane
// ones.  A lane's early return saves nothing while another lane of the wave goes on, and each one is a saved and restored exec mask:
// the rejections before the division are gathered into one flag (nothing passes them: the 64 lanes skip the division), the ones
// after it into a second.  Lanes that have failed carry on with values nobody reads (1 / 0 included: no traps on the device).
FT_DEV bool tri_hit_wave(cdp T, const Ray& r, bool live, double& t_out) {
    const double v0x = T[0], v0y = T[1], v0z = T[2], e1x = T[3], e1y = T[4], e1z = T[5], e2x = T[6], e2y = T[7], e2z = T[8];
    const double hx = r.dy * e2z - r.dz * e2y, hy = e2x * r.dz - e2z * r.dx, hz = r.dx * e2y - r.dy * e2x;
    const double a = e1x * hx + e1y * hy + e1z * hz;
    const double sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    const double sh = sx * hx + sy * hy + sz * hz;
    const double abs_a = fabs(a), lim = abs_a * 1.000000000000004;
    const bool pos = a > 0.0, neg = a < 0.0;
    bool ok = live & !((a > -kEps) & (a < kEps));
    ok = ok & !(((sh < 0.0) & pos) | ((sh > 0.0) & neg) | (fabs(sh) > lim));
    if (!__any(ok)) return false;                                   // the usual end in a dense mesh: the triangle lies beside the whole bundle
    const double qx = sy * e1z - sz * e1y, qy = e1x * sz - e1z * sx, qz = sx * e1y - sy * e1x;
    const double dq = r.dx * qx + r.dy * qy + r.dz * qz;
    ok = ok & !(((dq < 0.0) & pos) | ((dq > 0.0) & neg) | (fabs(dq) > lim));
    if (!__any(ok)) return false;
    const double f = 1.0 / a;
    const double u = f * sh;
    const double v = f * dq;
    const double t = f * (e2x * qx + e2y * qy + e2z * qz);
    t_out = t;
    return ok & !((u < 0.0) | (u > 1.0)) & !((v < 0.0) | (u + v > 1.0)) & (t > kEps);
}

// BoundingBox.intersects (BoundingBox.fs:32-58), inverse direction precomputed per ray.
FT_DEV bool aabb_hit(cdp nd, const Ray& r, double ivx, double ivy, double ivz, double* entry = nullptr, double* exit = nullptr) {
    struct { double bmin[3], bmax[3]; } n = {{nd[0], nd[1], nd[2]}, {nd[3], nd[4], nd[5]}};
    const bool nx = ivx < 0.0, ny = ivy < 0.0, nz = ivz < 0.0;
    double tmin = ((nx ? n.bmax[0] : n.bmin[0]) - r.ox) * ivx;
    double tmax = ((nx ? n.bmin[0] : n.bmax[0]) - r.ox) * ivx;
    const double tymin = ((ny ? n.bmax[1] : n.bmin[1]) - r.oy) * ivy;
    const double tymax = ((ny ? n.bmin[1] : n.bmax[1]) - r.oy) * ivy;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    tmin = fs_max(tymin, tmin);
    tmax = fs_min(tymax, tmax);
    const double tzmin = ((nz ? n.bmax[2] : n.bmin[2]) - r.oz) * ivz;
    const double tzmax = ((nz ? n.bmin[2] : n.bmax[2]) - r.oz) * ivz;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    tmin = fs_max(tzmin, tmin);
    tmax = fs_min(tzmax, tmax);
    if (entry) *entry = tmin;
    if (exit) *exit = tmax;
    return (tmin < __builtin_inf()) && (tmax > -__builtin_inf());
}

template <class Emit>
FT_DEV void mesh_hits(const Scene& S, uint32_t mesh_idx, const Ray& r, bool active, int32_t* stack, Emit&& emit) {
    const int32_t root = S.meshes[4 * mesh_idx];
    if (root < 0) {                                                // top-level Leaf: brute force, no AABB (BspMesh.fs:95-97)
        const uint32_t first = S.bsp_leaves[2 * (~root)], count = S.bsp_leaves[2 * (~root) + 1];
        cdp T = S.tris + 9ull * first;
        for (uint32_t k = 0; k < count; ++k, T += 9) {             // wave-uniform loop: triangle data comes through scalar loads
            double t;
            if (tri_hit(T, r, t)) emit(t, 0u, first + k);
        }
        return;
    }
    // BspMesh.intersect (BspMesh.fs:67-76): at a branch whose box the ray's line meets, all hits of the
    // RIGHT child come before those of the LEFT child.  Per-lane DFS with the pending left children on an
    // LDS stack.
    const double ivx = 1.0 / r.dx, ivy = 1.0 / r.dy, ivz = 1.0 / r.dz;
    int sp = 0;
    int cur = active ? root : kDone;
    while (__any(cur != kDone)) {
        while (cur >= 0) {                                         // descend through branches
            cdp nd = S.nodes + 8ull * (uint32_t)cur;
            if (aabb_hit(nd, r, ivx, ivy, ivz)) { cip ch = reinterpret_cast<cip>(nd + 6); stack[sp * kBlock] = ch[0]; ++sp; cur = ch[1]; }
            else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
            else cur = kDone;
        }
        if (cur != kDone) {                                        // a leaf: every triangle, in list order
            const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
            for (uint32_t k = 0; k < count; ++k) {
                double t;
                if (tri_hit(S.tris + 9ull * (first + k), r, t)) emit(t, 0u, first + k);
            }
            if (sp > 0) { --sp; cur = stack[sp * kBlock]; } else cur = kDone;
        }
    }
}

FT_DEV void to_model(cdp M, bool xform, const Ray& r, Ray& m) {   // Transform.fs:85
    if (xform) {
        m.ox = M[0] * r.ox + M[1] * r.oy + M[2] * r.oz + M[3];
        m.oy = M[4] * r.ox + M[5] * r.oy + M[6] * r.oz + M[7];
        m.oz = M[8] * r.ox + M[9] * r.oy + M[10] * r.oz + M[11];
        m.dx = M[0] * r.dx + M[1] * r.dy + M[2] * r.dz;
        m.dy = M[4] * r.dx + M[5] * r.dy + M[6] * r.dz;
        m.dz = M[8] * r.dx + M[9] * r.dy + M[10] * r.dz;
    } else m = r;
}

struct LeafHead { uint32_t kind, flags, material, mesh; };
FT_DEV LeafHead leaf_head(const Scene& S, uint32_t leaf) {
    cup h = reinterpret_cast<cup>(S.leaves + 16ull * leaf + 12);
    return {h[0], h[1], h[2], h[3]};
}

// Math.quadratic (Math.fs:4-10): far root first.
FT_DEV bool quadratic(double a, double b, double c, double& r0, double& r1) {
    const double disc = b * b - 4.0 * a * c;
    if (disc < 0.0) return false;
    const double sq = sqrt(disc), twoa = 2.0 * a;
    r0 = (-b + sq) / twoa; r1 = (-b - sq) / twoa;
    return true;
}

template <bool MESH, class Emit>
FT_DEV void leaf_hits(const Scene& S, uint32_t leaf, const LeafHead& H, const Ray& rw, bool active, int32_t* stack, Emit&& emit) {
    Ray r;
    to_model(S.leaves + 16ull * leaf, (H.flags & LF_XFORM) != 0, rw, r);
    switch (H.kind) {                                              // wave-uniform
        case LK_SPHERE: {                                          // Sphere.fs:11-21
            const double a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
            const double b = 2.0 * dot3(r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
            const double c = dot3(r.ox, r.oy, r.oz, r.ox, r.oy, r.oz) - 1.0;
            double t0, t1;
            if (quadratic(a, b, c, t0, t1)) { emit(t0, 0u, 0u); emit(t1, 1u, 0u); }
            break;
        }
        case LK_PLANE: {                                           // Plane.fs:28-33
            double t;
            if (plane_t(-r.oy, r.dy, t)) emit(t, 0u, 0u);
            break;
        }
        case LK_SQUARE: {                                          // Cube.fs:9-15
            double t;
            if (plane_t(-r.oy, r.dy, t)) {
                const double px = r.ox + t * r.dx, pz = r.oz + t * r.dz;
                if (px >= 0.0 && px <= 1.0 && pz >= 0.0 && pz <= 1.0) emit(t, 0u, 0u);
            }
            break;
        }
        case LK_CIRCLE: {                                          // Cylinder.fs:22
            double t;
            if (plane_t(-r.oy, r.dy, t)) {
                const double px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;
                if (sqrt(px * px + py * py + pz * pz) < 1.0) emit(t, 0u, 0u);
            }
            break;
        }
        case LK_CUBE: {                                            // Cube.fs:17-25, in the [0,1]^3 frame behind translate(-.5)
            const double qx = r.ox + 0.5, qy = r.oy + 0.5, qz = r.oz + 0.5;
            double t;
            // bottom (y=0, flipped) and top (y=1): clip on x,z
            if (plane_t(-qy, r.dy, t)) { const double u = qx + t * r.dx, v = qz + t * r.dz; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 0u, 0u); }
            if (plane_t(-(qy - 1.0), r.dy, t)) { const double u = qx + t * r.dx, v = qz + t * r.dz; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 1u, 0u); }
            // left (x=0) and right (x=1, flipped): local frame (y,-x,z), clip on y,z
            if (plane_t(qx, -r.dx, t)) { const double u = qy + t * r.dy, v = qz + t * r.dz; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 2u, 0u); }
            if (plane_t(qx - 1.0, -r.dx, t)) { const double u = qy + t * r.dy, v = qz + t * r.dz; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 3u, 0u); }
            // front (z=0) and back (z=1, flipped): local frame (x,-z,y), clip on x,y
            if (plane_t(qz, -r.dz, t)) { const double u = qx + t * r.dx, v = qy + t * r.dy; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 4u, 0u); }
            if (plane_t(qz - 1.0, -r.dz, t)) { const double u = qx + t * r.dx, v = qy + t * r.dy; if (u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) emit(t, 5u, 0u); }
            break;
        }
        case LK_CONE: {                                            // Cone.fs:7-27
            const double oy = r.oy - 1.0;
            const double a = r.dx * r.dx + r.dz * r.dz - r.dy * r.dy;
            const double b = 2.0 * (r.ox * r.dx + r.oz * r.dz - oy * r.dy);
            const double c = r.ox * r.ox + r.oz * r.oz - oy * oy;
            double t0, t1;
            if (quadratic(a, b, c, t0, t1)) {
                double py = (oy + t0 * r.dy) + 1.0; if (py >= 0.0 && py <= 1.0) emit(t0, 0u, 0u);
                py = (oy + t1 * r.dy) + 1.0;        if (py >= 0.0 && py <= 1.0) emit(t1, 1u, 0u);
            }
            break;
        }
        case LK_CYLINDER: {                                        // Cylinder.fs:8-20
            const double a = r.dx * r.dx + r.dz * r.dz;
            const double b = 2.0 * (r.ox * r.dx + r.oz * r.dz);
            const double c = r.ox * r.ox + r.oz * r.oz - 1.0;
            double t0, t1;
            if (quadratic(a, b, c, t0, t1)) {
                double py = r.oy + t0 * r.dy; if (py >= 0.0 && py <= 1.0) emit(t0, 0u, 0u);
                py = r.oy + t1 * r.dy;        if (py >= 0.0 && py <= 1.0) emit(t1, 1u, 0u);
            }
            break;
        }
        case LK_SOLIDCYL: {                                        // Cylinder.fs:25-29: [top; bottom; sides]
            double t;
            const double oyt = r.oy - 1.0;                         // top: translate (0,1,0) circle
            if (plane_t(-oyt, r.dy, t)) { const double px = r.ox + t * r.dx, py = oyt + t * r.dy, pz = r.oz + t * r.dz; if (sqrt(px * px + py * py + pz * pz) < 1.0) emit(t, 0u, 0u); }
            // bottom: rotate Z 180 circle — local frame (-x,-y,z)
            if (plane_t(r.oy, -r.dy, t)) { const double px = -r.ox + t * -r.dx, py = -r.oy + t * -r.dy, pz = r.oz + t * r.dz; if (sqrt(px * px + py * py + pz * pz) < 1.0) emit(t, 1u, 0u); }
            const double a = r.dx * r.dx + r.dz * r.dz;
            const double b = 2.0 * (r.ox * r.dx + r.oz * r.dz);
            const double c = r.ox * r.ox + r.oz * r.oz - 1.0;
            double t0, t1;
            if (quadratic(a, b, c, t0, t1)) {
                double py = r.oy + t0 * r.dy; if (py >= 0.0 && py <= 1.0) emit(t0, 2u, 0u);
                py = r.oy + t1 * r.dy;        if (py >= 0.0 && py <= 1.0) emit(t1, 3u, 0u);
            }
            break;
        }
        default:                                                   // LK_MESH
            if (MESH) mesh_hits(S, H.mesh, r, active, stack, emit);
            break;
    }
}

// Closest / any-hit query of a top-level-Leaf mesh through the exact BVH (ft_flat.h): same per-triangle
// arithmetic as the linear scan, boxes pruned only when they cannot hold a usable hit, t ties resolved
// by list index.  `r` is the ray in the mesh's model space.
template <bool ANY>
FT_DEV void mesh_bvh_query(const Scene& S, int32_t bvh_root, const Ray& r, Query<ANY>& q, uint32_t leaf, bool lit, int32_t* stack) {
    const double ivx = 1.0 / r.dx, ivy = 1.0 / r.dy, ivz = 1.0 / r.dz;
    double bound = ANY ? q.max_dist : q.best_t;                    // a hit at t >= bound cannot change the query's result
    uint32_t best_tri = 0xFFFFFFFFu;
    bool found = false;
    int sp = 0;
    int cur = (q.active && !(ANY && q.blocked)) ? bvh_root : kDone;
    while (__any(cur != kDone)) {
        while (cur >= 0) {
            cdp nd = S.nodes + 8ull * (uint32_t)cur;
            double t0 = (nd[0] - r.ox) * ivx, t1 = (nd[3] - r.ox) * ivx;
            double tmin = fmin(t0, t1), tmax = fmax(t0, t1);       // fmin/fmax drop NaNs (0 * inf on a slab plane): conservative
            t0 = (nd[1] - r.oy) * ivy; t1 = (nd[4] - r.oy) * ivy;
            tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1));
            t0 = (nd[2] - r.oz) * ivz; t1 = (nd[5] - r.oz) * ivz;
            tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1));
            if (tmax >= fmax(tmin, 0.0) && tmin <= bound) {        // triangle hits need t > 1e-7 (Triangle.fs:62), so boxes behind the origin are out
                cip ch = reinterpret_cast<cip>(nd + 6);
                const uint32_t axis = reinterpret_cast<cup>(nd + 7)[0];
                const double da = axis == 0 ? r.dx : axis == 1 ? r.dy : r.dz;
                const int near = da >= 0.0 ? ch[0] : ch[1], far = da >= 0.0 ? ch[1] : ch[0];
                stack[sp * kBlock] = far; ++sp; cur = near;        // near child first: the bound shrinks sooner
            } else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
            else cur = kDone;
        }
        if (cur != kDone) {
            const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
            for (uint32_t k = 0; k < count; ++k) {
                double t;
                if (!tri_hit(S.tris + 9ull * (first + k), r, t)) continue;
                if (ANY) { if (t < bound) { q.blocked = true; sp = 0; break; } }
                else {
                    const uint32_t orig = S.tri_orig[first + k];
                    if (t < bound || (found && t == bound && orig < best_tri)) { bound = t; best_tri = orig; found = true; }
                }
            }
            if (sp > 0) { --sp; cur = stack[sp * kBlock]; } else cur = kDone;
        }
    }
    if (!ANY && found) q.hit(bound, leaf, best_tri, lit);
}

// Closest / any-hit over a reference-shaped BSP tree (BspMesh.fs:67-76).  Nodes are visited in the reference's
// order (right subtree, then left; leaf triangles in list order), so "strictly smaller t wins" reproduces the
// stable sort; a node is entered iff the reference's own box test passes AND the box can still hold a usable
// hit (its entry distance is not beyond the current bound, with a margin far above the rounding of either side, and it does not
// lie wholly behind the ray's origin: a triangle hit needs t > 1e-7, Triangle.fs:62, and the sign of a computed distance is exact).
// PACKET = the wave walks the tree together (uniform stack in the lanes of a VGPR, scalar loads).
template <bool ANY, bool PACKET>
FT_DEV void mesh_bsp_query(const Scene& S, int32_t root, const Ray& r, Query<ANY>& q, uint32_t leaf, bool lit, int32_t* stack) {
    bool alive = q.active && !(ANY && q.blocked);
    if (!__any(alive)) return;
    const double ivx = 1.0 / r.dx, ivy = 1.0 / r.dy, ivz = 1.0 / r.dz;
    double bound = ANY ? q.max_dist : q.best_t;
    uint32_t best_tri = 0u;
    bool found = false;
    int sp = 0, stack_lanes = 0;
    int cur = PACKET ? root : (alive ? root : kDone);
    for (;;) {
        if (PACKET) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            if (cur >= 0) {
                cdp nd = S.nodes + 8ull * (uint32_t)cur;
                double entry = 0.0, exit = 0.0;
                const bool enter = alive && aabb_hit(nd, r, ivx, ivy, ivz, &entry, &exit) && !(entry > bound * (1.0 + 1e-12) + 1e-12) && !(exit < 0.0);
                if (__any(enter)) {
                    cip ch = reinterpret_cast<cip>(nd + 6);
                    const int left = ch[0];
                    stack_lanes = ((int)lane_id() == sp) ? left : stack_lanes;
                    ++sp; cur = ch[1];
                    continue;
                }
            } else {
                const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
                for (uint32_t k = 0; k < count; ++k) {
                    double t;
                    if (alive && tri_hit(S.tris + 9ull * (first + k), r, t)) {
                        if (ANY) { if (t < bound) { q.blocked = true; alive = false; } }
                        else if (t < bound) { bound = t; best_tri = first + k; found = true; }
                    }
                }
                if (ANY) { if (!__any(alive)) break; }
            }
            if (sp == 0) break;
            --sp;
            cur = __builtin_amdgcn_readlane(stack_lanes, sp);
        } else {
            if (!__any(cur != kDone)) break;
            while (cur >= 0) {
                cdp nd = S.nodes + 8ull * (uint32_t)cur;
                double entry = 0.0, exit = 0.0;
                if (aabb_hit(nd, r, ivx, ivy, ivz, &entry, &exit) && !(entry > bound * (1.0 + 1e-12) + 1e-12) && !(exit < 0.0)) {
                    cip ch = reinterpret_cast<cip>(nd + 6); stack[sp * kBlock] = ch[0]; ++sp; cur = ch[1];
                } else if (sp > 0) { --sp; cur = stack[sp * kBlock]; }
                else cur = kDone;
            }
            if (cur != kDone) {
                const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
                for (uint32_t k = 0; k < count; ++k) {
                    double t;
                    if (!tri_hit(S.tris + 9ull * (first + k), r, t)) continue;
                    if (ANY) { if (t < bound) { q.blocked = true; sp = 0; break; } }
                    else if (t < bound) { bound = t; best_tri = first + k; found = true; }
                }
                if (sp > 0) { --sp; cur = stack[sp * kBlock]; } else cur = kDone;
            }
        }
    }
    if (!ANY && found) q.hit(bound, leaf, best_tri, lit);
}

// The reference-shaped BSP walked by a COHERENT wavefront two levels at a time.  The host (ft_scene.cpp, widen_bsp) collapses a
// branch N and its two children into one 40-double record in the node array:
//   [0..5] box of N's RIGHT child R, [6..11] of its LEFT child L; [12..35] boxes of the grandchildren in the reference's visiting
//   order RR, RL, LR, LL (BspMesh.fs:73-75: right before left); [36..37] int32 child[4]: >= 0 the 64-byte unit of that grandchild's
//   own record, < 0 ~leaf, INT32_MIN empty (all-NaN box).  A child of N that is a leaf takes the first slot of its half.
// Branch boxes are the reference's own (BspMesh.fs:49) and are tested with the reference's arithmetic, (b - o) * (1 / d) per plane
// (BoundingBox.fs:41-56); a grandchild is entered by a lane only if its parent's box passes too, so the intermediate node gates its
// children exactly as the recursion does.  Leaves carry no box in the reference (BspMesh.fs:71); theirs is the inflated bound of their
// triangles, which can only reject a leaf none of whose triangles the ray can hit.  Williams' test decides on NaN (0 * inf: a zero
// direction component, origin on a slab plane) by falling through its comparisons; with every direction component of every live lane
// comfortably non-zero no NaN or infinity can arise and the test is max(entries) <= min(exits), which is what runs here - a wave with
// such a lane returns false and takes the one-node-at-a-time walk above.  Children are visited in the reference's order, so "strictly
// smaller t wins" still reproduces the stable sort; the pushes are static (LL, LR, RL; RR is taken directly).
template <bool ANY>
FT_DEV bool mesh_bsp_packet(const Scene& S, int32_t root, int32_t wide_root, const Ray& r, Query<ANY>& q, uint32_t leaf, bool lit) {
    bool alive = q.active && !(ANY && q.blocked);
    if (!__any(alive)) return true;
    const double adx = fabs(r.dx), ady = fabs(r.dy), adz = fabs(r.dz);
    if (__any(alive && !(adx >= 1e-280 && ady >= 1e-280 && adz >= 1e-280 && adx < 1e280 && ady < 1e280 && adz < 1e280))) return false;
    // dead lanes walk along with a harmless ray: no overflow, no NaN, and reach = -inf keeps them out of every decision
    const double ivx = alive ? 1.0 / r.dx : 1.0, ivy = alive ? 1.0 / r.dy : 1.0, ivz = alive ? 1.0 / r.dz : 1.0;
    const double ox = alive ? r.ox : 0.0, oy = alive ? r.oy : 0.0, oz = alive ? r.oz : 0.0;
    double bound = ANY ? q.max_dist : q.best_t;                    // a hit at t >= bound cannot change the query's result
    auto reach_of = [](double b) { const double m = b * (1.0 + 1e-12) + 1e-12; return m != m ? __builtin_inf() : m; };   // how far a box may begin and still matter
    double reach = alive ? reach_of(bound) : -__builtin_inf();
    // entry / exit distances of a box by the reference's arithmetic
#define FT_SLAB(bx, tmin, tmax) \
    double tmin, tmax; { double t0 = ((bx)[0] - ox) * ivx, t1 = ((bx)[3] - ox) * ivx; tmin = fmin(t0, t1); tmax = fmax(t0, t1); \
      t0 = ((bx)[1] - oy) * ivy; t1 = ((bx)[4] - oy) * ivy; tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1)); \
      t0 = ((bx)[2] - oz) * ivz; t1 = ((bx)[5] - oz) * ivz; tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1)); }
#define FT_ENTERED(tmin, tmax) (__builtin_amdgcn_ballot_w64(tmax >= fmax(tmin, 0.0)) & __builtin_amdgcn_ballot_w64(tmin <= reach))
    {   // the root's own box (its parent's record would have held it)
        cdp nd = S.nodes + 8ull * (uint32_t)root;
        FT_SLAB(nd, tmin, tmax);
        const bool in = (tmax >= fmax(tmin, 0.0)) & (tmin <= reach);
        alive = alive & in; reach = in ? reach : -__builtin_inf();
        if (!__any(alive)) return true;
    }
    uint32_t best_tri = 0u;
    bool found = false;
    int stack_lanes = 0, sp = 0, cur = wide_root;
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        sp = __builtin_amdgcn_readfirstlane(sp);
        if (cur >= 0) {
            cdp nd = S.nodes + 8ull * (uint32_t)cur;
            const int32_t ch[4] = {reinterpret_cast<cip>(nd + 36)[0], reinterpret_cast<cip>(nd + 36)[1], reinterpret_cast<cip>(nd + 36)[2], reinterpret_cast<cip>(nd + 36)[3]};
            unsigned long long half[2], m[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) { FT_SLAB(nd + 6 * h, tmin, tmax); half[h] = FT_ENTERED(tmin, tmax); }
#pragma unroll
            for (int c = 0; c < 4; ++c) { FT_SLAB(nd + 12 + 6 * c, tmin, tmax); m[c] = FT_ENTERED(tmin, tmax) & half[c >> 1]; }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FT_PUSH(c) asm("s_mov_b32 m0, %1\n\tv_writelane_b32 %0, %2, m0\n\ts_cmp_lg_u64 %3, 0\n\ts_addc_u32 %1, %1, 0" : "+v"(stack_lanes), "+s"(sp) : "s"(ch[c]), "s"(m[c]) : "m0", "scc")
            FT_PUSH(3); FT_PUSH(2); FT_PUSH(1);
#undef FT_PUSH
#pragma clang diagnostic pop
            if (m[0]) { cur = ch[0]; continue; }
        } else {
            const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
            for (uint32_t k = 0; k < count; ++k) {                 // list order; wave-uniform: scalar loads
                double t = 0.0;
                const bool h = tri_hit_wave(S.tris + 9ull * (first + k), r, alive, t);
                if (ANY) {
                    const bool b = h & (t < bound);
                    q.blocked = q.blocked | b; alive = alive & !b; reach = b ? -__builtin_inf() : reach;
                } else if (__any(h)) {
                    const bool nearer = h & (t < bound);
                    bound = nearer ? t : bound; best_tri = nearer ? first + k : best_tri; found = found | nearer;
                    reach = nearer ? reach_of(t) : reach;
                }
            }
            if (ANY) { if (!__any(alive)) break; }
        }
        if (sp == 0) break;
        --sp;
        cur = __builtin_amdgcn_readlane(stack_lanes, sp);
    }
#undef FT_SLAB
#undef FT_ENTERED
    if (!ANY && found) q.hit(bound, leaf, best_tri, lit);
    return true;
}

// The same query for a COHERENT wavefront (primary rays of one 8x8 pixel block and their shadow rays):
// the 64 rays walk the BVH together.  The node stack and the current node are wave-uniform (the stack lives
// in the lanes of one VGPR: lane i holds entry i, popped with v_readlane), node and triangle data come through scalar loads,
// and there is no per-lane control flow: a node is entered if ANY live lane's slab test passes.  Lanes that
// would have pruned a node on their own only run tests that cannot produce a usable hit for them (a
// triangle lies inside its boxes), so the per-ray result equals mesh_bvh_query's.
template <bool ANY>
FT_DEV void mesh_bvh_packet(const Scene& S, int32_t wide_root, const Ray& r, Query<ANY>& q, uint32_t leaf, bool lit) {
    bool alive = q.active && !(ANY && q.blocked);
    if (!__any(alive)) return;
    // 1/d, kept finite: a direction component of (nearly) zero stands for 1e150 of its sign, which orders the two planes of a slab
    // exactly like the infinite value would against every finite distance, without the inf - inf of the fused form below.
    const double ivx = fabs(r.dx) < 1e-150 ? copysign(1e150, r.dx) : 1.0 / r.dx, ivy = fabs(r.dy) < 1e-150 ? copysign(1e150, r.dy) : 1.0 / r.dy,
                 ivz = fabs(r.dz) < 1e-150 ? copysign(1e150, r.dz) : 1.0 / r.dz;
    const double oix = r.ox * ivx, oiy = r.oy * ivy, oiz = r.oz * ivz;
    double bound = ANY ? q.max_dist : q.best_t;
    double reach = alive ? bound : -__builtin_inf();               // how far a box may lie and still matter to this lane: nowhere for a dead one
    uint32_t best_tri = 0xFFFFFFFFu;
    bool found = false;
    int stack_lanes = 0;                                           // lane i holds stack entry i (at most 3 per level of the wide tree)
    int sp = 0;
    int cur = wide_root;
    // The boxes of a node's four children live in the node (ft_flat.h): one scalar-load round trip decides four subtrees.
    const unsigned long long live0 = __ballot(alive);
    // majority direction per axis (bit a set: most live rays travel towards +a): which child is nearer
    const uint32_t oct = (2 * __popcll(__ballot(alive && r.dx >= 0.0)) >= __popcll(live0) ? 1u : 0u) | (2 * __popcll(__ballot(alive && r.dy >= 0.0)) >= __popcll(live0) ? 2u : 0u) |
                         (2 * __popcll(__ballot(alive && r.dz >= 0.0)) >= __popcll(live0) ? 4u : 0u);
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        sp = __builtin_amdgcn_readfirstlane(sp);                    // wave-uniform by construction; said here so that the stack arithmetic stays on the scalar unit
        if (cur >= 0) {
            cdp nd = S.wide + (unsigned long long)kWideNodeDoubles * (uint32_t)cur;
            // children and axes are read up front, with the first boxes: fetched one by one behind each child's test, every
            // one of them was a scalar-load round trip of its own
            const int32_t ch[4] = {reinterpret_cast<cip>(nd + 24)[0], reinterpret_cast<cip>(nd + 24)[1], reinterpret_cast<cip>(nd + 24)[2], reinterpret_cast<cip>(nd + 24)[3]};
            const uint32_t axes = reinterpret_cast<cup>(nd + 26)[0];
            unsigned long long m[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                cdp bx = nd + 6 * c;
                // (b - o) / d as one fused multiply-add per plane, b * (1/d) - o * (1/d).  Its rounding error is of the order of
                // 1e-16 |o| in space, far inside the 1e-7 x extent by which the builder inflates every box.
                double t0 = __builtin_fma(bx[0], ivx, -oix), t1 = __builtin_fma(bx[3], ivx, -oix);
                double tmin = fmin(t0, t1), tmax = fmax(t0, t1);
                t0 = __builtin_fma(bx[1], ivy, -oiy); t1 = __builtin_fma(bx[4], ivy, -oiy);
                tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1));
                t0 = __builtin_fma(bx[2], ivz, -oiz); t1 = __builtin_fma(bx[5], ivz, -oiz);
                tmin = fmax(tmin, fmin(t0, t1)); tmax = fmin(tmax, fmax(t0, t1));
                // dead lanes carry reach = -inf, so the two comparisons are the whole test (no short-circuit: no exec-mask games)
                // (one ballot per comparison: the compiler turns a ballot of anything but a bare comparison into a select and a second compare);
                // an absent child's box is tested like any other - the builders fill it with NaN, which no comparison passes - so that no
                // branch stands between the node's loads
                m[c] = __builtin_amdgcn_ballot_w64(tmax >= fmax(tmin, 0.0)) & __builtin_amdgcn_ballot_w64(tmin <= reach);
            }
            // At most one child entered (the usual case near the leaves): no order to work out, nothing to push.
            {
                uint32_t n_in = 0u;                                  // counted on the scalar unit (the portable sum goes through two vector selects)
                asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, 0\n\ts_cmp_lg_u64 %2, 0\n\ts_addc_u32 %0, %0, 0\n\ts_cmp_lg_u64 %3, 0\n\ts_addc_u32 %0, %0, 0\n\ts_cmp_lg_u64 %4, 0\n\ts_addc_u32 %0, %0, 0"
                    : "+s"(n_in) : "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]) : "scc");
                if (n_in == 1u) { cur = m[0] ? ch[0] : m[1] ? ch[1] : m[2] ? ch[2] : ch[3]; continue; }
                if (n_in == 0u) { if (sp == 0) break; --sp; cur = __builtin_amdgcn_readlane(stack_lanes, sp); continue; }
            }
            // Visiting order, nearest first by the majority directions: halves by the node's axis, slots within a half by the child's.
            // The entered children go on the stack far to near and the common pop below takes the nearest; one of eight fixed
            // sequences is picked by three bits, so every push names its child statically (selecting m[c] / ch[c] by a computed c
            // cost ~250 scalar instructions per node - eight-byte select chains - against ~80 vector ones: the scalar pipe, one
            // per CU, was the busier half of this kernel).
            const uint32_t order = (((oct >> (axes & 3u)) & 1u) << 2) | (((oct >> ((axes >> 8) & 3u)) & 1u) << 1) | ((oct >> ((axes >> 16) & 3u)) & 1u);
            // a push is one v_writelane at the top slot (free when nothing is entered: the write is then simply not kept) and a scalar add;
            // the lane select goes through m0 (one scalar operand per vector instruction), which the compiler reserves and nothing else here uses
            // (naming it as clobbered keeps the operands out of it; the compiler remarks that the register is reserved)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FT_PUSH(c) asm("s_mov_b32 m0, %1\n\tv_writelane_b32 %0, %2, m0\n\ts_cmp_lg_u64 %3, 0\n\ts_addc_u32 %1, %1, 0" : "+v"(stack_lanes), "+s"(sp) : "s"(ch[c]), "s"(m[c]) : "m0", "scc")
            // the nearest child is not pushed and popped again: when it is entered the walk goes straight on with it
#define FT_NEXT(c) if (m[c]) { cur = ch[c]; continue; }
            switch (order) {                                        // bit 2: left half first; bit 1: slot 0 before 1; bit 0: slot 2 before 3
                case 7: FT_PUSH(3); FT_PUSH(2); FT_PUSH(1); FT_NEXT(0); break;     // near to far 0 1 2 3
                case 6: FT_PUSH(2); FT_PUSH(3); FT_PUSH(1); FT_NEXT(0); break;     // 0 1 3 2
                case 5: FT_PUSH(3); FT_PUSH(2); FT_PUSH(0); FT_NEXT(1); break;     // 1 0 2 3
                case 4: FT_PUSH(2); FT_PUSH(3); FT_PUSH(0); FT_NEXT(1); break;     // 1 0 3 2
                case 3: FT_PUSH(1); FT_PUSH(0); FT_PUSH(3); FT_NEXT(2); break;     // 2 3 0 1
                case 2: FT_PUSH(1); FT_PUSH(0); FT_PUSH(2); FT_NEXT(3); break;     // 3 2 0 1
                case 1: FT_PUSH(0); FT_PUSH(1); FT_PUSH(3); FT_NEXT(2); break;     // 2 3 1 0
                default: FT_PUSH(0); FT_PUSH(1); FT_PUSH(2); FT_NEXT(3); break;    // 3 2 1 0
            }
#undef FT_NEXT
#undef FT_PUSH
#pragma clang diagnostic pop
        } else {
            const uint32_t first = S.bsp_leaves[2 * (~cur)], count = S.bsp_leaves[2 * (~cur) + 1];
            for (uint32_t k = 0; k < count; ++k) {                 // wave-uniform: scalar loads
                double t = 0.0;
                const bool h = tri_hit_wave(S.tris + 9ull * (first + k), r, alive, t);
                if (ANY) {
                    const bool b = h & (t < bound);
                    q.blocked = q.blocked | b; alive = alive & !b; reach = b ? -__builtin_inf() : reach;
                } else if (__any(h)) {                              // a live lane's reach IS its nearest distance so far
                    const uint32_t orig = S.tri_orig[first + k];
                    const bool nearer = h & ((t < reach) | (found & (t == reach) & (orig < best_tri)));
                    reach = nearer ? t : reach; best_tri = nearer ? orig : best_tri; found = found | nearer;
                }
            }
            if (ANY) { if (!__any(alive)) break; }
        }
        if (sp == 0) break;
        --sp;
        cur = __builtin_amdgcn_readlane(stack_lanes, sp);
    }
    if (!ANY && found) q.hit(reach, leaf, best_tri, lit);
}

// ---------------------------------------------------------------------------------------------
// Wave-level pre-test of the item culls.  OP_CULL tests one item against the 64 rays of the wave, 64 lanes x ~40
// FP64 instructions per item; scenes with tens of items spend most of a coherent batch there.  Here the roles are
// swapped once per query: the rays of the wave are bounded by a cone (apex = origin of the first live lane, padded by
// the largest origin distance; axis = that lane's direction; half-angle = largest deviation) and lane k tests ITEM k's
// bounding sphere against that cone, so up to 64 items cost one pass of ~25 FP32 instructions.  Conservative by
// construction: a sphere is dropped only when it lies wholly beyond the tangent plane of the cone nearest to it, all
// float roundings are covered by explicit slack, and items with a parallel-sensitive face direction (Plane.fs:13-16)
// that some ray of the wave is nearly parallel to are always kept.  Survivors still run the exact per-ray test of OP_CULL.
FT_DEV float row16_reduce_min(float v) {
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));    // quad_perm [1,0,3,2]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));    // quad_perm [2,3,0,1]
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));   // row_half_mirror
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));   // row_mirror
    return v;
}
FT_DEV float wave_min(float v) {                                    // all 64 lanes must be executing
    v = row16_reduce_min(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
constexpr int kSparseLanes = 8;                                     // at most this many live rays: exact_cull tests lane = item
struct ItemMask { unsigned long long lo, hi; bool valid; };   // bit k: top-level item k may be hit by some ray of the wave (items >= 128: not covered)
// A bundle of rays bounded by a cone: apex c (origins within rho of it), unit axis a, half-angle given by cos_t (rounded down) /
// sin_t (rounded up); par_rows = face directions some ray of the bundle may be nearly parallel to.
struct Cone { float ax, ay, az, cx, cy, cz, cos_t, sin_t, rho; uint32_t par_rows; float far; };   // far: no usable hit lies further from the apex than this (+inf: unbounded)
// One item (8-float record: centre, radius, face-direction mask, ...) against one cone: false only when no ray inside the cone can
// give a usable hit on the item.
FT_DEV bool cone_may_reach(float ix, float iy, float iz, float radius, uint32_t rows, const Cone& B, float origin_mag) {
    const float vx = ix - B.cx, vy = iy - B.cy, vz = iz - B.cz;
    const float slack = 1e-5f * (1.0f + origin_mag + fabsf(ix) + fabsf(iy) + fabsf(iz));
    const float reach = radius * 1.0001f + B.rho + slack;          // sphere radius + origin spread + rounding slack
    const float h = vx * B.ax + vy * B.ay + vz * B.az;
    const float v2 = vx * vx + vy * vy + vz * vz;
    const float w = sqrtf(fmaxf(0.0f, v2 - h * h));
    const float out_of_range = B.far + reach;                      // the whole sphere lies further from the apex than any usable hit
    if (v2 > out_of_range * out_of_range * 1.0001f && !((rows & B.par_rows) != 0u)) return false;
    // lower bound of the distance from the centre to the cone: beyond its nearest tangent plane in front of the apex;
    // behind the apex the cone also lies within the half-space (x - apex).axis >= 0
    const float beyond = h > 0.0f ? w * B.cos_t - h * B.sin_t : fmaxf(w * B.cos_t, -h);
    return !(beyond > reach) || (rows & B.par_rows) != 0u;
}
// Lane k tests top-level ITEM k: bit k of the result is clear only when no ray inside the cone can give a usable hit on it.  The records are
// requested (item_recs) before the cone is worked out - some 150 instructions, two wave reductions - so their way from memory is covered.
struct ItemRecs { float x[2], y[2], z[2], r[2]; uint32_t rows[2]; };
FT_DEV ItemRecs item_recs(const Scene& S) {
    ItemRecs R{{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0u, 0u}};
    const int n_pass = S.n_items > 64 ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int item = pass * 64 + (int)lane_id();
        if (item < S.n_items) {
            const float* I = S.cull_items + 8 * (item + (int)opaque_zero());   // (worked out here: hoisted out of the batch loop, the address was spilled)
            R.x[pass] = I[0]; R.y[pass] = I[1]; R.z[pass] = I[2]; R.r[pass] = I[3]; R.rows[pass] = __float_as_uint(I[4]);
        }
    }
    return R;
}
FT_DEV ItemMask items_in_cone(const Scene& S, const Cone& B, const ItemRecs& R) {
    ItemMask M{~0ull, ~0ull, true};
    const float origin_mag = fabsf(B.cx) + fabsf(B.cy) + fabsf(B.cz);
    const int n_pass = S.n_items > 64 ? 2 : 1;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int item = pass * 64 + (int)lane_id();
        bool keep = true;
        if (item < S.n_items) keep = cone_may_reach(R.x[pass], R.y[pass], R.z[pass], R.r[pass], R.rows[pass], B, origin_mag);
        const unsigned long long km = __ballot(keep && item < S.n_items);
        if (pass == 0) M.lo = km; else M.hi = km;
    }
    if (n_pass == 1) M.hi = 0ull;
    return M;
}
// Face directions (Plane.fs:13-16: a ray within 2 kEps of parallel to a plane hits it at its own origin) that some ray of the wave is
// nearly parallel to: items using one of them are kept whatever the cone says.  Per direction that is an exact FP64 product per lane and a
// wave vote - forty scalar loads and twenty votes per query on night-house, for a mask that is almost always empty.  So lane k first
// asks of direction k whether ANY unit vector u within the cone could have |row . u| < 2 kEps / |d| (|d| >= 1e-3 is the caller's
// condition): |row . u| >= |row . a| cos t - |row| sin t, in floats with the roundings covered; only the directions that cannot be
// ruled out this way (none, unless the bundle grazes a face) get the exact test.
FT_DEV uint32_t rows_nearly_parallel(const Scene& S, const Ray& r, bool live, float ax, float ay, float az, float cos_t, float sin_t) {
    bool cand = false;
    if ((int)lane_id() < S.n_cull_rows) {
        const float* Rf = S.cull_items + 8 * S.n_items + 4 * (int)lane_id();
        cand = !(fabsf(Rf[0] * ax + Rf[1] * ay + Rf[2] * az) * cos_t > Rf[3] * (sin_t + 2e-5f) + 2.2e-4f);
    }
    uint32_t todo = (uint32_t)__ballot(cand), par_rows = 0u;
    while (todo) {
        const uint32_t k = (uint32_t)__builtin_ctz(todo); todo &= todo - 1u;
        cdp Rw = S.cull_rows + 3u * k;
        if (__any(live && fabs(dot3(Rw[0], Rw[1], Rw[2], r.dx, r.dy, r.dz)) < 2.0 * kEps)) par_rows |= 1u << k;
    }
    return par_rows;
}
FT_DEV float wave_max(float v) { return -wave_min(-v); }
// Shadow rays towards a point light (Shading.fs:38-42: d = normalise (position - point), maxDistance = |position - point|): every ray
// of the wave passes through the light, so the bundle is a cone whose APEX is the light - no origin spread to pad it with - and a
// usable hit (0 <= t < maxDistance, Scene.fs:121) lies between the light and the ray's origin: no further from the apex than the
// longest segment.  Items behind the light fall behind the apex, items beyond the surface patch beyond `far`.
FT_DEV ItemMask bundle_cull_to_light(const Scene& S, const Ray& r, bool live, cdp light, double max_dist) {
    ItemMask M{~0ull, ~0ull, false};
    const unsigned long long lm = __ballot(live);
    if (lm == 0ull) return M;
    const ItemRecs recs = item_recs(S);
    float ex = -(float)r.dx, ey = -(float)r.dy, ez = -(float)r.dz;  // from the light towards the ray's origin
    const float l2 = ex * ex + ey * ey + ez * ez;
    const float far_lane = (float)max_dist;
    if (__any(live && !(l2 > 0.25f && l2 < 4.0f && far_lane < 1e30f))) return M;   // the directions are unit vectors unless the point sits on the light
    const float inv = __builtin_amdgcn_rsqf(l2);
    ex *= inv; ey *= inv; ez *= inv;
    const int first = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)lm) - 1);
    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), first)), ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ey), first)),
                az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ez), first));
    const float cos_t = wave_min(live ? ax * ex + ay * ey + az * ez : 1.0f) - 1e-5f;
    if (!(cos_t > 0.3f)) return M;
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t)) + 1e-5f;
    const float cx = (float)light[0], cy = (float)light[1], cz = (float)light[2];
    const float far = wave_max(live ? far_lane : 0.0f);
    const float rho = 1e-5f * (1.0f + fabsf(cx) + fabsf(cy) + fabsf(cz) + far);     // the float images of the light and of the directions
    const uint32_t par_rows = rows_nearly_parallel(S, r, live, ax, ay, az, cos_t, sin_t);   // (|d| is within [0.5, 2] here)
    return items_in_cone(S, Cone{ax, ay, az, cx, cy, cz, cos_t, sin_t, rho, par_rows, far * 1.0001f + rho}, recs);
}
FT_DEV ItemMask bundle_cull(const Scene& S, const Ray& r, bool live) {
    ItemMask M{~0ull, ~0ull, false};
    if (S.n_items < 3 || S.n_cull_rows < 0) return M;
    const unsigned long long lm = __ballot(live);
    if (lm == 0ull) return M;
    const ItemRecs recs = item_recs(S);
    float dx = (float)r.dx, dy = (float)r.dy, dz = (float)r.dz;
    const float l2 = dx * dx + dy * dy + dz * dz;
    if (__any(live && !(l2 > 1e-6f && l2 < 1e30f))) return M;       // tiny / huge / NaN directions: no bound (rows_nearly_parallel counts on |d| >= 1e-3)
    const float inv = __builtin_amdgcn_rsqf(l2);
    dx *= inv; dy *= inv; dz *= inv;
    const float ox = (float)r.ox, oy = (float)r.oy, oz = (float)r.oz;
    const int first = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)lm) - 1);
    const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dx), first)), ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dy), first)),
                az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dz), first));
    const float cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ox), first)), cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oy), first)),
                cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(oz), first));
    const float ex = ox - cx, ey = oy - cy, ez = oz - cz;
    const float cos_dev = ax * dx + ay * dy + az * dz;
    const float spread2 = ex * ex + ey * ey + ez * ez;
    const float cos_t = wave_min(live ? cos_dev : 1.0f) - 1e-5f;    // cos of the half-angle, made smaller (cone wider)
    const float rho2 = -wave_min(live ? -spread2 : 0.0f);
    if (!(cos_t > 0.3f) || !(rho2 < 1e30f)) return M;               // a wide bundle bounds nothing
    const float sin_t = sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t)) + 1e-5f;   // sine of the widened half-angle, rounded up
    const float rho = sqrtf(rho2) * 1.0001f;
    // rays of the wave nearly parallel to a face direction: items using that direction are kept (exact FP64 test as in OP_CULL)
    const uint32_t par_rows = rows_nearly_parallel(S, r, live, ax, ay, az, cos_t, sin_t);
    return items_in_cone(S, Cone{ax, ay, az, cx, cy, cz, cos_t, sin_t, rho, par_rows, __builtin_inff()}, recs);
}

// Exact skip test of one top-level item (cull record C: centre, radius^2, number of face directions, the directions) for one ray.
template <class P>
FT_DEV bool item_missed_by(P C, double ox, double oy, double oz, double dx, double dy, double dz) {
    const double ocx = ox - C[0], ocy = oy - C[1], ocz = oz - C[2];
    const double dd = dot3(dx, dy, dz, dx, dy, dz), b = dot3(ocx, ocy, ocz, dx, dy, dz), cc = dot3(ocx, ocy, ocz, ocx, ocy, ocz);
    // (1) the squared distance from the centre to the ray's LINE exceeds the inflated radius: no hit at any t;
    // (2) the origin is outside the sphere and moving away: every hit of the item has t < 0, which neither
    //     closest (Scene.fs:115) nor lightIsBocked (Scene.fs:121) ever uses (CSG state inside the item is moot).
    // (A distance test against the light / the closest hit so far was measured: its two square roots per item cost
    //  more than the extra skips return.)
    bool miss = (cc * dd - b * b) > (C[3] * dd + 1e-12 * (cc * dd)) && dd > 0.0;
    if (cc > C[3] && b > 0.0) miss = true;
    const int n_rows = (int)C[4];
    for (int k = 0; k < n_rows; ++k)                               // near-parallel to a plane-derived face: Plane.fs:13-16 may hit at the origin
        if (fabs(dot3(C[5 + 3 * k], C[6 + 3 * k], C[7 + 3 * k], dx, dy, dz)) < 2.0 * kEps) miss = false;
    return miss;
}
// ... for this lane's ray, the record through scalar loads (see OP_CULL).
FT_DEV bool item_missed(const Scene& S, uint32_t item, const Ray& r) { return item_missed_by(S.culls + 24ull * item, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz); }
FT_DEV double readlane_f64(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The same test for every item up front (incoherent waves): the loads of consecutive records do not depend on each other,
// unlike the walk through the program from one OP_CULL to the next, and the wave then visits only what some lane needs.
FT_DEV ItemMask exact_cull(const Scene& S, const Ray& r, bool live) {
    ItemMask M{0ull, 0ull, false};
    const int n = S.n_items < 128 ? S.n_items : 128;
    if (S.n_items < 3) return ItemMask{~0ull, ~0ull, false};
    const unsigned long long lm = __ballot(live);
    if (__popcll(lm) <= kSparseLanes) {
        // Few live rays (the late bounces: a handful of lanes per wave): the roles swap.  Lane k holds ITEM k's record and the live
        // rays are broadcast one at a time, so the wave pays a few dozen instructions per ray instead of a scalar-load round trip
        // and a test per item (measured on hollow-sphere x1: the tail's eight bounces are a chain of such walks).
        for (int pass = 0; pass * 64 < n; ++pass) {
            const int item = pass * 64 + (int)lane_id();
            cdp C = S.culls + 24ull * (uint32_t)(item < n ? item : 0);   // per-lane record: vector loads (the address differs per lane)
            bool need = false;
            unsigned long long todo = lm;
            while (todo) {
                const int src = (int)__builtin_ctzll(todo); todo &= todo - 1ull;
                if (!item_missed_by(C, readlane_f64(r.ox, src), readlane_f64(r.oy, src), readlane_f64(r.oz, src), readlane_f64(r.dx, src), readlane_f64(r.dy, src), readlane_f64(r.dz, src))) need = true;
            }
            const unsigned long long km = __ballot(need && item < n);
            if (pass == 0) M.lo = km; else M.hi = km;
        }
        M.valid = true;
        return M;
    }
    // A float image of the same test comes first: it may only say "certainly missed" (every rounding of the float arithmetic is covered
    // by explicit slack, a ray nearly parallel to one of the item's face directions is never turned away), and the exact test - a
    // 192-byte record and ~45 FP64 instructions - only runs for the items some lane's float test could not rule out.  An incoherent
    // wave meets most of a scene's items this way (hollow-sphere: 26 per query, 4 of them needed).
    const float ox = (float)r.ox, oy = (float)r.oy, oz = (float)r.oz;
    float dx = (float)r.dx, dy = (float)r.dy, dz = (float)r.dz;
    const float dd = dx * dx + dy * dy + dz * dz;
    const bool tame = live && dd > 1e-30f && dd < 1e30f && fabsf(ox) < 1e15f && fabsf(oy) < 1e15f && fabsf(oz) < 1e15f;   // else: no float verdicts for this lane
    { const float inv = __builtin_amdgcn_rsqf(dd); dx *= inv; dy *= inv; dz *= inv; }   // a unit direction (to a few ulps: the 1e-4 slack below covers it) drops |d|^2 out of every comparison
    uint32_t par_lane = 0;                                         // face directions this lane's ray is nearly parallel to (Plane.fs:13-16)
    const bool rows_known = S.n_cull_rows >= 0;                    // (more than 32 distinct directions in the scene: the table does not exist)
    for (int k = 0; k < S.n_cull_rows; ++k) {
        cdp Rw = S.cull_rows + 3u * (uint32_t)k;
        if (fabs(dot3(Rw[0], Rw[1], Rw[2], r.dx, r.dy, r.dz)) < 2.000001 * kEps) par_lane |= 1u << k;
    }
    // The float records sit in the lanes (lane j: items j and 64 + j, one vector load each before the loop) and are handed round by
    // v_readlane: fetched one by one through scalar loads, every item of the loop began with a memory round trip of its own - the
    // fixed ~17 us a batch of incoherent rays cost whatever it held was mostly this loop, twice (closest, then shadow).
    struct ItemRec { float x, y, z, r2, rows; };                  // r2: the bounding sphere's radius squared, inflated (+inf: unbounded - no comparison below holds)
    auto load_rec = [&](int item) {
        const float* I = S.cull_items + 8 * ((item < n ? item : 0) + (int)opaque_zero());
        const float rad = I[3];
        return ItemRec{I[0], I[1], I[2], rad < 1e30f ? rad * rad * 1.001f : __builtin_inff(), I[4]};
    };
    const ItemRec lo = load_rec((int)lane_id());
    ItemRec hi{0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (n > 64) hi = load_rec(64 + (int)lane_id());
    auto lane_of = [](float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); };
    for (int k = 0; k < n; ++k) {
        const int src = k & 63;
        const bool first = k < 64;                                  // wave-uniform
        const float I[5] = {lane_of(first ? lo.x : hi.x, src), lane_of(first ? lo.y : hi.y, src), lane_of(first ? lo.z : hi.z, src),
                            lane_of(first ? lo.r2 : hi.r2, src), lane_of(first ? lo.rows : hi.rows, src)};
        const float cx = ox - I[0], cy = oy - I[1], cz = oz - I[2];
        const uint32_t rows = __float_as_uint(I[4]);
        const float b = cx * dx + cy * dy + cz * dz, cc = cx * cx + cy * cy + cz * cz, b2 = b * b;
        const float margin = cc * 0.9999f - I[3];                   // |c|^2 less the slack for every rounding above, less the inflated radius^2
        const bool line_misses = margin > b2;                       // the line passes the bounding sphere by a margin: |c|^2 - (c.u)^2 > r^2 + slack
        const bool leaves = margin > 0.0f && b > 0.0f && b2 > 1e-6f * cc;   // outside it and moving away: every hit has t < 0
        const bool certainly_missed = tame && rows_known && (line_misses || leaves) && (rows & par_lane) == 0u;
        if (!__any(live && !certainly_missed)) continue;
        // What the float test could not rule out is kept as it is: the mask only has to hold every item some lane can hit, and the FP64
        // image of the same test (item_missed: a 192-byte record through scalar loads, one exposed round trip per surviving item - 14 K
        // of the 21 K cycles this function took per query on hollow-sphere) turned away almost nothing the float test had let through.
        // Only a scene without the direction table (more than 32 distinct face directions) still needs it: there the float test says nothing.
        if (!rows_known && !__any(live && !item_missed(S, (uint32_t)k, r))) continue;
        if (k < 64) M.lo |= 1ull << k; else M.hi |= 1ull << (k - 64);
    }
    M.valid = true;
    return M;
}

#ifdef FT_ITEM_COUNTS
// Diagnostic build only: shader-clock cycles waves spend in sections of the tracing kernels (s_memtime; summed over waves).
// Every wave adds to 48 words of its own (no-return atomics on addresses nobody shares: shared words made the build eight times slower).
__device__ unsigned long long g_clk[8192 * 48];
#define FT_CLK_SLOT() (((blockIdx.x * (kBlock / 64) + threadIdx.x / 64) & 8191u) * 48u)
#define FT_CLK_NOW() __builtin_amdgcn_s_memtime()
#define FT_CLK_ADD(k, t0) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); if (lane_id() == 0) atomicAdd(&g_clk[FT_CLK_SLOT() + 16 + (k)], t1_ - (t0)); } while (0)
#define FT_CLK_INC(k) do { if (lane_id() == 0) atomicAdd(&g_clk[FT_CLK_SLOT() + 16 + (k)], 1ull); } while (0)
#else
#define FT_CLK_NOW() 0ull
#define FT_CLK_ADD(k, t0) do { (void)(t0); } while (0)
#define FT_CLK_INC(k) do { } while (0)
#endif
#ifdef FT_ITEM_COUNTS
// Diagnostic build only (tools/item_counts.py): how many top-level items a wave's query evaluates.  [4 * kind + k], kind = ANY * 2 + incoherent:
// k = 0 queries, 1 items evaluated, 2 live lanes, 3 items the mask offered (before each item's own OP_CULL).
#define FT_COUNT(k, v) do { const unsigned long long v_ = (unsigned long long)(v); if (lane_id() == 0) atomicAdd(&g_clk[FT_CLK_SLOT() + 4 * ((ANY ? 2 : 0) + (coherent ? 0 : 1)) + (k)], v_); } while (0)
#else
#define FT_COUNT(k, v) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------
// The scene program interpreter: Scene.intersect (Scene.fs:67-104) + closest / lightIsBocked.
// MESH = false compiles the triangle / BSP / BVH code out: scenes without meshes then run kernels with
// markedly fewer registers.
template <bool ANY, bool MESH>
FT_DEV void trace(const Scene& S, const Ray& r, Query<ANY>& q, uint32_t* lds, bool& overflow, bool coherent = false, cdp to_light = nullptr) {
    HitList L;
    L.init(lds, S.csg_cap, S.csg_rows, S.lane_fold);
    int32_t* stack = reinterpret_cast<int32_t*>(lds + 4 * S.csg_rows * kBlock) + threadIdx.x;
    // Coherent waves visit only the top-level items their ray bundle can reach (ascending, so ties between items still
    // go to the earlier one); everything else walks the whole program, every item behind its own OP_CULL.
    ItemMask IM{~0ull, ~0ull, false};
    const bool live = ANY ? (q.active && !q.blocked) : q.active;
    const unsigned long long clk0 = FT_CLK_NOW();
    if (coherent) IM = (ANY && to_light && S.n_items >= 3 && S.n_cull_rows >= 0) ? bundle_cull_to_light(S, r, live, to_light, q.max_dist) : bundle_cull(S, r, live);
    const bool exact_mask = !IM.valid;                             // the bundle bounded nothing (or was not tried): per-ray tests up front
    if (exact_mask) IM = exact_cull(S, r, live);
    FT_CLK_ADD(4 * ((ANY ? 2 : 0) + (coherent ? 0 : 1)) + 0, clk0);
    const unsigned long long clk1 = FT_CLK_NOW();
    FT_COUNT(0, 1); FT_COUNT(2, __popcll(__ballot(live))); FT_COUNT(3, IM.valid ? __popcll(IM.lo) + __popcll(IM.hi) : S.n_items);
    uint32_t item_end = 0xFFFFFFFFu;
    for (uint32_t pc = 0;; ++pc) {
        if (IM.valid && (item_end == 0xFFFFFFFFu || pc >= item_end)) {
            int k;
            if (IM.lo) { k = (int)__builtin_ctzll(IM.lo); IM.lo &= IM.lo - 1ull; }
            else if (IM.hi) { k = 64 + (int)__builtin_ctzll(IM.hi); IM.hi &= IM.hi - 1ull; }
            else if (S.n_items > 128) { k = 128; IM.valid = false; }       // the tail beyond the mask runs linearly
            else break;
            const uint32_t at = S.item_pc[k];
            pc = at & 0x7FFFFFFFu;
            // its OP_CULL was already evaluated for every lane (exact mask) - or, for the closest hits of a bundle, turns almost nothing
            // away that the cone test let through (measured: 1.23 items offered, 1.22 evaluated per query on hollow-sphere; 1.48 / 1.46 on
            // night-house) and costs ~45 FP64 instructions per item; shadow bundles keep it (it rejects 12 - 40 % of what they are offered)
            if (IM.valid && (at >> 31) && (exact_mask || !ANY)) pc += 2;
            item_end = IM.valid ? (S.item_pc[k + 1] & 0x7FFFFFFFu) : 0xFFFFFFFFu;
        }
        const uint32_t ins = S.program[pc];                        // wave-uniform: scalar load
        const uint32_t op = ins & 0xFFu, arg = ins >> 8;
        if (op == OP_END) break;
        switch (op) {
            case OP_LEAF_FOLD: {
                const LeafHead H = leaf_head(S, arg);
                const bool lit = (H.flags & LF_LIT) != 0;
                if (ANY && !lit) break;                            // an unlit object never blocks light (Scene.fs:121)
                FT_COUNT(1, 1);
                if (MESH && H.kind == LK_MESH) {
                    const int32_t bvh = S.meshes[4 * H.mesh + 3];
                    if (bvh >= 0) {
                        Ray rm;
                        to_model(S.leaves + 16ull * arg, (H.flags & LF_XFORM) != 0, r, rm);
                        if (coherent) mesh_bvh_packet<ANY>(S, S.mesh_wide[H.mesh], rm, q, arg, lit);
                        else mesh_bvh_query<ANY>(S, bvh, rm, q, arg, lit, stack);
                        break;
                    }
                    const int32_t bsp_root = S.meshes[4 * H.mesh];
                    if (bsp_root >= 0) {
                        Ray rm;
                        to_model(S.leaves + 16ull * arg, (H.flags & LF_XFORM) != 0, r, rm);
                        if (coherent) {
                            const int32_t wide_root = S.mesh_wide[H.mesh];     // the tree two levels at a time (none: deeper than the packet's stack)
                            if (wide_root == INT32_MIN || !mesh_bsp_packet<ANY>(S, bsp_root, wide_root, rm, q, arg, lit)) mesh_bsp_query<ANY, true>(S, bsp_root, rm, q, arg, lit, stack);
                        } else mesh_bsp_query<ANY, false>(S, bsp_root, rm, q, arg, lit, stack);
                        break;
                    }
                }
                leaf_hits<MESH>(S, arg, H, r, q.active, stack, [&](double t, uint32_t sub, uint32_t tri) { q.hit(t, arg | (sub << ID_SUB_SHIFT), tri, lit); });
                break;
            }
            case OP_LEAF_PUSH: {
                const LeafHead H = leaf_head(S, arg);
                const uint32_t tag = arg | ((H.flags & LF_LIT) ? ID_LIT : 0u);
                leaf_hits<MESH>(S, arg, H, r, q.active, stack, [&](double t, uint32_t sub, uint32_t tri) { if (q.active) L.push(t, tag | (sub << ID_SUB_SHIFT), tri); });
                break;
            }
            case OP_CULL: {
                // Exact skip of a whole item: taken only when NO lane of the wave can produce a hit on it.
                const bool miss = item_missed(S, arg, r);
                bool need = q.active && !miss;
                if (ANY) need = need && !q.blocked;
                ++pc;
                if (!__any(need)) pc += S.program[pc];
                break;
            }
            case OP_CSG_PAIR: {
                // Csg.constructedSolid (Csg.fs:74-94) over two bare primitives with the hit lists in registers.  The
                // stable sort of [A hits; B hits] by t is the merge of the two (stably sorted) operands with ties to A.
                FT_COUNT(1, 1);
                const uint32_t leaf_a = S.program[pc + 1], wb = S.program[pc + 2];
                const uint32_t leaf_b = wb & ID_LEAF_MASK, cop = (wb >> 24) & 3u;
                const bool fold = ((wb >> 26) & 1u) != 0;
                pc += 2;
                struct Two { double t0, t1; uint32_t s0, s1; int n; };
                Two A{0.0, 0.0, 0u, 0u, 0}, Bh{0.0, 0.0, 0u, 0u, 0};
                const LeafHead HA = leaf_head(S, leaf_a);
                leaf_hits<false>(S, leaf_a, HA, r, q.active, stack, [&](double t, uint32_t sub, uint32_t) {
                    if (q.active) { if (A.n == 0) { A.t0 = t; A.s0 = sub; } else if (A.n == 1) { A.t1 = t; A.s1 = sub; } ++A.n; } });
                if ((cop == 1u || cop == 2u) && !__any(A.n > 0)) { pc += arg; break; }    // intersect / subtract with no A hit: empty (Csg.fs:27-44)
                const LeafHead HB = leaf_head(S, leaf_b);
                leaf_hits<false>(S, leaf_b, HB, r, q.active, stack, [&](double t, uint32_t sub, uint32_t) {
                    if (q.active) { if (Bh.n == 0) { Bh.t0 = t; Bh.s0 = sub; } else if (Bh.n == 1) { Bh.t1 = t; Bh.s1 = sub; } ++Bh.n; } });
                const bool odd = A.n > 2 || Bh.n > 2 || (A.n > 0 && A.t0 != A.t0) || (A.n > 1 && A.t1 != A.t1) || (Bh.n > 0 && Bh.t0 != Bh.t0) || (Bh.n > 1 && Bh.t1 != Bh.t1);
                if (__any(odd)) break;                             // rare (parallel-ray hits of Plane.fs:13-16, NaN): the generic sequence follows
                pc += arg;
                if (!__any(A.n + Bh.n > 0)) break;
                if (A.n == 2 && A.t1 < A.t0) { const double t = A.t0; A.t0 = A.t1; A.t1 = t; const uint32_t u = A.s0; A.s0 = A.s1; A.s1 = u; }
                if (Bh.n == 2 && Bh.t1 < Bh.t0) { const double t = Bh.t0; Bh.t0 = Bh.t1; Bh.t1 = t; const uint32_t u = Bh.s0; Bh.s0 = Bh.s1; Bh.s1 = u; }
                uint32_t take, flip;
                switch (cop) {                                      // rule tables as in csg_merge
                    case 0: take = 0xC3u; flip = 0x00u; break;
                    case 1: take = 0x3Cu; flip = 0x00u; break;
                    case 2: take = 0x41u; flip = 0x28u; break;
                    default: take = 0xC3u; flip = 0x3Cu; break;
                }
                const uint32_t tag_a = leaf_a | ((HA.flags & LF_LIT) ? ID_LIT : 0u), tag_b = leaf_b | ((HB.flags & LF_LIT) ? ID_LIT : 0u);
                if (fold) {
                    // The result goes straight into the query, which only asks for the smallest usable t (ties: the earlier hit of the merged
                    // sequence) or for any usable t: the walk along the merged sequence is not needed, only every hit's intersection type.
                    // When hit k of A is reached, A's own flag is (k == 1) and B's flag is the parity of the B hits in front of it - those
                    // with a strictly smaller t (equal t: A first, Csg.fs:78-79); when hit k of B is reached, B's own flag is (k == 1) and
                    // A's the parity of the A hits with t <= its own.  Same tables, same Take / Flip, and the four hits enter the query in
                    // an order (A's, then B's) that agrees with the merged one wherever t ties.
                    const bool a0 = A.n > 0, a1 = A.n > 1, b0 = Bh.n > 0, b1 = Bh.n > 1;
                    auto in_b_at = [&](double t) { return (b0 && Bh.t0 < t) != (b1 && Bh.t1 < t); };
                    auto in_a_at = [&](double t) { return (a0 && A.t0 <= t) != (a1 && A.t1 <= t); };
                    auto emit = [&](bool has, double t, uint32_t id0, uint32_t index) {
                        const uint32_t type = (0x53714620u >> (4 * index)) & 0xF;
                        if ((flip >> type) & 1u) id0 ^= ID_FLIP;
                        if (has && (((take | flip) >> type) & 1u)) q.hit(t, id0, 0u, (id0 & ID_LIT) != 0);
                    };
                    if (__any(a0)) emit(a0, A.t0, tag_a | (A.s0 << ID_SUB_SHIFT), 0u + (in_b_at(A.t0) ? 1u : 0u));
                    if (__any(a1)) emit(a1, A.t1, tag_a | (A.s1 << ID_SUB_SHIFT), 2u + (in_b_at(A.t1) ? 1u : 0u));
                    if (__any(b0)) emit(b0, Bh.t0, tag_b | (Bh.s0 << ID_SUB_SHIFT), 4u + (in_a_at(Bh.t0) ? 2u : 0u));
                    if (__any(b1)) emit(b1, Bh.t1, tag_b | (Bh.s1 << ID_SUB_SHIFT), 5u + (in_a_at(Bh.t1) ? 2u : 0u));
                    break;
                }
                int ia = 0, ib = 0;
                bool in_a = false, in_b = false;
#pragma unroll
                for (int step = 0; step < 4; ++step) {
                    const bool has_a = ia < A.n, has_b = ib < Bh.n;
                    if (!__any(has_a || has_b)) break;
                    const double ta = ia == 0 ? A.t0 : A.t1, tb = ib == 0 ? Bh.t0 : Bh.t1;
                    const bool side_b = has_b && (!has_a || tb < ta);
                    const double t = side_b ? tb : ta;
                    uint32_t id0 = side_b ? (tag_b | ((ib == 0 ? Bh.s0 : Bh.s1) << ID_SUB_SHIFT)) : (tag_a | ((ia == 0 ? A.s0 : A.s1) << ID_SUB_SHIFT));
                    const uint32_t type = (0x53714620u >> (4 * ((side_b ? 4 : 0) + (in_a ? 2 : 0) + (in_b ? 1 : 0)))) & 0xF;
                    if (has_a || has_b) {
                        if (side_b) { in_b = !in_b; ++ib; } else { in_a = !in_a; ++ia; }
                        if ((flip >> type) & 1u) id0 ^= ID_FLIP;
                        if (((take | flip) >> type) & 1u) {
                            if (fold) q.hit(t, id0, 0u, (id0 & ID_LIT) != 0);
                            else L.push(t, id0, 0u);
                        }
                    }
                }
                break;
            }
            case OP_MARK: L.mark(); break;
            case OP_CSG: csg_merge(L, arg); break;
            case OP_SKIP_IF_EMPTY: {
                const int seg_a = (int)(L.marks_lo & 0xFF);
                if (!__any(q.active && L.len > seg_a)) { L.pop_mark(); pc += arg; }
                break;
            }
            default: {                                             // OP_FOLD_LIST
                if (__any(L.len > 0)) {
                    for (int e = 0; e < L.len; ++e) {
                        const uint32_t id0 = L.id0_of(e);
                        q.hit(L.t_of(e), id0, L.id1_of(e), (id0 & ID_LIT) != 0);
                    }
                }
                L.len = 0;
                break;
            }
        }
        if (ANY) { if (__all(q.blocked || !q.active)) break; }    // every lane already in shadow
    }
    FT_CLK_ADD(4 * ((ANY ? 2 : 0) + (coherent ? 0 : 1)) + 1, clk1);
    overflow = L.overflow;
}

// ---------------------------------------------------------------------------------------------
// Surface point, normal and material of a closest hit (recomputed from t + hit identity; the
// reference computes them for every candidate hit, the values for the winner are the same).
struct Surface { V3 p, n; uint32_t material; double u, v; };   // (u,v) = TextureCoords of the hit (Ray.fs:20,26), filled when the material is textured

FT_DEV void cylinder_side(const Ray& r, double t, V3& p, V3& n) {   // Cylinder.fs:14-17
    p = {r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz};
    const V3 nn = normalise(V3{p.x, 0.0, p.z});
    n = (dot3(nn.x, nn.y, nn.z, r.dx, r.dy, r.dz) < 0.0) ? nn : V3{-nn.x, -nn.y, -nn.z};
}

template <bool TEXTURED>
FT_DEV Surface surface_at(const Scene& S, const Ray& rw, double t, uint32_t id0, uint32_t id1) {
    const uint32_t leaf = id0 & ID_LEAF_MASK, sub = (id0 >> ID_SUB_SHIFT) & ID_SUB_MASK;
    const LeafHead H = leaf_head(S, leaf);
    cdp M = S.leaves + 16ull * leaf;
    const bool xform = (H.flags & LF_XFORM) != 0;
    Ray r;
    to_model(M, xform, rw, r);
    V3 p, n;
    double tu = 0.0, tv = 0.0;                                     // newIntersection: uv = (0,0) (Ray.fs:29)
    const bool textured = TEXTURED && reinterpret_cast<cip>(S.materials + 8ull * H.material + 6)[1] >= 0;
    switch (H.kind) {
        case LK_SPHERE:
            p = {r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz}; n = normalise(p);
            if (textured) { tu = 0.5 + atan2(n.z, n.x) / (2.0 * 3.14159265358979323846); tv = 0.5 - asin(n.y) / 3.14159265358979323846; }   // Sphere.setUV (Sphere.fs:6-10)
            break;
        case LK_PLANE: case LK_SQUARE: case LK_CIRCLE:
            p = {r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz}; n = {0.0, 1.0, 0.0}; tu = p.x; tv = p.z;   // Plane.setUV (Plane.fs:28-30)
            break;
        case LK_CUBE: {
            const double qx = r.ox + 0.5, qy = r.oy + 0.5, qz = r.oz + 0.5;
            const double hx = qx + t * r.dx, hy = qy + t * r.dy, hz = qz + t * r.dz;
            tu = sub < 2 ? hx : sub < 4 ? hy : hx; tv = sub < 4 ? hz : hy;   // each face is a square: (x,z) of its own frame
            p = {(qx + t * r.dx) - 0.5, (qy + t * r.dy) - 0.5, (qz + t * r.dz) - 0.5};
            n = {sub == 2 ? -1.0 : sub == 3 ? 1.0 : 0.0, sub == 0 ? -1.0 : sub == 1 ? 1.0 : 0.0, sub == 4 ? -1.0 : sub == 5 ? 1.0 : 0.0};
            break;
        }
        case LK_CONE: {
            const double oy = r.oy - 1.0;
            const double qx = r.ox + t * r.dx, qy = oy + t * r.dy, qz = r.oz + t * r.dz;
            p = {qx, qy + 1.0, qz};
            const V3 nn = normalise(V3{qx, -qy, qz});
            n = (dot3(nn.x, nn.y, nn.z, r.dx, r.dy, r.dz) < 0.0) ? nn : V3{-nn.x, -nn.y, -nn.z};
            break;
        }
        case LK_CYLINDER: cylinder_side(r, t, p, n); break;
        case LK_SOLIDCYL: {
            if (sub >= 2) cylinder_side(r, t, p, n);
            else if (sub == 0) { const double oyt = r.oy - 1.0; p = {r.ox + t * r.dx, (oyt + t * r.dy) + 1.0, r.oz + t * r.dz}; n = {0.0, 1.0, 0.0}; tu = p.x; tv = p.z; }
            else { p = {r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz}; n = {0.0, -1.0, 0.0}; tu = -p.x; tv = p.z; }
            break;
        }
        default: {                                                 // triangle (Triangle.fs:63-64)
            cdp T = S.tris + 9ull * id1;
            const double len = sqrt(dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz));
            const V3 nd = normalise(V3{r.dx, r.dy, r.dz});
            const double k = t * len;
            p = {r.ox + nd.x * k, r.oy + nd.y * k, r.oz + nd.z * k};
            n = normalise(V3{T[4] * T[8] - T[5] * T[7], T[6] * T[5] - T[8] * T[3], T[3] * T[7] - T[4] * T[6]});   // edge1 .** edge2
            break;
        }
    }
    if (xform) {                                                   // Transform.fs:86: p <- modelToWorld*p, n <- normalise(normalToWorld*n)
        cdp W = S.m2w + 12ull * leaf;
        p = {W[0] * p.x + W[1] * p.y + W[2] * p.z + W[3], W[4] * p.x + W[5] * p.y + W[6] * p.z + W[7], W[8] * p.x + W[9] * p.y + W[10] * p.z + W[11]};
        n = normalise(V3{M[0] * n.x + M[4] * n.y + M[8] * n.z, M[1] * n.x + M[5] * n.y + M[9] * n.z, M[2] * n.x + M[6] * n.y + M[10] * n.z});   // (W2M^T) n
    }
    const bool flip = (((H.flags & LF_FLIP) != 0) != ((id0 & ID_FLIP) != 0));
    if (flip) n = {-n.x, -n.y, -n.z};
    return {p, n, H.material, tu, tv};
}

// Wave-cooperative grab of the next 64-item batch from a persistent work cursor.
// Texture.grid under its uv functions (Textures/Texture.fs:8-29), then the hueShift rotations that follow it.
FT_DEV void textured_colour(const Scene& S, const MaterialV& mat, double u, double v, double col[3]) {
    cdp T = S.textures + 48ull * (uint32_t)mat.texture;
    const int n_ops = (int)T[6];
    for (int k = 0; k < n_ops; ++k) {
        const double kind = T[9 + 3 * k], a = T[10 + 3 * k], b = T[11 + 3 * k];
        if (kind == 0.0) { u = u / a; v = v / b; }
        else { const double x = a * u + 0.0 * 0.0 + b * v, z = -b * u + 0.0 * 0.0 + a * v; u = x; v = z; }
    }
    const double ru = fabs(u - floor(u)), rv = fabs(v - floor(v));                 // Texture.repeat
    if (T[7] != 0.0) {                                                             // ImageTexture.image (Textures/Image.fs:27-35): nearest texel
        const double w = T[0], h = T[1];
        const double x = floor(ru * w), y = floor(rv * h);
        // index = y*(3*width) + 3*x as the reference computes it (x == width wraps into the next row).  Past the last
        // byte the reference raises IndexOutOfRange; this path reads the last texel instead (and texel 0 for NaN).
        double idx = y * (3.0 * w) + 3.0 * x;
        const double last = 3.0 * (w * h - 1.0);
        idx = idx >= 0.0 ? (idx <= last ? idx : last) : 0.0;
        const uint8_t* px = S.tex_pixels + (uint64_t)T[8] + (uint64_t)idx;
        col[0] = (double)px[0] / 255.0; col[1] = (double)px[1] / 255.0; col[2] = (double)px[2] / 255.0;
    } else {
        const bool first = (ru < 0.5 && rv < 0.5) ? true : (ru < 0.5) ? false : (ru > 0.5 && rv > 0.5);
        cdp c = first ? T : T + 3;
        col[0] = c[0]; col[1] = c[1]; col[2] = c[2];
    }
    for (uint32_t h = 0; h < mat.hue_rot; ++h) { const double r = col[0], g = col[1], b = col[2]; col[0] = b; col[1] = r; col[2] = g; }   // CommonTypes.fs:90
}

// Work distribution.  The batches of a launch (batch b = rays b*B .. b*B+B-1) are split into kWorkGroups interleaved
// classes (b mod 64); wave w pulls batches of class w mod 64 from that class's own cursor, one returning atomic per
// batch, issued one batch ahead so its latency hides behind the current batch.  Why 64 cursors and not one: a single
// device-scope word serves about 88 dequeues per microsecond (MI355X_MICROARCH.md row "dequeue"); with one shared
// cursor that capped every kernel at 5.6 Grays/s (one atomic per batch) and still cost 0.2 ms per 8.4 M-ray launch
// of an EMPTY scene with one atomic per 16 batches (all waves queue on the word together).  Purely static striding
// removes the atomics but loses 20-50 % on scenes whose batches differ widely in cost.  64 lines, about 80 waves
// each, keep every word far below its service rate and keep the balancing dynamic.
struct BatchCursor {
    uint32_t* ctr; uint32_t cls;
    FT_DEV BatchCursor(uint32_t* counters) {
#ifdef FT_AB_NO_UNIFORM_WAVE
        const uint32_t wave = blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
#else
        const uint32_t wave = blockIdx.x * (kBlock / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));   // uniform: the cursor's address stays scalar
#endif
        cls = wave % (uint32_t)kWorkGroups;
        ctr = counters + 16u * cls;
    }
    FT_DEV uint32_t grab() {                                        // index of the next batch of this wave's class
        uint32_t k = 0;
        if (lane_id() == 0) k = atomicAdd(ctr, 1u);
        return cls + (uint32_t)kWorkGroups * (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    }
};
//
// Rays per batch: a wave's cost grows with the number of DISTINCT scene items its rays touch, so when a launch has
// few rays (late bounces: a few hundred incoherent reflection rays), they are spread thinly — 32, 16, ... 1 per batch —
// over the otherwise idle SIMDs instead of packing 64 unrelated rays into one wave.
FT_DEV uint32_t batch_lanes_for(uint32_t n, int lane_fold, int n_simd) {
    // A batch costs a fixed part (the walk through the scene program: ~17 us of issue time on an otherwise idle SIMD for one
    // incoherent ray) plus ~1 us per further ray, and the SIMD, not the wave, is what that time is spent on.  So the rays are spread
    // only until every SIMD has a batch or two (measured on hollow-sphere x1: one to two batches per SIMD 0.89 ms, at most one 0.95,
    // two to four 0.95, one per WAVE - round 1's rule, a level of 15 K rays as 7.5 K two-ray batches - 0.98).
    uint32_t b = 64u / (uint32_t)lane_fold;                         // folded lanes lend their LDS columns to the live ones (HitList)
    while (b > 1u && n < b * (uint32_t)n_simd) b >>= 1;
    return b;
}

// Per-render statistics: a wave sums what it counts in registers and adds it, once per launch and per non-zero counter, to one of
// kStatStripes copies of the counters (no-return atomics, 128 waves per copy at most); the host sums the copies.  Atomics on ONE
// shared set of words at the end of every launch (thousands of waves finishing together) cost about 10 % of the bunny frame.
FT_DEV RenderCounters* my_stats(FrameCounters* fc) { return &fc->stats[(blockIdx.x * (kBlock / 64) + threadIdx.x / 64) % (uint32_t)kStatStripes]; }
FT_DEV void wave_add(unsigned long long* dst, unsigned long long v) { if (lane_id() == 0 && v) atomicAdd(dst, v); }
FT_DEV void wave_add(double* dst, double v) { if (lane_id() == 0 && v != 0.0) atomicAdd(dst, v); }   // sums of integers below 2^53: exact in any order

// ---------------------------------------------------------------------------------------------
// Seeded counter-based stream standing in for the reference's unseeded System.Random (Jitter.fs:27, Image.fs:101):
//   key = sm64(sm64(sm64(seed ^ sample) ^ (depth << 32 | light << 8 | purpose)));  u_n = (sm64(key + n) >> 11) * 2^-53
// (sm64 = splitmix64's output function; sample = pixel_id * spp + s; purpose 1 = soft shadow, 2 = depth of field).
// Every draw is a pure function of its key, so frames do not depend on tiling, chunking or traversal order.
FT_DEV unsigned long long sm64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    unsigned long long key, n;
    FT_DEV double next() { return (double)(sm64(key + n++) >> 11) * (1.0 / 9007199254740992.0); }
};
FT_DEV Rng make_rng(unsigned long long seed, unsigned long long sample, uint32_t depth, uint32_t light, uint32_t purpose) {
    return {sm64(sm64(sm64(seed ^ sample) ^ (((unsigned long long)depth << 32) | ((unsigned long long)light << 8) | purpose))), 0ull};
}
// Jitter.jitterVector (Jitter.fs:26-39): the orthonormal frame around `v`, then one direction per call of jittered().
struct JitterFrame {
    V3 nv, i, j; double m;
    FT_DEV JitterFrame(V3 v, double tan_half_angle) {
        nv = normalise(v); m = tan_half_angle;
        const V3 g = nv.x > 0.9 ? V3{0.0, 1.0, 0.0} : V3{1.0, 0.0, 0.0};
        i = normalise(V3{g.y * nv.z - g.z * nv.y, nv.x * g.z - nv.z * g.x, g.x * nv.y - g.y * nv.x});          // generator .** normalised
        j = V3{i.y * nv.z - i.z * nv.y, nv.x * i.z - nv.z * i.x, i.x * nv.y - i.y * nv.x};                     // i .** normalised
    }
    // A frame around a wave-uniform vector (a light's direction) is wave-uniform: into scalar registers with it (uniform_f64) - nine
    // doubles that otherwise sit in vector registers across every shadow trace of the light's samples.
    FT_DEV void make_uniform() {
        nv = {uniform_f64(nv.x), uniform_f64(nv.y), uniform_f64(nv.z)}; i = {uniform_f64(i.x), uniform_f64(i.y), uniform_f64(i.z)};
        j = {uniform_f64(j.x), uniform_f64(j.y), uniform_f64(j.z)}; m = uniform_f64(m);
    }
    FT_DEV V3 jittered(Rng& rng) const {
        double x, y;
        for (;;) { x = 2.0 * rng.next() - 1.0; y = 2.0 * rng.next() - 1.0; if ((x * x + y * y) > 1.0) continue; break; }   // Jitter.circle (Jitter.fs:15-21)
        const double a = m * x, b = m * y;
        return normalise(V3{(nv.x + a * i.x) + b * j.x, (nv.y + a * i.y) + b * j.y, (nv.z + a * i.z) + b * j.z});
    }
};
// a / b for the two divisors of a launch (pixels per chunk, row stride) through the divisor's reciprocal: exact, because
// (a + 0.5) / b is at least 0.5 / b away from every integer while the product's rounding error is below quotient * 2^-51
// and quotient * b < 2^32 (four FP64 instructions in place of the ~25 of an integer division).
FT_DEV uint32_t div_by(uint32_t a, double inv_b) { return (uint32_t)(((double)a + 0.5) * inv_b); }

// Pixels per sample plane of the chunk: the active count written by k_classify, or the host's when nothing was classified; and how
// the chunk's samples are numbered (see slot_at).
struct Pix { uint32_t n; double inv; uint32_t n_blocks; double inv_blocks; uint32_t group_log2; };
FT_DEV Pix pix_count(PrimaryArg g) {
    const PixCount* c = g->counts;
    uint32_t n = g->n_pix;
    if (c) {                                                        // this chunk's window [pix_base, pix_base + n_pix) of the frame's active list
        const uint32_t n_active = to_const_as(c)->n_pix, first = g->pix_base, cap = g->n_pix;
        n = n_active > first ? (n_active - first < cap ? n_active - first : cap) : 0u;
    }
    const uint32_t nb = n >> 6;                                     // one division per batch is cheaper than a launch that stores the reciprocals
    return {n, uniform_f64(1.0 / (double)n), nb, uniform_f64(1.0 / (double)(nb ? nb : 1u)), (n & 63u) ? 0u : (uint32_t)g->group_log2};   // grouped numbering needs whole blocks (k_resolve: the same rule)
}

// How a chunk's samples are numbered.  A slot is a sample's place in the colour planes (acc) and the unit a wavefront's lane takes.
// Plain numbering (group_log2 = 0): slot = s * n_pix + pixel, sample plane by sample plane, so 64 consecutive slots are one 8x8 pixel
// block under one jitter offset.  Grouped numbering (G = 2^group_log2 divides the sample count, the list is made of whole blocks):
// the G x 64 samples of one block under G consecutive offsets are dealt out the other way round - 64 consecutive slots are 64 / G of
// the block's pixels under all G offsets, a bundle 1 / G as wide on the image plane, which enters fewer BVH nodes and passes fewer
// cull tests per ray.  Either way a wavefront's 64 slots are 512 contiguous bytes of each colour plane.
//   batch b = slot / 64 = (s0 + k) * n_blocks + blk,  lane = slot % 64 = (s - s0) * (64 / G) + j
//   for sample s = s0 + (s - s0) with s0 a multiple of G, pixel = blk * 64 + k * (64 / G) + j of the list
struct SlotAt { uint32_t s, pl; };
FT_DEV SlotAt slot_at(const Pix& px, uint32_t batch, uint32_t lane) {          // grouped numbering only
    const uint32_t sb = div_by(batch, px.inv_blocks), blk = batch - sb * px.n_blocks, k = sb & ((1u << px.group_log2) - 1u), pw = 6u - px.group_log2;
    return {(sb - k) + (lane >> pw), (blk << 6) + (k << pw) + (lane & ((1u << pw) - 1u))};
}
FT_DEV SlotAt slot_at(const Pix& px, uint32_t slot) {
    if (px.group_log2 == 0u) { const uint32_t s = div_by(slot, px.inv); return {s, slot - s * px.n}; }
    return slot_at(px, slot >> 6, slot & 63u);
}

// Pixel id behind entry `at` of the chunk's list.  In a classified frame the list is the frame's ACTIVE list, kept as a map from
// its 64-pixel blocks to the blocks of the original pixel list (k_classify): the ids themselves are never copied.
FT_DEV uint32_t list_pixel(PrimaryArg g, uint32_t at) {
    const uint32_t* map = g->block_map;
    return g->pixel_ids[map ? map[at >> 6] * 64u + (at & 63u) : at];
}
FT_DEV unsigned long long sample_id(PrimaryArg g, const Pix& px, uint32_t slot) {
    const SlotAt at = slot_at(px, slot);
    const uint32_t pid = g->pixel_ids ? list_pixel(g, g->pix_base + at.pl) : at.pl;   // no pixel list: the slot is the sample (ft_debug_colour)
    return (unsigned long long)pid * (unsigned long long)g->spp + at.s;
}

// Primary rays are never stored: k_primary generates them from the sample's place (ImagePlane.rayThroughPixel, Image.fs:83-89).
FT_DEV uint32_t primary_pixel(PrimaryArg g, const SlotAt& at) { return list_pixel(g, g->pix_base + at.pl); }   // the one memory access a primary ray needs
FT_DEV Ray primary_ray_from(PrimaryArg g, uint32_t s, uint32_t pid) {
    const uint32_t py = div_by(pid, g->inv_stride), px = pid - py * g->stride;
    const double centre_x = g->cam.tlx + (double)px * g->cam.pw, centre_y = g->cam.tly - (double)py * g->cam.ph;
    const double ox = g->jitter[2 * s], oy = g->jitter[2 * s + 1];
    const double jx = centre_x + ox * g->cam.pw, jy = centre_y + oy * g->cam.ph;
    Ray r{g->cam.o[0], g->cam.o[1], g->cam.o[2],
          (g->cam.k[0] + jx * g->cam.i[0]) + jy * g->cam.j[0], (g->cam.k[1] + jx * g->cam.i[1]) + jy * g->cam.j[1], (g->cam.k[2] + jx * g->cam.i[2]) + jy * g->cam.j[2]};
    if (g->cam.has_focus) {                                         // ImagePlane.depthOfFieldJitter (Image.fs:91-94, Ray.fs:15-18)
        Rng rng = make_rng(g->seed, (unsigned long long)pid * (unsigned long long)g->spp + s, 0u, 0u, 2u);
        const double f = g->cam.focal_length;
        const V3 o1{r.ox + f * r.dx, r.oy + f * r.dy, r.oz + f * r.dz};                               // shiftOrigin focalLength
        const V3 d1 = JitterFrame(V3{r.dx, r.dy, r.dz}, g->cam.tan_half_aperture).jittered(rng);        // jitterDirection
        r = {o1.x + -f * d1.x, o1.y + -f * d1.y, o1.z + -f * d1.z, d1.x, d1.y, d1.z};                  // shiftOrigin -focalLength
    }
    return r;
}

// Kernel variants: FANCY = Oren-Nayar and grid textures compiled in (libm-heavy code: acos, tan, atan2 ...),
// SOFT = softdirectional lights, MESH = triangle meshes.  Scenes that lack a feature run a leaner kernel.
//
// Two passes over the lights keep the live state across the shadow traces small (p, n and a few words instead
// of the whole fragment state): pass 1 only decides visibility (one byte per light: occluded sample count),
// pass 2 reloads the ray and the material and evaluates the shaders.
// getLightsOnPoint (Shading.fs:109-117), visibility half: one shadow query per light (per sample of a soft light) from the
// surface point; byte l of (vis_lo, vis_hi) = number of occluded samples of light l.
template <bool SOFT, bool MESH, class SeedFn>
FT_DEV void light_visibility(const Scene& S, const Surface& sf, bool lit, unsigned long long sample, SeedFn&& seed_of, int bounce, bool coherent, uint32_t* lds,
                             unsigned long long& vis_lo, unsigned long long& vis_hi, unsigned long long& n_shadow_wave, unsigned long long& n_ovf_wave) {
    const int n_lights = S.n_lights;
    vis_lo = 0ull; vis_hi = 0ull;
    for (int l = 0; l < n_lights; ++l) {                       // wave-uniform; getLightsOnPoint (Shading.fs:109-117)
        cdp lp = S.lights + 12ull * (uint32_t)l;               // scalar loads
        const uint32_t kind = reinterpret_cast<cup>(lp + 10)[0];
        const double sox = sf.p.x + 0.0001 * sf.n.x, soy = sf.p.y + 0.0001 * sf.n.y, soz = sf.p.z + 0.0001 * sf.n.z;
        unsigned long long occluded = 0ull;
        bool overflow = false;
        if (SOFT && kind == LT_SOFT) {                         // softShadowLightIntensity (Shading.fs:24-31)
            const int samples = reinterpret_cast<cip>(lp + 10)[1];
            const JitterFrame frame(V3{-lp[0], -lp[1], -lp[2]}, lp[11]);   // (moved into scalar registers - make_uniform - the nine doubles cost MORE scratch: 176 -> 248 B in k_bounce<F,T,T>)
            Rng rng = make_rng(seed_of(), sample, (uint32_t)bounce, (uint32_t)l, 1u);
            for (int k = 0; k < samples; ++k) {                // wave-uniform count; each lane draws its own direction
                const V3 dj = frame.jittered(rng);
                Query<true> qs;
                qs.active = lit; qs.blocked = false; qs.best_t = 0; qs.id0 = 0; qs.id1 = 0; qs.max_dist = 1.7976931348623157e308;
                bool ovf = false;
                if (__any(lit)) trace<true, MESH>(S, Ray{sox, soy, soz, dj.x, dj.y, dj.z}, qs, lds, ovf, false);
                if (qs.blocked) ++occluded;
                overflow = overflow || ovf;
                n_shadow_wave += (unsigned long long)__popcll(__ballot(lit));
            }
        } else {
            Query<true> q;
            q.active = lit; q.blocked = false; q.best_t = 0; q.id0 = 0; q.id1 = 0;
            Ray sr;
            if (kind == LT_POINT) {                            // shadowLightIntensity (Shading.fs:33-42)
                const double ddx = lp[0] - sox, ddy = lp[1] - soy, ddz = lp[2] - soz;
                q.max_dist = sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
                const V3 dn = normalise(V3{ddx, ddy, ddz});
                sr = {sox, soy, soz, dn.x, dn.y, dn.z};
            } else {
                sr = {sox, soy, soz, -lp[0], -lp[1], -lp[2]};
                q.max_dist = 1.7976931348623157e308;           // System.Double.MaxValue
            }
            if (__any(lit)) trace<true, MESH>(S, sr, q, lds, overflow, coherent, kind == LT_POINT ? lp : nullptr);
            n_shadow_wave += (unsigned long long)__popcll(__ballot(lit));
            occluded = q.blocked ? 1ull : 0ull;
        }
        n_ovf_wave += (unsigned long long)__popcll(__ballot(overflow && lit));
        if (l < 8) vis_lo |= occluded << (8 * l); else vis_hi |= occluded << (8 * (l - 8));
    }
}

// The shaders of Shading.fs:50-107 over all lights for one hit: sum of specular + diffuse fragments (the reflection shader
// travels with the path weight).
template <bool FANCY, bool SOFT>
FT_DEV void shade_lights(const Scene& S, const Surface& sf, const MaterialV& mat, const Ray& r, bool active, bool lit,
                         unsigned long long vis_lo, unsigned long long vis_hi, double& cr, double& cg, double& cb) {
    const int n_lights = S.n_lights;
    cr = 0.0; cg = 0.0; cb = 0.0;                                  // sum over fragments (Seq.sumBy shader, Shading.fs:139)
    for (int l = 0; l < n_lights; ++l) {
        if (active && !lit) { cr += mat.colour[0]; cg += mat.colour[1]; cb += mat.colour[2]; }   // shadeIfRequired (Shading.fs:100-104)
        if (!lit) continue;
        cdp lp = S.lights + 12ull * (uint32_t)l;
        const uint32_t kind = reinterpret_cast<cup>(lp + 10)[0];
        const double occluded = (double)((l < 8 ? vis_lo >> (8 * l) : vis_hi >> (8 * (l - 8))) & 0xFFull);
        double intensity; V3 ld;
        if (kind == LT_POINT) {                                // Light.attenuate (Light.fs:16-17), lightDirection (Shading.fs:44-48)
            const double sox = sf.p.x + 0.0001 * sf.n.x, soy = sf.p.y + 0.0001 * sf.n.y, soz = sf.p.z + 0.0001 * sf.n.z;
            const double ddx = lp[0] - sox, ddy = lp[1] - soy, ddz = lp[2] - soz;
            const double dist = sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
            intensity = occluded != 0.0 ? 0.0 : 1.0 / (lp[3] + dist * (lp[4] + dist * lp[5]));
            ld = normalise(V3{sf.p.x - lp[0], sf.p.y - lp[1], sf.p.z - lp[2]});
        } else {
            if (SOFT && kind == LT_SOFT) { const double samples = (double)reinterpret_cast<cip>(lp + 10)[1]; intensity = (samples - occluded) / samples; }
            else intensity = occluded != 0.0 ? 0.0 : 1.0;
            ld = {lp[0], lp[1], lp[2]};
        }
        const double lcr = intensity * lp[6], lcg = intensity * lp[7], lcb = intensity * lp[8];   // scaleColour (Image.fs:25-26)
        double fr = 0.0, fg = 0.0, fb = 0.0;
        {                                                      // specularShader (Shading.fs:78-87)
            const V3 nn = normalise(sf.n);
            const double k2 = 2.0 * dot3(ld.x, ld.y, ld.z, nn.x, nn.y, nn.z);
            const V3 rl = normalise(V3{ld.x - k2 * nn.x, ld.y - k2 * nn.y, ld.z - k2 * nn.z});         // Vector.reflect (CommonTypes.fs:72)
            const V3 vd = normalise(V3{r.dx, r.dy, r.dz});
            // intensity = (view . -reflected) ** shineyness.  With shineyness <= 0 the shader is black whatever the power is, so
            // the power is only evaluated when some lane needs it; integral exponents up to 64 (the usual case) go through
            // square-and-multiply, everything else through pow.
            const bool wants = active && mat.shineyness > 0.0;
            double si = 0.0;
            if (__any(wants)) {                                   // each lane takes its own route: a ray's result must not depend on its wave
                const double base = dot3(vd.x, vd.y, vd.z, -rl.x, -rl.y, -rl.z);
                const bool small_int = mat.shineyness <= 64.0 && mat.shineyness == floor(mat.shineyness);
                if (__any(wants && small_int)) {
                    const uint32_t e = (wants && small_int) ? (uint32_t)mat.shineyness : 0u;
                    double b = base, pw = 1.0;
                    for (uint32_t bit = 0; __any((e >> bit) != 0u); ++bit) { if ((e >> bit) & 1u) pw *= b; b *= b; }
                    if (wants && small_int) si = pw;
                }
                // Math.Pow proper only exists in the FANCY variants (the host routes every scene with such an exponent there): inlined into the
                // lean kernels its seventeen polynomial constants were hoisted out of the batch loop and spilled - 136 bytes of scratch per lane,
                // stored by every wave of every launch before its first batch
#ifdef FT_AB_POW_ALL
                constexpr bool kPow = true;
#else
                constexpr bool kPow = FANCY;
#endif
                if (kPow) { if (__any(wants && !small_int)) { const double pw = pow(base, mat.shineyness); if (wants && !small_int) si = pw; } }
            }
            if (!(mat.shineyness <= 0.0 || si <= 0.0)) { fr = lcr * si; fg = lcg * si; fb = lcb * si; }
        }
        // reflectionShader is carried by the path weight (below)
        if (!FANCY || mat.roughness == 0.0) {                  // diffuseShader -> lambertianDiffuse (Shading.fs:65-76)
            const double di = dot3(-ld.x, -ld.y, -ld.z, sf.n.x, sf.n.y, sf.n.z);
            fr = fr + di * (mat.colour[0] * lcr); fg = fg + di * (mat.colour[1] * lcg); fb = fb + di * (mat.colour[2] * lcb);
        } else {                                               // roughDiffuse: Oren-Nayar (Shading.fs:50-63); the light colour is not used (sic)
            const double rough = mat.roughness * mat.roughness;
            const V3 nn = normalise(sf.n), nv = normalise(V3{-r.dx, -r.dy, -r.dz}), nl = normalise(V3{-ld.x, -ld.y, -ld.z});
            const double ray_angle = acos(dot3(nn.x, nn.y, nn.z, nv.x, nv.y, nv.z)), light_angle = acos(dot3(nn.x, nn.y, nn.z, nl.x, nl.y, nl.z));
            const double alpha = fs_max(ray_angle, light_angle), beta = fs_min(ray_angle, light_angle);
            const double A = 1.0 - 0.5 * rough / (rough + 0.33), B = 0.45 * rough / (rough + 0.09);
            const double kl = dot3(-ld.x, -ld.y, -ld.z, nn.x, nn.y, nn.z), kv = dot3(-r.dx, -r.dy, -r.dz, nn.x, nn.y, nn.z);
            const V3 tl = normalise(V3{-ld.x - kl * nn.x, -ld.y - kl * nn.y, -ld.z - kl * nn.z});   // perpendicularComponent (CommonTypes.fs:77-79)
            const V3 tr = normalise(V3{-r.dx - kv * nn.x, -r.dy - kv * nn.y, -r.dz - kv * nn.z});
            const double di = cos(light_angle) * (A + (B * fs_max(0.0, dot3(tl.x, tl.y, tl.z, tr.x, tr.y, tr.z)) * sin(alpha) * tan(beta)));
            fr = fr + di * mat.colour[0]; fg = fg + di * mat.colour[1]; fb = fb + di * mat.colour[2];
        }
        cr += fr; cg += fg; cb += fb;
    }
}

// ---------------------------------------------------------------------------------------------
// k_primary: bounce 0 as ONE kernel (the north star's fused megakernel, used where the work is coherent).  A wave takes a batch of
// 64 primary rays - one 8x8 pixel block for one jitter offset - generates them (Image.fs:83-89), finds their closest hits as a
// packet, and for the lanes that hit goes straight on to the shadow queries and the shaders (Shading.fs:109-139) with the ray,
// the hit and the surface still in registers: no hit record, no hit list, no second generation of the ray, and the block's shadow
// rays stay one tight bundle instead of being compacted with those of other blocks.  Every live lane stores its sample's colour
// (Colour.Zero for a miss: Scene.fs:116), so the accumulator planes are written in whole lines; reflection rays are spawned into
// the bounce-1 wavefront buffer for k_bounce.
struct PrimaryArgs {
    DevScene S; Primary gen; RayBuf next;
    double* acc; FrameCounters* fc;
    uint32_t acc_stride; int32_t max_depth;
};
// Resident workgroups per CU (measured on the config scenes, profiles/r02_a_fused_vs_split.txt: 2 -> 3 -> 4 is faster at every
// step although 4 leaves 128 registers per lane and the compiler parks cold state in scratch: the path waits on dependent scalar
// loads, and a fourth wave per SIMD covers more of that than the spill traffic costs).
// A fifth wave pays on the leanest variant only (no meshes, no soft lights; measured: night-house-det 3.60 -> 3.30 ms, the others even
// or worse), and only where the scene's LDS lets five workgroups live on a CU: hollow-sphere's hit lists allow four, and the
// five-workgroup build - fewer registers, more scratch - costs it 3 %.  So the lean variant exists twice and the host picks.
#ifndef FT_PRIMARY_BLOCKS
#define FT_PRIMARY_BLOCKS 4
#endif
#ifndef FT_LEAN_BLOCKS
#define FT_LEAN_BLOCKS 5
#endif
template <bool FANCY, bool SOFT, bool MESH, int BLOCKS>
__global__ __launch_bounds__(kBlock, BLOCKS) void k_primary(PrimaryArgs) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const FT_CONST PrimaryArgs* K = kernel_args<PrimaryArgs>();
    const Scene S = scene_view(K->S);
    ChunkCounters* cc = &K->fc->cc;
    const Pix px = pix_count(&K->gen);
    const uint32_t n_pix = px.n;
    const uint32_t n = n_pix * (uint32_t)K->gen.spp;
    const int n_lights = S.n_lights;
    unsigned long long n_shadow_wave = 0, n_refl_wave = 0, n_ovf_wave = 0, n_hit_wave = 0;
    // grouped numbering (slot_at): whole wavefronts; plain: few rays are spread over the SIMDs
    const uint32_t B = px.group_log2 ? 64u : batch_lanes_for(n, S.lane_fold, S.n_simd);
    const uint32_t n_batches = (n + B - 1) / B;
    const bool coherent = K->S.coherent_waves != 0;
    auto at_of = [&](uint32_t batch) -> SlotAt { return px.group_log2 ? slot_at(px, batch, lane_id()) : slot_at(px, batch * B + lane_id()); };
    BatchCursor cursor(&cc->work_trace[0][0]);
    uint32_t bi = cursor.grab(), bi_next = cursor.grab();
    uint32_t pid_next = 0;
    if (bi < n_batches && bi * B + lane_id() < n && lane_id() < B) pid_next = primary_pixel(&K->gen, at_of(bi));
    for (; bi < n_batches; bi = bi_next, bi_next = cursor.grab()) {
        const uint32_t i = bi * B + lane_id();
        const uint32_t pid = pid_next;
        const bool active = i < n && lane_id() < B;
        const unsigned long long clk_batch = FT_CLK_NOW();          // (diagnostic build: sections of a batch, slots 22 .. 27)
        // ---- closest hit; the geometry sees the offset ray (Shading.fs:135), the shaders the original one (Shading.fs:137)
        Ray ro{0, 0, 0, 0, 0, 0};
        {
            const FT_CONST PrimaryArgs* Kb = fresh(K);              // camera, pixel list: loaded here, dead before the trace
            if (bi_next < n_batches && bi_next * B + lane_id() < n && lane_id() < B) pid_next = primary_pixel(&Kb->gen, at_of(bi_next));
            if (active) {
                const Ray r = primary_ray_from(&Kb->gen, at_of(bi).s, pid);
                ro = {r.ox + 0.0001 * r.dx, r.oy + 0.0001 * r.dy, r.oz + 0.0001 * r.dz, r.dx, r.dy, r.dz};   // slightOffset (Shading.fs:129)
            }
        }
        Query<false> q;
        q.active = active; q.best_t = __builtin_inf(); q.id0 = ID_MISS; q.id1 = 0; q.max_dist = 0.0; q.blocked = false;
        bool overflow;
        const unsigned long long clk_a = FT_CLK_NOW();
        FT_CLK_ADD(22, clk_batch);
        trace<false, MESH>(S, ro, q, lds, overflow, coherent);
        FT_CLK_ADD(23, clk_a);
        n_ovf_wave += (unsigned long long)__popcll(__ballot(overflow && active));
        const bool hit = active && q.id0 != ID_MISS;
        const unsigned long long hit_mask = __ballot(hit);
        double cr = 0.0, cg = 0.0, cb = 0.0;                        // a sample whose primary ray hits nothing is Colour.Zero
        if (hit_mask) {
            // ---- shade
            Surface sf{{0, 0, 0}, {0, 1, 0}, 0, 0.0, 0.0};
            bool lit = false;
            unsigned long long sample = 0ull;
            if (hit) {
                sf = surface_at<FANCY>(S, ro, q.best_t, q.id0, q.id1);
                lit = reinterpret_cast<cup>(S.materials + 8ull * sf.material + 6)[0] != 0;
                if (SOFT) sample = (unsigned long long)pid * (unsigned long long)fresh(K)->gen.spp + at_of(bi).s;
            }
            unsigned long long vis_lo, vis_hi;
            const unsigned long long clk_b = FT_CLK_NOW();
            light_visibility<SOFT, MESH>(S, sf, lit, sample, [&]() { return fresh(K)->gen.seed; }, 0, coherent, lds, vis_lo, vis_hi, n_shadow_wave, n_ovf_wave);
            FT_CLK_ADD(24, clk_b);
            MaterialV mat = material_at(S, sf.material);
            if (FANCY) { if (hit && mat.texture >= 0) textured_colour(S, mat, sf.u, sf.v, mat.colour); }
            // the view ray is generated again here rather than kept in registers across the shadow traces (same arithmetic, same value; keeping only
            // its point on the image plane - two doubles - across them was measured too: scratch 48 -> 72 B/lane, the headline even, the CSG scenes 2.5 % slower)
            const FT_CONST PrimaryArgs* K2 = fresh(K);
            const Ray rv = hit ? primary_ray_from(&K2->gen, at_of(bi).s, pid) : Ray{0, 0, 0, 0, 0, 0};
            shade_lights<FANCY, SOFT>(S, sf, mat, rv, hit, lit, vis_lo, vis_hi, cr, cg, cb);
            // reflectionShader (Shading.fs:89-98), see k_bounce: one ray of weight L * reflectance stands for the L identical sub-traces
            const bool spawn = lit && mat.reflectance > 0.0 && 0 < K2->max_depth;
            const unsigned long long m = __ballot(spawn);
            const uint32_t cnt = (uint32_t)__popcll(m);
            uint32_t dst = 0;
            if (lane_id() == 0 && cnt) dst = atomicAdd(&K2->fc->cc.n_rays[1], cnt);
            dst = __builtin_amdgcn_readfirstlane(dst);
            if (spawn) {
                const uint32_t o = dst + lanes_below(m);
                const FT_CONST RayBuf& next = K2->next;
                const double k2 = 2.0 * dot3(rv.dx, rv.dy, rv.dz, sf.n.x, sf.n.y, sf.n.z);
                next.ox[o] = sf.p.x; next.oy[o] = sf.p.y; next.oz[o] = sf.p.z;
                next.dx[o] = rv.dx - k2 * sf.n.x; next.dy[o] = rv.dy - k2 * sf.n.y; next.dz[o] = rv.dz - k2 * sf.n.z;
                next.w[o] = 1.0 * (mat.reflectance * uniform_f64((double)n_lights));
                next.slot[o] = i;
            }
            n_refl_wave += cnt;
            n_hit_wave += (unsigned long long)__popcll(hit_mask);
        }
        if (active) {                                               // the sample's first contribution: a plain store (path weight 1)
            const FT_CONST PrimaryArgs* Ka = fresh(K);
            double* acc = Ka->acc; const uint32_t acc_stride = Ka->acc_stride;
            acc[i] = 1.0 * cr; acc[(size_t)acc_stride + i] = 1.0 * cg; acc[2 * (size_t)acc_stride + i] = 1.0 * cb;
        }
        FT_CLK_ADD(25, clk_batch); FT_CLK_INC(26);
    }
    RenderCounters* mine = my_stats(fresh(K)->fc);
    wave_add(&mine->rays_shadow, n_shadow_wave);
    wave_add(&mine->rays_shadow_primary, n_shadow_wave);
    wave_add(&mine->rays_reflect, n_refl_wave);
    wave_add(&mine->rays_reflect_primary, n_refl_wave);
    wave_add(&mine->hits_total, n_hit_wave);
    wave_add(&mine->hits_primary, n_hit_wave);
    wave_add(&mine->csg_overflow, n_ovf_wave);
    // what the F# recursion would trace (Shading.fs:109-139), depth 0
    wave_add(&mine->ref_equiv, uniform_f64((double)fresh(K)->S.shadow_rays_per_hit * (double)n_hit_wave + (double)n_lights * (double)n_refl_wave));
}

// ---------------------------------------------------------------------------------------------
// k_bounce: one level of the reflection tree (bounce k >= 1) as one kernel - the same fusion as k_primary, for rays that come out
// of the wavefront buffer instead of the camera: closest hit, shadow queries, shaders, accumulation into the sample's colour and
// the spawn of the level's reflection rays, compacted by wave ballot + prefix sum into the other half of the ping-pong buffer, so
// every lane of the next level is live.  The host launches one k_bounce per level, as many as the previous frame of the same
// scene had levels with rays, plus one (a launch that finds no rays returns at once; the last one launched FOLLOWS whatever it
// still spawns to the end in registers, so a frame that goes deeper than its predecessor is complete all the same); round 1 ran a closest / shade pair per large
// level and one path-following kernel for the rest, whose waves kept dragging a few live lanes through eight levels (measured on
// hollow-sphere x1: 0.76 of the frame's 1.05 ms).  Few rays are spread thinly over the grid (batch_lanes_for).
struct BounceArgs {
    DevScene S; Primary gen; RayBuf rays; RayBuf next;
    double* acc; FrameCounters* fc;
    uint32_t acc_stride; int32_t bounce, max_depth;
    int32_t follow;                                                // the last level launched: rays it spawns are followed to their end in registers, not queued
};
#ifndef FT_BOUNCE_BLOCKS
#define FT_BOUNCE_BLOCKS 4
#endif
template <bool FANCY, bool SOFT, bool MESH>
#ifndef FT_BOUNCE_LEAN
#define FT_BOUNCE_LEAN FT_BOUNCE_BLOCKS
#endif
__global__ __launch_bounds__(kBlock, FANCY ? 2 : (!SOFT && !MESH ? FT_BOUNCE_LEAN : FT_BOUNCE_BLOCKS)) void k_bounce(BounceArgs) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const FT_CONST BounceArgs* K = kernel_args<BounceArgs>();
    const int bounce = K->bounce;
    ChunkCounters* cc = &K->fc->cc;
    const uint32_t n = cc->n_rays[bounce];
    if (n == 0u) return;
    const Scene S = scene_view(K->S);
    const Pix px = pix_count(&K->gen);
    const int n_lights = S.n_lights;
    const bool follow = K->follow != 0;
    unsigned long long n_shadow_wave = 0, n_refl_wave = 0, n_ovf_wave = 0, n_hit_wave = 0;
    double ref_wave = 0.0;
    double mult0 = 1.0;                                             // copies of a ray of this level in the F# recursion (Shading.fs:109-139): L^bounce, exactly
    const double lights_f = uniform_f64((double)n_lights);         // wave-uniform doubles live in scalar registers (uniform_f64)
    for (int k = 0; k < bounce; ++k) mult0 *= lights_f;
    mult0 = uniform_f64(mult0);
    const uint32_t B = batch_lanes_for(n, S.lane_fold, S.n_simd);
    const uint32_t n_batches = (n + B - 1) / B;
    BatchCursor cursor(&cc->work_trace[bounce][0]);
    for (uint32_t bi = cursor.grab(), bi_next = cursor.grab(); bi < n_batches; bi = bi_next, bi_next = cursor.grab()) {
        const uint32_t i = bi * B + lane_id();
        bool alive = i < n && lane_id() < B;
        const unsigned long long clk_batch = FT_CLK_NOW();
        Ray r{0, 0, 0, 0, 0, 0};
        double w = 0.0; uint32_t slot = 0;
        if (alive) {
            const FT_CONST RayBuf& rays = fresh(K)->rays;           // buffers: loaded here, dead before the trace
            r = {rays.ox[i], rays.oy[i], rays.oz[i], rays.dx[i], rays.dy[i], rays.dz[i]}; w = rays.w[i]; slot = rays.slot[i];
        }
        double mult = mult0;
        for (int depth = bounce;; ++depth, mult = uniform_f64(mult * lights_f)) {   // one pass, unless this launch follows its rays to the end
            // (Measured in round 3: keeping the ray in memory across the level's traces - read for the closest hit, read again behind the
            //  shadow queries, a followed ray parked in the other buffer - frees fourteen registers on paper and changes the scratch of the
            //  variants by -20 .. +20 bytes either way: the spills come from inside the item evaluation, not from what lives around it.)
            // ---- closest hit; the geometry sees the offset ray (Shading.fs:135), the shaders the original one (Shading.fs:137)
            const Ray ro{r.ox + 0.0001 * r.dx, r.oy + 0.0001 * r.dy, r.oz + 0.0001 * r.dz, r.dx, r.dy, r.dz};   // slightOffset (Shading.fs:129)
            Query<false> q;
            q.active = alive; q.best_t = __builtin_inf(); q.id0 = ID_MISS; q.id1 = 0; q.max_dist = 0.0; q.blocked = false;
            bool overflow;
            const unsigned long long clk_a = FT_CLK_NOW();
            trace<false, MESH>(S, ro, q, lds, overflow, false);
            FT_CLK_ADD(16, clk_a);
            n_ovf_wave += (unsigned long long)__popcll(__ballot(overflow && alive));
            const bool hit = alive && q.id0 != ID_MISS;
            const unsigned long long hit_mask = __ballot(hit);
            if (hit_mask == 0ull) break;                            // a ray that hits nothing adds Colour.Zero
            // ---- shade
            Surface sf{{0, 0, 0}, {0, 1, 0}, 0, 0.0, 0.0};
            bool lit = false;
            unsigned long long sample = 0ull;
            if (hit) {
                sf = surface_at<FANCY>(S, ro, q.best_t, q.id0, q.id1);
                lit = reinterpret_cast<cup>(S.materials + 8ull * sf.material + 6)[0] != 0;
                if (SOFT) sample = sample_id(&fresh(K)->gen, px, slot);
            }
            unsigned long long vis_lo, vis_hi;
            FT_CLK_ADD(17, clk_a);
            const unsigned long long clk_b = FT_CLK_NOW();
            light_visibility<SOFT, MESH>(S, sf, lit, sample, [&]() { return fresh(K)->gen.seed; }, depth, false, lds, vis_lo, vis_hi, n_shadow_wave, n_ovf_wave);
            FT_CLK_ADD(18, clk_b);
            const unsigned long long clk_c = FT_CLK_NOW();
            MaterialV mat = material_at(S, sf.material);
            if (FANCY) { if (hit && mat.texture >= 0) textured_colour(S, mat, sf.u, sf.v, mat.colour); }
            double cr, cg, cb;
            shade_lights<FANCY, SOFT>(S, sf, mat, r, hit, lit, vis_lo, vis_hi, cr, cg, cb);
            const FT_CONST BounceArgs* K2 = fresh(K);
            if (hit) {                                              // one ray per sample per level: no write conflicts, fixed order
                double* acc = K2->acc; const uint32_t acc_stride = K2->acc_stride;
                acc[slot] += w * cr; acc[(size_t)acc_stride + slot] += w * cg; acc[2 * (size_t)acc_stride + slot] += w * cb;
            }
            // reflectionShader (Shading.fs:89-98): every one of the L fragments adds reflectance * colour(reflected ray); those L sub-traces
            // are identical (deterministic lights, or streams keyed without the parent light), so one ray carries weight L * reflectance.
            const bool spawn = lit && mat.reflectance > 0.0 && depth < K2->max_depth;
            const unsigned long long m = __ballot(spawn);
            const uint32_t cnt = (uint32_t)__popcll(m);
            n_refl_wave += cnt;
            n_hit_wave += (unsigned long long)__popcll(hit_mask);
            ref_wave = uniform_f64(ref_wave + mult * ((double)K2->S.shadow_rays_per_hit * (double)__popcll(hit_mask) + lights_f * (double)cnt));
            const double k2 = 2.0 * dot3(r.dx, r.dy, r.dz, sf.n.x, sf.n.y, sf.n.z);
            if (!follow) {                                          // the level's reflection rays, compacted into the other buffer
                uint32_t dst = 0;
                if (lane_id() == 0 && cnt) dst = atomicAdd(&K2->fc->cc.n_rays[depth + 1], cnt);
                dst = __builtin_amdgcn_readfirstlane(dst);
                if (spawn) {
                    const uint32_t o = dst + lanes_below(m);
                    const FT_CONST RayBuf& next = K2->next;
                    next.ox[o] = sf.p.x; next.oy[o] = sf.p.y; next.oz[o] = sf.p.z;
                    next.dx[o] = r.dx - k2 * sf.n.x; next.dy[o] = r.dy - k2 * sf.n.y; next.dz[o] = r.dz - k2 * sf.n.z;
                    next.w[o] = w * (mat.reflectance * lights_f);
                    next.slot[o] = slot;
                }
                FT_CLK_ADD(19, clk_c);
                break;
            }
            FT_CLK_ADD(19, clk_c);
            if (cnt == 0u) break;
            if (lane_id() == 0) atomicAdd(&K2->fc->cc.n_rays[depth + 1], cnt);   // counted all the same: the host sizes the next frame's launches from these
            if (spawn) {                                            // followed in registers: same ray, same weight as the queued one would carry
                r = {sf.p.x, sf.p.y, sf.p.z, r.dx - k2 * sf.n.x, r.dy - k2 * sf.n.y, r.dz - k2 * sf.n.z};
                w = w * (mat.reflectance * lights_f);
            }
            alive = spawn;
        }
        FT_CLK_ADD(20, clk_batch); FT_CLK_INC(21);
    }
    RenderCounters* mine = my_stats(fresh(K)->fc);
    wave_add(&mine->rays_shadow, n_shadow_wave);
    wave_add(&mine->rays_reflect, n_refl_wave);
    wave_add(&mine->hits_total, n_hit_wave);
    wave_add(&mine->csg_overflow, n_ovf_wave);
    wave_add(&mine->ref_equiv, ref_wave);                           // what the F# recursion would trace (Shading.fs:109-139)
}

// ---------------------------------------------------------------------------------------------
// k_classify: which 64-pixel blocks of the frame can see anything at all, and the list of those that can.  One LANE per block (an
// 8x8 tile of the pixel list), one workgroup per 256 consecutive blocks.  The rays of a block - ALL samples of its pixels: the caller's
// jitter offsets lie within +-extent pixels (the reference's in the unit disc, Jitter.fs:15-21) - have directions k + jx i + jy j
// with (jx, jy) in a rectangle; their angle to any axis inside that pyramid is largest at one of its four corners, so the cone
// around the centre ray through the four outermost corners bounds them all.  The cone is tested against the bounding sphere of
// every top-level item (cone_may_reach, conservative: explicit slack, items with a face direction some ray of the block may be
// parallel to are kept, Plane.fs:13-16), a bare mesh that survives also against the <= 64 boxes holding its triangles.  A block
// nothing can be hit from is finished: block_pos = -1, k_resolve writes its pixels as Colour.Zero (Scene.fs:116) and none of its
// 64 x spp rays is ever generated.  The other blocks are appended to the frame's active pixel list IN BLOCK ORDER (neighbouring
// blocks stay neighbours; later stages batch by list position; the list is a map from its blocks to those of the pixel list, no
// pixel id is copied: a wave copying the ids of 64 kept blocks one after the other took 35 us) without a second kernel: workgroups take their 256-block segment in
// ticket order, publish how many blocks they keep, and add up the counts of the tickets before theirs (all of them already running, so
// the wait is bounded by the slowest classification; it is also bounded by a poll limit that fails the frame rather than hang).
// The host only runs this for pinhole cameras, pixel lists made of 8x8 tiles and scenes made of bounded items.
struct ClassifyArgs {
    DevScene S; Primary gen;                                        // gen.pixel_ids / n_pix: the frame's full pixel list (8x8 tiles)
    ClassifyOut out; FrameCounters* fc;
    double jitter_extent;                                           // max(1, largest |offset| of the caller's jitter pattern), in pixels
    uint32_t epoch;                                                 // tags this frame's entries of out.wave_counts (never cleared)
};
constexpr uint32_t kClassifyPollLimit = 1u << 22;

constexpr uint32_t kClassifyBlock = 256;                            // threads = pixel blocks per workgroup
__global__ __launch_bounds__(kClassifyBlock) void k_classify(ClassifyArgs) {
    __shared__ uint32_t sh_ticket, sh_before, sh_wave_keep[kClassifyBlock / 64];
    const FT_CONST ClassifyArgs* K = kernel_args<ClassifyArgs>();
    const Scene S = scene_view(K->S);
    const PrimaryArg g = &K->gen;
    const uint32_t n_blocks = g->n_pix / 64u;
#ifdef FT_STAMPS
    // diagnostic build only: s_memrealtime (100 MHz) at the phases of every workgroup, into the words behind wave_counts[4096]
#define FT_STAMP(k) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(K->out.wave_counts + 4096)[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FT_STAMP(k) do { } while (0)
#endif
    FT_STAMP(0);
    if (threadIdx.x == 0) sh_ticket = atomicAdd(&K->fc->classify_ticket, 1u);
    __syncthreads();
    FT_STAMP(1);
    const uint32_t ticket = sh_ticket, wave = threadIdx.x / 64u;
    const uint32_t blk = ticket * kClassifyBlock + threadIdx.x;
    const bool valid = blk < n_blocks;
    // Every lane runs the whole classification (lanes past the end of the list redo the last block): the box test below hands
    // lane b the role of BOX b, so no lane may sit out.
    const uint32_t lb = valid ? blk : n_blocks - 1u;
    // the tile's first and last pixel are its top-left and bottom-right corners
    const uint32_t pid0 = g->pixel_ids[(size_t)lb * 64u], pid1 = g->pixel_ids[(size_t)lb * 64u + 63u];
    const uint32_t y0 = div_by(pid0, g->inv_stride), x0 = pid0 - y0 * g->stride, y1 = div_by(pid1, g->inv_stride), x1 = pid1 - y1 * g->stride;
    if (__builtin_amdgcn_readfirstlane((int)(x0 + y0 + x1 + y1)) == -12345) return;   // (keeps the loads ahead of the stamp in the diagnostic build; never true)
    FT_STAMP(2);
    const double ext = K->jitter_extent * 1.000001;
    const double jxa = g->cam.tlx + ((double)x0 - ext) * g->cam.pw, jxb = g->cam.tlx + ((double)x1 + ext) * g->cam.pw;
    const double jya = g->cam.tly - ((double)y1 + ext) * g->cam.ph, jyb = g->cam.tly - ((double)y0 - ext) * g->cam.ph;
    double d[5][3];                                                 // centre, then the corners in order around the block
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        const double jx = c == 0 ? 0.5 * (jxa + jxb) : ((c == 1 || c == 4) ? jxa : jxb), jy = c == 0 ? 0.5 * (jya + jyb) : (c <= 2 ? jya : jyb);
        d[c][0] = (g->cam.k[0] + jx * g->cam.i[0]) + jy * g->cam.j[0]; d[c][1] = (g->cam.k[1] + jx * g->cam.i[1]) + jy * g->cam.j[1];
        d[c][2] = (g->cam.k[2] + jx * g->cam.i[2]) + jy * g->cam.j[2];
    }
    float ax = 0.f, ay = 0.f, az = 0.f, cos_dev = 1.0f;
    bool finite = true;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        float fx = (float)d[c][0], fy = (float)d[c][1], fz = (float)d[c][2];
        const float l2 = fx * fx + fy * fy + fz * fz;
        finite = finite && l2 > 1e-30f && l2 < 1e30f;
        const float inv = __builtin_amdgcn_rsqf(l2);
        fx *= inv; fy *= inv; fz *= inv;
        if (c == 0) { ax = fx; ay = fy; az = fz; } else cos_dev = fminf(cos_dev, ax * fx + ay * fy + az * fz);
    }
    const float cos_t = cos_dev - 1e-5f;                            // cos of the half-angle, made smaller (cone wider)
    const bool bounded = finite && cos_t > 0.3f;                    // a wide or degenerate bundle bounds nothing: the block stays
    // face directions some ray of the block may be nearly parallel to (Plane.fs:13-16): row . d is affine in (jx, jy), so over the
    // block's rectangle it lies between its values at the four corners
    uint32_t par_rows = 0;
    for (int k = 0; k < S.n_cull_rows; ++k) {
        cdp Rw = S.cull_rows + 3u * (uint32_t)k;
        double lo = __builtin_inf(), hi = -__builtin_inf();
#pragma unroll
        for (int c = 1; c < 5; ++c) { const double v = dot3(Rw[0], Rw[1], Rw[2], d[c][0], d[c][1], d[c][2]); lo = fmin(lo, v); hi = fmax(hi, v); }
        if (lo < 2.000001 * kEps && hi > -2.000001 * kEps) par_rows |= 1u << k;
    }
    const float cox = (float)g->cam.o[0], coy = (float)g->cam.o[1], coz = (float)g->cam.o[2];
    const float origin_mag = fabsf(cox) + fabsf(coy) + fabsf(coz);
    const Cone B{ax, ay, az, cox, coy, coz, cos_t, sqrtf(fmaxf(0.0f, 1.0f - cos_t * cos_t)) + 1e-5f, 1e-5f * (1.0f + origin_mag), par_rows, __builtin_inff()};
    bool keep = !bounded;
    for (int item = 0; item < S.n_items; ++item) {                  // wave-uniform: the item record comes through scalar loads
        if (!__any(!keep)) break;
        const FT_CONST float* I = to_const_as(S.cull_items) + 8u * (uint32_t)item;
        bool reach = !keep && cone_may_reach(I[0], I[1], I[2], I[3], __float_as_uint(I[4]), B, origin_mag);
        const uint32_t n_box = __float_as_uint(I[6]);
        if (n_box != 0u && __any(reach)) {
            // A bare mesh that survived its bounding sphere: the pyramid through the block's corners, taken into the mesh's model space,
            // must reach one of its coarse boxes (a box wholly behind one side plane of the pyramid is out of reach).
            const uint32_t first_box = __float_as_uint(I[5]), leaf = __float_as_uint(I[7]);
            const LeafHead Hm = leaf_head(S, leaf);
            cdp Mw = S.leaves + 16ull * leaf;
            Ray corner[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) to_model(Mw, (Hm.flags & LF_XFORM) != 0, Ray{g->cam.o[0], g->cam.o[1], g->cam.o[2], d[c + 1][0], d[c + 1][1], d[c + 1][2]}, corner[c]);
            float pn[4][3]; uint32_t sided = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {                           // side plane through corner rays c and c+1, oriented by the opposite corner
                const Ray& a = corner[c]; const Ray& bq = corner[(c + 1) & 3]; const Ray& opp = corner[(c + 2) & 3];
                float nx = (float)(a.dy * bq.dz - a.dz * bq.dy), ny = (float)(a.dz * bq.dx - a.dx * bq.dz), nz = (float)(a.dx * bq.dy - a.dy * bq.dx);
                const float side = nx * (float)opp.dx + ny * (float)opp.dy + nz * (float)opp.dz;
                if (side < 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
                pn[c][0] = nx; pn[c][1] = ny; pn[c][2] = nz; if (side != 0.0f) sided |= 1u << c;
            }
            const float ox = (float)corner[0].ox, oy = (float)corner[0].oy, oz = (float)corner[0].oz;   // the camera in model space: the same for every block
            // The boxes sit in the lanes (lane k: box k of the current 64) and are broadcast one at a time; every lane tests ITS block.
            // Groups of eight neighbouring boxes (siblings in the tree they were cut from) are tried first as one box: a block whose
            // pyramid misses the union of a group skips its eight members, and the walk ends as soon as every block has found a box
            // (measured: one block at a time against 64 boxes in the lanes took 25 us for a wave of blocks inside the bounding sphere).
            auto may_reach = [&](const float (&q)[6]) {
                bool in = true;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float nx = pn[c][0], ny = pn[c][1], nz = pn[c][2];
                    // the box corner furthest along the inward normal
                    const float qx = (nx > 0.0f ? q[3] : q[0]) - ox, qy = (ny > 0.0f ? q[4] : q[1]) - oy, qz = (nz > 0.0f ? q[5] : q[2]) - oz;
                    const float dd = nx * qx + ny * qy + nz * qz;
                    const float slack = 1e-4f * (fabsf(nx * qx) + fabsf(ny * qy) + fabsf(nz * qz));
                    if (dd < -slack && ((sided >> c) & 1u)) in = false;     // wholly outside this side of the pyramid
                }
                return in;
            };
            bool found = false;
            for (uint32_t b0 = 0; b0 < n_box && __any(reach && !found); b0 += 64u) {   // all 64 lanes are executing here (no enclosing per-lane branch)
                const bool has_box = b0 + lane_id() < n_box;
                float bx[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};           // no box: an empty one
                if (has_box) { const float* Bx = S.coarse_boxes + 6u * (first_box + b0 + lane_id()); for (int k = 0; k < 6; ++k) bx[k] = Bx[k]; }
                float gx[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {                        // union over the lane's group of eight
                    float v = bx[k];
                    for (int off = 1; off < 8; off <<= 1) { const float o2 = __shfl_xor(v, off); v = k < 3 ? fminf(v, o2) : fmaxf(v, o2); }
                    gx[k] = v;
                }
                const uint32_t n_here = n_box - b0 < 64u ? n_box - b0 : 64u;
                for (uint32_t g0 = 0; g0 < n_here; g0 += 8u) {       // wave-uniform
                    float q[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) q[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx[k]), (int)g0));
                    const bool gin = reach && !found && may_reach(q);
                    if (!__any(gin)) continue;
                    const uint32_t g1 = g0 + 8u < n_here ? g0 + 8u : n_here;
                    for (uint32_t bb = g0; bb < g1; ++bb) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) q[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx[k]), (int)bb));
                        if (gin && !found && may_reach(q)) found = true;
                        if (!__any(gin && !found)) break;
                    }
                    if (!__any(reach && !found)) break;
                }
            }
            reach = reach && found;
        }
        keep = keep || reach;
    }
    keep = keep && valid;
    FT_STAMP(3);
    // ---- compaction in block order
    const FT_CONST ClassifyOut& out = K->out;
    const unsigned long long km = __ballot(keep);
    const uint32_t n_keep = (uint32_t)__popcll(km);
    if (lane_id() == 0) sh_wave_keep[wave] = n_keep;
    __syncthreads();
    uint32_t n_keep_wg = 0, before_wave = 0;
    for (uint32_t w = 0; w < kClassifyBlock / 64; ++w) { if (w < wave) before_wave += sh_wave_keep[w]; n_keep_wg += sh_wave_keep[w]; }
    const uint32_t epoch = K->epoch;
    // Only the word itself travels between workgroups (no data is published behind it), so relaxed device-scope atomics are enough: an
    // acquire in the poll loop would invalidate the XCD's L2 on every poll, for every wave on it (measured: 238 us instead of 45).
    // One workgroup per 256 blocks keeps the words few: every reader loads all the words before its own, and a device-scope line
    // serves a few hundred loads per microsecond (one WAVE per 64 blocks: 128 K loads on 16 lines, 35 us of a 45 us kernel).
    if (threadIdx.x == 0) __hip_atomic_store(&out.wave_counts[ticket], (epoch << 10) | (n_keep_wg + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
        uint32_t before = 0;
        bool timed_out = false;
        for (uint32_t j0 = 0; j0 < ticket; j0 += 64u) {             // kept blocks of every earlier ticket (they are all running: tickets are taken at workgroup start)
            const uint32_t j = j0 + lane_id();
            uint32_t v = (epoch << 10) | 1u;
            if (j < ticket) {
                uint32_t polls = 0;
                for (;;) {
                    v = __hip_atomic_load(&out.wave_counts[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 10) == epoch || ++polls > kClassifyPollLimit) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if ((v >> 10) != epoch) { timed_out = true; v = (epoch << 10) | 1u; }
            }
            uint32_t c = (v & 0x3FFu) - 1u;
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
            before += c;
        }
        if (__any(timed_out) && lane_id() == 0) atomicOr(&K->fc->classify_error, 1u);
        if (lane_id() == 0) sh_before = before;
    }
    __syncthreads();
    FT_STAMP(4);
    const uint32_t before = sh_before + before_wave;                // kept blocks before this wave's first
    const uint32_t pos = before + lanes_below(km);
    if (valid) out.block_pos[blk] = keep ? (int32_t)pos : -1;
    if (keep) out.pos_block[pos] = blk;                             // the active list as a block map: pixel ids stay where they are
    if (threadIdx.x == 0 && (ticket + 1u) * kClassifyBlock >= n_blocks && ticket * kClassifyBlock < n_blocks)
        K->fc->counts.n_pix = 64u * (sh_before + n_keep_wg);        // the last segment publishes the length of the list
    const unsigned long long n_valid = (unsigned long long)__popcll(__ballot(valid));
    wave_add(&K->fc->stats[(ticket * (kClassifyBlock / 64) + wave) % (uint32_t)kStatStripes].pixels_culled, 64ull * (n_valid - (unsigned long long)n_keep));
    FT_STAMP(5);
}

// Image.write's toByte (Image.fs:36; Math.clamp, Math.fs:12-16): clamp to [0, 1] (NaN passes the clamp), * 255, truncate.
FT_DEV uint32_t to_byte(double x) {
    if (x > 1.0) x = 1.0; else if (x < 0.0) x = 0.0;
    x = x * 255.0;
    return (x != x) ? 0u : (uint32_t)(uint8_t)x;
}
FT_DEV void write_pixel(double* out_rgb, uint8_t* out_rgba, size_t o, double r, double g, double b) {
    if (out_rgb) { out_rgb[3 * o] = r; out_rgb[3 * o + 1] = g; out_rgb[3 * o + 2] = b; }
    if (out_rgba) reinterpret_cast<uint32_t*>(out_rgba)[o] = to_byte(r) | (to_byte(g) << 8) | (to_byte(b) << 16) | 0xFF000000u;   // R, G, B, A = 255 in memory order
}

// k_resolve: every pixel of the chunk's window of the active list gets the mean of its samples, summed from Zero in sample order and
// divided once (JitteredSampling.blendPixels, Image.fs:112-116: Array.average = sum, then DivideByInt, CommonTypes.fs:43-48); with
// block_pos the launch also writes Colour.Zero for the pixels of every block k_classify finished.  Each pixel of the frame is
// written once, as FP64 RGB and / or as RGBA8 bytes.
// The end of a frame: the last workgroup to get here copies what the host wants of the counters into the pinned report and clears
// the counters for the next frame.  Every other workgroup has finished with them (the ticket is taken after a workgroup's last
// access) and the kernels that wrote them ended before this one began.
FT_DEV void hand_over_frame(FrameCounters* fc, FrameReport* report, unsigned long long* cells) {   // cells: 8 KB of LDS
    __shared__ uint32_t last;
    const uint32_t t = threadIdx.x;
    __syncthreads();
    // no fence: what is handed over was written by earlier kernels, and this workgroup's own reads of the counters returned long ago
    // (a __threadfence here is an L2 write-back per workgroup: it made k_resolve take 314 us instead of 18)
    if (t == 0) {
        const uint32_t stripe = blockIdx.x & 63u, in_stripe = (gridDim.x - stripe + 63u) / 64u, n_stripes = gridDim.x < 64u ? gridDim.x : 64u;
        last = 0u;
        if (atomicAdd(&fc->report_stripe[stripe], 1u) == in_stripe - 1u) last = atomicAdd(&fc->report_ticket, 1u) == n_stripes - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    // The one workgroup that goes on reads what every other workgroup of the frame wrote (the stripes: no-return atomics, performed at the
    // L2 / memory side) and clears counters other workgroups of THIS launch read at their start.  Invariant: every read a workgroup makes of
    // `fc` precedes its ticket in program order and its value was consumed (loop bounds, the window) before the ticket was taken, so by the
    // time the last ticket is out no read of `fc` is outstanding anywhere.  The tickets stay relaxed (a fence per workgroup is an L2
    // write-back each: 314 us instead of 18); one agent-scope acquire here, in the one workgroup that reads and clears, makes the order
    // part of the memory model instead of a property of gfx950's in-order return of consumed loads.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    static_assert(sizeof(RenderCounters) == 128 && kStatStripes * 16 == 4 * kBlock, "the stripes are read as 4 x 256 eight-byte cells");
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&fc->stats[0]);
    for (uint32_t k = 0; k < 4; ++k) cells[k * kBlock + t] = src[k * kBlock + t];           // all loads in flight at once
    unsigned long long* rep = cells;                                // the report is assembled over the cells once they are summed
    __syncthreads();
    unsigned long long sum = 0ull;
    if (t < 16) {                                                   // one 8-byte field of RenderCounters per thread, summed over the stripes in order
        if (t == 4) { double s = 0.0; for (int k = 0; k < kStatStripes; ++k) s += __longlong_as_double((long long)cells[16 * k + 4]); sum = (unsigned long long)__double_as_longlong(s); }
        else for (int k = 0; k < kStatStripes; ++k) sum += cells[16 * k + t];
    }
    uint32_t word = 0u;
    if (t >= 64 && t < 64 + kMaxBounce + 2) word = fc->cc.n_rays[t - 64];
    else if (t == 128) word = fc->counts.n_pix;
    else if (t == 129) word = fc->classify_error;
    __syncthreads();
    uint32_t* words = reinterpret_cast<uint32_t*>(rep);
    if (t < 16) rep[t] = sum;
    else if (t >= 64 && t < 64 + kMaxBounce + 2) words[32 + (t - 64)] = word;
    else if (t == 128 || t == 129) words[32 + kMaxBounce + 2 + (t - 128)] = word;
    else if (t >= 130 && t < 142) words[34 + kMaxBounce + 2 + (t - 130)] = 0u;
    __syncthreads();
    if (t < 64) reinterpret_cast<uint32_t*>(report)[t] = words[t];
    __syncthreads();
    uint4* z = reinterpret_cast<uint4*>(fc);
    for (uint32_t w = threadIdx.x; w < sizeof(FrameCounters) / 16; w += kBlock) z[w] = uint4{0u, 0u, 0u, 0u};
}
__global__ __launch_bounds__(kBlock) void k_report(FrameCounters* fc, FrameReport* report) {
    __shared__ unsigned long long cells[kStatStripes * 16];
    hand_over_frame(fc, report, cells);
}

// k_resolve under grouped numbering (slot_at): the G samples of a pixel under consecutive offsets lie 64 / G doubles apart in ONE
// 512-byte run shared with its 64 / G - 1 neighbours, and the 64 pixels of a block in G such runs.  A wave reads the G runs of the three
// colour planes whole (3 G coalesced loads in flight per lane), passes them through LDS plane by plane, and every lane then sums its own
// pixel's samples in sample order: each colour is fetched once, in full lines (reading them pixel by pixel touches four times as many
// lines per instruction).  The tile is the wave's own: the LDS keeps one wave's accesses in order, so a compiler barrier is all that
// stands between the stores and the loads of other lanes' words.
template <int GL, class Emit>
FT_DEV void resolve_grouped(const ResolveArgs& a, uint32_t n_pix, double* T, Emit&& emit) {
    constexpr uint32_t G = 1u << GL, pw = 6u - GL, ppw = 1u << pw, row = 64u + ppw;   // rows padded so that the 64 / G-lane groups fall on different banks
    const uint32_t nb = n_pix >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, waves_total = gridDim.x * (kBlock / 64);
    const uint32_t mine = (lane >> pw) * row + (lane & (ppw - 1u));
    auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    for (uint32_t blk = blockIdx.x * (kBlock / 64) + wave; blk < nb; blk += waves_total) {
        double sum[3] = {0.0, 0.0, 0.0};
        for (uint32_t s0 = 0; s0 < (uint32_t)a.spp; s0 += G) {
            double v[3][G];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (uint32_t k = 0; k < G; ++k) v[p][k] = a.acc[(size_t)p * a.acc_stride + ((((size_t)(s0 + k)) * nb + blk) << 6) + lane];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (uint32_t k = 0; k < G; ++k) T[k * row + lane] = v[p][k];
                wave_sync();
#pragma unroll
                for (uint32_t ds = 0; ds < G; ++ds) sum[p] += T[mine + (ds << pw)];
                wave_sync();
            }
        }
        emit((blk << 6) + lane, sum[0], sum[1], sum[2]);
    }
}

__global__ __launch_bounds__(kBlock) void k_resolve(ResolveArgs a) {
    __shared__ double tile[(kBlock / 64) * 8 * 136];                // per wave 8 x 136 doubles >= G rows of 64 + 64 / G for G = 2 .. 16 (and 8 KB for the hand-over)
    uint32_t n_pix = a.n_pix_host;                                  // all pixels of the chunk, or its window of the frame's active list (k_classify)
    if (a.counts) { const uint32_t n_active = a.counts->n_pix; n_pix = n_active > a.first ? (n_active - a.first < a.n_pix_host ? n_active - a.first : a.n_pix_host) : 0u; }
    const double spp = (double)a.spp;
    const uint32_t group_log2 = (n_pix & 63u) ? 0u : a.group_log2;    // as pix_count decides it
    auto emit = [&](uint32_t q, double r, double g, double b) {
        const uint32_t al = a.first + q;                            // position in the active list -> position in the original pixel list
        const uint32_t p = a.pos_block ? a.pos_block[al >> 6] * 64u + (al & 63u) : al;
        write_pixel(a.out_rgb, a.out_rgba, a.pixel_ids ? (size_t)a.pixel_ids[p] : (size_t)p, r / spp, g / spp, b / spp);
    };
    if (group_log2 == 0u) {
        for (uint32_t q = blockIdx.x * kBlock + threadIdx.x; q < n_pix; q += gridDim.x * kBlock) {
            double r = 0.0, g = 0.0, b = 0.0;
            for (int s = 0; s < a.spp; ++s) {
                const size_t i = (size_t)s * n_pix + q;
                r += a.acc[i]; g += a.acc[(size_t)a.acc_stride + i]; b += a.acc[2 * (size_t)a.acc_stride + i];
            }
            emit(q, r, g, b);
        }
    } else {
        double* T = tile + (threadIdx.x >> 6) * (8 * 136);
        switch (group_log2) {
            case 1: resolve_grouped<1>(a, n_pix, T, emit); break;
            case 2: resolve_grouped<2>(a, n_pix, T, emit); break;
            case 3: resolve_grouped<3>(a, n_pix, T, emit); break;
            default: resolve_grouped<4>(a, n_pix, T, emit); break;
        }
    }
    __syncthreads();                                                // the tile serves the hand-over next
    if (a.block_pos) {                                              // one wave per finished block: Colour.Zero for its 64 pixels
        const uint32_t lane = threadIdx.x & 63u;
        for (uint32_t blk = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; blk < a.n_blocks_total; blk += gridDim.x * (kBlock / 64)) {
            if (a.block_pos[blk] >= 0) continue;
            const uint32_t p = blk * 64u + lane;
            write_pixel(a.out_rgb, a.out_rgba, a.pixel_ids ? (size_t)a.pixel_ids[p] : (size_t)p, 0.0, 0.0, 0.0);
        }
    }
    if (a.report) hand_over_frame(a.fc, a.report, reinterpret_cast<unsigned long long*>(tile));
}

__global__ __launch_bounds__(kBlock) void k_resolve_corner(const double* __restrict__ acc, uint32_t acc_stride, uint32_t w, uint32_t h,
                                                            const uint32_t* __restrict__ out_index, double* __restrict__ out_rgb, uint8_t* __restrict__ out_rgba) {
    const uint32_t n = w * h, cs = w + 1;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n; p += gridDim.x * kBlock) {
        const uint32_t y = p / w, x = p - y * w;
        const uint32_t c[4] = {y * cs + x, y * cs + x + 1, (y + 1) * cs + x, (y + 1) * cs + x + 1};   // Image.fs:139
        double r = 0.0, g = 0.0, b = 0.0;                          // Seq.average: sum in corner order, / 4
        for (int k = 0; k < 4; ++k) { r += acc[c[k]]; g += acc[(size_t)acc_stride + c[k]]; b += acc[2 * (size_t)acc_stride + c[k]]; }
        write_pixel(out_rgb, out_rgba, out_index ? (size_t)out_index[p] : (size_t)p, r / 4.0, g / 4.0, b / 4.0);
    }
}

__global__ __launch_bounds__(kBlock) void k_debug_closest(DevScene Sg, const double* __restrict__ o, const double* __restrict__ d, uint32_t n,
                                                           int32_t* hit, double* t, double* p, double* nrm, double* colour, unsigned long long* overflow_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Scene S = scene_view(Sg);
    const uint32_t B = 64u / (uint32_t)S.lane_fold, n_batches = (n + B - 1) / B;
    for (uint32_t b = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; b < n_batches; b += gridDim.x * (kBlock / 64)) {
        const uint32_t i = b * B + lane_id();
        Query<false> q;
        q.active = i < n && lane_id() < B; q.best_t = __builtin_inf(); q.id0 = ID_MISS; q.id1 = 0; q.max_dist = 0; q.blocked = false;
        Ray r{0, 0, 0, 0, 0, 0};
        if (q.active) r = {o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]};
        bool overflow;
        trace<false, true>(S, r, q, lds, overflow);
        if (q.active) {
            const bool h = q.id0 != ID_MISS;
            hit[i] = h ? 1 : 0;
            Surface sf{{0, 0, 0}, {1, 0, 0}, 0, 0.0, 0.0};
            double col[3] = {1, 1, 1};
            if (h) { sf = surface_at<true>(S, r, q.best_t, q.id0, q.id1); MaterialV m = material_at(S, sf.material); if (m.texture >= 0) textured_colour(S, m, sf.u, sf.v, m.colour); col[0] = m.colour[0]; col[1] = m.colour[1]; col[2] = m.colour[2]; }
            t[i] = h ? q.best_t : 0.0;
            p[3 * i] = sf.p.x; p[3 * i + 1] = sf.p.y; p[3 * i + 2] = sf.p.z;
            nrm[3 * i] = sf.n.x; nrm[3 * i + 1] = sf.n.y; nrm[3 * i + 2] = sf.n.z;
            colour[3 * i] = col[0]; colour[3 * i + 1] = col[1]; colour[3 * i + 2] = col[2];
            if (overflow) atomicAdd(overflow_count, 1ull);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_debug_blocked(DevScene Sg, const double* __restrict__ o, const double* __restrict__ d,
                                                           const double* __restrict__ max_dist, uint32_t n, int32_t* blocked, unsigned long long* overflow_count) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Scene S = scene_view(Sg);
    const uint32_t B = 64u / (uint32_t)S.lane_fold, n_batches = (n + B - 1) / B;
    for (uint32_t b = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; b < n_batches; b += gridDim.x * (kBlock / 64)) {
        const uint32_t i = b * B + lane_id();
        Query<true> q;
        q.active = i < n && lane_id() < B; q.blocked = false; q.best_t = 0; q.id0 = 0; q.id1 = 0; q.max_dist = 0;
        Ray r{0, 0, 0, 0, 0, 0};
        if (q.active) { r = {o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]}; q.max_dist = max_dist[i]; }
        bool overflow;
        trace<true, true>(S, r, q, lds, overflow);
        if (q.active) { blocked[i] = q.blocked ? 1 : 0; if (overflow) atomicAdd(overflow_count, 1ull); }
    }
}

} // namespace

typedef void (*PrimaryKernel)(PrimaryArgs);
static PrimaryKernel primary_variant(int v) {                      // bit 0 FANCY, bit 1 SOFT, bit 2 MESH, bit 3: the lean variant built for five workgroups per CU
    switch (v & 15) {
        case 0: return k_primary<false, false, false, FT_PRIMARY_BLOCKS>;
        case 8: return k_primary<false, false, false, FT_LEAN_BLOCKS>;
        case 1: case 9: return k_primary<true, false, false, 2>;
        case 2: case 10: return k_primary<false, true, false, FT_PRIMARY_BLOCKS>;
        case 3: case 11: return k_primary<true, true, false, 2>;
        case 4: case 12: return k_primary<false, false, true, FT_PRIMARY_BLOCKS>;
        case 5: case 13: return k_primary<true, false, true, 2>;
        case 6: case 14: return k_primary<false, true, true, FT_PRIMARY_BLOCKS>;
        default: return k_primary<true, true, true, 2>;
    }
}

typedef void (*BounceKernel)(BounceArgs);
static BounceKernel bounce_variant(int v) {
    switch (v & 7) {
        case 0: return k_bounce<false, false, false>;
        case 1: return k_bounce<true, false, false>;
        case 2: return k_bounce<false, true, false>;
        case 3: return k_bounce<true, true, false>;
        case 4: return k_bounce<false, false, true>;
        case 5: return k_bounce<true, false, true>;
        case 6: return k_bounce<false, true, true>;
        default: return k_bounce<true, true, true>;
    }
}

// ============================================================================================ launchers
static int blocks_for(uint32_t n, int grid) { uint32_t need = (n + kBlock - 1) / kBlock; if (need < 1) need = 1; return (int)(need < (uint32_t)grid ? need : (uint32_t)grid); }

void launch_primary(const Launch& L, const DevScene& S, const Primary& gen, RayBuf next, double* acc, uint32_t acc_stride, int max_depth, FrameCounters* fc) {
    const PrimaryArgs a{S, gen, next, acc, fc, acc_stride, max_depth};
    hipLaunchKernelGGL(primary_variant(L.variant), dim3(L.grid), dim3(kBlock), L.lds_bytes, L.stream, a);
}
void launch_classify(const Launch& L, const DevScene& S, const Primary& gen_list, const ClassifyOut& out, double jitter_extent, uint32_t epoch, FrameCounters* fc) {
    const ClassifyArgs a{S, gen_list, out, fc, jitter_extent, epoch};
    const uint32_t n_blocks = gen_list.n_pix / 64u;
    hipLaunchKernelGGL(k_classify, dim3((n_blocks + kClassifyBlock - 1u) / kClassifyBlock), dim3(kClassifyBlock), 0, L.stream, a);
}
void launch_bounce(const Launch& L, const DevScene& S, const Primary& gen, RayBuf rays, RayBuf next, double* acc, uint32_t acc_stride, int bounce, int max_depth, bool follow, FrameCounters* fc) {
    const BounceArgs a{S, gen, rays, next, acc, fc, acc_stride, bounce, max_depth, follow ? 1 : 0};
    hipLaunchKernelGGL(bounce_variant(L.variant), dim3(L.grid), dim3(kBlock), L.lds_bytes, L.stream, a);
}
void launch_resolve(const Launch& L, const ResolveArgs& a) {
    const uint32_t work = a.block_pos ? (a.n_blocks_total * 64u > a.n_pix_host ? a.n_blocks_total * 64u : a.n_pix_host) : a.n_pix_host;
    // L.grid: one resident round of workgroups (occupancy_blocks_resolve x CUs; the grouped path holds 3 x 16 colours per lane: fewer waves fit
    // than the tracing kernels' grids assume), so that the hand-over at the end waits for one generation of workgroups, not for several
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(work, L.grid)), dim3(kBlock), 0, L.stream, a);
}
#ifdef FT_ITEM_COUNTS
extern "C" int ft_debug_item_counts(unsigned long long out[48], int reset) {      // [0..15] item counters, [16..47] section clocks, summed over the waves' slots
    static unsigned long long host[8192 * 48];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clk), sizeof host) != hipSuccess) return -1;
    for (int k = 0; k < 48; ++k) { out[k] = 0; for (int w = 0; w < 8192; ++w) out[k] += host[w * 48 + k]; }
    if (reset) { static const unsigned long long z[8192 * 48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
void launch_report(const Launch& L, FrameCounters* fc, FrameReport* report) { hipLaunchKernelGGL(k_report, dim3(1), dim3(kBlock), 0, L.stream, fc, report); }
void launch_resolve_corner(const Launch& L, const double* acc, uint32_t acc_stride, uint32_t w, uint32_t h, const uint32_t* out_index, double* out_rgb, uint8_t* out_rgba) {
    hipLaunchKernelGGL(k_resolve_corner, dim3(blocks_for(w * h, L.grid * 4)), dim3(kBlock), 0, L.stream, acc, acc_stride, w, h, out_index, out_rgb, out_rgba);
}
void launch_debug_closest(const Launch& L, const DevScene& S, const double* o, const double* d, uint32_t n, int32_t* hit, double* t,
                          double* p, double* nrm, double* colour, unsigned long long* overflow) {
    hipLaunchKernelGGL(k_debug_closest, dim3(blocks_for(n, L.grid)), dim3(kBlock), L.lds_bytes, L.stream, S, o, d, n, hit, t, p, nrm, colour, overflow);
}
void launch_debug_blocked(const Launch& L, const DevScene& S, const double* o, const double* d, const double* max_dist, uint32_t n,
                          int32_t* blocked, unsigned long long* overflow) {
    hipLaunchKernelGGL(k_debug_blocked, dim3(blocks_for(n, L.grid)), dim3(kBlock), L.lds_bytes, L.stream, S, o, d, max_dist, n, blocked, overflow);
}

} // namespace ftk

// Resident workgroups per CU for the persistent grids (register- and LDS-limited).
namespace ftk {
static int clamp_blocks(int n) { return n < 1 ? 1 : (n > 8 ? 8 : n); }
// Resident workgroups per CU of k_primary for this scene; *variant gains bit 3 when the five-workgroup build of the lean variant fits.
int occupancy_blocks_primary(size_t lds_bytes, int* variant) {
    int n = 0;
    if ((*variant & 7) == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, primary_variant(8), kBlock, lds_bytes) == hipSuccess && n >= FT_LEAN_BLOCKS) { *variant |= 8; return clamp_blocks(n); }
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, primary_variant(*variant), kBlock, lds_bytes) != hipSuccess) n = 1;
    return clamp_blocks(n);
}
int occupancy_blocks_resolve() {
    int n = 0;
    return (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_resolve, kBlock, 0) == hipSuccess && n > 0) ? clamp_blocks(n) : 2;
}
int occupancy_blocks_bounce(size_t lds_bytes, int variant) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bounce_variant(variant), kBlock, lds_bytes) != hipSuccess) n = 1;
    return clamp_blocks(n);
}
} // namespace ftk
