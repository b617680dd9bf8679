// ft_bvh.hip — the exact BVH of a top-level-Leaf mesh built ON THE DEVICE: a linear BVH over the mesh's original
// triangles (Morton order of the centroids, Karras' parallel hierarchy, bottom-up boxes), emitted in the layouts the traversal
// kernels already walk (ft_flat.h: binary BspNode / BspLeaf records for per-lane traversal, 4-wide nodes for packets, <= 64 coarse
// boxes for k_classify).  It replaces the host's recursive median split (ft_scene.cpp, BspBuilder::bvh_build) behind the same
// `bspMesh 0 file` primitive (BspMesh.fs:88-97): the reference scans the triangle list linearly there (BspMesh.fs:95-97), and any
// tree that (a) holds every triangle once, (b) bounds them with inflated boxes and (c) lets ties go to the lower list index gives
// that scan's closest hit / any hit exactly (ft_flat.h) - which tree it is only changes the cost of the walk.
//
// Stages (all on the context's stream; n = triangles of the mesh, >= 8):
//   k_bvh_prepare    per triangle: its box from (v0, v0 + e1, v0 + e2) - the same records the hit test reads - and the mesh bounds
//   k_bvh_morton     30-bit Morton code of the box centre inside the mesh bounds
//   rocprim radix sort of (code, triangle) pairs - stable, so equal codes stay in list order
//   k_bvh_hierarchy  Karras 2012: internal node i covers a range of the sorted order and splits it at the highest differing bit
//                    (equal codes: by position), parents recorded
//   k_bvh_fit        leaves to root: the second child to arrive unites the boxes; also the height of every subtree
//   k_bvh_emit       subtrees of at most four triangles become leaves; BspNode / BspLeaf / sorted triangle copies / tri_orig /
//                    4-wide nodes written into the ranges the flattener reserved for this mesh
//   k_bvh_coarse     one thread: the level of the tree with at most 64 nodes, as float boxes rounded outward
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <math.h>
#include <stdint.h>

#include "ft_device.h"

using namespace ftd;

namespace ftk {
namespace {

constexpr uint32_t kLeafTris = 4;                                   // as the host builder (BspBuilder::kBvhLeafTris)

// Order-preserving map double -> uint64 for atomicMin / atomicMax on coordinates.
__device__ __forceinline__ unsigned long long ordered(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unordered(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

struct BuildState {                                                 // device scratch of one build
    unsigned long long lo[3], hi[3];                                // mesh bounds (ordered keys)
    unsigned long long extent;                                      // largest |coordinate| (ordered key of a non-negative double)
    uint32_t bad;                                                   // a non-finite coordinate was seen
    uint32_t height;                                                // height of the tree in nodes (root = its height)
};

__global__ void k_bvh_init(BuildState* st) {
    for (int a = 0; a < 3; ++a) { st->lo[a] = ordered(__builtin_inf()); st->hi[a] = ordered(-__builtin_inf()); }
    st->extent = ordered(0.0); st->bad = 0u; st->height = 0u;
}

// boxes: 6 doubles per triangle (lo xyz, hi xyz), exact min / max of the three vertices as the hit test sees them.
__global__ __launch_bounds__(256) void k_bvh_prepare(const double* __restrict__ tris, uint32_t first, uint32_t n, double* __restrict__ boxes, BuildState* st) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()}, ext = 0.0;
    bool bad = false;
    if (i < n) {
        const double* T = tris + 9ull * (first + i);
        for (int a = 0; a < 3; ++a) {
            const double v0 = T[a], v1 = T[a] + T[3 + a], v2 = T[a] + T[6 + a];
            lo[a] = fmin(v0, fmin(v1, v2)); hi[a] = fmax(v0, fmax(v1, v2));
            if (!(fabs(v0) < 1e300) || !(fabs(v1) < 1e300) || !(fabs(v2) < 1e300)) bad = true;
            ext = fmax(ext, fmax(fabs(v0), fmax(fabs(v1), fabs(v2))));
            boxes[6ull * i + a] = lo[a]; boxes[6ull * i + 3 + a] = hi[a];
        }
    }
    // wave reduction, then one atomic per wave and quantity
    for (int off = 32; off > 0; off >>= 1) {
        for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], off)); }
        ext = fmax(ext, __shfl_xor(ext, off));
    }
    if ((threadIdx.x & 63u) == 0u && !(lo[0] > hi[0])) {
        for (int a = 0; a < 3; ++a) { atomicMin(&st->lo[a], ordered(lo[a])); atomicMax(&st->hi[a], ordered(hi[a])); }
        atomicMax(&st->extent, ordered(ext));
    }
    if (__any(bad) && (threadIdx.x & 63u) == 0u) atomicOr(&st->bad, 1u);
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) {           // 10 bits -> every third bit
    v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void k_bvh_morton(const double* __restrict__ boxes, uint32_t n, const BuildState* st, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const double lo = unordered(st->lo[a]), hi = unordered(st->hi[a]);
        const double c = 0.5 * (boxes[6ull * i + a] + boxes[6ull * i + 3 + a]);
        const double w = hi - lo;
        double u = w > 0.0 ? (c - lo) / w : 0.0;
        u = fmin(fmax(u * 1024.0, 0.0), 1023.0);
        q[a] = (uint32_t)u;
    }
    keys[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);   // bit 3k+2: x, 3k+1: y, 3k: z
    vals[i] = i;
}

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees" (HPG 2012), sections 3-4.
// delta(i, j) = length of the common prefix of the keys at sorted positions i and j, ties broken by the position itself.
__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a != b) return __clz((int)(a ^ b));
    return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}
// Node references inside the build: internal node i -> i; sorted leaf k -> ~k.
__global__ __launch_bounds__(256) void k_bvh_hierarchy(const uint32_t* __restrict__ keys, uint32_t n, int32_t* __restrict__ left, int32_t* __restrict__ right,
                                                        uint32_t* __restrict__ range_first, uint32_t* __restrict__ range_last, int32_t* __restrict__ parent_of_node,
                                                        int32_t* __restrict__ parent_of_leaf, uint32_t* __restrict__ split_bit) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x), N = (int)n;
    if (i >= N - 1) return;
    const int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2) if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {                   // binary search for the split position
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int lref = (lo == gamma) ? ~gamma : gamma, rref = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lref; right[i] = rref;
    range_first[i] = (uint32_t)lo; range_last[i] = (uint32_t)hi;
    if (lref >= 0) parent_of_node[lref] = i; else parent_of_leaf[~lref] = i;
    if (rref >= 0) parent_of_node[rref] = i; else parent_of_leaf[~rref] = i;
    if (i == 0) parent_of_node[0] = -1;
    // the bit the two halves differ in: bit p of the code is axis 2 - p % 3 (x = 0); equal codes split by position: axis 0
    const uint32_t ka = keys[gamma], kb = keys[gamma + 1];
    split_bit[i] = ka != kb ? (uint32_t)(2 - (31 - __clz((int)(ka ^ kb))) % 3) : 0u;
}

// Boxes of the internal nodes, leaves to root: every leaf walks up; the first child to reach a node leaves, the second unites.
__global__ __launch_bounds__(256) void k_bvh_fit(const double* __restrict__ tri_boxes, const uint32_t* __restrict__ vals, uint32_t n, const int32_t* __restrict__ left,
                                                  const int32_t* __restrict__ right, const int32_t* __restrict__ parent_of_node, const int32_t* __restrict__ parent_of_leaf,
                                                  uint32_t* __restrict__ arrived, double* __restrict__ node_boxes, uint32_t* __restrict__ node_height, BuildState* st) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    int32_t node = parent_of_leaf[k];
    while (node >= 0) {
        __threadfence();                                            // this thread's box of the child below is visible before the counter moves
        if (atomicAdd(&arrived[node], 1u) == 0u) return;            // the sibling subtree is not done yet: its thread carries on from here
        __threadfence();
        double lo[3], hi[3];
        uint32_t h = 0;
        const int32_t ch[2] = {left[node], right[node]};
        for (int c = 0; c < 2; ++c) {
            const volatile double* b = ch[c] >= 0 ? node_boxes + 6ull * (uint32_t)ch[c] : tri_boxes + 6ull * vals[~ch[c]];
            for (int a = 0; a < 3; ++a) { const double l = b[a], u = b[3 + a]; lo[a] = c == 0 ? l : fmin(lo[a], l); hi[a] = c == 0 ? u : fmax(hi[a], u); }
            const uint32_t hc = ch[c] >= 0 ? ((const volatile uint32_t*)node_height)[ch[c]] : 0u;
            h = hc > h ? hc : h;
        }
        for (int a = 0; a < 3; ++a) { node_boxes[6ull * (uint32_t)node + a] = lo[a]; node_boxes[6ull * (uint32_t)node + 3 + a] = hi[a]; }
        node_height[node] = h + 1u;
        if (node == 0) st->height = h + 1u;
        node = parent_of_node[node];
    }
}

// What the emitters read of a finished hierarchy, whichever builder made it: internal node i (0 <= i < n_internal, root 0) covers the sorted
// positions range_first[i] .. range_last[i], has box node_boxes[i] and children left[i] / right[i] (>= 0 an internal node, < 0 ~j: build
// leaf j); build leaf j holds the positions leaf_first[j] .. + leaf_count[j] - 1 (at most kLeafTris of them) and has box leaf_boxes[j].
// The linear BVH has one build leaf per triangle (j = its sorted position) and lets internal nodes of at most kLeafTris triangles stand
// as leaves (scene_ref); the surface-area builder stops splitting at kLeafTris and makes those ranges build leaves itself.
struct EmitArgs {
    const double* tris_in; uint32_t first_global; uint32_t n;
    const uint32_t* vals; const int32_t* left; const int32_t* right; const uint32_t* range_first; const uint32_t* range_last; const uint32_t* split_bit;
    const double* leaf_boxes; const double* node_boxes; const BuildState* st;
    const uint32_t* leaf_first; const uint32_t* leaf_count; const uint32_t* counts;   // counts[0] = n_internal, counts[1] = build leaves
    // outputs: the ranges reserved for this mesh in the scene's arrays
    BspNode* nodes; uint32_t node_base;        // n - 1 records
    BspLeaf* leaves; uint32_t leaf_base;       // (n - 1) + n records: internal node i as a leaf -> leaf_base + i; sorted triangle k alone -> leaf_base + n - 1 + k
    double* tris_out; uint32_t* tri_orig; uint32_t tri_base;   // n sorted triangle records
    double* wide; uint32_t wide_base;          // n - 1 records of kWideNodeDoubles
};
__device__ __forceinline__ double pad_of(const BuildState* st) { return 1e-7 * unordered(st->extent) + 1e-300; }   // as the host builder: pruning can never drop a real hit
__device__ __forceinline__ uint32_t range_size(const EmitArgs& a, int32_t ref) { return ref < 0 ? a.leaf_count[~ref] : a.range_last[ref] - a.range_first[ref] + 1u; }
// The reference the traversal kernels use for build node `ref`: >= 0 a BspNode index, < 0 ~(BspLeaf index).
__device__ __forceinline__ int32_t scene_ref(const EmitArgs& a, int32_t ref) {
    if (ref < 0) return ~(int32_t)(a.leaf_base + (a.n - 1u) + (uint32_t)~ref);
    return range_size(a, ref) <= kLeafTris ? ~(int32_t)(a.leaf_base + (uint32_t)ref) : (int32_t)(a.node_base + (uint32_t)ref);
}
__device__ __forceinline__ int32_t wide_ref(const EmitArgs& a, int32_t ref) {
    const int32_t r = scene_ref(a, ref);
    return r < 0 ? r : (int32_t)(a.wide_base + (uint32_t)ref);
}
__device__ __forceinline__ void padded_box(const EmitArgs& a, int32_t ref, double pad, double out[6]) {
    const double* b = ref >= 0 ? a.node_boxes + 6ull * (uint32_t)ref : a.leaf_boxes + 6ull * (uint32_t)~ref;
    for (int k = 0; k < 3; ++k) { out[k] = b[k] - pad; out[3 + k] = b[3 + k] + pad; }
}

__global__ __launch_bounds__(256) void k_bvh_emit(EmitArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const double pad = pad_of(a.st);
    if (i < a.n) {                                                  // sorted copy of triangle i, its list index, its single-triangle leaf
        const uint32_t src = a.vals[i];
        const double* T = a.tris_in + 9ull * (a.first_global + src);
        double* O = a.tris_out + 9ull * (a.tri_base + i);
        for (int k = 0; k < 9; ++k) O[k] = T[k];
        a.tri_orig[a.tri_base + i] = a.first_global + src;
        a.leaves[a.leaf_base + (a.n - 1u) + i] = i < a.counts[1] ? BspLeaf{a.tri_base + a.leaf_first[i], a.leaf_count[i]} : BspLeaf{0u, 0u};   // build leaf i
    }
    if (i >= a.counts[0]) return;
    // internal node i: as a leaf (used when its range is small) and as binary / 4-wide nodes (used otherwise)
    a.leaves[a.leaf_base + i] = BspLeaf{a.tri_base + a.range_first[i], a.range_last[i] - a.range_first[i] + 1u};
    BspNode nd;
    double box[6];
    padded_box(a, (int32_t)i, pad, box);
    for (int k = 0; k < 3; ++k) { nd.bmin[k] = box[k]; nd.bmax[k] = box[3 + k]; }
    nd.left = scene_ref(a, a.left[i]); nd.right = scene_ref(a, a.right[i]); nd.axis = a.split_bit[i]; nd.pad = 0u;
    a.nodes[a.node_base + i] = nd;
    // 4-wide node (ft_flat.h): the two children of each child; a leaf child takes one slot of its half
    double* w = a.wide + (unsigned long long)kWideNodeDoubles * (a.wide_base + i);
    int32_t child[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    uint32_t axes = a.split_bit[i];
    const int32_t halves[2] = {a.left[i], a.right[i]};
    for (int k = 0; k < 24; ++k) w[k] = __builtin_nan("");          // an empty slot's box fails every slab test by itself (ft_flat.h)
    for (int h = 0; h < 2; ++h) {
        const int32_t c = halves[h];
        if (scene_ref(a, c) < 0) { child[2 * h] = scene_ref(a, c); padded_box(a, c, pad, w + 6 * (2 * h)); continue; }
        axes |= a.split_bit[c] << (8 * (h + 1));
        const int32_t gk[2] = {a.left[c], a.right[c]};
        for (int k = 0; k < 2; ++k) { child[2 * h + k] = wide_ref(a, gk[k]); padded_box(a, gk[k], pad, w + 6 * (2 * h + k)); }
    }
    int32_t* wc = reinterpret_cast<int32_t*>(w + 24);
    wc[0] = child[0]; wc[1] = child[1]; wc[2] = child[2]; wc[3] = child[3];
    reinterpret_cast<uint32_t*>(w + 26)[0] = axes; reinterpret_cast<uint32_t*>(w + 26)[1] = 0u;
    w[27] = 0.0;
}

__device__ __forceinline__ float round_down(double v) { float f = (float)v; if ((double)f > v) f = __uint_as_float(f > 0.0f ? __float_as_uint(f) - 1u : (f < 0.0f ? __float_as_uint(f) + 1u : 0x80000001u)); return f; }
__device__ __forceinline__ float round_up(double v) { return -round_down(-v); }
// <= 64 boxes that together hold every triangle: one level of the tree (the same rule as the host flattener), rounded outward with
// room for the float arithmetic of k_classify's test; `count` slots are always filled (the last box repeats).
__global__ void k_bvh_coarse(EmitArgs a, float* __restrict__ coarse, uint32_t count) {
    __shared__ int32_t frontier[2][64];
    if (threadIdx.x != 0) return;
    int cur = 0; uint32_t nf = 1;
    frontier[0][0] = 0;
    for (;;) {
        uint32_t nn = 0; bool any_inner = false, fits = true;
        for (uint32_t k = 0; k < nf && fits; ++k) {
            const int32_t c = frontier[cur][k];
            const bool inner = c >= 0 && range_size(a, c) > kLeafTris;
            if (nn + (inner ? 2u : 1u) > 64u) { fits = false; break; }
            if (!inner) { frontier[cur ^ 1][nn++] = c; continue; }
            any_inner = true;
            frontier[cur ^ 1][nn++] = a.left[c]; frontier[cur ^ 1][nn++] = a.right[c];
        }
        if (!any_inner || !fits) break;
        cur ^= 1; nf = nn;
    }
    for (uint32_t k = 0; k < count; ++k) {
        const int32_t c = frontier[cur][k < nf ? k : nf - 1u];
        const double* b = c >= 0 ? a.node_boxes + 6ull * (uint32_t)c : a.leaf_boxes + 6ull * (uint32_t)~c;
        for (int x = 0; x < 3; ++x) {
            const double lo = b[x], hi = b[3 + x];
            const double pad = 1e-5 * (fabs(lo) + fabs(hi) + (hi - lo)) + 1e-30;
            coarse[6u * k + x] = round_down(lo - pad); coarse[6u * k + 3 + x] = round_up(hi + pad);
        }
    }
}

// The linear BVH's build leaves: sorted triangle k alone.
__global__ __launch_bounds__(256) void k_bvh_single_leaves(const double* __restrict__ tri_boxes, const uint32_t* __restrict__ vals, uint32_t n, uint32_t* __restrict__ leaf_first,
                                                            uint32_t* __restrict__ leaf_count, double* __restrict__ leaf_boxes, uint32_t* __restrict__ counts) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k == 0u) { counts[0] = n - 1u; counts[1] = n; }
    if (k >= n) return;
    leaf_first[k] = k; leaf_count[k] = 1u;
    for (int a = 0; a < 6; ++a) leaf_boxes[6ull * k + a] = tri_boxes[6ull * vals[k] + a];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The surface-area builder: top down, a level of the tree per round, over the Morton-sorted triangles.  A node is a range of the
// sorted order; a round (i) bins the centroids of every open node's triangles into kBins slabs of the node's box along each axis
// (count and box per bin, device-scope atomics), (ii) prices the 3 x (kBins - 1) planes of every open node as
// area(left) x count(left) + area(right) x count(right) and takes the cheapest, (iii) partitions every open node's range stably by
// the side of its plane - one stable radix sort of the positions by (range start, side) - and (iv) opens the children that still
// hold more than kLeafTris triangles.  A node no plane divides (coincident centroids), or one 28 levels down, is halved by position and
// hands its own box to both halves.  The tree only decides which boxes a ray looks into: no pixel depends on it (ft_flat.h).
constexpr int kBins = 16;
constexpr uint32_t kBinWords = 7;                                    // ordered lo xyz, hi xyz, count
struct SahState { uint32_t n_internal, n_leaves, level_begin, level_end, level, pad[3]; };

__global__ void k_sah_begin(SahState* s, uint32_t n, uint32_t* range_first, uint32_t* range_last, double* node_boxes, const BuildState* st, int32_t* node_of, uint32_t* seg_first) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p == 0u) {
        s->n_internal = 1u; s->n_leaves = 0u; s->level_begin = 0u; s->level_end = 1u; s->level = 0u;
        range_first[0] = 0u; range_last[0] = n - 1u;
        for (int a = 0; a < 3; ++a) { node_boxes[a] = unordered(st->lo[a]); node_boxes[3 + a] = unordered(st->hi[a]); }
    }
    if (p < n) { node_of[p] = 0; seg_first[p] = 0u; }
}
__global__ __launch_bounds__(256) void k_sah_clear_bins(const SahState* s, unsigned long long* bins) {
    const uint32_t open = s->level_end - s->level_begin;
    const unsigned long long words = (unsigned long long)open * 3u * kBins * kBinWords;
    for (unsigned long long w = blockIdx.x * 256ull + threadIdx.x; w < words; w += (unsigned long long)gridDim.x * 256ull) {
        const uint32_t f = (uint32_t)(w % kBinWords);
        bins[w] = f < 3u ? ordered(__builtin_inf()) : f < 6u ? ordered(-__builtin_inf()) : 0ull;
    }
}
__device__ __forceinline__ int bin_of(double c, double lo, double hi) {
    const double w = hi - lo;
    if (!(w > 0.0)) return 0;
    const double u = (c - lo) / w * (double)kBins;
    return u >= (double)(kBins - 1) ? kBins - 1 : (u > 0.0 ? (int)u : 0);
}
__global__ __launch_bounds__(256) void k_sah_bin(const SahState* s, uint32_t n, const uint32_t* __restrict__ vals, const int32_t* __restrict__ node_of,
                                                  const double* __restrict__ tri_boxes, const double* __restrict__ node_boxes, unsigned long long* bins) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int32_t o = node_of[p];
    if (o < 0 || (uint32_t)o < s->level_begin) return;              // in a finished part of the tree
    const double* tb = tri_boxes + 6ull * vals[p];
    const double* nb = node_boxes + 6ull * (uint32_t)o;
    unsigned long long* B = bins + (unsigned long long)((uint32_t)o - s->level_begin) * 3u * kBins * kBinWords;
    for (int a = 0; a < 3; ++a) {
        unsigned long long* w = B + (unsigned long long)(a * kBins + bin_of(0.5 * (tb[a] + tb[3 + a]), nb[a], nb[3 + a])) * kBinWords;
        for (int k = 0; k < 3; ++k) { atomicMin(&w[k], ordered(tb[k])); atomicMax(&w[3 + k], ordered(tb[3 + k])); }
        atomicAdd(&w[6], 1ull);
    }
}
// One WAVE per open node (a thread per node spent the upper rounds of the tree reading 336 words one after the other): lane l < 48
// holds bin l % 16 of axis l / 16, prefix and suffix unions run over the 16 lanes of an axis, lane b prices plane b of its axis, the
// cheapest of the 45 wins; lane 0 makes the two children (an internal node or a build leaf each) and their boxes.
struct BinBox { double lo[3], hi[3]; uint32_t cnt; };
__device__ __forceinline__ BinBox shfl_box(const BinBox& b, int src) {
    BinBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = __shfl(b.lo[k], src); r.hi[k] = __shfl(b.hi[k], src); }
    r.cnt = __shfl(b.cnt, src);
    return r;
}
__device__ __forceinline__ void unite(BinBox& a, const BinBox& b) { for (int k = 0; k < 3; ++k) { a.lo[k] = fmin(a.lo[k], b.lo[k]); a.hi[k] = fmax(a.hi[k], b.hi[k]); } a.cnt += b.cnt; }
__device__ __forceinline__ double half_area(const BinBox& b) { const double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2]; return dx * dy + dy * dz + dz * dx; }
__global__ __launch_bounds__(256) void k_sah_split(SahState* s, const unsigned long long* __restrict__ bins, uint32_t* range_first, uint32_t* range_last, int32_t* left, int32_t* right,
                                                   uint32_t* split_bit, uint32_t* plane, uint32_t* n_left, double* node_boxes, uint32_t* leaf_first, uint32_t* leaf_count) {
    const uint32_t o = s->level_begin + blockIdx.x * 4u + threadIdx.x / 64u, lane = threadIdx.x & 63u;
    if (o >= s->level_end) return;                                  // (whole waves leave together)
    const double inf = __builtin_inf();
    const uint32_t bin = lane % kBins;
    BinBox mine{{inf, inf, inf}, {-inf, -inf, -inf}, 0u};
    if (lane < 3u * kBins) {
        const unsigned long long* w = bins + ((unsigned long long)(o - s->level_begin) * 3u * kBins + lane) * kBinWords;
        mine.cnt = (uint32_t)w[6];
        if (mine.cnt) for (int k = 0; k < 3; ++k) { mine.lo[k] = unordered(w[k]); mine.hi[k] = unordered(w[3 + k]); }
    }
    BinBox pre = mine, suf = mine;                                  // inclusive unions over bins [0, bin] and [bin, kBins) of the lane's axis
    for (int d = 1; d < kBins; d <<= 1) {
        const BinBox a = shfl_box(pre, (int)lane - d), b = shfl_box(suf, (int)lane + d);
        if ((int)bin - d >= 0) unite(pre, a);
        if ((int)bin + d < kBins) unite(suf, b);
    }
    const BinBox before = shfl_box(pre, (int)lane - 1);             // bins [0, bin): what lies left of plane `bin`
    double cost = inf;
    if (lane < 3u * kBins && bin >= 1u && before.cnt && suf.cnt && s->level < 28u) cost = half_area(before) * (double)before.cnt + half_area(suf) * (double)suf.cnt;
    double best = cost; uint32_t who = lane;
    for (int d = 32; d >= 1; d >>= 1) { const double c2 = __shfl_xor(best, d); const uint32_t w2 = __shfl_xor(who, d); if (c2 < best || (c2 == best && w2 < who)) { best = c2; who = w2; } }
    const uint32_t first = range_first[o], size = range_last[o] - first + 1u;
    const double* nb = node_boxes + 6ull * o;
    BinBox lb, rb; uint32_t best_axis, best_plane, best_left;
    if (!(best < inf)) {                                            // no plane divides the node: halves by position, each under the node's own box
        best_axis = 3u; best_plane = 0u; best_left = size / 2u;
        for (int k = 0; k < 3; ++k) { lb.lo[k] = rb.lo[k] = nb[k]; lb.hi[k] = rb.hi[k] = nb[3 + k]; }
    } else {
        best_axis = who / kBins; best_plane = who % kBins;
        lb = shfl_box(before, (int)who); rb = shfl_box(suf, (int)who); best_left = lb.cnt;
    }
    if (lane != 0u) return;
    split_bit[o] = best_axis == 3u ? 0u : best_axis; plane[o] = best_axis << 8 | best_plane; n_left[o] = best_left;
    const uint32_t sizes[2] = {best_left, size - best_left}, firsts[2] = {first, first + best_left};
    int32_t refs[2];
    for (int c = 0; c < 2; ++c) {
        if (sizes[c] > kLeafTris) {
            const uint32_t id = atomicAdd(&s->n_internal, 1u);
            range_first[id] = firsts[c]; range_last[id] = firsts[c] + sizes[c] - 1u;
            const BinBox& bx = c == 0 ? lb : rb;
            for (int k = 0; k < 3; ++k) { node_boxes[6ull * id + k] = bx.lo[k]; node_boxes[6ull * id + 3 + k] = bx.hi[k]; }
            refs[c] = (int32_t)id;
        } else {
            const uint32_t j = atomicAdd(&s->n_leaves, 1u);
            leaf_first[j] = firsts[c]; leaf_count[j] = sizes[c];
            refs[c] = ~(int32_t)j;
        }
    }
    left[o] = refs[0]; right[o] = refs[1];
}
// Sort key of every position: (start of its segment, side of its node's plane).  Positions outside the open nodes keep their place.
__global__ __launch_bounds__(256) void k_sah_keys(const SahState* s, uint32_t n, const uint32_t* __restrict__ vals, const int32_t* __restrict__ node_of, const uint32_t* __restrict__ seg_first,
                                                   const double* __restrict__ tri_boxes, const double* __restrict__ node_boxes, const uint32_t* __restrict__ plane,
                                                   const uint32_t* __restrict__ n_left, const uint32_t* __restrict__ range_first, uint32_t* __restrict__ keys) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int32_t o = node_of[p];
    uint32_t side = 0u;
    if (o >= 0 && (uint32_t)o >= s->level_begin) {
        const uint32_t axis = plane[o] >> 8, b = plane[o] & 0xFFu;
        if (axis == 3u) side = p - range_first[o] >= n_left[o] ? 1u : 0u;
        else { const double* tb = tri_boxes + 6ull * vals[p]; const double* nb = node_boxes + 6ull * (uint32_t)o; side = (uint32_t)bin_of(0.5 * (tb[axis] + tb[3 + axis]), nb[axis], nb[3 + axis]) >= b ? 1u : 0u; }
    }
    keys[p] = side;                                                 // the stable partition of every segment is a scan of these flags away (k_sah_scatter)
}
// The stable partition of every open node's range by side: rights[p] = flags set before position p (exclusive scan over all positions), so
// within the segment that begins at f = seg_first[p] there are rights[p] - rights[f] of them before p and (p - f) minus that many lefts;
// the lefts of an open node fill its first n_left positions in order, the rights the rest.  Positions outside open nodes carry flag 0
// and keep their place.  (Round 3: this replaced a radix sort of (segment, side) keys over all n positions per round.)
__global__ __launch_bounds__(256) void k_sah_scatter(const SahState* s, uint32_t n, const uint32_t* __restrict__ vin, uint32_t* __restrict__ vout, const int32_t* __restrict__ node_of,
                                                      const uint32_t* __restrict__ seg_first, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ rights,
                                                      const uint32_t* __restrict__ n_left, const uint32_t* __restrict__ plane) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int32_t o = node_of[p];
    uint32_t dst = p;
    if (o >= 0 && (uint32_t)o >= s->level_begin) {
        const uint32_t f = seg_first[p], before_r = rights[p] - rights[f], before_l = (p - f) - before_r;
        dst = flags[p] ? f + n_left[o] + before_r : f + before_l;
    }
    vout[dst] = vin[p];
}
// After the partition: every position of an open node moves into the child that now covers it.
__global__ __launch_bounds__(256) void k_sah_descend(const SahState* s, uint32_t n, int32_t* node_of, uint32_t* seg_first, const uint32_t* __restrict__ range_first,
                                                      const uint32_t* __restrict__ n_left, const int32_t* __restrict__ left, const int32_t* __restrict__ right) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    const int32_t o = node_of[p];
    if (o < 0 || (uint32_t)o < s->level_begin) return;
    const bool second = p - range_first[o] >= n_left[o];
    const int32_t child = second ? right[o] : left[o];
    node_of[p] = child >= 0 ? child : -1;
    seg_first[p] = range_first[o] + (second ? n_left[o] : 0u);
}
__global__ void k_sah_next_level(SahState* s) { s->level_begin = s->level_end; s->level_end = s->n_internal; s->level += 1u; }
__global__ __launch_bounds__(256) void k_sah_leaf_boxes(const SahState* s, const uint32_t* __restrict__ vals, const double* __restrict__ tri_boxes, const uint32_t* __restrict__ leaf_first,
                                                         const uint32_t* __restrict__ leaf_count, double* __restrict__ leaf_boxes, uint32_t* __restrict__ counts, BuildState* st) {
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j == 0u) { counts[0] = s->n_internal; counts[1] = s->n_leaves; st->height = s->level; }   // rounds made = levels of internal nodes
    if (j >= s->n_leaves) return;
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()}, hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    for (uint32_t k = 0; k < leaf_count[j]; ++k) { const double* tb = tri_boxes + 6ull * vals[leaf_first[j] + k]; for (int a = 0; a < 3; ++a) { lo[a] = fmin(lo[a], tb[a]); hi[a] = fmax(hi[a], tb[3 + a]); } }
    for (int a = 0; a < 3; ++a) { leaf_boxes[6ull * j + a] = lo[a]; leaf_boxes[6ull * j + 3 + a] = hi[a]; }
}

#define BVH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

} // namespace

// Build the BVH of triangles [first_global, first_global + n) of `tris` into the reserved ranges: kind 0 the linear BVH, 1 the binned
// surface-area tree (meshes beyond 4 Mi triangles get the linear one: a round's bins would not fit).  Returns hipSuccess, or an error;
// *height receives the height of the binary tree in nodes (0: the mesh holds a non-finite coordinate and nothing was written).
hipError_t build_lbvh(hipStream_t stream, const LbvhTarget& t, uint32_t* height, int kind) {
    const uint32_t n = t.n;
    const bool sah = kind == 1 && n <= (4u << 20);
    hipError_t err = hipSuccess;
    char* scratch = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_bytes = 0, sort_bytes2 = 0;
    *height = 0;
    uint32_t key_bits = 1; while ((1ull << key_bits) < 2ull * n) ++key_bits;
    // scratch: state | tri boxes | node boxes | keys x2 | vals x2 | left right first last parent_node parent_leaf split arrived height | leaves | (surface-area builder) ...
    const size_t n8 = ((size_t)n + 7) / 8 * 8;
    const size_t open_max = (size_t)n / (kLeafTris + 1) + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
    const size_t o_state = take(sizeof(BuildState)), o_tb = take(n8 * 48), o_nb = take(n8 * 48), o_k0 = take(n8 * 4), o_k1 = take(n8 * 4), o_v0 = take(n8 * 4), o_v1 = take(n8 * 4),
                 o_l = take(n8 * 4), o_r = take(n8 * 4), o_f = take(n8 * 4), o_la = take(n8 * 4), o_pn = take(n8 * 4), o_pl = take(n8 * 4), o_sb = take(n8 * 4), o_ar = take(n8 * 4), o_h = take(n8 * 4),
                 o_lf = take(n8 * 4), o_lc = take(n8 * 4), o_lb = take(n8 * 48), o_cnt = take(16),
                 o_sah = take(sizeof(SahState)), o_bins = take(sah ? open_max * 3 * kBins * kBinWords * 8 : 8);
    {
        BVH_HIP(hipMalloc(reinterpret_cast<void**>(&scratch), off));
        BuildState* st = reinterpret_cast<BuildState*>(scratch + o_state);
        double* tb = reinterpret_cast<double*>(scratch + o_tb); double* nb = reinterpret_cast<double*>(scratch + o_nb);
        uint32_t* k0 = reinterpret_cast<uint32_t*>(scratch + o_k0); uint32_t* k1 = reinterpret_cast<uint32_t*>(scratch + o_k1);
        uint32_t* v0 = reinterpret_cast<uint32_t*>(scratch + o_v0); uint32_t* v1 = reinterpret_cast<uint32_t*>(scratch + o_v1);
        int32_t* left = reinterpret_cast<int32_t*>(scratch + o_l); int32_t* right = reinterpret_cast<int32_t*>(scratch + o_r);
        uint32_t* rf = reinterpret_cast<uint32_t*>(scratch + o_f); uint32_t* rl = reinterpret_cast<uint32_t*>(scratch + o_la);
        int32_t* pn = reinterpret_cast<int32_t*>(scratch + o_pn); int32_t* pl = reinterpret_cast<int32_t*>(scratch + o_pl);
        uint32_t* sb = reinterpret_cast<uint32_t*>(scratch + o_sb); uint32_t* ar = reinterpret_cast<uint32_t*>(scratch + o_ar); uint32_t* hh = reinterpret_cast<uint32_t*>(scratch + o_h);
        uint32_t* lf = reinterpret_cast<uint32_t*>(scratch + o_lf); uint32_t* lc = reinterpret_cast<uint32_t*>(scratch + o_lc); double* lb = reinterpret_cast<double*>(scratch + o_lb);
        uint32_t* counts = reinterpret_cast<uint32_t*>(scratch + o_cnt);
        SahState* ss = reinterpret_cast<SahState*>(scratch + o_sah);
        unsigned long long* bins = reinterpret_cast<unsigned long long*>(scratch + o_bins);
        const dim3 grid((n + 255u) / 256u), block(256);
        hipLaunchKernelGGL(k_bvh_init, dim3(1), dim3(1), 0, stream, st);
        hipLaunchKernelGGL(k_bvh_prepare, grid, block, 0, stream, t.tris, t.first_global, n, tb, st);
        hipLaunchKernelGGL(k_bvh_morton, grid, block, 0, stream, tb, n, st, k0, v0);
        BVH_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, k0, k1, v0, v1, n, 0, 30, stream));
        if (sah) { BVH_HIP(rocprim::exclusive_scan(nullptr, sort_bytes2, k0, k1, 0u, n, rocprim::plus<uint32_t>(), stream)); if (sort_bytes2 > sort_bytes) sort_bytes = sort_bytes2; }
        BVH_HIP(hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 16));
        BVH_HIP(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, k0, k1, v0, v1, n, 0, 30, stream));
        const uint32_t* vals = v1;                                  // sorted position -> triangle of the mesh
        if (!sah) {
            BVH_HIP(hipMemsetAsync(ar, 0, n8 * 4, stream));
            hipLaunchKernelGGL(k_bvh_hierarchy, grid, block, 0, stream, k1, n, left, right, rf, rl, pn, pl, sb);
            hipLaunchKernelGGL(k_bvh_fit, grid, block, 0, stream, tb, v1, n, left, right, pn, pl, ar, nb, hh, st);
            hipLaunchKernelGGL(k_bvh_single_leaves, grid, block, 0, stream, tb, v1, n, lf, lc, lb, counts);
        } else {
            // the Morton order is where the rounds start: neighbours in space are neighbours in the arrays the atomics of a round hit
            int32_t* node_of = pn; uint32_t* seg_first = reinterpret_cast<uint32_t*>(pl); uint32_t* plane = ar; uint32_t* n_left = hh;   // (the linear builder's arrays, unused here)
            uint32_t *vin = v1, *vout = v0, *kin = k0, *kout = k1;
            BVH_HIP(hipMemsetAsync(left, 0, n8 * 4, stream)); BVH_HIP(hipMemsetAsync(right, 0, n8 * 4, stream));
            BVH_HIP(hipMemsetAsync(rf, 0, n8 * 4, stream)); BVH_HIP(hipMemsetAsync(rl, 0, n8 * 4, stream)); BVH_HIP(hipMemsetAsync(sb, 0, n8 * 4, stream));
            hipLaunchKernelGGL(k_sah_begin, grid, block, 0, stream, ss, n, rf, rl, nb, st, node_of, seg_first);
            // A level cannot be the last before every range is down to a leaf's size: n halves at best per level, so the first
            // log2(n / kLeafTris) - 1 rounds need no answer from the device - the host launches for 2^round open nodes at most (the kernels
            // bound themselves by the device's own counters) and only the later rounds wait for the level's size (a readback and a
            // synchronise per round were 0.4 of the build's 6 ms).
            int sure_rounds = 0;
            while ((2ull << sure_rounds) * (unsigned long long)kLeafTris < (unsigned long long)n) ++sure_rounds;
            for (int round = 0; round < 64; ++round) {
                uint32_t open = (uint32_t)std::min<unsigned long long>(1ull << std::min(round, 31), open_max);
                if (round >= sure_rounds) {
                    SahState h{};
                    BVH_HIP(hipMemcpyAsync(&h, ss, sizeof h, hipMemcpyDeviceToHost, stream));
                    BVH_HIP(hipStreamSynchronize(stream));
                    open = h.level_end - h.level_begin;
                    if (open == 0u) break;
                    if (open > open_max) { err = hipErrorInvalidValue; goto done; }   // (cannot happen: every open node holds more than kLeafTris triangles)
                }
                const uint32_t words_blocks = (uint32_t)std::min<unsigned long long>(((unsigned long long)open * 3u * kBins * kBinWords + 255u) / 256u, 4096ull);
                hipLaunchKernelGGL(k_sah_clear_bins, dim3(words_blocks), block, 0, stream, ss, bins);
                hipLaunchKernelGGL(k_sah_bin, grid, block, 0, stream, ss, n, vin, node_of, tb, nb, bins);
                hipLaunchKernelGGL(k_sah_split, dim3((open + 3u) / 4u), block, 0, stream, ss, bins, rf, rl, left, right, sb, plane, n_left, nb, lf, lc);
                hipLaunchKernelGGL(k_sah_keys, grid, block, 0, stream, ss, n, vin, node_of, seg_first, tb, nb, plane, n_left, rf, kin);
                BVH_HIP(rocprim::exclusive_scan(sort_tmp, sort_bytes, kin, kout, 0u, n, rocprim::plus<uint32_t>(), stream));
                hipLaunchKernelGGL(k_sah_scatter, grid, block, 0, stream, ss, n, vin, vout, node_of, seg_first, kin, kout, n_left, plane);
                { uint32_t* x = vin; vin = vout; vout = x; }
                hipLaunchKernelGGL(k_sah_descend, grid, block, 0, stream, ss, n, node_of, seg_first, rf, n_left, left, right);
                hipLaunchKernelGGL(k_sah_next_level, dim3(1), dim3(1), 0, stream, ss);
            }
            vals = vin;
            hipLaunchKernelGGL(k_sah_leaf_boxes, grid, block, 0, stream, ss, vals, tb, lf, lc, lb, counts, st);
        }
        const EmitArgs ea{t.tris, t.first_global, n, vals, left, right, rf, rl, sb, lb, nb, st, lf, lc, counts,
                          t.nodes, t.node_base, t.leaves, t.leaf_base, t.tris, t.tri_orig, t.tri_base, t.wide, t.wide_base};
        hipLaunchKernelGGL(k_bvh_emit, grid, block, 0, stream, ea);
        if (t.coarse_count) hipLaunchKernelGGL(k_bvh_coarse, dim3(1), dim3(64), 0, stream, ea, t.coarse, t.coarse_count);
        BVH_HIP(hipGetLastError());
        BuildState h{};
        BVH_HIP(hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, stream));
        BVH_HIP(hipStreamSynchronize(stream));
        *height = h.bad ? 0u : h.height;
    }
done:
    if (sort_tmp) (void)hipFree(sort_tmp);
    if (scratch) (void)hipFree(scratch);
    return err;
}

} // namespace ftk
