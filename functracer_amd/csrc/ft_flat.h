// ft_flat.h — the flat, HBM-resident scene representation shared by the host flattener
// (ft_scene.cpp) and the gfx950 kernels (ft_kernels.hip).  DESIGN.md §"Data layout in HBM".
//
// The reference evaluates a tree of closures per ray (Scene.fs:67-104).  Here the tree is
// compiled once, on the host, into
//   * leaves:   one per primitive INSTANCE with the world->model / model->world matrices of all
//               enclosing Transform nodes pre-composed (Transform.fs:80-87; t is invariant under
//               them), the flipNormals parity and the statically resolved material;
//   * a program: a linear post-order instruction stream over the leaves that reproduces
//               Ray.group concatenation order (Ray.fs:34) and Csg.constructedSolid (Csg.fs:74-94);
//   * meshes:   clipped triangles + BSP nodes exactly as BspMesh.compile builds them
//               (BspMesh.fs:51-65).
// Everything a wavefront reads while tracing is wave-uniform and is fetched with scalar loads.
#ifndef FT_FLAT_H
#define FT_FLAT_H
#include <stdint.h>

namespace ftd {

enum LeafKind : uint32_t {
    LK_SPHERE = 0,    // Sphere.fs:11-21
    LK_PLANE = 1,     // Plane.fs:28-33
    LK_SQUARE = 2,    // Cube.fs:9-15
    LK_CIRCLE = 3,    // Cylinder.fs:22
    LK_CUBE = 4,      // Cube.fs:17-25 (six squares, evaluated in the cube's own frame)
    LK_CONE = 5,      // Cone.fs:7-27
    LK_CYLINDER = 6,  // Cylinder.fs:8-20
    LK_SOLIDCYL = 7,  // Cylinder.fs:25-29 (top disc, bottom disc, sides)
    LK_MESH = 8       // BspMesh.fs:67-76, 95-97; also single Triangle primitives (Triangle.fs:43-66)
};

enum LeafFlags : uint32_t {
    LF_FLIP = 1u,          // odd number of flipNormals on the path (Ray.fs:36)
    LF_XFORM = 2u,         // at least one Transform node encloses the leaf (normal is re-normalised, Transform.fs:86)
    LF_LIT = 4u            // material.applyLighting (casts shadows, Scene.fs:121)
};

// 128 bytes; array element i lives at leaf_w2m + 16*i doubles.
struct Leaf {
    double w2m[12];        // rows of the 3x4 world->model matrix
    uint32_t kind;
    uint32_t flags;
    uint32_t material;
    uint32_t mesh;         // index into meshes[] for LK_MESH
    uint32_t pad[4];
};

struct Material {          // Ray.fs:4-10; 64 bytes
    double colour[3];
    double roughness, reflectance, shineyness;
    uint32_t apply_lighting;
    int32_t texture;       // -1 = none
    uint32_t hue_rot;      // number of hueShift channel rotations applied after the colour source (Ray.fs:51-55)
    uint32_t pad;
};

// Scene.Texture (Scene.fs:47-53): a Grid or an Image under at most kMaxUvOps uv functions, outermost (applied first) first.
// 48 doubles.  op kind 0 = Texture.scale (u/a, v/b) (Texture.fs:14-16); kind 1 = Texture.rotate with
// a = cos, b = sin of the angle (Texture.fs:18-22).  (The reference nests TextureFunction without bound; its scenes use two at most.
// The ops are read one by one through scalar loads, so their number costs the kernels nothing but the record's bytes.)
constexpr int kMaxUvOps = 13;
struct Texture {             // Scene.Texture (Scene.fs:47-53) flattened; 384 bytes = 48 doubles
    double c1[3], c2[3];     // Grid: the two colours.  Image: c1[0] = width, c1[1] = height
    double n_ops;
    double kind;             // 0 = Texture.grid, 1 = ImageTexture.image
    double pixel_base;       // Image: byte offset of its Rgb24 rows in DevScene::tex_pixels
    double ops[kMaxUvOps][3];// uv functions outermost first: {0, sx, sy} scale | {1, cos a, sin a} rotate
};

enum LightKind : uint32_t { LT_DIRECTIONAL = 0, LT_SOFT = 1, LT_POINT = 2 };
struct Light {             // Light.fs:7-14; 96 bytes
    double v[3];           // normalised direction | position
    double falloff[3];
    double colour[3];
    double scatter;
    uint32_t kind;
    int32_t samples;
    double tan_half_scatter;  // tan (scattering / 2) (Jitter.fs:30), evaluated on the host
};

struct Mesh {
    int32_t root;          // >= 0: BSP branch node index; < 0: ~leaf index (top-level Leaf, brute force, no AABB)
    uint32_t n_source_tris;
    uint32_t max_depth;    // BspMesh.maxDepth of the compiled tree
    int32_t bvh_root;      // top-level-Leaf meshes only: root of the device-side BVH (>= 0 node, < 0 ~leaf), or INT32_MIN = none
};
// 4-wide BVH node for packet traversal (28 doubles): the boxes of up to four children live in the PARENT, so one scalar-load
// round trip decides four subtrees (the two children of each child of a binary BVH node, collapsed):
//   [6*c .. 6*c+5] = lo xyz, hi xyz of child c (c = 0,1: the left half; 2,3: the right half)
//   [24], [25]     = int32 child[4]: >= 0 wide node, < 0 ~leaf index, INT32_MIN empty slot (whose box is all NaN: no ray passes it)
//   [26]           = uint32 axes: split axis of the binary node | of its left child << 8 | of its right child << 16
constexpr int kWideNodeDoubles = 28;
// The reference-shaped BSP of a `bspMesh depth` primitive (root >= 0) has a two-levels-at-a-time form as well, for the same walk:
// 40 doubles = five BspNode slots of the node array per branch (ft_scene.cpp, widen_bsp): the boxes of the branch's two children,
// those of its four grandchildren in the reference's visiting order (right-right, right-left, left-right, left-left), int32 child[4]
// at doubles 36..37 (>= 0 slot of the grandchild's record, < 0 ~leaf, INT32_MIN empty).  mesh_wide[mesh] = slot of the root's record.
struct BspNode {           // 64 bytes; BspMesh.fs:12-19 (also used for BVH nodes)
    double bmin[3], bmax[3];
    int32_t left, right;   // >= 0 branch node; < 0: ~leaf index
    uint32_t axis;         // BVH nodes: split axis (left = lower centroids)
    uint32_t pad;
};
struct BspLeaf { uint32_t first_tri, n_tris; };
// Triangles: 9 doubles each = v0, edge1 = v1 - v0, edge2 = v2 - v0 (Triangle.fs:45-46 evaluated once on the host).
// tri_orig[k] = index of the same triangle in the mesh's reference-order list (identity except inside BVH leaves).
//
// The BVH is an acceleration structure the reference does not have.  A top-level Leaf is "every triangle,
// in list order" (BspMesh.fs:53, 95-97); closest-hit / any-hit over that list only depend on the set of hits
// and, for equal t, on list order.  The BVH visits the ORIGINAL (unclipped) triangles with the same
// per-triangle arithmetic, prunes only boxes that cannot hold a closer hit (boxes are inflated, so rounding
// cannot prune a real hit) and breaks t ties by list index, so the result is identical to the linear scan.
// Hit lists under CSG still use the linear scan (order matters there).

// Program words: opcode in the low 8 bits, argument in the high 24.
enum Op : uint32_t {
    OP_END = 0,
    OP_LEAF_FOLD = 1,      // intersect leaf, fold every hit straight into the running closest / any-hit
    OP_LEAF_PUSH = 2,      // intersect leaf, append hits to the per-lane hit list (inside a CSG subtree)
    OP_MARK = 3,           // push the current list length (start of an operand segment)
    OP_CSG = 4,            // merge the two topmost segments with rule table arg (ft_csg_op)
    OP_FOLD_LIST = 5,      // fold the per-lane list into closest / any-hit and clear it
    OP_CULL = 6,           // arg = cull record; next word = number of program words of the item that follows.
                           // If no lane of the wave can possibly hit the item, the item is skipped.
    OP_SKIP_IF_EMPTY = 7,  // after operand A of subtract / intersect: if no lane has an A hit the result is empty for
                           // every lane (Csg.fs:27-44 never Take/Flip a B hit while outside A), so pop A's mark and skip arg words
    OP_CSG_PAIR = 8        // a CSG node over two bare primitives, merged in registers.  arg = words of the generic sequence
                           // (MARK, LEAF_PUSH A, ..., CSG[, FOLD_LIST]) that follows the two operand words
                           //   word 1 = leaf A;  word 2 = leaf B | op << 24 | fold << 26 (fold: top level, result goes to the query)
                           // and is skipped when the fast path applies to every lane of the wave (each operand gave at most two
                           // hits, none NaN); otherwise execution simply continues into the generic sequence.
};

// Conservative bound of one top-level item (a leaf or an outermost CSG subtree), 24 doubles.
// Skipping is exact: the line misses an inflated bounding sphere of every leaf box of the item, AND the
// ray is not near-parallel to any plane-derived face of it (Plane.intersect's parallel-ray rule,
// Plane.fs:13-16, can return a hit at the ray origin wherever that origin is).
struct CullRecord {
    double centre[3];
    double radius2;        // (r * (1 + 1e-6) + 1e-9)^2
    double n_rows;         // number of parallel-sensitive directions, <= 6
    double rows[6][3];     // world-space rows of world->model matrices whose dot with d is a plane denominator
    double pad;
};
inline uint32_t make_op(uint32_t op, uint32_t arg) { return op | (arg << 8); }

// Hit identity: id0 = leaf | sub << 24 | lit << 29 | flip << 30 | sideB << 31 (sideB only while sorting); id1 = triangle index.
enum : uint32_t { ID_LEAF_MASK = 0x00FFFFFFu, ID_SUB_SHIFT = 24, ID_SUB_MASK = 0x7u, ID_LIT = 1u << 29, ID_FLIP = 1u << 30, ID_SIDE_B = 1u << 31, ID_MISS = 0xFFFFFFFFu };

} // namespace ftd
#endif
