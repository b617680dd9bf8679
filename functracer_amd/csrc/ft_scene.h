// ft_scene.h — host-side scene graph arena (mirror of Scene.fs:8-53, 107-110) and its
// compilation to the flat HBM layout of ft_flat.h.  Pure C++; no HIP here.
#ifndef FT_SCENE_H
#define FT_SCENE_H
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/functracer_hip.h"
#include "ft_flat.h"

namespace fth {

struct GraphNode {
    enum Kind { Prim, TriangleP, Mesh, Transform, MaterialF, HueShift, IgnoreLight, Texture, Group, Csg } kind = Prim;
    int32_t prim = 0;                 // ft_primitive_kind
    double tri[9] = {0};              // TriangleP
    int32_t depth = 0;                // Mesh: BspMesh.bspMesh depth
    std::vector<double> tris;         // Mesh: n x 9 (a,b,c)
    std::vector<ft_transform> xf;     // Transform: one basic transform or a Composed list
    ft_material mat{};                // MaterialF
    int32_t op = 0;                   // Csg
    std::vector<int32_t> children;
    double ca[3] = {0}, cb[3] = {0};  // Texture grid colours
    std::vector<double> uv_ops;
    std::vector<uint8_t> pixels;      // Texture image: Rgb24 rows (empty = grid)
    int32_t img_w = 0, img_h = 0;
};

struct FlatScene {
    std::vector<ftd::Leaf> leaves;
    std::vector<double> m2w;          // 12 per leaf
    std::vector<ftd::Material> materials;
    std::vector<ftd::Light> lights;
    std::vector<ftd::Texture> textures;
    std::vector<uint8_t> tex_pixels;  // Rgb24 rows of all image textures, back to back
    std::vector<uint32_t> program;
    std::vector<ftd::Mesh> meshes;
    std::vector<ftd::BspNode> nodes;
    std::vector<ftd::BspLeaf> bsp_leaves;
    std::vector<double> tris;         // 9 per triangle: v0, e1, e2
    std::vector<uint32_t> tri_orig;   // 1 per triangle (see ft_flat.h)
    std::vector<double> wide;         // 28 doubles per 4-wide BVH node (ft_flat.h), walked by coherent wavefronts
    std::vector<int32_t> mesh_wide;   // per mesh: root of its 4-wide BVH, or INT32_MIN
    std::vector<float> coarse_boxes;  // 6 floats per box (lo, hi; model space, rounded outward): <= 64 boxes per mesh that cover all its triangles
    std::vector<uint32_t> mesh_coarse;// per mesh: first box, box count (0 = none) - k_classify tests pixel blocks against them
    std::vector<ftd::CullRecord> culls;
    std::vector<uint32_t> item_pc;    // program counter of every top-level item, in order, + one sentinel (the OP_END word)
    std::vector<float> cull_items;    // 8 floats per top-level item: centre, radius (rounded up; +inf = unbounded), row mask (bits), pad - the wave-level pre-test
    std::vector<double> cull_rows;    // 3 per distinct parallel-sensitive direction of the whole scene (<= 32, else the pre-test is off)
    bool unbounded = false;           // some top-level item has no bounds (a plane): every pixel block can see something, k_classify has nothing to do
    bool cull_bundle = true;          // false: more than 32 distinct directions
    std::vector<double> mesh_bounds;  // 6 per mesh: model-space AABB of the source triangles (lo > hi when empty)
    int32_t csg_capacity = 0;         // per-lane hit-list entries needed (0 = scene has no CSG)
    int32_t stack_capacity = 0;       // per-lane BSP / BVH stack entries needed (0 = no trees)
    int32_t bsp_stack_capacity = 0;   // the part of it the reference-shaped BSP trees need
    int64_t bvh_nodes = 0, bvh_leaves = 0, bvh_tris = 0;   // device-side BVH additions (not part of BspMesh.compile)
    bool any_reflective = false;
    bool any_texture = false;
    bool mesh_under_csg = false;
    // Meshes whose exact BVH is built on the device after the upload (ft_bvh.hip): the flattener only reserves their ranges.
    struct BvhJob { uint32_t mesh, first_global, n, node_base, leaf_base, tri_base, wide_base, coarse_first, coarse_count; };
    std::vector<BvhJob> bvh_jobs;
};

struct SceneGraph {
    std::vector<GraphNode> nodes;
    int32_t root = -1;
    std::vector<ftd::Light> lights;
    int32_t csg_mesh_capacity = 32;
    // Non-default fast mode: ignore the `depth` of bspMesh and trace every mesh through the device-side BVH over
    // its ORIGINAL triangles.  The reference clips triangles at BSP planes; unclipped triangles give the same surface but the
    // hit arithmetic differs in the last bits, so pixels may differ at the 1e-12 level (and on silhouette ties).
    bool mesh_unclipped_bvh = false;
    // Who builds the exact BVH of top-level-Leaf meshes: true = the device (linear BVH, ft_bvh.hip; device contexts' default),
    // false = the host's recursive surface-area sweep (host-only contexts, and the fallback when a device build is refused).
    bool device_bvh = false;
    int64_t device_bvh_min_tris = 0;   // with device_bvh: meshes with fewer triangles than this get the host's SAH tree (better tree, slower build)

    bool valid(int32_t id) const { return id >= 0 && id < (int32_t)nodes.size(); }
    // Returns FT_OK or a negative ft_status with err set.
    int32_t flatten(FlatScene& out, std::string& err) const;
};

// BspMesh.compile (BspMesh.fs:51-65) on the host: appends nodes / leaves / clipped triangles to
// the flat arrays and returns the root reference (>= 0 branch node, < 0 ~leaf) via mesh.
int32_t build_bsp(const double* tris_abc, int64_t n_tris, int32_t depth, FlatScene& out, ftd::Mesh& mesh, std::string& err, bool device_bvh = false);

// Triangle.slice (Triangle.fs:24-41) exposed for the known-answer tests of the product's own builder.
int32_t slice_triangle(const double p0[3], const double n[3], const double tri[9],
                       std::vector<double>& above, std::vector<double>& below, std::string& err);

} // namespace fth
#endif
