// ft_scene.cpp — scene-graph flattening and host-side BSP build for the HIP path.
// Reference citations are relative to /root/reference/FuncTracer/.
#include "ft_scene.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

namespace fth {
namespace {

// ------------------------------------------------------------------ 4x4 matrices (Transform.fs:7-22, 47-71)
struct Mat4 { double a[16]; };

Mat4 mat_identity() { Mat4 m{}; for (int i = 0; i < 4; ++i) m.a[5 * i] = 1.0; return m; }

// Same accumulation order as Matrix (*) (Transform.fs:11-14): sum from 0 over c = 0..3.
Mat4 mat_mul(const Mat4& l, const Mat4& r) {
    Mat4 o{};
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) {
            double s = 0.0;
            for (int c = 0; c < 4; ++c) s = s + l.a[4 * row + c] * r.a[4 * c + col];
            o.a[4 * row + col] = s;
        }
    return o;
}

double vec_len(const double v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// matrix (Transform.fs:55-69) for one basic transform; `inverse` (Transform.fs:47-50) is applied by the caller.
Mat4 basic_matrix(int kind, const double v[3], double angle) {
    Mat4 m = mat_identity();
    if (kind == FT_TRANSLATE) {
        m.a[3] = v[0]; m.a[7] = v[1]; m.a[11] = v[2];
    } else if (kind == FT_SCALE) {
        m.a[0] = v[0]; m.a[5] = v[1]; m.a[10] = v[2];
    } else {
        const double ux = v[0], uy = v[1], uz = v[2];
        const double c = std::cos(angle), invc = 1.0 - c, s = std::sin(angle);
        m.a[0] = c + invc * ux * ux;      m.a[1] = invc * ux * uy - s * uz; m.a[2] = invc * ux * uz + s * uy;
        m.a[4] = invc * ux * uy + s * uz; m.a[5] = c + invc * uy * uy;      m.a[6] = invc * uy * uz - s * ux;
        m.a[8] = invc * ux * uz - s * uy; m.a[9] = invc * uy * uz + s * ux; m.a[10] = c + invc * uz * uz;
    }
    return m;
}

struct BasicXf { int kind; double v[3]; double angle; };

BasicXf canonical(const ft_transform& t) {
    BasicXf b{t.kind, {t.v[0], t.v[1], t.v[2]}, t.angle};
    if (t.kind == FT_ROTATE) {                       // Transform.rotate normalises the axis (Transform.fs:37-38, CommonTypes.fs:63-67)
        double l = vec_len(b.v);
        if (!(l < 0.0000001)) { double s = 1.0 / l; b.v[0] = s * b.v[0]; b.v[1] = s * b.v[1]; b.v[2] = s * b.v[2]; }
    }
    return b;
}
BasicXf inverse_of(const BasicXf& b) {               // Transform.fs:47-50
    BasicXf r = b;
    if (b.kind == FT_TRANSLATE) { r.v[0] = -b.v[0]; r.v[1] = -b.v[1]; r.v[2] = -b.v[2]; }
    else if (b.kind == FT_SCALE) { r.v[0] = 1.0 / b.v[0]; r.v[1] = 1.0 / b.v[1]; r.v[2] = 1.0 / b.v[2]; }
    else r.angle = -b.angle;
    return r;
}
// matrix of a (possibly Composed) transform level and of its inverse (Transform.fs:51, 70-71):
// Composed [t1..tn] = M_n * (... * (M_1 * I)); inverse = Composed (rev (map inverse)).
void level_matrices(const std::vector<ft_transform>& ts, Mat4& m2w, Mat4& w2m) {
    std::vector<BasicXf> bs;
    for (auto& t : ts) bs.push_back(canonical(t));
    if (bs.size() == 1) {
        m2w = basic_matrix(bs[0].kind, bs[0].v, bs[0].angle);
        BasicXf inv = inverse_of(bs[0]);
        w2m = basic_matrix(inv.kind, inv.v, inv.angle);
        return;
    }
    m2w = mat_identity();
    for (size_t i = 0; i < bs.size(); ++i) m2w = mat_mul(basic_matrix(bs[i].kind, bs[i].v, bs[i].angle), m2w);
    w2m = mat_identity();
    for (size_t i = bs.size(); i-- > 0;) { BasicXf inv = inverse_of(bs[i]); w2m = mat_mul(basic_matrix(inv.kind, inv.v, inv.angle), w2m); }
}

// ------------------------------------------------------------------ flatten
struct MatFn { int kind; const GraphNode* node; };   // GraphNode::MaterialF / HueShift / IgnoreLight / Texture

struct WalkCtx {
    Mat4 m2w = mat_identity(), w2m = mat_identity();
    bool xform = false;
    bool bounds = true;                              // false under operand B of subtract / intersect: the result lies within operand A
    std::vector<MatFn> fns;                          // outermost first
};

struct Flattener {
    const SceneGraph& g;
    FlatScene& out;
    std::string& err;
    std::map<int32_t, uint32_t> mesh_of_node;
    int32_t status = FT_OK;
    int csg_depth = 0, max_csg_depth = 0;
    int cur_list = 0, max_list = 0;                  // static bound on the per-lane hit-list length
    std::vector<bool> leaf_bounds;                   // per leaf: its box is part of its item's bounds (WalkCtx::bounds)
    std::map<const GraphNode*, size_t> image_base;   // image texture node -> offset of its pixels in out.tex_pixels

    Flattener(const SceneGraph& g_, FlatScene& o, std::string& e) : g(g_), out(o), err(e) {}

    uint32_t resolve_material(const WalkCtx& c) {
        // Apply the enclosing scene functions innermost first; the outermost Material wins and resets
        // applyLighting because it replaces the whole record (Ray.fs:47-59, SceneParser.fs:100-105).
        ftd::Material m{};
        m.colour[0] = m.colour[1] = m.colour[2] = 1.0;                       // Ray.mattWhite, Ray.fs:11
        m.apply_lighting = 1; m.texture = -1; m.hue_rot = 0;
        for (size_t i = c.fns.size(); i-- > 0;) {
            const MatFn& f = c.fns[i];
            if (f.kind == GraphNode::MaterialF) {
                const ft_material& s = f.node->mat;
                m.colour[0] = s.colour[0]; m.colour[1] = s.colour[1]; m.colour[2] = s.colour[2];
                m.roughness = s.roughness; m.reflectance = s.reflectance; m.shineyness = s.shineyness;
                m.apply_lighting = s.apply_lighting ? 1u : 0u; m.texture = -1; m.hue_rot = 0;
            } else if (f.kind == GraphNode::IgnoreLight) {
                m.apply_lighting = 0;
            } else if (f.kind == GraphNode::HueShift) {                       // Colour.hueShift: (r,g,b) -> (b,r,g), CommonTypes.fs:90
                if (m.texture >= 0) m.hue_rot = (m.hue_rot + 1) % 3;
                else { double r = m.colour[0], gg = m.colour[1], b = m.colour[2]; m.colour[0] = b; m.colour[1] = r; m.colour[2] = gg; }
            } else if (f.kind == GraphNode::Texture) {
                ftd::Texture t{};
                for (int a = 0; a < 3; ++a) { t.c1[a] = f.node->ca[a]; t.c2[a] = f.node->cb[a]; }
                if (!f.node->pixels.empty()) {                               // ImageTexture.image (Textures/Image.fs:20-36)
                    auto it = image_base.find(f.node);
                    if (it == image_base.end()) {
                        it = image_base.emplace(f.node, out.tex_pixels.size()).first;
                        out.tex_pixels.insert(out.tex_pixels.end(), f.node->pixels.begin(), f.node->pixels.end());
                    }
                    t.kind = 1.0; t.pixel_base = (double)it->second;
                    t.c1[0] = (double)f.node->img_w; t.c1[1] = (double)f.node->img_h; t.c1[2] = 0.0;
                    t.c2[0] = t.c2[1] = t.c2[2] = 0.0;
                }
                size_t n_ops = f.node->uv_ops.size() / 3;
                if (n_ops > (size_t)ftd::kMaxUvOps) { status = FT_ERR_UNSUPPORTED; err = "more than 13 nested texture functions on one texture"; n_ops = ftd::kMaxUvOps; }   // flatten fails; keep the record well-formed
                t.n_ops = (double)n_ops;
                for (size_t k = 0; k < n_ops; ++k) {
                    const double kind = f.node->uv_ops[3 * k], a = f.node->uv_ops[3 * k + 1], b = f.node->uv_ops[3 * k + 2];
                    t.ops[k][0] = kind;
                    if (kind == 0.0) { t.ops[k][1] = a; t.ops[k][2] = b; }
                    else { t.ops[k][1] = std::cos(a); t.ops[k][2] = std::sin(a); }   // matrix (rotate (0,1,0) angle), Transform.fs:60-69
                }
                size_t idx = 0;
                for (; idx < out.textures.size(); ++idx) if (std::memcmp(&out.textures[idx], &t, sizeof t) == 0) break;
                if (idx == out.textures.size()) out.textures.push_back(t);
                m.texture = (int32_t)idx; m.hue_rot = 0;
            }
        }
        if (m.reflectance > 0.0 && m.apply_lighting) out.any_reflective = true;
        for (size_t i = 0; i < out.materials.size(); ++i)
            if (std::memcmp(&out.materials[i], &m, sizeof m) == 0) return (uint32_t)i;
        out.materials.push_back(m);
        return (uint32_t)out.materials.size() - 1;
    }

    void emit_leaf(uint32_t kind, uint32_t mesh, const WalkCtx& c, bool flip, bool in_csg, int max_hits) {
        ftd::Leaf L{};
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) L.w2m[4 * r + k] = c.w2m.a[4 * r + k];
        L.kind = kind; L.mesh = mesh;
        L.material = resolve_material(c);
        L.flags = (flip ? ftd::LF_FLIP : 0u) | (c.xform ? ftd::LF_XFORM : 0u) | (out.materials[L.material].apply_lighting ? ftd::LF_LIT : 0u);
        uint32_t id = (uint32_t)out.leaves.size();
        out.leaves.push_back(L);
        leaf_bounds.push_back(c.bounds);
        for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) out.m2w.push_back(c.m2w.a[4 * r + k]);
        out.program.push_back(ftd::make_op(in_csg ? ftd::OP_LEAF_PUSH : ftd::OP_LEAF_FOLD, id));
        if (in_csg) { cur_list += max_hits; if (cur_list > max_list) max_list = cur_list; }
    }

    uint32_t mesh_for(int32_t node_id, const double* tris, int64_t n, int32_t depth) {
        if (node_id >= 0) { auto it = mesh_of_node.find(node_id); if (it != mesh_of_node.end()) return it->second; }
        ftd::Mesh m{};
        int32_t rc = build_bsp(tris, n, depth, out, m, err, g.device_bvh && n >= g.device_bvh_min_tris);
        if (rc != FT_OK) { status = rc; return 0; }
        out.meshes.push_back(m);
        {
            const double inf = std::numeric_limits<double>::infinity();
            double b[6] = {inf, inf, inf, -inf, -inf, -inf};
            for (int64_t i = 0; i < 3 * n; ++i) for (int a = 0; a < 3; ++a) { double v = tris[3 * i + a]; if (v < b[a]) b[a] = v; if (v > b[3 + a]) b[3 + a] = v; }
            out.mesh_bounds.insert(out.mesh_bounds.end(), b, b + 6);
        }
        uint32_t idx = (uint32_t)out.meshes.size() - 1;
        if (node_id >= 0) mesh_of_node[node_id] = idx;
        if ((int32_t)m.max_depth + 1 > out.stack_capacity && m.root >= 0) out.stack_capacity = (int32_t)m.max_depth + 1;
        if ((int32_t)m.max_depth + 1 > out.bsp_stack_capacity && m.root >= 0) out.bsp_stack_capacity = (int32_t)m.max_depth + 1;
        return idx;
    }

    // ---- top-level item culling -------------------------------------------------------------
    struct ItemMark { size_t prog_at, leaf_at; bool open; };
    ItemMark begin_item(bool in_csg) {
        if (in_csg) return {0, 0, false};
        ItemMark m{out.program.size(), out.leaves.size(), true};
        out.program.push_back(ftd::make_op(ftd::OP_CULL, 0));
        out.program.push_back(0);
        return m;
    }
    bool model_box(const ftd::Leaf& L, double lo[3], double hi[3]) const {
        switch (L.kind) {
            case ftd::LK_SPHERE: lo[0] = lo[1] = lo[2] = -1; hi[0] = hi[1] = hi[2] = 1; return true;
            case ftd::LK_SQUARE: lo[0] = 0; lo[1] = 0; lo[2] = 0; hi[0] = 1; hi[1] = 0; hi[2] = 1; return true;
            case ftd::LK_CIRCLE: lo[0] = -1; lo[1] = 0; lo[2] = -1; hi[0] = 1; hi[1] = 0; hi[2] = 1; return true;
            case ftd::LK_CUBE: lo[0] = lo[1] = lo[2] = -0.5; hi[0] = hi[1] = hi[2] = 0.5; return true;
            case ftd::LK_CONE: case ftd::LK_CYLINDER: case ftd::LK_SOLIDCYL: lo[0] = -1; lo[1] = 0; lo[2] = -1; hi[0] = 1; hi[1] = 1; hi[2] = 1; return true;
            case ftd::LK_MESH: {
                const double* b = &out.mesh_bounds[6 * (size_t)L.mesh];
                for (int a = 0; a < 3; ++a) { lo[a] = b[a]; hi[a] = b[3 + a]; }
                return lo[0] <= hi[0];
            }
            default: return false;                                  // LK_PLANE: unbounded
        }
    }
    void unbounded_item(size_t prog_at, size_t leaf_at) {
        out.item_pc.push_back((uint32_t)prog_at);
        ftd::CullRecord never{};                                    // cull records are indexed by item: this one can never report a miss
        never.radius2 = std::numeric_limits<double>::infinity();
        out.culls.push_back(never);
        const float rec[8] = {0.f, 0.f, 0.f, std::numeric_limits<float>::infinity(), 0.f, 0.f, 0.f, 0.f};
        (void)leaf_at;
        out.unbounded = true;
        out.cull_items.insert(out.cull_items.end(), rec, rec + 8);
    }
    void end_item(const ItemMark& m) {
        if (!m.open) return;
        auto drop = [&]() { out.program.erase(out.program.begin() + (long)m.prog_at, out.program.begin() + (long)m.prog_at + 2); if (out.program.size() > m.prog_at) unbounded_item(m.prog_at, m.leaf_at); };
        if (status != FT_OK || out.leaves.size() == m.leaf_at) { drop(); return; }
        const double inf = std::numeric_limits<double>::infinity();
        double blo[3] = {inf, inf, inf}, bhi[3] = {-inf, -inf, -inf};
        std::vector<std::array<double, 3>> pts, rows;
        for (size_t li = m.leaf_at; li < out.leaves.size(); ++li) {
            const ftd::Leaf& L = out.leaves[li];
            double lo[3], hi[3];
            // A - B and A & B lie within A (Csg.fs:27-44 never keep a B hit outside A), so operand B does not widen the item's
            // bounds; its face directions still count (a parallel-ray hit of Plane.fs:13-16 can sit anywhere).
            const bool bounding = leaf_bounds[li];
            if (bounding && !model_box(L, lo, hi)) { drop(); return; }
            const double* W = &out.m2w[12 * li];
            for (int corner = 0; bounding && corner < 8; ++corner) {
                const double x = (corner & 1) ? hi[0] : lo[0], y = (corner & 2) ? hi[1] : lo[1], z = (corner & 4) ? hi[2] : lo[2];
                std::array<double, 3> q = {W[0] * x + W[1] * y + W[2] * z + W[3], W[4] * x + W[5] * y + W[6] * z + W[7], W[8] * x + W[9] * y + W[10] * z + W[11]};
                for (int a = 0; a < 3; ++a) { if (!(std::fabs(q[a]) < 1e300)) { drop(); return; } if (q[a] < blo[a]) blo[a] = q[a]; if (q[a] > bhi[a]) bhi[a] = q[a]; }
                pts.push_back(q);
            }
            auto add_row = [&](int r) {                             // world-space vector whose dot with d is a plane denominator of this leaf
                std::array<double, 3> v = {L.w2m[4 * r], L.w2m[4 * r + 1], L.w2m[4 * r + 2]};
                for (auto& e : rows) if (e == v) return;
                rows.push_back(v);
            };
            if (L.kind == ftd::LK_SQUARE || L.kind == ftd::LK_CIRCLE || L.kind == ftd::LK_SOLIDCYL || L.kind == ftd::LK_PLANE) add_row(1);
            if (L.kind == ftd::LK_CUBE) { add_row(0); add_row(1); add_row(2); }
        }
        // More face directions than a CullRecord holds: no per-ray OP_CULL for this item, but the wave-level tests (which keep the
        // directions in a scene-wide table and a mask per item) still get its bounding sphere.
        const bool exact = rows.size() <= 6;
        if (!exact) out.program.erase(out.program.begin() + (long)m.prog_at, out.program.begin() + (long)m.prog_at + 2);
        ftd::CullRecord R{};
        double r2 = 0.0;
        for (int a = 0; a < 3; ++a) R.centre[a] = 0.5 * (blo[a] + bhi[a]);
        for (auto& q : pts) { double d2 = 0; for (int a = 0; a < 3; ++a) d2 += (q[a] - R.centre[a]) * (q[a] - R.centre[a]); if (d2 > r2) r2 = d2; }
        const double r = std::sqrt(r2) * (1.0 + 1e-6) + 1e-9;
        R.radius2 = exact ? r * r : inf;                            // inf: the per-ray test (were it asked) can never report a miss
        R.n_rows = exact ? (double)rows.size() : 0.0;
        for (size_t k = 0; exact && k < rows.size(); ++k) for (int a = 0; a < 3; ++a) R.rows[k][a] = rows[k][a];
        out.culls.push_back(R);
        {   // compact copy for the wave-level pre-test (k_closest / k_shade, bundle_cull)
            uint32_t mask = 0;
            for (auto& v : rows) {
                size_t k = 0;
                for (; k < out.cull_rows.size() / 3; ++k) if (out.cull_rows[3 * k] == v[0] && out.cull_rows[3 * k + 1] == v[1] && out.cull_rows[3 * k + 2] == v[2]) break;
                if (k == out.cull_rows.size() / 3) { if (k >= 32) { out.cull_bundle = false; break; } out.cull_rows.insert(out.cull_rows.end(), v.begin(), v.end()); }
                mask |= 1u << k;
            }
            float rf = (float)r; while ((double)rf < r) rf = std::nextafter(rf, std::numeric_limits<float>::infinity());
            float bits; std::memcpy(&bits, &mask, 4);
            float rec[8] = {(float)R.centre[0], (float)R.centre[1], (float)R.centre[2], rf, bits, 0.f, 0.f, 0.f};
            if (out.leaves.size() == m.leaf_at + 1 && out.leaves[m.leaf_at].kind == ftd::LK_MESH) {   // a bare mesh: [5] first coarse box, [6] count, [7] its leaf
                const uint32_t mesh = out.leaves[m.leaf_at].mesh, w[3] = {out.mesh_coarse[2 * mesh], out.mesh_coarse[2 * mesh + 1], (uint32_t)m.leaf_at};
                std::memcpy(&rec[5], w, sizeof w);
            }
            out.cull_items.insert(out.cull_items.end(), rec, rec + 8);
            out.item_pc.push_back((uint32_t)m.prog_at | (exact ? 0x80000000u : 0u));   // top bit: the item starts with its OP_CULL pair
        }
        if (!exact) return;
        out.program[m.prog_at] = ftd::make_op(ftd::OP_CULL, (uint32_t)out.culls.size() - 1);
        out.program[m.prog_at + 1] = (uint32_t)(out.program.size() - (m.prog_at + 2));
    }

    // A solid every line crosses an even number of times: sphere, cube, solidCylinder and their CSG / group combinations.  Only then
    // does A - B (A & B) lie within A for the purposes of the item bounds: the reference decides "inside A" by counting crossings
    // along the whole LINE (Csg.fs:81-93, hits at negative t included), so a flat or open operand A (square, circle, cone,
    // cylinder, triangles) crossed once BEHIND the ray origin leaves the ray "inside A" and operand B's hits far from A are kept.
    bool closed_solid(int32_t id) const {
        const GraphNode& n = g.nodes[id];
        switch (n.kind) {
            case GraphNode::Prim: return n.prim == FT_PRIM_SPHERE || n.prim == FT_PRIM_CUBE || n.prim == FT_PRIM_SOLID_CYLINDER;
            case GraphNode::Transform: case GraphNode::MaterialF: case GraphNode::HueShift: case GraphNode::IgnoreLight: case GraphNode::Texture: return closed_solid(n.children[0]);
            case GraphNode::Group: case GraphNode::Csg: { for (int32_t ch : n.children) if (!closed_solid(ch)) return false; return !n.children.empty(); }
            default: return false;                                  // triangles, meshes
        }
    }

    // A primitive (not a mesh) under any chain of single-child scene functions: one leaf with at most a few hits.
    bool bare_primitive(int32_t id) const {
        for (;;) {
            const GraphNode& n = g.nodes[id];
            switch (n.kind) {
                case GraphNode::Prim: return true;
                case GraphNode::Transform: case GraphNode::MaterialF: case GraphNode::HueShift: case GraphNode::IgnoreLight: case GraphNode::Texture: id = n.children[0]; break;
                default: return false;
            }
        }
    }

    void walk(int32_t id, const WalkCtx& c, bool in_csg) {
        if (status != FT_OK) return;
        const GraphNode& n = g.nodes[id];
        switch (n.kind) {
            case GraphNode::Prim: {
                static const uint32_t kind_of[8] = {ftd::LK_CIRCLE, ftd::LK_SQUARE, ftd::LK_CUBE, ftd::LK_SPHERE, ftd::LK_PLANE, ftd::LK_CONE, ftd::LK_SOLIDCYL, ftd::LK_CYLINDER};
                static const int max_hits[8] = {1, 1, 6, 2, 1, 2, 4, 2};
                ItemMark im = begin_item(in_csg);
                emit_leaf(kind_of[n.prim], 0, c, false, in_csg, max_hits[n.prim]);
                end_item(im);
                break;
            }
            case GraphNode::TriangleP: {
                uint32_t mesh = mesh_for(-1, n.tri, 1, 0);
                ItemMark im = begin_item(in_csg);
                if (status == FT_OK) emit_leaf(ftd::LK_MESH, mesh, c, false, in_csg, 1);
                end_item(im);
                break;
            }
            case GraphNode::Mesh: {
                uint32_t mesh = mesh_for(id, n.tris.data(), (int64_t)(n.tris.size() / 9), g.mesh_unclipped_bvh ? 0 : n.depth);
                ItemMark im = begin_item(in_csg);
                if (status == FT_OK) {
                    if (in_csg) out.mesh_under_csg = true;
                    emit_leaf(ftd::LK_MESH, mesh, c, false, in_csg, g.csg_mesh_capacity);
                }
                end_item(im);
                break;
            }
            case GraphNode::Transform: {
                Mat4 m, w;
                level_matrices(n.xf, m, w);
                WalkCtx c2 = c;
                c2.m2w = mat_mul(c.m2w, m);          // outer levels act last on points: M_outer * M_inner
                c2.w2m = mat_mul(w, c.w2m);          // and first on rays:              W_inner * W_outer
                c2.xform = true;
                walk(n.children[0], c2, in_csg);
                break;
            }
            case GraphNode::MaterialF: case GraphNode::HueShift: case GraphNode::IgnoreLight: case GraphNode::Texture: {
                WalkCtx c2 = c;
                c2.fns.push_back({(int)n.kind, &n});
                walk(n.children[0], c2, in_csg);
                break;
            }
            case GraphNode::Group: {
                // Runs of bare Triangle primitives (the `mesh` keyword, SceneParser.fs:116-126) become one
                // brute-force triangle list: identical hit sequence, one leaf instead of thousands.
                size_t i = 0;
                while (i < n.children.size() && status == FT_OK) {
                    const GraphNode& ch = g.nodes[n.children[i]];
                    if (ch.kind == GraphNode::TriangleP) {
                        std::vector<double> run;
                        while (i < n.children.size() && g.nodes[n.children[i]].kind == GraphNode::TriangleP) {
                            const double* t = g.nodes[n.children[i]].tri; run.insert(run.end(), t, t + 9); ++i;
                        }
                        uint32_t mesh = mesh_for(-1, run.data(), (int64_t)(run.size() / 9), 0);
                        ItemMark im = begin_item(in_csg);
                        if (status == FT_OK) emit_leaf(ftd::LK_MESH, mesh, c, false, in_csg, (int)std::min<size_t>(run.size() / 9, (size_t)g.csg_mesh_capacity));
                        end_item(im);
                    } else {
                        walk(n.children[i], c, in_csg); ++i;
                    }
                }
                break;
            }
            case GraphNode::Csg: {
                ++csg_depth; if (csg_depth > max_csg_depth) max_csg_depth = csg_depth;
                int before = cur_list;
                ItemMark im = begin_item(in_csg);
                const bool pair = bare_primitive(n.children[0]) && bare_primitive(n.children[1]);
                const size_t pair_at = out.program.size(), first_leaf = out.leaves.size();
                if (pair) { out.program.push_back(0); out.program.push_back(0); out.program.push_back(0); }
                out.program.push_back(ftd::make_op(ftd::OP_MARK, 0));
                walk(n.children[0], c, true);
                const bool a_gates = n.op == FT_CSG_SUBTRACT || n.op == FT_CSG_INTERSECT;   // no A hit => empty result
                WalkCtx cb = c;
                if (a_gates && closed_solid(n.children[0])) cb.bounds = false;
                const size_t skip_at = out.program.size();
                if (a_gates) out.program.push_back(ftd::make_op(ftd::OP_SKIP_IF_EMPTY, 0));
                out.program.push_back(ftd::make_op(ftd::OP_MARK, 0));
                walk(n.children[1], cb, true);
                out.program.push_back(ftd::make_op(ftd::OP_CSG, (uint32_t)n.op));
                if (a_gates) out.program[skip_at] = ftd::make_op(ftd::OP_SKIP_IF_EMPTY, (uint32_t)(out.program.size() - skip_at - 1));
                --csg_depth;
                if (!in_csg) { out.program.push_back(ftd::make_op(ftd::OP_FOLD_LIST, 0)); cur_list = before; }
                if (pair && status == FT_OK && out.leaves.size() == first_leaf + 2) {
                    out.program[pair_at] = ftd::make_op(ftd::OP_CSG_PAIR, (uint32_t)(out.program.size() - (pair_at + 3)));
                    out.program[pair_at + 1] = (uint32_t)first_leaf;
                    out.program[pair_at + 2] = (uint32_t)(first_leaf + 1) | ((uint32_t)n.op << 24) | (in_csg ? 0u : 1u << 26);
                } else if (pair) {
                    out.program.erase(out.program.begin() + (long)pair_at, out.program.begin() + (long)pair_at + 3);
                }
                end_item(im);
                break;
            }
        }
    }
};

} // namespace

int32_t SceneGraph::flatten(FlatScene& out, std::string& err) const {
    out = FlatScene();
    if (!valid(root)) { err = "scene has no objects (ft_scene_set_objects)"; return FT_ERR_STATE; }
    Flattener f(*this, out, err);
    WalkCtx c;
    f.walk(root, c, false);
    if (f.status != FT_OK) return f.status;
    out.item_pc.push_back((uint32_t)out.program.size());
    out.program.push_back(ftd::make_op(ftd::OP_END, 0));
    out.lights = lights;
    out.csg_capacity = f.max_list;
    if (f.max_csg_depth > 8) { err = "CSG nesting deeper than 8 levels is not supported on the device path"; return FT_ERR_UNSUPPORTED; }
    if (out.csg_capacity > 255) { err = "a CSG subtree can produce more than 255 hits per ray; lower csg_mesh_capacity"; return FT_ERR_UNSUPPORTED; }
    if (out.leaves.size() > ftd::ID_LEAF_MASK) { err = "too many primitive instances"; return FT_ERR_UNSUPPORTED; }
    if (out.textures.empty()) out.textures.push_back(ftd::Texture{});
    for (auto& l : out.lights) if (l.kind == ftd::LT_SOFT) {
        l.tan_half_scatter = std::tan(l.scatter / 2.0);
        if (l.samples > 255) { err = "softdirectional lights with more than 255 samples are not supported on the device path"; return FT_ERR_UNSUPPORTED; }
    }
    // (Measured in round 3: shading the lights sixteen at a time in a loop, with the sums carried in light order across each round's shadow
    //  traces, costs every kernel variant 52 - 136 bytes of scratch per lane - k_primary<F,F,T,4> 56 -> 108, k_bounce<F,F,F> 0 -> 72 - whether
    //  or not a scene has a seventeenth light.  No scene of the reference has more than three.)
    if (out.lights.size() > 16) { err = "more than 16 lights are not supported on the device path"; return FT_ERR_UNSUPPORTED; }
    if (out.tris.empty()) { out.tris.assign(9, 0.0); out.tri_orig.assign(1, 0u); }   // keep device pointers non-null
    if (out.culls.empty()) out.culls.push_back(ftd::CullRecord{});
    return FT_OK;
}

// ====================================================================== BSP build (host)
namespace {

struct P3 { double x, y, z; };
struct Tri3 { P3 a, b, c; };
struct SplitPlane { P3 p0, n; };

inline double dot3(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline P3 diff(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

inline bool above_plane(const SplitPlane& pl, P3 q) { return dot3(diff(q, pl.p0), pl.n) >= 0.0; }   // Plane.isAbove, Plane.fs:22-23

// Triangle.edgeIntersection (Triangle.fs:8-10): Plane.intersect (Plane.fs:9-20) along the normalised edge.
bool edge_cut(const SplitPlane& pl, P3 from, P3 to, P3& out) {
    P3 d = diff(to, from);
    double len = std::sqrt(dot3(d, d));
    if (!(len < 0.0000001)) { double s = 1.0 / len; d = {s * d.x, s * d.y, s * d.z}; }                // CommonTypes.fs:63-67
    const double eps = 0.0000001;
    double num = dot3(diff(pl.p0, from), pl.n);
    double den = dot3(d, pl.n);
    if (std::fabs(den) < eps) { if (num < eps) { out = from; return true; } return false; }
    double t = num / den;
    out = {from.x + t * d.x, from.y + t * d.y, from.z + t * d.z};
    return true;
}

// Triangle.slice' (Triangle.fs:13-22): `lone` is alone on its side; outputs one piece on the lone side
// and two on the other, winding preserved.
bool cut_lone(const SplitPlane& pl, P3 lone, P3 q, P3 r, std::vector<Tri3>& lone_side, std::vector<Tri3>& pair_side) {
    P3 lq, lr, ql, rl;
    if (!edge_cut(pl, lone, q, lq) || !edge_cut(pl, lone, r, lr) || !edge_cut(pl, q, lone, ql) || !edge_cut(pl, r, lone, rl)) return false;
    lone_side.push_back({lone, lq, lr});
    pair_side.push_back({ql, q, r});
    pair_side.push_back({r, rl, ql});
    return true;
}

// Triangle.slice (Triangle.fs:24-41).
bool split_triangle(const SplitPlane& pl, const Tri3& t, std::vector<Tri3>& above, std::vector<Tri3>& below) {
    const bool ua = above_plane(pl, t.a), ub = above_plane(pl, t.b), uc = above_plane(pl, t.c);
    if (ua == ub && ub == uc) { (ua ? above : below).push_back(t); return true; }
    if (ua == ub) return cut_lone(pl, t.c, t.a, t.b, uc ? above : below, uc ? below : above);   // c is alone
    if (ua == uc) return cut_lone(pl, t.b, t.c, t.a, ub ? above : below, ub ? below : above);   // b is alone
    return cut_lone(pl, t.a, t.b, t.c, ua ? above : below, ua ? below : above);                 // a is alone
}

struct BspBuilder {
    FlatScene& out;
    std::string& err;
    uint32_t max_depth = 0;
    bool failed = false;

    int32_t make_leaf(const std::vector<Tri3>& ts) {
        ftd::BspLeaf L{(uint32_t)(out.tris.size() / 9), (uint32_t)ts.size()};
        for (auto& t : ts) {                                                    // v0, edge1, edge2 (Triangle.fs:45-46)
            const double rec[9] = {t.a.x, t.a.y, t.a.z, t.b.x - t.a.x, t.b.y - t.a.y, t.b.z - t.a.z, t.c.x - t.a.x, t.c.y - t.a.y, t.c.z - t.a.z};
            out.tri_orig.push_back((uint32_t)(out.tris.size() / 9));
            out.tris.insert(out.tris.end(), rec, rec + 9);
        }
        out.bsp_leaves.push_back(L);
        return ~(int32_t)(out.bsp_leaves.size() - 1);
    }

    // ---- device-side BVH over the triangles of a top-level Leaf (see ft_flat.h) -----------------
    struct Box { double lo[3], hi[3]; };
    static constexpr size_t kBvhLeafTris = 4;
    uint32_t bvh_depth = 0;
    int32_t bvh_build(const std::vector<Tri3>& ts, const std::vector<Box>& boxes, std::vector<uint32_t>& idx, size_t lo, size_t hi,
                      uint32_t first_global, double pad, uint32_t level) {
        if (level + 1 > bvh_depth) bvh_depth = level + 1;
        if (hi - lo <= kBvhLeafTris) {                                          // leaf: a reordered copy of the triangles + their list indices
            ftd::BspLeaf L{(uint32_t)(out.tris.size() / 9), (uint32_t)(hi - lo)};
            for (size_t k = lo; k < hi; ++k) {
                const Tri3& t = ts[idx[k]];
                const double rec[9] = {t.a.x, t.a.y, t.a.z, t.b.x - t.a.x, t.b.y - t.a.y, t.b.z - t.a.z, t.c.x - t.a.x, t.c.y - t.a.y, t.c.z - t.a.z};
                out.tri_orig.push_back(first_global + idx[k]);
                out.tris.insert(out.tris.end(), rec, rec + 9);
            }
            out.bsp_leaves.push_back(L);
            Box lb{{1e308, 1e308, 1e308}, {-1e308, -1e308, -1e308}};            // the leaf's own (inflated) box, for the 4-wide nodes
            for (size_t k = lo; k < hi; ++k) for (int a = 0; a < 3; ++a) { lb.lo[a] = std::min(lb.lo[a], boxes[idx[k]].lo[a] - pad); lb.hi[a] = std::max(lb.hi[a], boxes[idx[k]].hi[a] + pad); }
            leaf_box[out.bsp_leaves.size() - 1] = lb;
            return ~(int32_t)(out.bsp_leaves.size() - 1);
        }
        const double inf = std::numeric_limits<double>::infinity();
        Box b{{inf, inf, inf}, {-inf, -inf, -inf}}, cb = b;
        for (size_t k = lo; k < hi; ++k) {
            const Box& t = boxes[idx[k]];
            for (int a = 0; a < 3; ++a) {
                if (t.lo[a] < b.lo[a]) b.lo[a] = t.lo[a];
                if (t.hi[a] > b.hi[a]) b.hi[a] = t.hi[a];
                const double c = 0.5 * (t.lo[a] + t.hi[a]);
                if (c < cb.lo[a]) cb.lo[a] = c;
                if (c > cb.hi[a]) cb.hi[a] = c;
            }
        }
        // Split by the surface-area heuristic, swept exactly: along each axis the triangles are ordered by centroid and every cut is
        // priced as area(left) x count(left) + area(right) x count(right); ties and degenerate sweeps fall back to the median of the
        // widest axis.  (The tree only decides which boxes a ray looks into: no pixel depends on it.)
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        size_t mid = (lo + hi) / 2;
        auto by_centroid = [&](int ax) {
            return [&boxes, ax](uint32_t x, uint32_t y) {
                const double cx = boxes[x].lo[ax] + boxes[x].hi[ax], cy = boxes[y].lo[ax] + boxes[y].hi[ax];
                return cx < cy || (cx == cy && x < y);
            };
        };
        if (hi - lo <= 65536) {                                    // above that a level's three sorts cost more than its cut returns: medians
            const size_t n = hi - lo;
            auto area = [](const Box& q) { const double dx = q.hi[0] - q.lo[0], dy = q.hi[1] - q.lo[1], dz = q.hi[2] - q.lo[2]; return dx * dy + dy * dz + dz * dx; };
            double best = std::numeric_limits<double>::infinity();
            int best_axis = -1; size_t best_cut = 0;
            std::vector<double> right_area(n);
            std::vector<uint32_t> order(idx.begin() + (long)lo, idx.begin() + (long)hi);
            for (int ax = 0; ax < 3; ++ax) {
                std::sort(order.begin(), order.end(), by_centroid(ax));
                Box acc{{inf, inf, inf}, {-inf, -inf, -inf}};
                for (size_t k = n; k-- > 1;) {
                    for (int a = 0; a < 3; ++a) { acc.lo[a] = std::min(acc.lo[a], boxes[order[k]].lo[a]); acc.hi[a] = std::max(acc.hi[a], boxes[order[k]].hi[a]); }
                    right_area[k] = area(acc);
                }
                acc = Box{{inf, inf, inf}, {-inf, -inf, -inf}};
                for (size_t k = 1; k < n; ++k) {                    // cut before element k: [0, k) | [k, n)
                    for (int a = 0; a < 3; ++a) { acc.lo[a] = std::min(acc.lo[a], boxes[order[k - 1]].lo[a]); acc.hi[a] = std::max(acc.hi[a], boxes[order[k - 1]].hi[a]); }
                    const double cost = area(acc) * (double)k + right_area[k] * (double)(n - k);
                    if (cost < best) { best = cost; best_axis = ax; best_cut = k; }
                }
            }
            if (best_axis >= 0 && std::isfinite(best)) { axis = best_axis; mid = lo + best_cut; }
            if (level > 24) mid = (lo + hi) / 2;                    // a lopsided sweep must not outgrow the walkers' stacks (40 levels): medians from here on
        }
        std::sort(idx.begin() + (long)lo, idx.begin() + (long)hi, by_centroid(axis));
        const int32_t node = (int32_t)out.nodes.size();
        out.nodes.push_back(ftd::BspNode{});
        const int32_t l = bvh_build(ts, boxes, idx, lo, mid, first_global, pad, level + 1);
        const int32_t r = bvh_build(ts, boxes, idx, mid, hi, first_global, pad, level + 1);
        ftd::BspNode& nd = out.nodes[(size_t)node];
        for (int a = 0; a < 3; ++a) { nd.bmin[a] = b.lo[a] - pad; nd.bmax[a] = b.hi[a] + pad; }   // inflated: pruning can never drop a real hit
        nd.left = l; nd.right = r; nd.axis = (uint32_t)axis;
        return node;
    }
    // Collapse the binary BVH two levels at a time into 4-wide nodes whose child boxes sit in the parent (ft_flat.h).
    std::map<size_t, Box> leaf_box;
    Box box_of(int32_t child) const {
        if (child < 0) return leaf_box.at((size_t)~child);
        const ftd::BspNode& nd = out.nodes[(size_t)child];
        return Box{{nd.bmin[0], nd.bmin[1], nd.bmin[2]}, {nd.bmax[0], nd.bmax[1], nd.bmax[2]}};
    }
    int32_t widen(int32_t n) {
        const size_t at = out.wide.size();
        out.wide.resize(at + ftd::kWideNodeDoubles, 0.0);
        int32_t child[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        Box box[4] = {};
        uint32_t axes = out.nodes[(size_t)n].axis;
        const int32_t halves[2] = {out.nodes[(size_t)n].left, out.nodes[(size_t)n].right};
        for (int h = 0; h < 2; ++h) {
            const int32_t c = halves[h];
            if (c < 0) { child[2 * h] = c; box[2 * h] = box_of(c); continue; }              // a leaf takes one slot of its half
            axes |= out.nodes[(size_t)c].axis << (8 * (h + 1));
            const int32_t gk[2] = {out.nodes[(size_t)c].left, out.nodes[(size_t)c].right};
            for (int k = 0; k < 2; ++k) { box[2 * h + k] = box_of(gk[k]); child[2 * h + k] = gk[k] < 0 ? gk[k] : widen(gk[k]); }
        }
        double* w = &out.wide[at];
        const double absent = std::numeric_limits<double>::quiet_NaN();   // an empty slot's box fails every slab test by itself (ft_flat.h)
        for (int c = 0; c < 4; ++c) for (int a = 0; a < 3; ++a) { w[6 * c + a] = child[c] == INT32_MIN ? absent : box[c].lo[a]; w[6 * c + 3 + a] = child[c] == INT32_MIN ? absent : box[c].hi[a]; }
        std::memcpy(w + 24, child, sizeof child);
        std::memcpy(w + 26, &axes, sizeof axes);
        return (int32_t)(at / ftd::kWideNodeDoubles);
    }
    int32_t bvh_for_leaf(const std::vector<Tri3>& ts, uint32_t first_global) {
        if (ts.size() < 8) return INT32_MIN;
        std::vector<Box> boxes(ts.size());
        double extent = 0.0;
        for (size_t i = 0; i < ts.size(); ++i) {
            const P3 p[3] = {ts[i].a, ts[i].b, ts[i].c};
            Box& b = boxes[i];
            for (int a = 0; a < 3; ++a) { b.lo[a] = std::numeric_limits<double>::infinity(); b.hi[a] = -b.lo[a]; }
            for (auto& q : p) {
                const double v[3] = {q.x, q.y, q.z};
                for (int a = 0; a < 3; ++a) { if (!(std::fabs(v[a]) < 1e300)) return INT32_MIN; if (v[a] < b.lo[a]) b.lo[a] = v[a]; if (v[a] > b.hi[a]) b.hi[a] = v[a]; if (std::fabs(v[a]) > extent) extent = std::fabs(v[a]); }
            }
        }
        std::vector<uint32_t> idx(ts.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
        return bvh_build(ts, boxes, idx, 0, ts.size(), first_global, 1e-7 * extent + 1e-300, 0);
    }

    // The reference-shaped BSP two levels at a time, for the packet walk of coherent wavefronts (ft_kernels.hip, mesh_bsp_packet):
    // branch n and its two children become one record of 40 doubles = five 64-byte units of the node array,
    //   [0..5] box of the RIGHT child, [6..11] of the LEFT child, [12..35] boxes of the grandchildren in the reference's visiting
    //   order right-right, right-left, left-right, left-left (BspMesh.fs:73-75), [36..37] int32 child[4]: >= 0 the unit of that
    //   grandchild's own record, < 0 ~leaf, INT32_MIN empty (all-NaN box: no comparison passes it).
    // Branch boxes are BspMesh.compile's own (BspMesh.fs:49), untouched.  A leaf has no box in the reference (BspMesh.fs:71): it gets
    // the bound of its triangles, inflated, which can only turn away rays that cannot hit any of them.  A child of n that is a leaf
    // takes the first slot of its half, with its box in the half's slot too.
    static constexpr int kBspWideUnits = 5;
    Box bsp_leaf_box(int32_t leaf_ref, double pad) const {
        const double nan = std::numeric_limits<double>::quiet_NaN();
        const ftd::BspLeaf& L = out.bsp_leaves[(size_t)~leaf_ref];
        if (L.n_tris == 0) return Box{{nan, nan, nan}, {nan, nan, nan}};       // an empty leaf is never worth entering
        Box b{{1e308, 1e308, 1e308}, {-1e308, -1e308, -1e308}};
        for (uint32_t k = 0; k < L.n_tris; ++k) {
            const double* T = &out.tris[9 * (size_t)(L.first_tri + k)];
            for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) {
                const double q = T[a] + (v == 1 ? T[3 + a] : v == 2 ? T[6 + a] : 0.0);
                b.lo[a] = std::min(b.lo[a], q); b.hi[a] = std::max(b.hi[a], q);
            }
        }
        for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
        return b;
    }
    int32_t widen_bsp(int32_t n, double pad) {
        const size_t unit = out.nodes.size();
        out.nodes.resize(unit + kBspWideUnits, ftd::BspNode{});
        const double nan = std::numeric_limits<double>::quiet_NaN();
        double rec[40];
        for (double& v : rec) v = nan;
        int32_t child[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
        auto put = [&](int slot, const Box& b) { for (int a = 0; a < 3; ++a) { rec[6 * slot + a] = b.lo[a]; rec[6 * slot + 3 + a] = b.hi[a]; } };
        auto exact = [&](int32_t c) { const ftd::BspNode& nd = out.nodes[(size_t)c]; return Box{{nd.bmin[0], nd.bmin[1], nd.bmin[2]}, {nd.bmax[0], nd.bmax[1], nd.bmax[2]}}; };
        const int32_t halves[2] = {out.nodes[(size_t)n].right, out.nodes[(size_t)n].left};   // right before left
        for (int h = 0; h < 2; ++h) {
            const int32_t c = halves[h];
            if (c < 0) { const Box b = bsp_leaf_box(c, pad); put(h, b); put(2 + 2 * h, b); child[2 * h] = c; continue; }
            put(h, exact(c));
            const int32_t gk[2] = {out.nodes[(size_t)c].right, out.nodes[(size_t)c].left};
            for (int k = 0; k < 2; ++k) {
                if (gk[k] < 0) { put(2 + 2 * h + k, bsp_leaf_box(gk[k], pad)); child[2 * h + k] = gk[k]; }
                else { put(2 + 2 * h + k, exact(gk[k])); child[2 * h + k] = widen_bsp(gk[k], pad); }
            }
        }
        std::memcpy(rec + 36, child, sizeof child);
        rec[38] = rec[39] = 0.0;
        std::memcpy(reinterpret_cast<double*>(&out.nodes[unit]), rec, sizeof rec);
        return (int32_t)unit;
    }

    // BspMesh.compile (BspMesh.fs:51-65); returns a child reference.
    int32_t compile(int depth_left, const std::vector<Tri3>& ts, uint32_t level) {
        if (failed) return -1;
        if (depth_left == 0) return make_leaf(ts);
        const double inf = std::numeric_limits<double>::infinity();            // BoundingBox.pointsBoundry, BoundingBox.fs:9-22
        P3 lo{inf, inf, inf}, hi{-inf, -inf, -inf};
        auto grow = [&](P3 q) {
            lo.x = std::min(lo.x, q.x); lo.y = std::min(lo.y, q.y); lo.z = std::min(lo.z, q.z);
            hi.x = std::max(hi.x, q.x); hi.y = std::max(hi.y, q.y); hi.z = std::max(hi.z, q.z);
        };
        for (auto& t : ts) { grow(t.a); grow(t.b); grow(t.c); }
        const double wx = std::fabs(hi.x - lo.x) / 2.0, wy = std::fabs(hi.y - lo.y) / 2.0, wz = std::fabs(hi.z - lo.z) / 2.0;  // optimalSplit, BspMesh.fs:30-41
        SplitPlane pl;
        if (wx > wy && wx > wz) pl = {{(lo.x + hi.x) / 2.0, 0.0, 0.0}, {1.0, 0.0, 0.0}};
        else if (wy > wz) pl = {{0.0, (lo.y + hi.y) / 2.0, 0.0}, {0.0, 1.0, 0.0}};
        else pl = {{0.0, 0.0, (lo.z + hi.z) / 2.0}, {0.0, 0.0, 1.0}};
        std::vector<Tri3> left, right;                                          // left = above pieces, right = below (BspMesh.fs:42-46)
        for (auto& t : ts)
            if (!split_triangle(pl, t, left, right)) {
                failed = true;
                err = "BSP build: a triangle edge parallel to the split plane has no intersection (Triangle.fs:10 takes .Value of None)";
                return -1;
            }
        if (left.size() >= ts.size() || right.size() >= ts.size()) return make_leaf(ts);   // BspMesh.fs:59-60
        int32_t idx = (int32_t)out.nodes.size();
        out.nodes.push_back(ftd::BspNode{});
        if (level + 1 > max_depth) max_depth = level + 1;
        int32_t l = compile(depth_left - 1, left, level + 1);
        int32_t r = compile(depth_left - 1, right, level + 1);
        ftd::BspNode& nd = out.nodes[(size_t)idx];
        nd.bmin[0] = lo.x; nd.bmin[1] = lo.y; nd.bmin[2] = lo.z; nd.bmax[0] = hi.x; nd.bmax[1] = hi.y; nd.bmax[2] = hi.z;
        nd.left = l; nd.right = r;
        return idx;
    }
};

} // namespace

int32_t build_bsp(const double* tris_abc, int64_t n_tris, int32_t depth, FlatScene& out, ftd::Mesh& mesh, std::string& err, bool device_bvh) {
    if (n_tris < 0 || (n_tris > 0 && !tris_abc) || depth < 0) { err = "bad mesh arguments"; return FT_ERR_INVALID; }
    std::vector<Tri3> ts((size_t)n_tris);
    for (int64_t i = 0; i < n_tris; ++i) {
        const double* v = tris_abc + 9 * i;
        ts[(size_t)i] = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}};
    }
    BspBuilder b{out, err};
    int32_t root = (n_tris == 0) ? b.make_leaf(ts) : b.compile(depth, ts, 0);   // an empty mesh is an empty group
    if (b.failed) return FT_ERR_BUILD;
    mesh.root = root; mesh.n_source_tris = (uint32_t)n_tris; mesh.max_depth = b.max_depth;
    mesh.bvh_root = INT32_MIN;
    int32_t wide_root = INT32_MIN;
    bool deferred = false;
    if (root < 0 && device_bvh && ts.size() >= 8 && ts.size() < (1u << 28)) {
        // top-level Leaf, BVH built on the device after the upload: reserve its ranges here (ft_bvh.hip documents what lands where)
        const uint32_t n = (uint32_t)ts.size();
        FlatScene::BvhJob job{(uint32_t)out.meshes.size(), out.bsp_leaves[(size_t)~root].first_tri, n, (uint32_t)out.nodes.size(), (uint32_t)out.bsp_leaves.size(),
                              (uint32_t)(out.tris.size() / 9), (uint32_t)(out.wide.size() / ftd::kWideNodeDoubles), (uint32_t)(out.coarse_boxes.size() / 6), 64u};
        out.nodes.resize(out.nodes.size() + (n - 1), ftd::BspNode{});
        out.bsp_leaves.resize(out.bsp_leaves.size() + (2 * (size_t)n - 1), ftd::BspLeaf{0, 0});
        out.tris.resize(out.tris.size() + 9 * (size_t)n, 0.0);
        out.tri_orig.resize(out.tri_orig.size() + n, 0u);
        out.wide.resize(out.wide.size() + (size_t)ftd::kWideNodeDoubles * (n - 1), 0.0);
        for (uint32_t k = 0; k < job.coarse_count; ++k) { const float all[6] = {-3e38f, -3e38f, -3e38f, 3e38f, 3e38f, 3e38f}; out.coarse_boxes.insert(out.coarse_boxes.end(), all, all + 6); }   // until the build: everything
        out.bvh_jobs.push_back(job);
        out.bvh_nodes += n - 1; out.bvh_leaves += 2 * (int64_t)n - 1; out.bvh_tris += n;
        mesh.bvh_root = (int32_t)job.node_base;
        wide_root = (int32_t)job.wide_base;
        out.mesh_wide.push_back(wide_root);
        out.mesh_coarse.push_back(job.coarse_first); out.mesh_coarse.push_back(job.coarse_count);
        deferred = true;
    } else if (root < 0) {                                                      // top-level Leaf: add the exact BVH for closest / any-hit queries
        const size_t n0 = out.nodes.size(), l0 = out.bsp_leaves.size(), t0 = out.tris.size() / 9;
        mesh.bvh_root = b.bvh_for_leaf(ts, out.bsp_leaves[(size_t)~root].first_tri);
        wide_root = mesh.bvh_root >= 0 ? b.widen(mesh.bvh_root) : INT32_MIN;
        out.bvh_nodes += (int64_t)(out.nodes.size() - n0); out.bvh_leaves += (int64_t)(out.bsp_leaves.size() - l0); out.bvh_tris += (int64_t)(out.tris.size() / 9 - t0);
        if (mesh.bvh_root >= 0 && (int32_t)b.bvh_depth + 1 > out.stack_capacity) out.stack_capacity = (int32_t)b.bvh_depth + 1;
    }
    if (deferred) return FT_OK;
    if (root >= 0 && b.max_depth <= 40) {                                       // a real tree: its two-levels-at-a-time form (3 stack entries per wide level, 64 at most)
        const ftd::BspNode& top = out.nodes[(size_t)root];
        double extent = 0.0;
        for (int a = 0; a < 3; ++a) extent = std::max(extent, std::max(std::fabs(top.bmin[a]), std::fabs(top.bmax[a])));
        if (extent < 1e300) {
            const size_t n0 = out.nodes.size();
            wide_root = b.widen_bsp(root, 1e-7 * extent + 1e-300);
            out.bvh_nodes += (int64_t)(out.nodes.size() - n0);                  // not part of BspMesh.compile's tree
        }
    }
    out.mesh_wide.push_back(wide_root);
    {   // <= 64 boxes that together hold every triangle of the mesh: one level of the binary tree (BVH of a top-level Leaf, or the BSP itself)
        const int32_t top = root < 0 ? mesh.bvh_root : root;
        const uint32_t first = (uint32_t)(out.coarse_boxes.size() / 6);
        uint32_t count = 0;
        if (top >= 0 && top != INT32_MIN) {
            std::vector<int32_t> frontier{top};
            for (;;) {
                std::vector<int32_t> next;
                bool any_inner = false;
                for (int32_t c : frontier) {
                    if (c < 0) { next.push_back(c); continue; }
                    any_inner = true;
                    next.push_back(out.nodes[(size_t)c].left); next.push_back(out.nodes[(size_t)c].right);
                }
                if (!any_inner || next.size() > 64) break;
                frontier.swap(next);
            }
            for (int32_t c : frontier) {
                double lo[3] = {1e308, 1e308, 1e308}, hi[3] = {-1e308, -1e308, -1e308};
                if (c >= 0) { const ftd::BspNode& nd = out.nodes[(size_t)c]; for (int a = 0; a < 3; ++a) { lo[a] = nd.bmin[a]; hi[a] = nd.bmax[a]; } }
                else {
                    const ftd::BspLeaf& L = out.bsp_leaves[(size_t)~c];
                    for (uint32_t k = 0; k < L.n_tris; ++k) {
                        const double* T = &out.tris[9 * (size_t)(L.first_tri + k)];
                        for (int v = 0; v < 3; ++v) for (int a = 0; a < 3; ++a) {
                            const double q = T[a] + (v == 1 ? T[3 + a] : v == 2 ? T[6 + a] : 0.0);
                            lo[a] = std::min(lo[a], q); hi[a] = std::max(hi[a], q);
                        }
                    }
                    if (L.n_tris == 0) continue;                                // an empty leaf holds nothing
                }
                float rec[6];
                for (int a = 0; a < 3; ++a) {                                   // outward, with room for the float arithmetic of the test
                    const double pad = 1e-5 * (std::fabs(lo[a]) + std::fabs(hi[a]) + (hi[a] - lo[a])) + 1e-30;
                    rec[a] = std::nextafter((float)(lo[a] - pad), -std::numeric_limits<float>::infinity());
                    rec[3 + a] = std::nextafter((float)(hi[a] + pad), std::numeric_limits<float>::infinity());
                }
                out.coarse_boxes.insert(out.coarse_boxes.end(), rec, rec + 6);
                ++count;
            }
        }
        out.mesh_coarse.push_back(first); out.mesh_coarse.push_back(count);
    }
    return FT_OK;
}

int32_t slice_triangle(const double p0[3], const double n[3], const double tri[9], std::vector<double>& above, std::vector<double>& below, std::string& err) {
    SplitPlane pl{{p0[0], p0[1], p0[2]}, {n[0], n[1], n[2]}};
    Tri3 t{{tri[0], tri[1], tri[2]}, {tri[3], tri[4], tri[5]}, {tri[6], tri[7], tri[8]}};
    std::vector<Tri3> a, b;
    if (!split_triangle(pl, t, a, b)) { err = "edge parallel to plane"; return FT_ERR_BUILD; }
    auto dump = [](const std::vector<Tri3>& v, std::vector<double>& o) {
        o.clear();
        for (auto& q : v) { const double w[9] = {q.a.x, q.a.y, q.a.z, q.b.x, q.b.y, q.b.z, q.c.x, q.c.y, q.c.z}; o.insert(o.end(), w, w + 9); }
    };
    dump(a, above); dump(b, below);
    return FT_OK;
}

} // namespace fth
