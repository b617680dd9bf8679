// host_api.cpp — C ABI over the host-side scene model: parse, lower through a builder table,
// jitter pattern, PNG output.  Reference citations are relative to /root/reference/FuncTracer/.
#include "host_api.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <vector>

#include "Scene.hpp"

using namespace FuncTracer;

struct fth_scene {
    SceneOptions options;
    Scene scene;
};

namespace {

void put_err(char* err, int32_t len, const std::string& m) {
    if (err && len > 0) { std::snprintf(err, (size_t)len, "%s", m.c_str()); }
}

struct Lowerer {
    const fth_builder* b;
    void* ctx;
    int32_t status = FT_OK;
    std::map<const SceneGraph*, ft_node> memo;      // shared subtrees (repeat, SceneParser.fs:241-251) lower once

    ft_node check(ft_node n) { if (n < 0 && status == FT_OK) status = n; return n; }

    static void basic(const Transform& t, std::vector<ft_transform>& out) {
        if (t.kind == Transform::Composed) { for (auto& p : t.parts) basic(p, out); return; }
        ft_transform x{};
        x.kind = t.kind == Transform::Translate ? FT_TRANSLATE : t.kind == Transform::Scale ? FT_SCALE : FT_ROTATE;
        x.v[0] = t.v[0]; x.v[1] = t.v[1]; x.v[2] = t.v[2]; x.angle = t.angle;
        out.push_back(x);
    }

    ft_node lower(const SceneGraphPtr& g) {
        if (status != FT_OK) return status;
        auto it = memo.find(g.get());
        if (it != memo.end()) return it->second;
        ft_node r = FT_ERR_INVALID;
        switch (g->kind) {
            case SceneGraph::PrimitiveN:
                if (g->primitive == Primitive::Triangle) {
                    const Triangle& t = g->triangle;
                    const double v[9] = {t.a[0], t.a[1], t.a[2], t.b[0], t.b[1], t.b[2], t.c[0], t.c[1], t.c[2]};
                    r = b->sg_triangle(ctx, v);
                } else if (g->primitive == Primitive::BspMesh) {
                    std::vector<double> flat;
                    if (g->meshTriangles) {
                        flat.reserve(g->meshTriangles->size() * 9);
                        for (auto& t : *g->meshTriangles) { const double v[9] = {t.a[0], t.a[1], t.a[2], t.b[0], t.b[1], t.b[2], t.c[0], t.c[1], t.c[2]}; flat.insert(flat.end(), v, v + 9); }
                    }
                    r = b->sg_bsp_mesh(ctx, g->bspDepth, flat.data(), (int64_t)(flat.size() / 9));
                } else {
                    r = b->sg_primitive(ctx, (int32_t)g->primitive);   // enum order matches ft_primitive_kind (Scene.fs:10-17)
                }
                break;
            case SceneGraph::SceneFunctionN: {
                ft_node child = lower(g->nodes[0]);
                if (status != FT_OK) return status;
                const SceneFunction& f = g->function;
                switch (f.kind) {
                    case SceneFunction::TransformF: { std::vector<ft_transform> ts; basic(f.transform, ts); r = b->sg_transform(ctx, ts.data(), (int32_t)ts.size(), child); break; }
                    case SceneFunction::MaterialF: {
                        ft_material m{};
                        m.colour[0] = f.material.colour.r; m.colour[1] = f.material.colour.g; m.colour[2] = f.material.colour.b;
                        m.roughness = f.material.roughness; m.reflectance = f.material.reflectance; m.shineyness = f.material.shineyness;
                        m.apply_lighting = f.material.applyLighting ? 1 : 0;
                        r = b->sg_material(ctx, &m, child);
                        break;
                    }
                    case SceneFunction::TextureF: {
                        std::vector<double> ops;
                        for (auto& tf : f.texture.functions) { ops.push_back(tf.kind == TextureFunction::Scale ? 0.0 : 1.0); ops.push_back(tf.a); ops.push_back(tf.b); }
                        if (f.texture.kind == Texture::Image) {
                            if (!f.texture.pixels || !b->sg_texture_image) { status = FT_ERR_UNSUPPORTED; return status; }
                            r = b->sg_texture_image(ctx, f.texture.pixels->data(), f.texture.width, f.texture.height, ops.data(), (int32_t)(ops.size() / 3), child);
                            break;
                        }
                        const double ca[3] = {f.texture.c1.r, f.texture.c1.g, f.texture.c1.b}, cb[3] = {f.texture.c2.r, f.texture.c2.g, f.texture.c2.b};
                        r = b->sg_texture_grid(ctx, ca, cb, ops.data(), (int32_t)(ops.size() / 3), child);
                        break;
                    }
                    case SceneFunction::HueShift: r = b->sg_hue_shift(ctx, f.angle, child); break;
                    default: r = b->sg_ignore_light(ctx, child); break;
                }
                break;
            }
            case SceneGraph::Group: {
                std::vector<ft_node> kids;
                for (auto& n : g->nodes) { kids.push_back(lower(n)); if (status != FT_OK) return status; }
                r = b->sg_group(ctx, kids.data(), (int32_t)kids.size());
                break;
            }
            default: {
                ft_node a = lower(g->nodes[0]); if (status != FT_OK) return status;
                ft_node c = lower(g->nodes[1]); if (status != FT_OK) return status;
                int32_t op = g->kind == SceneGraph::Union ? FT_CSG_UNION : g->kind == SceneGraph::Intersect ? FT_CSG_INTERSECT : g->kind == SceneGraph::Subtract ? FT_CSG_SUBTRACT : FT_CSG_EXCLUDE;
                r = b->sg_csg(ctx, op, a, c);
                break;
            }
        }
        check(r);
        memo[g.get()] = r;
        return r;
    }
};

// ---- minimal PNG (stored deflate) -------------------------------------------------------------
uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0) {
    if (!crc_ready) { for (uint32_t i = 0; i < 256; ++i) { uint32_t x = i; for (int k = 0; k < 8; ++k) x = (x & 1) ? 0xEDB88320u ^ (x >> 1) : x >> 1; crc_table[i] = x; } crc_ready = true; }
    c = ~c;
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
void be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4); td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size()));
}

} // namespace

extern "C" {

fth_scene* fth_parse_scene(const char* text, const char* base_dir, char* err, int32_t err_len) {
    if (!text) { put_err(err, err_len, "null text"); return nullptr; }
    auto* s = new fth_scene();
    std::string e;
    if (!parseScene(text, base_dir ? base_dir : "", s->options, s->scene, e)) { put_err(err, err_len, e); delete s; return nullptr; }
    return s;
}
fth_scene* fth_parse_scene_file(const char* path, char* err, int32_t err_len) {
    if (!path) { put_err(err, err_len, "null path"); return nullptr; }
    std::string text;
    if (!readWholeFile(path, text)) { put_err(err, err_len, std::string("cannot open ") + path); return nullptr; }
    std::string p(path);
    size_t slash = p.find_last_of('/');
    std::string dir = slash == std::string::npos ? "." : p.substr(0, slash);
    return fth_parse_scene(text.c_str(), dir.c_str(), err, err_len);
}
void fth_scene_free(fth_scene* s) { delete s; }

int32_t fth_scene_options(const fth_scene* s, fth_options* out) {
    if (!s || !out) return FT_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    const Camera& c = s->options.camera;
    for (int k = 0; k < 3; ++k) { out->camera.o[k] = c.o[k]; out->camera.look_at[k] = c.lookAt[k]; out->camera.up[k] = c.up[k]; }
    out->camera.fov_y = c.fovY; out->camera.aspect_ratio = c.aspectRatio;
    out->camera.has_focus = c.hasFocus ? 1 : 0; out->camera.focal_length = c.focus.focalLength; out->camera.aperture_angular_size = c.focus.apetureAngularSize;
    out->res_h = s->options.resolution.h; out->res_v = s->options.resolution.v;
    out->samples = s->options.samplingStrategy.samplesPerPixel; out->corner = s->options.samplingStrategy.corner ? 1 : 0;
    return FT_OK;
}
int32_t fth_scene_counts(const fth_scene* s, int32_t* n_objects, int32_t* n_lights) {
    if (!s) return FT_ERR_INVALID;
    if (n_objects) *n_objects = (int32_t)s->scene.objects->nodes.size();
    if (n_lights) *n_lights = (int32_t)s->scene.lights.size();
    return FT_OK;
}

int32_t fth_scene_lower(const fth_scene* s, const fth_builder* b, void* ctx) {
    if (!s || !b) return FT_ERR_INVALID;
    int32_t rc = b->scene_clear(ctx);
    if (rc != FT_OK) return rc;
    Lowerer L{b, ctx, FT_OK, {}};
    ft_node root = L.lower(s->scene.objects);
    if (L.status != FT_OK) return L.status;
    for (const Light& l : s->scene.lights) {
        const double v[3] = {l.v[0], l.v[1], l.v[2]}, col[3] = {l.colour.r, l.colour.g, l.colour.b};
        const double fo[3] = {l.falloff.constant, l.falloff.linear, l.falloff.quadratic};
        if (l.kind == Light::Directional) rc = b->scene_add_directional(ctx, v, col);
        else if (l.kind == Light::SoftDirectional) rc = b->scene_add_soft_directional(ctx, v, l.samples, l.scattering, col);
        else rc = b->scene_add_positional(ctx, v, fo, col);
        if (rc != FT_OK) return rc;
    }
    if ((rc = b->scene_set_objects(ctx, root)) != FT_OK) return rc;
    return b->scene_commit(ctx);
}

int32_t fth_parse_colour(const char* text, double rgb[3]) {
    Colour c;
    if (!text || !rgb || !parseColour(text, c)) return -1;
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
    return 0;
}

int64_t fth_parse_ply(const char* text, double* out, int64_t cap, char* err, int32_t err_len) {
    std::vector<Triangle> tris; std::string e;
    bool ok = false;
    try { ok = text && parsePly(text, tris, e); } catch (const std::exception& x) { e = std::string("ply parser: ") + x.what(); }   // nothing may cross the C boundary
    if (!ok) { put_err(err, err_len, e); return -1; }
    if (out) for (int64_t i = 0; i < (int64_t)tris.size() && i < cap; ++i) {
        const Triangle& t = tris[(size_t)i];
        const double v[9] = {t.a[0], t.a[1], t.a[2], t.b[0], t.b[1], t.b[2], t.c[0], t.c[1], t.c[2]};
        std::memcpy(out + 9 * i, v, sizeof v);
    }
    return (int64_t)tris.size();
}

int64_t fth_load_image(const char* path, int32_t* width, int32_t* height, uint8_t* out, int64_t cap, char* err, int32_t err_len) {
    std::vector<uint8_t> rgb; std::string e; int w = 0, h = 0;
    bool ok = false;
    try { ok = path && loadImageRgb24(path, w, h, rgb, e); } catch (const std::exception& x) { e = std::string("image loader: ") + x.what(); }
    if (!ok) { put_err(err, err_len, e); return FT_ERR_INVALID; }
    if (width) *width = w;
    if (height) *height = h;
    if (out && cap >= (int64_t)rgb.size()) std::memcpy(out, rgb.data(), rgb.size());
    return (int64_t)rgb.size();
}

int32_t fth_jitter_pattern(uint64_t seed, int32_t n, double* out_xy) {
    if (n < 0 || (n > 0 && !out_xy)) return FT_ERR_INVALID;
    uint64_t state = seed;
    auto next_u = [&]() {                                           // splitmix64
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);      // [0,1), like NextDouble()
    };
    for (int i = 0; i < n; ++i) {                                   // Jitter.circle (Jitter.fs:15-21): rejection from [-1,1]^2
        for (;;) {
            double x = 2.0 * next_u() - 1.0, y = 2.0 * next_u() - 1.0;   // uniform, Jitter.fs:9-10
            if ((x * x + y * y) > 1.0) continue;
            out_xy[2 * i] = x; out_xy[2 * i + 1] = y;
            break;
        }
    }
    return FT_OK;
}

int32_t fth_write_png(const char* path, const uint8_t* rgba, int32_t width, int32_t height) {
    if (!path || !rgba || width < 1 || height < 1) return FT_ERR_INVALID;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * ((size_t)width * 4 + 1));
    for (int y = 0; y < height; ++y) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * width * 4, rgba + (size_t)(y + 1) * width * 4); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, bsum = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521u; bsum = (bsum + a) % 65521u; }
    for (size_t off = 0; off < raw.size(); off += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
    }
    be32(z, (bsum << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr; be32(ihdr, (uint32_t)width); be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", {});
    std::FILE* f = std::fopen(path, "wb");
    if (!f) return FT_ERR_INVALID;
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return w == out.size() ? FT_OK : FT_ERR_INVALID;
}

} // extern "C"
