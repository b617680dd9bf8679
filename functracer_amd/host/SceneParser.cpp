// SceneParser.cpp — the `.scene` DSL of SceneParser.fs:11-366 and the ASCII PLY layout of
// PlyParser.fs:20-69 as a hand-written recursive-descent parser (the reference uses FParsec).
// Same grammar, same section order (options, objects, lights), same units (angles in degrees).
#include <cstdio>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>

#include "Scene.hpp"

namespace FuncTracer {

namespace {
const double PI = 3.14159265358979323846;
double degToRad(double d) { return d * (PI / 180.0); }             // CommonTypes.fs:98-99
Vec3 normalise(Vec3 v) {                                           // CommonTypes.fs:63-67
    double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (l < 0.0000001) return v;
    double s = 1.0 / l;
    return {s * v[0], s * v[1], s * v[2]};
}
} // namespace

Transform translate(Vec3 v) { Transform t; t.kind = Transform::Translate; t.v = v; return t; }
Transform scale(Vec3 v) { Transform t; t.kind = Transform::Scale; t.v = v; return t; }
Transform rotate(Vec3 axis, double angleRad) { Transform t; t.kind = Transform::Rotate; t.v = normalise(axis); t.angle = angleRad; return t; }
Transform compose(const std::vector<Transform>& ts) {
    Transform c; c.kind = Transform::Composed;
    for (auto& t : ts) { if (t.kind == Transform::Composed) for (auto& u : t.parts) c.parts.push_back(u); else c.parts.push_back(t); }
    return c;
}

namespace {

struct ParseError { std::string msg; };

using GraphFn = std::function<SceneGraphPtr(SceneGraphPtr)> ;

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string baseDir;
    explicit Parser(const std::string& text, const std::string& dir) : s(text), baseDir(dir) {}

    [[noreturn]] void fail(const std::string& what) {
        int line = 1, col = 1;
        for (size_t k = 0; k < i && k < s.size(); ++k) { if (s[k] == '\n') { ++line; col = 1; } else ++col; }
        throw ParseError{"Error in Ln: " + std::to_string(line) + " Col: " + std::to_string(col) + ": expecting " + what};
    }
    bool eof() const { return i >= s.size(); }
    char peek() const { return eof() ? '\0' : s[i]; }
    bool isNewline(char c) const { return c == '\n' || c == '\r'; }
    void ws() { while (!eof() && (s[i] == ' ' || s[i] == '\t')) ++i; }                           // SceneParser.fs:18
    void ws1() { if (eof() || !(s[i] == ' ' || s[i] == '\t')) fail("space or tab"); ws(); }      // :19
    void anyWhitespace() { while (!eof() && (s[i] == ' ' || s[i] == '\t' || isNewline(s[i]))) ++i; }  // :21-22
    bool skipTriviaOnce() {                                                                       // :24-25
        if (eof()) return false;
        if (isNewline(s[i])) { if (s[i] == '\r' && i + 1 < s.size() && s[i + 1] == '\n') ++i; ++i; return true; }
        if (s[i] == ';') { while (!eof() && !isNewline(s[i])) ++i; if (!eof()) { if (s[i] == '\r' && i + 1 < s.size() && s[i + 1] == '\n') ++i; ++i; } return true; }
        return false;
    }
    bool skipTrivia1() { bool any = false; while (skipTriviaOnce()) any = true; return any; }
    bool lookingAtCI(const char* kw) const {
        size_t k = 0;
        for (; kw[k]; ++k) { if (i + k >= s.size() || std::tolower((unsigned char)s[i + k]) != std::tolower((unsigned char)kw[k])) return false; }
        return true;
    }
    bool acceptCI(const char* kw) { if (!lookingAtCI(kw)) return false; i += std::string(kw).size(); return true; }   // skipStringCI
    void expectChar(char c) { if (peek() != c) fail(std::string("'") + c + "'"); ++i; }
    void keyword(const char* kw) { if (!acceptCI(kw)) fail(std::string("'") + kw + "'"); anyWhitespace(); }    // pkeyword, :52-53

    double number(bool allowMinus = true) {                                                       // numberLiteral, :32-50
        size_t st = i;
        if (allowMinus && peek() == '-') ++i;
        size_t digits = 0;
        while (!eof() && std::isdigit((unsigned char)s[i])) { ++i; ++digits; }
        if (peek() == '.') { ++i; while (!eof() && std::isdigit((unsigned char)s[i])) { ++i; ++digits; } }
        if (digits == 0) { i = st; fail("number"); }
        if (peek() == 'e' || peek() == 'E') {
            size_t save = i; ++i;
            if (peek() == '+' || peek() == '-') ++i;
            if (std::isdigit((unsigned char)peek())) { while (!eof() && std::isdigit((unsigned char)s[i])) ++i; } else i = save;
        }
        return std::strtod(s.substr(st, i - st).c_str(), nullptr);
    }
    double pfloat() {                                                                             // FParsec pfloat: optional sign
        bool neg = false;
        if (peek() == '+' || peek() == '-') { neg = peek() == '-'; ++i; }
        double v = number(false);
        return neg ? -v : v;
    }
    int integer() {
        size_t st = i;
        if (peek() == '-' || peek() == '+') ++i;
        size_t d0 = i;
        while (!eof() && std::isdigit((unsigned char)s[i])) ++i;
        if (i == d0) { i = st; fail("integer"); }
        return std::atoi(s.substr(st, i - st).c_str());
    }
    Vec3 triple() {                                                                               // ptriple, :55-60
        expectChar('('); anyWhitespace();
        Vec3 v;
        v[0] = number(); ws();
        for (int k = 1; k < 3; ++k) { expectChar(','); ws(); v[k] = number(); ws(); }
        anyWhitespace(); expectChar(')');
        return v;
    }
    std::array<double, 2> pair() {                                                                // ppair, :62-67
        expectChar('('); anyWhitespace();
        std::array<double, 2> v;
        v[0] = number(); ws(); expectChar(','); ws(); v[1] = number(); ws();
        anyWhitespace(); expectChar(')');
        return v;
    }
    Colour colour() {                                                                             // pcolour, :69-87
        if (peek() == '(') { Vec3 v = triple(); return {v[0], v[1], v[2]}; }
        if (peek() == '#') {
            ++i;
            if (i + 6 > s.size()) fail("6 hex digits");
            double c[3];
            for (int k = 0; k < 3; ++k) {
                std::string h = s.substr(i + 2 * k, 2);
                char* end = nullptr;
                long v = std::strtol(h.c_str(), &end, 16);
                if (end != h.c_str() + 2) fail("hex colour");
                c[k] = (double)v / 255.0;
            }
            i += 6;
            return {c[0], c[1], c[2]};
        }
        double x = number();
        return {x, x, x};
    }
    std::string file() {                                                                          // pfile, :89-91
        expectChar('"');
        size_t st = i;
        while (!eof() && s[i] != '"') ++i;
        if (eof()) fail("closing quote");
        std::string f = s.substr(st, i - st);
        ++i;
        return f;
    }
    std::string resolvePath(std::string f) const {
        for (auto& ch : f) if (ch == '\\') ch = '/';
        bool absolute = !f.empty() && (f[0] == '/' || (f.size() > 1 && f[1] == ':'));
        if (absolute || baseDir.empty()) return f;
        return baseDir + "/" + f;
    }
    std::shared_ptr<const std::vector<Triangle>> loadPly(const std::string& f) {
        std::string path = resolvePath(f);
        std::string text;
        if (!readWholeFile(path, text)) throw ParseError{"cannot open mesh file: " + path};
        auto tris = std::make_shared<std::vector<Triangle>>();
        std::string err;
        if (!parsePly(text, *tris, err)) throw ParseError{err};                              // SceneParser.fs:123-124, 135-136 raise
        return tris;
    }

    Material material() {                                                                         // pmaterial, :99-111
        Material m = mattWhite();
        if (acceptCI("diffuse")) { anyWhitespace(); m.colour = colour(); ws1(); }
        if (acceptCI("roughness")) { anyWhitespace(); m.roughness = pfloat(); ws1(); }
        if (acceptCI("reflectance")) { anyWhitespace(); m.reflectance = pfloat(); ws1(); }
        if (acceptCI("shineyness")) { anyWhitespace(); m.shineyness = pfloat(); }
        return m;
    }

    Texture texture() {                                                                           // :159-185
        if (acceptCI("grid")) { anyWhitespace(); Texture t; t.kind = Texture::Grid; t.c1 = colour(); ws1(); t.c2 = colour(); return t; }
        if (acceptCI("image")) {                                                                  // ImageTexture.image, Textures/Image.fs:20-36
            anyWhitespace(); Texture t; t.kind = Texture::Image; t.source = file();
            auto px = std::make_shared<std::vector<uint8_t>>(); std::string e;
            if (!loadImageRgb24(resolvePath(t.source), t.width, t.height, *px, e)) throw ParseError{e};   // the reference raises while parsing too
            t.pixels = px;
            return t;
        }
        if (peek() == '(') {
            ++i; anyWhitespace();
            TextureFunction f{};
            if (acceptCI("scale")) { anyWhitespace(); auto p = pair(); f.kind = TextureFunction::Scale; f.a = p[0]; f.b = p[1]; }
            else if (acceptCI("rotate")) { anyWhitespace(); f.kind = TextureFunction::Rotate; f.a = degToRad(number()); }
            else fail("texture function");
            anyWhitespace();
            Texture inner = texture();
            anyWhitespace(); expectChar(')');
            inner.functions.insert(inner.functions.begin(), f);      // TextureFunction(t, f): f is applied to uv first
            return inner;
        }
        fail("texture");
    }

    static SceneGraphPtr fnNode(SceneFunction f, SceneGraphPtr g) {
        auto n = std::make_shared<SceneGraph>(); n->kind = SceneGraph::SceneFunctionN; n->function = std::move(f); n->nodes = {std::move(g)}; return n;
    }

    bool geometryFunctionAhead() const {
        return lookingAtCI("IgnoreLight") || lookingAtCI("texture") || lookingAtCI("hueShift") || lookingAtCI("material") || lookingAtCI("repeat") ||
               lookingAtCI("scale") || lookingAtCI("translate") || lookingAtCI("rotate") || peek() == '(';
    }

    GraphFn geometryFunction() {                                                                  // :263
        if (acceptCI("IgnoreLight")) { return [](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::IgnoreLight; return fnNode(f, g); }; }   // :253
        if (acceptCI("texture")) { anyWhitespace(); Texture t = texture(); return [t](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::TextureF; f.texture = t; return fnNode(f, g); }; }
        if (acceptCI("hueShift")) { anyWhitespace(); double a = pfloat(); return [a](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::HueShift; f.angle = a; return fnNode(f, g); }; }
        if (acceptCI("material")) { anyWhitespace(); Material m = material(); return [m](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::MaterialF; f.material = m; return fnNode(f, g); }; }
        if (acceptCI("repeat")) {                                                                 // :241-251
            anyWhitespace(); int count = integer(); anyWhitespace();
            GraphFn f = geometryFunction(); anyWhitespace();
            return [count, f](SceneGraphPtr g) {
                auto grp = std::make_shared<SceneGraph>(); grp->kind = SceneGraph::Group;
                SceneGraphPtr cur = g;
                for (int k = 0; k <= count; ++k) { cur = f(cur); grp->nodes.push_back(cur); }   // [f g; f (f g); ...], count+1 items
                return SceneGraphPtr(grp);
            };
        }
        if (acceptCI("scale")) {                                                                  // :194-198
            anyWhitespace();
            Vec3 v;
            if (peek() == '(') v = triple(); else { double x = number(); v = {x, x, x}; }
            ws1();
            return [v](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::TransformF; f.transform = scale(v); return fnNode(f, g); };
        }
        if (acceptCI("translate")) { anyWhitespace(); Vec3 v = triple(); return [v](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::TransformF; f.transform = translate(v); return fnNode(f, g); }; }  // :215-219
        if (acceptCI("rotate")) {                                                                 // :204-213
            anyWhitespace(); Vec3 axis = triple(); ws1(); double deg = pfloat();
            return [axis, deg](SceneGraphPtr g) { SceneFunction f; f.kind = SceneFunction::TransformF; f.transform = rotate(axis, degToRad(deg)); return fnNode(f, g); };
        }
        if (peek() == '(') {                                                                      // composed, :235-239: (f) . (g)  =  f >> g
            ++i; anyWhitespace(); GraphFn f1 = geometryFunction(); anyWhitespace(); expectChar(')');
            anyWhitespace(); expectChar('.'); anyWhitespace();
            expectChar('('); anyWhitespace(); GraphFn f2 = geometryFunction(); anyWhitespace(); expectChar(')');
            return [f1, f2](SceneGraphPtr g) { return f2(f1(g)); };
        }
        fail("geometry function");
    }

    bool primitiveAhead() const {
        static const char* names[] = {"mesh", "bspMesh", "circle", "square", "cube", "sphere", "plane", "cone", "solidCylinder", "cylinder"};
        for (auto n : names) if (lookingAtCI(n)) return true;
        return false;
    }
    bool geometryAhead() const { return peek() == '(' || primitiveAhead(); }

    SceneGraphPtr primitive() {                                                                   // :116-154
        auto prim = [](Primitive p) { auto n = std::make_shared<SceneGraph>(); n->kind = SceneGraph::PrimitiveN; n->primitive = p; return SceneGraphPtr(n); };
        if (acceptCI("mesh")) {                                                                   // Group of Triangle primitives, :116-126
            anyWhitespace();
            auto tris = loadPly(file());
            auto grp = std::make_shared<SceneGraph>(); grp->kind = SceneGraph::Group;
            for (auto& t : *tris) { auto n = std::make_shared<SceneGraph>(); n->kind = SceneGraph::PrimitiveN; n->primitive = Primitive::Triangle; n->triangle = t; grp->nodes.push_back(n); }
            return grp;
        }
        if (acceptCI("bspMesh")) {                                                                // :128-141
            anyWhitespace(); int depth = integer(); ws1();
            auto tris = loadPly(file());
            auto n = std::make_shared<SceneGraph>(); n->kind = SceneGraph::PrimitiveN; n->primitive = Primitive::BspMesh; n->bspDepth = depth; n->meshTriangles = tris;
            return n;
        }
        if (acceptCI("circle")) return prim(Primitive::Circle);
        if (acceptCI("square")) return prim(Primitive::Square);
        if (acceptCI("cube")) return prim(Primitive::Cube);
        if (acceptCI("sphere")) return prim(Primitive::Sphere);
        if (acceptCI("plane")) return prim(Primitive::Plane);
        if (acceptCI("cone")) return prim(Primitive::Cone);
        if (acceptCI("solidCylinder")) return prim(Primitive::SolidCylinder);
        if (acceptCI("cylinder")) return prim(Primitive::Cylinder);
        fail("primitive");
    }

    SceneGraphPtr geometry() {                                                                    // :264
        if (peek() != '(') return primitive();
        ++i; anyWhitespace();
        SceneGraphPtr r = appliedFunction();
        anyWhitespace(); expectChar(')');
        return r;
    }
    SceneGraphPtr binary(SceneGraph::Kind k) {                                                    // :221-226
        anyWhitespace();
        SceneGraphPtr a = geometry(); ws1(); SceneGraphPtr b = geometry();
        auto n = std::make_shared<SceneGraph>(); n->kind = k; n->nodes = {a, b};
        return n;
    }
    SceneGraphPtr appliedFunction() {                                                             // :255-261
        if (acceptCI("union")) return binary(SceneGraph::Union);
        if (acceptCI("subtract")) return binary(SceneGraph::Subtract);
        if (acceptCI("intersect")) return binary(SceneGraph::Intersect);
        if (acceptCI("exclude")) return binary(SceneGraph::Exclude);
        if (acceptCI("group")) {                                                                  // :228-231
            anyWhitespace();
            auto grp = std::make_shared<SceneGraph>(); grp->kind = SceneGraph::Group;
            while (geometryAhead()) { grp->nodes.push_back(geometry()); anyWhitespace(); }
            return grp;
        }
        GraphFn f = geometryFunction();                                                           // applied, :93-97
        anyWhitespace();
        return f(geometry());
    }

    void camera(SceneOptions& o) {                                                                // :272-294
        Camera c;
        keyword("pos"); c.o = triple(); ws1();
        keyword("lookat"); c.lookAt = triple(); ws1();
        keyword("up"); c.up = normalise(triple()); ws1();
        keyword("fov"); c.fovY = degToRad(number(false)); ws1();
        keyword("ratio"); c.aspectRatio = number(false); ws();
        if (acceptCI("focus")) { anyWhitespace(); auto p = pair(); c.hasFocus = true; c.focus = {p[0], degToRad(p[1])}; }
        o.camera = c;
    }

    void run(SceneOptions& options, Scene& scene) {                                               // pscenegraph, :353-358
        options = SceneOptions{};
        while (skipTriviaOnce()) {}
        for (;;) {                                                                                // poptions, :315-317
            if (acceptCI("camera")) { anyWhitespace(); camera(options); }
            else if (acceptCI("samples")) {                                                       // :299-306
                anyWhitespace();
                if (acceptCI("corner")) options.samplingStrategy = {true, 1};
                else options.samplingStrategy = {false, integer()};
            }
            else if (acceptCI("res")) { anyWhitespace(); int h = integer(); ws1(); int v = integer(); options.resolution = {h, v}; }   // :308-313
            else break;
            ws();
            if (!skipTrivia1()) break;
        }
        auto root = std::make_shared<SceneGraph>(); root->kind = SceneGraph::Group;               // { objects = Group objects }, :354
        while (geometryAhead()) {                                                                 // pobjects, :266-270
            root->nodes.push_back(geometry()); ws();
            if (!skipTrivia1()) break;
        }
        scene.objects = root;
        scene.lights.clear();
        for (;;) {                                                                                // plights, :349-351
            Light l;
            if (acceptCI("directional")) {                                                        // :319-326
                anyWhitespace(); keyword("dir"); l.kind = Light::Directional; l.v = triple(); ws1(); keyword("colour"); l.colour = colour();
            } else if (acceptCI("softdirectional")) {                                             // :328-337
                anyWhitespace(); keyword("dir"); l.kind = Light::SoftDirectional; l.v = triple(); ws1();
                keyword("samples"); l.samples = integer(); ws1();
                keyword("scatter"); l.scattering = degToRad(number(false)); ws1();
                keyword("colour"); Vec3 c = triple(); l.colour = {c[0], c[1], c[2]};
            } else if (acceptCI("positional")) {                                                  // :339-347
                anyWhitespace(); keyword("pos"); l.kind = Light::Point; l.v = triple(); ws1();
                keyword("falloff"); Vec3 f = triple(); l.falloff = {f[0], f[1], f[2]}; ws1();
                keyword("colour"); Vec3 c = triple(); l.colour = {c[0], c[1], c[2]};
            } else break;
            scene.lights.push_back(l);
            ws();
            if (!skipTrivia1()) break;
        }
        if (!eof()) fail("end of input");
    }
};

} // namespace

bool readWholeFile(const std::string& path, std::string& out) {
    out.clear();
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    const bool ok = !std::ferror(f);                                // a directory opens but cannot be read (EISDIR)
    std::fclose(f);
    return ok;
}

bool parseScene(const std::string& text, const std::string& baseDir, SceneOptions& options, Scene& scene, std::string& error) {
    Parser p(text, baseDir);
    try { p.run(options, scene); }
    catch (const ParseError& e) { error = e.msg; return false; }
    catch (const std::exception& e) { error = std::string("scene parser: ") + e.what(); return false; }   // nothing may cross the C boundary
    return true;
}

bool parseColour(const std::string& text, Colour& out) {
    Parser p(text, "");
    try { out = p.colour(); } catch (const ParseError&) { return false; }
    return true;
}

// ------------------------------------------------------------------ PlyParser.fs:14-69
bool parsePly(const std::string& text, std::vector<Triangle>& triangles, std::string& error) {
    std::istringstream in(text);
    std::string line;
    auto getl = [&](std::string& l) { if (!std::getline(in, l)) return false; if (!l.empty() && l.back() == '\r') l.pop_back(); return true; };
    if (!getl(line) || line != "ply") { error = "PLY: expecting 'ply'"; return false; }           // magicNumber, :20
    long vertexCount = -1, faceCount = -1;
    bool ended = false;
    while (getl(line)) {                                                                          // pheader, :31-40
        if (line.rfind("format", 0) == 0 || line.rfind("comment", 0) == 0 || line.rfind("property", 0) == 0) continue;
        if (line.rfind("element vertex ", 0) == 0) { vertexCount = std::atol(line.c_str() + 15); continue; }
        if (line.rfind("element face ", 0) == 0) { faceCount = std::atol(line.c_str() + 13); continue; }
        if (line == "end_header") { ended = true; break; }
        error = "PLY: unexpected header line: " + line; return false;
    }
    if (!ended || vertexCount < 0 || faceCount < 0) { error = "PLY: incomplete header"; return false; }
    std::vector<Vec3> vertexes((size_t)vertexCount);
    for (long k = 0; k < vertexCount; ++k) {                                                      // pvertex, :42-49: x y z confidence intensity
        if (!getl(line)) { error = "PLY: missing vertex line"; return false; }
        const char* c = line.c_str(); char* end = nullptr;
        double v[5];
        for (int j = 0; j < 5; ++j) {
            v[j] = std::strtod(c, &end);
            if (end == c) { error = "PLY: vertex line needs 5 numbers (x y z confidence intensity)"; return false; }
            c = end;
            if (j < 4) { if (*c != ' ') { error = "PLY: vertex fields are separated by one space"; return false; } ++c; }
        }
        vertexes[(size_t)k] = {v[0], v[1], v[2]};
    }
    triangles.clear(); triangles.reserve((size_t)faceCount);
    for (long k = 0; k < faceCount; ++k) {                                                        // pface, :51-57: "3 a b c"
        if (!getl(line)) { error = "PLY: missing face line"; return false; }
        if (line.rfind("3 ", 0) != 0) { error = "PLY: only triangular faces ('3 a b c') are accepted"; return false; }
        long idx[3]; const char* c = line.c_str() + 2; char* end = nullptr;
        for (int j = 0; j < 3; ++j) {
            idx[j] = std::strtol(c, &end, 10);
            if (end == c || idx[j] < 0 || idx[j] >= vertexCount) { error = "PLY: bad face index"; return false; }
            c = end; if (j < 2) { if (*c != ' ') { error = "PLY: face indices are separated by one space"; return false; } ++c; }
        }
        triangles.push_back({vertexes[(size_t)idx[0]], vertexes[(size_t)idx[1]], vertexes[(size_t)idx[2]]});
    }
    return true;
}

} // namespace FuncTracer
