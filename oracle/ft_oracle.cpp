// ft_oracle.cpp — CPU restatement of FuncTracer's render loop.  TEST INFRASTRUCTURE ONLY
// (see ft_oracle.h).  Every function cites the reference file:line it follows; citations are
// relative to /root/reference/FuncTracer/.  The structure deliberately mirrors the reference:
// a Geometry is "ray -> all intersections along the infinite line" (Ray.fs:31), transforms are
// applied per nesting level with 4x4 matrices (Transform.fs:80-87), CSG merges stable-sorted hit
// lists (Csg.fs:74-94), closest = stable sort + skip negatives + head (Scene.fs:112-116) and
// shading is the literal recursion of Shading.fs:131-139.
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off (no FMA contraction, SSE2 doubles) — the .NET JIT
// the reference runs on does not contract either.
//
// Known deviations from a bit-exact .NET run (all far below the 1e-4 parity tolerance):
//   * Math.quadratic uses `b ** 2.0` (Math.Pow, Math.fs:5); restated as b*b.
//   * libm sqrt/tan/sin/cos/pow/atan2/asin/acos stand in for System.Math (<= 1 ulp apart).
//   * Sorting sequences that contain NaN keys is left undefined (the reference's behaviour
//     there depends on F#'s generic comparison of NaN).
//   * System.Random streams are unseeded in the reference (Image.fs:101, Jitter.fs:27): the
//     jitter pattern is an explicit input; soft lights / depth of field draw from the seeded
//     counter-based stream documented at `struct Rng` below.
#include "ft_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- CommonTypes.fs
struct V3 { double x, y, z; };                                   // Vector / Point / Colour (CommonTypes.fs:4,28,42)
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }          // :5-6, :29-30, :44-45
inline V3 scale(V3 v, double s) { return {s * v.x, s * v.y, s * v.z}; }          // :7-10
inline V3 neg(V3 v) { return {-v.x, -v.y, -v.z}; }                               // :11-12
inline V3 sub(V3 a, V3 b) { return add(a, neg(b)); }                             // :13-14 (v1 + -v2)
inline V3 psub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }         // Point - Point :34-35
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // :15-16
inline V3 cross(V3 a, V3 b) {                                                    // :17-18
    return {a.y * b.z - a.z * b.y, b.x * a.z - b.z * a.x, a.x * b.y - a.y * b.x};
}
inline double length(V3 v) { return std::sqrt(dot(v, v)); }                      // :19
inline V3 normalise(V3 v) {                                                      // :63-67
    double l = length(v);
    if (l < 0.0000001) return v;
    return scale(v, 1.0 / l);
}
inline V3 reflect(V3 n, V3 v) { return sub(v, scale(n, 2.0 * dot(v, n))); }      // :72  v-(2.0*(v.*n)*n)
inline double angleBetween(V3 a, V3 b) { return std::acos(dot(normalise(a), normalise(b))); } // :74-75
inline V3 perpendicularComponent(V3 a, V3 b) {                                   // :77-79
    V3 na = normalise(a);
    return sub(b, scale(na, dot(b, na)));
}
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }         // Colour (*) :46-47
inline V3 hueShiftColour(V3 c) { return {c.z, c.x, c.y}; }                       // :90 (b,r,g)
const double PI = 3.14159265358979323846;                                        // System.Math.PI

// ---------------------------------------------------------------- Math.fs
inline int quadratic(double a, double b, double c, double roots[2]) {            // Math.fs:4-10
    double discriminant = b * b - 4.0 * a * c;                                   // b ** 2.0 (see header note)
    if (discriminant < 0.0) return 0;
    double sq = std::sqrt(discriminant);
    double twoa = 2.0 * a;
    roots[0] = (-b + sq) / twoa;
    roots[1] = (-b - sq) / twoa;
    return 2;
}
inline double clamp01(double x) { if (x > 1.0) return 1.0; if (x < 0.0) return 0.0; return x; } // Math.fs:12-16

// ---------------------------------------------------------------- Ray.fs
struct Material { V3 colour; double roughness, reflectance, shineyness; bool applyLighting; }; // Ray.fs:4-10
const Material mattWhite = {{1.0, 1.0, 1.0}, 0.0, 0.0, 0.0, true};                             // Ray.fs:11
struct Ray { V3 o, d; };                                                                       // Ray.fs:13
struct Hit { double t; V3 p; V3 n; Material material; double u, v; };                          // Ray.fs:21-27
inline Hit newIntersection() { return {0.0, {0, 0, 0}, {1, 0, 0}, mattWhite, 0.0, 0.0}; }      // Ray.fs:29
typedef std::vector<Hit> Hits;
typedef std::function<void(const Ray&, Hits&)> Geometry;                                       // Ray.fs:31 (appends)

Geometry group(std::vector<Geometry> xs) {                                                     // Ray.fs:34
    return [xs](const Ray& r, Hits& out) { for (auto& g : xs) g(r, out); };
}
template <class F> Geometry mapHits(Geometry g, F f) {   // "g >> Seq.map f"
    return [g, f](const Ray& r, Hits& out) {
        size_t first = out.size();
        g(r, out);
        for (size_t i = first; i < out.size(); ++i) f(out[i]);
    };
}
Geometry flipNormals(Geometry g) { return mapHits(g, [](Hit& h) { h.n = scale(h.n, -1.0); }); }      // Ray.fs:36
Geometry ignoreLight(Geometry g) { return mapHits(g, [](Hit& h) { h.material.applyLighting = false; }); } // Ray.fs:47
Geometry setMaterial(Material m, Geometry g) { return mapHits(g, [m](Hit& h) { h.material = m; }); }  // Ray.fs:49
Geometry hueShift(Geometry g) { return mapHits(g, [](Hit& h) { h.material.colour = hueShiftColour(h.material.colour); }); } // Ray.fs:51-55
typedef std::function<V3(double, double)> Texture;
Geometry textureDiffuse(Texture tex, Geometry g) {                                                   // Ray.fs:57-59
    return mapHits(g, [tex](Hit& h) { h.material.colour = tex(h.u, h.v); });
}

// ---------------------------------------------------------------- Textures/Texture.fs
inline double repeatOne(double x) {                                              // Texture.fs:9-11
    double a = std::fabs(x - std::floor(x));
    return (a < 0.0) ? 1.0 - a : a;
}
Texture gridTexture(V3 c1, V3 c2) {                                              // Texture.fs:24-29
    return [c1, c2](double u0, double v0) {
        double u = repeatOne(u0), v = repeatOne(v0);
        if (u < 0.5 && v < 0.5) return c1;
        if (u < 0.5) return c2;
        if (u > 0.5 && v > 0.5) return c1;
        return c2;
    };
}
// ImageTexture.image (Textures/Image.fs:20-36): nearest texel of an Rgb24 image.  Where the reference would raise
// IndexOutOfRange (index past the pixel array) the last texel is returned - the one stated deviation, shared with the device path.
Texture imageTexture(std::shared_ptr<const std::vector<uint8_t>> pixels, int width, int height) {
    return [pixels, width, height](double u0, double v0) {
        double u = repeatOne(u0), v = repeatOne(v0);
        double x = std::floor(u * (double)width), y = std::floor(v * (double)height);
        double index = y * (3.0 * (double)width) + 3.0 * x;
        double last = 3.0 * ((double)width * (double)height - 1.0);
        if (!(index >= 0.0)) index = 0.0;
        if (index > last) index = last;
        size_t i = (size_t)index;
        return V3{(double)(*pixels)[i] / 255.0, (double)(*pixels)[i + 1] / 255.0, (double)(*pixels)[i + 2] / 255.0};
    };
}

// ---------------------------------------------------------------- Transform.fs
struct M4 { double m[4][4]; };                                                   // Transform.fs:7-9
inline M4 mmul(const M4& a, const M4& b) {                                       // :11-14 (List.sumBy from 0)
    M4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
            for (int c = 0; c < 4; ++c) s = s + a.m[j][c] * b.m[c][i];
            r.m[j][i] = s;
        }
    return r;
}
inline V3 mulV(const M4& m, V3 v) {                                              // :15-18
    return {m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z,
            m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
            m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z};
}
inline V3 mulP(const M4& m, V3 p) {                                              // :19-22
    return {m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3],
            m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3],
            m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3]};
}
struct Xf {                                                                      // Transform.fs:25-30
    enum Kind { Translate, Scale, Rotate, Composed } kind;
    V3 v; double angle; std::vector<Xf> ts;
};
Xf xfTranslate(V3 v) { return {Xf::Translate, v, 0.0, {}}; }                     // :32
Xf xfScale(V3 s) { return {Xf::Scale, s, 0.0, {}}; }                             // :35
Xf xfRotate(V3 axis, double angle) { return {Xf::Rotate, normalise(axis), angle, {}}; } // :37-38
Xf xfCompose(const std::vector<Xf>& ts) {                                        // :41-45
    Xf r{Xf::Composed, {0, 0, 0}, 0.0, {}};
    for (auto& t : ts) { if (t.kind == Xf::Composed) for (auto& u : t.ts) r.ts.push_back(u); else r.ts.push_back(t); }
    return r;
}
Xf xfInverse(const Xf& t) {                                                      // :47-51
    switch (t.kind) {
        case Xf::Translate: return {Xf::Translate, neg(t.v), 0.0, {}};
        case Xf::Scale: return {Xf::Scale, {1.0 / t.v.x, 1.0 / t.v.y, 1.0 / t.v.z}, 0.0, {}};
        case Xf::Rotate: return {Xf::Rotate, t.v, -t.angle, {}};
        default: {
            Xf r{Xf::Composed, {0, 0, 0}, 0.0, {}};
            for (auto it = t.ts.rbegin(); it != t.ts.rend(); ++it) r.ts.push_back(xfInverse(*it));
            return r;
        }
    }
}
M4 identity() { M4 r; for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) r.m[j][i] = (i == j) ? 1.0 : 0.0; return r; } // :53
M4 matrix(const Xf& t) {                                                         // :55-71
    M4 r;
    switch (t.kind) {
        case Xf::Translate:
            for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i)
                r.m[j][i] = (i == j) ? 1.0 : (i < 3) ? 0.0 : (j == 0) ? t.v.x : (j == 1) ? t.v.y : t.v.z;
            return r;
        case Xf::Scale:
            for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i)
                r.m[j][i] = (i != j) ? 0.0 : (i == 0) ? t.v.x : (i == 1) ? t.v.y : (i == 2) ? t.v.z : 1.0;
            return r;
        case Xf::Rotate: {
            double ux = t.v.x, uy = t.v.y, uz = t.v.z;
            double c = std::cos(t.angle), invc = 1.0 - c, s = std::sin(t.angle);
            double rows[4][4] = {
                {c + invc * ux * ux, invc * ux * uy - s * uz, invc * ux * uz + s * uy, 0.0},
                {invc * ux * uy + s * uz, c + invc * uy * uy, invc * uy * uz - s * ux, 0.0},
                {invc * ux * uz - s * uy, invc * uy * uz + s * ux, c + invc * uz * uz, 0.0},
                {0.0, 0.0, 0.0, 1.0}};
            std::memcpy(r.m, rows, sizeof rows);
            return r;
        }
        default: {
            // List.foldBack (*) (ts |> List.rev |> List.map matrix) identity = M_n * (... * (M_1 * I))
            M4 acc = identity();
            for (auto& u : t.ts) acc = mmul(matrix(u), acc);
            return acc;
        }
    }
}
M4 transpose(const M4& a) { M4 r; for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) r.m[j][i] = a.m[i][j]; return r; } // :73-74
Geometry transform(const Xf& t, Geometry object) {                               // :80-87
    M4 modelToWorld = matrix(t);
    M4 worldToModel = matrix(xfInverse(t));
    M4 normalToWorld = transpose(worldToModel);                                  // inverse >> matrix >> transpose, :77-78
    return [=](const Ray& r, Hits& out) {
        Ray r2{mulP(worldToModel, r.o), mulV(worldToModel, r.d)};
        size_t first = out.size();
        object(r2, out);
        for (size_t i = first; i < out.size(); ++i) {
            out[i].p = mulP(modelToWorld, out[i].p);
            out[i].n = normalise(mulV(normalToWorld, out[i].n));
        }
    };
}

// ---------------------------------------------------------------- Sphere.fs
void sphere(const Ray& r, Hits& out) {                                           // Sphere.fs:11-21
    V3 ov = r.o;
    double a = dot(r.d, r.d);
    double b = 2.0 * dot(ov, r.d);
    double c = dot(ov, ov) - 1.0;
    double roots[2];
    int n = quadratic(a, b, c, roots);
    for (int i = 0; i < n; ++i) {
        double t = roots[i];
        Hit h = newIntersection();
        h.t = t; h.p = add(r.o, scale(r.d, t)); h.n = normalise(h.p);
        h.u = 0.5 + std::atan2(h.n.z, h.n.x) / (2.0 * PI);                      // setUV, Sphere.fs:6-10
        h.v = 0.5 - std::asin(h.n.y) / PI;
        out.push_back(h);
    }
}

// ---------------------------------------------------------------- Plane.fs
struct Plane { V3 p0, n; };
bool planeIntersect(const Plane& pl, const Ray& r, Hit& h) {                     // Plane.fs:9-20
    const double eps = 0.0000001;
    double num = dot(psub(pl.p0, r.o), pl.n);
    double denom = dot(r.d, pl.n);
    h = newIntersection();
    if (std::fabs(denom) < eps) {
        if (num < eps) { h.t = 0.0; h.p = r.o; h.n = pl.n; return true; }
        return false;
    }
    double t = num / denom;
    h.t = t; h.p = add(r.o, scale(r.d, t)); h.n = pl.n;
    return true;
}
inline bool isAbove(const Plane& pl, V3 point) { return dot(psub(point, pl.p0), pl.n) >= 0.0; }  // Plane.fs:22-23
void plane(const Ray& r, Hits& out) {                                            // Plane.fs:28-33
    Hit h;
    if (planeIntersect({{0, 0, 0}, {0, 1, 0}}, r, h)) { h.u = h.p.x; h.v = h.p.z; out.push_back(h); }
}

// ---------------------------------------------------------------- Cube.fs
void square(const Ray& r, Hits& out) {                                           // Cube.fs:9-15
    size_t first = out.size();
    plane(r, out);
    if (out.size() > first) {
        V3 p = out.back().p;
        if (!((p.x >= 0.0) && (p.x <= 1.0) && (p.z >= 0.0) && (p.z <= 1.0))) out.pop_back();
    }
}
double degToRad(double d) { return d * (PI / 180.0); }                           // CommonTypes.fs:98-99
Geometry makeCube() {                                                            // Cube.fs:17-25
    Geometry sq = square;
    Geometry bottom = flipNormals(sq);
    Geometry top = transform(xfTranslate({0.0, 1.0, 0.0}), sq);
    Geometry left = transform(xfRotate({0, 0, 1}, degToRad(90.0)), sq);
    Geometry right = flipNormals(transform(xfTranslate({1, 0, 0}), left));
    Geometry front = transform(xfRotate({1, 0, 0}, degToRad(-90.0)), sq);
    Geometry back = flipNormals(transform(xfTranslate({0, 0, 1}), front));
    Geometry combined = group({bottom, top, left, right, front, back});
    return transform(xfTranslate({-0.5, -0.5, -0.5}), combined);
}

// ---------------------------------------------------------------- Cone.fs
void cone(const Ray& r, Hits& out) {                                             // Cone.fs:7-27
    double ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dy = r.d.y, dz = r.d.z;
    oy = oy - 1.0;
    double a = dx * dx + dz * dz - dy * dy;
    double b = 2.0 * (ox * dx + oz * dz - oy * dy);
    double c = ox * ox + oz * oz - oy * oy;
    double roots[2];
    int n = quadratic(a, b, c, roots);
    for (int i = 0; i < n; ++i) {
        double t = roots[i];
        V3 q = add(V3{ox, oy, oz}, scale(r.d, t));
        V3 p = {q.x, q.y + 1.0, q.z};
        V3 nn = normalise(V3{q.x, -q.y, q.z});
        Hit h = newIntersection();
        h.t = t; h.p = p; h.n = (dot(nn, r.d) < 0.0) ? nn : neg(nn);
        if (h.p.y >= 0.0 && h.p.y <= 1.0) out.push_back(h);
    }
}

// ---------------------------------------------------------------- Cylinder.fs
void cylinder(const Ray& r, Hits& out) {                                         // Cylinder.fs:8-20
    double ox = r.o.x, oz = r.o.z, dx = r.d.x, dz = r.d.z;
    double a = dx * dx + dz * dz;
    double b = 2.0 * (ox * dx + oz * dz);
    double c = ox * ox + oz * oz - 1.0;
    double roots[2];
    int n = quadratic(a, b, c, roots);
    for (int i = 0; i < n; ++i) {
        double t = roots[i];
        V3 p = add(r.o, scale(r.d, t));
        V3 nn = normalise(V3{p.x, 0.0, p.z});
        Hit h = newIntersection();
        h.t = t; h.p = p; h.n = (dot(nn, r.d) < 0.0) ? nn : neg(nn);
        if (p.y >= 0.0 && p.y <= 1.0) out.push_back(h);
    }
}
void circle(const Ray& r, Hits& out) {                                           // Cylinder.fs:22
    size_t first = out.size();
    plane(r, out);
    if (out.size() > first) {
        if (!(length(psub(out.back().p, V3{0, 0, 0})) < 1.0)) out.pop_back();
    }
}
Geometry makeSolidCylinder() {                                                   // Cylinder.fs:25-29
    Geometry top = transform(xfTranslate({0.0, 1.0, 0.0}), circle);
    Geometry bottom = transform(xfRotate({0.0, 0.0, 1.0}, degToRad(180.0)), circle);
    Geometry sides = cylinder;
    return group({top, bottom, sides});
}

// ---------------------------------------------------------------- Triangle.fs
struct Tri { V3 a, b, c; };
struct BuildError { std::string msg; };
V3 edgeIntersection(const Plane& p, V3 a, V3 b) {                                // Triangle.fs:8-10
    Hit h;
    if (!planeIntersect(p, Ray{a, normalise(psub(b, a))}, h))
        throw BuildError{"Triangle.edgeIntersection: edge parallel to the split plane has no intersection (Option.Value on None, Triangle.fs:10)"};
    return h.p;
}
void slicePrime(const Plane& plane, const Tri& t, std::vector<Tri>& single, std::vector<Tri>& two) { // Triangle.fs:13-22
    V3 a = t.a, b = t.b, c = t.c;
    single.push_back({a, edgeIntersection(plane, a, b), edgeIntersection(plane, a, c)});
    two.push_back({edgeIntersection(plane, b, a), b, c});
    two.push_back({c, edgeIntersection(plane, c, a), edgeIntersection(plane, b, a)});
}
void slice(const Plane& plane, const Tri& t, std::vector<Tri>& above, std::vector<Tri>& below) { // Triangle.fs:24-41
    bool aAbove = isAbove(plane, t.a), bAbove = isAbove(plane, t.b), cAbove = isAbove(plane, t.c);
    std::vector<Tri> fst, snd;
    if (aAbove == bAbove && bAbove == cAbove) {
        fst.push_back(t);
    } else if (aAbove == bAbove) {
        slicePrime(plane, {t.c, t.a, t.b}, snd, fst);                            // |> flip true
    } else if (aAbove == cAbove) {
        slicePrime(plane, {t.b, t.c, t.a}, snd, fst);                            // |> flip true
    } else {
        slicePrime(plane, {t.a, t.b, t.c}, fst, snd);
    }
    if (!aAbove) std::swap(fst, snd);                                            // |> flip (not aAbove)
    for (auto& x : fst) above.push_back(x);
    for (auto& x : snd) below.push_back(x);
}
bool triangleHit(const Tri& tri, const Ray& ray, Hit& h) {                       // Triangle.fs:43-66
    const double epsilon = 0.0000001;
    V3 edge1 = psub(tri.b, tri.a);
    V3 edge2 = psub(tri.c, tri.a);
    V3 hh = cross(ray.d, edge2);
    double a = dot(edge1, hh);
    if (a > -epsilon && a < epsilon) return false;
    double f = 1.0 / a;
    V3 s = psub(ray.o, tri.a);
    double u = f * dot(s, hh);
    if (u < 0.0 || u > 1.0) return false;
    V3 q = cross(s, edge1);
    double v = f * dot(ray.d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(edge2, q);
    if (t > epsilon) {
        h = newIntersection();
        h.t = t;
        h.p = add(ray.o, scale(normalise(ray.d), t * length(ray.d)));
        h.n = normalise(cross(edge1, edge2));
        return true;
    }
    return false;
}

// ---------------------------------------------------------------- BoundingBox.fs
struct Aabb { V3 min, max; };
Aabb pointsBoundry(const std::vector<Tri>& tris) {                               // BoundingBox.fs:9-22 via BspMesh.fs:49
    const double inf = std::numeric_limits<double>::infinity();
    Aabb b{{inf, inf, inf}, {-inf, -inf, -inf}};
    auto acc = [&](V3 p) {
        b.min.x = std::min(b.min.x, p.x); b.min.y = std::min(b.min.y, p.y); b.min.z = std::min(b.min.z, p.z);
        b.max.x = std::max(b.max.x, p.x); b.max.y = std::max(b.max.y, p.y); b.max.z = std::max(b.max.z, p.z);
    };
    for (auto& t : tris) { acc(t.a); acc(t.b); acc(t.c); }
    return b;
}
inline double fsMax(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? std::numeric_limits<double>::quiet_NaN() : (a < b ? b : a); } // F# max on float = Math.Max
inline double fsMin(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? std::numeric_limits<double>::quiet_NaN() : (a < b ? a : b); }
bool aabbIntersects(const Aabb& box, const Ray& ray) {                           // BoundingBox.fs:32-58
    const double inf = std::numeric_limits<double>::infinity();
    double t0 = -inf, t1 = inf;
    const V3 bounds[2] = {box.min, box.max};
    V3 inv = {1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z};
    int sign[3] = {inv.x < 0.0 ? 1 : 0, inv.y < 0.0 ? 1 : 0, inv.z < 0.0 ? 1 : 0};
    double tmin = (bounds[sign[0]].x - ray.o.x) * inv.x;
    double tmax = (bounds[1 - sign[0]].x - ray.o.x) * inv.x;
    double tymin = (bounds[sign[1]].y - ray.o.y) * inv.y;
    double tymax = (bounds[1 - sign[1]].y - ray.o.y) * inv.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    tmin = fsMax(tymin, tmin);
    tmax = fsMin(tymax, tmax);
    double tzmin = (bounds[sign[2]].z - ray.o.z) * inv.z;
    double tzmax = (bounds[1 - sign[2]].z - ray.o.z) * inv.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    tmin = fsMax(tzmin, tmin);
    tmax = fsMin(tzmax, tmax);
    return (tmin < t1) && (tmax > t0);
}

// ---------------------------------------------------------------- BspMesh.fs
// (struct-of-arrays copy of a leaf's triangles: see triangleCandidates below)
struct TriSoA {
    std::vector<double> ax, ay, az, e1x, e1y, e1z, e2x, e2y, e2z;
    void build(const std::vector<Tri>& ts) {
        for (auto& t : ts) {
            V3 e1 = psub(t.b, t.a), e2 = psub(t.c, t.a);
            ax.push_back(t.a.x); ay.push_back(t.a.y); az.push_back(t.a.z);
            e1x.push_back(e1.x); e1y.push_back(e1.y); e1z.push_back(e1.z); e2x.push_back(e2.x); e2y.push_back(e2.y); e2z.push_back(e2.z);
        }
    }
};
struct BspNode {                                                                 // BspMesh.fs:12-19
    bool isLeaf; std::vector<Tri> tris;       // Leaf of Geometry = group of triangles
    TriSoA soa;
    Aabb aabb; std::unique_ptr<BspNode> left, right;
};
void optimalSplit(const Aabb& aabb, const std::vector<Tri>& tris, std::vector<Tri>& left, std::vector<Tri>& right) { // :30-46
    double widthx = std::fabs(aabb.max.x - aabb.min.x) / 2.0;
    double widthy = std::fabs(aabb.max.y - aabb.min.y) / 2.0;
    double widthz = std::fabs(aabb.max.z - aabb.min.z) / 2.0;
    Plane pl;
    if (widthx > widthy && widthx > widthz) pl = {{(aabb.min.x + aabb.max.x) / 2.0, 0.0, 0.0}, {1, 0, 0}};
    else if (widthy > widthz) pl = {{0.0, (aabb.min.y + aabb.max.y) / 2.0, 0.0}, {0, 1, 0}};
    else pl = {{0.0, 0.0, (aabb.min.z + aabb.max.z) / 2.0}, {0, 0, 1}};
    for (auto& t : tris) slice(pl, t, left, right);
}
std::unique_ptr<BspNode> compile(int maxDepth, const std::vector<Tri>& tris) {   // BspMesh.fs:51-65
    auto node = std::make_unique<BspNode>();
    auto makeLeaf = [&]() { node->isLeaf = true; node->tris = tris; node->soa.build(tris); };
    if (maxDepth == 0) { makeLeaf(); return node; }
    Aabb aabb = pointsBoundry(tris);
    std::vector<Tri> left, right;
    optimalSplit(aabb, tris, left, right);
    size_t triCount = tris.size();
    if (left.size() >= triCount || right.size() >= triCount) { makeLeaf(); return node; }
    node->isLeaf = false; node->aabb = aabb;
    node->left = compile(maxDepth - 1, left);
    node->right = compile(maxDepth - 1, right);
    return node;
}
// Speed only (the oracle doubles as bench.py's CPU baseline): the first two rejections of triangleHit - |a| < epsilon, u outside [0,1] -
// evaluated for a whole leaf at once over a struct-of-arrays copy of its triangles, with exactly the operations triangleHit performs
// (same products, same order, no contraction: the build has -ffp-contract=off), so that the compiler can use vector registers.  The
// survivors go through triangleHit itself, from the top; a triangle this pass turns away is one triangleHit would have turned away at
// the same comparison, so the hit sequence is unchanged bit for bit (tests/test_oracle_golden.py, tests/golden/frames.npz).
__attribute__((target_clones("avx512f", "avx2", "default"), optimize("O3")))
void triangleCandidates(const TriSoA& s, size_t n, const Ray& ray, unsigned char* keep) {
    const double epsilon = 0.0000001;
    const double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z, ox = ray.o.x, oy = ray.o.y, oz = ray.o.z;
    const double *ax = s.ax.data(), *ay = s.ay.data(), *az = s.az.data(), *e1x = s.e1x.data(), *e1y = s.e1y.data(), *e1z = s.e1z.data(),
                 *e2x = s.e2x.data(), *e2y = s.e2y.data(), *e2z = s.e2z.data();
    for (size_t i = 0; i < n; ++i) {
        const double hx = dy * e2z[i] - dz * e2y[i], hy = e2x[i] * dz - e2z[i] * dx, hz = dx * e2y[i] - dy * e2x[i];   // cross(ray.d, edge2)
        const double a = e1x[i] * hx + e1y[i] * hy + e1z[i] * hz;                                                        // dot(edge1, hh)
        const double f = 1.0 / a;
        const double sx = ox - ax[i], sy = oy - ay[i], sz = oz - az[i];
        const double u = f * (sx * hx + sy * hy + sz * hz);
        keep[i] = !((a > -epsilon) & (a < epsilon)) & !((u < 0.0) | (u > 1.0));
    }
}
void leafHits(const BspNode& n, const Ray& r, Hits& out) {                       // Leaf(triangles |> Seq.map triangle |> group), :53
    Hit h;
    const size_t count = n.tris.size();
    if (count < 16) { for (auto& t : n.tris) if (triangleHit(t, r, h)) out.push_back(h); return; }
    static thread_local std::vector<unsigned char> keep;
    if (keep.size() < count) keep.resize(count);
    triangleCandidates(n.soa, count, r, keep.data());
    for (size_t i = 0; i < count; ++i) if (keep[i] && triangleHit(n.tris[i], r, h)) out.push_back(h);
}
void bspIntersect(const BspNode& tree, const Ray& r, Hits& out) {                // BspMesh.fs:67-76 (tree is a Branch)
    if (!aabbIntersects(tree.aabb, r)) return;
    auto node = [&](const BspNode& n) { if (n.isLeaf) leafHits(n, r, out); else bspIntersect(n, r, out); };
    node(*tree.right);                                                           // left |> Seq.append right  ⇒ right first
    node(*tree.left);
}
int bspMaxDepth(const BspNode& n) { return n.isLeaf ? 0 : 1 + std::max(bspMaxDepth(*n.left), bspMaxDepth(*n.right)); } // :78-81
void bspLeafStats(const BspNode& n, int64_t& leaves, int64_t& tris) {            // :83-86
    if (n.isLeaf) { leaves++; tris += (int64_t)n.tris.size(); } else { bspLeafStats(*n.left, leaves, tris); bspLeafStats(*n.right, leaves, tris); }
}
Geometry bspMesh(int depth, const std::vector<Tri>& tris) {                      // BspMesh.fs:88-97
    std::shared_ptr<BspNode> tree(compile(depth, tris).release());
    if (tree->isLeaf) return [tree](const Ray& r, Hits& out) { leafHits(*tree, r, out); };
    return [tree](const Ray& r, Hits& out) { bspIntersect(*tree, r, out); };
}

// ---------------------------------------------------------------- Csg.fs
enum IType { OutsideIntoA, OutsideIntoB, BIntoAB, AIntoAB, ABleaveA, ABleaveB, AIntoOutside, BIntoOutside }; // Csg.fs:5-13
enum Rule { Take, Discard, Flip };                                               // Csg.fs:15
Rule unionRules(IType t) { switch (t) { case OutsideIntoA: case OutsideIntoB: case AIntoOutside: case BIntoOutside: return Take; default: return Discard; } } // :19-25
Rule subtractRules(IType t) { switch (t) { case OutsideIntoA: return Take; case AIntoAB: return Flip; case ABleaveB: return Flip; case AIntoOutside: return Take; default: return Discard; } } // :27-33
Rule intersectRules(IType t) { switch (t) { case BIntoAB: case AIntoAB: case ABleaveA: case ABleaveB: return Take; default: return Discard; } } // :35-44
Rule excludeRules(IType t) { switch (t) { case OutsideIntoA: case OutsideIntoB: case AIntoOutside: case BIntoOutside: return Take; default: return Flip; } } // :46-55
IType getIntersectionType(bool hitA, bool inA, bool inB) {                       // Csg.fs:59-72
    if (hitA) {
        if (inA && inB) return ABleaveA;
        if (!inA && inB) return BIntoAB;
        if (inA && !inB) return AIntoOutside;
        return OutsideIntoA;
    }
    if (inA && inB) return ABleaveB;
    if (!inA && inB) return BIntoOutside;
    if (inA && !inB) return AIntoAB;
    return OutsideIntoB;
}
Geometry constructedSolid(Rule (*rules)(IType), Geometry a, Geometry b) {        // Csg.fs:74-94
    return [=](const Ray& r, Hits& out) {
        Hits ha, hb;
        a(r, ha); b(r, hb);
        std::vector<std::pair<Hit, bool>> merged;                                // bool = HitA
        for (auto& h : ha) merged.push_back({h, true});
        for (auto& h : hb) merged.push_back({h, false});
        std::stable_sort(merged.begin(), merged.end(), [](const auto& x, const auto& y) { return x.first.t < y.first.t; }); // Seq.sortBy is stable
        bool insideA = false, insideB = false;
        for (auto& e : merged) {
            IType it = getIntersectionType(e.second, insideA, insideB);
            Rule action = rules(it);
            if (e.second) insideA = !insideA; else insideB = !insideB;
            if (action == Take) out.push_back(e.first);
            else if (action == Flip) { Hit h = e.first; h.n = scale(h.n, -1.0); out.push_back(h); }
        }
    };
}

// ---------------------------------------------------------------- Light.fs
struct Light { int kind; V3 v; double falloff[3]; V3 colour; int samples; double scatter; }; // 0 directional, 1 soft, 2 point
inline double attenuate(const double f[3], double distance) { return 1.0 / (f[0] + distance * (f[1] + distance * f[2])); } // Light.fs:16-17

// ---------------------------------------------------------------- Jitter.fs on a seeded counter-based stream
// The reference draws from `System.Random()` (unseeded: Jitter.fs:27, Image.fs:101), so no two runs of it
// agree.  Here every draw is a pure function of (seed, sample id, depth, light index, purpose, draw number):
//   key = sm64(sm64(sm64(seed ^ sample) ^ (depth << 32 | light << 8 | purpose)));  u_n = (sm64(key + n) >> 11) * 2^-53
// with sm64 = splitmix64's output function.  The same definition is implemented independently in the HIP
// path (ft_kernels.hip); sample = pixel_id * spp + s, purpose 1 = soft shadow, 2 = depth of field.
inline uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    uint64_t key, n;
    Rng(uint64_t seed, uint64_t sample, uint32_t depth, uint32_t light, uint32_t purpose)
        : key(sm64(sm64(sm64(seed ^ sample) ^ (((uint64_t)depth << 32) | ((uint64_t)light << 8) | purpose)))), n(0) {}
    double nextDouble() { return (double)(sm64(key + n++) >> 11) * (1.0 / 9007199254740992.0); }   // [0,1) like Random.NextDouble
};
inline double uniform(Rng& r) { return 2.0 * r.nextDouble() - 1.0; }             // Jitter.fs:9-10
inline void jitterCircle(Rng& r, double& x, double& y) {                         // Jitter.circle, Jitter.fs:15-21 (rejection)
    for (;;) { x = uniform(r); y = uniform(r); if ((x * x + y * y) > 1.0) continue; return; }
}
std::vector<V3> jitterVector(Rng& random, int count, double maxAngle, V3 vector) {   // Jitter.fs:26-39
    V3 normalised = normalise(vector);
    double maxOffsetMagnitude = std::tan(maxAngle / 2.0);
    V3 generator = (normalised.x > 0.9) ? V3{0, 1, 0} : V3{1, 0, 0};
    V3 i = normalise(cross(generator, normalised));
    V3 j = cross(i, normalised);
    std::vector<V3> out;
    for (int k = 0; k < count; ++k) {                                            // Jitter.pattern random circle count
        double x, y; jitterCircle(random, x, y);
        out.push_back(normalise(add(add(normalised, scale(i, maxOffsetMagnitude * x)), scale(j, maxOffsetMagnitude * y))));
    }
    return out;
}
struct Stream { uint64_t seed, sample; int maxDepth; };                          // identifies the ray tree a draw belongs to

// ---------------------------------------------------------------- Scene.fs
bool closest(Hits& hits, Hit& out) {                                             // Scene.fs:112-116
    std::stable_sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) { return a.t < b.t; });
    for (auto& h : hits) { if (0.0 > h.t) continue; out = h; return true; }      // skipWhile (0.0 > t) on a sorted list, then head
    return false;
}
// NB: skipWhile on the sorted sequence == first element with !(0.0 > t) scanning from the front.

// ---------------------------------------------------------------- Image.fs
struct ImagePlane { V3 origin, originToCentre, i, j; int resH, resV; double pw, ph, tlx, tly; }; // Image.fs:55-63
ImagePlane imagePlaneCreate(const ft_camera& c, int resH, int resV) {            // Image.fs:48-53, 67-81
    V3 o{c.o[0], c.o[1], c.o[2]}, la{c.look_at[0], c.look_at[1], c.look_at[2]}, up{c.up[0], c.up[1], c.up[2]};
    V3 k = normalise(psub(la, o));
    V3 i = normalise(cross(up, k));
    V3 j = cross(k, i);
    double height = std::tan(c.fov_y / 2.0) * 2.0;
    double width = height * c.aspect_ratio;
    double pixelHeight = height / (double)(resH - 1);                            // sic: resH (Image.fs:71)
    double pixelWidth = width / (double)(resV - 1);                              // sic: resV (Image.fs:72)
    return {o, k, i, j, resH, resV, pixelWidth, pixelHeight, -width / 2.0 + pixelWidth / 2.0, height / 2.0 - pixelHeight / 2.0};
}
Ray rayThroughPixel(const ImagePlane& ip, int px, int py, double jitterX, double jitterY) { // Image.fs:83-89
    double centreX = ip.tlx + (double)px * ip.pw, centreY = ip.tly - (double)py * ip.ph;
    double jx = centreX + jitterX * ip.pw, jy = centreY + jitterY * ip.ph;
    V3 via = add(add(ip.originToCentre, scale(ip.i, jx)), scale(ip.j, jy));
    return {ip.origin, via};
}

// ---------------------------------------------------------------- scene graph nodes (Scene.fs:33-53)
struct Node {
    enum Kind { Prim, TriangleP, Mesh, Transform, MaterialF, HueShift, IgnoreLight, Group, Csg, Texture } kind;
    int prim = 0; Tri tri{}; int depth = 0; std::vector<Tri> tris;
    Xf xf{Xf::Composed, {0, 0, 0}, 0.0, {}}; Material mat = mattWhite; int op = 0;
    std::vector<int> children;
    V3 ca{}, cb{}; std::vector<double> uvOps;
    std::shared_ptr<const std::vector<uint8_t>> pixels; int imgW = 0, imgH = 0;      // Texture.Image
};

struct Counters { std::atomic<uint64_t> shadow{0}, reflect{0}; };

} // namespace

struct fto_context {
    std::vector<Node> nodes;
    int root = -1;
    std::vector<Light> lights;
    Geometry geometry;
    bool committed = false;
    std::string err;
    Counters counters;
};

namespace {

Geometry build(fto_context* ctx, int id) {                                       // Scene.intersect, Scene.fs:67-104
    const Node& n = ctx->nodes[id];
    switch (n.kind) {
        case Node::Prim:                                                         // intersectPrimitive, Scene.fs:20-30
            switch (n.prim) {
                case FT_PRIM_CIRCLE: return circle;
                case FT_PRIM_SQUARE: return square;
                case FT_PRIM_CUBE: return makeCube();
                case FT_PRIM_SPHERE: return sphere;
                case FT_PRIM_PLANE: return plane;
                case FT_PRIM_CONE: return cone;
                case FT_PRIM_SOLID_CYLINDER: return makeSolidCylinder();
                default: return cylinder;
            }
        case Node::TriangleP: { Tri t = n.tri; return [t](const Ray& r, Hits& out) { Hit h; if (triangleHit(t, r, h)) out.push_back(h); }; }
        case Node::Mesh: return bspMesh(n.depth, n.tris);
        case Node::Transform: return transform(n.xf, build(ctx, n.children[0]));
        case Node::MaterialF: return setMaterial(n.mat, build(ctx, n.children[0]));
        case Node::HueShift: return hueShift(build(ctx, n.children[0]));
        case Node::IgnoreLight: return ignoreLight(build(ctx, n.children[0]));
        case Node::Texture: {                                                    // Scene.fs:68-75
            Texture tex = n.pixels ? imageTexture(n.pixels, n.imgW, n.imgH) : gridTexture(n.ca, n.cb);
            // uvOps are listed outermost-first; each wraps the texture built so far from the inside out.
            for (int k = (int)n.uvOps.size() / 3 - 1; k >= 0; --k) {
                int kind = (int)n.uvOps[3 * k]; double a = n.uvOps[3 * k + 1], b = n.uvOps[3 * k + 2];
                Texture inner = tex;
                if (kind == 0) tex = [inner, a, b](double u, double v) { return inner(u / a, v / b); };          // Texture.scale, Texture.fs:14-16
                else { M4 m = matrix(xfRotate({0.0, 1.0, 0.0}, a));                                               // Texture.rotate, Texture.fs:18-22
                       tex = [inner, m](double u, double v) { V3 q = mulV(m, V3{u, 0.0, v}); return inner(q.x, q.z); }; }
            }
            return textureDiffuse(tex, build(ctx, n.children[0]));
        }
        case Node::Group: { std::vector<Geometry> gs; for (int c : n.children) gs.push_back(build(ctx, c)); return group(gs); }
        default: {
            Geometry a = build(ctx, n.children[0]), b = build(ctx, n.children[1]);
            switch (n.op) {
                case FT_CSG_UNION: return constructedSolid(unionRules, a, b);
                case FT_CSG_INTERSECT: return constructedSolid(intersectRules, a, b);
                case FT_CSG_SUBTRACT: return constructedSolid(subtractRules, a, b);
                default: return constructedSolid(excludeRules, a, b);
            }
        }
    }
}

// Speed only: a ray's hit list comes out of a per-thread pool instead of a fresh allocation (the recursion nests, so a pool, not one list).
struct HitsLease {
    static std::vector<Hits>& pool() { static thread_local std::vector<Hits> p; return p; }
    Hits hits;
    HitsLease() { auto& p = pool(); if (!p.empty()) { hits = std::move(p.back()); p.pop_back(); hits.clear(); } }
    ~HitsLease() { pool().push_back(std::move(hits)); }
};
bool lightIsBlocked(fto_context* ctx, double maxDistance, const Ray& ray) {      // Scene.fs:119-121
    ctx->counters.shadow.fetch_add(1, std::memory_order_relaxed);
    HitsLease lease; Hits& hits = lease.hits;
    ctx->geometry(ray, hits);
    for (auto& i : hits) if (i.t >= 0.0 && i.t < maxDistance && i.material.applyLighting) return true;
    return false;
}

// ---------------------------------------------------------------- Shading.fs
double shadowLightIntensity(fto_context* ctx, const Light& light, V3 point, const Stream& st, int depth, int lightIndex) {   // Shading.fs:24-42
    if (light.kind == 1) {                                                       // softShadowLightIntensity, Shading.fs:24-31
        Rng rng(st.seed, st.sample, (uint32_t)depth, (uint32_t)lightIndex, 1);
        int occluded = 0;
        for (V3 d : jitterVector(rng, light.samples, light.scatter, neg(light.v)))
            if (lightIsBlocked(ctx, std::numeric_limits<double>::max(), Ray{point, d})) ++occluded;
        return (double)(light.samples - occluded) / (double)light.samples;
    }
    if (light.kind == 0) return lightIsBlocked(ctx, std::numeric_limits<double>::max(), Ray{point, neg(light.v)}) ? 0.0 : 1.0;
    V3 d = psub(light.v, point);
    double distance = length(d);
    if (lightIsBlocked(ctx, distance, Ray{point, normalise(d)})) return 0.0;
    return attenuate(light.falloff, distance);
}
V3 lightDirection(const Light& light, V3 atPoint) {                              // Shading.fs:44-48
    if (light.kind == 2) return normalise(psub(atPoint, light.v));
    return light.v;
}
struct Fragment { const Hit* intersection; V3 lightColour, lightDirection; const Ray* viewRay; }; // Shading.fs:10-15

V3 roughDiffuse(const Fragment& f) {                                             // Shading.fs:50-63
    const Hit& ix = *f.intersection;
    double roughness = ix.material.roughness * ix.material.roughness;            // ** 2.0
    double rayAngle = angleBetween(ix.n, neg(f.viewRay->d));
    double lightAngle = angleBetween(ix.n, neg(f.lightDirection));
    double alpha = fsMax(rayAngle, lightAngle);
    double beta = fsMin(rayAngle, lightAngle);
    double A = 1.0 - 0.5 * roughness / (roughness + 0.33);
    double B = 0.45 * roughness / (roughness + 0.09);
    V3 tangentLight = normalise(perpendicularComponent(ix.n, neg(f.lightDirection)));
    V3 tangentRay = normalise(perpendicularComponent(ix.n, neg(f.viewRay->d)));
    double intensity = std::cos(lightAngle) * (A + (B * fsMax(0.0, dot(tangentLight, tangentRay)) * std::sin(alpha) * std::tan(beta)));
    return scale(ix.material.colour, intensity);                                 // scaleColour; light colour is NOT used (sic)
}
V3 lambertianDiffuse(const Fragment& f) {                                        // Shading.fs:65-70
    const Hit& ix = *f.intersection;
    double intensity = dot(neg(f.lightDirection), ix.n);
    return scale(cmul(ix.material.colour, f.lightColour), intensity);
}
V3 diffuseShader(const Fragment& f) {                                            // Shading.fs:72-76
    if (f.intersection->material.roughness == 0.0) return lambertianDiffuse(f);
    return roughDiffuse(f);
}
V3 specularShader(const Fragment& f) {                                           // Shading.fs:78-87
    const Hit& ix = *f.intersection;
    V3 normal = normalise(ix.n);
    double shineyness = ix.material.shineyness;
    V3 reflectedLightDirection = normalise(reflect(normal, f.lightDirection));
    V3 viewDirection = normalise(f.viewRay->d);
    double intensity = std::pow(dot(viewDirection, neg(reflectedLightDirection)), shineyness);
    if (shineyness <= 0.0 || intensity <= 0.0) return {0, 0, 0};
    return scale(f.lightColour, intensity);
}

V3 getColourForRay(fto_context* ctx, int recursionLimit, const Ray& ray, const Stream& st);

V3 shadeFragment(fto_context* ctx, int recursionLimit, const Fragment& f, const Stream& st) {      // shadeIfRequired(multiPartShader [specular; reflection; diffuse]); Shading.fs:100-107, Program.fs:59
    const Hit& ix = *f.intersection;
    if (!ix.material.applyLighting) return ix.material.colour;
    V3 sum = {0, 0, 0};                                                          // Seq.sumBy starts from Zero
    sum = add(sum, specularShader(f));
    V3 refl = {0, 0, 0};                                                         // reflectionShader, Shading.fs:89-98
    if (ix.material.reflectance > 0.0) {
        V3 reflectedDirection = reflect(ix.n, f.viewRay->d);
        V3 c = {0, 0, 0};                                                        // getColourForDirection, Shading.fs:132-134
        if (!(recursionLimit <= 0)) {
            ctx->counters.reflect.fetch_add(1, std::memory_order_relaxed);
            c = getColourForRay(ctx, recursionLimit - 1, Ray{ix.p, reflectedDirection}, st);
        }
        refl = scale(c, ix.material.reflectance);
    }
    sum = add(sum, refl);
    sum = add(sum, diffuseShader(f));
    return sum;
}

V3 getColourForRay(fto_context* ctx, int recursionLimit, const Ray& ray, const Stream& st) {   // Shading.fs:131-139
    Ray offset{add(ray.o, scale(ray.d, 0.0001)), ray.d};                         // slightOffset, Shading.fs:129
    HitsLease lease; Hits& hits = lease.hits;
    ctx->geometry(offset, hits);
    Hit ix;
    if (!closest(hits, ix)) return {0, 0, 0};
    V3 shadowRayOrigin = add(ix.p, scale(ix.n, 0.0001));                         // getLightsOnPoint, Shading.fs:109-117
    V3 total = {0, 0, 0};
    int lightIndex = 0;
    for (const Light& light : ctx->lights) {                                     // createFragments + Seq.sumBy shader, Shading.fs:119-127,139
        double intensity = shadowLightIntensity(ctx, light, shadowRayOrigin, st, st.maxDepth - recursionLimit, lightIndex++);
        Fragment f{&ix, scale(light.colour, intensity), lightDirection(light, ix.p), &ray};
        total = add(total, shadeFragment(ctx, recursionLimit, f, st));
    }
    return total;
}

void setErr(fto_context* c, const std::string& m) { if (c) c->err = m; }
int newNode(fto_context* c, Node&& n) { c->nodes.push_back(std::move(n)); c->committed = false; return (int)c->nodes.size() - 1; }
bool validNode(fto_context* c, ft_node id) { return c && id >= 0 && id < (ft_node)c->nodes.size(); }

} // namespace

// ================================================================= C ABI
extern "C" {

int32_t fto_create(fto_context** out) { if (!out) return FT_ERR_INVALID; *out = new fto_context(); return FT_OK; }
void fto_destroy(fto_context* ctx) { delete ctx; }
const char* fto_last_error(const fto_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

ft_node fto_sg_primitive(fto_context* c, int32_t kind) {
    if (!c || kind < 0 || kind > FT_PRIM_CYLINDER) return FT_ERR_INVALID;
    Node n; n.kind = Node::Prim; n.prim = kind; return newNode(c, std::move(n));
}
ft_node fto_sg_triangle(fto_context* c, const double v[9]) {
    if (!c || !v) return FT_ERR_INVALID;
    Node n; n.kind = Node::TriangleP; n.tri = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}}; return newNode(c, std::move(n));
}
ft_node fto_sg_bsp_mesh(fto_context* c, int32_t depth, const double* tris, int64_t n_tris) {
    if (!c || (!tris && n_tris > 0) || n_tris < 0) return FT_ERR_INVALID;
    Node n; n.kind = Node::Mesh; n.depth = depth; n.tris.resize((size_t)n_tris);
    for (int64_t i = 0; i < n_tris; ++i) { const double* v = tris + 9 * i; n.tris[(size_t)i] = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}}; }
    return newNode(c, std::move(n));
}
ft_node fto_sg_transform(fto_context* c, const ft_transform* ts, int32_t n, ft_node child) {
    if (!validNode(c, child) || !ts || n < 1) return FT_ERR_INVALID;
    std::vector<Xf> xs;
    for (int i = 0; i < n; ++i) {
        V3 v{ts[i].v[0], ts[i].v[1], ts[i].v[2]};
        if (ts[i].kind == FT_TRANSLATE) xs.push_back(xfTranslate(v));
        else if (ts[i].kind == FT_SCALE) xs.push_back(xfScale(v));
        else if (ts[i].kind == FT_ROTATE) xs.push_back(xfRotate(v, ts[i].angle));
        else return FT_ERR_INVALID;
    }
    Node nd; nd.kind = Node::Transform; nd.xf = (n == 1) ? xs[0] : xfCompose(xs); nd.children = {child};
    return newNode(c, std::move(nd));
}
ft_node fto_sg_material(fto_context* c, const ft_material* m, ft_node child) {
    if (!validNode(c, child) || !m) return FT_ERR_INVALID;
    Node n; n.kind = Node::MaterialF; n.mat = {{m->colour[0], m->colour[1], m->colour[2]}, m->roughness, m->reflectance, m->shineyness, m->apply_lighting != 0}; n.children = {child};
    return newNode(c, std::move(n));
}
ft_node fto_sg_hue_shift(fto_context* c, double, ft_node child) {
    if (!validNode(c, child)) return FT_ERR_INVALID;
    Node n; n.kind = Node::HueShift; n.children = {child}; return newNode(c, std::move(n));
}
ft_node fto_sg_ignore_light(fto_context* c, ft_node child) {
    if (!validNode(c, child)) return FT_ERR_INVALID;
    Node n; n.kind = Node::IgnoreLight; n.children = {child}; return newNode(c, std::move(n));
}
ft_node fto_sg_group(fto_context* c, const ft_node* children, int32_t n) {
    if (!c || n < 0 || (n > 0 && !children)) return FT_ERR_INVALID;
    Node nd; nd.kind = Node::Group;
    for (int i = 0; i < n; ++i) { if (!validNode(c, children[i])) return FT_ERR_INVALID; nd.children.push_back(children[i]); }
    return newNode(c, std::move(nd));
}
ft_node fto_sg_csg(fto_context* c, int32_t op, ft_node a, ft_node b) {
    if (!validNode(c, a) || !validNode(c, b) || op < 0 || op > FT_CSG_EXCLUDE) return FT_ERR_INVALID;
    Node n; n.kind = Node::Csg; n.op = op; n.children = {a, b}; return newNode(c, std::move(n));
}
ft_node fto_sg_texture_image(fto_context* c, const uint8_t* rgb24, int32_t width, int32_t height, const double* uv_ops, int32_t n_uv_ops, ft_node child) {
    if (!validNode(c, child) || !rgb24 || width <= 0 || height <= 0 || n_uv_ops < 0 || (n_uv_ops > 0 && !uv_ops)) return FT_ERR_INVALID;
    Node n; n.kind = Node::Texture; n.ca = {0, 0, 0}; n.cb = {0, 0, 0};
    n.pixels = std::make_shared<std::vector<uint8_t>>(rgb24, rgb24 + (size_t)width * height * 3); n.imgW = width; n.imgH = height;
    n.uvOps.assign(uv_ops, uv_ops + 3 * n_uv_ops); n.children = {child};
    return newNode(c, std::move(n));
}
ft_node fto_sg_texture_grid(fto_context* c, const double ca[3], const double cb[3], const double* uv_ops, int32_t n_uv_ops, ft_node child) {
    if (!validNode(c, child) || !ca || !cb || n_uv_ops < 0 || (n_uv_ops > 0 && !uv_ops)) return FT_ERR_INVALID;
    Node n; n.kind = Node::Texture; n.ca = {ca[0], ca[1], ca[2]}; n.cb = {cb[0], cb[1], cb[2]};
    n.uvOps.assign(uv_ops, uv_ops + 3 * n_uv_ops); n.children = {child};
    return newNode(c, std::move(n));
}

int32_t fto_scene_clear(fto_context* c) { if (!c) return FT_ERR_INVALID; c->nodes.clear(); c->lights.clear(); c->root = -1; c->committed = false; c->geometry = nullptr; return FT_OK; }
int32_t fto_scene_set_objects(fto_context* c, ft_node root) { if (!validNode(c, root)) return FT_ERR_INVALID; c->root = root; c->committed = false; return FT_OK; }
int32_t fto_scene_add_directional(fto_context* c, const double dir[3], const double colour[3]) {   // Light.directional, Light.fs:19-20
    if (!c || !dir || !colour) return FT_ERR_INVALID;
    Light l{}; l.kind = 0; l.v = normalise(V3{dir[0], dir[1], dir[2]}); l.colour = {colour[0], colour[1], colour[2]};
    c->lights.push_back(l); return FT_OK;
}
int32_t fto_scene_add_soft_directional(fto_context* c, const double dir[3], int32_t samples, double scatter_rad, const double colour[3]) {  // Light.fs:22-23
    if (!c || !dir || !colour || samples < 1) return FT_ERR_INVALID;
    Light l{}; l.kind = 1; l.v = normalise(V3{dir[0], dir[1], dir[2]}); l.colour = {colour[0], colour[1], colour[2]}; l.samples = samples; l.scatter = scatter_rad;
    c->lights.push_back(l); return FT_OK;
}
int32_t fto_scene_add_positional(fto_context* c, const double pos[3], const double falloff[3], const double colour[3]) { // Light.fs:25-26
    if (!c || !pos || !falloff || !colour) return FT_ERR_INVALID;
    Light l{}; l.kind = 2; l.v = {pos[0], pos[1], pos[2]}; l.falloff[0] = falloff[0]; l.falloff[1] = falloff[1]; l.falloff[2] = falloff[2];
    l.colour = {colour[0], colour[1], colour[2]};
    c->lights.push_back(l); return FT_OK;
}
int32_t fto_scene_commit(fto_context* c) {
    if (!c || c->root < 0) { setErr(c, "no objects set"); return FT_ERR_STATE; }
    try { c->geometry = build(c, c->root); }
    catch (const BuildError& e) { setErr(c, e.msg); return FT_ERR_BUILD; }
    c->committed = true; return FT_OK;
}

int32_t fto_render(fto_context* c, const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t spp, const double* jitter_xy,
                   int32_t max_depth, uint64_t seed, const ft_rect* tiles, int32_t n_tiles, double* out_rgb, int32_t threads, fto_stats* stats) {
    if (!c || !cam || !out_rgb || res_h < 2 || res_v < 2 || spp < 0 || (spp > 0 && !jitter_xy)) return FT_ERR_INVALID;
    if (!c->committed) { setErr(c, "scene not committed"); return FT_ERR_STATE; }
    auto t0 = std::chrono::steady_clock::now();
    const bool corner = spp == 0;                                                // CornerSampling.strategy, Image.fs:125-150
    ImagePlane ip = imagePlaneCreate(*cam, res_h, res_v);
    // pixel list in y-major, x order (Image.fs:104), restricted to the tiles
    std::vector<int32_t> pixels;
    if (!tiles || n_tiles <= 0) { pixels.resize((size_t)res_h * res_v); for (size_t i = 0; i < pixels.size(); ++i) pixels[i] = (int32_t)i; }
    else for (int k = 0; k < n_tiles; ++k) for (int y = tiles[k].y0; y < tiles[k].y0 + tiles[k].h; ++y) for (int x = tiles[k].x0; x < tiles[k].x0 + tiles[k].w; ++x)
        if (x >= 0 && x < res_h && y >= 0 && y < res_v) pixels.push_back(y * res_h + x);
    // The rays to shade: spp jittered rays per pixel (Image.fs:100-110), or one ray per pixel CORNER (Image.fs:128-132).
    const int stride = res_h + 1;
    std::vector<int32_t> corners;                                                // corner ids y*stride + x needed by the tile pixels
    std::vector<int32_t> cornerSlot;
    if (corner) {
        cornerSlot.assign((size_t)stride * (res_v + 1), -1);
        for (int32_t pix : pixels) {
            int x = pix % res_h, y = pix / res_h;
            for (int id : {y * stride + x, y * stride + x + 1, (y + 1) * stride + x, (y + 1) * stride + x + 1})
                if (cornerSlot[(size_t)id] < 0) { cornerSlot[(size_t)id] = (int32_t)corners.size(); corners.push_back(id); }
        }
    }
    const int64_t nRays = corner ? (int64_t)corners.size() : (int64_t)pixels.size() * spp;
    std::vector<V3> colours((size_t)nRays);
    c->counters.shadow = 0; c->counters.reflect = 0;
    int nthreads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    const int64_t chunk = 1000;                                                  // Shading.fs:143
    std::atomic<int64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            int64_t begin = next.fetch_add(chunk);
            if (begin >= nRays) break;
            int64_t end = std::min(begin + chunk, nRays);
            for (int64_t r = begin; r < end; ++r) {
                Ray ray; uint64_t sample;
                if (corner) {
                    int32_t id = corners[(size_t)r];
                    ray = rayThroughPixel(ip, id % stride, id / stride, -0.5, 0.5);             // Image.fs:131
                    sample = (uint64_t)id;
                } else {
                    int32_t pix = pixels[(size_t)(r / spp)]; int s = (int)(r % spp);
                    ray = rayThroughPixel(ip, pix % res_h, pix / res_h, jitter_xy[2 * s], jitter_xy[2 * s + 1]);
                    sample = (uint64_t)pix * (uint64_t)spp + (uint64_t)s;
                }
                if (cam->has_focus) {                                            // ImagePlane.depthOfFieldJitter, Image.fs:91-94; Ray.fs:15-18
                    Rng rng(seed, sample, 0, 0, 2);
                    Ray shifted{add(ray.o, scale(ray.d, cam->focal_length)), ray.d};             // shiftOrigin focalLength
                    shifted.d = jitterVector(rng, 1, cam->aperture_angular_size, shifted.d)[0];  // jitterDirection
                    ray = Ray{add(shifted.o, scale(shifted.d, -cam->focal_length)), shifted.d};  // shiftOrigin -focalLength
                }
                colours[(size_t)r] = getColourForRay(c, max_depth, ray, Stream{seed, sample, max_depth});   // shadeRay, Shading.fs:142
            }
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
    for (size_t i = 0; i < pixels.size(); ++i) {
        double* o = out_rgb + 3 * (size_t)pixels[i];
        V3 acc = {0, 0, 0};
        if (corner) {                                                            // Seq.average over the four corners, Image.fs:138-141
            int x = pixels[i] % res_h, y = pixels[i] / res_h;
            for (int id : {y * stride + x, y * stride + x + 1, (y + 1) * stride + x, (y + 1) * stride + x + 1}) acc = add(acc, colours[(size_t)cornerSlot[(size_t)id]]);
            o[0] = acc.x / 4.0; o[1] = acc.y / 4.0; o[2] = acc.z / 4.0;
        } else {                                                                 // blendPixels: Array.average, Image.fs:112-116
            for (int s = 0; s < spp; ++s) acc = add(acc, colours[i * spp + s]);
            o[0] = acc.x / (double)spp; o[1] = acc.y / (double)spp; o[2] = acc.z / (double)spp;  // DivideByInt, CommonTypes.fs:43
        }
    }
    if (stats) {
        stats->rays_primary = (uint64_t)nRays; stats->rays_shadow = c->counters.shadow; stats->rays_reflect = c->counters.reflect;
        stats->rays_traced = stats->rays_primary + stats->rays_shadow + stats->rays_reflect;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->threads = nthreads;
    }
    return FT_OK;
}

int32_t fto_closest(fto_context* c, const double* o, const double* d, int64_t n, int32_t* hit, double* t, double* p, double* nrm, double* colour) {
    if (!c || !c->committed) return FT_ERR_STATE;
    for (int64_t i = 0; i < n; ++i) {
        Hits hits; Hit h;
        c->geometry(Ray{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}}, hits);
        bool ok = closest(hits, h);
        if (hit) hit[i] = ok ? 1 : 0;
        if (!ok) h = newIntersection();
        if (t) t[i] = ok ? h.t : 0.0;
        if (p) { p[3 * i] = h.p.x; p[3 * i + 1] = h.p.y; p[3 * i + 2] = h.p.z; }
        if (nrm) { nrm[3 * i] = h.n.x; nrm[3 * i + 1] = h.n.y; nrm[3 * i + 2] = h.n.z; }
        if (colour) { colour[3 * i] = h.material.colour.x; colour[3 * i + 1] = h.material.colour.y; colour[3 * i + 2] = h.material.colour.z; }
    }
    return FT_OK;
}
int32_t fto_all_hits(fto_context* c, const double* o, const double* d, int64_t n, int32_t cap, int32_t* counts, double* t, double* p, double* nrm) {
    if (!c || !c->committed) return FT_ERR_STATE;
    for (int64_t i = 0; i < n; ++i) {
        Hits hits;
        c->geometry(Ray{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}}, hits);
        counts[i] = (int32_t)hits.size();
        for (int k = 0; k < cap && k < (int)hits.size(); ++k) {
            size_t j = (size_t)i * cap + k;
            if (t) t[j] = hits[k].t;
            if (p) { p[3 * j] = hits[k].p.x; p[3 * j + 1] = hits[k].p.y; p[3 * j + 2] = hits[k].p.z; }
            if (nrm) { nrm[3 * j] = hits[k].n.x; nrm[3 * j + 1] = hits[k].n.y; nrm[3 * j + 2] = hits[k].n.z; }
        }
    }
    return FT_OK;
}
int32_t fto_blocked(fto_context* c, const double* o, const double* d, const double* max_dist, int64_t n, int32_t* blocked) {
    if (!c || !c->committed) return FT_ERR_STATE;
    for (int64_t i = 0; i < n; ++i)
        blocked[i] = lightIsBlocked(c, max_dist[i], Ray{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}}) ? 1 : 0;
    return FT_OK;
}
int32_t fto_colour_for_ray(fto_context* c, const double* o, const double* d, int64_t n, int32_t max_depth, double* rgb) {
    if (!c || !c->committed) return FT_ERR_STATE;
    for (int64_t i = 0; i < n; ++i) {
        V3 col = getColourForRay(c, max_depth, Ray{{o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}}, Stream{0, (uint64_t)i, max_depth});
        rgb[3 * i] = col.x; rgb[3 * i + 1] = col.y; rgb[3 * i + 2] = col.z;
    }
    return FT_OK;
}
int32_t fto_ray_through_pixel(const ft_camera* cam, int32_t res_h, int32_t res_v, int32_t px, int32_t py, double jx, double jy, double o[3], double d[3]) {
    ImagePlane ip = imagePlaneCreate(*cam, res_h, res_v);
    Ray r = rayThroughPixel(ip, px, py, jx, jy);
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
    return FT_OK;
}
int32_t fto_image_plane(const ft_camera* cam, int32_t res_h, int32_t res_v, double out[13]) {
    ImagePlane ip = imagePlaneCreate(*cam, res_h, res_v);
    out[0] = ip.pw; out[1] = ip.ph; out[2] = ip.tlx; out[3] = ip.tly;
    out[4] = ip.i.x; out[5] = ip.i.y; out[6] = ip.i.z; out[7] = ip.j.x; out[8] = ip.j.y; out[9] = ip.j.z;
    out[10] = ip.originToCentre.x; out[11] = ip.originToCentre.y; out[12] = ip.originToCentre.z;
    return FT_OK;
}
int32_t fto_aabb_intersects(const double bmin[3], const double bmax[3], const double o[3], const double d[3]) {
    return aabbIntersects(Aabb{{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmax[2]}}, Ray{{o[0], o[1], o[2]}, {d[0], d[1], d[2]}}) ? 1 : 0;
}
int32_t fto_slice(const double p0[3], const double n[3], const double tri[9], double* above, int32_t* n_above, double* below, int32_t* n_below) {
    std::vector<Tri> a, b;
    try { slice(Plane{{p0[0], p0[1], p0[2]}, {n[0], n[1], n[2]}}, Tri{{tri[0], tri[1], tri[2]}, {tri[3], tri[4], tri[5]}, {tri[6], tri[7], tri[8]}}, a, b); }
    catch (const BuildError&) { return FT_ERR_BUILD; }
    auto dump = [](const std::vector<Tri>& v, double* out) { for (size_t i = 0; i < v.size(); ++i) { const Tri& t = v[i]; double w[9] = {t.a.x, t.a.y, t.a.z, t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z}; std::memcpy(out + 9 * i, w, sizeof w); } };
    *n_above = (int32_t)a.size(); *n_below = (int32_t)b.size();
    dump(a, above); dump(b, below);
    return FT_OK;
}
int32_t fto_bsp_stats(const double* tris, int64_t n_tris, int32_t depth, int64_t out[3]) {
    std::vector<Tri> ts((size_t)n_tris);
    for (int64_t i = 0; i < n_tris; ++i) { const double* v = tris + 9 * i; ts[(size_t)i] = {{v[0], v[1], v[2]}, {v[3], v[4], v[5]}, {v[6], v[7], v[8]}}; }
    try {
        auto tree = compile(depth, ts);
        out[0] = bspMaxDepth(*tree); out[1] = 0; out[2] = 0; bspLeafStats(*tree, out[1], out[2]);
    } catch (const BuildError&) { return FT_ERR_BUILD; }
    return FT_OK;
}
int32_t fto_quadratic(double a, double b, double c, double roots[2]) { return quadratic(a, b, c, roots); }
int32_t fto_quantise_rgba8(const double* rgb, int64_t n, uint8_t* out) {         // Image.fs:36: Math.clamp c * 255.0 |> byte
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { double v = clamp01(rgb[3 * i + k]) * 255.0; out[4 * i + k] = (v != v) ? 0 : (uint8_t)v; }
        out[4 * i + 3] = 255;
    }
    return FT_OK;
}

} // extern "C"
