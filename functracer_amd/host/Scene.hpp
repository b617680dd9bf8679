// Scene.hpp — C++ mirror of the F# data model the reference's callers build and consume:
// Scene.fs:8-65, 107-110 (Primitive, SceneGraph, SceneFunction, Texture, SceneOptions, Scene),
// Image.fs:9-33 (Camera, Focus, Resolution), Light.fs:5-14, Ray.fs:4-11 (Material),
// Transform.fs:25-45.  Same names and meaning as the F# types; this is the host side that sits
// above the C ABI because no F# toolchain exists in the build image (INTEGRATION.md shows the
// equivalent DllImport shim).
#ifndef FUNCTRACER_HOST_SCENE_HPP
#define FUNCTRACER_HOST_SCENE_HPP
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace FuncTracer {

using Vec3 = std::array<double, 3>;

struct Colour { double r = 0, g = 0, b = 0; };                      // CommonTypes.fs:42

struct Material {                                                  // Ray.fs:4-10
    Colour colour{1, 1, 1};
    double roughness = 0, reflectance = 0, shineyness = 0;
    bool applyLighting = true;
};
inline Material mattWhite() { return Material{}; }                 // Ray.fs:11

struct Transform {                                                 // Transform.fs:25-30
    enum Kind { Translate, Scale, Rotate, Composed } kind = Translate;
    Vec3 v{0, 0, 0};
    double angle = 0;                                              // radians
    std::vector<Transform> parts;                                  // Composed
};
Transform translate(Vec3 v);                                       // Transform.fs:32
Transform scale(Vec3 v);                                           // Transform.fs:35
Transform rotate(Vec3 axis, double angleRad);                      // Transform.fs:37-38
Transform compose(const std::vector<Transform>& ts);               // Transform.fs:41-45

struct Triangle { Vec3 a, b, c; };                                 // Triangle.fs:5

enum class Primitive { Circle, Square, Cube, Sphere, Plane, Cone, SolidCylinder, Cylinder, Triangle, BspMesh };  // Scene.fs:8-18

struct TextureFunction { enum Kind { Scale, Rotate } kind; double a = 1, b = 1; };   // Scene.fs:51-53 (Rotate: a = radians)
struct Texture {                                                   // Scene.fs:47-50
    enum Kind { Grid, Image } kind = Grid;
    Colour c1, c2;
    std::string source;                                            // Image: path as written in the scene
    std::shared_ptr<const std::vector<uint8_t>> pixels;            // Image: Rgb24 rows, top first (image.SavePixelData(), Textures/Image.fs:23)
    int width = 0, height = 0;
    std::vector<TextureFunction> functions;                        // outermost first
};

struct SceneGraph;
using SceneGraphPtr = std::shared_ptr<const SceneGraph>;

struct SceneFunction {                                             // Scene.fs:41-46
    enum Kind { TransformF, MaterialF, TextureF, HueShift, IgnoreLight } kind = TransformF;
    Transform transform;
    Material material;
    Texture texture;
    double angle = 0;
};

struct SceneGraph {                                                // Scene.fs:33-40
    enum Kind { PrimitiveN, SceneFunctionN, Group, Union, Intersect, Subtract, Exclude } kind = Group;
    Primitive primitive = Primitive::Sphere;
    Triangle triangle{};                                           // Primitive.Triangle
    int bspDepth = 0;                                              // Primitive.BspMesh: depth + triangles are kept (the F# closure hides them)
    std::shared_ptr<const std::vector<Triangle>> meshTriangles;
    SceneFunction function;
    std::vector<SceneGraphPtr> nodes;                              // Group children | [a; b] | [child]
};

struct Falloff { double constant = 1, linear = 0, quadratic = 0; }; // Light.fs:5
struct Light {                                                     // Light.fs:7-14
    enum Kind { Directional, SoftDirectional, Point } kind = Directional;
    Vec3 v{0, 0, 0};                                               // direction (as given; normalised by the library like Light.fs:19-23) | position
    int samples = 1;
    double scattering = 0;                                         // radians
    Falloff falloff;
    Colour colour;
};

struct Focus { double focalLength = 0, apetureAngularSize = 0; };  // Image.fs:9
struct Camera {                                                    // Image.fs:10-17
    Vec3 o{0, 0, 0}, lookAt{0, 0, 1}, up{0, 1, 0};
    double fovY = 50.0 * (3.14159265358979323846 / 180.0);
    double aspectRatio = 1.0;
    bool hasFocus = false;
    Focus focus;
};
struct Resolution { int h = 400, v = 400; };                       // Image.fs:28 (resH, resV)
struct SamplingStrategy { bool corner = false; int samplesPerPixel = 8; };  // Image.fs:20-23, 118-122, 146-150

struct SceneOptions {                                              // Scene.fs:56-65 (Default)
    Camera camera;
    SamplingStrategy samplingStrategy;
    Resolution resolution;
};
struct Scene {                                                     // Scene.fs:107-110
    SceneGraphPtr objects;
    std::vector<Light> lights;
};

// SceneParser.parse (SceneParser.fs:360-366): returns true and fills options/scene, or false with `error`.
bool parseScene(const std::string& text, const std::string& baseDir, SceneOptions& options, Scene& scene, std::string& error);
// PlyParser.parse (PlyParser.fs:65-69).
bool parsePly(const std::string& text, std::vector<Triangle>& triangles, std::string& error);
// Image.Load<Rgb24> (Textures/Image.fs:21-26) for local PNG / PPM files; ImageLoader.cpp.
// Whole file into `out`; false for anything that cannot be read as a regular file (missing, a directory, an I/O error).
bool readWholeFile(const std::string& path, std::string& out);
bool loadImageRgb24(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgb, std::string& error);
// Parsers.pcolour (SceneParser.fs:85-87), exposed for the reference's own colour tests.
bool parseColour(const std::string& text, Colour& out);

} // namespace FuncTracer
#endif
