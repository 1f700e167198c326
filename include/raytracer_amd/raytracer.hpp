// raytracer.hpp — C++ mirror of the reference's public (Rust) scene API, on top of the C ABI in
// include/rt_amd.h.  The reference is Rust and this image has no Rust toolchain, so the host side
// above the C ABI is written in C++ with the same names, argument meaning and error behaviour:
//
//   World{objects, lights}            world.rs:15-21        Material{...}         material.rs:37-102
//   Camera::try_from(CameraBuilder)   camera.rs:77-136      Pattern3D::{Solid,..} pattern.rs:16-56
//   Camera::render(&World) -> Canvas  camera.rs:163-208     Light::{Point,Area}   light.rs:14-144
//   Canvas::to_image()                canvas.rs:31-45       Transform::*          transform.rs:141-287
//   Shape::{Sphere,Plane,Cube,Cylinder,Triangle,Group}, ShapeBuilder, CylinderBuilder,
//   TriangleBuilder, Group{push,extend,divide}, GroupBuilder       shape.rs:33-77, shape/*.rs
//
// Scenes are built exactly as in examples/*.rs (see scenes/examples.inc).  `flatten()` compiles a
// World into the flat structure-of-arrays buffers of rt_scene_desc (f64 -> f32, rounded once) and
// Camera::render() hands them to the HIP kernels through rt_scene_create / rt_render.  Nothing here
// intersects or shades anything on the CPU: without librt_amd.so and a GPU, render() fails loudly.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../rt_amd.h"

namespace rt {

constexpr double EPSILON = 1e-5;  // float.rs:3
constexpr double INF = std::numeric_limits<double>::infinity();

inline bool approx(double a, double b) {  // float.rs:5-11
    if (std::isinf(a) && std::isinf(b)) return a == b;
    return std::fabs(a - b) <= EPSILON;
}

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------------------------ tuples
struct Vector {
    double x = 0, y = 0, z = 0;
    double magnitude() const { return std::sqrt(x * x + y * y + z * z); }
    double dot(const Vector& r) const { return x * r.x + y * r.y + z * r.z; }
    Vector cross(const Vector& r) const { return {y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x}; }
    Vector operator+(const Vector& r) const { return {x + r.x, y + r.y, z + r.z}; }
    Vector operator-(const Vector& r) const { return {x - r.x, y - r.y, z - r.z}; }
    Vector operator*(double s) const { return {x * s, y * s, z * s}; }
    Vector operator-() const { return {0.0 - x, 0.0 - y, 0.0 - z}; }
    // tuple.rs:102-104,228-236: fails (here: throws) for a vector of magnitude <= 1e-5
    Vector normalize() const {
        double m = magnitude();
        if (approx(m, 0.0)) throw Error("tried to normalize a null vector");
        return (*this) * (1.0 / m);
    }
    bool is_null() const { return approx(x, 0.0) && approx(y, 0.0) && approx(z, 0.0); }
};
struct Point {
    double x = 0, y = 0, z = 0;
    Vector operator-(const Point& r) const { return {x - r.x, y - r.y, z - r.z}; }
    Point operator+(const Vector& r) const { return {x + r.x, y + r.y, z + r.z}; }
};

struct Color {
    double red = 0, green = 0, blue = 0;
};
namespace color {  // color/consts.rs:8-54
constexpr Color WHITE{1.0, 1.0, 1.0};
constexpr Color BLACK{0.0, 0.0, 0.0};
constexpr Color RED{1.0, 0.0, 0.0};
constexpr Color GREEN{0.0, 1.0, 0.0};
constexpr Color BLUE{0.0, 0.0, 1.0};
constexpr Color LIGHT_SKY_BLUE{0.5294, 0.8078, 0.9804};
constexpr Color DIRT{0.6078, 0.4627, 0.3255};
}  // namespace color

// ------------------------------------------------------------------------------------ transforms
// transform.rs:133-312.  4x4 row-major f64; inverse by cofactors (matrix.rs:136-151).
struct Transform {
    std::array<std::array<double, 4>, 4> m{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};

    static Transform translation(double x, double y, double z) {
        Transform t;
        t.m[0][3] = x, t.m[1][3] = y, t.m[2][3] = z;
        return t;
    }
    static Transform scaling(double x, double y, double z) {
        if (approx(x * y * z, 0.0)) throw Error("components cannot be scaled to zero");
        Transform t;
        t.m[0][0] = x, t.m[1][1] = y, t.m[2][2] = z;
        return t;
    }
    static Transform rotation_x(double r) {
        Transform t;
        t.m[1][1] = std::cos(r), t.m[1][2] = -std::sin(r), t.m[2][1] = std::sin(r), t.m[2][2] = std::cos(r);
        return t;
    }
    static Transform rotation_y(double r) {
        Transform t;
        t.m[0][0] = std::cos(r), t.m[0][2] = std::sin(r), t.m[2][0] = -std::sin(r), t.m[2][2] = std::cos(r);
        return t;
    }
    static Transform rotation_z(double r) {
        Transform t;
        t.m[0][0] = std::cos(r), t.m[0][1] = -std::sin(r), t.m[1][0] = std::sin(r), t.m[1][1] = std::cos(r);
        return t;
    }
    static Transform shearing(double xy, double xz, double yx, double yz, double zx, double zy) {
        if (approx(xz * yx * zy + xy * yz * zx - xy * yx - xz * zx - yz * zy + 1.0, 0.0))
            throw Error("result of `xz * yx * zy + xy * yz * zx - xy * yx - xz * zx - yz * zy` cannot equal `-1`");
        Transform t;
        t.m[0][1] = xy, t.m[0][2] = xz, t.m[1][0] = yx, t.m[1][2] = yz, t.m[2][0] = zx, t.m[2][1] = zy;
        return t;
    }
    static Transform view(const Point& from, const Point& to, const Vector& up) {  // transform.rs:251-275
        Vector to_from = to - from;
        if (approx(to_from.magnitude(), 0.0)) throw Error("`from` and `to` points cannot be equal");
        if (approx(up.magnitude(), 0.0)) throw Error("up direction cannot be null");
        Vector forward = to_from.normalize();
        Vector left = forward.cross(up.normalize());
        if (left.is_null()) throw Error("`from` and `up` vectors cannot be collinear");
        Vector true_up = left.cross(forward);
        Transform o;
        o.m[0] = {left.x, left.y, left.z, 0.0};
        o.m[1] = {true_up.x, true_up.y, true_up.z, 0.0};
        o.m[2] = {-forward.x, -forward.y, -forward.z, 0.0};
        return o * translation(-from.x, -from.y, -from.z);
    }
    Transform operator*(const Transform& r) const {
        Transform out;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc = acc + m[i][k] * r.m[k][j];
                out.m[i][j] = acc;
            }
        return out;
    }
    Point operator*(const Point& p) const {
        return {((m[0][0] * p.x + m[0][1] * p.y) + m[0][2] * p.z) + m[0][3],
                ((m[1][0] * p.x + m[1][1] * p.y) + m[1][2] * p.z) + m[1][3],
                ((m[2][0] * p.x + m[2][1] * p.y) + m[2][2] * p.z) + m[2][3]};
    }
    double minor3(int r0, int r1, int r2, int c0, int c1, int c2) const {
        return m[r0][c0] * (m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1]) -
               m[r0][c1] * (m[r1][c0] * m[r2][c2] - m[r1][c2] * m[r2][c0]) +
               m[r0][c2] * (m[r1][c0] * m[r2][c1] - m[r1][c1] * m[r2][c0]);
    }
    double cofactor(int row, int col) const {
        int r[3], c[3], ri = 0, ci = 0;
        for (int i = 0; i < 4; ++i) {
            if (i != row) r[ri++] = i;
            if (i != col) c[ci++] = i;
        }
        double mn = minor3(r[0], r[1], r[2], c[0], c[1], c[2]);
        return ((row + col) & 1) ? -mn : mn;
    }
    Transform inverse() const {
        double det = 0.0;
        for (int c = 0; c < 4; ++c) det = det + m[0][c] * cofactor(0, c);
        if (approx(det, 0.0)) throw Error("tried to inverse a singular matrix");
        Transform inv;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) inv.m[j][i] = cofactor(i, j) / det;
        return inv;
    }
    bool is_affine() const { return m[3][0] == 0.0 && m[3][1] == 0.0 && m[3][2] == 0.0 && m[3][3] == 1.0; }
};

// ------------------------------------------------------------------------------------ patterns, materials
struct Pattern3DSpec {  // pattern.rs:39-56
    Color color_a, color_b;
    Transform transform, transform_inverse;
    static Pattern3DSpec make(Color a, Color b, const Transform& t) { return {a, b, t, t.inverse()}; }
};
struct Pattern3D {  // pattern.rs:16-31
    int kind = RT_PAT_SOLID;
    Color solid = color::WHITE;
    Pattern3DSpec spec{};
    static Pattern3D Solid(Color c) {
        Pattern3D p;
        p.solid = c;
        return p;
    }
    static Pattern3D with(int kind, const Pattern3DSpec& s) {
        Pattern3D p;
        p.kind = kind, p.spec = s;
        return p;
    }
    static Pattern3D Stripe(const Pattern3DSpec& s) { return with(RT_PAT_STRIPE, s); }
    static Pattern3D Gradient(const Pattern3DSpec& s) { return with(RT_PAT_GRADIENT, s); }
    static Pattern3D Ring(const Pattern3DSpec& s) { return with(RT_PAT_RING, s); }
    static Pattern3D Checker(const Pattern3DSpec& s) { return with(RT_PAT_CHECKER, s); }
};
namespace material {  // material.rs:11-29
constexpr double VACUUM_INDEX_OF_REFRACTION = 1.0;
constexpr double AIR_INDEX_OF_REFRACTION = 1.00029;
constexpr double WATER_INDEX_OF_REFRACTION = 1.333;
constexpr double GLASS_INDEX_OF_REFRACTION = 1.458;
constexpr double DIAMOND_INDEX_OF_REFRACTION = 2.417;
}  // namespace material
struct Material {  // material.rs:37-102
    Pattern3D pattern = Pattern3D::Solid(color::WHITE);
    double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
    double index_of_refraction = material::VACUUM_INDEX_OF_REFRACTION;
    double reflectivity = 0.0, transparency = 0.0;
};

// ------------------------------------------------------------------------------------ bounding boxes
struct BoundingBox {  // shape/bounding_box.rs:5-147
    Point min{INF, INF, INF}, max{-INF, -INF, -INF};
    void add(const Point& p) {
        min = {std::fmin(p.x, min.x), std::fmin(p.y, min.y), std::fmin(p.z, min.z)};
        max = {std::fmax(p.x, max.x), std::fmax(p.y, max.y), std::fmax(p.z, max.z)};
    }
    void merge(const BoundingBox& b) {
        add(b.min);
        add(b.max);
    }
    static bool between(double x, double lo, double hi) { return (approx(x, lo) || x > lo) && (approx(x, hi) || x < hi); }
    bool contains_point(const Point& p) const {
        return between(p.x, min.x, max.x) && between(p.y, min.y, max.y) && between(p.z, min.z, max.z);
    }
    bool contains(const BoundingBox& b) const { return contains_point(b.min) && contains_point(b.max); }
    BoundingBox transform(const Transform& t) const {
        BoundingBox out;
        for (int i = 0; i < 8; ++i)
            out.add(t * Point{(i & 4) ? max.x : min.x, (i & 2) ? max.y : min.y, (i & 1) ? max.z : min.z});
        return out;
    }
    std::pair<BoundingBox, BoundingBox> split() const {  // bounding_box.rs:86-142
        double dx = std::fabs(min.x - max.x), dy = std::fabs(min.y - max.y), dz = std::fabs(min.z - max.z);
        double largest = dx;  // Iterator::max_by keeps the last of equal maxima
        if (!(largest > dy)) largest = dy;
        if (!(largest > dz)) largest = dz;
        Point lo = min, hi = max;
        if (approx(largest, dx)) lo.x = hi.x = min.x + dx / 2.0;
        else if (approx(largest, dy)) lo.y = hi.y = min.y + dy / 2.0;
        else lo.z = hi.z = min.z + dz / 2.0;
        return {BoundingBox{min, hi}, BoundingBox{lo, max}};
    }
};

// ------------------------------------------------------------------------------------ shapes
struct ShapeBuilder {
    Material material{};
    Transform transform{};
};
struct CylinderBuilder {
    Material material{};
    Transform transform{};
    double min = -INF, max = INF;
    bool closed = false;
};
struct TriangleBuilder {
    Material material{};
    std::array<Point, 3> vertices{};
};
struct Group;

struct Shape {
    enum Kind { CUBE, CYLINDER, GROUP, PLANE, SMOOTH_TRIANGLE, SPHERE, TRIANGLE } kind = SPHERE;
    // ObjectCache, shape/object.rs:6-12
    Material material{};
    Transform transform{}, transform_inverse{};
    BoundingBox bounding_box{}, parent_space_bounding_box{};
    double cyl_min = -INF, cyl_max = INF;
    bool closed = false;
    Point v0{};
    Vector e0{}, e1{}, normal{}, n0{}, n1{}, n2{};
    std::vector<Shape> children;  // Group only, group.rs:36-39

    static Shape make(Kind k, const Material& m, const Transform& t, const BoundingBox& bb) {  // object.rs:43-51
        Shape s;
        s.kind = k, s.material = m, s.transform = t, s.transform_inverse = t.inverse();
        s.bounding_box = bb, s.parent_space_bounding_box = bb.transform(t);
        return s;
    }
    static Shape Sphere(const ShapeBuilder& b = {}) {
        return make(SPHERE, b.material, b.transform, {Point{-1, -1, -1}, Point{1, 1, 1}});
    }
    static Shape Cube(const ShapeBuilder& b = {}) {
        return make(CUBE, b.material, b.transform, {Point{-1, -1, -1}, Point{1, 1, 1}});
    }
    static Shape Plane(const ShapeBuilder& b = {}) {
        return make(PLANE, b.material, b.transform, {Point{-INF, 0, -INF}, Point{INF, 0, INF}});
    }
    static Shape Cylinder(const CylinderBuilder& b = {}) {
        Shape s = make(CYLINDER, b.material, b.transform, {Point{-1, b.min, -1}, Point{1, b.max, 1}});
        s.cyl_min = b.min, s.cyl_max = b.max, s.closed = b.closed;
        return s;
    }
    static Shape Triangle(const TriangleBuilder& b) {  // triangle.rs:68-101
        Vector e0 = b.vertices[1] - b.vertices[0], e1 = b.vertices[2] - b.vertices[0];
        Vector cr = e1.cross(e0);
        if (approx(cr.magnitude(), 0.0)) throw Error("triangle sides most not be collinear");
        BoundingBox bb;
        for (const auto& v : b.vertices) bb.add(v);
        Shape s = make(TRIANGLE, b.material, Transform{}, bb);
        s.v0 = b.vertices[0], s.e0 = e0, s.e1 = e1, s.normal = cr.normalize();
        return s;
    }
    static Shape SmoothTriangle(const TriangleBuilder& b, Vector n0, Vector n1, Vector n2) {
        Shape s = Triangle(b);
        s.kind = SMOOTH_TRIANGLE, s.n0 = n0, s.n1 = n1, s.n2 = n2;
        return s;
    }
    static Shape Group(rt::Group g);
};

struct GroupBuilder {
    std::vector<Shape> children{};
    Transform transform{};
};

struct Group {  // shape/group.rs:36-243
    std::vector<Shape> children;
    Transform transform{}, transform_inverse{};
    BoundingBox bounding_box_acc{};  // object_cache.bounding_box, merged at push time

    static Group from(const GroupBuilder& b) {
        Group g;
        g.transform = b.transform, g.transform_inverse = b.transform.inverse();
        g.extend(b.children);
        return g;
    }
    static void apply_transform_to_child(Shape& child, const Transform& t) {  // group.rs:81-94
        if (child.kind == Shape::GROUP)
            for (auto& c : child.children) apply_transform_to_child(c, t);
        child.transform = t * child.transform;
        child.transform_inverse = child.transform.inverse();
        child.parent_space_bounding_box = child.bounding_box.transform(child.transform);
    }
    void push(Shape child) {  // group.rs:72-79
        apply_transform_to_child(child, transform);
        bounding_box_acc.merge(child.parent_space_bounding_box);
        children.push_back(std::move(child));
    }
    void extend(const std::vector<Shape>& cs) {
        for (const auto& c : cs) push(c);
    }
    BoundingBox bounding_box() const {  // group.rs:233-242
        BoundingBox bb;
        for (const auto& c : children) bb.merge(c.parent_space_bounding_box);
        return bb;
    }
    void divide(size_t threshold);  // group.rs:168-231
};

inline Shape Shape::Group(rt::Group g) {
    Shape s;
    s.kind = GROUP;
    s.transform = g.transform, s.transform_inverse = g.transform_inverse;
    s.bounding_box = g.bounding_box_acc;
    s.children = std::move(g.children);
    return s;
}
inline void Group::divide(size_t threshold) {
    if (threshold <= children.size()) {
        auto halves = bounding_box().split();
        std::vector<Shape> side[2];
        for (size_t i = 0; i < children.size();) {  // group.rs:188-219, Vec::swap_remove order
            const BoundingBox& cb = children[i].parent_space_bounding_box;
            int which = halves.first.contains(cb) ? 0 : (halves.second.contains(cb) ? 1 : -1);
            if (which < 0) {
                ++i;
                continue;
            }
            children[i].transform = transform_inverse * children[i].transform;
            side[which].push_back(std::move(children[i]));
            if (i + 1 != children.size()) children[i] = std::move(children.back());
            children.pop_back();
        }
        for (auto& cs : side) {
            if (cs.empty()) continue;
            Group sub;  // make_subgroup, group.rs:221-231
            for (auto& c : cs) sub.push(std::move(c));
            push(Shape::Group(std::move(sub)));
        }
    }
    for (auto& child : children)
        if (child.kind == Shape::GROUP) {
            Group sub;
            sub.transform = child.transform, sub.transform_inverse = child.transform_inverse;
            sub.bounding_box_acc = child.bounding_box;
            sub.children = std::move(child.children);
            sub.divide(threshold);
            child.bounding_box = sub.bounding_box_acc;
            child.children = std::move(sub.children);
        }
}

// ------------------------------------------------------------------------------------ lights
struct PointLight {
    Point position{};
    Color intensity{};
};
struct AreaLightBuilder {  // light.rs:92-115
    Point corner{};
    Vector horizontal_dir{};
    size_t horizontal_cells = 1;
    Vector vertical_dir{};
    size_t vertical_cells = 1;
    Color intensity{};
};
struct Light {
    int kind = RT_LIGHT_POINT;
    rt::Point position{};  // point: position; area: corner
    Vector uvec{}, vvec{};
    size_t usteps = 1, vsteps = 1;
    Color intensity{};
    static Light Point(const PointLight& p) {
        Light l;
        l.position = p.position, l.intensity = p.intensity;
        return l;
    }
    static Light Area(const AreaLightBuilder& b) {  // light.rs:117-144
        if (approx(static_cast<double>(b.horizontal_cells), 0.0) || approx(static_cast<double>(b.vertical_cells), 0.0))
            throw Error("tried to divide a tuple by zero");
        Light l;
        l.kind = RT_LIGHT_AREA;
        l.position = b.corner;
        l.uvec = b.horizontal_dir * (1.0 / static_cast<double>(b.horizontal_cells));
        l.vvec = b.vertical_dir * (1.0 / static_cast<double>(b.vertical_cells));
        l.usteps = b.horizontal_cells, l.vsteps = b.vertical_cells;
        l.intensity = b.intensity;
        return l;
    }
};

struct World {
    std::vector<Shape> objects;
    std::vector<Light> lights;
};

// ------------------------------------------------------------------------------------ flatten
// Owns the flat SoA buffers an rt_scene_desc points into.
struct FlatScene {
    std::vector<uint8_t> prim_type, pat_type, light_type;
    std::vector<float> prim_inv, prim_param, node_min, node_max, mat_param, pat_color_a, pat_color_b, pat_inv;
    std::vector<float> light_pos, light_uvec, light_vvec, light_color;
    std::vector<uint32_t> prim_material, item_ref, item_next, mat_pattern, light_usteps, light_vsteps;
    std::vector<double> prim_inv64, prim_param64, mat_ior64, pat_inv64;  // the f64 values themselves (rt_exact.h)
    std::vector<double> light_geom64;                                    // position (area: corner), uvec, vvec: nine a light
    std::vector<double> node_box64;                                      // a Group's box, min then max: six a node

    rt_scene_desc desc() const {
        rt_scene_desc d{};
        d.abi_version = RT_ABI_VERSION;
        d.n_prims = static_cast<uint32_t>(prim_type.size());
        d.prim_type = prim_type.data(), d.prim_inv = prim_inv.data(), d.prim_material = prim_material.data();
        d.prim_param = prim_param.data();
        d.n_nodes = static_cast<uint32_t>(node_min.size() / 3);
        d.node_min = node_min.data(), d.node_max = node_max.data();
        d.n_items = static_cast<uint32_t>(item_ref.size());
        d.item_ref = item_ref.data(), d.item_next = item_next.data();
        d.n_materials = static_cast<uint32_t>(mat_pattern.size());
        d.mat_param = mat_param.data(), d.mat_pattern = mat_pattern.data();
        d.n_patterns = static_cast<uint32_t>(pat_type.size());
        d.pat_type = pat_type.data(), d.pat_color_a = pat_color_a.data(), d.pat_color_b = pat_color_b.data();
        d.pat_inv = pat_inv.data();
        d.n_lights = static_cast<uint32_t>(light_type.size());
        d.light_type = light_type.data(), d.light_pos = light_pos.data(), d.light_uvec = light_uvec.data();
        d.light_vvec = light_vvec.data(), d.light_usteps = light_usteps.data(), d.light_vsteps = light_vsteps.data();
        d.light_color = light_color.data();
        d.prim_inv64 = prim_inv64.data(), d.prim_param64 = prim_param64.data();
        d.mat_ior64 = mat_ior64.data(), d.pat_inv64 = pat_inv64.data();
        d.light_geom64 = light_geom64.data();
        d.node_box64 = node_box64.data();
        return d;
    }
};

namespace detail {
inline void push3(std::vector<float>& v, double a, double b, double c) {
    v.push_back(static_cast<float>(a)), v.push_back(static_cast<float>(b)), v.push_back(static_cast<float>(c));
}
inline void push_affine(std::vector<float>& v, const Transform& t) {
    if (!t.is_affine()) throw Error("transform is not affine");
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) v.push_back(static_cast<float>(t.m[i][j]));
}
inline void push_affine64(std::vector<double>& v, const Transform& t) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) v.push_back(t.m[i][j]);
}
inline bool same_color(const Color& a, const Color& b) { return a.red == b.red && a.green == b.green && a.blue == b.blue; }
inline bool same_transform(const Transform& a, const Transform& b) { return a.m == b.m; }

struct Flattener {
    FlatScene out;
    std::vector<Pattern3D> pats;
    std::vector<Material> mats;

    uint32_t pattern_index(const Pattern3D& p) {
        for (size_t i = 0; i < pats.size(); ++i) {  // de-duplicate bitwise-identical patterns
            const Pattern3D& q = pats[i];
            if (q.kind != p.kind) continue;
            if (p.kind == RT_PAT_SOLID ? same_color(q.solid, p.solid)
                                       : (same_color(q.spec.color_a, p.spec.color_a) && same_color(q.spec.color_b, p.spec.color_b) &&
                                          same_transform(q.spec.transform_inverse, p.spec.transform_inverse)))
                return static_cast<uint32_t>(i);
        }
        pats.push_back(p);
        out.pat_type.push_back(static_cast<uint8_t>(p.kind));
        if (p.kind == RT_PAT_SOLID) {
            push3(out.pat_color_a, p.solid.red, p.solid.green, p.solid.blue);
            push3(out.pat_color_b, p.solid.red, p.solid.green, p.solid.blue);
            push_affine(out.pat_inv, Transform{});
            push_affine64(out.pat_inv64, Transform{});
        } else {
            push3(out.pat_color_a, p.spec.color_a.red, p.spec.color_a.green, p.spec.color_a.blue);
            push3(out.pat_color_b, p.spec.color_b.red, p.spec.color_b.green, p.spec.color_b.blue);
            push_affine(out.pat_inv, p.spec.transform_inverse);
            push_affine64(out.pat_inv64, p.spec.transform_inverse);
        }
        return static_cast<uint32_t>(pats.size() - 1);
    }
    uint32_t material_index(const Material& m) {
        uint32_t pi = pattern_index(m.pattern);
        for (size_t i = 0; i < mats.size(); ++i) {
            const Material& q = mats[i];
            if (out.mat_pattern[i] == pi && q.ambient == m.ambient && q.diffuse == m.diffuse && q.specular == m.specular &&
                q.shininess == m.shininess && q.index_of_refraction == m.index_of_refraction &&
                q.reflectivity == m.reflectivity && q.transparency == m.transparency)
                return static_cast<uint32_t>(i);
        }
        mats.push_back(m);
        out.mat_pattern.push_back(pi);
        const double p[8] = {m.ambient, m.diffuse, m.specular, m.shininess, m.index_of_refraction, m.reflectivity, m.transparency, 0.0};
        for (double v : p) out.mat_param.push_back(static_cast<float>(v));
        out.mat_ior64.push_back(m.index_of_refraction);
        return static_cast<uint32_t>(mats.size() - 1);
    }
    void add_shape(const Shape& s) {
        if (s.kind == Shape::GROUP) {
            BoundingBox bb;  // exactly Group::bounding_box(), group.rs:233-242
            for (const auto& c : s.children) bb.merge(c.parent_space_bounding_box);
            uint32_t node = static_cast<uint32_t>(out.node_min.size() / 3);
            push3(out.node_min, bb.min.x, bb.min.y, bb.min.z);
            push3(out.node_max, bb.max.x, bb.max.y, bb.max.z);
            for (double v : {bb.min.x, bb.min.y, bb.min.z, bb.max.x, bb.max.y, bb.max.z}) out.node_box64.push_back(v);
            size_t item = out.item_ref.size();
            out.item_ref.push_back(node | RT_ITEM_NODE);
            out.item_next.push_back(0);
            for (const auto& c : s.children) add_shape(c);
            out.item_next[item] = static_cast<uint32_t>(out.item_ref.size());
            return;
        }
        uint32_t prim = static_cast<uint32_t>(out.prim_type.size());
        int type = RT_PRIM_SPHERE;
        switch (s.kind) {
            case Shape::SPHERE: type = RT_PRIM_SPHERE; break;
            case Shape::PLANE: type = RT_PRIM_PLANE; break;
            case Shape::CUBE: type = RT_PRIM_CUBE; break;
            case Shape::CYLINDER: type = RT_PRIM_CYLINDER; break;
            case Shape::TRIANGLE: type = RT_PRIM_TRIANGLE; break;
            case Shape::SMOOTH_TRIANGLE: type = RT_PRIM_SMOOTH_TRIANGLE; break;
            default: break;
        }
        out.prim_type.push_back(static_cast<uint8_t>(type));
        push_affine(out.prim_inv, s.transform_inverse);
        push_affine64(out.prim_inv64, s.transform_inverse);
        out.prim_material.push_back(material_index(s.material));
        size_t base = out.prim_param.size();
        out.prim_param.resize(base + 24, 0.0f);
        out.prim_param64.resize(base + 24, 0.0);
        float* p = out.prim_param.data() + base;
        double* p64 = out.prim_param64.data() + base;
        if (s.kind == Shape::CYLINDER) {
            p[0] = static_cast<float>(s.cyl_min), p[1] = static_cast<float>(s.cyl_max), p[2] = s.closed ? 1.0f : 0.0f;
            p64[0] = s.cyl_min, p64[1] = s.cyl_max, p64[2] = s.closed ? 1.0 : 0.0;
        } else if (s.kind == Shape::TRIANGLE || s.kind == Shape::SMOOTH_TRIANGLE) {
            const double v[21] = {s.v0.x, s.v0.y, s.v0.z, s.e0.x, s.e0.y, s.e0.z, s.e1.x, s.e1.y, s.e1.z,
                                  s.normal.x, s.normal.y, s.normal.z, s.n0.x, s.n0.y, s.n0.z,
                                  s.n1.x, s.n1.y, s.n1.z, s.n2.x, s.n2.y, s.n2.z};
            for (int i = 0; i < 21; ++i) p[i] = static_cast<float>(v[i]), p64[i] = v[i];
        }
        out.item_ref.push_back(prim);
        out.item_next.push_back(static_cast<uint32_t>(out.item_ref.size()));
    }
};
}  // namespace detail

inline FlatScene flatten(const World& world) {
    detail::Flattener f;
    for (const auto& o : world.objects) f.add_shape(o);
    for (const auto& l : world.lights) {
        f.out.light_type.push_back(static_cast<uint8_t>(l.kind));
        detail::push3(f.out.light_pos, l.position.x, l.position.y, l.position.z);
        detail::push3(f.out.light_uvec, l.uvec.x, l.uvec.y, l.uvec.z);
        detail::push3(f.out.light_vvec, l.vvec.x, l.vvec.y, l.vvec.z);
        for (double v : {l.position.x, l.position.y, l.position.z, l.uvec.x, l.uvec.y, l.uvec.z, l.vvec.x, l.vvec.y, l.vvec.z})
            f.out.light_geom64.push_back(v);
        f.out.light_usteps.push_back(static_cast<uint32_t>(l.usteps));
        f.out.light_vsteps.push_back(static_cast<uint32_t>(l.vsteps));
        detail::push3(f.out.light_color, l.intensity.red, l.intensity.green, l.intensity.blue);
    }
    return std::move(f.out);
}

// ------------------------------------------------------------------------------------ canvas, camera
struct Canvas {  // canvas.rs:8-45 — dense, row-major, row 0 = top
    size_t width = 0, height = 0;
    std::vector<float> rgb;
    Color pixel_at(size_t x, size_t y) const {
        const float* p = &rgb[(y * width + x) * 3];
        return {p[0], p[1], p[2]};
    }
    std::vector<uint8_t> to_image() const {  // via rt_quantize_u8 (device kernel)
        std::vector<uint8_t> img(rgb.size());
        if (rt_quantize_u8(rgb.data(), rgb.size(), img.data()) != RT_OK) throw Error(rt_last_error());
        return img;
    }
};

struct CameraBuilder {
    size_t width = 0, height = 0;
    double field_of_view = 0;
    Transform transform{};
};

struct Camera {
    size_t hsize = 0, vsize = 0;
    double field_of_view = 0, pixel_size = 0, half_height = 0, half_width = 0;
    Transform transform{}, transform_inverse{};

    static Camera try_from(const CameraBuilder& b) {  // camera.rs:95-136
        if (approx(std::fmod(b.field_of_view, M_PI), 0.0)) throw Error("field of view angle cannot be straight");
        if (b.width == 0 || b.height == 0) throw Error("camera cannot have null dimensions");
        Camera c;
        c.hsize = b.width, c.vsize = b.height, c.field_of_view = b.field_of_view;
        double half_view = std::tan(b.field_of_view / 2.0);
        double aspect = static_cast<double>(b.width) / static_cast<double>(b.height);
        if (aspect < 1.0) c.half_width = half_view * aspect, c.half_height = half_view;
        else c.half_width = half_view, c.half_height = half_view / aspect;
        c.pixel_size = (c.half_width * 2.0) / static_cast<double>(b.width);
        c.transform = b.transform, c.transform_inverse = b.transform.inverse();
        return c;
    }
    rt_camera_desc desc() const {
        if (!transform_inverse.is_affine()) throw Error("camera transform is not affine");
        rt_camera_desc d{};
        d.hsize = static_cast<uint32_t>(hsize), d.vsize = static_cast<uint32_t>(vsize);
        d.pixel_size = static_cast<float>(pixel_size);
        d.half_width = static_cast<float>(half_width), d.half_height = static_cast<float>(half_height);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) d.inv[i * 4 + j] = static_cast<float>(transform_inverse.m[i][j]);
        d.has_f64 = 1u;
        d.pixel_size64 = pixel_size, d.half_width64 = half_width, d.half_height64 = half_height;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) d.inv64[i * 4 + j] = transform_inverse.m[i][j];
        return d;
    }
    // camera.rs:163-208.  `device` is the HIP ordinal; options default to the reference's constants.
    Canvas render(const World& world, const rt_options* options = nullptr, int device = 0, rt_stats* stats = nullptr) const {
        FlatScene flat = flatten(world);
        rt_scene_desc sd = flat.desc();
        rt_scene* scene = nullptr;
        if (rt_scene_create(&sd, device, &scene) != RT_OK) throw Error(rt_last_error());
        rt_camera_desc cd = desc();
        rt_options opt{};
        if (options) opt = *options;
        uint32_t rows = 0;
        if (rt_output_rows(&cd, &opt, &rows) != RT_OK) {
            rt_scene_destroy(scene);
            throw Error(rt_last_error());
        }
        Canvas canvas;
        canvas.width = hsize, canvas.height = rows;
        canvas.rgb.assign(static_cast<size_t>(rows) * hsize * 3, 0.0f);
        int rc = rt_render(scene, &cd, &opt, canvas.rgb.data(), stats);
        rt_scene_destroy(scene);
        if (rc != RT_OK) throw Error(rt_last_error());
        return canvas;
    }
    // `camera.render(&world).to_image()` — the last two lines of every example of the reference (e.g.
    // examples/cover-image.rs:196) — as ONE call: the frame is quantised on the GPU (canvas.rs:31-45) and a byte per
    // channel crosses PCIe instead of a float.  rows x hsize x 3 bytes, row 0 = top.
    std::vector<uint8_t> render_image(const World& world, const rt_options* options = nullptr, int device = 0, rt_stats* stats = nullptr) const {
        FlatScene flat = flatten(world);
        rt_scene_desc sd = flat.desc();
        rt_scene* scene = nullptr;
        if (rt_scene_create(&sd, device, &scene) != RT_OK) throw Error(rt_last_error());
        rt_camera_desc cd = desc();
        rt_options opt{};
        if (options) opt = *options;
        uint32_t rows = 0;
        if (rt_output_rows(&cd, &opt, &rows) != RT_OK) {
            rt_scene_destroy(scene);
            throw Error(rt_last_error());
        }
        std::vector<uint8_t> img(static_cast<size_t>(rows) * hsize * 3, 0);
        int rc = rt_render_u8(scene, &cd, &opt, img.data(), stats);
        rt_scene_destroy(scene);
        if (rc != RT_OK) throw Error(rt_last_error());
        return img;
    }
};

struct Scene {
    World world;
    Camera camera;
};

}  // namespace rt
