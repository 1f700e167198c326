// rt_oracle.hpp — f64 CPU restatement of regexPattern/raytracer's per-pixel path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under raytracer_amd/ or include/ may include, link or
// call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it
// (as the checker / the reported CPU baseline, never as the thing shipped).
//
// Pinning: the reference is Rust and no cargo/rustc exists in this image, so it cannot be
// built or run here.  This restatement is pinned by the known-answer values held by the
// reference's own #[test] functions (transcribed as values in tests/cpp/oracle_kat.cpp);
// full-frame output of the five examples is "parity unpinned" (the reference ships no golden
// images) — for those this file IS the reference statement, by construction from the source.
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Semantics kept on purpose: f64 everywhere, left-to-right accumulation, `powi` as repeated
// multiply, `%` = fmod, `hypot`, approximate comparisons with EPSILON = 1e-5, `max_by` keeps the
// last / `min_by` the first of "equal" elements, IEEE division by zero, sort with the approximate
// comparator (insertion sort == what rustc's sort_unstable_by does for <= 20 elements), the
// per-light duplication of reflected + refracted colour, `as u8` saturating truncation.
#pragma once

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace oracle {

using f64 = double;

// ---------------------------------------------------------------------------------------------
// float.rs:3-37
// ---------------------------------------------------------------------------------------------
constexpr f64 EPSILON = 1e-5;
constexpr f64 INF = std::numeric_limits<f64>::infinity();

inline bool approx(f64 a, f64 b) {  // float.rs:5-11
    if (std::isinf(a) && std::isinf(b)) return a == b;
    return std::fabs(a - b) <= EPSILON;
}
inline bool ge(f64 a, f64 b) { return approx(a, b) || a > b; }  // float.rs:21-23
inline bool le(f64 a, f64 b) { return approx(a, b) || a < b; }  // float.rs:25-27

enum class Ordering { Less, Equal, Greater };
inline Ordering partial_cmp(f64 a, f64 b) {  // float.rs:29-37 (NaN falls through to Greater)
    if (approx(a, b)) return Ordering::Equal;
    if (a < b) return Ordering::Less;
    return Ordering::Greater;
}

// Iterator::max_by / min_by of Rust's core: max_by keeps the LAST of equal maxima
// (fold: Greater => acc, else new), min_by keeps the FIRST (fold: Greater => new, else acc).
template <class Cmp>
inline f64 max_by3(f64 a, f64 b, f64 c, Cmp cmp) {
    f64 m = a;
    m = (cmp(m, b) == Ordering::Greater) ? m : b;
    m = (cmp(m, c) == Ordering::Greater) ? m : c;
    return m;
}
template <class Cmp>
inline f64 min_by3(f64 a, f64 b, f64 c, Cmp cmp) {
    f64 m = a;
    m = (cmp(m, b) == Ordering::Greater) ? b : m;
    m = (cmp(m, c) == Ordering::Greater) ? c : m;
    return m;
}
inline Ordering exact_cmp(f64 a, f64 b) {  // f64::partial_cmp(..).unwrap() on non-NaN input
    if (a < b) return Ordering::Less;
    if (a > b) return Ordering::Greater;
    return Ordering::Equal;
}

inline f64 powi(f64 x, int n) {  // compiler-rt __powidf2: square-and-multiply
    bool recip = n < 0;
    f64 r = 1.0;
    for (;;) {
        if (n & 1) r *= x;
        n /= 2;
        if (n == 0) break;
        x *= x;
    }
    return recip ? 1.0 / r : r;
}

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct RayStats {  // one "ray" = one World::intersect call in reference semantics
    uint64_t primary = 0, reflect = 0, refract = 0, shadow = 0;
    uint64_t shaded_hits = 0, light_cells = 0, pattern_lookups = 0;
    uint64_t prim_tests[7] = {0, 0, 0, 0, 0, 0, 0};  // indexed by Shape::Kind (GROUP = box tests)
    void add(const RayStats& r) {
        primary += r.primary, reflect += r.reflect, refract += r.refract, shadow += r.shadow;
        shaded_hits += r.shaded_hits, light_cells += r.light_cells, pattern_lookups += r.pattern_lookups;
        for (int i = 0; i < 7; ++i) prim_tests[i] += r.prim_tests[i];
    }
};
// Per-thread sink for primitive-test counts (set by Camera::render workers; null = off).
inline thread_local RayStats* tls_stats = nullptr;
inline thread_local bool tls_debug = false;  // oracle_probe_pixels(debug): print every shaded path node
// Path trace of one pixel for the parity census (scripts/census_classify.py): one event per shaded path
// node — 24 doubles: node, t, point[3], normal[3], inside, n1, n2, pattern colour[3], light intensity of
// the first four lights, reflection traced, refraction traced, shape kind, 3 spare.
constexpr int kTraceEventDoubles = 24;
inline thread_local std::vector<f64>* tls_trace = nullptr;

// ---------------------------------------------------------------------------------------------
// tuple.rs:74-236 — Point (w = 1) and Vector (w = 0).  The w lane is carried implicitly: every
// operation below is the x/y/z part of the reference's 4-lane arithmetic, in the same order.
// ---------------------------------------------------------------------------------------------
struct Vector {
    f64 x = 0, y = 0, z = 0;
    static Vector make(f64 x, f64 y, f64 z) { return Vector{x, y, z}; }
    f64 magnitude() const { return std::sqrt(x * x + y * y + z * z); }  // tuple.rs:92-94
    f64 dot(const Vector& r) const { return x * r.x + y * r.y + z * r.z; }  // tuple.rs:107-109
    Vector cross(const Vector& r) const {  // tuple.rs:112-118
        return Vector{y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x};
    }
    Vector operator+(const Vector& r) const { return Vector{x + r.x, y + r.y, z + r.z}; }
    Vector operator-(const Vector& r) const { return Vector{x - r.x, y - r.y, z - r.z}; }
    Vector operator*(f64 s) const { return Vector{x * s, y * s, z * s}; }  // tuple.rs:208-218
    Vector operator-() const { return Vector{0.0 - x, 0.0 - y, 0.0 - z}; }  // tuple.rs:200-206
    // tuple.rs:228-236: division is multiplication by the reciprocal and fails for |rhs| <= eps.
    bool try_div(f64 rhs, Vector& out) const {
        if (approx(rhs, 0.0)) return false;
        out = (*this) * (1.0 / rhs);
        return true;
    }
    bool try_normalize(Vector& out) const { return try_div(magnitude(), out); }  // tuple.rs:102-104
    Vector normalize() const {
        Vector o;
        if (!try_normalize(o)) throw Error("tried to normalize a null vector");
        return o;
    }
    Vector reflect(const Vector& n) const { return (*this) - n * 2.0 * dot(n); }  // tuple.rs:121-123
    bool operator==(const Vector& r) const { return approx(x, r.x) && approx(y, r.y) && approx(z, r.z); }
};

struct Point {
    f64 x = 0, y = 0, z = 0;
    static Point make(f64 x, f64 y, f64 z) { return Point{x, y, z}; }
    Point operator+(const Vector& r) const { return Point{x + r.x, y + r.y, z + r.z}; }
    Point operator-(const Vector& r) const { return Point{x - r.x, y - r.y, z - r.z}; }
    Vector operator-(const Point& r) const { return Vector{x - r.x, y - r.y, z - r.z}; }
    bool operator==(const Point& r) const { return approx(x, r.x) && approx(y, r.y) && approx(z, r.z); }
};

// ---------------------------------------------------------------------------------------------
// color.rs:49-103, color/consts.rs
// ---------------------------------------------------------------------------------------------
struct Color {
    f64 red = 0, green = 0, blue = 0;
    Color operator+(const Color& r) const { return Color{red + r.red, green + r.green, blue + r.blue}; }
    Color operator-(const Color& r) const { return Color{red - r.red, green - r.green, blue - r.blue}; }
    Color operator*(f64 s) const { return Color{red * s, green * s, blue * s}; }
    Color operator*(const Color& r) const { return Color{red * r.red, green * r.green, blue * r.blue}; }
    bool operator==(const Color& r) const {
        return approx(red, r.red) && approx(green, r.green) && approx(blue, r.blue);
    }
};
namespace color {
constexpr Color WHITE{1.0, 1.0, 1.0};
constexpr Color BLACK{0.0, 0.0, 0.0};
constexpr Color RED{1.0, 0.0, 0.0};
constexpr Color GREEN{0.0, 1.0, 0.0};
constexpr Color BLUE{0.0, 0.0, 1.0};
constexpr Color LIGHT_SKY_BLUE{0.5294, 0.8078, 0.9804};
constexpr Color DIRT{0.6078, 0.4627, 0.3255};
}  // namespace color

// ---------------------------------------------------------------------------------------------
// matrix.rs:39-198 — 4x4 f64, cofactor inverse, fold-from-0.0 products.
// ---------------------------------------------------------------------------------------------
struct Matrix4 {
    f64 m[4][4];

    static Matrix4 identity() {
        Matrix4 r{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = (i == j) ? 1.0 : 0.0;
        return r;
    }
    Matrix4 transpose() const {  // matrix.rs:100-110
        Matrix4 r{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.m[i][j] = m[j][i];
        return r;
    }
    Matrix4 operator*(const Matrix4& rhs) const {  // matrix.rs:168-182
        Matrix4 r{};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                f64 acc = 0.0;
                for (int k = 0; k < 4; ++k) acc = acc + m[i][k] * rhs.m[k][j];
                r.m[i][j] = acc;
            }
        return r;
    }
    static f64 det2(const f64 a[2][2]) { return a[0][0] * a[1][1] - a[0][1] * a[1][0]; }  // :39-43
    static f64 sign(int row, int col) { return ((row + col) & 1) ? -1.0 : 1.0; }  // (-1)^(r+c)
    static f64 det3(const f64 a[3][3]) {  // matrix.rs:82-96 (row-0 cofactor expansion)
        f64 acc = 0.0;
        for (int col = 0; col < 3; ++col) {
            f64 sub[2][2];
            int rr = 0;
            for (int i = 0; i < 3; ++i) {
                if (i == 0) continue;
                int cc = 0;
                for (int j = 0; j < 3; ++j) {
                    if (j == col) continue;
                    sub[rr][cc++] = a[i][j];
                }
                ++rr;
            }
            acc = acc + a[0][col] * (sign(0, col) * det2(sub));
        }
        return acc;
    }
    f64 minor(int row, int col) const {  // matrix.rs:112-122
        f64 sub[3][3];
        int rr = 0;
        for (int i = 0; i < 4; ++i) {
            if (i == row) continue;
            int cc = 0;
            for (int j = 0; j < 4; ++j) {
                if (j == col) continue;
                sub[rr][cc++] = m[i][j];
            }
            ++rr;
        }
        return det3(sub);
    }
    f64 cofactor(int row, int col) const { return sign(row, col) * minor(row, col); }  // :124-126
    f64 determinant() const {  // matrix.rs:128-134
        f64 acc = 0.0;
        for (int col = 0; col < 4; ++col) acc = acc + m[0][col] * cofactor(0, col);
        return acc;
    }
    bool try_inverse(Matrix4& inv) const {  // matrix.rs:136-151
        f64 det = determinant();
        if (approx(det, 0.0)) return false;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) inv.m[j][i] = cofactor(i, j) / det;
        return true;
    }
    bool operator==(const Matrix4& r) const {  // matrix.rs:25-37
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                if (!approx(m[i][j], r.m[i][j])) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// transform.rs:133-312
// ---------------------------------------------------------------------------------------------
struct Transform {
    Matrix4 mat = Matrix4::identity();

    static Transform translation(f64 x, f64 y, f64 z) {  // transform.rs:141-148
        Transform t;
        t.mat.m[0][3] = x;
        t.mat.m[1][3] = y;
        t.mat.m[2][3] = z;
        return t;
    }
    static Transform scaling(f64 x, f64 y, f64 z) {  // transform.rs:157-166
        if (approx(x * y * z, 0.0)) throw Error("components cannot be scaled to zero");
        Transform t;
        t.mat.m[0][0] = x;
        t.mat.m[1][1] = y;
        t.mat.m[2][2] = z;
        return t;
    }
    static Transform rotation_x(f64 r) {  // transform.rs:169-176
        Transform t;
        t.mat.m[1][1] = std::cos(r);
        t.mat.m[1][2] = -std::sin(r);
        t.mat.m[2][1] = std::sin(r);
        t.mat.m[2][2] = std::cos(r);
        return t;
    }
    static Transform rotation_y(f64 r) {  // transform.rs:179-186
        Transform t;
        t.mat.m[0][0] = std::cos(r);
        t.mat.m[0][2] = std::sin(r);
        t.mat.m[2][0] = -std::sin(r);
        t.mat.m[2][2] = std::cos(r);
        return t;
    }
    static Transform rotation_z(f64 r) {  // transform.rs:189-196
        Transform t;
        t.mat.m[0][0] = std::cos(r);
        t.mat.m[0][1] = -std::sin(r);
        t.mat.m[1][0] = std::sin(r);
        t.mat.m[1][1] = std::cos(r);
        return t;
    }
    static Transform shearing(f64 xy, f64 xz, f64 yx, f64 yz, f64 zx, f64 zy) {  // :208-227
        if (approx(xz * yx * zy + xy * yz * zx - xy * yx - xz * zx - yz * zy + 1.0, 0.0))
            throw Error("invalid relation between shearing components");
        Transform t;
        t.mat.m[0][1] = xy;
        t.mat.m[0][2] = xz;
        t.mat.m[1][0] = yx;
        t.mat.m[1][2] = yz;
        t.mat.m[2][0] = zx;
        t.mat.m[2][1] = zy;
        return t;
    }
    static Transform view(const Point& from, const Point& to, const Vector& up) {  // :251-275
        Vector forward, upn;
        if (!(to - from).try_normalize(forward)) throw Error("`from` and `to` points cannot be equal");
        if (!up.try_normalize(upn)) throw Error("up direction cannot be null");
        Vector left = forward.cross(upn);
        if (left == Vector{0.0, 0.0, 0.0}) throw Error("`from` and `up` vectors cannot be collinear");
        Vector true_up = left.cross(forward);
        Transform o;
        o.mat.m[0][0] = left.x, o.mat.m[0][1] = left.y, o.mat.m[0][2] = left.z;
        o.mat.m[1][0] = true_up.x, o.mat.m[1][1] = true_up.y, o.mat.m[1][2] = true_up.z;
        o.mat.m[2][0] = -forward.x, o.mat.m[2][1] = -forward.y, o.mat.m[2][2] = -forward.z;
        return o * translation(-from.x, -from.y, -from.z);
    }
    Transform inverse() const {  // transform.rs:277-282
        Transform r;
        if (!mat.try_inverse(r.mat)) throw Error("tried to inverse a singular matrix");
        return r;
    }
    Transform transpose() const { return Transform{mat.transpose()}; }
    Transform operator*(const Transform& r) const { return Transform{mat * r.mat}; }
    // matrix.rs:184-198: each row is fold(0.0, acc + m[i][k]*rhs[k]) over x,y,z,w.
    Point operator*(const Point& p) const {
        const auto& a = mat.m;
        return Point{(((0.0 + a[0][0] * p.x) + a[0][1] * p.y) + a[0][2] * p.z) + a[0][3] * 1.0,
                     (((0.0 + a[1][0] * p.x) + a[1][1] * p.y) + a[1][2] * p.z) + a[1][3] * 1.0,
                     (((0.0 + a[2][0] * p.x) + a[2][1] * p.y) + a[2][2] * p.z) + a[2][3] * 1.0};
    }
    Vector operator*(const Vector& v) const {
        const auto& a = mat.m;
        return Vector{(((0.0 + a[0][0] * v.x) + a[0][1] * v.y) + a[0][2] * v.z) + a[0][3] * 0.0,
                      (((0.0 + a[1][0] * v.x) + a[1][1] * v.y) + a[1][2] * v.z) + a[1][3] * 0.0,
                      (((0.0 + a[2][0] * v.x) + a[2][1] * v.y) + a[2][2] * v.z) + a[2][3] * 0.0};
    }
    bool operator==(const Transform& r) const { return mat == r.mat; }
};

// ---------------------------------------------------------------------------------------------
// ray.rs:12-22
// ---------------------------------------------------------------------------------------------
struct Ray {
    Point origin;
    Vector direction;
    Point position(f64 t) const { return origin + direction * t; }
    Ray transform(const Transform& t) const { return Ray{t * origin, t * direction}; }
};

// ---------------------------------------------------------------------------------------------
// pattern.rs:16-105
// ---------------------------------------------------------------------------------------------
struct Pattern3DSpec {
    Color color_a, color_b;
    Transform transform, transform_inverse;
    static Pattern3DSpec make(Color a, Color b, Transform t) {  // pattern.rs:48-55
        return Pattern3DSpec{a, b, t, t.inverse()};
    }
    bool operator==(const Pattern3DSpec& r) const {
        return color_a == r.color_a && color_b == r.color_b && transform == r.transform &&
               transform_inverse == r.transform_inverse;
    }
};

struct Pattern3D {
    enum Kind { SOLID, STRIPE, GRADIENT, RING, CHECKER } kind = SOLID;
    Color solid = color::WHITE;
    Pattern3DSpec spec{};

    static Pattern3D Solid(Color c) {
        Pattern3D p;
        p.kind = SOLID;
        p.solid = c;
        return p;
    }
    static Pattern3D with(Kind k, const Pattern3DSpec& s) {
        Pattern3D p;
        p.kind = k;
        p.spec = s;
        return p;
    }
    static Pattern3D Stripe(const Pattern3DSpec& s) { return with(STRIPE, s); }
    static Pattern3D Gradient(const Pattern3DSpec& s) { return with(GRADIENT, s); }
    static Pattern3D Ring(const Pattern3DSpec& s) { return with(RING, s); }
    static Pattern3D Checker(const Pattern3DSpec& s) { return with(CHECKER, s); }

    Transform transform_inverse() const {  // pattern.rs:98-105
        return kind == SOLID ? Transform{} : spec.transform_inverse;
    }
    Color color_at(const Point& p) const {  // pattern.rs:68-96
        const f64 x = p.x, y = p.y, z = p.z;
        switch (kind) {
            case SOLID: return solid;
            case STRIPE: return approx(std::fmod(std::floor(x), 2.0), 0.0) ? spec.color_a : spec.color_b;
            case GRADIENT: return spec.color_a + (spec.color_b - spec.color_a) * (x - std::floor(x));
            case RING:
                return approx(std::fmod(std::floor(std::hypot(x, z)), 2.0), 0.0) ? spec.color_a : spec.color_b;
            case CHECKER:
                return approx(std::fmod(std::floor(x) + std::floor(y) + std::floor(z), 2.0), 0.0)
                           ? spec.color_a
                           : spec.color_b;
        }
        return solid;
    }
    bool operator==(const Pattern3D& r) const {
        if (kind != r.kind) return false;
        return kind == SOLID ? solid == r.solid : spec == r.spec;
    }
};

// ---------------------------------------------------------------------------------------------
// material.rs:11-115
// ---------------------------------------------------------------------------------------------
namespace material {
constexpr f64 VACUUM_INDEX_OF_REFRACTION = 1.0;
constexpr f64 AIR_INDEX_OF_REFRACTION = 1.00029;
constexpr f64 WATER_INDEX_OF_REFRACTION = 1.333;
constexpr f64 GLASS_INDEX_OF_REFRACTION = 1.458;
constexpr f64 DIAMOND_INDEX_OF_REFRACTION = 2.417;
}  // namespace material

struct Material {  // defaults: material.rs:89-102
    Pattern3D pattern = Pattern3D::Solid(color::WHITE);
    f64 ambient = 0.1;
    f64 diffuse = 0.9;
    f64 specular = 0.9;
    f64 shininess = 200.0;
    f64 index_of_refraction = material::VACUUM_INDEX_OF_REFRACTION;
    f64 reflectivity = 0.0;
    f64 transparency = 0.0;
    bool operator==(const Material& r) const {  // material.rs:104-115
        return pattern == r.pattern && approx(ambient, r.ambient) && approx(diffuse, r.diffuse) &&
               approx(index_of_refraction, r.index_of_refraction) && approx(reflectivity, r.reflectivity) &&
               approx(shininess, r.shininess) && approx(specular, r.specular) &&
               approx(transparency, r.transparency);
    }
};

// ---------------------------------------------------------------------------------------------
// shape/bounding_box.rs:5-147
// ---------------------------------------------------------------------------------------------
struct BoundingBox {
    Point min{INF, INF, INF};
    Point max{-INF, -INF, -INF};

    void add(const Point& p) {  // bounding_box.rs:39-47 (f64::min/max ignore NaN == fmin/fmax)
        min.x = std::fmin(p.x, min.x), min.y = std::fmin(p.y, min.y), min.z = std::fmin(p.z, min.z);
        max.x = std::fmax(p.x, max.x), max.y = std::fmax(p.y, max.y), max.z = std::fmax(p.z, max.z);
    }
    void merge(const BoundingBox& r) {  // :49-52
        add(r.min);
        add(r.max);
    }
    static bool is_between_range(f64 x, f64 lo, f64 hi) { return ge(x, lo) && le(x, hi); }  // :145-147
    bool contains_point(const Point& p) const {  // :54-58
        return is_between_range(p.x, min.x, max.x) && is_between_range(p.y, min.y, max.y) &&
               is_between_range(p.z, min.z, max.z);
    }
    bool contains(const BoundingBox& o) const { return contains_point(o.min) && contains_point(o.max); }
    BoundingBox transform(const Transform& t) const {  // :64-79
        const Point corners[8] = {min,
                                  Point{min.x, min.y, max.z},
                                  Point{min.x, max.y, min.z},
                                  Point{min.x, max.y, max.z},
                                  Point{max.x, min.y, min.z},
                                  Point{max.x, min.y, max.z},
                                  Point{max.x, max.y, min.z},
                                  max};
        BoundingBox b;
        for (const auto& c : corners) b.add(t * c);
        return b;
    }
    bool intersect(const Ray& ray) const;  // :81-84 (defined after the cube slab test)
    void split(BoundingBox& left, BoundingBox& right) const {  // :86-142
        f64 dx = std::fabs(min.x - max.x), dy = std::fabs(min.y - max.y), dz = std::fabs(min.z - max.z);
        f64 largest = max_by3(dx, dy, dz, exact_cmp);
        f64 x0 = min.x, y0 = min.y, z0 = min.z, x1 = max.x, y1 = max.y, z1 = max.z;
        if (approx(largest, dx)) {
            f64 tmp = x0 + dx / 2.0;
            x0 = tmp, x1 = tmp;
        } else if (approx(largest, dy)) {
            f64 tmp = y0 + dy / 2.0;
            y0 = tmp, y1 = tmp;
        } else {
            f64 tmp = z0 + dz / 2.0;
            z0 = tmp, z1 = tmp;
        }
        left = BoundingBox{min, Point{x1, y1, z1}};
        right = BoundingBox{Point{x0, y0, z0}, max};
    }
    bool operator==(const BoundingBox& r) const { return min == r.min && max == r.max; }
};

// cube.rs:79-97
inline void check_axis(f64 origin, f64 direction, f64 mn, f64 mx, f64& tmin, f64& tmax) {
    f64 tmin_num = mn - origin, tmax_num = mx - origin;
    if (ge(std::fabs(direction), EPSILON)) {  // approximate ge: true for every finite direction
        tmin = tmin_num / direction;
        tmax = tmax_num / direction;
    } else {
        tmin = tmin_num * INF;
        tmax = tmax_num * INF;
    }
    if (tmin > tmax) std::swap(tmin, tmax);
}
// cube.rs:39-76
inline void intersect_box_with_bounding_box(const Ray& ray, const BoundingBox& bb, f64& tmin, f64& tmax) {
    f64 xtmin, xtmax, ytmin, ytmax, ztmin, ztmax;
    check_axis(ray.origin.x, ray.direction.x, bb.min.x, bb.max.x, xtmin, xtmax);
    check_axis(ray.origin.y, ray.direction.y, bb.min.y, bb.max.y, ytmin, ytmax);
    check_axis(ray.origin.z, ray.direction.z, bb.min.z, bb.max.z, ztmin, ztmax);
    tmin = max_by3(xtmin, ytmin, ztmin, partial_cmp);
    tmax = min_by3(xtmax, ytmax, ztmax, partial_cmp);
}
inline bool BoundingBox::intersect(const Ray& ray) const {
    f64 tmin, tmax;
    intersect_box_with_bounding_box(ray, *this, tmin, tmax);
    return tmin < tmax;
}

// ---------------------------------------------------------------------------------------------
// shape/object.rs:6-52, shape.rs:33-137, shape/*.rs
// ---------------------------------------------------------------------------------------------
struct ObjectCache {
    Material material{};
    Transform transform{};
    Transform transform_inverse{};
    BoundingBox bounding_box{};
    BoundingBox parent_space_bounding_box{};
    static ObjectCache make(const Material& m, const Transform& t, const BoundingBox& bb) {  // object.rs:43-51
        return ObjectCache{m, t, t.inverse(), bb, bb.transform(t)};
    }
    bool operator==(const ObjectCache& r) const {
        return material == r.material && transform == r.transform &&
               transform_inverse == r.transform_inverse && bounding_box == r.bounding_box &&
               parent_space_bounding_box == r.parent_space_bounding_box;
    }
};

struct ShapeBuilder {  // shape.rs:70-77
    Material material{};
    Transform transform{};
};
struct CylinderBuilder {  // cylinder.rs:50-86
    Material material{};
    Transform transform{};
    f64 min = -INF;
    f64 max = INF;
    bool closed = false;
};
struct TriangleBuilder {  // triangle.rs:59-66
    Material material{};
    std::array<Point, 3> vertices{};
};

struct Shape;
struct Intersection {  // intersection.rs:9-14
    f64 t = 0;
    const Shape* object = nullptr;
    bool has_uv = false;
    f64 u = 0, v = 0;
};

struct Group;

struct Shape {
    enum Kind { CUBE, CYLINDER, GROUP, PLANE, SMOOTH_TRIANGLE, SPHERE, TRIANGLE } kind = SPHERE;
    ObjectCache cache{};
    // cylinder.rs:42-47
    f64 cyl_min = -INF, cyl_max = INF;
    bool closed = false;
    // triangle.rs:48-56, smooth_triangle.rs:14-19
    Point v0{}, v1{}, v2{};
    Vector e0{}, e1{}, normal{}, n0{}, n1{}, n2{};
    // group.rs:36-39
    std::vector<Shape> children;

    static Shape Sphere(const ShapeBuilder& b = {}) {  // sphere.rs:21-35
        Shape s;
        s.kind = SPHERE;
        s.cache = ObjectCache::make(b.material, b.transform, BoundingBox{Point{-1, -1, -1}, Point{1, 1, 1}});
        return s;
    }
    static Shape Cube(const ShapeBuilder& b = {}) {  // cube.rs:22-36
        Shape s;
        s.kind = CUBE;
        s.cache = ObjectCache::make(b.material, b.transform, BoundingBox{Point{-1, -1, -1}, Point{1, 1, 1}});
        return s;
    }
    static Shape Plane(const ShapeBuilder& b = {}) {  // plane.rs:22-36
        Shape s;
        s.kind = PLANE;
        s.cache = ObjectCache::make(b.material, b.transform,
                                    BoundingBox{Point{-INF, 0.0, -INF}, Point{INF, 0.0, INF}});
        return s;
    }
    static Shape Cylinder(const CylinderBuilder& b = {}) {  // cylinder.rs:88-114
        Shape s;
        s.kind = CYLINDER;
        s.cache = ObjectCache::make(b.material, b.transform,
                                    BoundingBox{Point{-1.0, b.min, -1.0}, Point{1.0, b.max, 1.0}});
        s.cyl_min = b.min, s.cyl_max = b.max, s.closed = b.closed;
        return s;
    }
    static Shape Triangle(const TriangleBuilder& b) {  // triangle.rs:68-101
        Shape s;
        s.kind = TRIANGLE;
        s.v0 = b.vertices[0], s.v1 = b.vertices[1], s.v2 = b.vertices[2];
        s.e0 = s.v1 - s.v0;
        s.e1 = s.v2 - s.v0;
        if (!s.e1.cross(s.e0).try_normalize(s.normal)) throw Error("triangle sides most not be collinear");
        BoundingBox bb;
        bb.add(s.v0), bb.add(s.v1), bb.add(s.v2);
        s.cache = ObjectCache::make(b.material, Transform{}, bb);
        return s;
    }
    // smooth_triangle.rs:14-19 (the reference only builds these inside model.rs)
    static Shape SmoothTriangle(const TriangleBuilder& b, Vector n0, Vector n1, Vector n2) {
        Shape s = Triangle(b);
        s.kind = SMOOTH_TRIANGLE;
        s.n0 = n0, s.n1 = n1, s.n2 = n2;
        return s;
    }
    static Shape Group(oracle::Group g);

    bool operator==(const Shape& r) const {  // derive(PartialEq) on shape.rs:32 and every variant
        if (this == &r) return true;
        if (kind != r.kind) return false;
        if (!(cache == r.cache)) return false;
        switch (kind) {
            case CYLINDER:
                return approx(cyl_min, r.cyl_min) && approx(cyl_max, r.cyl_max) && closed == r.closed;
            case TRIANGLE:
            case SMOOTH_TRIANGLE:
                if (!(v0 == r.v0 && v1 == r.v1 && v2 == r.v2 && e0 == r.e0 && e1 == r.e1 && normal == r.normal))
                    return false;
                return kind == TRIANGLE || (n0 == r.n0 && n1 == r.n1 && n2 == r.n2);
            case GROUP:
                if (children.size() != r.children.size()) return false;
                for (size_t i = 0; i < children.size(); ++i)
                    if (!(children[i] == r.children[i])) return false;
                return true;
            default: return true;
        }
    }

    // ---- per-variant local intersections ------------------------------------------------------
    void sphere_intersect(const Ray& lr, std::vector<Intersection>& xs) const {  // sphere.rs:38-72
        Vector o = lr.origin - Point{0.0, 0.0, 0.0};
        f64 a = lr.direction.dot(lr.direction);
        f64 b = 2.0 * lr.direction.dot(o);
        f64 c = o.dot(o) - 1.0;
        f64 disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return;
        f64 t0 = (-b - std::sqrt(disc)) / (2.0 * a);
        f64 t1 = (-b + std::sqrt(disc)) / (2.0 * a);
        xs.push_back(Intersection{t0, this});
        xs.push_back(Intersection{t1, this});
    }
    void plane_intersect(const Ray& r, std::vector<Intersection>& xs) const {  // plane.rs:39-51
        if (!approx(r.direction.y, 0.0)) xs.push_back(Intersection{-r.origin.y / r.direction.y, this});
    }
    void cube_intersect(const Ray& r, std::vector<Intersection>& xs) const {  // cube.rs:101-122
        f64 tmin, tmax;
        intersect_box_with_bounding_box(r, cache.bounding_box, tmin, tmax);
        if (tmin > tmax) return;
        xs.push_back(Intersection{tmin, this});
        xs.push_back(Intersection{tmax, this});
    }
    static bool check_cap(const Ray& r, f64 t) {  // cylinder.rs:220-225
        f64 x = r.origin.x + t * r.direction.x;
        f64 z = r.origin.z + t * r.direction.z;
        return le(x * x + z * z, 1.0);
    }
    void cylinder_caps(const Ray& r, std::vector<Intersection>& xs) const {  // cylinder.rs:186-217
        if (!closed || approx(r.direction.y, 0.0)) return;
        f64 t = (cyl_min - r.origin.y) / r.direction.y;
        if (check_cap(r, t)) xs.push_back(Intersection{t, this});
        t = (cyl_max - r.origin.y) / r.direction.y;
        if (check_cap(r, t)) xs.push_back(Intersection{t, this});
    }
    void cylinder_intersect(const Ray& r, std::vector<Intersection>& xs) const {  // cylinder.rs:126-170
        f64 a = r.direction.x * r.direction.x + r.direction.z * r.direction.z;
        if (approx(a, 0.0)) return cylinder_caps(r, xs);
        f64 b = 2.0 * r.origin.x * r.direction.x + 2.0 * r.origin.z * r.direction.z;
        f64 c = r.origin.x * r.origin.x + r.origin.z * r.origin.z - 1.0;
        f64 disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return;
        f64 t0 = (-b - std::sqrt(disc)) / (2.0 * a);
        f64 t1 = (-b + std::sqrt(disc)) / (2.0 * a);
        if (t0 > t1) std::swap(t0, t1);
        f64 y0 = r.origin.y + t0 * r.direction.y;
        if (cyl_min < y0 && y0 < cyl_max) xs.push_back(Intersection{t0, this});
        f64 y1 = r.origin.y + t1 * r.direction.y;
        if (cyl_min < y1 && y1 < cyl_max) xs.push_back(Intersection{t1, this});
        cylinder_caps(r, xs);
    }
    void triangle_intersect(const Ray& r, std::vector<Intersection>& xs) const {  // triangle.rs:104-133
        Vector dir_cross_e1 = r.direction.cross(e1);
        f64 det = e0.dot(dir_cross_e1);
        if (approx(std::fabs(det), 0.0)) return;
        f64 f = 1.0 / det;
        Vector p0_to_origin = r.origin - v0;
        f64 u = f * p0_to_origin.dot(dir_cross_e1);
        if (!(u >= 0.0 && u <= 1.0)) return;
        Vector origin_cross_e0 = p0_to_origin.cross(e0);
        f64 v = f * r.direction.dot(origin_cross_e0);
        if (v < 0.0 || (u + v) > 1.0) return;
        xs.push_back(Intersection{f * e1.dot(origin_cross_e0), this, true, u, v});
    }
    void group_intersect(const Ray& world_ray, std::vector<Intersection>& xs) const;  // group.rs:106-119

    // shape.rs:99-116
    std::vector<Intersection> intersect(const Ray& ray) const {
        std::vector<Intersection> xs;
        if (tls_stats) tls_stats->prim_tests[kind]++;
        Ray object_ray = ray.transform(cache.transform_inverse);  // shape.rs:79-81 (computed for groups too)
        switch (kind) {
            case CUBE: cube_intersect(object_ray, xs); break;
            case CYLINDER: cylinder_intersect(object_ray, xs); break;
            case PLANE: plane_intersect(object_ray, xs); break;
            case SMOOTH_TRIANGLE: triangle_intersect(object_ray, xs); break;  // smooth_triangle.rs:22-24
            case SPHERE: sphere_intersect(object_ray, xs); break;
            case TRIANGLE: triangle_intersect(object_ray, xs); break;
            case GROUP: group_intersect(ray, xs); break;  // the UNTRANSFORMED world ray
        }
        return xs;
    }

    // ---- normals ------------------------------------------------------------------------------
    Vector local_normal_at(const Point& p, const Intersection& hit) const {
        switch (kind) {
            case CUBE: {  // cube.rs:125-141
                f64 ax = std::fabs(p.x), ay = std::fabs(p.y), az = std::fabs(p.z);
                f64 mc = max_by3(ax, ay, az, exact_cmp);
                if (approx(mc, ax)) return Vector{p.x, 0.0, 0.0};
                if (approx(mc, ay)) return Vector{0.0, p.y, 0.0};
                return Vector{0.0, 0.0, p.z};
            }
            case CYLINDER: {  // cylinder.rs:172-184
                f64 dist = p.x * p.x + p.z * p.z;
                if (dist < 1.0 && ge(p.y, cyl_max - EPSILON)) return Vector{0.0, 1.0, 0.0};
                if (dist < 1.0 && le(p.y, cyl_min + EPSILON)) return Vector{0.0, -1.0, 0.0};
                return Vector{p.x, 0.0, p.z};
            }
            case PLANE: return Vector{0.0, 1.0, 0.0};  // plane.rs:53-55
            case SMOOTH_TRIANGLE:  // smooth_triangle.rs:26-33
                return n1 * hit.u + n2 * hit.v + n0 * (1.0 - hit.u - hit.v);
            case SPHERE: return p - Point{0.0, 0.0, 0.0};  // sphere.rs:74-76
            case TRIANGLE: return normal;  // triangle.rs:135-137
            case GROUP: throw Error("a group is never asked for its normal");  // shape.rs:134
        }
        return Vector{};
    }
    Vector normal_at(const Point& point, const Intersection& hit) const {  // shape.rs:83-96,118-137
        Point object_point = cache.transform_inverse * point;
        Vector object_normal = local_normal_at(object_point, hit);
        Vector world_normal = cache.transform_inverse.transpose() * object_normal;  // w forced to 0
        return world_normal.normalize();
    }
};

// ---------------------------------------------------------------------------------------------
// intersection.rs:105-120 — approximate comparator + insertion sort (what sort_unstable_by does
// for slices of <= 20 elements on the rustc releases contemporary with the reference; for longer
// slices tie order among |dt| <= 1e-5 is toolchain-dependent and therefore unpinned).
// ---------------------------------------------------------------------------------------------
inline void sort_intersections(std::vector<Intersection>& xs) {
    for (size_t i = 1; i < xs.size(); ++i) {
        Intersection key = xs[i];
        size_t j = i;
        while (j > 0) {
            const f64 a = key.t, b = xs[j - 1].t;
            bool less = !approx(a, b) && a < b;
            if (!less) break;
            xs[j] = xs[j - 1];
            --j;
        }
        xs[j] = key;
    }
}
inline bool hit_of(std::vector<Intersection>& xs, Intersection& out) {  // intersection.rs:117-120
    sort_intersections(xs);
    for (const auto& i : xs)
        if (i.t > 0.0) {
            out = i;
            return true;
        }
    return false;
}
inline bool intersection_eq(const Intersection& a, const Intersection& b) {  // intersection.rs:30-37
    if (!approx(a.t, b.t)) return false;
    if (!(*a.object == *b.object)) return false;
    if (a.has_uv != b.has_uv) return false;
    return !a.has_uv || (approx(a.u, b.u) && approx(a.v, b.v));
}

// ---------------------------------------------------------------------------------------------
// shape/group.rs:36-243
// ---------------------------------------------------------------------------------------------
struct GroupBuilder {
    std::vector<Shape> children{};
    Transform transform{};
};

struct Group {
    std::vector<Shape> children;
    ObjectCache object_cache{};  // Default: identity transforms, empty (inf,-inf) boxes

    static Group from(const GroupBuilder& b) {  // group.rs:51-68
        Group g;
        g.object_cache.transform = b.transform;
        g.object_cache.transform_inverse = b.transform.inverse();
        g.extend(b.children);
        return g;
    }
    static void apply_transform_to_child(Shape& child, const Transform& transform) {  // group.rs:81-94
        if (child.kind == Shape::GROUP)
            for (auto& c : child.children) apply_transform_to_child(c, transform);
        Transform nt = transform * child.cache.transform;
        child.cache.transform = nt;
        child.cache.transform_inverse = nt.inverse();
        child.cache.parent_space_bounding_box = child.cache.bounding_box.transform(nt);
    }
    void push(Shape child) {  // group.rs:72-79
        apply_transform_to_child(child, object_cache.transform);
        object_cache.bounding_box.merge(child.cache.parent_space_bounding_box);
        children.push_back(std::move(child));
    }
    void extend(const std::vector<Shape>& cs) {  // group.rs:97-104
        for (const auto& c : cs) push(c);
    }
    BoundingBox bounding_box() const {  // group.rs:233-242
        BoundingBox bb;
        for (const auto& c : children) bb.merge(c.cache.parent_space_bounding_box);
        return bb;
    }
    void partition_children(std::vector<Shape>& left, std::vector<Shape>& right) {  // group.rs:188-219
        BoundingBox lb, rb;
        bounding_box().split(lb, rb);
        size_t i = 0;
        while (i < children.size()) {
            Shape& child = children[i];
            BoundingBox cb = child.cache.parent_space_bounding_box;
            bool to_left = lb.contains(cb);
            bool to_right = !to_left && rb.contains(cb);
            if (to_left || to_right) {
                child.cache.transform = object_cache.transform_inverse * child.cache.transform;
                // Vec::swap_remove(i): take element i, move the LAST element into its slot.
                Shape taken = std::move(children[i]);
                if (i + 1 != children.size()) children[i] = std::move(children.back());
                children.pop_back();
                (to_left ? left : right).push_back(std::move(taken));
            } else {
                ++i;
            }
        }
    }
    void make_subgroup(std::vector<Shape>& cs);  // group.rs:221-231
    void divide(size_t threshold);               // group.rs:168-186
};

inline Shape Shape::Group(oracle::Group g) {
    Shape s;
    s.kind = GROUP;
    s.cache = g.object_cache;
    s.children = std::move(g.children);
    return s;
}
inline oracle::Group group_view_take(Shape& s) {  // move a GROUP shape back into a Group value
    oracle::Group g;
    g.object_cache = s.cache;
    g.children = std::move(s.children);
    return g;
}
inline void Group::make_subgroup(std::vector<Shape>& cs) {
    Group sub;
    for (auto& c : cs) sub.push(std::move(c));
    push(Shape::Group(std::move(sub)));
}
inline void Group::divide(size_t threshold) {
    if (threshold <= children.size()) {
        std::vector<Shape> left, right;
        partition_children(left, right);
        if (!left.empty()) make_subgroup(left);
        if (!right.empty()) make_subgroup(right);
    }
    for (auto& child : children)
        if (child.kind == Shape::GROUP) {
            Group sub = group_view_take(child);
            sub.divide(threshold);
            child.cache = sub.object_cache;
            child.children = std::move(sub.children);
        }
}
inline void Shape::group_intersect(const Ray& ray, std::vector<Intersection>& xs) const {
    BoundingBox bb;  // group.rs:107,233-242: re-merged from the children on every ray
    for (const auto& c : children) bb.merge(c.cache.parent_space_bounding_box);
    if (!bb.intersect(ray)) return;
    for (const auto& c : children) {
        std::vector<Intersection> cx = c.intersect(ray);
        xs.insert(xs.end(), cx.begin(), cx.end());
    }
    sort_intersections(xs);
}

// ---------------------------------------------------------------------------------------------
// light.rs:14-216
// ---------------------------------------------------------------------------------------------
struct PointLight {
    Point position{};
    Color intensity{};
};
struct AreaLightBuilder {
    Point corner{};
    Vector horizontal_dir{};
    size_t horizontal_cells = 1;
    Vector vertical_dir{};
    size_t vertical_cells = 1;
    Color intensity{};
};
struct AreaLight {
    Point corner{};
    Vector uvec{};
    size_t usteps = 1;
    Vector vvec{};
    size_t vsteps = 1;
    size_t samples = 1;
    Color intensity{};
    static AreaLight from(const AreaLightBuilder& b) {  // light.rs:117-144
        AreaLight a;
        a.corner = b.corner;
        if (!b.horizontal_dir.try_div(static_cast<f64>(b.horizontal_cells), a.uvec) ||
            !b.vertical_dir.try_div(static_cast<f64>(b.vertical_cells), a.vvec))
            throw Error("tried to divide a tuple by zero");
        a.usteps = b.horizontal_cells, a.vsteps = b.vertical_cells;
        a.samples = a.usteps * a.vsteps;
        a.intensity = b.intensity;
        return a;
    }
    template <class Jitter>
    Point point_on_light(size_t u, size_t v, Jitter&& jitter) const {  // light.rs:211-216
        f64 ju = jitter();  // the first draw goes to u
        f64 jv = jitter();
        return corner + uvec * (static_cast<f64>(u) + ju) + vvec * (static_cast<f64>(v) + jv);
    }
};
struct Light {
    enum Kind { AREA, POINT } kind = POINT;
    PointLight point{};
    AreaLight area{};
    static Light Point(const PointLight& p) {
        Light l;
        l.kind = POINT;
        l.point = p;
        return l;
    }
    static Light Area(const AreaLightBuilder& b) {
        Light l;
        l.kind = AREA;
        l.area = AreaLight::from(b);
        return l;
    }
    std::vector<oracle::Point> cells() const {  // light.rs:157-171
        std::vector<oracle::Point> out;
        if (kind == AREA) {
            for (size_t v = 0; v < area.vsteps; ++v)
                for (size_t u = 0; u < area.usteps; ++u)
                    out.push_back(area.point_on_light(u, v, [] { return 0.5; }));
        } else {
            out.push_back(point.position);
        }
        return out;
    }
    Color effective_color() const { return kind == AREA ? area.intensity : point.intensity; }  // :173-178
    size_t samples() const { return kind == AREA ? area.samples : 1; }
};

// ---------------------------------------------------------------------------------------------
// Jitter sources.  The reference draws `thread_rng().gen::<u8>() as f64 / 255.0` (light.rs:149-152).
// CENTER (0.5) is the deterministic parity mode; COUNTER is the counter-based hash shared bit for
// bit with the HIP kernel; THREAD_RNG mimics the reference's independent stream (statistical
// comparisons only); SEQUENCE is the cyclic MockJitter of the reference's tests (light.rs:227-234).
// ---------------------------------------------------------------------------------------------
enum class JitterMode { CENTER = 0, COUNTER = 1, THREAD_RNG = 2, SEQUENCE = 3 };

inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// jitter = one byte of hash(seed, pixel, node, light, cell) / 255: the top byte for the
// u draw (axis 0, drawn first: light.rs:215), the next byte for the v draw (axis 1)
inline uint32_t jitter_pixel_key(uint32_t seed, uint32_t pixel) { return fmix32(pixel ^ (seed * 0x9E3779B9u)); }
inline f64 counter_jitter(uint32_t pixel_key, uint32_t node, uint32_t light, uint32_t cell, uint32_t axis) {
    // node < 256 for RECURSION_DEPTH <= 7; deeper trees (the product allows depth 8: nodes up to 510)
    // fold the bits the shift drops into the key, so no two nodes of a pixel share a stream
    uint32_t k = (node << 24) | (light << 16) | cell;
    uint32_t h = fmix32(pixel_key + k * 0x85EBCA77u + (node >> 8) * 0x27D4EB2Fu);
    return static_cast<f64>(axis == 0 ? (h >> 24) : ((h >> 16) & 0xFFu)) / 255.0;
}


struct TraceCtx {
    JitterMode mode = JitterMode::CENTER;
    uint32_t pixel_key = 0;  // jitter_pixel_key(seed, y*W+x)
    uint32_t node = 0;       // heap index in the path tree: root 0, reflection 2i+1, refraction 2i+2
    std::mt19937* rng = nullptr;             // THREAD_RNG
    const std::vector<f64>* sequence = nullptr;  // SEQUENCE
    size_t* seq_pos = nullptr;
    RayStats* stats = nullptr;

    f64 draw(uint32_t light, uint32_t cell, uint32_t axis) const {
        switch (mode) {
            case JitterMode::CENTER: return 0.5;
            case JitterMode::COUNTER: return counter_jitter(pixel_key, node, light, cell, axis);
            case JitterMode::THREAD_RNG: return static_cast<f64>((*rng)() & 0xFFu) / 255.0;
            case JitterMode::SEQUENCE: {
                f64 v = (*sequence)[*seq_pos % sequence->size()];
                ++*seq_pos;
                return v;
            }
        }
        return 0.5;
    }
    TraceCtx child(uint32_t n) const {
        TraceCtx c = *this;
        c.node = n;
        return c;
    }
};

// ---------------------------------------------------------------------------------------------
// intersection.rs:17-141
// ---------------------------------------------------------------------------------------------
struct Computation {
    Vector eyev{};
    bool inside = false;
    Intersection intersection{};
    f64 n1 = 1.0, n2 = 1.0;
    Vector normalv{};
    Point over_point{}, point{};
    Vector reflectv{};
    Point under_point{};

    f64 schlick() const {  // intersection.rs:124-141
        f64 cos = eyev.dot(normalv);
        if (n1 > n2) {
            f64 n = n1 / n2;
            f64 sin2_t = powi(n, 2) * (1.0 - powi(cos, 2));
            if (sin2_t > 1.0) return 1.0;
            cos = std::sqrt(1.0 - sin2_t);
        }
        f64 r0 = powi((n1 - n2) / (n1 + n2), 2);
        return r0 + (1.0 - r0) * powi(1.0 - cos, 5);
    }
};

inline void find_n1_and_n2(const Intersection& hit, const std::vector<Intersection>& xs, f64& n1, f64& n2) {
    // intersection.rs:71-103
    n1 = 1.0, n2 = 1.0;
    std::vector<const Shape*> visited;
    for (const auto& i : xs) {
        bool is_hit = intersection_eq(i, hit);
        if (is_hit && !visited.empty()) n1 = visited.back()->cache.material.index_of_refraction;
        size_t idx = visited.size();
        for (size_t k = 0; k < visited.size(); ++k)
            if (*visited[k] == *i.object) {
                idx = k;
                break;
            }
        if (idx < visited.size())
            visited.erase(visited.begin() + static_cast<long>(idx));
        else
            visited.push_back(i.object);
        if (is_hit) {
            if (!visited.empty()) n2 = visited.back()->cache.material.index_of_refraction;
            break;
        }
    }
}

inline Computation prepare_computation(const Intersection& self, const Ray& ray,
                                       const std::vector<Intersection>& xs) {  // intersection.rs:40-69
    Computation c;
    c.point = ray.position(self.t);
    c.eyev = -ray.direction;
    Vector normalv = self.object->normal_at(c.point, self);
    c.inside = normalv.dot(c.eyev) < 0.0;
    c.normalv = c.inside ? -normalv : normalv;
    c.reflectv = ray.direction.reflect(c.normalv);
    c.over_point = c.point + c.normalv * EPSILON;
    c.under_point = c.point - c.normalv * EPSILON;
    c.intersection = self;
    find_n1_and_n2(self, xs, c.n1, c.n2);
    return c;
}

// ---------------------------------------------------------------------------------------------
// world.rs:11-133, light.rs:147-209, material.rs:118-162
// ---------------------------------------------------------------------------------------------
constexpr uint8_t RECURSION_DEPTH = 5;  // world.rs:11

struct World {
    std::vector<Shape> objects;
    std::vector<Light> lights;

    std::vector<Intersection> intersect(const Ray& ray) const {  // world.rs:32-41
        std::vector<Intersection> xs;
        for (const auto& obj : objects) {
            std::vector<Intersection> ox = obj.intersect(ray);
            xs.insert(xs.end(), ox.begin(), ox.end());
        }
        sort_intersections(xs);
        return xs;
    }

    bool is_shadowed(const Point& light_position, const Point& point, const TraceCtx& ctx) const {  // :73-92
        Vector point_to_light = light_position - point;
        f64 distance = point_to_light.magnitude();
        Vector dir;
        if (!point_to_light.try_normalize(dir)) return false;
        if (ctx.stats) ctx.stats->shadow++;
        Ray shadow_ray{point, dir};
        std::vector<Intersection> xs = intersect(shadow_ray);
        Intersection h;
        if (!hit_of(xs, h)) return false;
        return h.t < distance;
    }

    f64 intensity_at(const Light& light, uint32_t light_index, const Point& point, const TraceCtx& ctx) const {
        if (light.kind == Light::POINT)  // light.rs:182-188
            return is_shadowed(light.point.position, point, ctx) ? 0.0 : 1.0;
        // light.rs:192-209
        const AreaLight& a = light.area;
        f64 total = 0.0;
        for (size_t v = 0; v < a.vsteps; ++v)
            for (size_t u = 0; u < a.usteps; ++u) {
                uint32_t cell = static_cast<uint32_t>(v * a.usteps + u);
                uint32_t axis = 0;
                oracle::Point lp = a.point_on_light(u, v, [&] { return ctx.draw(light_index, cell, axis++); });
                if (!is_shadowed(lp, point, ctx)) total += 1.0;
            }
        return total / static_cast<f64>(a.samples);
    }

    Color lighting(const Material& m, const Shape& object, const Light& light, const Point& point,
                   const Vector& eyev, const Vector& normalv, f64 light_intensity,
                   RayStats* stats = nullptr) const {  // material.rs:118-162
        // pattern.rs:58-66: object space, then pattern space
        Point object_point = object.cache.transform_inverse * point;
        Point pattern_point = m.pattern.transform_inverse() * object_point;
        Color effective_color = m.pattern.color_at(pattern_point) * light.effective_color();
        if (stats) stats->pattern_lookups++;
        Color ambient = effective_color * m.ambient;
        Color light_shade = color::BLACK;
        size_t light_samples = light.samples();
        for (const auto& cell : light.cells()) {
            if (stats) stats->light_cells++;
            Vector lightv;
            if (!(cell - point).try_normalize(lightv)) lightv = Vector{0.0, 0.0, 0.0};
            f64 light_dot_normal = lightv.dot(normalv);
            if (ge(light_dot_normal, 0.0)) {
                Color diffuse_contrib = effective_color * m.diffuse * light_dot_normal;
                light_shade = light_shade + diffuse_contrib;
                Vector reflectv = (-lightv).reflect(normalv);
                f64 reflect_dot_eye = reflectv.dot(eyev);
                if (reflect_dot_eye > 0.0) {
                    f64 factor = std::pow(reflect_dot_eye, m.shininess);
                    Color specular_contrib = light.effective_color() * m.specular * factor;
                    light_shade = light_shade + specular_contrib;
                }
            }
        }
        return ambient + (light_shade * (1.0 / static_cast<f64>(light_samples))) * light_intensity;
    }

    Color reflected_color(const Computation& comps, uint8_t depth, const TraceCtx& ctx) const {  // :94-107
        f64 reflectiveness = comps.intersection.object->cache.material.reflectivity;
        if (approx(reflectiveness, 0.0) || depth == 0) return color::BLACK;
        Ray r{comps.over_point, comps.reflectv};
        if (ctx.stats) ctx.stats->reflect++;
        return color_at(r, depth - 1, ctx.child(2 * ctx.node + 1), false) * reflectiveness;
    }
    Color refracted_color(const Computation& comps, uint8_t depth, const TraceCtx& ctx) const {  // :109-133
        f64 transparency = comps.intersection.object->cache.material.transparency;
        f64 n_ratio = comps.n1 / comps.n2;
        f64 cos_i = comps.eyev.dot(comps.normalv);
        f64 sin2_t = powi(n_ratio, 2) * (1.0 - powi(cos_i, 2));
        bool tir = sin2_t > 1.0;
        if (approx(transparency, 0.0) || depth == 0 || tir) return color::BLACK;
        f64 cos_t = std::sqrt(1.0 - sin2_t);
        Vector direction = comps.normalv * (n_ratio * cos_i - cos_t) - comps.eyev * n_ratio;
        Ray r{comps.under_point, direction};
        if (ctx.stats) ctx.stats->refract++;
        return color_at(r, depth - 1, ctx.child(2 * ctx.node + 2), false) * transparency;
    }

    Color shade_hit(const Computation& comps, uint8_t depth, const TraceCtx& ctx) const {  // world.rs:43-71
        Color acc = color::BLACK;
        if (ctx.stats) ctx.stats->shaded_hits++;
        uint32_t li = 0;
        size_t trace_at = 0;
        if (tls_trace) {  // census only: the decisions taken at this node (the per-light re-traces repeat them)
            const Shape& o = *comps.intersection.object;
            const Material& m = o.cache.material;
            const Color pc = m.pattern.color_at(m.pattern.transform_inverse() * (o.cache.transform_inverse * comps.over_point));
            const f64 nr = comps.n1 / comps.n2, ci = comps.eyev.dot(comps.normalv);
            const bool tir = powi(nr, 2) * (1.0 - powi(ci, 2)) > 1.0;
            trace_at = tls_trace->size();
            const f64 ev[kTraceEventDoubles] = {f64(ctx.node), comps.intersection.t, comps.point.x, comps.point.y, comps.point.z,
                                                comps.normalv.x, comps.normalv.y, comps.normalv.z, comps.inside ? 1.0 : 0.0, comps.n1, comps.n2,
                                                pc.red, pc.green, pc.blue, -1.0, -1.0, -1.0, -1.0,
                                                (!approx(m.reflectivity, 0.0) && depth > 0) ? 1.0 : 0.0,
                                                (!approx(m.transparency, 0.0) && depth > 0 && !tir) ? 1.0 : 0.0, f64(int(o.kind)), 0, 0, 0};
            tls_trace->insert(tls_trace->end(), ev, ev + kTraceEventDoubles);
        }
        for (const auto& light : lights) {
            const Shape& object = *comps.intersection.object;
            const Material& material = object.cache.material;
            f64 light_intensity = intensity_at(light, li, comps.over_point, ctx);
            if (tls_trace && li < 4) (*tls_trace)[trace_at + 14 + li] = light_intensity;
            Color surface = lighting(material, object, light, comps.over_point, comps.eyev, comps.normalv,
                                     light_intensity, ctx.stats);
            Color reflected = reflected_color(comps, depth, ctx);  // recomputed per light (quirk)
            Color refracted = refracted_color(comps, depth, ctx);
            Color reflectance;
            if ((material.reflectivity * material.transparency) > 0.0) {
                f64 r = comps.schlick();
                reflectance = reflected * r + refracted * (1.0 - r);
            } else {
                reflectance = reflected + refracted;
            }
            acc = acc + surface + reflectance;
            if (tls_debug)
                std::printf("  [oracle] node=%u depth=%u light=%u t=%.9g kind=%d p=(%.9g %.9g %.9g) n=(%.9g %.9g %.9g) inside=%d n1=%.6g n2=%.6g I=%.6g surf=(%.9g %.9g %.9g) refl=(%.9g %.9g %.9g) refr=(%.9g %.9g %.9g)\n",
                            ctx.node, unsigned(depth), li, comps.intersection.t, int(object.kind), comps.point.x, comps.point.y, comps.point.z,
                            comps.normalv.x, comps.normalv.y, comps.normalv.z, int(comps.inside), comps.n1, comps.n2, light_intensity,
                            surface.red, surface.green, surface.blue, reflected.red, reflected.green, reflected.blue,
                            refracted.red, refracted.green, refracted.blue);
            ++li;
        }
        return acc;
    }

    Color color_at(const Ray& ray, uint8_t depth, const TraceCtx& ctx = {}, bool primary = true) const {  // :24-30
        if (ctx.stats && primary) ctx.stats->primary++;
        std::vector<Intersection> xs = intersect(ray);
        Intersection h;
        if (!hit_of(xs, h)) return color::BLACK;
        return shade_hit(prepare_computation(h, ray, xs), depth, ctx);
    }
};

// ---------------------------------------------------------------------------------------------
// canvas.rs:8-45 (the HashMap store is replaced by a dense vector; unwritten pixels are BLACK)
// ---------------------------------------------------------------------------------------------
inline uint8_t quantize_u8(f64 c) {  // Rust `(c * 255.0) as u8`: truncate, saturate, NaN -> 0
    f64 v = c * 255.0;
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return static_cast<uint8_t>(v);
}
struct Canvas {
    size_t width = 0, height = 0;
    std::vector<Color> pixels;
    Canvas(size_t w, size_t h) : width(w), height(h), pixels(w * h, color::BLACK) {}
    const Color& pixel_at(size_t x, size_t y) const { return pixels[y * width + x]; }
    void write_pixel(size_t x, size_t y, const Color& c) { pixels[y * width + x] = c; }
    std::vector<uint8_t> to_image() const {  // canvas.rs:31-45, row-major RGB8
        std::vector<uint8_t> img(width * height * 3);
        for (size_t i = 0; i < pixels.size(); ++i) {
            img[3 * i + 0] = quantize_u8(pixels[i].red);
            img[3 * i + 1] = quantize_u8(pixels[i].green);
            img[3 * i + 2] = quantize_u8(pixels[i].blue);
        }
        return img;
    }
};

// ---------------------------------------------------------------------------------------------
// camera.rs:64-228
// ---------------------------------------------------------------------------------------------
struct CameraBuilder {
    size_t width = 0, height = 0;
    f64 field_of_view = 0;
    Transform transform{};
};

struct RenderOptions {
    JitterMode jitter = JitterMode::CENTER;
    uint32_t seed = 0;
    unsigned threads = 0;  // 0 => env RENDER_THREADS or 8 (camera.rs:16,167-170)
    size_t row_begin = 0, row_end = 0;  // row_end == 0 => all rows
    RayStats* stats = nullptr;
    // the `remaining` argument of World::color_at (world.rs:24); Camera::render passes RECURSION_DEPTH (camera.rs:193)
    uint8_t depth = RECURSION_DEPTH;
};

struct Camera {
    size_t hsize = 0, vsize = 0;
    f64 field_of_view = 0, pixel_size = 0, half_height = 0, half_width = 0;
    Transform transform{}, transform_inverse{};

    static Camera try_from(const CameraBuilder& b) {  // camera.rs:95-136
        if (approx(std::fmod(b.field_of_view, M_PI), 0.0)) throw Error("field of view angle cannot be straight");
        if (b.width == 0 || b.height == 0) throw Error("camera cannot have null dimensions");
        Camera c;
        c.hsize = b.width, c.vsize = b.height, c.field_of_view = b.field_of_view;
        f64 half_view = std::tan(b.field_of_view / 2.0);
        f64 aspect = static_cast<f64>(b.width) / static_cast<f64>(b.height);
        if (aspect < 1.0) {
            c.half_width = half_view * aspect;
            c.half_height = half_view;
        } else {
            c.half_width = half_view;
            c.half_height = half_view / aspect;
        }
        c.pixel_size = (c.half_width * 2.0) / static_cast<f64>(b.width);
        c.transform = b.transform;
        c.transform_inverse = b.transform.inverse();
        return c;
    }
    Ray ray_for_pixel(size_t x, size_t y) const {  // camera.rs:210-228
        f64 xoffset = (static_cast<f64>(x) + 0.5) * pixel_size;
        f64 yoffset = (static_cast<f64>(y) + 0.5) * pixel_size;
        f64 world_x = half_width - xoffset;
        f64 world_y = half_height - yoffset;
        Point pixel = transform_inverse * Point{world_x, world_y, -1.0};
        Point origin = transform_inverse * Point{0.0, 0.0, 0.0};
        Vector direction = (pixel - origin).normalize();
        return Ray{origin, direction};
    }
    Color color_for_pixel(const World& world, size_t x, size_t y, const RenderOptions& opt, TraceCtx& ctx) const {
        ctx.pixel_key = jitter_pixel_key(opt.seed, static_cast<uint32_t>(y * hsize + x));
        ctx.node = 0;
        return world.color_at(ray_for_pixel(x, y), opt.depth, ctx, true);
    }
    // camera.rs:163-208: one task per image row on a pool of RENDER_THREADS workers, each row is
    // buffered and written to the canvas under a mutex.  (The progress bar is not restated.)
    Canvas render(const World& world, const RenderOptions& opt = {}) const {
        Canvas image(hsize, vsize);
        std::mutex mutex;
        unsigned threads = opt.threads;
        if (threads == 0) {
            threads = 8;
            if (const char* e = std::getenv("RENDER_THREADS")) {
                char* end = nullptr;
                long v = std::strtol(e, &end, 10);
                if (end != e && *end == '\0' && v > 0) threads = static_cast<unsigned>(v);
            }
        }
        const size_t y0 = opt.row_begin, y1 = opt.row_end ? opt.row_end : vsize;
        std::atomic<size_t> next_row{y0};
        std::vector<RayStats> tstats(threads);
        auto worker = [&](unsigned tid) {
            std::mt19937 rng(opt.seed * 7919u + tid + 1u);
            TraceCtx ctx;
            ctx.mode = opt.jitter;
            ctx.rng = &rng;
            ctx.stats = opt.stats ? &tstats[tid] : nullptr;
            tls_stats = ctx.stats;
            for (;;) {
                size_t y = next_row.fetch_add(1);
                if (y >= y1) break;
                std::vector<std::pair<size_t, Color>> buffer;
                buffer.reserve(hsize);
                for (size_t x = 0; x < hsize; ++x) buffer.emplace_back(x, color_for_pixel(world, x, y, opt, ctx));
                std::lock_guard<std::mutex> lock(mutex);
                for (const auto& px : buffer) image.write_pixel(px.first, y, px.second);
            }
            tls_stats = nullptr;
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker, t);
        worker(0);
        for (auto& t : pool) t.join();
        if (opt.stats)
            for (const auto& s : tstats) opt.stats->add(s);
        return image;
    }
};

struct Scene {
    World world;
    Camera camera;
};

}  // namespace oracle
