/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the checker.
 *
 * orc_math.h: CPU restatement of the reference's affine algebra and spectrum arithmetic.
 *   Vector3/Point3/CoVector3/Matrix3   Affine.h:11-292   (double; one struct V3 stands for all three,
 *                                                         the arithmetic is identical component-wise)
 *   Spectrum                            RadiometryColor.h:16-69 (10 x float; scalars promoted through double)
 *   Complex                             Complex.h:9-35
 * Build with -ffp-contract=off: every expression below must round exactly like the reference's
 * (x86-64 SSE2, no fused multiply-add).
 */
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace orc {

static const double kPi    = 3.14159265358979323846264338327950288419716939937510;   /* RayTracer2020.h:13 */
static const double kSqrt3 = 1.73205080756887729352744634150587236694280525381038;   /* RayTracer2020.h:15 */
static const double kEps   = 2.0e-7;                                                 /* RayTracer2020.h:17 */
static const double kInf   = 1.0e+37;                                                /* RayTracer2020.h:18 */
enum { NSPEC = 10 };

struct V3 {
    double x, y, z;
    V3() : x(0.0), y(0.0), z(0.0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double  at(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
/* Affine.h:21-25, 65-67, 249-253 */
static inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(const V3 &a)              { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(const V3 &a, double f)    { return V3(a.x * f, a.y * f, a.z * f); }
static inline V3 operator/(const V3 &a, double f)    { return V3(a.x / f, a.y / f, a.z / f); }
/* Affine.h:32-35, 255-256, 263-266 */
static inline double dot(const V3 &u, const V3 &v)   { return u.x * v.x + u.y * v.y + u.z * v.z; }
/* Affine.h:38-41 */
static inline V3 cross(const V3 &u, const V3 &v)     { return V3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
/* Affine.h:44-53 */
static inline double norm(const V3 &v)               { return sqrt(dot(v, v)); }
static inline V3 normalize(const V3 &v)              { return v / norm(v); }

struct M3 {
    double m[3][3];
};
/* Affine.h:191-205: v' = M*v */
static inline V3 mul(const M3 &a, const V3 &v)
{
    return V3(v.x * a.m[0][0] + v.y * a.m[0][1] + v.z * a.m[0][2],
              v.x * a.m[1][0] + v.y * a.m[1][1] + v.z * a.m[1][2],
              v.x * a.m[2][0] + v.y * a.m[2][1] + v.z * a.m[2][2]);
}
/* Affine.h:286-292: covector row times matrix */
static inline V3 mul(const V3 &c, const M3 &a)
{
    return V3(c.x * a.m[0][0] + c.y * a.m[1][0] + c.z * a.m[2][0],
              c.x * a.m[0][1] + c.y * a.m[1][1] + c.z * a.m[2][1],
              c.x * a.m[0][2] + c.y * a.m[1][2] + c.z * a.m[2][2]);
}
/* Affine.h:173-189 */
static inline M3 mul(const M3 &a, const M3 &b)
{
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}

struct Spec {
    float v[NSPEC];
};
/* RadiometryColor.h:19-47 */
static inline Spec operator+(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NSPEC; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline Spec operator*(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NSPEC; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
static inline Spec operator*(const Spec &a, double f)      { Spec r; for (int i = 0; i < NSPEC; i++) r.v[i] = float(a.v[i] * f); return r; }
/* RadiometryColor.h:53-58 */
static inline Spec operator/(const Spec &a, double f)      { Spec r; if (f) f = 1.0 / f; for (int i = 0; i < NSPEC; i++) r.v[i] = float(a.v[i] * f); return r; }
/* RadiometryColor.h:59 */
static inline bool operator==(const Spec &a, const Spec &b) { for (int i = 0; i < NSPEC; i++) if (a.v[i] != b.v[i]) return false; return true; }
static inline Spec spec_const(float f) { Spec r; for (int i = 0; i < NSPEC; i++) r.v[i] = f; return r; }

struct Rgb { float r, g, b; };

/* Complex.h:9-35 */
struct Cx {
    double re, im;
    Cx() : re(0), im(0) {}
    Cx(double a, double b) : re(a), im(b) {}
    double abs2() const { return re * re + im * im; }
};
static inline Cx operator+(Cx a, Cx b) { return Cx(a.re + b.re, a.im + b.im); }
static inline Cx operator-(Cx a, Cx b) { return Cx(a.re - b.re, a.im - b.im); }
static inline Cx operator*(Cx a, double f) { return Cx(a.re * f, a.im * f); }
static inline Cx operator/(Cx a, Cx c)
{
    double den, q;
    if (fabs(c.re) < fabs(c.im)) {
        q = c.re / c.im;
        den = c.re * q + c.im;
        return Cx((a.re * q + a.im) / den, (a.im * q - a.re) / den);
    }
    q = c.im / c.re;
    den = c.im * q + c.re;
    return Cx((a.im * q + a.re) / den, (a.im - a.re * q) / den);
}

} // namespace orc
