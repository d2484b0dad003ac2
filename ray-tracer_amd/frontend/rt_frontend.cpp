/*
 * rt_frontend.cpp -- implementation of the host scene API (rt_frontend.h): scene construction
 * helpers, Optimize(), GetExtent(), flattening to RtSceneDesc, and RayTracer::Render, which calls
 * the HIP library through the C ABI (include/rt_amd.h).  No ray is ever intersected here.
 */
#include <stdlib.h>
#include <dlfcn.h>
#include <map>
#include <string>
#include "rt_frontend.h"
#include "../../include/rt_amd.h"
#include "../csrc/rt_tables.h"

/* ------------------------------------------------------------------------------ constants ---- */
static Spectrum table_spectrum(const char *name)
{
    Spectrum s;
    int dt = 0, cnt = 0;
    const void *p = rt_table_find(name, &dt, &cnt);
    if (!p || dt != RT_TABLE_F32 || cnt != NSPEC) { fprintf(stderr, "rt_frontend: missing table %s\n", name); abort(); }
    memcpy(s.rgf, p, sizeof s.rgf);
    return s;
}
static DisplayRGB table_rgb(const char *name)
{
    int dt = 0, cnt = 0;
    const float *p = (const float *)rt_table_find(name, &dt, &cnt);
    if (!p || dt != RT_TABLE_F32 || cnt != 3) { fprintf(stderr, "rt_frontend: missing table %s\n", name); abort(); }
    DisplayRGB c; c.red = p[0]; c.grn = p[1]; c.blu = p[2];
    return c;
}
#define SPEC(n) Spectrum n = table_spectrum(#n)
SPEC(g_sRed); SPEC(g_sGreen); SPEC(g_sBlue); SPEC(g_sYellow); SPEC(g_sCyan); SPEC(g_sMagenta); SPEC(g_sWhite); SPEC(g_sOrange);
SPEC(g_sBlack); SPEC(g_sEqualWhite); SPEC(g_sD6500);
SPEC(g_sWater); SPEC(g_sCobaltGlass); SPEC(g_sCopper); SPEC(g_sGold); SPEC(g_sRuby); SPEC(g_sEmerald); SPEC(g_sSapphire); SPEC(g_sWhiskey);
SPEC(g_sEarthSky); SPEC(g_sSolar); SPEC(g_sIllumD65);
#define RGBC(n) DisplayRGB n = table_rgb(#n)
RGBC(g_rgbIron); RGBC(g_rgbCopper); RGBC(g_rgbGold); RGBC(g_rgbZinc); RGBC(g_rgbZilver); RGBC(g_rgbAluminum); RGBC(g_rgbSilicon);
DisplayRGB SpectrumToRGB[NSPEC];
Vector2 g_rgvBlueNoise[BLUE_N];
int g_BlueNoiseQuasiIndex2[BLUE_N], g_BlueNoiseQuasiIndex3[BLUE_N];
double g_uCamera, g_vCamera, g_uLight, g_vLight, g_uMicroFacet, g_vMicroFacet;

namespace {
struct TableInit {
    TableInit()
    {
        int dt, cnt;
        const float *m = (const float *)rt_table_find("SpectrumToRGB", &dt, &cnt);
        for (int i = 0; i < NSPEC; i++) { SpectrumToRGB[i].red = m[3 * i]; SpectrumToRGB[i].grn = m[3 * i + 1]; SpectrumToRGB[i].blu = m[3 * i + 2]; }
        const double *b = (const double *)rt_table_find("g_rgvBlueNoise", &dt, &cnt);
        for (int i = 0; i < BLUE_N; i++) g_rgvBlueNoise[i] = Vector2(b[2 * i], b[2 * i + 1]);
        /* the reference ships these two permutations precomputed (BlueNoiseTile.cpp:1607, :1810) */
        BuildQuasiIndex(g_BlueNoiseQuasiIndex2, 2, BLUE_N);
        BuildQuasiIndex(g_BlueNoiseQuasiIndex3, 3, BLUE_N);
    }
} s_table_init;
}

const Spectrum *rt_spectrum_by_name(const char *name)
{
    static std::map<std::string, Spectrum> cache;
    std::map<std::string, Spectrum>::iterator it = cache.find(name);
    if (it != cache.end()) return &it->second;
    int dt, cnt;
    if (!rt_table_find(name, &dt, &cnt) || dt != RT_TABLE_F32 || cnt != NSPEC) return 0;
    cache[name] = table_spectrum(name);
    return &cache[name];
}

DisplayRGB Spectrum::sRGB() const
{
    DisplayRGB c(0.0, 0.0, 0.0);
    for (int i = 0; i < NSPEC; i++) {
        double w = rgf[i] * (5.0 / 10.0);
        c.red = c.red + float(SpectrumToRGB[i].red * w);
        c.grn = c.grn + float(SpectrumToRGB[i].grn * w);
        c.blu = c.blu + float(SpectrumToRGB[i].blu * w);
    }
    return c;
}

/* RadiometryColor.cpp:947-969 */
Spectrum RGBtoSpectrum(DisplayRGB c, int n6500)
{
    Spectrum s;
    if (c.red < c.grn && c.red < c.blu)
        s = g_sWhite * c.red + g_sGreen * (c.grn - c.red) + g_sBlue * (c.blu - c.red);
    else if (c.grn < c.red && c.grn < c.blu)
        s = g_sWhite * c.grn + g_sRed * (c.red - c.grn) + g_sBlue * (c.blu - c.grn);
    else
        s = g_sWhite * c.blu + g_sRed * (c.red - c.blu) + g_sGreen * (c.grn - c.blu);
    if (n6500) s = s / g_sD6500;
    return s;
}
Spectrum RGBtoSpectrum(double f) { return g_sWhite * f; }

/* RadiometryColor.cpp:985-1012 */
Spectrum BlackBody(double kelvin)
{
    if (kelvin < 0.0) return g_sBlack;
    const double h = 6.62606876e-34, c = 2.99792458e+8, kB = 1.3806503e-23;
    double c1 = 2.0 * D_PI * h * c * c, c2 = h * c / kB;
    Spectrum sd;
    for (int i = 0; i < NSPEC; i++) {
        double lambda = 1.0e-9 * (380 + 40 * double(i));
        double x = c2 / lambda / kelvin, em1;
        if (x < -0.0001 || x > 0.0001) em1 = exp(x) - 1.0;
        else em1 = x * (1.0 + 0.5 * x * (1.0 + x / 3.0 * (1.0 + 0.25 * x)));
        double planck = c1 / (lambda * lambda * lambda * lambda * lambda * em1);
        sd.rgf[i] = float(planck / 30932027310080.0) / 12.0f;
    }
    return sd * 1.98;
}

/* ---------------------------------------------------------------------------------- PRNG ---- */
unsigned ML_nMarsagliaX = 886459, ML_nMarsagliaC = 361290869;
unsigned RandomUnsigned()
{
    uint64_t n = (uint64_t)1965537969u * ML_nMarsagliaX + ML_nMarsagliaC;
    ML_nMarsagliaX = (unsigned)(n & 0xFFFFFFFFu);
    ML_nMarsagliaC = (unsigned)(n >> 32);
    return ML_nMarsagliaX;
}
double RandomDouble()
{
    double hi = double(RandomUnsigned());       /* first draw is the high word (left operand first) */
    double lo = double(RandomUnsigned());
    return (hi + lo * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0);
}
float RandomFloat() { return float(RandomUnsigned()) * (1.0f / 4294967296.0f); }
unsigned RandomLessThanN(unsigned mod)
{
    unsigned rem = 0xFFFFFFFFu % mod, n;
    do { n = RandomUnsigned(); } while (n <= rem);
    return n % mod;
}
void ML_InitializeRandom(unsigned x, unsigned c) { ML_nMarsagliaX = x; ML_nMarsagliaC = c; }

/* Radical-inverse permutation of 0..n-1 in base `prime` (Patterns.cpp:1697-1736): element k of the
   result is the k-th index when indices are ordered by their digit-reversed value; ties keep index
   order.  Computed directly from the reversed digit string instead of the reference's recursive
   bucket sort. */
void BuildQuasiIndex(int rgn[], int prime, int n)
{
    rt_build_quasi_index(rgn, prime, n);
}

void FibonacciSphere(double rgf[], int i, int n)       /* Patterns.cpp:1850-1865 */
{
    double u = fmod(double(i) / D_PHI, 1.0), v = double(i) / double(n);
    double z = 2.0 * v - 1.0, r = sqrt(1.0 - z * z);
    rgf[0] = r * sin(2.0 * D_PI * u);
    rgf[1] = r * cos(2.0 * D_PI * u);
    rgf[2] = z;
}

/* ------------------------------------------------------------------------------ matrices ---- */
static void basis_rotate(int i1, int i2, double rad, Matrix3 &mat, bool initial)   /* Affine.cpp:36-65 */
{
    double s = sin(rad), c = cos(rad);
    if (initial) {
        mat.m[i1][i1] = c; mat.m[i1][i2] = -s; mat.m[i2][i1] = s; mat.m[i2][i2] = c;
        return;
    }
    for (int j = 0; j < 3; j++) {
        double x = c * mat.m[i1][j] - s * mat.m[i2][j];
        mat.m[i2][j] = s * mat.m[i1][j] + c * mat.m[i2][j];
        mat.m[i1][j] = x;
    }
}
Matrix3 EulerMatrix(double r0, double r1, double r2, const char *cv)              /* Affine.cpp:67-97 */
{
    Matrix3 mat(1.0);
    const double r[3] = { r0, r1, r2 };
    for (int k = 0; k < 3; k++) {
        int a, b;
        switch (cv[k]) {
        case 'x': a = 1; b = 2; break;
        case 'y': a = 2; b = 0; break;
        case 'z': a = 0; b = 1; break;
        default: continue;
        }
        basis_rotate(a, b, r[k], mat, k == 0);
    }
    return mat;
}
static Matrix3 axis_angle(const Vector3 &n, double s, double c)
{
    Matrix3 mat;
    double t = 1.0 - c;
    mat.m[0][0] = t * n.x * n.x + c;
    mat.m[1][1] = t * n.y * n.y + c;
    mat.m[2][2] = t * n.z * n.z + c;
    mat.m[0][1] = t * n.x * n.y + s * n.z;
    mat.m[1][0] = t * n.x * n.y - s * n.z;
    mat.m[0][2] = t * n.x * n.z - s * n.y;
    mat.m[2][0] = t * n.x * n.z + s * n.y;
    mat.m[1][2] = t * n.y * n.z + s * n.x;
    mat.m[2][1] = t * n.y * n.z - s * n.x;
    return mat;
}
Matrix3 RotateAboutAxis(const Vector3 &axis, double rad)                          /* Affine.cpp:99-120 */
{
    return axis_angle(Normalize(axis), sin(-rad), cos(-rad));
}
Matrix3 RotateFromTo(const Vector3 &from, const Vector3 &to)                      /* Affine.cpp:122-153 */
{
    Vector3 a = Normalize(from), b = Normalize(to), n = Cross(a, b);
    double s = Norm(n), c = Dot(a, b);
    if (s < 1.0e-8) {
        if (c > 0.0) return Matrix3(1.0);
        return RotateAboutAxis(Vector3(1.0, 0.0, 0.0), D_PI);
    }
    n = n / s;
    return axis_angle(n, -s, c);
}
Matrix3 InverseMatrix(const Matrix3 &a)                                           /* Affine.cpp:6-31 */
{
    double f1 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    double f2 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    double f3 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    double det = a.m[0][0] * f1 + a.m[0][1] * f2 + a.m[0][2] * f3;
    double di = det ? (1.0 / det) : 0.0;
    return Matrix3(f1 * di, -(a.m[0][1] * a.m[2][2] - a.m[0][2] * a.m[2][1]) * di, (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * di,
                   f2 * di, (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * di, -(a.m[0][0] * a.m[1][2] - a.m[0][2] * a.m[1][0]) * di,
                   f3 * di, -(a.m[0][0] * a.m[2][1] - a.m[0][1] * a.m[2][0]) * di, (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * di);
}

/* -------------------------------------------------------------------------------- solids ---- */
static int s_serial = 0;
Solid::Solid() : nIsAffine(0), nSerial(s_serial++) {}
void rt_frontend_reset_ids() { s_serial = 0; }

#define T_  1.0 / D_SQRT3
#define P1_ D_PHI / D_SQRT3
#define P2_ (D_PHI - 1.0) / D_SQRT3
#define I1_ 0.52573111211913
#define I2_ 0.85065080835204
CoVector3 rgvTetrahedron[4] = { CoVector3(T_, T_, T_), CoVector3(T_, -T_, -T_), CoVector3(-T_, T_, -T_), CoVector3(-T_, -T_, T_) };
CoVector3 rgvCube[8] = { CoVector3(T_, T_, T_), CoVector3(T_, T_, -T_), CoVector3(T_, -T_, T_), CoVector3(T_, -T_, -T_),
                         CoVector3(-T_, T_, T_), CoVector3(-T_, T_, -T_), CoVector3(-T_, -T_, T_), CoVector3(-T_, -T_, -T_) };
CoVector3 rgvOctahedron[6] = { CoVector3(1.0, 0.0, 0.0), CoVector3(-1.0, 0.0, 0.0), CoVector3(0.0, 1.0, 0.0),
                               CoVector3(0.0, -1.0, 0.0), CoVector3(0.0, 0.0, 1.0), CoVector3(0.0, 0.0, -1.0) };
CoVector3 rgvDodecahedron[20] = {
    CoVector3(T_, T_, T_), CoVector3(T_, T_, -T_), CoVector3(T_, -T_, T_), CoVector3(T_, -T_, -T_),
    CoVector3(-T_, T_, T_), CoVector3(-T_, T_, -T_), CoVector3(-T_, -T_, T_), CoVector3(-T_, -T_, -T_),
    CoVector3(0.0, P2_, P1_), CoVector3(0.0, P2_, -P1_), CoVector3(0.0, -P2_, P1_), CoVector3(0.0, -P2_, -P1_),
    CoVector3(P1_, 0.0, P2_), CoVector3(P1_, 0.0, -P2_), CoVector3(-P1_, 0.0, P2_), CoVector3(-P1_, 0.0, -P2_),
    CoVector3(P2_, P1_, 0.0), CoVector3(P2_, -P1_, 0.0), CoVector3(-P2_, P1_, 0.0), CoVector3(-P2_, -P1_, 0.0) };
CoVector3 rgvIcosahedron[12] = {
    CoVector3(I1_, 0.0, I2_), CoVector3(I1_, 0.0, -I2_), CoVector3(-I1_, 0.0, I2_), CoVector3(-I1_, 0.0, -I2_),
    CoVector3(0.0, I2_, I1_), CoVector3(0.0, I2_, -I1_), CoVector3(0.0, -I2_, I1_), CoVector3(0.0, -I2_, -I1_),
    CoVector3(I2_, I1_, 0.0), CoVector3(-I2_, I1_, 0.0), CoVector3(I2_, -I1_, 0.0), CoVector3(-I2_, -I1_, 0.0) };

Prism::Prism(int n)                                     /* Solids.cpp:871-881 */
{
    nSides = n + 2;
    rgv = new CoVector3[n + 2];
    int i;
    for (i = 0; i < n; i++) rgv[i] = CoVector3(sin(2.0 * i * D_PI / n), cos(2.0 * i * D_PI / n), 0.0);
    rgv[i++] = CoVector3(0.0, 0.0, 1.0);
    rgv[i++] = CoVector3(0.0, 0.0, -1.0);
}
Fibonacci::Fibonacci(int n)                             /* Solids.cpp:883-892 */
{
    nSides = n;
    rgv = new CoVector3[n];
    for (int i = 0; i < n; i++) { double f[3]; FibonacciSphere(f, i, n); rgv[i] = CoVector3(f[0], f[1], f[2]); }
}

Extent ConeHyperCylinder::GetExtent()                   /* Solids.cpp:360-371 */
{
    double r = Max(1.0, R), z;
    if (R <= 1.0) z = 1.0; else z = sqrt(R * R / (R * R - 1.0));
    return Extent(Point3(-r, -r, -z), Point3(r, r, +z));
}

Extent AffineSolid::GetExtent()                         /* Solids.cpp:922-947 */
{
    if (psChild == 0) return Extent(0.0);
    Extent e = psChild->GetExtent();
    Point3 c[2] = { e.pMin, e.pMax };
    for (int i = 0; i < 8; i++) {
        Point3 p(c[(i >> 0) & 1].x, c[(i >> 1) & 1].y, c[(i >> 2) & 1].z);
        Point3 q = mTransform * p + vTranslate;
        if (i == 0) e.pMin = e.pMax = q;
        else { e.pMin = Min(e.pMin, q); e.pMax = Max(e.pMax, q); }
    }
    return e;
}
Extent Union::GetExtent()                               /* Solids.cpp:1077-1097 */
{
    if (psLeft && psRight) { Extent l = psLeft->GetExtent(), r = psRight->GetExtent(); return Extent(Min(l.pMin, r.pMin), Max(l.pMax, r.pMax)); }
    if (psLeft) return psLeft->GetExtent();
    if (psRight) return psRight->GetExtent();
    return Extent(0.0);
}
Extent Intersection::GetExtent()                        /* Solids.cpp:1160-1175 */
{
    if (psLeft && psRight) { Extent l = psLeft->GetExtent(), r = psRight->GetExtent(); return Extent(Max(l.pMin, r.pMin), Min(l.pMax, r.pMax)); }
    return Extent(0.0);
}
Extent Difference::GetExtent() { return psLeft ? psLeft->GetExtent() : Extent(0.0); }     /* Solids.cpp:1198-1204 */

AngularClip::AngularClip(double f1, double f2, Solid *p) : ps(p)        /* Solids.cpp:1206-1222 */
{
    if (f2 - f1 <= D_PI) { nIntersect = 1; fAng1 = f1; fAng2 = f2; }
    else { nIntersect = 0; fAng1 = f2; fAng2 = f1 + 2.0 * D_PI; }
    cv1 = CoVector3(-cos(fAng1), +sin(fAng1), 0.0);
    cv2 = CoVector3(+cos(fAng2), -sin(fAng2), 0.0);
}

/* Optimize(): merge affine chains and thread the parent links the shader walks to transform
   normals (Solids.cpp:957-983, 1489-1528). */
static AffineSolid *s_enclosing_affine = 0;
Solid *AffineSolid::Optimize()
{
    if (psChild == 0) return 0;
    while (psChild->nIsAffine) {
        AffineSolid *c = (AffineSolid *)psChild;
        Matrix3 m = mTransform * c->mTransform, mi = c->mInverse * mInverse;
        Vector3 v = vTranslate + mTransform * c->vTranslate;
        mTransform = m; mInverse = mi; vTranslate = v;
        psChild = c->psChild;
    }
    paParent = s_enclosing_affine;
    AffineSolid *saved = s_enclosing_affine;
    s_enclosing_affine = this;
    psChild = psChild->Optimize();
    s_enclosing_affine = saved;
    return this;
}
Solid *Material::Optimize() { paAffineTransforms = s_enclosing_affine; psChild = psChild->Optimize(); return this; }
Solid *Surface::Optimize() { paAffineTransforms = s_enclosing_affine; psChild = psChild->Optimize(); return this; }

static const int kLeftCount[13] = { 0, 0, 1, 1, 2, 3, 3, 3, 4, 4, 5, 5, 6 };
Solid *rt_make_union(Solid **ps, int n, int top)
{
    if (n <= 0) return 0;
    if (n == 1) return ps[0];
    if (n <= 5 && ps[0] == 0) return rt_make_union(ps + 1, n - 1, top);
    int l = n <= 12 ? kLeftCount[n] : n / 2;
    Solid *a = rt_make_union(ps, l, top), *b = rt_make_union(ps + l, n - l, top);
    return top ? (Solid *)new TopUnion(a, b) : (Solid *)new Union(a, b);
}
Solid *rt_make_intersection(Solid **ps, int n)
{
    if (n <= 0) return 0;
    if (n == 1) return ps[0];
    int l = n <= 12 ? kLeftCount[n] : n / 2;
    return new Intersection(rt_make_intersection(ps, l), rt_make_intersection(ps + l, n - l));
}

/* ---- shape library: Solids.cpp:1624-2009 ---- */
static Solid *four_cylinders(double a, double b, double h, double r)
{
    Solid *c = new Translate(+a, +b, 0.0, new Scale(r, r, h, new Cylinder));
    c = new Union(new Translate(+a, -b, 0.0, new Scale(r, r, h, new Cylinder)), c);
    c = new Union(new Translate(-a, +b, 0.0, new Scale(r, r, h, new Cylinder)), c);
    c = new Union(new Translate(-a, -b, 0.0, new Scale(r, r, h, new Cylinder)), c);
    return c;
}
Solid *Cuboid(double r, double x, double y, double z)
{
    Solid *cubes = new Scale(x - r, y - r, z, new Cube);
    cubes = new Union(new Scale(x - r, y, z - r, new Cube), cubes);
    cubes = new Union(new Scale(x, y - r, z - r, new Cube), cubes);
    Solid *ps = new Union(four_cylinders(x - r, y - r, z - r, r), cubes);
    ps = new Union(new Rotate(0.5 * D_PI, 0.0, 0.0, four_cylinders(x - r, z - r, y - r, r)), ps);
    ps = new Union(new Rotate(0.0, 0.5 * D_PI, 0.0, four_cylinders(z - r, y - r, x - r, r)), ps);
    Solid *sph = 0;
    for (int k = 0; k < 8; k++) {
        double sx = (k & 4) ? -1.0 : 1.0, sy = (k & 2) ? -1.0 : 1.0, sz = (k & 1) ? -1.0 : 1.0;
        Solid *one = new Translate(sx * (x - r), sy * (y - r), sz * (z - r), new Scale(r, r, r, new Sphere));
        sph = sph ? (Solid *)new Union(one, sph) : one;
    }
    return new BoundingBox(new Union(sph, ps));
}
Solid *CylinderFromTo(const Point3 &from, const Point3 &to, double r)
{
    double L = Norm(to - from);
    if (L == 0.0) printf("Zero length cylinder\n");
    Solid *ps = new Scale(r, r, 0.5 * L, new Translate(0.0, 0.0, -1.0, new Cylinder));
    ps = new Rotate(Vector3(0.0, 0.0, -1.0), to - from, ps);
    return new Translate(from, ps);
}
static Point3 as_point(const CoVector3 &c) { return Point3(c.x, c.y, c.z); }
static Solid *cage(const CoVector3 *v, int nv, double maxlen, double r, double rs, bool first_vertex_from_tetra)
{
    Solid *ps = 0;
    if (r) {
        for (int i = 0; i < nv; i++)
            for (int j = i + 1; j < nv; j++) {
                Point3 p1 = as_point(v[i]), p2 = as_point(v[j]);
                if (maxlen > 0.0 && Norm(p1 - p2) > maxlen) continue;
                Solid *c = CylinderFromTo(p1, p2, r);
                ps = ps ? (Solid *)new Union(ps, c) : c;
            }
    }
    if (rs) {
        for (int i = 0; i < nv; i++) {
            if (ps) ps = new Union(ps, new Translate(as_point(v[i]), new Scale(rs, rs, rs, new Sphere)));
            else ps = new Translate(as_point(first_vertex_from_tetra ? rgvTetrahedron[i] : v[i]), new Scale(rs, rs, rs, new Sphere));
        }
    }
    return new BoundingBox(ps);
}
/* Solids.cpp:801-857.  Corner k of the extent takes x, y, z from pMin / pMax by bits 0, 1, 2 of k.  For the plane, the corners
   with cv * p - fFromOrigin < 0 are kept (in their order) and every kept corner is joined to every dropped one; where such a
   segment meets the plane is one more point (kept corner outer, dropped corner inner).  The box of all those points is the
   answer -- the initial extent again when fewer than two points are left.  Only rgcv[0] is looked at: the loop of the reference
   ends in an unconditional break (:845). */
Extent ClipPlanes(const Extent &eInitial, const CoVector3 rgcv[], int nv, double fFromOrigin)
{
    std::vector<Point3> in, out;
    const Point3 *c2[2] = { &eInitial.pMin, &eInitial.pMax };
    for (int k = 0; k < 8; k++) in.push_back(Point3(c2[k & 1]->x, c2[(k >> 1) & 1]->y, c2[(k >> 2) & 1]->z));
    if (nv > 0) {
        const CoVector3 &cv = rgcv[0];
        std::vector<Point3> kept;
        for (size_t k = 0; k < in.size(); k++) ((cv * in[k] - fFromOrigin < 0.0) ? kept : out).push_back(in[k]);
        in = kept;
        for (size_t i = 0; i < kept.size(); i++)
            for (size_t j = 0; j < out.size(); j++) {
                const double t = (fFromOrigin - cv * kept[i]) / (cv * (out[j] - kept[i]));
                in.push_back(kept[i] + (out[j] - kept[i]) * t);
            }
    }
    if (in.size() < 2) return eInitial;
    Extent e(in[0], in[0]);
    for (size_t i = 1; i < in.size(); i++) { e.pMin = Min(e.pMin, in[i]); e.pMax = Max(e.pMax, in[i]); }
    return e;
}

Solid *TetrahedronCage(double r, double rs) { return cage(rgvTetrahedron, 4, 0.0, r, rs, false); }
Solid *IcosahedronCage(double r, double rs) { return cage(rgvIcosahedron, 12, 1.1, r, rs, true); }     /* Solids.cpp:1766 uses Tetra(i) when r == 0 */
Solid *DodecahedronCage(double r, double rs) { return cage(rgvDodecahedron, 20, 0.8, r, rs, false); }
Solid *HexahedronCage(double r, double rs) { return cage(rgvCube, 8, 1.2, r, rs, false); }
Solid *OctahedronCage(double r, double rs) { return cage(rgvOctahedron, 6, 1.9, r, rs, false); }
/* Solids.cpp:1880-1917: turns of a helix out of half tori, tilted alternately by asin(zRepeat / 4) */
Solid *PseudoHelix(double r, double zRepeat, double fAng)
{
    const double tilt = asin(zRepeat / 4.0), c = sqrt(1.0 - zRepeat * zRepeat / 16.0), e = 0.00001;
    Solid *ps = 0;
    for (int n = 0; fAng >= 0.0 * D_PI; fAng -= 2.0 * D_PI, n++) {
        Solid *a, *b = 0;
        if (fAng < D_PI) a = new AngularClip(0.0 - e, fAng + e, new Torus(r));
        else {
            a = new AngularClip(0.0 - e, D_PI + e, new Torus(r));
            b = new AngularClip(D_PI - e, (fAng < 2.0 * D_PI ? fAng : 2.0 * D_PI) + e, new Torus(r));
            b = new Translate(0.0, -c, 0.0, new Rotate(-tilt, 0.0, 0.0, new Translate(0.0, 1.0, 0.0, b)));
        }
        a = new Translate(0.0, -c, 0.0, new Rotate(+tilt, 0.0, 0.0, new Translate(0.0, 1.0, 0.0, a)));
        if (b) a = new Union(a, b);
        a = new BoundingBox(a);
        ps = n == 0 ? a : (Solid *)new Union(ps, new Translate(0.0, 0.0, -zRepeat * n, a));
    }
    return new BoundingBox(new Translate(0.0, 0.0, -zRepeat / 2.0, new Scale(1.0, 1.0 / c, 1.0, ps)));
}
Solid *Diskoid(double r) { return new BoundingBox(new Union(new Torus(r), new Scale(1.0, 1.0, r, new Cylinder))); }
Solid *Rodoid(double r)
{
    return new BoundingBox(MakeUnion(new Scale(r, r, 1.0, new Cylinder),
                                     new Translate(0.0, 0.0, +1.0, new Scale(r, r, r, new Sphere)),
                                     new Translate(0.0, 0.0, -1.0, new Scale(r, r, r, new Sphere))));
}
Solid *Squareoid(double r)
{
    double f90 = 90.0 * D_PI / 180.0;
    return new BoundingBox(MakeUnion(
        new Scale(1.0, 1.0, r, new Cube),
        new Translate(+1.0, +1.0, 0.0, new Scale(r, r, r, new Sphere)),
        new Translate(+1.0, -1.0, 0.0, new Scale(r, r, r, new Sphere)),
        new Translate(-1.0, +1.0, 0.0, new Scale(r, r, r, new Sphere)),
        new Translate(-1.0, -1.0, 0.0, new Scale(r, r, r, new Sphere)),
        new Translate(+1.0, 0.0, 0.0, new Rotate(f90, 0.0, 0.0, new Scale(r, r, 1.0, new Cylinder))),
        new Translate(-1.0, 0.0, 0.0, new Rotate(f90, 0.0, 0.0, new Scale(r, r, 1.0, new Cylinder))),
        new Translate(0.0, +1.0, 0.0, new Rotate(0.0, f90, 0.0, new Scale(r, r, 1.0, new Cylinder))),
        new Translate(0.0, -1.0, 0.0, new Rotate(0.0, f90, 0.0, new Scale(r, r, 1.0, new Cylinder)))));
}
static void knob_geometry(double r, double z, double &rs, double &zDelta)
{
    double fTan = (1.0 - SSS(r)) / z;
    double fCos = 1.0 / sqrt(1.0 + fTan * fTan);
    rs = 1.0 / fCos;
    zDelta = z + fTan;
}
Solid *Knoboid(double r, double z)
{
    double rs, zd;
    knob_geometry(r, z, rs, zd);
    return new BoundingBox(new Union(new Translate(0.0, 0.0, zd, new Scale(rs, rs, rs, new Sphere)),
        new Scale(1.0, 1.0, z, new Intersection(new Scale(1.0, 1.0, 1.0, new Slab), new ConeHyperCylinder(r)))));
}
Solid *Knoboid2(double r, double z)
{
    double rs, zd;
    knob_geometry(r, z, rs, zd);
    return new BoundingBox(new Union(new Scale(rs, rs, rs, new Sphere),
        new Translate(0.0, 0.0, -zd, new Scale(1.0, 1.0, z, new Intersection(new Scale(1.0, 1.0, 1.0, new Slab), new ConeHyperCylinder(r))))));
}
Solid *Hyperboloidoid(double r, double rt, double z)
{
    double fTan = (1.0 - r * r) / z;
    double fCos = 1.0 / sqrt(1.0 + fTan * fTan);
    double fSin = fTan * fCos;
    fCos *= rt; fSin *= rt;
    double zd = z + fSin, fS = 1.0 - fCos;
    return new BoundingBox(new Union(new Translate(0.0, 0.0, zd, new Scale(fS, fS, fS, Diskoid(rt / fS))),
        new Scale(1.0, 1.0, z, new Intersection(new Cube, new ConeHyperCylinder(r)))));
}
Solid *Slaboid(double x, double y, double z, double r)
{
    return new BoundingBox(MakeUnion(
        new Scale(x - r, y, z, new Cube), new Scale(x, y - r, z, new Cube),
        new Translate(+x - r, +y - r, 0.0, new Scale(r, r, z, new Cylinder)),
        new Translate(+x - r, -y + r, 0.0, new Scale(r, r, z, new Cylinder)),
        new Translate(-x + r, +y - r, 0.0, new Scale(r, r, z, new Cylinder)),
        new Translate(-x + r, -y + r, 0.0, new Scale(r, r, z, new Cylinder))));
}
/* Solids.cpp:2011-2049: a box with two rounding radii -- r for the four upright edges, r2 for the rims of the two faces.
   Three groups in the reference's child order: plates and corner posts, eight corner rounds, eight rim rods. */
Solid *Slaboidoid(double x, double y, double z, double r, double r2)
{
    const double f90 = 90.0 * D_PI / 180.0, rc = r - r2;
    static const int sgn[4][2] = { { +1, +1 }, { +1, -1 }, { -1, +1 }, { -1, -1 } };
    Solid *core[8], *corner[8], *rim[8];
    core[0] = new Scale(x - r, y, z - r2, new Cube);
    core[1] = new Scale(x, y - r, z - r2, new Cube);
    core[2] = new Scale(x - r, y - r2, z, new Cube);
    core[3] = new Scale(x - r2, y - r, z, new Cube);
    for (int k = 0; k < 4; k++)
        core[4 + k] = new Translate(sgn[k][0] * (x - r), sgn[k][1] * (y - r), 0.0, new Scale(r, r, z - r2, new Cylinder));
    for (int k = 0; k < 8; k++) {
        double sx = (k & 4) ? -1.0 : 1.0, sy = (k & 2) ? -1.0 : 1.0, sz = (k & 1) ? -1.0 : 1.0;
        corner[k] = new Translate(sx * (x - r), sy * (y - r), sz * (z - r2), new Scale(rc, rc, rc, Diskoid(r2 / rc)));
    }
    for (int k = 0; k < 4; k++) {
        rim[k] = new Translate(sgn[k][0] * (x - r2), 0.0, sgn[k][1] * (z - r2), new Scale(r2, y - r, r2, new Rotate(f90, 0.0, 0.0, new Cylinder)));
        rim[4 + k] = new Translate(0.0, sgn[k][0] * (y - r2), sgn[k][1] * (z - r2), new Scale(x - r, r2, r2, new Rotate(0.0, f90, 0.0, new Cylinder)));
    }
    return new BoundingBox(MakeUnion(rt_make_union(core, 8, 0), rt_make_union(corner, 8, 0), rt_make_union(rim, 8, 0)));
}

/* Solids.cpp:2173-2230: text out of cylinders and spheres, one boxed union per glyph, 12 font units per character cell and
   per unit of length.  The glyphs (Solids.cpp:2062-2171) are data: table "rgFont" of tables.bin, 8 codes per glyph --
   x in the high nibble, y in the low one, 0xFE lifts the pen, 0xFF ends the glyph.  Lower case is drawn as small capitals. */
Solid *SolidText(double xStart, double yStart, const char *sz, int nJust, double r)
{
    int dt = 0, cnt = 0;
    const int *font = (const int *)rt_table_find("rgFont", &dt, &cnt);
    if (!font || dt != RT_TABLE_I32 || cnt % 8) { fprintf(stderr, "rt_frontend: missing table rgFont\n"); abort(); }
    const int nGlyphs = cnt / 8;
    double xJust = 0.0;
    if (nJust == JUSTIFIED_RIGHT) xJust = -12.0 * strlen(sz);
    else if (nJust != JUSTIFIED_LEFT) xJust = -12.0 * strlen(sz) / 2.0;
    Solid *text = 0;
    for (int i = 0; sz[i]; i++) {
        const double xOrg = xStart + 12 * i + xJust, yOrg = yStart;
        double xPen = xOrg, yPen = yOrg;
        int glyph = (unsigned char)sz[i] - ' ';
        if (glyph < 0 || glyph >= nGlyphs) glyph = 0;
        const bool small = sz[i] >= 'a' && sz[i] <= 'z';
        bool down = false;
        Solid *letter = 0;
        for (int j = 0; j < 8; j++) {
            int code = font[8 * glyph + j];
            if (code == 0xFF) break;
            if (code == 0xFE) { down = false; continue; }
            double dx = (code >> 4) & 15, dy = code & 15;
            if (small) dy *= 0.75;
            if (down) {
                Point3 from(xPen / 12.0, -yPen / 12.0, 0.0), to((xOrg + dx) / 12.0, -(yOrg + dy) / 12.0, 0.0);
                letter = MakeUnion(letter, CylinderFromTo(from, to, r), (Solid *)new Translate(to, new Scale(r, r, r, new Sphere)));
            }
            xPen = xOrg + dx;
            yPen = yOrg + dy;
            letter = MakeUnion(letter, (Solid *)new Translate(xPen / 12.0, -yPen / 12.0, 0.0, new Scale(r, r, r, new Sphere)));
            down = true;
        }
        if (letter) text = MakeUnion(text, (Solid *)new BoundingBox(letter));
    }
    return new BoundingBox(text ? text : new NullSolid);
}

/* --------------------------------------------------------- texture / propagator identities ---- */
static volatile int s_texfn_tag;
#define TEXFN(name) void name(const Point3 &, Hit *, RayTracer *) { s_texfn_tag = __LINE__ * 64 + (int)sizeof(#name); }
TEXFN(Bumpy) TEXFN(MicroFacets) TEXFN(MicroFacets2) TEXFN(Wrinkled) TEXFN(Faceted20) TEXFN(Dented) TEXFN(Wavy)
TEXFN(Fibers) TEXFN(Blistered) TEXFN(Hammered) TEXFN(Tectonic) TEXFN(QuasiMicroFacet)
TEXFN(VoronoiBump) TEXFN(VoronoiAntiBump) TEXFN(VoronoiCrinkled)
TEXFN(Voronoi) TEXFN(Agate) TEXFN(Marble) TEXFN(Clouds) TEXFN(Wood) TEXFN(Spots) TEXFN(Veins) TEXFN(SeaScape)
TEXFN(EarthTexture) TEXFN(Earth2Texture) TEXFN(MarsTexture) TEXFN(MoonTexture) TEXFN(VenusTexture) TEXFN(MercuryTexture) TEXFN(SkyTexture)
Propagation ColoredGlass(const Point3 &, const Point3 &, Material *, RayTracer *) { Propagation p; p.cTransmittance = g_sEqualWhite; p.cSterisent = g_sBlack; return p; }
Propagation Foggy(const Point3 &, const Point3 &, Material *, RayTracer *) { Propagation p; p.cTransmittance = g_sEqualWhite; p.cSterisent = g_sBlack; return p; }
/* TextureMaps.cpp:162-168 is an empty stub; here the name selects the device's Rayleigh propagator */
Propagation Atmosphere(const Point3 &, const Point3 &, Material *, RayTracer *) { Propagation p; p.cTransmittance = g_sEqualWhite; p.cSterisent = g_sBlack; return p; }
int rt_texture_id(TextureMap f)
{
    if (!f) return RT_TEX_NONE;
    static const struct { TextureMap f; int id; } map[] = {
        { Bumpy, RT_TEX_BUMPY }, { MicroFacets, RT_TEX_MICROFACETS }, { MicroFacets2, RT_TEX_MICROFACETS2 }, { Wrinkled, RT_TEX_WRINKLED },
        { Faceted20, RT_TEX_FACETED20 }, { Dented, RT_TEX_DENTED }, { Wavy, RT_TEX_WAVY }, { Fibers, RT_TEX_FIBERS },
        { Blistered, RT_TEX_BLISTERED }, { Hammered, RT_TEX_HAMMERED }, { Tectonic, RT_TEX_TECTONIC }, { QuasiMicroFacet, RT_TEX_QUASIMICROFACET },
        { VoronoiBump, RT_TEX_VORONOIBUMP }, { VoronoiAntiBump, RT_TEX_VORONOIANTIBUMP }, { VoronoiCrinkled, RT_TEX_VORONOICRINKLED },
        { Voronoi, RT_TEX_VORONOI }, { Agate, RT_TEX_AGATE }, { Marble, RT_TEX_MARBLE }, { Clouds, RT_TEX_CLOUDS }, { Wood, RT_TEX_WOOD },
        { Spots, RT_TEX_SPOTS }, { Veins, RT_TEX_VEINS }, { SeaScape, RT_TEX_SEASCAPE },
        { EarthTexture, RT_TEX_EARTH }, { Earth2Texture, RT_TEX_EARTH2 }, { MarsTexture, RT_TEX_MARS }, { MoonTexture, RT_TEX_MOON },
        { VenusTexture, RT_TEX_VENUS }, { MercuryTexture, RT_TEX_MERCURY }, { SkyTexture, RT_TEX_SKY } };
    for (size_t i = 0; i < sizeof map / sizeof map[0]; i++) if (map[i].f == f) return map[i].id;
    return -1;
}
int rt_propagator_id(Propagator f)
{
    if (!f) return RT_PROP_NONE;
    if (f == ColoredGlass) return RT_PROP_COLOREDGLASS;
    if (f == Foggy) return RT_PROP_FOGGY;
    if (f == Atmosphere) return RT_PROP_ATMOSPHERE;
    return -1;
}

/* ---------------------------------------------------------------------------- flattening ---- */
int RtSerializer::add(int type, const Solid *who)
{
    RtNode n;
    memset(&n, 0, sizeof n);
    n.type = type; n.id = who ? who->nSerial : -1; n.left = n.right = n.link = -1; n.data = (int)data.size();
    nodes.push_back(n);
    owner.push_back(who);
    pending_link.push_back(0);
    return (int)nodes.size() - 1;
}
int RtSerializer::push(const double *p, int n) { int o = (int)data.size(); data.insert(data.end(), p, p + n); return o; }
void RtSerializer::resolve_links()
{
    std::map<const Solid *, int> where;
    for (size_t i = 0; i < owner.size(); i++) where[owner[i]] = (int)i;
    for (size_t i = 0; i < nodes.size(); i++) {
        nodes[i].link = -1;
        if (pending_link[i]) {
            std::map<const Solid *, int>::iterator it = where.find(pending_link[i]);
            if (it != where.end()) nodes[i].link = it->second;
        }
    }
}
static int emit_child(RtSerializer &s, Solid *c) { return c ? c->Emit(s) : -1; }

int NullSolid::Emit(RtSerializer &s) { return s.add(RT_NODE_NULL, this); }
int Quadric::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_QUADRIC, this);
    double p[3] = { P, Q, R };
    s.push(p, 3); s.nodes[i].ndata = 3; s.nodes[i].i0 = nClip;
    return i;
}
int ConeHyperCylinder::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_CONEHYPERCYL, this);
    double p[4] = { P, Q, Quadric::R, R };
    s.push(p, 4); s.nodes[i].ndata = 4; s.nodes[i].i0 = nClip;
    return i;
}
int ConvexPolyhedron::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_POLYHEDRON, this);
    for (int k = 0; k < nSides; k++) { double p[3] = { rgv[k].x, rgv[k].y, rgv[k].z }; s.push(p, 3); }
    s.nodes[i].ndata = 3 * nSides; s.nodes[i].i0 = nSides;
    return i;
}
int Slab::Emit(RtSerializer &s) { return s.add(RT_NODE_SLAB, this); }
int Torus::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_TORUS, this);
    s.push(&fMinorRadius, 1); s.nodes[i].ndata = 1;
    return i;
}
void MeshSolid::AddVertex(double x, double y, double z)
{
    Point3 p(x, y, z);
    if (rgfVertices.empty()) e = Extent(p, p);
    else { e.pMin = Min(e.pMin, p); e.pMax = Max(e.pMax, p); }          /* ObjectFormat.cpp:51-56 */
    rgfVertices.push_back(x); rgfVertices.push_back(y); rgfVertices.push_back(z);
}
int MeshSolid::LoadObjText(const char *sz)
{
    const char *p = sz;
    while (*p) {
        const char *eol = strchr(p, '\n');
        std::string line(p, eol ? (size_t)(eol - p) : strlen(p));
        p = eol ? eol + 1 : p + line.size();
        if (line.size() > 2 && line[0] == 'v' && line[1] == ' ') {
            double x, y, z;
            if (sscanf(line.c_str(), "v %lf%lf%lf", &x, &y, &z) != 3) return 0;
            AddVertex(x, y, z);
        } else if (line.size() > 2 && line[0] == 'f' && line[1] == ' ') {
            int a, b, c, t;
            if (sscanf(line.c_str(), "f %d/%d %d/%d %d/%d", &a, &t, &b, &t, &c, &t) != 6 &&
                sscanf(line.c_str(), "f %d %d %d", &a, &b, &c) != 3) return 0;
            const int n = nVertices();
            if (a < 1 || b < 1 || c < 1 || a > n || b > n || c > n) return 0;
            AddTriangle(a - 1, b - 1, c - 1);
        }
    }
    return nVertices();
}
int MeshSolid::LoadObjFile(const char *szFile)
{
    FILE *pf = fopen(szFile, "r");
    if (!pf) return 0;
    std::string all;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, pf)) > 0) all.append(buf, n);
    fclose(pf);
    return LoadObjText(all.c_str());
}
int MeshSolid::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_MESH, this);
    if (!rgfVertices.empty()) s.push(&rgfVertices[0], (int)rgfVertices.size());
    std::vector<double> idx(rgnTriangles.begin(), rgnTriangles.end());
    if (!idx.empty()) s.push(&idx[0], (int)idx.size());
    s.nodes[i].ndata = (int)(rgfVertices.size() + idx.size());
    s.nodes[i].i0 = nVertices(); s.nodes[i].i1 = nTriangles();
    return i;
}
MeshSolid *MeshSphere(int nu, int nv) { return MeshBlob(nu, nv, 0.0, 0); }
MeshSolid *MeshBlob(int nu, int nv, double amp, int k)
{
    MeshSolid *m = new MeshSolid;
    if (nu < 3) nu = 3;
    if (nv < 2) nv = 2;
    m->AddVertex(0.0, 0.0, 1.0);                               /* north pole: 0 */
    for (int j = 1; j < nv; j++) {
        double th = D_PI * j / nv;
        for (int i = 0; i < nu; i++) {
            double ph = 2.0 * D_PI * i / nu;
            const double r = (amp != 0.0) ? 1.0 + amp * sin(k * ph) * sin(k * th) : 1.0;
            m->AddVertex(r * (sin(th) * cos(ph)), r * (sin(th) * sin(ph)), r * cos(th));
        }
    }
    m->AddVertex(0.0, 0.0, -1.0);                              /* south pole */
    const int south = 1 + (nv - 1) * nu;
    for (int i = 0; i < nu; i++) {
        int i1 = (i + 1) % nu;
        m->AddTriangle(0, 1 + i, 1 + i1);
        for (int j = 0; j + 2 < nv; j++) {
            int a = 1 + j * nu + i, b = 1 + j * nu + i1, c = 1 + (j + 1) * nu + i, d = 1 + (j + 1) * nu + i1;
            m->AddTriangle(a, c, d);
            m->AddTriangle(a, d, b);
        }
        m->AddTriangle(south, 1 + (nv - 2) * nu + i1, 1 + (nv - 2) * nu + i);
    }
    return m;
}
MeshSolid *MeshTorus(int nu, int nv, double r)
{
    MeshSolid *m = new MeshSolid;
    if (nu < 3) nu = 3;
    if (nv < 3) nv = 3;
    for (int i = 0; i < nu; i++) {
        double ph = 2.0 * D_PI * i / nu;
        for (int j = 0; j < nv; j++) {
            double th = 2.0 * D_PI * j / nv;
            double R = 1.0 + r * cos(th);
            m->AddVertex(R * cos(ph), R * sin(ph), r * sin(th));
        }
    }
    for (int i = 0; i < nu; i++) {
        int i1 = (i + 1) % nu;
        for (int j = 0; j < nv; j++) {
            int j1 = (j + 1) % nv;
            int a = i * nv + j, b = i1 * nv + j, c = i1 * nv + j1, d = i * nv + j1;
            m->AddTriangle(a, b, c);
            m->AddTriangle(a, c, d);
        }
    }
    return m;
}
MeshSolid *MeshIcosahedron()
{
    MeshSolid *m = new MeshSolid;
    const double t = (1.0 + sqrt(5.0)) / 2.0, s = 1.0 / sqrt(1.0 + t * t);
    const double P[12][3] = { {-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t}, {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1} };
    for (int i = 0; i < 12; i++) m->AddVertex(P[i][0] * s, P[i][1] * s, P[i][2] * s);
    static const int f[20][3] = { {0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                  {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1} };
    for (int i = 0; i < 20; i++) m->AddTriangle(f[i][0], f[i][1], f[i][2]);
    return m;
}
int AffineSolid::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_AFFINE, this);
    double p[21];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { p[3 * r + c] = mTransform.m[r][c]; p[9 + 3 * r + c] = mInverse.m[r][c]; }
    p[18] = vTranslate.x; p[19] = vTranslate.y; p[20] = vTranslate.z;
    s.push(p, 21); s.nodes[i].ndata = 21;
    s.pending_link[i] = paParent;
    int c = emit_child(s, psChild);
    s.nodes[i].left = c;
    return i;
}
int BinarySolid::EmitBinary(RtSerializer &s, int type)
{
    int i = s.add(type, this);
    int l = emit_child(s, psLeft);
    int r = emit_child(s, psRight);
    s.nodes[i].left = l; s.nodes[i].right = r;
    return i;
}
int Union::Emit(RtSerializer &s) { return EmitBinary(s, RT_NODE_UNION); }
int TopUnion::Emit(RtSerializer &s) { return EmitBinary(s, RT_NODE_TOPUNION); }
int Intersection::Emit(RtSerializer &s) { return EmitBinary(s, RT_NODE_INTERSECTION); }
int Difference::Emit(RtSerializer &s) { return EmitBinary(s, RT_NODE_DIFFERENCE); }
int AngularClip::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_ANGULARCLIP, this);
    double p[6] = { cv1.x, cv1.y, cv1.z, cv2.x, cv2.y, cv2.z };
    s.push(p, 6); s.nodes[i].ndata = 6; s.nodes[i].i0 = nIntersect;
    int c = emit_child(s, ps);
    s.nodes[i].left = c;
    return i;
}
int BoundingBox::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_BOUNDINGBOX, this);
    double p[6] = { ex.pMin.x, ex.pMin.y, ex.pMin.z, ex.pMax.x, ex.pMax.y, ex.pMax.z };
    s.push(p, 6); s.nodes[i].ndata = 6;
    int c = emit_child(s, ps);
    s.nodes[i].left = c;
    return i;
}
int Material::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_MATERIAL, this);
    double p[2 + NSPEC] = { fDensity, fRefractiveIndex };
    for (int k = 0; k < NSPEC; k++) p[2 + k] = color.rgf[k];
    s.push(p, 2 + NSPEC); s.nodes[i].ndata = 2 + NSPEC;
    s.nodes[i].i0 = nLuster; s.nodes[i].i1 = rt_texture_id(Texture); s.nodes[i].i2 = rt_propagator_id(Propagate);
    s.pending_link[i] = paAffineTransforms;
    int c = emit_child(s, psChild);
    s.nodes[i].left = c;
    return i;
}
int Surface::Emit(RtSerializer &s)
{
    int i = s.add(RT_NODE_SURFACE, this);
    s.nodes[i].i0 = nRoughness; s.nodes[i].i1 = rt_texture_id(Texture);
    s.pending_link[i] = paAffineTransforms;
    int c = emit_child(s, psChild);
    s.nodes[i].left = c;
    return i;
}

void PerspectiveCamera::Describe(RtCamera &c) const { memset(&c, 0, sizeof c); c.type = RT_CAM_PERSPECTIVE; c.org[0] = pOrg.x; c.org[1] = pOrg.y; c.org[2] = pOrg.z; c.focal = focal; }
void StereographicCamera::Describe(RtCamera &c) const { memset(&c, 0, sizeof c); c.type = RT_CAM_STEREOGRAPHIC; c.org[0] = pOrg.x; c.org[1] = pOrg.y; c.org[2] = pOrg.z; c.focal = focal; }
void OrthographicCamera::Describe(RtCamera &c) const { memset(&c, 0, sizeof c); c.type = RT_CAM_ORTHOGRAPHIC; c.org[0] = pOrg.x; c.org[1] = pOrg.y; c.org[2] = pOrg.z; c.focal = focal; }
void ApertureCamera::Describe(RtCamera &c) const
{
    StereographicCamera::Describe(c);
    c.type = RT_CAM_APERTURE;
    c.vX[0] = vX.x; c.vX[1] = vX.y; c.vX[2] = vX.z; c.vY[0] = vY.x; c.vY[1] = vY.y; c.vY[2] = vY.z;
    c.focus_distance = fFocusDistance;
}

void FastBlueSamplingTile::QIndex1(int prime)                  /* Sampling.cpp:535-548 */
{
    if (prime > 0) {
        BuildQuasiIndex(g_BlueNoiseQuasiIndex2, 2, BLUE_N);
    } else {
        for (int i = 0; i < BLUE_N; i++) g_BlueNoiseQuasiIndex2[i] = i;
        if (prime < 0) RandomShuffle(g_BlueNoiseQuasiIndex2, BLUE_N);
    }
}

static char s_err[512];
const char *rt_frontend_last_error() { return s_err; }
static std::string s_asset_path;
void SetAssetPath(const char *szDirectory) { s_asset_path = szDirectory ? szDirectory : ""; }

int rt_flatten(const RayTracer &rt, RtSceneBlob &out, char *err, int errlen)
{
    RtSerializer &s = out.ser;
    s.nodes.clear(); s.data.clear(); s.lights.clear(); s.owner.clear(); s.pending_link.clear();
    int root = rt.psModel ? rt.psModel->Emit(s) : -1;
    s.resolve_links();
    for (size_t i = 0; i < s.nodes.size(); i++) {
        if ((s.nodes[i].type == RT_NODE_MATERIAL || s.nodes[i].type == RT_NODE_SURFACE) && s.nodes[i].i1 < 0) { snprintf(err, errlen, "node %d: TextureMap is not one of the TextureMaps.h routines", (int)i); return -1; }
        if (s.nodes[i].type == RT_NODE_MATERIAL && s.nodes[i].i2 < 0) { snprintf(err, errlen, "node %d: unknown Propagator", (int)i); return -1; }
    }
    for (PointLight *pl = rt.plPointLights; pl; pl = pl->plNext) {
        RtLight l;
        memset(&l, 0, sizeof l);
        l.type = pl->IsArea() ? RT_LIGHT_AREA : RT_LIGHT_POINT;
        l.no_shadow = pl->nNoShadow;
        l.org[0] = pl->pOrg.x; l.org[1] = pl->pOrg.y; l.org[2] = pl->pOrg.z;
        if (pl->IsArea()) {
            const AreaLight *al = (const AreaLight *)pl;
            l.vA[0] = al->vA.x; l.vA[1] = al->vA.y; l.vA[2] = al->vA.z; l.vB[0] = al->vB.x; l.vB[1] = al->vB.y; l.vB[2] = al->vB.z;
        }
        memcpy(l.flux, pl->flux.rgf, sizeof l.flux);
        s.lights.push_back(l);
    }
    RtSceneDesc &d = out.desc;
    memset(&d, 0, sizeof d);
    d.n_nodes = (int)s.nodes.size(); d.root = root; d.nodes = s.nodes.empty() ? 0 : &s.nodes[0];
    d.n_data = (int)s.data.size(); d.data = s.data.empty() ? 0 : &s.data[0];
    d.n_lights = (int)s.lights.size(); d.lights = s.lights.empty() ? 0 : &s.lights[0];
    d.has_ambient = rt.plAmbient != 0;
    if (rt.plAmbient) memcpy(d.ambient_flux, rt.plAmbient->flux.rgf, sizeof d.ambient_flux);
    if (!rt.pcCamera) { snprintf(err, errlen, "RayTracer::pcCamera is null"); return -1; }
    rt.pcCamera->Describe(d.camera);
    if (!rt.psSample) { snprintf(err, errlen, "RayTracer::psSample is null"); return -1; }
    int st = rt.psSample->DeviceSampler();
    if (st < 0) { snprintf(err, errlen, "this SamplingTile has no device path"); return -1; }
    d.sampler.type = st;
    d.sampler.venera = rt.psSample->nVenera;
    out.qi2.assign(g_BlueNoiseQuasiIndex2, g_BlueNoiseQuasiIndex2 + BLUE_N);
    out.qi3.assign(g_BlueNoiseQuasiIndex3, g_BlueNoiseQuasiIndex3 + BLUE_N);
    d.sampler.qindex2 = &out.qi2[0];
    d.sampler.qindex3 = &out.qi3[0];
    const double aux[6] = { g_uCamera, g_vCamera, g_uLight, g_vLight, g_uMicroFacet, g_vMicroFacet };
    memcpy(d.sampler.aux_init, aux, sizeof aux);
    d.sampler.prng_x = ML_nMarsagliaX; d.sampler.prng_c = ML_nMarsagliaC;
    /* the planet image maps the scene's textures name (LoadSkyMaps .. LoadMoonMaps, TextureMaps.cpp:1086-1234: file names, gamma and
       wrap modes as there) */
    static const struct { int tex; int slot[3]; } users[] = {
        { RT_TEX_EARTH, { RT_IMG_EARTH, RT_IMG_EARTHBUMP, RT_IMG_EARTHWATER } }, { RT_TEX_EARTH2, { RT_IMG_EARTH, RT_IMG_EARTHBUMP, -1 } },
        { RT_TEX_MARS, { RT_IMG_MARS, RT_IMG_MARSBUMP, -1 } }, { RT_TEX_MOON, { RT_IMG_MOON, RT_IMG_MOONBUMP, -1 } },
        { RT_TEX_VENUS, { RT_IMG_VENUS, RT_IMG_VENUSBUMP, -1 } }, { RT_TEX_MERCURY, { RT_IMG_MERCURY, RT_IMG_MERCURYBUMP, -1 } },
        { RT_TEX_SKY, { RT_IMG_SKY, -1, -1 } } };
    static const struct { const char *file; double gamma; } files[RT_IMG__COUNT] = {
        { "SkyColorMap.bmp", GAMMA_SRGB }, { "EarthColorMap.bmp", GAMMA_SRGB }, { "EarthBump.bmp", 1.0 }, { "EarthWater.bmp", 1.0 },
        { "MarsColorMap.bmp", GAMMA_SRGB }, { "MarsBump.bmp", 1.0 }, { "MoonColorMap2.bmp", GAMMA_SRGB }, { "MoonBump.bmp", 1.0 },
        { "VenusColorMap.bmp", GAMMA_SRGB }, { "VenusBump.bmp", 1.0 }, { "MercuryColorMap.bmp", GAMMA_SRGB }, { "MercuryBump.bmp", 1.0 } };
    bool need[RT_IMG__COUNT] = { false };
    bool any = false;
    for (size_t i = 0; i < s.nodes.size(); i++) {
        if (s.nodes[i].type != RT_NODE_MATERIAL && s.nodes[i].type != RT_NODE_SURFACE) continue;
        for (size_t u = 0; u < sizeof users / sizeof users[0]; u++)
            if (s.nodes[i].i1 == users[u].tex)
                for (int k = 0; k < 3; k++) if (users[u].slot[k] >= 0) { need[users[u].slot[k]] = true; any = true; }
    }
    for (size_t i = 0; i < out.image_store.size(); i++) delete out.image_store[i];
    out.image_store.clear(); out.images.clear();
    if (any) {
        const char *env = getenv("RT_AMD_ASSETS");
        const std::string dir = !s_asset_path.empty() ? s_asset_path : std::string(env ? env : "");
        if (dir.empty()) { snprintf(err, errlen, "the scene uses the planet image-map textures (TextureMaps.cpp:955-1243), whose BMP files the reference does not ship: give their directory with SetAssetPath() or RT_AMD_ASSETS"); return -1; }
        RtImageMap none;
        memset(&none, 0, sizeof none);
        out.images.assign(RT_IMG__COUNT, none);
        for (int k = 0; k < RT_IMG__COUNT; k++) {
            if (!need[k]) continue;
            Image *im = new Image;
            out.image_store.push_back(im);
            const std::string path = dir + "/" + files[k].file;
            if (!im->ReadBMP(path.c_str(), files[k].gamma)) { snprintf(err, errlen, "cannot read %s (Image::ReadBMP, Image.cpp:95-288)", path.c_str()); return -1; }
            im->m_bWrapX = WRAP_AROUND; im->m_bWrapY = WRAP_MIRROR;
            RtImageMap &m = out.images[k];
            m.width = im->m_nWidth; m.height = im->m_nHeight; m.channels = im->m_nChannels; m.wrap_x = im->m_bWrapX; m.wrap_y = im->m_bWrapY; m.pixels = im->m_pfImage;
        }
        d.n_images = RT_IMG__COUNT; d.images = &out.images[0];
    }
    return 0;
}

int Image::NewImage(int w, int h, int channels, int bWrapX, int bWrapY)
{
    int64_t n = (int64_t)w * h * channels, old = (int64_t)m_nWidth * m_nHeight * m_nChannels;
    if (!m_pfImage || n != old) { delete[] m_pfImage; m_pfImage = new float[n]; }
    m_nWidth = w; m_nHeight = h; m_nChannels = channels;
    m_bWrapX = bWrapX; m_bWrapY = bWrapY;
    return m_pfImage != 0;
}

static inline int ML_Modulus(int n, int m) { n = n % m; if (n < 0) n += m; return n; }      /* Image.cpp:50-57 */
static inline int ML_Wrap(int &i, int nLength, int nWrap)                                   /* Image.cpp:59-83 */
{
    int bDraw = 1;
    if (nWrap == WRAP_AROUND) i = ML_Modulus(i, nLength);
    else if (nWrap == WRAP_MIRROR) { i = ML_Modulus(i, 2 * nLength); if (i >= nLength) i = 2 * nLength - i - 1; }
    else { bDraw = 0; if (i < 0) i = 0; else if (i >= nLength) i = nLength - 1; else bDraw = 1; }
    return bDraw;
}
int Image::Wrap(int &iCol, int &jRow) const { return ML_Wrap(iCol, m_nWidth, m_bWrapX) & ML_Wrap(jRow, m_nHeight, m_bWrapY); }

void Image::SetGamma(double fGamma)             /* Image.cpp:497-517, the decode half */
{
    if (fGamma == GAMMA_SRGB) {
        for (int n = 0; n < 256; n++) {
            double f = double(n) * (1.0 / 255.0);
            if (f <= 0.03928) f = f / 12.92;
            else f = float(pow((f + 0.055) / 1.055, 2.4));
            m_rgfGammaT[n] = float(f);
        }
    } else
        for (int n = 0; n < 256; n++) m_rgfGammaT[n] = float(pow(double(n) * (1.0 / 255.0), fGamma));
}

int Image::ReadBMP(const char *szFileName, double fGamma)
{
    FILE *pf = fopen(szFileName, "rb");
    if (!pf) return 0;
    std::vector<unsigned char> line;
    unsigned char head[54], pal[256][4];
    unsigned mask[3] = { 0, 0, 0 };     /* red, green, blue */
    int shift[3] = { 0, 0, 0 };
    int ok = 0;
    do {
        if (fread(head, 1, 54, pf) != 54 || head[0] != 'B' || head[1] != 'M') break;
        #define U32(o) ((unsigned)head[o] | ((unsigned)head[(o) + 1] << 8) | ((unsigned)head[(o) + 2] << 16) | ((unsigned)head[(o) + 3] << 24))
        #define U16(o) ((unsigned)head[o] | ((unsigned)head[(o) + 1] << 8))
        const unsigned width = U32(18), height = U32(22), bits = U16(28), compression = U32(30), clr_used = U32(46);
        #undef U32
        #undef U16
        if (width == 0 || height == 0 || width > (1u << 20) || height > (1u << 20)) break;
        if (bits != 1 && bits != 4 && bits != 8 && bits != 16 && bits != 24 && bits != 32) break;
        if (compression == 0) {
            if (bits == 16) { mask[0] = 0x7C00; mask[1] = 0x03E0; mask[2] = 0x001F; }
            else if (bits == 32) { mask[0] = 0x00FF0000; mask[1] = 0x0000FF00; mask[2] = 0x000000FF; }
        } else if (compression == 3) {
            unsigned char m[12];
            if (fread(m, 1, 12, pf) != 12) break;
            for (int k = 0; k < 3; k++) mask[k] = (unsigned)m[4 * k] | ((unsigned)m[4 * k + 1] << 8) | ((unsigned)m[4 * k + 2] << 16) | ((unsigned)m[4 * k + 3] << 24);
        } else
            break;                      /* only RGB and BITFIELD compression */
        if (bits == 16 || bits == 32) {
            bool bad = false;
            for (int k = 0; k < 3; k++) {
                for (shift[k] = 0; shift[k] < 32; shift[k]++) if (mask[k] & (1u << shift[k])) break;
                if (shift[k] == 32) bad = true;         /* (the reference would divide by zero / shift by 32) */
            }
            if (bad) break;
        }
        int nColors;
        if (clr_used) nColors = (int)clr_used;
        else if (bits == 1) nColors = 2;
        else if (bits == 4) nColors = 16;
        else if (bits == 8) nColors = 256;
        else nColors = 0;
        if (nColors < 0 || nColors > 256) break;
        if (nColors && fread(pal, 1, (size_t)4 * nColors, pf) != (size_t)4 * nColors) break;
        int nChannels;
        if (nColors == 0) nChannels = bits >= 16 ? 3 : 1;
        else {
            nChannels = 1 + 2 * (bits >= 16);
            for (int i = 0; i < nColors; i++) if (pal[i][2] != pal[i][1] || pal[i][2] != pal[i][0]) { nChannels = 3; break; }
        }
        const size_t nLineSize = 4 * (((size_t)width * bits + 31) >> 5);
        line.resize(nLineSize);
        if (!NewImage((int)width, (int)height, nChannels)) break;
        SetGamma(fGamma);
        bool short_read = false;
        for (int j = m_nHeight - 1; j >= 0 && !short_read; --j) {
            if (fread(&line[0], 1, nLineSize, pf) != nLineSize) { short_read = true; break; }
            for (int i = 0; i < m_nWidth; i++) {
                int iRed = 0, iGrn = 0, iBlu = 0;
                unsigned nPixel;
                switch (bits) {
                case 1:  iRed = (line[i >> 3] >> (7 - (i & 7))) & 1; break;
                case 4:  iRed = (line[i >> 1] >> (4 - 4 * (i & 1))) & 15; break;
                case 8:  iRed = line[i]; break;
                case 16:
                    nPixel = line[2 * i] | (line[2 * i + 1] << 8);
                    iBlu = (int)((nPixel & mask[2]) >> shift[2]); iGrn = (int)((nPixel & mask[1]) >> shift[1]); iRed = (int)((nPixel & mask[0]) >> shift[0]);
                    iBlu = (255 * iBlu) / (int)(mask[2] >> shift[2]); iGrn = (255 * iGrn) / (int)(mask[1] >> shift[1]); iRed = (255 * iRed) / (int)(mask[0] >> shift[0]);
                    break;
                case 24: iRed = line[3 * i + 2]; iGrn = line[3 * i + 1]; iBlu = line[3 * i]; break;
                default:
                    nPixel = (unsigned)line[4 * i] | ((unsigned)line[4 * i + 1] << 8) | ((unsigned)line[4 * i + 2] << 16) | ((unsigned)line[4 * i + 3] << 24);
                    iBlu = (int)((nPixel & mask[2]) >> shift[2]); iGrn = (int)((nPixel & mask[1]) >> shift[1]); iRed = (int)((nPixel & mask[0]) >> shift[0]);
                    break;
                }
                if (bits <= 8) {
                    if (iRed >= nColors) { iRed = iGrn = iBlu = 0; }        /* (an index past the colour map: the reference reads its uninitialised tail) */
                    else { iBlu = pal[iRed][0]; iGrn = pal[iRed][1]; iRed = pal[iRed][2]; }
                }
                #define DECODE(n) ((n) < 0 ? 0.0f : ((n) > 255 ? 1.0f : m_rgfGammaT[n]))    /* GammaDecode, Image.h:197-205 */
                m_pfImage[i + m_nWidth * (int64_t)j] = DECODE(iRed);
                if (m_nChannels >= 3) {
                    m_pfImage[i + m_nWidth * (int64_t)(j + m_nHeight)] = DECODE(iGrn);
                    m_pfImage[i + m_nWidth * (int64_t)(j + 2 * m_nHeight)] = DECODE(iBlu);
                }
                #undef DECODE
            }
        }
        if (short_read) break;
        ok = 1;
    } while (0);
    fclose(pf);
    if (!ok) { delete[] m_pfImage; m_pfImage = 0; m_nWidth = m_nHeight = m_nChannels = 0; }
    return ok;
}

int Image::WriteBMP(const char *szFileName, double fGamma)
{
    if (m_pfImage == 0 || m_nChannels != 3) return 0;
    int64_t n = rt_bmp_size(m_nWidth, m_nHeight);
    std::vector<unsigned char> bytes((size_t)n);
    if (rt_bmp_encode(m_pfImage, m_nWidth, m_nHeight, fGamma, &bytes[0], n)) { snprintf(s_err, sizeof s_err, "rt_bmp_encode: %s", rt_last_error()); return 0; }
    FILE *pf = fopen(szFileName, "wb");
    if (!pf) return 0;
    size_t w = fwrite(&bytes[0], 1, (size_t)n, pf);
    fclose(pf);
    return w == (size_t)n ? 1 : 0;
}

int Video::NewVideo(const char *szFileName, Image &im)
{
    char h[128];
    if (im.m_nChannels != 3) return 0;
    int n = rt_y4m_header(im.m_nWidth, im.m_nHeight, h, (int)sizeof h);
    if (n < 0) { snprintf(s_err, sizeof s_err, "rt_y4m_header: %s", rt_last_error()); return 0; }
    hFile = fopen(szFileName, "wb");
    if (!hFile) return 0;
    m_nHeight = im.m_nHeight; m_nWidth = im.m_nWidth;
    return fwrite(h, 1, (size_t)n, hFile) == (size_t)n ? 1 : 0;
}

int Video::WriteFrame(Image &im)
{
    if (!hFile || im.m_nChannels != 3 || im.m_nWidth != m_nWidth || im.m_nHeight != m_nHeight) return 0;
    int64_t n = rt_y4m_frame_size(m_nWidth, m_nHeight);
    std::vector<unsigned char> bytes((size_t)n);
    if (rt_y4m_encode_frame(im.m_pfImage, m_nWidth, m_nHeight, m_fGamma, &bytes[0], n)) {
        snprintf(s_err, sizeof s_err, "rt_y4m_encode_frame: %s", rt_last_error());
        return 0;
    }
    return fwrite(&bytes[0], 1, (size_t)n, hFile) == (size_t)n ? 1 : 0;
}

int Video::Close()
{
    if (hFile) fclose(hFile);
    hFile = 0; m_nHeight = m_nWidth = 0;
    return 1;
}

int RayTracer::Render(Image &im, double spp)
{
    RtSceneBlob blob;
    if (rt_flatten(*this, blob, s_err, sizeof s_err)) return -1;
    if (im.m_nChannels != 3) { snprintf(s_err, sizeof s_err, "Render needs a 3-channel Image"); return -1; }
    rt_scene *sc = 0;
    if (rt_scene_upload(&blob.desc, &sc)) { snprintf(s_err, sizeof s_err, "rt_scene_upload: %s", rt_last_error()); return -1; }
    RtRenderDesc rd;
    memset(&rd, 0, sizeof rd);
    rd.width = im.m_nWidth; rd.height = im.m_nHeight; rd.spp = spp;
    int rc = rt_render(sc, &rd, im.m_pfImage, &stats);
    if (rc) snprintf(s_err, sizeof s_err, "rt_render: %s", rt_last_error());
    rt_scene_free(sc);
    if (!rc && blob.desc.sampler.type == RT_SAMPLER_JITTER) {      /* the tile consumed four RandomUnsigned per sample */
        uint32_t st[2] = { ML_nMarsagliaX, ML_nMarsagliaC };
        rt_random_skip(st, 4ull * (uint64_t)stats.samples);
        ML_nMarsagliaX = st[0]; ML_nMarsagliaC = st[1];
    }
    return rc ? -1 : (int)stats.samples;
}
