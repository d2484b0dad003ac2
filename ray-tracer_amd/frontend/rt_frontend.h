/*
 * rt_frontend.h -- host-side scene API of the MI355X render path.
 *
 * Keeps the reference's C++ scene-construction interface (class names, constructor signatures and
 * public member names of RayTracer2020.h, Affine.h, RadiometryColor.h, Solids.h, Lights.h,
 * Cameras.h, Sampling.h, TextureMaps.h, MonteCarlo.h) so that scene scripts written like
 * TestScene.cpp / DistributedRayTracing.cpp compile against it unchanged:
 *
 *     scene.psModel = new Union(new Material(0.9, s, MAT_METAL, new Translate(0,3,0, new Sphere)), ...);
 *     scene.psModel = scene.psModel->Optimize();
 *     scene.Render(im, 16.0);
 *
 * What differs is everything behind RayTracer::Render (RayTracer2020.cpp:33-38): no node can
 * intersect a ray on the host.  Render() flattens the object graph into an RtSceneDesc
 * (include/rt_scene_desc.h) and hands it to the HIP library through the C ABI of include/rt_amd.h.
 * There is no CPU fallback: without librt_amd.so Render() fails with an error.
 */
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/rt_scene_desc.h"

#define D_PI        3.14159265358979323846264338327950288419716939937510
#define D_SQRT2     1.41421356237309504880168872420969807856967187537694
#define D_SQRT3     1.73205080756887729352744634150587236694280525381038
#define D_PHI       1.61803398874989484820458683436563811772030917980576
#define EPSILON     RT_EPSILON
#ifdef INFINITY
#undef INFINITY
#endif
#define INFINITY    RT_INFINITY
#define NSPEC       RT_NSPEC
#define BLUE_NSQRT  RT_BLUE_NSQRT
#define BLUE_N      RT_BLUE_N

/* ---------------------------------------------------------------- affine algebra (Affine.h) ---- */
struct Vector3 {
    double x, y, z;
    Vector3() : x(0), y(0), z(0) {}
    Vector3(double f) : x(f), y(f), z(f) {}
    Vector3(double a, double b, double c) : x(a), y(b), z(c) {}
    Vector3 operator+(const Vector3 &v) const { return Vector3(x + v.x, y + v.y, z + v.z); }
    Vector3 operator-(const Vector3 &v) const { return Vector3(x - v.x, y - v.y, z - v.z); }
    Vector3 operator-() const { return Vector3(-x, -y, -z); }
    Vector3 operator*(double f) const { return Vector3(x * f, y * f, z * f); }
    Vector3 operator/(double f) const { return Vector3(x / f, y / f, z / f); }
};
inline double  Dot(const Vector3 &u, const Vector3 &v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline Vector3 Cross(const Vector3 &u, const Vector3 &v) { return Vector3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
inline double  Norm(const Vector3 &v) { return sqrt(Dot(v, v)); }
inline Vector3 Normalize(const Vector3 &v) { return v / Norm(v); }

struct Point3 {
    double x, y, z;
    Point3() : x(0), y(0), z(0) {}
    Point3(double a, double b, double c) : x(a), y(b), z(c) {}
    Point3  operator+(const Vector3 &v) const { return Point3(x + v.x, y + v.y, z + v.z); }
    Point3  operator-(const Vector3 &v) const { return Point3(x - v.x, y - v.y, z - v.z); }
    Vector3 operator-(const Point3 &p) const { return Vector3(x - p.x, y - p.y, z - p.z); }
};
inline double Max(double a, double b) { return a > b ? a : b; }
inline double Min(double a, double b) { return a < b ? a : b; }
inline Point3 Max(const Point3 &a, const Point3 &b) { return Point3(Max(a.x, b.x), Max(a.y, b.y), Max(a.z, b.z)); }
inline Point3 Min(const Point3 &a, const Point3 &b) { return Point3(Min(a.x, b.x), Min(a.y, b.y), Min(a.z, b.z)); }

struct Matrix3 {
    double m[3][3];
    Matrix3() { memset(m, 0, sizeof m); }
    Matrix3(double d) { memset(m, 0, sizeof m); m[0][0] = m[1][1] = m[2][2] = d; }
    Matrix3(double a, double b, double c, double d, double e, double f, double g, double h, double i)
    { m[0][0] = a; m[0][1] = b; m[0][2] = c; m[1][0] = d; m[1][1] = e; m[1][2] = f; m[2][0] = g; m[2][1] = h; m[2][2] = i; }
    Matrix3 operator*(const Matrix3 &o) const
    {
        Matrix3 r;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                r.m[i][j] = m[i][0] * o.m[0][j] + m[i][1] * o.m[1][j] + m[i][2] * o.m[2][j];
        return r;
    }
    Vector3 operator*(const Vector3 &v) const
    { return Vector3(v.x * m[0][0] + v.y * m[0][1] + v.z * m[0][2], v.x * m[1][0] + v.y * m[1][1] + v.z * m[1][2], v.x * m[2][0] + v.y * m[2][1] + v.z * m[2][2]); }
    Point3 operator*(const Point3 &v) const
    { return Point3(v.x * m[0][0] + v.y * m[0][1] + v.z * m[0][2], v.x * m[1][0] + v.y * m[1][1] + v.z * m[1][2], v.x * m[2][0] + v.y * m[2][1] + v.z * m[2][2]); }
    Matrix3 Transpose() const { return Matrix3(m[0][0], m[1][0], m[2][0], m[0][1], m[1][1], m[2][1], m[0][2], m[1][2], m[2][2]); }
};
inline Matrix3 ScaleMatrix(double x, double y, double z) { return Matrix3(x, 0, 0, 0, y, 0, 0, 0, z); }
Matrix3 EulerMatrix(double r0, double r1, double r2, const char *convention = "xyz");
Matrix3 RotateAboutAxis(const Vector3 &axis, double radians);
Matrix3 RotateFromTo(const Vector3 &from, const Vector3 &to);
Matrix3 InverseMatrix(const Matrix3 &mat);

struct CoVector3 {
    double x, y, z;
    CoVector3() : x(0), y(0), z(0) {}
    CoVector3(double f) : x(f), y(f), z(f) {}
    CoVector3(double a, double b, double c) : x(a), y(b), z(c) {}
    CoVector3 operator+(const CoVector3 &v) const { return CoVector3(x + v.x, y + v.y, z + v.z); }
    CoVector3 operator-() const { return CoVector3(-x, -y, -z); }
    CoVector3 operator*(double f) const { return CoVector3(x * f, y * f, z * f); }
    double    operator*(const Vector3 &v) const { return x * v.x + y * v.y + z * v.z; }
    double    operator*(const Point3 &p) const { return x * p.x + y * p.y + z * p.z; }
};

struct Vector2 {
    double x, y;
    Vector2() : x(0), y(0) {}
    Vector2(double a, double b) : x(a), y(b) {}
    Vector2 operator+(const Vector2 &v) const { return Vector2(x + v.x, y + v.y); }
    Vector2 operator*(double f) const { return Vector2(x * f, y * f); }
    Vector2 operator/(double f) const { return Vector2(x / f, y / f); }
};

struct Vector5 {
    union { double rgf[5]; struct { double x, y, z, a, b; }; };
    Vector5() { for (int i = 0; i < 5; i++) rgf[i] = 0; }
    Vector5(double x1, double x2, double x3, double x4, double x5) : x(x1), y(x2), z(x3), a(x4), b(x5) {}
    Vector5 operator/(double f) const { return Vector5(x / f, y / f, z / f, a / f, b / f); }
    double  Norm() { return sqrt(x * x + y * y + z * z + a * a + b * b); }
};

/* ------------------------------------------------- colour and radiometry (RadiometryColor.h) ---- */
struct DisplayRGB {
    float red, grn, blu;
    DisplayRGB() : red(0), grn(0), blu(0) {}
    DisplayRGB(double r, double g, double b) : red(float(r)), grn(float(g)), blu(float(b)) {}
    DisplayRGB(double f) : red(float(f)), grn(float(f)), blu(float(f)) {}
};

struct Spectrum {
    float rgf[NSPEC];
    Spectrum operator+(const Spectrum &s) const { Spectrum r; for (int i = 0; i < NSPEC; i++) r.rgf[i] = rgf[i] + s.rgf[i]; return r; }
    Spectrum operator-(const Spectrum &s) const { Spectrum r; for (int i = 0; i < NSPEC; i++) r.rgf[i] = rgf[i] - s.rgf[i]; return r; }
    Spectrum operator*(const Spectrum &s) const { Spectrum r; for (int i = 0; i < NSPEC; i++) r.rgf[i] = rgf[i] * s.rgf[i]; return r; }
    Spectrum operator/(const Spectrum &s) const { Spectrum r; for (int i = 0; i < NSPEC; i++) r.rgf[i] = s.rgf[i] ? rgf[i] / s.rgf[i] : 0.0f; return r; }
    Spectrum operator*(double f) const { Spectrum r; for (int i = 0; i < NSPEC; i++) r.rgf[i] = float(rgf[i] * f); return r; }
    Spectrum operator/(double f) const { Spectrum r; if (f) f = 1.0 / f; for (int i = 0; i < NSPEC; i++) r.rgf[i] = float(rgf[i] * f); return r; }
    float &operator[](int i) { return rgf[i]; }
    DisplayRGB sRGB() const;
};
typedef Spectrum Color, Flux, Irradiance, Radiance, Reflectance, Transmittance;

/* Constant spectra (values from ray-tracer_amd/data/tables.bin, embedded in the library). */
extern Spectrum g_sRed, g_sGreen, g_sBlue, g_sYellow, g_sCyan, g_sMagenta, g_sWhite, g_sOrange;
extern Spectrum g_sBlack, g_sEqualWhite, g_sD6500;
extern Spectrum g_sWater, g_sCobaltGlass, g_sCopper, g_sGold, g_sRuby, g_sEmerald, g_sSapphire, g_sWhiskey;
extern Spectrum g_sEarthSky, g_sSolar, g_sIllumD65;
extern DisplayRGB g_rgbIron, g_rgbCopper, g_rgbGold, g_rgbZinc, g_rgbZilver, g_rgbAluminum, g_rgbSilicon;
extern DisplayRGB SpectrumToRGB[NSPEC];
Spectrum RGBtoSpectrum(DisplayRGB rgb, int n6500 = 0);
Spectrum RGBtoSpectrum(double f);
Spectrum BlackBody(double kelvin);
const Spectrum *rt_spectrum_by_name(const char *name);     /* any `g_sXxx` of the tables, 0 if unknown */

/* --------------------------------------------------- pseudo-random numbers (MonteCarlo.h) ---- */
extern unsigned ML_nMarsagliaX, ML_nMarsagliaC;
unsigned RandomUnsigned();
double   RandomDouble();
float    RandomFloat();
unsigned RandomLessThanN(unsigned n);
void     ML_InitializeRandom(unsigned x = 886459, unsigned c = 361290869);
template <class T> void RandomShuffle(T items[], int n)
{
    for (int j = n - 1; j > 0; --j) {
        int i = (int)RandomLessThanN((unsigned)(j + 1));
        T t = items[i]; items[i] = items[j]; items[j] = t;
    }
}
void BuildQuasiIndex(int rgn[], int prime, int n);          /* Patterns.h */
void FibonacciSphere(double rgf[], int i, int n);

/* --------------------------------------------------------------------------- Image (Image.h) ---- */
#define WRAP_CLAMP   0                          /* Image.h:17-19 */
#define WRAP_AROUND  1
#define WRAP_MIRROR  2
#define GAMMA_SRGB   -1.0                       /* Image.h:21 */
struct Image {
    float *m_pfImage;
    int    m_nHeight, m_nWidth, m_nChannels;
    int    m_bWrapX, m_bWrapY;
    float  m_rgfGammaT[256];                    /* decode table of the last SetGamma (Image.cpp:497-517) */
    Image() : m_pfImage(0), m_nHeight(0), m_nWidth(0), m_nChannels(0), m_bWrapX(WRAP_MIRROR), m_bWrapY(WRAP_MIRROR) {}
    ~Image() { delete[] m_pfImage; }
    int NewImage(int w, int h, int channels = 1, int bWrapX = WRAP_MIRROR, int bWrapY = WRAP_MIRROR);
    /* Image::ReadBMP (Image.cpp:95-288): version-3 BMP of 1, 4, 8, 16, 24 or 32 bits per pixel, BI_RGB or BI_BITFIELDS, to linear
       floats through the gamma table.  Returns 1, or 0 on error (and frees the image) like the reference. */
    int ReadBMP(const char *szFileName, double fGamma = GAMMA_SRGB);
    void SetGamma(double fGamma);
    int Wrap(int &iCol, int &jRow) const;       /* Image.cpp:50-89 */
    float Get(int iCol, int jRow, int k = 0) const      /* Image.h:226-236 */
    {
        int i = iCol, j = jRow;
        if ((unsigned)i >= (unsigned)m_nWidth || (unsigned)j >= (unsigned)m_nHeight) Wrap(i, j);
        return m_pfImage[i + m_nWidth * (int64_t)(j + m_nHeight * k)];
    }
    DisplayRGB GetRGB(int i, int j) const { return DisplayRGB(Get(i, j, 0), Get(i, j, 1), Get(i, j, 2)); }
    /* Image::WriteBMP (Image.cpp:416-474), 3-channel images: clamp to gamut + gamma-encode on the GPU (rt_bmp_encode),
       then the file is written here.  Returns 1, or 0 on error like the reference. */
    int WriteBMP(const char *szFileName, double fGamma = -1.0 /* GAMMA_SRGB */);
private:
    Image(const Image &);
    Image &operator=(const Image &);
};

/* ----------------------------------------------------------------------------- Video (Video.h) ---- */
struct Video {                                  /* YUV4MPEG2 writer, Video.cpp:8-80; frames encoded on the GPU (rt_y4m_encode_frame) */
    FILE *hFile;
    int   m_nHeight, m_nWidth;
    float m_fGamma;
    Video() : hFile(0), m_nHeight(0), m_nWidth(0), m_fGamma(2.0) {}
    int   NewVideo(const char *szFileName, Image &im);
    int   WriteFrame(Image &im);
    int   Close();
    float SetGamma(float fGamma) { return m_fGamma = fGamma; }
};

/* -------------------------------------------------------- rays, extents (RayTracer2020.h) ---- */
struct Ray {
    Point3 pOrg; Vector3 vDir; double tMin; int bPrimary;
    Ray() : tMin(INFINITY), bPrimary(0) {}
    Ray(const Point3 &p, const Vector3 &v, int cam = 0) : pOrg(p), vDir(v), tMin(INFINITY), bPrimary(cam) {}
    Point3 operator()(double t) const { return pOrg + vDir * t; }
};

struct Extent {
    Point3 pMin, pMax;
    Extent() {}
    Extent(double w) : pMin(-w, -w, -w), pMax(w, w, w) {}
    Extent(const Point3 &a, const Point3 &b) : pMin(a), pMax(b) {}
    int NullExtent() const { return (pMax.x <= pMin.x) || (pMax.y <= pMin.y) || (pMax.z <= pMin.z); }
};

struct RtSerializer;
struct AffineSolid;
struct Hit;         /* opaque: hit lists only exist on the device */
struct RayTracer;

struct Solid {
    int nIsAffine;
    int nSerial;                                /* creation order; becomes RtNode.id */
    Solid();
    virtual ~Solid() {}
    virtual Extent GetExtent() { return Extent(Point3(0, 0, 0), Point3(0, 0, 0)); }
    virtual Solid *Optimize() { return this; }
    virtual int    Emit(RtSerializer &s) = 0;   /* append this subtree to a scene description, return node index */
};

/* ------------------------------------------------------------------------ solids (Solids.h) ---- */
struct NullSolid : Solid { int Emit(RtSerializer &s); };

struct Quadric : Solid {
    double P, Q, R; int nClip;
    Quadric(double p, double q, double r, int clip) : P(p), Q(q), R(r), nClip(clip) {}
    Extent GetExtent() { return Extent(1.0); }
    int Emit(RtSerializer &s);
};
struct Sphere : Quadric { Sphere() : Quadric(1.0, 0.0, -1.0, 0) {} };
struct HollowSphere : Quadric { HollowSphere() : Quadric(1.0, 0.0, -1.0, -1) {} };
struct Cylinder : Quadric { Cylinder() : Quadric(0.0, 0.0, -1.0, 1) {} };
struct Cone : Quadric { Cone() : Quadric(-0.25, -0.5, -0.25, 0) {} };
struct ConeFX : Quadric { ConeFX() : Quadric(-0.25, 0.0, 0.0, 0) {} };
struct ConeUnit : Quadric { ConeUnit() : Quadric(-1.0, 0.0, 0.0, 0) {} };
struct Hyperboloid1 : Quadric { Hyperboloid1(double r) : Quadric(-1.0, 0.0, -r * r, 0) {} };
struct Hyperboloid2 : Quadric { Hyperboloid2(double r) : Quadric(-1.0, 0.0, +r * r, 0) {} };
struct Paraboloid : Quadric { Paraboloid() : Quadric(0.0, -0.5, 0.0, 1) {} };
inline double SSS(double x) { return x < 0.0 ? -x * x : x * x; }
struct ConeHyperCylinder : Quadric {
    double R;
    ConeHyperCylinder(double r) : Quadric(SSS(r) - 1.0, 0.0, -SSS(r), 0), R(r) {}
    Extent GetExtent();
    int Emit(RtSerializer &s);
};

extern CoVector3 rgvTetrahedron[4], rgvCube[8], rgvOctahedron[6], rgvDodecahedron[20], rgvIcosahedron[12];
/* ClipPlanes (Solids.h:128, Solids.cpp:801-857): the box around what half space(s) cv * p < fFromOrigin leave of an extent's corner set.
   As checked in, the reference stops after the first plane; so does this (TestClipPlanes, Test.cpp:1822-1830). */
Extent ClipPlanes(const Extent &eInitial, const CoVector3 rgcv[], int nv, double fFromOrigin);
struct ConvexPolyhedron : Solid {
    CoVector3 *rgv; int nSides;
    ConvexPolyhedron() : rgv(0), nSides(0) {}
    ConvexPolyhedron(CoVector3 *p, int n) : rgv(p), nSides(n) {}
    Extent GetExtent() { return Extent(1.0); }
    int Emit(RtSerializer &s);
};
struct Slab : Solid {
    Extent GetExtent() { return Extent(Point3(-INFINITY, -INFINITY, -1.0), Point3(INFINITY, INFINITY, 1.0)); }
    int Emit(RtSerializer &s);
};
/* the face-normal table of each Platonic solid is its dual's vertex table (Solids.h:146-164) */
struct Tetrahedron : ConvexPolyhedron { Tetrahedron() : ConvexPolyhedron(rgvTetrahedron, 4) {} };
struct Cube : ConvexPolyhedron { Cube() : ConvexPolyhedron(rgvOctahedron, 6) {} };
struct Octahedron : ConvexPolyhedron { Octahedron() : ConvexPolyhedron(rgvCube, 8) {} };
struct Dodecahedron : ConvexPolyhedron { Dodecahedron() : ConvexPolyhedron(rgvIcosahedron, 12) {} };
struct Icosahedron : ConvexPolyhedron { Icosahedron() : ConvexPolyhedron(rgvDodecahedron, 20) {} };
struct Prism : ConvexPolyhedron { Prism(int n); };
struct Fibonacci : ConvexPolyhedron { Fibonacci(int n); };

struct Torus : Solid {
    double fMinorRadius;
    Torus(double r) : fMinorRadius(r) {}
    Extent GetExtent() { return Extent(Point3(-1.0 - fMinorRadius, -1.0 - fMinorRadius, -fMinorRadius), Point3(1.0 + fMinorRadius, 1.0 + fMinorRadius, fMinorRadius)); }
    int Emit(RtSerializer &s);
};

/* A closed triangle mesh as a Solid (policy Q14): vertices and triangles in the form ObjectFormat.cpp:12-28 holds them,
   read by the grammar of LoadObjFile (:31-75: "v x y z" and "f a/b c/d e/f", plus plain "f a b c"; OBJ indices are
   1-based), or made procedurally.  Winding: counter-clockwise seen from outside, so Cross(e1, e2) points out. */
struct MeshSolid : Solid {
    std::vector<double> rgfVertices;    /* x y z per vertex */
    std::vector<int>    rgnTriangles;   /* iV0 iV1 iV2 per triangle, 0-based */
    Extent e;
    MeshSolid() {}
    int  nVertices() const { return (int)(rgfVertices.size() / 3); }
    int  nTriangles() const { return (int)(rgnTriangles.size() / 3); }
    void AddVertex(double x, double y, double z);
    void AddTriangle(int a, int b, int c) { rgnTriangles.push_back(a); rgnTriangles.push_back(b); rgnTriangles.push_back(c); }
    int  LoadObjText(const char *sz);           /* returns the number of vertices read, 0 on a malformed face */
    int  LoadObjFile(const char *szFile);
    Extent GetExtent() { return e; }
    int Emit(RtSerializer &s);
};
MeshSolid *MeshSphere(int nLongitude, int nLatitude);          /* unit sphere, 2 * nLongitude * (nLatitude - 1) triangles */
MeshSolid *MeshBlob(int nLongitude, int nLatitude, double fAmplitude, int nLobes);   /* star-shaped: radius 1 + a sin(k phi) sin(k theta) */
MeshSolid *MeshTorus(int nMajor, int nMinor, double fMinorRadius);     /* same shape as Torus(r), 2 * nMajor * nMinor triangles */
MeshSolid *MeshIcosahedron();                                   /* 20 triangles, vertices on the unit sphere */

#define SURF_MATTE      RT_SURF_MATTE
#define SURF_BURNISHED  RT_SURF_BURNISHED
#define SURF_SMOOTH     RT_SURF_SMOOTH
#define SURF_SHINY      RT_SURF_SHINY
#define SURF_POLISHED   RT_SURF_POLISHED
#define SURF_WET        RT_SURF_WET
#define MAT_NOSHADE     RT_MAT_NOSHADE
#define MAT_PLASTIC     RT_MAT_PLASTIC
#define MAT_METAL       RT_MAT_METAL
#define MAT_TRANSPARENT RT_MAT_TRANSPARENT
#define MAT_COPPER      RT_MAT_COPPER
#define MAT_GOLD        RT_MAT_GOLD
#define MAT_IRON        RT_MAT_IRON
#define MAT_ALUMINUM    RT_MAT_ALUMINUM
#define MAT_OSMIUM      RT_MAT_OSMIUM
#define MAT_SILVER      RT_MAT_SILVER
#define MAT_TITANIUM    RT_MAT_TITANIUM
#define REFRACT_AIR       1.000293
#define REFRACT_WATER     1.333
#define REFRACT_ETHANOL   1.3571
#define REFRACT_ICE       1.36
#define REFRACT_QUARTZ    1.4298
#define REFRACT_PYREX     1.474
#define REFRACT_CORNING   1.5078
#define REFRACT_LEADGLASS 1.71
#define REFRACT_SAPPHIRE  1.77
#define REFRACT_DIAMOND   2.42

struct Material;
struct Propagation { Transmittance cTransmittance; Radiance cSterisent; };
typedef void (*TextureMap)(const Point3 &p, Hit *ph, RayTracer *scene);
typedef Propagation (*Propagator)(const Point3 &p0, const Point3 &p1, Material *pm, RayTracer *scene);
/* The call-outs are evaluated on the device; on the host these functions are identities whose
   ADDRESS selects the device routine (TextureMaps.h). */
void Bumpy(const Point3 &, Hit *, RayTracer *);
void MicroFacets(const Point3 &, Hit *, RayTracer *);
void MicroFacets2(const Point3 &, Hit *, RayTracer *);
void Wrinkled(const Point3 &, Hit *, RayTracer *);
void Faceted20(const Point3 &, Hit *, RayTracer *);
void Dented(const Point3 &, Hit *, RayTracer *);
void Wavy(const Point3 &, Hit *, RayTracer *);
void Fibers(const Point3 &, Hit *, RayTracer *);
void Blistered(const Point3 &, Hit *, RayTracer *);
void Hammered(const Point3 &, Hit *, RayTracer *);
void Tectonic(const Point3 &, Hit *, RayTracer *);
void QuasiMicroFacet(const Point3 &, Hit *, RayTracer *);
void VoronoiBump(const Point3 &, Hit *, RayTracer *);
void VoronoiAntiBump(const Point3 &, Hit *, RayTracer *);
void VoronoiCrinkled(const Point3 &, Hit *, RayTracer *);
/* solid colour textures (TextureMaps.cpp:518-683) and SeaScape (:497-513) */
void Voronoi(const Point3 &, Hit *, RayTracer *);
void Agate(const Point3 &, Hit *, RayTracer *);
void Marble(const Point3 &, Hit *, RayTracer *);
void Clouds(const Point3 &, Hit *, RayTracer *);
void Wood(const Point3 &, Hit *, RayTracer *);
void Spots(const Point3 &, Hit *, RayTracer *);
void Veins(const Point3 &, Hit *, RayTracer *);
void SeaScape(const Point3 &, Hit *, RayTracer *);
/* the planet image maps (TextureMaps.cpp:1098-1243).  The reference reads "EarthColorMap.bmp", "EarthBump.bmp", ... from the working
   directory at the first hit and does not ship them; here rt_flatten reads the files a scene's textures name from the asset
   directory -- SetAssetPath(), or the environment's RT_AMD_ASSETS -- and fails when one is missing. */
void EarthTexture(const Point3 &, Hit *, RayTracer *);
void Earth2Texture(const Point3 &, Hit *, RayTracer *);
void MarsTexture(const Point3 &, Hit *, RayTracer *);
void MoonTexture(const Point3 &, Hit *, RayTracer *);
void VenusTexture(const Point3 &, Hit *, RayTracer *);
void MercuryTexture(const Point3 &, Hit *, RayTracer *);
void SkyTexture(const Point3 &, Hit *, RayTracer *);
void SetAssetPath(const char *szDirectory);     /* null or "": back to RT_AMD_ASSETS */
Propagation ColoredGlass(const Point3 &, const Point3 &, Material *, RayTracer *);
Propagation Foggy(const Point3 &, const Point3 &, Material *, RayTracer *);
Propagation Atmosphere(const Point3 &, const Point3 &, Material *, RayTracer *);
int rt_texture_id(TextureMap f);        /* -1: not a device routine */
int rt_propagator_id(Propagator f);

struct Surface : Solid {
    TextureMap Texture; AffineSolid *paAffineTransforms; Solid *psChild; int nRoughness;
    Surface(int rough, TextureMap t, Solid *ps) : Texture(t), paAffineTransforms(0), psChild(ps), nRoughness(rough) {}
    Surface(int rough, Solid *ps) : Texture(0), paAffineTransforms(0), psChild(ps), nRoughness(rough) {}
    Extent GetExtent() { return psChild->GetExtent(); }
    Solid *Optimize();
    int Emit(RtSerializer &s);
};

struct Material : Solid {
    double fDensity, fRefractiveIndex; Color color; TextureMap Texture; Propagator Propagate;
    AffineSolid *paAffineTransforms; Solid *psChild; int nLuster;
    Material(double f, const Color &c, int luster, Solid *ps)
        : fDensity(f), fRefractiveIndex(1.0 + f), color(c), Texture(0), Propagate(0), paAffineTransforms(0), psChild(ps), nLuster(luster) {}
    Material(double f, double n, const Color &c, int luster, Propagator prop, Solid *ps)
        : fDensity(f), fRefractiveIndex(n), color(c), Texture(0), Propagate(prop), paAffineTransforms(0), psChild(ps), nLuster(luster) {}
    Material(double f, const Color &c, int luster, TextureMap t, Solid *ps)
        : fDensity(f), fRefractiveIndex(1.0 + f), color(c), Texture(t), Propagate(0), paAffineTransforms(0), psChild(ps), nLuster(luster) {}
    Extent GetExtent() { return psChild->GetExtent(); }
    Solid *Optimize();
    int Emit(RtSerializer &s);
};

struct AffineSolid : Solid {
    Matrix3 mTransform, mInverse; Vector3 vTranslate; Solid *psChild; AffineSolid *paParent;
    AffineSolid(Solid *ps) : psChild(ps), paParent(0) { nIsAffine = 1; }
    Extent GetExtent();
    Solid *Optimize();
    int Emit(RtSerializer &s);
};
struct Scale : AffineSolid {
    Scale(double x, double y, double z, Solid *ps) : AffineSolid(ps)
    {
        if (x == 0.0 || y == 0.0 || z == 0.0) printf("Afine Scale singular %g %g %g\n", x, y, z);
        mTransform = ScaleMatrix(x, y, z); mInverse = ScaleMatrix(1.0 / x, 1.0 / y, 1.0 / z); vTranslate = Vector3(0.0, 0.0, 0.0);
    }
};
struct Rotate : AffineSolid {
    Rotate(double x, double y, double z, Solid *ps, const char *sz = "xyz") : AffineSolid(ps)
    { mTransform = EulerMatrix(x, y, z, sz); mInverse = mTransform.Transpose(); vTranslate = Vector3(0.0, 0.0, 0.0); }
    Rotate(const Vector3 &axis, double rad, Solid *ps) : AffineSolid(ps)
    { mTransform = RotateAboutAxis(axis, rad); mInverse = mTransform.Transpose(); vTranslate = Vector3(0.0, 0.0, 0.0); }
    Rotate(const Vector3 &from, const Vector3 &to, Solid *ps) : AffineSolid(ps)
    { mTransform = RotateFromTo(from, to); mInverse = mTransform.Transpose(); vTranslate = Vector3(0.0, 0.0, 0.0); }
};
struct Translate : AffineSolid {
    Translate(double x, double y, double z, Solid *ps) : AffineSolid(ps) { mTransform = Matrix3(1.0); mInverse = Matrix3(1.0); vTranslate = Vector3(x, y, z); }
    Translate(const Vector3 &v, Solid *ps) : AffineSolid(ps) { mTransform = Matrix3(1.0); mInverse = Matrix3(1.0); vTranslate = v; }
    Translate(const Point3 &p, Solid *ps) : AffineSolid(ps) { mTransform = Matrix3(1.0); mInverse = Matrix3(1.0); vTranslate = p - Point3(0.0, 0.0, 0.0); }
};

struct BinarySolid : Solid {
    Solid *psLeft, *psRight;
    BinarySolid(Solid *a, Solid *b) : psLeft(a), psRight(b) {}
    Solid *Optimize() { if (psLeft) psLeft = psLeft->Optimize(); if (psRight) psRight = psRight->Optimize(); return this; }
    int EmitBinary(RtSerializer &s, int type);
};
struct Union : BinarySolid { Union(Solid *a, Solid *b) : BinarySolid(a, b) {} Extent GetExtent(); int Emit(RtSerializer &s); };
struct TopUnion : Union { TopUnion(Solid *a, Solid *b) : Union(a, b) {} int Emit(RtSerializer &s); };
struct Intersection : BinarySolid { Intersection(Solid *a, Solid *b) : BinarySolid(a, b) {} Extent GetExtent(); int Emit(RtSerializer &s); };
struct Difference : BinarySolid { Difference(Solid *a, Solid *b) : BinarySolid(a, b) {} Extent GetExtent(); int Emit(RtSerializer &s); };

struct AngularClip : Solid {
    Solid *ps; CoVector3 cv1, cv2; double fAng1, fAng2; int nIntersect;
    AngularClip(double f1, double f2, Solid *p);
    Extent GetExtent() { return ps->GetExtent(); }
    Solid *Optimize() { if (ps) ps = ps->Optimize(); return this; }
    int Emit(RtSerializer &s);
};

struct BoundingBox : Solid {
    Extent ex; Solid *ps;
    BoundingBox(Solid *p) : ps(p) { ex = ps->GetExtent(); if (ex.NullExtent()) ps = new NullSolid; }
    Extent GetExtent() { return ex; }
    Solid *Optimize() { ps = ps->Optimize(); return this; }
    int Emit(RtSerializer &s);
};

/* balanced-tree builders (Solids.cpp:1251-1459): same split points as the reference's overloads */
Solid *rt_make_union(Solid **ps, int n, int top);
Solid *rt_make_intersection(Solid **ps, int n);
template <class... A> Solid *MakeUnion(A... a) { Solid *v[] = { a... }; return rt_make_union(v, (int)sizeof...(a), 0); }
template <class... A> Solid *MakeTopUnion(A... a) { Solid *v[] = { a... }; return rt_make_union(v, (int)sizeof...(a), 1); }
template <class... A> Solid *MakeIntersection(A... a) { Solid *v[] = { a... }; return rt_make_intersection(v, (int)sizeof...(a)); }

/* shape library (Solids.h:502-517) */
Solid *Cuboid(double r, double x = 1.0, double y = 1.0, double z = 1.0);
Solid *CylinderFromTo(const Point3 &from, const Point3 &to, double r);
Solid *TetrahedronCage(double r, double rs);
Solid *HexahedronCage(double r, double rs);
Solid *OctahedronCage(double r, double rs);
Solid *IcosahedronCage(double r, double rs);
Solid *DodecahedronCage(double r, double rs);
Solid *Diskoid(double r);
Solid *Rodoid(double r);
Solid *PseudoHelix(double r, double zRepeat, double fAng = 2.0 * D_PI);   /* Solids.cpp:1880-1917 */
Solid *Squareoid(double r);
Solid *Knoboid(double r, double z = 1.0);
Solid *Knoboid2(double r, double z = 1.0);
Solid *Hyperboloidoid(double r, double rt, double z = 1.0);
Solid *Slaboid(double x, double y, double z, double r);
Solid *Slaboidoid(double x, double y, double z, double r, double r2);
#define JUSTIFIED_RIGHT  0                      /* Image.h:30-32 */
#define JUSTIFIED_LEFT   1
#define JUSTIFIED_CENTER 2
Solid *SolidText(double xStart, double yStart, const char *sz, int nJust, double r);    /* Solids.cpp:2173-2230 */
#define DENSE_AIR       0.000293                /* Solids.h:226-234 */
#define DENSE_VENUS     0.014716
#define DENSE_WATER     0.333
#define DENSE_ICE       0.36
#define DENSE_PYREX     0.474
#define DENSE_GLASS     0.52
#define DENSE_LEADGLASS 0.71
#define DENSE_SAPPHIRE  0.77
#define DENSE_DIAMOND   1.42
#define DODECDIHEDRAL   116.56505117707798935157219372045

/* --------------------------------------------------- lights (RayTracer2020.h:143-172, Lights.h) ---- */
struct Light {
    Flux flux;
    Light() {}
    Light(DisplayRGB rgb) { flux = RGBtoSpectrum(rgb); }
    Light(const Spectrum &s) { flux = s; }
    virtual ~Light() {}
    virtual Point3 Origin() = 0;
};
struct PointLight : Light {
    Point3 pOrg; PointLight *plNext; int nNoShadow;
    PointLight(const Point3 &p, double power, PointLight *pl = 0) : Light(DisplayRGB(power)), pOrg(p), plNext(pl), nNoShadow(0) {}
    PointLight(const Point3 &p, Spectrum power, PointLight *pl = 0) : Light(power), pOrg(p), plNext(pl), nNoShadow(0) {}
    Point3 Origin() { return pOrg; }
    virtual int IsArea() const { return 0; }
};
struct AreaLight : PointLight {
    Vector3 vA, vB;
    AreaLight(double diam, const Point3 &p, Spectrum power, PointLight *pl = 0) : PointLight(p, power, pl)
    {
        Vector3 vN = Point3(0.0, 0.0, 0.0) - p;
        vA = Normalize(Cross(vN, Vector3(0.0, 1.0, 0.0))) * diam;
        vB = Normalize(Cross(vN, vA)) * diam;
    }
    int IsArea() const { return 1; }
};
struct AmbientLight : Light {
    AmbientLight(double background) : Light(DisplayRGB(background)) {}
    AmbientLight(Spectrum background) : Light(background) {}
    Point3 Origin() { return Point3(0.0, 0.0, 0.0); }
};
/* RayTracer::plAmbient is an AmbientLight* and Illuminate is not virtual in the reference
   (RayTracer2020.h:169,190), so a HavercosineLight lights the scene as plain ambient with sAbove. */
struct HavercosineLight : AmbientLight {
    Flux flux2;
    HavercosineLight(const Spectrum &above, const Spectrum &below) : AmbientLight(above), flux2(below) {}
};

/* ------------------------------------------------------------------------ cameras (Cameras.h) ---- */
struct Camera {
    Point3 pOrg; double focal;
    Camera() : focal(35.0) {}
    Camera(Point3 p, double f = 35.0) : pOrg(p), focal(f) {}
    virtual ~Camera() {}
    virtual void Describe(RtCamera &c) const = 0;
};
struct PerspectiveCamera : Camera {
    PerspectiveCamera(Point3 p = Point3(0.0, 0.0, 1.0), double f = 35.0) : Camera(p, f) {}
    void Describe(RtCamera &c) const;
};
struct StereographicCamera : Camera {
    StereographicCamera(Point3 p = Point3(0.0, 0.0, 1.0), double f = 35.0) : Camera(p, f) {}
    void Describe(RtCamera &c) const;
};
struct OrthographicCamera : Camera {
    OrthographicCamera(Point3 p = Point3(0.0, 0.0, 1.0), double f = 35.0) : Camera(p, f) {}
    void Describe(RtCamera &c) const;
};
struct ApertureCamera : StereographicCamera {
    Vector3 vX, vY; double fApertureRadius, fFocusDistance;
    ApertureCamera(double fstop, double fdist, Point3 p, double focal_) : StereographicCamera(p, focal_), fFocusDistance(fdist)
    {
        fApertureRadius = 0.001 * 0.5 * focal_ / fstop;
        vX = Vector3(fApertureRadius, 0.0, 0.0);
        vY = Vector3(0.0, fApertureRadius, 0.0);
    }
    void Describe(RtCamera &c) const;
};

/* ------------------------------------------------------------------ sampling tiles (Sampling.h) ---- */
extern double g_uCamera, g_vCamera, g_uLight, g_vLight, g_uMicroFacet, g_vMicroFacet;
extern Vector2 g_rgvBlueNoise[BLUE_N];
extern int g_BlueNoiseQuasiIndex2[BLUE_N], g_BlueNoiseQuasiIndex3[BLUE_N];

struct SamplingTile {
    int nSamples, nVenera;
    SamplingTile() : nSamples(0), nVenera(0) {}
    virtual ~SamplingTile() {}
    virtual int  Initialize() { return nSamples; }
    virtual void QIndex1(int) {}
    virtual int  DeviceSampler() const = 0;     /* RtSamplerType, -1 if this tile has no device path yet */
};
struct TestSamplingTile : SamplingTile { int DeviceSampler() const { return RT_SAMPLER_TEST; } };
/* DebugSamplingTile (Sampling.cpp:350-386): one ray through the image centre, the image filled with its colour */
struct DebugSamplingTile : SamplingTile { int DeviceSampler() const { return RT_SAMPLER_DEBUG; } };
/* JitterSamplingTile (Sampling.cpp:388-419): one sample per supersample pixel, jittered by the global MWC stream */
struct JitterSamplingTile : SamplingTile { int DeviceSampler() const { return RT_SAMPLER_JITTER; } };
/* BlueNoiseSamplingTile (Sampling.cpp:421-466): tiles of 40/sqrt(spp) output pixels, per-sample Kaiser splat */
struct BlueNoiseSamplingTile : SamplingTile { int DeviceSampler() const { return RT_SAMPLER_BLUENOISE; } };
struct FastBlueSamplingTile : SamplingTile {
    FastBlueSamplingTile() { nVenera = 0; }
    void QIndex1(int prime);
    int DeviceSampler() const { return RT_SAMPLER_FASTBLUE; }
};

/* AdaptiveBlueSamplingTile (Sampling.cpp:550-709): a FastBlue pass, then 1600 more samples in every 20x20 block of the
   supersampled image whose neighbourhood has a Roberts gradient above 0.30 */
struct AdaptiveBlueSamplingTile : FastBlueSamplingTile { int DeviceSampler() const { return RT_SAMPLER_ADAPTIVEBLUE; } };

/* ------------------------------------------------------------------- scene (RayTracer2020.h:188) ---- */
struct RayTracer {
    Solid *psModel; AmbientLight *plAmbient; PointLight *plPointLights; Camera *pcCamera; SamplingTile *psSample;
    int nRayLevel; double fAttenuation;
    RtStats stats;                  /* filled by Render */
    RayTracer() : psModel(0), plAmbient(0), plPointLights(0), pcCamera(0), psSample(0), nRayLevel(0), fAttenuation(1.0) { memset(&stats, 0, sizeof stats); }
    int Render(Image &im, double fSamplesPerPixel = 1.0);   /* returns the sample count, -1 on error (see rt_frontend_last_error) */
};

/* -------------------------------------------------------------------------------- flattening ---- */
struct RtSerializer {
    std::vector<RtNode>  nodes;
    std::vector<double>  data;
    std::vector<RtLight> lights;
    std::vector<const Solid *> owner;           /* node index -> object, to resolve Optimize() links */
    std::vector<const AffineSolid *> pending_link;
    int  add(int type, const Solid *who);
    int  push(const double *p, int n);
    void resolve_links();
};
struct RtSceneBlob {                            /* owns every array an RtSceneDesc points to */
    RtSerializer ser;
    std::vector<int32_t> qi2, qi3;
    std::vector<RtImageMap> images;             /* RT_IMG__COUNT entries when a planet texture is used */
    std::vector<Image *> image_store;
    RtSceneDesc desc;
    RtSceneBlob() {}
    ~RtSceneBlob() { for (size_t i = 0; i < image_store.size(); i++) delete image_store[i]; }
private:
    RtSceneBlob(const RtSceneBlob &);
    RtSceneBlob &operator=(const RtSceneBlob &);
};
int  rt_flatten(const RayTracer &rt, RtSceneBlob &out, char *err, int errlen);
const char *rt_frontend_last_error();
void rt_frontend_reset_ids();             /* restart the RtNode.id serial (ids are per scene build) */
