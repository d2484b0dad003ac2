/*
 * rt_scenes.cpp -- scene scripts written against the host scene API exactly the way the
 * reference's drivers write theirs, plus a small text front door for tests and tools.
 *
 * Named scenes (the synthetic, deterministic inputs of bench.py and the parity tests):
 *   S0  the as-checked-in driver scene: TestRayTraceScene (TestScene.cpp:1539-1628) = 41 RNDSPEC
 *       draws, InitializeCSystem (:1362-1378), MakeModelCSystem(0, 1000) (:1380-1424) -- a 0.5-radius
 *       neutronium ball and 256 small plastic spheres on the unit sphere --, MakeLightsLorenz
 *       (:1472-1487), StereographicCamera((0,0,8), 70) (:1593).
 *   S1  MakeModel(30) (:44-77): the nine-primitive zoo over a floor slab, with MakeLights (:1456-1470)
 *       (area light + point light, shadows on), same camera.
 *   S2  DistributedRayTracer's scene (DistributedRayTracing.cpp:56-150): IcosahedronCage in polished
 *       blue plastic over a floor inside a NOSHADE sky dome, ApertureCamera(1.4, 24, (0,0,9), 35)
 *       (the commented camera line :128), area + point light, QIndex1(0).
 *   S4  one frame of Laboratory() (LaboratoryAnimation.cpp:387-442): five bumpy spectral-metal balls, a stopcock, a
 *       five-neck Florence flask on a copper stand, an Erlenmeyer flask, a condenser envelope of length t with a
 *       spiral tube inside, all Pyrex; floor and sky dome of Environment (:368-385); HavercosineLight ambient, one
 *       point light, StereographicCamera((0,0,12)).  "S4" is the frame t = 4.5; "S4@t" any t of the animation
 *       (3 <= t <= 6 there).  "S4atm" / "S4atm@t": the same frame in hazy air (the Atmosphere propagator).
 * Argument evaluation order where the reference leaves it to the compiler (RandomDouble() calls
 * as constructor arguments): right to left, which is what both MSVC and g++ do.
 */
#include <stdlib.h>
#include <ctype.h>
#include <string>
#include "rt_frontend.h"

/* ------------------------------------------------------------------------------ S0 / S1 ---- */
static Spectrum random_spectrum()       /* RNDSPEC, TestScene.cpp:11 */
{
    double b = RandomDouble(), g = RandomDouble(), r = RandomDouble();
    return RGBtoSpectrum(DisplayRGB(r, g, b));
}

static void palette(Spectrum *rgs /*20*/)  /* TestScene.cpp:1566-1581 */
{
    rgs[0] = random_spectrum();
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < 20; i++) rgs[i] = random_spectrum();
    for (int i = 1; i <= 9; i += 2) rgs[i] = g_sCopper;
}

enum { NLORENZ = 256 };
static Vector5  s_csystem[NLORENZ];
static Spectrum s_cscolor[NLORENZ];

static void init_csystem()              /* TestScene.cpp:1362-1378 */
{
    Spectrum blue = g_sEqualWhite * 0.5 + g_sBlue * 0.5;
    for (int i = 0; i < NLORENZ; i++) {
        double e = RandomDouble(), d = RandomDouble(), c = RandomDouble(), b = RandomDouble(), a = RandomDouble();
        Vector5 v(a, b, c, d, e);
        s_csystem[i] = v / v.Norm();
        s_cscolor[i] = i == 0 ? g_sRed : blue;
    }
}

static Solid *model_csystem(int steps)  /* TestScene.cpp:1380-1424 */
{
    const double dt = 0.01;
    while (--steps >= 0) {
        for (int i = 0; i < NLORENZ; i++) {
            Vector5 v = s_csystem[i], dv;
            for (int j = 0; j < 5; j++)
                dv.rgf[j] = v.rgf[(j + 1) % 5] * v.rgf[(j + 2) % 5] + v.rgf[(j + 4) % 5] * v.rgf[(j + 3) % 5] - 2.0 * v.rgf[(j + 1) % 5] * v.rgf[(j + 4) % 5];
            for (int j = 0; j < 5; j++)
                v.rgf[j] += dv.rgf[j] * dt;
            s_csystem[i] = v / v.Norm();
        }
    }
    Solid *all = new Translate(0.0, 0.0, 0.0, new Scale(0.5, 0.5, 0.5, new Sphere));
    for (int i = 0; i < NLORENZ; i++) {
        Vector3 v3(s_csystem[i].x, s_csystem[i].y, s_csystem[i].z);
        Solid *ps = new Translate(v3 / Norm(v3), new Scale(0.05, 0.05, 0.05, new Sphere));
        if (i == 0)
            ps = new Translate(s_csystem[i].x, s_csystem[i].y, s_csystem[i].z, new Scale(0.1, 0.1, 0.1, new Sphere));
        all = new Union(new Material(0.9, s_cscolor[i], MAT_PLASTIC, ps), all);
    }
    return all;
}

static Solid *model_zoo(double deg, const Spectrum *rgs)     /* MakeModel, TestScene.cpp:44-77 */
{
    double a = deg * D_PI / 180.0;
    Solid *p[9];
    p[0] = new Translate(-3.0, -3.0, 0.0, new Rotate(0.0, a, 0.0, new Cylinder));
    p[1] = new Translate(-3.0, 0.0, 0.0, new Rotate(0.0, a, 0.0, new Torus(0.5)));
    p[2] = new Translate(-3.0, 3.0, 0.0, new Rotate(0.0, a, 0.0, new Intersection(new Sphere, new Hyperboloid2(0.125))));
    p[3] = new Translate(0.0, -3.0, 0.0, new Rotate(0.0, a, 0.0, new Intersection(new Sphere, new Hyperboloid1(0.125))));
    p[4] = new Translate(0.0, 0.0, 0.0, new Rotate(0.0, a, 0.0, new Difference(new Icosahedron, new Cone)));
    p[5] = new Translate(0.0, 3.0, 0.0, new Rotate(0.0, a, 0.0, new Dodecahedron));
    p[6] = new Translate(3.0, -3.0, 0.0, new Rotate(0.0, a, 0.0, new Scale(1.2, 0.8, 0.8, new Sphere)));
    p[7] = new Translate(3.0, 0.0, 0.0, new Rotate(0.0, a, 0.0, new Union(new Torus(0.2), new Sphere)));
    p[8] = new Translate(3.0, 3.0, 0.0, new Rotate(0.0, a, 0.0, new Difference(new Cube, new Scale(0.5, 0.5, 1.2, new Cylinder))));
    for (int i = 0; i < 9; i++) p[i] = new Material(0.9, rgs[i], MAT_METAL, p[i]);
    Solid *ps = p[0];
    for (int i = 1; i < 9; i++) ps = new Union(p[i], ps);
    ps = new Surface(SURF_BURNISHED, ps);
    return new Union(ps, new Translate(0.0, 5.25, -10.0, new Scale(12.0, 1.0, 12.0, new Cube)));
}

static void lights_two(RayTracer *rt)   /* MakeLights, TestScene.cpp:1456-1470 */
{
    Spectrum s6500 = BlackBody(6500.0);
    rt->plAmbient = new AmbientLight(s6500 * 0.02);
    rt->plPointLights = new AreaLight(1.0, Point3(10.0, -10.0, 10.0), s6500 * 10000.0);
    rt->plPointLights = new PointLight(Point3(-10.0, -10.0, -10.0), s6500 * 500.0, rt->plPointLights);
}

static void lights_far(RayTracer *rt)   /* MakeLightsLorenz, TestScene.cpp:1472-1487 */
{
    Spectrum s6500 = BlackBody(6500.0);
    rt->plAmbient = new AmbientLight(s6500 * 0.02);
    rt->plPointLights = new AreaLight(1.0, Point3(100.0, -100.0, 100.0), s6500 * 1000000.0);
    rt->plPointLights->nNoShadow = 1;
}

static void reset_globals()
{
    ML_InitializeRandom();
    g_uCamera = g_vCamera = g_uLight = g_vLight = g_uMicroFacet = g_vMicroFacet = 0.0;
    BuildQuasiIndex(g_BlueNoiseQuasiIndex2, 2, BLUE_N);
}

static int build_S0(RayTracer &rt)
{
    Spectrum rgs[20];
    reset_globals();
    palette(rgs);
    lights_far(&rt);
    rt.pcCamera = new StereographicCamera(Point3(+0.0, 0.0, 8.0), 70.0);
    init_csystem();
    rt.psModel = model_csystem(1000);
    return 0;
}

static int build_S1(RayTracer &rt)
{
    Spectrum rgs[20];
    reset_globals();
    palette(rgs);
    lights_two(&rt);
    rt.pcCamera = new StereographicCamera(Point3(+0.0, 0.0, 8.0), 70.0);
    rt.psModel = model_zoo(30.0, rgs);
    return 0;
}

/* ------------------------------------------------------------------------------------ S2 ---- */
static Solid *lab_environment(Solid *ps, RayTracer &rt)     /* EnvironmentLab, DistributedRayTracing.cpp:56-77 */
{
    Spectrum s6500 = BlackBody(6500.0);
    double f20 = 0.0 * D_PI / 180.0;
    rt.plAmbient = new HavercosineLight(s6500 * 0.002, g_sBlack);
    rt.plPointLights = new AreaLight(1.0, Point3(9.0, -6.5, 12.0), s6500 * 8000.0 * 1.2);
    rt.plPointLights = new PointLight(Point3(10.0, -10.0, -25.0), s6500 * 1000.0, rt.plPointLights);
    return MakeUnion(
        ps,
        new Translate(0.0, 1.5, -10.0, new Rotate(f20, 0.0, 0.0, new Scale(16.0, 0.1, 16.0, new Material(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube)))),
        new Scale(1000.0, 1000.0, 1000.0, new Material(0.9, g_sEarthSky * 2.0, MAT_NOSHADE, new HollowSphere)));
}

static int build_S2(RayTracer &rt)      /* DistributedRayTracer, DistributedRayTracing.cpp:111-150 */
{
    reset_globals();
    FastBlueSamplingTile *tile = new FastBlueSamplingTile;
    tile->QIndex1(0);
    rt.psSample = tile;
    rt.pcCamera = new ApertureCamera(1.4, 9.0 + 15.0, Point3(0.0, +0.0, 9.0), 35.0);
    double ang = 0.0 * D_PI / 180.0;
    Solid *ps = MakeUnion(new Translate(+0.0, 0.0, 0.0, new Rotate(0.0, ang, 0.0, IcosahedronCage(0.05, 0.1))));
    Spectrum blue = (g_sEqualWhite + g_sBlue) * 0.5;
    ps = new Surface(SURF_POLISHED, 0, new Material(0.7, REFRACT_PYREX, blue, MAT_PLASTIC, 0, ps));
    rt.psModel = lab_environment(ps, rt);
    return 0;
}

/* ------------------------------------------------------------------------------------ S4 ----
 * The glassware grammar of LaboratoryAnimation.cpp: every vessel is (outer shape) minus (inner shape), wall THICK. */
namespace lab {
const double THICK = 0.05, TUBE = 0.25, f90 = 90.0 * D_PI / 180.0;

Solid *shell(Solid *outer, Solid *inner) { return new Difference(outer, inner); }
Solid *rim(double z) { return new Translate(0.0, 0.0, z, new Scale(TUBE + THICK, TUBE + THICK, 2.0 * THICK, new Sphere)); }

Solid *connect_part(double r, double rr, double w, double theta, double grow, double eps)   /* one wall of Connect, :14-66 */
{
    const double R1 = TUBE, R2 = 3.0 / 4.0 * TUBE;
    Solid *hyp = MakeIntersection(new Scale(1.0, 1.0, 1.0, new ConeHyperCylinder(r / rr)),
                                  new Scale(1.0, 1.0, 0.5 + grow, new Translate(0.0, 0.0, 1.0 - (grow != 0.0 ? 0.00001 : 0.0), new Cube)));
    hyp = new Translate(0.0, 0.0, w - 3.0 * TUBE - (grow != 0.0 ? 0.0 : 0.00001), new Scale(rr, rr, 2.0 * TUBE + grow, hyp));
    Solid *t1 = new Scale(R1, R1, R1, new AngularClip(D_PI - eps, 1.5 * D_PI + eps, new Torus(r / R1)));
    t1 = new Translate(w - 2.0 * TUBE, 0.0, w - 3.0 * TUBE, new Rotate(f90, 0.0, 0.0, t1));
    const double eps2 = grow != 0.0 ? 0.0001 : 0.0;
    Solid *t2 = new Scale(R2, R2, R2, new AngularClip(theta - eps2, 1.0 * D_PI + eps2, new Torus(r / R2)));
    t2 = new Translate(TUBE, R2, w - 3.0 * TUBE - TUBE, t2);
    return MakeUnion(hyp, t1, t2);
}
Solid *Connect(double fAng)                         /* :14-66 */
{
    const double w = 3.0 * TUBE, r = 0.5 * TUBE;
    const double theta = 0.5 * D_PI - atan(3.0 / 4.0 / 1.0);
    Solid *outer = connect_part(r, TUBE, w, theta, 0.0, 0.00001);
    Solid *inner = connect_part(r - THICK, TUBE - THICK, w, theta, 0.0001, 0.001);
    Solid *ps = shell(outer, inner);
    ps = new Translate(0.0, 0.0, 2.0 * TUBE - w - TUBE, ps);
    return new BoundingBox(new Rotate(0.0, 0.0, fAng + theta, ps));
}
Solid *bead(double r, double r1)
{
    return new Difference(new Scale(r1, r1, r1, new Sphere), new Rotate(0.5 * D_PI, 0.0, 0.0, new Scale(r - THICK, r - THICK, r1 + 0.001, new Cylinder)));
}
Solid *SpiralTube(double L)                         /* :96-137 */
{
    const double w = 3.0 * TUBE, cap1 = -w - 1.5 * TUBE, lMain = L + 2.0 * cap1, cap2 = -L - cap1;
    const double r = 0.5 * TUBE, R = w - 2.0 * r, z = 2.0 * TUBE, fAng = 2.0 * D_PI * lMain / z;
    double fRad = fAng + 0.5 * D_PI;
    while (fRad > 2.0 * D_PI) fRad -= 2.0 * D_PI;
    const double r1 = r * 1.05;
    Solid *ps = new Scale(R, R, R, PseudoHelix(r / R, z / R, fAng));
    ps = new Union(ps, new Rotate(0.0, 0.0, 0.5 * D_PI, new Translate(R, 0.0, 0.0, bead(r, r1))));
    ps = new Union(ps, new Rotate(0.0, 0.0, -fRad + 1.0 * D_PI, new Translate(R, 0.0, -lMain, bead(r, r1))));
    Solid *ps2 = new Scale(R, R, R, PseudoHelix((r - THICK) / R, z / R, fAng));
    ps = new Translate(0.0, 0.0, cap1, new Difference(ps, ps2));
    ps = new Union(ps, new Rotate(0.0, 0.0, 1.0 * D_PI, new Translate(0.0, 0.0, cap1 + w, new Scale(1.0, -1.0, +1.0, Connect(0.0)))));
    ps = new Union(ps, new Rotate(0.0, 0.0, -fRad + 0.5 * D_PI, new Translate(0.0, 0.0, cap2 - w, new Scale(1.0, +1.0, -1.0, Connect(0.0)))));
    ps = new Rotate(0.0, 0.5 * D_PI, 0.0, new Rotate(-f90, 0.0, 0.0, ps));
    return new BoundingBox(ps);
}
Solid *CondensorEnvelope(double L)                  /* :139-179 */
{
    const double w = 3.0 * TUBE, end1 = 0.0, cap1 = -w - 1.5 * TUBE, zMain = cap1, lMain = L + 2.0 * cap1, cap2 = -L - cap1, end2 = -L - end1;
    const double h = 0.5 * TUBE;
    Solid *ps = MakeUnion(
        rim(end1),
        new Translate(0.0, 0.0, end1, new Scale(TUBE, TUBE, 2.0 * TUBE, new Translate(0.0, 0.0, -1.0, new Cylinder))),
        new Translate(0.0, 0.0, cap1, new Scale(w, w, w, new Sphere)),
        new Translate(w + h, 0.0, cap1, new Rotate(0.0, f90, 0.0, new Scale(h, h, TUBE, new Cylinder))),
        new Translate(0.0, 0.0, zMain, new Scale(w, w, 0.5 * lMain, new Translate(0.0, 0.0, -1.0, new Cylinder))),
        new Translate(w + h, 0.0, cap2, new Rotate(0.0, f90, 0.0, new Scale(h, h, TUBE, new Cylinder))),
        new Translate(0.0, 0.0, cap2, new Scale(w, w, w, new Sphere)),
        new Translate(0.0, 0.0, end2, new Scale(TUBE, TUBE, 2.0 * TUBE, new Translate(0.0, 0.0, +1.0, new Cylinder))),
        rim(-L));
    const double ti = TUBE - THICK, wi = w - THICK, hi = h - THICK;
    Solid *ps2 = MakeUnion(
        new Translate(0.0, 0.0, end1, new Scale(ti, ti, 2.5 * TUBE, new Translate(0.0, 0.0, -0.75, new Cylinder))),
        new Translate(0.0, 0.0, cap1, new Scale(wi, wi, wi, new Sphere)),
        new Translate(w + h, 0.0, cap1, new Rotate(0.0, f90, 0.0, new Scale(hi, hi, 1.01 * TUBE, new Cylinder))),
        new Translate(0.0, 0.0, zMain, new Scale(wi, wi, 0.5 * lMain, new Translate(0.0, 0.0, -1.0, new Cylinder))),
        new Translate(w + h, 0.0, cap2, new Rotate(0.0, f90, 0.0, new Scale(hi, hi, 1.01 * TUBE, new Cylinder))),
        new Translate(0.0, 0.0, cap2, new Scale(wi, wi, wi, new Sphere)),
        new Translate(0.0, 0.0, end2, new Scale(ti, ti, 2.5 * TUBE, new Translate(0.0, 0.0, +0.75, new Cylinder))));
    return new BoundingBox(new Rotate(-f90, 0.0, 0.0, shell(ps, ps2)));
}
Solid *capped_barrel(double half)                   /* a tube section with a plain and a flanged end, StopCock :189-196 */
{
    return new Union(new Scale(TUBE, TUBE, half, new Cylinder),
                     new Union(new Translate(0.0, 0.0, +half, new Scale(TUBE, TUBE, 2.0 * THICK, new Sphere)), rim(-half)));
}
Solid *StopCock(double deg)                         /* :181-206 */
{
    const double fAng = deg * D_PI / 180.0, ti = TUBE - THICK, tp = TUBE - THICK - 0.01;
    Solid *sh = new Union(capped_barrel(2.0 * TUBE), new Rotate(-f90, 0.0, 0.0, capped_barrel(1.5 * TUBE)));
    sh = new Difference(sh, new Scale(ti, ti, 2.5 * (TUBE + THICK), new Cylinder));
    sh = new Difference(sh, new Rotate(-f90, 0.0, 0.0, new Scale(ti, ti, 2.0 * (TUBE + THICK), new Cylinder)));
    Solid *plug = new Union(new Union(new Scale(tp, tp, 2.0 * TUBE, new Cylinder),
                                      new Translate(0.0, 0.0, -2.0 * TUBE, new Scale(tp, tp, ti, new Sphere))),
                            new Translate(0.0, 0.0, -2.5 * (TUBE + THICK), new Scale(0.3 * TUBE, 1.5 * TUBE, 0.5 * TUBE, new Sphere)));
    plug = new Difference(plug, new Rotate(-f90 * 0.8, 0.0, 0.0, new Scale(0.25 * TUBE, 0.25 * TUBE, TUBE + THICK, new Cylinder)));
    return new BoundingBox(new Rotate(0.0, f90, 0.0, new Union(sh, new Rotate(0.0, 0.0, fAng, plug))));
}
Solid *ErlenmeyerFlask(double r, double fNeck)      /* :232-256 */
{
    const double rt = 0.15, z = 2.0, k = 1.01 + rt;
    Solid *body = new Scale(r, r, r, new Intersection(new Scale(k, k, k, new Translate(0.0, 0.0, 1.0, new Cube)), Hyperboloidoid(TUBE / r, rt, z)));
    Solid *neck = new Union(new Translate(0.0, 0.0, 0.0, new Scale(TUBE, TUBE, fNeck, new Cylinder)), rim(0.0 - fNeck));
    Solid *ps = new Rotate(-f90, 0.0, 0.0, new Union(neck, body));
    const double ri = r - 2.0 * THICK;
    body = new Scale(ri, ri, r - THICK, new Intersection(new Scale(k, k, k, new Translate(0.0, 0.0, 1.0, new Cube)),
                                                          Hyperboloidoid((TUBE - THICK) / ri, rt, z)));
    neck = new Translate(0.0, 0.0, 0.0, new Scale(TUBE - THICK, TUBE - THICK, fNeck + 2.1 * THICK, new Cylinder));
    return new BoundingBox(shell(ps, new Rotate(-f90, 0.0, 0.0, new Union(neck, body))));
}
Solid *flask_neck(double r, double fNeck)
{
    return new Union(new Rotate(f90, 0.0, 0.0, new Translate(0.0, 0.0, r, new Scale(TUBE, TUBE, fNeck, new Cylinder))),
                     new Rotate(f90, 0.0, 0.0, rim(r + fNeck)));
}
Solid *flask_bore(double r, double fNeck)
{
    return new Rotate(f90, 0.0, 0.0, new Translate(0.0, 0.0, r, new Scale(TUBE - THICK, TUBE - THICK, fNeck + 2.1 * THICK, new Cylinder)));
}
Solid *FlorenceFlask(double r, double fNeck, int nNecks)    /* :258-288 */
{
    const double f20 = 20.0 * D_PI / 180.0, fRad = nNecks > 1 ? 2.0 * D_PI / double(nNecks - 1) : 0.0;
    Solid *ps = new Union(new Scale(r, r, r, new Sphere), flask_neck(r, fNeck));
    for (int i = 0; i < nNecks - 1; i++) ps = new Union(ps, new Rotate(2.0 * f20, fRad * i + 0.5 * D_PI, 0.0, flask_neck(r, fNeck)));
    Solid *ps2 = new Union(new Scale(r - THICK, r - THICK, r - THICK, new Sphere), flask_bore(r, fNeck));
    for (int i = 0; i < nNecks - 1; i++) ps2 = new Union(ps2, new Rotate(2.0 * f20, fRad * i + 0.5 * D_PI, 0.0, flask_bore(r, fNeck)));
    return new BoundingBox(new Translate(0.0, r, 0.0, shell(ps, ps2)));
}
Solid *Stand()                                      /* :312-328 */
{
    const double f = 1.05 / sqrt(2.0);
    Solid *ps = new Torus(0.05 / f);
    ps = new Union(ps, new Translate(1.0, 0.0, -0.5, new Scale(0.05, 0.05, 0.5, new Cylinder)));
    ps = new Union(ps, new Translate(-0.5, +0.866, -0.5, new Scale(0.05, 0.05, 0.5, new Cylinder)));
    ps = new Union(ps, new Translate(-0.5, -0.866, -0.5, new Scale(0.05, 0.05, 0.5, new Cylinder)));
    ps = new Translate(0.0, f, 0.0, new Rotate(0.5 * D_PI, 0.0, 0.0, new Scale(f, f, f, ps)));
    return new BoundingBox(new Material(0.9, g_sCopper, MAT_COPPER, new Surface(SURF_POLISHED, 0, ps)));
}
Solid *Glass(Solid *ps)                             /* :345-350 */
{
    return new Material(0.33, REFRACT_PYREX, g_sWater, MAT_TRANSPARENT, ColoredGlass, new Surface(SURF_POLISHED, 0, ps));
}
Solid *metal_ball(double x, double y, double z, const Spectrum &c, int luster)
{
    return new Translate(x, y, z, new Material(0.9, c, luster, new Surface(SURF_POLISHED, Bumpy, new Sphere)));
}
Solid *Environment(Solid *ps)                       /* :368-385 */
{
    Spectrum sBlue = (g_sBlue + g_sEqualWhite) * 0.5;
    const double f20 = 20.0 * D_PI / 180.0;
    ps = new Union(ps, new Translate(0.0, 5.0, -10.0, new Scale(12.0, 1.0, 12.0, new Material(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube))));
    ps = new Union(ps, new Scale(100.0, 100.0, 100.0, new Material(0.9, sBlue, MAT_PLASTIC, new HollowSphere)));
    return new Rotate(-f20, 0.0, 0.0, ps);
}
}   // namespace lab

/* Laboratory(), LaboratoryAnimation.cpp:387-442, the frame at time t.  haze: the room filled with air whose Rayleigh
   coefficient is scaled a thousandfold (BASELINE configs[4] "with Atmosphere scattering"; the reference's Atmosphere is a
   stub, so this is documented new behaviour in the Propagator slot -- DESIGN.md 7b): a sphere of air larger than the
   sky dome, so no ray ever meets its boundary. */
static int build_S4(RayTracer &rt, double t, bool haze = false)
{
    using namespace lab;
    reset_globals();
    Spectrum sBlue = (g_sBlue + g_sEqualWhite) * 0.5, s6500 = BlackBody(6500.0);
    rt.plAmbient = new HavercosineLight(sBlue * 0.02, s6500 * 0.02);
    rt.plPointLights = new PointLight(Point3(9.0, -6.5, 12.0), s6500 * 10000.0);
    rt.psSample = new FastBlueSamplingTile;
    rt.pcCamera = new StereographicCamera(Point3(0.0, 0.0, 12.0));
    Solid *ps = MakeUnion(
        metal_ball(-6.5, 2.5, -8.0, sBlue, MAT_GOLD),
        metal_ball(-3.1, 2.5, -9.3, sBlue, MAT_IRON),
        metal_ball(-0.9, 2.5, -10.7, sBlue, MAT_ALUMINUM),
        metal_ball(+3.3, 2.5, -9.0, sBlue, MAT_OSMIUM),
        metal_ball(+6.4, 2.5, -10.9, sBlue, MAT_COPPER),
        Glass(new Translate(+2.5, 3.5, -0.9, new Rotate(0.5 * D_PI, 0.05 * D_PI, 0.0, StopCock(-20.0)))),
        Glass(new Translate(-4.5, 2.8 - 1.3, -2.0, FlorenceFlask(1.2, 0.8, 5))),
        new Translate(-4.5, 2.8, -2.0, Stand()),
        Glass(new Translate(+4.0, 0.5, -4.5, ErlenmeyerFlask(1.5, 0.5))),
        Glass(new Translate(+0.0, 0.5 * t, -0.5, new Rotate(0.0, -30.0 * D_PI / 180.0, 0.0, CondensorEnvelope(t)))),
        Glass(new Translate(+0.0, 0.5 * t, -0.5, SpiralTube(t))));
    rt.psModel = Environment(ps);
    if (haze)
        rt.psModel = new Union(rt.psModel, new Scale(150.0, 150.0, 150.0, new Material(DENSE_AIR, REFRACT_AIR, g_sEqualWhite * 1000.0, MAT_TRANSPARENT, Atmosphere, new Sphere)));
    return 0;
}

/* ---------------------------------------------------------------------- text front door ---- */
namespace {
struct Parser {
    const char *p;
    std::string err;
    RayTracer *rt;
    explicit Parser(const char *s, RayTracer *r) : p(s), rt(r) {}
    void ws() { while (*p && (isspace((unsigned char)*p))) p++; }
    bool fail(const std::string &m) { if (err.empty()) err = m + " near '" + std::string(p).substr(0, 24) + "'"; return false; }
    bool open() { ws(); if (*p == '(') { p++; return true; } return false; }
    bool close() { ws(); if (*p == ')') { p++; return true; } return fail("expected ')'"); }
    std::string word()
    {
        ws();
        std::string w;
        while (*p && !isspace((unsigned char)*p) && *p != '(' && *p != ')') w += *p++;
        return w;
    }
    double num()
    {
        ws();
        char *e;
        double v = strtod(p, &e);
        if (e == p) { fail("expected number"); return 0.0; }
        p = e;
        return v;
    }
    bool peek_open() { ws(); return *p == '('; }
    bool peek_close() { ws(); return *p == ')'; }

    Spectrum spectrum()
    {
        Spectrum s = g_sBlack;
        if (open()) {
            std::string w = word();
            if (w == "scale") { s = spectrum(); double f = num(); s = s * f; }
            else if (w == "add") { Spectrum a = spectrum(), b = spectrum(); s = a + b; }
            else if (w == "blackbody") s = BlackBody(num());
            else if (w == "rgb") { double r = num(), g = num(), b = num(); s = RGBtoSpectrum(DisplayRGB(r, g, b)); }
            else if (w == "rgb6500") { double r = num(), g = num(), b = num(); s = RGBtoSpectrum(DisplayRGB(r, g, b), 1); }
            else if (w == "gray") s = RGBtoSpectrum(num());
            else fail("unknown spectrum form " + w);
            close();
            return s;
        }
        std::string w = word();
        const Spectrum *t = rt_spectrum_by_name(w.c_str());
        if (!t) { fail("unknown spectrum " + w); return s; }
        return *t;
    }
    int luster(const std::string &w)
    {
        static const char *n[] = { "noshade", "plastic", "metal", "transparent", "copper", "gold", "iron", "aluminum", "osmium", "silver", "titanium" };
        for (int i = 0; i < 11; i++) if (w == n[i]) return i;
        fail("unknown luster " + w);
        return MAT_PLASTIC;
    }
    int rough(const std::string &w)
    {
        static const char *n[] = { "", "matte", "burnished", "smooth", "shiny", "polished", "wet" };
        for (int i = 1; i < 7; i++) if (w == n[i]) return i;
        fail("unknown roughness " + w);
        return SURF_MATTE;
    }
    TextureMap texture(const std::string &w)
    {
        if (w == "none") return 0;
        static const struct { const char *n; TextureMap f; } t[] = {
            { "bumpy", Bumpy }, { "microfacets", MicroFacets }, { "microfacets2", MicroFacets2 }, { "wrinkled", Wrinkled }, { "faceted20", Faceted20 },
            { "dented", Dented }, { "wavy", Wavy }, { "fibers", Fibers }, { "blistered", Blistered }, { "hammered", Hammered },
            { "tectonic", Tectonic }, { "quasimicrofacet", QuasiMicroFacet }, { "voronoibump", VoronoiBump },
            { "voronoiantibump", VoronoiAntiBump }, { "voronoicrinkled", VoronoiCrinkled },
            { "voronoi", Voronoi }, { "agate", Agate }, { "marble", Marble }, { "clouds", Clouds }, { "wood", Wood }, { "spots", Spots },
            { "veins", Veins }, { "seascape", SeaScape },
            { "earth", EarthTexture }, { "earth2", Earth2Texture }, { "mars", MarsTexture }, { "moon", MoonTexture }, { "venus", VenusTexture },
            { "mercury", MercuryTexture }, { "sky", SkyTexture } };
        for (size_t i = 0; i < sizeof t / sizeof t[0]; i++) if (w == t[i].n) return t[i].f;
        fail("unknown texture " + w);
        return 0;
    }
    Propagator propagator(const std::string &w)
    {
        if (w == "none") return 0;
        if (w == "coloredglass") return ColoredGlass;
        if (w == "foggy") return Foggy;
        if (w == "atmosphere") return Atmosphere;
        fail("unknown propagator " + w);
        return 0;
    }
    int children(Solid **v, int maxn)
    {
        int n = 0;
        while (peek_open() && n < maxn) v[n++] = solid();
        return n;
    }
    Solid *solid()
    {
        if (!open()) { fail("expected '('"); return 0; }
        std::string w = word();
        Solid *r = 0;
        if (w == "sphere") r = new Sphere;
        else if (w == "hollowsphere") r = new HollowSphere;
        else if (w == "cylinder") r = new Cylinder;
        else if (w == "cone") r = new Cone;
        else if (w == "conefx") r = new ConeFX;
        else if (w == "coneunit") r = new ConeUnit;
        else if (w == "paraboloid") r = new Paraboloid;
        else if (w == "hyperboloid1") r = new Hyperboloid1(num());
        else if (w == "hyperboloid2") r = new Hyperboloid2(num());
        else if (w == "conehypercylinder") r = new ConeHyperCylinder(num());
        else if (w == "quadric") { double P = num(), Q = num(), R = num(); int c = (int)num(); r = new Quadric(P, Q, R, c); }
        else if (w == "cube") r = new Cube;
        else if (w == "tetrahedron") r = new Tetrahedron;
        else if (w == "octahedron") r = new Octahedron;
        else if (w == "dodecahedron") r = new Dodecahedron;
        else if (w == "icosahedron") r = new Icosahedron;
        else if (w == "prism") r = new Prism((int)num());
        else if (w == "fibonacci") r = new Fibonacci((int)num());
        else if (w == "slab") r = new Slab;
        else if (w == "torus") r = new Torus(num());
        else if (w == "null") r = new NullSolid;
        else if (w == "meshsphere") { int a = (int)num(), b = (int)num(); r = MeshSphere(a, b); }
        else if (w == "meshblob") { int a = (int)num(), b = (int)num(); double c = num(); int k = (int)num(); r = MeshBlob(a, b, c, k); }
        else if (w == "meshtorus") { int a = (int)num(), b = (int)num(); double c = num(); r = MeshTorus(a, b, c); }
        else if (w == "meshicosahedron") r = MeshIcosahedron();
        else if (w == "mesh") {                 /* (mesh nVertices nTriangles  x y z ...  a b c ...) with 0-based indices */
            int nv = (int)num(), nt = (int)num();
            MeshSolid *m = new MeshSolid;
            for (int k = 0; k < nv && err.empty(); k++) { double x = num(), y = num(), z = num(); m->AddVertex(x, y, z); }
            for (int k = 0; k < nt && err.empty(); k++) {
                int a = (int)num(), b = (int)num(), c = (int)num();
                if (a < 0 || b < 0 || c < 0 || a >= nv || b >= nv || c >= nv) { fail("mesh index out of range"); break; }
                m->AddTriangle(a, b, c);
            }
            r = m;
        }
        else if (w == "objfile") {              /* (objfile path): the grammar of LoadObjFile, ObjectFormat.cpp:31-75 */
            std::string path = word();
            MeshSolid *m = new MeshSolid;
            if (!m->LoadObjFile(path.c_str())) fail("cannot read OBJ file " + path);
            r = m;
        }
        else if (w == "translate") { double x = num(), y = num(), z = num(); r = new Translate(x, y, z, solid()); }
        else if (w == "scale") { double x = num(), y = num(), z = num(); r = new Scale(x, y, z, solid()); }
        else if (w == "rotate") { double x = num(), y = num(), z = num(); r = new Rotate(x, y, z, solid()); }
        else if (w == "rotateaxis") { double x = num(), y = num(), z = num(), a = num(); r = new Rotate(Vector3(x, y, z), a, solid()); }
        else if (w == "rotatefromto") { double a = num(), b = num(), c = num(), x = num(), y = num(), z = num(); r = new Rotate(Vector3(a, b, c), Vector3(x, y, z), solid()); }
        else if (w == "union") { Solid *a = solid(), *b = solid(); r = new Union(a, b); }
        else if (w == "topunion") { Solid *a = solid(), *b = solid(); r = new TopUnion(a, b); }
        else if (w == "intersection") { Solid *a = solid(), *b = solid(); r = new Intersection(a, b); }
        else if (w == "difference") { Solid *a = solid(), *b = solid(); r = new Difference(a, b); }
        else if (w == "makeunion" || w == "maketopunion" || w == "makeintersection") {
            Solid *v[64];
            int n = children(v, 64);
            r = w == "makeintersection" ? rt_make_intersection(v, n) : rt_make_union(v, n, w == "maketopunion");
        }
        else if (w == "angularclip") { double a = num(), b = num(); r = new AngularClip(a, b, solid()); }
        else if (w == "boundingbox") r = new BoundingBox(solid());
        else if (w == "material") {
            /* (material density refr|auto color luster texture propagator child) */
            double d = num();
            ws();
            bool autoN = false;
            double n = 0.0;
            if (isalpha((unsigned char)*p)) { std::string a = word(); autoN = (a == "auto"); if (!autoN) fail("expected refractive index or auto"); }
            else n = num();
            Spectrum c = spectrum();
            int l = luster(word());
            TextureMap t = texture(word());
            Propagator pr = propagator(word());
            Solid *ch = solid();
            Material *m = autoN ? new Material(d, c, l, t, ch) : new Material(d, n, c, l, pr, ch);
            if (autoN) m->Propagate = pr; else m->Texture = t;
            r = m;
        }
        else if (w == "surface") { int ro = rough(word()); TextureMap t = texture(word()); r = new Surface(ro, t, solid()); }
        else if (w == "cuboid") { double a = num(), x = num(), y = num(), z = num(); r = Cuboid(a, x, y, z); }
        else if (w == "cylinderfromto") { double a = num(), b = num(), c = num(), x = num(), y = num(), z = num(), rr = num(); r = CylinderFromTo(Point3(a, b, c), Point3(x, y, z), rr); }
        else if (w == "tetrahedroncage") { double a = num(), b = num(); r = TetrahedronCage(a, b); }
        else if (w == "hexahedroncage") { double a = num(), b = num(); r = HexahedronCage(a, b); }
        else if (w == "octahedroncage") { double a = num(), b = num(); r = OctahedronCage(a, b); }
        else if (w == "icosahedroncage") { double a = num(), b = num(); r = IcosahedronCage(a, b); }
        else if (w == "dodecahedroncage") { double a = num(), b = num(); r = DodecahedronCage(a, b); }
        else if (w == "diskoid") r = Diskoid(num());
        else if (w == "rodoid") r = Rodoid(num());
        else if (w == "squareoid") r = Squareoid(num());
        else if (w == "knoboid") { double a = num(), z = num(); r = Knoboid(a, z); }
        else if (w == "knoboid2") { double a = num(), z = num(); r = Knoboid2(a, z); }
        else if (w == "hyperboloidoid") { double a = num(), b = num(), z = num(); r = Hyperboloidoid(a, b, z); }
        else if (w == "slaboid") { double x = num(), y = num(), z = num(), a = num(); r = Slaboid(x, y, z, a); }
        else fail("unknown solid " + w);
        close();
        return r;
    }
    bool scene()
    {
        if (!open() || word() != "scene") return fail("expected (scene ...)");
        while (open()) {
            std::string w = word();
            if (w == "model") rt->psModel = solid();
            else if (w == "ambient") rt->plAmbient = new AmbientLight(spectrum());
            else if (w == "pointlight") {
                double x = num(), y = num(), z = num();
                Spectrum s = spectrum();
                PointLight *pl = new PointLight(Point3(x, y, z), s, rt->plPointLights);
                if (!peek_close() && word() == "noshadow") pl->nNoShadow = 1;
                rt->plPointLights = pl;
            }
            else if (w == "arealight") {
                double d = num(), x = num(), y = num(), z = num();
                Spectrum s = spectrum();
                PointLight *pl = new AreaLight(d, Point3(x, y, z), s, rt->plPointLights);
                if (!peek_close() && word() == "noshadow") pl->nNoShadow = 1;
                rt->plPointLights = pl;
            }
            else if (w == "camera") {
                std::string k = word();
                if (k == "aperture") { double fs = num(), fd = num(), x = num(), y = num(), z = num(), f = num(); rt->pcCamera = new ApertureCamera(fs, fd, Point3(x, y, z), f); }
                else {
                    double x = num(), y = num(), z = num(), f = num();
                    if (k == "stereographic") rt->pcCamera = new StereographicCamera(Point3(x, y, z), f);
                    else if (k == "perspective") rt->pcCamera = new PerspectiveCamera(Point3(x, y, z), f);
                    else if (k == "orthographic") rt->pcCamera = new OrthographicCamera(Point3(x, y, z), f);
                    else fail("unknown camera " + k);
                }
            }
            else if (w == "sampler") {
                std::string k = word();
                if (k == "test") rt->psSample = new TestSamplingTile;
                else if (k == "debug") rt->psSample = new DebugSamplingTile;
                else if (k == "jitter") rt->psSample = new JitterSamplingTile;
                else if (k == "bluenoise") rt->psSample = new BlueNoiseSamplingTile;
                else if (k == "adaptive") { rt->psSample = new AdaptiveBlueSamplingTile; if (!peek_close()) rt->psSample->QIndex1((int)num()); }
                else if (k == "fastblue") { rt->psSample = new FastBlueSamplingTile; if (!peek_close()) rt->psSample->QIndex1((int)num()); }
                else fail("unknown sampler " + k);
            }
            else if (w == "aux") { g_uCamera = num(); g_vCamera = num(); g_uLight = num(); g_vLight = num(); g_uMicroFacet = num(); g_vMicroFacet = num(); }
            else fail("unknown scene item " + w);
            if (!err.empty() || !close()) return false;
        }
        return close() && err.empty();
    }
};
}

int rt_build_conformance_scene(const char *spec, RayTracer &rt, std::string *err);     /* rt_scenes_conformance.cpp */

/* ------------------------------------------------------------------------ C entry points ---- */
struct rtf_scene {
    RayTracer   rt;
    RtSceneBlob blob;
};
static char s_scene_err[512];

extern "C" const char *rtf_last_error() { return s_scene_err[0] ? s_scene_err : rt_frontend_last_error(); }

/* spec: "S0" | "S1" | "S2" | "S4" | "S4@t" | a conformance scene name (rt_scenes_conformance.cpp) | "(scene ...)".  sampler: "", "test", "jitter" or "fastblue" overrides the scene's tile.
   optimize != 0 calls psModel->Optimize() before flattening (what every reference driver does). */
extern "C" rtf_scene *rtf_scene_build(const char *spec, const char *sampler, int optimize)
{
    rt_frontend_reset_ids();
    rtf_scene *s = new rtf_scene;
    s_scene_err[0] = 0;
    std::string name(spec);
    int rc = 0;
    if (name == "S0") rc = build_S0(s->rt);
    else if (name == "S1") rc = build_S1(s->rt);
    else if (name == "S2") rc = build_S2(s->rt);
    else if (name == "S4") rc = build_S4(s->rt, 4.5);
    else if (name.compare(0, 3, "S4@") == 0) rc = build_S4(s->rt, atof(name.c_str() + 3));
    else if (name == "S4atm") rc = build_S4(s->rt, 4.5, true);
    else if (name.compare(0, 6, "S4atm@") == 0) rc = build_S4(s->rt, atof(name.c_str() + 6), true);
    else if (name[0] != '(') {
        std::string why;
        int k = rt_build_conformance_scene(spec, s->rt, &why);
        if (k == 0) why = "unknown scene name " + name;
        if (k <= 0) { snprintf(s_scene_err, sizeof s_scene_err, "%s", why.c_str()); delete s; return 0; }
    }
    else {
        reset_globals();
        Parser ps(spec, &s->rt);
        if (!ps.scene()) { snprintf(s_scene_err, sizeof s_scene_err, "scene text: %s", ps.err.c_str()); delete s; return 0; }
    }
    if (rc) { delete s; return 0; }
    if (sampler && !strcmp(sampler, "test")) s->rt.psSample = new TestSamplingTile;
    else if (sampler && !strcmp(sampler, "debug")) s->rt.psSample = new DebugSamplingTile;
    else if (sampler && !strcmp(sampler, "jitter")) s->rt.psSample = new JitterSamplingTile;
    else if (sampler && !strcmp(sampler, "bluenoise")) s->rt.psSample = new BlueNoiseSamplingTile;
    else if (sampler && !strcmp(sampler, "adaptive")) s->rt.psSample = new AdaptiveBlueSamplingTile;
    else if (sampler && !strcmp(sampler, "fastblue") && !(s->rt.psSample && s->rt.psSample->DeviceSampler() == RT_SAMPLER_FASTBLUE)) s->rt.psSample = new FastBlueSamplingTile;
    if (!s->rt.psSample) s->rt.psSample = new TestSamplingTile;
    if (!s->rt.pcCamera) s->rt.pcCamera = new StereographicCamera(Point3(0.0, 0.0, 8.0), 70.0);
    if (optimize && s->rt.psModel) s->rt.psModel = s->rt.psModel->Optimize();
    if (rt_flatten(s->rt, s->blob, s_scene_err, sizeof s_scene_err)) { delete s; return 0; }
    return s;
}
extern "C" const RtSceneDesc *rtf_scene_desc(rtf_scene *s) { return &s->blob.desc; }
extern "C" void rtf_scene_free(rtf_scene *s) { delete s; }

/* RayTracer::Render through the front end (scene -> flatten -> C ABI -> planes) */
extern "C" int rtf_scene_render(rtf_scene *s, int width, int height, double spp, float *planes, RtStats *stats)
{
    Image im;
    im.NewImage(width, height, 3);
    int n = s->rt.Render(im, spp);
    if (n < 0) { snprintf(s_scene_err, sizeof s_scene_err, "%s", rt_frontend_last_error()); return -1; }
    memcpy(planes, im.m_pfImage, sizeof(float) * 3 * (size_t)width * height);
    if (stats) *stats = s->rt.stats;
    return n;
}

/* GetExtent() of the scene root and of a scene given as text: 6 doubles */
extern "C" int rtf_extent(const char *solid_text, double *out6)
{
    RayTracer rt;
    Parser ps(solid_text, &rt);
    Solid *sd = ps.solid();
    if (!sd || !ps.err.empty()) { snprintf(s_scene_err, sizeof s_scene_err, "solid text: %s", ps.err.c_str()); return -1; }
    Extent e = sd->GetExtent();
    out6[0] = e.pMin.x; out6[1] = e.pMin.y; out6[2] = e.pMin.z; out6[3] = e.pMax.x; out6[4] = e.pMax.y; out6[5] = e.pMax.z;
    return 0;
}

/* ClipPlanes(Extent(half), table, n, from_origin) for one of the polyhedron normal tables ("tetrahedron", "cube", "octahedron",
   "dodecahedron", "icosahedron"): TestClipPlanes, Test.cpp:1822-1830 */
extern "C" int rtf_clip_planes(double half, const char *table, double from_origin, double *out6)
{
    const CoVector3 *t = 0;
    int n = 0;
    const std::string w(table ? table : "");
    if (w == "tetrahedron") { t = rgvTetrahedron; n = 4; } else if (w == "cube") { t = rgvCube; n = 8; } else if (w == "octahedron") { t = rgvOctahedron; n = 6; }
    else if (w == "dodecahedron") { t = rgvDodecahedron; n = 20; } else if (w == "icosahedron") { t = rgvIcosahedron; n = 12; }
    else { snprintf(s_scene_err, sizeof s_scene_err, "unknown normal table %s", w.c_str()); return -1; }
    const Extent e = ClipPlanes(Extent(half), t, n, from_origin);
    out6[0] = e.pMin.x; out6[1] = e.pMin.y; out6[2] = e.pMin.z; out6[3] = e.pMax.x; out6[4] = e.pMax.y; out6[5] = e.pMax.z;
    return 0;
}

extern "C" void rtf_random_double(unsigned x, unsigned c, int n, double *out)
{
    ML_InitializeRandom(x, c);
    for (int i = 0; i < n; i++) out[i] = RandomDouble();
}

/* Video::NewVideo / WriteFrame / Close over nframes images float[nframes][3][H][W]; returns frames written */
extern "C" int rtf_write_video(const char *path, const float *frames, int nframes, int width, int height, float gamma)
{
    Image im;
    Video vd;
    im.NewImage(width, height, 3);
    vd.SetGamma(gamma);
    if (!vd.NewVideo(path, im)) { snprintf(s_scene_err, sizeof s_scene_err, "NewVideo: %s", rt_frontend_last_error()); return -1; }
    int n = 0;
    for (; n < nframes; n++) {
        memcpy(im.m_pfImage, frames + (size_t)n * 3 * width * height, sizeof(float) * 3 * (size_t)width * height);
        if (!vd.WriteFrame(im)) { snprintf(s_scene_err, sizeof s_scene_err, "WriteFrame: %s", rt_frontend_last_error()); break; }
    }
    vd.Close();
    return n;
}

/* the directory of the planet image maps (SetAssetPath) */
extern "C" void rtf_set_asset_path(const char *dir) { SetAssetPath(dir); }

/* Image::ReadBMP + Image::Get through the front end, for the tests: read a file, report width, height, channels, copy the planes out,
   and read single pixels through the wrap modes */
extern "C" void *rtf_image_read_bmp(const char *path, double gamma, int *whc3)
{
    Image *im = new Image;
    if (!im->ReadBMP(path, gamma)) { snprintf(s_scene_err, sizeof s_scene_err, "ReadBMP failed: %s", path); delete im; return 0; }
    whc3[0] = im->m_nWidth; whc3[1] = im->m_nHeight; whc3[2] = im->m_nChannels;
    return im;
}
extern "C" void rtf_image_planes(void *h, float *out) { Image *im = (Image *)h; memcpy(out, im->m_pfImage, sizeof(float) * (size_t)im->m_nWidth * im->m_nHeight * im->m_nChannels); }
extern "C" float rtf_image_get(void *h, int wrap_x, int wrap_y, int i, int j, int k) { Image *im = (Image *)h; im->m_bWrapX = wrap_x; im->m_bWrapY = wrap_y; return im->Get(i, j, k); }
extern "C" void rtf_image_free(void *h) { delete (Image *)h; }
