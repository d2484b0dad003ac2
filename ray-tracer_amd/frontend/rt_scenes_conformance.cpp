/*
 * rt_scenes_conformance.cpp -- the reference's remaining scene scripts as named scenes of the front end (SURVEY.md
 * section 8(f) item 4): the same object graphs, written against the same scene API, rendered by the same path.  They are
 * the conformance suite of tests/test_conformance_scenes.py (GPU against the CPU oracle on the same description) and the
 * cheapest way to meet every limit of the flattener (nested regions and boxes, nested affines, hit-list depth).
 *
 * Names (argument after '@', default in brackets):
 *   Model2..Model9, ModelA..ModelV [@angle 30]   TestScene.cpp:79-1145  (MakeModel2 ... MakeModelV; L4, L5 and U use the
 *                                                planet image maps, which the reference does not ship: they need an
 *                                                asset directory -- SetAssetPath() / RT_AMD_ASSETS -- and fail without)
 *   Fib[@N 200]          MakeModelFib (:1147-1174): Fibonacci polyhedron with two SolidText labels
 *   Ruby[@angle 20]      MakeModelRuby (:1194-1224): bumpy metal dodecahedron cages around hollow gem stones
 *   Letters              MakeModelLetters (:1226-1235): SolidText("Hello World!")
 *   SetOperations        MakeModelSetOperations (:1237-1278): union / intersection / difference with missing operands
 *   HyperTest[@angle]    MakeModelHyperTest (:1280-1289)
 *   Lorenz[@steps 200]   InitializeLorenz + MakeModelLorenz (:1300-1358), the driver's far light
 *   TestTop              MakeModel_TestTop (:1428-1454): boxed TopUnions of PRNG-placed ellipsoids
 *   Venera[@t 0]         Venera() frame t (Venera.cpp:144-670): the lander (landing ring with flanges and bolts, hull,
 *                        struts, two cameras, colour chips) in EnvironmentLab
 *   VeneraVenus[@t 0]    the same lander in EnvironmentVenus (:237-259)
 *   VeneraAtmosphere[@t] VeneraVenus under Venus's air: the Atmosphere propagator (new behaviour, not a reference scene)
 *   Megacycles[@t 0]     Megacycles() frame t (Megacycles.cpp:8-349): 189 robot unicyclists on a second-order Peano curve
 *   Unicyclist[@t 0]     one rider of that scene, close up
 * Lights and cameras: the drivers' own where a driver exists (Venera, Megacycles, Lorenz); the TestScene models get
 * MakeLights (:1456-1470) and the driver's StereographicCamera((0,0,8), 70) (:1593), moved where the model's comment says so.
 */
#include <stdlib.h>
#include <string>
#include "rt_frontend.h"

namespace {
inline double rad(double deg) { return deg * D_PI / 180.0; }
inline Solid *at(double x, double y, double z, Solid *ps) { return new Translate(x, y, z, ps); }
inline Solid *size(double x, double y, double z, Solid *ps) { return new Scale(x, y, z, ps); }
inline Solid *size(double s, Solid *ps) { return new Scale(s, s, s, ps); }
inline Solid *turn(double x, double y, double z, Solid *ps) { return new Rotate(x, y, z, ps); }
inline Solid *mat(double d, const Spectrum &c, int luster, Solid *ps) { return new Material(d, c, luster, ps); }
inline Solid *mat(double d, double n, const Spectrum &c, int luster, Propagator pr, Solid *ps) { return new Material(d, n, c, luster, pr, ps); }
inline Solid *tex(double d, const Spectrum &c, int luster, TextureMap t, Solid *ps) { return new Material(d, c, luster, t, ps); }
inline Solid *surf(int rough, TextureMap t, Solid *ps) { return new Surface(rough, t, ps); }
inline Solid *surf(int rough, Solid *ps) { return new Surface(rough, ps); }
inline Solid *ball(double r) { return size(r, new Sphere); }
inline Solid *floor_slab(double y, double z = -10.0, double w = 12.0) { return at(0.0, y, z, size(w, 1.0, w, new Cube)); }
inline Solid *join(Solid *a, Solid *b) { return new Union(a, b); }

Spectrum s_rgs[20];
Spectrum pale(const Spectrum &c, const Spectrum &white) { return c * 0.5 + white * 0.5; }       /* "sBlue = g_sBlue * 0.5 + g_sWhite * 0.5" */
Spectrum mix(const Spectrum &a, const Spectrum &b) { return (a + b) * 0.5; }                    /* "(g_sBlue + g_sEqualWhite) * 0.5" */

Solid *Glass(Solid *ps) { return mat(0.33, REFRACT_PYREX, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, ps)); }           /* TestScene.cpp:16-21 */
Solid *Plastic(Solid *ps) { return mat(0.33, REFRACT_WATER, mix(g_sBlue, g_sEqualWhite), MAT_PLASTIC, 0, surf(SURF_POLISHED, 0, ps)); }   /* :23-31 */
Solid *Plastic2(Solid *ps) { return mat(0.33, REFRACT_WATER, mix(g_sBlue, g_sEqualWhite), MAT_PLASTIC, 0, surf(SURF_SHINY, 0, ps)); }     /* :33-41 */

/* ------------------------------------------------------------------------------------------- TestScene.cpp models ---- */
Solid *Model2(double a)         /* :79-108 ten tori, every roughness x {plastic, metal} */
{
    Solid *all = 0;
    int i = 0;
    double x = -6.0;
    for (int rough = SURF_MATTE; rough <= SURF_POLISHED; rough++, x += 3.0) {
        double y = -2.0;
        for (int luster = MAT_PLASTIC; luster <= MAT_METAL; luster++, y += 4.0, i++) {
            Solid *one = surf(rough, mat(0.9, s_rgs[i], luster, at(x, y, 0.0, turn(0.0, rad(a), 0.0, new Torus(0.5)))));
            all = all ? join(one, all) : one;
        }
    }
    return join(all, floor_slab(5.25));
}

Solid *bumped_sphere(double a, const Spectrum &c, int rough, TextureMap t) { return turn(0.0, rad(a), 0.0, mat(0.9, c, MAT_PLASTIC, surf(rough, t, new Sphere))); }

Solid *Model3(double a)         /* :110-132 */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    Solid *all = at(-3.0, 0.0, 0.0, bumped_sphere(a, blue, SURF_SHINY, Bumpy));
    all = join(at(0.0, 0.0, 0.0, bumped_sphere(a, blue, SURF_SHINY, Wrinkled)), all);
    return join(at(3.0, 0.0, 0.0, bumped_sphere(a, blue, SURF_SHINY, Dented)), all);
}

Solid *Model4(double x)         /* :137-164 a bumpy ball moving into and out of a bite taken from a matte one */
{
    Spectrum blue = pale(g_sBlue, g_sWhite), orange = g_sRed * 0.5 + g_sGreen * 0.25 + g_sWhite * 0.25;
    Solid *base = at(-1.5, 0.0, 0.0, mat(0.9, orange, MAT_PLASTIC, surf(SURF_MATTE, 0, new Sphere)));
    if (x > 60.0) {
        x = -1.0 + 2.5 * (x - 60.0) / 60.0;
        base = new Difference(base, at(-1.0, 0.0, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_BURNISHED, Bumpy, new Sphere))));
    } else
        x = 1.5 - 2.5 * x / 60.0;
    return join(base, at(x, 0.0, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_BURNISHED, Bumpy, new Sphere))));
}

Solid *Model5(double a)         /* :166-194 gold, bumpy copper, transparent and plastic polished spheres */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    Solid *all = at(-3.0, 0.0, -3.0, turn(0.0, 0.0, 0.0, mat(0.9, g_sGold, MAT_METAL, surf(SURF_POLISHED, 0, new Sphere))));
    all = join(at(3.0, 0.0, -3.0, turn(0.0, 0.0, 0.0, mat(0.9, g_sCopper, MAT_METAL, surf(SURF_POLISHED, Bumpy, new Sphere)))), all);
    all = join(at(0.0, 0.0, 0.0, turn(0.0, 0.0, 0.0, mat(0.9, blue, MAT_TRANSPARENT, surf(SURF_POLISHED, 0, new Sphere)))), all);
    all = turn(0.0, rad(a), 0.0, join(at(0.0, 0.0, -3.0, turn(0.0, 0.0, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_POLISHED, 0, new Sphere)))), all));
    return join(all, floor_slab(3.0));
}

/* :196-233 (cubes) and :403-442 (17-sided prisms): three water blocks holding a copper, a Pyrex and an air ball */
Solid *water_blocks(double a, bool prisms)
{
    struct Block { static Solid *make(bool p) { return p ? (Solid *)new Rotate(0.5 * D_PI, 0.0, 0.0, new Prism(17)) : (Solid *)new Cube; } };
    double r = rad(a);
    Solid *all = at(-3.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.33, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Block::make(prisms)))));
    all = join(all, at(0.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.9, g_sCopper, MAT_METAL, surf(SURF_POLISHED, 0, ball(0.5))))));
    all = join(all, at(0.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.33, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Block::make(prisms))))));
    all = join(at(-3.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.5, REFRACT_PYREX, g_sEqualWhite, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, ball(0.5))))), all);
    all = join(all, at(3.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.33, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Block::make(prisms))))));
    all = join(all, at(3.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.9, REFRACT_AIR, g_sEqualWhite, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, ball(0.5))))));
    return turn(rad(-20.0), 0.0, 0.0, join(all, floor_slab(3.0)));
}

Solid *Model7(double a)         /* :235-244 */
{
    return join(surf(SURF_POLISHED, mat(0.5, g_sWhite, MAT_TRANSPARENT, turn(0.0, rad(a), 0.0, new Cube))), floor_slab(3.0));
}

Solid *Model8(double a)         /* :246-284 six spheres, the middle front one bumpy water */
{
    Spectrum blue = pale(g_sBlue, g_sEqualWhite);
    double r = rad(a);
    Solid *all = at(-3.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_SHINY, 0, new Sphere))));
    all = join(at(0.0, 0.0, 0.0, turn(0.0, r, 0.0, mat(0.9, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, Bumpy, size(1.0, 1.0, 1.0, new Sphere))))), all);
    static const double pos[4][2] = { { 3.0, 0.0 }, { -3.0, -6.0 }, { 0.0, -6.0 }, { 3.0, -6.0 } };
    for (int k = 0; k < 4; k++)
        all = join(at(pos[k][0], 0.0, pos[k][1], turn(0.0, r, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_SHINY, 0, new Sphere)))), all);
    return turn(rad(-10.0), 0.0, 0.0, join(all, floor_slab(3.0)));
}

Solid *Model9(double)           /* :286-314 a tumbler with two lumps of ice, cut open by a cube */
{
    Solid *tumbler = mat(DENSE_PYREX, g_sBlack, MAT_TRANSPARENT, surf(SURF_POLISHED, 0,
        join(new Difference(new Cylinder, at(0.0, 0.0, 0.2, size(0.9, 0.9, 1.0, new Cylinder))), at(0.0, 0.0, 1.0, new Torus(0.06)))));
    Solid *ice = mat(DENSE_ICE, REFRACT_ICE, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, Bumpy,
        join(at(0.1, 0.1, 0.4, ball(0.5)), at(-0.1, 0.0, -0.4, ball(0.5)))));
    Solid *all = join(tumbler, at(0.0, 0.0, 0.0, ice));
    all = size(0.1, turn(rad(90.0), 0.0, 0.0, all));
    all = new Difference(all, at(0.0, 0.0, 1.0, surf(SURF_POLISHED, 0, new Cube)));
    return join(all, floor_slab(3.0));
}

inline double lane(double a, double phase) { return fmod(a / 10 + phase, 8.0) - 4.0; }

Solid *ModelA(double a)         /* :316-342 four balls sliding through a Pyrex and a water cube */
{
    Solid *all = floor_slab(3.0);
    all = join(at(2.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, new Cube))), all);
    all = join(at(-2.0, 0.0, 0.0, mat(0.5, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, new Cube))), all);
    all = join(at(lane(a, 0.0), 0.0, 0.0, mat(0.9, g_sCopper, MAT_METAL, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    all = join(at(lane(a, 2.0), 0.0, 0.0, mat(0.9, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    all = join(at(lane(a, 4.0), 0.0, 0.0, mat(0.9, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    all = join(at(lane(a, 6.0), 0.0, 0.0, mat(0.9, REFRACT_AIR, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    return turn(rad(-40.0), rad(20.0), 0.0, all);
}

Solid *ModelB(double) { return at(3.0, 0.0, 0.0, new Sphere); }       /* :344-353 */

/* the retort of MakeModelC / MakeModelM (:358-383, :865-880): (bulb + spout + neck + lip) minus the same shapes shrunk, water glass, boxed */
Solid *retort(double yaw)
{
    const double f20 = rad(20.0), f90 = rad(90.0);
    Solid *spout = new Intersection(new ConeFX, at(0.0, 0.0, 1.0, new Slab));
    Solid *outer = join(new Sphere,
        join(turn(f20, 0.0, 0.0, at(0.0, -0.6, -4.0, size(0.4, 0.4, 2.0, spout))),
            join(turn(f90, 0.0, 0.0, at(0.0, 0.0, 1.0, size(0.3, 0.3, 0.4, new Cylinder))),
                 turn(f90, 0.0, 0.0, at(0.0, 0.0, 1.4, size(0.35, 0.35, 0.1, new Sphere))))));
    spout = new Intersection(new ConeFX, at(0.0, 0.0, 1.0, new Slab));
    Solid *inner = join(size(0.9, new Sphere),
        join(turn(f20, 0.0, 0.0, at(0.0, -0.6, -4.0, size(0.3, 0.3, 2.0, spout))),
            join(turn(f90, 0.0, 0.0, at(0.0, 0.0, 1.0, size(0.2, 0.2, 0.6, new Cylinder))),
                 turn(f20, 0.0, 0.0, at(0.0, -0.6, -6.0, size(1.0, 1.0, 3.0, new Cube))))));
    Solid *ps = turn(0.0, yaw, 0.0, new Difference(outer, inner));
    return new BoundingBox(mat(0.33, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, ps)));
}
Solid *tripod()                 /* :385-393: a torus ring on three legs, polished copper, boxed */
{
    const double f = 1.05 / sqrt(2.0);
    Solid *ps = new Torus(0.05 / f);
    ps = join(ps, at(1.0, 0.0, -0.5, size(0.05, 0.05, 0.5, new Cylinder)));
    ps = join(ps, at(-0.5, +0.866, -0.5, size(0.05, 0.05, 0.5, new Cylinder)));
    ps = join(ps, at(-0.5, -0.866, -0.5, size(0.05, 0.05, 0.5, new Cylinder)));
    ps = at(0.0, f, 0.0, turn(0.5 * D_PI, 0.0, 0.0, size(f, ps)));
    return new BoundingBox(mat(0.9, g_sCopper, MAT_METAL, surf(SURF_POLISHED, 0, ps)));
}

Solid *ModelC(double a)         /* :358-401 */
{
    Solid *glass = retort(rad(a - 90.0));
    Solid *ps = join(tripod(), glass);
    ps = join(ps, floor_slab(2.5));
    ps = turn(-rad(20.0) * 0.7, 0.0, 0.0, ps);
    return join(ps, size(100.0, new HollowSphere));
}

Solid *ModelE(double a)         /* :444-463 */
{
    Spectrum blue = pale(g_sBlue, g_sEqualWhite);
    Solid *ps = mat(0.5, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, new Cube));
    ps = at(0.0, 1.5, 0.0, size(1.9, 1.0, 1.9, ps));
    ps = join(ps, at(3.0, 0.0, -3.0, turn(0.0, rad(a), 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_SHINY, 0, new Sphere)))));
    ps = join(ps, floor_slab(1.9));
    return turn(rad(-20.0), 0.0, 0.0, ps);
}

Solid *ChainLink(double a)      /* :465-480 two half tori joined by two straight rods, boxed */
{
    const double r = 0.5, g = 0.1;
    Solid *h1 = new Difference(new Torus(r), at(+0.5 * (1.0 + r), 0.0, 0.0, size(0.5 * (1.0 + r), (1.01 + r), r + 0.01, new Cube)));
    Solid *h2 = new Difference(new Torus(r), at(-0.5 * (1.0 + r), 0.0, 0.0, size(0.5 * (1.0 + r), (1.01 + r), r + 0.01, new Cube)));
    Solid *ps = join(at(-r - g, 0.0, 0.0, h1), at(+r + g, 0.0, 0.0, h2));
    ps = join(at(0.0, +1.0, 0.0, size(r + 0.000001 + g, r, r, turn(0.0, 0.5 * D_PI, 0.0, new Cylinder))), ps);
    ps = join(at(0.0, -1.0, 0.0, size(r + 0.000001 + g, r, r, turn(0.0, 0.5 * D_PI, 0.0, new Cylinder))), ps);
    return turn(rad(a), 0.0, 0.0, new BoundingBox(ps));
}

Solid *ModelF(double)           /* :482-501 */
{
    const double d = 1.0 + 0.5 + 0.5 + 0.1 + 0.1;
    Solid *ps = ChainLink(0.0);
    ps = join(ps, at(d, 0.0, 0.0, ChainLink(90.0)));
    ps = join(ps, at(2.0 * d, 0.0, 0.0, ChainLink(0.0)));
    ps = join(ps, at(-d, 0.0, 0.0, ChainLink(90.0)));
    ps = join(ps, at(0.0, 3, -10.0, size(12.0, 1.0, 12.0, new Cube)));
    return turn(rad(-20.0), 0.0, 0.0, ps);
}

/* :503-535 (G) and :672-709 (K: a tiny ball at x in front, the plastic ball becomes water, no yaw) */
Solid *cuboids_and_balls(double a, bool k, double x)
{
    const double r = 0.075;
    Spectrum emerald = g_sEmerald * 0.01, blue = mix(g_sBlue, g_sEqualWhite);
    Solid *all = floor_slab(3.0);
    if (k) all = join(at(x, 0.0, -0.05, ball(0.015)), all);
    all = join(at(2.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, Cuboid(r)))), all);
    all = join(at(-2.0, 0.0, 0.0, mat(0.5, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Cuboid(r)))), all);
    all = join(at(lane(a, 0.0), 0.0, 0.0, mat(0.9, REFRACT_PYREX, g_sCopper, MAT_GOLD, 0, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    all = join(at(lane(a, 2.0), 0.0, 0.0, mat(0.9, REFRACT_PYREX, emerald, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, Bumpy, ball(0.5)))), all);
    if (k) all = join(at(lane(a, 4.0), 0.0, 0.0, mat(0.9, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    else   all = join(at(lane(a, 4.0), 0.0, 0.0, mat(0.9, REFRACT_WATER, blue, MAT_PLASTIC, ColoredGlass, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    all = join(at(lane(a, 6.0), 0.0, 0.0, mat(0.9, REFRACT_AIR, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, ball(0.5)))), all);
    return turn(rad(-40.0), k ? 0.0 : rad(20.0), 0.0, all);
}

Solid *ModelG2(double)          /* :537-557 */
{
    Solid *all = floor_slab(3.0);
    all = join(at(-2.0, 0.0, 0.0, mat(0.5, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Cuboid(0.075)))), all);
    return turn(rad(-40.0), rad(20.0), 0.0, all);
}

Solid *ModelH(double a)         /* :559-614 a chain of ten links of five materials running through the two cuboids */
{
    const double r = 0.075, d = 1.0 + 0.5 + 0.5 + 0.1 + 0.1;
    Spectrum blue = mix(g_sEqualWhite, g_sBlue);
    Solid *cubes = floor_slab(3.0);
    cubes = join(at(2.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, Cuboid(r)))), cubes);
    cubes = join(at(-2.0, 0.0, 0.0, mat(0.5, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, Cuboid(r)))), cubes);
    /* link k of the reference's sequence: phase (in units of d), link angle, material kind */
    enum { COPPER, PYREX, WATER, AIR, PLASTIC };
    static const struct { double phase, ang; int kind; } seq[10] = {
        { 0.0, 0.0, COPPER }, { 0.5, 90.0, PYREX }, { 1.0, 0.0, WATER }, { 2.0, 0.0, AIR }, { 1.5, 90.0, PLASTIC },
        { 2.5, 90.0, COPPER }, { 3.0, 0.0, PYREX }, { 3.5, 90.0, WATER }, { 4.5, 90.0, AIR }, { 4.0, 0.0, PLASTIC } };
    Solid *chain = 0;
    for (int k = 0; k < 10; k++) {
        Solid *link = surf(seq[k].kind == PLASTIC ? SURF_SHINY : SURF_POLISHED, 0, size(0.5, ChainLink(seq[k].ang)));
        switch (seq[k].kind) {
        case COPPER:  link = mat(0.9, g_sCopper, MAT_METAL, link); break;
        case PYREX:   link = mat(0.9, REFRACT_PYREX, g_sBlack, MAT_TRANSPARENT, 0, link); break;
        case WATER:   link = mat(0.9, REFRACT_WATER, g_sWater, MAT_TRANSPARENT, ColoredGlass, link); break;
        case AIR:     link = mat(0.9, REFRACT_AIR, g_sBlack, MAT_TRANSPARENT, 0, link); break;
        default:      link = mat(0.9, REFRACT_PYREX, blue, MAT_PLASTIC, 0, link); break;
        }
        link = at(fmod(a + seq[k].phase * d, 5.0 * d) - 2.5 * d, 0.0, 0.0, link);
        chain = chain ? join(link, chain) : link;
    }
    chain = size(1.0 / 1.5, chain);
    return turn(rad(-40.0), rad(20.0), 0.0, join(cubes, chain));
}

Solid *ModelI(double)           /* :616-632 */
{
    Spectrum blue = mix(g_sEqualWhite, g_sBlue);
    Solid *all = at(-3.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, g_sWater, MAT_TRANSPARENT, 0, surf(SURF_POLISHED, 0, new Sphere)));
    all = join(at(+0.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, blue, MAT_PLASTIC, 0, surf(SURF_POLISHED, Bumpy, new Sphere))), all);
    all = join(at(+3.0, 0.0, 0.0, mat(0.5, REFRACT_PYREX, g_sCopper, MAT_METAL, 0, surf(SURF_POLISHED, 0, new Sphere))), all);
    return join(all, floor_slab(3.0));
}

/* the five-ball layouts of MakeModelJ / L / L3 (:634-813): front row of three, back row of two, a wide floor, pitched 30 degrees */
static const double kFive[5][3] = { { -2.5, 1.0, 0.0 }, { 0.0, 1.0, 0.0 }, { 2.5, 1.0, 0.0 }, { -1.25, -0.5, -1.5 }, { +1.25, -0.5, -1.5 } };
Solid *five_balls(Solid *b[5], const int where[5])
{
    Solid *all = at(kFive[where[0]][0], kFive[where[0]][1], kFive[where[0]][2], b[0]);
    for (int k = 1; k < 5; k++) all = join(at(kFive[where[k]][0], kFive[where[k]][1], kFive[where[k]][2], b[k]), all);
    all = join(all, at(0.0, 3.0, 0.0, size(32.0, 1.0, 32.0, new Cube)));
    return turn(rad(-30.0), 0.0, 0.0, all);
}

Solid *ModelJ(double a)         /* :634-670 five bumpy spectral metals */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    static const int luster[5] = { MAT_COPPER, MAT_GOLD, MAT_IRON, MAT_ALUMINUM, MAT_OSMIUM };
    static const TextureMap bump[5] = { VoronoiCrinkled, Dented, Bumpy, MicroFacets2, Tectonic };
    static const int where[5] = { 0, 1, 2, 3, 4 };
    Solid *b[5];
    for (int k = 0; k < 5; k++) b[k] = turn(-rad(a), 0.0, 0.0, mat(0.9, blue, luster[k], surf(SURF_POLISHED, bump[k], new Sphere)));
    return five_balls(b, where);
}

Solid *ModelL(double a)         /* :711-747 the solid colour textures */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    static const TextureMap t[5] = { Agate, Marble, Clouds, Voronoi, Wood };
    static const int where[5] = { 0, 1, 2, 4, 3 };
    Solid *b[5];
    for (int k = 0; k < 5; k++)
        b[k] = turn(-rad(a), 0.0, 0.0, tex(0.9, blue, MAT_PLASTIC, t[k], surf(SURF_BURNISHED, 0, k == 4 ? (Solid *)new Cylinder : (Solid *)new Sphere)));
    return five_balls(b, where);
}

Solid *ModelL2(double a)        /* :749-771 a wooden cylinder */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    Solid *ps = turn(-rad(a), rad(25.0), 0.0, at(0.0, -0.0, 0.0, surf(SURF_BURNISHED, 0, tex(0.9, blue, MAT_PLASTIC, Wood, at(0.0, 0.0, 0.0, new Cylinder)))));
    ps = join(at(0.0, 0.0, 0.0, ps), at(0.0, 3.0, 0.0, size(32.0, 1.0, 32.0, new Cube)));
    return turn(rad(-30.0), 0.0, 0.0, ps);
}

Solid *ModelL3(double a)        /* :773-813 the later bump maps */
{
    Spectrum c[5] = { pale(g_sRed, g_sWhite), pale(g_sGreen, g_sWhite), pale(g_sBlue, g_sWhite),
                      g_sRed * 0.25 + g_sBlue * 0.25 + g_sWhite * 0.5, pale(g_sOrange, g_sWhite) };
    static const int rough[5] = { SURF_SMOOTH, SURF_SMOOTH, SURF_POLISHED, SURF_SHINY, SURF_SHINY };
    static const TextureMap bump[5] = { Bumpy, VoronoiCrinkled, Blistered, Dented, Tectonic };
    static const int where[5] = { 2, 0, 1, 4, 3 };
    Solid *b[5];
    for (int k = 0; k < 5; k++) b[k] = turn(-rad(a), 0.0, 0.0, mat(0.9, c[k], MAT_PLASTIC, surf(rough[k], bump[k], new Sphere)));
    return five_balls(b, where);
}

Solid *ModelM(double a)         /* :851-910 the retort between a marble dodecahedron and a bumpy ellipsoid */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    const double yaw = rad(a - 120.0);
    Solid *glass = retort(yaw);
    Solid *ps = join(tripod(), glass);
    Solid *b = new BoundingBox(turn(0.0, yaw, 0.0, tex(0.9, blue, MAT_PLASTIC, Marble, surf(SURF_SMOOTH, 0, size(0.7, 1.2, 0.7, new Dodecahedron)))));
    ps = join(at(3.0, 0.0, 0.0, b), ps);
    b = new BoundingBox(turn(0.0, yaw, 0.0, mat(0.9, blue, MAT_PLASTIC, surf(SURF_SMOOTH, Bumpy, size(0.7, 1.2, 0.7, new Sphere)))));
    ps = join(at(-3.0, 0.0, 0.0, b), ps);
    ps = join(ps, floor_slab(2.5));
    return turn(-rad(20.0) * 0.7, 0.0, 0.0, ps);
}

Solid *ModelN(double a)         /* :912-940 one plastic ball per roughness */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    static const int rough[4] = { SURF_BURNISHED, SURF_SMOOTH, SURF_SHINY, SURF_POLISHED };
    static const double x[4] = { 3.3, 1.1, -1.1, -3.3 };
    Solid *ps = 0;
    for (int k = 0; k < 4; k++) {
        Solid *b = at(x[k], 0.0, 0.0, turn(0.0, rad(a - 120.0), 0.0, tex(0.9, blue, MAT_PLASTIC, 0, surf(rough[k], 0, size(1.0, 1.0, 1.0, new Sphere)))));
        ps = ps ? join(ps, b) : b;
    }
    return ps;
}

Solid *ModelO(double a)         /* :942-978 where a texture sits relative to Scale / Rotate / Material decides its coordinates */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    const double r = rad(a);
    Solid *ps = at(-3.0, 1.0, 0.0, turn(0.0, r, 0.0, tex(0.9, blue, MAT_PLASTIC, 0, surf(SURF_SMOOTH, Bumpy, size(1.5, 0.5, 0.5, new Sphere)))));
    ps = join(ps, at(0.0, 1.0, 0.0, turn(0.0, r, 0.0, tex(0.9, blue, MAT_PLASTIC, 0, size(1.5, 0.5, 0.5, surf(SURF_SMOOTH, Bumpy, new Sphere))))));
    ps = join(ps, at(3.0, 1.0, 0.0, surf(SURF_SMOOTH, Bumpy, turn(0.0, r, 0.0, tex(0.9, blue, MAT_PLASTIC, 0, size(1.5, 0.5, 0.5, new Sphere))))));
    ps = join(ps, at(-3.0, -1.0, 0.0, turn(0.0, r, 0.0, tex(0.9, blue, MAT_PLASTIC, Marble, surf(SURF_SMOOTH, 0, size(1.5, 0.5, 0.5, new Sphere))))));
    ps = join(ps, at(0.0, -1.0, 0.0, turn(0.0, r, 0.0, size(1.5, 0.5, 0.5, surf(SURF_SMOOTH, 0, tex(0.9, blue, MAT_PLASTIC, Marble, new Sphere))))));
    ps = join(ps, at(3.0, -1.0, 0.0, tex(0.9, blue, MAT_PLASTIC, Marble, surf(SURF_SMOOTH, 0, turn(0.0, r, 0.0, size(1.5, 0.5, 0.5, new Sphere))))));
    return ps;
}

Solid *helix_stack(double zRepeat, double yaw, int turns_of_yaw)
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    Solid *ps = surf(SURF_POLISHED, mat(0.9, blue, MAT_PLASTIC, PseudoHelix(0.1, zRepeat)));
    for (int k = 1; k < 5; k++)
        ps = join(ps, at(0.0, 0.0, zRepeat * k, surf(SURF_POLISHED, mat(0.9, blue, MAT_PLASTIC, PseudoHelix(0.1, zRepeat)))));
    while (turns_of_yaw-- > 0) ps = turn(0.0, yaw, 0.0, ps);
    return ps;
}
Solid *ModelP(double a)         /* :980-1024 two stacks of five pseudo-helix turns */
{
    Spectrum blue = pale(g_sBlue, g_sWhite);
    const double f20 = rad(10.0);
    Solid *s1 = at(+1.25, 0.0, 0.0, helix_stack(0.1, rad(a), 1));
    /* the second stack's translations are 0.3, 0.6, 0.9, 1.2 as written (not 0.3 * k recomputed) */
    Solid *ps = surf(SURF_POLISHED, mat(0.9, blue, MAT_PLASTIC, PseudoHelix(0.1, 0.3)));
    static const double z[4] = { 0.3, 0.6, 0.9, 1.2 };
    for (int k = 0; k < 4; k++) ps = join(ps, at(0.0, 0.0, z[k], surf(SURF_POLISHED, mat(0.9, blue, MAT_PLASTIC, PseudoHelix(0.1, 0.3)))));
    ps = turn(0.0, rad(a), 0.0, turn(0.0, rad(a), 0.0, ps));
    Solid *s2 = at(-1.25, 0.0, 0.0, ps);
    ps = turn(0.5 * D_PI, 0.0, 0.0, join(s1, s2));
    ps = join(ps, at(0.0, 2.5, -10.0, size(12.0, 1.0, 12.0, mat(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube))));
    ps = turn(-f20 * 0.7, 0.0, 0.0, ps);
    return join(ps, size(100.0, mat(0.9, blue, MAT_PLASTIC, new HollowSphere)));
}

Solid *ModelQ(double a)         /* :1026-1066 six shapes of the library */
{
    const double r = rad(a);
    Solid *ps = MakeUnion(
        at(-3.0, +1.5, 0.0, turn(0.0, r, 0.0, Cuboid(0.1, 0.5, 0.7))),
        at(+0.0, +1.5, 0.0, turn(0.0, r, 0.0, IcosahedronCage(0.1, 0.15))),
        at(+3.0, +1.5, 0.0, turn(0.0, r, 0.0, PseudoHelix(0.1, 0.3, 7.0 * D_PI))),
        at(-3.0, -1.5, 0.0, turn(0.0, r, 0.0, join(Diskoid(0.1), Rodoid(0.1)))),
        at(+0.0, -1.1, 0.0, turn(0.1, r, 0.1, Slaboidoid(0.5, 1.1, 0.2, 0.1, 0.025))),
        at(+3.0, -1.5, 0.0, turn(0.0, r, 0.0, size(0.5, Knoboid(0.4)))));
    return surf(SURF_POLISHED, mat(0.9, mix(g_sBlue, g_sWhite), MAT_PLASTIC, ps));
}

Solid *ModelR(double a) { return turn(0.25 * D_PI, 0.0, 0.0, PseudoHelix(0.2, 0.5, rad(a))); }             /* :1068-1081 (the code before its first return) */
Solid *ModelS(double r) { return turn(0.0, 0.5 * D_PI, 0.0, Plastic(Knoboid(r, 1.0))); }                     /* :1083-1093 */

Solid *ModelL4(double a)        /* :815-831 the Earth on a sphere.  Camera z = 5.0 */
{
    return at(0.0, 0.0, 0.0, turn(rad(90.0), rad(a), 0.0,
           tex(0.9, pale(g_sBlue, g_sWhite), MAT_PLASTIC, EarthTexture, surf(SURF_SHINY, 0, turn(0.0, 0.0, 0.0, new Sphere)))));
}
Solid *ModelL5(double a)        /* :833-849 the Earth on a torus.  Camera z = 5.0 */
{
    const double x = rad(-90.0);
    return at(0.0, 0.0, 0.0, turn(x, rad(a), 0.0,
           tex(0.9, pale(g_sBlue, g_sWhite), MAT_PLASTIC, EarthTexture, surf(SURF_SHINY, 0, turn(-x, 0.0, 0.0, new Torus(0.3))))));
}
Solid *ModelU(double a)         /* :1111-1129 the Earth (no water mask) inside a textured sky dome */
{
    Solid *ps = new Rotate(0.5 * D_PI, rad(a), 0.0, tex(0.9, g_sEqualWhite, MAT_PLASTIC, Earth2Texture, surf(SURF_SMOOTH, 0, new Sphere)), "xyz");
    Solid *sky = size(100.0, tex(0.9, g_sEarthSky * 2.0, MAT_NOSHADE, SkyTexture, new HollowSphere));
    sky = turn(rad(90.0), 0.0, 0.0, sky);
    sky = turn(0.0, 0.0, rad(33.0), sky);
    return join(ps, sky);
}

Solid *ModelT(double)           /* :1095-1109 a ring of 36 spheres around the camera */
{
    Solid *ps = turn(0.0, 0.0, 0.0, at(0.0, 0.0, 5.0, new Sphere));
    for (int i = 10; i < 360; i += 10) ps = join(ps, turn(0.0, rad(double(i)), 0.0, at(0.0, 0.0, 25.0, new Sphere)));
    return at(0.0, 0.0, 5.0, ps);
}

Solid *ModelV(double a)         /* :1131-1145 */
{
    Solid *ps = turn(0.0, rad(a), 0.0, tex(0.9, mix(g_sBlue, g_sWhite), MAT_PLASTIC, SeaScape, surf(SURF_POLISHED, 0, size(1.2, new Sphere))));
    return join(ps, at(0.0, 2.5, -10.0, size(12.0, 1.0, 12.0, mat(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube))));
}

Solid *ModelFib(int n, double a)        /* :1147-1174 */
{
    char szN[16], szPhi[16];
    snprintf(szN, sizeof szN, "N = %4d", n);
    snprintf(szPhi, sizeof szPhi, "PHI = %6.3f", D_PHI);
    Solid *ps = turn(0.0, rad(a), 0.0, Plastic2(new Fibonacci(n)));
    ps = join(ps, Plastic2(at(1.5 - 0.6, 1.5, 0.0, size(0.2, SolidText(0.0, 0.0, szN, JUSTIFIED_LEFT, 0.1)))));
    return join(ps, Plastic2(at(-1.5 - 0.9, 1.5, 0.0, size(0.2, SolidText(0.0, 0.0, szPhi, JUSTIFIED_LEFT, 0.1)))));
}

Solid *ModelRuby(double a)      /* :1176-1224 three gem-stone shells inside bumpy metal cages */
{
    const double tilt = ((180.0 - DODECDIHEDRAL) / 2.0) * D_PI / 180.0, yaw = rad(a);
    const double scale = 50.0, wall = 0.005 * scale, fD = 0.79465447229176612295553092832759;
    static const int luster[3] = { MAT_IRON, MAT_COPPER, MAT_GOLD };
    static const double x[3] = { +0.0, +2.5, -2.5 };
    Spectrum absorb[3] = { g_sRuby * 0.4 / scale, g_sSapphire * 1.5 / scale, g_sEmerald * 0.4 / scale };
    Solid *v[6];
    for (int k = 0; k < 3; k++) {
        v[2 * k] = at(x[k], +0.0, +0.0, turn(0.0, yaw, 0.0, turn(tilt, 0.0, 0.0,
            size(0.5, surf(SURF_POLISHED, 0, mat(0.9, g_sWhite, luster[k], size(2.0, DodecahedronCage(0.05, 0.1))))))));
        Solid *shell = new Difference(size(fD + wall, new Dodecahedron), size(fD - wall, new Dodecahedron));
        v[2 * k + 1] = at(x[k], +0.0, +0.0, turn(0.0, yaw, 0.0, turn(tilt, 0.0, 0.0,
            surf(SURF_POLISHED, 0, mat(0.8, REFRACT_SAPPHIRE, absorb[k], MAT_TRANSPARENT, ColoredGlass, shell)))));
    }
    Solid *ps = rt_make_union(v, 6, 0);
    ps = join(ps, at(0.0, 2.5, -10.0, size(12.0, 1.0, 12.0, mat(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube))));
    ps = turn(rad(-20.0), 0.0, 0.0, ps);
    return join(ps, size(1000.0, mat(0.9, g_sEarthSky * 2.0, MAT_NOSHADE, new HollowSphere)));
}

Solid *ModelLetters(double) { return SolidText(0.0, 0.0, "Hello World!", JUSTIFIED_CENTER, 0.05); }        /* :1226-1235 */

Solid *ModelSetOperations(double)       /* :1237-1278 every set operation with both, without its left and without its right operand */
{
    Solid *all = 0;
    for (int k = 0; k < 3; k++) {
        Solid *l[3], *r[3];
        for (int j = 0; j < 3; j++) {       /* k = 1: the +0.5 operand is missing, k = 2: the -0.5 one */
            l[j] = at(+0.5, 0.0, 0.0, k == 1 ? 0 : new Sphere);
            r[j] = at(-0.5, 0.0, 0.0, k == 2 ? 0 : new Sphere);
        }
        Solid *u = new Union(l[0], r[0]), *i = new Intersection(l[1], r[1]), *d = new Difference(l[2], r[2]);
        const double y = k == 0 ? 0.0 : k == 1 ? +2.5 : -2.5;
        all = all ? MakeUnion(all, at(-3.0, y, 0.0, u), at(+0.0, y, 0.0, i), at(+3.0, y, 0.0, d))
                  : MakeUnion(at(-3.0, y, 0.0, u), at(+0.0, y, 0.0, i), at(+3.0, y, 0.0, d));
    }
    return Plastic(new BoundingBox(all));
}

Solid *ModelHyperTest(double a) { return turn(0.0, rad(a), 0.0, new Intersection(new Slab, new ConeHyperCylinder(0.4))); }   /* :1280-1289 */

Solid *ModelLorenz(double a, int steps)         /* :1291-1358 256 points on the Lorenz attractor after `steps` Euler steps */
{
    const double RHO = 28.0, SIGMA = 10.0, BETA = 8.0 / 3.0, dt = 0.01;
    enum { N = 256 };
    static Vector3 pos[N];
    Vector3 crit1(+sqrt(BETA * (RHO - 1.0)), +sqrt(BETA * (RHO - 1.0)), RHO - 1.0), crit2(-sqrt(BETA * (RHO - 1.0)), -sqrt(BETA * (RHO - 1.0)), RHO - 1.0);
    Spectrum blue = g_sEqualWhite * 0.5 + g_sBlue * 0.5;
    for (int i = 0; i < N; i++) { double z = RandomDouble(), y = RandomDouble(), x = RandomDouble(); pos[i] = Vector3(x, y, z); }
    while (--steps >= 0)
        for (int i = 0; i < N; i++) {
            Vector3 v = pos[i];
            double dx = SIGMA * (v.y - v.x), dy = v.x * (RHO - v.z) - v.y, dz = v.x * v.y - BETA * v.z;
            pos[i] = v + Vector3(dx, dy, dz) * dt;
        }
    Solid *all = join(new Translate(crit1, ball(0.45)), new Translate(crit2, ball(0.45)));
    for (int i = 0; i < N; i++)
        all = join(mat(0.9, i == 0 ? g_sRed : blue, MAT_PLASTIC, new Translate(pos[i], ball(i == 0 ? 0.45 : 0.15))), all);
    all = at(0.0, 0.0, -25.0, all);
    all = turn(0.5 * D_PI, 0.0, 0.0, all);
    return turn(0.0, rad(a), 0.0, all);
}

Solid *ModelTestTop(double)     /* :1426-1454; RND draws right to left within each Translate(RND, -RND, RND, ...) */
{
    Solid *group[3];
    for (int g = 0; g < 3; g++) {
        Solid *e[4];
        for (int k = 0; k < 4; k++) {
            double z = 10.0 * RandomDouble() - 5.0, y = 10.0 * RandomDouble() - 5.0, x = 10.0 * RandomDouble() - 5.0;
            e[k] = at(x, -y, z, k < 3 ? size(0.5, 1.0, 1.0, new Sphere) : (Solid *)new Sphere);
        }
        group[g] = new BoundingBox(rt_make_union(e, 4, 1));
    }
    return new BoundingBox(rt_make_union(group, 3, 1));
}

void two_lights(RayTracer &rt)          /* MakeLights, :1456-1470 */
{
    Spectrum s6500 = BlackBody(6500.0);
    rt.plAmbient = new AmbientLight(s6500 * 0.02);
    rt.plPointLights = new AreaLight(1.0, Point3(10.0, -10.0, 10.0), s6500 * 10000.0);
    rt.plPointLights = new PointLight(Point3(-10.0, -10.0, -10.0), s6500 * 500.0, rt.plPointLights);
}
void far_light(RayTracer &rt)           /* MakeLightsLorenz, :1472-1487 */
{
    Spectrum s6500 = BlackBody(6500.0);
    rt.plAmbient = new AmbientLight(s6500 * 0.02);
    rt.plPointLights = new AreaLight(1.0, Point3(100.0, -100.0, 100.0), s6500 * 1000000.0);
    rt.plPointLights->nNoShadow = 1;
}

/* ------------------------------------------------------------------------------------------------- Venera.cpp ---- */
namespace venera {
const double RINGWIDE = 0.17, RINGHIGH = RINGWIDE * 0.75, RINGBOLTS = 120.0;
const double FLANGE = 0.075, FLANGEX = 0.022, FTHICK = 0.0025, FRAD = FLANGEX * 0.5, FBOLT = 0.5 * 0.012;
const double PHULL = 0.41, PHIGH = 0.12, RSTRUT = 0.02;

Solid *Titanium(Solid *ps) { return mat(0.8, REFRACT_WATER, g_sEqualWhite, MAT_TITANIUM, 0, surf(SURF_BURNISHED, 0, ps)); }       /* :144-149 */
Solid *Titanium2(Solid *ps) { return mat(0.8, REFRACT_WATER, g_sEqualWhite, MAT_OSMIUM, 0, surf(SURF_SMOOTH, 0, ps)); }
Solid *Titanium3(Solid *ps) { return mat(0.8, REFRACT_WATER, g_sEqualWhite, MAT_TITANIUM, 0, surf(SURF_SMOOTH, 0, ps)); }
Solid *Quartz(Solid *ps) { return mat(1.0, REFRACT_QUARTZ, g_sBlack, MAT_TRANSPARENT, ColoredGlass, surf(SURF_POLISHED, 0, ps)); }
Solid *KG_25(Solid *ps) { return mat(0.9, REFRACT_WATER, mix(g_sEqualWhite, g_sYellow), MAT_PLASTIC, 0, surf(SURF_MATTE, 0, ps)); }
Solid *bolt() { return size(FBOLT, new Prism(6)); }

Solid *ColorChips()             /* VeneraColorChips, :221-235 */
{
    static const char *names[5] = { "g_sEqualWhite", "g_sCoolGray", "g_sCoolRed", "g_sCoolGrn", "g_sCoolBlu" };
    Solid *v[5];
    for (int k = 0; k < 5; k++) v[k] = at(-5.0 + 2.0 * k, 0.0, 0.0, mat(0.9, *rt_spectrum_by_name(names[k]), MAT_PLASTIC, size(1.0, 1.0, 0.1, new Cube)));
    return size(0.25, new BoundingBox(rt_make_union(v, 5, 0)));
}

Solid *HalfSlaboid(double x, double y, double z, double r)      /* :306-315 */
{
    Solid *ps = at(0.0, y - r, 0.0, Slaboid(x, y + r, z, r));
    ps = new Difference(ps, size(1.0, 1.01 * y, 1.0, at(0.0, -1.0, 0.0, new Cube)));
    return join(ps, turn(0.0, 0.5 * D_PI, 0.0, size(z, z, x, new Cylinder)));
}

/* a plate with its four bolts (the two plates of Flange1/Flange2, :325-338) */
void bolted_plate(Solid **v, double high)
{
    v[0] = HalfSlaboid(FLANGEX, high, FTHICK, FRAD);
    v[1] = at(FLANGEX - FRAD, FRAD, 0.0, bolt());
    v[2] = at(FLANGEX - FRAD, 2.0 * high - FRAD, 0.0, bolt());
    v[3] = at(-FLANGEX + FRAD, FRAD, 0.0, bolt());
    v[4] = at(-FLANGEX + FRAD, 2.0 * high - FRAD, 0.0, bolt());
}

Solid *RingFlange(double deg, bool with_tab)    /* Flange1 (:317-347) and Flange2 (:349-379: one more tab) */
{
    Solid *web = at(0.0, FLANGE - FTHICK, -(FLANGEX - FTHICK), turn(0.0, 0.5 * D_PI, 0.0, Slaboid(FLANGEX, FLANGE, FTHICK, FRAD)));
    web = turn(+0.25 * D_PI, 0.0, 0.0, size(1.0, 1.0, 1.6, turn(-0.25 * D_PI, 0.0, 0.0, web)));
    Solid *cut = size(2.0 * FTHICK, 2.0 * FLANGE, 2.0 * FLANGEX, at(0.0, 1.0, -1.0, new Cube));
    Solid *v[8], *w[5];
    bolted_plate(v, FLANGE);
    bolted_plate(w, 0.7 * FLANGE);
    int n = 5;
    v[n++] = turn(-0.5 * D_PI, 0.0, 0.0, rt_make_union(w, 5, 0));
    if (with_tab) v[n++] = turn(0.75 * D_PI, 0.0, 0.0, HalfSlaboid(1.5 * FLANGEX, 0.3 * FLANGE, FTHICK, FRAD));
    v[n++] = new Difference(at(0.0, -0.4 * FLANGE, 0.55 * FLANGE, web), cut);
    Solid *ps = Titanium3(rt_make_union(v, n, 0));
    return new BoundingBox(turn(0.0, 0.0, rad(deg), at(0.0, 1.0 - RINGWIDE - 2.0 * FTHICK, RINGHIGH + 2.0 * FTHICK, ps)));
}

Solid *HullFlange(double deg)   /* Flange3, :381-395 */
{
    const double PF3 = -atan2(FLANGEX, PHULL);
    Solid *ps = size(PHULL + FTHICK, PHULL + FTHICK, PHULL, new Cylinder);
    Solid *lip = at(0.0, +PHULL, 0.0, turn(-0.25 * D_PI, 0.0, 0.0, at(0.0, -PHULL, 0.0, size(PHULL + 0.3 * FLANGE, PHULL + 0.3 * FLANGE, FTHICK, new Cylinder))));
    ps = new Difference(join(ps, lip), size(PHULL, PHULL, PHULL + FTHICK, new Cylinder));
    ps = new Intersection(ps, at(0.0, +PHULL, 0.0, turn(0.5 * D_PI, 0.0, 0.0, Slaboid(1.5 * FLANGEX, FLANGEX, FLANGEX, FRAD))));
    ps = turn(0.0, 0.0, rad(deg), turn(PF3, 0.0, 0.0, new BoundingBox(ps)));
    return Titanium3(ps);
}

Solid *LandingRing()            /* :397-458 */
{
    const double wide = 1.0 + 0.25 * RINGHIGH;
    Solid *ps = at(0.0, 0.0, 1.0, new Difference(size(wide, wide, 1.0, new Cylinder), size(1.0 - RINGWIDE, 1.0 - RINGWIDE, 1.01, new Cylinder)));
    ps = size(1.0, 1.0, 0.5 * RINGHIGH, ps);
    ps = new Intersection(ps, size(wide, wide, 4.0, at(0.0, 0.0, 1.0, new ConeUnit)));
    Solid *groove = new Difference(new Slab, size(1.0 - RINGWIDE + 0.01, 1.0 - RINGWIDE + 0.01, 1.01, new Cylinder));
    ps = new Difference(ps, size(1.0, 1.0, 0.5 * RINGHIGH, groove));
    const double step = 2.0 * D_PI / RINGBOLTS;
    for (double f = 0.0; f < 2.0 * D_PI; f += 6.0 * step) {     /* twenty boxed groups of six bolts, each wrapping the ring in one more Titanium */
        Solid *six[6];
        for (int k = 0; k < 6; k++) six[k] = turn(0.0, 0.0, double(k) * step + f, at(1.0, 0.0, 0.0, turn(0.0, 0.5 * D_PI, 0.0, bolt())));
        ps = Titanium(join(ps, at(0.0, 0.0, RINGHIGH - 3.0 * FBOLT, new BoundingBox(rt_make_union(six, 6, 0)))));
    }
    const double s = 1.0 - 0.4 * RINGWIDE;
    Solid *pad = Titanium(new Difference(size(s, s, 1.25, new Torus(0.6 * RINGWIDE)), size(s, s, 1.25, new Torus(0.5 * RINGWIDE))));
    Solid *groups[3];
    for (int g = 0; g < 3; g++) {
        Solid *four[4];
        for (int k = 0; k < 4; k++) four[k] = RingFlange(120.0 * g + 30.0 * k, (k & 1) != 0);
        groups[g] = new BoundingBox(rt_make_union(four, 4, 0));
    }
    return new BoundingBox(MakeUnion(ps, pad, rt_make_union(groups, 3, 0)));
}

Solid *Body()                   /* :460-499 (the insulation is built and left out there; here it is not built) */
{
    Solid *hull = join(size(PHULL, new Sphere), size(PHULL + 0.02, PHULL + 0.02, 0.01, new Cylinder));
    for (double f = 0; f < 1.9999 * D_PI; f += 10.0 * D_PI / 180.0)
        hull = join(hull, turn(0.0, 0.0, f, at(0.0, PHULL + 0.01, 0.0, size(0.005, 0.005, 0.015, new Prism(6)))));
    hull = join(Titanium(hull), KG_25(size(PHULL - 0.01, new Sphere)));
    hull = new Difference(hull, size(PHULL - 0.04, new Sphere));
    const double f50 = (90.0 - 50.0) * D_PI / 180.0;
    hull = new Difference(hull, turn(0.0, +f50, 0.0, at(0.0, 0.0, PHULL, size(0.02, 0.02, 0.05, new Cylinder))));
    hull = new Difference(hull, turn(0.0, -f50, 0.0, at(0.0, 0.0, PHULL, size(0.02, 0.02, 0.05, new Cylinder))));
    Solid *six[6];
    for (int k = 0; k < 6; k++) six[k] = HullFlange(60.0 * k);
    return new BoundingBox(at(0.0, 0.0, PHIGH + PHULL, MakeUnion(hull, (Solid *)new BoundingBox(rt_make_union(six, 6, 0)))));
}

Solid *ShockAbsorber(const Point3 &p1, const Point3 &p2)        /* :501-512 */
{
    return MakeUnion(CylinderFromTo(p1, p2, RSTRUT), (Solid *)new Translate(p1, ball(RSTRUT)), (Solid *)new Translate(p2, ball(RSTRUT)));
}
Solid *Strut(double deg)        /* :514-537 */
{
    Point3 p1(1.0 - RINGWIDE - 0.5 * RSTRUT, 0.0, RINGHIGH + 1.0 * RSTRUT), p2(PHULL, 0.0, PHULL + PHIGH - FLANGEX);
    const double a1 = (30.0 - 1.5) * D_PI / 180.0, a2 = 5.0 * D_PI / 180.0;
    Vector3 r1 = Point3(p1.x * cos(a1), p1.x * sin(a1), p1.z) - p1, r2 = Point3(p2.x * cos(a2), p2.x * sin(a2), p2.z) - p2;
    Vector3 r1b = Point3(p1.x * cos(-a1), p1.x * sin(-a1), p1.z) - p1, r2b = Point3(p2.x * cos(-a2), p2.x * sin(-a2), p2.z) - p2;
    Solid *ps = Titanium2(MakeUnion(ShockAbsorber(p1, p2), ShockAbsorber(p1 + r1, p2 + r2), ShockAbsorber(p1 + r1b, p2 + r2b)));
    return new BoundingBox(turn(0.0, 0.0, rad(deg) + 0.5 * D_PI, ps));
}

Solid *Telephotometer(double deg)       /* Camera(), :552-601: modelled in centimetres */
{
    const double CRAD = 10.5 * 0.5, CQUARTZ = 12 * 0.5, CTRIM = (12.0 + 1.0) * 0.5 * sqrt(2.0) - 0.7, CDELTA = (12.0 + 1.0) * 0.5;
    const double CLONG = 16.0, CSTEM = CLONG - 7.0, CSTDEL = (CLONG - 6.0) / 2.0;
    Solid *ps = MakeUnion(
        at(0.0, 0.0, 0.5, size(CRAD, CRAD, 0.5, new Cylinder)),
        at(0.0, 0.0, CSTDEL, size(5.5 / 2.0, 5.5 / 2.0, CSTEM / 2, new Cylinder)),
        turn(0.0, 0.0, 0.0 * D_PI / 1.5, at(5.5 / 2.0, 0.0, CSTDEL, size(1.0, 0.25, CSTEM / 2.0, new Cylinder))),
        turn(0.0, 0.0, 1.0 * D_PI / 1.5, at(5.5 / 2.0, 0.0, CSTDEL, size(1.0, 0.25, CSTEM / 2.0, new Cylinder))),
        turn(0.0, 0.0, 2.0 * D_PI / 1.5, at(5.5 / 2.0, 0.0, CSTDEL, size(1.0, 0.25, CSTEM / 2.0, new Cylinder))),
        at(0.0, 0.0, CLONG + 3.5 / 2.0, size(CQUARTZ + 0.5, CQUARTZ + 0.5, 3.5 / 2.0, new Cylinder)),
        at(0.0, 0.0, CLONG - 3.5 / 2.0 - 6.0, size(CQUARTZ + 0.5, CQUARTZ + 0.5, 3.5 / 2.0, new Cylinder)));
    ps = new Difference(ps, at(0.0, 0.0, CSTDEL + 1.0, size(4.0 / 2.0, 4.0 / 2.0, CSTDEL + 1.1, new Cylinder)));
    ps = new Difference(ps, at(0.0, 0.0, CLONG - 2.5, size(4.9, new Sphere)));
    ps = new Difference(ps, at(0.0, 0.0, CLONG - 3.5, size(4.9, new Sphere)));
    ps = new Intersection(ps, MakeUnion(
        at(0.0, 0.0, CSTDEL, size(CRAD + 0.1, CRAD + 0.1, CSTDEL + 0.01, new Cylinder)),
        at(0.0, 0.0, CLONG + 3.5 - CDELTA, size(CTRIM, new Sphere)),
        at(0.0, 0.0, CLONG - 6.0 - 3.5 + CDELTA, size(CTRIM, new Sphere))));
    Solid *v[7];
    for (int k = 0; k < 6; k++) v[k] = turn(0.0, 0.0, double(k) * D_PI / 3.0, at(0.0, 4.0, 0.0, size(0.5, 0.5, 1.5, new Prism(6))));
    v[6] = ps;
    ps = Titanium(size(0.01, rt_make_union(v, 7, 0)));
    Solid *window = new Difference(size(CQUARTZ, CQUARTZ, 6.5 / 2.0, new Cylinder), size(CQUARTZ - 1.0, CQUARTZ - 1.0, 12.0 / 2.0 + 0.01, new Cylinder));
    window = Quartz(size(0.01, at(0.0, 0.0, CLONG - 3.0, window)));
    ps = join(ps, window);
    ps = at(0.0, 0.0, PHULL + PHIGH, turn(0.0, (90.0 - 50.0) * D_PI / 180.0, 0.0, at(0.0, 0.0, PHULL, ps)));
    return new BoundingBox(turn(0.0, 0.0, rad(deg), ps));
}

Solid *Lander()                 /* VeneraLander, :603-614 */
{
    Solid *struts[6];
    for (int k = 0; k < 6; k++) struts[k] = Strut(60.0 * k);
    return turn(rad(90.0), 0.0, 0.0, MakeUnion(
        turn(+0.0 * D_PI / 180.0, 0.0, 0.0, at(0.0, 1.5, -0.2, ColorChips())),
        LandingRing(), Telephotometer(0.0), Telephotometer(180.0), rt_make_union(struts, 6, 0), Body()));
}

Spectrum ground_colour()        /* :244-250: Venera-14 ground over sky, the blue end extrapolated linearly */
{
    Spectrum g = *rt_spectrum_by_name("g_sVenera14Ground") / *rt_spectrum_by_name("g_sVenera14Sky");
    double f = g.rgf[5] - g.rgf[4];
    for (int k = 1; k <= 4; k++) g.rgf[4 - k] = float(g.rgf[4] - double(k) * f);
    return g;
}

Solid *InLab(Solid *ps, RayTracer &rt)          /* EnvironmentLab, :261-285 */
{
    Spectrum s6500 = BlackBody(6500.0);
    rt.plAmbient = new HavercosineLight(g_sEarthSky * 0.025, g_sEqualWhite * 0.00625);
    rt.plPointLights = new PointLight(Point3(9.0, -6.5, 12.0), s6500 * 8000.0 * 1.0);
    rt.plPointLights = new PointLight(Point3(-10.0, -10.0, -10.0), s6500 * 100.0, rt.plPointLights);
    ps = join(ps, at(0.0, 1.0, -10.0, size(12.0, 0.1, 12.0, mat(0.9, g_sEqualWhite, MAT_PLASTIC, new Cube))));
    ps = join(ps, size(1000.0, mat(0.9, g_sEarthSky * 2.0, MAT_NOSHADE, new HollowSphere)));
    return turn(-rad(10.0), 0.0, 0.0, ps);
}
Solid *OnVenus(Solid *ps, RayTracer &rt)        /* EnvironmentVenus, :237-259 */
{
    Spectrum sky = *rt_spectrum_by_name("g_sVenera13Sky");
    rt.plAmbient = new HavercosineLight(sky * 0.15 / 250.0, *rt_spectrum_by_name("g_sVenera14Ground") * 0.15 / 250.0);
    rt.plPointLights = new PointLight(Point3(-10.0, -10.0, -10.0), BlackBody(6500.0) * 0.0);
    ps = join(ps, at(0.0, 3.0, -10.0, size(20.0, 1.0, 20.0, mat(0.9, ground_colour() * 4.0, MAT_PLASTIC, new Cube))));
    ps = join(ps, size(1000.0, mat(0.9, sky * 1.5, MAT_NOSHADE, new HollowSphere)));
    return turn(-rad(10.0), 0.0, 0.0, ps);
}
/* not in the reference: the Venus of OnVenus under its own air -- refractive index of the surface atmosphere (VENUS_REFRACT,
   TextureMaps.cpp:99) in the Atmosphere propagator, a sphere larger than the sky dome so no ray meets its boundary */
Solid *UnderVenusAir(Solid *ps)
{
    return join(ps, size(1500.0, mat(DENSE_VENUS, 1.014716, g_sEqualWhite, MAT_TRANSPARENT, Atmosphere, new Sphere)));
}
}   // namespace venera

/* --------------------------------------------------------------------------------------------- Megacycles.cpp ---- */
namespace mega {
typedef Solid *(*Piece)(double t);

void limb(Solid **v, const double (*p)[6], int n) { for (int k = 0; k < n; k++) v[k] = at(p[k][0], p[k][1], p[k][2], size(p[k][3], p[k][4], p[k][5], new Sphere)); }

Solid *arm(double shoulder, double elbow)
{
    static const double fore[4][6] = { { 0.0, 0.0, 0.0, 0.13, 0.20, 0.13 }, { 0.0, -0.5, 0.0, 0.12, 0.50, 0.12 }, { 0.0, -1.0, 0.0, 0.10, 0.10, 0.10 }, { 0.0, -1.2, 0.0, 0.12, 0.24, 0.06 } };
    Solid *f[4];
    limb(f, fore, 4);
    return turn(0.0, 0.0, rad(shoulder), MakeUnion(
        at(0.0, -1.1, 0.0, turn(0.0, 0.0, rad(elbow), rt_make_union(f, 4, 0))),
        at(0.0, -0.55, 0.0, size(0.23, 0.55, 0.16, new Sphere)),
        at(0.0, -0.10, -0.1, turn(-rad(35.0), 0.0, 0.0, size(0.22, 0.33, 0.22, new Sphere)))));
}
Solid *leg(double hip, double knee)
{
    static const double calf[4][6] = { { 0.0, 0.00, 0.0, 0.20, 0.20, 0.20 }, { 0.0, -0.75, 0.0, 0.20, 0.75, 0.20 }, { 0.0, -1.50, 0.0, 0.20, 0.12, 0.12 }, { 0.2, -1.60, 0.0, 0.45, 0.10, 0.20 } };
    Solid *c[4];
    limb(c, calf, 4);
    return turn(0.0, 0.0, rad(hip), MakeUnion(
        at(0.0, -1.6, 0.0, turn(0.0, 0.0, rad(knee), rt_make_union(c, 4, 0))),
        at(0.0, -0.8, 0.0, size(0.30, 0.80, 0.27, new Sphere))));
}

Solid *Robot(double lShoulder, double lElbow, double rShoulder, double rElbow, double lHip, double lKnee, double rHip, double rKnee)  /* :8-77 */
{
    static const double body[9][6] = {
        { 0.0, 1.6, 0.00, 0.20, 0.40, 0.20 }, { 0.0, 1.2, 0.15, 0.30, 0.40, 0.55 }, { 0.0, 1.2, -0.15, 0.30, 0.40, 0.55 },
        { 0.0, 0.8, 0.00, 0.30, 0.80, 0.50 }, { 0.0, -0.5, 0.00, 0.35, 0.35, 0.55 }, { 0.0, 0.34, 0.00, 0.25, 0.25, 0.35 },
        { 0.0, 0.17, 0.00, 0.25, 0.25, 0.35 }, { 0.0, 0.00, 0.00, 0.25, 0.25, 0.35 }, { 0.0, -0.17, 0.00, 0.25, 0.25, 0.35 } };
    Solid *t[10];
    t[0] = at(0.0, 2.1, 0.00, turn(0.0, 0.0, rad(20.0), size(0.32, 0.48, 0.32, new Sphere)));
    limb(t + 1, body, 9);
    Solid *trunk = new BoundingBox(rt_make_union(t, 10, 0));
    Solid *lArm = new BoundingBox(at(0.0, 1.3, 0.7, arm(lShoulder, lElbow)));
    Solid *rArm = new BoundingBox(at(0.0, 1.3, -0.7, size(1.0, 1.0, -1.0, arm(rShoulder, rElbow))));
    Solid *lLeg = new BoundingBox(at(0.0, -0.5, 0.3, leg(lHip, lKnee)));
    Solid *rLeg = new BoundingBox(at(0.0, -0.5, -0.3, size(1.0, 1.0, -1.0, leg(rHip, rKnee))));
    Solid *ps = new BoundingBox(MakeTopUnion(trunk, lArm, rArm, lLeg, rLeg));
    Spectrum orange = g_sRed + g_sGreen * 0.43 + g_sBlue * 0.07;
    return surf(SURF_SHINY, mat(0.9, orange, MAT_PLASTIC, ps));
}

void pedal(double s, double &hip, double &knee)         /* :88-101: leg angles from the pedal angle */
{
    double f = 0.8 * sin(s), g = 0.8 * cos(s), h = 16 / 3.0 - 1.0, l = 8.0 / 3.0;
    double d = sqrt((h - g) * (h - g) + f * f), v = sqrt(l * l - d * d / 4.0), a = atan(2.0 * v / d), c = atan(f / (g - h));
    knee = -360.0 * a / D_PI;
    hip = 180.0 * (a + c) / D_PI;
}

Solid *Unicyclist(double t)     /* :81-125 */
{
    double lHip, lKnee, rHip, rKnee, s = D_PI * (2.0 - t);
    pedal(s, lHip, lKnee);
    pedal(s + D_PI, rHip, rKnee);
    Solid *ps = Robot(20.0 * cos(D_PI * (t + 1.0)), 15.0 + 15.0 * cos(D_PI * (t + 1.0)), 20.0 * cos(D_PI * t), 15.0 + 15.0 * cos(D_PI * t), lHip, lKnee, rHip, rKnee);
    ps = at(0.0, -18.5 / 3, 0.0, size(5.0 / 3.0, -5.0 / 3.0, -5.0 / 3.0, turn(0.0, 0.5 * D_PI, 0.0, ps)));
    ps = new TopUnion(ps, at(0.0, -1.0, 0.0, turn(0.0, 0.5 * D_PI, 0.0, size(1.27324, 1.27324, 0.1, new Cylinder))));
    return at(0.0, -0.13662, 0.0, size(0.5, ps));
}

/* an edge of the curve: a wall and three riders -- straight on (eF), the last one turning right (eR) or left (eL); :127-158 */
Solid *edge(double t, int bend)
{
    Solid *last = bend == 0 ? at(0.0, 0.0, 10.0 + 2.0 * t, Unicyclist(t))
                            : at(2.0 * bend, 0.0, 10.0, turn(0.0, 0.25 * D_PI * t * bend, 0.0, at(-2.0 * bend, 0.0, 0.0, Unicyclist(t))));
    return new BoundingBox(MakeTopUnion(at(0.0, 32.0, 6.0, size(2.0, 32.0, 8.0, new Cube)),
                                        at(0.0, 0.0, 2.0 + 2.0 * t, Unicyclist(t)), at(0.0, 0.0, 6.0 + 2.0 * t, Unicyclist(t)), last));
}
Solid *eF(double t) { return edge(t, 0); }
Solid *eR(double t) { return edge(t, +1); }
Solid *eL(double t) { return edge(t, -1); }

Solid *Clock1(double t, Solid *ps)      /* :160-168 */
{
    return new BoundingBox(MakeTopUnion(eR(t), at(0.0, 0.0, 12.0, turn(0.0, 0.5 * D_PI, 0.0, eR(t))), at(12.0, 0.0, 12.0, turn(0.0, D_PI, 0.0, ps))));
}
Solid *Counter1(double t, Solid *ps)    /* :188-196 */
{
    return new BoundingBox(MakeTopUnion(at(12.0, 0.0, 0.0, eL(t)), at(12.0, 0.0, 12.0, turn(0.0, -0.5 * D_PI, 0.0, eL(t))), at(0.0, 0.0, 12.0, turn(0.0, -D_PI, 0.0, ps))));
}
Solid *Clock1R(double t) { return Clock1(t, eR(t)); }
Solid *Clock1F(double t) { return Clock1(t, eF(t)); }
Solid *Clock1L(double t) { return Clock1(t, eL(t)); }
Solid *Counter1R(double t) { return Counter1(t, eR(t)); }
Solid *Counter1F(double t) { return Counter1(t, eF(t)); }
Solid *Counter1L(double t) { return Counter1(t, eL(t)); }

Solid *Clockwise(double w, Solid *q1, Solid *e1, Solid *q2, Solid *e2, Solid *q3, Solid *e3, Solid *q4)        /* :216-228 */
{
    return new BoundingBox(MakeTopUnion(
        at(0.0, 0.0, w, turn(0.0, 0.5 * D_PI, 0.0, q1)), at(0.0, 0.0, w, e1), at(0.0, 0.0, 12.0 + w, q2),
        at(w, 0.0, 12.0 + w, turn(0.0, 0.5 * D_PI, 0.0, e2)), at(12.0 + w, 0.0, 12.0 + w, q3),
        at(12.0 + w + w, 0.0, 12.0 + w, turn(0.0, D_PI, 0.0, e3)), at(12.0 + w + w, 0.0, 0.0, turn(0.0, -0.5 * D_PI, 0.0, q4))));
}
Solid *Counterwise(double w, Solid *q1, Solid *e1, Solid *q2, Solid *e2, Solid *q3, Solid *e3, Solid *q4)      /* :230-242 */
{
    return new BoundingBox(MakeTopUnion(
        at(0.0, 0.0, w, turn(0.0, 0.5 * D_PI, 0.0, q1)), at(0.0, 0.0, 12.0 + w, turn(0.0, D_PI, 0.0, e1)), at(0.0, 0.0, 12.0 + w, q2),
        at(12.0 + w, 0.0, 12.0 + w, turn(0.0, -0.5 * D_PI, 0.0, e2)), at(12.0 + w, 0.0, 12.0 + w, q3),
        at(12.0 + w + w, 0.0, w, e3), at(12.0 + w + w, 0.0, 0.0, turn(0.0, -0.5 * D_PI, 0.0, q4))));
}

/* EvenPeanos (:258-268) builds seven second-order pieces; Peano2 (:280-290) uses [4], [1], [0] and [6] (= [4] again) */
Solid *even_piece(int k, double t, double w)
{
    Piece cR = Clock1R, cF = Clock1F, cL = Clock1L, ccR = Counter1R, ccF = Counter1F, ccL = Counter1L;
    if (k <= 2) return Clockwise(w, ccR(t), eF(t), cL(t), eL(t), cF(t), eR(t), k == 0 ? ccR(t) : k == 1 ? ccF(t) : ccL(t));
    return Counterwise(w, k == 3 ? cR(t) : k == 5 ? cL(t) : cF(t), eL(t), ccF(t), eR(t), ccR(t), eF(t), cL(t));
}
Solid *Peano2(double t)
{
    return Clockwise(12.0 + 12.0 + 12.0, even_piece(4, t, 12.0), eR(t), even_piece(1, t, 12.0), eF(t), even_piece(0, t, 12.0), eF(t), even_piece(6, t, 12.0));
}

Solid *Outdoors(Solid *ps, RayTracer &rt)       /* Environment, :292-307 */
{
    rt.plAmbient = new HavercosineLight(g_sEarthSky * 0.025, g_sEqualWhite * 0.00625);
    rt.plPointLights = new PointLight(Point3(90.0, -65.0, 120.0), BlackBody(6500.0) * 800000.0 * 1.0);
    return new TopUnion(ps, size(1000.0, mat(0.9, g_sEarthSky * 2.0, MAT_NOSHADE, new HollowSphere)));
}
}   // namespace mega
}   // namespace

/* Returns 1 when `spec` names a conformance scene (rt filled in), 0 when it does not, -1 with *err set when it is refused. */
int rt_build_conformance_scene(const char *spec, RayTracer &rt, std::string *err)
{
    std::string name(spec), arg;
    size_t p = name.find('@');
    if (p != std::string::npos) { arg = name.substr(p + 1); name = name.substr(0, p); }
    const bool has = !arg.empty();
    const double a = has ? atof(arg.c_str()) : 30.0;

    static const struct { const char *n; Solid *(*f)(double); double cam_z; } models[] = {
        { "Model2", Model2, 8.0 }, { "Model3", Model3, 9.0 }, { "Model4", Model4, 8.0 }, { "Model5", Model5, 9.0 }, { "Model7", Model7, 8.0 },
        { "Model8", Model8, 9.0 }, { "Model9", Model9, 8.0 }, { "ModelA", ModelA, 8.0 }, { "ModelB", ModelB, 8.0 }, { "ModelC", ModelC, 8.0 },
        { "ModelE", ModelE, 8.0 }, { "ModelF", ModelF, 8.0 }, { "ModelG2", ModelG2, 8.0 }, { "ModelH", ModelH, 8.0 }, { "ModelI", ModelI, 8.0 },
        { "ModelJ", ModelJ, 9.0 }, { "ModelL", ModelL, 9.0 }, { "ModelL2", ModelL2, 9.0 }, { "ModelL3", ModelL3, 9.0 }, { "ModelL4", ModelL4, 5.0 }, { "ModelL5", ModelL5, 5.0 },
        { "ModelU", ModelU, 5.0 }, { "ModelM", ModelM, 8.0 },
        { "ModelN", ModelN, 8.0 }, { "ModelO", ModelO, 8.0 }, { "ModelP", ModelP, 8.0 }, { "ModelQ", ModelQ, 8.0 }, { "ModelR", ModelR, 8.0 },
        { "ModelT", ModelT, 5.0 }, { "ModelV", ModelV, 8.0 }, { "Letters", ModelLetters, 8.0 },
        { "SetOperations", ModelSetOperations, 8.0 }, { "HyperTest", ModelHyperTest, 8.0 }, { "TestTop", ModelTestTop, 12.0 } };
    int which = -1;
    for (size_t k = 0; k < sizeof models / sizeof models[0]; k++) if (name == models[k].n) which = (int)k;
    const bool extra = name == "Model6" || name == "ModelD" || name == "ModelG" || name == "ModelK" || name == "ModelS" || name == "Ruby" || name == "Fib" || name == "Lorenz";
    const bool driver = name == "Venera" || name == "VeneraVenus" || name == "VeneraAtmosphere" || name == "Megacycles" || name == "Unicyclist";
    if (which < 0 && !extra && !driver) return 0;

    /* every scene starts from the default PRNG seeds and auxiliary dimensions */
    ML_InitializeRandom();
    g_uCamera = g_vCamera = g_uLight = g_vLight = g_uMicroFacet = g_vMicroFacet = 0.0;
    BuildQuasiIndex(g_BlueNoiseQuasiIndex2, 2, BLUE_N);

    if (name == "Venera" || name == "VeneraVenus" || name == "VeneraAtmosphere") {        /* Venera(), Venera.cpp:626-670: frame t of 360 */
        rt.psSample = new FastBlueSamplingTile;
        rt.pcCamera = new StereographicCamera(Point3(0.0, -0.4, 9.0), 105.0);
        Solid *ps = turn(0.0, rad(has ? a : 0.0), 0.0, venera::Lander());
        rt.psModel = name == "Venera" ? venera::InLab(ps, rt) : venera::OnVenus(ps, rt);
        if (name == "VeneraAtmosphere") rt.psModel = venera::UnderVenusAir(rt.psModel);
        return 1;
    }
    if (name == "Megacycles") {                             /* Megacycles(), Megacycles.cpp:317-349: the frame at time t in [0, 2) */
        rt.psSample = new DebugSamplingTile;
        rt.pcCamera = new StereographicCamera(Point3(17.0 - 0.5, -7.0, 172.0), 5000.0);
        Solid *ps = turn(0.0, (90 + 30.0) * D_PI / 180.0, 0.0, mega::Peano2(has ? a : 0.0));
        ps = turn(-15.0 * D_PI / 180.0, 0.0, 0.0, ps);
        rt.psModel = mega::Outdoors(ps, rt);
        return 1;
    }
    if (name == "Unicyclist") {
        rt.psSample = new FastBlueSamplingTile;
        rt.pcCamera = new StereographicCamera(Point3(0.0, -2.0, 14.0), 70.0);
        rt.psModel = mega::Outdoors(turn(0.0, rad(-60.0), 0.0, mega::Unicyclist(has ? a : 0.0)), rt);
        return 1;
    }

    /* the TestScene.cpp models: the driver's 41 RNDSPEC draws first (:1566-1581; arguments drawn right to left) */
    for (int pass = 0; pass < 3; pass++)
        for (int i = 0; i < (pass == 0 ? 1 : 20); i++) {
            double b = RandomDouble(), g = RandomDouble(), r = RandomDouble();
            s_rgs[i] = RGBtoSpectrum(DisplayRGB(r, g, b));
        }
    for (int i = 1; i <= 9; i += 2) s_rgs[i] = g_sCopper;
    if (name == "Lorenz") {                                 /* the driver's light and a camera that sees the attractor */
        far_light(rt);
        rt.pcCamera = new StereographicCamera(Point3(0.0, 0.0, 80.0), 70.0);
        rt.psModel = ModelLorenz(0.0, has ? atoi(arg.c_str()) : 200);
        return 1;
    }
    double cam_z = 8.0;
    Solid *model;
    if (which >= 0) { model = models[which].f(a); cam_z = models[which].cam_z; }
    else if (name == "Model6") model = water_blocks(a, false);
    else if (name == "ModelD") model = water_blocks(a, true);
    else if (name == "ModelG") model = cuboids_and_balls(a, false, 0.0);
    else if (name == "ModelK") model = cuboids_and_balls(20.0, true, has ? a : 0.3);
    else if (name == "ModelS") model = ModelS(has ? a : 0.4);
    else if (name == "Ruby") model = ModelRuby(has ? a : 20.0);
    else model = ModelFib(has ? atoi(arg.c_str()) : 200, 30.0);
    two_lights(rt);
    rt.pcCamera = new StereographicCamera(Point3(+0.0, 0.0, cam_z), 70.0);
    rt.psModel = model;
    return 1;
}
