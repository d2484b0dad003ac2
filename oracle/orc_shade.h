/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h).
 * Scene container + shading entry points (restating Shade.cpp, Lights.h/.cpp, Cameras.h).
 */
#pragma once
#include "orc_solids.h"

namespace orc {

struct Light {
    int  type, no_shadow;
    V3   org, vA, vB;
    Spec flux;
};

struct Scene {
    std::vector<Node*> nodes;       /* index == RtSceneDesc node index */
    Node   *root;
    std::vector<Light> lights;
    bool    has_ambient;
    Spec    ambient;
    RtCamera cam;
    int     sampler_type, venera;
    std::vector<int> qi2, qi3;
    double  aux_init[6];
    uint32_t prng_x, prng_c;    /* MWC state at Render time (JitterSamplingTile) */
    std::vector<std::vector<float> > image_px;      /* own copies of RtSceneDesc.images */
    std::vector<RtImageMap> images;
    Scene() : root(0), has_ambient(false), sampler_type(0), venera(0), prng_x(886459), prng_c(361290869) {}
    ~Scene() { for (size_t i = 0; i < nodes.size(); i++) delete nodes[i]; }
};

/* spectral constants the shader needs (RadiometryColor.cpp:616-790, :26-37, :336-372), read from tables.bin */
struct Tables {
    bool  loaded;
    Spec  metal_n[11], metal_k[11];     /* indexed by MAT_COPPER..MAT_TITANIUM */
    float spec2rgb[NSPEC][3];
    Spec  chan[3];                      /* g_sRedChannel, g_sGrnChannel, g_sBluChannel */
    std::vector<double> blue;           /* g_rgvBlueNoise, 1600 x 2 */
    std::vector<double> disk128;        /* DiskSamples128, 128 x 2 (LowDiscrepancy.cpp:129) */
    Spec  s_gwhite, s_d6500, s_ocean;                 /* g_sWhite, g_sD6500, s_sOcean: the planet image maps */
    Spec  s_green, s_white, s_orange, s_red, s_blue;  /* g_sGreen, g_sEqualWhite, g_sOrange, g_sRed, g_sBlue: the solid colour textures' palette */
    Tables() : loaded(false) {}
};
extern Tables g_t;

float band_noise(const float *p, int ndim);
float gradient_band_noise(float *g, const float *p, int ndim);
float gradient_voronoi(float *grad, const float *p, int npow);
float fractal_noise_export(const float *p, int ndim, int levels, int absolute);    /* FractalNoise, Texture.cpp:200-241 */
void  fresnel_probe(double fNE, double n1, double n2, int metal_luster, double *F, double *fNT, float *spec10);    /* Shade.cpp:90-153 */
void  apply_texture(int tex, const V3 &p, Hit &h, const Ctx &cx, const RtImageMap *images = 0);
void  transform_normals_export(Hit &h, const Ctx &cx);   /* TransformNormals, Shade.cpp:264-284 */

Ray  camera_emit(const RtCamera &c, double vx, double vy, const Ctx &cx);
Spec visual_ray(Scene &s, const Ray &ray, Ctx &cx);
Spec light_ray(Scene &s, const Ray &ray, const Light &l, Ctx &cx);
Rgb  spectrum_srgb(const Spec &s);
Rgb  spectrum_vrgb(const Spec &s);

} // namespace orc
