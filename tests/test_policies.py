"""The stated departures from the reference (policies Q1-Q3 and Q10-Q14), one test each.

Each of these is a place where the reference's behaviour is undefined, leaks state between rays, or does not exist; the
oracle and the device implement the SAME stated policy, so the comparisons below are device == oracle, and every test
also asserts the property that tells the policy from the reference's quirk.  They are EXCLUDED from any claim of parity
with the reference itself: a reader who finds a difference here is looking at a documented policy, not at a bug.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LIGHTS = """(ambient (scale (blackbody 6500) 0.02))
 (pointlight -10 -10 10 (scale (blackbody 6500) 2000))
 (camera stereographic 0 0 8 70) (sampler fastblue)"""


@pytest.fixture(scope="module")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return rt


def _pair(rt, oracle, spec, sampler="test"):
    opt = rt.FrontScene(spec, sampler, optimize=True)
    raw = rt.FrontScene(spec, sampler, optimize=False)
    return opt, raw, rt.GpuScene(opt.desc), oracle.OracleScene(raw.desc)


def _fan(n, z=8.0, spread=0.3, seed=1):
    rng = np.random.default_rng(seed)
    org = np.tile(np.array([0.0, 0.0, z]), (n, 1))
    dr = np.concatenate([rng.uniform(-spread, spread, (n, 2)), -np.ones((n, 1))], axis=1)
    return org, dr


def test_Q1_havercosine_light_is_plain_ambient(gpu, oracle):
    """Q1: AmbientLight::Illuminate is not virtual (RayTracer2020.h:169), so a HavercosineLight is
    never dispatched and acts as plain ambient with flux = sAbove.  S4 (Laboratory) is lit by one: device == oracle there
    (test_laboratory_frames_match_oracle); here: the ambient term does not depend on the surface normal."""
    spec = "(scene (model (material 0.9 auto g_sRed plastic none none (sphere))) (ambient (scale (blackbody 6500) 0.05)) (camera stereographic 0 0 8 70))"
    opt, raw, g, o = _pair(gpu, oracle, spec)
    org, dr = _fan(4000, spread=0.1)
    a, b = g.radiance(org, dr), o.radiance(org, dr)
    assert np.array_equal(a, b)
    lit = a.sum(axis=1) > 0
    assert lit.mean() > 0.5 and np.ptp(a[lit], axis=0).max() == 0.0      # one value whatever the normal


def test_Q2_noshade_medium_does_not_leak_depth(gpu, oracle):
    """Q2 (quirk 2): the reference's early `return pm->color` for MAT_NOSHADE (Shade.cpp:436-437) skips --nRayDepth, so
    after eleven such rays every later ray of the process returns black.  Policy: no leak -- the 5000th ray through a
    NOSHADE medium gets the same radiance as the first."""
    spec = """(scene (model (union (scale 30 30 30 (material 0.001 1.0 g_sBlue noshade none none (hollowsphere)))
                                  (material 0.9 auto g_sRed plastic none none (sphere)))) %s)""" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    n = 5000
    org = np.tile(np.array([0.0, 0.0, 8.0]), (n, 1))
    dr = np.tile(np.array([0.9, 0.1, -1.0]), (n, 1))        # misses the ball, ends in the dome's medium
    a, b = g.radiance(org, dr), o.radiance(org, dr)
    assert np.array_equal(a, b)
    assert a[0].sum() > 0 and np.ptp(a, axis=0).max() == 0.0


def test_Q3_spectral_metals_reflect_with_fresnel_one_and_keep_their_material(gpu, oracle):
    """Q3 (quirk 3): for MAT_COPPER..MAT_TITANIUM the reference never assigns fFresnel (Shade.cpp:446-466; read at :491,
    :494 -- uninitialised) and overwrites the shared Material::color per hit (:447).  Policy: fFresnel = 1.0 (polished
    metals always reflect) and the Fresnel colour belongs to the hit: the same ray gives the same radiance whatever rays
    came before it, and a polished copper ball shows the red ball beside it."""
    spec = """(scene (model (makeunion
       (translate -1.1 0 0 (material 0.9 auto g_sCopper copper none none (surface polished none (sphere))))
       (translate 1.1 0 0 (material 0.9 auto g_sRed plastic none none (surface shiny none (sphere))))
       (translate 0 2.2 -1 (scale 8 0.2 8 (material 0.9 auto g_sEqualWhite plastic none none (cube)))))) %s)""" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    org, dr = _fan(6000, spread=0.35, seed=3)
    a, b = g.radiance(org, dr), o.radiance(org, dr)
    assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
    again = g.radiance(org[::-1].copy(), dr[::-1].copy())[::-1]          # other rays first: no state carried between rays
    assert np.array_equal(a, again)
    st = gpu.RtStats()
    img, st = g.render(64, 64, 1)
    assert st.rays_secondary > 0                                         # the polished metal did spawn reflection rays


def test_Q10_root_slot_read_past_the_end_counts_as_far(gpu, oracle):
    """Q10: HitWork::HitList reads rgfRoots[iFirst] with iFirst == nRoots when every root lies behind the origin
    (Solids.cpp:194-197): an uninitialised slot.  Policy: it reads as "farther than epsilon", so an origin inside a solid
    with no crossing ahead gets the synthetic t = 0 hit -- a ray that leaves a complemented sphere (HollowSphere, solid
    outside) behind it travels in its material."""
    spec = "(scene (model (material 0.9 auto g_sRed plastic none none (hollowsphere))) %s)" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    n = 2000
    rng = np.random.default_rng(2)
    org = np.tile(np.array([0.0, 0.0, 3.0]), (n, 1))
    dr = np.concatenate([rng.uniform(-0.2, 0.2, (n, 2)), np.ones((n, 1))], axis=1)     # away from the sphere: both roots behind
    gr, orr = g.intersect(org, dr, max_hits=4), o.intersect(org, dr, max_hits=4)
    for x, y in zip(gr, orr):
        assert np.array_equal(x, y)
    assert (gr[0] == 1).all() and (gr[1][:, 0] == 0.0).all()            # one synthetic hit at t = 0, entering the material
    assert (gr[4][:, 0] != -1).all()                                     # material id: not vacuum


def test_Q11_hits_without_a_surface_shade_as_matte(gpu, oracle):
    """Q11: a synthetic t = 0 hit has ps == 0 (Solids.cpp:187-192); the reference would dereference it if such a hit were
    shaded as a boundary (Shade.cpp:468).  Policy: a hit without a surface reads as matte.  The merge rules make a t = 0 hit
    the first of its list, where the walk consumes it as the medium the ray starts in (Shade.cpp:424-432), so the policy is
    a guard rather than a visible behaviour: a camera inside a plastic solid gets the surface-less hit, starts in its
    material, shades the way out, and device == oracle."""
    spec = "(scene (model (scale 20 20 20 (material 0.9 auto g_sGreen plastic none none (sphere)))) %s)" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    org, dr = _fan(2000, spread=0.5, seed=4)
    a, b = g.radiance(org, dr), o.radiance(org, dr)
    assert np.array_equal(a, b) and np.isfinite(a).all()
    gr = g.intersect(org, dr, max_hits=4)
    assert (gr[1][:, 0] == 0.0).all() and (gr[5][:, 0] == -9).all()      # first hit: t = 0, no surface


def test_Q12_recorded_affine_of_a_bare_primitive_is_the_static_one(gpu, oracle):
    """Q12: hits record s_paLastHitAffine, a static the last AffineSolid::Intersect left behind (Solids.cpp:170, :213): a
    primitive with no affine of its own inherits whatever affine was visited before it.  Policy: the static (depth-first)
    image of that value, the same in oracle (which keeps the static across rays, as the reference does) and device: the
    bare sphere after a translated one gets the translated one's normal transform, for every ray, whatever came before."""
    spec = """(scene (model (union (translate 2.5 0 0 (scale 1 2 1 (material 0.9 auto g_sRed plastic none none (sphere))))
                                  (material 0.9 auto g_sGreen plastic none none (sphere)))) %s)""" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    org, dr = _fan(6000, spread=0.45, seed=5)
    gr, orr = g.intersect(org, dr, max_hits=6, transform=1), o.intersect(org, dr, max_hits=6, transform=1)
    for x, y in zip(gr, orr):
        assert np.array_equal(x, y)
    back = g.intersect(org[::-1].copy(), dr[::-1].copy(), max_hits=6, transform=1)
    assert np.array_equal(back[2][::-1], gr[2])                          # normals do not depend on the order of the rays
    assert (gr[0] > 0).mean() > 0.1


def test_Q13_texture_colour_belongs_to_the_hit(gpu, oracle):
    """Q13: the solid colour textures write pm->color / pm->nLuster of the SHARED Material (TextureMaps.cpp:530-683), so in
    the reference a ray's colour can leak into the next ray that meets the material without evaluating the texture.  Policy:
    the colour belongs to the hit.  Same rays in reverse order: same radiance; device == oracle to the libm tolerance."""
    spec = """(scene (model (makeunion
       (translate -1.1 0 0 (surface shiny none (material 0.9 auto g_sWhite plastic marble none (sphere))))
       (translate 1.1 0 0 (surface shiny none (material 0.9 auto g_sWhite plastic clouds none (sphere)))))) %s)""" % LIGHTS
    opt, raw, g, o = _pair(gpu, oracle, spec)
    org, dr = _fan(6000, spread=0.35, seed=6)
    a, b = g.radiance(org, dr), o.radiance(org, dr)
    bad = ~np.isclose(a, b, rtol=1e-4, atol=1e-6).all(axis=1)
    assert bad.mean() < 2e-3
    again = g.radiance(org[::-1].copy(), dr[::-1].copy())[::-1]
    assert np.array_equal(a, again)
    lit = a.sum(axis=1) > 0
    assert np.unique(np.round(a[lit], 5), axis=0).shape[0] > 100         # the texture is visible: many colours


def test_Q14_a_mesh_is_a_solid(gpu, oracle):
    """Q14: the reference reads OBJ meshes but never ray-traces them (ObjectFormat.cpp:79-102 is uncalled and wrong
    twice).  Policy: a closed mesh is a Solid whose boundary crossings are its triangle hits.  Pinned on a shape the
    reference has: a cube of twelve triangles gives Cube's hit lists (t to 1e-12, same materials, same counts), outside
    and inside, alone and under Difference; device == oracle bit for bit (tests/test_gpu_parity.py has the large cases)."""
    from conftest import MESH_CUBE
    body = "(difference (material 0.9 auto g_sRed plastic none none %s) (scale 0.5 0.5 2 (cylinder)))"
    lights = LIGHTS
    _, _, gm, om = _pair(gpu, oracle, "(scene (model (rotate 0.3 0.2 0.1 %s)) %s)" % (body % MESH_CUBE, lights))
    _, _, gq, _ = _pair(gpu, oracle, "(scene (model (rotate 0.3 0.2 0.1 %s)) %s)" % (body % "(cube)", lights))
    org, dr = _fan(6000, spread=0.3, seed=7)
    org[3000:] = np.random.default_rng(8).uniform(-0.9, 0.9, (3000, 3)) * np.array([1, 1, 0.3])     # inside, too
    a, b, q = gm.intersect(org, dr, max_hits=8), om.intersect(org, dr, max_hits=8), gq.intersect(org, dr, max_hits=8)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    same = a[0] == q[0]
    assert same.mean() > 0.999                                           # rays through an edge shared by two triangles may differ
    assert np.allclose(a[1][same], q[1][same], rtol=1e-12, atol=1e-12) and np.array_equal(a[4][same], q[4][same])
    assert (a[0] >= 2).mean() > 0.3
