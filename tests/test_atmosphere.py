"""The Atmosphere propagator (BASELINE configs[4] "with Atmosphere scattering").

NEW BEHAVIOUR, parity unpinned by construction: the reference's `Atmosphere` is an empty stub that returns an uninitialised
struct (TextureMaps.cpp:162-168) and its `Rayleigh()` stores nothing (:123-149).  What is built fills the Propagator slot
(Solids.h:267-274) with Rayleigh extinction from the constants of that function -- K = 1e36 * 1e3 * 24 pi^3 *
((n^2-1)/(n^2+2))^2 / N per km over the wavelength in nm to the fourth, N from n = sqrt(1 + alpha N) with
POLARIZABILITY_AIR, the material's colour as a per-band scale, scene units metres -- and airlight (1 - T) x the scene's
ambient radiance.  The CPU test checks the oracle against that formula evaluated here; the GPU tests check the HIP path
against the oracle (exp() is the device's: tolerance, not equality).
"""
import numpy as np
import pytest

AIR = """(scene
 (model (material 0.000293 %s (scale g_sEqualWhite %g) transparent none atmosphere (surface polished none (scale %g %g %g (sphere)))))
 (ambient (scale g_sEqualWhite 0.5))
 (camera stereographic 0 0 8 70) (sampler fastblue))"""

HAZE = """(scene
 (model (makeunion
   (scale 60 60 60 (material 0.000293 1.000293 (scale g_sEqualWhite 4000) transparent none atmosphere (sphere)))
   (translate -1.5 0 -2 (material 0.9 auto g_sRed plastic none none (surface shiny none (sphere))))
   (translate 1.5 0 -14 (material 0.9 auto g_sRed plastic none none (surface shiny none (sphere))))
   (translate 0 0.5 -6 (material 0.9 1.5 (scale g_sEmerald 0.01) transparent none coloredglass (surface polished none (scale 0.8 0.8 0.8 (sphere)))))
   (translate 0 2.2 -8 (scale 14 0.2 14 (material 0.9 auto g_sEqualWhite plastic none none (cube))))))
 (ambient (scale (blackbody 6500) 0.05))
 (pointlight -10 -10 10 (scale (blackbody 6500) 2000))
 (arealight 1.0 9 -6.5 12 (scale (blackbody 6500) 3000))
 (camera stereographic 0 0 8 70) (sampler fastblue))"""


def _expected_tau(n, scale, metres):
    n2 = n * n
    ll = (n2 - 1.0) / (n2 + 2.0)
    N = (n2 - 1.0) / 2.133e-29
    K = 1.0e36 * 1.0e3 * (24.0 * np.pi ** 3) * ll * ll / N
    lam = 380.0 + 40.0 * np.arange(10)
    return scale * K / lam ** 4 * metres / 1000.0


@pytest.mark.parametrize("n,scale,radius", [(1.000293, 1000.0, 40.0), (1.014716, 1.0, 900.0), (1.000293, 1.0, 8000.0)])
def test_rayleigh_extinction_and_airlight_follow_the_stated_formula(rt, oracle, n, scale, radius):
    """A ray from the centre of a ball of air under ambient light only.  The propagator returns T = exp(-tau) and the
    sterisent (1 - T) x ambient; VisualRay adds the sterisent and then multiplies the sum by the segment's transmittance
    (Shade.cpp:441, :503 -- the slot's semantics, the same for Foggy), so the radiance is T (1 - T) x ambient, tau per band
    from the formula above (blue scatters (740/380)^4 = 14.4 times as much as red); Earth's air comes out at 0.0105 per km at
    540 nm, the textbook value."""
    fs = rt.FrontScene(AIR % (repr(n), scale, radius, radius, radius), "", optimize=False)
    o = oracle.OracleScene(fs.desc)
    rad = o.radiance(np.array([[0.0, 0.0, 0.0]]), np.array([[0.3, -0.2, 1.0]]) / np.sqrt(1.13))[0].astype(np.float64)
    tau = _expected_tau(n, scale, radius)
    want = np.exp(-tau) * (1.0 - np.exp(-tau)) * 0.5
    # the polished boundary adds Fresnel-weighted terms of order ((n-1)/2)^2 <= 6e-5
    assert np.allclose(rad, want, rtol=2e-4, atol=1e-7), (rad, want)
    assert abs(tau[0] / tau[9] - (740.0 / 380.0) ** 4) < 1e-9
    if scale == 1.0 and n < 1.001:
        assert abs(_expected_tau(n, 1.0, 1000.0)[4] - 0.0113) < 1.5e-3      # per km at 540 nm


def test_atmosphere_is_named_in_the_scene_api(rt):
    fs = rt.FrontScene("S4atm", "", optimize=True)
    d = fs.desc.contents
    props = [d.nodes[i].i2 for i in range(d.n_nodes) if d.nodes[i].type == 12]        # RT_NODE_MATERIAL: i2 = RtPropagatorId
    assert d.n_nodes == rt.FrontScene("S4", "", optimize=True).desc.contents.n_nodes + 4
    assert 3 in props                                                                  # RT_PROP_ATMOSPHERE


@pytest.fixture(scope="module")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return rt


def _rel_rmse(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(1e-30, np.sqrt((b ** 2).mean())))


@pytest.mark.gpu
def test_atmosphere_radiance_matches_oracle(gpu, oracle):
    """Rays through haze: visual segments, shadow rays through the medium (Shade.cpp:333-336), glass inside it."""
    opt = gpu.FrontScene(HAZE, "", optimize=True); raw = gpu.FrontScene(HAZE, "", optimize=False)
    g = gpu.GpuScene(opt.desc); o = oracle.OracleScene(raw.desc)
    rng = np.random.default_rng(5)
    n = 4000
    org = np.tile(np.array([0.0, 0.0, 8.0]), (n, 1)) + rng.normal(0, 0.05, (n, 3))
    dr = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.35, 0.35, n), -np.ones(n)], 1)
    aux = rng.uniform(-1, 1, (n, 6))
    a = g.radiance(org, dr, aux); b = o.radiance(org, dr, aux)
    assert b.mean() > 1e-3 and np.abs(a - b).max() <= 2e-5 * max(1.0, float(np.abs(b).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", [HAZE, "S4atm", "VeneraAtmosphere@30"], ids=["haze", "S4atm", "VeneraAtmosphere"])
def test_atmosphere_frames_match_oracle(gpu, oracle, name):
    opt = gpu.FrontScene(name, "fastblue", optimize=True); raw = gpu.FrontScene(name, "fastblue", optimize=False)
    g = gpu.GpuScene(opt.desc); o = oracle.OracleScene(raw.desc)
    w, h, spp = 96, 54, 4
    img, st = g.render(w, h, spp)
    ost = gpu.RtStats()
    ref, _, n = o.render(w, h, spp, gpu.RtRenderDesc(w, h, float(spp), 0, 0), ost, threads=8)
    assert st.samples == n and st.hit_overflow == 0
    assert abs(st.rays() - ost.rays()) <= max(4, 1e-4 * ost.rays())
    assert _rel_rmse(img, ref) <= 1e-5 and np.abs(img - ref).max() <= 1e-3 * max(1.0, float(np.abs(ref).max()))
    # and the haze is there: the same frame without the propagator is a different picture
    if name == "S4atm":
        fs = gpu.FrontScene("S4", "fastblue", optimize=True)       # (kept alive: the description points into it)
        clear, _ = gpu.GpuScene(fs.desc).render(w, h, spp)
        assert _rel_rmse(img, clear) > 0.05
