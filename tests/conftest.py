import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def entry():
    import __graft_entry__ as g
    return g


@pytest.fixture(scope="session")
def rt(entry):
    """ray-tracer_amd package (ctypes bindings of the C ABI and of the host front end)."""
    return entry.load_package()


@pytest.fixture(autouse=True)
def _library_defaults(request):
    """Every GPU test starts from, and leaves behind, the library's shipped tunables: a test that changes one and fails
    (or restores a wrong value) cannot change what the tests after it exercise."""
    if "gpu" not in request.keywords:
        yield
        return
    import __graft_entry__ as g
    rt = g.load_package()
    global _DEFAULTS
    if _DEFAULTS is None:
        _DEFAULTS = rt.get_options()       # first GPU test of the session: nothing has been set yet
    rt.set_options(_DEFAULTS)
    yield
    rt.set_options(_DEFAULTS)


_DEFAULTS = None


@pytest.fixture(scope="session")
def oracle(entry):
    """CPU oracle binding -- test infrastructure only."""
    return entry.load_oracle()


SCENE_GLASS = """(scene
 (model (makeunion
   (translate -1.2 0 0 (material 0.9 1.474 (scale g_sEmerald 0.01) transparent none coloredglass (surface polished none (sphere))))
   (translate 1.3 0.2 -0.5 (rotate 0.3 0.5 0.1 (material 0.7 1.333 g_sWater transparent none coloredglass (surface polished none (cube)))))
   (translate 0 -0.3 -2.5 (material 0.9 auto g_sGold gold none none (surface polished none (scale 1.5 1.5 1.5 (sphere)))))
   (translate 0 2.2 -1 (scale 8 0.2 8 (material 0.9 auto g_sEqualWhite plastic none none (cube))))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 9 -6.5 12 (scale (blackbody 6500) 8000))
 (pointlight -10 -10 10 (scale (blackbody 6500) 2000))
 (camera stereographic 0 0 8 70)
 (sampler fastblue))"""

SCENE_BUMPY = """(scene
 (model (makeunion
   (translate -1.5 0 0 (rotate 0.2 0.4 0 (surface shiny bumpy (material 0.9 auto g_sRed plastic none none (torus 0.4)))))
   (translate 1.5 0 0 (surface wet microfacets (material 0.9 auto g_sCopper metal none none (scale 1 1.3 1 (sphere)))))
   (translate 0 -1.8 0 (surface smooth none (material 0.9 auto g_sBlue plastic faceted20 none (dodecahedron))))
   (translate 0 1.6 0.5 (surface polished quasimicrofacet (material 0.9 auto g_sSilverRefraction silver none none (cylinder))))
   (translate 0 3.2 -1 (scale 9 0.2 9 (surface burnished fibers (cube))))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 10 -10 10 (scale (blackbody 6500) 10000))
 (camera aperture 2.0 8 0 0 8 50)
 (sampler fastblue 0))"""

SCENE_CSG = """(scene
 (model (union
   (surface burnished none (makeunion
     (translate -2 -2 0 (material 0.9 auto g_sRed metal none none (difference (cube) (scale 0.5 0.5 1.2 (cylinder)))))
     (translate 2 -2 0 (material 0.5 auto g_sGreen plastic none none (intersection (sphere) (hyperboloid1 0.125))))
     (translate -2 2 0 (rotate 0.4 0.2 0 (material 0.9 auto g_sYellow metal none none (union (torus 0.3) (scale 0.6 0.6 0.6 (sphere))))))
     (translate 2 2 0 (material 0.9 auto g_sCyan plastic none none (boundingbox (knoboid 0.5 1.0))))
     (translate 0 0 0 (rotate 0.5 0.5 0 (material 0.7 auto g_sOrange metal none none (angularclip 0.3 2.5 (torus 0.35)))))
     (translate 0 -3.5 0 (scale 0.7 0.7 0.7 (material 0.9 auto g_sMagenta plastic none none (icosahedroncage 0.05 0.1))))
     (translate 0 3.5 0 (scale 0.6 0.6 0.6 (material 0.9 auto g_sWhite plastic none none (cuboid 0.2 1 1 1))))))
   (translate 0 5.25 -10 (scale 12 1 12 (cube)))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 10 -10 10 (scale (blackbody 6500) 10000))
 (pointlight -10 -10 -10 (scale (blackbody 6500) 500))
 (camera stereographic 0 0 9 70)
 (sampler fastblue))"""


# thirteen disjoint spheres in a row along the view axis, distinct densities: rays down the axis collect 26 hits, so the
# hit arena fills up, merges pile dead records under live lists and the arena must be compacted
SCENE_ROW = """(scene
 (model (makeunion
     (translate 0 0 -0 (scale 0.3 0.3 0.3 (material 0.3 auto g_sRed plastic none none (sphere))))
     (translate 0 0 -1 (scale 0.3 0.3 0.3 (material 0.35 auto g_sGreen plastic none none (sphere))))
     (translate 0 0 -2 (scale 0.3 0.3 0.3 (material 0.4 auto g_sYellow plastic none none (sphere))))
     (translate 0 0 -3 (scale 0.3 0.3 0.3 (material 0.45 auto g_sCyan plastic none none (sphere))))
     (translate 0 0 -4 (scale 0.3 0.3 0.3 (material 0.5 auto g_sOrange plastic none none (sphere))))
     (translate 0 0 -5 (scale 0.3 0.3 0.3 (material 0.55 auto g_sMagenta plastic none none (sphere))))
     (translate 0 0 -6 (scale 0.3 0.3 0.3 (material 0.6 auto g_sWhite plastic none none (sphere))))
     (translate 0 0 -7 (scale 0.3 0.3 0.3 (material 0.65 auto g_sRed plastic none none (sphere))))
     (translate 0 0 -8 (scale 0.3 0.3 0.3 (material 0.7 auto g_sGreen plastic none none (sphere))))
     (translate 0 0 -9 (scale 0.3 0.3 0.3 (material 0.75 auto g_sYellow plastic none none (sphere))))
     (translate 0 0 -10 (scale 0.3 0.3 0.3 (material 0.8 auto g_sCyan plastic none none (sphere))))
     (translate 0 0 -11 (scale 0.3 0.3 0.3 (material 0.85 auto g_sOrange plastic none none (sphere))))
     (translate 0 0 -12 (scale 0.3 0.3 0.3 (material 0.9 auto g_sMagenta plastic none none (sphere))))))
 (ambient (scale (blackbody 6500) 0.02))
 (pointlight -10 -10 10 (scale (blackbody 6500) 500))
 (camera stereographic 0 0 9 70)
 (sampler fastblue))"""


# the fractal-noise bump maps (Wrinkled, Dented, Hammered, Tectonic: GradientFractalNoise, TextureMaps.cpp:267-405) and a
# hit that queues two call-outs, the inner one Dented (which may overwrite the perturbed normal without flagging it)
SCENE_FRACTAL = """(scene
 (model (makeunion
   (translate -2.2 -1.2 0 (surface shiny wrinkled (material 0.9 auto g_sRed plastic none none (sphere))))
   (translate 0 -1.2 0 (rotate 0.3 0.1 0 (surface smooth dented (material 0.9 auto g_sGreen plastic none none (cube)))))
   (translate 2.2 -1.2 0 (surface wet hammered (material 0.9 auto g_sCopper metal none none (scale 1 1.2 1 (sphere)))))
   (translate -1.2 1.2 0 (surface smooth tectonic (material 0.9 auto g_sYellow plastic none none (cylinder))))
   (translate 1.2 1.2 0 (surface shiny bumpy (material 0.9 auto g_sCyan plastic dented none (sphere))))
   (translate -3.2 1.2 0 (surface shiny voronoibump (material 0.9 auto g_sWhite plastic none none (sphere))))
   (translate 3.2 1.2 0 (surface shiny voronoicrinkled (material 0.9 auto g_sOrange plastic voronoiantibump none (sphere))))
   (translate 0 3.0 -1 (scale 9 0.2 9 (surface burnished wrinkled (cube))))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 10 -10 10 (scale (blackbody 6500) 10000))
 (camera stereographic 0 0 9 70)
 (sampler fastblue))"""


# the solid colour textures (TextureMaps.cpp:518-683): every one on a plastic solid; Marble on a metal (colour enters the
# BRDF and the reflectance); Clouds on coloured glass (the propagator reads the textured colour); Spots on glass (the
# luster a texture sets decides whether shadow rays pass); SeaScape (no effect) on a surface
SCENE_COLOUR = """(scene
 (model (makeunion
   (translate -3.3 -1.3 0 (surface shiny none (material 0.9 auto g_sWhite plastic voronoi none (sphere))))
   (translate -1.1 -1.3 0 (surface smooth none (material 0.9 auto g_sWhite plastic agate none (rotate 0.3 0.2 0 (cube)))))
   (translate 1.1 -1.3 0 (surface polished none (material 0.9 auto g_sWhite metal marble none (sphere))))
   (translate 3.3 -1.3 0 (surface polished none (material 0.9 1.5 (scale g_sEqualWhite 0.01) transparent clouds coloredglass (sphere))))
   (translate -3.3 1.1 0 (surface burnished none (material 0.9 auto g_sWhite plastic wood none (cylinder))))
   (translate -1.1 1.1 0 (surface polished none (material 0.9 1.5 g_sWhite transparent spots none (sphere))))
   (translate 1.1 1.1 0 (surface shiny seascape (material 0.9 auto g_sWhite plastic veins none (sphere))))
   (translate 3.3 1.1 0 (surface wet none (material 0.9 auto g_sWhite plastic spots none (scale 0.8 1.1 0.8 (sphere)))))
   (translate 0 3.0 -1 (scale 9 0.2 9 (surface matte none (material 0.9 auto g_sEqualWhite plastic clouds none (cube)))))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 10 -10 10 (scale (blackbody 6500) 10000))
 (pointlight -8 -9 12 (scale (blackbody 6500) 4000))
 (camera stereographic 0 0 9 70)
 (sampler fastblue))"""


# Foggy propagator (TextureMaps.cpp:39-57) around and inside glass, seen through each of the other two cameras
MESH_CUBE = ("(mesh 8 12  -1 -1 -1  1 -1 -1  1 1 -1  -1 1 -1  -1 -1 1  1 -1 1  1 1 1  -1 1 1 "
             "  0 2 1  0 3 2  4 5 6  4 6 7  0 1 5  0 5 4  3 6 2  3 7 6  0 4 7  0 7 3  1 2 6  1 6 5)")

# closed triangle meshes as Solids (policy Q14) in every CSG position; no torus and no exp(), so the device must be bit exact
SCENE_MESH = """(scene
 (model (union
   (surface burnished none (makeunion
     (translate -2 -1.6 0 (material 0.9 auto g_sRed metal none none (difference %s (scale 0.5 0.5 1.2 (cylinder)))))
     (translate 2 -1.6 0 (rotate 0.3 0.2 0 (material 0.5 auto g_sGreen plastic none none (intersection (meshicosahedron) (scale 0.9 0.9 0.9 (sphere))))))
     (translate -2 1.6 0 (rotate 0.9 0.3 0 (material 0.9 auto g_sYellow metal none none (meshtorus 24 12 0.35))))
     (translate 2 1.6 0 (material 0.9 1.5 g_sWhite transparent none none (surface polished none (meshsphere 16 9))))
     (scale 0.8 0.8 0.8 (material 0.9 auto g_sCyan plastic none none (union (meshsphere 12 7) (translate 0.7 0 0 (meshicosahedron)))))))
   (translate 0 4 -6 (scale 12 1 12 (cube)))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 10 -10 10 (scale (blackbody 6500) 10000))
 (pointlight -10 -10 10 (scale (blackbody 6500) 800))
 (camera stereographic 0 0 9 70)
 (sampler fastblue))""" % MESH_CUBE

SCENE_FOG_PERSPECTIVE = """(scene
 (model (makeunion
   (translate 0 0 -1 (scale 4 4 4 (material 0.2 1.0 (scale g_sEqualWhite 0.002) transparent none foggy (surface polished none (sphere)))))
   (translate -1.0 0 -1 (material 0.9 auto g_sRed plastic none none (surface shiny none (sphere))))
   (translate 1.2 0.3 -0.5 (material 0.9 1.5 (scale g_sEmerald 0.01) transparent none coloredglass (surface polished none (scale 0.7 0.7 0.7 (sphere)))))
   (translate 0 2.2 -1 (scale 8 0.2 8 (material 0.9 auto g_sEqualWhite plastic none none (cube))))))
 (ambient (scale (blackbody 6500) 0.02))
 (pointlight -10 -10 10 (scale (blackbody 6500) 2000))
 (camera perspective 0 0 8 50)
 (sampler fastblue))"""
SCENE_FOG_ORTHOGRAPHIC = SCENE_FOG_PERSPECTIVE.replace("(camera perspective 0 0 8 50)", "(camera orthographic 0 0 8 4)")
