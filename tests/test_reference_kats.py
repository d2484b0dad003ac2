"""The reference's own spot checks (void TestXxx() functions that print numbers) as known-answer tests.

The reference holds no expected outputs for them -- its author read the printed numbers -- so every case here is pinned twice:
by values derived BY HAND from the reference's source (on the CPU, against the oracle and the host front end) and, under -m gpu,
by the HIP path against the oracle on the same inputs:

  TestRefraction   Test.cpp:1529-1549       hit list from an origin inside a rotated transparent cube
  TestLightRayBug  TestScene.cpp:1500-1517  one ray through an emerald glass sphere, two placements of the second light
  TestFresnel      Test.cpp:1467-1515       sweep of the angle for the real, complex and spectral Fresnel laws
  TestBandNoise    Test.cpp:1635-1653       50 points drawn from the MWC generator's default seed
  TestVoronoi      Test.cpp:3798-3836       a slab of a 1/32 grid
  TestClipPlanes   Test.cpp:1822-1830       ClipPlanes(Extent(3.0), rgvCube, 8, 1.0)
"""
import ctypes as C
import math

import numpy as np
import pytest

S6500 = "(blackbody 6500)"
REFRACT_PYREX = 1.474       # Solids.h:243

# TestRefraction (Test.cpp:1538-1543): Surface(POLISHED, Material(0.5, g_sWhite, TRANSPARENT, Rotate(0, 20 deg, 0, Cube)))
REFRACTION = """(scene
 (model (surface polished none (material 0.5 auto g_sWhite transparent none none (rotate 0 %r 0 (cube)))))
 (ambient (scale %s 0.02))
 (pointlight 10 -10 10 (scale %s 10000)))""" % (20.0 * math.pi / 180.0, S6500, S6500)


def _light_bug_scene(x):
    """TestLightRayBug (TestScene.cpp:1500-1517) with MakeLights3(x) (:1489-1498): the light at (x, 0, 0) is created last and so
    heads the list; the scene text prepends like the reference's constructors do."""
    return """(scene
 (model (material 0.9 %r (scale g_sEmerald 0.01) transparent none coloredglass (surface polished none (sphere))))
 (ambient (scale %s 0.02))
 (pointlight 10 -10 10 (scale %s 10000))
 (pointlight %r 0 0 (scale %s 100)))""" % (REFRACT_PYREX, S6500, S6500, x, S6500)


def _densities(fs):
    """RtNode.id -> density of every Material node of a flattened scene (-1: vacuum)."""
    d = fs.desc.contents
    out = {d.nodes[i].id: d.data[d.nodes[i].data] for i in range(d.n_nodes) if d.nodes[i].type == 12}
    out[-1] = 0.0
    return out


def _band_noise_points():
    """TestBandNoise's 50 points: rgf[k] = float(10*RandomDouble()), three draws per point from the default seeds."""
    import __graft_entry__ as g
    ob = g.load_oracle()
    return (10.0 * ob.random_double(150)).astype(np.float32).reshape(50, 3)


def _voronoi_grid():
    """TestVoronoi's lattice: rgf = (i, j, k) / 32 (SCALE, Test.cpp:3772) -- frame i = 0 and a strip of the 1920 x 1080 image."""
    j, k = np.meshgrid(np.arange(0, 1920, 7), np.arange(0, 1080, 9), indexing="ij")
    p = np.stack([np.zeros(j.size), j.ravel() / 32.0, k.ravel() / 32.0], axis=1)
    p2 = p.copy(); p2[:, 0] = 599.0 / 32.0          # the last frame
    return np.concatenate([p, p2]).astype(np.float32)


# ---------------------------------------------------------------------------------------------- CPU: oracle vs values by hand
def test_refraction_hit_list_by_hand(rt, oracle):
    """What TestRefraction prints: the origin (0,0,0) is inside the cube, so HitList prepends a t = 0 hit entering the material
    (Solids.cpp:187-192); the ray (0,0,-1) leaves through the face whose normal it makes 20 degrees with, at t = 1 / cos 20."""
    fs = rt.FrontScene(REFRACTION, "test", optimize=False)
    dens = _densities(fs)
    cnt, t, _, _, mat, _ = oracle.OracleScene(fs.desc).intersect(np.zeros((1, 3)), np.array([[0.0, 0.0, -1.0]]), max_hits=4)
    assert cnt[0] == 2
    assert [dens[int(m)] for m in mat[0, :2]] == [0.5, 0.0]
    assert t[0, 0] == 0.0 and abs(t[0, 1] - 1.0 / math.cos(math.radians(20.0))) < 1e-15
    assert ("t=%0.1f d=%0.1f, t=%0.1f d=%0.1f" % (t[0, 0], 0.5, t[0, 1], 0.0)) == "t=0.0 d=0.5, t=1.1 d=0.0"      # PrintHitList's text


def test_light_ray_bug_hit_list_by_hand(rt, oracle):
    """The ray (0,0,5) -> (0,0,-1) enters the unit glass sphere at t = 4 and leaves at t = 6, both placements alike; the radiance
    is finite, positive and depends on where the second light stands (outside at x = -1.1, inside the glass at x = -0.9)."""
    org, dr = np.array([[0.0, 0.0, 5.0]]), np.array([[0.0, 0.0, -1.0]])
    rad = []
    for x in (29.0 / 10 - 4.0, 31.0 / 10 - 4.0):
        fs = rt.FrontScene(_light_bug_scene(x), "test", optimize=False)
        d = fs.desc.contents
        assert d.n_lights == 2 and abs(d.lights[0].org[0] - x) < 1e-15 and d.lights[1].org[0] == 10.0      # MakeLights3's list order
        o = oracle.OracleScene(fs.desc)
        cnt, t, _, _, mat, _ = o.intersect(org, dr, max_hits=4)
        dens = _densities(fs)
        assert cnt[0] == 2 and list(t[0, :2]) == [4.0, 6.0] and [dens[int(m)] for m in mat[0, :2]] == [0.9, 0.0]
        rad.append(o.radiance(org, dr)[0])
    assert np.isfinite(rad).all() and rad[0].min() > 0 and rad[1].min() > 0
    assert not np.array_equal(rad[0], rad[1])


def test_fresnel_laws_by_hand(rt, oracle):
    """TestFresnel's sweep, against closed forms: normal incidence ((n1 - n2) / (n1 + n2))^2 for the real law and
    ((n - 1)^2 + k^2) / ((n + 1)^2 + k^2) per band for the spectral one (NormalRefl, Test.cpp:1455-1464); grazing incidence and
    every cosine below zero reflect everything; beyond the critical angle from the dense side F = 1 and fNT is left alone."""
    ang = np.arange(0.0, 181.0, 10.0)
    cos = np.cos(np.pi * ang / 180.0)
    F, T, sp = oracle.fresnel_probe(cos, 1.0, 1.5, 4)           # MAT_COPPER
    assert abs(F[0] - 0.04) < 1e-15 and abs(T[0] - 1.0) < 1e-15
    assert np.all(F[ang >= 90.0] >= 1.0 - 1e-12) and np.all(np.diff(F[:10]) > 0)        # rises monotonically to 1 at 90 degrees
    # Snell: n1 sin(a) = n2 sin(b), fNT = cos(b)
    assert np.allclose(T[:10], np.sqrt(1.0 - (np.sin(np.radians(ang[:10])) / 1.5) ** 2), rtol=0, atol=1e-15)
    L = rt.rt_lib()
    L.rt_table_find.restype = C.c_void_p
    L.rt_table_find.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    dt, cnt = C.c_int(), C.c_int()
    n = np.ctypeslib.as_array(C.cast(L.rt_table_find(b"g_sCopperRefraction", C.byref(dt), C.byref(cnt)), C.POINTER(C.c_float)), shape=(10,)).astype(np.float64)
    k = np.ctypeslib.as_array(C.cast(L.rt_table_find(b"g_sCopperExtinction", C.byref(dt), C.byref(cnt)), C.POINTER(C.c_float)), shape=(10,)).astype(np.float64)
    normal = ((n - 1.0) ** 2 + k * k) / ((n + 1.0) ** 2 + k * k)
    assert np.allclose(sp[0], normal, rtol=2e-7) and np.all(sp[ang > 90.0] == 1.0)
    assert np.all(sp[:9] > 0.2) and np.all(sp[:9] <= 1.0)
    # from glass into air: total internal reflection beyond asin(1 / 1.5) = 41.8 degrees
    F2, _, _ = oracle.fresnel_probe(cos[:10], 1.5, 1.0, 0)
    assert abs(F2[0] - 0.04) < 1e-15 and np.all(F2[ang[:10] >= 50.0] == 1.0) and np.all(F2[ang[:10] <= 40.0] < 1.0)
    # equal indices: nothing is reflected (the special case of Shade.cpp:105-106)
    F3, T3, sp3 = oracle.fresnel_probe(cos[:9], 1.3, 1.3, 0)
    assert np.all(F3 == 0.0) and np.allclose(T3, cos[:9], rtol=0, atol=1e-14) and np.all(sp3 == 0.0)


def test_band_noise_points_by_hand(rt, oracle):
    """TestBandNoise: the 50 points come from RandomDouble with the default seeds (MonteCarlo.cpp:9-10) -- the first coordinate by
    hand from the generator's recurrence -- and both noises stay in the range their gain allows; BandNoise at lattice points
    depends on the hash alone: the cubic B-spline weights (1/6, 4/6, 1/6) per axis, checked against a direct evaluation."""
    p = _band_noise_points()
    a, x, c = 1965537969, 886459, 361290869
    draws = []
    for _ in range(2):
        v = a * x + c
        x, c = v & 0xFFFFFFFF, v >> 32
        draws.append(x)
    assert p[0, 0] == np.float32(10.0 * ((draws[0] + draws[1] / 4294967296.0) / 4294967296.0))
    band, frac, vor = oracle.noise_probe2(p)
    assert np.all(np.isfinite(band)) and band.min() > -0.6 and band.max() < 1.6 and 0.3 < band.mean() < 0.7
    assert frac.min() > -0.6 and frac.max() < 1.6 and 0.3 < frac.mean() < 0.7 and np.all(vor >= 0.0) and vor.max() < 3.0

    def lattice(v):                                     # LatticeValue, Texture.cpp:11-16
        v = (1099087573 * v) & 0xFFFFFFFF
        s = v & 31
        return ((v >> s) | (v << (32 - s))) & 0xFFFFFFFF if s else v

    def knot(ix, iy, iz):                               # Spline's recursion (Texture.cpp:34-62): the hash of a dimension seeds the next lower one
        h = lattice((ix ^ 1099087573) & 0xFFFFFFFF)
        h = lattice((iy ^ h) & 0xFFFFFFFF)
        h = lattice((iz ^ h) & 0xFFFFFFFF)
        return np.float32(h) - np.float32(0.5 * 4294967296.0)

    q = np.array([[3.0, 5.0, 2.0]], dtype=np.float32)
    got = oracle.noise_probe2(q)[0][0]
    w = [1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0]
    tot = 0.0
    for i in range(3):
        for j in range(3):
            for k in range(3):
                tot += w[i] * w[j] * w[k] * float(knot(3 - 1 + i, 5 - 1 + j, 2 - 1 + k))
    want = 1.27 ** 3 * tot * (0.9 / 4294967296.0) + 0.5
    assert abs(float(got) - want) < 2e-6, (got, want)


def test_voronoi_grid_by_hand(oracle):
    """TestVoronoi's lattice: the value is the squared distance to the nearest feature point, one per unit cell at cell + hash
    in [0, 1)^3 -- so it never exceeds the 3 of a cell's diagonal (the reference's own remark, Test.cpp:3770), it is continuous
    across cell borders, and a brute-force search over the 27 cells around a point finds the same minimum."""
    p = _voronoi_grid()
    _, _, vor = oracle.noise_probe2(p)
    assert np.all(vor >= 0.0) and vor.max() <= 3.0 and vor.mean() < 0.6
    eps = np.float32(1e-3)
    a = np.array([[2.0 - eps, 7.3, 4.6], [2.0 + eps, 7.3, 4.6]], dtype=np.float32)
    va = oracle.noise_probe2(a)[2]
    assert abs(float(va[0]) - float(va[1])) < 1e-2

    def lattice(v):
        v = (1099087573 * v) & 0xFFFFFFFF
        s = v & 31
        return ((v >> s) | (v << (32 - s))) & 0xFFFFFFFF if s else v

    def feature(nx, ny, nz):                            # VHash3, Texture.cpp:319-332
        f = np.float32(1.0 / 4294967296.0)
        n = lattice((nx ^ 1099087573) & 0xFFFFFFFF)
        n = lattice((ny ^ n ^ 1099087573) & 0xFFFFFFFF)
        n = lattice((nz ^ n ^ 1099087573) & 0xFFFFFFFF)
        x = np.float32(nx) + np.float32(n) * f
        n = lattice(n ^ 1099087573)
        y = np.float32(ny) + np.float32(n) * f
        n = lattice(n ^ 1099087573)
        z = np.float32(nz) + np.float32(n) * f
        return x, y, z

    for q in p[[5, 777, 4242, len(p) - 3]]:
        c = np.floor(q).astype(int)
        best = min((float(q[0] - f[0]) ** 2 + float(q[1] - f[1]) ** 2 + float(q[2] - f[2]) ** 2)
                   for f in (feature(c[0] + i, c[1] + j, c[2] + k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)))
        got = float(oracle.noise_probe2(q.reshape(1, 3))[2][0])
        assert abs(got - best) <= 1e-6 * max(1.0, best), (q, got, best)


def test_clip_planes_by_hand(rt):
    """TestClipPlanes: ClipPlanes(Extent(3.0), rgvCube, 8, 1.0).  As checked in the routine looks at the first plane only
    (Solids.cpp:845).  Plane (1,1,1)/sqrt 3 at distance 1 keeps the corners with x + y + z < sqrt 3 -- (-3,-3,-3) and the three
    with one +3 -- whose box already is the whole extent.  With the plane at -4 only (-3,-3,-3) is kept and joined to the other
    seven; the farthest cut, towards a corner with one +3, lies at -3 + (9 - 4 sqrt 3): the box shrinks to [-3, 6 - 4 sqrt 3]^3.
    At -6 nothing is kept and the extent comes back as it was."""
    assert np.array_equal(rt.clip_planes(3.0, "cube", 1.0), [-3, -3, -3, 3, 3, 3])
    e = rt.clip_planes(3.0, "cube", -4.0)
    assert np.array_equal(e[:3], [-3, -3, -3]) and np.allclose(e[3:], 6.0 - 4.0 * math.sqrt(3.0), rtol=0, atol=1e-14)
    assert np.array_equal(rt.clip_planes(3.0, "cube", -6.0), [-3, -3, -3, 3, 3, 3])
    # the octahedron table's first plane is x < d: the box is cut there
    assert np.allclose(rt.clip_planes(2.0, "octahedron", 0.5), [-2, -2, -2, 0.5, 2, 2], rtol=0, atol=1e-15)


# ---------------------------------------------------------------------------------------------- GPU: HIP path vs oracle
@pytest.fixture(scope="module")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return rt


def _fan(n, seed, org, spread):
    rng = np.random.default_rng(seed)
    o = np.tile(np.asarray(org, dtype=np.float64), (n, 1)) + rng.normal(0, 0.05, (n, 3))
    d = rng.normal(0, 1, (n, 3))
    d[:, 2] = -np.abs(d[:, 2]) - spread
    return o, d


@pytest.mark.gpu
def test_refraction_scene_on_gpu(gpu, oracle):
    """TestRefraction through the C ABI: the printed ray and a fan of rays from inside the cube -- hit lists (no libm call on the
    path) bit for bit, and the radiance the commented-out rt.Shade() would have printed."""
    opt = gpu.FrontScene(REFRACTION, "test", optimize=True)
    raw = gpu.FrontScene(REFRACTION, "test", optimize=False)
    g, o = gpu.GpuScene(opt.desc), oracle.OracleScene(raw.desc)
    org, dr = _fan(4000, 21, (0.0, 0.0, 0.0), 0.2)
    org[0] = 0.0; dr[0] = (0.0, 0.0, -1.0)
    a, b = g.intersect(org, dr, max_hits=6), o.intersect(org, dr, max_hits=6)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[0][0] == 2 and a[1][0, 0] == 0.0 and abs(a[1][0, 1] - 1.0 / math.cos(math.radians(20.0))) < 1e-15
    assert (a[0] == 2).mean() > 0.9                      # from inside: the t = 0 hit and the way out
    ra, rb = g.radiance(org, dr), o.radiance(org, dr)
    assert np.array_equal(ra, rb) and ra[0].min() > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("x", [29.0 / 10 - 4.0, 31.0 / 10 - 4.0], ids=["light_outside", "light_inside"])
def test_light_ray_bug_on_gpu(gpu, oracle, x):
    """TestLightRayBug through the C ABI, both placements of MakeLights3: the ray of the reference and a fan around it.  The glass
    has a ColoredGlass propagator (exp of the device), hence a tolerance on the radiance; hit lists are exact."""
    text = _light_bug_scene(x)
    opt = gpu.FrontScene(text, "test", optimize=True)
    raw = gpu.FrontScene(text, "test", optimize=False)
    g, o = gpu.GpuScene(opt.desc), oracle.OracleScene(raw.desc)
    org, dr = _fan(4000, 22, (0.0, 0.0, 5.0), 4.0)
    org[0] = (0.0, 0.0, 5.0); dr[0] = (0.0, 0.0, -1.0)
    a, b = g.intersect(org, dr, max_hits=4), o.intersect(org, dr, max_hits=4)
    for p, q in zip(a, b):
        assert np.array_equal(p, q)
    assert a[0][0] == 2 and list(a[1][0, :2]) == [4.0, 6.0]
    ra, rb = g.radiance(org, dr), o.radiance(org, dr)
    assert np.isfinite(ra).all() and ra[0].min() > 0.0
    assert np.allclose(ra, rb, rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_fresnel_sweep_on_gpu(gpu, oracle):
    """TestFresnel's sweep (0 .. 180 degrees in steps of 10) and a dense one, for every spectral metal and both directions through
    a glass boundary: +, -, *, / and sqrt only, so the device must agree with the oracle bit for bit."""
    ang = np.concatenate([np.arange(0.0, 181.0, 10.0), np.linspace(0.0, 180.0, 3001)])
    cos = np.cos(np.pi * ang / 180.0)
    for luster in range(4, 11):                          # MAT_COPPER .. MAT_TITANIUM
        for n1, n2 in ((1.0, 1.5), (1.474, 1.0), (1.333, 1.333)):
            F, T, sp = gpu.probe_fresnel(cos, n1, n2, luster)
            oF, oT, osp = oracle.fresnel_probe(cos, n1, n2, luster)
            assert np.array_equal(F, oF) and np.array_equal(T, oT) and np.array_equal(sp, osp)
    assert sp.min() >= 0.0 and sp.max() == 1.0


@pytest.mark.gpu
def test_band_noise_and_voronoi_points_on_gpu(gpu, oracle):
    """TestBandNoise's 50 MWC-seeded points and TestVoronoi's 1/32 lattice: BandNoise, FractalNoise and Voronoi are integer hashes
    and float arithmetic -- bit exact on the device."""
    for p in (_band_noise_points(), _voronoi_grid()):
        a, b = gpu.probe_noise(p), oracle.noise_probe2(p)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert len(set(a[0].tolist())) > 1000


# ---------------------------------------------------------------------------------------------- scene scripts vs counts by hand
# The parity tests hand BOTH sides a scene built by the same front end, so a slip in transcribing one of the reference's scene
# scripts would be invisible to them.  Here the graphs the front end builds (before Optimize) are compared with a census made BY
# HAND from the reference's scripts: nodes by type, material lusters, surface roughnesses, texture and propagator call-outs, and
# the root extent worked out from the script's numbers (the oracle's GetExtent over the serialised graph).
T_QUADRIC, T_POLY, T_TORUS, T_AFFINE, T_UNION, T_INTER, T_DIFF, T_BBOX, T_MAT, T_SURF = 1, 2, 4, 5, 6, 8, 9, 11, 12, 13
C30, S30 = math.cos(math.radians(30.0)), math.sin(math.radians(30.0))
FLOOR_Y3 = ([-12, 2, -22], [12, 4, 2])         # Translate(0, 3, -10, Scale(12, 1, 12, Cube))
I2 = 0.85065080835204                           # Solids.cpp:653

CENSUS = {
    # TestScene.cpp:1380-1424 MakeModelCSystem: a ball of 0.5 and 256 balls, each Translate(Scale(Sphere)) under a plastic Material
    "S0": dict(types={T_QUADRIC: 257, T_AFFINE: 514, T_UNION: 256, T_MAT: 256}, lusters={1: 256}),
    # TestScene.cpp:44-77 MakeModel(30): nine solids of every primitive type under metal Materials, one Surface over them all, a floor
    "S1": dict(types={T_QUADRIC: 9, T_TORUS: 2, T_POLY: 4, T_AFFINE: 22, T_INTER: 2, T_DIFF: 2, T_UNION: 10, T_MAT: 9, T_SURF: 1},
               lusters={2: 9}, rough={2: 1}, poly_sides=[6, 6, 12, 20], extent=([-12, -4, -22], [12, 6.25, 2])),
    # DistributedRayTracing.cpp:56-109: IcosahedronCage(0.05, 0.1) = 30 CylinderFromTo (four affines each) + 12 balls in a BoundingBox
    # (Solids.cpp:1659-1675, :1736-1770), Plastic() around it, a floor slab and a NOSHADE sky dome joined by MakeUnion
    "S2": dict(types={T_QUADRIC: 43, T_POLY: 1, T_AFFINE: 150, T_UNION: 43, T_BBOX: 1, T_MAT: 3, T_SURF: 1}, lusters={1: 2, 0: 1}, rough={5: 1},
               bbox=([-(I2 + 0.1)] * 3, [I2 + 0.1] * 3), extent=([-1000] * 3, [1000] * 3)),
    # TestScene.cpp:79-108: ten tori, SURF_MATTE..SURF_POLISHED x {plastic, metal}, at x = -6 .. 6, y = -2, 2; floor at y = 5.25
    "Model2@0": dict(types={T_TORUS: 10, T_POLY: 1, T_AFFINE: 22, T_UNION: 10, T_MAT: 10, T_SURF: 10}, lusters={1: 5, 2: 5},
                     rough={1: 2, 2: 2, 3: 2, 4: 2, 5: 2}, extent=([-12, -3.5, -22], [12, 6.25, 2])),
    # :110-132: three shiny spheres, Bumpy / Wrinkled / Dented, each rotated by 30 degrees about y, at x = -3, 0, 3
    "Model3@30": dict(types={T_QUADRIC: 3, T_AFFINE: 6, T_UNION: 2, T_MAT: 3, T_SURF: 3}, lusters={1: 3}, rough={4: 3}, tex={1: 1, 4: 1, 6: 1},
                      extent=([-3 - C30 - S30, -1, -C30 - S30], [3 + C30 + S30, 1, C30 + S30])),
    # :137-164, x <= 60: a matte ball at -1.5 and a bumpy burnished one at 1.5 - 2.5 x / 60
    "Model4@30": dict(types={T_QUADRIC: 2, T_AFFINE: 2, T_UNION: 1, T_MAT: 2, T_SURF: 2}, rough={1: 1, 2: 1}, tex={1: 1},
                      extent=([-2.5, -1, -1], [1.25, 1, 1])),
    # x > 60: the matte ball minus a bumpy one at -1, and the moving ball at -1 + 2.5 (x - 60) / 60
    "Model4@90": dict(types={T_QUADRIC: 3, T_AFFINE: 3, T_UNION: 1, T_DIFF: 1, T_MAT: 3, T_SURF: 3}, rough={1: 1, 2: 2}, tex={1: 2},
                      extent=([-2.5, -1, -1], [1.25, 1, 1])),
    # :166-194: gold, bumpy copper, transparent and plastic polished balls, turned by 30 degrees about y as a group; floor at y = 3
    "Model5@30": dict(types={T_QUADRIC: 4, T_POLY: 1, T_AFFINE: 11, T_UNION: 4, T_MAT: 4, T_SURF: 4}, lusters={2: 2, 3: 1, 1: 1}, rough={5: 4}, tex={1: 1},
                      extent=([-12, -1, -22], [12, 4, 4 * S30 + C30])),
    # :196-233: three water cubes, a copper, a Pyrex and an air ball of radius 0.5
    "Model6": dict(types={T_POLY: 4, T_QUADRIC: 3, T_AFFINE: 18, T_UNION: 6, T_MAT: 6, T_SURF: 6}, lusters={3: 5, 2: 1}, rough={5: 6}, prop={1: 3}),
    # :235-244: the transparent cube of TestRefraction over the floor
    "Model7@30": dict(types={T_POLY: 2, T_AFFINE: 3, T_UNION: 1, T_MAT: 1, T_SURF: 1}, lusters={3: 1}, rough={5: 1},
                      extent=([-12, -1, -22], [12, 4, 2])),
    # :246-284: six balls, the middle front one bumpy water under a Scale(1, 1, 1)
    "Model8@30": dict(types={T_QUADRIC: 6, T_POLY: 1, T_AFFINE: 16, T_UNION: 6, T_MAT: 6, T_SURF: 6}, lusters={1: 5, 3: 1}, rough={4: 5, 5: 1}, tex={1: 1}, prop={1: 1}),
    # :286-314: a tumbler (cylinder minus cylinder, torus rim), two lumps of bumpy ice, cut open by a cube; the whiskey is built and left out
    "Model9": dict(types={T_QUADRIC: 4, T_TORUS: 1, T_POLY: 2, T_AFFINE: 13, T_UNION: 4, T_DIFF: 2, T_MAT: 2, T_SURF: 3}, lusters={3: 2}, rough={5: 3}, tex={1: 1}, prop={1: 1}),
    # :316-342: a Pyrex and a water cube, four balls sliding through them
    "ModelA@30": dict(types={T_POLY: 3, T_QUADRIC: 4, T_AFFINE: 13, T_UNION: 6, T_MAT: 6, T_SURF: 6}, lusters={3: 5, 2: 1}, rough={5: 6}, prop={1: 2}),
    # :344-353
    "ModelB": dict(types={T_QUADRIC: 1, T_AFFINE: 1}, extent=([2, -1, -1], [4, 1, 1])),
}


def _count(values):
    out = {}
    for v in values:
        out[v] = out.get(v, 0) + 1
    return out


@pytest.mark.parametrize("name", sorted(CENSUS))
def test_scene_scripts_against_a_census_by_hand(rt, oracle, name):
    want = CENSUS[name]
    fs = rt.FrontScene(name, "test", optimize=False)
    d = fs.desc.contents
    nodes = [d.nodes[i] for i in range(d.n_nodes)]
    assert _count(n.type for n in nodes) == want["types"]
    mats = [n for n in nodes if n.type == T_MAT]
    surfs = [n for n in nodes if n.type == T_SURF]
    if "lusters" in want:
        assert _count(m.i0 for m in mats) == want["lusters"]
    if "rough" in want:
        assert _count(s.i0 for s in surfs) == want["rough"]
    assert _count(n.i1 for n in mats + surfs if n.i1) == want.get("tex", {})
    assert _count(m.i2 for m in mats if m.i2) == want.get("prop", {})
    if "poly_sides" in want:
        assert sorted(n.i0 for n in nodes if n.type == T_POLY) == want["poly_sides"]
    if "bbox" in want:
        (bb,) = [n for n in nodes if n.type == T_BBOX]
        assert np.allclose([d.data[bb.data + k] for k in range(6)], np.concatenate(want["bbox"]), rtol=0, atol=1e-12)
    if "extent" in want:
        e = oracle.OracleScene(fs.desc, run_optimize=False).node_extent(d.root)
        assert np.allclose(e, np.concatenate(want["extent"]), rtol=0, atol=1e-12), e


def test_scene_script_numbers_by_hand(rt):
    """Numbers the scripts put into their nodes.  S0: every small ball sits ON the unit sphere (Translate(v3 / Norm(v3)),
    TestScene.cpp:1411) with radius 0.05, ball 0 inside it with radius 0.1, the centre ball has radius 0.5; one nNoShadow area light
    at (100, -100, 100) (MakeLightsLorenz, :1472-1487).  S2: EnvironmentLab's lights in list order -- the point light is created
    last and heads the list -- and the aperture camera of DistributedRayTracing.cpp:128.  Model2: the tori's positions."""
    keep = [rt.FrontScene("S0", "fastblue", optimize=False), rt.FrontScene("S2", "", optimize=False), rt.FrontScene("Model2@0", "test", optimize=False)]      # the descriptions point into them
    d = keep[0].desc.contents
    aff = [d.nodes[i] for i in range(d.n_nodes) if d.nodes[i].type == T_AFFINE]
    tr = np.array([[d.data[a.data + 18 + k] for k in range(3)] for a in aff])
    diag = np.array([[d.data[a.data + 4 * k] for k in range(3)] for a in aff])
    is_tr = (diag == 1.0).all(axis=1)
    assert is_tr.sum() == 257 and (~is_tr).sum() == 257
    norms = np.linalg.norm(tr[is_tr], axis=1)
    assert (np.abs(norms - 1.0) < 1e-12).sum() == 255 and (norms == 0.0).sum() == 1 and ((norms > 0) & (norms < 1.0 - 1e-9)).sum() == 1
    assert _count(tuple(x) for x in diag[~is_tr]) == {(0.05, 0.05, 0.05): 255, (0.1, 0.1, 0.1): 1, (0.5, 0.5, 0.5): 1}
    assert d.n_lights == 1 and d.lights[0].type == 1 and d.lights[0].no_shadow == 1 and list(d.lights[0].org) == [100.0, -100.0, 100.0]
    assert d.has_ambient and d.camera.type == 1 and list(d.camera.org) == [0.0, 0.0, 8.0] and d.camera.focal == 70.0
    d = keep[1].desc.contents
    assert d.n_lights == 2 and d.lights[0].type == 0 and list(d.lights[0].org) == [10.0, -10.0, -25.0]
    assert d.lights[1].type == 1 and list(d.lights[1].org) == [9.0, -6.5, 12.0] and d.lights[0].no_shadow == 0 == d.lights[1].no_shadow
    assert d.camera.type == 3 and list(d.camera.org) == [0.0, 0.0, 9.0] and d.camera.focal == 35.0 and d.camera.focus_distance == 24.0
    assert d.sampler.type == 1 and [d.sampler.qindex2[i] for i in range(1600)] == list(range(1600))      # QIndex1(0): the identity (Sampling.cpp:543-544)
    d = keep[2].desc.contents
    aff = [d.nodes[i] for i in range(d.n_nodes) if d.nodes[i].type == T_AFFINE]
    tr = sorted(tuple(d.data[a.data + 18 + k] for k in range(3)) for a in aff if d.nodes[a.left].type == T_AFFINE and d.nodes[d.nodes[a.left].left].type == T_TORUS)
    assert tr == sorted((x, y, 0.0) for x in (-6.0, -3.0, 0.0, 3.0, 6.0) for y in (-2.0, 2.0))
