"""CPU tests (no GPU): the oracle against the reference's own golden data, the host front end, and the
C-ABI library's symbol table.  Run with -m "not gpu"."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_quasi_index_matches_reference_tables(rt, oracle):
    """BuildQuasiIndex (Patterns.cpp:1723-1736) vs the reference's checked-in outputs
    (LowDiscrepancy.cpp:262-333, BlueNoiseTile.cpp:1607-2417): oracle AND product, bit exact."""
    tabs = json.load(open(os.path.join(GOLD, "quasi_index_tables.json")))["tables"]
    assert len(tabs) == 8
    for key, want in tabs.items():
        p = int(re.search(r"_p(\d+)_", key).group(1))
        n = int(re.search(r"_N(\d+)$", key).group(1))
        want = np.array(want, dtype=np.int32)
        assert (oracle.build_quasi_index(p, n) == want).all(), key
        assert (rt.build_quasi_index(p, n) == want).all(), key


@pytest.mark.parametrize("n", [1, 2, 7, 16, 125, 1000])
def test_quasi_index_is_permutation(rt, oracle, n):
    for p in (2, 3, 5, 7):
        a = rt.build_quasi_index(p, n)
        assert sorted(a.tolist()) == list(range(n))
        assert (a == oracle.build_quasi_index(p, n)).all()


def test_mwc_generator_product_vs_oracle(rt, oracle):
    """RandomUnsigned (MonteCarlo.h:17-27), default seeds (MonteCarlo.cpp:9-10): integer, bit exact."""
    n = 4096
    st1 = np.array([886459, 361290869], dtype=np.uint32)
    st2 = st1.copy()
    a = np.zeros(n, dtype=np.uint32)
    b = np.zeros(n, dtype=np.uint32)
    rt.rt_lib().rt_random_unsigned(st1.ctypes.data_as(C.POINTER(C.c_uint32)), n, a.ctypes.data_as(C.POINTER(C.c_uint32)))
    oracle.lib().orc_random_unsigned(st2.ctypes.data_as(C.POINTER(C.c_uint32)), n, b.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert (a == b).all() and (st1 == st2).all()
    # first value by hand: (1965537969 * 886459 + 361290869) mod 2^32
    assert int(a[0]) == (1965537969 * 886459 + 361290869) & 0xFFFFFFFF
    # RandomDouble of the front end = (hi + lo/2^32)/2^32 of consecutive draws
    d = np.zeros(8)
    rt.frontend_lib().rtf_random_double(886459, 361290869, 8, d.ctypes.data_as(C.POINTER(C.c_double)))
    want = (a[0:16:2].astype(np.float64) + a[1:16:2].astype(np.float64) * (1.0 / 4294967296.0)) * (1.0 / 4294967296.0)
    assert (d == want).all()


SURVEY = json.load(open(os.path.join(GOLD, "survey_checksums.json")))["cases"]


@pytest.mark.parametrize("case", [c for c in SURVEY if c["spp"] == 1], ids=lambda c: "%s_%d_%dspp" % (c["scene"], c["width"], c["spp"]))
def test_oracle_reproduces_reference_checksums_1spp(rt, oracle, case):
    """The oracle (own Optimize + recursive tracer) on the scenes built by the front end vs the image
    checksums and ray counts the survey recorded from the reference itself."""
    _check_case(rt, oracle, case)


@pytest.mark.parametrize("case", [c for c in SURVEY if c["spp"] == 16], ids=lambda c: "%s_%d_%dspp" % (c["scene"], c["width"], c["spp"]))
def test_oracle_reproduces_reference_checksums_fastblue(rt, oracle, case):
    """BASELINE configs[0] exactly (512 x 512, 16 spp FastBlue): S1 (soft shadows, every primitive of the zoo) and S0 (the
    as-checked-in 257-sphere scene, brute-force union in the oracle as in the reference: about 40 s of CPU on 8 threads)."""
    _check_case(rt, oracle, case)


def _check_case(rt, oracle, case):
    fs = rt.FrontScene(case["scene"], case["sampler"], optimize=False)
    osc = oracle.OracleScene(fs.desc)
    assert osc.check_bounding_boxes() == 0
    rd = rt.RtRenderDesc(case["width"], case["height"], float(case["spp"]), 0, 0)
    st = rt.RtStats()
    img, _, n = osc.render(case["width"], case["height"], case["spp"], rd, st, threads=8)
    assert abs(float(img.astype(np.float64).sum()) - case["sum"]) < 1e-6
    if case["rays"] is not None:
        assert st.rays() == case["rays"]


def test_oracle_threads_do_not_change_the_image(rt, oracle):
    fs = rt.FrontScene("S1", "fastblue", optimize=False)
    osc = oracle.OracleScene(fs.desc)
    rd = rt.RtRenderDesc(48, 40, 4.0, 0, 0)
    a, _, _ = osc.render(48, 40, 4, rd, rt.RtStats(), threads=1)
    b, _, _ = osc.render(48, 40, 4, rd, rt.RtStats(), threads=5)
    assert (a == b).all()


def test_frontend_optimize_matches_oracle_optimize(rt, oracle, ):
    """Front end's Optimize() (affine chain merge, Solids.cpp:957-983) vs the oracle's own: identical hit lists."""
    from conftest import SCENE_CSG
    raw = rt.FrontScene(SCENE_CSG, "test", optimize=False)
    opt = rt.FrontScene(SCENE_CSG, "test", optimize=True)
    assert opt.desc.contents.n_nodes < raw.desc.contents.n_nodes      # chains were merged
    a = oracle.OracleScene(raw.desc, run_optimize=True)
    b = oracle.OracleScene(opt.desc, run_optimize=False)
    rng = np.random.default_rng(5)
    n = 4000
    org = np.tile(np.array([0.0, 0.0, 9.0]), (n, 1)) + rng.normal(0, 0.05, (n, 3))
    dr = np.concatenate([rng.uniform(-0.6, 0.6, (n, 2)), -np.ones((n, 1))], axis=1)
    ra = a.intersect(org, dr, max_hits=12, transform=1)
    rb = b.intersect(org, dr, max_hits=12, transform=1)
    for x, y in zip(ra, rb):
        assert np.array_equal(x, y)
    assert ra[0].max() >= 4


def test_extents_follow_reference_quirks(rt):
    """GetExtent of the front end: TestExtents cases (Test.cpp:1557-1581)."""
    L = rt.frontend_lib()
    out = np.zeros(6)

    def ext(txt):
        assert L.rtf_extent(txt.encode(), out.ctypes.data_as(C.POINTER(C.c_double))) == 0, L.rtf_last_error()
        return out.copy()
    assert (ext("(icosahedron)") == [-1, -1, -1, 1, 1, 1]).all()          # unit cube regardless (Solids.cpp:789)
    assert (ext("(torus 0.5)") == [-1.5, -1.5, -0.5, 1.5, 1.5, 0.5]).all()
    assert (ext("(translate 0 3 0 (scale 1 5 1 (sphere)))") == [-1, -2, -1, 1, 8, 1]).all()
    assert (ext("(intersection (sphere) (translate 0 1 0 (sphere)))") == [-1, 0, -1, 1, 1, 1]).all()
    assert (ext("(boundingbox (union (sphere) (translate 0 3 0 (sphere))))") == [-1, -1, -1, 1, 4, 1]).all()


def test_test_material_hit_lists(rt, oracle):
    """TestMaterial (Test.cpp:1242-1271): (t, density) lists of union / intersection / difference of a
    rotated torus (density 0.9) and a sphere (0.5) along the ray (0,0,-5) -> (0,0,1)."""
    body = "(material 0.9 auto g_sYellow plastic none none (rotate 1.5707963267948966 0 0 (torus 0.5))) (material 0.5 auto g_sBlue plastic none none (sphere))"
    org = np.array([[0.0, 0.0, -5.0]])
    dr = np.array([[0.0, 0.0, 1.0]])
    got = {}
    for op in ("union", "intersection", "difference"):
        fs = rt.FrontScene("(scene (model (%s %s)))" % (op, body), "test", optimize=False)
        ids = {fs.desc.contents.nodes[i].id: fs.desc.contents.data[fs.desc.contents.nodes[i].data]
               for i in range(fs.desc.contents.n_nodes) if fs.desc.contents.nodes[i].type == 12}
        ids[-1] = 0.0
        cnt, t, _, _, mat, _ = oracle.OracleScene(fs.desc).intersect(org, dr, max_hits=8)
        got[op] = [(round(float(t[0, k]), 9), ids[int(mat[0, k])]) for k in range(cnt[0])]
    assert got["union"] == [(3.5, 0.9), (4.5, 0.5), (5.5, 0.9), (6.5, 0.0)]
    assert got["intersection"] == [(4.0, 0.5), (4.5, 0.0), (5.5, 0.5), (6.0, 0.0)]
    assert got["difference"] == [(3.5, 0.9), (4.0, 0.0), (6.0, 0.9), (6.5, 0.0)]


def test_c_abi_exports_every_declared_symbol(rt):
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    names = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", hdr))
    assert {"rt_render", "rt_render_device", "rt_scene_upload", "rt_scene_free", "rt_last_error"} <= names
    L = rt.rt_lib()
    for n in sorted(names):
        assert hasattr(L, n), n


def test_upload_without_gpu_fails_loudly(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    fs = rt.FrontScene("S1", "test", optimize=True)
    with pytest.raises(rt.RtError) as e:
        rt.GpuScene(fs.desc)
    assert "no HIP device" in str(e.value) or "CPU path" in str(e.value)


def test_bmp_restatement_layout_and_gamma(oracle):
    """Image::WriteBMP restated: header fields (Image.cpp:330-414), bottom-up BGR lines padded to 4 bytes, and the
    sRGB encode table: 0 -> 0, 1 -> 255, monotone, every code used, linear 0.2140 = ((0.5 + 0.055)/1.055)^2.4 -> 127 or 128."""
    import struct
    img = np.zeros((3, 2, 3), dtype=np.float32)
    img[0, 0, 0] = 1.0          # red, top-left
    img[2, 1, 2] = 1.0          # blue, bottom-right
    b = oracle.bmp_encode(img)
    assert len(b) == 54 + 2 * 12 and b[:2].tobytes() == b"BM"
    size, _, off, hdr, w, h, planes, bits = struct.unpack("<IIIIIIHH", b[2:30].tobytes())
    assert (size, off, hdr, w, h, planes, bits) == (78, 54, 40, 3, 2, 1, 24)
    assert struct.unpack("<II", b[38:46].tobytes()) == (11811, 11811)          # 300 dpi
    first_line, second_line = b[54:66], b[66:78]
    assert list(first_line[6:9]) == [255, 0, 0] and list(second_line[0:3]) == [0, 0, 255]      # file line 0 = image row 1
    assert list(first_line[9:12]) == [0, 0, 0]                                                  # padding
    ramp = np.linspace(0.0, 1.0, 4096, dtype=np.float32)
    enc = oracle.bmp_encode(np.stack([ramp.reshape(1, -1)] * 3))[54:54 + 3 * 4096:3]
    assert enc[0] == 0 and enc[-1] == 255 and np.all(np.diff(enc.astype(int)) >= 0) and len(set(enc.tolist())) == 256
    grey = np.full((3, 1, 4), 0.2140, dtype=np.float32)
    assert oracle.bmp_encode(grey)[54] in (127, 128)


def test_y4m_restatement(rt, oracle):
    """Video::WriteFrame restated (Video.cpp:49-80): "FRAME\\n", then the Y, Cb, Cr planes in image row order; black is
    (16, 128, 128), white (235, 128, 128), primaries at the BT.601 studio-range corners; gamma 2 is a square root.  The
    header (host-only entry point) is the text NewVideo writes and sizes that are not multiples of 4 are refused."""
    img = np.zeros((3, 4, 4), dtype=np.float32)
    img[:, 0, 1] = 1.0                              # white
    img[0, 0, 2] = 1.0                              # red
    img[1, 0, 3] = 1.0                              # green
    img[2, 1, 0] = 1.0                              # blue
    img[:, 1, 1] = 0.25                             # sqrt -> 0.5 grey
    f = oracle.y4m_encode_frame(img)
    assert len(f) == 6 + 48 and f[:6].tobytes() == b"FRAME\n"
    Y, Cb, Cr = (f[6 + 16 * k:6 + 16 * (k + 1)].reshape(4, 4) for k in range(3))
    assert (Y[0, 0], Cb[0, 0], Cr[0, 0]) == (16, 128, 128)
    assert (Y[0, 1], Cb[0, 1], Cr[0, 1]) == (235, 128, 128)
    assert (Y[0, 2], Cb[0, 2], Cr[0, 2]) == (81, 90, 240)
    assert (Y[0, 3], Cb[0, 3], Cr[0, 3]) == (144, 53, 34)
    assert (Y[1, 0], Cb[1, 0], Cr[1, 0]) == (40, 240, 109)
    assert (Y[1, 1], Cb[1, 1], Cr[1, 1]) == (int(16 + 219 * 0.5), 128, 128)
    assert rt.y4m_header(1280, 720) == b"YUV4MPEG2 W1280 H720 F30:1 Ip A1:1 C444\n"
    with pytest.raises(rt.RtError):
        rt.y4m_header(1281, 720)


def test_random_skip_equals_stepping(rt):
    """rt_random_skip (MWC as multiplication by a modulo a 2^32 - 1) lands where RandomUnsigned would after n draws."""
    import ctypes as C
    L = rt.rt_lib()
    L.rt_random_skip.argtypes = [C.POINTER(C.c_uint32), C.c_uint64]
    L.rt_random_unsigned.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
    for n in (0, 1, 2, 255, 256, 100003):
        a = (C.c_uint32 * 2)(886459, 361290869)
        b = (C.c_uint32 * 2)(886459, 361290869)
        out = (C.c_uint32 * max(n, 1))()
        L.rt_random_unsigned(a, n, out)
        L.rt_random_skip(b, n)
        assert (a[0], a[1]) == (b[0], b[1]), n


def test_jitter_tile_restatement(rt, oracle):
    """JitterSamplingTile: one sample per supersample pixel inside its own pixel, y drawn before x, and an image that
    agrees with the blue-noise tile's up to sampling noise."""
    raw = rt.FrontScene("S0", "jitter", optimize=False)
    osc = oracle.OracleScene(raw.desc)
    v, tot = osc.jitter_samples(40, 24, 4, 0, 80 * 48)
    assert tot == 80 * 48
    i = np.arange(tot) % 80
    j = np.arange(tot) // 80
    assert np.all((v[:, 0] * 80 >= i) & (v[:, 0] * 80 < i + 1))
    assert np.all((v[:, 1] * 80 - (80 - 48) // 2 >= j) & (v[:, 1] * 80 - (80 - 48) // 2 < j + 1))
    rd = rt.RtRenderDesc(40, 24, 4.0, 0, 0)
    a, _, n = osc.render(40, 24, 4, rd, rt.RtStats(), threads=2)
    assert n == tot
    fbs = rt.FrontScene("S0", "fastblue", optimize=False)      # keep it alive: the description points into it
    fb = oracle.OracleScene(fbs.desc)
    b, _, _ = fb.render(40, 24, 4, rd, rt.RtStats(), threads=2)
    assert np.abs(a - b).mean() < 0.02 and np.abs(a).mean() > 0.001


def test_bluenoise_tile_restatement(rt, oracle):
    """BlueNoiseSamplingTile restated: ceil(W/fSqrt) x ceil(H/fSqrt) tiles of 1600 samples, positions inside their
    tile, auxiliary dimensions inside the unit disk and repeating with period 128; the image agrees with the FastBlue
    tile's up to sampling noise."""
    raw = rt.FrontScene("S0", "bluenoise", optimize=False)
    osc = oracle.OracleScene(raw.desc)
    w, h, spp = 50, 30, 4
    o8, xy, tot = osc.bluenoise_samples(w, h, spp, 0, 3 * 2 * 1600)
    assert tot == 3 * 2 * 1600                                  # fSqrt = 20
    tile = np.arange(tot) // 1600
    assert np.all((xy[:, 0] >= 20 * (tile // 2)) & (xy[:, 0] <= 20 * (tile // 2 + 1)))        # the tile holds points at exactly 1.0
    assert np.all((xy[:, 1] >= 20 * (tile % 2)) & (xy[:, 1] <= 20 * (tile % 2 + 1)))
    assert np.all(np.hypot(o8[:, 2::2], o8[:, 3::2]) <= 1.0 + 1e-9) and np.array_equal(o8[:128, 2:], o8[128:256, 2:])
    assert np.allclose(o8[:, 0], xy[:, 0] / 50) and np.allclose(o8[:, 1], (xy[:, 1] + 10) / 50)
    rd = rt.RtRenderDesc(w, h, float(spp), 0, 0)
    a, _, n = osc.render(w, h, spp, rd, rt.RtStats(), threads=2)
    assert n == tot
    fbs = rt.FrontScene("S0", "fastblue", optimize=False)
    b, _, _ = oracle.OracleScene(fbs.desc).render(w, h, spp, rd, rt.RtStats(), threads=2)
    assert np.abs(a - b).mean() < 0.02 and np.abs(a).mean() > 0.001


def test_adaptive_tile_restatement(rt, oracle):
    """AdaptiveBlueSamplingTile restated: the base pass is the FastBlue pass (same return value), refinement happens
    only where there is contrast (some blocks, not all), costs 1600 primary rays per block, and moves the image towards
    a higher-rate FastBlue render at the refined places without touching flat regions."""
    w, h, spp = 60, 40, 4
    rd = rt.RtRenderDesc(w, h, float(spp), 0, 0)
    ads = rt.FrontScene("S0", "adaptive", optimize=False)
    fbs = rt.FrontScene("S0", "fastblue", optimize=False)
    sta, stf = rt.RtStats(), rt.RtStats()
    a, _, na = oracle.OracleScene(ads.desc).render(w, h, spp, rd, sta, threads=2)
    b, _, nb = oracle.OracleScene(fbs.desc).render(w, h, spp, rd, stf, threads=2)
    assert na == nb and stf.refined_blocks == 0
    nblk = (120 // 20) * (80 // 20)
    assert 0 < sta.refined_blocks < nblk and sta.rays_primary == stf.rays_primary + 1600 * sta.refined_blocks
    changed = np.abs(a - b).max(axis=0) > 0
    assert 0.02 < changed.mean() < 0.98 and np.abs(a - b).max() < 0.5


def test_solid_colour_textures_restatement(rt, oracle):
    """The solid colour textures restated (TextureMaps.cpp:518-683, policy Q13): a textured sphere renders, differs from
    the untextured one, and SeaScape (whose loop discards what it computes, :497-513) changes nothing."""
    from conftest import SCENE_COLOUR
    rd = rt.RtRenderDesc(48, 32, 1.0, 0, 0)
    def img(text):
        fs = rt.FrontScene(text, "test", optimize=False)
        return oracle.OracleScene(fs.desc).render(48, 32, 1, rd, rt.RtStats(), threads=2)[0]
    a = img(SCENE_COLOUR)
    assert np.isfinite(a).all() and a.mean() > 1e-3
    b = img(SCENE_COLOUR.replace(" seascape ", " none "))
    assert np.array_equal(a, b)
    plain = SCENE_COLOUR
    for name in ("voronoi", "agate", "marble", "clouds", "wood", "spots", "veins"):
        plain = plain.replace(" %s none " % name, " none none ").replace(" %s coloredglass " % name, " none coloredglass ")
    c = img(plain)
    assert (np.abs(a - c).max(axis=0) > 1e-4).mean() > 0.3


def _mesh_arrays(fs):
    d = fs.desc.contents
    for i in range(d.n_nodes):
        nd = d.nodes[i]
        if nd.type == 15:       # RT_NODE_MESH
            arr = np.ctypeslib.as_array(d.data, shape=(nd.data + nd.ndata,))[nd.data:nd.data + nd.ndata].copy()
            return arr[:3 * nd.i0].reshape(-1, 3), arr[3 * nd.i0:].reshape(-1, 3).astype(int)
    raise AssertionError("no mesh node")


@pytest.mark.parametrize("body", ["(meshicosahedron)", "(meshsphere 16 9)", "(meshsphere 3 2)", "(meshtorus 24 12 0.35)"])
def test_mesh_generators_make_closed_outward_meshes(rt, body):
    """Policy Q14 asks for closed meshes wound counter-clockwise seen from outside: every directed edge once, its reverse
    once, positive signed volume."""
    fs = rt.FrontScene("(scene (model %s))" % body, "test", optimize=True)
    v, f = _mesh_arrays(fs)
    edges = {}
    for a, b, c in f:
        for e in ((a, b), (b, c), (c, a)):
            edges[e] = edges.get(e, 0) + 1
    assert all(n == 1 for n in edges.values()) and all((y, x) in edges for (x, y) in edges)
    assert sum(np.dot(v[a], np.cross(v[b], v[c])) for a, b, c in f) > 0


def test_obj_grammar_of_LoadObjFile(rt, tmp_path):
    """LoadObjFile's grammar (ObjectFormat.cpp:46-66): "v x y z" and "f a/b c/d e/f" (texture indices ignored), 1-based;
    plain "f a b c" too; other lines skipped; a face naming a vertex that does not exist is an error."""
    from conftest import MESH_CUBE
    want_v, want_f = _mesh_arrays(rt.FrontScene("(scene (model %s))" % MESH_CUBE, "test"))
    lines = ["# a cube", "o cube"] + ["v %r %r %r" % tuple(p) for p in want_v.tolist()] + ["vt 0 0"]
    for k, (a, b, c) in enumerate(want_f):
        lines.append("f %d/1 %d/1 %d/1" % (a + 1, b + 1, c + 1) if k % 2 else "f %d %d %d" % (a + 1, b + 1, c + 1))
    path = tmp_path / "cube.obj"
    path.write_text("\n".join(lines) + "\n")
    v, f = _mesh_arrays(rt.FrontScene("(scene (model (objfile %s)))" % path, "test"))
    assert np.array_equal(v, want_v) and np.array_equal(f, want_f)
    L = rt.frontend_lib()
    out = np.zeros(6)
    assert L.rtf_extent(("(translate 0 3 0 (objfile %s))" % path).encode(), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert (out == [-1, 2, -1, 1, 4, 1]).all()
    path.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n")
    with pytest.raises(Exception):
        rt.FrontScene("(scene (model (objfile %s)))" % path, "test")


def test_mesh_solid_restatement_against_the_same_solid_made_of_planes(rt, oracle):
    """Policy Q14 pinned on shapes the reference does have: a cube of twelve triangles must give the hit lists of Cube
    (ConvexPolyhedron, Solids.cpp:660-760) from outside, from inside and inside a Difference -- same counts, materials and
    surfaces, t to rounding, normals parallel -- and a finely tessellated sphere must approach Sphere."""
    from conftest import MESH_CUBE
    rng = np.random.default_rng(1)
    n = 4000
    dr = np.concatenate([rng.uniform(-0.4, 0.4, (n, 2)), -np.ones((n, 1))], axis=1)
    outside = np.tile(np.array([0.3, 0.2, 6.0]), (n, 1)) + rng.normal(0, 0.05, (n, 3))
    inside = rng.normal(0, 0.2, (n, 3))
    wrap = "(material 0.9 auto g_sRed plastic none none (rotate 0.3 0.4 0.1 (difference %s (scale 0.5 0.5 1.2 (cylinder)))))"
    for org in (outside, inside):
        fa = rt.FrontScene("(scene (model %s))" % (wrap % MESH_CUBE), "test")
        fb = rt.FrontScene("(scene (model %s))" % (wrap % "(cube)"), "test")
        a = oracle.OracleScene(fa.desc).intersect(org, dr, max_hits=8, transform=1)
        b = oracle.OracleScene(fb.desc).intersect(org, dr, max_hits=8, transform=1)
        assert (b[0] > 0).mean() > 0.2
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
        assert np.allclose(a[1], b[1], rtol=0, atol=1e-12)
        na = a[2] / np.maximum(1e-300, np.linalg.norm(a[2], axis=2, keepdims=True))
        nb = b[2] / np.maximum(1e-300, np.linalg.norm(b[2], axis=2, keepdims=True))
        assert np.allclose(na, nb, atol=1e-12)
    fa = rt.FrontScene("(scene (model (meshsphere 256 128)))", "test")
    fb = rt.FrontScene("(scene (model (sphere)))", "test")
    a = oracle.OracleScene(fa.desc).intersect(outside[:400], dr[:400] * [0.4, 0.4, 1.0], max_hits=4)
    b = oracle.OracleScene(fb.desc).intersect(outside[:400], dr[:400] * [0.4, 0.4, 1.0], max_hits=4)
    both = (a[0] == 2) & (b[0] == 2)
    assert both.sum() > 100 and (a[0] != b[0]).mean() < 0.02          # silhouette rays only
    assert np.abs(a[1][both] - b[1][both]).max() < 5e-3


def test_laboratory_scene_script(rt, oracle):
    """S4 = one frame of Laboratory() (LaboratoryAnimation.cpp:387-442).  Structure of what the front end builds: eleven
    objects + floor + sky dome, six ColoredGlass Pyrex materials, five spectral metals with Bumpy surfaces and a copper
    stand; the condenser grows with t (more helix turns = more nodes); Optimize() merges the affine chains; the oracle
    renders it (finite, lit, several rays per sample through the glass)."""
    fa = rt.FrontScene("S4@3.0", "fastblue", optimize=False)        # keep them alive: the descriptions point into them
    fb = rt.FrontScene("S4@6.0", "fastblue", optimize=False)
    assert fb.desc.contents.n_nodes > fa.desc.contents.n_nodes > 400
    fs = rt.FrontScene("S4", "fastblue", optimize=False)
    d = fs.desc.contents
    mats = [d.nodes[i] for i in range(d.n_nodes) if d.nodes[i].type == 12]
    assert sum(1 for m in mats if m.i2 == 1) == 5 and sum(1 for m in mats if 4 <= m.i0 <= 10) == 6      # ColoredGlass; spectral metals
    assert d.n_lights == 1 and d.has_ambient and d.camera.type == 1 and d.camera.org[2] == 12.0
    fo = rt.FrontScene("S4", "fastblue", optimize=True)
    assert fo.desc.contents.n_nodes < d.n_nodes
    st = rt.RtStats()
    img, _, n = oracle.OracleScene(fs.desc).render(64, 36, 1, rt.RtRenderDesc(64, 36, 1.0, 0, 0), st, threads=4)
    assert np.isfinite(img).all() and img.mean() > 0.05 and st.rays() > 5 * n


def test_host_spectrum_builders_against_their_formulas(rt):
    """a16: the front end's BlackBody (RadiometryColor.cpp:985-1012) and RGBtoSpectrum (:947-969), evaluated here from the
    formulas: Planck's law at the ten band wavelengths, scaled as the reference scales it, and the three-way decomposition of
    an RGB triple into white plus two primaries (spectra from the extracted tables)."""
    import ctypes as C
    L = rt.frontend_lib()
    fs = rt.FrontScene("(scene (model (material 0.9 auto (blackbody 6500) plastic none none (sphere))) (camera perspective 0 0 6 40))", "", optimize=False)
    d = fs.desc.contents
    mats = [i for i in range(d.n_nodes) if d.nodes[i].type == 12]
    col = np.array([d.data[d.nodes[mats[0]].data + 2 + k] for k in range(10)])
    h, c, kB = 6.62606876e-34, 2.99792458e+8, 1.3806503e-23
    lam = 1e-9 * (380 + 40 * np.arange(10.0))
    planck = 2 * np.pi * h * c * c / (lam ** 5 * (np.exp(h * c / kB / lam / 6500.0) - 1.0))
    want = (planck / 30932027310080.0).astype(np.float32) / np.float32(12.0)
    want = (want.astype(np.float64) * 1.98).astype(np.float32)
    assert np.allclose(col, want, rtol=2e-6), (col, want)
    # RGBtoSpectrum: red smallest -> white * r + green * (g - r) + blue * (b - r)
    fs2 = rt.FrontScene("(scene (model (material 0.9 auto (rgb 0.2 0.5 0.9) plastic none none (sphere))) (camera perspective 0 0 6 40))", "", optimize=False)
    d2 = fs2.desc.contents
    m2 = [i for i in range(d2.n_nodes) if d2.nodes[i].type == 12][0]
    got = np.array([d2.data[d2.nodes[m2].data + 2 + k] for k in range(10)])
    def table(name):
        f = rt.FrontScene("(scene (model (material 0.9 auto %s plastic none none (sphere))) (camera perspective 0 0 6 40))" % name, "", optimize=False)
        dd = f.desc.contents
        mm = [i for i in range(dd.n_nodes) if dd.nodes[i].type == 12][0]
        return np.array([dd.data[dd.nodes[mm].data + 2 + k] for k in range(10)], dtype=np.float32)
    W, G, B = table("g_sWhite"), table("g_sGreen"), table("g_sBlue")
    r, g_, b = np.float32(0.2), np.float32(0.5), np.float32(0.9)
    f32 = lambda spec, x: (spec.astype(np.float64) * float(x)).astype(np.float32)
    want2 = f32(W, r) + f32(G, g_ - r) + f32(B, b - r)
    assert np.allclose(got, want2, rtol=1e-6, atol=1e-7), (got, want2)
