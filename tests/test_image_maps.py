"""The planet image-map textures (PlanetMapping, TextureMaps.cpp:955-1243) with their Image plumbing: Image::ReadBMP
(Image.cpp:95-288), the wrap modes of Image::Get (Image.cpp:50-89, Image.h:226-236), SampleLerp / SampleLerpRGB / SampleGradient
(Image.cpp:1262-1307, Image.h:187-189).

The reference does not ship its maps (EarthColorMap.bmp, ... are read from the working directory), so every asset here is SYNTHETIC:
written by this file into a temporary directory, same file names, smaller sizes.  Parity is GPU against the CPU oracle on the same
description plus expectations worked out by hand below; "parity unpinned" beyond that (no output of the reference exists for
ModelL4 / ModelL5 / ModelU)."""
import math
import os
import struct

import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------ BMP files made by hand
def write_bmp(path, width, height, bits, rows, palette=None, masks=None, clr_used=0):
    """rows: top row first; per pixel a palette index (bits <= 8), a packed integer (16, 32) or an (r, g, b) triple (24)."""
    line = 4 * ((width * bits + 31) >> 5)
    body = bytearray()
    for row in reversed(rows):
        b = bytearray(line)
        for i, v in enumerate(row):
            if bits == 1:
                b[i >> 3] |= (v & 1) << (7 - (i & 7))
            elif bits == 4:
                b[i >> 1] |= (v & 15) << (4 - 4 * (i & 1))
            elif bits == 8:
                b[i] = v
            elif bits == 16:
                b[2 * i:2 * i + 2] = struct.pack("<H", v)
            elif bits == 24:
                b[3 * i:3 * i + 3] = bytes((v[2], v[1], v[0]))
            else:
                b[4 * i:4 * i + 4] = struct.pack("<I", v)
        body += b
    pal = b"".join(bytes((c[2], c[1], c[0], 0)) for c in (palette or []))
    mk = struct.pack("<III", *masks) if masks else b""
    off = 54 + len(mk) + len(pal)
    head = b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off)
    head += struct.pack("<IIIHHIIIIII", 40, width, height, 1, bits, 3 if masks else 0, 0, 2835, 2835, clr_used, 0)
    with open(path, "wb") as f:
        f.write(head + mk + pal + bytes(body))


def gamma_table(gamma):
    """Image::SetGamma, Image.cpp:497-517 (math.pow is the C library's pow, as the front end's)."""
    t = np.zeros(256, np.float32)
    for n in range(256):
        f = n * (1.0 / 255.0)
        if gamma == -1.0:
            f = f / 12.92 if f <= 0.03928 else float(np.float32(math.pow((f + 0.055) / 1.055, 2.4)))
        else:
            f = math.pow(f, gamma)
        t[n] = np.float32(f)
    return t


def test_read_bmp_every_bit_depth_by_hand(rt, tmp_path):
    rng = np.random.RandomState(5)
    w, h = 13, 7                                    # 13: every depth needs padding bytes
    srgb, lin = gamma_table(-1.0), gamma_table(1.0)
    # 24 bits, sRGB
    px = rng.randint(0, 256, (h, w, 3))
    write_bmp(str(tmp_path / "a24.bmp"), w, h, 24, [[tuple(int(c) for c in px[j, i]) for i in range(w)] for j in range(h)])
    im = rt.BmpImage(str(tmp_path / "a24.bmp"))
    assert (im.width, im.height, im.channels) == (w, h, 3)
    assert np.array_equal(im.planes, np.stack([srgb[px[:, :, k]] for k in range(3)]))
    # 8 bits with a grey colour map -> one channel; gamma 1.0 as the bump maps are read
    idx = rng.randint(0, 256, (h, w))
    grey = [(i, i, i) for i in range(256)]
    write_bmp(str(tmp_path / "g8.bmp"), w, h, 8, idx.tolist(), palette=grey)
    im = rt.BmpImage(str(tmp_path / "g8.bmp"), 1.0)
    assert im.channels == 1 and np.array_equal(im.planes[0], lin[idx])
    # 8 bits with a coloured map -> three channels through the map
    pal = [tuple(int(c) for c in rng.randint(0, 256, 3)) for _ in range(256)]
    write_bmp(str(tmp_path / "c8.bmp"), w, h, 8, idx.tolist(), palette=pal)
    im = rt.BmpImage(str(tmp_path / "c8.bmp"))
    pa = np.array(pal)
    assert im.channels == 3 and np.array_equal(im.planes, np.stack([srgb[pa[idx, k]] for k in range(3)]))
    # 4 bits and 1 bit
    idx4 = rng.randint(0, 16, (h, w))
    write_bmp(str(tmp_path / "c4.bmp"), w, h, 4, idx4.tolist(), palette=pal[:16])
    im = rt.BmpImage(str(tmp_path / "c4.bmp"))
    assert np.array_equal(im.planes, np.stack([srgb[pa[idx4, k]] for k in range(3)]))
    idx1 = rng.randint(0, 2, (h, w))
    write_bmp(str(tmp_path / "m1.bmp"), w, h, 1, idx1.tolist(), palette=[(0, 0, 0), (255, 255, 255)])
    im = rt.BmpImage(str(tmp_path / "m1.bmp"))
    assert im.channels == 1 and np.array_equal(im.planes[0], srgb[255 * idx1])
    # 16 bits: 5-5-5 without masks, 5-6-5 with BI_BITFIELDS: "iRed = (255*iRed) / (mask >> shift)"
    v = rng.randint(0, 1 << 15, (h, w))
    write_bmp(str(tmp_path / "h555.bmp"), w, h, 16, v.tolist())
    im = rt.BmpImage(str(tmp_path / "h555.bmp"))
    want = [srgb[(255 * ((v >> s) & 31)) // 31] for s in (10, 5, 0)]
    assert im.channels == 3 and np.array_equal(im.planes, np.stack(want))
    v = rng.randint(0, 1 << 16, (h, w))
    write_bmp(str(tmp_path / "h565.bmp"), w, h, 16, v.tolist(), masks=(0xF800, 0x07E0, 0x001F))
    im = rt.BmpImage(str(tmp_path / "h565.bmp"))
    want = [srgb[(255 * ((v >> 11) & 31)) // 31], srgb[(255 * ((v >> 5) & 63)) // 63], srgb[(255 * (v & 31)) // 31]]
    assert np.array_equal(im.planes, np.stack(want))
    # 32 bits, 8-8-8 (the top byte is ignored)
    v = rng.randint(0, 1 << 31, (h, w)).astype(np.int64) * 2 + rng.randint(0, 2, (h, w))
    write_bmp(str(tmp_path / "x32.bmp"), w, h, 32, v.tolist())
    im = rt.BmpImage(str(tmp_path / "x32.bmp"))
    assert np.array_equal(im.planes, np.stack([srgb[(v >> s) & 255] for s in (16, 8, 0)]))


def test_read_bmp_errors_like_the_reference(rt, tmp_path):
    with pytest.raises(rt.RtError):
        rt.BmpImage(str(tmp_path / "absent.bmp"))
    (tmp_path / "junk.bmp").write_bytes(b"PNG" + bytes(100))
    with pytest.raises(rt.RtError):
        rt.BmpImage(str(tmp_path / "junk.bmp"))
    write_bmp(str(tmp_path / "ok.bmp"), 8, 4, 24, [[(1, 2, 3)] * 8] * 4)
    data = (tmp_path / "ok.bmp").read_bytes()
    (tmp_path / "short.bmp").write_bytes(data[:-20])                # a truncated raster
    with pytest.raises(rt.RtError):
        rt.BmpImage(str(tmp_path / "short.bmp"))
    rle = bytearray(data); rle[30] = 1                              # BI_RLE8: "only RGB and BITFIELD compression supported"
    (tmp_path / "rle.bmp").write_bytes(bytes(rle))
    with pytest.raises(rt.RtError):
        rt.BmpImage(str(tmp_path / "rle.bmp"))


def wrap_by_hand(i, n, mode):
    """ML_Wrap, Image.cpp:59-83; Python's % is already the non-negative modulus ML_Modulus builds."""
    if mode == 1:
        return i % n
    if mode == 2:
        i = i % (2 * n)
        return 2 * n - i - 1 if i >= n else i
    return min(max(i, 0), n - 1)


def test_image_get_wrap_modes_by_hand(rt, tmp_path):
    w, h = 5, 3
    idx = np.arange(w * h).reshape(h, w)
    write_bmp(str(tmp_path / "w.bmp"), w, h, 8, idx.tolist(), palette=[(i, i, i) for i in range(256)])
    im = rt.BmpImage(str(tmp_path / "w.bmp"), 1.0)
    lin = gamma_table(1.0)
    for wx in (0, 1, 2):
        for wy in (0, 1, 2):
            for i in (-23, -11, -6, -5, -1, 0, 2, 4, 5, 6, 9, 10, 14, 37):
                for j in (-8, -4, -3, -1, 0, 1, 2, 3, 5, 6, 11):
                    if 0 <= i < w and 0 <= j < h:
                        want = idx[j, i]                # in range: no wrapping at all
                    else:
                        want = idx[wrap_by_hand(j, h, wy), wrap_by_hand(i, w, wx)]      # out of range in either: BOTH are wrapped
                    assert im.get(i, j, 0, wx, wy) == lin[want], (wx, wy, i, j)


# ------------------------------------------------------------------------------------------------ synthetic assets
def make_assets(d, w=256, h=128, flat_mars=False):
    """Same file names as LoadSkyMaps .. LoadMoonMaps (TextureMaps.cpp:1086-1234), synthetic contents."""
    os.makedirs(d, exist_ok=True)
    y, x = np.mgrid[0:h, 0:w]
    grey = [(i, i, i) for i in range(256)]

    def colour(fx, fy, ph):
        r = 127.5 + 127.5 * np.sin(2 * np.pi * (fx * x / w + ph))
        g = 127.5 + 127.5 * np.sin(2 * np.pi * (fy * y / h + 0.3 + ph))
        b = 127.5 + 127.5 * np.sin(2 * np.pi * (fx * x / w + fy * y / h + 0.6))
        return np.stack([r, g, b], -1).astype(np.int64)

    def put24(name, a):
        write_bmp(os.path.join(d, name), w, h, 24, [[tuple(int(c) for c in a[j, i]) for i in range(w)] for j in range(h)])

    def put8(name, a):
        write_bmp(os.path.join(d, name), w, h, 8, a.astype(np.int64).tolist(), palette=grey)

    put24("EarthColorMap.bmp", colour(5, 3, 0.0))
    put8("EarthBump.bmp", 127.5 + 127.5 * np.sin(2 * np.pi * 7 * x / w) * np.cos(2 * np.pi * 5 * y / h))
    put8("EarthWater.bmp", 255 * ((np.sin(2 * np.pi * 3 * x / w) + np.cos(2 * np.pi * 2 * y / h)) > 0.4))
    put24("SkyColorMap.bmp", colour(2, 1, 0.25))
    if flat_mars:
        put24("MarsColorMap.bmp", np.tile(np.array([200, 120, 80]), (h, w, 1)))
        put8("MarsBump.bmp", np.full((h, w), 90))
    else:
        put24("MarsColorMap.bmp", colour(3, 4, 0.5))
        put8("MarsBump.bmp", 127.5 + 127.5 * np.cos(2 * np.pi * 9 * x / w + 2 * np.pi * 4 * y / h))
    return d


MARS = "(scene (model (rotate 0.3 0.5 0.1 (material 0.9 auto g_sEqualWhite plastic mars none (surface smooth none (sphere))))) " \
       "(pointlight 5 -6 9 (scale g_sEqualWhite 900)) (ambient (scale g_sEqualWhite 0.02)) (camera stereographic 0 0 5 70))"
TORUS_MOON = "(scene (model (union (translate -1.2 0 0 (material 0.9 auto g_sEqualWhite plastic earth none (surface polished none (rotate 1.1 0.2 0 (torus 0.35))))) " \
             "(translate 1.3 0 0 (scale 0.8 1.1 0.8 (material 0.9 auto g_sEqualWhite metal mars none (surface matte none (sphere))))))) " \
             "(pointlight 5 -6 9 (scale g_sEqualWhite 900)) (camera stereographic 0 0 5 70))"


def test_planet_scenes_need_an_asset_directory(rt, tmp_path, monkeypatch):
    monkeypatch.delenv("RT_AMD_ASSETS", raising=False)
    rt.set_asset_path(None)
    for name in ("ModelL4", "ModelL5", "ModelU"):
        with pytest.raises(rt.RtError, match="image-map"):
            rt.FrontScene(name, "", optimize=True)
    rt.set_asset_path(str(tmp_path))                                # a directory without the files
    with pytest.raises(rt.RtError, match="EarthColorMap.bmp"):
        rt.FrontScene("ModelL4", "", optimize=True)
    make_assets(str(tmp_path))
    try:
        fs = rt.FrontScene("ModelL4", "", optimize=True)
        d = fs.desc.contents
        assert d.n_images == 12
        got = {k: (d.images[k].width, d.images[k].height, d.images[k].channels, d.images[k].wrap_x, d.images[k].wrap_y) for k in range(12) if d.images[k].pixels}
        # EarthTexture: colour (3 channels), bump and water (grey), all WRAP_AROUND in x and WRAP_MIRROR in y (TextureMaps.cpp:1114-1122)
        assert got == {1: (256, 128, 3, 1, 2), 2: (256, 128, 1, 1, 2), 3: (256, 128, 1, 1, 2)}
        # ModelU: Earth2Texture (no water mask) and SkyTexture
        fu = rt.FrontScene("ModelU", "", optimize=True)
        du = fu.desc.contents
        assert sorted(k for k in range(12) if du.images[k].pixels) == [0, 1, 2]
        monkeypatch.setenv("RT_AMD_ASSETS", str(tmp_path))         # the environment works as well
        rt.set_asset_path(None)
        rt.FrontScene("ModelL5", "", optimize=True)
    finally:
        rt.set_asset_path(None)


def test_flat_maps_equal_a_plain_material_by_hand(rt, oracle, tmp_path):
    """CPU.  A colour map of ONE colour and a bump map of one height: PlanetMapping then gives every hit the colour
    RGBtoSpectrum(GammaDecode(200, 120, 80), 1), roughness SURF_SHINY and a zero perturbation -- the picture of the same sphere with
    that material and a shiny surface, which the scene text can say directly."""
    make_assets(str(tmp_path), 64, 32, flat_mars=True)
    rt.set_asset_path(str(tmp_path))
    try:
        tex = rt.FrontScene(MARS, "fastblue", optimize=False)
        srgb = gamma_table(-1.0)
        r, g, b = (repr(float(srgb[v])) for v in (200, 120, 80))
        plain = rt.FrontScene(MARS.replace("g_sEqualWhite plastic mars", "(rgb6500 %s %s %s) plastic none" % (r, g, b)).replace("surface smooth", "surface shiny"), "fastblue", optimize=False)
        rd = rt.RtRenderDesc(48, 32, 4.0, 0, 0)
        a, _, n = oracle.OracleScene(tex.desc).render(48, 32, 4, rd, rt.RtStats(), threads=4)
        b_, _, _ = oracle.OracleScene(plain.desc).render(48, 32, 4, rd, rt.RtStats(), threads=4)
        assert float(b_.max()) > 0.05
        assert np.abs(a - b_).max() <= 2e-6 * float(b_.max())          # (the perturbed normal is Normalize(n) + 0: last-bit differences)
        # and with real structure in the maps the picture changes
        make_assets(str(tmp_path), 64, 32, flat_mars=False)
        bumpy = rt.FrontScene(MARS, "fastblue", optimize=False)
        c, _, _ = oracle.OracleScene(bumpy.desc).render(48, 32, 4, rd, rt.RtStats(), threads=4)
        assert np.abs(c - a).max() > 0.01
    finally:
        rt.set_asset_path(None)


def test_planet_mapping_probe_by_hand(rt, oracle, tmp_path):
    """CPU.  One ray down the z axis onto the unit sphere under EarthTexture: the hit is the north pole of the map -- the stereographic
    branch at its centre (0.25, 0.25) * 8192 = texel (2048, 2048), which a 64 x 32 map addresses as (2048 - 0.5) wrapped around in x and
    mirrored in y: texels x = 63 | 0, y = 0 | 0 (2047 mod 64 = 63, 2048 mod 64 = 0; 2047 mod 64 = 63 -> 64 - 63 - 1 = 0 mirrored, 2048 -> 0).
    Water mask zero there -> the colour is RGBtoSpectrum of the average of texels (63, 0) and (0, 0)."""
    d = str(tmp_path)
    os.makedirs(d, exist_ok=True)
    w, h = 64, 32
    rng = np.random.RandomState(11)
    px = rng.randint(0, 256, (h, w, 3))
    write_bmp(os.path.join(d, "EarthColorMap.bmp"), w, h, 24, [[tuple(int(c) for c in px[j, i]) for i in range(w)] for j in range(h)])
    grey = [(i, i, i) for i in range(256)]
    write_bmp(os.path.join(d, "EarthBump.bmp"), w, h, 8, np.full((h, w), 77).tolist(), palette=grey)
    write_bmp(os.path.join(d, "EarthWater.bmp"), w, h, 8, np.zeros((h, w), np.int64).tolist(), palette=grey)
    rt.set_asset_path(d)
    try:
        fs = rt.FrontScene("(scene (model (material 0.9 auto g_sEqualWhite plastic earth none (surface matte none (sphere)))) "
                           "(ambient (scale g_sEqualWhite 1.0)) (camera orthographic 0 0 5 70))", "", optimize=False)
        o = oracle.OracleScene(fs.desc)
        rad = o.radiance(np.array([[0.0, 0.0, 5.0]]), np.array([[0.0, 0.0, -1.0]]))[0]
    finally:
        rt.set_asset_path(None)
    srgb = gamma_table(-1.0)
    # dx = 0.5 exactly (2048 - 0.5 - 2047), dy = 0.5 with both rows the same texel row 0
    a, b = srgb[px[0, 63]], srgb[px[0, 0]]
    rgb = a + ((b - a).astype(np.float64) * 0.5).astype(np.float32)

    def table(name):
        f = rt.FrontScene("(scene (model (material 0.9 auto %s plastic none none (sphere))) (camera perspective 0 0 6 40))" % name, "", optimize=False)
        dd = f.desc.contents
        mm = [i for i in range(dd.n_nodes) if dd.nodes[i].type == 12][0]
        return np.array([dd.data[dd.nodes[mm].data + 2 + k] for k in range(10)], dtype=np.float32)
    W, R, G, B, D = (table(n) for n in ("g_sWhite", "g_sRed", "g_sGreen", "g_sBlue", "g_sD6500"))
    f32 = lambda spec, x: (spec.astype(np.float64) * float(x)).astype(np.float32)
    r_, g_, b_ = rgb
    if r_ < g_ and r_ < b_:
        s = f32(W, r_) + f32(G, g_ - r_) + f32(B, b_ - r_)
    elif g_ < r_ and g_ < b_:
        s = f32(W, g_) + f32(R, r_ - g_) + f32(B, b_ - g_)
    else:
        s = f32(W, b_) + f32(R, r_ - b_) + f32(G, g_ - b_)
    s = s / D
    # ambient light of flux 1 on the hit: R = colour * E, E = flux * pi (Lights.cpp:33-37)
    want = (s.astype(np.float32) * (np.float32(1.0) * np.float64(math.pi)).astype(np.float32)).astype(np.float32)
    assert np.allclose(rad, want, rtol=3e-6), (rad, want)


@pytest.fixture(scope="module")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return rt


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ModelL4", "ModelL4@105", "ModelL5", "ModelU", "ModelU@200", MARS, TORUS_MOON])
def test_planet_scene_matches_oracle(gpu, oracle, tmp_path, name):
    """GPU against the oracle on synthetic maps; the path goes through asin / atan2 / log / tan of two C libraries, so the bar is the
    libm bar of tests/test_conformance_scenes.py."""
    make_assets(str(tmp_path))
    gpu.set_asset_path(str(tmp_path))
    try:
        opt = gpu.FrontScene(name, "fastblue", optimize=True)
        raw = gpu.FrontScene(name, "fastblue", optimize=False)
    finally:
        gpu.set_asset_path(None)
    g = gpu.GpuScene(opt.desc)
    o = oracle.OracleScene(raw.desc)
    w, h, spp = 96, 54, 4
    img, st = g.render(w, h, spp)
    ost = gpu.RtStats()
    ref, _, n = o.render(w, h, spp, gpu.RtRenderDesc(w, h, float(spp), 0, 0), ost, threads=8)
    assert st.samples == n and st.hit_overflow == 0
    assert float(ref.max()) > 0.0 and float(ref.std()) > 1e-3
    assert abs(st.rays() - ost.rays()) <= max(4, 1e-3 * ost.rays())
    a, b = img.astype(np.float64), ref.astype(np.float64)
    assert float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())) <= 1e-5
    assert np.abs(img - ref).max() <= 1e-2 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.gpu
def test_bmp_written_by_the_gpu_reads_back(gpu, tmp_path):
    """ReadBMP is the inverse of WriteBMP (rt_bmp_encode) on values the 8-bit code can hold: decode(encode(T[n])) == T[n]."""
    srgb = gamma_table(-1.0)
    rng = np.random.RandomState(3)
    codes = rng.randint(0, 256, (3, 9, 14))
    planes = srgb[codes]
    (tmp_path / "rt.bmp").write_bytes(bytes(gpu.bmp_encode(planes)))
    im = gpu.BmpImage(str(tmp_path / "rt.bmp"))
    assert (im.width, im.height, im.channels) == (14, 9, 3) and np.array_equal(im.planes, planes)
