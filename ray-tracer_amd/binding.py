"""ctypes bindings: struct mirrors of include/rt_scene_desc.h and the entry points of
librt_amd.so (include/rt_amd.h) and librt_frontend.so (frontend/rt_scenes.cpp)."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_RT = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "librt_amd.so")   # RT_AMD_LIB: A/B builds of the same library
LIB_FRONTEND = os.path.join(HERE, "librt_frontend.so")
TABLES = os.path.join(HERE, "data", "tables.bin")
NSPEC = 10
RT_SAMPLER_TEST, RT_SAMPLER_FASTBLUE, RT_SAMPLER_JITTER, RT_SAMPLER_BLUENOISE, RT_SAMPLER_ADAPTIVEBLUE, RT_SAMPLER_DEBUG = 0, 1, 2, 3, 4, 5      # enum RtSamplerType, include/rt_scene_desc.h
BLUE_N = 1600


class RtNode(C.Structure):
    _fields_ = [("type", C.c_int32), ("id", C.c_int32), ("left", C.c_int32), ("right", C.c_int32),
                ("i0", C.c_int32), ("i1", C.c_int32), ("i2", C.c_int32), ("link", C.c_int32),
                ("data", C.c_int32), ("ndata", C.c_int32)]


RT_LIGHT_POINT, RT_LIGHT_AREA = 0, 1      # enum RtLightType, include/rt_scene_desc.h


class RtLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("no_shadow", C.c_int32), ("org", C.c_double * 3),
                ("vA", C.c_double * 3), ("vB", C.c_double * 3), ("flux", C.c_float * NSPEC)]


class RtCamera(C.Structure):
    _fields_ = [("type", C.c_int32), ("pad_", C.c_int32), ("org", C.c_double * 3), ("focal", C.c_double),
                ("vX", C.c_double * 3), ("vY", C.c_double * 3), ("focus_distance", C.c_double)]


class RtSampler(C.Structure):
    _fields_ = [("type", C.c_int32), ("venera", C.c_int32), ("qindex2", C.POINTER(C.c_int32)),
                ("qindex3", C.POINTER(C.c_int32)), ("aux_init", C.c_double * 6), ("prng_x", C.c_uint32), ("prng_c", C.c_uint32)]


class RtImageMap(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("wrap_x", C.c_int32), ("wrap_y", C.c_int32),
                ("pad_", C.c_int32), ("pixels", C.POINTER(C.c_float))]


class RtSceneDesc(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("root", C.c_int32), ("nodes", C.POINTER(RtNode)),
                ("n_data", C.c_int32), ("n_lights", C.c_int32), ("data", C.POINTER(C.c_double)),
                ("lights", C.POINTER(RtLight)), ("has_ambient", C.c_int32), ("ambient_flux", C.c_float * NSPEC),
                ("camera", RtCamera), ("sampler", RtSampler),
                ("n_images", C.c_int32), ("pad_", C.c_int32), ("images", C.POINTER(RtImageMap))]


class RtRenderDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_double), ("row0", C.c_int32), ("row1", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("samples", C.c_int64), ("rays_primary", C.c_int64), ("rays_shadow", C.c_int64),
                ("rays_secondary", C.c_int64), ("hit_overflow", C.c_int64), ("seconds", C.c_double),
                ("seconds_trace", C.c_double), ("seconds_reconstruct", C.c_double), ("refined_blocks", C.c_int64),
                ("rays_fast", C.c_int64), ("rays_visual0", C.c_int64)]

    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


_frontend = None
_rt = None


class RtError(RuntimeError):
    pass


def rt_lib():
    """librt_amd.so -- the HIP library.  Fails loudly when it has not been built."""
    global _rt
    if _rt is None:
        if not os.path.exists(LIB_RT):
            raise RtError("librt_amd.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        L = C.CDLL(LIB_RT, mode=C.RTLD_GLOBAL)
        L.rt_last_error.restype = C.c_char_p
        L.rt_scene_upload.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(C.c_void_p)]
        L.rt_scene_free.argtypes = [C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(RtRenderDesc), C.POINTER(C.c_float), C.POINTER(RtStats)]
        L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RtRenderDesc), C.c_void_p, C.POINTER(RtStats), C.c_void_p]
        L.rt_render_device_static.argtypes = [C.c_void_p, C.POINTER(RtRenderDesc), C.c_void_p, C.POINTER(RtStats), C.c_void_p]
        L.rt_static_scratch_size.restype = C.c_int64
        L.rt_probe_samples.argtypes = [C.c_void_p, C.POINTER(RtRenderDesc), C.c_int64, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.rt_probe_camera.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_probe_intersect.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rt_probe_radiance.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float)]
        L.rt_probe_sample_rgb.argtypes = [C.c_void_p, C.POINTER(RtRenderDesc), C.POINTER(C.c_float), C.c_int64, C.POINTER(RtStats)]
        L.rt_build_quasi_index.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
        L.rt_random_unsigned.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.rt_set_option.argtypes = [C.c_char_p, C.c_int64]
        L.rt_get_option.argtypes = [C.c_char_p]
        L.rt_get_option.restype = C.c_int64
        _rt = L
    return _rt


def frontend_lib():
    global _frontend
    if _frontend is None:
        rt_lib()  # the front end links against the C ABI
        if not os.path.exists(LIB_FRONTEND):
            raise RtError("librt_frontend.so is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_FRONTEND, mode=C.RTLD_GLOBAL)
        L.rtf_last_error.restype = C.c_char_p
        L.rtf_scene_build.restype = C.c_void_p
        L.rtf_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.rtf_scene_desc.restype = C.POINTER(RtSceneDesc)
        L.rtf_scene_desc.argtypes = [C.c_void_p]
        L.rtf_scene_free.argtypes = [C.c_void_p]
        L.rtf_scene_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float), C.POINTER(RtStats)]
        L.rtf_write_video.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_float]
        L.rtf_extent.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
        L.rtf_random_double.argtypes = [C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_double)]
        _frontend = L
    return _frontend


class FrontScene:
    """A scene built by the host front end (named scene S0/S1/S2 or scene text), flattened."""

    def __init__(self, spec, sampler="", optimize=True):
        L = frontend_lib()
        self._L = L
        self.h = L.rtf_scene_build(spec.encode(), sampler.encode(), 1 if optimize else 0)
        if not self.h:
            raise RtError(L.rtf_last_error().decode())
        self.desc = L.rtf_scene_desc(self.h)

    def node_ids(self):
        d = self.desc.contents
        return [d.nodes[i].id for i in range(d.n_nodes)]

    def render(self, width, height, spp):
        planes = np.zeros((3, height, width), dtype=np.float32)
        st = RtStats()
        n = self._L.rtf_scene_render(self.h, width, height, float(spp), _p(planes, C.c_float), C.byref(st))
        if n < 0:
            raise RtError(self._L.rtf_last_error().decode())
        return planes, st

    def close(self):
        if self.h:
            self._L.rtf_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuScene:
    """rt_scene handle on the current HIP device (C ABI)."""

    def __init__(self, desc):
        self.L = rt_lib()
        h = C.c_void_p()
        rc = self.L.rt_scene_upload(desc, C.byref(h))
        if rc:
            raise RtError("rt_scene_upload: %d %s" % (rc, self.L.rt_last_error().decode()))
        self.h = h

    def _chk(self, rc, what):
        if rc:
            raise RtError("%s: %d %s" % (what, rc, self.L.rt_last_error().decode()))

    def render(self, width, height, spp, row0=0, row1=0):
        planes = np.zeros((3, height, width), dtype=np.float32)
        rd = RtRenderDesc(width, height, float(spp), row0, row1)
        st = RtStats()
        self._chk(self.L.rt_render(self.h, C.byref(rd), _p(planes, C.c_float), C.byref(st)), "rt_render")
        return planes, st

    def render_device(self, width, height, spp, dev_ptr, stream=0, row0=0, row1=0):
        rd = RtRenderDesc(width, height, float(spp), row0, row1)
        st = RtStats()
        self._chk(self.L.rt_render_device(self.h, C.byref(rd), C.c_void_p(dev_ptr), C.byref(st), C.c_void_p(stream)), "rt_render_device")
        return st

    def render_device_static(self, width, height, spp, dev_ptr, stream=0, row0=0, row1=0):
        """rt_render_device_static: the same render out of the library's static scratch array, no allocation."""
        rd = RtRenderDesc(width, height, float(spp), row0, row1)
        st = RtStats()
        self._chk(self.L.rt_render_device_static(self.h, C.byref(rd), C.c_void_p(dev_ptr), C.byref(st), C.c_void_p(stream)), "rt_render_device_static")
        return st

    def samples(self, width, height, spp, first, n):
        rd = RtRenderDesc(width, height, float(spp), 0, 0)
        out8 = np.zeros((n, 8)); xy = np.zeros((n, 2)); tot = C.c_int64()
        self._chk(self.L.rt_probe_samples(self.h, C.byref(rd), first, n, _p(out8, C.c_double), _p(xy, C.c_double), C.byref(tot)), "rt_probe_samples")
        return out8, xy, tot.value

    def camera(self, v2, aux=None):
        v2 = np.ascontiguousarray(v2, dtype=np.float64); n = len(v2)
        org = np.zeros((n, 3)); dr = np.zeros((n, 3))
        a = None if aux is None else _p(np.ascontiguousarray(aux, dtype=np.float64), C.c_double)
        self._chk(self.L.rt_probe_camera(self.h, n, _p(v2, C.c_double), a, _p(org, C.c_double), _p(dr, C.c_double)), "rt_probe_camera")
        return org, dr

    def intersect(self, org, dr, aux=None, max_hits=8, transform=0):
        org = np.ascontiguousarray(org, dtype=np.float64); dr = np.ascontiguousarray(dr, dtype=np.float64); n = len(org)
        cnt = np.zeros(n, dtype=np.int32); t = np.zeros((n, max_hits)); nn = np.zeros((n, max_hits, 3))
        leaf = np.zeros((n, max_hits), dtype=np.int32); mat = np.zeros_like(leaf); surf = np.zeros_like(leaf)
        auxa = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
        a = None if auxa is None else _p(auxa, C.c_double)
        self._chk(self.L.rt_probe_intersect(self.h, n, _p(org, C.c_double), _p(dr, C.c_double), a, max_hits, transform,
                                            _p(cnt, C.c_int), _p(t, C.c_double), _p(nn, C.c_double), _p(leaf, C.c_int), _p(mat, C.c_int), _p(surf, C.c_int)), "rt_probe_intersect")
        return cnt, t, nn, leaf, mat, surf

    def radiance(self, org, dr, aux=None):
        org = np.ascontiguousarray(org, dtype=np.float64); dr = np.ascontiguousarray(dr, dtype=np.float64); n = len(org)
        out = np.zeros((n, NSPEC), dtype=np.float32)
        auxa = None if aux is None else np.ascontiguousarray(aux, dtype=np.float64)
        a = None if auxa is None else _p(auxa, C.c_double)
        self._chk(self.L.rt_probe_radiance(self.h, n, _p(org, C.c_double), _p(dr, C.c_double), a, _p(out, C.c_float)), "rt_probe_radiance")
        return out

    def sample_rgb(self, width, height, spp, max_samples):
        rd = RtRenderDesc(width, height, float(spp), 0, 0)
        out = np.zeros((max_samples, 3), dtype=np.float32); st = RtStats()
        self._chk(self.L.rt_probe_sample_rgb(self.h, C.byref(rd), _p(out, C.c_float), max_samples, C.byref(st)), "rt_probe_sample_rgb")
        return out[:st.samples], st

    def close(self):
        if self.h:
            self.L.rt_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


OPTION_NAMES = ("chunk_samples", "bvh", "root_cull", "lds_chains", "fuse_splat", "cell_order", "full_tables", "full_waves", "persistent",
                "lds_tables", "tile_mask", "alt_chain", "tile_need", "tile_cand", "time_kernels", "fast_path", "fast_split", "cull_f32", "tile_aperture", "push_wrappers", "level_sync", "expect_empty", "rgb_sorted", "mesh_pre", "direct_full", "sched", "sort_rays", "pool_cache", "level_copy", "matte_shape")


def get_options():
    """Current values of the library's tunables (rt_get_option)."""
    L = rt_lib()
    return {n: int(L.rt_get_option(n.encode())) for n in OPTION_NAMES}


def direct_launches():
    """FULL kernels launched without a LITE launch before them since the library was loaded (option direct_full; read-only)."""
    return int(rt_lib().rt_get_option(b"direct_launches"))


def set_options(opts):
    L = rt_lib()
    for n, v in opts.items():
        if L.rt_set_option(n.encode(), int(v)):
            raise RtError("rt_set_option(%s, %d): %s" % (n, v, L.rt_last_error().decode()))


def build_quasi_index(prime, n):
    out = np.zeros(n, dtype=np.int32)
    rt_lib().rt_build_quasi_index(_p(out, C.c_int), prime, n)
    return out


def probe_noise(p3):
    """rt_probe_noise: BandNoise, FractalNoise (5 levels, lacunarity 1.9) and Voronoi of the device at points p3[n][3]."""
    L = rt_lib()
    p3 = np.ascontiguousarray(p3, dtype=np.float32); n = len(p3)
    band = np.zeros(n, dtype=np.float32); frac = np.zeros(n, dtype=np.float32); vor = np.zeros(n, dtype=np.float32)
    L.rt_probe_noise.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if L.rt_probe_noise(n, _p(p3, C.c_float), _p(band, C.c_float), _p(frac, C.c_float), _p(vor, C.c_float)):
        raise RtError("rt_probe_noise: %s" % L.rt_last_error().decode())
    return band, frac, vor


def probe_fresnel(cos_ne, n1, n2, metal_luster=0, spectral=True):
    """rt_probe_fresnel: FresnelLaw_Real (F, fNT) for (n1, n2) and, with spectral, Fresnel_Spectral of a metal luster seen from n1."""
    L = rt_lib()
    c = np.ascontiguousarray(cos_ne, dtype=np.float64); n = len(c)
    F = np.zeros(n); T = np.zeros(n); sp = np.zeros((n, NSPEC), dtype=np.float32)
    L.rt_probe_fresnel.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float)]
    if L.rt_probe_fresnel(n, _p(c, C.c_double), float(n1), float(n2), int(metal_luster), _p(F, C.c_double), _p(T, C.c_double), _p(sp, C.c_float) if spectral else None):
        raise RtError("rt_probe_fresnel: %s" % L.rt_last_error().decode())
    return F, T, sp


def clip_planes(half, table, from_origin):
    """ClipPlanes(Extent(half), rgv<table>, n, from_origin) of the front end (Solids.cpp:801-857): 6 doubles, min then max."""
    L = frontend_lib()
    out = np.zeros(6)
    L.rtf_clip_planes.argtypes = [C.c_double, C.c_char_p, C.c_double, C.POINTER(C.c_double)]
    if L.rtf_clip_planes(float(half), table.encode(), float(from_origin), _p(out, C.c_double)):
        raise RtError(L.rtf_last_error().decode())
    return out


def set_asset_path(directory):
    """SetAssetPath of the front end: where the planet image maps (EarthColorMap.bmp, ...) are read from; None = RT_AMD_ASSETS."""
    L = frontend_lib()
    L.rtf_set_asset_path.argtypes = [C.c_char_p]
    L.rtf_set_asset_path.restype = None
    L.rtf_set_asset_path(directory.encode() if directory else None)


class BmpImage(object):
    """Image::ReadBMP of the front end (Image.cpp:95-288): .planes is [channels][H][W] linear floats, row 0 = top; get() reads one
    pixel through Image::Get with the given wrap modes (Image.cpp:50-89)."""

    def __init__(self, path, gamma=-1.0):
        L = frontend_lib()
        L.rtf_image_read_bmp.argtypes = [C.c_char_p, C.c_double, C.POINTER(C.c_int)]
        L.rtf_image_read_bmp.restype = C.c_void_p
        L.rtf_image_planes.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.rtf_image_planes.restype = None
        L.rtf_image_get.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rtf_image_get.restype = C.c_float
        L.rtf_image_free.argtypes = [C.c_void_p]
        L.rtf_image_free.restype = None
        whc = (C.c_int * 3)()
        self._L = L
        self.h = L.rtf_image_read_bmp(path.encode(), float(gamma), whc)
        if not self.h:
            raise RtError(L.rtf_last_error().decode())
        self.width, self.height, self.channels = int(whc[0]), int(whc[1]), int(whc[2])
        self.planes = np.zeros((self.channels, self.height, self.width), dtype=np.float32)
        L.rtf_image_planes(self.h, _p(self.planes, C.c_float))

    def get(self, i, j, k=0, wrap_x=2, wrap_y=2):
        return float(self._L.rtf_image_get(self.h, wrap_x, wrap_y, int(i), int(j), int(k)))

    def __del__(self):
        if getattr(self, "h", None):
            self._L.rtf_image_free(self.h)
            self.h = None


def bmp_encode(planes, gamma=-1.0):
    """Image::WriteBMP on the GPU (rt_bmp_encode): the bytes of the 24-bit BMP file of a [3][H][W] float image."""
    L = rt_lib()
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, h, w = planes.shape
    L.rt_bmp_size.restype = C.c_int64
    L.rt_bmp_size.argtypes = [C.c_int, C.c_int]
    L.rt_bmp_encode.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_double, C.POINTER(C.c_ubyte), C.c_int64]
    n = L.rt_bmp_size(w, h)
    out = np.zeros(n, dtype=np.uint8)
    rc = L.rt_bmp_encode(planes.ctypes.data_as(C.POINTER(C.c_float)), w, h, float(gamma), out.ctypes.data_as(C.POINTER(C.c_ubyte)), n)
    if rc:
        raise RtError("rt_bmp_encode: %s" % L.rt_last_error().decode())
    return out


def y4m_header(width, height):
    """Video::NewVideo's stream header (rt_y4m_header)."""
    L = rt_lib()
    L.rt_y4m_header.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(128)
    n = L.rt_y4m_header(width, height, buf, 128)
    if n < 0:
        raise RtError("rt_y4m_header: %s" % L.rt_last_error().decode())
    return buf.raw[:n]


def y4m_encode_frame(planes, gamma=2.0):
    """Video::WriteFrame on the GPU (rt_y4m_encode_frame): b"FRAME\\n" + Y, Cb, Cr planes of a [3][H][W] float image."""
    L = rt_lib()
    planes = np.ascontiguousarray(planes, dtype=np.float32)
    _, h, w = planes.shape
    L.rt_y4m_frame_size.restype = C.c_int64
    L.rt_y4m_frame_size.argtypes = [C.c_int, C.c_int]
    L.rt_y4m_encode_frame.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float, C.POINTER(C.c_ubyte), C.c_int64]
    n = L.rt_y4m_frame_size(w, h)
    out = np.zeros(n, dtype=np.uint8)
    rc = L.rt_y4m_encode_frame(planes.ctypes.data_as(C.POINTER(C.c_float)), w, h, float(gamma), out.ctypes.data_as(C.POINTER(C.c_ubyte)), n)
    if rc:
        raise RtError("rt_y4m_encode_frame: %s" % L.rt_last_error().decode())
    return out


def write_video(path, frames, gamma=2.0):
    """The front end's Video class (NewVideo, WriteFrame per image, Close) over a list of [3][H][W] float images."""
    L = frontend_lib()
    fr = np.ascontiguousarray(np.stack(frames), dtype=np.float32)
    n, _, h, w = fr.shape
    rc = L.rtf_write_video(path.encode(), _p(fr, C.c_float), n, w, h, float(gamma))
    if rc < 0:
        raise RtError(L.rtf_last_error().decode())
    return rc
