"""rt_render_device_static (include/rt_amd.h): the render entry point that allocates nothing, against rt_render_device.

Called through the C ABI with device pointers.  Bar: the image and the ray counts of rt_render_device, bit for bit, on
two calls in a row, the second with a smaller input (the static scratch array still holds the first call's bytes); the
device's free memory does not move across a call; a render beyond the array's size is rejected with a clear error.
"""
import numpy as np
import pytest

from conftest import SCENE_GLASS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(rt):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return rt


def _both(g, torch, w, h, spp, r0=0, r1=0):
    a = torch.zeros((3, h, w), dtype=torch.float32, device="cuda")
    b = torch.full((3, h, w), 7.0, dtype=torch.float32, device="cuda")
    sa = g.render_device(w, h, spp, a.data_ptr(), 0, r0, r1)
    sb = g.render_device_static(w, h, spp, b.data_ptr(), 0, r0, r1)
    torch.cuda.synchronize()
    lo, hi = (r0, r1) if r1 > 0 else (0, h)
    return a[:, lo:hi], b[:, lo:hi], sa, sb


# S4 (the Laboratory: deep CSG, glass with propagators, bumpy spectral metals) and the glass scene run kernel sets on this path that
# rt_render_device does not use for them (no hit-record pool, no ray sorting, no direct FULL launches): same frames all the same
@pytest.mark.parametrize("scene,sampler", [("S1", "fastblue"), ("S0", "fastblue"), ("S1", "test"), ("S1", "jitter"), ("S0", "adaptive"), ("S1", "bluenoise"),
                                           ("S4", "fastblue"), (SCENE_GLASS, "fastblue")],
                         ids=["S1-fastblue", "S0-fastblue", "S1-test", "S1-jitter", "S0-adaptive", "S1-bluenoise", "S4-fastblue", "glass-fastblue"])
def test_static_render_equals_render_device_twice_in_a_row(gpu, scene, sampler):
    import torch
    fs = gpu.FrontScene(scene, sampler, optimize=True)
    g = gpu.GpuScene(fs.desc)
    for (w, h, spp, r0, r1) in ((160, 96, 4, 0, 0), (64, 40, 4, 0, 0), (160, 96, 4, 30, 61)):      # large, then smaller, then a row band
        a, b, sa, sb = _both(g, torch, w, h, spp, r0, r1)
        assert torch.equal(a, b), (w, h, spp)
        assert (sa.samples, sa.rays_primary, sa.rays_shadow, sa.rays_secondary, sa.refined_blocks) == (sb.samples, sb.rays_primary, sb.rays_shadow, sb.rays_secondary, sb.refined_blocks)
        assert float(a.abs().sum()) > 0.0


def test_static_render_leaves_free_memory_alone_and_rejects_what_does_not_fit(gpu):
    import torch
    fs = gpu.FrontScene("S1", "fastblue", optimize=True)
    warm = gpu.GpuScene(fs.desc)
    out = torch.zeros((3, 128, 128), dtype=torch.float32, device="cuda")
    warm.render_device_static(128, 128, 4, out.data_ptr(), 0)       # the runtime sets up the kernels' private segments on first launch
    g = gpu.GpuScene(fs.desc)                                       # a scene that owns no render buffers yet
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    st = g.render_device_static(128, 128, 4, out.data_ptr(), 0)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert st.samples > 0 and free1 == free0, (free0, free1)
    assert gpu.rt_lib().rt_static_scratch_size() == 128 << 20
    with pytest.raises(gpu.RtError, match="scratch"):
        g.render_device_static(3840, 2160, 64, out.data_ptr(), 0)
    st2 = g.render_device_static(128, 128, 4, out.data_ptr(), 0)    # still usable after the rejection
    assert st2.rays() == st.rays()
