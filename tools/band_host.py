"""Host share of one band render of S0 4K / 64 spp: time until rt_render_device returns against time until the stream is idle."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
W, H, spp = 3840, 2160, 64
fs = rt.FrontScene("S0", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for (a, b) in ((0, 0), (1045, 1235), (0, 390)):
    for k in range(3):
        gs.render_device(W, H, spp, planes.data_ptr(), st, a, b)
    torch.cuda.synchronize()
    ret, tot = [], []
    for k in range(6):
        t0 = time.perf_counter()
        gs.render_device(W, H, spp, planes.data_ptr(), st, a, b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ret.append(t1 - t0); tot.append(t2 - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gs.render_device(W, H, spp, planes.data_ptr(), st, a, b); e1.record(); torch.cuda.synchronize()
    print("rows [%d, %d): call returns after %.3f ms, stream idle after %.3f ms, GPU time by events %.3f ms" % (a, b, 1e3 * min(ret), 1e3 * min(tot), e0.elapsed_time(e1)))
