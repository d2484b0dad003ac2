import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
W, H, spp = 3840, 2160, 64
fs = rt.FrontScene("S0", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
gs.render_device(W, H, spp, planes.data_ptr(), 0)
torch.cuda.synchronize()
for (a, b) in ((0, 8), (1000, 1008), (1000, 1064), (1000, 1270), (1058, 1240)):
    ts = []
    for k in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gs.render_device(W, H, spp, planes.data_ptr(), 0, a, b)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("rows [%d, %d): %.2f ms" % (a, b, 1e3 * min(ts)))
