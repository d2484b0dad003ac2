import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
W, H, spp = 1920, 1080, 4.0
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
for t in (3.0, 3.1, 3.2):
    fs = rt.FrontScene("S4@%g" % t, "fastblue", optimize=True)
    gs = rt.GpuScene(fs.desc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = gs.render_device(W, H, spp, planes.data_ptr(), 0, 0, 0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    gs.close(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("t=%g render %.1f ms, free %.1f ms" % (t, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
