import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
rt = ge.load_package(); ob = ge.load_oracle()
torch.cuda.set_device(0)
for name in ("S4", "S4@3.0", "S4@6.0"):
    fs = rt.FrontScene(name, "fastblue", optimize=True)
    raw = rt.FrontScene(name, "fastblue", optimize=False)
    try:
        gs = rt.GpuScene(fs.desc)
        img, st = gs.render(192, 108, 4)
    except Exception as e:
        print(name, "GPU failed:", e); continue
    ost = rt.RtStats()
    ref, _, n = ob.OracleScene(raw.desc).render(192, 108, 4, rt.RtRenderDesc(192, 108, 4.0, 0, 0), ost, threads=16)
    d = np.abs(img - ref)
    print(name, "samples", st.samples, n, "rays", st.rays(), ost.rays(), "overflow", st.hit_overflow, "max abs", d.max(), "rel rmse", float(np.sqrt((d**2).mean())/np.sqrt((ref**2).mean())), "frac pixels > 1e-3", float((d.max(axis=0) > 1e-3).mean()))
fs = rt.FrontScene("S4", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, 1080, 1920), dtype=torch.float32, device="cuda")
gs.render_device(1920, 1080, 4.0, planes.data_ptr(), torch.cuda.current_stream().cuda_stream, 0, 0); torch.cuda.synchronize()
t0 = time.perf_counter(); st = gs.render_device(1920, 1080, 4.0, planes.data_ptr(), torch.cuda.current_stream().cuda_stream, 0, 0); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("S4 1920x1080 4 spp (the reference's own frame size): %.1f ms, %d rays, %.1f M rays/s" % (dt * 1e3, st.rays(), st.rays() / dt / 1e6))
