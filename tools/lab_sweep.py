"""Every n-th frame of the Laboratory animation (LaboratoryAnimation.cpp:410: t = 3.0 ... 6.0 in 121 steps) at the reference's
size, once each: does it render (no dynamic limit hit), how long does it take, how many rays.  usage: lab_sweep.py [stride]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H, spp = 1920, 1080, 4.0
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
times = rt.sharding.laboratory_times()
bad = 0
for i in list(range(0, len(times), stride)) + [len(times) - 1]:
    fs = rt.FrontScene("S4@%.17g" % times[i], "fastblue", optimize=True)
    gs = rt.GpuScene(fs.desc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        st = gs.render_device(W, H, spp, planes.data_ptr(), 0, 0, 0)
        torch.cuda.synchronize()
        print("frame %3d  t = %.4f  %7.1f ms  %11d rays  overflow %d" % (i, times[i], 1e3 * (time.perf_counter() - t0), st.rays(), st.hit_overflow), flush=True)
    except Exception as e:
        bad += 1
        print("frame %3d  t = %.4f  FAILED: %s" % (i, times[i], e), flush=True)
sys.exit(1 if bad else 0)
