import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
W, H, spp = 1920, 1080, 4.0
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
names = ["Venera@0", "Venera@0.5", "Venera@1", "VeneraVenus@0.5", "VeneraAtmosphere@0.5", "Megacycles@0", "Megacycles@0.5", "Megacycles@1", "Unicyclist@0.3",
         "ModelK", "ModelV", "Model9", "Fib@200", "Ruby", "Letters", "SetOperations", "HyperTest", "Lorenz@2000", "TestTop", "S4atm", "S4atm@5.5"]
bad = 0
for nm in names:
    try:
        fs = rt.FrontScene(nm, "fastblue", optimize=True)
        gs = rt.GpuScene(fs.desc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = gs.render_device(W, H, spp, planes.data_ptr(), 0, 0, 0)
        torch.cuda.synchronize()
        print("%-22s %8.1f ms  %11d rays  overflow %d  finite %s" % (nm, 1e3 * (time.perf_counter() - t0), st.rays(), st.hit_overflow, bool(torch.isfinite(planes).all())), flush=True)
    except Exception as e:
        bad += 1
        print("%-22s FAILED: %s" % (nm, e), flush=True)
sys.exit(1 if bad else 0)
