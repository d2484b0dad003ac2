"""Kernel trace of ONE narrow band render of S0 4K / 64 spp (what a band costs whatever its height).  Run under rocprofv3 --kernel-trace:
the script renders the whole frame once, then the band [1000, 1008) four times."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
W, H, spp = 3840, 2160, 64
fs = rt.FrontScene("S0", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 1008)
for k in range(5):
    gs.render_device(W, H, spp, planes.data_ptr(), 0, a, b)
    torch.cuda.synchronize()
