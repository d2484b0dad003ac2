import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
fs = rt.FrontScene("S4", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, 1080, 1920), dtype=torch.float32, device="cuda")
st = gs.render_device(1920, 1080, 4.0, planes.data_ptr(), torch.cuda.current_stream().cuda_stream, 0, 0); torch.cuda.synchronize()
print("rays", st.rays_primary, st.rays_shadow, st.rays_secondary)
