import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
rt = ge.load_package(); ob = ge.load_oracle()
torch.cuda.set_device(0)
fs = rt.FrontScene("S0", "fastblue", optimize=True); raw = rt.FrontScene("S0", "fastblue", optimize=False)
o = ob.OracleScene(raw.desc)
w, h, spp = 96, 64, 16
ref, _, n = o.render(w, h, spp, rt.RtRenderDesc(w, h, float(spp), 0, 0), rt.RtStats(), threads=8)
for opts in ({}, {"cull_f32": 0}, {"fast_path": 0}, {"fast_split": 0}, {"tile_cand": 0}):
    rt.set_options(opts)
    g = rt.GpuScene(fs.desc)
    img, st = g.render(w, h, spp)
    d = np.abs(img - ref)
    print(opts, "equal", np.array_equal(img, ref), "ndiff", int((d > 0).sum()), "max", float(d.max()), "fast", st.rays_fast, "visual0", st.rays_visual0)
    for k, v in opts.items(): rt.set_options({k: 1 if k != "tile_cand" else 2})
