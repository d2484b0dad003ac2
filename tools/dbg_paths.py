import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
scene = sys.argv[1] if len(sys.argv) > 1 else "S1"
w, h, spp = 960, 540, 16
fs = rt.FrontScene(scene, "fastblue", optimize=True)
for opts in ({}, {"push_wrappers": 0}, {"tile_mask": 0}, {"tile_cand": 0}):
    rt.set_options(opts)
    g = rt.GpuScene(fs.desc)
    img, st = g.render(w, h, spp)
    print(opts, "samples", st.samples, "fast", st.rays_fast, "visual0", st.rays_visual0, "primary", st.rays_primary, "sum", float(img.sum()))
    for k in opts: rt.set_options({k: 2 if k == "tile_cand" else 1})
