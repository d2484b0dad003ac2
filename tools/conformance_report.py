"""GPU vs oracle over the conformance scenes (tests/test_conformance_scenes.py): one line per scene."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as g
from test_conformance_scenes import EXACT, LIBM
rt = g.load_package(); ob = g.load_oracle()
torch.cuda.set_device(0)
w, h, spp = 96, 54, 4
for n in (sys.argv[1:] or EXACT + LIBM):
    try:
        opt = rt.FrontScene(n, "fastblue", optimize=True); raw = rt.FrontScene(n, "fastblue", optimize=False)
        t0 = time.time(); gs = rt.GpuScene(opt.desc); img, st = gs.render(w, h, spp); t1 = time.time()
        ost = rt.RtStats(); ref, _, cnt = ob.OracleScene(raw.desc).render(w, h, spp, rt.RtRenderDesc(w, h, float(spp), 0, 0), ost, threads=16); t2 = time.time()
        d = np.abs(img.astype(np.float64) - ref)
        rr = float(np.sqrt((d ** 2).mean()) / max(1e-30, np.sqrt((ref.astype(np.float64) ** 2).mean())))
        print("%-16s %s eq=%d rays %d/%d (%d %d %d | %d %d %d) rmse %.3g max %.3g ovf %d gpu %.2fs cpu %.2fs" % (n, "EXACT" if n in EXACT else "libm ", int(np.array_equal(img, ref)), st.rays(), ost.rays(),
              st.rays_primary, st.rays_shadow, st.rays_secondary, ost.rays_primary, ost.rays_shadow, ost.rays_secondary, rr, float(d.max()), st.hit_overflow, t1 - t0, t2 - t1), flush=True)
    except Exception as e:
        print(n, "FAILED", str(e)[:300], flush=True)
