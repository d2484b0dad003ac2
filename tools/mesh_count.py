"""Node visits and triangle tests per mesh ray.  Needs the instrumented library:
    RT_EXTRA_DEFS=-DRT_MESH_COUNT python3 -c 'import __graft_entry__ as g; g.build(force=True)'
(rebuild without it afterwards)."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import __graft_entry__ as ge, bench
rt = ge.load_package()
torch.cuda.set_device(0)
L = rt.rt_lib()
if not hasattr(L, 'rt_debug_mesh_counts'):
    raise SystemExit(__doc__)
W,H,spp = 960,540,4.0
planes = torch.zeros((3,H,W), dtype=torch.float32, device="cuda")
CASES = (("ico","(meshicosahedron)"),("1.6K","(meshblob 40 21 0.08 5)"),("100K","(meshblob 320 157 0.08 5)"),("1M","(meshblob 1000 501 0.08 5)"),("sph100K","(meshsphere 320 157)"),("10M","(meshblob 3200 1563 0.08 5)"))
for name, body in [c for c in CASES if len(sys.argv) < 2 or c[0] in sys.argv[1:]]:
    fs = rt.FrontScene(bench.MESH_SCENE.replace("(meshblob %d %d 0.08 5)", body), "fastblue", optimize=True)
    gs = rt.GpuScene(fs.desc)
    out = (C.c_ulonglong*8)()
    L.rt_debug_mesh_counts(out, 1)
    st = gs.render_device(W,H,spp,planes.data_ptr(),torch.cuda.current_stream().cuda_stream,0,0)
    torch.cuda.synchronize()
    L.rt_debug_mesh_counts(out, 1)
    r = max(1,out[0])
    print("%-8s rays %d (shadow %d)  mesh rays %9d  nodes/ray %6.1f  tris/ray %6.1f  crossings/ray %5.2f  wave node-iters per ray-wave %7.1f  wave leaf-iters per ray-wave %7.1f" % (name, st.rays(), st.rays_shadow, out[0], out[1]/r, out[3]/r, out[6]/r, out[4]*64/r, out[5]*64/r))
