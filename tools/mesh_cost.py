#!/usr/bin/env python3
"""Frame time of the bench's mesh scene with the mesh replaced by other solids: separates the fixed cost of the FULL
kernels from the cost of walking the mesh's BVH.  python tools/mesh_cost.py [width height spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
import bench
rt = ge.load_package()
W, H, spp = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 16.0)
torch.cuda.set_device(0)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for name, body in (("sphere", "(sphere)"), ("torus", "(scale 0.8 0.8 0.8 (torus 0.3))"), ("icosahedron mesh", "(meshicosahedron)"), ("blob 1.6K", "(meshblob 40 21 0.08 5)"),
                   ("blob 100K", "(meshblob 320 157 0.08 5)"), ("blob 1M", "(meshblob 1000 501 0.08 5)"), ("blob 10M", "(meshblob 3200 1563 0.08 5)")):
    text = bench.MESH_SCENE.replace("(meshblob %d %d 0.08 5)", body)
    fs = rt.FrontScene(text, "fastblue", optimize=True)
    t0 = time.perf_counter()
    gs = rt.GpuScene(fs.desc)
    t_up = time.perf_counter() - t0
    gs.render_device(W, H, spp, planes.data_ptr(), stream, 0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        st = gs.render_device(W, H, spp, planes.data_ptr(), stream, 0, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print("%-18s upload %6.2f s  frame %8.2f ms  rays %10d  %7.1f M rays/s  overflow %d" % (name, t_up, dt * 1e3, st.rays(), st.rays() / dt / 1e6, st.hit_overflow), flush=True)
