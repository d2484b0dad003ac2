"""Lane utilisation of the scene interpreter's loop (intersect_scene).  Needs the instrumented library:
    RT_EXTRA_DEFS=-DRT_MESH_COUNT python3 -c 'import __graft_entry__ as g; g.build(force=True)'
(rebuild without it afterwards).  usage: diverge_count.py [scene [width height spp]]"""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
L = rt.rt_lib()
if not hasattr(L, 'rt_debug_counts32'):
    raise SystemExit(__doc__)
scene = sys.argv[1] if len(sys.argv) > 1 else "S4"
W, H, spp = (int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (960, 540, 4.0)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
fs = rt.FrontScene(scene, "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
out = (C.c_ulonglong * 32)()
ops = (C.c_ulonglong * 96)()
L.rt_debug_counts32(out, 1)
L.rt_debug_op_times(ops, 1)
st = gs.render_device(W, H, spp, planes.data_ptr(), torch.cuda.current_stream().cuda_stream, 0, 0)
torch.cuda.synchronize()
L.rt_debug_counts32(out, 1)
for name, b in (("FULL", 8), ("LITE", 16)):
    it, nd, act, rays = out[b], out[b + 1], out[b + 2], out[b + 3]
    if out[b + 5]:
        print("%s %s: lanes still in the loop per pass of the loop %.1f of 64 (running or waiting for their class; %d passes)" % (scene, name, out[b + 4] / float(out[b + 5]), out[b + 5]))
    if it:
        print("%s %s: %d rays, %.1f wave iterations per 64 rays, %.1f instructions per ray, active lanes per iteration %.1f of 64, distinct opcodes per iteration %.2f"
              % (scene, name, rays, it * 64.0 / max(1, rays), act / max(1, rays), act / it, nd / it))
L.rt_debug_op_times(ops, 1)
NAMES = "END BOUND REFBOX PUSH_AFFINE POP_AFFINE QUADRIC POLY SLAB TORUS MATERIAL SURFACE UNION INTER DIFF ANGCLIP SKIP_EMPTY RETAG REGION RET CHAIN MESH".split()
tot = float(sum(ops[:32])) or 1.0
print("scheduled interpreter (FULL), per opcode leading a round: share of time, rounds, ticks per round, lanes per round")
for i, nm in enumerate(NAMES):
    if ops[32 + i]:
        print("  %-12s %5.1f %%  %11d rounds  %8.1f ticks  %5.1f lanes" % (nm, 100.0 * ops[i] / tot, ops[32 + i], ops[i] / ops[32 + i], ops[64 + i] / ops[32 + i]))
