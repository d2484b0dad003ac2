#!/usr/bin/env python3
"""One-GPU rehearsal of the multi-GPU split: render every rank's band of one frame in turn and report the per-band
times for equal-height and cost-balanced bands (the frame time on N GPUs is the slowest band)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

rt = ge.load_package()
W, H, spp, scene = 3840, 2160, 64, (sys.argv[1] if len(sys.argv) > 1 else "S0")
fs = rt.FrontScene(scene, "fastblue", optimize=True)       # keep the front-end scene alive: the description points into it
gs = rt.GpuScene(fs.desc)
planes = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
gs.render_device(W, H, spp, planes.data_ptr(), 0)          # warm-up, allocations
torch.cuda.synchronize()
t0 = time.perf_counter(); gs.render_device(W, H, spp, planes.data_ptr(), 0); torch.cuda.synchronize()
full = time.perf_counter() - t0
ps = rt.FrontScene(scene, "test", optimize=True)
prev, _ = rt.GpuScene(ps.desc).render(W // 16, H // 16, 1)
weights = [float(x) for x in os.environ.get("RT_HIT_WEIGHTS", "").split(",") if x]       # sweep of the cost model's weight
cands = [("equal", None)] + ([("w%g" % w, w) for w in weights] if weights else [("balanced", "default")])
for world in (2, 4, 8):
    for name, wt in cands:
        if wt is None: bands = rt.sharding.row_bands(H, world)
        elif wt == "default": bands = rt.sharding.balanced_bands(rt.sharding.row_cost_from_preview(prev, H), world)
        else: bands = rt.sharding.balanced_bands(rt.sharding.row_cost_from_preview(prev, H, hit_weight=wt), world)
        if True:
            ts = []
            for r in range(world):
                if bands[r] == bands[r + 1]:
                    ts.append(0.0); continue
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gs.render_device(W, H, spp, planes.data_ptr(), 0, bands[r], bands[r + 1])
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print("%s world %d %-8s full %.1f ms  slowest band %.1f ms  -> efficiency %.2f   bands %s" %
                  (scene, world, name, 1e3 * full, 1e3 * max(ts), full / (world * max(ts)), bands))

# measured feedback (sharding.rebalance, what bench.py does before its warm-up at N > 1): bands from the preview model, then
# three rounds of "every rank times its band, the model is rescaled band by band, the bands are cut again"
for world in (2, 4, 8):
    cost = rt.sharding.row_cost_from_preview(prev, H)
    bands = rt.sharding.balanced_bands(cost, world)
    for rnd in range(4):
        ts = []
        for r in range(world):
            if bands[r] == bands[r + 1]:
                ts.append(0.0); continue
            gs.render_device(W, H, spp, planes.data_ptr(), 0, bands[r], bands[r + 1])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gs.render_device(W, H, spp, planes.data_ptr(), 0, bands[r], bands[r + 1])
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("%s world %d feedback round %d  slowest band %.2f ms  -> efficiency %.2f   bands %s" % (scene, world, rnd, 1e3 * max(ts), full / (world * max(ts)), bands))
        cost, bands = rt.sharding.rebalance(cost, bands, ts)
