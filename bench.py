#!/usr/bin/env python3
"""bench.py -- rays/s of the render hot path on MI355X.

A "step" is one RayTracer::Render of the workload: sample generation, camera rays, CSG intersection,
shading with shadow / secondary rays and reconstruction, with the scene already resident in HBM and the
image left in HBM (rt_render_device).  At N > 1 the image rows are split across ranks (no data-path
collective; the finished row bands are gathered to rank 0 with RCCL inside the timed region -- the gather of frame
k is queued asynchronously and runs beside the render of frame k + 1 into a second buffer; the last one is waited
for before the clock stops).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scene S0|S1|S2|S4|S4@t|MESH100K|MESH1M|MESH10M|"(scene ...)"] [--width W --height H --spp S]
                  [--frames F] [--no-cpu-baseline]

Default workload = BASELINE.json configs[1]: the as-checked-in TestScene.cpp scene ("S0": 257 spheres,
MakeLightsLorenz, stereographic camera) at 3840x2160, 64 spp FastBlue = 530,841,600 samples.
--frames F (BASELINE configs[4], "frame- and tile-parallel"): a step is a batch of F frames PER RANK of Laboratory()'s
animation (LaboratoryAnimation.cpp:410-436), whole frames dealt round robin to the ranks, finished frames collected on
rank 0 inside the timed region; per-GPU work is fixed as N grows ("scaling": "weak").
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of a traced ray, SURVEY.md section 8d weights, flops per event
W_AFFINE, W_QUADRIC, W_BOX, W_MERGE, W_SHADE, W_SHADE_LIGHT = 33.0, 57.0, 30.0, 6.0, 250.0, 60.0
FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = half of the 157.3 TF FP32 vector peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0               # HBM3E, MI355X_MICROARCH.md
# PMC measurement of the same workload, collected by tools/pmc_traffic.sh in runs of their own (a copy of the profiles/
# file named inside, kept where GPU runs can read it): NOT measured by this run
TRAFFIC_FILE = os.path.join(ROOT, "ray-tracer_amd", "data", "pmc_traffic.json")


def library_fingerprint():
    """sha256 (first 16 hex digits) over the sources librt_amd.so is built from.  tools/pmc_traffic.py stores it with a PMC
    measurement; `traffic` is reported only while it still names the kernels that are running (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ray-tracer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# BASELINE configs[3]: a synthetic mesh of about ten million triangles (a lobed, star-shaped closed surface of
# 2 * 3200 * 1562 = 9,996,800 triangles) on a floor, one area light and one point light: every ray walks the mesh's BVH
MESH_SCENE = """(scene
 (model (union
   (rotate 0.5 0.3 0 (material 0.9 auto g_sOrange plastic none none (surface shiny none (meshblob %d %d 0.08 5))))
   (translate 0 1.6 -1 (scale 8 0.2 8 (material 0.9 auto g_sEqualWhite plastic none none (cube))))))
 (ambient (scale (blackbody 6500) 0.02))
 (arealight 1.0 9 -6.5 12 (scale (blackbody 6500) 8000))
 (pointlight -10 -10 10 (scale (blackbody 6500) 2000))
 (camera perspective 0 0 6 40))"""
NAMED_SCENES = {"MESH10M": MESH_SCENE % (3200, 1563), "MESH1M": MESH_SCENE % (1000, 501), "MESH100K": MESH_SCENE % (320, 157)}


def note(msg):
    """progress on stderr: a long stage (a 10 M-triangle upload, the CPU baseline) must not look like a hang"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): this process -- which has not touched the
    GPU and never will -- starts N fresh children, one rank per GPU (RANK, LOCAL_RANK, WORLD_SIZE, 127.0.0.1 and a free
    port), relays rank 0's JSON line and exits non-zero if any child does.  The reference's own marker for where this
    parallelism belongs: "This is where forking multiple threads can happen" (Sampling.cpp:313-314, RayTracer2020.h:182).
    A child that fails takes the others down with it (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc, out0 = 0, b""
    live = set(range(n))
    while live:
        for r in sorted(live):
            if r == 0:
                try:
                    out0 += procs[0].communicate(timeout=0.2)[0] or b""
                except subprocess.TimeoutExpired:
                    continue
            elif procs[r].poll() is None:
                continue
            live.discard(r)
            if procs[r].returncode != 0:
                rc = rc or procs[r].returncode or 1
                print("[bench] rank %d exited with %d" % (r, procs[r].returncode), file=sys.stderr, flush=True)
                for q in sorted(live):              # exactly the children started above
                    procs[q].kill()
        if live and 0 not in live:
            time.sleep(0.05)
    for line in out0.decode(errors="replace").splitlines():     # the JSON line to stdout; anything a library printed beside it to stderr
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr, flush=True)
    return rc if rc >= 0 else 1


def launcher_selftest():
    """What a child does under RT_BENCH_LAUNCHER_SELFTEST=1 (tests/test_multi_rank.py, no GPU): join the group the launcher
    described, run one collective, rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if os.environ.get("RT_BENCH_LAUNCHER_SELFTEST") == "fail" and rank == world - 1:
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "sum_of_ranks_plus_one": float(t[0]),
                          "local_rank": int(os.environ["LOCAL_RANK"]), "master": os.environ["MASTER_ADDR"]}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="S0")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=float, default=64.0)
    ap.add_argument("--frames", type=int, default=0, help="frame-parallel animation: frames of Laboratory() per rank and step (0 = one frame split by rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opts", default=os.environ.get("RT_OPTS", ""), help="library tunables for A/B runs: name=value,...")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("RT_BENCH_LAUNCHER_SELFTEST"):
        return launcher_selftest()

    import torch
    import __graft_entry__ as ge
    rt = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the render path has no CPU fallback)")
    # rehearsal of the N > 1 path on a one-GPU box (tools, not the driver): every rank on cuda:0, gloo instead of RCCL
    rehearse = os.environ.get("RT_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if rehearse and world > 2:
        # the ranks share ONE device's memory here, and each sizes its queue pool from what is free: smaller passes, so that all fit
        rt.rt_lib().rt_set_option(b"chunk_samples", max(1 << 20, (1 << 26) // world))
    for kv in filter(None, args.opts.split(",")):
        name, val = kv.split("=")
        if rt.rt_lib().rt_set_option(name.encode(), int(val)):
            raise SystemExit("bad --opts entry %r" % kv)
    if args.frames > 0:
        return frames_mode(args, rt, torch, dist, world, rank, rehearse)
    W, H, spp = args.width, args.height, args.spp
    scene_name = args.scene
    args.scene = NAMED_SCENES.get(scene_name, scene_name)
    fs = rt.FrontScene(args.scene, "fastblue", optimize=True)
    note("scene built")
    gs = rt.GpuScene(fs.desc)
    note("scene uploaded")
    # contiguous row bands; every rank traces its own halo, so the data path has no exchange
    sh = rt.sharding
    if world > 1:
        # cost-balanced bands from a tiny 1-spp preview that every rank renders for itself (deterministic, no exchange)
        ps = rt.FrontScene(args.scene, "test", optimize=True)
        ph, pw = max(16, H // 16), max(16, W // 16)
        preview, _ = rt.GpuScene(ps.desc).render(pw, ph, 1)
        cost = sh.row_cost_from_preview(preview, H)
        rows = sh.balanced_bands(cost, world)
        # ... then corrected by measurement, before any warm-up or timed step: every rank renders its band (the second of two
        # renders is timed with events on its own GPU), the times are all-gathered (N floats) and every rank cuts the same new
        # bands (sharding.rebalance).  A renderer of an animation would do the same from the frame before.
        calib = torch.zeros((3, H, W), dtype=torch.float32, device="cuda")
        cstream = torch.cuda.current_stream().cuda_stream
        for _ in range(int(os.environ.get("RT_BENCH_REBALANCE_ROUNDS", "3"))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms = 0.0
            if rows[rank + 1] > rows[rank]:
                gs.render_device(W, H, spp, calib.data_ptr(), cstream, rows[rank], rows[rank + 1])
                e0.record()
                gs.render_device(W, H, spp, calib.data_ptr(), cstream, rows[rank], rows[rank + 1])
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1)
            mine = torch.tensor([ms], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            got = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            cost, rows = sh.rebalance(cost, rows, [float(t[0]) for t in got])
        del calib
        note("bands after measured rebalancing: %s" % rows)
    else:
        rows = sh.row_bands(H, world)
    row0, row1 = rows[rank], rows[rank + 1]
    # two frame buffers: the gather of frame k (RCCL, its own stream) runs beside the render of frame k + 1
    frames = [torch.zeros((3, H, W), dtype=torch.float32, device="cuda") for _ in range(2 if world > 1 else 1)]
    stream = torch.cuda.current_stream().cuda_stream
    state = {"k": 0, "pending": None}

    def step():
        planes = frames[state["k"] % len(frames)]
        state["k"] += 1
        st = gs.render_device(W, H, spp, planes.data_ptr(), stream, row0, row1 if world > 1 else 0)
        g = sh.gather_bands_async(planes, rows, rank, world, dist)
        if state["pending"] is not None:
            state["pending"].finish()       # the previous frame is complete on rank 0 from here on
        state["pending"] = g
        return st

    def drain():
        if state["pending"] is not None:
            state["pending"].finish()
            state["pending"] = None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    note("warmup done")
    # HIP events around the tracing kernels' launches of the timed region, recorded on the render stream and read back
    # after it (rt_probe_work / rt_probe_fast_time): nothing inside the region waits for them
    rt.rt_lib().rt_set_option(b"time_kernels", 1)
    barrier()
    t0 = time.perf_counter()
    rays = samples = rays_fast = rays_interp = 0
    t_trace = 0.0
    for _ in range(args.steps):
        st = step()
        rays += st.rays()
        samples += st.samples
        rays_fast += st.rays_fast
        rays_interp += st.rays_visual0
        t_trace += st.seconds_trace
    drain()                 # the last frame's gather belongs to the timed region
    barrier()
    dt = time.perf_counter() - t0
    note("timed steps done")
    rt.rt_lib().rt_set_option(b"time_kernels", 0)
    tot = torch.tensor([float(rays), float(samples), dt, t_trace], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[2])
    rays_all, samples_all = float(tot[0]), float(tot[1])
    # Row bands overlap: the samples of a band's halo are traced by two ranks.  The job is ONE frame per step, so the rays that
    # count are the frame's, not the sum over ranks: scale by (samples of the frame) / (samples the ranks traced).
    halo_factor = 1.0
    if world > 1 and rank == 0 and samples_all > 0:
        frame_samples = float(gs.samples(W, H, spp, 0, 0)[2]) * args.steps
        halo_factor = samples_all / frame_samples
        rays_all, samples_all = rays_all / halo_factor, frame_samples

    if rank == 0:
        res = {
            "metric": "rays/sec (primary+secondary)", "value": rays_all / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s %dx%d %g spp FastBlue (BASELINE configs[1] when S0 3840x2160 64, configs[3] when MESH10M)" % (scene_name, W, H, spp),
                       "scene": scene_name, "width": W, "height": H, "spp": spp, "samples_per_step": samples_all / args.steps,
                       "rays_per_step": rays_all / args.steps, "parallelism": "rows%d" % world, "bands": rows,
                       "halo_factor": halo_factor,     # samples traced by all ranks / samples of the frame (rays and samples above are the frame's)
                       # how rank 0's primary rays were traced: the rest ended in the generator (they miss every box of the scene)
                       "primary_rays_per_step_by_path": {"k_fast": rays_fast / args.steps, "k_visual": rays_interp / args.steps,
                                                          "ended_in_generator": (st.rays_primary - st.rays_fast - st.rays_visual0)},
                       **({"rehearsal": "all ranks on one GPU, gloo"} if rehearse else {})},
            "sec_per_frame": dt / args.steps,
        }
        res["roofline"] = roofline(rt, gs, fs, W, H, spp, st, args.steps, rays_fast, res["config"]["workload"])
        if scene_name in MESH_WALK_BYTES:
            res["roofline"] = mesh_roofline(rt, gs, scene_name, res["roofline"])
        note("roofline probe done")
        if not args.no_cpu_baseline:
            res["cpu_baseline"], res["cpu_baseline_all_cores"] = cpu_baseline(ge, rt, args.scene, scene_name)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def frames_mode(args, rt, torch, dist, world, rank, rehearse):
    """Frame-parallel animation (BASELINE configs[4] without its Atmosphere stub): every rank renders whole
    frames of Laboratory() -- frame i of a batch goes to rank i % world -- and rank 0 collects them.  Scene construction,
    Optimize() and upload are the reference driver's work outside RayTracer::Render (LaboratoryAnimation.cpp:412-432) and
    happen before the clock starts.  Fixed work per GPU: weak scaling."""
    sh = rt.sharding
    times = sh.laboratory_times()
    F, W, H, spp = args.frames, args.width, args.height, args.spp
    per_step = F * world
    need = (args.warmup + args.steps) * per_step
    assign = sh.frame_assignment(per_step, world)
    scenes = {}
    for step_i in range(args.warmup + args.steps):
        for j in assign[rank]:
            gi = (step_i * per_step + j) % len(times)
            if gi not in scenes:
                fs = rt.FrontScene("S4@%.17g" % times[gi], "fastblue", optimize=True)
                scenes[gi] = (fs, rt.GpuScene(fs.desc))
    note("%d scenes built and uploaded on rank 0 (of %d frames in all)" % (len(scenes), need))
    bufs = [torch.zeros((3, H, W), dtype=torch.float32, device="cuda") for _ in range(F)]
    stream = torch.cuda.current_stream().cuda_stream
    owner = {i: r for r in range(world) for i in assign[r]}

    uses = {}
    for step_i in range(args.warmup + args.steps):
        for j in assign[rank]:
            gi = (step_i * per_step + j) % len(times)
            uses[gi] = uses.get(gi, 0) + 1

    def step(step_i):
        rays = samples = 0
        mine = {}
        done = []
        for n, j in enumerate(assign[rank]):
            gi = (step_i * per_step + j) % len(times)
            tq = time.perf_counter()
            st = scenes[gi][1].render_device(W, H, spp, bufs[n].data_ptr(), stream)
            if os.environ.get("RT_BENCH_FRAME_TIMES"):
                torch.cuda.synchronize(); note("frame %d (t = %.4f): %.1f ms" % (gi, times[gi], 1e3 * (time.perf_counter() - tq)))
            rays += st.rays()
            samples += st.samples
            mine[j] = bufs[n]
            done.append(gi)
        got = sh.gather_frames(mine, owner, rank, world, dist)
        for gi in done:             # a frame that is not rendered again gives its queues (tens of GB) back
            uses[gi] -= 1
            if uses[gi] == 0 and not os.environ.get("RT_BENCH_KEEP_SCENES"):
                torch.cuda.synchronize()
                del scenes[gi]
        return rays, samples, len(got)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    rays = samples = 0
    for k in range(args.steps):
        r, s, _ = step(args.warmup + k)
        rays += r
        samples += s
    barrier()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(rays), float(samples), dt], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(mx[2])
    if rank == 0:
        res = {
            "metric": "rays/sec (primary+secondary)", "value": float(tot[0]) / dt, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Laboratory() animation, %d frames per rank and step, %dx%d %g spp FastBlue (BASELINE configs[4] without its Atmosphere stub)" % (F, W, H, spp),
                       "scene": "S4@t", "width": W, "height": H, "spp": spp, "frames_per_step": per_step, "parallelism": "frames%d" % world,
                       "rays_per_step": float(tot[0]) / args.steps, "samples_per_step": float(tot[1]) / args.steps,
                       **({"rehearsal": "all ranks on one GPU, gloo"} if rehearse else {})},
            "sec_per_frame": dt / (args.steps * per_step) * world,
            "roofline": {"bound": "valu_fp64", "achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                         "note": "frame-parallel mode times whole frames; the kernel roofline is reported by the one-frame mode (--scene S4)"},
        }
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


# SURVEY 8d's unit for the mesh workload: visited BVH nodes x node bytes + tested triangles x triangle bytes per ray, counted by
# the device on the BVH it walks (instrumented build, tools/mesh_count.py, 960x540 4 spp of the same scene: every ray of the
# scene is walked through the mesh): 15.6 nodes of 64 B and 7.8 triangles of 80 B, plus the 48-byte ray the walk reads
MESH_WALK_COUNTS = {"MESH10M": (15.6, 7.8)}        # BVH nodes visited, triangles tested per ray
MESH_WALK_BYTES = {k: n * 64.0 + t * 80.0 + 48.0 for k, (n, t) in MESH_WALK_COUNTS.items()}
MESH_WALK_BYTES_SURVEY = {k: n * 32.0 + t * 48.0 for k, (n, t) in MESH_WALK_COUNTS.items()}     # SURVEY 8d's literal weights: 32-byte nodes, 48-byte triangles


def mesh_roofline(rt, gs, name, other):
    """Mesh workloads: the dominant kernel is k_mesh_cross (every ray of a level walked through the mesh's BVH before the
    tracing kernels run), HBM-bound in SURVEY 8d's terms.  Durations: HIP events around its launches inside the timed region."""
    L = rt.rt_lib()
    L.rt_probe_mesh_time.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    t = np.zeros(3, dtype=np.float64)
    L.rt_probe_mesh_time(gs.h, t.ctypes.data_as(C.POINTER(C.c_double)))
    ms, launches, rays = t
    if launches <= 0 or rays <= 0:
        return dict(other, note="no k_mesh_cross launch was timed (option mesh_pre < 2?): figures of the tracing kernel")
    b, b8d = MESH_WALK_BYTES[name], MESH_WALK_BYTES_SURVEY[name]
    gbs = b * rays / (ms * 1e-3) / 1e9
    gbs8d = b8d * rays / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "k_mesh_cross", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": None, "bytes_per_ray": b, "rays_per_launch": rays / launches, "ms_per_launch": ms / launches,
            "in_kernel_rays_per_s": rays / (ms * 1e-3), "launches_timed": launches,
            "survey_8d_weights": {"bytes_per_ray": b8d, "achieved": gbs8d, "frac": gbs8d / HBM_PEAK_GBS,
                                  "note": "the same event counts priced at SURVEY 8d's literal 32 B per node and 48 B per triangle"},
            "note": "bytes_per_ray = %.1f BVH nodes x 64 B + %.1f triangles x 80 B (the structures this library walks) + 48 B ray; the counts are the "
                    "device's own, from the counting build (tools/mesh_count.py, -DRT_MESH_COUNT) on this scene at 960x540, 4 spp -- not re-measured by this run; "
                    "the walk is bound by memory-level parallelism (dependent node reads), not by bandwidth: DESIGN.md section 6" % MESH_WALK_COUNTS[name],
            "tracing_kernel": {k: other.get(k) for k in ("kernel", "bound", "achieved", "unit", "frac", "rays_per_launch", "ms_per_launch")}}


def roofline(rt, gs, fs, W, H, spp, st, steps, rays_fast_total, workload):
    """The dominant kernel of the timed steps and its roofs.

    Analytic-CSG workloads are FP64 VALU work, so the roof is the FP64 vector peak: "bound": "valu_fp64",
    achieved = SURVEY 8d's per-event flop weights x IDEAL-culling event counts (counted by the device on a sample of the
    workload's camera rays that reach the tracing kernels) x rays per launch / launch duration.  For S0 SURVEY 8d itself
    prices "~ 8 box tests" per ray -- that is the headline count; the formula count (2 log2 K + 2 per leaf reached: 21 on
    S0) is printed beside it.  S0-like workloads (most primary rays can only touch bare chains of the root region): the
    kernel is k_fast, which reads an 8-byte queue entry per ray and writes 12 bytes of RGB; other workloads: k_visual, which
    streams a frame record per ray (56 B read + 4 B flags + 144 B per shaded boundary + 56 B per queued shadow ray, + 16 B per
    auxiliary pair).  Those algorithmic bytes against the HBM roof are the "hbm" view.  Durations are HIP events on the render
    stream recorded inside the timed region and read after it.  traffic = HBM bytes per launch from PMC counters (separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 correction) of the profiles/ run named in traffic_source -- runs of their own, and
    reported only when they were made with this very build of the library (library_fingerprint) on launches of this size."""
    _, _, total = gs.samples(W, H, spp, 0, 0)
    ntile = total // 1600
    picks = sorted(set(int(x) for x in np.linspace(0, ntile - 1, 64)))
    s8 = np.concatenate([gs.samples(W, H, spp, t * 1600, 1600)[0] for t in picks], axis=0)
    m = len(s8)
    org, dr = gs.camera(s8[:, :2], s8[:, 2:])
    work = np.zeros(9, dtype=np.float64)
    fast = np.zeros(2, dtype=np.float64)
    L = rt.rt_lib()
    L.rt_probe_work.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rt_probe_fast_time.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.rt_probe_fast_time(gs.h, fast.ctypes.data_as(C.POINTER(C.c_double)))
    rc = L.rt_probe_work(gs.h, m, org.ctypes.data_as(C.POINTER(C.c_double)), dr.ctypes.data_as(C.POINTER(C.c_double)), work.ctypes.data_as(C.POINTER(C.c_double)))
    empty = {"bound": "valu_fp64", "achieved": None, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
    if rc:
        return dict(empty, note="rt_probe_work failed: %s" % L.rt_last_error().decode())
    leaf, affine, boxes_ideal, merged, boundaries, lights, ms_per_launch, rays_per_launch, m_reach = work
    if m_reach <= 0:
        return dict(empty, note="no sampled ray reached a tracing kernel")
    frac_reach = m_reach / m
    m = m_reach         # per-ray figures are over the rays the tracing kernels process
    def flops(boxes):
        return (leaf * (W_AFFINE + W_QUADRIC) + boxes * W_BOX + merged * W_MERGE + boundaries * W_SHADE + lights * W_SHADE_LIGHT) / m
    # SURVEY 8d prices S0 with "~ 8 box tests" per ray; the formula 2 log2 K + 2 per leaf reached gives 21 there.  The headline
    # fraction uses the survey's own count where it names one (S0) and the formula elsewhere; both are printed.
    per_ray_formula = flops(boxes_ideal)
    per_ray_survey = flops(8.0 * m) if workload.startswith("S0 ") else per_ray_formula
    events = {"leaf_tests": leaf / m, "ideal_box_tests_formula_2log2K_plus_2_per_leaf": boxes_ideal / m, "box_tests_survey_8d": 8.0 if workload.startswith("S0 ") else None,
              "merged_hits": merged / m, "shaded_boundaries": boundaries / m, "light_terms": lights / m}
    d = fs.desc.contents
    area = any(d.lights[i].type == rt.RT_LIGHT_AREA for i in range(d.n_lights))
    fast_ms, fast_launches = fast
    visual_rays = max(1, st.rays_primary + st.rays_secondary)
    if fast_launches > 0 and rays_fast_total > 0.5 * (rays_fast_total + st.rays_visual0 * steps):
        kernel, ms_per_launch, rays_per_launch = "k_fast", fast_ms / fast_launches, rays_fast_total / fast_launches
        # one launch = the pair k_fast_hit + k_fast_shade (option fast_split, the default): 8-byte queue entry in, a 44-byte boundary
        # record written and read back, 12 bytes of RGB out; the single kernel k_fast (fast_split = 0): 8 in, 12 out
        split = rt.get_options().get("fast_split", 1) == 1
        bytes_per_ray = (8.0 + 44.0 + 44.0 + 12.0) if split else (8.0 + 12.0)
        events["kernels_per_launch"] = "k_fast_hit + k_fast_shade (timed as one pair)" if split else "k_fast"
        events["box_tests_done"] = "none in these kernels: the generator tests the <= 16 boxes of the sample tile's list (k_tile_mask), no BVH walk"
    else:
        if ms_per_launch <= 0:
            return dict(empty, note="no tracing kernel launch was timed")
        kernel = "k_visual"
        bytes_per_ray = 56.0 + (16.0 if area else 0.0) + 4.0 + 144.0 * boundaries / m + 56.0 * st.rays_shadow / visual_rays
    sec = ms_per_launch * 1e-3
    gbs = bytes_per_ray * rays_per_launch / sec / 1e9
    ach = per_ray_survey * rays_per_launch / sec / 1e12
    ach_formula = per_ray_formula * rays_per_launch / sec / 1e12
    traffic = source = None
    traffic_note = "no PMC measurement of this workload and kernel is on file"
    try:
        with open(TRAFFIC_FILE) as f:
            tf = json.load(f)
        if workload.startswith(tf["workload"]) and tf["kernel"] == kernel and abs(tf["rays_per_launch"] - rays_per_launch) <= 0.02 * rays_per_launch:
            if tf.get("library_fingerprint") == library_fingerprint():
                traffic, source = tf["hbm_bytes_per_launch"], tf["source"] + " (PMC passes of their own over the same workload with this very library: fingerprint %s)" % tf["library_fingerprint"]
                traffic_note = None
            else:
                traffic_note = "the PMC measurement on file (%s) is of another build of the library: not reported" % tf.get("source")
    except (OSError, ValueError, KeyError):
        pass
    # analytic-CSG workloads are FP64 VALU work, not HBM work: the roof is the FP64 vector peak
    out = {"bound": "valu_fp64", "kernel": kernel, "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TFLOPS,
           "traffic": traffic, "traffic_source": source, "flops_per_ray": per_ray_survey,
           "flops_per_ray_basis": "SURVEY 8d weights x ideal-culling event counts" + (", 8 box tests per ray as SURVEY 8d prices S0" if workload.startswith("S0 ") else ", box tests by the formula 2 log2 K + 2 per leaf reached"),
           "with_formula_box_tests": {"flops_per_ray": per_ray_formula, "achieved": ach_formula, "frac": ach_formula / FP64_VECTOR_PEAK_TFLOPS},
           "rays_per_launch": rays_per_launch, "ms_per_launch": ms_per_launch, "in_kernel_rays_per_s": rays_per_launch / sec,
           "rays_reaching_tracing_kernels": frac_reach,
           "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_ray": bytes_per_ray,
                   "note": "algorithmic bytes of the same kernel against the HBM roof: not what bounds it"},
           "events_per_ray": events}
    if traffic_note:
        out["traffic_note"] = traffic_note
    return out


def cpu_baseline(ge, rt, scene, name):
    """The oracle (CPU port of the reference's algorithm, oracle/ -- "kind": "port": the reference itself needs Win32
    headers and cannot be built here without stand-ins) timed on this host: a bounded sample of the same workload -- the
    same scene and sampler at 512x288, 16 spp (same rays per sample) -- first on 1 thread (the reference is single-threaded),
    then on the box's share of host cores (OpenMP over samples; results do not depend on the thread count).
    For the mesh workloads the oracle tests every triangle per ray (the reference has no hierarchy over triangles
    either: ObjectFormat.cpp holds flat arrays), so the sample is a few hundred rays."""
    ob = ge.load_oracle()
    sampler = "test" if name.startswith("MESH") else "fastblue"     # one ray per pixel: FastBlue would cast a whole tile of 1600
    raw = rt.FrontScene(scene, sampler, optimize=False)
    osc = ob.OracleScene(raw.desc)
    note("oracle scene built")
    w, h, spp = (512, 288, 16) if name == "S0" else (640, 360, 16)     # about 15-25 s of single-thread work
    if name.startswith("MESH"):
        w, h, spp = {"MESH10M": (12, 8, 1), "MESH1M": (40, 24, 1)}.get(name, (128, 72, 1))
    if name.startswith("S4"):
        w, h, spp = 160, 90, 4
    out = []
    cores = min(16, len(os.sched_getaffinity(0)))       # the GPU box gives a one-GPU job 16 host cores
    for threads in (1, cores):
        rd = rt.RtRenderDesc(w, h, float(spp), 0, 0)
        st = rt.RtStats()
        t0 = time.perf_counter()
        osc.render(w, h, spp, rd, st, threads=threads)
        dt = time.perf_counter() - t0
        out.append({"value": st.rays() / dt, "unit": "rays/s", "cores": threads, "kind": "port",
                    "sample": "%s %dx%d %d spp %s, %d rays in %.1f s on %d thread%s (oracle/, the CPU restatement of the reference; the reference itself is single-threaded)"
                              % (name, w, h, spp, "FastBlue" if sampler == "fastblue" else "one ray per pixel", st.rays(), dt, threads, "" if threads == 1 else "s")})
        note("cpu baseline on %d thread(s) done" % threads)
    return out[0], out[1]


if __name__ == "__main__":
    main()
