"""Multi-rank path on CPU (gloo, world size 2 and 3): band arithmetic, halo planning and the one gather per frame.

The GPU never enters here: each rank takes its band from an image rendered by the ORACLE (test infrastructure), so
what is tested is the host logic bench.py and a multi-GPU caller use -- ray-tracer_amd/sharding.py and
rt_plan_rows (include/rt_amd.h)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, height, width, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    rt = ge.load_package()
    ob = ge.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        raw = rt.FrontScene("S0", "test", optimize=False)
        osc = ob.OracleScene(raw.desc)
        rd = rt.RtRenderDesc(width, height, 1.0, 0, 0)
        full, _, _ = osc.render(width, height, 1, rd, rt.RtStats(), threads=1)
        bands = rt.sharding.row_bands(height, world)
        mine = torch.zeros((3, height, width), dtype=torch.float32)
        mine[:, bands[rank]:bands[rank + 1], :] = torch.from_numpy(full[:, bands[rank]:bands[rank + 1], :])
        rt.sharding.gather_bands(mine, bands, rank, world, dist)
        ok = bool(np.array_equal(mine.numpy(), full)) if rank == 0 else True
        # the pipelined form bench.py uses: two frames in flight, finish() of frame k after frame k + 1 was queued
        frames = []
        for k in range(2):
            f = torch.zeros((3, height, width), dtype=torch.float32)
            f[:, bands[rank]:bands[rank + 1], :] = torch.from_numpy(full[:, bands[rank]:bands[rank + 1], :]) * float(k + 1)
            frames.append(f)
        g0 = rt.sharding.gather_bands_async(frames[0], bands, rank, world, dist)
        g1 = rt.sharding.gather_bands_async(frames[1], bands, rank, world, dist)
        g0.finish(); g1.finish()
        if rank == 0:
            ok = ok and bool(np.array_equal(frames[0].numpy(), full)) and bool(np.array_equal(frames[1].numpy(), full * np.float32(2.0)))
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 37), (3, 50)])
def test_band_gather_gloo(world, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world * 7 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, 48, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _frames_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import __graft_entry__ as ge
    rt = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_frames = 5
        assign = rt.sharding.frame_assignment(n_frames, world)
        owner = {i: r for r in range(world) for i in assign[r]}
        # every frame is a constant image of its own index: what arrives where is all that is tested
        mine = {i: torch.full((3, 4, 6), float(i + 1)) for i in assign[rank]}
        got = rt.sharding.gather_frames(mine, owner, rank, world, dist)
        if rank == 0:
            q.put(sorted(got) == list(range(n_frames)) and all(bool((got[i] == float(i + 1)).all()) for i in got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_parallel_gather_gloo(world):
    """Frame-parallel animation (Laboratory()'s loop, LaboratoryAnimation.cpp:410-436): whole frames are dealt round robin
    to the ranks, nothing is exchanged while rendering, rank 0 collects the finished frames."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + world * 11 + (os.getpid() % 200)
    procs = [ctx.Process(target=_frames_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_laboratory_frame_times_and_assignment():
    import __graft_entry__ as ge
    rt = ge.load_package()
    ts = rt.sharding.laboratory_times()
    assert ts[0] == 3.0 and ts[1] == 3.0 + 3.0 / 120.0 and ts[-1] <= 6.0 < ts[-1] + 3.0 / 120.0 and len(ts) in (120, 121)
    for n, w in ((121, 8), (5, 2), (3, 8), (0, 4)):
        a = rt.sharding.frame_assignment(n, w)
        assert len(a) == w and sorted(i for r in a for i in r) == list(range(n))
        assert max(len(r) for r in a) - min(len(r) for r in a) <= 1


def test_row_bands_partition():
    import __graft_entry__ as ge
    rt = ge.load_package()
    for h in (1, 7, 2160, 2161):
        for w in (1, 2, 3, 8):
            b = rt.sharding.row_bands(h, w)
            assert b[0] == 0 and b[-1] == h and all(b[i] <= b[i + 1] for i in range(w))


@pytest.mark.parametrize("width,height,spp,world", [(64, 48, 4, 2), (96, 54, 16, 3), (80, 100, 9, 8)])
def test_halo_covers_every_contributing_sample(rt, oracle, width, height, spp, world):
    """Every sample whose bilinear splat touches a supersample row that the vertical filter reads for a band must
    lie in the tile rows rt_plan_rows assigns to that band: then bands need no exchange."""
    raw = rt.FrontScene("S0", "fastblue", optimize=False)
    osc = oracle.OracleScene(raw.desc)
    _, _, total = osc.samples(width, height, spp, 0, 0)
    _, xy, _ = osc.samples(width, height, spp, 0, total)
    L = rt.rt_lib()
    whole = rt.sharding.plan_rows(L, rt.RtRenderDesc, rt.RT_SAMPLER_FASTBLUE, width, height, spp, 0, 0)
    assert whole["tile_y0"] == 0 and whole["tiles_y"] == whole["tiles_y_total"]
    assert total == whole["tiles_x"] * whole["tiles_y_total"] * 1600
    bands = rt.sharding.row_bands(height, world)
    # the reference's loop order is tile x outer, tile y inner (Sampling.cpp:498-499)
    tile_y = (np.arange(total) // 1600) % whole["tiles_y_total"]
    y = xy[:, 1]
    for r in range(world):
        if bands[r] == bands[r + 1]:
            continue
        p = rt.sharding.plan_rows(L, rt.RtRenderDesc, rt.RT_SAMPLER_FASTBLUE, width, height, spp, bands[r], bands[r + 1])
        lo_row = np.floor(y - 0.5)             # a splat touches rows floor(y - 0.5) and the next (Image.cpp:1317-1330)
        touches = (lo_row + 1 >= p["src_y0"]) & (lo_row <= p["src_y1"] - 1)
        inside = (tile_y >= p["tile_y0"]) & (tile_y < p["tile_y0"] + p["tiles_y"])
        assert not np.any(touches & ~inside), "band %d misses contributing samples" % r


def test_balanced_bands_equalise_cost():
    import __graft_entry__ as ge
    rt = ge.load_package()
    rng = np.random.default_rng(5)
    for h, w in ((2160, 8), (100, 3), (17, 4), (5, 8)):
        cost = (1.0 + 6.0 * (rng.random(h) < 0.3) * rng.random(h)).tolist()
        b = rt.sharding.balanced_bands(cost, w)
        assert len(b) == w + 1 and b[0] == 0 and b[-1] == h and all(b[i] <= b[i + 1] for i in range(w))
        if h >= 20 * w:
            sums = [sum(cost[b[i]:b[i + 1]]) for i in range(w)]
            assert max(sums) <= 1.15 * (sum(cost) / w) + max(cost)
    assert rt.sharding.balanced_bands([1.0] * 10, 1) == [0, 10]
    # a preview with geometry only in the middle pulls the band boundaries towards the middle
    prev = np.zeros((3, 32, 32), dtype=np.float32)
    prev[:, 12:20, :] = 1.0
    cost = rt.sharding.row_cost_from_preview(prev, 320)
    b = rt.sharding.balanced_bands(cost, 4)
    assert b[1] > 80 and b[3] < 240 and (b[2] - b[1]) < 80


def test_rebalance_converges_on_the_true_cost():
    """sharding.rebalance: bands cut from a wrong cost model (flat) converge to equal measured times within a few rounds when
    every rank reports the time of its band under the true cost; deterministic, bands stay a partition."""
    import __graft_entry__ as ge
    rt = ge.load_package()
    rng = np.random.default_rng(11)
    h, world = 2160, 8
    true = 1.0 + 9.0 * np.exp(-((np.arange(h) - 1100.0) / 300.0) ** 2) + 0.2 * rng.random(h)     # a hump in the middle of the picture
    cost = [1.0] * h
    bands = rt.sharding.balanced_bands(cost, world)
    first = None
    for it in range(4):
        times = [float(true[bands[r]:bands[r + 1]].sum()) + 0.3 for r in range(world)]         # + a fixed cost per call
        eff = sum(times) / (world * max(times))
        first = eff if first is None else first
        cost, bands = rt.sharding.rebalance(cost, bands, times)
        assert len(bands) == world + 1 and bands[0] == 0 and bands[-1] == h and all(bands[i] <= bands[i + 1] for i in range(world))
    times = [float(true[bands[r]:bands[r + 1]].sum()) + 0.3 for r in range(world)]
    assert first < 0.6 and sum(times) / (world * max(times)) > 0.93
    # a rank without a measurement (an empty band, a failed timer) leaves the model of its rows alone
    c2, b2 = rt.sharding.rebalance([1.0] * 100, [0, 50, 50, 100], [1.0, 0.0, 1.0])
    assert b2[0] == 0 and b2[-1] == 100 and len(c2) == 100


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus N` with no launcher around it (the driver's own command line) must start N ranks itself:
    the parent never touches the GPU, gives every child RANK / LOCAL_RANK / WORLD_SIZE / 127.0.0.1 and a free port, and
    relays rank 0's one JSON line.  The children run the launcher's self-test instead of a render (no GPU here)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RT_BENCH_LAUNCHER_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["launcher_selftest"] is True and res["n_gpus"] == 2 and res["sum_of_ranks_plus_one"] == 3.0
    assert res["local_rank"] == 0 and res["master"] == "127.0.0.1"
    r3 = _run_bench(["--gpus", "3"], {"RT_BENCH_LAUNCHER_SELFTEST": "1"})
    assert r3.returncode == 0 and json.loads(r3.stdout.strip())["n_gpus"] == 3


def test_bench_launcher_fails_when_a_rank_fails():
    """a child that exits non-zero ends the job: the others are stopped (they would wait in the collective) and the
    launcher's exit code says so"""
    r = _run_bench(["--gpus", "2"], {"RT_BENCH_LAUNCHER_SELFTEST": "fail"}, timeout=120)
    assert r.returncode != 0
    assert "rank 1 exited with 3" in r.stderr


_NCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as ge
rt = ge.load_package()
torch.cuda.set_device(0)
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = %r
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
W, H, spp = 96, 64, 4
fs = rt.FrontScene("S1", "fastblue", optimize=True)
gs = rt.GpuScene(fs.desc)
ref, _ = gs.render(W, H, spp)
stream = torch.cuda.current_stream().cuda_stream
frames = [torch.zeros((3, H, W), dtype=torch.float32, device="cuda") for _ in range(2)]
bands = [0, H]
pend = None
for k in range(3):          # bench.py's pipeline: the gather of frame k beside the render of frame k + 1
    planes = frames[k %% 2]
    gs.render_device(W, H, spp, planes.data_ptr(), stream, 0, 0)
    g = rt.sharding.gather_bands_async(planes, bands, 0, 1, dist, force=True)
    assert g.work is not None and g.out is not None and g.out[0].is_cuda
    if pend is not None:
        pend.finish()
    pend = g
pend.finish()
torch.cuda.synchronize()
for f in frames:
    assert np.array_equal(f.cpu().numpy(), ref)
out = rt.sharding.gather_bands(frames[0].clone(), bands, 0, 1, dist, force=True)
assert np.array_equal(out.cpu().numpy(), ref)
got = rt.sharding.gather_frames({0: frames[0], 1: frames[1] * 2.0}, {0: 0, 1: 0}, 0, 1, dist, force=True)
assert sorted(got) == [0, 1] and got[0].is_cuda
assert np.array_equal(got[0].cpu().numpy(), ref) and np.array_equal(got[1].cpu().numpy(), ref * np.float32(2.0))
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("NCCL-WORLD1-OK")
"""


@pytest.mark.gpu
def test_rccl_gather_at_world_size_one():
    """The RCCL branch of bench.py (backend "nccl", device tensors) on the one GPU there is: a group of one rank goes
    through the same init, gather_bands_async / finish pipeline, gather_frames and reductions as N ranks would, in a
    process of its own (so that RCCL's buffers and streams do not stay behind in the test process)."""
    import subprocess
    port = str(29300 + os.getpid() % 500)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _NCCL_WORLD1 % (ROOT, port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL-WORLD1-OK" in r.stdout, r.stdout + r.stderr
