"""Row-band sharding of one frame over ranks: one process per GPU, each rank renders a
contiguous band of output rows with its own halo of samples, rank 0 gathers the bands.  No collective on the
data path other than that one gather per frame."""
import ctypes as C


def row_bands(height, world):
    """Band boundaries: rank r renders rows [b[r], b[r+1])."""
    return [(height * r) // world for r in range(world + 1)]


def balanced_bands(row_cost, world):
    """Band boundaries with (nearly) equal summed cost: row_cost[j] > 0 is a relative cost estimate of output row j.
    Deterministic in its inputs, so every rank derives the same bands without talking to the others."""
    h = len(row_cost)
    if world <= 1 or h == 0:
        return [0, h]
    total = float(sum(row_cost))
    bands, acc, j = [0], 0.0, 0
    for r in range(1, world):
        target = total * r / world
        while j < h and acc + row_cost[j] <= target:
            acc += row_cost[j]
            j += 1
        j = max(j, bands[-1])
        bands.append(min(j, h))
    bands.append(h)
    return bands


def row_cost_from_preview(preview, height, hit_weight=6.6):
    """Relative cost of each of `height` output rows from a small preview image [3][h][w] of the same view: every row
    pays for sample generation and reconstruction (1.0), rows whose pixels see geometry also pay for intersection and
    shading (hit_weight x the fraction of non-black pixels; 6.6 is the measured ratio on the MI355X kernels)."""
    import numpy as np
    hit = (np.abs(np.asarray(preview)).sum(axis=0) > 0).mean(axis=1)          # fraction of hit pixels per preview row
    hp = len(hit)
    # a row also pays for its neighbours' samples (filter halo): smooth over +-1 preview row
    sm = np.convolve(np.pad(hit, 1, mode="edge"), np.ones(3) / 3.0, mode="valid")
    idx = np.minimum((np.arange(height) * hp) // max(height, 1), hp - 1)
    return (1.0 + hit_weight * sm[idx]).tolist()


def rebalance(row_cost, bands, times, damping=1.0):
    """Measured feedback for the cost model: rank r needed times[r] for rows [bands[r], bands[r+1]) whose modelled cost was
    the sum of row_cost over them; every row of that band is rescaled so that the model reproduces the measurement (rows
    keep their relative cost within a band), and the bands are cut again.  A rank with an empty band or without a measurement
    leaves its rows as they are.  Deterministic: every rank computes the same bands from the same gathered times.
    Returns (new_row_cost, new_bands)."""
    world = len(bands) - 1
    cost = list(row_cost)
    tot_c = sum(cost)
    tot_t = sum(t for t in times if t > 0.0)
    if world <= 1 or tot_c <= 0.0 or tot_t <= 0.0:
        return cost, list(bands)
    for r in range(world):
        a, b = bands[r], bands[r + 1]
        c = sum(cost[a:b])
        if b <= a or c <= 0.0 or not times[r] > 0.0:
            continue
        f = (times[r] / tot_t) / (c / tot_c)            # measured share / modelled share
        f = 1.0 + damping * (f - 1.0)
        f = min(4.0, max(0.25, f))
        for j in range(a, b):
            cost[j] *= f
    return cost, balanced_bands(cost, world)


def plan_rows(rt_lib, desc_cls, sampler, width, height, spp, row0, row1):
    """rt_plan_rows (include/rt_amd.h): supersample rows and tile rows a band needs.  Host only."""
    out = (C.c_int * 8)()
    rd = desc_cls(width, height, float(spp), row0, row1)
    rt_lib.rt_plan_rows.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    rc = rt_lib.rt_plan_rows(int(sampler), C.byref(rd), out)
    if rc:
        raise RuntimeError("rt_plan_rows failed: %d" % rc)
    keys = ("src_y0", "src_y1", "tile_y0", "tiles_y", "nW", "nH", "tiles_x", "tiles_y_total")
    return dict(zip(keys, list(out)))


class PendingGather:
    """A band gather in flight (gather_bands_async).  finish() waits for it and, on rank 0, puts the bands in place."""

    def __init__(self, planes, bands, rank, world, work, out, slab=None):
        self.planes, self.bands, self.rank, self.world, self.work, self.out = planes, bands, rank, world, work, out
        self.slab = slab                # kept alive until the collective has read it

    def finish(self):
        if self.work is not None:
            self.work.wait()            # RCCL: the current stream waits for the collective; gloo: the host does
            self.work = None
        self.slab = None
        if self.rank == 0 and self.out is not None:
            for r in range(self.world):
                n = self.bands[r + 1] - self.bands[r]
                self.planes[:, self.bands[r]:self.bands[r + 1], :] = self.out[r][:, :n, :].to(self.planes.device)
            self.out = None
        return self.planes


def gather_bands_async(planes, bands, rank, world, dist, force=False):
    """gather_bands without waiting: the collective is queued behind the work already on the current stream (the band's
    render) and runs beside whatever is queued next -- the following frame's render into ANOTHER buffer.  The caller
    calls finish() on the returned object before it reads planes on rank 0 or reuses the buffer.
    force: run the collective at world size 1 too (a group of one: how the RCCL path is exercised on a one-GPU box)."""
    import torch
    if world == 1 and not (force and dist is not None):
        return PendingGather(planes, bands, rank, world, None, None)
    maxrows = max(bands[r + 1] - bands[r] for r in range(world))
    row0, row1 = bands[rank], bands[rank + 1]
    slab = torch.zeros((planes.shape[0], maxrows, planes.shape[2]), dtype=planes.dtype, device=planes.device)
    slab[:, : row1 - row0, :] = planes[:, row0:row1, :]
    if slab.is_cuda and dist.get_backend() == "gloo":      # one-GPU rehearsals: gloo gathers host tensors
        slab = slab.cpu()
    out = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    work = dist.gather(slab, out, dst=0, async_op=True)
    return PendingGather(planes, bands, rank, world, work, out, slab)


def gather_bands(planes, bands, rank, world, dist, force=False):
    """planes: torch tensor [3][H][W] whose rows bands[rank]:bands[rank+1] this rank has rendered.  After the call
    rank 0's planes holds every band.  One torch.distributed.gather (RCCL on GPUs, gloo on CPU) of equal-size slabs."""
    import torch
    if world == 1 and not (force and dist is not None):
        return planes
    maxrows = max(bands[r + 1] - bands[r] for r in range(world))
    row0, row1 = bands[rank], bands[rank + 1]
    slab = torch.zeros((planes.shape[0], maxrows, planes.shape[2]), dtype=planes.dtype, device=planes.device)
    slab[:, : row1 - row0, :] = planes[:, row0:row1, :]
    if slab.is_cuda and dist.get_backend() == "gloo":      # one-GPU rehearsals: gloo gathers host tensors
        slab = slab.cpu()
    out = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
    dist.gather(slab, out, dst=0)
    if rank == 0:
        for r in range(world):
            planes[:, bands[r]:bands[r + 1], :] = out[r][:, : bands[r + 1] - bands[r], :].to(planes.device)
    return planes


# ---- frame-parallel animation ("S4: frame-parallel across ranks first") ----------------------------

def laboratory_times():
    """The frame times of Laboratory()'s loop, `for (t = 3.0; t <= 6.0; t += 3.0/120.0)` (LaboratoryAnimation.cpp:410):
    accumulated in double exactly as the reference accumulates them."""
    out, t = [], 3.0
    while t <= 6.0:
        out.append(t)
        t += 3.0 / 120.0
    return out


def frame_assignment(n_frames, world):
    """Which frames each rank renders: frame i goes to rank i % world (whole frames, no exchange while rendering; the
    finished frames are what rank 0 collects).  With fewer frames than ranks the surplus ranks get none -- such a job is
    better split by rows (row_bands)."""
    return [list(range(r, n_frames, world)) for r in range(world)]


def gather_frames(frames, owner_of, rank, world, dist, force=False):
    """Collect whole frames on rank 0.  frames: {frame index: tensor [3][H][W]} rendered by this rank; owner_of: frame
    index -> rank, for every frame of the batch.  One torch.distributed.gather per round of `world` frames (RCCL on
    GPUs, gloo on CPU): in round k every rank that has a k-th frame sends it.  Returns {index: tensor} on rank 0."""
    import torch
    if world == 1 and not (force and dist is not None):
        return dict(frames)
    per_rank = [[i for i in sorted(owner_of) if owner_of[i] == r] for r in range(world)]
    rounds = max(len(p) for p in per_rank)
    like = next(iter(frames.values())) if frames else None
    shape = [None]
    if rank == 0:
        shape[0] = tuple(like.shape) if like is not None else None
    dist.broadcast_object_list(shape, src=0)
    out_all = {}
    for k in range(rounds):
        mine = per_rank[rank][k] if k < len(per_rank[rank]) else None
        if mine is not None:
            slab = frames[mine]
        else:
            dev = like.device if like is not None else "cpu"
            slab = torch.zeros(shape[0], dtype=torch.float32, device=dev)
        if slab.is_cuda and dist.get_backend() == "gloo":
            slab = slab.cpu()
        out = [torch.empty_like(slab) for _ in range(world)] if rank == 0 else None
        dist.gather(slab.contiguous(), out, dst=0)
        if rank == 0:
            for r in range(world):
                if k < len(per_rank[r]):
                    out_all[per_rank[r][k]] = out[r]
    return out_all
