"""Sweep driver: the reference's user-level `for` loop over sweep points (example/Example1.ipynb lambda sweep,
example/Example3.ipynb (Wx,Wy) sweep) as one batched, optionally multi-GPU job.

Sharding: sweep points are independent, so rank r of R owns a contiguous block of the flattened sweep
(`shard_range`) -- or, when the cost of a point varies along the sweep (geometry sweeps: eigensolver iteration counts and
fp64 re-solves differ per shape), every R-th point (`shard_indices(..., cyclic=True)`); there is no data-path
collective.  The only communication is one all_gather of the requested S-parameters at the end (`gather_sweep`, RCCL over xGMI
on GPUs; payload is a few KB, latency-bound).
"""
import os

import numpy as np
import torch

from .batched import BatchedRCWA

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_indices(n_items, rank, world, cyclic=False):
    """Global indices (numpy int64, ascending) of the sweep points rank `rank` solves: the contiguous block of `shard_range`, or
    rank, rank + world, rank + 2 world, ... (cyclic: neighbouring points -- similar cost -- go to different ranks)."""
    if cyclic:
        return np.arange(int(rank), int(n_items), int(world), dtype=np.int64)
    lo, hi = shard_range(n_items, rank, world)
    return np.arange(lo, hi, dtype=np.int64)


def gather_sweep(local, n_items, group=None, cyclic=False):
    """all_gather of per-point results: `local` is this rank's [m_r, ...] block, in the order of shard_indices(n_items, rank, world,
    cyclic).  Returns the full [n_items, ...] tensor in sweep order on every rank.  Works with unequal block sizes (pads to the largest)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [len(shard_indices(n_items, r, world, cyclic)) for r in range(world)]
    sizes = [(0, c) for c in counts]
    mmax = max(counts)
    pad = torch.zeros((mmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if pad.is_complex():
        buf = torch.view_as_real(pad).contiguous()
    else:
        buf = pad.contiguous()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = []
    for r, (lo, hi) in enumerate(sizes):
        t = out[r][: hi - lo]
        parts.append(torch.view_as_complex(t) if pad.is_complex() else t)
    full = torch.cat(parts, dim=0)
    if not cyclic:
        return full
    # rank-major -> sweep order
    order = np.concatenate([shard_indices(n_items, r, world, True) for r in range(world)])
    inv = torch.as_tensor(np.argsort(order), device=full.device)
    return full.index_select(0, inv)


def asih_eps_table():
    """eps(lambda) of a-Si:H on linspace(400,700,128) nm (values of example/Materials.py `aSiH.apply(l)**2`, stored as data)."""
    z = np.load(os.path.join(_DATA, "asih_eps_400_700_128.npz"))
    return z["lam"], z["eps"]


def rectangle_density(nx, ny, Lx, Ly, Wx, Wy, Cx, Cy, theta=0.0, edge_sharpness=1000.0, dtype=torch.float64, device="cpu"):
    from .geometry import geometry
    g = geometry(Lx=Lx, Ly=Ly, nx=nx, ny=ny, edge_sharpness=edge_sharpness, dtype=dtype, device=torch.device(device))
    return g.rectangle(Wx=Wx, Wy=Wy, Cx=Cx, Cy=Cy, theta=theta)


def _solve_chunk(freq, layers, order, L, eps_in, eps_out, inc_ang, azi_ang, dtype, precision, engine, orders,
                 polarization, direction, port, check_info, eig_route="auto", route_hint=None):
    """layers: list of (thickness, eps[, mu]); thickness scalar or [b]; eps/mu scalar, [b] or [b,nx,ny]."""
    sim = BatchedRCWA(freq, order, L, dtype=dtype, precision=precision, engine=engine, keep_coupling=False, fold_layers=True,
                      eig_route=eig_route, route_hint=route_hint)
    if eps_in is not None:
        sim.add_input_layer(eps=eps_in)
    if eps_out is not None:
        sim.add_output_layer(eps=eps_out)
    sim.set_incident_angle(inc_ang, azi_ang)
    for lay in layers:
        sim.add_layer(*lay)
    sim.solve_global_smatrix()
    return sim.S_parameters([list(o) for o in orders], direction=direction, port=port, polarization=polarization)


# HBM footprint of one sweep point, in units of one n x n complex128 matrix (n = 2 (2 ox + 1)(2 oy + 1)): measured on MI355X with the caching
# allocator -- single patterned layer, order [15,15], 128 points: 75.8 GB allocated / 111.8 GB reserved = 10.0 / 14.8 matrices per point; 4-layer stack
# with the streaming cascade, order [21,21], 64 points: 228 / 247 GB = 16.3 / 17.6.  precision="native" halves the element.
_POINT_MATRICES = {1: 15.0, 2: 18.0}          # layers == 1 / layers >= 2 (what the allocator RESERVES, which is what must fit)
_HEADROOM = 0.10                               # fraction of the device memory a sweep leaves free


def auto_chunk(B, order, n_layers, precision, device, dtype=torch.complex64, streams=1):
    """Largest number of points solved in lock-step that fits the free HBM of `device` with _HEADROOM to spare (a multiple of 8 when it
    is cut: the mixed-precision eigensolver and its iteration groups want batches of at least 8).  Raises with the numbers when not even
    one point fits -- instead of an allocator error in the middle of a solve."""
    if device.type != "cuda":
        return B
    n = 2 * (2 * order[0] + 1) * (2 * order[1] + 1)
    # element size of the COMPUTE dtype: complex128 unless a complex64 problem is solved natively (BatchedRCWA: precision="native" only
    # halves the element of complex64 problems); `streams` chunks are resident at once when the sweep is dealt to several streams
    elem = 8 if (precision == "native" and dtype == torch.complex64) else 16
    per_point = _POINT_MATRICES[1 if n_layers <= 1 else 2] * n * n * elem * max(1, int(streams))
    free, total = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)      # the caching allocator's idle blocks are ours to reuse
    budget = free - _HEADROOM * total
    fit = int(budget // per_point)
    if fit < 1:
        raise RuntimeError("torcwa_amd sweep: one sweep point at order %s needs about %.1f GB of HBM (%d x %d complex matrices x %.0f), but only "
                           "%.1f GB are free on %s (%.1f GB total, %.0f %% kept as headroom); free memory or lower the order"
                           % (list(order), per_point / 1e9, n, n, _POINT_MATRICES[1 if n_layers <= 1 else 2], free / 1e9, device, total / 1e9, 100 * _HEADROOM))
    if fit >= B:
        return B
    return fit if fit < 8 else fit - fit % 8


def _slice(v, lo, hi, B):
    """Per-point quantities are [B] vectors or [B,nx,ny] grids; a 2-D tensor is a grid SHARED by all points and is never cut
    (a 128 x 128 grid in a 128-point sweep is not a per-point quantity)."""
    if torch.is_tensor(v) and v.dim() in (1, 3) and v.shape[0] == B:
        return v[lo:hi]
    return v


def solve_stack_sweep(freq, layers, order, L, *, eps_in=None, eps_out=None, inc_ang=0.0, azi_ang=0.0, dtype=torch.complex64,
                      precision="high", engine=None, chunk=None, streams=1, orders=((0, 0),), polarization="xx",
                      direction="forward", port="transmission", check_info=True, eig_route="auto"):
    """B sweep points of a multi-layer stack (BASELINE.json configs 2-4): the reference's per-point Python loop
    (example/Example1-1.ipynb, Example3.ipynb) as chunks of a batched solve.  `layers` as in `_solve_chunk`, with
    per-point quantities carrying a leading dimension B = len(freq).  Returns the requested S-parameter [B, len(orders)].

    eig_route: "auto" (mixed-precision eigensolver; once a chunk of THIS call had to redo matrices in fp64, the remaining layers and chunks
    of this call use the all-fp64 route -- BatchedRCWA._eig_call), "mixed" or "fp64"."""
    from .engine import default_engine
    B = freq.shape[0]
    eng = engine if engine is not None else default_engine()
    old_check, eng.check_info = eng.check_info, check_info         # restored below: the engine may be shared with other solvers
    # chunk=None: as many points in lock-step as the free HBM holds (the reference's per-point loop cannot run out of memory; neither must this)
    chunk = auto_chunk(B, order, len(layers), precision, freq.device, dtype=dtype, streams=streams) if not chunk else int(chunk)
    if streams > 1 and chunk >= B:
        chunk = -(-B // streams)
    spans = [(lo, min(B, lo + chunk)) for lo in range(0, B, chunk)]
    outs = [None] * len(spans)
    route_hint = {}                     # shared by the chunks of this call only

    def run(i):
        lo, hi = spans[i]
        lays = [tuple(_slice(v, lo, hi, B) for v in lay) for lay in layers]
        outs[i] = _solve_chunk(freq[lo:hi], lays, order, L, _slice(eps_in, lo, hi, B), _slice(eps_out, lo, hi, B), _slice(inc_ang, lo, hi, B),
                               _slice(azi_ang, lo, hi, B), dtype, precision, engine, orders, polarization, direction, port, check_info,
                               eig_route=eig_route, route_hint=route_hint)

    dev = freq.device
    try:
        _run_spans(run, spans, streams, dev)
    finally:
        eng.check_info = old_check
    return torch.cat(outs, dim=0)


def _run_spans(run, spans, streams, dev):
    import threading
    if streams <= 1 or len(spans) == 1 or dev.type != "cuda":
        for i in range(len(spans)):
            run(i)
    else:
        cur = torch.cuda.current_stream(dev)
        pool = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        errors = []

        def worker(w):
            try:
                with torch.cuda.device(dev), torch.cuda.stream(pool[w]):
                    pool[w].wait_stream(cur)
                    for i in range(w, len(spans), streams):
                        run(i)
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        for st in pool:
            cur.wait_stream(st)


def solve_single_layer_sweep(freq, eps_grids, thickness, order, L, **kw):
    """B sweep points of a 1-patterned-layer stack (configs 2 and 4 of BASELINE.json): freq [B], eps_grids [B,nx,ny].

    chunk   : points solved in lock-step by one batched solver (bounds the HBM footprint; default None = as many as the free HBM holds with
              10 % headroom, `auto_chunk`).  At order [15,15] (n = 1922) a point costs about 0.6 GB allocated / 0.9 GB reserved, so about 256 points
              fit the 288 GB of an MI355X, and larger chunks are faster (measured, round 5: 15.7 / 22.1 / 27.9 / 32.2 layer-solves/s at 16 / 32 / 64 / 128 points).
    streams : number of HIP streams / host threads the chunks are dealt to (default 1: one lock-step chunk is faster on MI355X -- two half
              sweeps on two threads 18.3 layer-solves/s against 33.6, and 28.9 - 31.3 with CU-masked streams that keep the other half's GEMM
              grids off a reserved set of compute units: profiles/r06_ab/cumask.txt).
    """
    return solve_stack_sweep(freq.to(eps_grids.device), [(thickness, eps_grids)], order, L, **kw)
