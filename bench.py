#!/usr/bin/env python3
"""bench.py -- RCWA layer-solves/s on MI355X (BASELINE.json metric), one process per GPU.

Workload (BASELINE.json configs[1]): single patterned layer (Example-1 rectangle 180x100 nm of a-Si:H on glass, 300 nm
thick, 300x300 grid), Fourier order [15,15] (n = 2N = 1922), wavelength sweep linspace(400,700,128) nm, complex64 I/O.
A "step" = one pass of the hot path (conv-matrix -> P,Q -> eig -> layer S-matrix -> Redheffer with the input half
space -> S-parameter read-out) over one batch of `--batch` sweep points whose permittivity grids are already resident
in HBM.  N GPUs: every rank runs its own batch (weak scaling, no data-path collective); the only communication is
the final all_gather of the S-parameters (RCCL).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def make_inputs(batch, order, device, rank, seed_shift=0):
    from torcwa_amd.sweep import asih_eps_table, rectangle_density
    lam, eps_si = asih_eps_table()
    idx = (np.arange(batch) + rank * batch + seed_shift) % len(lam)
    lam_b, eps_b = lam[idx], eps_si[idx]
    dens = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float32, device=device)
    eps_t = torch.as_tensor(eps_b, dtype=torch.complex64, device=device)
    grids = dens[None] * eps_t[:, None, None] + (1. - dens[None])            # complex64 [B,300,300], as a c64 user builds it
    freq = torch.as_tensor(1.0 / lam_b, dtype=torch.float64, device=device)
    return freq, grids.contiguous(), lam_b


def run_step(freq, grids, order, engine, precision, chunk, streams):
    from torcwa_amd.sweep import solve_single_layer_sweep
    return solve_single_layer_sweep(freq, grids, 300., order, [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex64,
                                    precision=precision, engine=engine, chunk=chunk, streams=streams, check_info=False)


def cpu_baseline(order, lam_nm, eps_si, threads):
    """The reference's CPU path (oracle port, same op sequence), timed on the host cores on a bounded sample:
    one layer-solve in complex64 (the timed baseline, as the reference runs it) and the same point in complex128
    (the parity oracle)."""
    from oracle import rcwa_oracle as orc
    torch.set_num_threads(threads)
    dens = orc.rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float32)
    eps = (dens * complex(eps_si) + (1. - dens)).to(torch.complex64)
    t0 = time.perf_counter()
    s, lays, S, C = orc.solve_stack(1.0 / float(lam_nm), order, [300., 300.], [(300., eps, 1.0)], dtype=torch.complex64, eps_in=1.46 ** 2)
    v = orc.s_parameters(s, S, [0, 0])
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    s, lays, S, C = orc.solve_stack(1.0 / float(lam_nm), order, [300., 300.], [(300., eps.to(torch.complex128), 1.0)], dtype=torch.complex128, eps_in=1.46 ** 2)
    v128 = orc.s_parameters(s, S, [0, 0])
    dt128 = time.perf_counter() - t1
    return dt, complex(v[0]), dt128, complex(v128[0])


# MI355X dense matrix-core peaks for the arithmetic type of the path.  f32: 157.3 TF (MI355X_MICROARCH.md, "Peak FP32
# (matrix)"); f64: 78.6 TF (AMD MI355X datasheet; the guide only states that the f32 matrix rate equals the vector rate
# and f64 runs at half of it).  HBM3E: 8 TB/s spec.
PEAK_TFLOPS = {"high": 78.6, "native": 157.3}
PEAK_HBM_GBS = 8000.0


def roofline(engine, precision, elapsed, batch_per_gpu=None):
    """Live figures from the HIP events libtrx recorded around its dominant kernels during the timed region."""
    import ctypes
    tags = []
    for tag in range(6):
        buf = (ctypes.c_double * 6)()
        engine.lib.check(engine.lib.prof_get(tag, ctypes.addressof(buf)))
        launches, timed, flops, nbytes, ms, flops_all = list(buf)
        tags.append({"kernel": engine.lib.prof_tag_name(tag).decode(), "launches": int(launches), "timed_launches": int(timed),
                     "ms_timed": ms, "flops_timed": flops, "bytes_timed": nbytes, "flops_all": flops_all})
    times = [{"kernel": t["kernel"], "launches": t["launches"], "timed_launches": t["timed_launches"],
              "avg_us": (1e3 * t["ms_timed"] / t["timed_launches"]) if t["timed_launches"] else None,
              "share_of_wall": (t["ms_timed"] * (t["launches"] / max(t["timed_launches"], 1)) / (1e3 * elapsed)) if t["timed_launches"] else None}
             for t in tags]
    # dominant kernel with a known algorithmic work figure: the N,N complex GEMM on the matrix cores
    g = tags[0]
    roof = None
    if g["timed_launches"] > 0 and g["ms_timed"] > 0:
        ach = g["flops_timed"] / (g["ms_timed"] * 1e-3) / 1e12
        roof = {"kernel": g["kernel"], "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                "frac": ach / PEAK_TFLOPS[precision], "traffic": None,
                "avg_launch_us": 1e3 * g["ms_timed"] / g["timed_launches"], "launches_timed": g["timed_launches"],
                "algorithmic_flops_per_launch": g["flops_timed"] / g["timed_launches"],
                "algorithmic_bytes_per_launch": g["bytes_timed"] / g["timed_launches"],
                "note": "8 real flops per complex MAC x m*n*k*batch of each launch; events on the launch stream"}
        h = tags[5]
        if h["timed_launches"] > 0 and h["ms_timed"] > 0:
            gbs = h["bytes_timed"] / (h["ms_timed"] * 1e-3) / 1e9
            roof["secondary"] = {"kernel": h["kernel"], "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": gbs / PEAK_HBM_GBS, "avg_launch_us": 1e3 * h["ms_timed"] / h["timed_launches"]}
        a = tags[3]    # QR slab updates: the work is data dependent and counted on the device (all launches); time = avg of the timed launches
        if a["timed_launches"] > 0 and a["ms_timed"] > 0 and a["flops_all"] > 0:
            tf = a["flops_all"] / (a["ms_timed"] * 1e-3 * a["launches"] / a["timed_launches"]) / 1e12
            roof["qr_slab_updates"] = {"kernel": a["kernel"], "bound": "mfma", "achieved": tf, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
                                       "frac": tf / PEAK_TFLOPS[precision], "avg_launch_us": 1e3 * a["ms_timed"] / a["timed_launches"],
                                       "note": "flops = 8 ww^2 (2n - ww) per matrix and window step, summed on the device over all launches; up to 4 iteration "
                                               "groups run this kernel concurrently on their own streams, so the event-bracketed time of a launch includes "
                                               "the share of the GPU the other groups take: this is the rate ONE group sees, the aggregate is up to 4x"}
    # HBM traffic of the same kernel from the separate rocprofv3 --pmc passes (profiles/scripts/pmc_bench.sh; counters cannot be
    # collected from inside this process).  Only reported when the committed summary was taken at this batch size.
    if roof is not None:
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_bench.json")
        try:
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            if pmc.get("batch") == batch_per_gpu:
                kk = [v for k_, v in pmc["kernels"].items() if k_.startswith("gemm_mfma_kernel<double, 0, 0" if precision == "high" else "gemm_mfma_kernel<float, 0, 0")]
                if kk:
                    tot = sum(v["bytes_per_launch_corrected"] * v["launches"] for v in kk)
                    n_l = sum(v["launches"] for v in kk)
                    roof["traffic"] = tot / n_l
                    roof["traffic_note"] = ("HBM bytes per launch, average over the launches of this kernel in one step: 2*FETCH_SIZE + WRITE_SIZE from "
                                            "profiles/r01_pmc_bench.json (separate --pmc passes of the same command)")
        except (OSError, ValueError, KeyError):
            pass
    return roof, times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="sweep points per step per GPU")
    ap.add_argument("--order", type=int, default=15)
    ap.add_argument("--chunk", type=int, default=0, help="points solved concurrently (0 = whole batch)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams (host threads) the chunks of a step are dealt to")
    ap.add_argument("--precision", default="high", choices=["high", "native"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of one extra (untimed) step, top entries to stderr")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=device)

    import torcwa_amd
    from torcwa_amd.sweep import gather_sweep
    engine = torcwa_amd.Engine(device=device)
    order = [args.order, args.order]
    n = 2 * (2 * args.order + 1) ** 2
    chunk = args.chunk if args.chunk > 0 else -(-args.batch // max(1, args.streams))
    freq, grids, lam_b = make_inputs(args.batch, order, device, rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        try:
            out = run_step(freq, grids, order, engine, args.precision, chunk, args.streams)
        except (RuntimeError, torcwa_amd.TrxError) as e:
            # an untimed warm-up step may hit a device that is still releasing the memory of a previous process: free the
            # allocator cache, wait and try once more (the timed steps below are never retried)
            if w > 0:
                raise
            print("bench: warm-up step failed (%s); retrying once" % str(e).splitlines()[0], file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            time.sleep(10.0)
            out = run_step(freq, grids, order, engine, args.precision, chunk, args.streams)
    barrier()
    # HIP-event timing of the dominant kernels, recorded by libtrx on the launch stream during the timed region
    engine.lib.prof_reset()
    engine.lib.prof_enable(1)
    ms0 = torch.cuda.memory_stats(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run_step(freq, grids, order, engine, args.precision, chunk, args.streams)
    barrier()
    elapsed = time.perf_counter() - t0
    engine.lib.prof_enable(0)
    ms1 = torch.cuda.memory_stats(device)
    mem = {"peak_reserved_GB": ms1.get("reserved_bytes.all.peak", 0) / 1e9, "peak_allocated_GB": ms1.get("allocated_bytes.all.peak", 0) / 1e9,
           "device_mallocs_in_timed_region": ms1.get("segment.all.allocated", 0) - ms0.get("segment.all.allocated", 0),
           "device_frees_in_timed_region": ms1.get("segment.all.freed", 0) - ms0.get("segment.all.freed", 0),
           "alloc_retries": ms1.get("num_alloc_retries", 0)}
    n_fail = engine.failures()
    if n_fail:
        raise SystemExit(f"bench invalid: {n_fail} numerical failures (info != 0) inside the timed region")
    if args.host_profile and rank == 0:
        import cProfile, pstats, sys
        pr = cProfile.Profile()
        pr.enable()
        run_step(freq, grids, order, engine, args.precision, chunk, args.streams)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        full = gather_sweep(out, args.batch * world)       # the one collective of the job: final gather (RCCL)
    else:
        full = out
    total_solves = args.batch * world * args.steps
    value = total_solves / elapsed

    if rank == 0:
        res = {
            "metric": "RCWA layer-solves/sec (complex64 I/O) at Fourier order [%d,%d]" % (args.order, args.order),
            "value": value, "unit": "layer-solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "c128 arithmetic (complex64 I/O)" if args.precision == "high" else "c64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single patterned layer, order=[%d,%d] (n=%d), 300x300 grid, %d-lambda sweep per GPU, "
                                   "glass input half-space" % (args.order, args.order, n, args.batch),
                       "batch_per_gpu": args.batch, "chunk": chunk, "streams": args.streams, "precision": args.precision},
            "txx00_sample": [float(full[0, 0].real), float(full[0, 0].imag)], "numerical_failures": 0, "hbm": mem,
        }
        res["roofline"], res["kernel_times"] = roofline(engine, args.precision, elapsed, args.batch)
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads if args.cpu_threads > 0 else max(1, (os.cpu_count() or 2) // 2)
            from torcwa_amd.sweep import asih_eps_table
            lam, eps_si = asih_eps_table()
            dt, v, dt128, v128 = cpu_baseline(order, lam[0], eps_si[0], threads)
            got = complex(full[0, 0])
            res["parity_sample"] = {"point": "lambda=%.1f nm, txx(0,0)" % lam[0], "gpu": [got.real, got.imag],
                                    "oracle_c128": [v128.real, v128.imag], "rel_err_vs_c128_oracle": abs(got - v128) / abs(v128),
                                    "oracle_c64": [v.real, v.imag], "rel_err_of_c64_oracle_vs_c128_oracle": abs(v - v128) / abs(v128),
                                    "oracle_c128_seconds": dt128}
            res["cpu_baseline"] = {"value": 1.0 / dt, "unit": "layer-solves/s", "cores": threads, "kind": "port",
                                   "sample": "1 layer-solve (lambda=%.1f nm) of the same workload, complex64, oracle/rcwa_oracle.py on torch-CPU; "
                                             "txx00=%.6f%+.6fj" % (lam[0], v.real, v.imag)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
