#!/usr/bin/env python3
"""bench.py -- RCWA layer-solves/s on MI355X (BASELINE.json metric), one process per GPU.

Default workload (BASELINE.json configs[1]): single patterned layer (Example-1 rectangle 180x100 nm of a-Si:H on glass, 300 nm
thick, 300x300 grid), Fourier order [15,15] (n = 2N = 1922), wavelength sweep linspace(400,700,128) nm, complex64 I/O.
A "step" = one pass of the hot path (conv-matrix -> P,Q -> eig -> layer S-matrix -> Redheffer with the input half space ->
S-parameter read-out) over one batch of sweep points whose permittivity grids are already resident in HBM.

    python bench.py                                  # 1 GPU, 128-lambda sweep, the driver's BENCH line
    python bench.py --gpus 8                         # spawns 8 ranks itself (torch.distributed.run, RCCL); weak scaling: 128
                                                     # points per GPU, PLUS the strong-scaling leg north_star asks for (the SAME
                                                     # 128-lambda sweep split 8 ways) reported under "strong_scaling"
    python bench.py --gpus 8 --scaling strong        # only the split sweep (value = its throughput)
    python bench.py --gpus 8 --config 4              # configs[3]: 16x16x16 (Wx,Wy,lambda) sweep = 4096 solves, sharded (strong)
    python bench.py --config 3                       # configs[2]: 4 patterned layers, order [21,21], 64-lambda sweep (4 layer-solves per point)
    python bench.py --config 5                       # configs[4]: topology-optimisation step, order [25,25], complex128, forward + adjoint

Under `python -m torch.distributed.run ... bench.py --gpus N ...` (the driver's launch) the ranks are already there and nothing
is spawned.  No data-path collective exists: every rank solves its own contiguous block of the flattened sweep
(torcwa_amd.sweep.shard_range); the only communication is one all_gather of the S-parameters at the end.

Prints ONE JSON line on rank 0 (driver contract) with `roofline` (live HIP-event timing of the dominant kernels + the
per-layer-solve T_roof/T_measured of SURVEY.md 8(d)) and, at N = 1, `cpu_baseline` (the oracle port of the reference's CPU
path timed on the host cores).
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X dense matrix-core peaks for the arithmetic type of the path.  f32: 157.3 TF (MI355X_MICROARCH.md, "Peak FP32 (matrix)");
# f64: 78.6 TF (AMD MI355X datasheet; a register-only MFMA loop sustains 77.5 TF, profiles/r01_mfma_peak.txt).  HBM3E: 8 TB/s.
PEAK_TFLOPS = {"high": 78.6, "native": 157.3}
PEAK_HBM_GBS = 8000.0
EMU = os.environ.get("TRX_BENCH_EMU") == "1"       # launcher-plumbing test on CPU (tests/test_bench_launch.py); never a measurement


# ---------------------------------------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config: 2 = lambda sweep (configs[1]), 3 = 4-layer stack (configs[2]), 4 = (Wx,Wy,lambda) sweep (configs[3]), "
                         "5 = forward + adjoint of one topology-optimisation step (configs[4]; N > 1: independent replicas)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="default: weak for configs 2, 3, 5, strong for config 4")
    ap.add_argument("--batch", type=int, default=0, help="configs 2, 3: sweep points per GPU (weak) / in total (strong); 0 = 128 (config 2), 64 (config 3)")
    ap.add_argument("--points", type=int, default=4096, help="config 4: total sweep points (16^3 grid, cycled if larger)")
    ap.add_argument("--order", type=int, default=0, help="Fourier order [o,o]; 0 = the config's own: 15 (2, 4), 21 (3), 25 (5)")
    ap.add_argument("--grid", type=int, default=300, help="permittivity grid is grid x grid")
    ap.add_argument("--chunk", type=int, default=0, help="points solved in lock-step (0 = min(local points, 128); config 3: 64; config 4: what the sweep driver "
                    "fits into the free HBM with 10 %% headroom, torcwa_amd.sweep.auto_chunk)")
    ap.add_argument("--cyclic", action="store_true", help="config 4, N > 1: rank r solves points r, r + N, ... instead of a contiguous block (SURVEY.md 8(e))")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams (host threads) the chunks of a step are dealt to")
    ap.add_argument("--precision", default="high", choices=["high", "native"])
    ap.add_argument("--eig-route", default="auto", choices=["auto", "mixed", "fp64"], help="eigensolver route of the sweep drivers (torcwa_amd.sweep.solve_stack_sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-leg", action="store_true", help="N > 1, weak scaling: skip the additional strong-scaling measurement")
    ap.add_argument("--cpu-points", type=int, default=3, help="sweep points the CPU baseline times (>= 1)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--host-profile", action="store_true", help="cProfile of one extra (untimed) step, top entries to stderr")
    args = ap.parse_args(argv)
    if args.order <= 0:
        args.order = {2: 15, 3: 21, 4: 15, 5: 25}[args.config]
    if args.batch <= 0:
        args.batch = {2: 128, 3: 64, 4: 128, 5: 1}[args.config]
    if args.config == 5:
        args.batch = 1
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: re-execute this script under torch.distributed.run with N ranks (one per
    GPU; rendezvous on 127.0.0.1) and pass its exit code on.  Rank 0 of that job prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------------
# workloads: sweep point g (global index) -> (frequency, permittivity grid)
# ---------------------------------------------------------------------------------------------------------------------------
def make_inputs(config, idx, grid, device):
    """Inputs of the sweep points with global indices `idx` (numpy int array), built on `device` (resident before the timed
    region).  config 2: fixed rectangle 180 x 100, lambda_g = linspace(400,700,128)[g mod 128].  config 4: g -> (iw, jw, kl) of
    the 16^3 grid Wx, Wy in linspace(50,250,16), lambda in linspace(400,700,16) (example/Example3.ipynb:85-101 at BASELINE size)."""
    from torcwa_amd.geometry import geometry
    from torcwa_amd.materials import asih_nk
    from torcwa_amd.sweep import asih_eps_table
    geo = geometry(Lx=300., Ly=300., nx=grid, ny=grid, edge_sharpness=1000., dtype=torch.float32, device=device)
    if config == 2:
        lam_t, eps_t = asih_eps_table()
        k = idx % len(lam_t)
        lam, eps_si = lam_t[k], eps_t[k]
        dens = geo.rectangle(Wx=180., Wy=100., Cx=150., Cy=150.)[None]
    else:
        iw, jw, kl = np.unravel_index(idx % 4096, (16, 16, 16))
        wv = np.linspace(50., 250., 16)
        lam = np.linspace(400., 700., 16)[kl]
        eps_si = (asih_nk(torch.from_numpy(lam)) ** 2).numpy()
        wx = torch.as_tensor(wv[iw], dtype=torch.float32, device=device)[:, None, None]
        wy = torch.as_tensor(wv[jw], dtype=torch.float32, device=device)[:, None, None]
        dens = geo.rectangle(Wx=wx, Wy=wy, Cx=150., Cy=150.)
    eps_c = torch.as_tensor(eps_si, dtype=torch.complex64, device=device)
    grids = dens * eps_c[:, None, None] + (1. - dens)                      # complex64 [B,grid,grid], as a complex64 user builds it
    freq = torch.as_tensor(1.0 / lam, dtype=torch.float64, device=device)
    return freq, grids.contiguous(), lam, eps_si


def make_inputs_stack(idx, nl, grid, device):
    """config 3 (Example1-1 style): four 200 nm layers, each the 180 x 100 nm a-Si:H rectangle rotated by 0 / 30 / 60 / 90 degrees in
    SU-8 (n = 1.6), glass input half space; sweep point g -> the (g mod nl)-th of nl wavelengths spread over 400-700 nm."""
    from torcwa_amd.sweep import asih_eps_table, rectangle_density
    lam_t, eps_t = asih_eps_table()
    k = np.linspace(0, len(lam_t) - 1, nl).round().astype(int)[idx % nl]
    lam, eps_si = lam_t[k], eps_t[k]
    eps_c = torch.as_tensor(eps_si, dtype=torch.complex64, device=device)
    layers = []
    for th in (0., 30., 60., 90.):
        d = rectangle_density(grid, grid, 300., 300., 180., 100., 150., 150., theta=th / 180 * np.pi, dtype=torch.float32, device=device)
        layers.append((200., (d[None] * eps_c[:, None, None] + (1. - d[None]) * 1.6 ** 2).contiguous()))
    freq = torch.as_tensor(1.0 / lam, dtype=torch.float64, device=device)
    return freq, layers, lam, eps_si


def make_inputs_topopt(device):
    """config 5 (Example6 style): a smooth random density on the 700 x 300 nm cell (700 x 300 grid, mirror-symmetric in y, Gaussian
    blur of radius 20 via FFT like the notebook), silicon at 532 nm."""
    nx, ny = 700, 300
    rho = torch.rand(nx, ny, generator=torch.Generator().manual_seed(333), dtype=torch.float64)
    rho = (rho + torch.flip(rho, dims=[1])) / 2
    kx, ky = torch.fft.fftfreq(nx, d=1.0)[:, None], torch.fft.fftfreq(ny, d=1.0)[None, :]
    blur = torch.exp(-2 * (np.pi * 20.0) ** 2 * (kx ** 2 + ky ** 2) / 4)
    return torch.real(torch.fft.ifft2(torch.fft.fft2(rho) * blur)).clamp(0, 1).to(device)


TOPOPT_EPS = 12.011610263133004 + 0.525912014756j
_grad_norm = [None]


def run_step_topopt(rho0, order, engine):
    """One optimiser step of config 5: figure of merit (power into the +1 order, all four polarisation pairs) and its gradient with
    respect to the density, through the stabilised eigendecomposition adjoint (torcwa/torch_eig.py)."""
    import torcwa_amd
    rho = rho0.clone().requires_grad_(True)
    sim = torcwa_amd.rcwa(freq=1 / 532., order=order, L=[700., 300.], dtype=torch.complex128, device=rho0.device, stable_eig_grad=True, engine=engine)
    sim.add_input_layer(eps=1.46 ** 2)
    sim.set_incident_angle(inc_ang=0., azi_ang=0.)
    sim.add_layer(thickness=300., eps=rho * TOPOPT_EPS + (1. - rho))
    sim.solve_global_smatrix()
    t = [sim.S_parameters(orders=[1, 0], direction='forward', port='transmission', polarization=p, ref_order=[0, 0]) for p in ('xx', 'yx', 'xy', 'yy')]
    fom = sum(torch.abs(v) ** 2 for v in t).sum()
    fom.backward()
    _grad_norm[0] = float(rho.grad.norm())
    return fom.detach().to(torch.complex128).reshape(1, 1)


_side_stream = [None]


def run_step(freq, grids, order, engine, args, chunk):
    from torcwa_amd.sweep import solve_single_layer_sweep, solve_stack_sweep
    if os.environ.get("TRX_BENCH_SIDE_STREAM") and freq is not None and freq.device.type == "cuda" and torch.cuda.current_stream(freq.device) == torch.cuda.default_stream(freq.device):
        # experiment (profiles/scripts/r4_first_call.sh): run the step on a created stream instead of the legacy null stream, i.e. put
        # iteration group 0 of trx_eig's QR phase on another hardware queue than the one the null stream owns
        if _side_stream[0] is None:
            _side_stream[0] = torch.cuda.Stream(device=freq.device)
        _side_stream[0].wait_stream(torch.cuda.current_stream(freq.device))
        with torch.cuda.stream(_side_stream[0]):
            out = run_step(freq, grids, order, engine, args, chunk)
        torch.cuda.current_stream(freq.device).wait_stream(_side_stream[0])
        return out
    if args.config == 5:
        return run_step_topopt(grids, order, engine)
    if args.config == 3:
        return solve_stack_sweep(freq, grids, order, [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex64, precision=args.precision, engine=engine,
                                 chunk=chunk, streams=args.streams, orders=[(0, 0)], polarization="xx", check_info=False, eig_route=args.eig_route)
    return solve_single_layer_sweep(freq, grids, 300., order, [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex64,
                                    precision=args.precision, engine=engine, chunk=chunk, streams=args.streams, check_info=False, eig_route=args.eig_route)


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle port of the reference, test infrastructure -- only this leg imports it)
# ---------------------------------------------------------------------------------------------------------------------------
def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) if cores else max(1, logical // 2)), logical


def cpu_baseline(order, grid, lams, eps_sis, threads):
    """The reference's CPU path (oracle/rcwa_oracle.py: same op sequence -- 1 eig / 12 inv / 48 matmul per layer-solve), timed
    on the host cores on a BOUNDED sample: len(lams) sweep points of the same workload in complex64 with denormals NOT flushed
    (as the reference runs), one of them again with flush-to-zero (footnote), and every point again in complex128
    (the parity oracle of the timed sweep)."""
    from oracle import rcwa_oracle as orc
    torch.set_num_threads(threads)
    dens = orc.rectangle_density(grid, grid, 300., 300., 180., 100., 150., 150., dtype=torch.float32)

    def solve(lam, eps_si, cdt):
        eps = (dens * torch.tensor(complex(eps_si), dtype=torch.complex64) + (1. - dens)).to(torch.complex64).to(cdt)
        t0 = time.perf_counter()
        s, lays, S, C = orc.solve_stack(1.0 / float(lam), order, [300., 300.], [(300., eps, 1.0)], dtype=cdt, eps_in=1.46 ** 2)
        v = complex(orc.s_parameters(s, S, [0, 0])[0])
        return time.perf_counter() - t0, v

    secs, vals = [], []
    for lam, e in zip(lams, eps_sis):
        dt, v = solve(lam, e, torch.complex64)
        secs.append(dt)
        vals.append(v)
    torch.set_flush_denormal(True)
    dt_flush, _ = solve(lams[0], eps_sis[0], torch.complex64)
    torch.set_flush_denormal(False)
    dt128, v128 = [], []
    for lam, e in zip(lams, eps_sis):          # the parity oracle at EVERY baseline point
        dt, v = solve(lam, e, torch.complex128)
        dt128.append(dt)
        v128.append(v)
    return secs, vals, dt_flush, dt128, v128


# ---------------------------------------------------------------------------------------------------------------------------
# roofline
# ---------------------------------------------------------------------------------------------------------------------------
def csrc_sha16():
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "torcwa_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "torcwa_amd", "csrc", "*.hpp"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def csrc_file_hashes():
    out = {}
    for p in sorted(glob.glob(os.path.join(ROOT, "torcwa_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "torcwa_amd", "csrc", "*.hpp"))):
        out[os.path.basename(p)] = hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
    return out


def profile_valid_for(prof, tag):
    """A committed profile (kernel trace / counter passes) may be quoted for `tag` when the source files that define the tag's kernels are
    the ones it was taken on (per-file hashes in the profile; profiles without them: the hash of the whole csrc directory)."""
    per_file = prof.get("src_sha16")
    if not per_file:
        return prof.get("csrc_sha16") == csrc_sha16()
    files = next((v for k, v in _TAG_SOURCES.items() if tag.startswith(k)), None)
    now = csrc_file_hashes()
    if files is None:
        return per_file == now
    return all(per_file.get(f) == now.get(f) for f in files)


def layer_solve_roof(n, precision):
    """nominal work of one patterned layer-solve, F_dense = 160 n^3, F_eig = 100 n^3 real flops, B_eig,min =
    elem * n^3 / 3 bytes; T_roof = F_dense / P + max(F_eig / P, B_eig,min / 8 TB/s)."""
    n3 = float(n) ** 3
    out = {}
    for key, peak, elem in (("survey_fp32", 157.3e12, 8), ("fp64", 78.6e12, 16)):
        out[key] = 1e3 * (160 * n3 / peak + max(100 * n3 / peak, elem * n3 / 3 / 8e12))
    return out


# instrumented tags (torcwa_amd/csrc/prof.hip).  "gemm<N,N>" = one trx gemm() call with both operands untransposed: the large-tile kernel
# gemm_big_kernel plus the narrow-tile launches that take its thin remainders, or one gemm_mfma_kernel launch for small / panel shapes.
_BOUND = {"gemm<N,N>": "mfma", "gemm<other ops>": "mfma", "apply_links_kernel<0>": "mfma", "apply_links_kernel<1>": "mfma", "hess_gemv_kernel": "hbm",
          "gemm<N,N> fp32": "mfma", "gemm<other ops> fp32": "mfma",
          "qr_window_kernel": "latency", "qr_prepare_kernel": "latency", "hess_col_kernel": "latency", "lu_panel_kernel": "latency"}
# Latency-bound kernels (one wave / one workgroup per matrix walking a chain of dependent steps): the roof is the ISSUE-LIMITED time of the
# chain -- what the dependent instructions of one step cost when nothing else waits -- in shader cycles per step (the models are stated here,
# next to the measurement they are held against; the clock is the 2.4 GHz the matrix peaks are quoted at).  `achieved` = steps per second and
# matrix, `peak` = clock / cycles_min, frac = t_min / t_measured of a launch.
_CLOCK_HZ = 2.4e9
_LATENCY_MODEL = {
    # chain step of the bulge chase (qr_window_kernel, eig_qr.hip): rotation from two LDS operands (LDS round trip ~64 cycles + ~12 dependent
    # VALU incl. one quarter-rate v_rsq: ~90) ; left phase (LDS read round trip, 8 FMAs, write: ~100) ; barrier ; right phase (~100) ; barrier
    # (16 waves: ~30 each at best) => ~400 cycles.  Steps are counted on the device (chain steps per matrix).
    "qr_window_kernel": dict(unit="chain steps/s per matrix", cycles_min=400.0, steps="device",
                             model="rotation generate ~150 + left ~100 + right ~100 cycles + 2 barriers of a 16-wave workgroup ~50 = 400 cycles per chain step"),
    # rotation of the explicit-shift QR of the AED window (qr_prepare_kernel: small_schur): v_readlane broadcast of (f, g), rotation generate
    # (~90), own column pair update + LDS write + next row read (~64 + 20) => ~180 cycles; the right / U phases add ~1/3 (lanes independent).
    "qr_prepare_kernel": dict(unit="rotations/s per matrix", cycles_min=240.0, steps="device",
                              model="broadcast + rotation generate ~90 + column update and LDS round trip ~90 cycles per rotation of the left phase, + 1/3 for the lane-parallel right / U phases = 240"),
    # one Householder column (hess_col_kernel): combine the partial sums (one global round trip ~2000 cycles under load), one combined reduction over
    # the column (LDS tree, ~10 barriers x ~60), write reflector + T column (one more global round trip) => ~5000 cycles per column
    "hess_col_kernel": dict(unit="columns/s per matrix", cycles_min=5000.0, steps=1.0,
                            model="2 dependent global round trips (~2000 cycles each under load) + one block reduction (~10 barriers) per column = 5000 cycles"),
    # one 32-column LU panel (lu.hip; since round 6 in sub-blocks of 8 columns: per column a pivot search and a rank-1 update of the sub-block --
    # in LDS for panels of up to ~1000 rows, else one launch per column over W workgroups per matrix -- and per sub-block one rank-8 pass over the
    # rest of the panel): pivot search ~2500 + rank-1 update ~2000 cycles per column as the issue-limited floor of a column
    "lu_panel_kernel": dict(unit="panel columns/s per matrix", cycles_min=4500.0, steps=32.0,
                            model="pivot search ~2500 + rank-1 panel update ~2000 cycles per column, 32 columns per panel (launch latency of the per-column launches NOT included in the minimum)"),
}
# kernels of a rocprofv3 trace that belong to a tag (prefixes of profiles/kernel_stats.py's short names), and the source files that define them
# (a committed profile stays valid for a tag as long as THESE files are unchanged; other kernels may have moved on)
_TRACE_KEYS = {"gemm<N,N>": ("gemm_big_kernel<0, 0", "gemm_mfma_kernel<double, 0, 0"), "gemm<N,N> fp32": ("gemm_mfma_kernel<float, 0, 0",),
               "apply_links_kernel<0>": ("apply_links_kernel<float, 0", "apply_links_kernel<double, 0", "apply_links_kernel<float, 2", "apply_links_kernel<double, 2"),
               "apply_links_kernel<1>": ("apply_links_kernel<float, 1", "apply_links_kernel<double, 1"),
               "hess_gemv_kernel": ("hess_gemv_kernel",), "qr_prepare_kernel": ("qr_prepare_kernel",),
               "qr_window_kernel": ("qr_window_kernel",), "hess_col_kernel": ("hess_col_kernel",), "lu_panel_kernel": ("lu_panel_kernel", "lu_panel_lds_kernel", "lu_split_")}
_TAG_SOURCES = {"gemm": ("gemm.hip", "gemm_big.hip", "mfma.hpp", "common.hpp"), "apply": ("eig_qr.hip", "mfma.hpp", "common.hpp"),
                "qr_": ("eig_qr.hip", "common.hpp"), "hess": ("eig_hess.hip", "common.hpp"), "lu_": ("lu.hip", "common.hpp")}
# kernels of the eigensolver proper: with the mixed-precision route (libtrx default for complex128 input, n >= 256, batch >= 8) they run in fp32
_EIG_STAGE = ("apply_links_kernel<0>", "apply_links_kernel<1>", "qr_prepare_kernel", "qr_window_kernel", "hess_gemv_kernel", "hess_col_kernel")


def eig_is_mixed(args, n, chunk):
    return args.precision == "high" and os.environ.get("TRX_EIG_VEC", "0") in ("0", "3") and n >= 256 and chunk >= 8 and args.config != 5 and args.eig_route != "fp64"


def read_prof_tags(engine):
    """{tag name: (launches, timed launches, flops of the timed, bytes of the timed, ms of the timed, flops over ALL launches, bytes over ALL launches)} for every tag
    libtrx instruments (kernel tags and the wall-clock phase tags of trx_eig)."""
    import ctypes
    out = {}
    for tag in range(64):
        name = engine.lib.prof_tag_name(tag).decode()
        if name == "?":
            break
        buf = (ctypes.c_double * 7)()
        engine.lib.check(engine.lib.prof_get(tag, ctypes.addressof(buf)))
        out[name] = tuple(buf)
    return out


def phase_table(tags, py_phases, elapsed, steps):
    """Wall-clock phases of a step: the Python-level brackets of torcwa_amd.Engine (event pairs on the compute stream around the library calls of a
    layer-solve) and, inside trx_eig, the library's own phase brackets (balance / Hessenberg / QR from fork to join of its iteration groups /
    Schur vectors / Newton refinement).  share = of the measured step."""
    step_ms = 1e3 * elapsed / steps
    rows = []
    for name, ms in sorted(py_phases.items(), key=lambda kv: -kv[1]):
        rows.append({"phase": name, "ms_per_step": ms / steps, "share_of_step": ms / steps / step_ms})
    inner = []
    for name, v in tags.items():
        if name.startswith("phase:") and v[1] > 0:
            ms = v[4] / v[1] * v[0]
            inner.append({"phase": "trx_eig / " + name[6:], "ms_per_step": ms / steps, "share_of_step": ms / steps / step_ms, "calls_per_step": v[0] / steps})
    inner.sort(key=lambda r: -r["ms_per_step"])
    return {"step_ms": step_ms, "phases": rows, "inside_trx_eig": inner,
            "note": "event-timed on the compute stream, summed over the timed steps / steps; phases are sequential on that stream, so shares add up to ~1 "
                    "(the remainder is torch glue between the library calls)"}


def roofline(engine, args, elapsed, steps, n, units_per_step, chunk=128):
    """Live figures from the HIP events libtrx recorded (on the launch streams, uniformly sampled) during the timed region."""
    tags = read_prof_tags(engine)
    kernels = []
    for name, (launches, timed, flops_t, bytes_t, ms, flops_all, bytes_all) in tags.items():
        if name.startswith("phase:") or timed <= 0 or ms <= 0:          # phases: see phase_table; not launched, or no usable event timing (CPU emulator)
            continue
        fp32_kernel = name.endswith("fp32") or args.precision == "native" or (name in _EIG_STAGE and eig_is_mixed(args, n, chunk))
        peak_tf = PEAK_TFLOPS["native" if fp32_kernel else "high"]
        avg_us = 1e3 * ms / timed
        k = {"kernel": name, "launches": int(launches), "timed_launches": int(timed), "avg_us": avg_us, "arithmetic": "fp32" if fp32_kernel else "fp64",
             "est_total_ms_per_step": avg_us * launches / 1e3 / steps, "sum_over_wall": avg_us * launches / 1e6 / elapsed}
        bound = _BOUND.get(name)
        if name.startswith("apply_links_kernel") and flops_all > 0:
            # data-dependent work, counted on the device over ALL launches; time = uniform-sample average x launches
            k.update(bound="mfma", achieved=flops_all / (avg_us * 1e-6 * launches) / 1e12, peak=peak_tf, unit="TFLOP/s",
                     algorithmic_flops_per_launch=flops_all / launches,
                     note="flops = 8 ww^2 x (columns right of the band | rows above the window + n) per matrix and link (4M count; the banded product of a chase "
                          "unitary issues 13/16 of them); the iteration groups of the QR phase run their kernels concurrently on their own streams, so a "
                          "launch's event time includes the share of the GPU the others take")
        elif bound == "mfma" and flops_t > 0:
            # achieved: work of the SAMPLED launches / their event time.  Work per launch (what is scaled to the step and held against the
            # committed profiles): the EXACT mean over all launches -- a tag mixes a few 7 TFLOP calls with a thousand small ones, and which of
            # the big ones the sampling stride hits moved the sampled mean by 7 % between a 3-step and a 20-step run
            k.update(bound="mfma", achieved=flops_t / (ms * 1e-3) / 1e12, peak=peak_tf, unit="TFLOP/s",
                     algorithmic_flops_per_launch=flops_all / launches, algorithmic_bytes_per_launch=bytes_all / launches,
                     sampled_flops_per_launch=flops_t / timed)
        elif bound == "hbm" and bytes_t > 0:
            k.update(bound="hbm", achieved=bytes_t / (ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", algorithmic_bytes_per_launch=bytes_all / launches,
                     sampled_bytes_per_launch=bytes_t / timed)
        elif bound == "latency":
            m = _LATENCY_MODEL[name]
            steps_per_launch = (flops_all / launches) if m["steps"] == "device" else m["steps"]
            if steps_per_launch <= 0:
                continue
            if name == "qr_window_kernel" and bytes_all > 0:
                # fused launches: the far workgroups and the chase workgroup's catch-up apply the PREVIOUS launch's links from the left inside this
                # kernel; the library reports those flops in this tag's bytes slot (csrc/eig_qr.hip), apply_links_kernel<0> holds the stand-alone rest
                k["fused_left_update_tflop_per_step"] = bytes_all / steps / 1e12
            t_min_us = steps_per_launch * m["cycles_min"] / _CLOCK_HZ * 1e6
            k.update(bound="latency", achieved=steps_per_launch / (avg_us * 1e-6), peak=_CLOCK_HZ / m["cycles_min"], unit=m["unit"],
                     dependent_steps_per_launch=steps_per_launch, cycles_per_step_measured=avg_us * 1e-6 * _CLOCK_HZ / steps_per_launch,
                     cycles_per_step_min=m["cycles_min"], t_min_us_per_launch=t_min_us, model=m["model"],
                     note="latency-bound chain (one wave or workgroup per matrix): roof = issue-limited cycles of the dependent steps of one launch")
        else:
            continue
        k["frac"] = k["achieved"] / k["peak"]
        if name in ("gemm<N,N>", "gemm<other ops>") and not fp32_kernel and k["achieved"] > 0:
            # fp64 path: 3M complex product, three real MFMAs where the 8-flops-per-complex-MAC count has four
            k["issued_mfma_tflops"] = 0.75 * k["achieved"]
            k["issued_mfma_frac"] = 0.75 * k["frac"]
            k["flop_count_note"] = "achieved counts 8 real flops per complex MAC (TF-equivalent); the 3M product issues 6, so the matrix pipe is at issued_mfma_frac"
        k.update(profile_fracs(k, args, k["peak"], steps))
        kernels.append(k)
    if not kernels:
        return None
    ph = phase_table(tags, engine.phase_report(), elapsed, steps)
    # The kernels of the QR phase run on the phase's iteration groups (2 - 4 streams side by side), so their summed event times overlap each other:
    # wall share = event sum / (sum of the phase's kernel event times / the phase's own fork-to-join time).  Everything else runs on ONE stream.
    qr_tags = ("qr_window_kernel", "qr_prepare_kernel", "apply_links_kernel<0>", "apply_links_kernel<1>", "apply_links_kernel<2>")
    qr_sum = sum(k["est_total_ms_per_step"] for k in kernels if k["kernel"] in qr_tags)
    qr_wall = sum(r["ms_per_step"] for r in ph["inside_trx_eig"] if r["phase"] == "trx_eig / qr")
    qr_conc = max(1.0, qr_sum / qr_wall) if qr_wall > 0 else 1.0
    for k in kernels:
        conc = qr_conc if k["kernel"] in qr_tags else 1.0
        k["streams_side_by_side"] = conc
        k["wall_ms_per_step"] = k["est_total_ms_per_step"] / conc
    kernels.sort(key=lambda k: -k["wall_ms_per_step"])
    dom = dict(kernels[0])
    dom["chosen_by"] = ("largest wall-clock share among the instrumented kernels: summed event time (uniform-sample average x launches), and for the kernels "
                        "of the QR phase that sum divided by the %.2f group streams the phase ran side by side (sum of its kernels' event times / its "
                        "fork-to-join time)" % qr_conc)
    dom["traffic"], dom["traffic_note"] = pmc_traffic(dom, args, steps)
    if dom["traffic"] and dom.get("algorithmic_flops_per_launch"):
        # which roof binds, from the MEASURED traffic: time the bytes need at the HBM peak against the time the issued flops need at the matrix peak
        issued = dom["algorithmic_flops_per_launch"] * (0.75 if "issued_mfma_frac" in dom else 1.0)
        t_hbm, t_mfma = dom["traffic"] / (PEAK_HBM_GBS * 1e9), issued / (dom["peak"] * 1e12)
        dom["bound"] = "hbm" if t_hbm > t_mfma else "mfma"
        dom["bound_note"] = "measured traffic %.1f GB per call = %.2f ms at 8 TB/s vs %.2f ms of issued matrix-core work at the peak; %.1fx the algorithmic bytes" % (
            dom["traffic"] / 1e9, 1e3 * t_hbm, 1e3 * t_mfma, dom["traffic"] / max(dom.get("algorithmic_bytes_per_launch", 0.0), 1.0))
    t_meas = 1e3 * elapsed / (steps * units_per_step)
    roofs = layer_solve_roof(n, args.precision)
    layer = {
        "t_measured_ms": t_meas, "definition": "SURVEY.md 8(d): T_roof / T_measured per patterned layer-solve, nominal 260 n^3 flops + n^3/3 element reads",
        "t_roof_ms_at_fp32_peak": roofs["survey_fp32"], "frac_at_fp32_peak": roofs["survey_fp32"] / t_meas,
        "t_roof_ms_at_fp64_peak": roofs["fp64"], "frac_at_fp64_peak": roofs["fp64"] / t_meas,
        "priced_at": "fp64 (78.6 TF, 16-byte elements): the path delivers complex128 results (DESIGN.md section 3; S-matrix algebra and eigen-refinement "
                     "in fp64, the first stage of the mixed-precision eigensolver in fp32); the survey's own figure prices the same flops at the fp32 peak"
                     if args.precision == "high" else "fp32 (157.3 TF, 8-byte elements)"}
    red = sum(r["share_of_step"] for r in ph["phases"] if r["phase"].startswith("Redheffer"))
    # HEADLINE = the figure the target is defined on: the whole patterned layer-solve against its roofline, priced as the survey
    # wrote it -- 260 n^3 nominal real flops at the fp32 matrix peak (its eigensolver term is flop-bound there: 100 n^3 / 157.3 TF > the n^3 / 3
    # element stream at 8 TB/s), so T_roof / T_measured = (260 n^3 / T_measured) / 157.3 TF.  The dominant kernel is a sub-field.
    nominal_tf = 260.0 * float(n) ** 3 / (t_meas * 1e-3) / 1e12
    head = {"bound": "mfma", "achieved": nominal_tf, "peak": PEAK_TFLOPS["native"], "unit": "TFLOP/s", "frac": layer["frac_at_fp32_peak"],
            "traffic": dom["traffic"], "traffic_note": "of the dominant kernel, per call: " + str(dom["traffic_note"]),
            "scope": "one patterned layer-solve (conv-matrix + P,Q + eig + layer S-matrix + its Redheffer product), SURVEY.md 8(d): achieved = 260 n^3 nominal "
                     "real flops / measured time per layer-solve; frac = T_roof / T_measured = %.1f ms / %.1f ms" % (roofs["survey_fp32"], t_meas),
            "layer_solve": layer, "dominant_kernel": dom, "kernels": kernels, "phases": ph, "redheffer_share_of_step": red}
    return head


PROFILE_TAG = "r06"          # the committed profiles of this round: profiles/<tag>_kernel_profile.json, profiles/<tag>_pmc_bench.json


def _tag_kernels(prof, name):
    keys = _TRACE_KEYS.get(name)
    if not keys:
        return []
    return [v for k_, v in prof.get("kernels", {}).items() if k_.startswith(keys)]


def profile_fracs(k, args, peak, steps):
    """The same tag's roofline fraction recomputed from the COMMITTED profiles, so that the line and profiles/ can be held against each
    other: `frac_rocprof` = algorithmic work of the tag per step / the summed duration of its kernels per step in the rocprofv3
    --kernel-trace of the same command (in situ, kernels of the other iteration groups running next to it), `frac_alone` = the same with the
    durations of the --pmc passes (rocprofv3 serialises kernels there).  A gemm() call is one to three launches (large tile + peeled
    remainders), hence per-step sums on both sides.  Refused unless the profile was taken on the same sources OF THESE KERNELS at this batch."""
    out = {}
    work = k.get("algorithmic_flops_per_launch") if k.get("bound") == "mfma" else k.get("algorithmic_bytes_per_launch")
    name = k["kernel"]
    if not work or args.config != 2 or name not in _TRACE_KEYS:
        return out
    work_per_step = work * k["launches"] / steps
    for field, fn in (("frac_rocprof", "%s_kernel_profile.json" % PROFILE_TAG), ("frac_alone", "%s_pmc_bench.json" % PROFILE_TAG)):
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", fn)))
        except (OSError, ValueError):
            out[field], out[field + "_note"] = None, "profiles/%s not committed" % fn
            continue
        if not profile_valid_for(prof, name) or prof.get("batch") != args.batch:
            out[field], out[field + "_note"] = None, "profiles/%s was taken on other sources of this kernel or another batch size: refused" % fn
            continue
        kk = _tag_kernels(prof, name)
        if not kk:
            out[field], out[field + "_note"] = None, "kernel not in profiles/%s" % fn
            continue
        ms_per_step = sum(v.get("total_ms", v.get("ms_total", 0.0)) for v in kk) / max(prof.get("steps_traced", 1), 1)
        rate = work_per_step / (ms_per_step * 1e-3) / (1e12 if k.get("bound") == "mfma" else 1e9)
        out[field] = rate / peak
        out[field + "_note"] = "%.1f ms per step over %d launches per step in profiles/%s" % (
            ms_per_step, sum(v["launches"] for v in kk) // max(prof.get("steps_traced", 1), 1), fn)
    return out


def pmc_traffic(dom, args, steps):
    """HBM-side bytes per call of the dominant tag from the separate rocprofv3 --pmc passes (profiles/scripts/pmc_bench.sh; counters cannot
    be collected from inside this process): 2 x FETCH_SIZE + WRITE_SIZE summed over the tag's kernels, per step, divided by the calls per
    step.  Accepted only when the summary was taken on the same sources of these kernels and at this batch size -- otherwise null."""
    fn = "%s_pmc_bench.json" % PROFILE_TAG
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
    except (OSError, ValueError):
        return None, "profiles/%s not committed" % fn
    if not profile_valid_for(pmc, dom["kernel"]):
        return None, "profiles/%s was taken on other sources of this kernel: refused" % fn
    if pmc.get("batch") != args.batch or args.config != 2:
        return None, "profiles/%s was taken at another workload" % fn
    kk = _tag_kernels(pmc, dom["kernel"])
    if not kk:
        return None, "kernel not in the PMC summary"
    per_step = sum(v["bytes_per_launch_corrected"] * v["launches"] for v in kk) / max(pmc.get("steps_traced", 1), 1)
    calls_per_step = dom["launches"] / steps
    return per_step / calls_per_step, ("bytes per call crossing the L2 -> fabric boundary (2*FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are counted, "
                                       "so this bounds the HBM traffic from above), separate --pmc passes of the same command restricted to these kernels, "
                                       "same kernel sources: profiles/%s" % fn)


# ---------------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    import torcwa_amd
    from torcwa_amd.sweep import gather_sweep, shard_range
    if EMU:
        from tests.emu import emu_lib
        device = torch.device("cpu")
        engine = torcwa_amd.Engine(lib=emu_lib(), device="cpu")
        if world > 1:
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=device)        # "nccl" is RCCL on ROCm
        engine = torcwa_amd.Engine(device=device)
    world = dist.get_world_size() if world > 1 else 1                          # the world size RCCL actually formed

    scaling = args.scaling or ("strong" if args.config == 4 else "weak")
    if args.config == 5:
        scaling = "weak"                                                            # replicas only: one optimisation step per GPU
    layers_per_point = 4 if args.config == 3 else 1
    order = [args.order, args.order]
    n = 2 * (2 * args.order + 1) ** 2

    def sync():
        if not EMU:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    cyclic = bool(args.cyclic and args.config == 4)

    def local_block(total):
        from torcwa_amd.sweep import shard_indices
        return shard_indices(total, rank, world, cyclic=cyclic)

    def measure(idx, steps, warmup, profile):
        """W untimed + K timed steps over this rank's sweep points `idx`; returns (elapsed max over ranks, last result, inputs)."""
        if args.config == 3:
            freq, grids, lam, eps_si = make_inputs_stack(idx, args.batch, args.grid, device)
        elif args.config == 5:
            freq, grids, lam, eps_si = None, make_inputs_topopt(device), np.array([532.]), np.array([TOPOPT_EPS])
        else:
            freq, grids, lam, eps_si = make_inputs(args.config, idx, args.grid, device)
        # lock-step chunk: the 128-point sweep of config 2 is one chunk; the 512 points per GPU of config 4 go in chunks of 256
        # (156 GB allocated / 208 GB reserved of the 288 GB; measured 31.6 vs 28.6 layer-solves/s with chunks of 128); the 4-layer
        # stack of config 3 at n = 3698 in chunks of 32 (144 GiB peak)
        # config 3: with the streaming cascade (one layer resident) 64 points of the 4-layer stack take 228 GB allocated / 247 GB reserved of
        # the 288 GB (3.96 vs 3.44 layer-solves/s in chunks of 32: 114 GB); a smaller device falls back to 32
        big = EMU or torch.cuda.get_device_properties(device).total_memory >= 280e9
        chunk = args.chunk if args.chunk > 0 else max(1, min(len(idx), {2: 128, 3: 64 if big else 32, 4: 1 << 30, 5: 1}[args.config]))
        if args.config == 4 and args.chunk <= 0 and not EMU:
            # config 4 goes through the public driver WITHOUT a chunk argument: the driver sizes it from the free HBM (the figure is reported)
            from torcwa_amd.sweep import auto_chunk
            chunk = auto_chunk(len(idx), order, 1, args.precision, device)
        # (round 3 forced the all-fp64 route here: the mixed route's extra n^2 buffer did not fit at chunk 64; the refinement now builds its
        # update matrix in place, so the library default applies -- to be confirmed by the config-3 run)
        out = None
        for w in range(warmup):
            try:
                out = run_step(freq, grids, order, engine, args, chunk)
            except (RuntimeError, torcwa_amd.TrxError) as e:
                # an untimed warm-up step may hit a device that is still releasing the memory of a previous process: free the
                # allocator cache, wait and try once more (the timed steps below are never retried)
                if w > 0:
                    raise
                print("bench: warm-up step failed (%s); retrying once" % str(e).splitlines()[0], file=sys.stderr, flush=True)
                sync()
                if not EMU:
                    torch.cuda.empty_cache()
                time.sleep(10.0)
                out = run_step(freq, grids, order, engine, args, chunk)
        barrier()
        if profile:
            engine.lib.prof_reset()
            engine.lib.prof_enable(0 if os.environ.get("TRX_BENCH_NOPROF") == "1" else 1)
            engine.phase_report()                                                       # drop the warm-up's brackets
            engine.profile_phases = os.environ.get("TRX_BENCH_NOPROF") != "1"
            if not EMU:
                mstat["timed0"] = torch.cuda.memory_stats(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run_step(freq, grids, order, engine, args, chunk)
        barrier()
        elapsed = time.perf_counter() - t0
        engine.lib.prof_enable(0)
        engine.profile_phases = False
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        return elapsed, out, (freq, grids, lam, eps_si, chunk)

    # ---- primary measurement ------------------------------------------------------------------------------------------
    if args.config in (2, 3, 5):
        total = args.batch * world if scaling == "weak" else args.batch
        idx = np.arange(rank * args.batch, (rank + 1) * args.batch) if scaling == "weak" else local_block(total)
    else:
        total = args.points
        idx = local_block(total)
        if scaling == "weak":
            total, idx = args.points * world, np.arange(rank * args.points, (rank + 1) * args.points)
    mstat = {}
    ms0 = torch.cuda.memory_stats(device) if not EMU else {}
    elapsed, out, (freq, grids, lam, eps_si, chunk) = measure(idx, args.steps, args.warmup, profile=True)
    ms1 = torch.cuda.memory_stats(device) if not EMU else {}
    n_fail = engine.failures()
    if n_fail:
        raise SystemExit(f"bench invalid: {n_fail} numerical failures (info != 0) inside the timed region")
    roof = roofline(engine, args, elapsed, args.steps, n, len(idx) * layers_per_point, chunk) if rank == 0 else None
    if args.host_profile and rank == 0:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run_step(freq, grids, order, engine, args, chunk)
        sync()
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
    full = gather_sweep(out, total, cyclic=cyclic and scaling != "weak") if world > 1 else out      # the one collective of the job (RCCL all_gather, KB-sized)
    value = total * layers_per_point * args.steps / elapsed

    # ---- strong-scaling leg (north_star: ">= 6x strong scaling of a wavelength sweep at 8 GPUs") ------------------------------
    strong = None
    if world > 1 and scaling == "weak" and args.config == 2 and not args.no_strong_leg:
        sidx = local_block(args.batch)
        el_s, out_s, _ = measure(sidx, args.steps, 1, profile=False)
        full_s = gather_sweep(out_s, args.batch)
        n_fail = engine.failures()
        strong = {"workload": "the SAME %d-lambda sweep split over %d GPUs (%d-%d points per GPU)" % (args.batch, world, args.batch // world, -(-args.batch // world)),
                  "value": args.batch * args.steps / el_s, "unit": "layer-solves/s", "ms_per_step": 1e3 * el_s / args.steps, "steps": args.steps,
                  "gathered_points": int(full_s.shape[0]), "numerical_failures": int(n_fail),
                  "speedup": "value / (value of the n_gpus = 1 run of this script: same sweep on one GPU)"}

    if rank == 0:
        mem = None
        if not EMU:
            mem = {"peak_reserved_GB": ms1.get("reserved_bytes.all.peak", 0) / 1e9, "peak_allocated_GB": ms1.get("allocated_bytes.all.peak", 0) / 1e9,
                   "device_mallocs_in_run": ms1.get("segment.all.allocated", 0) - ms0.get("segment.all.allocated", 0),
                   "device_mallocs_in_timed_region": ms1.get("segment.all.allocated", 0) - mstat["timed0"].get("segment.all.allocated", 0),
                   "device_frees_in_timed_region": ms1.get("segment.all.freed", 0) - mstat["timed0"].get("segment.all.freed", 0),
                   "alloc_retries": ms1.get("num_alloc_retries", 0)}
        wl = {2: "configs[1]: single patterned layer, order=[%d,%d] (n=%d), %dx%d grid, %d-lambda sweep, glass input half-space"
                 % (args.order, args.order, n, args.grid, args.grid, args.batch),
              3: "configs[2]: Example1-1 style stack of 4 patterned layers (rotated rectangles in SU-8), order=[%d,%d] (n=%d), %dx%d grid, %d-lambda sweep "
                 "(4 layer-solves + 3 dense layer-layer star products per point)" % (args.order, args.order, n, args.grid, args.grid, args.batch),
              4: "configs[3]: Example3-style (Wx,Wy,lambda) sweep, %d independent single-layer solves, order=[%d,%d] (n=%d), %dx%d grid, sharded by contiguous blocks"
                 % (total, args.order, args.order, n, args.grid, args.grid),
              5: "configs[4]: Example6-style topology-optimisation step, order=[%d,%d] (n=%d), 700x300 grid, complex128: forward + adjoint (stabilised eig "
                 "gradient) of one patterned layer = one layer-solve; N > 1: independent replicas" % (args.order, args.order, n)}[args.config]
        res = {
            "metric": "RCWA layer-solves/sec (complex64 I/O) at Fourier order [%d,%d]" % (args.order, args.order),
            "value": value, "unit": "layer-solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "c128" if args.config == 5 else (("c128 results (mixed-precision eigensolver: fp32 eigendecomposition refined to fp64 by two Newton steps -- fp64 residual and update GEMMs, "
                                                        "the correction solved with ONE fp32 LU of the fp32 start; everything else fp64 MFMA); complex64 I/O" if eig_is_mixed(args, n, chunk) else "c128 arithmetic (fp64 MFMA; complex64 I/O)")
                                                       if args.precision == "high" else "c64"),
            "data": "synthetic" if not EMU else "synthetic -- CPU kernel-logic EMULATOR, launcher plumbing test, not a measurement",
            "config": {"workload": wl, "points_total": int(total), "points_per_gpu": int(len(idx)), "layer_solves_per_point": layers_per_point,
                       "chunk": int(chunk), "streams": args.streams,
                       "eig_route": ("--eig-route %s: mixed = fp32 eigendecomposition + fp64 Newton refinement, matrices the refinement cannot certify redone in fp64 "
                                     "inside trx_eig; auto = mixed until a call of the sweep had to redo matrices, then fp64 for the rest of THAT sweep call "
                                     "(last eig call of the run redid %d of its matrices)" % (args.eig_route, int(getattr(engine, "last_eig_fallback", 0)))
                                     if eig_is_mixed(args, n, chunk) else "fp64 (Hessenberg, multi-shift QR, Schur vectors)")
                                    if args.precision == "high" else "fp32 (Hessenberg, multi-shift QR, Schur vectors)",
                       "sharding": ("cyclic (rank r: points r, r + N, ...)" if cyclic else "contiguous blocks") if world > 1 else None,
                       "precision": args.precision, "backend": ("gloo" if EMU else "nccl (RCCL)") if world > 1 else None},
            "txx00_sample": [float(full[0, 0].real), float(full[0, 0].imag)], "gathered_points": int(full.shape[0]),
            "numerical_failures": 0, "hbm": mem, "csrc_sha16": csrc_sha16(),
        }
        if args.config == 5 and roof is not None:
            roof["layer_solve"]["note"] = "t_measured covers forward AND adjoint; T_roof prices the forward solve only"
        if args.config == 5:
            res["metric"] = "RCWA layer-solves/sec, forward + adjoint (complex128) at Fourier order [%d,%d]" % (args.order, args.order)
            res["fom"], res["grad_norm"] = float(full[0, 0].real), _grad_norm[0]
        if strong is not None:
            res["strong_scaling"] = strong
        res["roofline"] = roof
        if EMU:
            # the CPU kernel-logic emulator is a launcher-plumbing test: it must never look like a measurement
            res["emulator_plumbing"] = {"units_per_s": res["value"], "ms_per_step": res["ms_per_step"]}
            res["value"], res["ms_per_step"], res["roofline"] = None, None, None
        if not args.no_cpu_baseline and world == 1 and args.config == 2:
            model, phys, logical = host_cpu()
            threads = args.cpu_threads if args.cpu_threads > 0 else phys
            npts = max(1, min(args.cpu_points, len(lam)))
            pick = sorted(set(int(round(i * (len(lam) - 1) / max(npts - 1, 1))) for i in range(npts)))
            secs, vals, dt_flush, dt128, v128 = cpu_baseline(order, args.grid, [lam[i] for i in pick], [eps_si[i] for i in pick], threads)
            pts = []
            for j, i in enumerate(pick):
                got = complex(full[i, 0])
                pts.append({"point": "lambda=%.1f nm, txx(0,0)" % lam[i], "gpu": [got.real, got.imag], "oracle_c128": [v128[j].real, v128[j].imag],
                            "rel_err_vs_c128_oracle": abs(got - v128[j]) / abs(v128[j]), "oracle_c64": [vals[j].real, vals[j].imag],
                            "rel_err_of_c64_oracle_vs_c128_oracle": abs(vals[j] - v128[j]) / abs(v128[j]), "oracle_c128_seconds": dt128[j]})
            res["parity_sample"] = dict(pts[0], all_points=pts, max_rel_err_vs_c128_oracle=max(p_["rel_err_vs_c128_oracle"] for p_ in pts),
                                        gate="<= 1e-5 (north_star): complex64-I/O result of the timed sweep against the complex128 oracle at every CPU-baseline point")
            res["cpu_baseline"] = {"value": len(secs) / sum(secs), "unit": "layer-solves/s", "cores": threads, "kind": "port",
                                   "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
                                   "seconds_per_layer_solve": secs, "flush_denormal_seconds_footnote": dt_flush,
                                   "sample": "%d layer-solves of the same workload (lambda = %s nm), complex64, denormals not flushed (as the reference "
                                             "runs), oracle/rcwa_oracle.py on torch-CPU with %d threads = physical cores; footnote: the first point "
                                             "again with torch.set_flush_denormal(True)" % (len(secs), ", ".join("%.1f" % lam[i] for i in pick), threads)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
