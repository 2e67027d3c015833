"""CPU-only checks of host logic: C-ABI surface, loud failure without the extension, sweep sharding + gather (gloo, 2 ranks)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "trx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(trx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    """torcwa_amd/libtrx.so (gfx950 build) loads on CPU and exports every function include/trx.h declares."""
    from torcwa_amd.csrc import build
    from torcwa_amd import _lib
    path = build.build_gpu()
    dll = ctypes.CDLL(path)
    syms = _header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(dll, s), s
    assert sorted(_lib.exported_symbols()) == syms          # the ctypes binding covers exactly the header
    h = _lib.TrxLib(path)
    assert h.version() >= 100 and b"workspace" in h.strerror(-3)


def test_product_fails_loudly_without_gpu_or_extension(tmp_path):
    from torcwa_amd import _lib
    import torcwa_amd
    with pytest.raises(_lib.TrxError):
        _lib.TrxLib(str(tmp_path / "missing_libtrx.so"))
    if not torch.cuda.is_available():
        with pytest.raises(_lib.TrxError):
            torcwa_amd.Engine()                              # no CPU fallback
        with pytest.raises(_lib.TrxError):
            torcwa_amd.rcwa(freq=1 / 500., order=[1, 1], L=[300., 300.])


def test_shard_range_partitions():
    from torcwa_amd.sweep import shard_range
    for n in (0, 1, 7, 64, 128, 4096, 4099):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from torcwa_amd.sweep import shard_range, gather_sweep
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
M = 11                                   # uneven split: 6 + 5
lo, hi = shard_range(M, rank, world)
idx = torch.arange(lo, hi, dtype=torch.float64)
local = torch.complex(idx, -2 * idx)[:, None].repeat(1, 3)      # [m_r, 3] complex "S-parameters" = f(global index)
full = gather_sweep(local, M)
exp = torch.complex(torch.arange(M, dtype=torch.float64), -2 * torch.arange(M, dtype=torch.float64))[:, None].repeat(1, 3)
assert full.shape == (M, 3) and torch.equal(full, exp), (rank, full)
r = gather_sweep(idx[:, None], M)
assert torch.equal(r[:, 0], torch.arange(M, dtype=torch.float64))
# cyclic sharding (SURVEY.md 8(e)): rank r solves points r, r + world, ...; the gather restores sweep order
from torcwa_amd.sweep import shard_indices
mine = torch.as_tensor(shard_indices(M, rank, world, cyclic=True), dtype=torch.float64)
assert len(mine) == (6 if rank == 0 else 5) and float(mine[0]) == rank
fullc = gather_sweep(torch.complex(mine, -2 * mine)[:, None].repeat(1, 3), M, cyclic=True)
assert fullc.shape == (M, 3) and torch.equal(fullc, exp), (rank, fullc)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sweep_gather_two_ranks_gloo(tmp_path):
    """world_size-2 gloo run of the one collective of the sweep driver (final all_gather of per-point results)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script), ROOT]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok") >= 2


def test_shard_indices_cover_the_sweep():
    from torcwa_amd.sweep import shard_indices
    for n in (0, 1, 7, 4096, 4099):
        for w in (1, 3, 8):
            for cyc in (False, True):
                parts = [shard_indices(n, r, w, cyclic=cyc) for r in range(w)]
                assert sorted(np.concatenate(parts).tolist()) == list(range(n))
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_auto_chunk_fits_free_memory(monkeypatch):
    """solve_stack_sweep(chunk=None) sizes its lock-step chunk from the free HBM (15 / 18 n x n complex128 matrices per point, 10 % of the
    device kept free) and says so, with numbers, when not even one point fits."""
    from torcwa_amd import sweep
    dev = torch.device("cuda", 0)
    gb = 1e9
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda d=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda d=None: 0)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(280 * gb), int(288 * gb)))
    assert sweep.auto_chunk(128, [15, 15], 1, "high", dev) == 128                  # config 2: the whole sweep
    c4 = sweep.auto_chunk(4096, [15, 15], 1, "high", dev)                          # config 4's 4096 points: cut to what fits
    assert 200 <= c4 <= 296 and c4 % 8 == 0
    c3 = sweep.auto_chunk(64, [21, 21], 4, "high", dev)                            # config 3: [21,21], 4 layers
    per = 18 * 3698 ** 2 * 16
    assert 8 <= c3 <= 64 and c3 * per <= (280 - 28.8) * gb
    assert sweep.auto_chunk(4096, [15, 15], 1, "native", dev) >= 2 * c4 - 8        # fp32 arithmetic: half the bytes
    assert sweep.auto_chunk(4096, [15, 15], 1, "native", dev, dtype=torch.complex128) == c4          # a complex128 problem computes in complex128 whatever `precision` says
    assert sweep.auto_chunk(4096, [15, 15], 1, "high", dev, streams=2) <= c4 // 2                     # two chunks are resident at once on two streams
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d=None: (int(30 * gb), int(288 * gb)))
    with pytest.raises(RuntimeError, match="needs about"):
        sweep.auto_chunk(64, [21, 21], 4, "high", dev)
    assert sweep.auto_chunk(77, [15, 15], 1, "high", torch.device("cpu")) == 77


def test_homogeneity_rule_matches_the_reference():
    """rcwa.py:156-157: float / complex / 0-d tensor / 1-D tensor of length 1 are homogeneous; python ints raise; any other 1-D tensor is not
    homogeneous (batched extension: length B = one value per sweep point)."""
    from torcwa_amd.batched import BatchedRCWA
    obj = BatchedRCWA.__new__(BatchedRCWA)
    obj.B = 1
    assert obj._is_homogeneous(2.0) and obj._is_homogeneous(2j) and obj._is_homogeneous(torch.tensor(2.0)) and obj._is_homogeneous(torch.tensor([2.0]))
    assert not obj._is_homogeneous(torch.tensor([2.0, 3.0, 4.0])) and not obj._is_homogeneous(torch.ones(4, 4))
    with pytest.raises(AttributeError):
        obj._is_homogeneous(2)
    obj.B = 3
    assert obj._is_homogeneous(torch.tensor([2.0, 3.0, 4.0])) and not obj._is_homogeneous(torch.tensor([2.0, 3.0]))


def test_eig_route_policy_is_scoped_to_the_solver_or_sweep_call():
    """BatchedRCWA._eig_call: "auto" asks the library's automatic (mixed-precision) route until a call reports matrices redone in fp64, then the
    all-fp64 route -- for eigenproblems of that size handled by THIS solver object / sweep call only (the hint is a dict created per call)."""
    from torcwa_amd.batched import BatchedRCWA

    class FakeEngine:
        def __init__(self):
            self.calls, self.last_eig_fallback, self.next_fallback = [], 0, 0

        def eig(self, A, destroy=False, refine_steps=0, route=0):
            self.calls.append(route)
            self.last_eig_fallback = self.next_fallback if route != 1 else 0
            return None, None

        def eig_fallback_of_last_call(self):          # the per-thread count Engine.eig leaves behind (not the shared attribute)
            return self.last_eig_fallback

    def solver(eng, route, hint):
        o = BatchedRCWA.__new__(BatchedRCWA)
        o.engine, o.eig_route, o._route_hint = eng, route, hint
        return o

    eng = FakeEngine()
    A = torch.zeros(2, 6, 6)
    hint = {}
    a, b = solver(eng, "auto", hint), solver(eng, "auto", hint)        # two chunks of one sweep call share the hint
    a._eig_call(A, 2)
    eng.next_fallback = 1
    a._eig_call(A, 2)                   # this one had to redo a matrix
    b._eig_call(A, 2)                   # -> the other chunk goes to fp64 directly
    a._eig_call(torch.zeros(2, 8, 8), 2)        # another size: no hint
    c = solver(eng, "auto", {})         # a new sweep call / solver starts fresh
    c._eig_call(A, 2)
    solver(eng, "fp64", {})._eig_call(A, 2)
    solver(eng, "mixed", {})._eig_call(A, 2)
    assert eng.calls == [0, 0, 1, 0, 0, 1, 3]


def test_blockdiag2_algebra():
    from torcwa_amd.batched import BlockDiag2
    g = torch.Generator().manual_seed(0)
    mk = lambda: BlockDiag2(*[torch.randn(2, 5, generator=g, dtype=torch.complex128) for _ in range(4)])
    A, B = mk(), mk()
    assert torch.allclose((A @ B).dense(), A.dense() @ B.dense())
    assert torch.allclose(A.inv().dense(), torch.linalg.inv(A.dense()))
    assert torch.allclose((A + B).dense(), A.dense() + B.dense())


def test_bench_profile_bookkeeping(tmp_path, monkeypatch):
    """bench.py holds its live roofline figures against the committed profiles per TAG: a gemm<N,N> call is the large-tile kernel plus the
    narrow-tile launches of its peeled remainders, so work and kernel time are compared per step; a profile is accepted for a tag while the
    source files of THAT tag's kernels are unchanged, and refused otherwise."""
    import json
    import types
    import bench
    hashes = bench.csrc_file_hashes()
    assert "gemm_big.hip" in hashes and "common.hpp" in hashes
    prof = {"csrc_sha16": "x", "src_sha16": dict(hashes), "batch": 128, "steps_traced": 4,
            "kernels": {"gemm_big_kernel<0, 0, 4, 2, 2, 3>": {"launches": 40, "avg_us": 1000.0, "total_ms": 40.0},
                        "gemm_mfma_kernel<double, 0, 0, 2, 4, 16>": {"launches": 40, "avg_us": 50.0, "total_ms": 2.0},
                        "gemm_mfma_kernel<float, 0, 0, 4, 4, 16>": {"launches": 8, "avg_us": 10.0, "total_ms": 0.08},
                        "hess_gemv_kernel<float, 2>": {"launches": 400, "avg_us": 100.0, "total_ms": 40.0}}}
    assert bench.profile_valid_for(prof, "gemm<N,N>") and bench.profile_valid_for(prof, "hess_gemv_kernel")
    moved = dict(prof, src_sha16=dict(hashes, **{"eig_hess.hip": "0" * 16}))
    assert bench.profile_valid_for(moved, "gemm<N,N>") and not bench.profile_valid_for(moved, "hess_gemv_kernel")
    moved = dict(prof, src_sha16=dict(hashes, **{"gemm_big.hip": "0" * 16}))
    assert not bench.profile_valid_for(moved, "gemm<N,N>") and bench.profile_valid_for(moved, "hess_gemv_kernel")
    assert not bench.profile_valid_for({"csrc_sha16": "other"}, "gemm<N,N>")            # legacy profile without per-file hashes
    # per-step accounting: 10 calls per step of 1e12 flops each against (40 + 2) ms / 4 steps of kernel time
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / ("%s_kernel_profile.json" % bench.PROFILE_TAG)).write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "csrc_file_hashes", lambda: hashes)
    args = types.SimpleNamespace(config=2, batch=128, precision="high")
    k = {"kernel": "gemm<N,N>", "bound": "mfma", "algorithmic_flops_per_launch": 1e12, "launches": 20}
    out = bench.profile_fracs(k, args, 78.6, steps=2)
    assert abs(out["frac_rocprof"] - (10 * 1e12 / 10.5e-3 / 1e12) / 78.6) < 1e-9 and out["frac_alone"] is None


def test_gemm_big_vgpr_form_isa():
    """gemm_big.hip: the 8-wave layout must get the VGPR form of the fp64 MFMA (accumulators as ordinary registers, no v_accvgpr_* moves
    around the K slabs -- worth ~15 %): cross-compile the file (hipcc, gfx950, no GPU needed) and check the ISA."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "torcwa_amd", "csrc", "gemm_big.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gemm_big.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        text = open(out).read()
    kernels = re.findall(r"^(_ZN3trx[^:\s]*gemm_big_kernel[^:\s]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 9                     # 3 x 3 operand forms of the one tile layout
    for name, body in kernels:
        assert "v_accvgpr" not in body, name
        assert "v_mfma_f64_16x16x4_f64 v[" in body
