#!/usr/bin/env python3
"""Build the CPU kernel-logic emulator of libtrx (TEST INFRASTRUCTURE): the unmodified torcwa_amd/csrc/*.hip sources compiled
with clang++ against tests/hipemu/hip/hip_runtime.h (fibers) -> tests/hipemu/_build/libtrx_emu.so.

    python tests/hipemu/build_emu.py [--force]
"""
import os
import sys

EMU = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(EMU))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402

from torcwa_amd.csrc.build import _deps, _run, _sources, _stamp  # noqa: E402  (source list and stamping shared with the GPU build)

CLANGXX = os.environ.get("TRX_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")


def build_emu(verbose=False, force=False):
    emu = EMU
    bdir = os.path.join(emu, "_build")
    os.makedirs(bdir, exist_ok=True)
    out = os.path.join(bdir, "libtrx_emu.so")
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-I", emu, "-x", "c++", "-Wno-unused-result",
             "-Wno-unused-value", "-fno-strict-aliasing"]
    deps = _deps() + [os.path.join(emu, "hip", "hip_runtime.h")]
    objs, jobs = [], []
    for src in _sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        st = _stamp([src] + deps, " ".join(flags))
        stf = obj + ".stamp"
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(stf) or open(stf).read() != st:
            jobs.append((src, obj, stf, st))

    def comp(j):
        src, obj, stf, st = j
        _run([CLANGXX] + flags + ["-c", src, "-o", obj])
        open(stf, "w").write(st)
        if verbose:
            print("compiled(emu)", os.path.basename(src))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(comp, jobs))
    sw = os.path.join(bdir, "hipemu_switch.o")
    if not os.path.exists(sw):
        _run([CLANGXX, "-c", os.path.join(emu, "hipemu_switch.S"), "-o", sw])
    if jobs or not os.path.exists(out):
        _run([CLANGXX, "-shared", "-fPIC", "-o", out] + objs + [sw])
    return out


if __name__ == "__main__":
    print(build_emu(verbose=True, force="--force" in sys.argv))
