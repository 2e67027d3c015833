#!/usr/bin/env python3
"""Build libtrx.so (gfx950) in-tree with hipcc, or the CPU kernel-logic emulator build for tests.

    python torcwa_amd/csrc/build.py            # -> torcwa_amd/libtrx.so          (hipcc --offload-arch=gfx950)
    python torcwa_amd/csrc/build.py --emu      # -> tests/hipemu/_build/libtrx_emu.so (clang++, fibers; tests only)
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(CSRC)
ROOT = os.path.dirname(PKG)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = os.environ.get("TRX_CLANGXX", "/opt/rocm/lib/llvm/bin/clang++")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(ROOT, "include", "trx.h")])


def _stamp(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_gpu(verbose=False, force=False):
    out = os.path.join(PKG, "libtrx.so")
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
    deps = _deps()
    objs, jobs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        st = _stamp([src] + deps, " ".join(flags))
        stf = obj + ".stamp"
        objs.append(obj)
        if force or not os.path.exists(obj) or not os.path.exists(stf) or open(stf).read() != st:
            jobs.append((src, obj, stf, st))

    def comp(j):
        src, obj, stf, st = j
        _run([HIPCC] + flags + ["-c", src, "-o", obj])
        open(stf, "w").write(st)
        if verbose:
            print("compiled", os.path.basename(src))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(comp, jobs))
    if jobs or not os.path.exists(out):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_emu(verbose=False, force=False):
    emu = os.path.join(ROOT, "tests", "hipemu")
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
    if "--emu" in sys.argv:
        print(build_emu(verbose=True, force="--force" in sys.argv))
    else:
        print(build_gpu(verbose=True, force="--force" in sys.argv))
