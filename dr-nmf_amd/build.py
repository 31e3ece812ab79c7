"""Build libdrnmf.so (hipcc, gfx950 only) in-tree next to this file.

    python dr-nmf_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdrnmf.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # scalar kernel arguments arrive preloaded in SGPRs (no kernarg load on the critical path)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(HERE), "include", "drnmf.h"), os.path.abspath(__file__)]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and
                os.path.getmtime(obj) > max(os.path.getmtime(p) for p in _deps()
                                            if not p.endswith(".hip") or p == src)):
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
