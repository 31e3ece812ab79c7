"""Build libdrnmf.so (hipcc, gfx950 only) in-tree next to this file.

    python dr-nmf_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the
repo snapshot.
"""
import fcntl
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdrnmf.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# gfx950 only.  xnack-: what MI355X runs by default (XNACK off); the compiler then need not keep the
# address registers of in-flight loads intact for a replay: +0.6 % on the recurrent cell, +1.4 % on
# the GEMMs against the generic target.  A process with XNACK enabled (HSA_XNACK=1) cannot load it:
# set DRNMF_OFFLOAD_ARCH=gfx950 and rebuild.
ARCH = ["--offload-arch=" + os.environ.get("DRNMF_OFFLOAD_ARCH", "gfx950:xnack-")]
# DRNMF_TIMELINE=1: measurement build with s_memtime stamps in the cell kernels (tools/timeline.py)
# DRNMF_MEASURE=1: measurement build -- the ablation arguments / environment aids of DESIGN.md section 8 compiled in
FLAGS = ARCH + (["-DDRNMF_TIMELINE"] if os.environ.get("DRNMF_TIMELINE") else []) + \
    (["-DDRNMF_MEASURE"] if os.environ.get("DRNMF_MEASURE") else []) + os.environ.get("DRNMF_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # scalar kernel arguments arrive preloaded in SGPRs (no kernarg load on the critical path)
         "-mllvm", "-amdgpu-kernarg-preload-count=16",
         # `#pragma unroll` means it: with the default threshold the optimiser refused to unroll the output-tile
         # loops of gemm_nt_kernel under the powf epilogues of the general beta-divergence (24 "loop not
         # unrolled" warnings); acc[a][b] was then indexed dynamically = 832 bytes of scratch per lane
         "-mllvm", "-pragma-unroll-threshold=262144"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        [os.path.join(os.path.dirname(HERE), "include", "drnmf.h"), os.path.abspath(__file__)]


STAMP = LIB + ".srchash"


def _src_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy to the
    GPU box in any particular order)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(_deps()):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _src_hash()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    # one builder at a time: `bench.py --gpus N` starts N processes that all call build()
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():     # another process built it meanwhile
                return LIB
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # per-object stamp: this source + every header + the flags
        hh = hashlib.sha256(" ".join(FLAGS).encode())
        for p in sorted(q for q in _deps() if not q.endswith(".hip") or q == src):
            with open(p, "rb") as f:
                hh.update(f.read())
        want = hh.hexdigest()
        stamp = obj + ".srchash"
        if not force and os.path.exists(obj) and os.path.exists(stamp):
            with open(stamp) as f:
                if f.read().strip() == want:
                    continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, stamp, want, subprocess.Popen(cmd)))
    for src, stamp, want, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [HIPCC] + ARCH + ["-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)            # a process that already mapped the old file keeps it
    with open(STAMP, "w") as f:
        f.write(_src_hash() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
