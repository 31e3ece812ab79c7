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
FLAGS = ARCH + (["-DDRNMF_TIMELINE"] if os.environ.get("DRNMF_TIMELINE") else []) + os.environ.get("DRNMF_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # scalar kernel arguments arrive preloaded in SGPRs (no kernarg load on the critical path)
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


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


# ---- host-side sanitizer build (race detection / sanitizers) -------------------
# DRNMF_SANITIZE=1 python dr-nmf_amd/build.py, or build_sanitized(): the HOST half of every translation
# unit -- descriptor validation, workspace / parameter layouts, the graph cache, the C-ABI shims --
# compiled with AddressSanitizer + UndefinedBehaviorSanitizer (-Xarch_host: the device code is built as
# usual; GPU ASan is not available on this pool and is not wanted here), into
# dr-nmf_amd/build_asan/libdrnmf_asan.so.  Never loaded by the product: tests/test_sanitize.py links a
# plain-C driver against it and runs it on the CPU box.
ASAN_DIR = os.path.join(HERE, "build_asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libdrnmf_asan.so")
ASAN_FLAGS = ARCH + ["-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function",
                     "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
                     "-Xarch_host", "-fno-sanitize-recover=undefined",
                     "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def sanitizer_runtime_dir():
    """Directory of clang's shared ASan runtime (the sanitized library and its driver link -shared-libsan)."""
    pat = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "lib", "clang", "*", "lib",
                       "linux", "libclang_rt.asan-x86_64.so")
    hits = sorted(glob.glob(pat))
    return os.path.dirname(hits[-1]) if hits else None


def build_sanitized(verbose=False):
    os.makedirs(ASAN_DIR, exist_ok=True)
    h = hashlib.sha256(" ".join(ASAN_FLAGS).encode())
    for p in sorted(_deps()):
        with open(p, "rb") as f:
            h.update(f.read())
    want = h.hexdigest()
    stamp = ASAN_LIB + ".srchash"
    with open(ASAN_LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(ASAN_LIB) and os.path.exists(stamp):
                with open(stamp) as f:
                    if f.read().strip() == want:
                        return ASAN_LIB
            objs, procs = [], []
            for src in _sources():
                obj = os.path.join(ASAN_DIR, os.path.basename(src) + ".o")
                objs.append(obj)
                cmd = [HIPCC] + ASAN_FLAGS + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd, stderr=None if verbose else subprocess.DEVNULL)))
            for src, p in procs:
                if p.wait() != 0:
                    raise RuntimeError("hipcc (sanitized) failed on %s" % src)
            subprocess.check_call([HIPCC] + ARCH + ["-shared", "-fPIC", "-fsanitize=address,undefined",
                                                    "-shared-libsan", "-o", ASAN_LIB] + objs + ["-ldl"])
            with open(stamp, "w") as f:
                f.write(want + "\n")
            return ASAN_LIB
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


if __name__ == "__main__":
    if os.environ.get("DRNMF_SANITIZE"):
        print(build_sanitized(verbose=True))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
