"""Host-side sanitizer build of libdrnmf (race detection / sanitizers).

    python tools/sanitize/build_sanitized.py

The HOST half of every translation unit -- descriptor validation, workspace / parameter layouts, the
graph cache, the C-ABI shims -- compiled with AddressSanitizer + UndefinedBehaviorSanitizer (-Xarch_host:
the device code is built as usual; GPU ASan is not available on this pool and is not wanted here), into
dr-nmf_amd/build_asan/libdrnmf_asan.so.  Never loaded by the product and never run on the GPU box (this
directory and the build output are listed in .gpurunignore): tests/test_sanitize.py links a plain-C
driver against it and runs it on the CPU box."""
import fcntl
import glob
import hashlib
import importlib.util
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_spec = importlib.util.spec_from_file_location("drnmf_build", os.path.join(ROOT, "dr-nmf_amd", "build.py"))
_bm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bm)
HERE, HIPCC, ARCH, _sources, _deps = _bm.HERE, _bm.HIPCC, _bm.ARCH, _bm._sources, _bm._deps

ASAN_DIR = os.path.join(HERE, "build_asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libdrnmf_asan.so")
ASAN_FLAGS = ARCH + ["-O1", "-g", "-std=c++17", "-fPIC", "-Wno-unused-function",
                     "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
                     "-Xarch_host", "-fno-sanitize-recover=undefined",
                     "-mllvm", "-amdgpu-kernarg-preload-count=16",
                     "-mllvm", "-pragma-unroll-threshold=262144"]


DRIVER_FLAGS = ["-std=c99", "-Wall", "-Werror", "-g", "-pthread", "-fsanitize=address,undefined",
                "-fno-sanitize-recover=undefined", "-shared-libsan"]
RUN_ENV = {"ASAN_OPTIONS": "halt_on_error=1:detect_leaks=0:abort_on_error=0",
           "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


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
    print(build_sanitized(verbose=True))
