"""CPU-side AddressSanitizer + UndefinedBehaviorSanitizer run of the HOST half of libdrnmf (SURVEY.md
section 5 "race detection / sanitizers"; never on the GPU box -- GPU ASan is not available there): the
library's host code is rebuilt with -fsanitize=address,undefined (dr-nmf_amd/build.py: build_sanitized;
cached by source hash, ~2 min the first time) and tests/c_abi/sanitize_host.c drives every size query,
descriptor validator, layout rule and argument-check path of the C ABI through it."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location("drnmf_build", os.path.join(ROOT, "dr-nmf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_half_under_asan_and_ubsan(tmp_path):
    bm = _build_module()
    if not os.path.exists(bm.HIPCC):
        pytest.skip("no hipcc")
    rt = bm.sanitizer_runtime_dir()
    if rt is None:
        pytest.skip("clang's shared ASan runtime is not installed")
    lib = bm.build_sanitized()
    clang = os.path.join(os.path.dirname(os.path.dirname(bm.HIPCC)), "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = shutil.which("clang")
    assert clang, "no clang to build the driver with"
    exe = str(tmp_path / "sanitize_host")
    subprocess.run([clang, "-std=c99", "-Wall", "-Werror", "-g", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=undefined", "-shared-libsan",
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "sanitize_host.c"),
                    "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + rt], check=True)
    env = dict(os.environ)
    # leaks: the HIP runtime itself keeps allocations alive at exit; our own are checked by the explicit
    # create / destroy pairs under ASan's use-after-free / overflow instrumentation
    env["ASAN_OPTIONS"] = "halt_on_error=1:detect_leaks=0:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    for k in [k for k in env if k.startswith("DRNMF_")]:
        del env[k]
    r = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    report = r.stdout[-2000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, report
    assert r.returncode == 0, report
    assert "0 failed" in r.stdout, report
