"""CPU-side sanitizer run of the HOST half of libdrnmf (SURVEY.md section 5 "race detection / sanitizers";
never on the GPU box -- device-side sanitizers are not available there): the library's host code is
rebuilt with the address + undefined-behaviour sanitizers (tools/sanitize/build_sanitized.py; cached by
source hash, ~2 min the first time) and tests/c_abi/sanitize_host.c drives every size query, descriptor
validator, layout rule and argument-check path of the C ABI through it."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "sanitize", "build_sanitized.py")


def _tool():
    spec = importlib.util.spec_from_file_location("drnmf_build_sanitized", TOOL)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_host_half_under_the_sanitizers(tmp_path):
    if not os.path.exists(TOOL):
        pytest.skip("tools/sanitize is not shipped to this box (CPU-box test)")
    bm = _tool()
    if not os.path.exists(bm.HIPCC):
        pytest.skip("no hipcc")
    rt = bm.sanitizer_runtime_dir()
    if rt is None:
        pytest.skip("clang's shared sanitizer runtime is not installed")
    lib = bm.build_sanitized()
    clang = os.path.join(os.path.dirname(os.path.dirname(bm.HIPCC)), "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = shutil.which("clang")
    assert clang, "no clang to build the driver with"
    exe = str(tmp_path / "sanitize_host")
    subprocess.run([clang] + bm.DRIVER_FLAGS +
                   ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "sanitize_host.c"),
                    "-o", exe, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + rt], check=True)
    env = dict(os.environ)
    # leaks: the HIP runtime itself keeps allocations alive at exit; our own are covered by the explicit
    # create / destroy pairs under the use-after-free / overflow instrumentation
    env.update(bm.RUN_ENV)
    for k in [k for k in env if k.startswith("DRNMF_")]:
        del env[k]
    r = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    report = r.stdout[-2000:] + r.stderr[-6000:]
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, report
    assert r.returncode == 0, report
    assert "0 failed" in r.stdout, report
