"""SURVEY.md section 5 (race detection / sanitizers): the host side of the C-ABI under AddressSanitizer + UBSan in the
CPU suite.  `build.py --host-asan` instruments the host code of every translation unit (device code is compiled as
usual); the library is then loaded into a python that has the ASan runtime preloaded and driven through its
no-device paths: symbol table, error plumbing, argument checks of EVERY entry point, the switch table, the parameter
fingerprint (threaded), and the launch helpers' device-pointer guard firing on a heap pointer."""
import importlib.util
import os
import subprocess
import sys

import pytest

from conftest import REPO, PKG_ROOT


def test_host_side_is_clean_under_asan_ubsan():
    spec = importlib.util.spec_from_file_location("lnasr_build", os.path.join(PKG_ROOT, "build.py"))
    bld = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bld)
    rt = bld.asan_runtime()
    if rt is None:
        pytest.skip("no shared ASan runtime in this ROCm install")
    lib = bld.build(verbose=False, host_asan=True)
    env = dict(os.environ, LNASR_LIB=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for script in ("_asan_host_paths.py",):
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", script)], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        assert "asan host paths ok" in r.stdout
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    # the sanitizer really is in the library that was exercised
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms
