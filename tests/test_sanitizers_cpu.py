"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU builds (GPU sanitizers are not available on this pool): the C oracle
(oracle/vsp_ref.c) through its whole test file, and the kernel arithmetic headers (csrc/field.h, curve.h: the 32-bit-limb device types and
the 64-bit-limb host types of the library's finishing code) through theirs.  Each runs as a child pytest with the sanitized build
preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _runtime(name):
    p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip(name + " not installed")
    return p


def _child(test_file, env_extra):
    env = dict(os.environ, LD_PRELOAD=_runtime("libasan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98", **env_extra)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", test_file), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    assert " passed" in p.stdout


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    so = str(tmp_path / "libvsp_ref_san.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "vsp_ref.c")])
    _child("test_oracle.py", {"VSP_REF_SO": so})


def test_kernel_arithmetic_headers_under_address_and_undefined_behaviour_sanitizers():
    _runtime("libubsan.so")
    _child("test_device_math_cpu.py", {"VSP_MATHCHK_FLAGS": "-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined"})
