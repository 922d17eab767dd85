"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on
the CPU side only; GPU ASan is not available on the pool).  The oracle's own test files run once more
in a child process against `make -C oracle asan` (oracle/_build/libmirt_oracle_asan.so); any
out-of-bounds access, use-after-free or undefined behaviour in the restatement aborts that run."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _libasan() -> str:
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else ""


def test_oracle_suite_is_clean_under_asan_ubsan():
    libasan = _libasan()
    if not libasan:
        pytest.skip("gcc's libasan.so not found")
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], check=True, capture_output=True)
    env = dict(os.environ, MIRT_ORACLE_VARIANT="asan", LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    files = ["tests/test_oracle_kats.py", "tests/test_oracle_pt.py", "tests/test_golden.py", "tests/test_math_spec.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", *files],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, f"oracle tests failed under ASan/UBSan:\n{tail}"
    assert "passed" in r.stdout and "runtime error" not in (r.stdout + r.stderr), tail
