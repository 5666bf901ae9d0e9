"""ASan + UBSan over the host-side product code (FASTA ingest, special-region module, text generator, the C host
program's argument handling) -- CPU build only (tests/sanitize/Makefile; the full log incl. the oracle leg is
profiles/r03_sanitizers.txt).  The GPU pool has no sanitizer support, so this runs in the CPU suite alone."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = os.path.join(ROOT, "tests", "sanitize")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_ubsan():
    r = subprocess.run(["make", "-C", SAN, "host_sanitize", "cli_args_asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([os.path.join(SAN, "host_sanitize")], cwd=SAN, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "0 failures" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr
    # ThreadSanitizer over the threaded host code (it found the one race this code had: run ends looked for while other
    # runs were being sorted, special_host.cpp)
    r = subprocess.run(["make", "-C", SAN, "host_sanitize_tsan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([os.path.join(SAN, "host_sanitize_tsan")], cwd=SAN, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failures" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    r = subprocess.run(["sh", "cli_args.sh"], cwd=SAN, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "cli_args: ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
