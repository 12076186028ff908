"""Sanitizer builds of the host runtime (tools/sanitize_host.py): the TCP store under
ThreadSanitizer, and store + bucket planner + engine layout planner under AddressSanitizer and
UBSan (host pass of hipcc only; GPU sanitizers are not available and not used)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("which", ["tsan", "asan"])
def test_host_runtime_is_clean_under_sanitizers(which):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_host.py"), which],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "clean" in r.stdout and "store ok" in r.stdout
