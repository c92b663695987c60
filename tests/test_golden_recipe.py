"""The fixture recipe reproduces the fixtures: tools/check_golden.py re-runs the reference itself on
the inputs stored in tests/golden and compares every stored output byte for byte.  Build container only - the reference
does not travel to the GPU box, where this test is skipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference is only present in the build container")
def test_committed_fixtures_are_the_references_outputs_on_the_stored_inputs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_golden.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
