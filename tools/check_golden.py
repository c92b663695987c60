#!/usr/bin/env python3
"""Are the committed fixtures still the reference's answers?  Build container only (needs /root/reference).

Runs tools/gen_golden.py into a scratch directory - the reference's own modules on the INPUTS the committed fixtures hold -
and compares every file it writes with tests/golden byte for byte.  Nothing under tests/golden is touched.  Exit 0: every
stored output is what the reference produces today on the stored input (the recipe reproduces the fixtures); exit 1: the
files that differ are listed.  Exit 77 (a skip) when the reference is not present (the GPU box)."""
import filecmp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
# inputs (never rewritten by the default recipe) and hand-derived known answers (tests/test_ssw_ties.py owns those)
NOT_OUTPUTS = {"c1_reads.fa.gz", "c1_whitelist.npy", "ssw_tie_kats.json"}


def main():
    if not os.path.isdir("/root/reference"):
        print("check_golden: /root/reference is absent - nothing to compare against")
        return 77
    with tempfile.TemporaryDirectory(prefix="golden_check_") as tmp:
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_golden.py"), "--out", tmp], check=True, stdout=subprocess.DEVNULL)
        made = sorted(os.listdir(tmp))
        want = sorted(f for f in os.listdir(GOLDEN) if f not in NOT_OUTPUTS)
        bad = [f for f in want if f not in made or not filecmp.cmp(os.path.join(tmp, f), os.path.join(GOLDEN, f), shallow=False)]
        extra = [f for f in made if f not in want]
    if bad or extra:
        print("check_golden: FAILED - differ or missing: %s; written but not committed: %s" % (bad, extra))
        return 1
    print("check_golden: %d fixture files are the reference's outputs on the stored inputs" % len(want))
    return 0


if __name__ == "__main__":
    sys.exit(main())
