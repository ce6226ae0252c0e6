"""A time-boxed slice of every parity sweep of tools/gpu_fuzz_*.py inside the GPU suite: random sizes / parameters, HIP path
vs the oracle (bit-identical integer results, 1e-5 poses), so that a mismatch on an input no hand-written case covers fails
`pytest -m gpu` and not just a side script.  Each sweep runs its seeded case list for half a minute."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seconds,at_least", [("extractor", 400, 30, 6), ("matcher", 100, 30, 2), ("rest", 100, 30, 3),
                                                         ("more", 100, 30, 3), ("keyframes", 100, 30, 3)])
def test_fuzz_slice(gpu, tool, cases, seconds, at_least):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", f"gpu_fuzz_{tool}.py"), str(cases), str(seconds)], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"cases (\d+) of \d+ .*MISMATCHES(?: / ERRORS)? (\d+)", last)
    assert m, r.stdout[-2000:]
    assert int(m.group(2)) == 0, r.stdout[-4000:]
    assert int(m.group(1)) >= at_least, last
