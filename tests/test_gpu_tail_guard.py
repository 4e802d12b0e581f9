"""Out-of-bounds READS that no value check can see (round 6).  The whole GPU suite in one process died with "Memory access fault
by GPU" in a D = 16 fit: for D < 32 the waves of k_panel_fast beyond row D re-read rows 0 .. 31 of the D-row right operand -- values
they never use, past the end of the caller's array.  Inside an allocator block that is invisible; when the array ends its mapping it
is a fault.  The bug was a round old and every per-file run was green.  Here every array of every call sits at the END of its own
2 MiB allocation, freshly mapped for every repetition (tests/tail_guard_worker.py), in a child process: a fault kills the child
and fails the test.  Checked against the library before the fix: all seven shapes with D < 32 fail there, none here."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D,B", [(2, 1), (4, 2), (6, 3), (10, 5), (16, 8), (18, 2), (30, 8), (34, 4), (62, 16), (66, 40),
                                 (100, 17), (130, 64), (200, 130), (258, 20), (64, 200), (1000, 32)])
def test_arrays_at_the_end_of_their_mapping(D, B):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tail_guard_worker.py"), str(D), str(B)],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "tail guard ok" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
