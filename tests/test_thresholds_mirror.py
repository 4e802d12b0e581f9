"""CPU: tests/dispatch_thresholds.py mirrors the sources -- every threshold's snippet is present in the file it cites, the
numbers the table states are the numbers in the snippet, and the case list is large enough to mean something."""
import os
import re

from conftest import ROOT
import dispatch_thresholds as dt


def test_every_threshold_is_where_the_table_says():
    for name, axis, value, kinds, path, snippet in dt.THRESHOLDS:
        src = open(os.path.join(ROOT, path)).read()
        assert snippet in src, f"{name}: the dispatch line moved or changed in {path}; update tests/dispatch_thresholds.py"
        if axis in ("B", "n") and name != "panel row blocks of 64":
            nums = {int(x) for x in re.findall(r"\d+", snippet)}
            assert nums & {value, 2 * value}, (name, value, sorted(nums))      # the line quotes the number the table states (n2 = 2n)


def test_case_list_covers_every_kind_around_every_threshold():
    cs = dt.cases()
    assert len(cs) >= 150
    kinds = {k for k, _, _, _ in cs}
    assert kinds == {"gsm", "gsmf", "bam", "bamf", "potrf"}
    # each B / n / n2 threshold appears at value - 2 .. value + 2 for each of its kinds (even values only on the n2 axis)
    have = {(k, D, B) for k, D, B, _ in cs}
    for name, axis, value, tkinds, _p, _s in dt.THRESHOLDS:
        if axis == "D":
            continue
        for kind in tkinds:
            for off in (-2, 0, 2):
                v = value + off
                B = v // 2 if axis == "n2" else v
                D = dt.B_AXIS_DIM[kind]
                if kind in ("gsmf", "bamf") and 2 * B > D:
                    continue
                assert (kind, D, B) in have, (name, kind, D, B)
