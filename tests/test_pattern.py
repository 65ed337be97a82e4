"""The rBRIEF pattern table: both committed copies are identical and equal the reference's (when it is present)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    txt = open(path).read().split("*/", 1)[1]
    return [int(t) for t in re.findall(r"-?\d+", txt)]


def test_copies_identical_and_sane():
    a = _load(os.path.join(ROOT, "oracle", "orb_pattern.inc"))
    b = _load(os.path.join(ROOT, "orb_slam3_modified_amd", "csrc", "orb_pattern.inc"))
    assert a == b and len(a) == 1024 and min(a) == -13 and max(a) == 12
    r = max(np.hypot(a[i], a[i + 1]) for i in range(0, 1024, 2))
    assert 18.0 < r < 18.5   # rotated taps reach +-18: the describe kernel's 37x37 blurred window
    assert a[:8] == [8, -3, 9, 5, 4, 2, 7, -12]


def test_equals_reference_table():
    ref = "/root/reference/src/ORBextractor.cc"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference checkout not present (GPU box)")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_pattern
    assert gen_pattern.parse(ref) == _load(os.path.join(ROOT, "oracle", "orb_pattern.inc"))
