import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    arrays = np.load(os.path.join(GOLDEN_DIR, "pairs_v1.npz"), allow_pickle=False)
    with open(os.path.join(GOLDEN_DIR, "pairs_v1.json")) as f:
        meta = json.load(f)
    return arrays, meta


@pytest.fixture(scope="session")
def anchors():
    """tests/golden/pairs_v1_anchors.json: per fixture the fp64-blur score (an anchor that shares
    no rounding sequence with the HIP kernel) and the published-recursion scores + recorded gaps."""
    with open(os.path.join(GOLDEN_DIR, "pairs_v1_anchors.json")) as f:
        return {p["name"]: p for p in json.load(f)["pairs"]}


@pytest.fixture(scope="session")
def oracle():
    from oracle import ssimu2_oracle
    ssimu2_oracle.build()
    return ssimu2_oracle


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the gfx950 library; fails loudly when hipcc is missing."""
    from oavif_amd import build
    build.build()
    from oavif_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def scorer(hip_lib):
    """GPU scorer context (only request it from @pytest.mark.gpu tests)."""
    from oavif_amd import Ssimu2
    s = Ssimu2(0)
    yield s
    s.close()


@pytest.fixture(scope="session")
def iscorer(hip_lib):
    """Context of the INSTRUMENTED build (liboavif_hip_instr.so: include/ssimu2_hip_internal.h);
    only for tests that need stage timing, plane downloads or the experiment knobs."""
    from oavif_amd import Ssimu2
    s = Ssimu2(0, instrumented=True)
    yield s
    s.close()
