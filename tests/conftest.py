"""Shared fixtures.  `-m "not gpu"` = oracle vs known answers / golden fixtures, host logic, ABI symbol check;
`-m gpu` = HIP path vs the oracle through the C-ABI (bit-exact for integer work, 1e-4 relative for BA)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    o = graft.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def mvo():
    """The product package (ctypes over libmvo_hip.so)."""
    m = graft.load_package()
    if not os.path.exists(m.LIB_PATH):
        graft.build()
    return m


@pytest.fixture(scope="session")
def ctx(mvo):
    """A GPU context; fails loudly (no skip, no fallback) when the HIP path is unavailable."""
    c = mvo.Context(0)
    yield c
    c.close()


GOLDEN = os.path.join(ROOT, "tests", "golden")


def assert_struct_equal(a, b, what):
    assert len(a) == len(b), "%s: %d vs %d elements" % (what, len(a), len(b))
    if a.tobytes() != b.tobytes():
        for name in a.dtype.names:
            bad = np.nonzero(a[name] != b[name])[0]
            if len(bad):
                i = bad[0]
                raise AssertionError("%s: field %s differs at %d elements, first at %d: %r vs %r"
                                     % (what, name, len(bad), i, a[i], b[i]))
        raise AssertionError(what + ": byte mismatch (NaN payload?)")
