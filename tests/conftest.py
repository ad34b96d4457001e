import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The tests need the in-tree library (and the oracle's C build).  They normally travel prebuilt with the snapshot
    (`__graft_entry__.build()`); on a checkout without them, build once here.  The *product* never builds implicitly: it
    raises ``KernelLoadError`` when the library is missing (tests/test_host_cpu.py::test_missing_library_is_a_load_error)."""
    try:
        from brainevent_amd import _lib
        if _lib.needs_build():
            _lib.build()
        from oracle import oracle_c
        oracle_c.build()
    except Exception as e:      # surfaced by the tests that need the library
        print(f"[conftest] could not build the native libraries: {e!r}")


@pytest.fixture(scope='session')
def be():
    import brainevent_amd
    return brainevent_amd


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle_np
    return oracle_np
