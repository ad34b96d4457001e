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


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """BE_POISON_ALLOC=1 (robustness runs on the GPU box): before every gpu test, what the caching allocator will hand out next
    is filled with 0xFF bytes (NaN as a float, -1 as an integer, all spikes set as a mask), so a kernel or a host path that
    relies on `torch.empty` memory being zero — true in a fresh process, false in a long-running one — fails here instead of
    once in a while.  Large pool: one block of BE_POISON_GIB GiB (default 16; tests that allocate more get fresh pages beyond
    it), small pool: 512 blocks of 1 MiB."""
    if os.environ.get('BE_POISON_ALLOC') != '1' or request.node.get_closest_marker('gpu') is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gib = int(os.environ.get('BE_POISON_GIB', '16'))
        free = torch.cuda.mem_get_info()[0]
        big = torch.empty(min(gib << 30, int(free * 0.9)), dtype=torch.uint8, device='cuda').fill_(0xFF)
        small = [torch.empty(1 << 20, dtype=torch.uint8, device='cuda').fill_(0xFF) for _ in range(512)]
        tiny = [torch.empty(4096, dtype=torch.uint8, device='cuda').fill_(0xFF) for _ in range(512)]
        torch.cuda.synchronize()
        del big, small, tiny          # back to the caching allocator, contents intact
    yield
