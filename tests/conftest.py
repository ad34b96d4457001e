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


_POISON_WORDS = {'ff': -1, 'one': 1, 'f1': 0x3f800000, 'idx': 0x00030003}


def _install_poisoned_empty(pattern: str) -> None:
    """BE_POISON_ALLOC=ff|one|f1|idx: every device tensor that `torch.empty` / `empty_like` / `new_empty` hands out — in the
    product and in the tests, whatever its size (the mirror test allocates 170 GB) — is filled right away: 0xFF bytes (NaN,
    -1, every spike set), or PLAUSIBLE values that do not end in loud NaN / overflow paths: every 32-bit word 1 (a counter
    that looks armed, a directory that says "one entry", a denormal float), 1.0f, or 0x00030003 (a pair of valid uint16
    columns / a small int32 index).  A kernel or host path that reads such memory before writing it then gives a wrong
    number in the test that uses it, every time, instead of once in hundreds of runs on recycled pages."""
    import torch
    word = _POISON_WORDS[pattern]

    def fill(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() and t.is_contiguous():
            b = t.view(torch.uint8).reshape(-1)
            n4 = b.numel() // 4 * 4
            if n4 and b.data_ptr() % 4 == 0:
                b[:n4].view(torch.int32).fill_(word - (1 << 32) if word >= (1 << 31) else word)
            if b.numel() > n4:
                b[n4:].fill_(word & 0xff)
        return t

    for mod, name in ((torch, 'empty'), (torch, 'empty_like')):
        orig = getattr(mod, name)

        def wrapped(*a, __orig=orig, **k):
            return fill(__orig(*a, **k))
        setattr(mod, name, wrapped)
    orig_new = torch.Tensor.new_empty
    torch.Tensor.new_empty = lambda self, *a, **k: fill(orig_new(self, *a, **k))


if os.environ.get('BE_POISON_ALLOC') in _POISON_WORDS:
    _install_poisoned_empty(os.environ['BE_POISON_ALLOC'])


@pytest.fixture(autouse=True)
def _binned_conservation(request):
    """Every gpu test that builds a binned workspace checks, when it ends, that the workspace's conservation counters agree
    (`BinnedScatter.check_status`: entries in the active rows == tickets == accumulated + overflow over all the steps the test
    ran) — a lost or duplicated entry fails the test that produced it even where the test's own comparison would not see it."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    from brainevent_amd import _csr
    yield
    _csr.check_binned_status()           # every binned workspace still alive (the library keeps a weak registry of them)


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """BE_POISON_ALLOC=1 (robustness runs on the GPU box): before every gpu test, what the caching allocator will hand out next
    is filled with 0xFF bytes (NaN as a float, -1 as an integer, all spikes set as a mask), so a kernel or a host path that
    relies on `torch.empty` memory being zero — true in a fresh process, false in a long-running one — fails here instead of
    once in a while.  Large pool: one block of BE_POISON_GIB GiB (default 16; tests that allocate more get fresh pages beyond
    it), small pool: 512 blocks of 1 MiB."""
    if os.environ.get('BE_POISON_ALLOC') not in ('1', 'ff') or request.node.get_closest_marker('gpu') is None:
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
