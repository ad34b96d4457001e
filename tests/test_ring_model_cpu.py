"""The append protocol of the binned route (write-combining ring of two blocks per bin: tickets, generations, commits, flush
list, drain, directory — ``brainevent_amd/csrc/be_csr_binned.hip``) as a host C++ model (``tests/c/ring_model.cc``), run

* under **ThreadSanitizer** with waves as threads and random yields: the block words are plain memory, so any access the
  protocol's acquire / release edges do not order is reported, and every output is compared with the serial sum (each entry
  has a unique weight: a duplicate, a loss or a misplaced entry changes a sum) together with the conservation counters the
  library keeps (entries == tickets == accumulated + overflowed); geometries include regions that overflow (the float-atomic
  path), flush lists shorter than the blocks completed at once, lanes without an entry and all lanes on one bin;
* as an **exhaustive search** over every interleaving of two waves' DS instructions on a small case: every final state
  delivers every entry exactly once.

(sanitizers belong on the CPU build.)"""
import shutil
import subprocess
from pathlib import Path

import pytest

SRC = Path(__file__).resolve().parent / 'c' / 'ring_model.cc'


@pytest.fixture(scope='module')
def binaries(tmp_path_factory):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('no g++')
    d = tmp_path_factory.mktemp('ring_model')
    plain, tsan = d / 'ring', d / 'ring_tsan'
    subprocess.run([gxx, '-O2', '-std=c++17', '-pthread', str(SRC), '-o', str(plain)], check=True)
    r = subprocess.run([gxx, '-O1', '-g', '-std=c++17', '-fsanitize=thread', '-pthread', str(SRC), '-o', str(tsan)],
                       capture_output=True, text=True)
    return plain, (tsan if r.returncode == 0 else None)


def test_every_interleaving_of_two_waves_delivers_every_entry_once(binaries):
    r = subprocess.run([str(binaries[0]), 'explore'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'every entry delivered exactly once' in r.stdout


def test_threaded_model_is_race_free_under_thread_sanitizer(binaries):
    if binaries[1] is None:
        pytest.skip('this g++ has no ThreadSanitizer runtime')
    r = subprocess.run([str(binaries[1]), 'stress', '150'], capture_output=True, text=True, timeout=600,
                       env={'TSAN_OPTIONS': 'halt_on_error=1 exitcode=66'})
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert 'ThreadSanitizer' not in r.stderr
    assert '150 seeds, 0 bad' in r.stdout


def test_threaded_model_without_the_sanitizer(binaries):
    r = subprocess.run([str(binaries[0]), 'stress', '400'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '400 seeds, 0 bad' in r.stdout, r.stdout + r.stderr
