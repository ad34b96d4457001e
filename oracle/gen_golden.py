#!/usr/bin/env python3
"""Generate tests/golden/* in the build container (the reference never travels to the GPU box).

Sources of truth used here:
  1. the reference's own numpy-only golden model  brainevent/_jit_uniform/_test_util.py, loaded BY FILE PATH
     (it imports only math and numpy) — edges, uniform dense matrices, scalar RNG values;
  2. the reference's known-answer tests and docstring examples, transcribed as data (inputs + expected outputs)
     with their file:line.
Run:  python oracle/gen_golden.py   (needs /root/reference; writes small .npz / .json fixtures)
"""
import importlib.util
import json
import os
import sys

import numpy as np

REF = os.environ.get('BE_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def load_ref_util():
    path = os.path.join(REF, 'brainevent', '_jit_uniform', '_test_util.py')
    spec = importlib.util.spec_from_file_location('_ref_jit_uniform_test_util', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_normal_util(ru):
    """brainevent/_jit_normal/_test_util.py is numpy-only in its body but imports the uniform golden model through the
    package path (`from brainevent._jit_uniform._test_util import ...`).  The package itself cannot be imported here (jax /
    brainunit absent), so the already-loaded uniform util is registered under the module name the file asks for — nothing
    of the reference is rewritten or stubbed, the two files it consists of are executed as they are."""
    import types
    for name in ('brainevent', 'brainevent._jit_uniform'):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg
    sys.modules['brainevent._jit_uniform._test_util'] = ru
    path = os.path.join(REF, 'brainevent', '_jit_normal', '_test_util.py')
    spec = importlib.util.spec_from_file_location('_ref_jit_normal_test_util', path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    finally:
        for name in ('brainevent._jit_uniform._test_util', 'brainevent._jit_uniform', 'brainevent'):
            sys.modules.pop(name, None)
    return mod


def main():
    os.makedirs(OUT, exist_ok=True)
    ru = load_ref_util()
    rn = load_ref_normal_util(ru)

    # ---- 1a. scalar RNG values on a grid -------------------------------------------------------------
    grid = [0, 1, 2, 7, 42, 123, 0xFFFFFFFF, 0x80000000, 0x6d2b79f5, 99991]
    scal = {
        'mix32': [[x, int(ru.mix32(x))] for x in grid],
        'next': [[x, int(ru.light_rng_next(x))] for x in grid],
        'bounded': [[r, b, int(ru.fast_bounded_u32(r, b))] for r in grid for b in (1, 2, 19, 1999, 0xFFFFFFFF)],
        'init': [[s, r, c, l, int(ru.light_rng_init(s, r, c, l))] for s in (0, 42, 123) for r in (0, 5, 100000)
                 for c in (0, 3) for l in (0, 1, 31)],
        'initial_q': [],
        'uniform01': [[s, r, c, float(ru.hash_uniform01(s, r, c))] for s in (0, 42, 123) for r in (0, 7, 4000000)
                      for c in (0, 11, 3999999)],
        'conn_length': [[p, int(ru.conn_length(p))] for p in (0.0, 0.001, 0.01, 0.1, 0.2, 0.3, 0.5, 1.0)],
        'default_chunk_size': [[n, int(ru.default_chunk_size(n))] for n in (1, 3, 4, 5, 17, 30, 50, 4000000)],
    }
    for st in (1, 42, 0xdeadbeef, 0x12345678):
        for cl in (2, 3, 10, 20, 2000):
            q, s2 = ru.stationary_initial_q(st, cl)
            scal['initial_q'].append([st, cl, int(q), int(s2)])

    # ---- 1b. edge lists ------------------------------------------------------------------------------
    edges = {}
    for seed in (42, 123):
        for prob in (0.1, 0.2, 1.0):
            for (n_rows, n_cols) in ((13, 17), (20, 30), (100, 50)):
                for stride in (32, 4):
                    clen = ru.conn_length(prob)
                    e = np.array([(rr, rc) for (_, _, rr, rc) in
                                  ru.iter_edges(seed, clen, n_rows, n_cols, corder=True, stride=stride)], dtype=np.int32)
                    edges[f's{seed}_p{prob}_r{n_rows}_c{n_cols}_st{stride}'] = e.reshape(-1, 2)
    # small-prob case on a small walk with an explicit chunk size (the C3 parameters scaled down)
    e = np.array([(rr, rc) for (_, _, rr, rc) in
                  ru.iter_edges(42, ru.conn_length(0.001), 6, 5000, corder=True, stride=32, chunk_size=1250)], dtype=np.int32)
    edges['s42_p0.001_r6_c5000_st32'] = e.reshape(-1, 2)
    np.savez_compressed(os.path.join(OUT, 'jitc_edges.npz'), **edges)

    # ---- 1c. dense uniform matrices (all transpose / corder / mode combinations) ------------------------
    dense = {}
    for shape in ((13, 17), (20, 30)):
        for transpose in (False, True):
            for corder in (False, True):
                for mm in ('mv', 'mm'):
                    d = ru.dense_uniform_reference(np.float32(-1.5), np.float32(1.5), 0.2, 123, shape=shape,
                                                   transpose=transpose, corder=corder, matrix_mode=mm)
                    dense[f'{shape[0]}x{shape[1]}_t{int(transpose)}_c{int(corder)}_{mm}'] = d
    np.savez_compressed(os.path.join(OUT, 'jitu_dense.npz'), **dense)

    # ---- 1d. dense normal matrices from the reference's normal golden model (_jit_normal/_test_util.py:50-80) and a
    #          grid of its hash_normal01 values (:10-47) — central region and both tails ------------------------------
    dense_n = {}
    for shape in ((13, 17), (20, 30)):
        for transpose in (False, True):
            for corder in (False, True):
                for mm in ('mv', 'mm'):
                    d = rn.dense_normal_reference(np.float32(0.25), np.float32(1.5), 0.2, 123, shape=shape,
                                                  transpose=transpose, corder=corder, matrix_mode=mm)
                    dense_n[f'{shape[0]}x{shape[1]}_t{int(transpose)}_c{int(corder)}_{mm}'] = d
    np.savez_compressed(os.path.join(OUT, 'jitn_dense.npz'), **dense_n)
    nrm = [[s, r, c, float(rn.hash_normal01(s, r, c))] for s in (0, 42, 123) for r in range(0, 40, 3) for c in range(0, 40, 3)]
    us = np.array([float(ru.hash_uniform01(s, r, c)) for s, r, c, _ in nrm])
    assert (us < 0.02425).any() and (us > 0.97575).any(), "the normal01 grid must reach both Acklam tails"
    scal['normal01'] = nrm
    # the reference's own exact-value tests of the two hashes (data: inputs + expected outputs)
    scal['uniform01_reference_test'] = {
        'src': 'brainevent/_numba_random_test.py:58-70', 'compare': 'exact float32',
        'cases': [[42, 0, 0, 0.2929498553276062], [42, 3, 7, 0.548724353313446], [123, 19, 29, 0.5329357385635376],
                  [0, 1, 2, 0.31099069118499756], [0xFFFFFFFF, 65535, 123456, 0.8090267777442932]]}
    scal['normal01_reference_test'] = {
        'src': 'brainevent/_numba_random_test.py:81-93', 'compare': 'rtol 1e-6, atol 1e-6',
        'cases': [[42, 0, 0, -0.5447874069213867], [42, 3, 7, 0.12243907153606415], [123, 19, 29, 0.08265165984630585],
                  [0, 1, 2, -0.4930441081523895], [0xFFFFFFFF, 65535, 123456, 0.8743151426315308]]}
    json.dump(scal, open(os.path.join(OUT, 'light_rng_scalars.json'), 'w'))

    # ---- 2. known-answer tests transcribed from the reference's tests / docstrings -----------------------
    kat = [
        {'src': 'brainevent/_csr/binary_test.py:76-96', 'op': 'csrmv', 'w': [1., 2.], 'indices': [0, 1], 'indptr': [0, 2],
         'v': [True, False], 'shape': [1, 2], 'transpose': False, 'expect': [1.0]},
        {'src': 'brainevent/_csr/binary_test.py:125-145', 'op': 'csrmm', 'w': [1., 2.], 'indices': [0, 1], 'indptr': [0, 2],
         'B': [[True, False], [False, True]], 'shape': [1, 2], 'transpose': False, 'expect': [[1.0, 2.0]]},
        {'src': 'brainevent/_csr/binary_test.py:343-366', 'op': 'csrmv', 'w': [2.0], 'indices': [0, 2, 1, 2],
         'indptr': [0, 2, 4], 'v': [True, False, True], 'shape': [2, 3], 'transpose': False, 'expect': [4., 2.]},
        {'src': 'brainevent/_csr/binary_test.py:370-393', 'op': 'csrmv', 'w': [1., 2., 3., 4.], 'indices': [0, 2, 1, 2],
         'indptr': [0, 2, 4], 'v': [1.0, -1.0], 'shape': [2, 3], 'transpose': True, 'expect': [1., 0., 2.]},
        {'src': 'brainevent/_csr/binary_test.py:397-421', 'op': 'csrmm', 'w': [2.0], 'indices': [0, 2, 1, 2],
         'indptr': [0, 2, 4], 'B': [[True, False], [False, True], [True, True]], 'shape': [2, 3], 'transpose': False,
         'expect': [[4., 2.], [2., 4.]]},
        {'src': 'brainevent/_csr/binary_test.py:425-449', 'op': 'csrmm', 'w': [1., 2., 3., 4.], 'indices': [0, 2, 1, 2],
         'indptr': [0, 2, 4], 'B': [[1., -1.], [-1., 1.]], 'shape': [2, 3], 'transpose': True,
         'expect': [[1., 0.], [0., 3.], [2., 4.]]},
        {'src': 'brainevent/_event/binary.py:183-185', 'op': 'densemv', 'W': [[1., 2.], [3., 4.], [5., 6.]],
         's': [True, False, True], 'transpose': True, 'expect': [6., 8.]},
        {'src': 'brainevent/_event/binary.py:258-261', 'op': 'densemv', 'W': [[1., 2., 3.], [4., 5., 6.]],
         's': [True, False, True], 'transpose': False, 'expect': [4., 10.]},
        {'src': 'brainevent/_event/binary_test.py:45-60', 'op': 'densemv', 'W': [[1., 2.], [3., 4.], [5., 6.]],
         's': [0, 1, 1], 'transpose': True, 'expect': [8., 10.]},
        {'src': 'brainevent/_event/binary_test.py:75-99', 'op': 'densemv', 'W': [[1., 2., 3.], [4., 5., 6.]],
         's': [0, 1, 1], 'transpose': False, 'expect': [5., 11.]},
        {'src': 'brainevent/_fcn/binary.py:124-130', 'op': 'fcnmv', 'w': [1.0], 'indices': [[0, 1], [1, 2]],
         's': [True, False, True], 'shape': [2, 3], 'transpose': False, 'expect': [1., 1.]},
        {'src': 'brainevent/_fcn/binary.py:646-654', 'op': 'fcnmm', 'w': [1.0], 'indices': [[0, 1], [1, 2]],
         'M': [[True, False], [False, True], [True, True]], 'shape': [2, 3], 'transpose': False,
         'expect': [[1., 1.], [1., 2.]]},
    ]
    json.dump(kat, open(os.path.join(OUT, 'kat.json'), 'w'), indent=1)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    sys.exit(main())
