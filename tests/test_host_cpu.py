"""CPU tests of the host logic: C-ABI surface, validators, operator dispatch contract, loud failure
without a GPU.  No compute calls are made (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'brainevent_amd.h')


def declared_symbols():
    """Every function the public header declares, after macro expansion (gcc -E)."""
    src = subprocess.run(['gcc', '-E', '-P', HEADER], check=True, capture_output=True, text=True).stdout
    names = set(re.findall(r'\b(be_[A-Za-z0-9_]+)\s*\(', src))
    assert len(names) > 200, len(names)
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from brainevent_amd import _lib
    if _lib.needs_build():
        _lib.build()
    lib = ctypes.CDLL(str(_lib.lib_path()))
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing[:10]
    lib.be_build_arch.restype = ctypes.c_char_p
    assert lib.be_build_arch() == b'gfx950' and lib.be_version() >= 100


def test_declared_symbol_families_cover_the_hot_path():
    names = set(declared_symbols())
    for fam in ('be_binary_csrmv_t_hetero_f32_bool', 'be_binary_csrmv_nt_homo_bf16_float', 'be_binary_csrmm_t_hetero_f16_bool',
                'be_binary_fcnmv_scatter_homo_f32_bool', 'be_binary_fcnmm_gather_hetero_f64_float',
                'be_binary_densemv_transpose_f32_bool', 'be_binary_densemm_no_transpose_bf16_float',
                'be_binary_jitsmv_notrans_f32', 'be_binary_jitumv_trans_f16', 'be_binary_jitnmm_trans_f64',
                'be_scatter_plan_count', 'be_scatter_plan_fill', 'be_binary_csrmm_t_plan', 'be_pack_spikes',
                'be_compact_spikes', 'be_last_error', 'be_profile_enable'):
        assert fam in names, fam


def test_ops_fail_loudly_without_gpu():
    import brainevent_amd as be
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    w = np.ones(2, np.float32)
    with pytest.raises(be.KernelNotAvailableError):
        be.binary_csrmv(w, np.array([0, 1], np.int32), np.array([0, 2], np.int32), np.array([True]), shape=(1, 2), transpose=True)
    with pytest.raises(be.KernelNotAvailableError):
        be.BinaryArray(np.array([True, False])) @ np.ones((2, 3), np.float32)
    with pytest.raises(be.KernelNotAvailableError):
        be.binary_jitsmv(np.float32(1.0), 0.5, np.ones(4, bool), 1, shape=(3, 4))


def test_missing_library_is_a_load_error(monkeypatch, tmp_path):
    from brainevent_amd import _lib
    import brainevent_amd as be
    monkeypatch.setenv('BE_HIP_LIB', str(tmp_path / 'nope.so'))
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(be.KernelLoadError):
        _lib.lib()


def test_structure_validators():
    from brainevent_amd import _misc as M
    idx = torch.tensor([0, 2, 1], dtype=torch.int64)
    out = M._as_int32_indices(idx, 3, 'ctx')
    assert out.dtype == torch.int32
    with pytest.raises(ValueError):
        M._as_int32_indices(torch.tensor([0, 3]), 3, 'ctx')
    with pytest.raises(ValueError):
        M._as_int32_indices(torch.tensor([-1, 1]), 3, 'ctx')
    with pytest.raises(TypeError):
        M._as_int32_indices(torch.tensor([0.5]), 3, 'ctx')
    assert M._resolve_indptr_dtype(10) == torch.int32
    assert M._resolve_indptr_dtype(2 ** 31) == torch.int64
    with pytest.raises(OverflowError):
        M._resolve_indptr_dtype(2 ** 31, np.int32)
    with pytest.raises(ValueError):
        M._resolve_indptr_dtype(5, 'int32')
    ptr = torch.tensor([0, 2, 3], dtype=torch.int32)
    ind = torch.tensor([0, 1, 1], dtype=torch.int32)
    M._check_compressed_structure(ind, ptr, (2, 2), 'csr')
    with pytest.raises(ValueError):
        M._check_compressed_structure(ind, torch.tensor([1, 2, 3], dtype=torch.int32), (2, 2), 'csr')     # indptr[0] != 0
    with pytest.raises(ValueError):
        M._check_compressed_structure(ind, torch.tensor([0, 3, 2], dtype=torch.int32), (2, 2), 'csr')     # not monotone
    with pytest.raises(ValueError):
        M._check_compressed_structure(ind, torch.tensor([0, 2, 4], dtype=torch.int32), (2, 2), 'csr')     # indptr[-1] != nse
    with pytest.raises(ValueError):
        M._check_compressed_structure(ind, ptr, (3, 2), 'csr')                                            # length
    with pytest.raises(TypeError):
        M._check_compressed_structure(ind.long(), ptr, (2, 2), 'csr')
    assert M._normalize_chunk_size(17) == 5 and M._normalize_chunk_size(4_000_000) == 1_000_000
    with pytest.raises(ValueError):
        M._normalize_chunk_size(8, 0)
    assert (M._MV_STRIDE, M._MM_STRIDE) == (32, 4)


def test_operator_dispatch_contract():
    import warnings
    import brainevent_amd as be
    from brainevent_amd._op import OpKernel
    op = OpKernel('unit_test_op')
    op.def_kernel('a', 'gpu', lambda x: ('a', x))
    op.def_kernel('b', 'gpu', lambda x: ('b', x))
    assert op.available_backends('gpu') == ['a', 'b'] and op(1) == ('a', 1)
    assert op(1, backend='b') == ('b', 1)                            # per-call wins
    be.config.set_backend('gpu', 'b')
    try:
        assert op(1) == ('b', 1)                                     # global beats the per-operator default
        be.config.set_backend('gpu', 'zzz')
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter('always')
            assert op(1) == ('a', 1)                                 # unknown global backend: warn + default
        assert wlist
    finally:
        be.config.clear_backends()
    with pytest.raises(be.KernelFallbackExhaustedError):
        op(1, backend='zzz')                                         # unknown per-call backend: error
    op.set_default('gpu', 'b')
    assert op(1) == ('b', 1)
    with pytest.raises(be.KernelFallbackExhaustedError):
        op.set_default('gpu', 'zzz')
    assert 'unit_test_op' in be.get_all_primitive_names()
    op.def_tags('x', 'y')
    assert 'unit_test_op' in be.get_primitives_by_tags({'x'})
    for name in ('binary_csrmv', 'binary_csrmm', 'binary_densemv', 'binary_densemm', 'binary_fcnmv', 'binary_fcnmm',
                 'binary_jitsmv', 'binary_jitsmm', 'binary_jitumv', 'binary_jitumm', 'binary_jitnmv', 'binary_jitnmm'):
        assert be.get_registry()[name].available_backends('gpu') == ['hip']


def test_config_backend_api():
    import brainevent_amd as be
    be.config.clear_backends()
    assert be.config.get_backend('gpu') is None
    be.config.set_backend('gpu', 'hip')
    assert be.config.get_backend('gpu') == 'hip'
    be.config.set_backend('gpu', None)
    assert be.config.get_backend('gpu') is None
    with pytest.raises(ValueError):
        be.config.set_backend('tpu', 'x')
    with pytest.raises(TypeError):
        be.config.set_backend('gpu', 3)


def test_binaryarray_container():
    import brainevent_amd as be
    s = be.BinaryArray([True, False, True])
    assert s.shape == (3,) and s.ndim == 1 and s.size == 3 and s.dtype == np.bool_ and len(s) == 3
    assert isinstance(s.with_value(np.zeros(2, bool)), be.BinaryArray)
    assert s[0] and not s[1]
    m = be.BinaryArray(np.zeros((2, 3), bool))
    assert m.T.shape == (3, 2) and m.transpose().shape == (3, 2)
    assert be.BinaryArray(s).value is s.value
    with pytest.raises(be.MathError):
        be.BinaryArray(np.zeros((2, 2, 2), bool)) @ np.ones((2, 2), np.float32)


def test_jit_host_helpers():
    from brainevent_amd import _jitc as J
    assert J._initialize_conn_length(0.001) == 2000 and J._initialize_conn_length(1.0) == 2 and J._initialize_conn_length(0.0) == 0
    assert J._initialize_conn_length(0.3) == 7
    assert J._initialize_seed(np.array([5], np.int32)) == 5 and isinstance(J._initialize_seed(None), int)
    # fixed-point exponent keeps |w| * 2^e * n below 2^62
    for wmax, n in ((1.0, 10), (0.5, 10 ** 7), (123.0, 4_000_000), (1e-6, 100)):
        e = J._fixed_scale_exp(wmax, n)
        assert wmax * 2.0 ** e * n < 2.0 ** 62
    with pytest.raises(ValueError):
        J._validate_prob(1.5)
    with pytest.raises(ValueError):
        J._validate_prob(float('nan'))
    with pytest.raises(ValueError):
        J._validate_prob(np.array([0.1, 0.2]))


def test_event_representation_minimal_api():
    """The container contract of the reference's ``brainevent/_event/base_test.py:25-66`` (pytree round trip excepted)."""
    import brainevent_amd as be
    with pytest.raises(TypeError):
        be.EventRepresentation(np.array([1, 2, 3]))                 # abstract
    arr = be.BinaryArray(np.array([1, 2, 3]))
    assert arr.shape == (3,) and arr.ndim == 1 and arr.size == 3 and arr.dtype == arr.value.dtype
    assert arr[0] == 1 and list(arr) == [1, 2, 3]
    arr2 = arr.with_value(np.array([4, 5, 6]))
    assert isinstance(arr2, be.BinaryArray) and arr2 is not arr
    assert np.array_equal(arr.value, [1, 2, 3]) and np.array_equal(arr2.value, [4, 5, 6])
    assert np.array_equal(np.asarray(arr), [1, 2, 3])
    with pytest.raises(TypeError):
        arr[0] = 5                                                   # item assignment is not supported
    # encodings exist on the CPU as types (their constructors need the device)
    assert issubclass(be.BitPackedBinary, be.EventRepresentation) and hasattr(be.CompactBinary, 'from_array')


def test_operator_call_function_contract():
    """``def_call`` / ``call`` of the operator object (reference ``brainevent/_op/main.py:1084-1150``)."""
    import brainevent_amd as be
    from brainevent_amd._op import OpKernel
    op = OpKernel('test_call_contract_op')
    with pytest.raises(ValueError, match="No call function registered"):
        op.call(1, 2)
    op.def_call(lambda a, b, backend=None: a + b)
    assert op.call(1, 2) == 3
    with pytest.raises(ValueError):
        op.benchmark(platform='gpu')                       # no benchmark data registered
    for name in ('binary_csrmv', 'binary_csrmm', 'binary_fcnmv', 'binary_fcnmm', 'binary_densemv', 'binary_densemm',
                 'binary_jitsmv', 'binary_jitsmm', 'binary_jitumv', 'binary_jitumm', 'binary_jitnmv', 'binary_jitnmm'):
        prim = getattr(be, name + '_p')
        assert prim._call_fn is getattr(be, name + '_p_call')


def test_scatter_plan_auto_geometry():
    """Layout / slice-width rules of a plan built without explicit choices (measured: tools/exp_layouts.py)."""
    from brainevent_amd._csr import ScatterPlan as P
    U16, D8, H8 = P.LAYOUT_U16, P.LAYOUT_D8, P.LAYOUT_H8

    def geo(m, k, row, homo, **kw):
        lay, w = P.auto_geometry(m, k, m * row, homo, P.default_shift(k, homo), **kw)
        return lay, -(-k // w)

    # the headline configs: whole-LDS slices, 5 / 10 parts
    assert geo(1_000_000, 1_000_000, 10000, False) == (D8, 51)
    assert geo(1_000_000, 1_000_000, 10000, True) == (H8, 25)
    # long rows over few columns: slices narrowed until a block fits one decode pass (and the partial sums shrink with them)
    assert geo(100_000, 100_000, 10000, False) == (D8, 51)
    assert geo(100_000, 100_000, 10000, True) == (H8, 25)
    assert geo(700_000, 700_000, 16000, True) == (H8, 42)
    assert geo(100_000, 100_000, 10000, True, force='u16') == (U16, 25)
    # mid-size outputs: ~24 slices (25 x 10 parts) instead of the few the LDS capacity would allow, blocks of >= 32 entries
    assert geo(100_000, 100_000, 1000, True) == (U16, 25) and geo(100_000, 100_000, 1000, False) == (D8, 25)
    assert geo(100_000, 100_000, 300, True) == (U16, 9)
    # a post slice of an 8-way partition (1M stored rows x 125k outputs): ~24 * (k / m)^(1/3) slices
    assert geo(1_000_000, 125_000, 1250, False) == (D8, 12) and geo(1_000_000, 125_000, 1250, True) == (U16, 12)
    # short rows: h8 has nothing to gain (blocks of a line or two)
    assert geo(1_000_000, 1_000_000, 1000, True) == (U16, 32)
    # ... and d8 would spend most of its items on escapes (column gaps of ~1000): uint16 columns, 8 lanes per block
    assert geo(1_000_000, 1_000_000, 1000, False) == (U16, 64)
    assert geo(350_000, 350_000, 1000, False) == (U16, 25) and geo(200_000, 200_000, 1000, False) == (U16, 25)
    # short blocks pay per block: as few slices as the LDS capacity allows (a multiple of 8), not a CU-filling count
    assert geo(1_500_000, 1_500_000, 1000, False) == (U16, 96) and geo(2_500_000, 2_500_000, 1000, True) == (U16, 80)
    # ... and below 8 (weighted) / 10 (counted) entries per block the binned route takes over
    import torch
    from brainevent_amd._csr import choose_scatter_route as route
    w1, wn = torch.ones(1), torch.ones(2)
    assert route(1_500_000 * 1000, 1_500_000, 1_500_000, wn) == 'plan' and route(2_500_000 * 1000, 2_500_000, 2_500_000, wn) == 'binned'
    assert route(2_500_000 * 1000, 2_500_000, 2_500_000, w1) == 'plan' and route(4_000_000 * 1000, 4_000_000, 4_000_000, w1) == 'binned'
    assert geo(300_000, 300_000, 3000, True)[0] == U16
    # one slice, <= 1M entries: the single-launch kernel (d8 reaches 20000 columns), counted entries stay u16
    assert geo(4000, 4000, 80, False) == (D8, 1) and geo(4000, 18000, 80, False) == (D8, 1)
    assert geo(4000, 4000, 80, True) == (U16, 1)
    # sorted layouts not applicable (f64 weights, rows beyond the LDS sort): u16 with the same pass-sized slices
    assert geo(100_000, 100_000, 10000, False, delta_ok=False) == (U16, 51)
    # more than 1024 slices: u16
    assert geo(10_000_000, 30_000_000, 1000, False)[0] == U16


def test_data_representation_buffer_contract():
    """DataRepresentation (reference ``_data.py:35-60``): named buffers have to be registered before they are set; the
    structure operations are declared on the base and refused there; every container family derives from it."""
    import brainevent_amd as be

    class X(be.DataRepresentation):
        def __init__(self, buffers=None):
            self._init_buffers(buffers)

    x = X({'a': 1})
    assert x.buffers == {'a': 1}
    x.register_buffer('b')
    assert x.buffers['b'] is None
    x.set_buffer('b', 3)
    assert x.buffers == {'a': 1, 'b': 3}
    with pytest.raises(ValueError):
        x.set_buffer('c', 1)
    with pytest.raises(AssertionError):
        X(buffers=[1])
    for op in (x.tocsr, x.tocsc, x.tocoo, x.todense, X.fromdense, x.transpose):
        with pytest.raises(NotImplementedError):
            op()
    for cls in (be.CSR, be.CSC, be.FixedNumPerPre, be.FixedNumPerPost, be.JITCScalarR, be.JITCUniformC, be.JITCNormalR, be.Dense):
        assert issubclass(cls, be.DataRepresentation)
    assert issubclass(be.JITCScalarR, be.JITCScalarMatrix) and issubclass(be.JITCScalarC, be.JITCScalarMatrix)
    assert issubclass(be.JITCUniformR, be.JITCUniformMatrix) and issubclass(be.JITCNormalC, be.JITCNormalMatrix)


def test_reference_task_workspace_names_and_formula():
    """`hybrid_task_capacity` (brainevent/_csr/hybrid_config.py:298-324: rows longer than 128 contribute
    ceil(len / 4096) tasks; same validation errors) and the explicit task workspace of brainevent/_csr/binary.py:76-120 exist
    under the reference's names, so that code which builds / unpacks one keeps working (the kernels here do not read it)."""
    import numpy as np
    import pytest
    import torch
    import brainevent_amd as be
    from brainevent_amd import _csr as C
    lens = np.array([0, 1, 128, 129, 4096, 4097, 10000, 128 * 3])
    ptr = np.concatenate([[0], np.cumsum(lens)])
    want = sum(-(-int(l) // 4096) for l in lens if l > 128)
    assert be.hybrid_task_capacity(ptr) == want == C.hybrid_task_capacity(torch.tensor(ptr)) == 1 + 1 + 2 + 3 + 1
    with pytest.raises(ValueError):
        be.hybrid_task_capacity(np.zeros((2, 2), np.int64))
    with pytest.raises(ValueError):
        be.hybrid_task_capacity(np.array([], np.int64))
    with pytest.raises(ValueError):
        be.hybrid_task_capacity(np.array([0, 5, 3]))
    for maker in (C._make_binary_csrmv_workspace, C._make_binary_csrmv_benchmark_workspace, C._make_binary_task_workspace):
        ws = maker(ptr.astype(np.int32))
        cap, tb, te, st = ws                                   # unpacks like the reference's namedtuple
        assert cap == ws.task_capacity == want and tb.shape == te.shape == (want,) and tb.dtype == np.int32
        assert st.shape == (2,) and st.dtype == np.int32
    assert C._BinaryTaskWorkspace is C._BinaryCsrmvTaskWorkspace


def test_the_bench_line_compacts_below_the_drivers_eight_kilobytes():
    """`bench.compact_line` on a committed verbose line of the default run (every secondary present): strict JSON of at most 8000
    bytes that keeps the driver contract's keys on the headline, a number set per secondary and the prose once in `legend`."""
    import glob
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    paths = sorted(glob.glob(os.path.join(root, 'profiles', 'r0[5-9]_*bench*line*.json')))
    assert paths
    for path in paths:
        full = json.load(open(path))
        if 'secondary' not in full or 'legend' in full:
            continue
        c = bench.compact_line(full)
        txt = json.dumps(bench._finite(c), allow_nan=False, separators=(',', ':'))
        assert len(txt.encode()) <= bench.LINE_BYTE_BOUND, (path, len(txt))
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                  'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'legend'):
            assert k in c, k
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in c['roofline'], k
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in c['cpu_baseline'], k
        assert set(c['secondary']) == set(full['secondary'])
        for name, e in c['secondary'].items():
            assert 'value' in e or 'error' in e, name
