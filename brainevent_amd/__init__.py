"""brainevent_amd — MI355X-native event-driven sparse matmul (brainevent-compatible ``@`` surface).

Only the spike-triggered SpMV/SpMM hot path of chaobrain/brainevent is provided:
``BinaryArray @ {CSR, CSC, dense, JITC{Scalar,Normal,Uniform}{R,C}, FixedNumPerPre/PerPost}`` and the
functional ``binary_*`` operators, running hand-written HIP kernels (gfx950) through a C ABI.
"""
from ._version import __version__
__version_info__ = tuple(int(p) for p in __version__.split('.')[:3] if p.isdigit())
from ._error import (BrainEventError, MathError, KernelError, KernelNotAvailableError, KernelCompilationError,
                     KernelFallbackExhaustedError, KernelExecutionError, KernelLoadError, UnsupportedOperationError,
                     BenchmarkDataFnNotProvidedError, KernelToolchainError, CompilationError, KernelRegistrationError)
from . import config
from ._registry import get_registry, get_primitives_by_tags, get_all_primitive_names
from ._event import EventRepresentation, BinaryArray, BitPackedBinary, CompactBinary, bitpack
from ._csr import (CSR, CSC, ScatterPlan, BinnedScatter, check_binned_status, PlannedMatrix, Mirror, indexed_workspace, build_mirror_of, hybrid_task_capacity, binary_csrmv, binary_csrmm, binary_csrmv_indexed, binary_csrmm_indexed, binary_csrmv_indexed_p, binary_csrmm_indexed_p, binary_csrmv_p, binary_csrmm_p,
                   binary_csrmv_p_call, binary_csrmm_p_call)
from ._fcn import (FixedNumConn, FixedNumPerPre, FixedNumPerPost, binary_fcnmv, binary_fcnmm, binary_fcnmv_p,
                   binary_fcnmm_p, binary_fcnmv_p_call, binary_fcnmm_p_call)
from ._dense import (Dense, binary_densemv, binary_densemm, binary_densemv_p, binary_densemm_p, binary_densemv_p_call,
                     binary_densemm_p_call)
from ._convert import (csr_to_coo_index, coo_to_csc_index, coo2csr, csr_to_csc_index, csc_to_csr_index,
                       fixed_conn_num_csr_indptr, fixed_conn_num_csc_structure, fixed_conn_num_to_csc, CscBuilder)
from ._float import (csrmv, csrmm, csrmv_p, csrmm_p, csrmv_p_call, csrmm_p_call, fcnmv, fcnmm, fcnmv_p, fcnmm_p, fcnmv_p_call,
                     fcnmm_p_call)
from ._graph import GraphedStep, capture_step
from ._tuning import (ScatterTuning, DEFAULT_SCATTER_TUNING, get_scatter_tuning, save_scatter_tuning, apply_scatter_tuning,
                      tune_scatter_routes)
from ._neuron import lif_coba_step, lif_cuba_step
from ._op import OpKernel
XLACustomKernel = OpKernel      # the operator object under the reference's name (no XLA underneath)
from ._data import DataRepresentation
from ._jitc import (JITCScalarMatrix, JITCUniformMatrix, JITCNormalMatrix, JITCMatrix, JITCScalarR, JITCScalarC, JITCUniformR, JITCUniformC, JITCNormalR, JITCNormalC,
                    binary_jitsmv, binary_jitsmm, binary_jitumv, binary_jitumm, binary_jitnmv, binary_jitnmm,
                    binary_jitsmv_p, binary_jitsmm_p, binary_jitumv_p, binary_jitumm_p, binary_jitnmv_p, binary_jitnmm_p,
                    binary_jitsmv_p_call, binary_jitsmm_p_call, binary_jitumv_p_call, binary_jitumm_p_call,
                    binary_jitnmv_p_call, binary_jitnmm_p_call, JITCScatterShard, JITCGatherShard,
                    jitsmv, jitsmm, jitumv, jitumm, jitnmv, jitnmm, jitsmv_p, jitsmm_p, jitumv_p, jitumm_p, jitnmv_p, jitnmm_p,
                    jitsmv_p_call, jitsmm_p_call, jitumv_p_call, jitumm_p_call, jitnmv_p_call, jitnmm_p_call)
