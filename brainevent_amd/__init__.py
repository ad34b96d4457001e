"""brainevent_amd — MI355X-native event-driven sparse matmul (brainevent-compatible ``@`` surface).

Only the spike-triggered SpMV/SpMM hot path of chaobrain/brainevent is provided:
``BinaryArray @ {CSR, CSC, dense, JITC{Scalar,Normal,Uniform}{R,C}, FixedNumPerPre/PerPost}`` and the
functional ``binary_*`` operators, running hand-written HIP kernels (gfx950) through a C ABI.
"""
from ._version import __version__
from ._error import (BrainEventError, MathError, KernelError, KernelNotAvailableError, KernelCompilationError,
                     KernelFallbackExhaustedError, KernelExecutionError, KernelLoadError, UnsupportedOperationError)
from . import config
from ._registry import get_registry, get_primitives_by_tags, get_all_primitive_names
from ._event import EventRepresentation, BinaryArray
from ._csr import (CSR, CSC, ScatterPlan, binary_csrmv, binary_csrmm, binary_csrmv_p, binary_csrmm_p,
                   binary_csrmv_p_call, binary_csrmm_p_call)
from ._fcn import (FixedNumConn, FixedNumPerPre, FixedNumPerPost, binary_fcnmv, binary_fcnmm, binary_fcnmv_p,
                   binary_fcnmm_p, binary_fcnmv_p_call, binary_fcnmm_p_call)
from ._dense import (binary_densemv, binary_densemm, binary_densemv_p, binary_densemm_p, binary_densemv_p_call,
                     binary_densemm_p_call)
