"""Exception hierarchy for the hot path.

Mirrors the names a brainevent user catches (reference ``brainevent/_error.py:43-405``):
``BrainEventError`` > ``MathError`` / ``KernelError`` > ``KernelNotAvailableError`` /
``KernelFallbackExhaustedError`` / ``KernelExecutionError`` / ``KernelLoadError`` /
``KernelCompilationError``.  Toolchain (nvcc) errors of the reference have no counterpart: the
HIP library is built ahead of time.
"""


class BrainEventError(Exception):
    """Base class of every error raised by this package."""


class MathError(BrainEventError):
    """Invalid mathematical operation (e.g. ``@`` on a 0-d or 3-d event array)."""


class UnsupportedOperationError(BrainEventError):
    """The operation is outside the accelerated hot path."""


class KernelError(BrainEventError):
    """Base class for kernel build / load / dispatch / execution errors."""


class KernelNotAvailableError(KernelError):
    """No usable kernel: the HIP library loaded but no MI355X device is visible."""


class KernelCompilationError(KernelError):
    """hipcc failed while building ``libbrainevent_amd.so``."""


class KernelFallbackExhaustedError(KernelError):
    """The requested backend is not registered for the operator."""


class KernelExecutionError(KernelError):
    """A C-ABI call returned a negative status; carries the library's message."""


class KernelLoadError(KernelError):
    """``libbrainevent_amd.so`` is missing or cannot be loaded."""


class BenchmarkDataFnNotProvidedError(BrainEventError, ValueError):
    """``benchmark()`` was called on an operator without a registered data generator (``def_benchmark_data``)."""


class KernelToolchainError(KernelError):
    """The compilation toolchain (hipcc) is missing or incompatible."""


class CompilationError(KernelCompilationError):
    """Compilation failed; carries the compiler output like the reference's class of the same name."""

    def __init__(self, message: str, compiler_output: str = "", command: str = "", stage: str = "compile"):
        super().__init__(message)
        self.compiler_output, self.command, self.stage = compiler_output, command, stage


class KernelRegistrationError(KernelError):
    """Registering a kernel with the operator table failed."""
