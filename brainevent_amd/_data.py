"""``DataRepresentation``: the contract every weight-matrix container shares (reference ``brainevent/_data.py:35-230``,
minus ``brainunit.sparse.SparseMatrix`` and the pytree plumbing).

A container carries named *buffers* — per-matrix, non-differentiable state that travels with it, e.g. the scatter plan a
``CSR`` builds on first use (the reference keeps its task workspace there, ``_csr/main.py:148-161``) — and declares the
structure operations of a generic sparse weight matrix; a family that cannot support one refuses it explicitly."""
from typing import Dict, Optional

__all__ = ['DataRepresentation']


class DataRepresentation:
    """Base of ``CSR`` / ``CSC``, ``FixedNumPerPre`` / ``FixedNumPerPost``, the ``JITC*`` matrices and ``Dense``.

    ``buffers`` is a plain dict here (the reference exposes the registered names through a property of the same name);
    ``register_buffer`` / ``set_buffer`` keep the reference's semantics: a buffer has to be registered before it is set."""

    buffers: Dict
    # a numpy array on the LEFT of ``@`` defers to the container's ``__rmatmul__`` instead of treating it as a 0-d object
    __array_ufunc__ = None

    def _init_buffers(self, buffers: Optional[Dict]) -> None:
        if buffers is not None and not isinstance(buffers, dict):
            raise AssertionError("buffers must be a dictionary of name-value pairs.")
        self.buffers = dict(buffers) if buffers else {}

    def register_buffer(self, name, value=None):
        """Register a named buffer with a default value."""
        self.buffers[name] = value

    def set_buffer(self, name, value):
        """Update the value of a previously registered buffer."""
        if name not in self.buffers:
            raise ValueError(f"Buffer '{name}' not registered. Call register_buffer first.")
        self.buffers[name] = value

    # ---- common-API contract: declared here, overridden (or deliberately refused) by the families
    @classmethod
    def fromdense(cls, *args, **kwargs):
        raise NotImplementedError(f"{cls.__name__}.fromdense")

    def todense(self):
        raise NotImplementedError(f"{type(self).__name__}.todense")

    def tocoo(self):
        raise NotImplementedError(f"{type(self).__name__}.tocoo")

    def tocsr(self):
        raise NotImplementedError(f"{type(self).__name__}.tocsr")

    def tocsc(self):
        raise NotImplementedError(f"{type(self).__name__}.tocsc")

    def with_data(self, data):
        raise NotImplementedError(f"{type(self).__name__}.with_data")

    def transpose(self, axes=None):
        raise NotImplementedError(f"{type(self).__name__}.transpose")
