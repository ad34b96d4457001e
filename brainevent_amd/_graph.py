"""Capture a whole time step into a HIP graph and replay it.

The operators issue asynchronous launches on the current stream, allocate nothing themselves and never synchronise the
host, so a step function built from them (plus elementwise torch ops on preallocated state) captures as it is.  For small
and mid-size networks the step is bound by the host's launch rate (≈ 14–22 µs per operator call); a replayed graph removes
that: ``examples/coba_2005.py`` runs 177 µs per step eagerly and 87 µs replayed.  (This plays the part ``jax.jit`` plays
for the reference: trace once, launch cheaply — by recording the launches instead of tracing a program.)
"""
from typing import Any, Callable

import torch

from ._lib import require_device

__all__ = ['GraphedStep', 'capture_step']


class GraphedStep:
    """``step = capture_step(fn)``; then ``step()`` replays ``fn``'s launches and returns the (static) outputs of the
    captured call.  ``fn`` must read its inputs from tensors that stay at the same addresses (update them in place with
    ``copy_`` between replays) and must not synchronise or branch on device values."""

    def __init__(self, fn: Callable[[], Any], warmup: int = 3, repeat: int = 1):
        require_device()
        if repeat < 1:
            raise ValueError('capture_step: repeat must be >= 1')
        self.repeat = int(repeat)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):          # builds plans / workspaces outside the capture
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(self.repeat):          # repeat > 1: that many consecutive steps of an in-place step function per replay
                self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs

    def check(self) -> None:
        """Status of the binned workspaces the replays ran on (``brainevent_amd.check_binned_status``): a replayed step cannot raise,
        so a step that poisoned its outputs (NaN) or lost an entry is reported here.  Synchronises; call it at a sync point."""
        from ._csr import check_binned_status
        check_binned_status()


def capture_step(fn: Callable[[], Any], warmup: int = 3, repeat: int = 1) -> GraphedStep:
    """``repeat`` consecutive invocations of ``fn`` are recorded into the one graph (a step function that advances its state in
    place: one replay = ``repeat`` time steps, one host call instead of ``repeat``)."""
    return GraphedStep(fn, warmup=warmup, repeat=repeat)
