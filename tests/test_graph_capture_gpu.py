"""Every operator family captured into a HIP graph (torch.cuda.CUDAGraph) and replayed on new spike contents: the C-ABI
calls are asynchronous on the current stream, allocate nothing themselves and never synchronise the host, so a whole
time step can be captured once ("HIP streams and graphs instead of a tracing compiler")."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture_and_replay(make_out, spikes_buf, new_spikes_list):
    """make_out() -> tensor computed from spikes_buf; returns the outputs of graph replays after refilling spikes_buf."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            make_out()                                   # warm-up outside capture (plans, workspaces)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = make_out()
    res = []
    for s in new_spikes_list:
        spikes_buf.copy_(s)
        g.replay()
        torch.cuda.synchronize()
        res.append(out.clone())
    return res


@pytest.mark.parametrize('kind', ['csr_direct', 'csr_plan_d8', 'csr_plan_homo', 'csr_binned', 'csr_gather', 'csr_mm_plan', 'fcn',
                                  'jitc_scatter', 'jitc_gather', 'dense_mv', 'dense_mm_mfma', 'packed_events'])
def test_operator_captures_into_a_hip_graph(kind, monkeypatch):
    import brainevent_amd as be
    import brainevent_amd._csr as C
    rng = np.random.default_rng(7)
    dev = torch.device('cuda')
    m, k, nc = 3000, 40000, 200
    ptr = torch.arange(m + 1, dtype=torch.int32, device=dev) * nc
    idx = torch.tensor(rng.integers(0, k, m * nc).astype(np.int32), device=dev)
    w = torch.tensor(rng.random(m * nc).astype(np.float32), device=dev)
    w1 = torch.ones(1, device=dev)

    def eager(fn_of_spikes, spikes_list):
        return [fn_of_spikes(s).clone() for s in spikes_list]

    if kind.startswith('csr') or kind == 'packed_events':
        homo = kind == 'csr_plan_homo'
        csr = be.CSR((w1 if homo else w, idx, ptr), shape=(m, k), check_structure=False)
        if kind in ('csr_plan_d8', 'csr_plan_homo', 'csr_mm_plan', 'packed_events'):
            csr.buffers['scatter_plan'] = C.ScatterPlan.build(csr.data, idx, ptr, shape=(m, k))
        elif kind == 'csr_binned':
            csr.buffers['scatter_plan'] = C.BinnedScatter(w, m, k, idx.numel(), indices=idx)
        else:
            csr.buffers['scatter_plan'] = None
        if kind == 'csr_gather':
            buf = torch.zeros(k, dtype=torch.bool, device=dev)
            f = lambda s: csr @ be.BinaryArray(s)
            news = [torch.tensor(rng.random(k) < 0.1, device=dev) for _ in range(3)]
        elif kind == 'csr_mm_plan':
            buf = torch.zeros((4, m), dtype=torch.bool, device=dev)
            f = lambda s: be.BinaryArray(s) @ csr
            news = [torch.tensor(rng.random((4, m)) < 0.1, device=dev) for _ in range(3)]
        elif kind == 'packed_events':
            buf = torch.zeros(m, dtype=torch.bool, device=dev)
            f = lambda s: be.BinaryArray(s).bitpack() @ csr
            news = [torch.tensor(rng.random(m) < 0.1, device=dev) for _ in range(3)]
        else:
            buf = torch.zeros(m, dtype=torch.bool, device=dev)
            f = lambda s: be.BinaryArray(s) @ csr
            news = [torch.tensor(rng.random(m) < 0.1, device=dev) for _ in range(3)]
    elif kind == 'fcn':
        conn = be.FixedNumPerPre((w.view(m, nc), idx.view(m, nc)), shape=(m, k), check_indices=False)
        buf = torch.zeros(m, dtype=torch.bool, device=dev)
        f = lambda s: be.BinaryArray(s) @ conn
        news = [torch.tensor(rng.random(m) < 0.1, device=dev) for _ in range(3)]
    elif kind.startswith('jitc'):
        J = be.JITCUniformR((np.float32(0.1), np.float32(0.9), 0.01, 5), shape=(m, k), corder=(kind == 'jitc_scatter'))
        buf = torch.zeros(m, dtype=torch.bool, device=dev)
        f = lambda s: be.BinaryArray(s) @ J
        news = [torch.tensor(rng.random(m) < 0.1, device=dev) for _ in range(3)]
    else:
        W = torch.tensor(rng.standard_normal((4096, 1024)), dtype=torch.float16 if kind == 'dense_mm_mfma' else torch.float32,
                         device=dev)
        shp = (16, 4096) if kind == 'dense_mm_mfma' else (4096,)
        buf = torch.zeros(shp, dtype=torch.bool, device=dev)
        f = lambda s: be.BinaryArray(s) @ W
        news = [torch.tensor(rng.random(shp) < 0.2, device=dev) for _ in range(3)]

    ref = eager(f, news)
    got = _capture_and_replay(lambda: f(buf), buf, news)
    for a, b in zip(got, ref):
        if kind in ('csr_direct', 'csr_binned', 'fcn'):          # float-atomic merges: order dependent
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
        else:
            assert torch.equal(a, b), kind
