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


def test_capture_step_helper():
    """``be.capture_step``: a two-projection step with in-place state, replayed; equals the eager loop."""
    import brainevent_amd as be
    rng = np.random.default_rng(3)
    dev = torch.device('cuda')
    n, nc = 3000, 50
    ptr = torch.arange(n + 1, dtype=torch.int32, device=dev) * nc

    def proj(w):
        idx = torch.tensor(rng.integers(0, n, n * nc).astype(np.int32), device=dev)
        return be.CSR((torch.full((1,), w, device=dev), idx, ptr), shape=(n, n), check_structure=False).prepare()

    A1, A2 = proj(0.3), proj(-0.2)

    def make_state():
        return {'v': torch.zeros(n, device=dev), 'spk': torch.tensor(rng.random(n) < 0.05, device=dev)}

    def step_fn(st):
        cur = (be.BinaryArray(st['spk']) @ A1) + (be.BinaryArray(st['spk']) @ A2)
        st['v'].mul_(0.9).add_(cur).add_(0.11)
        st['spk'].copy_(st['v'] > 1.0)
        st['v'].masked_fill_(st['spk'], 0.0)
        return st['v']

    s0 = make_state()
    s_eager = {k: v.clone() for k, v in s0.items()}
    s_graph = {k: v.clone() for k, v in s0.items()}
    for _ in range(20):
        step_fn(s_eager)
    keep = {k: v.clone() for k, v in s_graph.items()}
    step = be.capture_step(lambda: step_fn(s_graph))           # warm-up + capture advance the state: restore it
    for k2 in s_graph:
        s_graph[k2].copy_(keep[k2])
    for _ in range(20):
        out = step()
    torch.cuda.synchronize()
    step.check()                                      # status of the binned workspaces the replays ran on (none poisoned, entries conserved)
    assert out is s_graph['v']
    torch.testing.assert_close(s_graph['v'], s_eager['v'], rtol=1e-5, atol=1e-5)
    assert torch.equal(s_graph['spk'], s_eager['spk'])


def test_coba_network_firing_rate_matches_the_reference_example():
    """the COBA E/I network of the reference's example (examples/COBA_2005.py:35-87; 4000 LIF neurons, 80
    synapses per neuron, both projections `BinaryArray @ CSR`) settles at the firing rate the reference reports for it,
    ≈ 50.6 Hz (examples/COBA_2005.py:100-110) — an end-to-end sanity number: ± 2 Hz over 1 s of simulated time, in the
    eager loop and as a captured HIP graph."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'coba_2005.py')
    spec = importlib.util.spec_from_file_location('coba_2005_example', path)
    coba = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(coba)
    n, _, rate = coba.run_graph(1.0, 10000)
    assert n == 4000 and abs(rate - 50.6) <= 2.0, rate
    n, _, rate_eager = coba.run(1.0, 3000)
    assert abs(rate_eager - 50.6) <= 4.0, rate_eager          # 0.3 s: includes the start-up transient


def test_fused_neuron_step_reproduces_the_elementwise_formulation_bit_for_bit():
    """`be.lif_coba_step` (one launch) against the same formulas written as elementwise torch ops (examples/coba_2005.py
    `run`): identical membrane potentials, conductances, refractory timers and spikes after 300 steps of the COBA network —
    the kernel rounds every operation separately, in the same order — and the ≈ 50.6 Hz of the reference's example as a
    replayed HIP graph of three launches per step."""
    import importlib.util
    import os
    import brainevent_amd as be
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'coba_2005.py')
    spec = importlib.util.spec_from_file_location('coba_2005_example_fused', path)
    coba = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(coba)
    dev = torch.device('cuda', 0)
    dt = 0.1
    n_exc, n_inh, n, E, I, g = coba.build(1.0, dev)
    V0 = torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g)
    # elementwise formulation
    V, ge, gi, refr = V0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    spk = torch.zeros(n, dtype=torch.bool, device=dev)
    # fused
    Vf, gef, gif, refrf = V0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    spkf = torch.zeros(n, dtype=torch.bool, device=dev)
    cnt = torch.zeros(n, device=dev)
    dec_e, dec_i = float(torch.exp(torch.tensor(-dt / 5.0))), float(torch.exp(torch.tensor(-dt / 10.0)))
    import math
    n_spikes = 0
    for t in range(300):
        ge = ge * math.exp(-dt / 5.0) + (be.BinaryArray(spk[:n_exc]) @ E)
        gi = gi * math.exp(-dt / 10.0) + (be.BinaryArray(spk[n_exc:]) @ I)
        I_syn = (ge * (0.0 - V) + gi * (-80.0 - V)) * 1e-3
        dV = (-(V - (-60.0)) + I_syn + 20.0) * (dt / 20.0)
        active = refr <= 0
        V = torch.where(active, V + dV, V)
        spk = active & (V >= -50.0)
        V = torch.where(spk, torch.full_like(V, -60.0), V)
        refr = torch.where(spk, torch.full_like(refr, 5.0), refr - dt)
        n_spikes += int(spk.sum().item())
        be.lif_coba_step(Vf, gef, gif, refrf, be.BinaryArray(spkf[:n_exc]) @ E, be.BinaryArray(spkf[n_exc:]) @ I, spkf, cnt, dt=dt)
        assert torch.equal(spkf, spk), t
    assert n_spikes > 1000 and int(cnt.sum().item()) == n_spikes
    assert torch.equal(Vf, V) and torch.equal(gef, ge) and torch.equal(gif, gi) and torch.equal(refrf, refr)
    n, _, rate, _, _ = coba.run_fused(1.0, 10000, graph=True)
    assert n == 4000 and abs(rate - 50.6) <= 2.0, rate
    with pytest.raises(ValueError):
        be.lif_coba_step(Vf, gef, gif, refrf, gef, gif, spkf[:10])


def test_fused_current_based_step_reproduces_the_elementwise_formulation_bit_for_bit():
    """`be.lif_cuba_step` (the current-based twin: i_syn = (g_exc + g_inh) * scale; reference examples/CUBA_2005.py:35-66) against the
    same formulas as elementwise torch ops (examples/cuba_2005.py `elementwise_step`): identical state and spikes over 400 steps of
    the CUBA network (one projection carries a negative shared weight), then the reference's 24-25 Hz as a replayed HIP graph."""
    import importlib.util
    import os
    import brainevent_amd as be
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'cuba_2005.py')
    spec = importlib.util.spec_from_file_location('cuba_2005_example', path)
    cuba = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cuba)
    dev = torch.device('cuda', 0)
    dt = 0.1
    n_exc, n_inh, n, E, I, g = cuba.build(1.0, dev)
    V0 = torch.empty(n, device=dev).normal_(-55.0, 2.0, generator=g)
    V, ge, gi, refr = V0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    spk = torch.zeros(n, dtype=torch.bool, device=dev)
    Vf, gef, gif, refrf = V0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    spkf = torch.zeros(n, dtype=torch.bool, device=dev)
    bits = torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)
    cnt = torch.zeros(n, device=dev)
    n_spikes = 0
    for t in range(400):
        V, ge, gi, refr, spk = cuba.elementwise_step(V, ge, gi, refr, spk, E, I, n_exc, dt)
        n_spikes += int(spk.sum().item())
        be.lif_cuba_step(Vf, gef, gif, refrf, be.BinaryArray(spkf[:n_exc]) @ E, be.BinaryArray(spkf[n_exc:]) @ I, spkf, cnt,
                         spike_bits=bits, dt=dt)
        assert torch.equal(spkf, spk), t
    assert n_spikes > 500 and int(cnt.sum().item()) == n_spikes
    assert torch.equal(Vf, V) and torch.equal(gef, ge) and torch.equal(gif, gi) and torch.equal(refrf, refr)
    assert float(gif.min().item()) < 0.0          # the inhibitory projection's negative weight arrived as such
    assert torch.equal(be.bitpack(spkf, 0).reshape(-1).to(torch.int32), bits)
    n, _, rate, _, _ = cuba.run_fused(1.0, 10000, graph=True)
    assert n == 4000 and 21.0 <= rate <= 28.0, rate          # reference: 22.4-25.0 Hz at scale 1 (CUBA_2005.py:98-125)
    with pytest.raises(ValueError):
        be.lif_cuba_step(Vf, gef, gif, refrf, gef, gif, spkf[:10])


def test_a_captured_jitc_scatter_survives_the_eviction_of_every_cached_workspace():
    """The armed scatter-workspace cache (`_jitc._armed`) is never used while a stream captures: a graph captured on the very stream
    that warmed up (and so already owns a cached, armed workspace) records a workspace of its OWN pool; evicting every cached
    workspace afterwards — and scribbling over fresh allocations that would reuse the freed memory — leaves the replays exact.  Two
    batch shapes of equal workspace size but different layouts get different cache entries."""
    import brainevent_amd as be
    from brainevent_amd import _jitc as J
    dev = torch.device('cuda')
    rng = np.random.default_rng(3)
    m, k = 3000, 40000
    M = be.JITCScalarR((np.float32(1.0), 0.01, 11), shape=(m, k), corder=True)
    buf = torch.zeros(m, dtype=torch.bool, device=dev)
    news = [torch.tensor(rng.random(m) < 0.1, device=dev) for _ in range(4)]
    ref = [(be.BinaryArray(s) @ M).clone() for s in news]
    s_cap = torch.cuda.Stream()
    s_cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_cap):
        be.BinaryArray(buf) @ M                          # warm-up ON the capture stream: a cached armed workspace for (device, s_cap)
        torch.cuda.synchronize()
        n_cached = len(J._armed)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s_cap):
            out = be.BinaryArray(buf) @ M
        assert len(J._armed) == n_cached                 # the captured call added nothing to the cache
    cached_ptrs = {t.data_ptr() for t in J._armed.values()}
    # evict everything: more workspaces of distinct sizes than the cache holds push out every earlier entry (disarmed and released)
    with torch.cuda.stream(s_cap):
        for i in range(J._ARMED_MAX + 1):
            J._armed_scatter_workspace((1 << 20) + 4096 * i)
    torch.cuda.synchronize()
    assert len(J._armed) == J._ARMED_MAX and all(k[-1] >= (1 << 20) for k in J._armed)
    for key in list(J._armed):
        J._drop_armed(key)
    junk = [torch.full((1 << 20,), -1, dtype=torch.int32, device=dev) for _ in range(16)]      # reuse of whatever was freed
    torch.cuda.synchronize()
    for s, r in zip(news, ref):
        buf.copy_(s)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, r)
    del junk
    # the layout is part of the key
    J._armed.clear()
    J._armed_scatter_workspace(1 << 16, ('mm', 8))
    J._armed_scatter_workspace(1 << 16, ('mm', 72))
    assert len(J._armed) == 2
    for key in list(J._armed):
        J._drop_armed(key)


@pytest.mark.parametrize('kind,scale', [('coba', 1.0), ('cuba', 1.0), ('coba', 12.0), ('cuba', 30.0)])
def test_one_scatter_for_both_projections_gives_the_same_spikes_bit_for_bit(kind, scale):
    """`run_fused(combined=True)`: the excitatory and the inhibitory projection stacked into one n x 2n matrix of weight 1 — ONE scatter
    per time step — with the weights applied inside the neuron step (`in_scale_exc` / `in_scale_inh`, be_lif_step_scaled_packed)
    against the two-projection step: identical membrane potentials and spikes after 2000 steps at several network sizes (planned
    single-workgroup, planned and binned routes), for the conductance- and the current-based network (negative inhibitory weight)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', f'{kind}_2005.py')
    spec = importlib.util.spec_from_file_location(f'{kind}_2005_example_combined', path)
    net = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(net)
    n1, _, rate1, V1, s1 = net.run_fused(scale, 2000, graph=False)
    n2, _, rate2, V2, s2 = net.run_fused(scale, 2000, graph=False, combined=True)
    assert n1 == n2 and rate1 == rate2 and rate1 > 5.0
    assert torch.equal(V1, V2) and torch.equal(s1, s2)
    n3, _, rate3, V3, s3 = net.run_fused(scale, 2000, graph=True, unroll=10, combined=True)
    assert rate3 > 5.0 and n3 == n1
