// be_neuron.hip — the neuron half of the COBA step loop: one fused state update of a population of
// conductance-based leaky integrate-and-fire neurons with exponential synapses, around the two `spikes @ CSR` scatters of
// a time step.  The reference example builds these dynamics from brainstate modules (examples/COBA_2005.py:35-87: LIF,
// V_rest -60 mV, V_th -50 mV, V_reset -60 mV, tau 20 ms, refractory 5 ms; Expon synapses tau 5 / 10 ms, COBA outputs with
// reversal 0 / -80 mV); as separate elementwise launches they cost ~20 launches per step, which is all a 4000-neuron
// network's step consists of.  Every operation is rounded separately (no FMA contraction), in the order the plain
// elementwise formulation applies them, so the fused step reproduces that formulation bit for bit.
#include "be_common.h"
#include <algorithm>

namespace {

struct LifCobaP {
  float dt, dt_over_tau, v_rest, v_th, v_reset, t_ref, e_exc, e_inh, decay_exc, decay_inh, i_ext, syn_scale, in_scale_exc, in_scale_inh;
};

// CUBA = current-based synapses (reference examples/CUBA_2005.py:35-66: CUBA outputs, i_syn = (g_exc + g_inh) * syn_scale; the
// inhibitory weight is negative there) instead of conductance-based ones; everything else is the same update.
template <bool CUBA>
__global__ void __launch_bounds__(256) k_lif_coba_step(float* __restrict__ V, float* __restrict__ ge, float* __restrict__ gi,
                                                       float* __restrict__ refr, const float* __restrict__ in_exc,
                                                       const float* __restrict__ in_inh, uint8_t* __restrict__ spikes,
                                                       uint32_t* __restrict__ spike_bits, float* __restrict__ spike_count,
                                                       int64_t n, LifCobaP p) {
#pragma clang fp contract(off)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = V[i];
    // (in_scale_*: the inputs are synaptic COUNTS of a projection with weight 1 and the weight is applied here — one rounding, the
    //  same one `count * w` gets inside the scatter; 1.0 leaves a weighted input as it is, bit for bit)
    const float g_e = ge[i] * p.decay_exc + in_exc[i] * p.in_scale_exc;
    const float g_i = gi[i] * p.decay_inh + in_inh[i] * p.in_scale_inh;
    const float i_syn = CUBA ? (g_e + g_i) * p.syn_scale : (g_e * (p.e_exc - v) + g_i * (p.e_inh - v)) * p.syn_scale;
    const float dv = (-(v - p.v_rest) + i_syn + p.i_ext) * p.dt_over_tau;
    const float r = refr[i];
    const bool active = r <= 0.f;
    const float vn = active ? v + dv : v;
    const bool s = active && vn >= p.v_th;
    V[i] = s ? p.v_reset : vn;
    refr[i] = s ? p.t_ref : r - p.dt;
    ge[i] = g_e;
    gi[i] = g_i;
    if (spikes != nullptr) spikes[i] = s ? 1 : 0;
    if (spike_bits != nullptr) {       // the packed form the scatters and the spike exchange consume (BE_SPIKE_BITS): one word per 32 neurons
      // (lanes past the population have left the loop: the ballot reads them as 0; a word's first lane is inside whenever
      //  the word holds a neuron, the grid stride is a multiple of 64)
      const uint64_t bal = __ballot(s);
      if ((threadIdx.x & 31) == 0) spike_bits[i >> 5] = (uint32_t)(bal >> (threadIdx.x & 32));
    }
    if (spike_count != nullptr && s) spike_count[i] += 1.f;
  }
}

template <bool CUBA>
int lif_step(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
             uint8_t* spikes_out, uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m,
             double v_rest, double v_th, double v_reset, double t_ref, double e_exc, double e_inh,
             double decay_exc, double decay_inh, double i_ext, double syn_scale, double in_scale_exc, double in_scale_inh,
             be_stream_t stream) {
  BE_REQUIRE(n >= 0, BE_ERR_INVALID, "n < 0");
  if (n == 0) return BE_OK;
  BE_REQUIRE(v && g_exc && g_inh && refractory && in_exc && in_inh && (spikes_out || spike_bits_out), BE_ERR_INVALID, "null pointer");
  BE_REQUIRE(tau_m > 0. && dt > 0., BE_ERR_INVALID, "dt and tau_m must be positive");
  // derived constants in double, rounded to f32 once (what a host formulation with Python / C doubles hands to f32 arrays)
  const LifCobaP p{(float)dt, (float)(dt / tau_m), (float)v_rest, (float)v_th, (float)v_reset, (float)t_ref, (float)e_exc, (float)e_inh,
                   (float)decay_exc, (float)decay_inh, (float)i_ext, (float)syn_scale, (float)in_scale_exc, (float)in_scale_inh};
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_lif_coba_step<CUBA>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), v, g_exc,
                     g_inh, refractory, in_exc, in_inh, spikes_out, spike_bits_out, spike_count, n, p);
  BE_LAUNCH_CHECK();
  return BE_OK;
}

}  // namespace

extern "C" {

int be_lif_coba_step_packed(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                            uint8_t* spikes_out, uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m,
                            double v_rest, double v_th, double v_reset, double t_ref, double e_exc, double e_inh,
                            double decay_exc, double decay_inh, double i_ext, double syn_scale, be_stream_t stream) {
  return lif_step<false>(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, spike_bits_out, spike_count, n, dt, tau_m, v_rest,
                         v_th, v_reset, t_ref, e_exc, e_inh, decay_exc, decay_inh, i_ext, syn_scale, 1.0, 1.0, stream);
}

int be_lif_step_scaled_packed(int current_based, float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc,
                              const float* in_inh, double in_scale_exc, double in_scale_inh, uint8_t* spikes_out,
                              uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest,
                              double v_th, double v_reset, double t_ref, double e_exc, double e_inh, double decay_exc,
                              double decay_inh, double i_ext, double syn_scale, be_stream_t stream) {
  if (current_based)
    return lif_step<true>(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, spike_bits_out, spike_count, n, dt, tau_m, v_rest,
                          v_th, v_reset, t_ref, 0., 0., decay_exc, decay_inh, i_ext, syn_scale, in_scale_exc, in_scale_inh, stream);
  return lif_step<false>(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, spike_bits_out, spike_count, n, dt, tau_m, v_rest,
                         v_th, v_reset, t_ref, e_exc, e_inh, decay_exc, decay_inh, i_ext, syn_scale, in_scale_exc, in_scale_inh, stream);
}

int be_lif_cuba_step_packed(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                            uint8_t* spikes_out, uint32_t* spike_bits_out, float* spike_count, int64_t n, double dt, double tau_m,
                            double v_rest, double v_th, double v_reset, double t_ref, double decay_exc, double decay_inh, double i_ext,
                            double syn_scale, be_stream_t stream) {
  return lif_step<true>(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, spike_bits_out, spike_count, n, dt, tau_m, v_rest,
                        v_th, v_reset, t_ref, 0., 0., decay_exc, decay_inh, i_ext, syn_scale, 1.0, 1.0, stream);
}

int be_lif_cuba_step(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                     uint8_t* spikes_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest, double v_th,
                     double v_reset, double t_ref, double decay_exc, double decay_inh, double i_ext, double syn_scale,
                     be_stream_t stream) {
  BE_REQUIRE(n == 0 || spikes_out, BE_ERR_INVALID, "null pointer");
  return be_lif_cuba_step_packed(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, nullptr, spike_count, n, dt, tau_m, v_rest,
                                 v_th, v_reset, t_ref, decay_exc, decay_inh, i_ext, syn_scale, stream);
}

int be_lif_coba_step(float* v, float* g_exc, float* g_inh, float* refractory, const float* in_exc, const float* in_inh,
                     uint8_t* spikes_out, float* spike_count, int64_t n, double dt, double tau_m, double v_rest, double v_th,
                     double v_reset, double t_ref, double e_exc, double e_inh, double decay_exc, double decay_inh, double i_ext,
                     double syn_scale, be_stream_t stream) {
  BE_REQUIRE(n == 0 || spikes_out, BE_ERR_INVALID, "null pointer");
  return be_lif_coba_step_packed(v, g_exc, g_inh, refractory, in_exc, in_inh, spikes_out, nullptr, spike_count, n, dt, tau_m, v_rest,
                                 v_th, v_reset, t_ref, e_exc, e_inh, decay_exc, decay_inh, i_ext, syn_scale, stream);
}

}  // extern "C"
