// The GRU cell arithmetic shared by the launch-per-step and the persistent bf16-operand kernels (gru.hip, gru_persist.hip), with
// floating-point contraction pinned off so that both evaluate bit-identical expressions whatever code surrounds the call:
// the persistent kernel's hand-off protocol is tested by exact equality against the per-step kernels.
// Gate order and formulas: torch.nn.GRU as used by the reference (morgana/utils.py:345-393).  Throughput (bf16) mode only: the
// sigmoid / tanh are the v_exp_f32 + v_rcp_f32 forms (about 2e-7 absolute error, far below the bf16 operand rounding of this
// mode); the fp32 parity kernels in gru.hip keep expf / tanhf.
#pragma once

#include "common.h"

struct mg_gru_cell_out {
    float r, z, n, hnew;
};

__device__ __forceinline__ mg_gru_cell_out mg_gru_cell(float xr, float xz, float xn, float hr, float hz, float hn, float hprev) {
#pragma clang fp contract(off)
    mg_gru_cell_out o;
    o.r = mg_sigmoid_fast(xr + hr);
    o.z = mg_sigmoid_fast(xz + hz);
    o.n = 2.f * mg_sigmoid_fast(2.f * (xn + o.r * hn)) - 1.f;      // tanh; saturates cleanly (exp2 -> inf -> rcp -> 0)
    o.hnew = (1.f - o.z) * o.n + o.z * hprev;
    return o;
}

// sum of the 4 waves' partial pre-activations plus the bias, in a fixed order
__device__ __forceinline__ float mg_gru_sum4(float a, float b, float c, float d, float bias) {
#pragma clang fp contract(off)
    return ((a + b) + (c + d)) + bias;
}

struct mg_gru_cell_grad {
    float dr, dz, dn, dnr, carry;
};

// dstate = carry-in + the 4 waves' partial sums of dhproj_{t+1} W_hh, in a fixed order
__device__ __forceinline__ float mg_gru_dstate(float carry_in, float a, float b, float c, float d) {
#pragma clang fp contract(off)
    return carry_in + ((a + b) + (c + d));
}

// gradient of one active GRU step: dstate = d loss / d h_t from the later steps, gout = d loss / d output_t
__device__ __forceinline__ mg_gru_cell_grad mg_gru_cell_bwd(float dstate, float gout, float r, float z, float n, float hn, float hprev) {
#pragma clang fp contract(off)
    mg_gru_cell_grad g;
    const float dh = dstate + gout;
    g.dn = dh * (1.f - z) * (1.f - n * n);
    g.dz = dh * (hprev - n) * z * (1.f - z);
    g.dr = g.dn * hn * r * (1.f - r);
    g.dnr = g.dn * r;
    g.carry = dh * z;
    return g;
}

// The exact-fp32 cell of the parity-mode kernels (expf / tanhf), shared by the launch-per-step and the persistent fp32 kernels
// with contraction pinned off for the same reason as above.
__device__ __forceinline__ mg_gru_cell_out mg_gru_cell_exact(float xr, float xz, float xn, float hr, float hz, float hn, float hprev) {
#pragma clang fp contract(off)
    mg_gru_cell_out o;
    o.r = mg_sigmoid(xr + hr);
    o.z = mg_sigmoid(xz + hz);
    o.n = tanhf(xn + o.r * hn);
    o.hnew = (1.f - o.z) * o.n + o.z * hprev;
    return o;
}
