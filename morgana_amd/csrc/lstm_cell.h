// The exact-fp32 LSTM cell arithmetic shared by the launch-per-step kernels (lstm.hip) and the persistent fp32 kernels
// (lstm_persist_f32.hip), with floating-point contraction pinned off so that both evaluate bit-identical expressions whatever code
// surrounds the call: the persistent kernels' hand-off protocol is tested by exact equality against the per-step kernels.
// Gate order (i, f, g, o) and formulas: torch.nn.LSTM as the reference wraps it (morgana/utils.py:345-393, models/RNN_SPSS.py:36-37).
#pragma once

#include "common.h"

// pre-activation of one gate: input projection + (the 4 waves' partial sums of h_{t-1} W_hh^T in a fixed order + bias)
__device__ __forceinline__ float mg_lstm_pre(float x, float a, float b, float c, float d, float bias) {
#pragma clang fp contract(off)
    return x + (((a + b) + (c + d)) + bias);
}

struct mg_lstm_cell_out {
    float i, f, g, o, c, h;
};

__device__ __forceinline__ mg_lstm_cell_out mg_lstm_cell_exact(float pre_i, float pre_f, float pre_g, float pre_o, float cprev) {
#pragma clang fp contract(off)
    mg_lstm_cell_out r;
    r.i = mg_sigmoid(pre_i);
    r.f = mg_sigmoid(pre_f);
    r.g = tanhf(pre_g);
    r.o = mg_sigmoid(pre_o);
    r.c = r.f * cprev + r.i * r.g;
    r.h = r.o * tanhf(r.c);
    return r;
}

// d loss / d h_t from the later steps: carry-in + the 4 waves' partial sums of dgates_{t+1} W_hh, in a fixed order
__device__ __forceinline__ float mg_lstm_dstate(float carry_in, float a, float b, float c, float d) {
#pragma clang fp contract(off)
    return carry_in + ((a + b) + (c + d));
}

struct mg_lstm_cell_grad {
    float di, df, dg, d_o, cc;
};

// gradient of one ACTIVE step: dh_state / dc_state = d loss / d (h_t, c_t) from the later steps, gout = d loss / d output_t;
// cc = the elementwise part of d loss / d c_{t-1} (all of d loss / d h_{t-1} goes through the matmul with these gate gradients)
__device__ __forceinline__ mg_lstm_cell_grad mg_lstm_cell_bwd(float dh_state, float dc_state, float gout, float s_i, float s_f, float s_g,
                                                              float s_o, float c_prev, float c_new) {
#pragma clang fp contract(off)
    mg_lstm_cell_grad r;
    const float dh = dh_state + gout;
    const float tc = tanhf(c_new);
    const float dc = dc_state + dh * s_o * (1.f - tc * tc);
    r.di = dc * s_g * s_i * (1.f - s_i);
    r.df = dc * c_prev * s_f * (1.f - s_f);
    r.dg = dc * s_i * (1.f - s_g * s_g);
    r.d_o = dh * tc * s_o * (1.f - s_o);
    r.cc = dc * s_f;
    return r;
}
