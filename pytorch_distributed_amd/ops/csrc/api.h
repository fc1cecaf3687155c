// Host-callable C ABI of the kernel library, for code inside the library that drives
// kernels itself (the native inference loop, infer_server.hip).  Python binds the same
// symbols through ctypes (ops/nn.py, ops/dqn_engine.py, ops/mlp_engine.py).
#pragma once
#include "common.h"

// forward-only head for the actor server: q = W5 relu(h1pre + b4) + b5, then epsilon-greedy
struct ActHeadArgs {
    float* h1pre; const float* b4; const float* w5; const float* b5;
    const float* eps; float* out; int out_ld;        // out[b] = {action, q[action], max q}
    float* q_out; int B, HID, A; unsigned long long seed; const unsigned long long* counter;
};

struct StageJob { const float* src; long ld_src; int src_col; int ncols; bf16_t* dst; long ld_dst; int dst_row0; float scale; };
struct StageJobs { StageJob j[6]; };

PDA_EXPORT void pda_conv_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                const void* const* bias, void* const* out, int ngroups, int batch,
                                int H, int W, int C, int KH, int KW, int S, int O, float scale, int act,
                                hipStream_t stream);
PDA_EXPORT void pda_conv_fwd(const void* const* x, const void* const* w, const void* const* bias,
                             void* const* out, int ngroups, int batch, int H, int W, int C, int KH, int KW,
                             int S, int O, int act, hipStream_t stream);
PDA_EXPORT void pda_linear_fwd(const void* const* x, const void* const* w, const void* const* bias,
                               void* const* out, void* const* out32, int ngroups, int M, int N, int K,
                               long ldx, long ldw, long ldo, long ldo32, int act, int ksplit, hipStream_t stream);
// tile choice of the GEMM entry points (gemm.hip): 0 = 64 x 64, else the large-tile variant for this problem
PDA_EXPORT int pda_gemm_large_tiles(int M, int N, int units, int best);
PDA_EXPORT void pda_act_head(const ActHeadArgs* args, hipStream_t stream);
PDA_EXPORT void pda_tick(void* step, void* counter, hipStream_t stream);
PDA_EXPORT void pda_mlp_stage(const StageJobs* jobs, int njobs, const void* idx, int B, int max_cols,
                              hipStream_t stream);
PDA_EXPORT void pda_bf16_rows_to_f32(const void* in, long ld_in, void* out, long ld_out, int rows, int cols,
                                     hipStream_t stream);

// Nature-CNN kernel family (nature_*.hip, shared header nature_cnn.h)
PDA_EXPORT int pda_conv1_supported(int H, int W, int C, int KH, int KW, int S, int O);
PDA_EXPORT void pda_conv1_fwd_u8(const void* const* frames, const void* const* idx, const void* const* w,
                                 const void* const* bias, void* const* out, int ngroups, int batch, int H, int W,
                                 int C, float scale, int act, hipStream_t stream);
PDA_EXPORT void pda_conv1_fwd_u8_x3(const void* const* frames, const void* const* idx, const void* const* w,
                                    long w_plane, const void* const* bias, void* const* out, long out_plane,
                                    int ngroups, int batch, int H, int W, int C, float scale, int act,
                                    hipStream_t stream);
PDA_EXPORT int pda_generic_conv1_forced();
PDA_EXPORT int pda_cnn_fwd_stack_supported(int H, int W, int C);
PDA_EXPORT void pda_cnn_fwd_stack(const void* const* frames, const void* const* idx, const void* const* ref,
                                  long frame_stride, const void* const* w1,
                                  const void* const* b1, const void* const* w2, const void* const* b2,
                                  const void* const* w3, const void* const* b3, void* const* a1, void* const* a2,
                                  void* const* a3, int ngroups, int batch, int H, int W, int C, float scale,
                                  hipStream_t stream);
