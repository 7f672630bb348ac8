// Kernels of the CNN-LSTM classifier shared by the inference path (cnnlstm.hip) and the training step
// (cnnlstm_train.hip).
#pragma once
#include "rsaf_common.h"

namespace rsaf {

// Persistent bidirectional LSTM recurrence over xproj[B][T][2][4H] (input projections + bias, gate order i,f,g,o)
// with whh[2][4H][H]; hout[B][T][2H].  With gates_save/c_save (training) the post-activation gates are stored in the
// xproj layout (gates_save may alias xproj) and the cell states in the hout layout.
int launch_lstm_rec(const float* xproj, const float* whh, float* hout, float* gates_save, float* c_save, int B, int T,
                    int H, hipStream_t s);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. waits for
// the write acknowledgements of the per-step global stores of the recurrences - a few hundred cycles on every time step
// for data no other wave of the workgroup reads.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0); vmcnt / expcnt left alone
    __builtin_amdgcn_s_barrier();
}

// Drain the vector-memory counter once in front of a persistent loop: the waitcnt pass merges the loop pre-header with
// the back edge, and loads still pending from the pre-header would otherwise force vmcnt(0) in every iteration.
__device__ __forceinline__ void vmem_drain() { __builtin_amdgcn_s_waitcnt(0x0f70); }   // vmcnt(0) only

}  // namespace rsaf
