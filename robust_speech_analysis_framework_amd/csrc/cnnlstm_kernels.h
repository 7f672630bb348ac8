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

}  // namespace rsaf
