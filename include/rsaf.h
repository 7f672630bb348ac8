/*
 * rsaf.h — C ABI of librsaf.so (gfx950 / MI355X).
 *
 * The reference (ayushpradhan-dev/robust-speech-analysis-framework) is pure Python and has no
 * FFI/plugin registry; its hot path is reached through four Python callables and one nn.Module
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds (ctypes) to
 * replace the third-party back-ends those callables drive.  Each entry point cites the reference
 * interface it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls enqueue work and
 *     return without synchronising unless stated otherwise;
 *   - return value: RSAF_OK (0) or an RSAF_ERR_* code; rsaf_last_error() gives the message of the
 *     calling thread's last failure; no exception crosses the boundary;
 *   - the caller owns every buffer; workspaces are sized with the matching *_workspace_bytes call;
 *   - plain C types only (no torch types).
 */
#ifndef RSAF_H
#define RSAF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSAF_OK 0
#define RSAF_ERR_ARG 1      /* bad argument / unsupported shape */
#define RSAF_ERR_HIP 2      /* a HIP runtime call failed */
#define RSAF_ERR_WORKSPACE 3 /* workspace too small */

#define RSAF_ABI_VERSION 7   /* 7: rsaf_gemm_f16x3 / rsaf_split_f16x2 / rsaf_f16x2_row_scales replace the bf16x6 entries (two fp16 planes with power-of-two row scales, three MFMA products).  6: params_host[18] of rsaf_mshds_pitch (per-depth Chebyshev tables behind sinc_cheb).  4: clip_info rows carry the sound's x1 / xmax (48 bytes); rsaf_resample_praat restates Sound_upsample for a rate ratio of 2.  5: the openSMILE-style chain is float64 end to end (lld / cand / functionals buffers are double) */

typedef void* rsaf_stream_t;

/* ---- library ------------------------------------------------------------------------------- */
int rsaf_abi_version(void);
const char* rsaf_last_error(void);
/* Upload the per-device constant tables (Hamming window, FFT twiddles, mel bank, DCT, ...).
 * Synchronous; called lazily by the first rsaf_smile_* call on a device otherwise. */
int rsaf_init_device(int device);

/* Per-kernel timing with HIP events on the launch stream (bench.py roofline object).
 * begin: start collecting; end: synchronise the recorded events and copy up to `cap` records.
 * A record accumulates all launches of one kernel family. */
typedef struct {
    char name[48];
    int64_t launches;
    double ms;        /* summed event time */
    double flops;     /* algorithmic FLOPs summed over launches (0 for byte-bound kernels) */
    double bytes;     /* algorithmic bytes summed over launches */
} rsaf_prof_record;
int rsaf_prof_begin(void);
int rsaf_prof_end(rsaf_prof_record* records_host, int cap, int* n_records_host);

/* ---- openSMILE-style chain ---------------------------------------------------------------------
 * Replaces the per-file SMILExtract subprocess of src/opensmile_extractor.py:62-75 driven by
 * Androids.conf:73-368 (cFramer .. cFunctionals), at the FILE'S OWN sample rate: cFramer's sizes are
 * seconds (Androids.conf:73-78) and SMILExtract analyses the file as it is.  Layout: all clips of a
 * batch (one sample rate per call) are concatenated in `wav` (float32 in [-1,1), mono); clip c occupies
 * samples [clip_off[c], clip_off[c+1]).  rsaf_smile_geometry gives frame = round(0.025 fs), hop =
 * round(0.010 fs) (floor(x + 0.5)) and the FFT length (next power of two: 256 / 512 / 1024 / 2048).
 * Frames: n_frames(n) = n < frame ? 0 : (n-frame)/hop + 1, frame_off is their exclusive prefix sum.
 * LLDs are written contour-major: lld[i * total_frames + frame_off[c] + t], i in [0, RSAF_SMILE_NLLD).
 * The whole chain computes, stores and hands over FLOAT64 (ABI 5): its decisions (peak enhancement, candidate ranking,
 * Viterbi path, jitter lags, roll-off crossings, maxPos / minPos) then coincide with the float64 CPU restatement. */
#define RSAF_SMILE_FRAME 400   /* at 16 kHz */
#define RSAF_SMILE_HOP 160     /* at 16 kHz */
#define RSAF_SMILE_NLLD 38
#define RSAF_SMILE_NFEAT 912
#define RSAF_SMILE_NCAND 6
int rsaf_smile_geometry(int sample_rate, int* frame_host, int* hop_host, int* nfft_host);
int64_t rsaf_smile_n_frames(int64_t n_samples, int sample_rate);
/* Androids.conf:73-186,258-280: framer, pre-emphasis, Hamming, FFT magnitude, mel/MFCC, RMS energy, ZCR,
 * intensity/loudness, 16 spectral descriptors (32 LLD rows), and cSpecScale + cPitchShs: per frame the
 * NCAND best sub-harmonic-summation candidates as (f0 Hz, voicing) double pairs, best score first, zeros in
 * empty slots -> cand[frame][NCAND][2] (caller-owned, 96 bytes per frame).  The six remaining LLD rows
 * (F0final, voicingFinalUnclipped, jitterLocal, jitterDDP, shimmerLocal, logHNR) are filled by
 * rsaf_smile_pitch_track.  octave_spectrum (may be NULL): the cSpecScale output ('hps' level of the config),
 * [total_frames][nfft/2 + 1] float64. */
int rsaf_smile_lld_batch(const float* wav, const int64_t* clip_off, const int64_t* frame_off,
                         int n_clips, int64_t max_clip_frames, int64_t total_frames, int sample_rate,
                         double* lld, double* cand, double* octave_spectrum, rsaf_stream_t stream);
/* Androids.conf:190-255: cPitchSmootherViterbi (fixed lag 30 frames) over the candidates, cValbasedSelector
 * (pitch zeroed where pcm_RMSenergy < 0.001: reads LLD row 0), cPitchJitter (waveform-matched periods on `wav`
 * driven by F0final) -> LLD rows 14, 15, 18..21.  back_workspace: 8 bytes per frame.
 * rsaf_smile_workspace_bytes(total_frames) covers cand + back_workspace when carved from one buffer. */
int64_t rsaf_smile_workspace_bytes(int64_t total_frames);
int rsaf_smile_pitch_track(const float* wav, const int64_t* clip_off, const int64_t* frame_off, int n_clips,
                           int64_t total_frames, int sample_rate, const double* cand, void* back_workspace,
                           double* lld, rsaf_stream_t stream);
/* Androids.conf:284-368 (sma3, delta regression W=2, 12 functionals).  window_frames = 0: statistics over the
 * whole clip; > 0: over the first window_frames frames of the full-length contours (the literal reading of
 * frameSize=0.025 / frameStep=0 at :355-356 keeps row 0 of a 3-frame-window output; see oracle/smile_oracle.py).
 * out: [n_clips, RSAF_SMILE_NFEAT] float64 in cCsvSink column order. */
int rsaf_smile_functionals(const double* lld, const int64_t* frame_off, int n_clips,
                           int64_t total_frames, int window_frames, double* out, rsaf_stream_t stream);

/* ---- exact-fp32 MFMA GEMM building block ---------------------------------------------------------
 * C[z][m][n] = act(alpha * sum_k A[z][m][k] * B[z](k,n) + bias[n] + R[z][m][n]).
 * Replaces the stock PyTorch ops the reference dispatches for every dense contraction of the path:
 * nn.Linear / nn.Conv1d inside transformers' Wav2Vec2Model (src/foundation_model_extractor.py:115)
 * and nn.Conv1d / the LSTM input projections of CNNLSTM (src/models.py:49-62,145-152).
 * B is [N,K] (b_kn = 0, K contiguous) or [K,N] (b_kn = 1).  z = z1*nz2 + z2 with the element strides
 * strides8_host = {sA1,sA2,sB1,sB2,sC1,sC2,sR1,sR2} (host array, may be NULL = all zero).
 * a_pad_k > 0: k=3/pad=1 convolution over a channels-last sequence read in place (A points one
 * row before the sequence, lda = Cin, K = 3*Cin, a_pad_k = Cin).  act: 0 none, 1 GELU(erf), 2 SiLU.
 * Requirements: A, B 16-byte aligned; lda, ldb, A/B strides multiples of 4; K % 4 == 0 (b_kn = 0)
 * or N % 4 == 0 (b_kn = 1); nz <= 65535. */
int rsaf_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* R,
                  int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr,
                  int nz, int nz2, const int64_t* strides8_host, int a_pad_k, int act, float alpha,
                  int b_kn, rsaf_stream_t stream);

/* ---- fp32-accurate GEMM on the fp16 matrix pipe: two-way operand splits, three products --------------------------
 * The same contraction and call sites as above (nn.Linear / nn.Conv1d inside Wav2Vec2Model,
 * src/foundation_model_extractor.py:115), from three v_mfma_f32_32x32x16_f16 products per term (csrc/gemm_f16x3.hip):
 * every operand row is multiplied by a power of two `scale[r]` that puts its largest magnitude into [2^14, 2^15) and
 * split into two fp16 planes, x * scale = hi + lo (11 + 11 significand bits); a b = (a_h b_h + a_h b_l + a_l b_h) /
 * (s_a s_b), fp32 accumulation.  Error against float64: at or below that of an fp32 FMA chain.
 *   rsaf_f16x2_row_scales: scale[r] from the exact maximum of row r of src[rows][K] (row stride ld); norm2 (may be NULL)
 *     receives the rows' Euclidean norms (the Cauchy-Schwarz bound a producer needs for a GEMM OUTPUT's scale).
 *   rsaf_split_f16x2: src * scale[r * scale_stride] -> planes (fp16 bit patterns, plane_stride elements apart), row-major
 *     [rows][K] or (panels != 0) k16 panels: element (r, k) at (k / 16) * (rows * 16) + r * 16 + k % 16.
 *   rsaf_gemm_f16x3: C (fp32, row stride ldc) and / or C_planes (the next GEMM's A, scaled by c_scale[m * c_scale_stride],
 *     which must keep |x| * c_scale below 65504), either may be NULL; a_scale[m * a_scale_stride], b_scale[n]: the scales
 *     the planes were built with; amax_out (may be NULL): atomicMax of the bit pattern of max |x| over everything
 *     written (zero it first).  Supported: {act 0, C} {act 0, C, R} {act 0, C_planes} {act 0, C, C_planes} {act 1 or 2, C}
 *     {act 1, C_planes}.  K % 16 == 0, N % 16 == 0.                                                                */
int rsaf_f16x2_row_scales(const float* src, int64_t rows, int K, int64_t ld, float* scale, float* norm2,
                          rsaf_stream_t stream);
int rsaf_split_f16x2(const float* src, int64_t rows, int K, int64_t ld, const float* scale, int scale_stride,
                     uint16_t* planes, int64_t plane_stride, int panels, rsaf_stream_t stream);
int rsaf_gemm_f16x3(const uint16_t* A_planes, int64_t a_plane_stride, const float* a_scale, int a_scale_stride,
                    const uint16_t* B_planes, int64_t b_plane_stride, const float* b_scale, float* C,
                    uint16_t* C_planes, int64_t c_plane_stride, const float* c_scale, int c_scale_stride,
                    uint32_t* amax_out, const float* bias, const float* R, int M, int N, int K, int64_t lda, int64_t ldb,
                    int64_t ldc, int64_t ldr, int act, float alpha, int a_panels, int b_panels, int c_panels,
                    rsaf_stream_t stream);

/* ---- CNN-LSTM-with-attention classifier forward ----------------------------------------------------
 * Replaces CNNLSTM.forward (src/models.py:161-193) in eval mode: x[B,T,input_dim] float32 ->
 * logits[B,num_classes].  Zero-padded frames are processed like any other frame (the reference's
 * collate_fn pads without a mask, src/dl_cv_strategies.py:81-84).
 * `weights`: one float32 device blob with eval-mode BatchNorm folded into the convolutions,
 * conv kernels stored tap-major ([Cout][tap*Cin + ci]), LSTM biases summed (b_ih + b_hh) and both
 * directions concatenated ([fwd 4H | reverse 4H] rows, gate order i,f,g,o).  Segment offsets (in
 * floats, 16-byte aligned) come from rsaf_cnnlstm_weight_offsets in this order:
 *   w1 b1 wsc bsc w2 b2 w3 b3 w4 b4  {wih_l bih_l whh_l} x lstm_layers  watt batt wfc bfc
 * (wsc/bsc = -1 when input_dim == channels: identity shortcut).
 * act: 1 = gelu, 2 = silu.  hidden must be 64 or 128, T >= 2, B <= 65535.                       */
int64_t rsaf_cnnlstm_weight_floats(int input_dim, int channels, int hidden, int num_classes, int lstm_layers);
int rsaf_cnnlstm_weight_offsets(int input_dim, int channels, int hidden, int num_classes, int lstm_layers,
                                int64_t* offsets_host, int cap, int* n_host);
int64_t rsaf_cnnlstm_workspace_bytes(int B, int T, int input_dim, int channels, int hidden, int lstm_layers);
int rsaf_cnnlstm_forward(const float* x, int B, int T, int input_dim, int channels, int hidden,
                         int num_classes, int lstm_layers, int act, const float* weights,
                         void* workspace, int64_t workspace_bytes, float* logits, rsaf_stream_t stream);
/* The same forward with the intermediate tensors the reference's sub-modules return copied out (any pointer may
 * be NULL): res1_out [B][T][C] = res_block1 output (src/models.py:175, channels-last), res2_out [B][T/2][C] =
 * res_block2 output (:178), lstm_out [B][T/2][2H] = nn.LSTM output (:184), pooled_out [B][2H] =
 * AttentionPooling output (:187). */
int rsaf_cnnlstm_forward_stages(const float* x, int B, int T, int input_dim, int channels, int hidden,
                                int num_classes, int lstm_layers, int act, const float* weights, void* workspace,
                                int64_t workspace_bytes, float* logits, float* res1_out, float* res2_out,
                                float* lstm_out, float* pooled_out, rsaf_stream_t stream);
/* ResidualBlock.forward (src/models.py:64-76) in eval mode on a channels-last sequence: x [B][T][Cin] ->
 * y [B][T][Cout] = act(BN2(conv2(act(BN1(conv1(x))))) + shortcut(x)), k = 3 / pad 1 / stride 1.  BatchNorm folded
 * into tap-major weights ([Cout][tap*Cin + ci]) as in the blob of rsaf_cnnlstm_forward; wsc/bsc = folded
 * conv1x1+BN shortcut, both NULL for the identity shortcut (Cin == Cout). */
int64_t rsaf_cnn_resblock_workspace_bytes(int B, int T, int out_channels);
int rsaf_cnn_resblock_forward(const float* x, int B, int T, int in_channels, int out_channels, int act,
                              const float* w1, const float* b1, const float* wsc, const float* bsc,
                              const float* w2, const float* b2, void* workspace, int64_t workspace_bytes,
                              float* y, rsaf_stream_t stream);
/* AttentionPooling.forward (src/models.py:94-107): seq [B][T][features] -> pooled [B][features] =
 * sum_t softmax_t(seq . watt + batt) * seq; features = 128 or 256. */
int rsaf_attnpool_forward(const float* seq, int B, int T, int features, const float* watt, const float* batt,
                          float* pooled, rsaf_stream_t stream);

/* ---- CNN-LSTM training step: forward in training mode + backward -------------------------------------
 * Replaces `out = model(seq)` under model.train() and the model part of `loss.backward()` in the reference's
 * training loops (src/dl_cv_strategies.py:118-125,241-243; module: src/models.py:64-76,161-193): BatchNorm1d on
 * batch statistics (biased variance over B*T rows; zero-padded frames count), dropout by caller-supplied masks
 * (float32 0 or 1/(1-p), NULL = no dropout): mask_block1 [B][T][C], mask_block2 [B][T/2][C],
 * mask_lstm_host = HOST array of lstm_layers-1 DEVICE pointers [B][T/2][2H] (or NULL), mask_fc [B][2H].
 * `params` / `grads`: float32 device blobs, segment offsets (floats) from rsaf_cnnlstm_train_param_offsets in
 * this order (conv kernels tap-major [Cout][tap][Cin]; g/be = BatchNorm weight/bias; b_l = b_ih + b_hh, both
 * directions stacked [fwd 4H | reverse 4H], gate order i,f,g,o):
 *   w1 b1 g1 be1  wsc bsc gsc besc  w2 b2 g2 be2  w3 b3 g3 be3  w4 b4 g4 be4  {wih_l b_l whh_l} x layers
 *   watt batt wfc bfc                                  (the four shortcut entries are -1 when input_dim == channels)
 * The gradient of b_l is the gradient of both b_ih and b_hh.
 * `saved` (rsaf_cnnlstm_train_saved_floats) carries the activations from forward to backward; `scratch`
 * (rsaf_cnnlstm_train_scratch_floats) may be reused between calls.  bn_stats_out (optional, device): [5][3][C]
 * mean / biased variance / rstd of bn1, shortcut BN, bn2 of block 1 and bn1, bn2 of block 2, for the caller's
 * running-statistics update.  backward consumes `saved` (the gates are overwritten): one backward per forward. */
int64_t rsaf_cnnlstm_train_param_floats(int input_dim, int channels, int hidden, int num_classes, int lstm_layers);
int rsaf_cnnlstm_train_param_offsets(int input_dim, int channels, int hidden, int num_classes, int lstm_layers,
                                     int64_t* offsets_host, int cap, int* n_host);
int64_t rsaf_cnnlstm_train_saved_floats(int B, int T, int input_dim, int channels, int hidden, int lstm_layers);
int64_t rsaf_cnnlstm_train_scratch_floats(int B, int T, int input_dim, int channels, int hidden, int lstm_layers);
int rsaf_cnnlstm_train_forward(const float* x, int B, int T, int input_dim, int channels, int hidden,
                               int num_classes, int lstm_layers, int act, const float* params,
                               const float* mask_block1, const float* mask_block2,
                               const float* const* mask_lstm_host, const float* mask_fc, float* saved,
                               int64_t saved_floats, float* scratch, int64_t scratch_floats, float* logits,
                               float* bn_stats_out, rsaf_stream_t stream);
int rsaf_cnnlstm_train_backward(const float* x, int B, int T, int input_dim, int channels, int hidden,
                                int num_classes, int lstm_layers, int act, const float* params,
                                const float* mask_block1, const float* mask_block2,
                                const float* const* mask_lstm_host, const float* mask_fc, float* saved,
                                int64_t saved_floats, float* scratch, int64_t scratch_floats,
                                const float* dlogits, float* grads, rsaf_stream_t stream);

/* ---- Wav2Vec2 frame embeddings for a batch of equal-length chunks -----------------------------------
 * Replaces, per chunk, `processor(chunk).input_values` + `Wav2Vec2Model(...)(input_values)
 * .last_hidden_state` (src/foundation_model_extractor.py:113-116; third-party transformers
 * modeling_wav2vec2.py / feature_extraction_wav2vec2.py:95), fp32 end to end.
 * Chunk i covers samples [chunk_start[i], chunk_start[i] + chunk_len) of `wav` (chunk_start: device
 * int64 array).  out: float32 rows of `hidden` floats; chunk i's T = rsaf_w2v2_frames(chunk_len) frames go
 * to rows out_row_start[i] .. +T (device int64 array; lets the caller lay chunks out in the
 * reference's np.vstack order, src/foundation_model_extractor.py:124) or, when out_row_start is
 * NULL, to rows i*T .. (i+1)*T.
 * `weights`: one float32 device blob; segment offsets (floats, 16-byte aligned) from
 * rsaf_w2v2_weight_offsets in this order:
 *   conv0[C][10] gn_gamma gn_beta conv1..conv6 ([C][tap*C+ci]) fp_ln_g fp_ln_b fp_w[H][C] fp_b
 *   pos_w ([G][C/G][tap*(H/G)+ci], weight norm folded) pos_b enc_ln_g enc_ln_b
 *   per layer: wqkv[3H][H] bqkv wo bo ln1_g ln1_b w1[I][H] b1 w2[H][I] b2 ln2_g ln2_b
 * Feature encoder geometry is wav2vec2's (kernels 10,3,3,3,3,2,2 / strides 5,2,2,2,2,2,2, GroupNorm
 * on layer 0, no conv bias, post-LN encoder).  n_chunks * max(heads, pos_groups) <= 65535.        */
int rsaf_w2v2_frames(int chunk_len);
int64_t rsaf_w2v2_weight_floats(int conv_dim, int hidden, int layers, int heads, int intermediate,
                                int pos_kernel, int pos_groups);
int rsaf_w2v2_weight_offsets(int conv_dim, int hidden, int layers, int heads, int intermediate,
                             int pos_kernel, int pos_groups, int64_t* offsets_host, int cap, int* n_host);
int64_t rsaf_w2v2_workspace_bytes(int n_chunks, int chunk_len, int conv_dim, int hidden, int layers,
                                  int heads, int intermediate, int pos_kernel, int pos_groups);
int rsaf_w2v2_forward(const float* wav, const int64_t* chunk_start, int n_chunks, int chunk_len,
                      int conv_dim, int hidden, int layers, int heads, int intermediate, int pos_kernel,
                      int pos_groups, float layer_norm_eps, const float* weights, void* workspace,
                      int64_t workspace_bytes, float* out, const int64_t* out_row_start,
                      rsaf_stream_t stream);
/* The same forward for windows of DIFFERENT lengths in one call (the reference's loop ends every file in a tail window of
 * its own length, src/foundation_model_extractor.py:103-108): chunk i has chunk_len[i] samples (device int32 array) and
 * T_i = rsaf_w2v2_frames(chunk_len[i]) frames, written to rows out_row_start[i] .. +T_i (or packed, window after window,
 * when out_row_start is NULL).  chunk_len_host: the same lengths on the host (launch geometry); they must be
 * NON-INCREASING.  Every window is normalised, GroupNorm-ed, zero-padded (positional conv) and attended over its own
 * frames only: the values are those of the per-window call.                                                          */
int64_t rsaf_w2v2_workspace_bytes_ragged(const int* chunk_len_host, int n_chunks, int conv_dim, int hidden, int layers,
                                         int heads, int intermediate, int pos_kernel, int pos_groups);
int rsaf_w2v2_forward_ragged(const float* wav, const int64_t* chunk_start, const int* chunk_len,
                             const int* chunk_len_host, int n_chunks, int conv_dim, int hidden, int layers, int heads,
                             int intermediate, int pos_kernel, int pos_groups, float layer_norm_eps,
                             const float* weights, void* workspace, int64_t workspace_bytes, float* out,
                             const int64_t* out_row_start, rsaf_stream_t stream);

/* ---- Praat-style analyses behind the MSHDS features (float64) ---------------------------------------
 * Replace the parselmouth/Praat calls of src/mshds_extractor.py: To Intensity (:41,198), To Pitch
 * (ac) (:104,143,178,270,355), To Harmonicity (cc) (:36,221), to_spectrogram + spectrum moments
 * (:356-369) and the statistics taken from them (:144-160,179-180,199-202,222,370-373).
 * `clip_info`: device array of n_clips records {int64 sample_off; int64 frame_off; double t1;
 * int32 n_samples; int32 n_frames; double x1; double xmax} (48 bytes) describing, for THIS analysis, where each clip's
 * samples start in `wav`, where its frames start in the per-frame output buffers, the time of its
 * first frame and its frame count (Praat's Sampled_shortTermAnalysis grid, computed by the host), and the
 * sound's own time axis as Praat's Sound object carries it: x1 = time of the first sample (dx / 2 for a sound read
 * from a file), [0, xmax] = its time domain (n dx for a file; after the Sound_resample of
 * src/mshds_extractor.py:418-419 the ORIGINAL duration, with the new grid centred in it).
 * Window tables (`window`, `window_r` = normalised autocorrelation of the window, `twiddle`) are
 * device float64 arrays prepared by the host.  All outputs are float64.                          */
int rsaf_mshds_frameout_doubles(void);   /* doubles per frame record: intensity, ncand, freq[16], strength[16] */
int rsaf_mshds_clip_peak(const float* wav, const void* clip_info, int n_clips, double* gpeak, rsaf_stream_t stream);
/* db_out[frame]; stats_out[clip][2] = {energy-mean dB, max/min of parabolic extrema} */
int rsaf_mshds_intensity(const float* wav, const void* clip_info, int n_clips, int max_frames,
                         const double* window, int half_window, double time_step, int subtract_mean,
                         double* db_out, double* stats_out, rsaf_stream_t stream);
/* params_host[18] = {dt, min_pitch, ceiling, voicing_thr, octave_cost, silence_thr, octave_jump_cost,
 *   voiced_unvoiced_cost, nsamp_window, nsamp_period, min_lag, max_lag, brent_ixmax, max_candidates,
 *   refine_depth, is_cc, dt_window, table_mode (0 | 1 | 2, see sinc_cheb)}.  sel_freq / sel_strength: the path finder's choice per frame.
 * stats_out[clip][8] = {n(f != 0), mean, population sd, mean after |z| <= 2, n voiced, mean Hz,
 *   sd in semitones (n-1), n after filter}.
 * sinc_cheb (may be NULL): [2 * refine_depth][16] Chebyshev coefficients on frac in [0, 1] of the sinc-interpolation
 *   weights of tap offsets -(depth-1) .. depth (mshds.sinc_cheb_table); with it the Brent refinement evaluates a
 *   16-term polynomial per step instead of the 2*depth-term sum whenever no candidate's depth is clipped.  With
 *   params_host[17] = 1 the table is followed by the tables of the clipped depths e = 1 .. refine_depth - 1
 *   ([2 e][16] each, depth e at offset 2 * refine_depth * 16 + 16 e (e - 1) doubles): candidates whose depth the array
 *   ends clip to e >= 3 (cross-correlation passes) then take the polynomial form too (ABI 6).
 *   With params_host[17] = 2 (single-threshold cross-correlation analyses; the harmonicity pass: depth 700) sinc_cheb holds
 *   ONE TABLE PER CELL instead (mshds.sinc_cell_tables): cells b = brent_ixmax + lag_lo - 1 .. brent_ixmax + lag_hi
 *   (lag_lo = max(min_lag, 2), lag_hi = min(max_lag, brent_ixmax) - 1), each [ntap_pad][16] with ntap_pad = max_lag + 1
 *   rounded up to a multiple of 4: row m = the sum of the rows of the cell's own depth min(depth, b + 1, RN - b - 1) that
 *   meet lag +m and lag -m of the symmetric correlation array; the coefficients of every cell of the batch are then built
 *   by a per-cell GEMM on the fp64 matrix pipe, clipped depths included (pitch_cell_coef_kernel).
 * workspace: per frame the correlation row (between the correlation kernel - fp64 FFT auto- / cross-correlation, one wave
 *   per frame - and the candidate kernel), the Chebyshev coefficient blocks of two candidate lists (2 x 4 KB) and a 128-byte
 *   record (between the candidate kernel and the refinement kernels: the Brent search runs one candidate per lane in its own
 *   kernel); at least rsaf_mshds_pitch_workspace_bytes_per_clip bytes, the clips are processed in groups of
 *   floor(workspace_bytes / that).  Environment RSAF_PITCH_INKERNEL=1: refinement inside the candidate kernel (the form
 *   before round 4; the tests' A/B reference). */
int64_t rsaf_mshds_pitch_workspace_bytes_per_clip(int max_frames, const double* params_host);
int rsaf_mshds_pitch(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* gpeak,
                     const double* window, const double* window_r, const double* params_host,
                     void* frame_out, unsigned char* psi, int* end_state, double* sel_freq, double* sel_strength,
                     double* stats_out, const double* sinc_cheb, void* workspace, int64_t workspace_bytes,
                     rsaf_stream_t stream);
/* The same analysis for two voicing thresholds at once (src/mshds_extractor.py:178 and :270 differ in nothing
 * else): one frame kernel computes the correlation and refines the union of the two candidate lists, the path
 * finder runs per threshold.  The *2 outputs have the shapes of their first-threshold counterparts. */
int rsaf_mshds_pitch_dual(const float* wav, const void* clip_info, int n_clips, int max_frames, const double* gpeak,
                          const double* window, const double* window_r, const double* params_host,
                          void* frame_out, unsigned char* psi, int* end_state, double* sel_freq, double* sel_strength,
                          double* stats_out, double voicing_threshold2, void* frame_out2, unsigned char* psi2,
                          int* end_state2, double* sel_freq2, double* sel_strength2, double* stats_out2,
                          const double* sinc_cheb, void* workspace, int64_t workspace_bytes, rsaf_stream_t stream);
/* _speechrate (src/mshds_extractor.py:11-125) from the 50 Hz / 16 ms intensity contour (rsaf_mshds_intensity)
 * and the 4-candidate pitch pass of :104.  out[clip][5] = Speaking_Rate, Articulation_Rate,
 * Phonation_Ratio, Pause_Rate, Mean_Pause_Dur.  workspace: n_clips * workspace_doubles(max_frames). */
int64_t rsaf_mshds_speechrate_workspace_doubles(int max_frames);
int rsaf_mshds_speechrate(const double* intensity_db, const void* clip_info, int n_clips, int max_frames,
                          double intensity_dt, const double* sel_freq, const void* pitch_clip_info,
                          double pitch_dt, double pitch_ceiling, double* workspace, double* out,
                          rsaf_stream_t stream);
/* _measureFormants (src/mshds_extractor.py:303-338): To Formant (burg) 5 ms / 5 formants / 5 kHz / 25 ms /
 * 50 Hz on a 10 kHz resampling of the clip, To Pitch (cc) (rsaf_mshds_pitch, is_cc), To PointProcess (cc),
 * then F1,B1,F2,B2 linearly interpolated at every pulse -> mean and sample SD.
 * resample_info: device array of {int64 sample_off; int64 out_off; double pos0; double x1o; int32 n_in;
 * int32 n_out; int32 table; int32 pad} (48 bytes).  The 10 kHz resampling is Praat's Sound_resample(10000, 500):
 * `lowpassed` = the clips after rsaf_praat_lowpass_batch (same layout as wav, float64); tables: [n_tables][5][table_stride]
 * float64 NUM_interpolate_sinc weights at full depth (tap k < 2*depth+1 belongs to input base + k - depth; zeros up to
 * table_stride >= rsaf_mshds_resample10k_table_stride(depth)); phase_base:
 * [n_tables][5] int32 = floor(pos0 + 1.6 r) (output m = 5q + r reads input 8q + base); outputs whose depth Praat
 * clips at the ends of the sound are evaluated directly.
 * frames_out: per frame {double f[5]; double b[5]} (NaN padded); pulses: [n_clips][max_pulses] times (unsorted);
 * stats out[clip][8] = mean/SD of F1, B1, F2, B2 in the reference's column order. */
int rsaf_mshds_resample10k_table_stride(int depth);
int rsaf_mshds_resample10k(const double* lowpassed, const void* resample_info, int n_clips, int max_out,
                           const double* tables, int table_stride, const int* phase_base, int depth, double* out,
                           rsaf_stream_t stream);
int rsaf_mshds_formants(const double* y10, const void* resample_info, const void* clip_info, int n_clips,
                        int max_frames, const double* window, int nsamp_window, double time_step, double dx_out,
                        double preemph_factor, void* frames_out, rsaf_stream_t stream);
/* pulses[clip][max_pulses] in ascending time, n_pulses[clip]; workspace sized by
 * rsaf_mshds_pulses_workspace_bytes (stretch table and per-stretch pulse slots). */
int64_t rsaf_mshds_pulses_workspace_bytes(int n_clips, int max_frames, double pitch_dt, double pitch_ceiling);
int rsaf_mshds_pulses(const float* wav, const void* pitch_clip_info, int n_clips, int max_frames, const double* sel_freq,
                      double pitch_dt, double pitch_ceiling, void* workspace, int64_t workspace_bytes, double* pulses,
                      int max_pulses, int* n_pulses, rsaf_stream_t stream);
/* Ltas (pitch-corrected) 0-5000 Hz in 100 Hz bands from the time-sorted pulses of rsaf_mshds_pulses, then
 * "Get slope" (50-1000 vs 1000-4000 Hz, dB) and the slope of the robust line fit over 100-5000 Hz:
 * out[clip][2] = {Spectral_Slope, Spectral_Tilt}, NaN where Praat raises.
 * Replaces call(snd, "To Ltas (pitch-corrected)...") + "Get slope" + "Report spectral tilt",
 * src/mshds_extractor.py:239-249.  clip_info rows need sample_off and n_samples only. */
int rsaf_mshds_ltas_slope_tilt(const float* wav, const void* clip_info, int n_clips, const double* pulses,
                               int max_pulses, const int* n_pulses, double shortest_period, double longest_period,
                               double max_period_factor, double* out, rsaf_stream_t stream);
/* Mean cepstral peak prominence (CPPS) of the voiced stretches: out[clip], NaN when no stretch exceeds 4 dB or
 * Praat would raise.  Voiced intervals from the time-sorted pulses ("To TextGrid (vuv)" 0.02 0.1, times rounded
 * to 6 decimals like "Down to Table"), each extracted, resampled to 10 kHz as Sound_resample (10000, 50) does it (FFT
 * low-pass of the extracted part over its own power of two, NUM_interpolate_sinc), "To PowerCepstrogram" 60 0.002 5000 50,
 * "Get CPPS" no 0.01 0.001 60 330 0.05 parabolic 0.001 0 Straight Robust.
 * Replaces src/mshds_extractor.py:272-297.  Workspaces (caller-owned, per clip): seg_table
 * [max_seg][rsaf_mshds_cpp_seg_doubles()] doubles, hdr [4] ints, resampled [cap_res], cepstrogram
 * [cap_frames][513], cpp_frames [cap_frames] doubles, lp_work [cap_work] complex doubles (cap_work >= longest clip +
 * 2000 max_seg covers every case); lowpassed: float64 scratch for the samples wav[lp_origin ...] of the clips of this
 * call (sample s of the batch at lowpassed[s - lp_origin]); lg_max = log2 of the first power of two >= longest clip +
 * 2000.  window1000 = the 1000-point Gaussian window of Sound_createGaussian, twiddle1024[k] = (cos, -sin)(2 pi k / 1024),
 * k < 512. */
int rsaf_mshds_cpp_seg_doubles(void);
int rsaf_mshds_cpp(const float* wav, const void* clip_info, int n_clips, const double* pulses, int max_pulses,
                   const int* n_pulses, const double* window1000, const double* twiddle1024, int max_seg, int cap_res,
                   int cap_frames, void* seg_table, int* hdr, double* resampled, double* cepstrogram, double* cpp_frames,
                   double* lowpassed, int64_t lp_origin, void* lp_work, int64_t cap_work, int lg_max, double* out,
                   rsaf_stream_t stream);
int rsaf_mshds_formant_stats(const void* frames, const void* clip_info, int n_clips, double time_step,
                             const double* pulses, int max_pulses, const int* n_pulses, double* out,
                             rsaf_stream_t stream);
int rsaf_mshds_hnr_mean(const double* sel_freq, const double* sel_strength, const void* clip_info, int n_clips,
                        double* out, rsaf_stream_t stream);
/* moments[frame][5] = {gate, CoG, SD, skewness, kurtosis}; stats_out[clip][4] = means over gated frames */
int rsaf_mshds_spectral_moments(const float* wav, const void* clip_info, const void* pitch_clip_info, int n_clips,
                                int max_frames, const double* sel_freq, double pitch_dt, double ceiling,
                                const double* window, const double* twiddle, int nsamp_window, int nfft,
                                int nbins, double time_step, double freq_step, double* moments,
                                double* stats_out, rsaf_stream_t stream);

/* ---- sample-rate conversion in front of the extractors ---------------------------------- */
/* Interleaved little-endian integer PCM (1 = unsigned 8 bit, 2, 3, 4 bytes per sample) -> float32 in [-1, 1),
 * channel mean in float32.  Replaces torchaudio.load + waveform.mean(dim=0), src/foundation_model_extractor.py:87-91
 * (and the mono conversion of parselmouth.Sound / cWaveSource monoMixdown). */
int rsaf_pcm_to_mono_f32(const void* pcm, int sample_width, int n_channels, int64_t n_frames, float* out,
                         rsaf_stream_t stream);
/* torchaudio.transforms.Resample(orig, new) with its defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff
 * 0.99): out[i * n_phase + p] = sum_k taps[p][k] * in[i * orig + tap_start[p] + k] (zero outside the input), with
 * orig/new already divided by their gcd, n_phase = new, n_out = ceil(new * n_in / orig).  The host builds the
 * taps (resample.sinc_hann_taps).  Replaces src/foundation_model_extractor.py:93-94. */
int rsaf_resample_sinc_hann(const float* in, int64_t n_in, const float* taps, const int* tap_start, int n_phase, int orig,
                            int taps_per_phase, float* out, int64_t n_out, rsaf_stream_t stream);
/* Praat Sound.resample(fs_out, precision) as Praat does it (published Sound_resample): when the rate goes down, a
 * brick-wall low-pass of the whole sound by a real FFT over the first power of two >= n_in + 2000 samples (bins from
 * floor(fs_out / fs_in * nfft) in Praat's packed order cleared), then NUM_interpolate_sinc(precision) on the new sample
 * grid centred in the old time domain; n_out = round(n_in / fs_in * fs_out).  `work`: device scratch of at least
 * rsaf_resample_praat_work_bytes(...) bytes (0 when the rate goes up: may be NULL).  Sounds of up to 2^24 - 2000
 * samples.  Not restated: the special case of a ratio of exactly 2 (Sound_upsample), which takes the general branch.
 * Replaces snd.resample(16000, 50), src/mshds_extractor.py:419. */
int64_t rsaf_resample_praat_work_bytes(int64_t n_in, double fs_in, double fs_out);
/* The low-pass step alone over a batch of sounds (the 10 kHz resampling inside To Formant (burg)): sigs = device array of
 * {int64 in_off; int64 out_off; int64 work_off; int32 n; int32 lg} (32 bytes): n float samples at in + in_off ->
 * n float64 samples at out + out_off, through 2^(lg-1) complex numbers at work + work_off (2^lg = first power of two
 * >= n + 2000, 11 <= lg <= lg_max <= 24); upfactor = new rate * old sample period < 1; work_complex = complex numbers
 * in `work`. */
/* Longest sound (samples) the whole-sound FFT low-pass takes: 2^26 - 2000 (25 min at 44.1 kHz, 69 min at 16 kHz); the
 * transform runs as two LDS passes of at most 8192 x 4096 complex points.  Longer input: rsaf_resample_praat /
 * rsaf_praat_lowpass_batch return RSAF_ERR_ARG; the MSHDS drop-in then gives NaN for the formant columns of that clip only
 * (16 kHz input) or the reference's per-file NaN row (other rates). */
int64_t rsaf_praat_lowpass_max_samples(void);
int rsaf_praat_lowpass_batch(const float* in, const void* sigs, int n_sigs, int lg_max, double upfactor, void* work,
                             int64_t work_complex, double* out, rsaf_stream_t stream);
int rsaf_resample_praat(const float* in, int64_t n_in, double fs_in, double fs_out, int precision, float* out,
                        int64_t n_out, void* work, int64_t work_bytes, rsaf_stream_t stream);

/* ---- session aggregation and batch assembly behind the extractors ---------------------- */
/* out[seg][col][2] = {mean, sample standard deviation (n-1)} over the rows row_index[seg_off[seg] .. seg_off[seg+1])
 * of rows[.][ld], NaN skipped, NaN when fewer than 1 / 2 values remain.
 * Replaces merged_df.groupby('unique_participant_id').agg(['mean', 'std']), src/utils.py:49. */
int rsaf_segment_mean_std(const double* rows, int64_t ld, const int* row_index, const int* seg_off, int n_seg, int width,
                          double* out, rsaf_stream_t stream);
/* dst[r][0..width) = src[src_row[r]][0..width), zeros where src_row[r] < 0.
 * Replaces np.vstack(participant_sequences), src/utils.py:96, and the zero padding of collate_fn,
 * src/dl_cv_strategies.py:81-84. */
int rsaf_gather_rows_f32(const float* src, int64_t ld_src, const int64_t* src_row, int64_t n_rows, int width, float* dst,
                         int64_t ld_dst, rsaf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RSAF_H */
