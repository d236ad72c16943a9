/*
 * hupr.h — C ABI of the MI355X (gfx950) radar->pose hot path.
 *
 * The reference (robert80203/HuPR-...) is pure Python with no native layer, so there is no
 * FFI in it to mirror symbol-for-symbol.  Each entry point below therefore cites the
 * reference *Python* interface it replaces (file:line relative to the reference root); the
 * Python host in hupr-..._amd/ binds these with ctypes (runtime.py) and re-exposes the
 * reference's own names (RadarObject.generateHeatmap, Normalize, HuPRNet.forward, ...).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; the caller owns every buffer
 *   - stream-ordered on `stream` (a hipStream_t passed as void*); never synchronises,
 *     never allocates; scratch comes from the caller (`ws`, `ws_bytes`)
 *   - returns 0 on success, a negative HUPR_ERR_* otherwise; hupr_last_error() gives a
 *     thread-local message
 */
#ifndef HUPR_H
#define HUPR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HUPR_OK 0
#define HUPR_ERR_ARG (-1)       /* bad argument (null pointer, unsupported shape)   */
#define HUPR_ERR_WORKSPACE (-2) /* workspace too small                               */
#define HUPR_ERR_LAUNCH (-3)    /* hipLaunch / runtime error                         */
#define HUPR_ERR_COMM (-4)      /* librccl missing / RCCL returned an error          */

typedef void* hupr_stream_t; /* hipStream_t */

int hupr_version(void);
const char* hupr_last_error(void);
/* kernels launched by this library in this process so far (every launch goes through one counting macro, csrc/hupr_common.h):
 * bench.py prints the difference over a step as `launches_per_step` — measurement aid, nothing in the reference to replace */
unsigned long long hupr_launch_count(void);

/* ------------------------------------------------------------------------------------------
 * (a1) FFT chain — replaces RadarObject.generateHeatmap, preprocessing/process_iwr1843.py:106-173
 *      (+ clutterRemoval :85-104, postProcessFFT3D :48-52).
 *
 * adc_iq : int16  [n_sf][4 rx][192 chirp][256 sample][2 (I,Q)]      786 432 B / sensor-frame
 * out    : float2 [n_sf][16 doppler][64 range][64 az][8 el]          complex64, 4 194 304 B
 * Window = rectangular, output = complex (the reference applies no window / magnitude).
 * ws must hold hupr_fft_chain_ws_bytes(n_sf) bytes.
 * ---------------------------------------------------------------------------------------- */
size_t hupr_fft_chain_ws_bytes(int n_sf);
int hupr_fft_chain_c64(const int16_t* adc_iq, int n_sf, void* out_c64, void* ws, size_t ws_bytes,
                       hupr_stream_t stream);

/* (a1+a2) FFT chain fused with the loader glue — replaces generateHeatmap + np.save/np.load +
 *      datasets/dataset.py:144-150 (keep Doppler 4..11, re/im split) + Normalize
 *      (datasets/base.py:13-24; per elevation channel (x-mean)/std, unbiased std).
 * out    : float  [n_sf][8 f][2 re/im][64 range][64 az][8 el]        2 097 152 B / sensor-frame
 *          == one (F,2,R,A,E) group-frame of HuPRNet's input. */
int hupr_fft_chain_loader_f32(const int16_t* adc_iq, int n_sf, float* out, void* ws, size_t ws_bytes,
                              hupr_stream_t stream);

/* (a1+a2+a3 seam) the same, with HuPRNet's elevation mean (models/networks.py:26-27, the first thing forward_chirp does to the
 *      loader tensor) folded in: means[n_sf][16 = 2 f + c][64 range][64 az] = mean over the 8 elevation bins of the normalised
 *      plane — 262 144 B / sensor-frame instead of 2 097 152 B (1 048 576 B algorithmic incl. the ADC read).
 *      Consumed by hupr_mnet_fwd_means_*; bit-identical to hupr_fft_chain_loader_f32 followed by hupr_mnet_fwd_*. */
int hupr_fft_chain_loader_means_f32(const int16_t* adc_iq, int n_sf, float* means, void* ws, size_t ws_bytes,
                                    hupr_stream_t stream);

/* (a1, opt-in variants) north_star asks for "Hanning windowing and magnitude"; the reference has neither
 *      (process_iwr1843.py:130-151 is bare np.fft.fft2 / np.fft.fft, np.abs only in the plotting helper :207-208), so
 *      both are flags that are OFF on every parity path:
 *        HUPR_FFT_HANN_RANGE    x[c,s] *= hann256[s]            before the range FFT   (np.hanning(256), symmetric)
 *        HUPR_FFT_HANN_DOPPLER  (x - mean_c x)[c] *= hann64[c]  before the Doppler FFT (after clutter removal :122-128)
 *        HUPR_FFT_MAGNITUDE     out = |X| as float [n_sf][16][64][64][8] (2 097 152 B) instead of complex64
 *      loader: 0 = complex cube, 1 = the fused loader epilogue of hupr_fft_chain_loader_f32, 2 = the elevation-mean planes of
 *      hupr_fft_chain_loader_means_f32 (window / zero-Doppler / order flags only).
 *      flags == 0 is exactly hupr_fft_chain_c64 / hupr_fft_chain_loader_f32 / hupr_fft_chain_loader_means_f32.
 *
 *      The zero-Doppler bin (Doppler index 8; the loader's slot f = 4).  Clutter removal (process_iwr1843.py:122-128) cancels
 *      it analytically; the reference keeps the fp64 rounding residue of its fft2 there (~2e-16 of the other bins, white, a
 *      pure function of the frame) and its Normalize (datasets/base.py:17-24) inflates that plane to a unit-variance input
 *      channel — an exactly-zero plane would be 0/0 = NaN in that Normalize.  Default: the bin carries a frame-keyed dither
 *      with the reference residue's statistics (integer hash of the exact chirp sums, 2^-53 of an ADC LSB per sample, restated
 *      bit for bit by oracle/fft_chain.py::zero_doppler_dither), transformed and normalised like every other bin.
 *        HUPR_FFT_ZERO_DOPPLER_EXACT  the bin is exactly 0 (round 3's behaviour; the loader epilogues then emit zeros in f = 4)
 *        HUPR_FFT_RANGE_FIRST         the range-first order of rounds 1-2 (the reference's order of operations in fp32: the bin
 *                                     holds that order's own fp32 rounding residue, ~1e-7 relative); slower (2.6 vs 4 TB/s) */
#define HUPR_FFT_HANN_RANGE 1
#define HUPR_FFT_HANN_DOPPLER 2
#define HUPR_FFT_MAGNITUDE 4
#define HUPR_FFT_ZERO_DOPPLER_EXACT 8
#define HUPR_FFT_RANGE_FIRST 16
int hupr_fft_chain_opts(const int16_t* adc_iq, int n_sf, void* out, int flags, int loader, void* ws, size_t ws_bytes,
                        hupr_stream_t stream);

/* (a2) loader glue alone on a precomputed cube (the .npy hand-off of the reference):
 * cube_c64 : float2 [n_sf][16][64][64][8]  ->  out float [n_sf][8][2][64][64][8]            */
int hupr_loader_normalize_c64(const void* cube_c64, int n_sf, float* out, hupr_stream_t stream);

/* (f1) DCA1000 raw capture -> ADC layout above — replaces RadarObject.getadcDataFromDCA1000,
 *      preprocessing/process_iwr1843.py:54-83 (2-lane LVDS groups [I0,I1,Q0,Q1]; per chirp [rx0][rx1][rx2][rx3] x 256).
 * raw : int16 stream of n_frames * 786 432 / 2 values;  adc_iq: int16 [n_frames][4][192][256][2]. */
int hupr_dca1000_deinterleave(const int16_t* raw, int16_t* adc_iq, int n_frames, hupr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * HuPRNet operators.  Activations are CHANNELS-LAST fp32: x[b][d][h][w][c] ("voxel stride" =
 * floats between consecutive voxels, >= c, lets an op read/write a channel slice of a wider
 * concat buffer).  Parameters keep the reference's state_dict shapes.
 * ---------------------------------------------------------------------------------------- */

/* Generic batched fp32 MFMA GEMM, row-major.  ta/tb: 0 = as stored, 1 = transposed.
 *   (ta,tb) = (0,1): C[m][n] = sum_k A[m][k] B[n][k]      attention S = Q K^T, dP = dO V^T   (models/layers.py:129)
 *   (0,0):          C[m][n] = sum_k A[m][k] B[k][n]      O = P V (:131), dQ = dS K, PRGCN W.x  (gcn_networks.py:25)
 *   (1,0):          C[m][n] = sum_k A[k][m] B[k][n]      dV = P^T dO, dK = dS^T Q
 * res (optional) is added in the epilogue (cross-attention residual, models/layers.py:146,148).
 * accumulate != 0: C += result. */
int hupr_gemm_f32(int ta, int tb, const float* A, const float* B, float* C, int M, int N, int K, long lda,
                  long ldb, long ldc, int batch, long a_batch_stride, long b_batch_stride, long c_batch_stride,
                  const float* res, long res_ld, long res_batch_stride, int accumulate, hupr_stream_t stream);

/* (a4,a6) stride-1 convolution as implicit GEMM — replaces nn.Conv3d / nn.Conv2d forward and, with
 * weights packed in mode 1, their input-gradient (models/layers.py:10-23,42-62,81-95,115-123,194-210).
 * x : [Bn][Di][Hi][Wi] voxels, in_ld floats apart, first Ci channels used (Ci % 32 == 0)
 * wp: packed weights [Co][kd*kh*kw][Ci]   (hupr_pack_conv_weights_f32)
 * y : [Bn][Do][Ho][Wo] voxels, out_ld floats apart, first Co channels written
 * bias [Co] and res (same voxel indexing, res_ld) optional. */
int hupr_conv_fwd_f32(const float* x, const float* wp, const float* bias, const float* res, float* y, int Bn,
                      int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int out_ld,
                      int res_ld, int kd, int kh, int kw, int pd, int ph, int pw, int accumulate,
                      hupr_stream_t stream);

/* weight gradient of the same convolution: dw in the PARAMETER layout (Co, Ci, kd, kh, kw). */
size_t hupr_conv_wgrad_ws_bytes(int Bn, int Do, int Ho, int Wo, int Ci, int Co, int kd, int kh, int kw);
int hupr_conv_wgrad_f32(const float* x, const float* dy, float* dw, int Bn, int Di, int Hi, int Wi, int Ci,
                        int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd, int kh, int kw, int pd,
                        int ph, int pw, void* ws, size_t ws_bytes, hupr_stream_t stream);

/* w (Co, Ci, taps) parameter layout -> mode 0: [Co][taps][Ci] (forward), mode 1: [Ci][taps reversed][Co] (dgrad) */
int hupr_pack_conv_weights_f32(const float* w, float* wp, int Co, int Ci, int taps, int mode, hupr_stream_t stream);

/* bf16-MFMA variants of the three entry points above: identical signatures and fp32 tensors in HBM;
 * operands are rounded to bf16 while staged into LDS, products accumulate in fp32 (16x the matrix rate).
 * The fp32 entry points remain the parity path (north_star: fp32 within 1e-3; bf16 for configs C2/C4). */
int hupr_gemm_bf16(int ta, int tb, const float* A, const float* B, float* C, int M, int N, int K, long lda,
                   long ldb, long ldc, int batch, long a_batch_stride, long b_batch_stride, long c_batch_stride,
                   const float* res, long res_ld, long res_batch_stride, int accumulate, hupr_stream_t stream);
int hupr_conv_fwd_bf16(const float* x, const float* wp, const float* bias, const float* res, float* y, int Bn,
                       int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int out_ld,
                       int res_ld, int kd, int kh, int kw, int pd, int ph, int pw, int accumulate,
                       hupr_stream_t stream);
int hupr_conv_wgrad_bf16(const float* x, const float* dy, float* dw, int Bn, int Di, int Hi, int Wi, int Ci,
                         int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd, int kh, int kw, int pd,
                         int ph, int pw, void* ws, size_t ws_bytes, hupr_stream_t stream);

/* Streaming form of the temporal merge Conv3d(C, C, (G,1,1)) for 64-channel maps (reference models/layers.py:208,218: the
 * level-1 merge): x bf16 (Bn, G, HW, 64) -> y fp32 (Bn, HW, 64).  The operand travels global -> LDS by LDS-DMA through a ring of
 * four 16 KB stages per persistent workgroup (three in flight while one is multiplied); the generic engine keeps one 8 KB tile
 * in flight and ran this 64 flop/B product at 2.2 TB/s.  wp_bf16: the weight packed [Co][G][Ci] (hupr_pack_conv_weights_bf16). */
int hupr_tmerge_stream_supported(int G, int HW, int Ci, int Co);
int hupr_tmerge_fwd_stream_bf16(const void* x, const void* wp_bf16, float* y, int Bn, int G, int HW, int Ci, int Co,
                                hupr_stream_t stream);
/* Its input gradient dx (bf16, (Bn,G,HW,64)) from dy (fp32, (Bn,HW,64)) — write-bound: the dy tile is converted to bf16 in LDS
 * once and multiplied by the G resident weight slices; wp1_bf16 = the mode-1 bf16 packing ([Ci][taps reversed][Co]) — and its
 * weight gradient dw (fp32, parameter layout (Co,Ci,G), overwritten; deterministic): x and dy both stream through the LDS-DMA
 * ring, the voxel axis is put along the MFMA K of both operands by ds_read_b64_tr_b16, every persistent workgroup leaves one
 * fp32 partial in `ws` (hupr_tmerge_wgrad_stream_ws_bytes) and a second kernel sums them in a fixed order. */
int hupr_tmerge_dgrad_stream_bf16(const float* dy, const void* wp1_bf16, void* dx, int Bn, int G, int HW, int Ci, int Co,
                                  hupr_stream_t stream);
int hupr_tmerge_wgrad_stream_supported(int G, int HW, int Ci, int Co);  /* the weight gradient also takes C = 128 / 256 (Ci == Co, G * C / 64 in {2,4,8}):
                                                                        a frame is C / 64 virtual 64-channel frames, the Co / 64 output blocks go to different workgroups */
size_t hupr_tmerge_wgrad_stream_ws_bytes(int Bn, int G, int HW, int Ci, int Co);
int hupr_tmerge_wgrad_stream_bf16(const void* x, const float* dy, float* dw, int Bn, int G, int HW, int Ci, int Co, void* ws,
                                  size_t ws_bytes, hupr_stream_t stream);
/* Mixed-storage variants for the temporal merges of Encoder3D (reference models/layers.py:195-197: Conv3d with
 * kernel (G,1,1) collapsing the frame axis): the feature maps arrive bf16-stored from the bf16-activation encoder,
 * the merged maps and all gradients of the parameters stay fp32.  x_bf16 / y_bf16 / dx_bf16 select the HBM storage
 * type of that tensor (0: fp32, 1: bf16); everything else as in the entry points above.
 * hupr_tmerge_dgrad_bf16: input gradient of such a merge (one output slice, no padding) as Bn*G batched GEMMs
 * dx[b,d] = dy[b] . W_d  with wp1 the mode-1 packed fp32 weights ([Ci][taps reversed][Co]). */
int hupr_conv_fwd_bf16_mixed(const void* x, int x_bf16, const float* wp, const float* bias, void* y, int y_bf16,
                             int Bn, int Di, int Hi, int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co,
                             int out_ld, int kd, int kh, int kw, int pd, int ph, int pw, hupr_stream_t stream);
int hupr_conv_wgrad_bf16_mixed(const void* x, int x_bf16, const float* dy, float* dw, int Bn, int Di, int Hi,
                               int Wi, int Ci, int in_ld, int Do, int Ho, int Wo, int Co, int dy_ld, int kd,
                               int kh, int kw, int pd, int ph, int pw, void* ws, size_t ws_bytes,
                               hupr_stream_t stream);
int hupr_tmerge_dgrad_bf16(const float* dy, const float* wp1, void* dx, int dx_bf16, int Bn, int G, int HW,
                           int Ci, int Co, hupr_stream_t stream);

/* LDS halo-tiled 3x3x3 / 1x3x3 "same" convolution on the bf16 matrix pipe (forward, and input gradient
 * with mode-1 packed weights): the input halo of a 128-voxel tile is staged once as bf16 and all taps
 * run from LDS.  wp_bf16 from hupr_pack_conv_weights_bf16 ([Co][kd*9][Ci] bf16). */
/* BatchNorm statistics fused into the convolution epilogue (BasicBlock3D: every 3x3x3 convolution is followed by a
 * BatchNorm3d, reference models/layers.py:55-65): bf16 activations, the 256-voxel persistent kernel, no bias / residual.
 * `stats`: hupr_conv3x3_halo_stats_rows() x [2][Co] doubles — per-workgroup column sums and sums of squares of the stored
 * (bf16-rounded) outputs; hupr_bn_train_finalize_f32 (below, with the BatchNorm entry points) turns them into the
 * BatchNorm coefficients, so the separate statistics pass over y is skipped. */
int hupr_conv3x3_halo_stats_supported(int Bn, int D, int H, int W, int Ci, int Co, int kd);
int hupr_conv3x3_halo_stats_rows(void);
int hupr_conv3x3_halo_bf16act_stats(const void* x, const void* wp_bf16, void* y, int Bn, int D, int H, int W, int Ci,
                                    int in_ld, int Co, int out_ld, int kd, void* stats, hupr_stream_t stream);
int hupr_pack_conv_weights_bf16(const float* w, void* wp_bf16, int Co, int Ci, int taps, int mode,
                                hupr_stream_t stream);
/* Repack many weights (both layouts each) in ONE launch.  descs_dev: device array of 48-byte records
 * { const float* w; void* wp0; void* wp1; int64 first; int32 co, ci, taps, kind } with kind 0 = fp32 layouts
 * (hupr_pack_conv_weights_f32 modes 0 and 1), 1 = bf16 layouts.  blocks_dev: one 16-byte record per workgroup
 * { int32 entry, layout; int64 start }: the workgroup writes destination elements [start, start + 2048) of that layout. */
int hupr_pack_conv_weights_table(const void* descs_dev, const void* blocks_dev, int n_blocks, hupr_stream_t stream);
int hupr_conv3x3_halo_supported(int D, int H, int W, int Ci, int kd, int kh, int kw, int pd, int ph, int pw);
int hupr_conv3x3_halo_bf16(const float* x, const void* wp_bf16, const float* bias, const float* res, float* y,
                           int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld, int kd,
                           hupr_stream_t stream);

/* Weight gradient of the same 3x3(x3) convolutions, halo-tiled (x halo + dy tile staged once per 128-voxel tile,
 * operands transposed by ds_read_b64_tr_b16).  Requires Ci % 64 == 0; dw in parameter layout (Co,Ci,kd,3,3). */
size_t hupr_conv3x3_wgrad_halo_ws_bytes(int Ci, int Co, int kd);
int hupr_conv3x3_wgrad_halo_bf16(const float* x, const float* dy, float* dw, int Bn, int D, int H, int W, int Ci,
                                 int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes, hupr_stream_t stream);

/* (a4) BatchNorm3d pieces (models/layers.py:46,49,53); x is [M voxels][C]. */
size_t hupr_bn_ws_bytes(int C);
int hupr_bn_train_stats_f32(const float* x, long M, int C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                            float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                            hupr_stream_t stream);
int hupr_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, int C, float* scale, float* shift,
                            hupr_stream_t stream);
/* y = act(x1*scale1+shift1 [+ x2*scale2+shift2]); act: 0 identity, 1 ReLU  (BasicBlock3D.forward :66-70) */
int hupr_scale_shift_act_f32(const float* x1, const float* scale1, const float* shift1, const float* x2,
                             const float* scale2, const float* shift2, float* y, long M, int C, int act,
                             hupr_stream_t stream);
/* Eval-mode BatchNorm(s) + optional ReLU from the module's own tensors (models/layers.py:57,60,64 in eval mode): the
 * coefficient launch (hupr_bn_eval_params_f32) and the apply launch as one — single-sample inference is launch-bound.
 * x2 (and its BatchNorm) may be null.  The bf16act form takes bf16-stored x1 / x2 / y. */
int hupr_bn_eval_act_f32(const float* x1, const float* gamma1, const float* beta1, const float* mean1, const float* var1,
                         float eps1, const float* x2, const float* gamma2, const float* beta2, const float* mean2,
                         const float* var2, float eps2, float* y, long M, int C, int act, hupr_stream_t stream);
int hupr_bn_eval_act_bf16act(const void* x1, const float* gamma1, const float* beta1, const float* mean1, const float* var1,
                             float eps1, const void* x2, const float* gamma2, const float* beta2, const float* mean2,
                             const float* var2, float eps2, void* y, long M, int C, int act, hupr_stream_t stream);
/* finalize only, from nblk rows of [2][C] double column sums produced elsewhere (hupr_conv3x3_halo_bf16act_stats) */
int hupr_bn_train_finalize_f32(const void* partial, int nblk, long M, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                               float* save_invstd, float* scale, float* shift, hupr_stream_t stream);
/* ... of two BatchNorms over tensors of one shape in one launch (the tail of a BasicBlock3D, reference models/layers.py:66-70) */
int hupr_bn_train_finalize2_f32(const void* partial1, int nblk1, const float* gamma1, const float* beta1, float* running_mean1,
                                float* running_var1, float momentum1, float eps1, float* save_mean1, float* save_invstd1,
                                float* scale1, float* shift1, const void* partial2, int nblk2, const float* gamma2,
                                const float* beta2, float* running_mean2, float* running_var2, float momentum2, float eps2,
                                float* save_mean2, float* save_invstd2, float* scale2, float* shift2, long M, int C,
                                hupr_stream_t stream);
int hupr_bn_bwd_f32(const float* dy, const float* y_mask, const float* x, const float* save_mean,
                    const float* save_invstd, const float* gamma, float* dx, float* dgamma, float* dbeta, long M,
                    int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream);

/* out[c] = sum_rows x[row][c] — bias gradient of Encoder3D.layer1.0 (models/layers.py:195); ws as hupr_bn_ws_bytes */
/* both branches of y = relu(bn_a(x1) + bn_b(x2)) (BasicBlock3D tail) in one statistics pass + one apply pass */
int hupr_bn_bwd2_f32(const float* dy, const float* y_mask, const float* x1, const float* mean1, const float* invstd1,
                     const float* gamma1, const float* x2, const float* mean2, const float* invstd2, const float* gamma2,
                     float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, long M, int C,
                     int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
/* The same two backward passes with the ReLU mask RECOMPUTED from x (x1, x2) and the forward pass's scale / shift
 * (the outputs of hupr_bn_train_stats_* / hupr_bn_eval_params_f32, i.e. exactly the expression
 * hupr_scale_shift_act_* evaluated) instead of read back from the activation: one tensor read less per pass. */
int hupr_bn_bwd_remask_f32(const float* dy, const float* fwd_scale, const float* fwd_shift, const float* x,
                           const float* save_mean, const float* save_invstd, const float* gamma, float* dx, float* dgamma,
                           float* dbeta, long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_bn_bwd2_remask_f32(const float* dy, const float* x1, const float* fwd_scale1, const float* fwd_shift1,
                            const float* mean1, const float* invstd1, const float* gamma1, const float* x2,
                            const float* fwd_scale2, const float* fwd_shift2, const float* mean2, const float* invstd2,
                            const float* gamma2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                            float* dbeta2, long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_colsum_f32(const float* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream);

/* (a6) nn.PReLU() with one shared slope (models/layers.py:26,32) */
int hupr_prelu_fwd_f32(const float* x, const float* alpha, float* y, long n, hupr_stream_t stream);
size_t hupr_prelu_ws_bytes(void);
int hupr_prelu_bwd_f32(const float* dy, const float* x, const float* alpha, float* dx, float* dalpha, long n,
                       void* ws, size_t ws_bytes, hupr_stream_t stream);

/* (a3) MNet front end — replaces HuPRNet.forward_chirp + MNet.forward (models/networks.py:23-33,
 * models/chirp_networks.py:17-21): elevation mean, the (F,2)->(2,F) .view, Conv3d(2->32,(2,1,1),s(2,1,1)),
 * MaxPool3d((4,1,1)).  x: (n_bg=B*G, 8, 2, pixels=R*A, 8) fp32; out: (n_bg, pixels, 32) channels-last.
 * means_or_null: (n_bg, pixels, 16) fp32 — the 16 elevation means per pixel; they are all the backward pass needs of
 * x (1/8 of its bytes), so a training forward saves them and the backward takes them instead of x. */
int hupr_mnet_fwd_f32(const float* x, const float* w, const float* bias, float* out, float* means_or_null, long n_bg,
                      int pixels, hupr_stream_t stream);
size_t hupr_mnet_bwd_ws_bytes(void);
int hupr_mnet_bwd_f32(const float* x_or_null, const float* means_or_null, const float* w, const float* bias,
                      const float* dy, float* dw, float* dbias, long n_bg, int pixels, void* ws, size_t ws_bytes,
                      hupr_stream_t stream);

/* tri-/bilinear align_corners=True resampling (models/layers.py:84,89,199,204; gcn_networks.py:49,63) */
int hupr_interp_linear_fwd_f32(const float* x, float* y, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                               int C, int in_ld, int out_ld, hupr_stream_t stream);
int hupr_interp_linear_bwd_f32(const float* dy, float* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                               int C, int in_ld, int out_ld, hupr_stream_t stream);

/* (a5) softmax over the key axis, in place on rows of length n (models/layers.py:131), and its backward */
int hupr_softmax_rows_f32(float* s, long rows, int n, hupr_stream_t stream);
int hupr_softmax_rows_bwd_f32(const float* p, float* dp_inout, long rows, int n, hupr_stream_t stream);

/* (a5) fused flash-style attention on the bf16 matrix pipe (C in {64,128}, N % 128 == 0): no N x N matrix in HBM.
 * K, Q, V, out, dout, dK, dQ, dV: (B,N,C) fp32 token-major; lse, Dq_scratch: (B,N) fp32. */
int hupr_attn_flash_supported(int N, int C);
int hupr_attn_fwd_bf16(const float* K, const float* Q, const float* V, float* out, float* lse, int Bn, int N, int C,
                       int residual, hupr_stream_t stream);
int hupr_attn_bwd_bf16(const float* K, const float* Q, const float* V, const float* out, const float* dout,
                       const float* lse, float* dK, float* dQ, float* dV, float* Dq_scratch, int Bn, int N, int C,
                       int residual, hupr_stream_t stream);

/* Same kernels fed with pre-rounded bf16 copies of the MFMA operands (hupr_cast_f32_to_bf16): every workgroup re-reads
 * all of K/V (or Q/dO), so the copies halve that traffic and drop the per-tile fp32->bf16 conversions; results are
 * bit-identical to the fp32-input entry points.  Vres / V32 / dout32: the fp32 tensors of the exact residual terms. */
int hupr_attn_fwd_bf16in(const void* K, const void* Q, const void* V, const float* Vres, float* out, float* lse, int Bn,
                         int N, int C, hupr_stream_t stream);
int hupr_attn_bwd_bf16in(const void* K, const void* Q, const void* V, const void* dO, const float* V32, const float* out,
                         const float* dout32, const float* lse, float* dK, float* dQ, float* dV, float* Dq_scratch,
                         int Bn, int N, int C, int residual, hupr_stream_t stream);
/* Strided forms for one MSCSA level (models/layers.py:150-163: eight 1x1 projections of the two maps feed four attentions):
 * the four projections of a map are one GEMM into a (B, N, 4C) bf16 tensor, K / Q point at column blocks of it with
 * row strides ldk / ldq (elements), and the backward writes dK / dQ into column blocks (lddk / lddq) of the matching
 * fp32 gradient tensors.  accumulate != 0 (non-residual form) adds onto the dV another attention left in place.
 * out16: optional bf16 copy of the output with row stride ld16 (a column block of the decoder's concatenated input);
 * dO: bf16, row stride lddo; dout32 null: the gradient arrived bf16-stored and dO is used for the exact terms too. */
int hupr_attn_fwd_bf16in_ld(const void* K, int ldk, const void* Q, int ldq, const void* V, const float* Vres, float* out,
                            float* lse, void* out16_or_null, int ld16, int Bn, int N, int C, hupr_stream_t stream);
/* Small batches (BASELINE config C2, B = 1 inference: N / 128 x Bn workgroups leave most of the 256 CUs idle): the same forward
 * with the keys split over a third grid dimension and a merge launch (flash-decoding).  hupr_attn_fwd_split_ws_bytes returns
 * the workspace that needs, or 0 when the one-pass kernel is used (ws may then be null): by default the split is taken for
 * single-sample calls only, so that batched runs keep the rounding their parity gates were measured with. */
size_t hupr_attn_fwd_split_ws_bytes(int Bn, int N, int C);
/* Up to four independent attentions of one shape (the four of an MSCSA level, reference models/layers.py:150-163) in as few launches as
 * fill the chip.  Where hupr_attn_fwd_split_ws_bytes() > 0 (single-sample inference — bound by launches, not work): ONE split launch and
 * ONE merge launch, ws: n_items times that size.  Otherwise (training batches; ws may be NULL): ONE launch of the one-pass kernel over all
 * items (levels 2 and 3), or one ping-pong launch per item (level-1 shape).  K / Q / V: bf16 with row strides ldk / ldq / C; Vres (fp32 V for the residual) and out16 (bf16 copy, stride ld16) may be null. */
typedef struct hupr_attn_item { const void* K; const void* Q; const void* V; const float* Vres; float* out; float* lse; void* out16; } hupr_attn_item;
int hupr_attn_fwd_bf16in_ld_ws_batch(const hupr_attn_item* items, int n_items, int ldk, int ldq, int ld16, int Bn, int N, int C,
                                     void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_attn_fwd_bf16in_ld_ws(const void* K, int ldk, const void* Q, int ldq, const void* V, const float* Vres, float* out,
                               float* lse, void* out16_or_null, int ld16, int Bn, int N, int C, void* ws, size_t ws_bytes,
                               hupr_stream_t stream);
int hupr_attn_bwd_bf16in_ld(const void* K, int ldk, const void* Q, int ldq, const void* V, const void* dO, int lddo,
                            const float* V32, const float* out, const float* dout32_or_null, const float* lse, float* dK,
                            int lddk, float* dQ, int lddq, float* dV, float* Dq_scratch, int Bn, int N, int C,
                            int residual, int accumulate, hupr_stream_t stream);
/* The "QS" forms (round 5): the query operand arrives pre-scaled, Qs = log2(e) Q as bf16 (the level's query projections multiply by
 * log2(e)-scaled weight rows: hupr_pack_conv_weights_table, block layout 3), so that K . Qs^T is the exponent of 2 and the score
 * leaves the matrix pipe as the argument of v_exp_f32: the MFMA chain of a score tile starts from minus the running maximum
 * (forward; kept until a tile exceeds it by more than 8 binary orders) or minus the stored log-sum-exp (backward) instead of 0.
 * Same semantics as the entry points above: out / lse (natural units) of softmax_keys(K Q^T), gradients with respect to K, the
 * UNSCALED Q and V.  Same reference lines (models/layers.py:126-133). */
int hupr_attn_fwd_bf16in_ld_ws_qs(const void* K, int ldk, const void* Qs, int ldq, const void* V, const float* Vres, float* out,
                                  float* lse, void* out16_or_null, int ld16, int Bn, int N, int C, void* ws, size_t ws_bytes,
                                  hupr_stream_t stream);
int hupr_attn_fwd_bf16in_ld_ws_batch_qs(const hupr_attn_item* items, int n_items, int ldk, int ldq, int ld16, int Bn, int N, int C,
                                        void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_attn_bwd_bf16in_ld_qs(const void* K, int ldk, const void* Qs, int ldq, const void* V, const void* dO, int lddo,
                               const float* V32, const float* out, const float* dout32_or_null, const float* lse, float* dK,
                               int lddk, float* dQ, int lddq, float* dV, float* Dq_scratch, int Bn, int N, int C, int residual,
                               int accumulate, hupr_stream_t stream);
/* The backward passes of up to four attentions of one shape and one set of strides (the four of an MSCSA level, reference
 * models/layers.py:150-163 under autograd) with a bf16-stored gradient dO: one row-sum launch, one dQ launch, and dK / dV launches
 * in as many rounds as the dV targets need (an item with `accumulate` adds onto the dV an EARLIER item writes).  Same results as n
 * calls of hupr_attn_bwd_bf16in_ld(_qs) with dout32 == NULL in array order; the level-1 shape is launched per item as before.
 * hupr_attn_fwd_bf16in_ld_ws_batch(_qs) is the forward counterpart (ws may be NULL for training batches). */
typedef struct hupr_attn_bwd_item {
    const void* K; const void* Q; const void* V; const void* dO;      /* bf16 operands (Q: Qs for the _qs form) */
    const float* V32; const float* out; const float* lse;             /* fp32: values, forward output, log-sum-exp */
    float* dK; float* dQ; float* dV; float* Dq;                        /* outputs (row strides lddk / lddq / C) and a (Bn, N) scratch of its own */
    int residual, accumulate;
} hupr_attn_bwd_item;
int hupr_attn_bwd_bf16in_ld_batch(const hupr_attn_bwd_item* items, int n_items, int ldk, int ldq, int lddo, int lddk, int lddq,
                                  int Bn, int N, int C, hupr_stream_t stream);
int hupr_attn_bwd_bf16in_ld_batch_qs(const hupr_attn_bwd_item* items, int n_items, int ldk, int ldq, int lddo, int lddk, int lddq,
                                     int Bn, int N, int C, hupr_stream_t stream);

/* (a7) PRGCN: y = act(t . A + bias) with t = W . x computed by hupr_gemm_f32 (gcn_networks.py:23-29,53-58) */
/* The 1x1 key-point head nn.Conv2d(32, 14, 1, bias=False) (reference models/layers.py:94) in plain fp32 FMAs: x [M][32],
 * w16 [16][32] (the 14 filters zero-padded to 16 output channels), y / dy [M][16]; backward writes dx [M][32] and / or
 * dw16 [16][32] (either may be null).  Used for the "head" precision region of bf16 runs (functional.PRECISION). */
size_t hupr_head1x1_ws_bytes(void);
int hupr_head1x1_fwd_f32(const float* x, const float* w16, float* y, long M, hupr_stream_t stream);
int hupr_head1x1_bwd_f32(const float* x, const float* w16, const float* dy, float* dx_or_null, float* dw16_or_null, long M,
                         void* ws, size_t ws_bytes, hupr_stream_t stream);
/* ... writing only the first out_rows (<= 16) filter rows of dw: out_rows = 14 lets dw be the (14, 32, 1, 1) parameter's own gradient
 * slot in a flat bucket (reference models/layers.py:94: nn.Conv2d(nf, numKeypoints, 1)) */
int hupr_head1x1_bwd_rows_f32(const float* x, const float* w16, const float* dy, float* dx_or_null, float* dw_or_null, int out_rows,
                              long M, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_gcn_adj_fwd_f32(const float* t, const float* adj, const float* bias, float* y, int Bn, int F, int K,
                         int ld, int relu, hupr_stream_t stream);
/* t as `slices` partial products [slices][Bn*F][ld] (the K slices of W x of a single-sample forward), summed here in slice order */
int hupr_gcn_adj_fwd_sliced_f32(const float* t, int slices, const float* adj, const float* bias, float* y, int Bn, int F,
                                int K, int ld, int relu, hupr_stream_t stream);
int hupr_gcn_adj_bwd_f32(const float* dy, const float* y, const float* adj, float* dt, float* gmasked,
                         float* dbias, int Bn, int F, int K, int ld, int relu, hupr_stream_t stream);
/* The feature products of a PRGCN layer on the fp32 matrix pipe, batch folded into the MFMA column axis in place (x, t, dt: (Bn, F, 16),
 * W: (F, F) as nn.Parameter of gcn_networks.py:18 stores it; ld must be 16, F a multiple of 64):
 *   hupr_gcn_wx_f32, trans_w = 0:  t[b][f][n]  = sum_g W[f][g] x[b][g][n]       support = W . x          (gcn_networks.py:25)
 *   hupr_gcn_wx_f32, trans_w = 1:  t[b][g][n]  = sum_f W[f][g] x[b][f][n]       its input gradient W^T . dt
 *   hupr_gcn_dw_f32:               dW[f][g]    = sum_{b,n} dt[b][f][n] x[b][g][n]                     its weight gradient */
int hupr_gcn_wx_f32(const float* W, const float* x, float* t, int Bn, int F, int ld, int trans_w, hupr_stream_t stream);
int hupr_gcn_dw_f32(const float* dt, const float* x, float* dW, int Bn, int F, int ld, hupr_stream_t stream);

/* (a8) sigmoid heads: channels-last logits (B,HW,ld) -> NCHW probabilities (B,K,HW)  (networks.py:40, gcn_networks.py:64) */
int hupr_sigmoid_to_nchw_f32(const float* x, float* y, int Bn, int HW, int K, int ld, hupr_stream_t stream);
int hupr_sigmoid_to_nchw_bwd_f32(const float* dy, const float* y, float* dx, int Bn, int HW, int K, int ld,
                                 hupr_stream_t stream);

/* (a9) loss / targets / decode (misc/losses.py:23-45, misc/utils.py:6-66, misc/metrics.py:10-38) */
size_t hupr_bce_ws_bytes(void);
int hupr_bce_fwd_f32(const float* p, const float* t, long n, float* loss, void* ws, size_t ws_bytes,
                     hupr_stream_t stream);
int hupr_bce_bwd_f32(const float* p, const float* t, const float* grad_out, float* dp, long n, hupr_stream_t stream);
/* Both losses of a step and their weighted sum (reference misc/losses.py:24-33: loss = alpha * BCE(heatmap, target) + beta * BCE(gcn
 * heatmap, target)) in two launches: loss3 = {loss, loss1, loss2}, the same floats as two hupr_bce_fwd_f32 calls and torch's scalar
 * arithmetic; the backward takes the gradient of `loss` (and optionally of loss2) and writes both dp.  ws: 2 x hupr_bce_ws_bytes(). */
int hupr_bce_pair_fwd_f32(const float* p1, const float* p2, const float* t, long n, float alpha, float beta, float* loss3, void* ws,
                          size_t ws_bytes, hupr_stream_t stream);
int hupr_bce_pair_bwd_f32(const float* p1, const float* p2, const float* t, const float* grad_loss, const float* grad_loss2_or_null,
                          float alpha, float beta, float* dp1, float* dp2, long n, hupr_stream_t stream);
int hupr_gaussian_targets_f32(const long long* joints, const float* patch, float* t, int BK, int H, int rad,
                              float stride, hupr_stream_t stream);
int hupr_argmax_rows_f32(const float* p, long rows, int n, int* idx, float* maxval, hupr_stream_t stream);

/* (a10) Adam with coupled L2 weight decay (tools/base.py:47), one flat launch */
int hupr_adam_step_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, float gscale, hupr_stream_t stream);
/* same, {learning rate, step count} read from device memory (2 floats): for steps replayed from a captured hipGraph */
int hupr_adam_step_dev_f32(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, const float* dev_state,
                           float beta1, float beta2, float eps, float weight_decay, float gscale, hupr_stream_t stream);

/* ---- bf16-activation variants ("bf16act") -------------------------------------------------------------
 * Same operators with the ACTIVATION tensors (x, y, dy, dx, residual) stored as bf16 in HBM; parameters,
 * statistics, weight gradients and all arithmetic stay fp32 (fp32 accumulate on the matrix pipe).  The
 * 3-D encoders (models/layers.py:160-191) run on them in bf16 mode: the layer-1 convolution sits at the
 * HBM/MFMA ridge with fp32 tensors (453 flop/B vs 312), bf16 storage doubles its intensity and halves every
 * BatchNorm / resampling pass.  Leading dimensions are in elements; bf16 conv inputs need in_ld % 8 == 0. */
int hupr_conv3x3_halo_bf16act(const void* x, const void* wp_bf16, const float* bias, const void* res, void* y,
                              int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld, int kd,
                              hupr_stream_t stream);
/* The same operator with a caller-supplied workspace.  Grids that would leave most of the chip idle (single-sample inference,
 * config C2 of BASELINE.json: a level-3 layer is 32 workgroups walking 36 weight stages each) split the reduction over
 * (channel chunk, kz plane) slices on blockIdx.y; every slice leaves fp32 partial sums [slice][voxel][Co] in ws and a second
 * launch sums them in slice order (+ bias, + residual) and rounds once.  hupr_conv3x3_halo_splitk_ws_bytes() returns 0 where
 * the one-launch form is used anyway (ws may then be null).  Same products as the plain form; the fp32 summation order differs. */
size_t hupr_conv3x3_halo_splitk_ws_bytes(int Bn, int D, int H, int W, int Ci, int Co, int kd);
int hupr_conv3x3_halo_bf16act_ws(const void* x, const void* wp_bf16, const float* bias, const void* res, void* y,
                                 int Bn, int D, int H, int W, int Ci, int in_ld, int Co, int out_ld, int res_ld, int kd,
                                 void* ws, size_t ws_bytes, hupr_stream_t stream);
/* First half only: the fp32 partial sums [slices][voxel][Co] stay in `part` (slices = hupr_conv3x3_halo_splitk_ws_bytes() /
 * (voxels * Co * 4), which must be > 0) for a consumer that sums them itself.  No bias, no residual. */
int hupr_conv3x3_halo_bf16act_partial(const void* x, const void* wp_bf16, int Bn, int D, int H, int W, int Ci, int in_ld,
                                      int Co, int kd, void* part, size_t part_bytes, hupr_stream_t stream);
/* Inference tails that sum K-sliced partials themselves, in slice order and rounded where the stored tensor would have been
 * (bit-identical to convolution -> reduce -> tail, one launch per convolution less).  x*: bf16 tensor (n* == 0) or n* fp32 slices
 * [n][M][C]; x2 may be null.  mode 0: y = relu?(bn1_eval(x1) [+ bn2_eval(x2)]) — the BasicBlock3D tails, reference
 * models/layers.py:55-70 in eval mode; mode 1: y = prelu(x1 [+ x2]) — the BasicBlock2D tails, :30-37. */
int hupr_infer_tail_bf16act(int mode, const void* x1, int n1, const float* gamma1, const float* beta1, const float* mean1,
                            const float* var1, float eps1, const void* x2, int n2, const float* gamma2, const float* beta2,
                            const float* mean2, const float* var2, float eps2, const float* alpha, int relu, void* y, long M,
                            int C, hupr_stream_t stream);
int hupr_conv3x3_wgrad_halo_bf16act(const void* x, const void* dy, float* dw, int Bn, int D, int H, int W, int Ci,
                                    int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes,
                                    hupr_stream_t stream);
/* The weight gradients of TWO convolutions of one input (main[0] and downsample[0] of a residual block, reference models/layers.py:55-65;
 * same weight shape, same dy stride) in one launch of the LDS-DMA kernel over 2 Co output channels and one reduction: the same partial
 * sums, element for element, as two hupr_conv3x3_wgrad_halo_bf16act calls.  Applies where ..._dual_supported returns 1 (bf16 storage,
 * Co % 64 == 0); ws: 2 x hupr_conv3x3_wgrad_halo_ws_bytes(Ci, Co, kd). */
int hupr_conv3x3_wgrad_halo_dual_supported(int Bn, int D, int H, int W, int Ci, int Co, int kd);
int hupr_conv3x3_wgrad_halo_bf16act_dual(const void* x, const void* dy_a, const void* dy_b, float* dw_a, float* dw_b, int Bn, int D, int H,
                                         int W, int Ci, int in_ld, int Co, int dy_ld, int kd, void* ws, size_t ws_bytes,
                                         hupr_stream_t stream);
int hupr_bn_train_stats_bf16act(const void* x, long M, int C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                                float* save_invstd, float* scale, float* shift, void* ws, size_t ws_bytes,
                                hupr_stream_t stream);
int hupr_scale_shift_act_bf16act(const void* x1, const float* scale1, const float* shift1, const void* x2,
                                 const float* scale2, const float* shift2, void* y, long M, int C, int act,
                                 hupr_stream_t stream);
int hupr_bn_bwd_bf16act(const void* dy, const void* y_mask, const void* x, const float* save_mean,
                        const float* save_invstd, const float* gamma, void* dx, float* dgamma, float* dbeta, long M,
                        int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_bn_bwd2_bf16act(const void* dy, const void* y_mask, const void* x1, const float* mean1, const float* invstd1,
                         const float* gamma1, const void* x2, const float* mean2, const float* invstd2, const float* gamma2,
                         void* dx1, void* dx2, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, long M, int C,
                         int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_bn_bwd_remask_bf16act(const void* dy, const float* fwd_scale, const float* fwd_shift, const void* x,
                               const float* save_mean, const float* save_invstd, const float* gamma, void* dx,
                               float* dgamma, float* dbeta, long M, int C, int train, void* ws, size_t ws_bytes,
                               hupr_stream_t stream);
int hupr_bn_bwd2_remask_bf16act(const void* dy, const void* x1, const float* fwd_scale1, const float* fwd_shift1,
                                const float* mean1, const float* invstd1, const float* gamma1, const void* x2,
                                const float* fwd_scale2, const float* fwd_shift2, const float* mean2, const float* invstd2,
                                const float* gamma2, void* dx1, void* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                float* dbeta2, long M, int C, int train, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_colsum_bf16act(const void* x, long M, int C, float* out, void* ws, size_t ws_bytes, hupr_stream_t stream);
int hupr_prelu_fwd_bf16act(const void* x, const float* alpha, void* y, long n, hupr_stream_t stream);
int hupr_prelu_bwd_bf16act(const void* dy, const void* x, const float* alpha, void* dx, float* dalpha, long n, void* ws,
                           size_t ws_bytes, hupr_stream_t stream);
/* The PReLU backward with the final sum of its slope gradient deferred (nn.PReLU under autograd, reference models/layers.py:21-37): dx now,
 * *n_partials partial sums left in the caller's `partials` buffer (>= hupr_prelu_ws_bytes()); hupr_sum_partials_multi finishes any number
 * of them in one launch per 16 — the twelve slope gradients of a step when their gradient bucket is complete.  Same sums. */
typedef struct hupr_sum_item { const void* partial; int n; float* out; } hupr_sum_item;
int hupr_prelu_bwd_partials_f32(const float* dy, const float* x, const float* alpha, float* dx, long n, void* partials,
                                size_t partials_bytes, int* n_partials, hupr_stream_t stream);
int hupr_prelu_bwd_partials_bf16act(const void* dy, const void* x, const float* alpha, void* dx, long n, void* partials,
                                    size_t partials_bytes, int* n_partials, hupr_stream_t stream);
int hupr_sum_partials_multi(const hupr_sum_item* items, int n_items, hupr_stream_t stream);
int hupr_mnet_fwd_bf16act(const float* x, const float* w, const float* bias, void* out, float* means_or_null, long n_bg,
                          int pixels, hupr_stream_t stream);
/* MNet front end from the fused loader's elevation-mean planes [n_bg][16][pixels] (hupr_fft_chain_loader_means_f32); also
 * leaves the pixel-major means [n_bg][pixels][16] hupr_mnet_bwd_* reads (means_or_null). */
int hupr_mnet_fwd_means_f32(const float* mean_planes, const float* w, const float* bias, float* out, float* means_or_null,
                            long n_bg, int pixels, hupr_stream_t stream);
int hupr_mnet_fwd_means_bf16act(const float* mean_planes, const float* w, const float* bias, void* out, float* means_or_null,
                                long n_bg, int pixels, hupr_stream_t stream);
int hupr_mnet_bwd_bf16act(const float* x_or_null, const float* means_or_null, const float* w, const float* bias,
                          const void* dy, float* dw, float* dbias, long n_bg, int pixels, void* ws, size_t ws_bytes,
                          hupr_stream_t stream);
int hupr_interp_linear_fwd_bf16act(const void* x, void* y, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                   int C, int in_ld, int out_ld, hupr_stream_t stream);
int hupr_interp_linear_bwd_bf16act(const void* dy, void* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                   int C, int in_ld, int out_ld, hupr_stream_t stream);
/* the same with dx += ...: dx already holds another consumer's gradient of the tensor (Encoder3D's level maps feed both a temporal
 * merge and the next level's down-sampling, models/layers.py:212-217) — replaces autograd's separate accumulation kernel */
int hupr_interp_linear_bwd_acc_f32(const float* dy, float* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                   int in_ld, int out_ld, hupr_stream_t stream);
int hupr_interp_linear_bwd_acc_bf16act(const void* dy, void* dx, int Bn, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                       int C, int in_ld, int out_ld, hupr_stream_t stream);
/* boundary casts of the bf16-activation region (n % 4 == 0) */
int hupr_cast_f32_to_bf16(const float* x, void* y, long n, hupr_stream_t stream);
int hupr_cast_bf16_to_f32(const void* x, float* y, long n, hupr_stream_t stream);

/* (a5) The four 1 x 1 projections of one map of an MSCSA level (models/layers.py:150-157: phi / theta, cross / self — bias-free
 *      nn.Conv2d(C, C, 1)) as ONE streaming product: Y (M, 4 C) bf16 = X (M, C) fp32 . Wc^T with Wc (4 C, C) fp32 the four weight
 *      matrices stacked in the order the caller reads their column blocks (functional.MSCSALevelFn: [phi_cross | theta_cross | phi_self |
 *      theta_self], query rows pre-scaled by log2 e for the QS attention kernels); M = B H W channels-last rows.  Same operand roundings
 *      as hupr_conv_fwd_bf16_mixed with a bf16 epilogue (which it replaces for C in {64, 128}: levels 1 and 2); HBM-bound. */
int hupr_mscsa_proj_supported(long M, int C);
int hupr_mscsa_proj_fwd_bf16(const float* X, const float* Wc, void* Y, long M, int C, hupr_stream_t stream);
/* its input gradient: dX (M, C) fp32 = dY (M, 4 C) fp32 . Wc (+ res (M, C) fp32 or null — the map's value gradient dV) */
int hupr_mscsa_proj_dgrad_supported(long M, int C);      /* C == 64 (level 1) */
int hupr_mscsa_proj_dgrad_f32(const float* dY, const float* Wc, const float* res, float* dX, long M, int C, hupr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * (e) Data-parallel exchange over RCCL / xGMI.  Nothing in the reference to mirror: it trains on one
 *     device (tools/base.py:14 `self.device = 'cuda'`, tools/run.py:76-79 forward / backward / step with no
 *     collective); BASELINE.json's configs 4-5 add 8-GPU data parallelism, whose only exchange is the sum
 *     all-reduce of the flat gradient buckets between `loss.backward()` (run.py:78) and `optimizer.step()`
 *     (run.py:79), plus one parameter broadcast at start-up.
 *
 *     librccl.so.1 is bound at run time (hupr_comm_load: the copy already mapped into the process wins, so
 *     the library never brings a second RCCL next to PyTorch's); one communicator per process = per GPU.
 *     hupr_comm_unique_id() is called on rank 0 and the HUPR_COMM_ID_BYTES blob is handed to every rank
 *     out of band (the Python host uses the torch.distributed store); hupr_comm_init_rank() is collective
 *     and binds the communicator to the CURRENT HIP device.  The two data entry points are in place,
 *     stream-ordered and capturable in a hipGraph; they never synchronise the host.
 * ---------------------------------------------------------------------------------------- */
#define HUPR_COMM_ID_BYTES 128
#define HUPR_COMM_F32 0
#define HUPR_COMM_BF16 1
typedef void* hupr_comm_t; /* ncclComm_t */
int hupr_comm_load(const char* librccl_path_or_null);
int hupr_comm_unique_id(void* id_out /* HUPR_COMM_ID_BYTES */);
int hupr_comm_init_rank(hupr_comm_t* comm_out, const void* id, int n_ranks, int rank);
int hupr_comm_destroy(hupr_comm_t comm);
/* ncclCommCount / ncclCommUserRank of a live communicator (what the bench line reports as rccl_ranks). */
int hupr_comm_info(hupr_comm_t comm, int* n_ranks_out, int* rank_out);
/* bucket[i] <- sum over ranks of bucket[i]  (count elements of dtype HUPR_COMM_F32 / HUPR_COMM_BF16) */
int hupr_allreduce_bucket(hupr_comm_t comm, void* bucket, size_t count, int dtype, hupr_stream_t stream);
/* bucket <- rank `root`'s bucket (initial parameter synchronisation) */
int hupr_broadcast_bucket(hupr_comm_t comm, void* bucket, size_t count, int dtype, int root, hupr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HUPR_H */
