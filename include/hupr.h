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

typedef void* hupr_stream_t; /* hipStream_t */

int hupr_version(void);
const char* hupr_last_error(void);

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

/* (a2) loader glue alone on a precomputed cube (the .npy hand-off of the reference):
 * cube_c64 : float2 [n_sf][16][64][64][8]  ->  out float [n_sf][8][2][64][64][8]            */
int hupr_loader_normalize_c64(const void* cube_c64, int n_sf, float* out, hupr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HUPR_H */
