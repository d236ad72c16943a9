/* hupr_debug.h — test and profiling aids of libhupr_hip.so.  NOT part of the operator contract (include/hupr.h): process-wide
 * switches that select an alternative kernel for a parity comparison in tests/ (same products, another summation order or launch
 * shape) or arm a profiling hook.  Every default is the product path; nothing in the package calls these outside tests/ and scripts/.
 */
#ifndef HUPR_DEBUG_H
#define HUPR_DEBUG_H
#ifdef __cplusplus
extern "C" {
#endif

void hupr_debug_attn_trace(void* dev_buf); /* profiling aid: device buffer of 3 x 2 x 4096 uint64 s_memtime stamps written by workgroup 0 of the ping-pong attention kernels, or null */
void hupr_debug_halo_variant(int v);  /* A/B aid: 0 auto, 1 force the 128-voxel kernel */
void hupr_debug_halo_ablate(int bits); /* profiling aid: bit0 skip halo fill, bit1 skip MFMA, bit2 skip stores */
void hupr_debug_halo_res_prefetch(int on);  /* A/B aid: 0 = the 256-voxel 16 x 16 x 32 convolution reads a residual in its immediate epilogue (rounds 4-5a); default 1: prefetched, deferred epilogue */
void hupr_debug_splitk_slices(int s);     /* test aid: slices per workgroup of the split-K reduction: 0 auto, 4, 16; + 256: the scattered-store kernel (hupr_k_splitk_reduce4) for convolution weight gradients too — same sums, the comparison the parity test makes */
void hupr_debug_wgrad_m16(int on);        /* A/B aid: 0 = the LDS-DMA weight gradient on v_mfma_f32_32x32x16_bf16 (rounds 2-4); default 1: v_mfma_f32_16x16x32_bf16 (round 5) */
void hupr_debug_wgrad_ci32(int on);       /* A/B aid: 0 sends Ci <= 32 weight gradients through the two-quadrant kernel (K halves only), 2 forces the K-quarter mode at any size, 3 the same on the 32 x 32 x 16 kernel (the rounds-3-5 path), 1 = default */
void hupr_debug_halo_trace(void* device_u64_4096); /* profiling aid: per-tile s_memtime stamps of workgroup 0 (null = off) */
void hupr_debug_attn_split(int mode);    /* 0 (default): split for Bn == 1 only; 1: every grid below 128 workgroups; -1: never */
void hupr_debug_halo_split_k(int on);     /* A/B aid: 0 = never slice the reduction of small grids */
void hupr_debug_halo_tiles(int mask);     /* test aid: which tiles of the 256-voxel convolution kernel (conv_halo256m_bf16.hip) are in use — bit 0: 4 x 8 x 8, bit 1: 2 x 8 x 16 (D % 4 != 0), bit 2: 1 x 16 x 16 (1 x 3 x 3 taps), bit 3: 8 x 8 x 8 (32 output channels, D = 8), bit 4: 4 x 8 x 8 on 64-byte rows (32 input channels); default 31.  A cleared bit sends those layers to the 128-voxel kernel (the comparison the parity tests make) */

#ifdef __cplusplus
}
#endif
#endif /* HUPR_DEBUG_H */
