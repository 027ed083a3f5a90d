/*
 * eavqa_test.h - TEST-ONLY entry points of libeavqa_hip.so (not part of the drop-in boundary of include/eavqa.h).
 *
 * eavqa_gemm / eavqa_attention_* choose a kernel from the shape.  The parity tests must cover every kernel on shapes
 * where the dispatcher would pick another one, and tools/gemm_bench.py times individual kernels: the *_ex forms below take
 * the choice as an ARGUMENT, so the library itself keeps no mutable state (the public forms pass 0).
 */
#ifndef EAVQA_TEST_H
#define EAVQA_TEST_H

#include "eavqa.h"

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with hidden visibility: exactly the entry points declared here are exported */
#pragma GCC visibility push(default)

/* eavqa_gemm with a kernel selector.  Bit fields of `knobs`:
 *   [3:0]   start-up delay (x 8 x 64 cycles) of every other co-resident workgroup of the round-1 128 x 128 LDS-DMA kernel (experiment)
 *   [6:4]   timing-only ablation variant of that kernel (RESULTS ARE WRONG when non-zero)
 *   [7]     general register-staged kernel instead of the LDS-DMA kernels (bf16, k-contiguous operands, K % 32 == 0)
 *   [13:8]  full-line (BK = 64) family of csrc/gemm_k64.hip: 0 by cost model, 1 never (round-1 dispatch), 2.. force entry id - 2 of
 *           K64_SHAPES (tile shape x plain / loader-consumer specialised; the table is in that file)
 *   [15:14] round-1 256 x 256 kernel: 0 by shape, 1 never, 2 always (K % 64 == 0)
 *   [17:16] 2 = 8-stage LDS ring of the round-1 128 x 128 kernel (experiment)
 *   [20:18] round-1 shaped tiles: 0 by cost model, 1 never, 2..6 always 128x80 / 128x96 / 256x128 / 256x160 / 256x192
 *   [24:21] 256 x 256 kernel, order of an XCD's tiles in time: 0 library default, 1 m fastest (round 2), 2..15 column groups of value - 1
 *   [25]    256 x 256 kernel: 1 keeps a ragged last tile row in the same launch (default: M % 256 <= 192 rows go to a second, small-tile
 *           launch when that saves a round of workgroups) */
int eavqa_gemm_ex(int dtype, int a_kc, int b_kc, int M, int N, int K,
                  const void* A, int64_t lda, const void* B, int64_t ldb,
                  void* C, int64_t ldc, int out_flags, float alpha,
                  const float* bias, int act,
                  const void* aux_in, void* aux_out, int64_t ld_aux,
                  const void* residual, int64_t ldr, void* stream, int knobs);
/* eavqa_gemm_ln with the same knobs (every kernel family carries the folded-LayerNorm epilogue; the tests force each) */
int eavqa_gemm_ln_ex(int dtype, int a_kc, int b_kc, int M, int N, int K,
                     const void* A, int64_t lda, const void* B, int64_t ldb,
                     void* C, int64_t ldc, int out_flags, float alpha,
                     const float* bias, int act,
                     const void* aux_in, void* aux_out, int64_t ld_aux,
                     const void* residual, int64_t ldr, const eavqa_gemm_ln_t* ln, void* stream, int knobs);

/* eavqa_attention_fwd / _bwd with a path selector: bit 0 keeps bf16 on the vector-ALU kernels (instead of the matrix-core
 * ones), bit 1 (backward) takes the dQ + dK/dV kernel pair even when the problem is one tile, bit 2 (forward) keeps the
 * streamed-tile matrix-core kernel where the K / V-resident one (hd 64, no mask, Sq == Sk <= 592) would be chosen, bit 3 (forward)
 * takes the resident kernel also for Sk <= 64; bit 2 (backward) keeps the round-2 padded-pitch one-tile kernel where the swizzled
 * hd = 64 one (bwd_fused64_kernel) would be chosen; bit 4 (forward, one query per sample): keep the round-3 decode kernel (V image in
 * LDS, 16 waves) where the round-4 all-in-registers one (<= 8 waves, K and V in flight from the first instruction) would be chosen; bit 5:
 * take the all-in-registers kernel also for key counts that need 2 x 20 loads per wave (measured slower than the LDS image). */
int eavqa_attention_fwd_ex(int dtype, int B, int H, int Sq, int Sk, int hd,
                           const void* q, int64_t ldq, const void* k, int64_t ldk,
                           const void* v, int64_t ldv, void* o, int64_t ldo,
                           int64_t q_batch_rows, int64_t kv_batch_rows,
                           const int32_t* key_mask, int64_t ld_mask, const int32_t* cu_seqlens, int causal, float scale,
                           float* lse, void* stream, int path);
int eavqa_attention_bwd_ex(int dtype, int B, int H, int Sq, int Sk, int hd,
                           const void* q, int64_t ldq, const void* k, int64_t ldk,
                           const void* v, int64_t ldv, const void* o, int64_t ldo,
                           const void* d_o, int64_t lddo,
                           void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                           const int32_t* key_mask, const int32_t* cu_seqlens, int causal, float scale,
                           const float* lse, float* delta, void* stream, int path);

/* eavqa_gemm_splitk with the depth of the per-wave weight-load window as an argument: 8 or 16 k-steps (KiB) in flight per wave;
 * 0 = the library default. */
int eavqa_gemm_splitk_ex(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                         float* partials, int ks, void* stream, int unroll);

/* eavqa_lm_block_forward with the decode-step structure as an argument, for A / B measurements and parity tests of alternative
 * structures against the shipped one: 0 = what the library does; values the build does not know behave like 0. */
int eavqa_lm_block_forward_ex(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act, float eps,
                              int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask,
                              void* workspace, int64_t workspace_bytes, void* stream, int route);

/* eavqa_t5_decoder_step with the step structure as an argument: 0 = what the library does (split-K projections with fused consumers where
 * every shape allows it), 1 = the round-3 call sequence (eavqa_gemm's M <= 64 kernels, separate RMSNorm / append / gate kernels). */
int eavqa_t5_decoder_step_ex(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                             int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                             const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                             void* workspace, int64_t workspace_bytes, void* stream, int route);

/* eavqa_gemm_decode with a selector: bits [3:0] force the 16-column fragments per workgroup (0 = by shape), bit 4 = plain instead of
 * non-temporal weight loads, bits [11:8] = split the rows over that many workgroups per column group (0 = all rows in one);
 * bit 7: every workgroup starts its walk over K at a different k-block; bits 5 / 6: timing-only ablations - the A / the B operand is not fetched (zeros arrive instead: RESULTS ARE WRONG). */
int eavqa_gemm_decode_ex(const eavqa_decode_gemm_t* args, void* stream, int sel);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* EAVQA_TEST_H */
