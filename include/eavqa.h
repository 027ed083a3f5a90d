/*
 * eavqa.h - C ABI of libeavqa_hip.so: the MI355X (gfx950) kernels behind the
 * CLIP-ViT -> mapping network -> causal-LM hot path of
 * rs-anderson/explicit-alignment-for-vqa-tasks.
 *
 * The reference has NO native code and no FFI: every entry point
 * below replaces arithmetic the reference delegates to PyTorch / HuggingFace.
 * Each declaration cites the reference (or third-party) code whose arithmetic it
 * performs.  Paths are relative to the reference repo; "HF:" means
 * transformers/ (5.15.0 in the build container, 4.12.5 pinned by the reference).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    named host_*; the caller owns and allocates every buffer;
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*):
 *    no allocation, no synchronisation, no global mutable state, so calls are
 *    re-entrant across streams and capturable into a hipGraph;
 *  - return value: 0 on success, a negative EAVQA_E_* code otherwise; nothing
 *    throws across the ABI; eavqa_strerror() names a code;
 *  - dtype: 0 = float32, 1 = bfloat16 (storage of activations / weights);
 *    accumulation is always float32;
 *  - "ld*" arguments are leading dimensions in ELEMENTS.
 */
#ifndef EAVQA_H
#define EAVQA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with hidden visibility: exactly the entry points declared here are exported */
#pragma GCC visibility push(default)

#define EAVQA_ABI_VERSION 1

enum { EAVQA_F32 = 0, EAVQA_BF16 = 1,
       EAVQA_F16 = 2 };   /* IEEE half: only as the STORAGE type of a frozen tower's residual stream (eavqa_layernorm_fwd x / y,
                             eavqa_gemm residual / C with EAVQA_GEMM_STREAM_F16); no kernel multiplies in it */

/* activation ids (HF:activations.py: gelu_new :59-66, quick_gelu :117-123; torch tanh/relu) */
enum { EAVQA_GEMM_OUT_F32 = 1, EAVQA_GEMM_RESIDUAL_LOWP = 2, EAVQA_GEMM_STREAM_F16 = 4 };     /* eavqa_gemm out_flags */
enum { EAVQA_ACT_NONE = 0, EAVQA_ACT_TANH = 1, EAVQA_ACT_RELU = 2, EAVQA_ACT_GELU_NEW = 3, EAVQA_ACT_QUICK_GELU = 4 };

enum {
    EAVQA_OK = 0,
    EAVQA_E_ARG = -1,       /* null pointer / non-positive size */
    EAVQA_E_ALIGN = -2,     /* pointer or leading dimension not aligned as required */
    EAVQA_E_SHAPE = -3,     /* shape not supported by the kernel */
    EAVQA_E_DTYPE = -4,     /* unknown dtype / activation id */
    EAVQA_E_LAUNCH = -5,    /* hipLaunch / hipFuncSetAttribute failed */
    EAVQA_E_ARCH = -6       /* device is not gfx950 */
};

int eavqa_abi_version(void);
const char* eavqa_strerror(int code);
/* 0 when the current device is gfx950, EAVQA_E_ARCH otherwise (host call, no stream). */
int eavqa_check_device(void);

/* ---------------------------------------------------------------- GEMM ---
 * C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
 *   a_kc != 0: A stored [M,K] (k contiguous, lda = row stride); else stored [K,M].
 *   b_kc != 0: B stored [N,K] (k contiguous, torch.nn.Linear weight layout); else
 *              stored [K,N] (HF Conv1D weight layout, HF:pytorch_utils.py Conv1D).
 * epilogue, in this order:
 *   v  = alpha*acc + bias[n]                 (bias: float32 [N] or NULL)
 *   if aux_out: aux_out[m,n] = v             (pre-activation, dtype = `dtype`)
 *   v  = act(v)                              (act id above)
 *   if aux_in:  v = v * act'(aux_in[m,n])    (backward of the activation; act() above is skipped)
 *   if residual: v += residual[m,n]          (float32 - or `dtype` with EAVQA_GEMM_RESIDUAL_LOWP -, leading dim ldr; may alias C
 *                                             when it has C's element type)
 *   C[m,n] = v                               (float32 with EAVQA_GEMM_OUT_F32, else `dtype`)
 * out_flags: EAVQA_GEMM_OUT_F32 (= 1: the historical `out_f32` argument) | EAVQA_GEMM_RESIDUAL_LOWP (= 2: the residual is a 16-bit
 *   type, not float32: the residual stream of a frozen, forward-only tower - halves the epilogue's stream traffic) |
 *   EAVQA_GEMM_STREAM_F16 (= 4: that 16-bit residual, and C when it is not float32, are IEEE half instead of `dtype`; bf16 operands
 *   only.  OpenAI CLIP itself runs its whole tower in fp16 on the GPU, extract_clip_embeddings_conceptual_captions.py:26,86; a
 *   bf16 stream rounds each of the 2 x n_layer residual sums to 8 bits and doubled the embedding error, half does not).
 * Replaces: torch.nn.Linear / HF Conv1D matmuls of clipcap.py:31-42 (MLP mapper),
 * :45-104 (mapper transformer), HF:models/gpt2/modeling_gpt2.py:186-226,229-243,698,
 * HF:models/opt/modeling_opt.py:137-181,228-248, HF:models/clip/modeling_clip.py:296-385,
 * and their autograd (dgrad for frozen weights, dgrad+wgrad for the mapper).
 * Requirements: contiguous dimension of A and of B a multiple of 8 (bf16) / 4 (f32)
 * elements, 16-byte aligned base pointers, lda/ldb multiples of the same.
 */
int eavqa_gemm(int dtype, int a_kc, int b_kc, int M, int N, int K,
               const void* A, int64_t lda, const void* B, int64_t ldb,
               void* C, int64_t ldc, int out_flags, float alpha,
               const float* bias, int act,
               const void* aux_in, void* aux_out, int64_t ld_aux,
               const void* residual, int64_t ldr, void* stream);

/* eavqa_gemm with the LayerNorm between two Linear layers of a FROZEN pre-LN decoder layer folded into its neighbours (round 4; HF
 * modeling_gpt2.py:246-309 `ln_1 -> c_attn`, `ln_2 -> c_fc`; modeling_opt.py:184-254 `self_attn_layer_norm -> q/k/v_proj`,
 * `final_layer_norm -> fc1`).  With W' = W * gamma (column-wise), c[n] = sum_k W'(n,k) and d = W beta + bias, prepared once at load:
 *     LayerNorm(x) W^T + bias  =  rstd (x W'^T - mean c) + d
 * so the normalisation is a per-row scale and a rank-1 term of the consuming product's epilogue, and no LayerNorm kernel runs:
 *   producer (the Linear whose result is the stream x: out-projection, FFN-down):
 *     copy_out  != NULL: the result a second time in `dtype`, leading dimension ld_copy (the consumer's A operand);
 *     stats_out != NULL: float32 [M, stats_ld, 2], stats_ld >= ceil(N / 64): (sum, sum of squares) of every result row (of the values as
 *                        stored in C before any rounding) spread over 64-column slots - a tile writes its whole sum into its first slot and
 *                        zeros into the others, so the consumer adds all stats_ld slots whatever tile width produced them;
 *   consumer (QKV projection, FFN-up):
 *     ln_stats != NULL: the producer's stats (leading dimension ln_ld slots, ln_parts of them summed), ln_cols = the length of a stream
 *                       row; A = the UN-normalised rows in `dtype`, B = W', bias = d, ln_c = c (float32 [N]):
 *                       v = rstd[m] (alpha acc - mean[m] c[n]) + bias[n], then the rest of eavqa_gemm's epilogue;
 *                       mean = sum / ln_cols, rstd = 1 / sqrt(sumsq / ln_cols - mean^2 + ln_eps);
 *     mean_out / rstd_out (both or neither): float32 [M], what eavqa_layernorm_fwd would have saved for eavqa_layernorm_bwd.
 * Either side may be used alone; every pointer of `ln` may be NULL.  k-contiguous operands only (a_kc and b_kc non-zero: the Linear layers of
 * a frozen LM); not available with the M <= 64 weight-streaming kernel (the call takes a tiled kernel instead). */
typedef struct {
    void* copy_out; int64_t ld_copy;
    float* stats_out; int32_t stats_ld;
    const float* ln_stats; int32_t ln_parts; int32_t ln_ld; int32_t ln_cols;
    const float* ln_c; float ln_eps;
    float* mean_out; float* rstd_out;
} eavqa_gemm_ln_t;
int eavqa_gemm_ln(int dtype, int a_kc, int b_kc, int M, int N, int K,
                  const void* A, int64_t lda, const void* B, int64_t ldb,
                  void* C, int64_t ldc, int out_flags, float alpha,
                  const float* bias, int act,
                  const void* aux_in, void* aux_out, int64_t ld_aux,
                  const void* residual, int64_t ldr, const eavqa_gemm_ln_t* ln, void* stream);

/* ----------------------------------------------------------- LayerNorm ---
 * torch.nn.LayerNorm over the last dim (ln_1/ln_2/ln_f HF:gpt2 :253-257,620;
 * self_attn_layer_norm/final_layer_norm HF:opt :196-205; CLIP layer_norm1/2, pre/post
 * HF:clip :340-343,909-912; mapper norm1/norm2 clipcap.py:131-135).
 * y: `dtype` (EAVQA_F32 / EAVQA_BF16 / EAVQA_F16).  x: float32 when x_kind == 1, `dtype` when 0, and explicitly bfloat16 / half
 * when x_kind == 2 / 3 (a half stream normalised into a bf16 GEMM operand).  mean/rstd (float32 [rows]) may be
 * NULL in inference.  gamma/beta float32 [cols].  cols % 4 == 0, cols <= 8192.
 */
int eavqa_layernorm_fwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx,
                        const float* gamma, const float* beta, float eps,
                        void* y, int64_t ldy, float* mean, float* rstd, void* stream);
/* dx[r,:] = (dres ? dres[r,:] : 0) + LayerNorm'(dy)[r,:]   (float32 out; dres may alias dx)
 * dgamma/dbeta: float32 [cols], ACCUMULATED with atomics when non-NULL (mapper only).
 * dx_lowp: optional second copy of dx in `dtype` (the next dgrad GEMM's A operand), or NULL. */
int eavqa_layernorm_bwd(int dtype, int x_f32, int rows, int cols, const void* x, int64_t ldx,
                        const void* dy, int64_t lddy, const float* gamma,
                        const float* mean, const float* rstd,
                        const float* dres, float* dx, int64_t lddx,
                        float* dgamma, float* dbeta, void* dx_lowp, int64_t ld_lowp, void* stream);
/* eavqa_layernorm_fwd / _bwd with the row quantiser of the fp8 path fused in (round 4; BASELINE configs[4]): the result that would have been
 * written in bfloat16 and read back by eavqa_quantize_rows_fp8 leaves as e4m3 bytes + one float32 scale per row (scale = amax / 448 of the
 * bf16-rounded row, 1 for an all-zero row) - the same bytes and scales as the two-kernel sequence, one pass over the row less.
 * x_kind as in eavqa_layernorm_fwd (1 float32, 2 bfloat16, 3 half); the backward's dy is bfloat16, dx stays float32. */
int eavqa_layernorm_fwd_fp8(int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, const float* beta,
                            float eps, void* yq, int64_t ldq, float* row_scale, float* mean, float* rstd, void* stream);
int eavqa_layernorm_bwd_fp8(int x_f32, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy,
                            const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx, int64_t lddx,
                            void* dxq, int64_t ldq, float* row_scale, void* stream);


/* ----------------------------------------------------------- attention ---
 * softmax(scale * q k^T + mask) v per (batch, head); eager formula
 * HF:models/gpt2/modeling_gpt2.py:54-72 (causal + key padding), HF:clip :249-277 (full),
 * clipcap.py:94-100 (mapper, layout bnmh == per-head softmax over keys).
 * q/k/v/o: `dtype`, element (b, s, h, d) at base[(b*batch_rows + s)*ld + h*hd + d] where
 * batch_rows is q_batch_rows for q/o and kv_batch_rows for k/v (0 = Sq / Sk; a KV cache
 * allocated [B, S_max, E] passes kv_batch_rows = S_max).
 * key_mask: int32, row b at key_mask + b*ld_mask (ld_mask 0 = Sk), 0 = padded key; or NULL.  causal: key j visible to query i iff
 * j <= i + (Sk - Sq).  Masked scores are replaced by -FLT_MAX (HF adds finfo.min), so a
 * fully masked row yields the uniform average over all Sk keys, never NaN.
 * cu_seqlens: int32 [B+1] or NULL.  Non-NULL = packed self-attention over rows with the padding removed
 * (eavqa_build_row_plan): sample b owns rows [cu[b], cu[b+1]) of q/k/v/o, Sq = Sk = that length (the Sq
 * argument is then the LONGEST sample, used for the grid and the lse pitch), key_mask must be NULL.
 * lse: float32 [B,H,Sq] log-sum-exp of the scaled masked scores (NULL in inference).
 * hd in {4,8,16,32,48,64,80,96,128,160,256,320,512}.
 */
int eavqa_attention_fwd(int dtype, int B, int H, int Sq, int Sk, int hd,
                        const void* q, int64_t ldq, const void* k, int64_t ldk,
                        const void* v, int64_t ldv, void* o, int64_t ldo,
                        int64_t q_batch_rows, int64_t kv_batch_rows,
                        const int32_t* key_mask, int64_t ld_mask, const int32_t* cu_seqlens, int causal, float scale,
                        float* lse, void* stream);
/* One decode step (Sq = 1, causal) that also APPENDS: k_new / v_new hold the new position's K / V rows (`dtype`, sample b at
 * k_new + b*ld_new, head h at + h*hd - e.g. columns E.. and 2E.. of the QKV projection's output); they are written to position Sk-1 of
 * each sample's cache (k_cache / v_cache [B, kv_batch_rows, ld]) and attended together with positions 0..Sk-2 already there.
 * bfloat16, hd % 8 == 0, hd <= 128, Sk <= 3584 (EAVQA_E_SHAPE otherwise).  Replaces the append + attention of one token of the
 * reference's full re-forward (src/models/clipcap.py:414-419). */
int eavqa_attention_decode(int dtype, int B, int H, int Sk, int hd, const void* q, int64_t ldq, void* k_cache, int64_t ldk,
                           void* v_cache, int64_t ldv, int64_t kv_batch_rows, const void* k_new, const void* v_new, int64_t ld_new,
                           void* o, int64_t ldo, const int32_t* key_mask, int64_t ld_mask, float scale, void* stream);
/* The same step straight from the split-K partial sums of the QKV projection (eavqa_gemm_splitk: qkv_partials [ks][B][3 H hd] fp32,
 * columns q | k | v): every lane sums the slices in index order, adds qkv_bias (may be NULL) and rounds to bfloat16 - bit for bit the
 * q / K / V that eavqa_splitk_finish would have stored - appends K / V at position Sk-1 and attends.  One kernel instead of three
 * (finish, append, attention) per layer of a decode step. */
int eavqa_attention_decode_splitk(int dtype, int B, int H, int Sk, int hd, const float* qkv_partials, int ks, const float* qkv_bias,
                                  void* k_cache, int64_t ldk, void* v_cache, int64_t ldv, int64_t kv_batch_rows, void* o, int64_t ldo,
                                  const int32_t* key_mask, int64_t ld_mask, float scale, void* stream);
/* Backward (dense batches only: batch_rows = Sq / Sk): dq/dk/dv in `dtype`, addressed as q/k/v
 * with leading dims lddq/lddk/lddv.
 * delta: float32 scratch [B,H,Sq] (rowsum(do*o), written by the call). */
int eavqa_attention_bwd(int dtype, int B, int H, int Sq, int Sk, int hd,
                        const void* q, int64_t ldq, const void* k, int64_t ldk,
                        const void* v, int64_t ldv, const void* o, int64_t ldo,
                        const void* d_o, int64_t lddo,
                        void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                        const int32_t* key_mask, const int32_t* cu_seqlens, int causal, float scale,
                        const float* lse, float* delta, void* stream);

/* ---------------------------------------- sequence assembly (indexing) ---
 * Integer / index work is bit-exact against the oracle.
 */
/* ClipCaptionModel.forward/generate clipcap.py:303-321,353-381: row b of the LM input is
 * [L prefix slots | T text tokens].  Writes src[b,s] = -(1 + b*prefix_row_stride + prefix_row_offset + s)
 * for s < L (row of the mapper output buffer holding prefix slot s of sample b; stride L, offset 0 for
 * the MLP mapper, stride clip_length+L, offset clip_length for the transformer mapper's residual
 * stream clipcap.py:220), else token id; mask_out[b,s] = 1 for s < L else question_mask; pos[b,s] = s
 * (pos_mode 0, GPT-2 HF:gpt2 :571-574) or cumsum(mask)*mask - 1 + 2 (pos_mode 1, OPT HF:opt :45-70).
 * tokens/question_mask int64 [B,T] (the reference hands over int64); outputs int32 [B,L+T]. */
int eavqa_build_prefix_rows(int B, int L, int T, const int64_t* tokens, const int64_t* question_mask,
                            int pos_mode, int prefix_row_stride, int prefix_row_offset,
                            int32_t* src, int32_t* mask_out, int32_t* pos, void* stream);
/* VCT0Model.insert_prefix_into_input src/models/vct0.py:494-533 (golden vectors
 * src/models/vct0_test.py:79-211): the n-th sentinel token of row b (ids special_token_id - i,
 * i = 0..n_img-1) expands into the L prefix slots of image n.  T_out = T + (L-1)*n_img.
 * Writes src (token id, or -(1 + (b*n_img + n)*L + l)), mask_out, pos (as above) and
 * status[b] = number of sentinels found (caller checks == n_img; the reference's .view fails). */
int eavqa_build_fewshot_rows(int B, int T, int L, int n_img, int64_t special_token_id,
                             const int64_t* tokens, const int64_t* question_mask, int pos_mode,
                             int32_t* src, int32_t* mask_out, int32_t* pos, int32_t* status, void* stream);
/* Row plan of the training forward.  pack != 0 drops every position whose mask is 0 (padding): because of the
 * causal + key-padding mask (HF:gpt2 :54-72) and ignore_index -100 such positions influence neither the loss nor
 * any attended position, so the loss and the mapper gradients are unchanged while every GEMM / LayerNorm /
 * attention runs on sum(lengths) rows instead of B*S.  pack == 0 keeps all B*S rows (identity plan).
 * Inputs: mask/src/pos int32 [B,S] (eavqa_build_prefix_rows), labels int64 [B,S] UNshifted or NULL.
 * Outputs: cu_seqlens int32 [B+1]; for each kept row r (sample-major, position order): src_rows[r], pos_rows[r],
 * flat_index[r] = b*S+s, row_labels[r] = labels[b, s+1] (or -100 at s = S-1 / labels NULL), buffers sized B*S. */
int eavqa_build_row_plan(int B, int S, int pack, const int32_t* mask, const int64_t* labels, const int32_t* src,
                         const int32_t* pos, int32_t* cu_seqlens, int32_t* src_rows, int32_t* pos_rows,
                         int64_t* row_labels, int32_t* flat_index, void* stream);
/* dst[(b*dst_batch_rows + dst_row0 + s)*ldd + c] = src[(b*src_batch_rows + s)*lds + c], b < B, s < S,
 * c < cols (`dtype` -> `dtype`): fills / appends the per-layer KV cache [B, S_max, E] from the QKV
 * projection rows (the reference re-runs the whole sequence instead, clipcap.py:414-419). cols % 4 == 0. */
int eavqa_copy_rows(int dtype, int B, int S, int cols, const void* src, int64_t lds, int64_t src_batch_rows,
                    void* dst, int64_t ldd, int64_t dst_batch_rows, int64_t dst_row0, void* stream);
/* Rows that carry a label (row_labels[r] >= 0, as produced by eavqa_build_row_plan): their row numbers, in order, into
 * sel_idx[0..capacity) and their labels into sel_labels; count[0] (optional) = how many there are.  Only these rows go
 * through the lm_head (HF computes all B*S rows of logits and ignores the rest in the loss, loss_utils.py:32-46). */
int eavqa_select_rows(int M, const int64_t* row_labels, int capacity, int32_t* sel_idx, int64_t* sel_labels, int32_t* count,
                      void* stream);
/* scatter = 0: dst[i, :] = src[idx[i], :];  scatter = 1: dst[idx[i], :] = src[i, :]  for i < n (`dtype`, cols % 8 (bf16) / 4). */
int eavqa_move_rows(int dtype, int scatter, int n, int cols, const void* src, int64_t ld_src, const int32_t* idx, void* dst,
                    int64_t ld_dst, void* stream);
/* dst[c, r] = src[r, c] (`dtype` -> `dtype`): the k-contiguous copy of a trainable [N,K] weight that its dgrad GEMM streams
 * (the frozen weights get theirs once at load; the MLP mapper's second Linear needs a fresh one per step). */
int eavqa_transpose(int dtype, int rows, int cols, const void* src, int64_t ld_src, void* dst, int64_t ld_dst, void* stream);
/* out[c] (+)= sum_r x[r, c]  (float32 out; `dtype` in): bias gradients of the mapper's Linear layers and
 * the gradient of TransformerMapper.prefix_const (clipcap.py:235-237).  accumulate != 0 adds to out. */
int eavqa_colsum(int dtype, int rows, int cols, const void* x, int64_t ldx, float* out, int accumulate, void* stream);
/* x[b,s,:] = (src >= 0 ? wte[src] : prefix_rows[-src-1]) + wpe[pos]   (float32 out, residual stream)
 * wte/wpe/prefix_rows in `dtype`; wpe may be NULL.  Also the append path of
 * _generate_from_embeddings clipcap.py:423,440-442.  E % 4 == 0. */
int eavqa_embed_assemble(int dtype, int rows, int E, const int32_t* src, const int32_t* pos,
                         const void* wte, int64_t ld_wte, const void* prefix_rows, int64_t ld_prefix,
                         const void* wpe, int64_t ld_wpe, float* x, int64_t ldx, void* stream);
/* backward of the above w.r.t. prefix_rows: dprefix[-src-1,:] = dx[row,:] for src < 0 (`dtype` out). */
int eavqa_embed_assemble_bwd(int dtype, int rows, int E, const int32_t* src, const float* dx, int64_t lddx,
                             void* dprefix, int64_t ld_dprefix, void* stream);

/* ClipCapExecutor.training_step label construction src/trainers/clipcap_exector.py:134-150
 * (mode 0) and the Conceptual-Captions collate data_loader_conceptual_captions.py:94-95 (mode 1).
 * Output int64 [B, L + T]: L leading -100 (clipcap.py:323-335) then the masked labels. */
int eavqa_build_labels(int mode, int B, int T, int L, const int64_t* input_ids, int64_t pad_token_id,
                       int64_t bos_token_id, int64_t* labels_out, void* stream);

/* -------------------------------------------------------- loss / argmax ---
 * ForCausalLMLoss HF:loss/loss_utils.py:49-71: labels shifted left by one, ignore_index -100,
 * mean over kept positions.  logits float32 [rows, ld] (first V columns valid), labels int64 [B,S]
 * UNshifted (rows = B*S, row r = b*S+s is scored against labels[b, s+1]); or, with S == 0, `labels` is one
 * already-shifted label per row (int64 [B], B = number of rows; packed rows of eavqa_build_row_plan).
 * Writes row_loss/row_lse float32 [rows] (0 for ignored rows), then loss[0] = sum/count and
 * count[0] (deterministic tree reduction, no atomics).  A label that is neither -100 nor in [0, V) makes loss[0] NaN
 * (torch.nn.functional.cross_entropy asserts on the device there; nothing throws across this ABI). */
int eavqa_ce_fwd(int B, int S, int V, const float* logits, int64_t ld, const int64_t* labels,
                 float* row_loss, float* row_lse, float* loss, float* count, void* stream);
/* loss[0] = NaN when count[0] > capacity: the scored-row compaction (eavqa_select_rows) was sized from a host-side label
 * count; an under-count would silently drop labelled rows, so it poisons the loss instead (no host synchronisation). */
int eavqa_guard_count(const int32_t* count, int capacity, float* loss, void* stream);
/* dlogits[r, v] = gscale[0] / count[0] * (softmax(logits[r])[v] - [v == label]) for kept rows, 0 for
 * ignored rows and for pad columns V..ldd-1 (`dtype` out). */
int eavqa_ce_bwd(int dtype, int B, int S, int V, const float* logits, int64_t ld, const int64_t* labels,
                 const float* row_lse, const float* count, const float* gscale,
                 void* dlogits, int64_t ldd, void* stream);
/* torch.argmax(logits[:, -1, :], -1) clipcap.py:420-421 (first maximal index) fused with the
 * finished-row bookkeeping of :426-461: raw[b] = argmax; emitted[b] = unfinished ? raw : pad;
 * unfinished[b] &= (emitted != eos) when eos >= 0.  logits float32 [B, ld].
 * logprob (NULL = skip): float32 [B], log_softmax(logits[b])[raw[b]] - the per-token score that the few-shot
 * ensembling sums (`torch.log(torch.stack(outputs.scores).softmax(-1))`, src/trainers/few_shot_vqa_executor.py:314).
 * any_unfinished (NULL = skip; eos >= 0 only): one int32 the caller has zeroed; set to 1 when some row is still unfinished after this
 * step - the `unfinished_sequences.max() == 0` test of clipcap.py:463 without a reduction on the host side of the ABI. */
int eavqa_greedy_pick(int B, int V, const float* logits, int64_t ld, int64_t pad_token_id, int64_t eos_token_id,
                      int32_t* raw, int64_t* emitted, int64_t ld_emitted, int32_t* unfinished, float* logprob,
                      int32_t* any_unfinished, void* stream);

/* ----------------------------------------------------------- optimiser ---
 * torch.optim.AdamW single-tensor update as configured at src/trainers/clipcap_exector.py:79-81
 * over one flat float32 parameter buffer; grad_scale multiplies the gradient first
 * (1/accumulate_grad_batches, 1/world_size).  shadow: optional `dtype` copy of the updated
 * parameters for the next forward.  step >= 1. */
int eavqa_adamw(int64_t n, float* param, const float* grad, float* m, float* v, int step,
                float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                int shadow_dtype, void* shadow, void* stream);

/* ------------------------------------------------------- CLIP patchify ---
 * CLIPVisionEmbeddings HF:models/clip/modeling_clip.py:202-219: the stride==kernel conv is a GEMM
 * over non-overlapping patches.  pixels float32 NCHW [B,3,img,img] -> patches `dtype`
 * [B*g*g, ldp] with column (c*ps + i)*ps + j, zero-filled up to ldp (ldp >= 3*ps*ps). */
int eavqa_patchify(int dtype, int B, int img, int ps, const float* pixels, void* patches, int64_t ldp, void* stream);
/* x[b,0,:] = cls + pos[0]; x[b,1+t,:] = patch_embed[b*g*g+t,:] + pos[1+t]   (float32 out).
 * patch_embed `dtype` [B*n_patch, ldpe]; cls/pos float32. */
int eavqa_vit_assemble(int dtype, int B, int n_patch, int W, const void* patch_embed, int64_t ldpe,
                       const float* cls, const float* pos, float* x, int64_t ldx, void* stream);

/* ------------------------------------------------- decoder-layer driver ---
 * All `n_layer` pre-LN decoder layers (HF:gpt2 :246-309 / HF:opt :184-254) for Sq NEW positions per sample in one
 * call, with a per-layer KV cache [B, S_max, E]: K/V of the new positions are appended at sequence index row0 and
 * attention (causal, key mask row b at key_mask + b*ld_mask) runs against positions [0, row0+Sq).  Prefill: row0 = 0,
 * Sq = prompt length; decode step t: row0 = prompt + t, Sq = 1 (the reference re-runs the whole sequence instead,
 * clipcap.py:414-419).  x: float32 residual stream [B*Sq, E], updated in place.  Weights are in the packed layouts of
 * eavqa_gemm's b_kc form ([out, in]); `layers` is a HOST array.  workspace: device scratch of at least
 * eavqa_lm_block_workspace_bytes(dtype, B*Sq, E, F) bytes.  Enqueue-only. */
typedef struct {
    const float* ln1_g; const float* ln1_b;
    const void* w_qkv; const float* b_qkv;      /* [3E, E], [3E] */
    const void* w_o; const float* b_o;          /* [E, E], [E] */
    const float* ln2_g; const float* ln2_b;
    const void* w_fc1; const float* b_fc1;      /* [F, E], [F] */
    const void* w_fc2; const float* b_fc2;      /* [E, F], [E] */
    void* k_cache; void* v_cache;               /* [B, S_max, E] in `dtype` */
} eavqa_lm_layer_t;
int64_t eavqa_lm_block_workspace_bytes(int dtype, int rows, int E, int F);
int eavqa_lm_block_forward(int dtype, int n_layer, const eavqa_lm_layer_t* layers, int E, int H, int F, int act, float eps,
                           int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* eavqa_lm_block_forward for an LM whose Linear weights are e4m3 bytes with one scale per tensor (BASELINE configs[4]; models/lm.py
 * Fp8Weight): `layers[l].w_*` point at the bytes ([out, in], k contiguous), `scales[l]` holds the four tensor scales; activations, KV
 * cache and biases as in the bf16 call.  Prefill quantises every GEMM's rows and multiplies on the fp8 matrix cores (eavqa_quantize_rows_fp8
 * + eavqa_gemm_fp8: the arithmetic of the re-forward loop); a decode step (Sq = 1, <= 64 rows) streams the e4m3 weights once through
 * eavqa_gemm_fp8_splitk.  E and F multiples of 128.  workspace >= eavqa_lm_block_fp8_workspace_bytes(B * Sq, E, F). */
typedef struct { float s_qkv, s_o, s_fc1, s_fc2; } eavqa_lm_layer_scales_t;
int64_t eavqa_lm_block_fp8_workspace_bytes(int rows, int E, int F);
int eavqa_lm_block_forward_fp8(int n_layer, const eavqa_lm_layer_t* layers, const eavqa_lm_layer_scales_t* scales, int E, int H, int F, int act,
                               float eps, int B, int Sq, int row0, int S_max, float* x, const int32_t* key_mask, int64_t ld_mask,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* One cached greedy step through all `n_layer` decoder layers of a frozen T5 (HF:models/t5/modeling_t5.py T5Block x n: self-attention with
 * the relative-position bias, cross-attention over the encoder output, (gated) feed-forward; RMSNorm, no biases, unscaled scores), for the
 * ONE new decoder position t - 1 of every sample: q / k / v of the new position, K / V appended to the per-layer cache [B, t_max, inner] at
 * row t - 1, one query at the end of the t cached keys (bias rel_bias[h * rel_ld + (j - (t - 1)) + rel_zero]); cross-attention against
 * cross_kv [B * S, 2 inner] = [K | V] of the encoder output (key mask row b at enc_mask + b * ld_mask).  x: float32 residual stream
 * [B, E] = the input embedding of position t - 1, overwritten; out [B, E] `dtype` = final RMSNorm of the stack.  `layers` is a HOST
 * array; weights in eavqa_gemm's b_kc layout ([out, in]; w_i = [wi_0; wi_1] when gated).  The call sequence of
 * FrozenT5.decode_step (the reference runs HF generate: src/models/vct0.py:452-491), enqueue-only, no allocation. */
typedef struct {
    const float* ln_sa; const void* w_qkv; const void* w_o;            /* [3 inner, E], [E, inner] */
    const float* ln_ca; const void* w_q_ca; const void* w_o_ca;        /* [inner, E], [E, inner] */
    const float* ln_ff; const void* w_i; const void* w_o_ff;           /* [F or 2F, E], [E, F] */
    void* k_cache; void* v_cache;                                       /* [B, t_max, inner] in `dtype` */
    const void* cross_kv;                                               /* [B * S, 2 inner] in `dtype` */
} eavqa_t5_dec_layer_t;
int64_t eavqa_t5_decoder_step_workspace_bytes(int dtype, int B, int E, int inner, int F, int gated);
int eavqa_t5_decoder_step(int dtype, int n_layer, const eavqa_t5_dec_layer_t* layers, const float* ln_final, int E, int inner, int H,
                          int F, int gated, int act, float eps, int B, int t, int t_max, int S, float* x, void* out,
                          const int32_t* enc_mask, int64_t ld_mask, const float* rel_bias, int64_t rel_ld, int rel_zero,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* ---- decode-step primitives (M = batch <= 64 new tokens; replaces the eager per-step loop of
 * `_generate_from_embeddings`, src/models/clipcap.py:414-419, once a KV cache exists) --------------------------------
 * eavqa_gemm_splitk: bf16 A [M,K] x B [N,K]^T, K cut into `ks` slices over workgroups; fp32 partial sums
 * partials[s][m][n] (ld = N, s-major) - no epilogue, the consumer sums the slices in order.  `ks` must divide K / 32;
 * eavqa_gemm_splitk_plan returns the recommended ks (0 if the shape is not supported). */
int eavqa_gemm_splitk_plan(int M, int N, int K);
int eavqa_gemm_splitk(int dtype, int M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb,
                      float* partials, int ks, void* stream);
/* out[m, n] = act(sum_s partials[s][m][n] + bias[n]) (+ residual[m, n]); the N columns are cut into n_seg (1..3) equal
 * segments, segment j written to out_j + m * ld_j (storage `dtype`, or fp32 when out_f32).  With n_seg = 3 and
 * out1 / out2 pointing at the cache rows this is "add the QKV bias, emit q and append k, v" in one pass. */
int eavqa_splitk_finish(int dtype, int M, int N, const float* partials, int ks, const float* bias, int act,
                        const float* residual, int64_t ld_residual, int out_f32, int n_seg,
                        void* out0, int64_t ld0, void* out1, int64_t ld1, void* out2, int64_t ld2, void* stream);
/* x = x_in + bias + sum_s partials[s] (written to x_out when non-NULL); y = LayerNorm(x) * gamma + beta in `dtype`.
 * ks = 0: plain LayerNorm of x_in. */
int eavqa_layernorm_splitk(int dtype, int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks,
                           const float* bias, float* x_out, int64_t ld_out, const float* gamma, const float* beta,
                           float eps, void* y, int64_t ldy, void* stream);

/* The decode step of an LM held in e4m3 (BASELINE configs[4]; the forward it must agree with is eavqa_quantize_rows_fp8 + eavqa_gemm_fp8,
 * HF:models/opt/modeling_opt.py:137-181,228-248 for one new token):
 * eavqa_gemm_fp8_splitk: partials[s][m][n] = a_row_scale[m] * b_scale * sum_{k in slice s} A[m,k] B[n,k], A [M,K] and B [N,K] e4m3 bytes
 *   (K % (128 ks) == 0, lda / ldb % 16 == 0; the block-scaled 16x16x128 MFMA of eavqa_gemm_fp8 with unit block scales): fp32 partial sums with the scales already applied, so eavqa_splitk_finish,
 *   eavqa_layernorm_splitk and eavqa_attention_decode_splitk consume them as they consume the bf16 kernel's; half the weight bytes per
 *   step.  eavqa_gemm_fp8_splitk_plan: recommended ks (0 = unsupported shape).
 * eavqa_layernorm_splitk_fp8: eavqa_layernorm_splitk whose output is the NEXT fp8 GEMM's operand: LayerNorm(x) rounded to bfloat16, then
 *   quantised row-wise exactly as eavqa_quantize_rows_fp8 does (yq e4m3 bytes [rows, ldq], row_scale float32 [rows]). */
int eavqa_gemm_fp8_splitk_plan(int M, int N, int K);
int eavqa_gemm_fp8_splitk(int M, int N, int K, const void* A, int64_t lda, const float* a_row_scale, const void* B, int64_t ldb,
                          float b_scale, float* partials, int ks, void* stream);
int eavqa_layernorm_splitk_fp8(int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks, const float* bias,
                               float* x_out, int64_t ld_out, const float* gamma, const float* beta, float eps, void* yq, int64_t ldq,
                               float* row_scale, void* stream);

/* The same three consumers for a T5 decoder step (HF:models/t5/modeling_t5.py T5Block; the reference reaches it through HF generate,
 * src/models/vct0.py:458-464):
 * eavqa_rmsnorm_splitk: x = x_in + sum_s partials[s] (written to x_out when non-NULL); y = T5LayerNorm(x) (HF:t5 :50-72; no mean, no bias).
 * eavqa_splitk_finish_gated: h[m, c] = act(sum_s P[s][m][c]) * sum_s P[s][m][F + c], P = partial sums [ks][M][2F] of x @ [wi_0; wi_1]^T
 *   (T5DenseGatedActDense, HF:t5 :97-123).
 * eavqa_attention_decode_splitk_rel: eavqa_attention_decode_splitk with T5's additive relative-position bias for the one query at
 *   position Sk - 1 (score(j) += rel_bias[h * rel_ld + (j - (Sk - 1)) + rel_zero], NULL = none) and, with part_cols = H * hd, for partial
 *   sums that hold q ALONE (cross-attention: nothing is appended, k / v = the encoder's projected rows); part_cols = 3 * H * hd: q | k | v
 *   as eavqa_attention_decode_splitk.  No bias vector (T5's projections have none). */
int eavqa_rmsnorm_splitk(int dtype, int rows, int cols, const float* x_in, int64_t ldx, const float* partials, int ks,
                         float* x_out, int64_t ld_out, const float* gamma, float eps, void* y, int64_t ldy, void* stream);
int eavqa_splitk_finish_gated(int dtype, int M, int F, const float* partials, int ks, int act, void* out, int64_t ld, void* stream);
int eavqa_attention_decode_splitk_rel(int dtype, int B, int H, int Sk, int hd, const float* partials, int ks, int part_cols,
                                      void* k, int64_t ldk, void* v, int64_t ldv, int64_t kv_batch_rows, void* o, int64_t ldo,
                                      const int32_t* key_mask, int64_t ld_mask, float scale, const float* rel_bias, int64_t rel_ld,
                                      int rel_zero, void* stream);

/* eavqa_gemm_decode: one weight-streaming GEMM of a decode step WITHOUT partial sums (csrc/decode_direct.hip):
 *   C[m, n] = epilogue(sum_k norm(A)[m, k] * B[n, k]),  M <= 64 rows, B = a frozen bf16 [N, K] weight read once, K % 64 == 0.
 * K is split over the eight waves of a workgroup that owns eavqa_gemm_decode_cols(M, N, K, a_kind, gated) columns of C, so the finished sums
 * exist inside the kernel and the whole epilogue of eavqa_gemm runs there (bias, act, residual, fp32 or bf16 C) - plus:
 *   a_kind 0: A is bf16 [M, K];  1 / 2: A is the fp32 residual stream and LayerNorm (HF:gpt2 :253-257, HF:opt :196-205) / T5 RMSNorm
 *     (HF:t5 :50-72) is applied while loading it: gamma (beta: LayerNorm only) fp32 [K]; the row statistics come from `stats_in`
 *     [M][n_stats_in][2] = (sum, sum of squared deviations from its own mean) of stats_in_cols consecutive columns each - what the
 *     GEMM that produced the stream left in its `stats_out` (n_stats_in = ceil(its N / its eavqa_gemm_decode_cols)); combined in
 *     index order, bitwise reproducible;
 *   gated_rows = F != 0: B is T5's [wi_0; wi_1] ([2F, K]), N = F, C[m, n] = act(row n) * (row F + n) (HF:t5 :97-123);
 *   n_seg (1..3): the N columns are cut into n_seg equal segments, segment j written to out[j] + m * ld_out[j] (n_seg = 3 with out[1] /
 *     out[2] at the cache rows of the new position: "emit q, append k and v" in the QKV projection itself);
 *   stats_out (NULL = skip): [M][ceil(N / cols)][2] statistics of the rows of C for the next normalising consumer.
 * Replaces, for one new token per sample, the nn.Linear / Conv1D matmuls and the LayerNorm in front of them of HF:gpt2 :246-309,
 * HF:opt :184-254, HF:t5 T5Block in the reference's greedy loops (src/models/clipcap.py:414-419, src/models/vct0.py:458-464). */
typedef struct {
    int M, N, K;
    const void* A; int64_t lda; int a_kind;
    const float* gamma; const float* beta; float eps;
    const float* stats_in; int n_stats_in; int stats_in_cols;
    const void* B; int64_t ldb; int gated_rows;
    const float* bias; int act;
    const float* residual; int64_t ld_residual;
    int out_f32; int n_seg; void* out[3]; int64_t ld_out[3];
    float* stats_out;
} eavqa_decode_gemm_t;
int eavqa_gemm_decode_cols(int M, int N, int K, int a_kind, int gated);     /* columns of C per workgroup; 0 = shape not supported */
int eavqa_gemm_decode(const eavqa_decode_gemm_t* args, void* stream);

/* ---- RICES retrieval (src/in_context_example_selection/get_question_knn.py:64-76: faiss.normalize_L2 +
 * IndexFlatIP.search with k = 2048).  The scores are eavqa_gemm in fp32 (queries [Nq, D] x database [Nd, D]^T).
 * eavqa_l2_normalize_rows: x[r, :] /= ||x[r, :]||_2 in place (rows of norm 0 untouched, as faiss).
 * eavqa_topk_rows: per row the k largest scores sorted descending (ties: smaller column first), k <= 2048, k <= cols;
 * out_val float32 [rows, k], out_idx int64 [rows, k] (faiss's D and I). */
int eavqa_l2_normalize_rows(int rows, int cols, float* x, int64_t ld, void* stream);
int eavqa_topk_rows(int rows, int cols, const float* scores, int64_t ld, int k, float* out_val, int64_t* out_idx, void* stream);

/* ---- fp8 path (BASELINE configs[4]: the frozen LM's Linear layers in OCP e4m3 on the block-scaled MFMA) -----------------------
 * eavqa_quantize_rows_fp8: x[r, :] (`dtype`: float32 or bfloat16) -> out[r, :] one e4m3 byte per element, row_scale[r] =
 * max|x[r, :]| / 448 (1 for an all-zero row), out = round-to-nearest-even(x / row_scale) (v_cvt_pk_fp8_f32).  cols % 4 == 0.
 * eavqa_gemm_fp8: C[M,N] = epilogue(alpha * b_scale * a_row_scale[m] * sum_k A(m,k) * B(n,k)), A [M,K] and B [N,K] e4m3 bytes,
 * k contiguous, K % 128 == 0, lda / ldb % 16 == 0; fp32 accumulation in v_mfma_scale_f32_16x16x128_f8f6f4 with unit block
 * scales; epilogue (bias, act, aux_in / aux_out, residual, C) exactly as eavqa_gemm with `dtype` = bfloat16.  Replaces the
 * nn.Linear matmuls of HF:models/opt/modeling_opt.py:137-181,228-248 (forward) and their dgrad when the LM is held in fp8.
 * `tile`: 0 = choose by shape; 1.. force a tile (tests / tools). */
int eavqa_quantize_rows_fp8(int dtype, int rows, int cols, const void* x, int64_t ldx, void* out, int64_t ld_out,
                            float* row_scale, void* stream);
int eavqa_gemm_fp8(int M, int N, int K, const void* A, int64_t lda, const float* a_row_scale, const void* B, int64_t ldb,
                   float b_scale, void* C, int64_t ldc, int out_f32, float alpha, const float* bias, int act,
                   const void* aux_in, void* aux_out, int64_t ld_aux, const float* residual, int64_t ldr, void* stream, int tile);

/* ------------------------------------------------------------- T5 / T0 encoder-decoder (VCT0Model, src/models/vct0.py:301-491) ---
 * T5LayerNorm (HF:models/t5/modeling_t5.py:50-72): y = gamma * x * rsqrt(mean(x^2) + eps); no mean subtraction, no bias.
 * x_kind as in eavqa_layernorm_fwd (1 float32, 0 `dtype`, 2 bfloat16, 3 half); rstd float32 [rows] or NULL. */
int eavqa_rmsnorm_fwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx, const float* gamma, float eps,
                      void* y, int64_t ldy, float* rstd, void* stream);
/* dx = (dres ? dres : 0) + RMSNorm'(dy) (float32; dres may alias dx), frozen gamma; dx_lowp: optional copy of dx in `dtype`. */
int eavqa_rmsnorm_bwd(int dtype, int x_kind, int rows, int cols, const void* x, int64_t ldx, const void* dy, int64_t lddy,
                      const float* gamma, const float* rstd, const float* dres, float* dx, int64_t lddx, void* dx_lowp,
                      int64_t ld_lowp, void* stream);
/* T5DenseGatedActDense (HF:models/t5/modeling_t5.py:97-123): u [rows, 2F] = x @ [wi_0; wi_1]^T (one eavqa_gemm);
 * fwd: h[:, c] = act(u[:, c]) * u[:, F + c];  bwd: du[:, c] = dh * u[:, F + c] * act'(u[:, c]), du[:, F + c] = dh * act(u[:, c]). */
int eavqa_gated_act_fwd(int dtype, int rows, int F, int act, const void* u, int64_t ldu, void* h, int64_t ldh, void* stream);
int eavqa_gated_act_bwd(int dtype, int rows, int F, int act, const void* u, int64_t ldu, const void* dh, int64_t lddh,
                        void* du, int64_t lddu, void* stream);
/* eavqa_attention_fwd / _bwd with T5's relative-position bias (HF:models/t5/modeling_t5.py:217-279, added to the unscaled scores,
 * :312-350): score(i, j) += rel_bias[h * rel_ld + (j - (i + Sk - Sq)) + rel_zero] (float32 table per head over key - query
 * offsets, at least -(Sk - 1) .. Sk - 1; NULL = no bias: the cross-attention).  No packed (cu_seqlens) form.  bfloat16 at head sizes 64 / 80 /
 * 96 / 128: the matrix-core kernels, forward and backward; float32 (and other head sizes): fp32 arithmetic on the vector-ALU kernels. */
int eavqa_attention_fwd_rel(int dtype, int B, int H, int Sq, int Sk, int hd, const void* q, int64_t ldq, const void* k, int64_t ldk,
                            const void* v, int64_t ldv, void* o, int64_t ldo, int64_t q_batch_rows, int64_t kv_batch_rows,
                            const int32_t* key_mask, int64_t ld_mask, int causal, float scale, const float* rel_bias,
                            int64_t rel_ld, int rel_zero, float* lse, void* stream);
int eavqa_attention_bwd_rel(int dtype, int B, int H, int Sq, int Sk, int hd, const void* q, int64_t ldq, const void* k, int64_t ldk,
                            const void* v, int64_t ldv, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                            void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                            const int32_t* key_mask, int causal, float scale, const float* rel_bias, int64_t rel_ld,
                            int rel_zero, const float* lse, float* delta, void* stream);

/* float32 -> `dtype` elementwise copy with row strides (casts the residual stream / pooled rows). */
int eavqa_cast_rows(int dtype, int rows, int64_t cols, const float* x, int64_t ldx, void* y, int64_t ldy, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* EAVQA_H */
