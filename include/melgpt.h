/* melgpt.h - C ABI of libmelgpt_hip.so: the MI355X (gfx950) kernels behind the
 * mel-spectrogram -> VQ-codebook -> GPT hot path of karchkha/MelSpec_GPT_VQVAE.
 *
 * The reference has no FFI layer of its own (it is pure PyTorch); this ABI is the
 * boundary a maintainer binds with ctypes from the reference's Python modules
 * (INTEGRATION.md shows the stubs).  Each entry point names the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; no torch types;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); launches are
 *     asynchronous, no entry point synchronises, allocates or frees;
 *   - return value: MELGPT_OK or a negative MELGPT_ERR_* (arguments are validated on the
 *     host BEFORE anything is launched; the Python shim turns errors into exceptions);
 *   - dtype codes: MELGPT_F32 (parity lane, exact-f32 MFMA) / MELGPT_BF16 (throughput
 *     lane, bf16 storage + f32 accumulation).
 */
#ifndef MELGPT_H
#define MELGPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MELGPT_ABI_VERSION 1

enum { MELGPT_F32 = 0, MELGPT_BF16 = 1 };

enum {
  MELGPT_OK = 0,
  MELGPT_ERR_BAD_ARG = -1,     /* null pointer / non-positive size */
  MELGPT_ERR_UNSUPPORTED = -2, /* shape or dtype outside what the kernel was built for */
  MELGPT_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after the launch */
  MELGPT_ERR_ALIGN = -4        /* pointer / stride not aligned as required */
};

int melgpt_abi_version(void);
const char* melgpt_strerror(int code);
/* Compute units the persistent kernels (one workgroup per CU for a whole launch: the 256-wide GEMM, the wide fused
 * convolution) leave unused, so that RCCL's all-reduce kernels running beside the backward pass of a data-parallel step
 * (Lightning DDP in the reference, GPT_VAE_train.py:172-174) always find free CUs and never delay a persistent
 * workgroup's whole tile list.  0 (default) = use every CU; per process. */
int melgpt_set_reserved_cus(int n);
int melgpt_get_reserved_cus(void);
/* The persistent GEMM has two K loops: a lockstep walk over a ring of five 32 KiB slots (csrc/gemm256.hip; every layout,
 * epilogue and tile-list form) and a two-group ping-pong over 16 KiB half-tiles (csrc/gemm8p.hip; static tile lists,
 * row-major A with either B layout and the weight-gradient form; faster where it applies).  1 (default;
 * MELGPT_GEMM_8P=0 in the environment turns it off) lets a launch take the ping-pong loop when its combination is built,
 * 0 keeps every launch on the ring; same results bit for bit; per process.  The counters say which loop the launches
 * since load took. */
int melgpt_set_gemm_pingpong(int on);
int melgpt_get_gemm_pingpong(void);
int melgpt_gemm_loop_launches(long long* ring, long long* pingpong);

/* ===================================================================== mel frontend
 * wav -> log-mel in one kernel = MelSpectrogram.__call__ + TRANSFORMS
 * (feature_extraction/extract_mel_spectrogram.py:36-38, 141-151) and, optionally, the CenterCrop + 2x-1 of
 * feature_extraction/extract_codes.py:42-43.  librosa-0.8.1 STFT semantics: n_fft = win_length = 1024, periodic Hann,
 * center=True with reflect padding, frames = 1 + n_samples/hop.
 *   wav        (n_clips, n_samples) f32, already padded/truncated to its final length (get_spectrogram :169-173)
 *   mel_basis  (n_mels, 513) f32 = librosa.filters.mel(...) (:26); band_lo/band_hi (n_mels,) first/last non-zero bin
 *   value      clip((log10(max(min_val, m)) * mult - sub + add) / div, clip_lo, clip_hi)     (:143-149)
 *   mel_out    optional (n_clips, n_mels, n_keep) f32, frames [0, n_keep)                      (TrimSpec :150)
 *   tile_out   optional (n_clips, n_mels, crop_len) f32|bf16 = 2*value - 1 of frames [crop0, crop0+crop_len)  */
int melgpt_mel_frontend_fwd(const float* wav, int n_clips, long long n_samples, int n_fft, int hop,
                            const float* mel_basis, const int* band_lo, const int* band_hi, int n_mels,
                            float min_val, float mult, float sub, float add, float div, float clip_lo,
                            float clip_hi, float* mel_out, int n_keep, void* tile_out, int tile_dtype, int crop0,
                            int crop_len, void* stream);
/* The transform tail alone = TRANSFORMS.transforms[1:] (feature_extraction/extract_mel_spectrogram.py:143-150:
 * LowerThresh, Log10, Multiply, Subtract, Add, Divide, Clip, TrimSpec) on mel magnitudes already in memory,
 * mel_in (n_clips, n_mels, n_frames) f32: the same device function the fused kernel ends with; outputs as above. */
int melgpt_mel_transforms_fwd(const float* mel_in, int n_clips, int n_mels, int n_frames, float min_val, float mult,
                              float sub, float add, float div, float clip_lo, float clip_hi, float* mel_out,
                              int n_keep, void* tile_out, int tile_dtype, int crop0, int crop_len, void* stream);

/* ===================================================================== VQ codebook
 * Nearest-neighbour lookup = VectorQuantizer.forward, vqvae/big_model_attn_gan.py:19-54
 * (distances :28-30, argmin :33 first-index tie-break, gather :36-40, MSE terms :43-45,
 * straight-through value :49, code histogram for the perplexity :50-51).
 *
 * Latent element (n, c), n in [0,N), c in [0,D):
 *     z[(n / inner) * stride_outer + (n % inner) * stride_inner + c * stride_c]   (element units)
 * which covers NCHW (inner = H*W, stride_outer = D*H*W, stride_inner = 1, stride_c = H*W) and
 * channels-last / flat (N,D) (inner = N, stride_inner = D, stride_c = 1).
 *
 * codebook   (K, D) f32 row-major.  Built for D = 256, K = 128 (big_model_attn_gan.py:522, minGPT.py:242).
 * indices    out (N,) int64 - row-major order n = b*H*W + h*W + w, as info[2] in the reference.
 * quantized  optional out, same dtype/addressing as z: x + (E[idx] - x)   (:49 forward value)
 * sq_err     optional out (grid,) f32: per-workgroup partial sums of (E[idx]-x)^2, reduced in
 *            fixed order by melgpt_vq_finalize (deterministic; no float atomics)
 * histogram  optional in/out (K,) int32 code counts (integer atomics); caller zeroes it.
 * `grid_out` (host pointer, optional) receives the number of workgroups = valid length of sq_err.
 * F32 lane: the cross term is an exact k-ordered f32 FMA chain (f32 MFMA); oracle/vq_argmin.c
 * reproduces it bit for bit.  BF16 lane: z is bf16, the codebook is rounded to bf16 in-kernel.
 */
int melgpt_vq_argmin_fwd(const void* z, int z_dtype, int64_t n_vectors, int dim, int64_t inner,
                         int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                         const float* codebook, int num_codes, int64_t* indices, void* quantized,
                         float* sq_err, int32_t* histogram, int* grid_out, void* stream);

/* Same, plus `distances`: optional out (N,K) f32 = the matrix of :28-30 (inspection / bit-exact tests). */
int melgpt_vq_argmin_fwd_ex(const void* z, int z_dtype, int64_t n_vectors, int dim, int64_t inner,
                            int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                            const float* codebook, int num_codes, int64_t* indices, void* quantized,
                            float* sq_err, int32_t* histogram, float* distances, int* grid_out,
                            void* stream);

/* max workgroups melgpt_vq_argmin_fwd will ever use (size sq_err with it). */
int melgpt_vq_max_grid(void);

/* Prepared codebook image for the bf16 lookup (indices only - the hot path of feature_extraction/extract_codes.py:
 * 48-50 and of the training step's VQ-encode).  Everything a launch of melgpt_vq_argmin_fwd derives from the codebook
 * (bf16 rounding, swizzled MFMA fragment layout, the |e|^2 chains) is computed ONCE per codebook version:
 *   conv_weight == NULL : plain image; melgpt_vq_lookup_image(fused = 0) returns the same bits as the bf16 lane of
 *                         melgpt_vq_argmin_fwd.
 *   conv_weight != NULL : the 1x1 `quant_conv` (big_model_attn_gan.py:578,607; weight (D,D) out x in, bias (D) or
 *                         NULL) is folded in:  argmin_k |W x + b - e_k|^2 = argmin_k (|e_k|^2 - 2 b.e_k) - 2 x.(W^T e_k),
 *                         so the lookup (fused = 1) runs on the ENCODER's output x and z = quant_conv(x) is never formed.
 *                         with_lo = 1 keeps W^T e_k as two bf16 planes (hi + lo, 2^-17 relative); 0 = one plane.
 * `image` is a device buffer of melgpt_vq_image_bytes(with_lo) bytes, 16-byte aligned; D = 256, K = 128.
 * melgpt_vq_lookup_image: z bf16, channel-contiguous (stride_c = 1) latents in the addressing of z above;
 * histogram optional (K,) int32 in/out. */
int64_t melgpt_vq_image_bytes(int with_lo);
int melgpt_vq_prepare_image(const float* codebook, int num_codes, int dim, const float* conv_weight,
                            const float* conv_bias, int with_lo, void* image, void* stream);
int melgpt_vq_lookup_image(const void* z, int64_t n_vectors, int dim, int64_t inner, int64_t stride_outer,
                           int64_t stride_inner, int64_t stride_c, const void* image, int with_lo, int fused,
                           int64_t* indices, int32_t* histogram, void* stream);

/* loss = mse + commitment*mse (:43-45), perplexity = exp(-sum p log(p+1e-10)) (:50-51).
 * out[0] = loss, out[1] = perplexity, out[2] = mse. */
int melgpt_vq_finalize(const float* sq_err, int n_partials, const int32_t* histogram, int num_codes,
                       int64_t n_vectors, int dim, float commitment_cost, float* out, void* stream);

/* get_codebook_entry, big_model_attn_gan.py:56-71: out(n, c) = codebook[indices[n]][c], same
 * addressing convention as z above. */
int melgpt_vq_gather(const int64_t* indices, int64_t n_vectors, const float* codebook, int num_codes,
                     int dim, void* out, int out_dtype, int64_t inner, int64_t stride_outer,
                     int64_t stride_inner, int64_t stride_c, void* stream);

/* one-hot `encodings` (N,K) f32 of :36-37 (API-compat output only; the hot path never builds it). */
int melgpt_vq_onehot(const int64_t* indices, int64_t n_vectors, int num_codes, float* encodings,
                     void* stream);

/* backward of VectorQuantizer.forward through `quantized` (STE, :49) and `loss` (:43-45):
 *   dz(n,c)        = g_q(n,c) + g_loss * 2*commitment/(N*D) * (x - E[idx])
 *   dcodebook[k,c] += sum_{n: idx[n]=k} g_loss * 2/(N*D) * (E[k] - x(n,c))       (f32 atomics)
 * g_q may be NULL (treated as 0); g_loss is a device scalar (may be NULL = 0). */
int melgpt_vq_bwd(const void* z, const void* g_quantized, int dtype, int64_t n_vectors, int dim,
                  int64_t inner, int64_t stride_outer, int64_t stride_inner, int64_t stride_c,
                  const float* codebook, int num_codes, const int64_t* indices, const float* g_loss,
                  float commitment_cost, void* dz, float* dcodebook, void* stream);

/* ===================================================================== GEMM (+ fused epilogues)
 * C[b][m,n] = epi( alpha * sum_k A[b](m,k) * B[b](n,k) ), row-major C with leading dimension ldc.
 *   a_kmajor = 0: A(m,k) = A[m*lda + k]   (nn.Linear input, reduction index contiguous)
 *   a_kmajor = 1: A(m,k) = A[k*lda + m]   (transposed operand, e.g. dY^T in a weight gradient)
 *   b_kmajor likewise for B(n,k): 0 = nn.Linear weight (out,in) as stored by the reference
 *   (transformer/minGPT.py:56-63,100-105,149); 1 = the same weight used in the input-gradient product.
 * epilogue order: v = alpha*acc + bias[n]; C2 = v (optional pre-activation copy);
 *   act: MELGPT_ACT_GELU -> exact-erf GELU (nn.GELU(), minGPT.py:102);
 *        MELGPT_ACT_GELU_GRAD -> v *= gelu'(R[m,n])  (R is the saved pre-activation, not a residual);
 *        MELGPT_ACT_GELU_DACT -> as MELGPT_ACT_GELU, but C2 receives gelu'(v) instead of v: the forward pass of
 *                                Linear -> GELU saves the activation's DERIVATIVE (the exponential is shared with the
 *                                activation itself), so that the backward epilogue is a multiplication:
 *        MELGPT_ACT_MUL       -> v *= R[m,n]  (R = the saved derivative, not a residual);
 *   dropout(drop_p) with Philox4x32-10 keyed by (seed, stream_id, element index)  (minGPT.py:88,104);
 *   v += R[m,n] (residual, minGPT.py:115,117);  if accumulate: v += C[m,n];  store as dtype or f32.
 * dtype: MELGPT_F32 -> exact f32 MFMA; MELGPT_BF16 -> bf16 operands, f32 accumulate.  bias is f32.
 * Alignment: all pointers 16 B; N % 4 == 0; K, lda, ldb and batch strides multiples of 16 bytes.   */
enum { MELGPT_ACT_NONE = 0, MELGPT_ACT_GELU = 1, MELGPT_ACT_GELU_GRAD = 2, MELGPT_ACT_GELU_DACT = 3, MELGPT_ACT_MUL = 4 };

int melgpt_gemm(const void* A, int a_kmajor, long long lda, long long strideA, const void* B, int b_kmajor,
                long long ldb, long long strideB, void* C, long long ldc, long long strideC, int M, int N,
                int K, int batch, int dtype, int out_f32, int accumulate, float alpha, const float* bias,
                int act, const void* R, long long ldr, long long strideR, void* C2, float drop_p,
                unsigned long long seed, unsigned stream_id, void* stream);

/* Weight gradient AND bias gradient of y = x W^T + b in one launch (torch.nn.Linear backward, the c_attn / c_proj / mlp
 * sites of transformer/minGPT.py:61-66,99-104): dW_part[z] (M x N, f32) = dY_z^T X_z over batch z's K reduction rows
 * (both operands K-major: dY is (K, M), X is (K, N), element strides lda / ldb, batch strides strideA / strideB), and
 * rowsum_part (melgpt_wgrad_rowsum_rows(N, batch) rows of M floats, row stride ld_rowsum) = partial sums over the
 * reduction rows of dY - one more MFMA per A fragment in the same K loop instead of a pass of its own over dY.  The caller
 * adds the batches of dW_part and ALL rows of rowsum_part in fixed order (melgpt_reduce_rows).  16-bit lane, shapes that
 * run on the persistent kernel; otherwise MELGPT_ERR_UNSUPPORTED before any launch (callers then use melgpt_gemm +
 * melgpt_colsum). */
int melgpt_wgrad_rowsum_rows(int N, int batch);
int melgpt_wgrad_rowsum(const void* dY, long long lda, long long strideA, const void* X, long long ldb, long long strideB,
                        float* dW_part, long long ldc, long long strideC, int M, int N, int K, int batch, int dtype,
                        float* rowsum_part, long long ld_rowsum, void* stream);

/* 3x3 / 1x1 convolution as implicit GEMM over an NHWC activation (torch.nn.Conv2d sites of
 * vqvae/big_model_attn_gan.py:85-99,108-112,151-159,176-186,247-251,313-317,403-422,578-579).
 *   x (B,H,W,Cin) dtype; wpack (Cout, KH, KW, Cin) dtype (repacked from the reference's OIHW f32);
 *   y (B,OH,OW,Cout); bias (Cout) f32 or NULL; residual (B,OH,OW,Cout) or NULL (ResnetBlock x+h, :135).
 *   pad_t/pad_l: zero padding at top/left (bottom/right implied by OH/OW: Downsample pads (0,1,0,1), :152);
 *   upsample = 1 folds F.interpolate(scale=2, 'nearest') (:183) into the input addressing.
 *   Cin must be a multiple of 64 (bf16) / 32 (f32).                                                 */
int melgpt_conv2d_nhwc(const void* x, int B, int H, int W, int Cin, const void* wpack, int Cout, int KH,
                       int KW, int stride, int pad_t, int pad_l, int OH, int OW, int upsample,
                       const float* bias, const void* residual, void* y, int dtype, void* stream);

/* Backward of the VQ-VAE's convolution and normalisation sites (LitVQVAE.forward is differentiable in the reference,
 * vqvae/big_model_attn_gan.py:622-634; no scored configuration trains the VQ-VAE, so these are correctness-first compositions
 * of the forward kernels - csrc/vqvae_bwd.hip).  NHWC, both numerics lanes.
 *   melgpt_conv3x3_bwd_data:   dx (B,H,W,Cin) = gradient of conv3x3(stride 1, pad 1) w.r.t. its input, from dy (B,H,W,Cout) and
 *                              the forward's packed weight (Cout,3,3,Cin).  Cout % 64 == 0 (16-bit) / % 32 (f32), Cin % 8 == 0.
 *   melgpt_conv3x3_bwd_weight: dw (Cout,3,3,Cin) f32 = gradient w.r.t. the packed weight, dbias (Cout) f32 or NULL = column sums
 *                              of dy; deterministic (fixed-order partial sums).  Cin, Cout rows 16-byte multiples.
 *   workspace: melgpt_conv3x3_bwd_workspace(...) bytes, 16-byte aligned (serves either call).
 *   melgpt_groupnorm_swish_bwd: gradient of y = swish?(GroupNorm(32)(x) * gamma + beta) (Normalize + nonlinearity, :139-140,
 *                              :164-166) from the forward's statistics (mean, rstd: (B*32,) f32): dx in dtype, dgamma / dbeta (C)
 *                              f32 or NULL; workspace melgpt_groupnorm_swish_bwd_workspace(B, HW, C) bytes. */
long long melgpt_conv3x3_bwd_workspace(int B, int H, int W, int Cin, int Cout, int dtype);
int melgpt_conv3x3_bwd_data(const void* dy, const void* wpack, void* dx, int B, int H, int W, int Cin, int Cout,
                            void* workspace, int dtype, void* stream);
int melgpt_conv3x3_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                              void* workspace, int dtype, void* stream);
/* The same for Downsample's convolution (F.pad (0,1,0,1) + conv3x3 stride 2, :151-159): x (B,H,W,Cin), dy (B,OH,OW,Cout) with
 * OH = (H - 2) / 2 + 1.  bwd_data: a stride-1 convolution of the zero-dilated gradient; bwd_weight: the four phase images of the
 * padded input make every tap a row offset again. */
long long melgpt_conv3x3_s2_bwd_workspace(int B, int H, int W, int Cin, int Cout, int dtype);
int melgpt_conv3x3_s2_bwd_data(const void* dy, const void* wpack, void* dx, int B, int H, int W, int Cin, int Cout,
                               void* workspace, int dtype, void* stream);
int melgpt_conv3x3_s2_bwd_weight(const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int Cin, int Cout,
                                 void* workspace, int dtype, void* stream);
/* Upsample (nearest x2 + conv3x3, :171-186): y (B,2H,2W,C) = the x2 tensor itself (the weight gradient's operand; the forward
 * folds it into the addressing), and its adjoint y (B,H,W,C) = 2 x 2 block sums of x (B,2H,2W,C). */
int melgpt_upsample2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
int melgpt_sumpool2_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* AttnBlock's softmax (:438-440) backward: dscores[r,c] = scale * probs[r,c] * (dprobs[r,c] - sum_c' dprobs[r,c'] probs[r,c']) for
 * c < n, zeros for n <= c < ld_dscores; probs / dscores in dtype, dprobs f32. */
int melgpt_softmax_bwd_rows(const void* probs, long long ld_probs, const float* dprobs, long long ld_dprobs, int n, long long rows,
                            float scale, void* dscores, long long ld_dscores, int dtype, void* stream);
/* im2col of a ONE-channel image for a 3 x 3 / pad 1 convolution: out ((B H W) x 32), column t < 9 = the tap (t / 3 - 1, t % 3 - 1),
 * the rest zeros - the gradient side of Encoder.conv_in (1 -> ch, :203-207) and Decoder.conv_out (ch -> 1, :355-359) as GEMMs. */
int melgpt_im2col_c1(const void* img, void* out, int B, int H, int W, int dtype, void* stream);
long long melgpt_groupnorm_swish_bwd_workspace(int B, int HW, int C);
int melgpt_groupnorm_swish_bwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const void* dy, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int swish,
                               float* workspace, int dtype, void* stream);

/* ===================================================================== attention (minGPT.py:72-90)
 * q/k/v: (B*T, >= H*64) matrices with row stride ld (elements), head h in columns [64h, 64h+64) - i.e. the
 * three column blocks of a packed QKV projection; no (B,H,T,hs) transposes are ever materialised.
 * mask: key <= query, plus the fully visible n_unmasked x n_unmasked corner (minGPT.py:65-69).
 * out (B*T, H*64) row stride ldo = softmax(mask(q k^T / sqrt(hs))) -> dropout(p) -> @ v, heads merged (:85).
 * lse (B,H,T) f32 is saved for the backward.  att: optional (B,H,T,T) f32 = post-softmax PRE-dropout
 * probabilities, the tensor CausalSelfAttention.forward returns (:90); NULL skips it.
 * Built for head_size 64 (all reference configs: 1024/16, 1472/23) and T <= 288 (block_size 265/266). */
int melgpt_attn_fwd(const void* q, const void* k, const void* v, long long ld, void* out, long long ldo,
                    float* lse, float* att, int B, int H, int T, int head_size, int n_unmasked, float drop_p,
                    unsigned long long seed, unsigned stream_id, int dtype, void* stream);

/* one KV-cached decoding step (replaces the full re-forward per sampled token of minGPT.py:293-360 /
 * decoders.py:89-123): qkv = the new token's packed projection rows (B, 3C) [key|query|value], row stride ld.
 * Appends its key / value at position `pos` of the caches (B * Tmax * C elements each, owned by this entry point: laid out
 * head-major, (B, H, Tmax, 64), so that a (batch, head)'s rows are contiguous; callers only allocate them) and writes
 * out (B, C) = softmax(q . K[0..pos] / sqrt(hs)) @ V[0..pos], heads merged.  att_row: optional (B, H, Tmax) f32
 * probabilities of this row (entries > pos untouched).  head_size 64, Tmax <= 320, eval mode (no dropout).
 * pos_dev: optional device int; when given the position is read from it (one captured HIP graph then serves every
 * decoding step) and `pos` is ignored. */
int melgpt_attn_decode(const void* qkv, long long ld, void* kcache, void* vcache, int B, int H, int head_size,
                       int Tmax, int pos, const int* pos_dev, void* out, float* att_row, int dtype, void* stream);
/* skinny-M linear layer of a decode step: y (M,N) = epi(x (M,K) @ W (N,K)^T + bias) (+ residual), W = nn.Linear.weight
 * layout; act in {MELGPT_ACT_NONE, MELGPT_ACT_GELU (exact erf)}; y in `dtype`, or f32 when out_f32 (always f32 for
 * dtype f32).  One wave per 4 output columns streams the weights once with every CU busy; M is walked 16 rows at a
 * time (weights re-read from L2).  N % 4 == 0, K % (16 / sizeof(T)) == 0.
 * ln_gamma / ln_beta (K,) f32, both or neither: the rows are LayerNorm-ed (eps ln_eps, minGPT.py:97-98) on the way in,
 * y = W LN(x) - the pre-LN of a transformer block folded into its qkv / fc1 layer. */
int melgpt_gemv_rows(const void* x, long long ldx, const void* W, long long ldw, const float* bias,
                     const void* residual, long long ldr, void* y, long long ldy, int M, int N, int K, int act,
                     int dtype, int out_f32, const float* ln_gamma, const float* ln_beta, float ln_eps, void* stream);
/* the same layer for 5 .. 128 rows, bf16: one workgroup per 16 output columns, its four waves split K and stream their part
 * of the weights once against all rows as MFMA 16x16x32 (decode steps at batch 5 .. 128; the tiled GEMM would run such a
 * problem on N / 128 workgroups).  M <= 128, N % 16 == 0, K % 128 == 0; residual / y rows 8-byte aligned.
 * ln_gamma / ln_beta (both or neither, as in melgpt_gemv_rows): y = W LN(x), statistics derived inside the kernel;
 * only for K in {512, 1024} and M <= 64, otherwise MELGPT_ERR_UNSUPPORTED (normalise first). */
int melgpt_linear_skinny(const void* x, long long ldx, const void* W, long long ldw, const float* bias,
                         const void* residual, long long ldr, void* y, long long ldy, int M, int N, int K, int act,
                         int dtype, int out_f32, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         void* stream);
/* The same product for 9 .. 128 rows with every byte a workgroup needs requested up front (decode steps at batch 16 ..
 * 128, transformer/minGPT.py:293-360 with a KV cache): a workgroup owns 16 output columns x one 1024-wide slice of K;
 * its rows of x go to LDS by LDS-DMA, its weights to registers, one wait, then MFMAs on on-chip operands.  K must be a
 * multiple of 1024; K > 1024 runs as K / 1024 slices whose f32 partial tiles (melgpt_linear_lds_workspace bytes) are
 * summed in slice order by a second launch that applies bias / GELU / residual.
 * Pre-LayerNorm folded in (ln_c1 / ln_c2 non-NULL, K = 1024): W is then the PREPARED weight W' of melgpt_ln_fold_prepare
 * and y = rstd (W' x - mu c1) + c2 = W LN(x) + b with the row statistics taken inside the launch; `bias` is ignored. */
long long melgpt_linear_lds_workspace(int M, int N, int K);
int melgpt_linear_lds(const void* x, long long ldx, const void* W, long long ldw, const float* bias, const void* residual,
                      long long ldr, void* y, long long ldy, int M, int N, int K, int act, int dtype, int out_f32,
                      const float* ln_c1, const float* ln_c2, float ln_eps, void* workspace, void* stream);
/* once per (weight, LayerNorm, bias) version: W' (N, K) = W diag(gamma) in the 16-bit format, c1[n] = sum_k W'[n][k],
 * c2[n] = sum_k W[n][k] beta[k] + bias[n] (bias may be NULL) - the block's pre-LN (transformer/minGPT.py:108-117: ln1
 * before attn's key/query/value, ln2 before mlp[0]; ln_f before head :187-188) folded into the Linear that follows it. */
int melgpt_ln_fold_prepare(const void* W, long long ldw, const float* bias, const float* gamma, const float* beta, int N,
                           int K, int dtype, void* W_folded, float* c1, float* c2, void* stream);

/* graph-replayed decoding helpers: x[b,:] = tok_emb[idx[b]] + pos_emb[*pos_dev] (minGPT.py:170-180 for one position);
 * *counter += 1 */
int melgpt_embed_decode(const long long* idx, const float* tok_emb, const float* pos_emb, const int* pos_dev, int B,
                        int C, int V, void* out, int dtype, void* stream);
int melgpt_incr_i32(int* counter, void* stream);

/* backward: dq/dk/dv (row stride ldg) from dout; probabilities are recomputed from lse, the dropout mask is
 * regenerated from (seed, stream_id).  delta (B,H,T) f32 is workspace (rowsum(dout*out)). Deterministic. */
int melgpt_attn_bwd(const void* q, const void* k, const void* v, long long ld, const void* out,
                    const void* dout, long long ldo, const float* lse, float* delta, void* dq, void* dk,
                    void* dv, long long ldg, int B, int H, int T, int head_size, int n_unmasked, float drop_p,
                    unsigned long long seed, unsigned stream_id, int dtype, void* stream);
/* 16-bit lane, causal mask (n_unmasked == 0), T <= 272: melgpt_attn_bwd runs as ONE launch that evaluates the
 * probabilities once (dK / dV in registers, dS through LDS for dQ).  on != 0 keeps the two-kernel path (dQ, then dK / dV)
 * that every other case uses; returns the previous setting.  MELGPT_ATTN_BWD_TWO_PASS=1 in the environment does the same. */
int melgpt_set_attn_bwd_two_pass(int on);
/* Forward kernel of the 16-bit lane when no attention map is asked for: -1 (default) = by shape - 32-row query tiles with
 * the logits of a tile row kept in registers (attn_fwd32_kernel) for the full-square mask (n_unmasked >= T: the GPT-VAE
 * encoder, encoders.py:21-30), the 16-row two-pass kernel under the causal mask, where the two tie; 1 / 0 = force the
 * 32-row / the 16-row kernel for every shape (the 16-row one always serves the f32 lane and the `att` output).  Returns the
 * previous mode; MELGPT_ATTN_FWD32=0|1 in the environment sets the initial one.  Same dropout bits from either kernel. */
int melgpt_set_attn_fwd32(int mode);

/* ===================================================================== row kernels
 * nn.LayerNorm(C, eps) (minGPT.py:97-98,141): y = (x-mean)*rstd*gamma+beta; mean/rstd (M,) f32 saved. */
int melgpt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                         float* rstd, long long M, int C, float eps, int dtype, void* stream);
/* dx = add_in + LN'(dy); dgamma/dbeta (+)= column sums (two-stage, deterministic).  workspace: f32
 * [melgpt_layernorm_bwd_nwaves(M) * 2 * C].  dgamma/dbeta may both be NULL. */
int melgpt_layernorm_bwd_nwaves(long long M);
int melgpt_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                         const float* rstd, const void* add_in, void* dx, float* dgamma, float* dbeta,
                         int accumulate, float* workspace, long long M, int C, int dtype, void* stream);
/* The same pass with a second output: dx_masked = dropout_apply(dx, drop_p, seed, stream_id) - the mask replay of the
 * branch that consumes dx next (the Block's backward, minGPT.py:107-119: x + drop(branch(LN(x)))), bit for bit what
 * melgpt_dropout_apply makes of the stored dx.  dx_masked NULL = melgpt_layernorm_bwd. */
int melgpt_layernorm_bwd_masked(const void* dy, const void* x, const float* gamma, const float* mean,
                                const float* rstd, const void* add_in, void* dx, float* dgamma, float* dbeta,
                                int accumulate, float* workspace, long long M, int C, void* dx_masked, float drop_p,
                                unsigned long long seed, unsigned stream_id, int dtype, void* stream);
/* out[n] (+)= sum_m a[m,n] (bias gradients).  workspace: f32 [melgpt_colsum_rows() * N]. */
int melgpt_colsum_rows(void);
int melgpt_colsum(const void* a, long long M, int N, long long lda, float* out, int accumulate,
                  float* workspace, int dtype, void* stream);
/* embedding stem (minGPT.py:170-180, 207-212): out[b,t,:] = drop((t < n_pre ? PRE : tok_emb[idx]) + pos_emb[t]);
 * idx is (B,Tt) int64 with row stride idx_ld (so z_indices[:, :-1] of :279 needs no copy);
 * PRE = pre_table[pre_idx[b*n_pre+t]] (GPTClass.embedder) or pre_vals[b,t,:] (explicit embeddings, f32). */
int melgpt_embed_fwd(const long long* idx, const float* tok_emb, const float* pos_emb,
                     const long long* pre_idx, const float* pre_table, const float* pre_vals, int n_pre, int B,
                     int Tt, long long idx_ld, int C, int V, void* out, int dtype, float drop_p,
                     unsigned long long seed, unsigned stream_id, void* stream);
/* gradients of the stem; every table row is reduced by one workgroup in ascending position order (no atomics) */
int melgpt_embed_bwd(const void* dx, const long long* idx, long long idx_ld, const long long* pre_idx, int n_pre,
                     int B, int Tt, int C, int V, int n_pre_rows, float* tok_grad, float* pos_grad, float* pre_table_grad,
                     float* pre_vals_grad, int accumulate, int dtype, float drop_p, unsigned long long seed,
                     unsigned stream_id, void* stream);
/* rows of the one-hot matrix behind  d tok_emb = OneHot^T @ dX  (zero rows for the n_pre prepended positions) */
int melgpt_onehot_rows(const long long* idx, long long idx_ld, int B, int Tt, int n_pre, int V, void* out,
                       int dtype, void* stream);
/* out[c] (+)= scale * sum_{r<R} partials[r*ld + c], fixed summation order (split-K / two-stage reductions) */
int melgpt_reduce_rows(const float* partials, int R, long long ld, long long ncols, float* out, int accumulate,
                       float scale, void* stream);
/* melgpt_reduce_rows for TWO jobs in one launch (scale 1): a weight gradient's split-K slabs and the row-sum partials of its
 * bias gradient behind melgpt_wgrad_rowsum (transformer/minGPT.py:56-63,100-105 backward).  The same bits as two calls. */
int melgpt_reduce_rows_pair(const float* part_a, int Ra, long long lda, long long ncols_a, float* out_a, int accumulate_a,
                            const float* part_b, int Rb, long long ldb, long long ncols_b, float* out_b, int accumulate_b,
                            void* stream);
/* F.cross_entropy pieces (minGPT.py:197,416; decoders.py:64-68): loss_rows[m] = lse[m] - logits[m,target[m]] */
int melgpt_cross_entropy_fwd(const float* logits, long long ld, const long long* target, long long M, int V,
                             float* loss_rows, float* lse, void* stream);
/* dlogits[m,v] = (softmax - onehot) * g_rows[m / g_group] * (*g_scalar) * g_scale  (g_rows / g_scalar may be NULL = 1;
 * g_group = tokens per sequence when the upstream gradient is per sequence, decoders.py:68) */
int melgpt_cross_entropy_bwd(const float* logits, long long ld, const long long* target, const float* lse,
                             const float* g_rows, int g_group, const float* g_scalar, float g_scale, long long M,
                             int V, void* dlogits, long long ldd, int dtype, void* stream);
/* out[r] = scale * sum_{j<n} in[r*n+j]: per-sequence sums of per-token losses (decoders.py:68 `.sum(-1)`) */
int melgpt_group_sum_f32(const float* in, long long groups, int n, float scale, float* out, void* stream);
/* one autoregressive decoding step (minGPT.py:345-358, decoders.py:108-121): logits (rows,V<=1024) f32 ->
 * /temperature -> top-k filter (top_k <= 0: off) -> softmax -> argmax (do_sample = 0) or one multinomial draw
 * (Philox uniform keyed by seed, step, row).  out (rows,) int64; probs_out optional (rows,V). */
int melgpt_sample_logits(const float* logits, long long ld, int rows, int V, float temperature, int top_k,
                         int do_sample, unsigned long long seed, unsigned step, long long* out, float* probs_out,
                         void* stream);
/* the same step with its number read on the device (graph-replayed decoding): step = *pos_dev + step_offset; the
 * picked token also goes to seq[row * seq_ld + *pos_dev] when seq is given. */
int melgpt_sample_logits_dev(const float* logits, long long ld, int rows, int V, float temperature, int top_k,
                             int do_sample, unsigned long long seed, const int* pos_dev, int step_offset,
                             long long* out, long long* seq, long long seq_ld, void* stream);
/* GPTEncoder.reparameterize + KL (encoders.py:62-104): stats (B,2nz) = [mu | logvar]; z = mu + eps*exp(logvar/2);
 * KL[b] = 0.5*sum(mu^2 + exp(logvar) - logvar - 1).  gen_eps != 0: eps (B,ns,nz) is DRAWN in-kernel (N(0,1) from
 * Philox + Box-Muller) and written; gen_eps == 0: eps is an input. */
int melgpt_vae_reparam_fwd(const float* stats, float* eps, int gen_eps, unsigned long long seed, int B, int ns,
                           int nz, float* z, float* kl, void* stream);
int melgpt_vae_reparam_bwd(const float* stats, const float* eps, const float* dz, const float* dkl, int B, int ns,
                           int nz, float* dstats, void* stream);
/* GPTEncoder.eval_inference_dist (encoders.py:106-134) and the density table inside calc_mi (:154-163):
 * log N(z; mu, exp(logvar)), statistics rows mu / logvar (X, nz) f32 at row pitch ld_stats.  pairwise == 0: z (X, S, nz),
 * out[x][s] = density of z[x][s] under row x's own statistics (out (X, S)); pairwise != 0: z (S, nz),
 * out[i][x] = density of z[i] under row x's statistics (out (S, X)). */
int melgpt_gauss_log_density(const float* z, const float* mu, const float* logvar, long long ld_stats, int X, int S,
                             int nz, int pairwise, float* out, void* stream);
/* GPTEncoder.calc_mi (encoders.py:136-170; utils.log_sum_exp :6-19): the mutual-information estimate
 * E log q(z|x) - E log q(z) with one reparameterised draw per row and the batch mixture as aggregate posterior.
 * eps (B, nz): the draw's noise - an input (gen_eps == 0) or drawn in-kernel and written (as melgpt_vae_reparam_fwd).
 * workspace: B * nz + B floats.  mi: one f32. */
int melgpt_vae_calc_mi(const float* mu, const float* logvar, long long ld_stats, float* eps, int gen_eps,
                       unsigned long long seed, int B, int nz, float* workspace, float* mi, void* stream);
int melgpt_sum_f32(const float* in, long long n, float scale, float* out, int accumulate, void* stream);
/* y = keep(x)/(1-p) with the same Philox mask an epilogue used for element i of a contiguous tensor */
int melgpt_dropout_apply(const void* x, void* y, long long n, float drop_p, unsigned long long seed,
                         unsigned stream_id, int dtype, void* stream);
/* the same replay AND out[n] (+)= sum_m y[m, n] in one pass over an (M, N) row-major tensor: the backward of
 * drop(linear(.)) (minGPT.py:88-89,101-105) needs the masked gradient and, for the bias, its column sums; the sums are
 * taken on y as stored and in melgpt_colsum's order (same bits as melgpt_dropout_apply followed by melgpt_colsum).
 * workspace: melgpt_colsum_rows() * N floats. */
int melgpt_dropout_apply_colsum(const void* x, void* y, long long M, int N, float drop_p, unsigned long long seed,
                                unsigned stream_id, float* out, int accumulate, float* workspace, int dtype,
                                void* stream);
int melgpt_cast(const void* x, int src_dtype, void* y, int dst_dtype, long long n, void* stream);
/* p[0 .. bytes) = 0 on `stream` (what `tensor.zero_()` does in the reference's host code, e.g. the F.pad / zero-initialised
 * buffers around minGPT.py:170-180; here: guard rows of the AttnBlock's q|k|v buffer, the prepended positions' slice of
 * the positional-embedding gradient) - so that no framework fill kernel runs inside a step. */
int melgpt_zero_bytes(void* p, long long bytes, void* stream);
/* torch.optim.AdamW step (minGPT.py:660-664) fused over a flat f32 buffer; optional bf16 shadow copy. */
int melgpt_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                 long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                 float grad_scale, void* stream);

/* VQ code ordering (minGPT.py:387-394 get_x, :431-456 make_idx/code_reader): (B,H,W) row-major <-> (B, W*H)
 * time-major sequence, position p = w*H + h.  reverse = 1 is code_reader(reverse=True). int64 in/out. */
int melgpt_codes_permute(const long long* in, long long* out, int B, int H, int W, int reverse, void* stream);

/* ===================================================================== VQ-VAE encoder / decoder pieces
 * All activations are NHWC ((B, H*W, C) row-major) in `dtype`.
 * GroupNorm(32, C, eps) (big_model_attn_gan.py:139-140): statistics per (batch, group) in two deterministic
 * stages (f32 partials per 512-pixel chunk, combined in f64); workspace f32 [B * nchunks(HW) * 32 * 2].   */
int melgpt_groupnorm_nchunks(int HW);
int melgpt_groupnorm_stats(const void* x, int B, int HW, int C, float eps, float* mean, float* rstd,
                           float* workspace, int dtype, void* stream);
/* y = act((x - mean)*rstd*gamma + beta), act = swish x*sigmoid(x) (:164-166) when swish != 0 */
int melgpt_groupnorm_apply(const void* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, void* y, int B, int HW, int C, int swish, int dtype, void* stream);
/* Normalize + nonlinearity (:139-140,164-166) of a SMALL image (H W <= 1152, 128-byte channel slabs, a group >= 16 bytes:
 * the 256 / 512-channel levels at 10 x 106 and 5 x 53) in ONE launch that reads the tensor once - statistics and
 * normalisation out of registers; mean / rstd (B*32,) f32: optional outputs.  MELGPT_ERR_UNSUPPORTED, nothing launched, for
 * every other shape: melgpt_groupnorm_stats + melgpt_groupnorm_apply then. */
int melgpt_groupnorm_fused(const void* x, const float* gamma, const float* beta, void* y, int B, int HW, int C,
                           float eps, int swish, float* mean, float* rstd, int dtype, void* stream);
/* Encoder.conv_in (:203-207): 3x3, pad 1, ONE input channel; x (B,H,W) x_dtype; w (Cout,1,3,3) f32 as stored. */
int melgpt_conv_in_c1(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype, int B,
                      int H, int W, int Cout, void* stream);
/* The same stem conv AND the GroupNorm(32) statistics (mean, rstd: (B*32,) f32, eps as torch.nn.GroupNorm) of its
 * output in one pass - what the first ResnetBlock's norm1 (vqvae/big_model_attn_gan.py:117 on conv_in's output, :259-263)
 * needs; statistics of the values as stored.  Cout == 128 (32 groups of 4).  workspace: melgpt_conv_in_c1_stats_workspace
 * floats. */
int melgpt_conv_in_c1_stats_workspace(int B, int H, int W);
int melgpt_conv_in_c1_stats(const void* x, int x_dtype, const float* w, const float* bias, void* y, int dtype, int B,
                            int H, int W, int Cout, float eps, float* mean, float* rstd, float* workspace, void* stream);
/* Decoder.conv_out (:355-359): 3x3, pad 1, ONE output channel; w tap-major (9, C) f32; y (B,H,W). */
int melgpt_conv_out_c1(const void* x, int dtype, const float* w_tap_major, const float* bias, void* y,
                       int y_dtype, int B, int H, int W, int C, void* stream);
/* AttnBlock softmax (:438-440): probs[r,c] = softmax_c(scale*scores[r,c]), c < n; columns n..ld_probs-1 = 0 */
int melgpt_softmax_rows(const float* scores, long long ld_scores, int n, long long rows, float scale,
                        void* probs, long long ld_probs, int dtype, void* stream);
/* torch Conv2d weight (O,I,KH,KW) f32 -> (O,KH,KW,I) in out_dtype: the B operand of melgpt_conv2d_nhwc */
int melgpt_repack_conv_weight(const float* w_oihw, void* out_ohwi, int out_dtype, int O, int I, int KH, int KW,
                              void* stream);

/* ResnetBlock's  norm -> swish -> conv3x3  (big_model_attn_gan.py:117-127) as ONE kernel: halo-tiled 3x3 conv
 * (stride 1, pad 1) whose input patch is normalised (GroupNorm(32) statistics from melgpt_groupnorm_stats, affine
 * gamma/beta) and swish-ed while it is staged into LDS; mean == NULL -> plain convolution.  residual/bias as in
 * melgpt_conv2d_nhwc.  Supported when the patch fits LDS: Cin*elem_size in [256, ~680] bytes (Cin 128/256 bf16,
 * 64/128 f32); otherwise MELGPT_ERR_UNSUPPORTED (callers fall back to groupnorm_apply + conv2d_nhwc). */
int melgpt_conv3x3_gn_nhwc(const void* x, int B, int H, int W, int Cin, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int swish, const void* wpack, int Cout,
                           const float* bias, const void* residual, void* y, int dtype, void* stream);

/* The same launch that ALSO yields the GroupNorm(32) statistics (eps out_eps) of its own output y - what the block's next
 * Normalize (:124), or the next ResnetBlock's norm1 (:117) after conv2 + shortcut, would otherwise read y again for:
 * per-tile partial sums leave the conv's epilogue (values as stored: bias and residual added, rounded to bf16),
 * melgpt_groupnorm_finalize adds them per image in tile order.  bf16, Cout == 128, shapes that run on the persistent
 * kernel; anything else: MELGPT_ERR_UNSUPPORTED before any launch (callers then use melgpt_conv3x3_gn_nhwc +
 * melgpt_groupnorm_stats).  residual may be null.  workspace: melgpt_conv3x3_gn_stats_workspace(B,H,W) floats. */
int melgpt_conv3x3_gn_stats_workspace(int B, int H, int W);
int melgpt_conv3x3_gn_nhwc_stats(const void* x, int B, int H, int W, int Cin, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int swish, const void* wpack, int Cout,
                                 const float* bias, const void* residual, void* y, int dtype, float out_eps,
                                 float* out_mean, float* out_rstd, float* workspace, void* stream);
/* second stage of melgpt_groupnorm_stats on its own: partial (B, nchunks, 32, 2) sums -> mean / rstd (B*32);
 * count = elements per (image, group) */
int melgpt_groupnorm_finalize(const float* partial, int nchunks, int B, double count, float eps, float* mean, float* rstd,
                              void* stream);

/* (B,C,HW) <-> (B,HW,C) copy with dtype conversion, for callers that hand over contiguous NCHW tensors */
int melgpt_permute_nchw_nhwc(const void* x, int x_dtype, void* y, int y_dtype, int B, int C, int HW,
                             int to_nhwc, void* stream);

/* ===================================================================== MelGAN generator glue (vocoder/modules.py:24-79)
 * Conv1d / ConvTranspose1d layers run as implicit GEMMs (melgpt_conv1d_nlc below) on a channels-last (B, L, C)
 * activation; the padded copy + single-channel convolution pair serves the last layer (Cout = 1):
 * y (B, L + 2 pad, C) = act(pad(x (B, L, C))): reflect != 0 is nn.ReflectionPad1d (pad < L), else zero rows;
 * act = LeakyReLU(slope), slope = 1 for none.  C % (16 / sizeof(T)) == 0. */
int melgpt_pad1d_act(const void* x, void* y, int B, int L, int C, int pad, int reflect, float slope, int dtype,
                     void* stream);
/* One Conv1d (stride 1) or one output phase of a ConvTranspose1d of the generator as ONE implicit GEMM on a channels-last
 * activation x (B, L, Cin) - no padded copy, no per-tap launches (vocoder/modules.py:23-79: WNConv1d / WNConvTranspose1d
 * with their ReflectionPad1d / LeakyReLU neighbours):
 *   y[b, l, :] = bias + sum_{t < KW} W_t f(x[b, l - pad_l + t * dilation, :])  (+ residual[b, l, :]) (+ y when accumulate)
 * f = LeakyReLU(in_slope) applied to the input operand (in_slope = 0: none); out_slope != 0 applies LeakyReLU(out_slope) to
 * bias + sum before the residual; positions outside [0, L) are reflected
 * (reflect != 0: nn.ReflectionPad1d, at most one reflection) or read as zeros.  wpack (Cout, KW * Cin): the taps' (Cout, Cin)
 * slices side by side along K.  y / residual rows of Cout elements with row strides ldy / ldr (a transposed
 * convolution's phase s writes rows s, s + r, ... of its output: ldy = r * Cout).  Cin, Cout multiples of 16 bytes. */
int melgpt_conv1d_nlc(const void* x, int B, int L, int Cin, const void* wpack, int Cout, int KW, int dilation,
                      int pad_l, int reflect, float in_slope, const float* bias, const void* residual, long long ldr,
                      int accumulate, float out_slope, void* y, long long ldy, int dtype, void* stream);
/* y (B, L) f32 = [tanh]( bias + sum_{t<K, c<C} xp[b, l + t, c] * w[t*C + c] ): the generator's last Conv1d(ngf, 1, 7)
 * + nn.Tanh on an already padded xp (B, L + K - 1, C); w (K*C) f32 = weight[0].T flattened tap-major. */
int melgpt_conv1d_out1(const void* xp, const float* w, const float* bias, float* y, int B, int L, int C, int K,
                       int do_tanh, int dtype, void* stream);
/* The same output layer WITHOUT the padded copy, 16-bit lane: y (B, L) f32 = [tanh]( bias + conv_K( reflect_pad(
 * LeakyReLU_slope( h (B, L, C) ) ) ) ), C in {32, 64}, odd K; otherwise MELGPT_ERR_UNSUPPORTED (use the pair above). */
int melgpt_conv1d_out1_fused(const void* h, const float* w, const float* bias, float* y, int B, int L, int C, int K,
                             float slope, int do_tanh, int dtype, void* stream);
/* One ResnetBlock of a generator stage of dim 32, 64 or 128 (vocoder/modules.py:48-64) in one pass over the
 * activation: y (B, L, C) = shortcut(x) + conv1( LeakyReLU( conv3_dilated( ReflectionPad1d( LeakyReLU(x) ) ) ) ).
 * wfrag: the three weight matrices as MFMA operand fragments in lane order - 5 (C/32) (C/16) fragments of 64 lanes x 16
 * bytes: [3 (C/32)][C/16] conv3 (tap-major), [C/32][C/16] shortcut, [C/32][C/16] conv1 with the accumulator's channel
 * order as its k-slots (csrc/vocoder.hip; vocoder/modules.py packs them).  b3 = conv3's bias, b1s = conv1's + the
 * shortcut's.  16-bit lane, L % 16 == 0.  C = 128 (L % 64 == 0) runs the same block with the 160 fragments filling the
 * LDS instead of registers.  Otherwise MELGPT_ERR_UNSUPPORTED (run the three convolutions separately). */
int melgpt_resblock_narrow(const void* x, void* y, const void* wfrag, const float* b3, const float* b1s, int B, int L,
                           int C, int dilation, float slope, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MELGPT_H */
