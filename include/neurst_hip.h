/*
 * libneurst_hip.so -- C ABI of the MI355X-native SpeechTransformer training hot path.
 *
 * The reference (bytedance/neurst) is pure Python on TensorFlow: it has no FFI
 * boundary of its own.  Every entry point below REPLACES the third-party
 * TensorFlow kernel (or Horovod call) that the cited reference line dispatches
 * to; INTEGRATION.md shows the ctypes stub a NeurST maintainer binds them with.
 * Citations are file:line relative to the reference repository root.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / STL types.
 *  - every pointer is a DEVICE pointer owned by the caller (the PyTorch caching
 *    allocator in our host); the library allocates nothing persistent.
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it.
 *  - return 0 on success, a negative NST_ERR_* otherwise; the message is
 *    available from nst_last_error_string() (thread local).  Never aborts,
 *    never throws.
 *  - row-major contiguous tensors unless a leading dimension is given.
 *  - weights are in the reference's TensorFlow layout ([in,out] dense kernels,
 *    q|k|v packed projection columns, [H*dh,out] output projection, HWIO conv
 *    kernels, [V,d] shared embedding) so TF checkpoints map 1:1.
 *  - dtype: NST_F32 or NST_BF16 activations; LayerNorm statistics, softmax,
 *    loss, gradients of parameters and optimizer state are always fp32.
 */
#ifndef NEURST_HIP_H_
#define NEURST_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NST_ABI_VERSION 10

enum { NST_F32 = 0, NST_BF16 = 1 };

enum {
  NST_OK = 0,
  NST_ERR_INVALID_ARG = -1,
  NST_ERR_LAUNCH = -2,
  NST_ERR_UNSUPPORTED = -3,
  NST_ERR_WORKSPACE = -4
};

int nst_abi_version(void);
const char* nst_last_error_string(void);

/* Dropout seed offset.  Every entry point that draws a dropout mask (nst_gemm, nst_attention_fwd, nst_ffn_fwd,
 * nst_layernorm_bwd_dropout, nst_embedding_*, nst_scale_*_dropout_*) keys its Philox generator with
 * (seed + OFFSET, stream_id, element index), where OFFSET is one device-resident 64-bit scalar owned by the library
 * (0 until set) that the kernels read WHEN THEY RUN.  A host that replays a captured HIP graph of a training step keeps
 * its seed arguments constant and enqueues nst_dropout_seed_offset_add(1) once per step (inside the graph): every replay
 * then draws fresh masks, and forward / backward kernels of one step still see the same value.  A host that passes a
 * new seed per step (the reference passes none: tf.nn.dropout draws from TF's global generator,
 * neurst/layers/common_layers.py:82,157) leaves the offset at 0.  Both calls are asynchronous on `stream`. */
/* nst_dropout_seed_offset_bind(scalar_dev): from now on (this host thread) launches use the caller's 8-byte device scalar as
 * OFFSET instead of the library's own (NULL: back to the library's).  The pointer is passed to each kernel at launch, so
 * several model instances in one process keep separate counters, and a captured graph keeps the scalar it was captured
 * with.  _set / _add act on the scalar currently bound. */
int nst_dropout_seed_offset_bind(uint64_t* scalar_dev);
int nst_dropout_seed_offset_set(uint64_t value, void* stream);
int nst_dropout_seed_offset_add(uint64_t delta, void* stream);

/* ------------------------------------------------------------------ LayerNorm
 * tf.keras.layers.LayerNormalization(epsilon, dtype=float32) over the last axis:
 *   neurst/layers/common_layers.py:64-65,77 (pre-norm), transformer_encoder.py:98-100,135,
 *   transformer_decoder.py:99-101,225 (output_ln).
 * x,y: [rows,d] dtype; gamma,beta: [d] f32; mean,rstd: [rows] f32 (saved for bwd). */
int nst_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int64_t rows, int d, float eps, int dtype, void* stream);
/* dgamma/dbeta [d] f32 are ACCUMULATED into (+=) when accumulate!=0, else overwritten.
 * dres (nullable, [rows,d] dtype) is added to dx: the gradient arriving through the residual branch of
 * PrePostProcessingWrapper (inputs + y, common_layers.py:85), so no separate add pass is needed.
 * workspace (nullable): with at least 512*2*d*4 bytes the per-workgroup partial sums of dgamma/dbeta are written to it
 * and reduced by a second tiny kernel; without it they are accumulated with float atomics, whose fan-in on 2*d
 * addresses costs more than the whole data pass. */
int nst_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                      int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ GEMM with fused epilogue
 * Replaces tf.einsum/tf.matmul/Dense on the path:
 *   MultiHeadDenseLayer.call  neurst/layers/common_layers.py:262-288 (qkv / q / kv / output projections)
 *   TransformerFFN.call       neurst/layers/common_layers.py:156-159 (dense1+relu+dropout, dense2)
 *   AudioConv2dSubsamplingLayer output_dense  neurst/layers/modalities/audio_modalities.py:107-108
 *   WordEmbeddingSharedWeights._top  neurst/layers/modalities/text_modalities.py:103-108 (tied logits)
 * and their gradients (TF autodiff of the same ops).
 *
 *   C[M,N] = epilogue( alpha * op(A)[M,K] @ op(B)[K,N] )
 * trans_a==0: A stored [M,K] (lda); trans_a==1: A stored [K,M] (lda).
 * trans_b==0: B stored [K,N] (ldb); trans_b==1: B stored [N,K] (ldb).
 * Epilogue, applied in this order to v = alpha*acc:
 *   +bias[col] (f32) -> relu -> dropout(p; Philox(seed,stream_id, row*N+col)) -> +residual[row,col]
 *   -> *gate (gate_src[row,col] > 0 ? gate_scale : 0)   [backward of relu+dropout from the saved activation]
 *   -> (v*emb_scale + posenc[row % posenc_period, col])   [PositionEmbeddingWrapper, common_layers.py:427-434]
 *   -> C = v  or  C += v (accumulate)
 * split_k > 1 partitions K over grid.z (weight gradients: small output, very long reduction).  Requires
 * out_dtype==NST_F32 and the plain alpha*A*B (+accumulate) epilogue.  With a caller-provided workspace of at least
 * split_k*M*N*4 bytes every split writes its partial tile with plain coalesced stores and a second kernel sums the
 * slabs into C; without one the partials are added atomically (much slower: ~36 G atomic elements/s). */
typedef struct {
  int M, N, K;
  int trans_a, trans_b;
  int64_t lda, ldb, ldc;
  int in_dtype;   /* dtype of A and B */
  int out_dtype;  /* dtype of C, residual and gate_src */
  float alpha;
  const float* bias;      /* [N] or NULL */
  int relu;
  float dropout_p;        /* 0 = off */
  uint64_t seed, stream_id;
  const void* residual;   /* [M,N] ld = ldr, or NULL */
  int64_t ldr;
  const void* gate_src;   /* [M,N] ld = ldg, or NULL */
  int64_t ldg;
  float gate_scale;
  const float* posenc;    /* [posenc_period, N] f32 or NULL */
  int posenc_period;
  float emb_scale;
  int accumulate;
  int split_k;            /* <=1: off */
  void* workspace;        /* split-K slabs (device), or NULL */
  int64_t workspace_bytes;
  /* colsum != NULL (requires trans_b == 0): colsum[N] f32 (+)= sum_k B[k, :] -- the bias gradient that goes with a
   * weight gradient dW = X^T.dZ (A = X, trans_a = 1, B = dZ).  Computed inside the MFMA loop (no extra pass over dZ)
   * when the operands are 16-byte aligned and the output is f32, by a separate column-sum pass otherwise.  With
   * split_k > 1 the workspace must hold split_k*(M+1)*N floats. */
  float* colsum;
  int colsum_accumulate;
  /* split_k > 1 with a workspace: reduce_job_out != NULL defers the second stage -- nst_gemm only writes the partial
   * slabs (which must then stay untouched in `workspace`) and fills *reduce_job_out (HOST memory); the caller later hands up
   * to 8 such jobs to ONE nst_splitk_reduce_multi launch.  The 92 weight gradients of a step otherwise pay 92 separate
   * reduce launches of 7-20 us each on the weight-gradient stream. */
  struct NstSplitkJob* reduce_job_out;
  /* Row dots per 64-column head, fused into the epilogue (bf16 output, N % 64 == 0, split_k == 1, 16-byte aligned rows):
   *   rowdot_dst[(b*rowdot_heads + h)*rowdot_rows + t] = sum_{c < 64} C[b*rowdot_rows + t][h*64 + c] * rowdot_src[same]
   * with C as stored (rounded to bf16).  The input gradient of the attention output projection (dO = dZ . Wo^T,
   * multi_head_attention.py:213-215 in reverse) passes the attention output O as rowdot_src and receives
   * delta = rowsum(dO o O) [B, H, Tq] -- the term the softmax backward subtracts -- without another pass over dO and O;
   * nst_attention_bwd then takes out == NULL ("delta is already there").  NULL rowdot_dst: off. */
  const void* rowdot_src;
  int64_t ldrs;            /* elements */
  float* rowdot_dst;
  int rowdot_rows, rowdot_heads;
} NstGemmDesc;

typedef struct NstSplitkJob {
  const float* slabs;      /* [split][M][N] f32 */
  float* C;                /* [M][N] f32, leading dimension ldc */
  int64_t ldc;
  int M, N, split, accumulate;
  const float* cs_parts;   /* [split][N] partial column sums or NULL */
  float* cs_out;           /* [N] */
  int cs_accumulate, reserved;
} NstSplitkJob;

int nst_gemm(const NstGemmDesc* desc, const void* A, const void* B, void* C, void* stream);
/* n weight gradients dW_i[M_i, N_i] (+)= X_i^T . dZ_i in ONE launch (ABI 7): descs[i] describes product i exactly as for
 * nst_gemm (trans_a = 1, trans_b = 0, bf16 in, f32 out, plain epilogue; accumulate, colsum / colsum_accumulate honoured;
 * split_k must be <= 1), A[i] / B[i] / C[i] are its operands.  Every 256 x 256 output tile of every product is one workgroup of
 * the same grid, so a whole layer stack's gradients fill the chip without split-K slabs (the reference leaves these products
 * to TF's gradient tape: tf.GradientTape.gradient in neurst/training/gradaccum_keras_model.py:162-197).  Operands must be
 * 16-byte aligned with 8-element granular extents; returns NST_ERR_UNSUPPORTED (nothing launched) when a product does not
 * qualify -- the caller then issues nst_gemm per product.  Up to 56 products travel in the kernel arguments; more (n <= 1024)
 * need `workspace` (device memory, 16-byte aligned, >= 72 * n bytes, untouched until the launch has run): the product table is
 * written there by tiny kernels on the same stream, so the call stays capturable into a HIP graph. */
int nst_gemm_wgrad_group(const NstGemmDesc* descs, const void* const* A, const void* const* B, void* const* C, int n,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* C (+)= sum of the slabs (and the column sums) of up to 8 deferred split-K products, one launch. */
int nst_splitk_reduce_multi(const NstSplitkJob* jobs_host, int njobs, void* stream);

/* Sequence masks from lengths (ABI 7): out[b][t] = t < len'(b) ? on_token : on_padding with len' = `halvings` times
 * ceil(len / stride) -- model_utils.py:44-75 (sequence_mask and 1 - sequence_mask), layer_utils.py:19-32 (padding * FLOAT_MIN:
 * on_padding = -1e9) and the conv-subsampled lengths of speech_transformer.py:179-189 (halvings = 2, stride = 2) in one launch.
 * lengths [B] int64 (device), out [B, T] f32. */
int nst_seq_mask(const int64_t* lengths, float* out, int B, int T, int halvings, int stride, float on_token, float on_padding,
                 void* stream);
/* Reductions of LabelSmoothedCrossEntropy (label_smoothed_cross_entropy.py:46-53, 141-157) in one launch (ABI 7):
 * nll_sum[b] = sum_t xent[b][t], n_tokens[b] = sum_t weights[b][t], loss = sum(nll_sum) / sum(n_tokens),
 * inv_tokens = 1 / sum(n_tokens) (the device scalar nst_ls_xent_bwd reads).  xent, weights [B, L] f32 dense. */
int nst_xent_reduce(const float* xent, const float* weights, int B, int L, float* nll_sum, float* n_tokens, float* loss,
                    float* inv_tokens, void* stream);

/* Column sums: out[N] (f32) (+)= sum_rows x[rows,N] -- bias gradients.  workspace (nullable, >= 256*N*4 bytes):
 * two-stage reduction without atomics, as for nst_layernorm_bwd. */
int nst_colsum(const void* x, float* out, int64_t rows, int n, int64_t ldx, int dtype, int accumulate, void* workspace,
               int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ fused scaled-dot-product attention
 * MultiHeadAttention.call / att_fn  neurst/layers/attentions/multi_head_attention.py:124-164, 203-215:
 *   q *= dh^-0.5; logits = q.k^T + bias; softmax; dropout(p) on the probabilities; out = P.v
 * q [B,Tq,H,dh] with row stride ldq (elements) between consecutive (b,t) rows (so the packed
 * q|k|v projection output can be passed without a split copy); same for k,v (ldk, ldv) and out (ldo).
 * key_bias: [B,Tk] f32 additive bias (padding * FLOAT_MIN, neurst/layers/layer_utils.py:19-32) or NULL.
 * causal!=0 adds FLOAT_MIN where key > query (lower_triangle_attention_bias, layer_utils.py:35-53).
 * lse [B,H,Tq] f32 = log-sum-exp of the biased logits, saved for backward. */
typedef struct {
  int B, H, Tq, Tk, dh;
  int64_t ldq, ldk, ldv, ldo;
  int dtype;
  float scale;       /* dh^-0.5 */
  int causal;
  float float_min;   /* -1e9 (compat.FLOAT_MIN) */
  float dropout_p;
  uint64_t seed, stream_id;
  /* dropout_p > 0: device buffer of nst_attention_dropout_mask_bytes(desc) bytes, 8-byte aligned.  The forward
   * call writes one keep bit per probability into it, the backward call of the same step reads it (no RNG in bwd). */
  void* dropout_mask;
  int64_t dropout_mask_bytes;
  /* batch strides (elements) of k and v (and of dk, dv); 0 = Tk * ldk / Tk * ldv (dense [B,Tk,*]).  A larger stride
   * lets incremental decoding attend over the filled prefix of a preallocated [B, Tmax, *] key / value cache
   * (multi_head_attention.py:254-290 concatenates instead). */
  int64_t bsk, bsv;
  /* with causal != 0: key j is masked for query i when j > i + causal_offset (0 = the lower-triangle bias; k - 1 = the
   * wait-k bias of layer_utils.py:56-78 for cross attention, band_part(ones, -1, k - 1)).  Must be >= 0. */
  int causal_offset;
  int reserved0;
  /* nst_attention_bwd, bf16: optional scratch of at least B*H*ceil(Tk/128)*128*ceil(Tq/64)*64*2 bytes (16-byte aligned).
   * With it the dK/dV kernel leaves the scaled dS^T there and dQ is one small product over it instead of a second pass that
   * recomputes both score products and all the element-wise work (the backward is VALU-issue bound).  NULL: two-pass form. */
  void* ds_workspace;
  int64_t ds_workspace_bytes;
} NstAttnDesc;

/* B*H*ceil(Tq/16)*ceil(Tk/64)*128 bytes */
int64_t nst_attention_dropout_mask_bytes(const NstAttnDesc* d);

int nst_attention_fwd(const NstAttnDesc* d, const void* q, const void* k, const void* v, const float* key_bias,
                      void* out, float* lse, void* stream);
/* dq,dk,dv share the layouts/strides of q,k,v (ldq,ldk,ldv); dout that of out. delta [B,H,Tq] f32 workspace.
 * out == NULL: delta already holds rowsum(dout o out) per head (NstGemmDesc.rowdot_dst of the GEMM that produced dout). */
int nst_attention_bwd(const NstAttnDesc* d, const void* q, const void* k, const void* v, const float* key_bias,
                      const void* out, const void* dout, const float* lse, float* delta, void* dq, void* dk,
                      void* dv, void* stream);

/* ------------------------------------------------------------------ conv2d-subsampling front end
 * AudioConv2dSubsamplingLayer.call  neurst/layers/modalities/audio_modalities.py:84-109.
 * Layer 1 (C_in = 1): pad 1 -> Conv2D 3x3 s2 VALID + bias -> LayerNorm(C, eps) -> ReLU, fused in one pass.
 *   src [B,T,F] dtype f32 (the feature tensor is always f32), w1 [3,3,1,C] f32 (HWIO), out [B,T1,F1,C] dtype.
 *   mean/rstd [B*T1*F1] f32 saved for backward (ignored / may be NULL when layer_norm==0).
 *   bf16 output with C == 256: the taps run on the matrix cores as hi/lo bf16 splits of the f32 operands (x_hi*w_hi + x_hi*w_lo +
 *   x_lo*w_hi: ~2^-16 relative to the f32 product, the same arithmetic nst_conv1_ln_relu_bwd recomputes); every other case f32 FMAs. */
int nst_conv1_ln_relu_fwd(const float* src, const float* w1, const float* b1, const float* gamma,
                          const float* beta, void* out, float* mean, float* rstd, int B, int T, int F, int C,
                          int layer_norm, float eps, int out_dtype, void* stream);
/* Recomputes the conv from src; dout is the gradient w.r.t. the ReLU output.  All parameter gradients f32,
 * accumulated (+=) when accumulate!=0. */
int nst_conv1_ln_relu_bwd(const float* src, const float* w1, const float* b1, const float* gamma,
                          const float* beta, const float* mean, const float* rstd, const void* dout, float* dw1,
                          float* db1, float* dgamma, float* dbeta, int B, int T, int F, int C, int layer_norm,
                          float eps, int dtype, int accumulate, void* stream);
/* Layer 2 as an implicit GEMM on MFMA (M = B*T2*F2, N = C, K = 9*C), x [B,T1,F1,C] dtype, w2 [3,3,C,C] dtype (HWIO).
 *   fwd:   y[B,T2,F2,C] = conv(x) + b2, ReLU fused when relu!=0 (layer_norm=False recipe); with LayerNorm the
 *          LN+ReLU follow as nst_layernorm_relu_fwd
 *   dgrad: dx[B,T1,F1,C] = conv_transpose(dy, w2)
 *   wgrad: dw2[3,3,C,C] f32 (+)= x (*) dy   */
int nst_conv2_fwd(const void* x, const void* w2, const float* b2, void* y, int B, int T1, int F1, int C, int relu,
                  int dtype, void* stream);
int nst_conv2_dgrad(const void* dy, const void* w2, void* dx, int B, int T1, int F1, int C, int dtype, void* stream);
/* workspace (nullable): split-K slabs, see nst_gemm; needs splits*(9*C+1)*C*4 bytes (query with workspace==NULL is
 * not needed: 64 MiB covers every supported shape; smaller workspaces fall back to atomic accumulation). */
/* db2 (nullable): [C] f32 (+)= sum over pixels of dy -- the conv bias gradient, produced by the same pass over dy. */
int nst_conv2_wgrad(const void* x, const void* dy, float* dw2, float* db2, int B, int T1, int F1, int C, int dtype,
                    int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* LayerNorm + ReLU (second conv layer; audio_modalities.py:102-104) -- same contract as nst_layernorm_*,
 * y = relu(LN(x)); backward takes dy w.r.t. the ReLU output and the saved y (gate y>0). */
int nst_layernorm_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                           int64_t rows, int d, float eps, int dtype, void* stream);
/* nst_layernorm_bwd that also emits dz = dropout_backward(dx) with the mask Philox(seed, stream_id, element index) of
 * the sublayer that consumes dx next (PrePostProcessingWrapper, common_layers.py:80-84): the pre-norm residual chain
 * feeds every dx through exactly that dropout mask, so the extra pass over dx is folded into this kernel. */
int nst_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                              const void* dres, void* dx, void* dz, float dropout_p, uint64_t seed, uint64_t stream_id,
                              float* dgamma, float* dbeta, int64_t rows, int d, int dtype, int accumulate,
                              void* workspace, int64_t workspace_bytes, void* stream);
int nst_layernorm_relu_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                           const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                           int accumulate, void* workspace, int64_t workspace_bytes, void* stream);

/* Deferred parameter-gradient stage of the LayerNorm backward.  dgamma / dbeta feed nothing in the backward chain, so
 * their cross-workgroup reduction need not sit between dx and the next GEMM: nst_layernorm_bwd_deferred runs the dx
 * kernel (all three variants: y != NULL -> ReLU gate, dz != NULL -> also emit the dropped gradient), leaves the partial
 * sums in `workspace` (which must then stay untouched) and fills *job_out (HOST memory); up to 16 such jobs go to ONE
 * nst_ln_finalize_multi launch, on any stream ordered after the dx kernels (the trainer uses its weight-gradient stream).
 * job_out->nblocks == 0 means nothing is pending (rows == 0, or the workspace was too small and the gradients were
 * accumulated directly). */
typedef struct NstLnFinalizeJob {
  const float* partial;    /* [nblocks][2][d] f32 */
  float* dgamma;           /* [d] */
  float* dbeta;            /* [d] */
  int nblocks, d, accumulate, reserved;
} NstLnFinalizeJob;
int nst_layernorm_bwd_deferred(const void* dy, const void* x, const void* y, const float* gamma, const float* mean,
                               const float* rstd, const void* dres, void* dx, void* dz, float dropout_p, uint64_t seed,
                               uint64_t stream_id, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                               int accumulate, void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out,
                               void* stream);
int nst_ln_finalize_multi(const NstLnFinalizeJob* jobs_host, int njobs, void* stream);
/* nst_layernorm_relu_bwd without the saved activation y (ABI 9): the ReLU gate is recomputed as LN(x) > 0 from x, the saved
 * statistics, gamma and beta -- the expression the forward evaluated -- which takes a quarter off the traffic of the
 * front end's 576 000-row backward (audio_modalities.py:102-104).  job_out == NULL: parameter gradients finished by this call.
 * Needs d % 8 == 0, d <= 1024, 16-byte aligned rows and a workspace (NST_ERR_UNSUPPORTED otherwise). */
int nst_layernorm_relu_bwd_regate(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, int dtype,
                                  int accumulate, void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out,
                                  void* stream);

/* fp32 residual stream of the pre-norm path (ABI 9).  PrePostProcessingWrapper (neurst/layers/common_layers.py:73-85) computes
 * inputs + dropout(layer(LN(inputs))) in float32; a bf16 path that rounds that sum to bf16 after every sub-layer carries 24
 * roundings through a 12-layer encoder -- measured as the one rounding class that puts the bf16 gradients outside 1e-2 of
 * the reference (profiles/r05_rounding_point_study_b32.json).  Here the sum stays f32: the sub-layer's last kernel writes
 * its contribution `delta` = dropout(layer(..)) as bf16 WITHOUT the residual, and the NEXT LayerNorm adds it:
 *     x_new = x + delta   (f32; written to x_out unless x_out == NULL)      y = LayerNorm(x_new)   (dtype, bf16)
 * x is f32 (x_dtype = NST_F32) or, for the first sub-layer of a stack, the bf16 embedding output (x_dtype = NST_BF16).
 * mean / rstd of x_new as in nst_layernorm_fwd.  Needs d % 8 == 0, d <= 1024, 16-byte aligned rows (NST_ERR_UNSUPPORTED
 * otherwise: the host then keeps the bf16 stream). */
int nst_add_layernorm_fwd(const void* x, int x_dtype, const void* delta, void* x_out, const float* gamma, const float* beta,
                          void* y, float* mean, float* rstd, int64_t rows, int d, float eps, int dtype, void* stream);
/* nst_layernorm_bwd / _bwd_dropout / _bwd_deferred in one entry for a saved input x of its own dtype (x_dtype = NST_F32 with
 * dtype = NST_BF16: the x_new of nst_add_layernorm_fwd; x_dtype == dtype: same as the entries above).  dz == NULL: no dropped
 * copy; job_out == NULL: the parameter gradients are finished by this call. */
int nst_layernorm_bwd_mixed(const void* dy, const void* x, int x_dtype, const float* gamma, const float* mean,
                            const float* rstd, const void* dres, void* dx, void* dz, float dropout_p, uint64_t seed,
                            uint64_t stream_id, float* dgamma, float* dbeta, int64_t rows, int d, int dtype, int accumulate,
                            void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out, void* stream);

/* ------------------------------------------------------------------ whole-row products of the pre-norm wrapper (ABI 10)
 * PrePostProcessingWrapper.call, neurst/layers/common_layers.py:73-85: inputs + dropout(layer(LayerNorm(inputs))).  With
 * d_model = 256 a workgroup owns complete output rows, so the wrapper's row-wise stages run in the epilogue of the product
 * that makes the row (bf16 operands, f32 accumulation; nst_rowgemm_supported says whether a shape qualifies: n = 256,
 * k a multiple of 64).
 *   A [rows, k] (lda);  W: trans_b == 0: [k, 256] (ldb) -- a dense kernel as stored, forward;  trans_b == 1: [256, k] (ldb) --
 *   the same kernel read as the input-gradient operand (dX = dZ . W^T).  Built orientations (NST_ERR_UNSUPPORTED otherwise):
 *   nst_gemm_add_layernorm_fwd trans_b == 0; nst_gemm_layernorm_bwd and nst_gemm_rowdot256 trans_b == 1. */
typedef struct NstRowGemmDesc {
  int64_t rows;
  int n, k;                 /* n must be 256 */
  int trans_b;
  int dtype;                /* NST_BF16 */
  int64_t lda, ldb;
  float dropout_p;          /* fwd: the wrapper's dropout on the product; bwd: the mask of the dz copy */
  float eps;                /* fwd: LayerNorm epsilon */
  uint64_t seed, stream_id; /* Philox key of that mask (element index row * 256 + col, as nst_gemm) */
} NstRowGemmDesc;
int nst_rowgemm_supported(int n, int k, int dtype);
/* The LAST product of a sub-layer (attention output_transform, multi_head_attention.py:219; feed-forward dense2,
 * common_layers.py:159) with the wrapper's dropout, the residual add and the NEXT wrapper's LayerNorm (common_layers.py:77) in
 * its epilogue:  delta = bf16(dropout(A . W + bias));  x_out = x + delta (f32, nullable);  y = LayerNorm(x + delta; gamma,
 * beta, eps) (bf16);  mean / rstd [rows] of the sum.  Equals nst_gemm (bias, dropout) followed by nst_add_layernorm_fwd. */
int nst_gemm_add_layernorm_fwd(const NstRowGemmDesc* desc, const void* A, const void* W, const float* bias, const float* x,
                               float* x_out, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                               void* stream);
/* The FIRST product of a sub-layer's backward chain (input gradient of qkv_transform / q_transform / dense1: g = A . W is the
 * gradient w.r.t. the LayerNorm output) with the LayerNorm backward in its epilogue:  dx = LayerNorm'(bf16(g); x, mean, rstd,
 * gamma) + dres (bf16; dres nullable);  dz (nullable) = dx under the dropout mask of the desc;  dgamma / dbeta as in
 * nst_layernorm_bwd_deferred (workspace >= ceil(rows / 32) * 2 * 256 * 4 bytes holds the per-workgroup partial sums; job_out
 * NULL: finished by this call).  Equals nst_gemm followed by nst_layernorm_bwd_mixed. */
int nst_gemm_layernorm_bwd(const NstRowGemmDesc* desc, const void* A, const void* W, const float* x, const float* gamma,
                           const float* mean, const float* rstd, const void* dres, void* dx, void* dz, float* dgamma,
                           float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out,
                           void* stream);
/* C [rows, 256] = bf16(A . W^T) and dst[(b * 4 + h) * rows_per_batch + t] = sum over head h's 64 columns of
 * C[b * rows_per_batch + t, .] o src[same] (f32): the attention output projection's input gradient together with the
 * delta = rowsum(dO o O) its attention backward needs (nst_gemm's rowdot epilogue on whole rows). */
int nst_gemm_rowdot256(const NstRowGemmDesc* desc, const void* A, const void* W, void* C, const void* src, float* dst,
                       int rows_per_batch, void* stream);
/* ------------------------------------------------------------------ target embedding
 * WordEmbeddingSharedWeights._bottom + PositionEmbeddingWrapper.call
 *   neurst/layers/modalities/text_modalities.py:84-93, neurst/layers/common_layers.py:415-434:
 *   out[b,t,:] = table[ids[b,t],:] * emb_scale + posenc[t,:]   (then dropout p)
 * table [V,d] dtype, ids int64 [rows], posenc [L,d] f32 or NULL, out [rows,d] dtype; rows = B*L. */
int nst_embedding_fwd(const void* table, const int64_t* ids, const float* posenc, void* out, int64_t rows, int L,
                      int d, int V, float emb_scale, float dropout_p, uint64_t seed, uint64_t stream_id, int dtype,
                      void* stream);
/* dtable [V,d] f32 += scatter-add of dout*emb_scale*(dropout mask) (sparse IndexedSlices made dense,
 * neurst/training/hvd_utils.py:72-73).  Always accumulates (the tied logits GEMM writes the dense part first). */
int nst_embedding_bwd(const void* dout, const int64_t* ids, float* dtable, int64_t rows, int d, int V,
                      float emb_scale, float dropout_p, uint64_t seed, uint64_t stream_id, int dtype, void* stream);

/* ------------------------------------------------------------------ elementwise dropout (+ scale, + posenc)
 * tf.nn.dropout at transformer_encoder.py:125-127 / transformer_decoder.py:212-214 and the
 * PositionEmbeddingWrapper scale+signal on the audio front-end output (common_layers.py:427-434):
 *   y = dropout_p( x*scale + posenc[row % period, :] )          (posenc may be NULL)
 * backward: dx = dy * mask * scale  (same Philox triple). */
int nst_scale_posenc_dropout_fwd(const void* x, const float* posenc, void* y, int64_t rows, int d, int period,
                                 float scale, float dropout_p, uint64_t seed, uint64_t stream_id, int dtype,
                                 void* stream);
int nst_scale_dropout_bwd(const void* dy, void* dx, int64_t n, float scale, float dropout_p, uint64_t seed,
                          uint64_t stream_id, int dtype, void* stream);

/* ------------------------------------------------------------------ label-smoothed cross entropy
 * LabelSmoothedCrossEntropy.__call__  neurst/criterions/label_smoothed_cross_entropy.py:94-157.
 * logits [rows,V] dtype (ldl), labels int64 [rows], weights f32 [rows] (sequence_mask(trg_length)).
 * fwd: xent[rows] f32 = (-(sum_v soft_v*log_softmax_v) - normalizing_constant) * weight ; lse[rows] f32 saved.
 * bwd: dlogits[rows,V] dtype = (softmax - soft_target) * weight[row] * gscale * (gscale_dev ? *gscale_dev : 1)
 *      (reduce_loss = sum(nll)/sum(n_tokens), label_smoothed_cross_entropy.py:46-53: pass 1/sum(n_tokens) either as
 *      the host value gscale or, to avoid a device->host sync, as the device scalar gscale_dev) */
int nst_ls_xent_fwd(const void* logits, const int64_t* labels, const float* weights, float* xent, float* lse,
                    int64_t rows, int V, int64_t ldl, float label_smoothing, int dtype, void* stream);
int nst_ls_xent_bwd(const void* logits, const int64_t* labels, const float* weights, const float* lse,
                    void* dlogits, int64_t rows, int V, int64_t ldl, float label_smoothing, float gscale,
                    const float* gscale_dev, int dtype, void* stream);

/* ------------------------------------------------------------------ data-parallel exchange: what the compute entry points guarantee
 * The exchange itself is at the end of this header (nst_comm_*, ABI v8: the Horovod calls of the reference --
 * hvd.DistributedOptimizer / hvd.allreduce, neurst/training/hvd_utils.py:46-98; rank-0 broadcast neurst/exps/trainer.py:285);
 * a host may just as well keep it in its own framework (ours does by default: RCCL through torch.distributed; the metric
 * reduce of neurst/training/callbacks.py:149-207 always stays there).  There is no nst_workspace_query_*: workspaces are
 * sized by the caller from the formulas given at each entry point.  What the compute entry points guarantee to either kind
 * of host:
 *   - all parameter gradients of a model are written into ONE flat fp32 buffer in forward registration order, so a
 *     bucket is a contiguous [start, end) slice the host can hand to RCCL (ncclAllReduce / torch.distributed) as is;
 *   - every entry point is asynchronous on the stream it is given and touches nothing outside its arguments: the host
 *     orders its communication stream after the compute and weight-gradient streams with plain HIP events;
 *   - the 1/world_size of hvd.Average is an ARGUMENT of the entry points that follow the exchange (grad_scale of
 *     nst_adam_update*, pre_scale of nst_grad_clip), nst_loss_scale_update runs on the already exchanged buffer.
 * Our host does this in neurst_amd/training/distributed.py (side stream, buckets = the slices the backward pass reports,
 * optional 16-bit wire).
 *
 * ------------------------------------------------------------------ optimizer
 * Keras Adam (neurst/models/speech_transformer.py:265-279, neurst/optimizers/__init__.py) over ONE flat buffer:
 *   m = b1*m+(1-b1)*g ; v = b2*v+(1-b2)*g^2 ; p -= lr_t * m/(sqrt(v)+eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
 * g is multiplied by grad_scale first (1/world_size for hvd.Average, neurst/training/hvd_utils.py:46-50).
 * p,m,v,g f32 [n]; shadow (bf16 copy of p for the compute path) may be NULL. */
int nst_adam_update(float* p, float* m, float* v, const float* g, uint16_t* shadow_bf16, int64_t n, float lr_t,
                    float beta1, float beta2, float eps, float grad_scale, void* stream);
/* Same update with the bias-corrected step size lr_t read from DEVICE memory when the kernel runs: a captured HIP graph of
 * the training step is replayed with a new lr_t per step (the host writes the scalar before each replay). */
int nst_adam_update_dev(float* p, float* m, float* v, const float* g, uint16_t* shadow_bf16, int64_t n, float lr_t,
                        const float* lr_t_dev, float beta1, float beta2, float eps, float grad_scale,
                        const float* loss_scale_state, void* stream);
/* lr_t_dev (nullable): overrides lr_t.  loss_scale_state (nullable): the 4-float state of nst_loss_scale_update -- the update is
 * skipped when its finite flag is 0 and the gradients are divided by the scale they carry otherwise
 * (tf.keras LossScaleOptimizer.apply_gradients under the reference's RevisedDynamicLossScale).
 *
 * Dynamic loss scale, neurst/training/revised_dynamic_loss_scale.py:48-107 (wrapped around the optimizer by
 * training_utils.handle_fp16_and_distributed_optimizer :373-419 when the compute dtype is float16; initial scale 2^15,
 * growth_steps 2000, multiplier 2).  state[4] f32 = {current scale, good-step counter, finite flag, scale of the checked
 * gradients}.  One call per optimizer step, AFTER the gradient exchange: checks grad[0..n) for inf / nan, records the flag and
 * the scale the gradients carry, then halves the scale (floor 1) and clears the counter on overflow, or counts a good step and
 * doubles the scale every growth_steps good steps.  workspace: >= 4 bytes, zero on first use (left zero). */
int nst_loss_scale_update(const float* grad, int64_t n, float* state, float growth_steps, float multiplier, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Gradient clipping of the flat gradient buffer after the data-parallel average (GradAccumKerasModel.train_step,
 * neurst/training/gradaccum_keras_model.py:228-233):  g *= pre_scale (the 1/world_size average), then
 *   clip_value > 0: g = clamp(g, -clip_value, +clip_value)                     (tf.clip_by_value)
 *   clip_norm  > 0: g = g * clip_norm / max(||g||_2, clip_norm)  PER TENSOR   (tf.clip_by_norm on each gradient)
 * exactly one of the two must be positive.  table (device, nentries x 16 bytes): {int64 off; int32 n (<= 4096); int32 seg}
 * -- entries of one tensor (segment) are consecutive; seg_first (device, int32[nseg + 1]): first entry of every
 * segment; workspace: nentries + nseg floats (clip_norm only).  Deterministic (no atomics). */
int nst_grad_clip(float* grad, const void* table, int nentries, const int32_t* seg_first, int nseg, float* workspace,
                  int64_t workspace_floats, float pre_scale, float clip_value, float clip_norm, void* stream);

/* f32 -> bf16 cast (weight shadow refresh), bf16 -> f32, and fill. */
int nst_cast_f32_to_bf16(const float* in, uint16_t* out, int64_t n, void* stream);
int nst_cast_bf16_to_f32(const uint16_t* in, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ host helper of the data feed
 * CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of a byte range, continuing from `crc` (0 to start): the
 * checksum of TFRecord frames (neurst/data/dataset_utils.py:256-326 reads them through TensorFlow).  Pure host code. */
uint32_t nst_crc32c(const void* data, int64_t n, uint32_t crc);

/* ------------------------------------------------------------------ probes (used by tests only)
 * nst_probe_mfma: writes the raw lane->value maps of the MFMA / LDS-transpose-read instructions the
 * kernels rely on, so the layout assumptions are verified on real hardware. */
int nst_probe_mfma(float* out_c16, float* out_c16_f32, uint16_t* out_tr, void* stream);
/* nst_probe_fetch (measurement only): `workgroups` workgroups of 4 waves each stream steps x 32 KB from
 * src + (workgroup % group_mod) * wg_stride (cyclically over `span` bytes, a multiple of 32 KB) into a ring of `ring` (2..4) LDS
 * slots with the LDS-DMA pattern of the stream kernels (mode 0) or with plain 16-byte register loads (mode 1) and do nothing
 * else: the time of the launch gives the fetch rate a kernel of that tile shape cannot exceed.  group_mod = workgroups: private
 * regions; 8: the workgroups of an XCD (ids go round-robin over the 8 XCDs) share a region, as the tiles of a split-K slice
 * do; wg_stride = 0: one region for all.  pattern (mode 0): how the 64 lanes of a 1 KB piece address it -- 0 contiguous, 1 / 2 the
 * swizzled reduction-major / row-major tile images of the stream GEMM, 3 / 4 the same with rows 4 KB apart.  sink: one float,
 * never written. */
int nst_probe_fetch(const void* src, int64_t wg_stride, int64_t span, int steps, int ring, int mode, int workgroups, int group_mod,
                    int pattern, float* sink, void* stream);

/* ------------------------------------------------------------------ fused position-wise feed-forward (d_model = 256, bf16)
 * TransformerFFN.call inside PrePostProcessingWrapper.call (pre-norm), neurst/layers/common_layers.py:145-160, 73-85:
 *     hidden = dropout(relu(x @ dense1/kernel + dense1/bias), ffn_dropout_rate)          [rows, filter_size]
 *     y      = residual + dropout(hidden @ dense2/kernel + dense2/bias, layer_postprocess_dropout_rate)
 * in ONE launch: the hidden tile never goes back through HBM between the two products (it is WRITTEN once, as the
 * activation the backward pass needs).  Replaces two nst_gemm calls (tf.keras Dense x2 + tf.nn.relu + tf.nn.dropout x2
 * + the residual add in the reference).  nst_ffn_bwd is the input-gradient half of TF autodiff over the same ops:
 *     dhidden = (dy @ dense2/kernel^T) * (hidden > 0 ? 1/(1-p_hidden) : 0)                [rows, filter_size]
 *     dx      = dhidden @ dense1/kernel^T (+ residual)
 * (the two weight gradients x^T.dhidden and hidden^T.dy remain nst_gemm reductions over the rows, fed by dhidden / hidden).
 *
 * Weight operands:
 *   nst_ffn_fwd reads TRANSPOSED bf16 copies  w1t [filter_size, 256] = dense1/kernel^T,  w2t [256, filter_size] =
 *   dense2/kernel^T (made by nst_transpose_bf16 whenever the weights change: once per optimizer step);
 *   nst_ffn_bwd reads the kernels in the reference's layout: w2 = dense2/kernel [filter_size, 256], w1 = dense1/kernel
 *   [256, filter_size].
 * Dropout masks are the library's Philox masks of nst_gemm: element (row, col) of the [rows, N] tensor, keyed by
 * (seed (+ *seed_offset), stream_id) -- the hidden mask over N = filter_size, the output mask over N = 256.
 * seed_offset (nullable): device scalar added to both seeds when the kernel runs, so a captured HIP graph draws new
 * masks at every replay (the host bumps the scalar between replays).
 * nst_ffn_supported(d_model, filter_size, dtype) != 0 tells whether this path exists for a shape (d_model 256,
 * filter_size a multiple of 128, bf16); otherwise the caller composes the same math from nst_gemm. */
typedef struct {
  int64_t rows;
  int d_model, filter_size, dtype;
  float hidden_dropout_p;          /* ffn_dropout_rate */
  uint64_t hidden_seed, hidden_stream_id;
  float output_dropout_p;          /* layer_postprocess_dropout_rate (forward only) */
  uint64_t output_seed, output_stream_id;
  const uint64_t* seed_offset;     /* device scalar or NULL */
  /* Optional gate bits (ABI 6).  The backward needs of `hidden` only the test hidden > 0; when nst_ffn_gate_bits_bytes(desc)
   * is > 0 the caller may hand nst_ffn_fwd a buffer of that many bytes (4-byte aligned, device memory): the forward fills it
   * with one bit per hidden element (layout private to the library), and nst_ffn_bwd given the SAME buffer reads it instead
   * of the [rows, filter_size] activation (1/16 of the traffic; `hidden` must still be passed and is what every other
   * path uses).  NULL: the activation is the gate.  0 from the query: this shape runs a kernel that has no bit path --
   * passing a buffer to nst_ffn_fwd is then an argument error. */
  void* gate_bits;
  int64_t gate_bits_bytes;
} NstFfnDesc;

int nst_ffn_supported(int d_model, int filter_size, int dtype);
int64_t nst_ffn_gate_bits_bytes(const NstFfnDesc* desc);   /* rows, d_model, filter_size, dtype are read */
/* b1 [filter_size] f32 (nullable = 0), b2 [256] f32 (nullable), residual [rows,256] (nullable); hidden [rows, filter_size]
 * and y [rows,256] are written. */
int nst_ffn_fwd(const NstFfnDesc* desc, const void* x, const void* w1t, const float* b1, const void* w2t, const float* b2,
                const void* residual, void* hidden, void* y, void* stream);
/* hidden: the activation nst_ffn_fwd saved; residual (nullable, [rows,256]) is added to dx (post-norm wrapper). */
int nst_ffn_bwd(const NstFfnDesc* desc, const void* dy, const void* hidden, const void* w2, const void* w1,
                const void* residual, void* dhidden, void* dx, void* stream);
/* The feed-forward pair (nst_ffn_fwd / nst_ffn_bwd above) with the same row stages behind its second product (ABI 10; the
 * eight-wave kernel's shapes with gate bits):
 *   nst_ffn_add_layernorm_fwd = nst_ffn_fwd (residual NULL, output dropout of the desc) followed by nst_add_layernorm_fwd on the
 *     float32 stream x_res: x_out = x_res + delta, y = LayerNorm(x_out; gamma, beta, eps), mean / rstd;
 *   nst_ffn_layernorm_bwd = nst_ffn_bwd (residual NULL) followed by nst_layernorm_bwd_mixed of the wrapper's LayerNorm (saved
 *     input x_ln f32, mean, rstd, gamma; dres added; dz = dropped copy of dx under (dz_p, dz_seed, dz_stream_id), nullable;
 *     dgamma / dbeta through workspace >= ceil(rows / 32) * 2 * 256 * 4 bytes, job_out as in nst_layernorm_bwd_deferred).
 * nst_ffn_ln_supported: 0 = no; 1 = one launch (>= 20 480 rows: every CU gets a 128-row tile); 2 = two launches (1 024 ..
 * 20 479 rows, e.g. the decoder's 9 600: too few row tiles for the chip, so the hidden dimension is split over S workgroups per
 * tile that leave f32 partial sums in `slabs` (nst_ffn_ln_slab_bytes(desc) bytes, 16-byte aligned, caller-owned scratch) and a
 * second launch adds them and runs the row stages).  Gate bits are required in both modes: rows * (filter_size / 32) * 4 bytes
 * (what nst_ffn_gate_bits_bytes returns where nst_ffn_fwd supports them); bits written by mode 2 are read by
 * nst_ffn_layernorm_bwd only. */
int nst_ffn_ln_supported(const NstFfnDesc* desc);
int64_t nst_ffn_ln_slab_bytes(const NstFfnDesc* desc);
int nst_ffn_add_layernorm_fwd(const NstFfnDesc* desc, const void* x, const void* w1t, const float* b1, const void* w2t,
                              const float* b2, const float* x_res, float* x_out, const float* gamma, const float* beta, float eps,
                              void* hidden, void* y, float* mean, float* rstd, void* slabs, int64_t slabs_bytes, void* stream);
int nst_ffn_layernorm_bwd(const NstFfnDesc* desc, const void* dy, const void* hidden, const void* w2, const void* w1,
                          const float* x_ln, const float* gamma, const float* mean, const float* rstd, const void* dres,
                          void* dhidden, void* dx, void* dz, float dz_p, uint64_t dz_seed, uint64_t dz_stream_id, float* dgamma,
                          float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes, NstLnFinalizeJob* job_out,
                          void* slabs, int64_t slabs_bytes, void* stream);

/* Batched bf16 transposes dst[cols, rows] = src[rows, cols]^T, one launch for a table of matrices (device memory):
 * tile0 = number of 64x64 tiles of all earlier jobs, tiles_c = ceil(cols / 64); total_tiles = sum over jobs. */
typedef struct {
  const void* src;
  void* dst;
  int rows, cols, tiles_c, tile0;
} NstTransposeJob;
int nst_transpose_bf16(const NstTransposeJob* jobs_dev, int njobs, int total_tiles, void* stream);

/* Batched strided block copies, one launch for a table of jobs (device memory): rows x row_bytes from src (pitch src_pitch
 * bytes) to dst (pitch dst_pitch).  Used for PACKED copies of weights that several layers apply to the same input: the six
 * cross-attention kv_transform kernels [d, 2d] of the decoder layers (neurst/layers/transformer_layers.py:213-234,
 * multi_head_attention.py:166-223) side by side as one [d, 6*2d] operand, so the projection of the encoder output is ONE GEMM
 * (and its input gradient one GEMM with K = 6*2d); refreshed once per optimizer step like the transposed copies.
 * block0 = number of 256-thread blocks of all earlier jobs, nblocks = blocks of this job; total_blocks = their sum. */
typedef struct {
  const void* src;
  void* dst;
  int rows, row_bytes;
  int64_t src_pitch, dst_pitch;
  int block0, nblocks;
} NstPack2dJob;
int nst_pack2d(const NstPack2dJob* jobs_dev, int njobs, int total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL (one process per GPU; ABI v8).  Replaces the Horovod calls of the reference's data-parallel
 * step: the averaged all-reduce of every gradient (neurst/training/hvd_utils.py:46-62, hvd.DistributedOptimizer /
 * hvd.Average; optional fp16 wire training_utils.py:381-384) and the broadcast of rank 0's variables
 * (neurst/exps/trainer.py:285).  The host moves the 128-byte unique id from rank 0 to the other ranks by whatever channel it
 * has (the reference: MPI under Horovod; here: the launcher's TCP store).  RCCL is bound at run time (librccl.so.1 of the
 * process, else of the loader path; NST_RCCL_PATH overrides): NST_ERR_UNSUPPORTED when there is none.
 *
 * A communicator owns a communication stream and its event fences; every call is asynchronous for the host:
 *   nst_comm_allreduce_bucket  the communication stream waits for everything queued so far on the `producers` streams (the
 *                              compute stream, the weight-gradient stream), then SUMS buf[0:count) over the ranks in place.
 *                              Buckets issued in the same order on every rank.  No 1/N: nst_adam_update's grad_scale does it.
 *   nst_comm_fence             `consumer` (the stream of the optimizer step) waits for every bucket issued so far.
 *   nst_comm_broadcast         buf[0:count) of `root` to all ranks, ordered with `stream` on both sides.
 * dtype: NST_F32, NST_BF16, NST_COMM_F16, NST_COMM_U8.  All ranks of a communicator call init / destroy collectively. */
/* Streams by priority class (-1 high, 0 default, 1 low; NST_ERR_UNSUPPORTED when the device lacks the class).  The HIP runtime
 * multiplexes the streams of one class onto a few hardware queues, and two BUSY streams that share a queue run in turns
 * (measured: 13.0 -> 22 ms per step when the step's stream and its weight-gradient stream met in one queue).
 * A host keeps the three concurrent activities of a step in three classes: the step on a high-priority stream, the
 * weight-gradient stream on a low-priority one, the exchange (the communicator's own stream) in the default class. */
int nst_stream_create(int priority_class, void** stream_out);
int nst_stream_destroy(void* stream);

#define NST_COMM_UNIQUE_ID_BYTES 128
enum { NST_COMM_F16 = 2, NST_COMM_U8 = 3 };
int nst_comm_unique_id(void* id, size_t id_bytes);
int nst_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out);
int nst_comm_info(void* comm, int* rank, int* world, int64_t* buckets_since_fence, int64_t* bytes_since_fence);
int nst_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, void* const* producers, int nproducers);
int nst_comm_fence(void* comm, void* consumer);
int nst_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream);
int nst_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* NEURST_HIP_H_ */
