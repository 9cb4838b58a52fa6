/*
 * tvl_hip.h -- C ABI of libtvl_hip.so: the MI355X (gfx950) kernels behind the
 * prompt-tuning hot path of naamiinepal/tunevlseg.
 *
 * The reference has no native code and no FFI: its hot path is
 * the PyTorch op sequence inside ``net(text_input, image_input)``
 * (reference src/models/image_text_mask_module.py:68-70,257-265).  Each entry
 * point below replaces the op sequence cited next to it.  Conventions:
 *
 *   - plain C symbols, device pointers + sizes + a hipStream_t (as void*);
 *   - return 0 on success, non-zero on bad arguments / launch failure
 *     (tvl_last_error() gives the text; the Python host raises RuntimeError);
 *   - no allocation, no synchronisation, no global state: re-entrant per stream
 *     and capturable into a hipGraph;
 *   - all tensors are fp32, row-major, unless stated; "ld*" are leading
 *     dimensions in elements.
 */
#ifndef TVL_HIP_H
#define TVL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tvlStream_t; /* hipStream_t */

/* Bumped on every incompatible change of a signature or struct below.  3: tvl_dicece_stats gained `work`, tvl_split_planes /
 * tvl_gemm_planes removed (round 2).  4: tvl_text_assemble gained `vocab` (rows of the embedding table: ids outside it give NaN rows
 * instead of a wild read); tvlGemmTp3Args gained workspace / workspace_bytes; tvl_resize_u8 / tvl_augment_u8 added.  5: tvl_dicece_loss, tvl_mlp64_*, tvl_h2_zero_rows added; tvl_upconv_taps_fwd gained `work`; tvl_attn_h2_bwd gained `only_block`, tvl_h2k_gather_rows added;
 * tvl_layernorm_fwd_h2 / _bwd_h2 gained max_slot / tag (largest row norm by tagged atomicMax).  6 (round 4): tvl_dicece_loss / tvl_adamw gained `nonfinite` (sticky device-side NaN / Inf flags), TVL_ACT_GELU, the DenseCLIP entries
 * (tvl_groupnorm_*, tvl_tconv2x2_unshuffle, tvl_colscale_*).  The Python binding refuses a library whose tvl_abi_version() differs. */
#define TVL_ABI_VERSION 6

const char* tvl_last_error(void);
int tvl_abi_version(void);
/* how the library was built: bit 0 = make EXPERIMENTS=1 (retired GEMM variants present), bit 1 = DIAG=1, bit 2 = POISON=1 (LDS poison debug build) */
int tvl_build_flags(void);

/* row' = (r / div) * mul + (r % div) + off   (div <= 0: identity) */
typedef struct { int32_t div, mul, off; } tvlRowMap;

enum { TVL_NT = 0, TVL_NN = 1, TVL_TN = 2 };
enum { TVL_ACT_NONE = 0, TVL_ACT_QUICK_GELU = 1, TVL_ACT_RELU = 2, TVL_ACT_SIGMOID = 3, TVL_ACT_GELU = 4 /* erf form, nn.GELU() */ };
/* OR into tvlGemmArgs.act: apply the activation AFTER the residual add (ResNet bottleneck tail relu(conv + identity),
 * reference cris_model/clip.py:75); default order is act first, then residual (transformer blocks) */
enum { TVL_ACT_POST_RESIDUAL = 0x100 };

/*
 * C = epilogue(alpha * op(A) . op(B)), exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   layout NT: A[M,K] (lda), B[N,K] (ldb)   -> nn.Linear forward  (HF modeling_clipseg.py:290-292,333,352-354)
 *   layout NN: A[M,K] (lda), B[K,N] (ldb)   -> data gradient  dX = dY . W
 *   layout TN: A[K,M] (lda), B[K,N] (ldb)   -> weight gradient dW = dY^T . X
 * epilogue, in this order, per element (m, n):
 *   v = alpha*acc; v += bias[n]; v *= act'(dact_aux[m,n]) (dact); pre_out[m,n] = v;
 *   v = act(v); v += residual[m,n]; C[map(m), n] = v      (with TVL_ACT_POST_RESIDUAL: v += residual; v = act(v))
 * a_map remaps the rows of A as stored (the M rows for NT/NN, the K rows for TN);
 * c_map remaps rows of C / pre_out / residual / dact_aux.
 */
typedef struct {
    int32_t layout, M, N, K;
    const float* A; int32_t lda;
    const float* B; int32_t ldb;
    float* C; int32_t ldc;
    const float* bias;
    const float* residual; int32_t ldr;
    int32_t act;
    float* pre_out;
    const float* dact_aux; int32_t ld_aux; int32_t dact;
    float alpha;
    tvlRowMap a_map, c_map;
} tvlGemmArgs;
int tvl_gemm_f32(const tvlGemmArgs* args, tvlStream_t stream);
/*
 * Same contract (fp32 in, fp32 out, same epilogue), NT layout only, computed on the bf16 matrix cores
 * (v_mfma_f32_32x32x16_bf16) with each fp32 operand split into `nsplit` bf16 pieces while it is staged:
 *   nsplit 3: 6 MFMAs per k-step, fp32-equivalent accuracy (three bf16 pieces hold all 24 significand bits)
 *   nsplit 2: 3 MFMAs, ~2^-16 relative per product;   nsplit 1: plain bf16 operands.
 */
int tvl_gemm_bf16s(const tvlGemmArgs* args, int32_t nsplit, tvlStream_t stream);
/* Skinny problems (few output tiles, deep K: the text towers' M = B*L-row GEMMs): the k-range is split over `splits`
 * workgroups per output tile, partials go to `workspace` (>= splits*M*N floats) and are summed in a fixed order by a second
 * kernel that applies the epilogue -- deterministic, 3-piece split only. */
int tvl_gemm_bf16s_splitk(const tvlGemmArgs* args, int32_t splits, float* workspace, int64_t workspace_floats, tvlStream_t stream);

/* LayerNorm over the last dim (nn.LayerNorm, eps 1e-5; HF:347,355,392,396).  mean/rstd may be NULL. */
int tvl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int64_t rows, int32_t cols, float eps, tvlStream_t stream);
/* dx = [dres +] LN'(dy); dgamma/dbeta (may be NULL) are ACCUMULATED with atomics (tiny trainable norms only). */
int tvl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      const float* dres, float* dx, float* dgamma, float* dbeta,
                      int64_t rows, int32_t cols, tvlStream_t stream);
/* LayerNorm between tp3 GEMMs: the forward writes ONLY the tp3 image of y [rows, cols] (cols % 16 == 0, <= 2048); the backward
 * writes dx twice, fp32 (residual-stream gradient) and tp3 (A operand of the next data-gradient GEMM); frozen gamma / beta. */
int tvl_layernorm_fwd_tp3(const float* x, const float* gamma, const float* beta, void* y_tp3, float* mean, float* rstd,
                          int64_t rows, int32_t cols, float eps, tvlStream_t stream);
int tvl_layernorm_bwd_tp3(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, void* dx_tp3, int64_t rows, int32_t cols, tvlStream_t stream);

/*
 * Multi-head softmax attention, flash style (HF eager_attention_forward, modeling_clipseg.py:232-252).
 * q/k/v: element (b, t, h, d) at ptr[b*bs + t*ts + h*dh + d] (packed QKV GEMM output or separate).
 * o: [B, T, H*dh] (ldo = row stride).  lse: [B, H, T] natural-log-sum-exp of the scaled scores.
 * causal != 0: key j visible to query i iff j <= i (reference coop_clipseg.py:229-233);
 * key_mask (may be NULL): int32 [B, Tk], 0 = padded key (coop_clipseg.py:236-246).
 * dh in {8,16,32,64}.  Tk = number of keys/values (0: same as T); Tk != T is cross-attention (CRIS decoder,
 * reference cris_model/layers.py:341-349: 676 visual queries x <=77 word keys with a key-padding mask), no causal.
 */
typedef struct {
    const float *q, *k, *v; int64_t q_bs, k_bs, v_bs; int32_t q_ts, k_ts, v_ts;
    float* o; int32_t ldo;
    float* lse;
    const int32_t* key_mask;
    int32_t B, H, T, dh, causal;
    float scale;
    int32_t Tk;
} tvlAttnFwdArgs;
int tvl_attn_fwd(const tvlAttnFwdArgs* a, tvlStream_t stream);

typedef struct {
    const float *q, *k, *v; int64_t q_bs, k_bs, v_bs; int32_t q_ts, k_ts, v_ts;
    const float* o; const float* d_o; int32_t ldo;
    const float* lse;
    float* delta;               /* workspace [B,H,T] */
    float *dq, *dk, *dv; int64_t dq_bs, dk_bs, dv_bs; int32_t dq_ts, dk_ts, dv_ts;
    const int32_t* key_mask;
    int32_t B, H, T, dh, causal;
    float scale;
    int32_t Tk;
} tvlAttnBwdArgs;
int tvl_attn_bwd(const tvlAttnBwdArgs* a, tvlStream_t stream);
/* The vision tower's attention (d_h = 64, no masks) between tp3 GEMMs (see "tp3" below): the forward writes O as the tp3 image of
 * [B*T, H*64] (a->o may be NULL), the backward takes that image for delta = rowsum(dO * O) and writes dQ | dK | dV as the tp3
 * image of the packed gradient [B*T, 3*H*64] -- the A operand of the QKV data-gradient GEMM (a->o, a->dq/dk/dv must be NULL). */
int tvl_attn_fwd_tp3(const tvlAttnFwdArgs* a, void* o_tp3, tvlStream_t stream);
/* The same attention with Q, K, V read from the tp3 image of the packed QKV matrix [B*T, 3*H*64] (written by the QKV GEMM's
 * epilogue): key tiles are filled by LDS-DMA, nothing is split in the kernel.  lse [B,H,T] may be NULL. */
int tvl_attn_tp3_fwd(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream);
/* Backward of tvl_attn_tp3_fwd on tp3 operands: packed QKV, O (the forward's output) and dO (tp3 image of [B*T, H*64], e.g. the
 * out-projection data gradient written by tvl_gemm_tp3 with C_tp3); lse from the forward; delta: [B, H, T] fp32 workspace.
 * Writes dQ | dK | dV as the tp3 image of the packed gradient [B*T, 3*H*64] (rows beyond B*T of the last block are not touched).
 * Replaces autograd through HF CLIPSegAttention (modeling_clipseg.py:232-252) for the vision tower, as tvl_attn_bwd_tp3 does. */
int tvl_attn_tp3_bwd(const void* qkv_tp3, const void* o_tp3, const void* do_tp3, const float* lse, float* delta, void* dqkv_tp3,
                     int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream);
/* diagnostics (tools/bench_attn.py): ablation variants / per-workgroup clock stamps; outputs of variants != 0, 16 are wrong by construction */
int tvl_attn_tp3_fwd_diag(const void* qkv_tp3, void* o_tp3, float* lse, int32_t B, int32_t H, int32_t T, float scale, int32_t variant,
                          int64_t* stamps, tvlStream_t stream);
int tvl_attn_bwd_tp3(const tvlAttnBwdArgs* a, const void* o_tp3, void* dqkv_tp3, tvlStream_t stream);

/* ---- token plumbing (reference vpt_context_learner.py:46-64, base_visual_learner.py:18-23,
 *      coop_context_learner.py:124-181, HF:190-206) ---- */
/* im2col for the 16x16/s16 patch conv: img [B,C,H,W] -> cols [B*gh*gw, C*ps*ps] */
int tvl_im2col_patch(const float* img, float* cols, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ps, tvlStream_t stream);
/* x0[b,t,:] = t==0 ? cls+pos[0] : t<=P ? patch[b*P+t-1]+pos[t] : ctx[(b*ctx_bs) + t-1-P]  ; T = 1+P+n */
int tvl_vision_assemble(const float* patch, const float* cls, const float* pos, const float* ctx, int64_t ctx_bs,
                        float* x0, int32_t B, int32_t P, int32_t n, int32_t D, tvlStream_t stream);
/* out[b,t,:] = (map[t] >= 0 ? table[ids[b*L + map[t]]] : ctx[b*ctx_bs + (-map[t]-1)*D]) + pos[t]  (ids int64) */
int tvl_text_assemble(const int64_t* ids, int32_t L, const int32_t* map, const float* table, int64_t vocab, const float* ctx, int64_t ctx_bs,
                      const float* pos, float* out, int32_t B, int32_t T, int32_t D, tvlStream_t stream);
/* out[b,t,:] = map[t] >= 0 ? x[b, map[t], :] : ctx[b*ctx_bs + (-map[t]-1)*D ...]   (x: [B, L, D]) */
int tvl_splice_rows(const float* x, int32_t L, const int32_t* map, const float* ctx, int64_t ctx_bs, float* out,
                    int32_t B, int32_t T, int32_t D, tvlStream_t stream);
/* x[b, row0+j, :] = src[b*src_bs + j*D ...]  for j < n  (src_bs = 0: broadcast one [n,D] block over the batch) */
int tvl_rows_overwrite(float* x, const float* src, int64_t src_bs, int32_t B, int32_t T, int32_t D, int32_t row0, int32_t n, tvlStream_t stream);
/* dst[(b*dst_bs) + j*D + c] (+)= sum_b? g[b, row0+j, c]; reduce_batch != 0 sums over b into one [n,D] block;
 * zero_src != 0 clears g[b,row0+j,:] afterwards (gradient cut of an in-place overwrite). accumulate != 0: += */
int tvl_rows_grad(float* g, float* dst, int32_t B, int32_t T, int32_t D, int32_t row0, int32_t n,
                  int32_t reduce_batch, int32_t zero_src, int32_t accumulate, tvlStream_t stream);
/* out[b,:] = x[b, idx[b], :]  /  dx[b, idx[b], :] += dout[b,:]  (dx pre-zeroed by caller) ; idx int32 */
int tvl_gather_rows(const float* x, const int32_t* idx, float* out, int32_t B, int32_t T, int32_t D, tvlStream_t stream);
int tvl_scatter_rows_add(const float* dout, const int32_t* idx, float* dx, int32_t B, int32_t T, int32_t D, tvlStream_t stream);

/* ---- decoder pieces (reference base_clipseg.py:82-172, vpt_clipseg.py:237-319, HF:549-586) ---- */
/* FiLM: y[b,t,c] = mul[b,c]*x[b,t,c] + add[b,c] */
int tvl_film_fwd(const float* x, const float* mul, const float* add, float* y, int32_t B, int32_t T, int32_t C, tvlStream_t stream);
/* dx = mul*dy ; dmul[b,c] = sum_t dy*x ; dadd[b,c] = sum_t dy  (dmul/dadd may be NULL) */
int tvl_film_bwd(const float* dy, const float* x, const float* mul, float* dx, float* dmul, float* dadd,
                 int32_t B, int32_t T, int32_t C, tvlStream_t stream);
/* ConvTranspose2d(k=s=ps) tail: logits[b, gy*ps+py, gx*ps+px] = a*(cols[(b*G*G+gy*G+gx), py*ps+px] + bias) + r*extra[...]
 * (extra may be NULL; a, r implement `+=` (1,1) or `(1-r)*..+r*..` mixing of the new last layer) */
int tvl_pixel_shuffle_fwd(const float* cols, const float* bias, const float* extra, float a, float r,
                          float* logits, int32_t B, int32_t G, int32_t ps, tvlStream_t stream);
/* dcols[(b,gy,gx),(py,px)] = a * dlogits[b, gy*ps+py, gx*ps+px] */
int tvl_pixel_unshuffle_bwd(const float* dlogits, float a, float* dcols, int32_t B, int32_t G, int32_t ps, tvlStream_t stream);
/* new last layer (reference base_clipseg.py:58-71): Upsample(x ps, bilinear, align_corners=False) ->
 * Conv2d(C->1, k, same, replicate).  The channel contraction is a GEMM done by the caller:
 * taps[(b,i,j), ky*k+kx] = sum_c feat[b,i,j,c] * w[c,ky,kx]   ([B*G*G, ldg]); then
 *   out[b,y,x] = bias + sum_{ky,kx} bilinear(taps[b,:,:,ky,kx])(clamp(y+ky-pl), clamp(x+kx-pl))
 * so the C x (G*ps)^2 upsampled map is never materialised. k <= 7. */
int tvl_upconv_taps_fwd(const float* taps, int32_t ldg, const float* bias, float* out, float* work /* B*G*k*G*ps floats: two separable passes; or null: one gathering pass */,
                        int32_t B, int32_t G, int32_t ps, int32_t k, tvlStream_t stream);
/* dtaps [B*G*G, ldg] (first k*k columns written); work: [B*k*G*ps*G] floats */
int tvl_upconv_taps_bwd(const float* dout, float* dtaps, int32_t ldg, float* work,
                        int32_t B, int32_t G, int32_t ps, int32_t k, tvlStream_t stream);

/* ---- loss + metrics (monai DiceCELoss(sigmoid) + torchmetrics Dice/Jaccard; reference
 *      image_text_mask_module.py:87-107,272-302, configs/model/vpt_clipseg.yaml:21-25) ---- */
/* per-sample sums over N = H*W pixels, one pass:
 *   fsum[b] = {sum p*t, sum p, sum t, sum bce}  (float64 x4)   p = sigmoid(logit)
 *   isum[b] = {TP, FP, FN, TN} of (p > thr) vs (int64)t  (int64 x4, bit-exact)
 * label (may be NULL): uint8 thresholded map.  work: tvl_dicece_work_doubles(B, N) doubles -- per-workgroup partial sums, added in
 * a fixed order by a second kernel, so fsum (hence the loss) is bitwise reproducible; the integer counts are exact anyway.
 * Definitions pinned by the hand-derived known-answer vectors tests/golden/loss_metric_kav.json. */
int64_t tvl_dicece_work_doubles(int32_t B, int64_t N);
int tvl_dicece_stats(const float* logits, const float* target, double* fsum, int64_t* isum, uint8_t* label, double* work,
                     int32_t B, int64_t N, float thr, tvlStream_t stream);
/* loss[0] = lambda_dice * mean_b[1 - (2 fsum[b][0] + smooth_nr) / (fsum[b][1] + fsum[b][2] + smooth_dr)] + lambda_ce * sum_b fsum[b][3] / (B N):
 * monai DiceCELoss(sigmoid=True) for one channel (mean over the batch of the per-sample Dice term; BCE-with-logits, mean over all
 * elements), float64 inside, fixed summation order, written as one fp32 scalar on the device (no host round trip, one launch instead of
 * the dozen tensor-library launches of the same arithmetic) */
int tvl_dicece_loss(const double* fsum, float* loss, int32_t B, int64_t N, float lambda_dice, float lambda_ce,
                    float smooth_nr, float smooth_dr, int32_t* nonfinite, tvlStream_t stream);   /* nonfinite (may be NULL): += 1 when the loss is NaN / Inf */
/* dlogits = gscale * ( lambda_dice * dDice/dlogit + lambda_ce * (p - t)/(B*N) ), using fsum from tvl_dicece_stats */
int tvl_dicece_bwd(const float* logits, const float* target, const double* fsum, float* dlogits,
                   int32_t B, int64_t N, float lambda_dice, float lambda_ce, float smooth_nr, float smooth_dr,
                   const float* gscale, tvlStream_t stream);

/* ---- feed-forward half of a CLIPSeg decoder layer in one kernel (reduce_dim = 64, hidden_act = relu; HF CLIPSegDecoderLayer,
 *      modeling_clipseg.py:393-410 as called from the reference's decoder loop, src/models/core_models/coop/base_clipseg.py via
 *      CLIPSegDecoder):   out = LayerNorm(x + W2 relu(W1 x + b1) + b2),  x [M, 64], W1 [F, 64], W2 [64, F], F % 128 == 0.
 * The F-wide intermediate never leaves the registers (csrc/mlp64.hip); two fp16 pieces per operand, three MFMAs per product.
 * tvl_mlp64_pack: the frozen weights once -> img (tvl_mlp64_image_bytes(F) bytes: four fragment-ordered images), scaled by
 * scale1 / scale2 = the powers of two that put max |W1| / max |W2| into [2^13, 2^14); the kernels get their inverses.
 * tvl_mlp64_fwd: also t2 = the LayerNorm's input and mean / rstd [M] (null when no backward follows).  w1_rownorm = max_f ||W1[f, :]||_2,
 * b1_max = max |b1| (bound of the hidden rows, which sets their fp16 scale).
 * tvl_mlp64_bwd: dx = d out / d x applied to dout (LayerNorm backward, both GEMM data gradients with the relu gate recomputed from
 * x, and the residual path); w2_colnorm = max_f ||W2[:, f]||_2. */
int64_t tvl_mlp64_image_bytes(int32_t F);
int tvl_mlp64_pack(const float* W1, const float* W2, int32_t F, float scale1, float scale2, void* img, tvlStream_t stream);
int tvl_mlp64_fwd(const float* x, const void* img, const float* b1, const float* b2, const float* gamma, const float* beta, float* out, float* t2,
                  float* mean, float* rstd, int64_t M, int32_t F, float inv_w1, float inv_w2, float w1_rownorm, float b1_max, float eps,
                  tvlStream_t stream);
int tvl_mlp64_bwd(const float* dout, const float* x, const float* t2, const float* mean, const float* rstd, const void* img, const float* b1,
                  const float* gamma, float* dx, int64_t M, int32_t F, float inv_w1, float inv_w2, float w2_colnorm, tvlStream_t stream);

/* ---- input side (row f2): decoded uint8 sample -> network input, on the device ----
 * tvl_normalize_u8: albumentations Normalize(mean, std, max_pixel_value=255) + ToTensorV2 of the reference's transforms
 *   (configs/experiment/coop/clipseg.yaml:78-127): img [B,H,W,3] uint8 (host-pointer mean3 / std3) -> out [B,3,H,W] float.
 * tvl_mask_u8: mask [n] uint8 -> float / 255 (reference image_text_mask_dataset.py:66-71). */
int tvl_normalize_u8(const uint8_t* img, float* out, int32_t B, int32_t H, int32_t W, const float* mean3, const float* std3, tvlStream_t stream);
int tvl_mask_u8(const uint8_t* mask, float* out, int64_t n, tvlStream_t stream);
/* tvl_resize_u8: albumentations.Resize of the WHOLE ragged batch (reference configs/experiment/coop/clipseg.yaml:80-84): image b =
 *   hw[2b] x hw[2b+1] x C uint8 at packed + offs[b] (device arrays) -> out [B,H,W,C] uint8.  mode 2 = cv2.INTER_CUBIC (OpenCV's 8-bit
 *   fixed-point definition), mode 0 = cv2.INTER_NEAREST (masks).
 * tvl_augment_u8: Affine (p-gated, cubic / nearest, BORDER_REPLICATE) + RandomBrightnessContrast + Normalize + ToTensorV2
 *   (clipseg.yaml:85-120) in one pass: img [B,H,W,3] uint8, mask [B,H,W] uint8 or null, params [B,8] = inverse affine 2x3 (dst -> src),
 *   alpha, beta; flags [B] bit 0 warp, bit 1 brightness/contrast; host-pointer mean3 / std3 -> out_img [B,3,H,W], out_mask [B,1,H,W] (/255). */
int tvl_resize_u8(const uint8_t* packed, const int64_t* offs, const int32_t* hw, int32_t B, int32_t C, int32_t H, int32_t W, int32_t mode,
                  uint8_t* out, tvlStream_t stream);
int tvl_augment_u8(const uint8_t* img, const uint8_t* mask, const float* params, const int32_t* flags, const float* mean3, const float* std3,
                   float* out_img, float* out_mask, int32_t B, int32_t H, int32_t W, tvlStream_t stream);

/* last-layer mix with the TRAINABLE residual_ratio read on the device (reference base_clipseg.py:150-155, coop_cris.py:240-242):
 * out = (1 - ratio[0]) * main + ratio[0] * extra;   y = (one_minus ? 1 - ratio[0] : ratio[0]) * x  (its gradient passes).
 * No host read of the scalar: the step stays free of device->host synchronisation and can be captured into a hipGraph. */
int tvl_mix(const float* main, const float* extra, const float* ratio, float* out, int64_t n, tvlStream_t stream);
int tvl_scale_dev(const float* x, const float* ratio, int32_t one_minus, float* y, int64_t n, tvlStream_t stream);

/* ---- optimiser + misc ---- */
/* torch.optim.AdamW step over a flat fp32 buffer (decoupled weight decay); step_t is 1-based */
int tvl_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int32_t step_t, float grad_scale, int32_t* nonfinite, tvlStream_t stream);   /* nonfinite (may be NULL): set to 1 when a gradient is NaN / Inf */
int tvl_fill(float* p, float val, int64_t n, tvlStream_t stream);
/* y = a*x + b*y */
int tvl_axpby(const float* x, float a, float* y, float b, int64_t n, tvlStream_t stream);
/* y[r,c] = act(x[r,c]) or dact: y = dy * act'(x) */
int tvl_bias_act(const float* x, const float* bias, float* y, int64_t rows, int32_t cols, int32_t act, tvlStream_t stream);
/* y[i] = keep(seed, i) ? x[i] / (1 - p) : 0 ; keep is a pure function of (seed, i): call again on dy for the backward
 * (nn.Dropout inside the SharedAttn learner's TransformerEncoderLayer, reference configs/model/shared_attn_clipseg.yaml:21) */
int tvl_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, tvlStream_t stream);
/* out = dy * act'(pre) */
int tvl_dact_mul(const float* dy, const float* pre, float* out, int64_t n, int32_t act, tvlStream_t stream);
/* CoCoOp shifted context (reference cocoop_context_learner.py:50-58): out[b,j,:] = bias[b,:] + cvec[j,:] and its gradients */
int tvl_outer_add(const float* bias, const float* cvec, float* out, int32_t B, int32_t n, int32_t D, tvlStream_t stream);
int tvl_outer_add_bwd(const float* dout, float* dbias, float* dcvec, int32_t B, int32_t n, int32_t D, tvlStream_t stream);
/* l2-normalise rows: y = x / ||x|| ; bwd */
int tvl_l2norm_fwd(const float* x, float* y, float* inv_norm, int32_t rows, int32_t cols, tvlStream_t stream);
int tvl_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int32_t rows, int32_t cols, tvlStream_t stream);
/* out[0] (+)= sum_i x[i]*y[i]  (y NULL: plain sum) */
int tvl_dot(const float* x, const float* y, float* out, int64_t n, int32_t accumulate, tvlStream_t stream);
/* colsum[c] (+)= sum_r x[r,c]  (bias gradients of small trainable Linears) */
int tvl_colsum(const float* x, float* out, int64_t rows, int32_t cols, int32_t accumulate, tvlStream_t stream);

/* ---- CRIS conv path (BASELINE configs[2]; reference src/models/components/cris_model/clip.py:18-274,
 *      layers.py:15-119,359-445, src/models/core_models/coop/coop_cris.py:203-242) ----
 * Feature maps are NHWC pixel matrices: element (b, y, x, c) at ptr[((b*H + y)*W + x)*ld + c] (ld >= C: a channel slice of a
 * wider concat buffer is addressed by pointer offset + ld).  A 1x1 conv (+ folded eval BatchNorm + ReLU) is tvl_gemm_* over
 * the map; a 3x3 conv is tvl_gemm_* over the im2col matrix; their data gradients are the same two calls with the
 * transposed / tap-flipped weight matrix. */
/* dst[r*ldd + c] = src[r*lds + c]: channel concat (torch.cat(dim=1) of NCHW maps, layers.py:424,429,440) and its split */
int tvl_copy2d(const float* src, int32_t lds, float* dst, int32_t ldd, int64_t rows, int32_t cols, tvlStream_t stream);
/* cols[(b,oy,ox), (ky*3+kx)*C + c] = x[b*sb + (oy*stride+ky-1)*sy + (ox*stride+kx-1)*sx + c*sc] (0 outside the map);
 * explicit element strides let the stem conv read the NCHW image; columns 9*C..ldc-1 are zero-filled. pad 1, stride 1|2. */
int tvl_im2col3x3(const float* x, int64_t sb, int64_t sy, int64_t sx, int64_t sc, float* cols, int32_t ldc,
                  int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, tvlStream_t stream);
/* The same conv WITHOUT the im2col matrix (implicit GEMM on the split-bf16 kernel): args->A = the NHWC map (lda = its row
 * stride), args->B = weights [N, ldb] in the column order above, args->M = B*Ho*Wo, args->K = 9*C; epilogue as tvl_gemm_*.
 * Needs C % 4 == 0; nsplit must be 3 (the fp32-equivalent split).  a_map is ignored. */
typedef struct { int32_t B, H, W, C, stride; } tvlConvGeom;
int tvl_conv3x3_bf16s(const tvlGemmArgs* args, const tvlConvGeom* geom, int32_t nsplit, tvlStream_t stream);
/*
 * "tp3": an fp32 matrix X[R, K] (K % 16 == 0) handed over as three bf16 pieces per element, x = p0 + p1 + p2 (p0, p1 by
 * truncating the running residual, p2 rounded: all 24 significand bits), stored in MFMA-fragment order:
 *   block (rb, kb) = rows 32*rb..+31, k = 16*kb..+15 at byte ((rb * K/16 + kb) * 3 + piece) * 1024,
 *   element (r, k) inside a piece at ((k % 16) / 8 * 32 + r % 32) * 16 + (k % 8) * 2;   rows are padded to a multiple of 32.
 * Frozen weights are packed once (tvl_tp3_pack); activations are written in this form by their producer (LayerNorm,
 * attention, the GEMM epilogue below), so the consumer GEMM fills LDS by DMA and spends nothing on splitting.
 * tvl_tp3_bytes: size of the image; tvl_tp3_unpack: back to fp32 (tests / debugging).
 */
int64_t tvl_tp3_bytes(int64_t rows, int32_t K);
int tvl_tp3_pack(const float* x, int64_t ldx, int64_t rows, int32_t K, void* out, tvlStream_t stream);
int tvl_tp3_unpack(const void* in, int64_t rows, int32_t K, float* y, int64_t ldy, tvlStream_t stream);
/*
 * C[M,N] = epilogue(alpha * A . B^T) with A = tp3 image of [a_rows >= M, K] and B = tp3 image of [b_rows >= N, K]
 * (nn.Linear forward with the weight as stored, HF modeling_clipseg.py:290-292,333,352-354; its data gradient with the
 * transposed weight packed once).  Same epilogue contract and order as tvl_gemm_f32 (no row maps; pre_out shares ldc).
 * Outputs: C (fp32, may be NULL) and / or C_tp3, the tp3 image of the final value as the next GEMM's A operand [M, N].
 * K % 16 == 0, N % 16 == 0.  tile_m: 0 = choose, else 128 | 192 | 256 rows per workgroup; variant: scheduling A/B switch.
 * workspace (optional, tvl_gemm_h2 / tvl_gemm_h2_out): >= 64 MiB of device memory owned by the calling stream; launches of more than
 * 512 tiles whose epilogue writes an image (QKV, fc1, its data gradient's dz) then walk their tiles with one persistent workgroup per
 * CU and park one partial tile per workgroup there (csrc/gemm_h2m_kernel.h).  NULL: one workgroup per tile.
 */
typedef struct {
    int32_t M, N, K;
    const void* A; int64_t a_rows;
    const void* B; int64_t b_rows;
    float* C; int32_t ldc;
    void* C_tp3;
    const float* bias;
    const float* residual; int32_t ldr;
    int32_t act;
    float* pre_out;
    const float* dact_aux; int32_t ld_aux; int32_t dact;
    float alpha;
    int32_t tile_m, variant;
    void* workspace; int64_t workspace_bytes;
    int32_t aux_blocked;   /* tvl_gemm_h2_out only: pre_out (written) / dact_aux (read) is not a row-major matrix but a private buffer of
                            * tvl_gemm_aux_floats(M, N) floats in the producing kernel's accumulator order -- fc1's z handed to its data
                            * gradient's QuickGELU' epilogue; both calls must have the same M and N.  2 (experiment, TVL_GEMM_ZHALF=1): the
                            * buffer holds QuickGELU'(z) as one fp16 per element instead of z as fp32 (half the bytes each way) */
    int32_t a_scale_one;   /* tvl_gemm_h2 / tvl_gemm_h2_out: a_row_scale points at ONE inverse scale for every row of A (a tensor-scaled
                            * activation image, e.g. the attention output that shares the QKV scale) instead of at M of them */
} tvlGemmTp3Args;
int tvl_gemm_tp3(const tvlGemmTp3Args* args, tvlStream_t stream);
/* floats in an aux_blocked buffer for an [M, N] result (whole 256 x 256 tiles), or -1 when the shape cannot use one */
int64_t tvl_gemm_aux_floats(int64_t M, int64_t N);

/* "h2": two fp16 pieces per element (x * s = h0 + h1, s an exact power of two per row or per tensor), same block order as tp3 with
 * 2 KiB per 32 x 16 block.  Three MFMAs per product instead of six at the same (fp32-equivalent) accuracy: csrc/gemm_h2.hip.
 * tvl_h2_pack: fp32 [rows, K] -> image + inverse scales (per_row != 0: inv_scale[rows], the A operand of an activation; else
 * inv_scale[1], a frozen weight; work = 4 bytes of device scratch for the per-tensor maximum).
 * tvl_gemm_h2: tvl_gemm_tp3's argument block with h2 images as A / B; the result is multiplied by alpha (= B's inverse scale) and by
 * a_row_scale[m] (A's inverse row scales, may be null); C_tp3 stays a tp3 image.  Replaces the same nn.Linear calls as tvl_gemm_tp3. */
int64_t tvl_h2_bytes(int64_t rows, int32_t K);
int tvl_h2_pack(const float* x, int64_t ldx, int64_t rows, int32_t K, void* out, float* inv_scale, float* row_norm /* [rows] or null */,
                int32_t per_row, void* work, tvlStream_t stream);
/* tvl_h2_pack of x * (mask > 0): a ReLU layer's data gradient gated while it is packed (mask = the layer's output) */
int tvl_h2_pack_masked(const float* x, int64_t ldx, const float* mask, int64_t ldm, int64_t rows, int32_t K, void* out, float* inv_scale,
                       float* row_norm /* [rows] or null */, int32_t per_row, void* work, tvlStream_t stream);
/* rows b*T + row0 .. + n - 1 (b < B) of an h2 image of [B*T, K] := 0: keeps the image that travels with a gradient valid across the gradient
 * cut of an in-place prompt overwrite (tvl_rows_grad with zero_src; reference base_visual_learner.py:18-23) */
int tvl_h2_zero_rows(void* img, int32_t K, int32_t B, int32_t T, int32_t row0, int32_t n, tvlStream_t stream);
int tvl_h2_absmax(const float* x, int64_t ldx, int64_t rows, int32_t K, void* bits /* 4 bytes: max |x| as float bits */, tvlStream_t stream);
int tvl_gemm_h2(const tvlGemmTp3Args* args, const float* a_row_scale, tvlStream_t stream);
/* ... with the result written as an h2 image (the next GEMM's A operand).  Its row scales come from the bound
 * |out[m, n]| <= out_row_norm[m] * out_mul + out_add (L2 norms of A's rows from A's producer; out_mul = max_n ||B row n||_2 times the
 * activation's Lipschitz bound; out_add = max |bias|); the inverse scales are written to out_inv_scale[M]. */
int tvl_gemm_h2_out(const tvlGemmTp3Args* args, const float* a_row_scale, void* c_h2, const float* out_row_norm, float out_mul, float out_add,
                    float* out_inv_scale, int32_t out_per_tensor /* != 0: one bound out_row_norm[0] and one scale out_inv_scale[0] */, tvlStream_t stream);
/* tvl_gemm_h2 with A scaled per (row, 64-column chunk of K): a_kscale[M][K / 64] inverse scales (the packed attention gradient written by
 * tvl_attn_h2_bwd with g_as_h2 != 0); plain fp32 output: the QKV data gradient. */
int tvl_gemm_h2_ks(const tvlGemmTp3Args* args, const float* a_kscale, tvlStream_t stream);
/* tvl_conv3x3_bf16s on h2 operands (3 MFMAs per product): 3x3 / pad 1 / stride 1, C % 32 == 0.  args->A = the h2 image of the NHWC pixel
 * matrix [B*H*W, C] packed with ONE scale (tvl_h2_pack, per_row = 0; a_scale points at its inverse) and followed by one all-zero 32-row
 * block, i.e. tvl_h2_bytes(B*H*W, C) + tvl_h2_bytes(32, C) bytes (the padding taps read it); args->a_rows = args->M = B*H*W;
 * args->B = the h2 image of the weights [N, 9*C] with the columns ordered (c / 16, ky, kx, c % 16) (taps innermost per 16-channel block:
 * consecutive k-slabs gather from the same rows); args->K = 9*C; fp32 output with the epilogue of tvl_gemm_h2.
 * Replaces the frozen 3x3 convs of reference model/layers.py:12-17,96-119,412-445 and clip.py:44-47 (eval BatchNorm folded). */
int tvl_conv3x3_h2(const tvlGemmTp3Args* args, const tvlConvGeom* geom, const float* a_scale /* [1] */, tvlStream_t stream);
/* Attention on two-piece fp16 operands (3 MFMAs per product; csrc/attention_h2.hip): packed QKV and dO as h2 images with ONE scale each
 * (tvl_gemm_h2_out in per-tensor mode), O and dQ | dK | dV as tp3 images.  delta: [B, H, T] fp32 workspace; dnorm_ws: [B, H] x 4 bytes. */
int tvl_attn_h2_fwd(const void* qkv_h2, const float* qkv_inv, void* o_img, int32_t o_as_h2 /* O as an h2 image sharing the QKV scale, else tp3 */,
                    float* lse, int32_t B, int32_t H, int32_t T, float scale, tvlStream_t stream);
int tvl_attn_h2_bwd(const void* qkv_h2, const float* qkv_inv, const void* o_img, int32_t o_is_h2, const void* do_h2, const float* do_inv, const float* lse,
                    float* delta, void* dnorm_ws, void* dqkv_img, int32_t g_as_h2 /* h2 image + g_kscale [B*T, 3*H], else tp3 */, float* g_kscale,
                    int32_t B, int32_t H, int32_t T, float scale,
                    int32_t only_block /* -1: all rows; >= 0: dQ | dK | dV only for rows 128*only_block .. +127 of every sample (the other rows of the image stay
                                          unwritten): the layer under the visual prompts needs the gradient of the prompt rows alone */,
                    tvlStream_t stream);
/* rows b*T + row0 .. + n - 1 (b < B) of such an h2 gradient image back to fp32: out [B*n, K] */
int tvl_h2k_gather_rows(const void* img, const float* kscale, int32_t K, int32_t B, int32_t T, int32_t row0, int32_t n, float* out, tvlStream_t stream);
/* LayerNorm forward / backward writing their result as an h2 operand (+ inv_scale[rows]); otherwise as tvl_layernorm_fwd_tp3 / _bwd_tp3.
 * max_slot (or null; needs row_norm): 8 bytes of device memory that receive max_m row_norm[m] without being cleared first -- every
 * workgroup does ONE atomicMax of (tag << 32 | float bits of its largest row norm) on the 64-bit word, so the value of the call with
 * the largest tag wins: the caller passes a tag larger than any the slot has seen (zero-initialised memory + a call counter), and the
 * consumer (tvl_gemm_h2_out, out_per_tensor) reads the LOW 32 bits as a float: its out_row_norm.  Replaces a reduction launch per GEMM. */
int tvl_layernorm_fwd_h2(const float* x, const float* gamma, const float* beta, void* y_h2, float* inv_scale, float* row_norm /* or null */, float* mean, float* rstd,
                         int64_t rows, int32_t cols, float eps, uint64_t* max_slot, uint32_t tag, tvlStream_t stream);
int tvl_layernorm_bwd_h2(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres,
                         float* dx, void* dx_h2, float* inv_scale, float* row_norm /* or null */, int64_t rows, int32_t cols,
                         uint64_t* max_slot, uint32_t tag, tvlStream_t stream);

/* nn.AvgPool2d(k) / F.avg_pool2d(x, k, k) on [B,H,W,C] (H, W divisible by k) and its gradient (H, W = input sizes) */
int tvl_avgpool_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, tvlStream_t stream);
int tvl_avgpool_bwd(const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t k, tvlStream_t stream);
/* F.interpolate(scale_factor=s, mode="bilinear") / nn.Upsample (align_corners=False), integer s; bwd: H, W = input sizes */
int tvl_bilinear_up_fwd(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t s, tvlStream_t stream);
int tvl_bilinear_up_bwd(const float* dy, int32_t lddy, float* dx, int32_t lddx, int32_t B, int32_t H, int32_t W, int32_t C, int32_t s, tvlStream_t stream);
/* tvl_bilinear_up_fwd written directly as the h2 image (one scale = the input's, amax_bits from tvl_h2_absmax over x) the next 3x3 conv reads (tvl_conv3x3_h2);
 * C % 16 == 0.  Replaces F.interpolate + the conv's own packing of reference layers.py:100-107. */
int tvl_bilinear_up_h2(const float* x, int32_t ldx, const void* amax_bits, void* out, float* inv_scale, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t s, tvlStream_t stream);
/* y[b] = a * F.interpolate(x[b], (Ho,Wo), mode="bicubic", align_corners=True) + r * extra[b]  (extra may be NULL);
 * single-channel maps [B,Hi,Wi] -> [B,Ho,Wo] (coop_cris.py:235,240-242); bwd: dx = a * bicubic^T(dy) */
int tvl_bicubic_ac_fwd(const float* x, float* y, const float* extra, float a, float r, int32_t B, int32_t Hi, int32_t Wi,
                       int32_t Ho, int32_t Wo, tvlStream_t stream);
int tvl_bicubic_ac_bwd(const float* dy, float a, float* dx, int32_t B, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, tvlStream_t stream);
/* predict tail (reference src/utils/save_utils.py:93-101): one probability map [Hi,Wi] -> TF.resize(size=(Ho,Wo), BICUBIC,
 * antialias=False) (= F.interpolate bicubic, align_corners=False) -> save_image's uint8(clamp(v*255 + 0.5, 0, 255)) */
int tvl_bicubic_resize_u8(const float* x, uint8_t* out, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, tvlStream_t stream);
/* Projector tail (layers.py:106-118): out[b,y,x] = word[b,9C] + sum_{c,ky,kx} word[b, c*9+ky*3+kx] * x[b, y+ky-1, x+kx-1, c]
 * (a grouped conv with one 3x3xC kernel per sample).  taps: workspace [B*H*W, 9]. */
int tvl_dynconv_fwd(const float* x, int32_t ldx, const float* word, int32_t ldw, float* taps, float* out,
                    int32_t B, int32_t H, int32_t W, int32_t C, tvlStream_t stream);
/* dx (may be NULL) [B*H*W, lddx], dword [B, ldw]; work: tvl_dynconv_bwd_work_floats(B,H,W,C) floats (deterministic two-stage sum) */
int64_t tvl_dynconv_bwd_work_floats(int32_t B, int32_t H, int32_t W, int32_t C);
int tvl_dynconv_bwd(const float* dout, const float* x, int32_t ldx, const float* word, int32_t ldw, float* dx, int32_t lddx,
                    float* dword, float* work, int32_t B, int32_t H, int32_t W, int32_t C, tvlStream_t stream);

/* ---- DenseCLIP (BASELINE configs[4]): reference src/models/components/denseclip/models.py:581-600,693-701 (the ViT-B/16 FPN taps of
 * CLIPVisionTransformer) and denseclip.py:157.  Maps are NHWC pixel matrices [B, H*W, C] with row stride ld* and sample stride batch_stride
 * (elements): a tap is the token matrix [B, 1 + H*W, C] minus its CLS row, read in place. */
/* nn.GroupNorm(1, C): per-sample mean / rstd over all H*W*C elements -> stats[2*b] = mean, stats[2*b + 1] = rstd.  Two stages, double sums in a
 * fixed order (bitwise reproducible); work: tvl_groupnorm_work_doubles(B, rows) doubles. */
int64_t tvl_groupnorm_work_doubles(int32_t B, int32_t rows);
int tvl_groupnorm_stats(const float* x, int64_t batch_stride, int32_t ldx, int32_t B, int32_t rows, int32_t C, float eps, float* stats, double* work,
                        tvlStream_t stream);
/* y = (x - mean_b) * rstd_b * gamma[c] + beta[c]; pool = 2: followed by nn.MaxPool2d(2, 2) in the same pass (fpn4, models.py:598-600); y is
 * [B*(H/pool)*(W/pool), ldy] contiguous over samples */
int tvl_groupnorm_apply(const float* x, int64_t batch_stride, int32_t ldx, const float* stats, const float* gamma, const float* beta, float* y, int32_t ldy,
                        int32_t B, int32_t H, int32_t W, int32_t C, int32_t pool, tvlStream_t stream);
/* nn.ConvTranspose2d(C, C, kernel_size=2, stride=2) is a GEMM [pixels, C] x [C, (dy, dx, co)] (kernel == stride: no overlap) whose output rows are the
 * pixels in blocked order; `levels` such GEMMs in a row (fpn1: two, the per-pixel BatchNorm / GELU between them do not care about pixel order)
 * leave [B*H*W*4^(levels-1), 4C]; this puts the result back in raster order: out [B, H*2^levels, W*2^levels, C], row stride ldo. */
int tvl_tconv2x2_unshuffle(const float* in, float* out, int32_t ldo, int32_t B, int32_t H, int32_t W, int32_t C, int32_t levels, tvlStream_t stream);
/* out[r, c] = a[r, c] + g[c] * b[r, c]  (text_embeddings + gamma * text_diff, denseclip.py:157); bwd: db = d * g (may be NULL), dg[c] = sum_r d * b (may be NULL) */
int tvl_colscale_add(const float* a, const float* b, const float* g, float* out, int64_t rows, int32_t cols, tvlStream_t stream);
int tvl_colscale_bwd(const float* d, const float* b, const float* g, float* db, float* dg, int64_t rows, int32_t cols, tvlStream_t stream);
/* out[b*rows + i, k] = full[b*T + skip + i, b*K + k]: the per-sample score maps (denseclip.py:162-165) as the diagonal blocks of ONE GEMM of all samples' pixel rows
 * against all samples' class vectors, full [B*T, ld >= B*K] */
int tvl_blockdiag_gather(const float* full, int32_t ld, float* out, int32_t B, int32_t T, int32_t skip, int32_t rows, int32_t K, tvlStream_t stream);
/* Cross-attention with few queries (Tq <= 32) and many keys, no masks: DenseCLIP's ContextDecoder (models.py:463-481,520-524: the K class embeddings attend over
 * 1 + H*W visual tokens).  The key dimension carries the parallelism and the probabilities are materialised ([B, H, Tq, Tk] fp32); all sums in a fixed order.
 * Matrices are [B*T, ld] with head h in columns h*dh ..; dh in {16, 32, 64}.
 *   tvl_fq_qk       S[b,h,q,k] = alpha * <A[b,q,h,:], Bm[b,k,h,:]>            S = Q K^T * scale;  dP = dO V^T
 *   tvl_fq_softmax  rows of S -> probabilities in place; lse[row] (natural log, may be NULL)
 *   tvl_fq_pk       O[b,q,h,:] = alpha * sum_k P[b,h,q,k] Bm[b,k,h,:]         O = P V;  dQ = scale * dS K
 *   tvl_fq_ds       dP := P * (dP - <dO[b,q,h,:], O[b,q,h,:]>)                 the softmax backward
 *   tvl_fq_tk       G[b,k,h,:] = alpha * sum_q W[b,h,q,k] A[b,q,h,:]          dV = P^T dO;  dK = scale * dS^T Q */
int tvl_fq_qk(const float* A, int32_t lda, const float* Bm, int32_t ldb, float* S, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh, float alpha, tvlStream_t stream);
int tvl_fq_softmax(float* S, float* lse, int64_t rows, int32_t Tk, tvlStream_t stream);
/* (tvl_fq_pk: P[b,h,q,k] at P + b*p_bs + h*p_hs + q*p_qs + k*p_ks, sample b of Bm at Bm + b*bm_bs: also the score map's text-side gradient dT[b] = dS[b]^T V[b] with
 * p_hs = 0, p_qs = 1, p_ks = K over dS [B*H*W, K]) */
int tvl_fq_pk(const float* P, int64_t p_bs, int64_t p_hs, int64_t p_qs, int64_t p_ks, const float* Bm, int64_t bm_bs, int32_t ldb, float* O, int32_t ldo, int32_t B, int32_t H,
              int32_t Tq, int32_t Tk, int32_t dh, float alpha, tvlStream_t stream);
int tvl_fq_ds(const float* P, float* dP, const float* dO, int32_t lddo, const float* O, int32_t ldo, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh, tvlStream_t stream);
int tvl_fq_tk(const float* W, const float* A, int32_t lda, float* G, int32_t ldg, int32_t B, int32_t H, int32_t Tq, int32_t Tk, int32_t dh, float alpha, tvlStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TVL_HIP_H */
