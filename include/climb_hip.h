/* climb_hip.h -- C ABI of libclimb_hip.so: the MI355X (gfx950) device side of CLiMB's ViLT continual-fine-tuning step.
 *
 * The reference (GLAMOR-USC/CLiMB) is pure Python on eager PyTorch; it has no FFI of its own.  Each entry point below
 * replaces the eager-op sequence the reference runs at the cited place (REF = CLiMB src/, HF = transformers
 * models/vilt/modeling_vilt.py 5.15.0, the third-party module REF/modeling/vilt.py:17 imports).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless a comment says otherwise
 *   - every launcher takes the hipStream_t to enqueue on as its last argument (`void* stream`), never synchronises,
 *     never allocates, and is safe to capture into a hipGraph
 *   - return value: 0 = ok, >0 = hipError_t, <0 = library code (-1 invalid argument, -2 unsupported shape);
 *     climb_error_string() renders either.  Nothing throws across this boundary.
 *   - dtype codes: 0 = fp32, 1 = bf16 (raw 16-bit), 2 = split (a (hi, lo) pair of 16-bit planes, see "split-operand arithmetic").  ld* = leading dimension in ELEMENTS.
 *   - token layout: activations are [B, S_pad, H] row-major with S_pad = roundup(T + 1 + NP, 32);
 *     rows [0,T) text, T image [CLS], [T+1, T+1+NP) patches in raster order, the rest zero padding (masked as keys).
 */
#ifndef CLIMB_HIP_H
#define CLIMB_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------------------------------------- */
int climb_version(void);
const char* climb_arch(void);
/* the 16-bit operand type this build of the library computes in: "bf16" (libclimb_hip.so) or "fp16" (libclimb_hip_f16.so, the same sources
 * with -DCLIMB_H16_F16=1).  Every `void*` 16-bit buffer and CLIMB_DT_BF16 below mean that type. */
const char* climb_h16(void);
const char* climb_error_string(int code);
int climb_device_sync(void);
/* diagnostic for bench.py's roofline (no counterpart in the reference): blocks x 4 waves each run `iters` rounds of 16 independent 32x32x16 MFMAs on
 * register operands read once from src (blocks*256*64 16-bit values); out: blocks*256 floats.  2 * 32*32*16 * 16 * iters * 4 * blocks flops. */
int climb_mfma_sustained_probe(const void* src, float* out, int blocks, int iters, void* stream);
/* tuning switches for A/B measurements: key 1 = waves per workgroup of the 128x128 bf16 NT GEMM (4 or 8); key 2 / 4 / 5 = allow the
 * 64x128 / 96x192 / 192x192 NT tile variants (0/1); key 3 = workgroup target of the 128x128 TN split; key 6 = waves per workgroup of the TN GEMM (4 or 8);
 * key 7 = persistent 256-row NT tiles (0 never, 1 auto, 2 / 3 force 256 / 192 columns, 4 two-workgroup variant); key 8 = k-loop-only probe of that kernel;
 * key 9 = its grid; key 10 = persistent TN kernel (0/1); key 11 = de-phasing of the two-workgroup variant; key 12 = query blocks per wave of the
 * bf16 attention forward (0 auto, 1, 2); key 13 = bf16 attention backward as one launch (1, default) or one launch per phase (0);
 * key 14 = split-along-K balancing of the 192-tile NT GEMMs (0 default: measured slower; 1 needs climb_set_nt_workspace);
 * key 15 = store-wave mode of the persistent NT kernel (0 default: measured slower);
 * key 16 = de-phased start of the persistent NT kernel's workgroups in multi-round launches: value % 1000 = hold-back unit (x 64 clocks), value / 1000 = k: 2 << k groups (0 = off) */
int climb_set_option(int key, int value);
/* current value of option `key` (9 = persistent NT grid: what a caller that shrinks it temporarily must put back; 22 = phase groups of the grouped
 * weight-gradient plan, read by climb_tn_grouped_plan when a plan is BUILT -- a caller that caches plans keys them by it); -1 = not readable */
int climb_get_option(int key);

/* ---- embeddings -------------------------------------------------------------------------------------------------- */
/* HF:237-269 TextEmbeddings.forward + HF:208-210: x[b,t,:] = LN(word[ids]+type[tt]+pos[t])*gamma+beta + modality[0].
 * ids/tts are int64 [B,T]; mean/rstd [B*T] are saved for the backward.
 * ids == NULL: `word` is an inputs_embeds matrix [B, emb_ld, H] (emb_ld >= T rows per sequence) used in place of the table lookup
 * (HF TextEmbeddings with inputs_embeds; REF/modeling/viltbert.py:142-147 feeds BERT's last hidden state this way). */
int climb_embed_text_fwd(const long* ids, const long* tts, const float* word, const float* type, const float* pos, const float* gamma, const float* beta, const float* mod0, float eps, float* x, int B, int T, int S_pad, int H, float* mean, float* rstd, int emb_ld, void* stream);
/* backward of the above: scatter-adds into dword (atomics), dpos += sum over the batch; part[ceil(B*T/R)][3][H], R = climb_embed_text_bwd_rows_per_block(), = {dgamma, dbeta,
 * dmodality0} partials; part2[T][2][H] = token-type partials (reduce over T with stride 2H); dpre [B*T,H] scratch */
int climb_embed_text_bwd_rows_per_block(void);
int climb_embed_text_bwd(const long* ids, const long* tts, const float* word, const float* type, const float* pos, const float* gamma, const float* mean, const float* rstd, const float* dres, int B, int T, int S_pad, int H, float* dword, float* dpos, float* dpre, float* part, float* part2, int emb_ld, void* stream);
/* HF:292-300 Conv2d(3,768,k=32,s=32) as a GEMM: out[(b*NP + py*gw + px), c*P*P + ky*P + kx] = pixels[b,c,py*P+ky,px*P+kx] */
int climb_im2col(const float* pixels, void* out, int out_dtype, int B, int C, int H, int W, int P, void* stream);
/* HF:96-99: per-sample valid patch extent {h, w} of a padded variable-resolution batch (pixel_mask int64 [B,H,W]) */
int climb_patch_grid_dims(const long* pixel_mask, int B, int H, int W, int P, int* dims, void* stream);
/* HF:168-173, :211-216 (+ HF:92-166 when dims != NULL): image rows of the embedding = proj/cls + position + modality[img_type[b]].
 * Canvas of NP = gh*gw patches in raster order; dims == NULL: all patches valid and gh = gw = g0 (position table used as is);
 * dims != NULL: patch (py,px) valid iff py < h_b && px < w_b, position table resized bilinearly (align_corners) to (h_b,w_b) on the
 * fly, invalid patches written as zero rows and masked in key_bias.  Padding rows zeroed.
 * compact != 0 (with dims): sample b's valid patches are packed in raster order of its own h_b x w_b grid into rows T+1..T+h_b*w_b
 * (the canvas only addresses `proj`): S_pad >= T + 1 + max_b h_b*w_b suffices, as in the reference (HF:136-159 keeps max_b(h*w) rows). */
int climb_assemble_image(const float* proj, const float* cls, const float* pos, const float* mod, const int* img_type, const int* dims, float* x, float* key_bias, int B, int T, int NP, int gw, int g0, int S_pad, int H, int compact, void* stream);
/* backward of the image rows; part[(NP+1)][ntypes][H] modality partials (reduce with climb_colreduce, stride ntypes*H);
 * with dims the position-table gradient is the transpose of the bilinear resize (gather kernel, g0 = 12) */
int climb_image_embed_bwd(const float* dres, const int* img_type, const int* dims, void* dproj, int dproj_dtype, float* dpos, float* dcls, float* part, int B, int T, int NP, int gw, int g0, int S_pad, int H, int ntypes, int compact, void* stream);
/* HF:623-627 additive key mask as a [B,S_pad] vector: 0 keep, -3e38 masked text token or padding row */
int climb_key_bias(const long* attn_mask, float* bias, int B, int T, int S, int S_pad, void* stream);

/* ---- LayerNorm ------------------------------------------------------------------------------------------------- */
/* nn.LayerNorm (HF:430-451 layernorm_before/after eps 1e-12; REF/modeling/vilt.py:192 head LN eps 1e-5); x fp32, y dtype */
int climb_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, void* y, long ldy, int y_dtype, float* mean, float* rstd, int M, int C, void* stream);
/* dxo = dres_in + LNbwd(dy); optional cast of dxo; part[ceil(M/32)][3][C] = {dgamma, dbeta, colsum(dxo)} partials */
int climb_layernorm_bwd(const void* dy, long lddy, int dtype, const float* x, long ldx, const float* mean, const float* rstd, const float* gamma, const float* dres_in, long ldr, float* dxo, long ldo, void* dcast, long ldc, float* part, int M, int C, void* stream);
int climb_layernorm_bwd_rows_per_block(void);
/* out[c] = beta*out[c] + sum_b part[b*stride + c]  (deterministic second stage of every column reduction) */
int climb_colreduce(const float* part, long stride, int nblk, float* out, int ncols, float beta, void* stream);
/* the {dgamma, dbeta, colsum} triple in ONE launch: out_k[c] = beta*out_k[c] + sum_b part[b*stride + k*ncols + c]; NULL outputs skipped */
int climb_colreduce3(const float* part, long stride, int nblk, float* out0, float* out1, float* out2, int ncols, float beta, void* stream);
/* many such triples in ONE launch (r03: the LayerNorm backwards of a whole group of layers): `segs` = device array of 48-byte records
 * { const float* part; long stride; float* out[3] (NULL = skip); int nblk; int ncols }: out_k[c] += sum_b part[b*stride + k*ncols + c] */
int climb_colreduce_batched(const void* segs, int nseg, int max_cols, void* stream);
/* bias gradients: part[ceil(M/64)][C] column sums of x (optionally also writes a bf16 cast of an fp32 x) */
int climb_colsum(const void* x, long ldx, int in_dtype, void* cast_bf16, long ldc, float* part, int M, int C, void* stream);
int climb_colsum_rows_per_block(void);

/* ---- GEMM ------------------------------------------------------------------------------------------------------ */
/* nn.Linear forward / input-grad / weight-grad (HF:325-327, :366-369, :397-400, :410-414; REF/modeling/vilt.py:190-195)
 * on v_mfma_f32_32x32x2_f32 (exact fp32):  C[m,n] = epi(sum_k A[m*sam+k*sak] * B[n*sbn+k*sbk] + bias[n]) + beta*C[m,n]
 * epi: 0 none, 1 GELU (aux_out = pre-activation), 2 + aux residual, 3 * gelu'(aux), 4 tanh, 5 SiLU (aux_out = pre-activation),
 * 6 * silu'(aux), 7 + aux + aux2 (adapter up-projection with both residuals).  allow_splitk != 0 lets skinny problems (few output tiles,
 * long K; epi 0 only) split K across workgroups with fp32 atomics (sum order then varies from run to run) */
int climb_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, long ldc, int M, int N, int K, const float* bias, int epi, const float* aux, long ldaux, float* aux_out, long ldauxo, float beta, const float* aux2, long ldaux2, int allow_splitk, void* stream);

/* r04: the skinny fp32 products of the pooler and the task heads (REF/modeling/vilt.py:179-203, HF:650-663; M = batch rows): one workgroup per
 * 16-column strip of C holding EVERY row, K split over its 8 waves and summed in LDS in a fixed order (no atomics: run-to-run identical).
 * C[m,n] = epi(sum_k A[m*lda+k] * B[n*sbn+k*sbk] + bias[n]); epi: 0 none, 1 tanh, 2 * (1 - aux^2) (tanh backward), 3 * gelu'(aux).
 * colsum != NULL (M <= 64): colsum[n] = colsum_beta * colsum[n] + sum_m C[m,n] (the bias gradient of the layer below);
 * acol != NULL (M <= 64): acol[k] = acol_beta * acol[k] + sum_m A[m,k] (the bias gradient of the layer that produced A).  lda % 4 == 0. */
int climb_skinny_f32(const float* A, long lda, const float* B, long sbn, long sbk, float* C, long ldc, int M, int N, int K, const float* bias, int epi, const float* aux, long ldaux, float* colsum, float colsum_beta, float* acol, float acol_beta, void* stream);

/* weight gradient of a head / pooler linear for M = batch rows (REF/modeling/vilt.py:190-195 backward): C[n,k] (ldc) += sum_m dY[m,n] * X[m,k], exact fp32 */
int climb_rank_update_f32(const float* dY, long lddy, const float* X, long ldx, float* C, long ldc, int M, int N, int K, void* stream);
/* REF/modeling/vilt.py:191-193 (head LayerNorm, eps 1e-5, then GELU): zn = LN(x), gz = gelu(zn), both [M, C] fp32 with leading dim ldy; C <= 1536 */
int climb_layernorm_gelu_fwd(const float* x, long ldx, const float* gamma, const float* beta, float eps, float* zn, float* gz, long ldy, float* mean, float* rstd, int M, int C, void* stream);

/* ---- attention ------------------------------------------------------------------------------------------------- */
/* HF:322-351 ViltSelfAttention: softmax(Q K^T / sqrt(d) + key_bias) V per (batch, head); scores never leave the CU.
 * qkv [B*S_pad, 3H] columns [q|k|v]; ctx [B*S_pad, H]; lse [B, heads, S_pad] saved for backward */
int climb_attn_fwd_f32(const float* qkv, const float* key_bias, float* ctx, float* lse, int B, int S_pad, int heads, int head_dim, void* stream);
/* The same attention with DROPOUT ON THE PROBABILITIES, forward only, S_pad <= 64: the frozen BERT of ViLT-BERT while the learner is in
 * train mode (REF/modeling/viltbert.py:115-120 never calls bert.eval(); HFB eager_attention_forward: softmax -> dropout -> P V).
 * qkv / ctx in `dtype` (0 fp32, 1 the library's 16-bit type); keep uint8 [B, heads, T, T] (1 = kept), drop_scale = 1 / (1 - p). */
int climb_attn_fwd_dropout(const void* qkv, const float* key_bias, const void* keep, void* ctx, int dtype, int B, int S_pad, int heads, int head_dim, int T, float drop_scale, void* stream);
/* delta[b,h,q] = sum_d dctx*ctx (softmax backward row term) */
int climb_attn_delta(const void* dctx, const void* ctx, int dtype, float* delta, int B, int S_pad, int heads, void* stream);
int climb_attn_bwd_f32(const float* qkv, const float* key_bias, const float* dctx, const float* lse, const float* delta, float* dqkv, int B, int S_pad, int heads, int head_dim, void* stream);

/* ---- heads / losses -------------------------------------------------------------------------------------------- */
/* op: 0 gelu(a) | 1 a*gelu'(b) | 2 a*(1-b^2) (tanh bwd) | 3 a*b*s (dropout) | 4 a*s | 5 a+b | 6 tanh(a) */
int climb_elementwise(int op, const float* a, const float* b, float* out, long n, float s, void* stream);
/* REF/train/visionlanguage_tasks/train_vqa.py:95,:157: loss = BCEWithLogits(mean)*N; dlogits = gscale*(sigmoid(x)-t)/B */
int climb_bce_logits(const float* logits, long ldl, const float* target, long ldt, float* dlogits, long ldd, float* loss, float* partials, int B, int N, float gscale, void* stream);
int climb_bce_workspace_floats(void);
/* REF/train/visionlanguage_tasks/train_nlvr2.py:80 nn.CrossEntropyLoss(); labels int64 */
int climb_cross_entropy(const float* logits, long ldl, const long* labels, float* dlogits, long ldd, float* loss, int B, int N, float gscale, void* stream);

/* ---- optimiser / continual-learning terms (flat parameter buffer) ------------------------------------------------ */
/* torch.optim.AdamW as built by REF/modeling/vilt.py:205-215.  seg_start[nseg+1] (int64, device) tensor offsets,
 * seg_group[nseg] (int8, device) group id or -1 = tensor skipped; groups = HOST array [ngroups][8] =
 * {lr, wd, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, 0}.  Optionally refreshes the bf16 weight shadow in the same pass. */
int climb_adamw(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, const long* seg_start, const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, void* stream);
/* r04: the same update over the ACTIVE spans of the flat buffer only (tensors the grouped weight-gradient launch updated in its epilogue, frozen
 * or untouched tensors are not walked).  spans (device) = nspans x { first element, elements, first 1024-element block, source }, ascending; nblocks =
 * blocks of all spans; seg_start / seg_group / groups as above.  zero_grad != 0: every gradient element it consumes is cleared (the
 * optimizer.zero_grad() of REF/train/visionlanguage_tasks/train_vqa.py:168-172 folded into the pass).  A span with source != 0 reads its gradient
 * from g16 (the data-parallel reducer's 16-bit payload buffer, laid out like g) times g16_scale instead of from g. */
int climb_adamw_spans(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks, const long* seg_start, const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, int zero_grad, const void* g16, float g16_scale, void* stream);
/* r05: the same pass with the EWC term (REF/cl_algorithms/ewc.py:75-87) folded in: elements below enc_n (the encoder range; star / fisher = theta* / F laid
 * out like it) get 2 lam F (theta - theta*) added to their gradient, and lam F (theta - theta*)^2 is ADDED to *ewc_loss (zero it once per step). */
int climb_adamw_spans_ewc(float* p, float* g, float* m, float* v, void* shadow_bf16, const long* spans, int nspans, long nblocks, const long* seg_start, const signed char* seg_group, int nseg, const float* groups, int ngroups, float gscale, int zero_grad, const void* g16, float g16_scale, const float* star, const float* fisher, long enc_n, float lam, float* ewc_loss, void* stream);
/* REF/cl_algorithms/ewc.py:75-87: loss_out = lam*sum F (theta-theta*)^2; grad += gscale*2*lam*F*(theta-theta*) if grad != NULL */
int climb_ewc_penalty(const float* theta, const float* star, const float* fisher, float* grad, long n, float lam, float gscale, float* partials, float* loss_out, void* stream);
int climb_ewc_workspace_floats(void);
/* REF/cl_algorithms/ewc.py:62-64: fisher += grad^2 */
int climb_fisher_accum(float* fisher, const float* grad, long n, void* stream);
int climb_scale(float* x, long n, float s, void* stream);
int climb_cast_bf16(const float* x, void* y, long n, void* stream);
/* data-parallel payload helpers: y[n] (f32) = scale * x[n] (bf16), and x[n] *= scale in place; n % 4 == 0, 16-byte aligned */
int climb_uncast_bf16_scale(const void* x, float* y, long n, float scale, void* stream);
int climb_scale_f32(float* x, long n, float scale, void* stream);
int climb_transpose_bf16(const void* in, void* out, int R, int C, void* stream);
/* n transposes in one launch: table[i] = {src_off, dst_off, R, C} in elements (int64, device); grid = (tiles_per_matrix, n) */
int climb_transpose_bf16_batched(const void* src, void* dst, const long* table, int n, int tiles_per_matrix, void* stream);

/* ---- bf16 throughput path (v_mfma_f32_32x32x16_bf16, fp32 accumulate) --------------------------------------------------- */
/* nn.Linear forward and input-gradient GEMMs (HF:325-327, :366-369, :397-400, :410-414):
 * C[M,N] (c_dtype) = epi(A[M,K] B[N,K]^T + bias); A,B bf16, K contiguous.  epi 1: aux_out (bf16) = pre-activation;
 * epi 2: aux = fp32 residual [M,N]; epi 3: aux = bf16 pre-activation (multiplies by gelu'); epi 5/6: SiLU / silu' likewise;
 * epi 7: aux = fp32 residual, aux2 = bf16 residual (Houlsby adapter up-projection: out = up(s) + sublayer_out + x).
 * r05, 16-bit C only: epi 8 = GELU whose aux_out receives gelu'(pre-activation) instead of the pre-activation (HF/modeling_vilt.py:393, 397-414: the
 * intermediate activation and what its backward needs); epi 9: C = (A B^T) * aux, aux = the 16-bit tensor epi 8 saved. */
int climb_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi, const void* aux, long ldaux, void* aux_out, long ldauxo, const void* aux2, long ldaux2, void* stream);
/* Houlsby bottleneck adapter forward in one launch (16-bit mode; replaces the two skinny NT GEMMs of engine.py's adapter branch, which follow
 * the GLAMOR fork's arithmetic as restated in climb_amd/cl_algorithms/adapters.py): z = y wd^T + bd; s = silu(z); out = resid + y + s wu^T + bu.
 * y [M,H] 16-bit, resid / out [M,H] fp32, wd [r,H] / wu [H,r] 16-bit, bd [r] / bu [H] fp32, z / s [M,r] 16-bit (saved for the backward).
 * H % 128 == 0, r % 16 == 0, r <= 64; CLIMB_EUNSUPPORTED otherwise (the caller keeps the two-GEMM path). */
int climb_adapter_fwd_bf16(const void* y, long ldy, const float* resid, long ldr, const void* wd, const float* bd, const void* wu, const float* bu, void* z, void* s, long ldz, float* out, long ldo, int M, int H, int r, void* stream);
/* the input-gradient half of the same adapter's backward, one launch: dz = (dout Wu) * silu'(z), dy = dres + dz Wd.  dout [M,H] 16-bit, dres [M,H]
 * fp32 (the residual gradient), wu_t [r,H] / wd_t [H,r] = the TRANSPOSED 16-bit weight shadows, z [M,r] the forward's saved pre-activation;
 * outputs dz [M,r] (kept for the down-projection's weight gradient) and dy [M,H], 16-bit */
int climb_adapter_bwd_bf16(const void* dout, long ldd, const float* dres, long ldr, const void* wu_t, const void* wd_t, const void* z, void* dz, long ldz, void* dy, long ldo, int M, int H, int r, void* stream);
/* weight gradient: C[N,K] (fp32) += A[M,N]^T B[M,K]; reduction over tokens via LDS transpose reads, split over M with fp32 atomics;
 * dbias (optional, fp32 [N]) += column sums of A = the bias gradient of the same layer (one extra MFMA against an all-ones operand) */
int climb_gemm_bf16_tn(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, float* dbias, void* stream);
/* optional scratch for the split partial sums of climb_gemm_bf16_tn (the library never allocates): with a registered buffer of
 * >= splits*N*K*4 bytes the partials are written as plain stores and summed by a second launch on the same stream; without one they
 * are accumulated with fp32 atomics.  64 MB covers every ViLT-B shape at 12288 tokens.  Caller-owned, stream-ordered use. */
int climb_set_tn_workspace(void* ptr, long bytes);
/* optional scratch for climb_gemm_bf16_nt (r03; used only under climb_set_option(14, 1)): with a registered buffer of >= climb_nt_workspace_bytes() the GEMMs that tile into exactly
 * 192 tiles of 256 x 192 with K >= 1536 (the layer's down-projection, HF:410-414, and the input gradients dhn / dxn) run on all 256 CUs --
 * four workgroups share three tiles along K and hand partial accumulator tiles over through this buffer; without it 64 CUs idle.
 * Caller-owned, stream-ordered use, one stream at a time. */
int climb_set_nt_workspace(void* ptr, long bytes);
int climb_nt_workspace_bytes(void);
/* Grouped weight gradients (r03): the dW GEMMs of SEVERAL layers (HF:325-327, :366-369, :397-414: dW2, dW1, dWo, dWqkv, and the patch
 * projection's, HF:292-300) as ONE persistent launch.  Every 256 x 256 output tile that fits a whole round of the chip reduces over all
 * tokens inside one workgroup and is added to C with a plain read-modify-write: no token split, no partial sums, no reduce launch;
 * the last (tiles mod nwg) tiles are cut stream-K style into equal shares whose sums meet in C through fp32 atomics.
 *   probs : device array of 72-byte records { const void* A [M,N] 16-bit, token-major; const void* B [M,K]; float* C [N,K]; float* dbias
 *           [N] or NULL; long lda, ldb, ldc; int M, N, K, reserved }   --  C += A^T B, dbias += column sums of A
 *   items : device int32 [n_items][8] = { problem, tile n, tile k, first reduction tile, end reduction tile, partial, 0, 0 }
 *   first : device int32 [nwg + 1]: workgroup b walks items first[b] .. first[b + 1] - 1
 * climb_tn_grouped_plan fills HOST copies of `items` (capacity `cap` records) and `first` from the problems' shapes (host arrays; M % 128,
 * N % 8, K % 8 required; `ragged` = 1 at the launch when some N or K is not a multiple of 256: surplus tile columns are computed on clamped
 * addresses and not stored -- the adapters' 768 x 48 / 48 x 768 gradients) for `nwg` workgroups (a multiple of 8; 256 on MI355X) and returns the number of items (or a negative code);
 * the caller uploads them once per shape and keeps them -- the launch itself allocates and copies nothing. */
int climb_tn_grouped_plan(int nprob, const int* M, const int* N, const int* K, int nwg, int* items_out, int cap, int* first_out);
int climb_gemm_bf16_tn_grouped(const void* probs, const void* items, const void* first, int nwg, int ragged, void* stream);
/* r04: the same launch with the optimizer in its epilogue (REF/modeling/vilt.py:205-215, REF/train/visionlanguage_tasks/train_vqa.py:160-170: nothing reads
 * a weight gradient between backward() and optimizer.step()).  opts: device array parallel to probs of 56-byte records { float* p, m, v [N,K] (leading
 * dimension = the problem's ldc); void* s [N,K] 16-bit shadow; void* st [K,N] transposed 16-bit shadow; long ldt; int fused, pad }.  A WHOLE tile of a problem
 * with fused = 1 applies AdamW to its elements -- g = gscale * (tile sum + C if grad_dirty) -- and writes p, m, v and both shadows, never C; everything
 * else behaves as climb_gemm_bf16_tn_grouped.  A problem may be fused only if none of its tiles is `partial` in the plan.  adam: HOST array of 8 floats
 * { lr, weight_decay, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t, gscale } (climb_adamw's group row + the gradient scale). */
int climb_gemm_bf16_tn_grouped_adamw(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam, int grad_dirty, void* stream);
/* r05: ... and the EWC term in that epilogue: `flat` = the fp32 parameter buffer the fused problems' p pointers point into, star / fisher laid out like its
 * encoder range; *loss += lam sum F (theta - theta*)^2 over the elements updated here.  Not for ragged plans or a non-zero gradient buffer (CLIMB_EUNSUPPORTED). */
int climb_gemm_bf16_tn_grouped_adamw_ewc(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam, int grad_dirty, const float* flat, const float* star, const float* fisher, float lam, float* loss, void* stream);
/* HF:322-351 in bf16: same contract as the _f32 entry points, qkv/ctx/dctx/dqkv are bf16.  The backward takes the forward's ctx and
 * computes delta itself (its first phase); `delta` [B,heads,S_pad] is scratch it writes */
int climb_attn_fwd_bf16(const void* qkv, const float* key_bias, void* ctx, float* lse, int B, int S_pad, int heads, int head_dim, void* stream);
int climb_attn_bwd_bf16(const void* qkv, const float* key_bias, const void* dctx, const void* ctx, const float* lse, float* delta, void* dqkv, int B, int S_pad, int heads, int head_dim, void* stream);

/* ---- split-operand arithmetic (r06): the fast mode INSIDE north_star's tolerance ---------------------------------------------------------
 * HF:325-327, :366-369, :397-414 (every nn.Linear of the layer, forward / input gradient / weight gradient) with each GEMM operand carried as a PAIR of
 * 16-bit planes, x ~= hi + lo, hi = rn16(x), lo = rn16(x - hi), and each product as three MFMA passes hi.hi + hi.lo + lo.hi accumulated in fp32: with
 * bf16 planes 16 significant bits per operand at fp32's range (the reference computes in fp32: REF/train/visionlanguage_tasks/train_vqa.py:135-174),
 * at a third of the 16-bit MFMA rate instead of the sixteenth v_mfma_f32_32x32x2_f32 runs at.
 * dtype code 2 = CLIMB_DT_SPLIT (accepted by climb_layernorm_fwd's y_dtype and climb_layernorm_bwd's dtype, where it means: dy fp32, dcast split): the
 * pointer names the hi plane [rows, ld]; the lo plane lies rows * ld elements behind it (a split tensor is [2][rows][ld] 16-bit). */
/* y (split, lo plane lo_off elements behind y) = f(x), x fp32 [M, C]: mode 0 f = x; 1 f = gelu(x) (erf form, HF:393); 2 f = x * gelu'(aux), aux fp32 [M, C].
 * With M = 1 and C = n it splits a flat buffer (the weight shadow: lo_off = the buffer's length). */
int climb_split_f32(const float* x, long ldx, void* y, long ldy, long lo_off, int M, int C, int mode, const float* aux, long ldaux, void* stream);
/* C[M,N] (fp32) = epi(A B^T + bias); A = split [M,K] (lda, lo plane a_lo elements behind), B = split [N,K] (ldb, b_lo); epi 0 none, 2 + aux (fp32 [M,N]).
 * K % 64 == 0.  Runs on the four-wave persistent kernel when the shape tiles into 192 x 192 (gemm_bf16_nt4.hip), on the 128 x 128 kernel otherwise. */
int climb_gemm_split_nt(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias, int epi, const float* aux, long ldaux, void* stream);
/* The MLP's activation epilogues on the four-wave kernel (HF:393, :397-414).  epi 10: C (fp32 [M,N]) = A B^T + bias (the pre-activation the backward needs), out (split
 * [M,N], ldo, lo plane o_lo elements behind) = gelu(C); epi 11: out = (A B^T) * gelu'(aux), aux = that fp32 pre-activation, C unused (may be NULL).  GELU in its erf form
 * to 1.5e-7.  climb_gemm_split_nt_takes_act(M, N, K) = 1 for the shapes it takes (192 x 192 tiles, (M / 192) % 8 == 0); otherwise the caller runs
 * climb_gemm_split_nt + climb_split_f32 (mode 1 / 2). */
int climb_gemm_split_nt_takes_act(int M, int N, int K);
int climb_gemm_split_nt_act(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, void* out, long ldo, long o_lo, int M, int N, int K, const float* bias, int epi, const float* aux, long ldaux, void* stream);
/* C[N,K] (fp32) += A^T B over M tokens; A = split [M,N], B = split [M,K]; dbias (optional) += column sums of A.  Three ordinary weight-gradient launches:
 * the path of shapes climb_gemm_split_tn_grouped does not take. */
int climb_gemm_split_tn(const void* A, long lda, long a_lo, const void* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, float* dbias, void* stream);
/* climb_gemm_bf16_tn_grouped over split operands: a problem's A / B name the hi planes of pairs whose lo plane DIRECTLY follows ([2 Mt, .]); its M field
 * (and the M handed to climb_tn_grouped_plan) = 3 Mt (three products per 64-token tile), and its `reserved` field = Mt / 64.  N % 256 == K % 256 == 0. */
int climb_gemm_split_tn_grouped(const void* probs, const void* items, const void* first, int nwg, void* stream);
/* ... with AdamW in its epilogue (climb_gemm_bf16_tn_grouped_adamw's contract for opts / adam / grad_dirty): a whole tile of a fused problem writes p, m, v and the
 * hi AND lo planes of the straight and the transposed shadow; s_lo / st_lo = elements from the hi to the lo plane of those two buffers */
int climb_gemm_split_tn_grouped_adamw(const void* probs, const void* items, const void* first, int nwg, const void* opts, const float* adam, int grad_dirty, long s_lo, long st_lo, void* stream);

/* HF:322-351 on split operands (csrc/attention_split.hip): the fp32 entry points' contract -- qkv / dctx fp32, lse and delta as above -- with every product as
 * three bf16 MFMA passes over (hi, lo) planes formed on the way into LDS / registers.  The forward writes ctx as fp32 (may be NULL) and / or as split planes
 * (ctx_split = the hi plane [B*S_pad, H], the lo plane ctx_lo elements behind; may be NULL); the backward writes d(qkv) as fp32 and / or split planes likewise. */
int climb_attn_fwd_split(const float* qkv, const float* key_bias, float* ctx, void* ctx_split, long ctx_lo, float* lse, int B, int S_pad, int heads, int head_dim, void* stream);
int climb_attn_bwd_split(const float* qkv, const float* key_bias, const float* dctx, const float* lse, const float* delta, float* dqkv, void* dqkv_split, long dqkv_lo, int B, int S_pad, int heads, int head_dim, void* stream);

/* ---- image pre-processing on the device (replaces the host call REF/modeling/vilt.py:86-96 -> ViltProcessor ->
 * transformers image_processing_pil_vilt.py:127-242 -> Pillow Resample.c).  All images of a batch live in byte arenas; `table` holds
 * 16 longs per image: byte offsets of its raw [sh][sw][3] pixels, its [sh][dw][3] intermediate and its [dh][dw][3] result, then
 * sh sw dh dw, then int offsets into `coef` of the horizontal coefficients / bounds and their row length, the same for vertical.
 * Coefficients are Pillow's 22-bit fixed point (computed by the host: climb_amd/data/image_pipeline.py). */
int climb_image_resample(const void* src, void* tmp, void* dst, const int* coef, const long* table, int n_images, long max_elems, void* stream);
/* rescale + normalise (256-entry table) + zero pad to [B,3,Hc,Wc] + pixel_mask [B,Hc,Wc] (int64, 1 = real pixel) */
int climb_image_normalize_pad(const void* img, const long* table, const float* lut, float* pixel_values, long* pixel_mask, int n_images, int Hc, int Wc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
