// Multi-head self-attention of the split-operand mode (r06; HF modeling_vilt.py:322-351): the data flow of attention_f32.hip -- one workgroup per
// (batch, head), K / V (backward: the inner operand pair) resident in LDS, one 32-row block of the outer operand per wave, scores computed "swapped" so
// that a lane owns one outer row (forward: one pass with an online softmax) -- with every product on
// v_mfma_f32_32x32x16_bf16 over (hi, lo) bf16 planes: a b = a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate.  16 significant bits per operand
// (split.hip) at 3/16 of the MFMA time of the exact-fp32 instruction: 12 MFMAs of 32 cycles per (32 x 32 x 64) product instead of 32 of 64.
//
// Operands arrive as fp32 (the QKV GEMM's output, d(ctx)) and are split on the way into LDS / registers; nothing 16-bit is read from HBM.  The
// probabilities and d(scores) are split in registers: in the 32 x 32 accumulator layout a lane's 16 values ARE an MFMA B operand of the next product
// (two k-steps of 8 keys each, in the lane's own key order), the matching A operand -- 8 strided rows of one column of a row-major LDS image -- is
// gathered by two ds_read_b64_tr_b16 per plane.
//
//   LDS image of a [rows][64] operand plane: bf16, row stride 144 B (128 + 16: the 16 rows a ds_read_b128 lane group touches start in 16 different
//   4-bank slots; the transpose reads see 2-way conflicts on half their lanes -- LDS time is a tenth of the MFMA time here); hi plane, then lo plane.
//
// Outputs: ctx as fp32 (the backward's softmax row term reads it) AND as split planes (the out-projection's operand); d(qkv) as split planes (only the
// two QKV gradient GEMMs read it).
#include "common.h"

#define AS_D 64
#define AS_PITCH 144            // bytes per LDS row
#define AS_MAXKEYS 192          // rows of the inner operands resident per chunk

typedef __attribute__((ext_vector_type(4))) unsigned int as_u32x4;

__device__ __forceinline__ int as_drow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ bf16x8 as_bits(as_u32x4 v) {
  union { as_u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}
// (hi, lo) planes of 8 consecutive fp32 values, optionally scaled by a power of two first (exact)
__device__ __forceinline__ void as_split8(const float4& a, const float4& b, float sc, bf16x8& hi, bf16x8& lo) {
  const float x[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
  as_u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
    l[i] = pack_bf16x2(x[2 * i] - h16lo_to_f32(h[i]), x[2 * i + 1] - h16hi_to_f32(h[i]));
  }
  hi = as_bits(h);
  lo = as_bits(l);
}
// the lane's 16 accumulator values as two B operands (k-steps of 8 rows: r = 0..7, 8..15), hi and lo planes
__device__ __forceinline__ void as_split16(const f32x16& v, bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    as_u32x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = v[8 * t + 2 * i], b = v[8 * t + 2 * i + 1];
      h[i] = pack_bf16x2(a, b);
      l[i] = pack_bf16x2(a - h16lo_to_f32(h[i]), b - h16hi_to_f32(h[i]));
    }
    hi[t] = as_bits(h);
    lo[t] = as_bits(l);
  }
}
// 32 rows x 64 columns of an outer operand as MFMA B fragments: lane (row l31, half) holds columns 16 ks + 8 half .. + 8 of its row
__device__ __forceinline__ void as_load_outer(const float* __restrict__ g, long ld, int l31, int half, float sc, bf16x8 (&hi)[4], bf16x8 (&lo)[4]) {
  const float* p = g + (long)l31 * ld + 8 * half;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) as_split8(ld4(p + 16 * ks), ld4(p + 16 * ks + 4), sc, hi[ks], lo[ks]);
}
// rows [0, nrows) x 64 fp32 columns of TWO strided global matrices -> their (hi, lo) LDS images.  Every thread requests a batch of up to 2 x AS_UB float4
// (forward: 2 -- its register budget is 168 for three waves per SIMD; backward: 4) before it converts the first one: with one load per loop iteration (r06, first version) a thread's 16 loads per chunk were 16 SERIAL round trips -- a
// third of the forward's time at S_pad = 192.
__device__ __forceinline__ void as_store_split(unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, int r, int c, const float4& v) {
  uint2 h, l;
  h.x = pack_bf16x2(v.x, v.y);
  h.y = pack_bf16x2(v.z, v.w);
  l.x = pack_bf16x2(v.x - h16lo_to_f32(h.x), v.y - h16hi_to_f32(h.x));
  l.y = pack_bf16x2(v.z - h16lo_to_f32(h.y), v.w - h16hi_to_f32(h.y));
  *reinterpret_cast<uint2*>(hi + r * AS_PITCH + c * 2) = h;
  *reinterpret_cast<uint2*>(lo + r * AS_PITCH + c * 2) = l;
}
template <int AS_UB>
__device__ __forceinline__ void as_stage2(unsigned char* __restrict__ hi0, unsigned char* __restrict__ lo0, const float* __restrict__ src0, long ld0,
                                          unsigned char* __restrict__ hi1, unsigned char* __restrict__ lo1, const float* __restrict__ src1, long ld1, int nrows, int tid,
                                          int nthreads) {
  const int n = nrows * 16;
  for (int e0 = tid; e0 < n; e0 += nthreads * AS_UB) {
    float4 v0[AS_UB], v1[AS_UB];
#pragma unroll
    for (int u = 0; u < AS_UB; ++u) {
      const int e = e0 + u * nthreads;
      if (e < n) {
        const int r = e >> 4, c = (e & 15) * 4;
        v0[u] = ld4(src0 + (long)r * ld0 + c);
        if (src1) v1[u] = ld4(src1 + (long)r * ld1 + c);
      }
    }
#pragma unroll
    for (int u = 0; u < AS_UB; ++u) {
      const int e = e0 + u * nthreads;
      if (e < n) {
        const int r = e >> 4, c = (e & 15) * 4;
        as_store_split(hi0, lo0, r, c, v0[u]);
        if (src1) as_store_split(hi1, lo1, r, c, v1[u]);
      }
    }
  }
}
// A operand, row-major: rows row0 + l31, columns 16 ks + 8 half .. + 8
__device__ __forceinline__ bf16x8 as_frag(const unsigned char* __restrict__ img, int row0, int l31, int half, int ks) {
  return as_bits(*reinterpret_cast<const as_u32x4*>(img + (row0 + l31) * AS_PITCH + (16 * ks + 8 * half) * 2));
}
// A operand, transposed: MFMA row = column dblk * 32 + l31 of the image, k = the lane's own accumulator row order: rows row0 + 16 t + 8 (jj >> 2) + 4 half + (jj & 3)
__device__ __forceinline__ bf16x8 as_frag_t(const unsigned char* __restrict__ img, int row0, int t, int dblk, int lane) {
  const int g16 = lane >> 4, i = lane & 15;
  const int row = row0 + 16 * t + 4 * (g16 >> 1) + (i >> 2);
  const int col = dblk * 32 + (g16 & 1) * 16 + 4 * (i & 3);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const unsigned char* p = img + row * AS_PITCH + col * 2;
  union { s16x4 h[2]; bf16x8 b; } c;
  c.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  c.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 8 * AS_PITCH));
  return c.b;
}
// acc += (a_hi, a_lo) x (b_hi, b_lo) without the lo.lo term
__device__ __forceinline__ void as_mma3(f32x16& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
}
// acc[rows of the image block][outer rows] = image block (row-major) x outer registers over the 64 columns
__device__ __forceinline__ void as_prod_rm(f32x16& acc, const unsigned char* __restrict__ ihi, const unsigned char* __restrict__ ilo, int row0, int l31, int half,
                                           const bf16x8 (&bh)[4], const bf16x8 (&bl)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) as_mma3(acc, as_frag(ihi, row0, l31, half, ks), as_frag(ilo, row0, l31, half, ks), bh[ks], bl[ks]);
}
// acc[dblk][column dblk*32 + .. of the image][outer rows] += image block^T x the lane's 16 split values (rows row0 .. row0 + 31 of the image)
__device__ __forceinline__ void as_prod_t(f32x16 (&acc)[2], const unsigned char* __restrict__ ihi, const unsigned char* __restrict__ ilo, int row0, int lane,
                                          const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int d = 0; d < 2; ++d) as_mma3(acc[d], as_frag_t(ihi, row0, t, d, lane), as_frag_t(ilo, row0, t, d, lane), bh[t], bl[t]);
}
__device__ __forceinline__ void as_zero(f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
}
// the lane's column of a transposed accumulator pair (acc[d][r] = out[row l31][column d*32 + drow(r, half)]): fp32 row and / or split planes
__device__ __forceinline__ void as_store_row(const f32x16 (&acc)[2], float* __restrict__ f32row, bf16_t* __restrict__ hirow, long lo_off, int half) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = make_float4(acc[d][4 * g], acc[d][4 * g + 1], acc[d][4 * g + 2], acc[d][4 * g + 3]);
      const int c = d * 32 + 8 * g + 4 * half;
      if (f32row) st4(f32row + c, v);
      if (hirow) split_st4(hirow + c, lo_off, v);
    }
}

// ---------------------------------------------------------------------------------------------- forward
// dynamic LDS: K hi | K lo | V hi | V lo (CK rows each) | bias_s[S_pad]
// (second launch bound = waves per SIMD the register budget must allow: 3 = two 6-wave workgroups per CU at 96-row chunks)
template <int NW>
__global__ __launch_bounds__(64 * NW, 3) void attn_fwd_split_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias, float* __restrict__ ctx,
                                                                 bf16_t* __restrict__ ctx_hi, long ctx_lo, float* __restrict__ lse_out, int S_pad, int heads, int CK,
                                                                 float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Kh = smem;
  unsigned char* Kl = Kh + CK * AS_PITCH;
  unsigned char* Vh = Kl + CK * AS_PITCH;
  unsigned char* Vl = Vh + CK * AS_PITCH;
  float* bias_s = reinterpret_cast<float*>(Vl + CK * AS_PITCH);
  const int H = heads * AS_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const float* Qg = qkv + (long)b * S_pad * ld + h * AS_D;
  const float* Kg = Qg + H;
  const float* Vg = Qg + 2 * H;
  const int NB = S_pad / 32;
  const int nchunks = (S_pad + CK - 1) / CK;
  for (int i = tid; i < S_pad; i += 64 * NW) bias_s[i] = key_bias[(long)b * S_pad + i];
  __syncthreads();

  for (int round = 0; round * NW < NB; ++round) {
    const int qb = round * NW + wid;
    const bool active = qb < NB;
    bf16x8 qh[4], ql[4];
    as_load_outer(Qg + (long)(active ? qb : 0) * 32 * ld, ld, l31, half, scale, qh, ql);          // Q / sqrt(d): a power of two, exact
    // ONE pass over the keys (online softmax): running maximum m and sum l per query, the unnormalised O^T rescaled whenever m moves; normalised at the end
    // (the exact-fp32 kernel takes two passes -- statistics, then exp(s - lse) -- at twice the score products; the backward recomputes P from lse either way)
    float m_run = -3.0e38f, l_run = 0.f;
    f32x16 o[2];
    as_zero(o[0]);
    as_zero(o[1]);
    for (int ch = 0; ch < nchunks; ++ch) {
      const int k0 = ch * CK, nk = min(CK, S_pad - k0);
      if (nchunks > 1 || round == 0) {
        __syncthreads();
        as_stage2<2>(Kh, Kl, Kg + (long)k0 * ld, ld, Vh, Vl, Vg + (long)k0 * ld, ld, nk, tid, 64 * NW);
        __syncthreads();
      }
      if (active)
        for (int kb = 0; kb < nk / 32; ++kb) {
          f32x16 s;
          as_zero(s);
          as_prod_rm(s, Kh, Kl, kb * 32, l31, half, qh, ql);
          float mx = -3.0e38f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[r] += bias_s[k0 + kb * 32 + as_drow(r, half)];
            mx = fmaxf(mx, s[r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float m_new = fmaxf(m_run, mx);
          const float alpha = __expf(m_run - m_new);
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            sum += s[r];
          }
          sum += __shfl_xor(sum, 32, 64);
          l_run = l_run * alpha + sum;
          m_run = m_new;
          if (alpha != 1.0f) {          // (wave-divergent only while the maximum still moves: the first blocks of a row)
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
          }
          bf16x8 ph[2], pl[2];
          as_split16(s, ph, pl);
          as_prod_t(o, Vh, Vl, kb * 32, lane, ph, pl);
        }
    }
    const float lse = m_run + __logf(l_run);
    {
      const float inv = 1.0f / l_run;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= inv;
    }
    if (active) {
      const long row = (long)b * S_pad + qb * 32 + l31;
      as_store_row(o, ctx ? ctx + row * H + h * AS_D : nullptr, ctx_hi ? ctx_hi + row * H + h * AS_D : nullptr, ctx_lo, half);
      if (half == 0) lse_out[((long)b * heads + h) * S_pad + qb * 32 + l31] = lse;
    }
  }
}

// measurement knobs (climb_set_option 23 / 24 / 25): rows of the inner operands resident per chunk (LDS = 4 planes x 144 B x rows: 192 rows = 111 KB = one
// workgroup per CU, 96 rows = 55 KB = two), waves per workgroup of the forward / the backward (0 = as_waves)
static int g_as_maxkeys = AS_MAXKEYS, g_as_nw_fwd = 0, g_as_nw_bwd = 0;
void climb_attn_split_set(int key, int v) {
  if (key == 23 && v >= 32 && v <= AS_MAXKEYS && v % 32 == 0) g_as_maxkeys = v;
  if (key == 24 && (v == 0 || v == 4 || v == 6 || v == 8)) g_as_nw_fwd = v;
  if (key == 25 && (v == 0 || v == 4 || v == 6 || v == 8)) g_as_nw_bwd = v;
}
static int as_waves(int NB) {          // waves per workgroup: one round of query blocks where four waves do not suffice, and then eight -- the two idle waves of
  return NB <= 4 ? 4 : 8;              // S_pad = 192 still stage (measured r06, tools/attn_split_bench.py: forward 62.5 / 58.0 us, backward 163.7 / 150.8 us with 6 / 8)
}
static int as_chunk(int S_pad) {          // equal chunks of whole 32-row blocks, at most g_as_maxkeys rows each
  const int nb = S_pad / 32, nch = (S_pad + g_as_maxkeys - 1) / g_as_maxkeys;
  return ((nb + nch - 1) / nch) * 32;
}

// qkv fp32 [B*S_pad, 3H] -> ctx fp32 [B*S_pad, H] (may be NULL) and / or ctx_split (hi plane; the lo plane ctx_lo elements behind; may be NULL), lse
extern "C" int climb_attn_fwd_split(const float* qkv, const float* key_bias, float* ctx, void* ctx_split, long ctx_lo, float* lse, int B, int S_pad, int heads,
                                    int head_dim, void* stream) {
  if (head_dim != AS_D || S_pad % 32 || S_pad <= 0 || (!ctx && !ctx_split) || !qkv || !key_bias || !lse) return CLIMB_EUNSUPPORTED;
  const int CK = as_chunk(S_pad);
  if (CK > AS_MAXKEYS) return CLIMB_EUNSUPPORTED;
  const size_t lds = (size_t)4 * CK * AS_PITCH + (size_t)S_pad * sizeof(float);
  const int nw = g_as_nw_fwd ? g_as_nw_fwd : as_waves(S_pad / 32);
  static size_t lds_set[3] = {0, 0, 0};
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ASF(NW_, IDX_)                                                                                                                       \
  do {                                                                                                                                       \
    if (lds > lds_set[IDX_]) {                                                                                                               \
      hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_split_kernel<NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
      if (e != hipSuccess) return (int)e;                                                                                                    \
      lds_set[IDX_] = lds;                                                                                                                   \
    }                                                                                                                                        \
    hipLaunchKernelGGL((attn_fwd_split_kernel<NW_>), dim3(B * heads), dim3(64 * NW_), lds, (hipStream_t)stream, qkv, key_bias, ctx,          \
                       (bf16_t*)ctx_split, ctx_lo, lse, S_pad, heads, CK, scale);                                                            \
  } while (0)
  if (nw == 4) ASF(4, 0); else if (nw == 6) ASF(6, 1); else ASF(8, 2);
#undef ASF
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ---------------------------------------------------------------------------------------------- backward
// Phase A (dQ): per query block, loop over key blocks with K and V resident:
//     S^T = K Q^T, P^T = exp(. - lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta) * scale, dQ^T += K^T dS^T
// Phase B (dK, dV): per key block, loop over query blocks with Q and dO resident:
//     S = Q K^T, P, dP = dO V^T, dS;  dV^T += dO^T P,  dK^T += Q^T dS
// dynamic LDS: X hi | X lo | Y hi | Y lo (CK rows each) | bias[S_pad] | lse[S_pad] | delta[S_pad]
template <int PHASE, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_split_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias, const float* __restrict__ dctx,
                                                                 const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                                 bf16_t* __restrict__ dqkv_hi, long dqkv_lo, int S_pad, int heads, int CK, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xh = smem;                       // phase A: K chunk; phase B: Q chunk
  unsigned char* Xl = Xh + CK * AS_PITCH;
  unsigned char* Yh = Xl + CK * AS_PITCH;         // phase A: V chunk; phase B: dO chunk
  unsigned char* Yl = Yh + CK * AS_PITCH;
  float* bias_s = reinterpret_cast<float*>(Yl + CK * AS_PITCH);
  float* lse_s = bias_s + S_pad;
  float* delta_s = lse_s + S_pad;
  const int H = heads * AS_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const float* Qg = qkv + (long)b * S_pad * ld + h * AS_D;
  const float* Kg = Qg + H;
  const float* Vg = Qg + 2 * H;
  const float* dOg = dctx + (long)b * S_pad * H + h * AS_D;
  const int NB = S_pad / 32;
  const int nchunks = (S_pad + CK - 1) / CK;
  for (int i = tid; i < S_pad; i += 64 * NW) {
    bias_s[i] = key_bias[(long)b * S_pad + i];
    lse_s[i] = lse[((long)b * heads + h) * S_pad + i];
    delta_s[i] = delta[((long)b * heads + h) * S_pad + i];
  }
  __syncthreads();          // (the per-lane statistics below are read before the first staging barrier)
  const float* O1g = PHASE == 0 ? Qg : Kg;   const long O1ld = ld;
  const float* O2g = PHASE == 0 ? dOg : Vg;  const long O2ld = PHASE == 0 ? H : ld;
  const float* I1g = PHASE == 0 ? Kg : Qg;   const long I1ld = ld;
  const float* I2g = PHASE == 0 ? Vg : dOg;  const long I2ld = PHASE == 0 ? ld : H;

  for (int round = 0; round * NW < NB; ++round) {
    const int ob = round * NW + wid;        // outer block (queries in phase A, keys in phase B)
    const bool active = ob < NB;
    bf16x8 r1h[4], r1l[4], r2h[4], r2l[4];
    as_load_outer(O1g + (long)(active ? ob : 0) * 32 * O1ld, O1ld, l31, half, scale, r1h, r1l);      // (Q or K) / sqrt(d): only the scores use them
    as_load_outer(O2g + (long)(active ? ob : 0) * 32 * O2ld, O2ld, l31, half, 1.0f, r2h, r2l);
    f32x16 acc[2][2];   // phase A: acc[0] = dQ^T (2 d-blocks); phase B: acc[0] = dK^T, acc[1] = dV^T
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int d = 0; d < 2; ++d) as_zero(acc[a][d]);
    const int my = ob * 32 + l31;  // this lane's outer index (query in A, key in B)
    const float my_lse = PHASE == 0 ? lse_s[active ? my : 0] : 0.f;
    const float my_delta = PHASE == 0 ? delta_s[active ? my : 0] : 0.f;
    const float my_bias = PHASE == 1 ? bias_s[active ? my : 0] : 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
      const int i0 = ch * CK, ni = min(CK, S_pad - i0);
      if (nchunks > 1 || round == 0) {
        __syncthreads();
        as_stage2<4>(Xh, Xl, I1g + (long)i0 * I1ld, I1ld, Yh, Yl, I2g + (long)i0 * I2ld, I2ld, ni, tid, 64 * NW);
        __syncthreads();
      }
      if (!active) continue;
      for (int ib = 0; ib < ni / 32; ++ib) {
        f32x16 s, dp;
        as_zero(s);
        as_zero(dp);
        // rows = inner block (keys in A, queries in B); cols = outer block
        as_prod_rm(s, Xh, Xl, ib * 32, l31, half, r1h, r1l);
        as_prod_rm(dp, Yh, Yl, ib * 32, l31, half, r2h, r2l);
        f32x16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int in = i0 + ib * 32 + as_drow(r, half);
          if (PHASE == 0) {   // in = key, lane = query
            p[r] = __expf(s[r] + bias_s[in] - my_lse);
            ds[r] = p[r] * (dp[r] - my_delta) * scale;
          } else {            // in = query, lane = key
            p[r] = __expf(s[r] + my_bias - lse_s[in]);
            ds[r] = p[r] * (dp[r] - delta_s[in]) * scale;
          }
        }
        bf16x8 dsh[2], dsl[2];
        as_split16(ds, dsh, dsl);
        as_prod_t(acc[0], Xh, Xl, ib * 32, lane, dsh, dsl);          // A: dQ^T += K^T dS^T;  B: dK^T += Q^T dS
        if (PHASE == 1) {                                            // dV^T += dO^T P
          bf16x8 ph[2], pl[2];
          as_split16(p, ph, pl);
          as_prod_t(acc[1], Yh, Yl, ib * 32, lane, ph, pl);
        }
      }
    }
    if (active) {
      const long row = (long)b * S_pad + my;
#pragma unroll
      for (int a = 0; a < (PHASE == 0 ? 1 : 2); ++a) {
        const long off = row * ld + h * AS_D + (PHASE == 0 ? 0 : (a == 0 ? H : 2 * H));
        as_store_row(acc[a], dqkv ? dqkv + off : nullptr, dqkv_hi ? dqkv_hi + off : nullptr, dqkv_lo, half);
      }
    }
  }
}

// d(qkv) as fp32 [B*S_pad, 3H] (may be NULL) and / or split planes (dqkv_split = the hi plane, the lo plane dqkv_lo elements behind; may be NULL)
extern "C" int climb_attn_bwd_split(const float* qkv, const float* key_bias, const float* dctx, const float* lse, const float* delta, float* dqkv, void* dqkv_split,
                                    long dqkv_lo, int B, int S_pad, int heads, int head_dim, void* stream) {
  if (head_dim != AS_D || S_pad % 32 || S_pad <= 0 || (!dqkv && !dqkv_split) || !qkv || !key_bias || !dctx || !lse || !delta) return CLIMB_EUNSUPPORTED;
  const int CK = as_chunk(S_pad);
  if (CK > AS_MAXKEYS) return CLIMB_EUNSUPPORTED;
  const size_t lds = (size_t)4 * CK * AS_PITCH + (size_t)3 * S_pad * sizeof(float);
  const int nw = g_as_nw_bwd ? g_as_nw_bwd : as_waves(S_pad / 32);
  static size_t lds_set[3] = {0, 0, 0};
  const float scale = 1.0f / sqrtf((float)head_dim);
#define ASB(NW_, IDX_)                                                                                                                            \
  do {                                                                                                                                            \
    if (lds > lds_set[IDX_]) {                                                                                                                    \
      hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_split_kernel<0, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
      if (e != hipSuccess) return (int)e;                                                                                                         \
      e = hipFuncSetAttribute((const void*)attn_bwd_split_kernel<1, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      if (e != hipSuccess) return (int)e;                                                                                                         \
      lds_set[IDX_] = lds;                                                                                                                        \
    }                                                                                                                                             \
    hipLaunchKernelGGL((attn_bwd_split_kernel<0, NW_>), dim3(B * heads), dim3(64 * NW_), lds, (hipStream_t)stream, qkv, key_bias, dctx, lse, delta, \
                       dqkv, (bf16_t*)dqkv_split, dqkv_lo, S_pad, heads, CK, scale);                                                              \
    LAUNCH_CHECK();                                                                                                                               \
    hipLaunchKernelGGL((attn_bwd_split_kernel<1, NW_>), dim3(B * heads), dim3(64 * NW_), lds, (hipStream_t)stream, qkv, key_bias, dctx, lse, delta, \
                       dqkv, (bf16_t*)dqkv_split, dqkv_lo, S_pad, heads, CK, scale);                                                              \
  } while (0)
  if (nw == 4) ASB(4, 0); else if (nw == 6) ASB(6, 1); else ASB(8, 2);
#undef ASB
  LAUNCH_CHECK();
  return CLIMB_OK;
}
