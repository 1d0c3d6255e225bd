// fp32 multi-head self-attention over the concatenated [text | image] tokens (HF modeling_vilt.py:322-351), forward
// and backward, on the exact-f32 matrix-core instruction v_mfma_f32_32x32x2_f32.  Reference-precision path.
//
// One workgroup (4 waves) per (batch, head).  All score products are computed "swapped" (S^T = K Q^T, keys in MFMA
// rows, queries in MFMA columns) so that a lane owns ONE query column: softmax statistics are per-lane scalars plus a
// single xor-32 shuffle, and the probability registers are directly the B operand of the next product
// (O^T = V^T P^T) with no cross-lane movement.  The score matrix never touches HBM.
//
//   qkv : [B*S_pad, 3H] fp32, row = token, columns [q | k | v], head h at columns h*64..h*64+63 of each third
//   bias: [B, S_pad] additive key bias (0 keep / -3e38 masked or padding)
//   ctx : [B*S_pad, H],  lse: [B, heads, S_pad] (log-sum-exp of the scaled, biased scores; saved for backward)
#include "common.h"

#define AF_D 64
#define AF_LDK 65          // padded row stride (floats): conflict-free "row = lane" reads
#define AF_MAXKEYS 192     // keys resident in LDS per chunk

__device__ __forceinline__ int drow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// copy rows [r0, r0+nrows) x 64 floats of a strided global matrix into LDS with row stride `lds_ld`
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, int lds_ld, const float* __restrict__ src, long ld, int nrows, int tid,
                                           int nthreads) {
  for (int e = tid; e < nrows * 16; e += nthreads) {
    int r = e >> 4, c = (e & 15) * 4;
    float4 v = ld4(src + (long)r * ld + c);
    float* d = dst + r * lds_ld + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// ---------------------------------------------------------------------------------------------- forward
// dynamic LDS: Ks[CK][65] | Vs[CK][64] | Ws[4][32][65] | bias_s[S_pad]
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias,
                                                           float* __restrict__ ctx, float* __restrict__ lse_out, int S_pad, int heads, int CK,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = Ks + CK * AF_LDK;
  float* Ws = Vs + CK * AF_D;
  float* bias_s = Ws + 4 * 32 * AF_LDK;
  const int H = heads * AF_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const float* Qg = qkv + (long)b * S_pad * ld + h * AF_D;
  const float* Kg = Qg + H;
  const float* Vg = Qg + 2 * H;
  const int NB = S_pad / 32;
  const int nchunks = (S_pad + CK - 1) / CK;
  for (int i = tid; i < S_pad; i += 256) bias_s[i] = key_bias[(long)b * S_pad + i];
  float* Wq = Ws + wid * 32 * AF_LDK;

  for (int round = 0; round * 4 < NB; ++round) {
    const int qb = round * 4 + wid;
    const bool active = qb < NB;
    __syncthreads();
    if (active) stage_rows(Wq, AF_LDK, Qg + (long)qb * 32 * ld, ld, 32, lane, 64);
    __syncthreads();
    float qreg[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) qreg[j] = Wq[l31 * AF_LDK + 2 * j + half];
    // pass 1: online max / sum over all keys -> lse
    float m_run = -3.0e38f, l_run = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int k0 = ch * CK, nk = min(CK, S_pad - k0);
      if (nchunks > 1 || round == 0) {
        __syncthreads();
        stage_rows(Ks, AF_LDK, Kg + (long)k0 * ld, ld, nk, tid, 256);
        if (nchunks == 1) stage_rows(Vs, AF_D, Vg + (long)k0 * ld, ld, nk, tid, 256);
        __syncthreads();
      }
      if (active)
        for (int kb = 0; kb < nk / 32; ++kb) {
          f32x16 s;
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
          for (int j = 0; j < 32; ++j)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(kb * 32 + l31) * AF_LDK + 2 * j + half], qreg[j], s, 0, 0, 0);
          float mx = -3.0e38f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s[r] = s[r] * scale + bias_s[k0 + kb * 32 + drow(r, half)];
            mx = fmaxf(mx, s[r]);
          }
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float m_new = fmaxf(m_run, mx);
          float sum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += __expf(s[r] - m_new);
          sum += __shfl_xor(sum, 32, 64);
          l_run = l_run * __expf(m_run - m_new) + sum;
          m_run = m_new;
        }
    }
    const float lse = m_run + __logf(l_run);
    // pass 2: P = exp(s - lse) (already normalised), O^T = V^T P^T
    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      const int k0 = ch * CK, nk = min(CK, S_pad - k0);
      if (nchunks > 1) {
        __syncthreads();
        stage_rows(Ks, AF_LDK, Kg + (long)k0 * ld, ld, nk, tid, 256);
        stage_rows(Vs, AF_D, Vg + (long)k0 * ld, ld, nk, tid, 256);
        __syncthreads();
      }
      if (active)
        for (int kb = 0; kb < nk / 32; ++kb) {
          f32x16 s;
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
          for (int j = 0; j < 32; ++j)
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(kb * 32 + l31) * AF_LDK + 2 * j + half], qreg[j], s, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = __expf(s[r] * scale + bias_s[k0 + kb * 32 + drow(r, half)] - lse);
          // reduction index of this MFMA step = key (kb*32 + drow(r, half)); B operand = the lane's own P register
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + (kb * 32 + drow(r, half)) * AF_D;
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[l31], s[r], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 + l31], s[r], o[1], 0, 0, 0);
          }
        }
    }
    if (active) {
      // O^T layout: col = query (l31), row = d
      float* cr = ctx + ((long)b * S_pad + qb * 32 + l31) * H + h * AF_D;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          st4(cr + d * 32 + 8 * g + 4 * half, make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]));
      if (half == 0) lse_out[((long)b * heads + h) * S_pad + qb * 32 + l31] = lse;
    }
  }
}

extern "C" int climb_attn_fwd_f32(const float* qkv, const float* key_bias, float* ctx, float* lse, int B, int S_pad, int heads, int head_dim,
                                  void* stream) {
  if (head_dim != AF_D || S_pad % 32 || S_pad <= 0) return CLIMB_EUNSUPPORTED;
  int CK = S_pad <= AF_MAXKEYS ? S_pad : ((S_pad / 32 + 1) / 2) * 32;
  if (CK > AF_MAXKEYS) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)(CK * AF_LDK + CK * AF_D + 4 * 32 * AF_LDK + S_pad) * sizeof(float);
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, key_bias, ctx, lse, S_pad, heads, CK,
                     1.0f / sqrtf((float)head_dim));
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ---------------------------------------------------------------------------------------------- delta
// delta[b,h,q] = sum_d dO[q,d] * O[q,d]   (softmax backward row term); one 16-lane group per (row, head)
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ dctx, const T* __restrict__ ctx, float* __restrict__ delta, int B, int S_pad, int heads) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long grp = gid >> 4;
  const int sub = (int)(gid & 15);
  if (grp >= (long)B * S_pad * heads) return;
  const long row = grp / heads;
  const int h = (int)(grp % heads);
  const long off = row * heads * AF_D + h * AF_D + sub * 4;
  float4 a = ld4(dctx + off), o = ld4(ctx + off);
  float s = a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
#pragma unroll
  for (int m = 8; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
  if (sub == 0) {
    const long b = row / S_pad, q = row % S_pad;
    delta[(b * heads + h) * S_pad + q] = s;
  }
}
extern "C" int climb_attn_delta(const void* dctx, const void* ctx, int dtype, float* delta, int B, int S_pad, int heads, void* stream) {
  long n = (long)B * S_pad * heads * 16;
  dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  if (dtype == CLIMB_DT_F32) hipLaunchKernelGGL((attn_delta_kernel<float>), grid, blk, 0, (hipStream_t)stream, (const float*)dctx, (const float*)ctx, delta, B, S_pad, heads);
  else if (dtype == CLIMB_DT_BF16) hipLaunchKernelGGL((attn_delta_kernel<bf16_t>), grid, blk, 0, (hipStream_t)stream, (const bf16_t*)dctx, (const bf16_t*)ctx, delta, B, S_pad, heads);
  else return CLIMB_EINVAL;
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ---------------------------------------------------------------------------------------------- backward
// Phase A (dQ): per query block, loop over key blocks with K and V resident:
//     S^T = K Q^T, P^T = exp(.-lse), dP^T = V dO^T, dS^T = P^T (dP^T - delta) * scale, dQ^T += K^T dS^T
// Phase B (dK, dV): per key block, loop over query blocks with Q and dO resident:
//     S = Q K^T, P, dP = dO V^T, dS;  dV^T += dO^T P,  dK^T += Q^T dS
// Each phase keeps the register operand of the "outer" block in 2x32 VGPRs and reads the "inner" side from LDS.
// dynamic LDS: Xs[CK][65] | Ys[CK][65] | Ws[4][32][65] | v1[S_pad] | v2[S_pad] | v3[S_pad]
template <int PHASE>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ key_bias,
                                                           const float* __restrict__ dctx, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, float* __restrict__ dqkv, int S_pad, int heads, int CK,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // phase A: K chunk; phase B: Q chunk
  float* Ys = Xs + CK * AF_LDK;           // phase A: V chunk; phase B: dO chunk
  float* Ws = Ys + CK * AF_LDK;
  float* bias_s = Ws + 4 * 32 * AF_LDK;
  float* lse_s = bias_s + S_pad;
  float* delta_s = lse_s + S_pad;
  const int H = heads * AF_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const float* Qg = qkv + (long)b * S_pad * ld + h * AF_D;
  const float* Kg = Qg + H;
  const float* Vg = Qg + 2 * H;
  const float* dOg = dctx + (long)b * S_pad * H + h * AF_D;
  const int NB = S_pad / 32;
  const int nchunks = (S_pad + CK - 1) / CK;
  for (int i = tid; i < S_pad; i += 256) {
    bias_s[i] = key_bias[(long)b * S_pad + i];
    lse_s[i] = lse[((long)b * heads + h) * S_pad + i];
    delta_s[i] = delta[((long)b * heads + h) * S_pad + i];
  }
  float* Ww = Ws + wid * 32 * AF_LDK;
  // outer-block operand sources / inner-chunk sources
  const float* O1g = PHASE == 0 ? Qg : Kg;   const long O1ld = ld;
  const float* O2g = PHASE == 0 ? dOg : Vg;  const long O2ld = PHASE == 0 ? H : ld;
  const float* I1g = PHASE == 0 ? Kg : Qg;   const long I1ld = ld;
  const float* I2g = PHASE == 0 ? Vg : dOg;  const long I2ld = PHASE == 0 ? ld : H;

  for (int round = 0; round * 4 < NB; ++round) {
    const int ob = round * 4 + wid;        // outer block (queries in phase A, keys in phase B)
    const bool active = ob < NB;
    float r1[32], r2[32];
    __syncthreads();
    if (active) stage_rows(Ww, AF_LDK, O1g + (long)ob * 32 * O1ld, O1ld, 32, lane, 64);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; ++j) r1[j] = Ww[l31 * AF_LDK + 2 * j + half];
    __syncthreads();
    if (active) stage_rows(Ww, AF_LDK, O2g + (long)ob * 32 * O2ld, O2ld, 32, lane, 64);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; ++j) r2[j] = Ww[l31 * AF_LDK + 2 * j + half];

    f32x16 acc[2][2];   // phase A: acc[0] = dQ^T (2 d-blocks); phase B: acc[0] = dK^T, acc[1] = dV^T
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][d][r] = 0.f;
    const int my = ob * 32 + l31;  // this lane's outer index (query in A, key in B)
    const float my_lse = PHASE == 0 ? lse_s[active ? my : 0] : 0.f;
    const float my_delta = PHASE == 0 ? delta_s[active ? my : 0] : 0.f;
    const float my_bias = PHASE == 1 ? bias_s[active ? my : 0] : 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
      const int i0 = ch * CK, ni = min(CK, S_pad - i0);
      if (nchunks > 1 || round == 0) {
        __syncthreads();
        stage_rows(Xs, AF_LDK, I1g + (long)i0 * I1ld, I1ld, ni, tid, 256);
        stage_rows(Ys, AF_LDK, I2g + (long)i0 * I2ld, I2ld, ni, tid, 256);
        __syncthreads();
      }
      if (!active) continue;
      for (int ib = 0; ib < ni / 32; ++ib) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
        // rows = inner block (keys in A, queries in B); cols = outer block
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          const int off = (ib * 32 + l31) * AF_LDK + 2 * j + half;
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[off], r1[j], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Ys[off], r2[j], dp, 0, 0, 0);
        }
        f32x16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int in = i0 + ib * 32 + drow(r, half);
          if (PHASE == 0) {   // in = key, lane = query
            p[r] = __expf(s[r] * scale + bias_s[in] - my_lse);
            ds[r] = p[r] * (dp[r] - my_delta) * scale;
          } else {            // in = query, lane = key
            p[r] = __expf(s[r] * scale + my_bias - lse_s[in]);
            ds[r] = p[r] * (dp[r] - delta_s[in]) * scale;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (ib * 32 + drow(r, half)) * AF_LDK;
          if (PHASE == 0) {   // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[row + l31], ds[r], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[row + 32 + l31], ds[r], acc[0][1], 0, 0, 0);
          } else {            // dK^T[d][key] += Q^T[d][q] dS[q][key];  dV^T[d][key] += dO^T[d][q] P[q][key]
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[row + l31], ds[r], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Xs[row + 32 + l31], ds[r], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ys[row + l31], p[r], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ys[row + 32 + l31], p[r], acc[1][1], 0, 0, 0);
          }
        }
      }
    }
    if (active) {
      float* orow = dqkv + ((long)b * S_pad + my) * ld + h * AF_D;
#pragma unroll
      for (int a = 0; a < (PHASE == 0 ? 1 : 2); ++a) {
        float* dst = orow + (PHASE == 0 ? 0 : (a == 0 ? H : 2 * H));
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            st4(dst + d * 32 + 8 * g + 4 * half, make_float4(acc[a][d][4 * g], acc[a][d][4 * g + 1], acc[a][d][4 * g + 2], acc[a][d][4 * g + 3]));
      }
    }
  }
}

extern "C" int climb_attn_bwd_f32(const float* qkv, const float* key_bias, const float* dctx, const float* lse, const float* delta, float* dqkv,
                                  int B, int S_pad, int heads, int head_dim, void* stream) {
  if (head_dim != AF_D || S_pad % 32 || S_pad <= 0) return CLIMB_EUNSUPPORTED;
  int CK = S_pad <= AF_MAXKEYS ? S_pad : ((S_pad / 32 + 1) / 2) * 32;
  if (CK > AF_MAXKEYS) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)(2 * CK * AF_LDK + 4 * 32 * AF_LDK + 3 * S_pad) * sizeof(float);
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_f32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_f32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  const float scale = 1.0f / sqrtf((float)head_dim);
  hipLaunchKernelGGL((attn_bwd_f32_kernel<0>), dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, key_bias, dctx, lse, delta, dqkv, S_pad,
                     heads, CK, scale);
  LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_f32_kernel<1>), dim3(B * heads), dim3(256), lds, (hipStream_t)stream, qkv, key_bias, dctx, lse, delta, dqkv, S_pad,
                     heads, CK, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ---------------------------------------------------------------------------------------------- forward with dropout on the probabilities
// The frozen BERT of ViLT-BERT while the learner is in train mode (REF/modeling/viltbert.py:115-120 never calls bert.eval(): its
// attention_probs_dropout is live; HFB eager_attention_forward: softmax -> dropout -> P V).  Text only: S_pad <= 64 keys, head_dim 64,
// forward only (BERT is frozen), 2 560 rows at 64 sequences -- a plain VALU kernel: one workgroup per (batch, head), K and V rows in LDS
// as fp32, one wave per query (lane = key for the scores and the softmax, lane = output dimension for P V).
// keep[b, h, q, k] (uint8, T x T per head, T <= S_pad the mask's own row length) is the keep-mask, drop_scale = 1 / (1 - p).
template <typename T16>
__global__ __launch_bounds__(256) void attn_fwd_dropout_kernel(const T16* __restrict__ qkv, const float* __restrict__ key_bias, const unsigned char* __restrict__ keep,
                                                               T16* __restrict__ ctx, int S_pad, int heads, int T, float drop_scale) {
  __shared__ float Ks[64][65], Vs[64][64], Qs[4][64], Ps[4][64];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, H = heads * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long row0 = (long)b * S_pad;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, d = i & 63;
    float kv = 0.f, vv = 0.f;
    if (r < S_pad) {
      kv = Act<T16>::ld(qkv + (row0 + r) * 3 * H + H + h * 64 + d);
      vv = Act<T16>::ld(qkv + (row0 + r) * 3 * H + 2 * H + h * 64 + d);
    }
    Ks[r][d] = kv;
    Vs[r][d] = vv;
  }
  __syncthreads();
  const float kb = lane < S_pad ? key_bias[(long)b * S_pad + lane] : -INFINITY;
  for (int q = w; q < S_pad; q += 4) {
    Qs[w][lane] = Act<T16>::ld(qkv + (row0 + q) * 3 * H + h * 64 + lane);
    __builtin_amdgcn_wave_barrier();
    float s = 0.f;
#pragma unroll 16
    for (int d = 0; d < 64; ++d) s += Qs[w][d] * Ks[lane][d];
    s = s * 0.125f + kb;
    float m = s;
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float p = (lane < S_pad && m > -INFINITY) ? __expf(s - m) : 0.f;
    float sum = p;
    for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
    p = sum > 0.f ? p / sum : 0.f;
    if (q < T && lane < T) p = keep[(((long)b * heads + h) * T + q) * T + lane] ? p * drop_scale : 0.f;      // rows / keys beyond T: padding, masked anyway
    Ps[w][lane] = p;
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
#pragma unroll 16
    for (int k = 0; k < 64; ++k) o += Ps[w][k] * Vs[k][lane];
    Act<T16>::st(ctx + (row0 + q) * H + h * 64 + lane, o);
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int climb_attn_fwd_dropout(const void* qkv, const float* key_bias, const void* keep, void* ctx, int dtype, int B, int S_pad, int heads,
                                      int head_dim, int T, float drop_scale, void* stream) {
  if (head_dim != 64 || S_pad <= 0 || S_pad > 64 || T <= 0 || T > S_pad || !keep) return CLIMB_EUNSUPPORTED;
  if (dtype == CLIMB_DT_F32)
    hipLaunchKernelGGL((attn_fwd_dropout_kernel<float>), dim3(B * heads), dim3(256), 0, (hipStream_t)stream, (const float*)qkv, key_bias,
                       (const unsigned char*)keep, (float*)ctx, S_pad, heads, T, drop_scale);
  else if (dtype == CLIMB_DT_BF16)
    hipLaunchKernelGGL((attn_fwd_dropout_kernel<bf16_t>), dim3(B * heads), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                       (const unsigned char*)keep, (bf16_t*)ctx, S_pad, heads, T, drop_scale);
  else return CLIMB_EINVAL;
  LAUNCH_CHECK();
  return CLIMB_OK;
}
