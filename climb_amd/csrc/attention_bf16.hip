// bf16 multi-head self-attention for the concatenated [text | image] tokens (HF modeling_vilt.py:322-351), forward and
// backward, on v_mfma_f32_32x32x16_bf16.  Throughput-mode twin of attention_f32.hip; same data layout and the same
// "swapped" formulation: every product has the reduction index in MFMA rows of the previous result, so softmax
// statistics are per-lane scalars (+ one xor-32 shuffle) and P / dS feed the next MFMA straight from registers.
//
// One workgroup per (batch, head); S <= 288 keys means a whole head's K and V (S_pad x 64 bf16 = 24.6 KB each at
// S_pad = 192) are LDS-resident: single pass over HBM, the S x S scores never leave the CU.
//
// LDS image of every [rows][64] operand: 128-B rows, 16-B chunks XOR-swizzled by swz(row).  It is read two ways:
//   * "row" fragments (8 consecutive d of one row)      -> ds_read_b128, conflict-free
//   * "column" fragments (4 consecutive rows at one d)  -> ds_read_b64_tr_b16 (LDS transpose read), conflict-free
// so K serves both S^T = K Q^T and dQ^T = K^T dS^T from one image, likewise V, Q, dO.
#include "common.h"

#define AB_D 64

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int aswz(int row) {
  int y = (row >> 1) & 7;
  return ((y & 1) << 2) | (y >> 1);
}
__device__ __forceinline__ bf16x8 to_bf16x8(u32x4 v) {
  union { u32x4 u; bf16x8 b; } c;
  c.u = v;
  return c.b;
}
__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// stage rows [0, nrows) x 64 bf16 of a strided global matrix into the swizzled LDS image with LDS-DMA
// (global_load_lds_dwordx4: asynchronous, no VGPRs, no ds_write pass; the destination is linear per wave-instruction, so
// the swizzle goes on the per-lane SOURCE address).  nrows is a multiple of 8; the caller's __syncthreads() drains vmcnt.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
__device__ __forceinline__ void stage_image(unsigned char* __restrict__ S, const bf16_t* __restrict__ src, long ld, int nrows, int tid, int nthreads) {
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  for (int j = wave; j < nrows / 8; j += nwaves) {
    const int slot = j * 64 + lane, row = slot >> 3, c = (slot & 7) ^ aswz(row);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(src + (long)row * ld + c * 8), (lds_void_t*)(S + j * 1024), 16, 0, 0);
  }
}
// row fragment: image row (row0 + lane&31), d = 16*ks + 8*(lane>>5) .. +7
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* __restrict__ S, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31), c = 2 * ks + (lane >> 5);
  return to_bf16x8(*reinterpret_cast<const u32x4*>(S + row * 128 + ((c ^ aswz(row)) << 4)));
}
// column fragment for output index d = d0 + (lane&31): rows r0 + 4*half + {0..3} and r0 + 8 + 4*half + {0..3}
// (exactly the rows whose P / dS values a lane holds in accumulator registers 8s..8s+7 of a 32x32 block)
__device__ __forceinline__ bf16x8 col_frag(const unsigned char* __restrict__ S, int r0, int d0, int lane) {
  const int g16 = lane >> 4, i = lane & 15;
  const int col = d0 + (g16 & 1) * 16 + 4 * (i & 3);      // first of this lane's 4 contiguous d (8 bytes inside one 16-B chunk)
  const int c = col >> 3, within = (col & 7) * 2;
  const int rowa = r0 + 4 * (g16 >> 1) + (i >> 2), rowb = rowa + 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + rowa * 128 + ((c ^ aswz(rowa)) << 4) + within));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + rowb * 128 + ((c ^ aswz(rowb)) << 4) + within));
  union { s16x4 h[2]; bf16x8 b; } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.b;
}
// 8 fp32 accumulator registers (8s..8s+7 of a block) -> bf16x8 MFMA operand
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
  union { unsigned int u[4]; bf16x8 b; } c;
#pragma unroll
  for (int j = 0; j < 4; ++j) c.u[j] = pack_bf16x2(v[8 * s + 2 * j], v[8 * s + 2 * j + 1]);
  return c.b;
}
// this lane's B operand for a 32-row register block: row (row0 + lane&31) of a global [.,64] bf16 matrix, 4 k-steps
__device__ __forceinline__ void load_rows(bf16x8 (&f)[4], const bf16_t* __restrict__ src, long ld, int row0, int lane) {
  const bf16_t* p = src + (long)(row0 + (lane & 31)) * ld + (lane >> 5) * 8;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = to_bf16x8(*reinterpret_cast<const u32x4*>(p + 16 * ks));
}
// O^T-style accumulator (col = row index of the output matrix, rows = d) -> 4 consecutive d per register group
__device__ __forceinline__ void store_acc(bf16_t* __restrict__ dst, const f32x16 (&o)[2], int half, float scale) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      st4(dst + d * 32 + 8 * g + 4 * half,
          make_float4(o[d][4 * g] * scale, o[d][4 * g + 1] * scale, o[d][4 * g + 2] * scale, o[d][4 * g + 3] * scale));
}

// ------------------------------------------------------------------------------------------------------ forward
// dynamic LDS: Ks[S_pad][128 B] | Vs[S_pad][128 B] | bias_s[S_pad] f32
__global__ __launch_bounds__(576) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                            bf16_t* __restrict__ ctx, float* __restrict__ lse_out, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = Ks + S_pad * 128;
  float* bias_s = reinterpret_cast<float*>(Vs + S_pad * 128);
  const int H = heads * AB_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int nwaves = blockDim.x >> 6;
  const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Qg + 2 * H;
  stage_image(Ks, Kg, ld, S_pad, tid, blockDim.x);
  stage_image(Vs, Vg, ld, S_pad, tid, blockDim.x);
  for (int i = tid; i < S_pad; i += blockDim.x) bias_s[i] = key_bias[(long)b * S_pad + i];
  __syncthreads();
  const int NB = S_pad / 32;
  for (int qb = wid; qb < NB; qb += nwaves) {
    bf16x8 qf[4];
    load_rows(qf, Qg, ld, qb * 32, lane);
    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;
    for (int kb = 0; kb < NB; ++kb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, kb * 32, ks, lane), qf[ks], s, 0, 0, 0);
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = s[r] * scale + bias_s[kb * 32 + crow(r, half)];
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
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      // O^T[d][q] += V^T[d][key] P^T[key][q]; MFMA k-slot (half, j) <-> key kb*32 + 16*st + 8*(j>>2) + 4*half + (j&3)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 pf = pack8(s, st);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Vs, kb * 32 + 16 * st, 0, lane), pf, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Vs, kb * 32 + 16 * st, 32, lane), pf, o[1], 0, 0, 0);
      }
    }
    store_acc(ctx + ((long)b * S_pad + qb * 32 + l31) * H + h * AB_D, o, half, 1.0f / l_run);
    if (half == 0) lse_out[((long)b * heads + h) * S_pad + qb * 32 + l31] = m_run + __logf(l_run);
  }
}


// one wave per 32-row block: the per-workgroup latency chain (stage K/V -> load Q -> scores -> softmax -> P.V -> store) is
// walked once instead of NB/3 times, and a CU holds 3 workgroups x NB waves
static int pick_waves(int NB) { return NB <= 9 ? NB : 8; }

extern "C" int climb_attn_fwd_bf16(const void* qkv, const float* key_bias, void* ctx, float* lse, int B, int S_pad, int heads, int head_dim,
                                   void* stream) {
  if (head_dim != AB_D || S_pad % 32 || S_pad <= 0 || S_pad > 512) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)S_pad * 256 + (size_t)S_pad * 4;
  static size_t lds_set = 0;          // raise the dynamic-LDS cap once per size (not a stream operation: keep it out of graph capture)
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3(B * heads), dim3(64 * pick_waves(S_pad / 32)), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                     (bf16_t*)ctx, lse, S_pad, heads, 1.0f / sqrtf((float)head_dim));
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ------------------------------------------------------------------------------------------------------ backward
// PHASE 0 (dQ):      outer = query block (Q, dO rows in registers), inner = keys (K, V images in LDS)
//     S^T = K Q^T, dP^T = V dO^T, P^T = exp(S^T*scale + bias - lse), dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T
// PHASE 1 (dK, dV):  outer = key block (K, V rows in registers), inner = queries (Q, dO images in LDS)
//     S = Q K^T, dP = dO V^T, P, dS;  dV^T += dO^T P,  dK^T += Q^T dS
// delta[q] = sum_d dO[q][d] O[q][d] (the softmax-backward row term) is computed by PHASE 0 itself, whose waves already hold
// their dO rows in registers, and written out for PHASE 1, which needs it for every query.
// dynamic LDS: Xs[S_pad][128 B] | Ys[S_pad][128 B] | bias_s | lse_s | delta_s
// launch bound 576 (= 9 waves, 3 per SIMD) caps the kernel at 168 VGPRs: three workgroups stay resident per CU, which is worth
// more than the 17 spilled dwords of phase 1 (measured: 63 us vs 80 us per layer with a 256-thread bound)
template <int PHASE>
__global__ __launch_bounds__(576) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ key_bias,
                                                            const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ ctx,
                                                            const float* __restrict__ lse, float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int S_pad, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Xs = smem;
  unsigned char* Ys = Xs + S_pad * 128;
  float* bias_s = reinterpret_cast<float*>(Ys + S_pad * 128);
  float* lse_s = bias_s + S_pad;
  float* delta_s = lse_s + S_pad;
  const int H = heads * AB_D, ld = 3 * H;
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int nwaves = blockDim.x >> 6;
  const bf16_t* Qg = qkv + (long)b * S_pad * ld + h * AB_D;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Qg + 2 * H;
  const bf16_t* dOg = dctx + (long)b * S_pad * H + h * AB_D;
  if (PHASE == 0) {
    stage_image(Xs, Kg, ld, S_pad, tid, blockDim.x);
    stage_image(Ys, Vg, ld, S_pad, tid, blockDim.x);
  } else {
    stage_image(Xs, Qg, ld, S_pad, tid, blockDim.x);
    stage_image(Ys, dOg, H, S_pad, tid, blockDim.x);
  }
  for (int i = tid; i < S_pad; i += blockDim.x) {
    bias_s[i] = key_bias[(long)b * S_pad + i];
    lse_s[i] = lse[((long)b * heads + h) * S_pad + i];
    if (PHASE == 1) delta_s[i] = delta[((long)b * heads + h) * S_pad + i];
  }
  __syncthreads();
  const int NB = S_pad / 32;
  for (int ob = wid; ob < NB; ob += nwaves) {
    bf16x8 f1[4], f2[4];
    if (PHASE == 0) { load_rows(f1, Qg, ld, ob * 32, lane); load_rows(f2, dOg, H, ob * 32, lane); }
    else            { load_rows(f1, Kg, ld, ob * 32, lane); load_rows(f2, Vg, ld, ob * 32, lane); }
    f32x16 acc1[2], acc2[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[d][r] = acc2[d][r] = 0.f;
    const int my = ob * 32 + l31;
    float my_delta = 0.f;
    if (PHASE == 0) {
      bf16x8 fo[4];
      load_rows(fo, ctx + (long)b * S_pad * H + h * AB_D, H, ob * 32, lane);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) my_delta += (float)fo[ks][e] * (float)f2[ks][e];
      my_delta += __shfl_xor(my_delta, 32, 64);
      if (half == 0) delta[((long)b * heads + h) * S_pad + my] = my_delta;
    }
    const float my_lse = lse_s[my], my_bias = bias_s[my];
    for (int ib = 0; ib < NB; ++ib) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Xs, ib * 32, ks, lane), f1[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ys, ib * 32, ks, lane), f2[ks], dp, 0, 0, 0);
      }
      f32x16 p, ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int in = ib * 32 + crow(r, half);
        if (PHASE == 0) {   // in = key, lane = query
          p[r] = __expf(s[r] * scale + bias_s[in] - my_lse);
          ds[r] = p[r] * (dp[r] - my_delta) * scale;
        } else {            // in = query, lane = key
          p[r] = __expf(s[r] * scale + my_bias - lse_s[in]);
          ds[r] = p[r] * (dp[r] - delta_s[in]) * scale;
        }
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8 dsf = pack8(ds, st);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Xs, ib * 32 + 16 * st, 0, lane), dsf, acc1[0], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Xs, ib * 32 + 16 * st, 32, lane), dsf, acc1[1], 0, 0, 0);
        if (PHASE == 1) {
          const bf16x8 pf = pack8(p, st);
          acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Ys, ib * 32 + 16 * st, 0, lane), pf, acc2[0], 0, 0, 0);
          acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(col_frag(Ys, ib * 32 + 16 * st, 32, lane), pf, acc2[1], 0, 0, 0);
        }
      }
    }
    bf16_t* orow = dqkv + ((long)b * S_pad + my) * ld + h * AB_D;
    if (PHASE == 0) store_acc(orow, acc1, half, 1.0f);
    else {
      store_acc(orow + H, acc1, half, 1.0f);
      store_acc(orow + 2 * H, acc2, half, 1.0f);
    }
  }
}

extern "C" int climb_attn_bwd_bf16(const void* qkv, const float* key_bias, const void* dctx, const void* ctx, const float* lse, float* delta,
                                   void* dqkv, int B, int S_pad, int heads, int head_dim, void* stream) {
  if (head_dim != AB_D || S_pad % 32 || S_pad <= 0 || S_pad > 512) return CLIMB_EUNSUPPORTED;
  size_t lds = (size_t)S_pad * 256 + (size_t)S_pad * 12;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)attn_bwd_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    lds_set = lds;
  }
  const float scale = 1.0f / sqrtf((float)head_dim);
  const int nthreads = 64 * ((S_pad / 32) % 3 == 0 ? 3 : 4);     // measured at S_pad = 192: 3-4 waves 63 us, 6 waves 78 us (register pressure)
  hipLaunchKernelGGL((attn_bwd_bf16_kernel<0>), dim3(B * heads), dim3(nthreads), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                     (const bf16_t*)dctx, (const bf16_t*)ctx, lse, delta, (bf16_t*)dqkv, S_pad, heads, scale);
  LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_bf16_kernel<1>), dim3(B * heads), dim3(nthreads), lds, (hipStream_t)stream, (const bf16_t*)qkv, key_bias,
                     (const bf16_t*)dctx, (const bf16_t*)ctx, lse, delta, (bf16_t*)dqkv, S_pad, heads, scale);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
