// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-identical to an
// fmaf chain).  This is the "reference precision" path of the library: every encoder GEMM in fp32 mode, and the
// pooler / task-head GEMMs in both modes.  Operands are addressed through (row, k) strides so one kernel serves
//   Y = X W^T (forward), dX = dY W (input grad) and dW = dY^T X (weight grad, reduction over tokens)
// without materialising a transpose.  C[m,n] = epi( sum_k A(m,k) * B(n,k) + bias[n] ) (+ beta * C[m,n]).
//
// Tiling: 256 threads = 4 waves in 2x2; block tile BM x BN (128x128 or 64x64), BK = 16.  Tiles are staged
// global -> registers -> LDS as [k][row] (row contiguous) so that the one-float-per-lane MFMA operands
// (lane l: A[i = l&31][k = l>>5]) are conflict-free ds_read_b32; the next tile's global loads are issued before the
// MFMAs of the current one.
#include "common.h"

#define GF_BK 16
#define GF_PAD 4

template <int BR>
struct TileRegs { float v[BR * GF_BK / 256]; };

// mode: 0 = scalar (any strides), 1 = k-contiguous float4, 2 = row-contiguous float4
template <int BR>
__device__ __forceinline__ void tile_load(TileRegs<BR>& t, const float* __restrict__ P, long s_r, long s_k, int r0, int k0, int R, int K,
                                          int mode) {
  constexpr int NE = BR * GF_BK / 256;
  const int tid = threadIdx.x;
  if (mode == 1) {
#pragma unroll
    for (int p = 0; p < NE / 4; ++p) {
      int row = tid / 4 + 64 * p, kq = (tid & 3) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < R) {
        const float* q = P + (long)(r0 + row) * s_r + (k0 + kq);
        if (k0 + kq + 3 < K) v = ld4(q);
        else {                                   // ragged K tail: element-wise, zero-filled
          if (k0 + kq < K) v.x = q[0];
          if (k0 + kq + 1 < K) v.y = q[1];
          if (k0 + kq + 2 < K) v.z = q[2];
        }
      }
      t.v[p * 4 + 0] = v.x; t.v[p * 4 + 1] = v.y; t.v[p * 4 + 2] = v.z; t.v[p * 4 + 3] = v.w;
    }
  } else if (mode == 2) {
    constexpr int TPK = BR / 4;        // threads per k row
    constexpr int KPP = 256 / TPK;     // k rows per pass
#pragma unroll
    for (int p = 0; p < NE / 4; ++p) {
      int k = tid / TPK + KPP * p, r4 = (tid % TPK) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + k < K) {
        const float* q = P + (long)(k0 + k) * s_k + (r0 + r4);
        if (r0 + r4 + 3 < R) v = ld4(q);
        else {                                   // ragged row tail (e.g. 3129 VQA labels)
          if (r0 + r4 < R) v.x = q[0];
          if (r0 + r4 + 1 < R) v.y = q[1];
          if (r0 + r4 + 2 < R) v.z = q[2];
        }
      }
      t.v[p * 4 + 0] = v.x; t.v[p * 4 + 1] = v.y; t.v[p * 4 + 2] = v.z; t.v[p * 4 + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      int e = tid + 256 * j;
      int r = e % BR, k = e / BR;
      float v = 0.f;
      if (r0 + r < R && k0 + k < K) v = P[(long)(r0 + r) * s_r + (long)(k0 + k) * s_k];
      t.v[j] = v;
    }
  }
}

template <int BR>
__device__ __forceinline__ void tile_store(const TileRegs<BR>& t, float* __restrict__ S, int mode) {
  constexpr int NE = BR * GF_BK / 256;
  constexpr int LD = BR + GF_PAD;
  const int tid = threadIdx.x;
  if (mode == 1) {
#pragma unroll
    for (int p = 0; p < NE / 4; ++p) {
      int row = tid / 4 + 64 * p, kq = (tid & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[(kq + j) * LD + row] = t.v[p * 4 + j];
    }
  } else if (mode == 2) {
    constexpr int TPK = BR / 4;
    constexpr int KPP = 256 / TPK;
#pragma unroll
    for (int p = 0; p < NE / 4; ++p) {
      int k = tid / TPK + KPP * p, r4 = (tid % TPK) * 4;
      *reinterpret_cast<float4*>(&S[k * LD + r4]) = make_float4(t.v[p * 4], t.v[p * 4 + 1], t.v[p * 4 + 2], t.v[p * 4 + 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      int e = tid + 256 * j;
      S[(e / BR) * LD + (e % BR)] = t.v[j];
    }
  }
}

// XEPI: the adapter epilogues (SiLU, SiLU', dual residual) are compiled only into their own instantiation -- with all seven cases in one
// kernel the 128 x 128 variant spilled 320 B per lane and the parity mode's GEMMs ran at 37 instead of 66 TF
template <int BM, int BN, bool XEPI = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak, int modeA, const float* __restrict__ B,
                                                       long sbn, long sbk, int modeB, float* C, long ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, int epi, const float* aux, long ldaux, float* aux_out,
                                                       long ldauxo, float beta, const float* aux2, long ldaux2, int kper) {
  constexpr int WM = BM / 2, WN = BN / 2;   // per-wave tile
  constexpr int TM = WM / 32, TN = WN / 32; // 32x32 MFMA blocks per wave
  __shared__ __attribute__((aligned(16))) float As[GF_BK * (BM + GF_PAD)];
  __shared__ __attribute__((aligned(16))) float Bs[GF_BK * (BN + GF_PAD)];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileRegs<BM> ta;
  TileRegs<BN> tb;
  // split-K (skinny GEMMs of the task heads: M = batch, long K): blockIdx.z owns k in [kbeg, kend) and adds its partial sum with
  // fp32 atomics into a C the launcher zeroed; kper == 0 means no split
  const int kbeg = kper ? blockIdx.z * kper : 0;
  const int Kfull = K;
  if (kper) K = min(K, kbeg + kper);
  tile_load<BM>(ta, A, sam, sak, m0, kbeg, M, K, modeA);
  tile_load<BN>(tb, B, sbn, sbk, n0, kbeg, N, K, modeB);
  const int half = lane >> 5, l31 = lane & 31;
  for (int k0 = kbeg; k0 < K; k0 += GF_BK) {
    __syncthreads();  // previous tile fully consumed
    tile_store<BM>(ta, As, modeA);
    tile_store<BN>(tb, Bs, modeB);
    __syncthreads();
    if (k0 + GF_BK < K) {
      tile_load<BM>(ta, A, sam, sak, m0, k0 + GF_BK, M, K, modeA);
      tile_load<BN>(tb, B, sbn, sbk, n0, k0 + GF_BK, N, K, modeB);
    }
#pragma unroll
    for (int kk = 0; kk < GF_BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(kk + half) * (BM + GF_PAD) + wm * WM + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + half) * (BN + GF_PAD) + wn * WN + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // epilogue: D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (m)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + l31;
      if (n >= N) continue;
      const float bv = (bias && kbeg == 0) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= M) continue;
        float v = acc[i][j][r] + bv;
        if (epi == EPI_GELU) { aux_out[(long)m * ldauxo + n] = v; v = gelu_f(v); }
        else if (epi == EPI_RESID) v += aux[(long)m * ldaux + n];
        else if (epi == EPI_DGELU) v *= dgelu_f(aux[(long)m * ldaux + n]);
        else if (epi == EPI_TANH) v = tanhf(v);
        if constexpr (XEPI) {
          if (epi == EPI_SILU) { aux_out[(long)m * ldauxo + n] = v; v = silu_f(v); }
          else if (epi == EPI_DSILU) v *= dsilu_f(aux[(long)m * ldaux + n]);
          else if (epi == EPI_RESID2) v += aux[(long)m * ldaux + n] + aux2[(long)m * ldaux2 + n];
        }
        float* cp = C + (long)m * ldc + n;
        if (kper) { atomicAdd(cp, v); continue; }
        if (beta != 0.f) v += beta * (*cp);
        *cp = v;
      }
    }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// C[m,n] (ldc) = epi(sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + bias[n]) + beta*C
__global__ void zero_rows_kernel(float* __restrict__ C, long ldc, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N) C[(long)blockIdx.y * ldc + n] = 0.f;
}

extern "C" int climb_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, float* C, long ldc, int M, int N, int K,
                              const float* bias, int epi, const float* aux, long ldaux, float* aux_out, long ldauxo, float beta, const float* aux2,
                              long ldaux2, int allow_splitk, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return CLIMB_EINVAL;
  if ((epi == EPI_RESID || epi == EPI_DGELU || epi == EPI_DSILU || epi == EPI_RESID2) && !aux) return CLIMB_EINVAL;
  if ((epi == EPI_GELU || epi == EPI_SILU) && !aux_out) return CLIMB_EINVAL;
  if (epi == EPI_RESID2 && !aux2) return CLIMB_EINVAL;
  auto pick = [](const float* P, long s_r, long s_k, int R, int K_) -> int {
    if (s_k == 1 && (s_r % 4) == 0 && al16(P)) return 1;
    if (s_r == 1 && (s_k % 4) == 0 && al16(P)) return 2;
    return 0;
  };
  const int modeA = pick(A, sam, sak, M, K), modeB = pick(B, sbn, sbk, N, K);
  hipStream_t st = (hipStream_t)stream;
  const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
  if (allow_splitk && epi == EPI_NONE && (beta == 0.f || beta == 1.f) && tiles64 <= 64 && K >= 512) {
    int splits = (int)(256 / tiles64);
    if (splits > K / 128) splits = K / 128;
    if (splits > 1) {
      int kper = ((K + splits - 1) / splits + GF_BK - 1) / GF_BK * GF_BK;
      splits = (K + kper - 1) / kper;
      if (beta == 0.f) {   // a kernel, not hipMemset2DAsync: memset2D nodes with a pitch did not replay correctly under hipGraph capture
        hipLaunchKernelGGL(zero_rows_kernel, dim3((N + 255) / 256, M), dim3(256), 0, st, C, ldc, N);
        LAUNCH_CHECK();
      }
      dim3 grid((N + 63) / 64, (M + 63) / 64, splits);
      hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), grid, dim3(256), 0, st, A, sam, sak, modeA, B, sbn, sbk, modeB, C, ldc, M, N, K, bias, epi, aux,
                         ldaux, aux_out, ldauxo, beta, aux2, ldaux2, kper);
      LAUNCH_CHECK();
      return CLIMB_OK;
    }
  }
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  // rank-K updates with a short K (the heads' weight gradients: K = batch) are all epilogue (read-modify-write of C): the smaller
  // tile quadruples the workgroups that share that memory work
  const bool xepi = epi == EPI_SILU || epi == EPI_DSILU || epi == EPI_RESID2;
#define F32_LAUNCH(BM_, BN_, X_, grid_)                                                                                                      \
  hipLaunchKernelGGL((gemm_f32_kernel<BM_, BN_, X_>), grid_, dim3(256), 0, st, A, sam, sak, modeA, B, sbn, sbk, modeB, C, ldc, M, N, K, bias, epi, aux, \
                     ldaux, aux_out, ldauxo, beta, aux2, ldaux2, 0)
  if (tiles128 >= 192 && K > 128) {
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    if (xepi) F32_LAUNCH(128, 128, true, grid); else F32_LAUNCH(128, 128, false, grid);
  } else {
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (xepi) F32_LAUNCH(64, 64, true, grid); else F32_LAUNCH(64, 64, false, grid);
  }
#undef F32_LAUNCH
  LAUNCH_CHECK();
  return CLIMB_OK;
}
