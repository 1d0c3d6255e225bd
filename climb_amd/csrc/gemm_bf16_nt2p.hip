// bf16 NT GEMM, 128 x 192 x 64 tiles, TWO independent persistent workgroups of 4 waves per CU.
//
// Why, after gemm_bf16_ntp.hip (one 8-wave workgroup per CU, 256-row tiles): measured there, the k-loop runs at ~1100-1300 TF but
// every tile ends with an epilogue whose global stores are an HBM-write burst (5-13 us per round of 256 tiles) during which the CU's
// matrix pipes idle -- and on gfx9 a wave cannot let its stores drain under the next tile's k-loop, because loads and stores share
// ONE counter (vmcnt): the counted waits of the next k-loop would wait for the stores.  Two workgroups per CU solve both: each SIMD
// hosts one wave of each, the workgroups run different tiles with their own barriers and their own vmcnt, so while one is in its
// epilogue (VALU, stores, operand fetch of the residual) the other's waves own the matrix pipe.  Halving the tile height also halves
// the scheduling quantum: 12288 x 2304 is 1152 tiles over 512 resident workgroups (2.25 "rounds" of half-size tiles) instead of
// 432 over 256 (1.69 rounds of full-size ones).
//
// Geometry.  Waves 2 (M) x 2 (N); a wave owns 64 x 96 = [3 n-blocks][2 m-blocks] of v_mfma_f32_32x32x16_bf16, exactly the wave tile
// of the 256 x 192 kernel.  LDS per workgroup: 2 k-tile buffers of { A image [128][64] = 16 KB | 3 B units of 8 KB } = 80 KB -- two
// workgroups fill the CU's 160 KB.  Staging units: A0 / A1 = rows 0..63 / 64..127, B_p = n-block p of both wave columns; every unit is
// 8 pieces of 1 KB = 2 LDS-DMA instructions per lane.
//
// Schedule: the 3-phase table of gemm_bf16_phase.h (variant 0) with ONE s_barrier per phase: there is no second wave group inside
// the workgroup to stagger against (the other workgroup plays that role), so phase p = { ds_read the fragments of n-block p (+ all of
// A in phase 0); issue this phase's units; counted vmcnt; s_barrier; 8 MFMAs }.  Hazards: data waited for before the barrier of
// phase w is read from phase w + 1 on; a unit last read in phase q is restaged from phase q + 2 on (its readers passed their
// lgkmcnt before the barrier of phase q + 1).  Wait counts come from ntp_wait(..., gb = 2).
#include "gemm_bf16_phase.h"

#define NT2_BM 128
#define NT2_NI 3
#define NT2_BN (64 * NT2_NI)
#define NT2_A_BYTES (NT2_BM * GB_BK * 2)            // 16 KB
#define NT2_BUF (NT2_A_BYTES + NT2_NI * NTP_B_UNIT)   // 40 KB
#define NT2_GB 2

struct Nt2Stage {
  unsigned a_off[4];            // [unit 0/1][i]
  unsigned b_off[NT2_NI][2];
  unsigned wid;
};

template <int P, bool D1, bool D2, int K_ = 0>
__device__ __forceinline__ void nt2_issue(const Nt2Stage& sg, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, unsigned char* __restrict__ smem, int T) {
  if constexpr (K_ < NTP_MAXI) {
    constexpr int e = ntp_sched(NT2_NI, P, K_);
    if constexpr (e >= 0) {
      constexpr int unit = e / 4, d = e % 4;
      if constexpr ((d == 1 && D1) || (d == 2 && D2)) {
        const int t = T + d;
        unsigned char* buf = smem + (t & 1) * NT2_BUF;
        const long koff = (long)t * (GB_BK * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (unit < 2)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(A) + koff + sg.a_off[unit * 2 + i]),
                                             (lds_void_t*)(buf + unit * (NT2_A_BYTES / 2) + (sg.wid * 2 + i) * 1024), 16, 0, 0);
          else
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(B) + koff + sg.b_off[unit - 2][i]),
                                             (lds_void_t*)(buf + NT2_A_BYTES + (unit - 2) * NTP_B_UNIT + (sg.wid * 2 + i) * 1024), 16, 0, 0);
        }
      }
    }
    nt2_issue<P, D1, D2, K_ + 1>(sg, A, B, smem, T);
  }
}

__device__ __forceinline__ bf16x8 nt2_frag(const unsigned char* __restrict__ p) { return as_bf16x8(*reinterpret_cast<const u32x4*>(p)); }

template <int P, bool D1, bool D2, int W>
__device__ __forceinline__ void nt2_phase(f32x16 (&acc)[NT2_NI][2], bf16x8 (&a)[2][4], const Nt2Stage& sg, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[4],
                                          const unsigned (&boff)[4]) {
  const unsigned char* buf = smem + (T & 1) * NT2_BUF;
  bf16x8 b[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b[ks] = nt2_frag(buf + NT2_A_BYTES + P * NTP_B_UNIT + boff[ks]);
  if constexpr (P == 0) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[jj][ks] = nt2_frag(buf + jj * 4096 + aoff[ks]);
  }
  nt2_issue<P, D1, D2>(sg, A, B, smem, T);
  if constexpr (W >= 0) wait_vmcnt<(W < 0 ? 0 : W)>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) acc[P][jj] = CLIMB_MFMA_H16(b[ks], a[jj][ks], acc[P][jj], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int P = 0>
__device__ __forceinline__ void nt2_ktile(f32x16 (&acc)[NT2_NI][2], bf16x8 (&a)[2][4], const Nt2Stage& sg, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[4],
                                          const unsigned (&boff)[4]) {
  if constexpr (P < NT2_NI) {
    constexpr int W = ntp_wait(NT2_NI, 6, MODE == 0 ? 2 : (MODE == 1 ? 4 : 5), P, NT2_GB);
    nt2_phase<P, MODE <= 1, MODE == 0, W>(acc, a, sg, A, B, smem, T, aoff, boff);
    nt2_ktile<MODE, P + 1>(acc, a, sg, A, B, smem, T, aoff, boff);
  }
}
template <int P = 0>
__device__ __forceinline__ void nt2_prologue(const Nt2Stage& sg, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, unsigned char* __restrict__ smem) {
  if constexpr (P < NT2_NI) {
    nt2_issue<P, false, true>(sg, A, B, smem, -2);
    nt2_prologue<P + 1>(sg, A, B, smem);
  } else if constexpr (P < 2 * NT2_NI) {
    nt2_issue<P - NT2_NI, true, true>(sg, A, B, smem, -1);
    nt2_prologue<P + 1>(sg, A, B, smem);
  }
}

// Epilogue of one wave (64 x 96): EVERY load -- bias, and the residual / pre-activation operands of both 32-row blocks -- is issued
// before the first store (see NtBias in gemm_bf16_nt.h: a load behind a store costs a full store drain on gfx9), then the two
// blocks are turned through the wave's staging region and stored.
template <typename TO, int EPI>
__device__ __forceinline__ void nt2_epilogue(const f32x16 (&acc)[NT2_NI][2], unsigned char* __restrict__ stage, int el, int mw, int nw, int M, int N,
                                             TO* __restrict__ C, long ldc, const float* __restrict__ bias, const void* __restrict__ aux, long ldaux,
                                             bf16_t* __restrict__ aux_out, long ldauxo) {
  NtBias<NT2_NI> bb;
  AuxRegs<EPI, NT2_NI * 4> ax0, ax1;
  nt_aux_prefetch_l<EPI, NT2_NI, 1>(ax0, el, mw, nw, M, N, aux, ldaux, nullptr, 0);
  nt_bias_preload<NT2_NI>(bb, bias, el, nw, N);
  {
    f32x16 blk[NT2_NI];
#pragma unroll
    for (int i = 0; i < NT2_NI; ++i) blk[i] = acc[i][0];
    nt_epi_stage<NT2_NI>(blk, stage, el);
  }
  __builtin_amdgcn_sched_barrier(0);          // block 0's accumulators are dead: room for block 1's operands
  nt_aux_prefetch_l<EPI, NT2_NI, 1>(ax1, el, mw + 32, nw, M, N, aux, ldaux, nullptr, 0);
  nt_aux_touch(ax0);
  nt_aux_touch(ax1);                          // both operand sets have landed before the first store is issued
  __builtin_amdgcn_sched_barrier(0);
  nt_epi_drain<TO, EPI, NT2_NI, 0>(ax0, bb, stage, el, mw, nw, M, N, C, ldc, aux_out, ldauxo);
  __builtin_amdgcn_sched_barrier(0);
  {
    f32x16 blk[NT2_NI];
#pragma unroll
    for (int i = 0; i < NT2_NI; ++i) blk[i] = acc[i][1];
    nt_epi_stage<NT2_NI>(blk, stage, el);
  }
  nt_epi_drain<TO, EPI, NT2_NI, 0>(ax1, bb, stage, el, mw + 32, nw, M, N, C, ldc, aux_out, ldauxo);
}

template <typename TO, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt2_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                               TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                               const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                                                               int dephase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x 40 KB
  // De-phasing: all workgroups start together and every tile costs the same, so without help the whole chip computes, then the whole
  // chip stores (an HBM-write burst with idle matrix pipes), in lock step.  The workgroups of the second residency slot (ids >= half the
  // grid) hold back `dephase` x 64 clocks once, so that on every CU one workgroup's epilogue falls under the other's k-loop.
  if (dephase > 0 && blockIdx.x >= gridDim.x / 2) {
    const long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (long)dephase * 64) __builtin_amdgcn_s_sleep(8);
  }
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1, half = lane >> 5, l31 = lane & 31;
  const int nbm = (M + NT2_BM - 1) / NT2_BM, nbn = (N + NT2_BN - 1) / NT2_BN;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cs = ((2 * ks + half) ^ swz(l31)) << 4;
    aoff[ks] = (wr * 64 + l31) * 128 + cs;
    boff[ks] = (wc * 32 + l31) * 128 + cs;
  }
  const int nk = K / GB_BK;
  for (int id = blockIdx.x; id < nbm * nbn; id += gridDim.x) {
    int tm, tn;
    nt_tile_id<16>(id, nbm, nbn, tm, tn);                   // supertile of 16 x 128 = 2048 rows
    const int m0 = tm * NT2_BM, n0 = tn * NT2_BN;
    Nt2Stage sg;
    sg.wid = wid;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane, r = u * 64 + (slot >> 3), c = (slot & 7) ^ swz(r);
        int arow = m0 + r;
        arow = arow < M ? arow : M - 1;
        sg.a_off[u * 2 + i] = (unsigned)(((long)arow * lda + c * 8) * 2);
      }
#pragma unroll
    for (int p = 0; p < NT2_NI; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane, r = slot >> 3, c = (slot & 7) ^ swz(r);       // r = 0..63: wave column r >> 5, row r & 31 of n-block p
        int brow = n0 + (r >> 5) * (32 * NT2_NI) + p * 32 + (r & 31);
        brow = brow < N ? brow : N - 1;
        sg.b_off[p][i] = (unsigned)(((long)brow * ldb + c * 8) * 2);
      }
    f32x16 acc[NT2_NI][2];
#pragma unroll
    for (int i = 0; i < NT2_NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[2][4];
    nt2_prologue(sg, A, B, smem);
    wait_vmcnt<ntp_wait(NT2_NI, 6, -1, NT2_NI - 1, NT2_GB)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int T = 0;
    for (; T + 2 < nk; ++T) nt2_ktile<0>(acc, a, sg, A, B, smem, T, aoff, boff);
    nt2_ktile<1>(acc, a, sg, A, B, smem, T, aoff, boff);
    nt2_ktile<2>(acc, a, sg, A, B, smem, T + 1, aoff, boff);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                            // every wave is past its last fragment read: the buffers become staging regions
    __builtin_amdgcn_sched_barrier(0);
    int el = lane;
    asm volatile("" : "+v"(el));                             // keeps the epilogue's address arithmetic out of the k-loop's registers
    nt2_epilogue<TO, EPI>(acc, smem + wid * (NT2_NI * 4096), el, m0 + wr * 64, n0 + wc * (32 * NT2_NI), M, N, C, ldc, bias, aux, ldaux, aux_out, ldauxo);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
}

static int g_nt2_dephase = 0;
void climb_nt2_set_dephase(int v) { g_nt2_dephase = v; }

template <typename TO, int EPI>
static int nt2_launch_one(int nwg, hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K,
                          const float* bias, const void* aux, long ldaux, bf16_t* aux_out, long ldauxo) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt2_kernel<TO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * NT2_BUF);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  hipLaunchKernelGGL((gemm_bf16_nt2_kernel<TO, EPI>), dim3(nwg), dim3(256), 2 * NT2_BUF, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo,
                     nwg > 256 ? g_nt2_dephase : 0);
  return CLIMB_OK;
}

// c_dtype 0 fp32 / 1 bf16.  K % 64 == 0, K >= 128, operands below 4 GB; epilogues NONE / GELU / RESID / DGELU.
int climb_nt2_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                     const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st) {
  if ((K % GB_BK) || K < 2 * GB_BK) return CLIMB_EUNSUPPORTED;
  if (((long)M * lda + K) * 2 >= (1L << 32) || ((long)N * ldb + K) * 2 >= (1L << 32)) return CLIMB_EUNSUPPORTED;
  int nwg = ((M + NT2_BM - 1) / NT2_BM) * ((N + NT2_BN - 1) / NT2_BN);
  if (nwg > 512) nwg = 512;                                  // two persistent workgroups per CU
#define LNT2(TO, E) return nt2_launch_one<TO, E>(nwg, st, A, lda, B, ldb, (TO*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo)
#define LNT2_EPI(TO)                    \
  switch (epi) {                        \
    case EPI_NONE: LNT2(TO, EPI_NONE);  \
    case EPI_GELU: LNT2(TO, EPI_GELU);  \
    case EPI_RESID: LNT2(TO, EPI_RESID); \
    case EPI_DGELU: LNT2(TO, EPI_DGELU); \
    default: break;                     \
  }
  if (c_dtype == CLIMB_DT_F32) { LNT2_EPI(float) }
  if (c_dtype == CLIMB_DT_BF16) { LNT2_EPI(bf16_t) }
#undef LNT2_EPI
#undef LNT2
  return CLIMB_EUNSUPPORTED;
}
