// bf16 MFMA GEMMs for the ViLT encoder linears (throughput mode): v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
//   NT  C[M,N] = epi(A[M,K] . B[N,K]^T + bias)      forward (B = W) and input-gradient (B = W^T shadow) GEMMs
//   TN  C[N,K] += A[M,N]^T . B[M,K]                 weight-gradient GEMM, reduction over the token dimension M
//
// Both: block tile 128x128 (NT also 64x128 and 96x192), BK = 64, double-buffered LDS filled by LDS-DMA with the next tile's
// loads in flight under the current tile's MFMAs (one barrier per k-tile).
// NT operands are issued "swapped" (MFMA A = the weight-side tile, MFMA B = the token-side tile): in the 32x32 accumulator
// layout a lane then owns ONE output row and 4 CONSECUTIVE output columns per register group, which the epilogue turns
// through LDS into whole-row (128-byte line) global accesses.
//
// LDS images:
//   NT: [128 rows][64 k] bf16, 128-B rows, 16-B chunks XOR-swizzled by swz(row) so that the ds_read_b128 fragment reads
//       (16-lane groups = 16 different rows, same k-chunk) hit 16 distinct 4-bank groups: conflict-free.
//   TN: [64 m][128 cols] bf16, 256-B rows, 16-B chunks XOR-swizzled by (row & 3) << 2; fragments are gathered with
//       ds_read_b64_tr_b16 (the gfx950 LDS transpose read: a 16-lane group loads a [4 m][16 col] block and each lane
//       receives one column = 4 consecutive reduction indices), two reads per 8-element operand.  The swizzle puts 4
//       consecutive m-rows in 4 different bank quarters: conflict-free.
// Staging is LDS-DMA (global_load_lds_dwordx4, swizzle applied to the per-lane SOURCE address) whenever the shape allows;
// a register-staged variant of each kernel handles ragged K / ragged reduction tails.
#include "gemm_bf16_nt.h"


// ---------------------------------------------------------------------------------------------------------------- NT
// stage one [128][64] tile: thread t handles chunks (row = t/8 + 32p, c = t%8), p = 0..3
__device__ __forceinline__ void nt_load(u32x4 (&r)[4], const bf16_t* __restrict__ P, long ld, int row0, int k0, int R, int K) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int row = row0 + (t >> 3) + 32 * p;
    row = row < R ? row : R - 1;                       // clamp: rows beyond R are never stored
    const int k = k0 + (t & 7) * 8;
    if (k < K) r[p] = *reinterpret_cast<const u32x4*>(P + (long)row * ld + k);
    else r[p] = (u32x4){0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ void nt_store(const u32x4 (&r)[4], unsigned char* __restrict__ S) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (t >> 3) + 32 * p, c = t & 7;
    *reinterpret_cast<u32x4*>(S + row * 128 + ((c ^ swz(row)) << 4)) = r[p];
  }
}


// 128 x 128 tile; waves arranged (8/MJ/2) x 2, each owning MJ*32 rows x 64 columns: MJ = 1 -> 8 waves (twice the waves per CU
// hiding LDS-DMA / L2 latency for the same LDS footprint, at 1.5x the fragment reads per MFMA); MJ = 2 -> 4 waves.
// (16 waves of 32 x 32 measured the same as 8 within noise on the multi-round shapes: 60 / 105 / 100 us against 62 / 103 / 102.)
// SPLIT (r06, split.hip): A / B are the hi planes of (hi, lo) pairs, the lo planes a_lo / b_lo elements behind; the reduction walks 3 K / 64 k-tiles --
// the three products A {hi, hi, lo} x B {hi, lo, hi} of one k-tile after each other (K % 64 == 0)
template <typename TO, int EPI, bool GLDS, int MJ, bool SPLIT = false>
__global__ __launch_bounds__(512 / MJ)
void gemm_bf16_nt_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb, TO* __restrict__ C, long ldc, int M, int N,
                         int K, const float* __restrict__ bias, const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                         const bf16_t* __restrict__ aux2, long ldaux2, long a_lo = 0, long b_lo = 0) {
  constexpr int NW = 8 / MJ;                       // waves per workgroup
  static_assert(GLDS || MJ == 2, "the register-staged fallback is written for 256 threads");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * GB_BM * GB_BK * 2];   // [buf][A|B][128][64] bf16 = 64 KB
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
  int tm, tn;
  nt_tile<8>((M + GB_BM - 1) / GB_BM, (N + GB_BN - 1) / GB_BN, tm, tn);
  const int m0 = tm * GB_BM, n0 = tn * GB_BN;
  f32x16 acc[2][MJ];   // [n block][m block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u32x4 ra[4], rb[4];
  const int ksp = K / GB_BK;
  const int nk = SPLIT ? 3 * ksp : (K + GB_BK - 1) / GB_BK;
  AuxRegs<EPI, 8 * MJ> ax;
  nt_aux_prefetch<EPI, 2, MJ>(ax, m0 + wm * (32 * MJ), n0 + wn * 64, M, N, aux, ldaux, aux2, ldaux2);
#define KOFF(kt_) (SPLIT ? ((kt_) / 3) * GB_BK : (kt_) * GB_BK)
#define KPA(kt_) (SPLIT && (kt_) % 3 == 2 ? A + a_lo : A)
#define KPB(kt_) (SPLIT && (kt_) % 3 == 1 ? B + b_lo : B)
  if constexpr (GLDS) {
    nt_glds<NW>(A, lda, m0, KOFF(0), M, smem);
    nt_glds<NW>(B, ldb, n0, KOFF(0), N, smem + GB_BM * GB_BK * 2);
  } else {
    nt_load(ra, A, lda, m0, KOFF(0), M, K);
    nt_load(rb, B, ldb, n0, KOFF(0), N, K);
    nt_store(ra, smem);
    nt_store(rb, smem + GB_BM * GB_BK * 2);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* As = smem + (kt & 1) * (2 * GB_BM * GB_BK * 2);
    unsigned char* Bs = As + GB_BM * GB_BK * 2;
    if (kt + 1 < nk) {
      if constexpr (GLDS) {
        unsigned char* An = smem + ((kt + 1) & 1) * (2 * GB_BM * GB_BK * 2);
        nt_glds<NW>(KPA(kt + 1), lda, m0, KOFF(kt + 1), M, An);
        nt_glds<NW>(KPB(kt + 1), ldb, n0, KOFF(kt + 1), N, An + GB_BM * GB_BK * 2);
      } else {
        nt_load(ra, KPA(kt + 1), lda, m0, KOFF(kt + 1), M, K);
        nt_load(rb, KPB(kt + 1), ldb, n0, KOFF(kt + 1), N, K);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fa[MJ], fb[2];
      const int c = 2 * ks + half;
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int row = wm * (32 * MJ) + j * 32 + l31;
        fa[j] = as_bf16x8(*reinterpret_cast<const u32x4*>(As + row * 128 + ((c ^ swz(row)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = wn * 64 + i * 32 + l31;
        fb[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bs + row * 128 + ((c ^ swz(row)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] = CLIMB_MFMA_H16(fb[i], fa[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (!GLDS) {
      if (kt + 1 < nk) {
        unsigned char* An = smem + ((kt + 1) & 1) * (2 * GB_BM * GB_BK * 2);
        nt_store(ra, An);
        nt_store(rb, An + GB_BM * GB_BK * 2);
      }
    }
    __syncthreads();    // with LDS-DMA in flight the compiler drains vmcnt(0) here: tile kt+1 has landed for every wave
  }
#undef KOFF
#undef KPA
#undef KPB
  // every wave is past the last k-tile (barrier above): the buffers become the per-wave staging regions (8 KB each)
  nt_epilogue<TO, EPI, 2, MJ>(acc, ax, smem + wid * 8192, m0 + wm * (32 * MJ), n0 + wn * 64, M, N, C, ldc, bias, aux_out, ldauxo);
}


// 64 x 128 tile variant (4 waves as 2 x 2, each 32 x 64; LDS 2 x 24 KB -> three workgroups per CU).
// Same fragments, swizzle, DMA staging and epilogue; used where 128 x 128 tiles leave the last round of workgroups mostly empty.
template <typename TO, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_nt64_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                             TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                             const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                                                             const bf16_t* __restrict__ aux2, long ldaux2) {
  constexpr int ABYTES = 64 * GB_BK * 2, BBYTES = GB_BN * GB_BK * 2, STAGE = ABYTES + BBYTES;     // 8 KB + 16 KB
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
  int tm, tn;
  nt_tile<16>((M + 63) / 64, (N + GB_BN - 1) / GB_BN, tm, tn);      // 16 M-tiles of 64 = the same 1024-row supertile
  const int m0 = tm * 64, n0 = tn * GB_BN;
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int nk = K / GB_BK;
  AuxRegs<EPI, 8> ax;
  nt_aux_prefetch<EPI, 2, 1>(ax, m0 + wm * 32, n0 + wn * 64, M, N, aux, ldaux, aux2, ldaux2);
  nt_glds<4, 64>(A, lda, m0, 0, M, smem);
  nt_glds<4, 128>(B, ldb, n0, 0, N, smem + ABYTES);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* As = smem + (kt & 1) * STAGE;
    const unsigned char* Bs = As + ABYTES;
    if (kt + 1 < nk) {
      unsigned char* An = smem + ((kt + 1) & 1) * STAGE;
      nt_glds<4, 64>(A, lda, m0, (kt + 1) * GB_BK, M, An);
      nt_glds<4, 128>(B, ldb, n0, (kt + 1) * GB_BK, N, An + ABYTES);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + half;
      const int rowa = wm * 32 + l31;
      const bf16x8 fa = as_bf16x8(*reinterpret_cast<const u32x4*>(As + rowa * 128 + ((c ^ swz(rowa)) << 4)));
      bf16x8 fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int rowb = wn * 64 + i * 32 + l31;
        fb[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bs + rowb * 128 + ((c ^ swz(rowb)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][0] = CLIMB_MFMA_H16(fb[i], fa, acc[i][0], 0, 0, 0);
    }
    __syncthreads();
  }
  nt_epilogue<TO, EPI, 2, 1>(acc, ax, smem + wid * 8192, m0 + wm * 32, n0 + wn * 64, M, N, C, ldc, bias, aux_out, ldauxo);
}


// 96 x 192 tile variant: 6 waves as 3 x 2, each 32 x 96 (three 32x32 blocks sharing one token fragment).
// 12288 x {768, 2304, 3072} outputs split into exactly {512, 1536, 2048} such tiles = {1, 3, 4} FULL rounds of two
// workgroups per CU (LDS 2 x 36 KB), where 128 x 128 gives 1.125 / 3.375 / 4.5 rounds: no under-filled last round.
#define N96_BM 96
#define N96_BN 192
template <typename TO, int EPI>
__global__ __launch_bounds__(384) void gemm_bf16_nt96_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                             TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                             const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                                                             const bf16_t* __restrict__ aux2, long ldaux2) {
  constexpr int ABYTES = N96_BM * GB_BK * 2, BBYTES = N96_BN * GB_BK * 2, STAGE = ABYTES + BBYTES;     // 12 KB + 24 KB
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
  int tm, tn;
  nt_tile<8>((M + N96_BM - 1) / N96_BM, (N + N96_BN - 1) / N96_BN, tm, tn);      // supertile: 768 rows x all N-tiles per L2
  const int m0 = tm * N96_BM, n0 = tn * N96_BN;
  f32x16 acc[3][1];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int nk = K / GB_BK;
  AuxRegs<EPI, 12> ax;
  nt_aux_prefetch<EPI, 3, 1>(ax, m0 + wm * 32, n0 + wn * 96, M, N, aux, ldaux, aux2, ldaux2);
  nt_glds<6, N96_BM>(A, lda, m0, 0, M, smem);
  nt_glds<6, N96_BN>(B, ldb, n0, 0, N, smem + ABYTES);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* As = smem + (kt & 1) * STAGE;
    const unsigned char* Bs = As + ABYTES;
    if (kt + 1 < nk) {
      unsigned char* An = smem + ((kt + 1) & 1) * STAGE;
      nt_glds<6, N96_BM>(A, lda, m0, (kt + 1) * GB_BK, M, An);
      nt_glds<6, N96_BN>(B, ldb, n0, (kt + 1) * GB_BK, N, An + ABYTES);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + half;
      const int rowa = wm * 32 + l31;
      const bf16x8 fa = as_bf16x8(*reinterpret_cast<const u32x4*>(As + rowa * 128 + ((c ^ swz(rowa)) << 4)));
      bf16x8 fb[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int rowb = wn * 96 + i * 32 + l31;
        fb[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bs + rowb * 128 + ((c ^ swz(rowb)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i][0] = CLIMB_MFMA_H16(fb[i], fa, acc[i][0], 0, 0, 0);
    }
    __syncthreads();
  }
  // 6 waves x 12 KB of staging = the 72 KB of k-loop buffers
  nt_epilogue<TO, EPI, 3, 1>(acc, ax, smem + wid * 12288, m0 + wm * 32, n0 + wn * 96, M, N, C, ldc, bias, aux_out, ldauxo);
}

// 192 x 192 tile, ONE workgroup of 12 waves per CU, three 48 KB LDS stages: the deep-prefetch variant.
// The 128 x 128 kernels keep one k-tile in flight; measured, their k-loop sits at 28-36 % MFMA utilisation because an
// LDS-DMA tile takes longer to land (~1-2 us under load) than the MFMAs of one k-tile last, and a second 64 KB workgroup per CU
// is all the LDS there is.  Here two k-tiles are always in flight (raw s_barrier + counted vmcnt, never vmcnt(0) in the loop),
// the tile reads 1.5x fewer operand bytes per flop, and 12288 x {768, 2304, 3072} outputs are exactly {1, 3, 4} rounds of 256 CUs.
// Waves: 6 (M) x 2 (N), each 32 x 96 = three 32x32 blocks sharing one token fragment (3 waves per SIMD).
// Measured alternative (r01, slower, removed): the same tile as a persistent kernel with two producer waves issuing all
// LDS-DMAs and 12 consumer waves whose epilogue stores drain under the next tile's k-loop -- qkv 66 / GELU 162 / xGELU' 118 us
// against 62 / 114 / 111 for two independent 128 x 128 workgroups per CU: with one workgroup per CU nothing computes while the
// consumers run the (VALU-heavy) epilogue.
// Also measured and removed: 128 x 128 tiles with BK = 32, three 16 KB stages and THREE workgroups per CU (24 waves, direct
// epilogue): 75 / 154 / 142 us on the same three shapes -- twice the barriers per k and a 64-byte-row LDS image cost more than the
// extra occupancy and prefetch depth return.
#define N192_T 192
#define N192_STAGE (2 * N192_T * GB_BK * 2)      // A + B image of one k-tile: 48 KB
template <typename TO, int EPI>
__global__ __launch_bounds__(768) void gemm_bf16_nt192_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                              TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                              const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                                                              const bf16_t* __restrict__ aux2, long ldaux2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 3 stages = 144 KB
  constexpr int ABYTES = N192_T * GB_BK * 2;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wm = wid >> 1, wn = wid & 1, half = lane >> 5, l31 = lane & 31;
  int tm, tn;
  nt_tile<8>((M + N192_T - 1) / N192_T, N / N192_T, tm, tn);
  const int m0 = tm * N192_T, n0 = tn * N192_T;
  f32x16 acc[3][1];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int nk = K / GB_BK;
  // prologue: k-tiles 0 and 1 in flight (4 DMA instructions per wave per stage)
  nt_glds<12, N192_T>(A, lda, m0, 0, M, smem);
  nt_glds<12, N192_T>(B, ldb, n0, 0, N, smem + ABYTES);
  if (nk > 1) {
    nt_glds<12, N192_T>(A, lda, m0, GB_BK, M, smem + N192_STAGE);
    nt_glds<12, N192_T>(B, ldb, n0, GB_BK, N, smem + N192_STAGE + ABYTES);
  }
  AuxRegs<EPI, 12> ax;
  nt_aux_prefetch<EPI, 3, 1>(ax, m0 + wm * 32, n0 + wn * 96, M, N, aux, ldaux, aux2, ldaux2);
  for (int kt = 0; kt < nk; ++kt) {
    // k-tile kt has landed for THIS wave once at most the 4 newest vector-memory operations are outstanding (returns are in
    // order, and k-tile kt+1's 4 DMAs were issued after it); the barrier extends that to every wave's share, and also says
    // every wave is done reading k-tile kt-1, whose stage is refilled next.
    if (kt + 1 < nk) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) {
      unsigned char* Sn = smem + ((kt + 2) % 3) * N192_STAGE;
      nt_glds<12, N192_T>(A, lda, m0, (kt + 2) * GB_BK, M, Sn);
      nt_glds<12, N192_T>(B, ldb, n0, (kt + 2) * GB_BK, N, Sn + ABYTES);
    }
    const unsigned char* As = smem + (kt % 3) * N192_STAGE;
    const unsigned char* Bs = As + ABYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + half;
      const int rowa = wm * 32 + l31;
      const bf16x8 fa = as_bf16x8(*reinterpret_cast<const u32x4*>(As + rowa * 128 + ((c ^ swz(rowa)) << 4)));
      bf16x8 fb[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int rowb = wn * 96 + i * 32 + l31;
        fb[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(Bs + rowb * 128 + ((c ^ swz(rowb)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) acc[i][0] = CLIMB_MFMA_H16(fb[i], fa, acc[i][0], 0, 0, 0);
    }
  }
  __syncthreads();      // every wave is past the last k-tile: the stages become 12 per-wave 12 KB staging regions
  nt_epilogue<TO, EPI, 3, 1>(acc, ax, smem + wid * 12288, m0 + wm * 32, n0 + wn * 96, M, N, C, ldc, bias, aux_out, ldauxo);
}

template <typename TO, int EPI>
static int nt192_launch(int nwg, hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K,
                        const float* bias, const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, const bf16_t* aux2, long ldaux2) {
  static bool configured = false;      // per instantiation; the attribute is sticky for the process
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt192_kernel<TO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * N192_STAGE);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  hipLaunchKernelGGL((gemm_bf16_nt192_kernel<TO, EPI>), dim3(nwg), dim3(768), 3 * N192_STAGE, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux,
                     aux_out, ldauxo, aux2, ldaux2);
  return CLIMB_OK;
}

static int g_tn_target = 512;
static int g_tn_waves = 8;     // measured (r01): 8 waves of 32 x 64 282 us per layer vs 4 waves of 64 x 64 318 us
static int g_nt_waves = 8;     // measured on MI355X (r01): 8 waves 537 TF vs 4 waves 514 TF average over the per-layer shapes
static int g_nt_96 = 1;        // 96x192 tiles when they tile the problem into full rounds
static int g_nt_192 = 1;       // 192x192 deep-prefetch tiles for large problems with N % 192 == 0
static int g_nt_256 = 1;       // persistent 256-row tiles (gemm_bf16_ntp.hip): 0 never, 1 auto (large problems), 2 / 3 force 256 / 192 columns
static int g_tn_p = 1;         // weight gradients on the persistent 256-row-tile kernel (gemm_bf16_tnp.hip) when the shape allows
static int g_nt_small_m = 1;   // 64x128 tiles for shapes whose 128x128 tiling quantises badly on 512 workgroup slots
void climb_attn_set_qb(int v);
void climb_attn_set_bwd_fused(int v);
void climb_ntsk_enable(int v);
void climb_ntp_set_sw(int v);
void climb_ntp_set_dephase(int v);
void climb_ntsk_set_workspace(void* ptr, long bytes);
void climb_skinny_set_probe(int v);
int climb_nt256_get_grid();
void climb_attn_set_1pp_grid(int v);
void climb_ln_set_rpw(int v);
void climb_attn_split_set(int key, int v);
// current value of a library option (only the ones a caller has to put back: 9 = persistent NT grid); -1 = not readable
int climb_tn_get_stagger();
extern "C" int climb_get_option(int key) { return key == 9 ? climb_nt256_get_grid() : (key == 22 ? climb_tn_get_stagger() : -1); }
extern "C" int climb_set_option(int key, int value) {
  if (key == 1 && (value == 4 || value == 8)) { g_nt_waves = value; return CLIMB_OK; }
  if (key == 2) { g_nt_small_m = value; return CLIMB_OK; }
  if (key == 4) { g_nt_96 = value; return CLIMB_OK; }
  if (key == 5) { g_nt_192 = value; return CLIMB_OK; }
  if (key == 7 && value >= 0 && value <= 4) { g_nt_256 = value; return CLIMB_OK; }
  if (key == 8) { climb_nt256_set_probe(value); return CLIMB_OK; }            // 1: k-loop only; 256 * gm: supertile height override
  if (key == 10 && (value == 0 || value == 1)) { g_tn_p = value; return CLIMB_OK; }
  if (key == 11 && value >= 0) { climb_nt2_set_dephase(value); return CLIMB_OK; }
  if (key == 12 && value >= 0 && value <= 2) { climb_attn_set_qb(value); return CLIMB_OK; }
  if (key == 13 && value >= 0 && value <= 4) { climb_attn_set_bwd_fused(value); return CLIMB_OK; }
  if (key == 14 && (value == 0 || value == 1)) { climb_ntsk_enable(value); return CLIMB_OK; }
  if (key == 15 && (value == 0 || value == 1)) { climb_ntp_set_sw(value); return CLIMB_OK; }
  if (key == 16 && value >= 0 && value < 4000) { climb_ntp_set_dephase(value); return CLIMB_OK; }
  if (key == 9 && value >= 0) { climb_nt256_set_grid(value); climb_nt4_set_grid(value); return CLIMB_OK; }
  if (key == 17 && value >= 0 && value <= 5) { climb_nt4_set(value); return CLIMB_OK; }
  if (key == 18 && value >= 0) { climb_nt4_set_probe(value); return CLIMB_OK; }
  if (key == 19 && value >= 0 && value <= 2) { climb_skinny_set_probe(value); return CLIMB_OK; }
  if (key == 20 && value >= 0) { climb_attn_set_1pp_grid(value); return CLIMB_OK; }
  if (key == 21 && value >= 1 && value <= 3) { climb_ln_set_rpw(value); return CLIMB_OK; }
  if (key == 22 && value >= 0 && value <= 116) { climb_tn_set_stagger(value); return CLIMB_OK; }
  if (key >= 23 && key <= 25) { climb_attn_split_set(key, value); return CLIMB_OK; }          // split-operand attention: rows per LDS chunk, waves per workgroup (fwd / bwd)
  if (key == 6 && (value == 4 || value == 8)) { g_tn_waves = value; return CLIMB_OK; }
  if (key == 3 && value > 0) { g_tn_target = value; return CLIMB_OK; }
  return CLIMB_EINVAL;
}

template <typename TO>
static int nt_dispatch(const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K, const float* bias, int epi,
                       const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, const bf16_t* aux2, long ldaux2, hipStream_t st) {
  const int nwg = ((M + GB_BM - 1) / GB_BM) * ((N + GB_BN - 1) / GB_BN);
  dim3 grid(nwg);
  const bool glds = (K % GB_BK) == 0;      // the DMA path cannot zero-fill a ragged K tail
  // tile-height choice: 2 workgroups of 128x128 (or 3 of 64x128) fit a CU; pick the tiling whose last wave of workgroups
  // is fuller (the 768-wide GEMMs leave 576 = 512 + 64 tiles of 128x128: a nearly empty second round)
  const int nwg64 = ((M + 63) / 64) * ((N + GB_BN - 1) / GB_BN);
  auto waste = [](int tiles, int slots) { int r = tiles % slots; return r == 0 ? 0.0 : (double)(slots - r) / ((tiles + slots - 1) / slots * (double)slots); };
  const int nwg96 = ((M + N96_BM - 1) / N96_BM) * ((N + N96_BN - 1) / N96_BN);
  const bool use96 = glds && g_nt_96 == 1 && (M % N96_BM) == 0 && (N % N96_BN) == 0 && nwg96 == 512;      // exactly one full round (measured: better than 64x128 there, worse than 128x128 on multi-round shapes)
  const bool use64 = glds && g_nt_small_m == 1 && waste(nwg64, 768) + 0.10 < waste(nwg, 512);
#define NT_ARGS A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo, aux2, ldaux2
  const int nwg192 = ((M + N192_T - 1) / N192_T) * (N / N192_T);
  // measured (r01, M = 12288): the single-round N = 768 GEMMs gain 15-25 % (K = 3072: 102 -> 79 us); on multi-round shapes the
  // epilogue-heavy kernels lose the second resident workgroup that computes while the first one stores, and come out 5-12 % slower
  const bool use192 = glds && g_nt_192 == 1 && (N % N192_T) == 0 && K >= 2 * GB_BK && nwg192 >= 160 && nwg192 <= 256;
  // persistent 256 x {256, 192} tiles (gemm_bf16_ntp.hip): the tile width whose last round of 256 workgroups wastes less
  const bool epi256 = epi == EPI_NONE || epi == EPI_GELU || epi == EPI_RESID || epi == EPI_DGELU || epi == EPI_GELUD || epi == EPI_MUL;
  if (glds && epi256 && g_nt_256 == 1) {      // (forcing one of the 8-wave tile shapes through option 7 also keeps this kernel out)
    int rc = climb_nt4_launch(A, lda, B, ldb, C, ldc, sizeof(TO) == 4 ? CLIMB_DT_F32 : CLIMB_DT_BF16, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo, st);
    if (rc == CLIMB_OK) { LAUNCH_CHECK(); return CLIMB_OK; }
    if (rc != CLIMB_EUNSUPPORTED) return rc;
  }
  if (glds && epi256 && K >= 2 * GB_BK && g_nt_256 != 0) {
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t192 = (long)((M + 255) / 256) * ((N + 191) / 192);
    const long c256 = (t256 + 255) / 256 * 256, c192 = (t192 + 255) / 256 * 192;       // rounds x tile width
    int bn = c192 < c256 ? 192 : 256;
    if (g_nt_256 == 2) bn = 256;
    if (g_nt_256 == 3) bn = 192;
    if (g_nt_256 == 4) {            // two 4-wave workgroups per CU, 128 x 192 tiles (gemm_bf16_nt2p.hip)
      int rc = climb_nt2_launch(A, lda, B, ldb, C, ldc, sizeof(TO) == 4 ? CLIMB_DT_F32 : CLIMB_DT_BF16, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo, st);
      if (rc == CLIMB_OK) { LAUNCH_CHECK(); return CLIMB_OK; }
      if (rc != CLIMB_EUNSUPPORTED) return rc;
    }
    const bool big = (long)M * N >= 2048L * 768;            // below that the 64 x 128 / 128 x 128 kernels fill the chip better
    // measured (r02, M = 12288): at K = 768 the single-round N = 768 shapes are a tie or better on the 192 x 192 three-stage kernel
    // (dctx 24.6 vs 26.9 us); from K = 2304 on the persistent kernel's k-loop wins (down 88 -> 81, dhn 74 -> 69, dxn 53 -> 51 us)
    // r03, after the supertile = one XCD's share fix: the persistent kernel wins or ties at K = 768 too (out-projection + residual 31.7 -> 28.2,
    // dctx 24.6 -> 24.1 us), so the 192 x 192 kernel (and its 52 B/lane spill) is off the step's path; it still serves tile counts in [160, 256]
    // that are not "big"
    const bool keep192 = false;
    if (g_nt_256 >= 2 || (big && !keep192)) {
      int rc = climb_nt256_launch(bn, A, lda, B, ldb, C, ldc, sizeof(TO) == 4 ? CLIMB_DT_F32 : CLIMB_DT_BF16, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo,
                                  aux2, ldaux2, st);
      if (rc == CLIMB_OK) { LAUNCH_CHECK(); return CLIMB_OK; }
      if (rc != CLIMB_EUNSUPPORTED) return rc;
    }
  }
#define NT_LAUNCH(E)                                                                                                           \
  do {                                                                                                                         \
    if (use192) {                                                                                                              \
      int rc = nt192_launch<TO, E>(nwg192, st, NT_ARGS);                                                                       \
      if (rc != CLIMB_OK) return rc;                                                                                           \
    } else if (use96) hipLaunchKernelGGL((gemm_bf16_nt96_kernel<TO, E>), dim3(nwg96), dim3(384), 0, st, NT_ARGS); \
    else if (use64) hipLaunchKernelGGL((gemm_bf16_nt64_kernel<TO, E>), dim3(nwg64), dim3(256), 0, st, NT_ARGS); \
    else if (glds && g_nt_waves == 8) hipLaunchKernelGGL((gemm_bf16_nt_kernel<TO, E, true, 1>), grid, dim3(512), 0, st, NT_ARGS);   \
    else if (glds) hipLaunchKernelGGL((gemm_bf16_nt_kernel<TO, E, true, 2>), grid, dim3(256), 0, st, NT_ARGS);                 \
    else hipLaunchKernelGGL((gemm_bf16_nt_kernel<TO, E, false, 2>), grid, dim3(256), 0, st, NT_ARGS);                          \
  } while (0)
  switch (epi) {
    case EPI_NONE: NT_LAUNCH(EPI_NONE); break;
    case EPI_GELU: NT_LAUNCH(EPI_GELU); break;
    case EPI_RESID: NT_LAUNCH(EPI_RESID); break;
    case EPI_DGELU: NT_LAUNCH(EPI_DGELU); break;
    case EPI_SILU: NT_LAUNCH(EPI_SILU); break;
    case EPI_DSILU: NT_LAUNCH(EPI_DSILU); break;
    case EPI_RESID2: NT_LAUNCH(EPI_RESID2); break;
    case EPI_GELUD:          // (16-bit outputs only: the saved derivative and the activation are GEMM operands of the backward)
      if constexpr (sizeof(TO) == 2) { NT_LAUNCH(EPI_GELUD); break; } else return CLIMB_EINVAL;
    case EPI_MUL:
      if constexpr (sizeof(TO) == 2) { NT_LAUNCH(EPI_MUL); break; } else return CLIMB_EINVAL;
    default: return CLIMB_EINVAL;
  }
#undef NT_LAUNCH
#undef NT_ARGS
  LAUNCH_CHECK();
  return CLIMB_OK;
}

static inline bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// r06 (split.hip): the 128 x 128 kernel over split operands -- every shape the four-wave kernel does not take (ragged M, small problems).  fp32 C.
int climb_nt_split_generic(const bf16_t* A, long lda, long a_lo, const bf16_t* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias,
                           int epi, const float* aux, long ldaux, hipStream_t st) {
  if (K % GB_BK) return CLIMB_EUNSUPPORTED;
  const int nwg = ((M + GB_BM - 1) / GB_BM) * ((N + GB_BN - 1) / GB_BN);
#define NTS(E) hipLaunchKernelGGL((gemm_bf16_nt_kernel<float, E, true, 1, true>), dim3(nwg), dim3(512), 0, st, A, lda, B, ldb, C, ldc, M, N, K, bias, (const void*)aux, ldaux, \
                                  (bf16_t*)nullptr, 0L, (const bf16_t*)nullptr, 0L, a_lo, b_lo)
  if (epi == EPI_NONE) NTS(EPI_NONE);
  else if (epi == EPI_RESID) NTS(EPI_RESID);
  else return CLIMB_EINVAL;
#undef NTS
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// C (c_dtype: 0 fp32 / 1 bf16) [M,N] = epi(A[M,K] B[N,K]^T + bias).  A, B bf16 with K contiguous; K % 8 == 0, N % 4 == 0.
// epi 1/5 (GELU/SiLU): aux_out (bf16 [M,N]) receives the pre-activation.  epi 2: aux = fp32 residual [M,N].  epi 3/6: aux = bf16
// pre-activation (x gelu'/silu').  epi 7: aux = fp32 residual and aux2 = bf16 second residual.  epi 8 (16-bit C only): aux_out receives
// gelu'(pre-activation) instead of the pre-activation; epi 9 (16-bit C only): C = acc * aux, aux = what epi 8 saved.
extern "C" int climb_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K,
                                  const float* bias, int epi, const void* aux, long ldaux, void* aux_out, long ldauxo, const void* aux2, long ldaux2,
                                  void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K % 8) || (N % 4) || (lda % 8) || (ldb % 8) || (ldc % 4) || !al16p(A) || !al16p(B) || !al16p(C)) return CLIMB_EINVAL;
  if ((epi == EPI_RESID || epi == EPI_DGELU || epi == EPI_DSILU || epi == EPI_RESID2 || epi == EPI_MUL) && !aux) return CLIMB_EINVAL;
  if ((epi == EPI_GELU || epi == EPI_SILU || epi == EPI_GELUD) && !aux_out) return CLIMB_EINVAL;
  if ((epi == EPI_GELUD || epi == EPI_MUL) && c_dtype != CLIMB_DT_BF16) return CLIMB_EINVAL;
  if (epi == EPI_RESID2 && !aux2) return CLIMB_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (c_dtype == CLIMB_DT_F32)
    return nt_dispatch<float>((const bf16_t*)A, lda, (const bf16_t*)B, ldb, (float*)C, ldc, M, N, K, bias, epi, aux, ldaux, (bf16_t*)aux_out, ldauxo, (const bf16_t*)aux2, ldaux2, st);
  if (c_dtype == CLIMB_DT_BF16)
    return nt_dispatch<bf16_t>((const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, M, N, K, bias, epi, aux, ldaux, (bf16_t*)aux_out, ldauxo, (const bf16_t*)aux2, ldaux2, st);
  return CLIMB_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------- TN
#define TN_BR 64            // reduction rows (tokens) per LDS tile
#define TN_PITCH 256        // bytes per LDS row: 128 cols * 2 B, 16-B chunks XOR-swizzled by tswz(row)
__device__ __forceinline__ int tswz(int row) { return (row & 3) << 2; }   // 4 consecutive m-rows -> 4 different bank quarters

// stage one [64 m][128 cols] tile: thread t handles (row = t/16 + 16p, chunk = t%16), p = 0..3
__device__ __forceinline__ void tn_load(u32x4 (&r)[4], const bf16_t* __restrict__ P, long ld, int m0, int c0, int m_end, int Ccols) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int m = m0 + (t >> 4) + 16 * p;
    int c = c0 + (t & 15) * 8;
    const bool ok = m < m_end && c < Ccols;              // rows past the end contribute zeros to the reduction
    r[p] = ok ? *reinterpret_cast<const u32x4*>(P + (long)m * ld + c) : (u32x4){0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ void tn_store(const u32x4 (&r)[4], unsigned char* __restrict__ S) {
  const int t = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (t >> 4) + 16 * p;
    *reinterpret_cast<u32x4*>(S + row * TN_PITCH + (((t & 15) ^ tswz(row)) << 4)) = r[p];
  }
}
// LDS-DMA staging of a FULL [64][128] tile (rows must all be inside the reduction range; columns are clamped)
template <int NW = 4>
__device__ __forceinline__ void tn_glds(const bf16_t* __restrict__ P, long ld, int m0, int c0, int Ccols, unsigned char* __restrict__ S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int j = wave * (16 / NW) + i;
    const int slot = j * 64 + lane, row = slot >> 4, c = (slot & 15) ^ tswz(row);
    int col = c0 + c * 8;
    col = col < Ccols ? col : Ccols - 8;
    __builtin_amdgcn_global_load_lds((gbl_void_t*)(P + (long)(m0 + row) * ld + col), (lds_void_t*)(S + j * 1024), 16, 0, 0);
  }
}
// 8-element MFMA operand for output index (col0 + lane&31) and reduction rows mrow0 + (lane>>5)*8 + 0..7
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* __restrict__ S, int mrow0, int col0, int lane) {
  const int g16 = lane >> 4, i = lane & 15;
  const int row = mrow0 + (g16 >> 1) * 8 + (i >> 2);
  const int col = col0 + (g16 & 1) * 16 + 4 * (i & 3);
  // rows `row` and `row + 4` share (row & 3), hence the same swizzle
  const unsigned char* p = S + row * TN_PITCH + ((((col >> 3) ^ tswz(row))) << 4) + (col & 7) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * TN_PITCH));
  union { s16x4 h[2]; bf16x8 b; } c;
  c.h[0] = lo;
  c.h[1] = hi;
  return c.b;
}

// grid: tiles * splits.  C[n][k] += sum_m A[m][n] * B[m][k]; ATOMIC = 1 when several splits accumulate into C.
// Measured alternatives (r01, slower, removed): 192 x 192 tiles with three stages and 12 waves (329 vs 305 us per layer) and
// 96 x 192 tiles filling 512 workgroups exactly (376 us): with two transpose reads per fragment the 64 x 64 wave tile's
// read-per-MFMA ratio matters more here than tile quantisation.  32-row stages (32 KB of LDS, three workgroups per CU): 299-321 us against 283.
template <bool ATOMIC, bool GLDS, int NI = 2>
__global__ __launch_bounds__(512 / NI) void gemm_bf16_tn_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                           float* __restrict__ C, long ldc, int M, int N, int K, int rows_per_split,
                                                           float* __restrict__ dbias) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TN_BR * TN_PITCH];   // [buf][A|B][64][256 B] = 64 KB
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  static_assert(GLDS || NI == 2, "register-staged fallback is written for 256 threads");
  constexpr int NW = 8 / NI;                     // NI = 2: 4 waves of 64 x 64; NI = 1: 8 waves of 32 x 64
  const int wn = wid >> 1, wk = wid & 1, half = lane >> 5, l31 = lane & 31;
  // 1-D grid, XCD-contiguous chunks: the workgroups one XCD runs together belong to the same token range (split), so its
  // L2 holds each dY / X panel once while all (n, k) tiles of that split consume them
  const int nbn = (N + 127) / 128, tiles = nbn * ((K + 127) / 128);
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int split = bid / tiles, tile = bid - split * tiles;
  // tile order inside a split: the XCD's contiguous chunk should span the NARROWER operand completely and only a slice of the wider
  // one (N = 3072, K = 768: a chunk of 6 k-tiles x 9 n-tiles re-reads X per XCD, not the 4x larger dY)
  const int nkt = (K + 127) / 128;
  const int ntile = nbn > nkt ? tile / nkt : tile % nbn, ktile = nbn > nkt ? tile % nkt : tile / nbn;
  const int n0 = ntile * 128, k0 = ktile * 128;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  f32x16 acc[NI][2];   // [n block][k block]
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient db[n] = sum_m dY[m][n] rides along as dY^T . 1: one extra MFMA per dY fragment against an all-ones operand
  // (every column of the result holds the column sum).  The reduction steps are dealt round-robin to the k-tiles that share
  // this n-tile, so every workgroup carries the same small share of the extra work (no slow tail).
  const bool do_bias = dbias != nullptr && wk == 0;
  f32x16 accb[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
  bf16x8 ones;
  {
    union { unsigned int u[4]; bf16x8 b; } c;
    c.u[0] = c.u[1] = c.u[2] = c.u[3] = CLIMB_H16_ONE_X2;
    ones = c.b;
  }
  u32x4 ra[4], rb[4];
  if constexpr (GLDS) {
    tn_glds<NW>(A, lda, mbeg, n0, N, smem);
    tn_glds<NW>(B, ldb, mbeg, k0, K, smem + TN_BR * TN_PITCH);
  } else {
    tn_load(ra, A, lda, mbeg, n0, mend, N);
    tn_load(rb, B, ldb, mbeg, k0, mend, K);
    tn_store(ra, smem);
    tn_store(rb, smem + TN_BR * TN_PITCH);
  }
  __syncthreads();
  const int nt = (mend - mbeg + TN_BR - 1) / TN_BR;
  for (int t = 0; t < nt; ++t) {
    const unsigned char* As = smem + (t & 1) * (2 * TN_BR * TN_PITCH);
    const unsigned char* Bs = As + TN_BR * TN_PITCH;
    if (t + 1 < nt) {
      if constexpr (GLDS) {
        unsigned char* An = smem + ((t + 1) & 1) * (2 * TN_BR * TN_PITCH);
        tn_glds<NW>(A, lda, mbeg + (t + 1) * TN_BR, n0, N, An);
        tn_glds<NW>(B, ldb, mbeg + (t + 1) * TN_BR, k0, K, An + TN_BR * TN_PITCH);
      } else {
        tn_load(ra, A, lda, mbeg + (t + 1) * TN_BR, n0, mend, N);
        tn_load(rb, B, ldb, mbeg + (t + 1) * TN_BR, k0, mend, K);
      }
    }
#pragma unroll
    for (int ms = 0; ms < TN_BR / 16; ++ms) {
      bf16x8 fa[NI], fb[2];
#pragma unroll
      for (int i = 0; i < NI; ++i) fa[i] = tn_frag(As, ms * 16, wn * (32 * NI) + i * 32, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[j] = tn_frag(Bs, ms * 16, wk * 64 + j * 32, lane);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = CLIMB_MFMA_H16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (do_bias && (t % nkt) == ktile) {
#pragma unroll
        for (int i = 0; i < NI; ++i) accb[i] = CLIMB_MFMA_H16(fa[i], ones, accb[i], 0, 0, 0);
      }
    }
    if constexpr (!GLDS) if (t + 1 < nt) {
      unsigned char* An = smem + ((t + 1) & 1) * (2 * TN_BR * TN_PITCH);
      tn_store(ra, An);
      tn_store(rb, An + TN_BR * TN_PITCH);
    }
    __syncthreads();
  }
  // D layout: col = lane&31 -> k, row = (r&3) + 8*(r>>2) + 4*half -> n ; a wave-instruction writes 2 rows x 32 consecutive floats
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + l31;
      if (k >= K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (32 * NI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n >= N) continue;
        float* cp = C + (long)n * ldc + k;
        if (ATOMIC) atomicAdd(cp, acc[i][j][r]);
        else *cp += acc[i][j][r];
      }
    }
  if (do_bias && l31 == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (32 * NI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < N) atomicAdd(dbias + n, accb[i][r]);       // several k-tiles (and splits) contribute
      }
  }
}


// Registers (ptr, bytes) as the scratch the persistent weight-gradient kernel writes its split partial sums to (NULL / 0 unregisters).
// The buffer is the caller's, must outlive every later climb_gemm_bf16_tn launch that uses it, and is used in stream order only.
extern "C" int climb_set_tn_workspace(void* ptr, long bytes) {
  if (bytes < 0 || (((uintptr_t)ptr) & 15)) return CLIMB_EINVAL;
  climb_tnp_set_workspace(ptr, bytes);
  return CLIMB_OK;
}

// Registers the scratch of the split-along-K NT kernel (gemm_bf16_ntp.hip: partial accumulator tiles + hand-over flags); NULL / 0
// unregisters.  climb_nt_workspace_bytes() says how much it wants; without it the N = 768 GEMMs run on 192 of the 256 CUs as before.
extern "C" int climb_nt_workspace_bytes(void) { return 192 * 96 * 512 * 4 + 4096; }
extern "C" int climb_set_nt_workspace(void* ptr, long bytes) {
  if (bytes < 0 || (((uintptr_t)ptr) & 15)) return CLIMB_EINVAL;
  climb_ntsk_set_workspace(ptr, bytes);
  return CLIMB_OK;
}

// C[N,K] (fp32, ldc) += A[M,N]^T B[M,K]; A, B bf16 row-major (lda, ldb).  N % 8 == 0, K % 8 == 0.
// dbias (optional, fp32 [N]) += column sums of A  (the bias gradient of the same linear layer, fused)
extern "C" int climb_gemm_bf16_tn(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K, float* dbias,
                                  void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (N % 8) || (K % 8) || (lda % 8) || (ldb % 8) || !al16p(A) || !al16p(B)) return CLIMB_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (g_tn_p == 1 && (long)N * K >= 256L * 192 && M >= 2048 && (((uintptr_t)C) & 3) == 0) {
    int rc = climb_tnp_launch((const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, dbias, st);
    if (rc == CLIMB_OK) { LAUNCH_CHECK(); return CLIMB_OK; }
    if (rc != CLIMB_EUNSUPPORTED) return rc;
  }
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  // small outputs (768 x 768: 36 tiles) are atomics-bound: fewer, longer splits (measured 37 vs 44 us at 360 vs 504 workgroups)
  // ... and sliver outputs (the adapters' 48 x 768 / 768 x 48: 6 tiles) want ~96 workgroups (measured 29 / 16 us against 58 / 22 at 384)
  const int target = tiles <= 8 ? g_tn_target * 3 / 16 : (tiles <= 48 ? g_tn_target * 3 / 4 : g_tn_target);
  int splits = target / tiles;                                    // as many token-range splits as fit ONE round of 2 workgroups per CU
                                                                  // (measured: 432 workgroups 579 TF vs 576 workgroups 436 TF)
  const int max_splits = (M + 4 * TN_BR - 1) / (4 * TN_BR);       // at least 4 LDS tiles of work per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rows = (M + splits - 1) / splits;
  rows = (rows + TN_BR - 1) / TN_BR * TN_BR;
  splits = (M + rows - 1) / rows;
  dim3 grid(tiles * splits), blk(256);
  const bool glds = (M % TN_BR) == 0 && N >= 8 && K >= 8;      // the DMA path cannot zero-fill a ragged reduction tail
#define TN_LAUNCH(AT, GL) hipLaunchKernelGGL((gemm_bf16_tn_kernel<AT, GL>), grid, blk, 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, rows, dbias)
  if (glds && g_tn_waves == 8) {
    if (splits > 1) hipLaunchKernelGGL((gemm_bf16_tn_kernel<true, true, 1>), grid, dim3(512), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, rows, dbias);
    else hipLaunchKernelGGL((gemm_bf16_tn_kernel<false, true, 1>), grid, dim3(512), 0, st, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, M, N, K, rows, dbias);
  } else if (splits > 1) { if (glds) TN_LAUNCH(true, true); else TN_LAUNCH(true, false); }
  else { if (glds) TN_LAUNCH(false, true); else TN_LAUNCH(false, false); }
#undef TN_LAUNCH
  LAUNCH_CHECK();
  return CLIMB_OK;
}
