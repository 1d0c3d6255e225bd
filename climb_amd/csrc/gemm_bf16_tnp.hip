// bf16 TN GEMM (weight gradients): C[N,K] += A[M,N]^T . B[M,K], reduction over the token dimension M, on the phase structure of
// gemm_bf16_ntp.hip: 256 (n) x 64 NI (k) output tiles, one workgroup of 8 waves per CU, NI phases per 64-token reduction tile with
// the staging schedule and the compile-time-derived vmcnt counts of gemm_bf16_phase.h.
//
// What differs from the NT kernel is only how the operands sit in memory: both are token-major, so an LDS image is [64 tokens][cols]
// and the MFMA fragments (output index x 8 consecutive tokens) are gathered with ds_read_b64_tr_b16, the gfx950 LDS transpose read
// (two reads per fragment), exactly as in the 128 x 128 weight-gradient kernel of gemm_bf16.hip.
//   A image  [64 tokens][256 n]  512-B rows = 32 KB; staging units A0 / A1 = tokens 0..31 / 32..63 (2 DMA instructions per lane each);
//            16-B chunks XOR-swizzled by (token & 3) << 2: the 4 token rows a transpose read touches sit in 4 different bank quarters.
//   B unit p [64 tokens][2 x 32 k] 128-B rows = 8 KB: the columns of k-block p of both wave columns (1 DMA instruction per lane);
//            chunks XOR-swizzled by ((token >> 1) & 1) << 2 for the same reason at this pitch.
// Waves 4 (n) x 2 (k); wave (wr, wc) owns n rows wr*64..+64 (2 n-blocks) x k columns wc*32 NI..+32 NI (NI k-blocks); phase p multiplies
// the A fragments of the whole reduction tile (read in phase 0, kept) with the fragments of k-block p.
//
// Work decomposition: (output tile, token-range split) pairs, splits chosen so that the pairs fill ONE round of 256 CUs; the partial
// sums of a tile's splits meet in C through fp32 atomics (global_atomic_add_f32), as before -- but 7 - 9 splits of 256-wide tiles move
// half the atomic traffic of the 12 - 18 splits the 128 x 128 kernel needs.
// The bias gradient db[n] = sum_m A[m][n] rides along on the VALU: the A fragments a lane holds ARE 8 tokens of one column n, so the
// lanes of wave column 0 add them up during the reduction tiles dealt to their workgroup (round-robin over the k-tiles of an n-row
// of tiles, so every workgroup carries the same small share).
#include <vector>
#include "gemm_bf16_phase.h"

__device__ __forceinline__ int tnp_aswz(int row) { return (row & 3) << 2; }
__device__ __forceinline__ int tnp_bswz(int row) { return ((row >> 1) & 1) << 2; }

// SPLIT (r06, split.hip): the operands are (hi, lo) plane pairs stacked along the token dimension ([2 M, .]: the lo plane follows the hi plane), and the
// reduction walks 3 M / 64 VIRTUAL tiles -- virtual tile v = t0 + t is product v % 3 of token tile v / 3: A {hi, hi, lo} against B {hi, lo, hi}, i.e. row tile
// v / 3 (+ mt for the lo plane; mt = M / 64) -- the three products of a token tile follow each other, so the repeated hi images come from the XCD's L2.
template <int NI, bool SPLIT = false>
struct TnpStage {
  unsigned a_off[4];       // per lane: byte offset (from A + first token row of the split) of its 4 DMA sources of the A image
  unsigned b_off[NI];      // per lane: byte offset (from B + first token row of the split) of its DMA source of B unit p
  unsigned wid;
  int mt, t0;              // SPLIT only
};

template <int NI, int P, bool D1, bool D2, int K_ = 0, bool SPLIT = false>
__device__ __forceinline__ void tnp_issue(const TnpStage<NI, SPLIT>& sg, const unsigned char* __restrict__ A, long a_tile, const unsigned char* __restrict__ B,
                                          long b_tile, unsigned char* __restrict__ smem, int T) {
  if constexpr (K_ < NTP_MAXI) {
    constexpr int BUF = NTP_A_BYTES + NI * NTP_B_UNIT;
    constexpr int e = ntp_sched(NI, P, K_);
    if constexpr (e >= 0) {
      constexpr int unit = e / 4, d = e % 4;
      if constexpr ((d == 1 && D1) || (d == 2 && D2)) {
        const int t = T + d;
        unsigned char* buf = smem + (t & 1) * BUF;
        int ta = t, tb = t;
        if constexpr (SPLIT) {
          const unsigned v = (unsigned)(sg.t0 + t), m = v / 3u, ph = v - 3u * m;
          ta = (int)m + (ph == 2u ? sg.mt : 0);
          tb = (int)m + (ph == 1u ? sg.mt : 0);
        }
        if constexpr (unit < 2) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(A + (long)ta * a_tile + sg.a_off[unit * 2 + i]),
                                             (lds_void_t*)(buf + unit * (NTP_A_BYTES / 2) + (sg.wid * 2 + i) * 1024), 16, 0, 0);
        } else {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(B + (long)tb * b_tile + sg.b_off[unit - 2]),
                                           (lds_void_t*)(buf + NTP_A_BYTES + (unit - 2) * NTP_B_UNIT + sg.wid * 1024), 16, 0, 0);
        }
      }
    }
    tnp_issue<NI, P, D1, D2, K_ + 1>(sg, A, a_tile, B, b_tile, smem, T);
  }
}

// 8-token MFMA operand: two transpose reads (token rows r..r+3 and r+4..r+7 of the lane's 16-lane block)
template <int PITCH>
__device__ __forceinline__ bf16x8 tnp_frag(const unsigned char* __restrict__ p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * PITCH));
  union { s16x4 h[2]; bf16x8 b; } c;
  c.h[0] = lo;
  c.h[1] = hi;
  return c.b;
}
__device__ __forceinline__ float tnp_sum8(bf16x8 v) {
  union { bf16x8 b; unsigned u[4]; } c;
  c.b = v;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += h16lo_to_f32(c.u[i]) + h16hi_to_f32(c.u[i]);
  return s;
}

template <int NI, int P, bool D1, bool D2, int W, bool SPLIT = false>
__device__ __forceinline__ void tnp_phase(f32x16 (&acc)[NI][2], bf16x8 (&a)[2][4], float (&bsum)[2], bool bias_tile, const TnpStage<NI, SPLIT>& sg,
                                          const unsigned char* __restrict__ A, long a_tile, const unsigned char* __restrict__ B, long b_tile,
                                          unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[2], unsigned boff) {
  constexpr int BUF = NTP_A_BYTES + NI * NTP_B_UNIT;
  const unsigned char* buf = smem + (T & 1) * BUF;
  bf16x8 b[4];
#pragma unroll
  for (int ms = 0; ms < 4; ++ms) b[ms] = tnp_frag<128>(buf + NTP_A_BYTES + P * NTP_B_UNIT + ms * (16 * 128) + boff);
  if constexpr (P == 0) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) a[j][ms] = tnp_frag<512>(buf + ms * (16 * 512) + aoff[j]);
  }
  tnp_issue<NI, P, D1, D2>(sg, A, a_tile, B, b_tile, smem, T);
  if constexpr (W >= 0) wait_vmcnt<(W < 0 ? 0 : W)>();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ms = 0; ms < 4; ++ms)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[P][j] = CLIMB_MFMA_H16(a[j][ms], b[ms], acc[P][j], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  if constexpr (P == 1) {          // bias share of this reduction tile (VALU, beside the other wave group's MFMAs)
    if (bias_tile) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) bsum[j] += tnp_sum8(a[j][ms]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <int NI, int MODE, int P = 0, bool SPLIT = false>
__device__ __forceinline__ void tnp_ktile(f32x16 (&acc)[NI][2], bf16x8 (&a)[2][4], float (&bsum)[2], bool bias_tile, const TnpStage<NI, SPLIT>& sg,
                                          const unsigned char* __restrict__ A, long a_tile, const unsigned char* __restrict__ B, long b_tile,
                                          unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[2], unsigned boff) {
  if constexpr (P < NI) {
    constexpr int W = ntp_wait(NI, 6, MODE == 0 ? 2 : (MODE == 1 ? 4 : 5), P);
    tnp_phase<NI, P, MODE <= 1, MODE == 0, W>(acc, a, bsum, bias_tile, sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
    tnp_ktile<NI, MODE, P + 1>(acc, a, bsum, bias_tile, sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
  }
}
template <int NI, int P = 0, bool SPLIT = false>
__device__ __forceinline__ void tnp_prologue(const TnpStage<NI, SPLIT>& sg, const unsigned char* __restrict__ A, long a_tile, const unsigned char* __restrict__ B,
                                             long b_tile, unsigned char* __restrict__ smem) {
  if constexpr (P < NI) {
    tnp_issue<NI, P, false, true>(sg, A, a_tile, B, b_tile, smem, -2);
    tnp_prologue<NI, P + 1>(sg, A, a_tile, B, b_tile, smem);
  } else if constexpr (P < 2 * NI) {
    tnp_issue<NI, P - NI, true, true>(sg, A, a_tile, B, b_tile, smem, -1);
    tnp_prologue<NI, P + 1>(sg, A, a_tile, B, b_tile, smem);
  }
}

// grid: min(tiles * splits, 256 * k) workgroups walking (tile, split) pairs.  M % 64 == 0, rows_per_split % 64 == 0, every split >= 128 rows.
template <int NI>
__global__ __launch_bounds__(512) void gemm_bf16_tnp_kernel(const bf16_t* __restrict__ Ag, long lda, const bf16_t* __restrict__ Bg, long ldb,
                                                            float* __restrict__ C, long ldc, int M, int N, int K, int rows_per_split, int splits,
                                                            float* __restrict__ dbias, float* __restrict__ slab) {
  constexpr int BK_ = 64 * NI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2, half = lane >> 5, l31 = lane & 31;
  const int nbn = (N + NTP_BM - 1) / NTP_BM, nbk = (K + BK_ - 1) / BK_, tiles = nbn * nbk, nwork = tiles * splits;
  // transpose-read lane geometry: 16-lane block g16 reads token rows (g16>>1)*8 + (i>>2) [+4], columns (g16&1)*16 + 4*(i&3) .. +3
  const int g16 = lane >> 4, i16 = lane & 15, trow = (g16 >> 1) * 8 + (i16 >> 2), tcol = (g16 & 1) * 16 + 4 * (i16 & 3);
  unsigned aoff[2], boff;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wr * 64 + j * 32 + tcol;
    aoff[j] = trow * 512 + (((col >> 3) ^ tnp_aswz(trow)) << 4) + (col & 7) * 2;
  }
  {
    const int col = wc * 32 + tcol;
    boff = trow * 128 + (((col >> 3) ^ tnp_bswz(trow)) << 4) + (col & 7) * 2;
  }
  const long a_tile = 64 * lda * 2, b_tile = 64 * ldb * 2;          // bytes from one 64-token reduction tile to the next
  for (int id = blockIdx.x; id < nwork; id += gridDim.x) {
    // XCD-contiguous order: the workgroups an XCD runs together share a token range (split), so its L2 holds each dY / X panel once
    int bid;
    {
      const int q = nwork / 8, rr = nwork % 8, xcd = id % 8, idx = id / 8;
      bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    }
    const int split = bid / tiles, tile = bid - split * tiles;
    const int tn = nbn > nbk ? tile / nbk : tile % nbn, tk = nbn > nbk ? tile % nbk : tile / nbn;
    const int n0 = tn * NTP_BM, k0 = tk * BK_;
    const int mbeg = split * rows_per_split, mend = min(M, mbeg + rows_per_split);
    const int nk = (mend - mbeg) / 64;                                 // >= 2 (launcher)
    const unsigned char* A = reinterpret_cast<const unsigned char*>(Ag) + (long)mbeg * lda * 2;
    const unsigned char* B = reinterpret_cast<const unsigned char*>(Bg) + (long)mbeg * ldb * 2;
    TnpStage<NI> sg;
    sg.wid = wid;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane, r = u * 32 + (slot >> 5), c = (slot & 31) ^ tnp_aswz(r);
        int col = n0 + c * 8;
        col = col < N ? col : N - 8;                                   // clamped columns are computed but never stored
        sg.a_off[u * 2 + i] = (unsigned)(((long)r * lda + col) * 2);
      }
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int slot = wid * 64 + lane, r = slot >> 3, c = (slot & 7) ^ tnp_bswz(r);
      int col = k0 + (c >> 2) * (32 * NI) + p * 32 + (c & 3) * 8;
      col = col < K ? col : K - 8;
      sg.b_off[p] = (unsigned)(((long)r * ldb + col) * 2);
    }
    f32x16 acc[NI][2];   // [k block][n block]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = dbias != nullptr && wc == 0;
    bf16x8 a[2][4];
    tnp_prologue<NI>(sg, A, a_tile, B, b_tile, smem);
    wait_vmcnt<ntp_wait(NI, 6, -1, NI - 1)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int T = 0;
    for (; T + 2 < nk; ++T) tnp_ktile<NI, 0>(acc, a, bsum, do_bias && (T % nbk) == tk, sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
    tnp_ktile<NI, 1>(acc, a, bsum, do_bias && (T % nbk) == tk, sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
    tnp_ktile<NI, 2>(acc, a, bsum, do_bias && ((T + 1) % nbk) == tk, sg, A, a_tile, B, b_tile, smem, T + 1, aoff, boff);
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // D layout: col = lane & 31 -> k, row = (r & 3) + 8 * (r >> 2) + 4 * half -> n: a wave-instruction adds 2 rows x 32 consecutive floats
#pragma unroll
    for (int p = 0; p < NI; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = k0 + wc * (32 * NI) + p * 32 + l31;
        if (k >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wr * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (n >= N) continue;
          if (slab) {                                   // this split's partial sum, plain stores: [split][N][K], reduced by tnp_reduce_kernel
            slab[((long)split * N + n) * K + k] = acc[p][j][r];
          } else {
            float* cp = C + (long)n * ldc + k;
            if (splits > 1) atomicAdd(cp, acc[p][j][r]);
            else *cp += acc[p][j][r];
          }
        }
      }
    if (do_bias) {
      // a lane summed the tokens 8*(lane>>5) .. +7 of every 16 for column n = lane & 31 of each n-block: fold the two lane halves
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);
        const int n = n0 + wr * 64 + j * 32 + l31;
        if (half == 0 && n < N) atomicAdd(dbias + n, s);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // the next work item's prologue overwrites the buffers
    __builtin_amdgcn_sched_barrier(0);
  }
}

// C[n][k] += sum over splits of slab[s][n][k]   (one float4 per thread-iteration; K % 4 == 0)
__global__ __launch_bounds__(256) void tnp_reduce_kernel(const float* __restrict__ slab, float* __restrict__ C, long ldc, int N, int K, int splits) {
  const long per = (long)N * K, n4 = per / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int n = (int)(e / K), k = (int)(e - (long)n * K);
    float4 a = ld4(C + (long)n * ldc + k);
    for (int s = 0; s < splits; ++s) {
      const float4 v = ld4(slab + s * per + e);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4(C + (long)n * ldc + k, a);
  }
}

#ifdef TNP_PROBE_REDUCE_ENTRY       // measurement build only (tools/probe/reduce_overlap.py)
extern "C" int climb_probe_tn_reduce(const float* slab, float* C, long ldc, int N, int K, int splits, void* stream) {
  const long n4 = (long)N * K / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(tnp_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, C, ldc, N, K, splits);
  return 0;
}
#endif
static float* g_tn_ws = nullptr;
static long g_tn_ws_bytes = 0;
// Scratch for the split partial sums (the library allocates nothing: the host registers a buffer once; without one, or when a launch
// needs more than it holds, the partial sums meet in C through fp32 atomics instead -- measured 1.3 TB/s against ~5 TB/s for the slabs).
void climb_tnp_set_workspace(void* ptr, long bytes) { g_tn_ws = (float*)ptr; g_tn_ws_bytes = ptr ? bytes : 0; }

template <int NI>
static int tnp_launch_one(int nwg, hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, float* C, long ldc, int M, int N, int K, int rows,
                          int splits, float* dbias) {
  constexpr int LDS = 2 * (NTP_A_BYTES + NI * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tnp_kernel<NI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  float* slab = nullptr;
  if (splits > 1 && (K % 4) == 0 && (ldc % 4) == 0 && (((uintptr_t)C) & 15) == 0 && (long)splits * N * K * 4 <= g_tn_ws_bytes) slab = g_tn_ws;
  hipLaunchKernelGGL((gemm_bf16_tnp_kernel<NI>), dim3(nwg), dim3(512), LDS, st, A, lda, B, ldb, C, ldc, M, N, K, rows, splits, dbias, slab);
#ifdef TNP_PROBE_NO_REDUCE          // measurement build only: what would hiding the reduce launches be worth at most
  if (false) {
#else
  if (slab) {
#endif
    const long n4 = (long)N * K / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(tnp_reduce_kernel, dim3(grid), dim3(256), 0, st, slab, C, ldc, N, K, splits);
  }
  return CLIMB_OK;
}

// C[N,K] (fp32, ldc) += A[M,N]^T B[M,K]; dbias (optional) += column sums of A.  Returns CLIMB_EUNSUPPORTED for shapes it does not take
// (the caller falls back to the 128 x 128 kernel): M % 64, N % 8, K % 8, too few tokens per split, operands of 4 GB or more.
int climb_tnp_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, float* C, long ldc, int M, int N, int K, float* dbias, hipStream_t st) {
  if ((M % 64) || (N % 8) || (K % 8) || N < 8 || K < 8) return CLIMB_EUNSUPPORTED;
  if (((long)M * lda + N) * 2 >= (1L << 32) || ((long)M * ldb + K) * 2 >= (1L << 32)) return CLIMB_EUNSUPPORTED;
  const int nbn = (N + NTP_BM - 1) / NTP_BM;
  // k-tile width: the one whose (tiles x splits) fills the 256 CUs better with whole 64-token reduction tiles
  int best_ni = 0, best_splits = 0, best_rows = 0;
  double best_cost = 1e30;
  for (int ni = 3; ni <= 4; ++ni) {
    const int bk = 64 * ni, tiles = nbn * ((K + bk - 1) / bk);
    int splits = 256 / tiles;
    if (splits < 1) splits = 1;
    const int mt = M / 64;
    if (splits > mt / 2) splits = mt / 2;                 // at least two reduction tiles per split (the pipeline's minimum)
    if (splits < 1) continue;
    const int rows = ((mt + splits - 1) / splits) * 64;
    splits = (M + rows - 1) / rows;
    if (M - (splits - 1) * rows < 128) continue;          // ragged last split shorter than two reduction tiles
    // cost ~ rounds x (reduction tiles per split) x tile width, + the fixed per-work-item cost of fill and atomics
    const int work = tiles * splits, rounds = (work + 255) / 256;
    const double cost = rounds * ((double)(rows / 64) + 6.0) * bk * ((double)(((N + 255) / 256) * 256) / N) * ((double)(((K + bk - 1) / bk) * bk) / K);
    if (cost < best_cost) { best_cost = cost; best_ni = ni; best_splits = splits; best_rows = rows; }
  }
  if (best_ni == 0) return CLIMB_EUNSUPPORTED;
  const int bk = 64 * best_ni, tiles = nbn * ((K + bk - 1) / bk);
  int nwg = tiles * best_splits;
#ifdef TNP_PROBE_GRID            // measurement build only (tools/probe/split_chip.py): cap the persistent grid
  if (nwg > TNP_PROBE_GRID) nwg = TNP_PROBE_GRID;
#endif
  if (nwg > 256) nwg = 256;
  if (best_ni == 4) return tnp_launch_one<4>(nwg, st, A, lda, B, ldb, C, ldc, M, N, K, best_rows, best_splits, dbias);
  return tnp_launch_one<3>(nwg, st, A, lda, B, ldb, C, ldc, M, N, K, best_rows, best_splits, dbias);
}

// ======================================================================================================================================
// Grouped weight gradients (r03): ONE persistent launch for the dW GEMMs of several layers, no token split for whole tiles.
//
// Nothing reads a weight gradient before the optimizer, so the host defers the four dW GEMMs of a layer (dW2, dW1, dWo, dWqkv: 108 output
// tiles of 256 x 256 at ViLT-B's shapes) and hands a group of layers to this kernel: 6 / 12 layers = 648 / 1296 tiles = 2.5 / 5 rounds of
// 256 CUs.  A tile now reduces over ALL tokens inside one workgroup (192 reduction tiles at 12288 tokens instead of ~38 per split), adds
// its sum to C with a plain read-modify-write, and needs neither the 47 MB of split partials per GEMM nor the reduce launch.  What does not
// fill a whole round -- the last `tiles mod 256` tiles -- is cut stream-K style into equal shares of reduction tiles, one per CU, whose
// partial sums meet in C through fp32 atomics (<= 2 partial tiles per CU and launch).
//
// The work list is data: `items` (tile, reduction range) sorted by workgroup, `first[b] .. first[b + 1]` = workgroup b's items;
// climb_tn_grouped_plan() builds it on the host, once per shape (the tables live in HBM and are reused every step).
struct TnGroupProblem {      // 72 bytes; include/climb_hip.h documents this layout
  const bf16_t* A;           // [M, N] token-major (lda)   -- dY
  const bf16_t* B;           // [M, K] token-major (ldb)   -- X
  float* C;                  // [N, K] (ldc), accumulated into
  float* dbias;              // [N] += column sums of A, or NULL
  long lda, ldb, ldc;
  int M, N, K, reserved;
};
struct TnGroupItem { int prob, tn, tk, kt0, kt1, partial, r0, r1; };      // 32 bytes; reduction tiles [kt0, kt1) of 64 tokens, kt1 - kt0 >= 2
// r04: the optimizer in the epilogue.  Nothing reads a weight gradient between the backward and optimizer.step() in the fused training step, and
// this launch is MFMA / power-bound with the memory system half idle: a WHOLE tile (the only writer of its elements) of a problem marked `fused`
// applies AdamW right there -- g = tile sum (+ what the gradient buffer already holds, when the host says it is not zero: the EWC penalty's
// 2 lam F (theta - theta*), REF/cl_algorithms/ewc.py:75-87, or an earlier accumulating backward), reads p, m, v, writes p, m, v, the 16-bit
// shadow [N,K] and the TRANSPOSED shadow [K,N] the input-gradient GEMMs read -- and never writes the gradient.  The flat AdamW pass then skips
// these tensors, the batched shadow transpose too.  A problem may be fused only if ALL its tiles are whole tiles (the host checks the plan).
struct TnGroupOpt {          // 56 bytes, parallel to the problems; include/climb_hip.h documents this layout
  float* p; float* m; float* v;      // [N, K] fp32, leading dimension = the problem's ldc
  bf16_t* s;                         // [N, K] 16-bit shadow, leading dimension ldc
  bf16_t* st;                        // [K, N] transposed 16-bit shadow, leading dimension ldt
  long ldt;
  int fused, pad;
};
// r05: the EWC term in the same epilogue (REF/cl_algorithms/ewc.py:75-87: loss += lam sum F (theta - theta*)^2, so d/dtheta = 2 lam F (theta - theta*)).
// theta* and F are laid out like the encoder range of the flat parameter buffer: element e of p <-> star[e - flat], fisher[e - flat].  The term is
// added to the tile sum in fp32 before the update, and lam F (theta - theta*)^2 of the elements this launch updates is added to *loss (atomics).
struct TnEwc { const float* flat; const float* star; const float* fisher; float* loss; float lam; int pad; };
struct TnAdam { AdamGroup g; float gscale; int grad_dirty; TnEwc ewc; unsigned s_lo_b, st_lo_b; };      // (SPLIT: byte distance from the hi to the lo plane of the straight / transposed shadow)
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// 4 x 4 transpose inside every quad of lanes: register i of lane b <-> register b of lane i (two exchange steps through DPP quad permutes)
__device__ __forceinline__ float tn_dpp_quad(float x, bool hi) {
  const int v = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, hi ? __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true) : __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ void tn_quad_transpose(float (&r)[4], int lane) {
  const bool o1 = lane & 1, o2 = lane & 2;
  float t0 = tn_dpp_quad(o1 ? r[0] : r[1], false), t1 = tn_dpp_quad(o1 ? r[2] : r[3], false);
  if (o1) { r[0] = t0; r[2] = t1; } else { r[1] = t0; r[3] = t1; }
  t0 = tn_dpp_quad(o2 ? r[0] : r[2], true);
  t1 = tn_dpp_quad(o2 ? r[1] : r[3], true);
  if (o2) { r[0] = t0; r[1] = t1; } else { r[2] = t0; r[3] = t1; }
}

// RAGGED: N, K any multiples of 8 (the adapters' 768 x 48 / 48 x 768 gradients ride in the same launch as 256 x 256 tiles whose surplus
// columns are computed on clamped addresses and never stored -- the FLOPs of those GEMMs are nothing, their launches and operand reads were).
template <bool RAGGED, bool FUSED = false, bool EWC = false, bool SPLIT = false>
__global__ __launch_bounds__(512) void gemm_bf16_tn_grouped_kernel(const TnGroupProblem* __restrict__ probs, const TnGroupItem* __restrict__ items,
                                                                   const int* __restrict__ first, const TnGroupOpt* __restrict__ opts = nullptr, TnAdam ad = TnAdam()) {
  constexpr int NI = 4, BK_ = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2, half = lane >> 5, l31 = lane & 31;
  const int g16 = lane >> 4, i16 = lane & 15, trow = (g16 >> 1) * 8 + (i16 >> 2), tcol = (g16 & 1) * 16 + 4 * (i16 & 3);
  unsigned aoff[2], boff;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wr * 64 + j * 32 + tcol;
    aoff[j] = trow * 512 + (((col >> 3) ^ tnp_aswz(trow)) << 4) + (col & 7) * 2;
  }
  {
    const int col = wc * 32 + tcol;
    boff = trow * 128 + (((col >> 3) ^ tnp_bswz(trow)) << 4) + (col & 7) * 2;
  }
  const int it_end = __builtin_amdgcn_readfirstlane(first[blockIdx.x + 1]);
  for (int it = __builtin_amdgcn_readfirstlane(first[blockIdx.x]); it < it_end; ++it) {
    const TnGroupItem item = items[it];
    const TnGroupProblem P = probs[__builtin_amdgcn_readfirstlane(item.prob)];
    const int tn = __builtin_amdgcn_readfirstlane(item.tn), tk = __builtin_amdgcn_readfirstlane(item.tk);
    const int kt0 = __builtin_amdgcn_readfirstlane(item.kt0), nk = __builtin_amdgcn_readfirstlane(item.kt1) - kt0;
    const bool partial = __builtin_amdgcn_readfirstlane(item.partial) != 0;
    const int N = P.N, K = P.K, nbk = (K + BK_ - 1) / BK_;
    const long lda = P.lda, ldb = P.ldb, ldc = P.ldc;
    const int n0 = tn * NTP_BM, k0 = tk * BK_;
    const long a_tile = 64 * lda * 2, b_tile = 64 * ldb * 2;
    const unsigned char* A = reinterpret_cast<const unsigned char*>(P.A) + (SPLIT ? 0L : (long)kt0 * a_tile);
    const unsigned char* B = reinterpret_cast<const unsigned char*>(P.B) + (SPLIT ? 0L : (long)kt0 * b_tile);
    TnpStage<NI, SPLIT> sg;
    sg.wid = wid;
    sg.mt = P.reserved;          // (SPLIT: reduction tiles per phase; the problem's M counts all three)
    sg.t0 = kt0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane, r = u * 32 + (slot >> 5), c = (slot & 31) ^ tnp_aswz(r);
        int col = n0 + c * 8;
        if (RAGGED) col = col < N ? col : N - 8;                        // clamped columns are computed but never stored
        sg.a_off[u * 2 + i] = (unsigned)(((long)r * lda + col) * 2);
      }
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int slot = wid * 64 + lane, r = slot >> 3, c = (slot & 7) ^ tnp_bswz(r);
      int col = k0 + (c >> 2) * (32 * NI) + p * 32 + (c & 3) * 8;
      if (RAGGED) col = col < K ? col : K - 8;
      sg.b_off[p] = (unsigned)(((long)r * ldb + col) * 2);
    }
    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const bool do_bias = P.dbias != nullptr && wc == 0;
    const int tb = (tk - kt0 % nbk + nbk) % nbk;        // reduction tile T of this item is dealt to k-tile column (kt0 + T) % nbk
    bf16x8 a[2][4];
    tnp_prologue<NI>(sg, A, a_tile, B, b_tile, smem);
    wait_vmcnt<ntp_wait(NI, 6, -1, NI - 1)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int T = 0;
    // (SPLIT: the column sums of A are those of its hi and lo planes once each -- product 1 of a token tile re-reads the hi plane and is left out; token
    // tiles, not virtual tiles, are dealt to the k-tile columns)
    auto bias_at = [&](int t_) {
      if constexpr (SPLIT) {
        const unsigned v = (unsigned)(kt0 + t_), m = v / 3u;
        return do_bias && (int)(m % (unsigned)nbk) == tk && (v - 3u * m) != 1u;
      } else {
        return do_bias && (t_ % nbk) == tb;
      }
    };
    for (; T + 2 < nk; ++T) tnp_ktile<NI, 0>(acc, a, bsum, bias_at(T), sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
    tnp_ktile<NI, 1>(acc, a, bsum, bias_at(T), sg, A, a_tile, B, b_tile, smem, T, aoff, boff);
    tnp_ktile<NI, 2>(acc, a, bsum, bias_at(T + 1), sg, A, a_tile, B, b_tile, smem, T + 1, aoff, boff);
    if (grp == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    float* Cb = P.C + (long)(n0 + wr * 64 + 4 * half) * ldc + k0 + wc * (32 * NI) + l31;
    if (RAGGED) {          // element guards; tiles this narrow are few and tiny next to their 192-tile reductions
      const int nb = n0 + wr * 64 + 4 * half, kb = k0 + wc * (32 * NI) + l31;
#pragma unroll
      for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dn = j * 32 + (r & 3) + 8 * (r >> 2);
            if (nb + dn < N && kb + p * 32 < K) {
              float* cp = Cb + (long)dn * ldc + p * 32;
              if (partial) atomicAdd(cp, acc[p][j][r]);
              else *cp += acc[p][j][r];
            }
          }
    } else if (partial) {
#pragma unroll
      for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) atomicAdd(Cb + (long)(j * 32 + (r & 3) + 8 * (r >> 2)) * ldc + p * 32, acc[p][j][r]);
    } else if (FUSED && opts[__builtin_amdgcn_readfirstlane(item.prob)].fused) {
      // whole tile of a fused problem: AdamW here.  Element (p, j, r) of this lane = weight [n][k], n = n0 + wr*64 + 4*half + j*32 + 8*(r>>2) + (r&3),
      // k = k0 + wc*128 + p*32 + l31: fp32 accesses are 128-byte rows of 32 lanes, the transposed shadow gets 4 consecutive n (8 bytes) per lane
      const TnGroupOpt O = opts[__builtin_amdgcn_readfirstlane(item.prob)];
      // MUBUF addressing: one per-lane offset (elements) + a scalar offset per element: no 64-bit address arithmetic in vector registers (the flat
      // form of this epilogue spilled 1.1 KB per lane next to the 128 accumulators)
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)O.p, 0, 0x7fffffff, 0x00020000), rm = __builtin_amdgcn_make_buffer_rsrc((void*)O.m, 0, 0x7fffffff, 0x00020000),
                                   rv = __builtin_amdgcn_make_buffer_rsrc((void*)O.v, 0, 0x7fffffff, 0x00020000), rg = __builtin_amdgcn_make_buffer_rsrc((void*)P.C, 0, 0x7fffffff, 0x00020000),
                                   rs = __builtin_amdgcn_make_buffer_rsrc((void*)O.s, 0, 0x7fffffff, 0x00020000), rt = __builtin_amdgcn_make_buffer_rsrc((void*)O.st, 0, 0x7fffffff, 0x00020000);
      // Memory shape.  In the accumulator layout a lane owns ONE k (l31) and four consecutive n per register group: 4-byte accesses, 7 per element --
      // measured +0.41 ms on the launch (instruction-bound).  A 4 x 4 transpose inside every quad of lanes (two DPP exchange steps) turns a register
      // group into one n (4 half + 8 q + l31 % 4) x four consecutive k (4 (l31 / 4) ..): p, m, v (and the gradient term) move as 16-byte accesses, 8 lanes
      // a 128-byte row; the update runs in that layout, the 16-bit shadow is stored from it (8 bytes), and the new weights are transposed back for
      // the [K,N] shadow (four consecutive n: 8 bytes).
      const int b4 = l31 & 3, a8 = l31 >> 2;
      const unsigned le = (unsigned)((4 * half + b4) * (int)ldc + 4 * a8);                                  // lane part of the [N,K] element index (transposed registers)
      const unsigned lt = (unsigned)(l31 * (int)O.ldt + 4 * half) * 2u;                                     // lane part of the [K,N] byte offset (accumulator layout)
      const unsigned se0 = (unsigned)((n0 + wr * 64) * (int)ldc + k0 + wc * (32 * NI));                     // uniform parts
      const unsigned st0 = (unsigned)((k0 + wc * (32 * NI)) * (int)O.ldt + n0 + wr * 64) * 2u;
      const float isb2 = rsqrtf(ad.g.bc2), step = ad.g.lr / ad.g.bc1;
      const bool dirty = ad.grad_dirty != 0;
      __amdgpu_buffer_rsrc_t rstar = rp, rfis = rp;
      if constexpr (EWC) {
        const long delta = O.p - ad.ewc.flat;          // this matrix inside the encoder range
        rstar = __builtin_amdgcn_make_buffer_rsrc((void*)(ad.ewc.star + delta), 0, 0x7fffffff, 0x00020000);
        rfis = __builtin_amdgcn_make_buffer_rsrc((void*)(ad.ewc.fisher + delta), 0, 0x7fffffff, 0x00020000);
      }
      const float two_lam = 2.f * ad.ewc.lam;
      float ewc_sum = 0.f;
#pragma unroll
      for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {          // one 32 x 32 block: every load before the first store (a load behind a store drains the store queue)
          f32x4 pv[4], mv[4], vv[4], gv[4];
          if constexpr (EWC) {
            // first p, theta*, F -> the term (and its share of the penalty's value); only then m, v: the working set stays that of the plain epilogue
            // (all six operands at once spilled 1 KB per lane next to the 128 accumulators).  Every load still precedes the block's first store.
            f32x4 sv[4], fv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned so = (se0 + (unsigned)((j * 32 + 8 * q) * (int)ldc + p * 32)) * 4u;
              pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, le * 4u, so, 0));
              sv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rstar, le * 4u, so, 0));
              fv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rfis, le * 4u, so, 0));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float d = pv[q][i] - sv[q][i], fd = fv[q][i] * d;
                ewc_sum = fmaf(fd, d, ewc_sum);
                gv[q][i] = two_lam * fd;          // the term, in the transposed-register layout the update runs in
              }
            __builtin_amdgcn_sched_barrier(0);          // (the scheduler otherwise hoists the loads below above the term: all six operands live at once)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned so = (se0 + (unsigned)((j * 32 + 8 * q) * (int)ldc + p * 32)) * 4u;
              mv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, le * 4u, so, 0));
              vv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, le * 4u, so, 0));
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned so = (se0 + (unsigned)((j * 32 + 8 * q) * (int)ldc + p * 32)) * 4u;
              pv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, le * 4u, so, 0));
              mv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, le * 4u, so, 0));
              vv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, le * 4u, so, 0));
              gv[q] = dirty ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, le * 4u, so, 0)) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float g4[4] = {acc[p][j][4 * q], acc[p][j][4 * q + 1], acc[p][j][4 * q + 2], acc[p][j][4 * q + 3]};
            tn_quad_transpose(g4, l31);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float pp = pv[q][i], mm = mv[q][i], v2 = vv[q][i];
              adamw_update(pp, mm, v2, (gv[q][i] + g4[i]) * ad.gscale, ad.g, isb2, step);
              pv[q][i] = pp; mv[q][i] = mm; vv[q][i] = v2;
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // Every register-side step (packs, the transposes back) BEFORE the 16-byte stores, and wait states behind them: a VALU write to a data
            // register of a buffer_store_dwordx4 in the slot right after it reached memory instead of the stored value (r06, measured: m / v elements
            // holding the transpose's select of two p values, 80 of 589 824 in one matrix; the compiler's hazard table exempts stores with an SGPR offset).
            const unsigned so = se0 + (unsigned)((j * 32 + 8 * q) * (int)ldc + p * 32);
            const unsigned sto = st0 + (unsigned)((p * 32) * (int)O.ldt + j * 32 + 8 * q) * 2u;
            u32x2_t w = {pack_bf16x2(pv[q][0], pv[q][1]), pack_bf16x2(pv[q][2], pv[q][3])};
            float t4[4] = {pv[q][0], pv[q][1], pv[q][2], pv[q][3]};
            tn_quad_transpose(t4, l31);          // back: four consecutive n of this lane's k
            u32x2_t wt = {pack_bf16x2(t4[0], t4[1]), pack_bf16x2(t4[2], t4[3])};
            u32x2_t wl = {0u, 0u}, wlt = {0u, 0u};
            if constexpr (SPLIT) {          // r06: the lo planes of both shadows (lo = rn16(w - hi), split.hip), the same two layouts
              float l4[4] = {pv[q][0] - h16lo_to_f32(w[0]), pv[q][1] - h16hi_to_f32(w[0]), pv[q][2] - h16lo_to_f32(w[1]), pv[q][3] - h16hi_to_f32(w[1])};
              wl = (u32x2_t){pack_bf16x2(l4[0], l4[1]), pack_bf16x2(l4[2], l4[3])};
              tn_quad_transpose(l4, l31);
              wlt = (u32x2_t){pack_bf16x2(l4[0], l4[1]), pack_bf16x2(l4[2], l4[3])};
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv[q]), rp, le * 4u, so * 4u, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mv[q]), rm, le * 4u, so * 4u, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv[q]), rv, le * 4u, so * 4u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(w, rs, le * 2u, so * 2u, 0);
            __builtin_amdgcn_raw_buffer_store_b64(wt, rt, lt, sto, 0);
            if constexpr (SPLIT) {
              __builtin_amdgcn_raw_buffer_store_b64(wl, rs, le * 2u, so * 2u + ad.s_lo_b, 0);
              __builtin_amdgcn_raw_buffer_store_b64(wlt, rt, lt, sto + ad.st_lo_b, 0);
            }
            asm volatile("s_nop 3");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if constexpr (EWC) {
        ewc_sum = wave_sum(ewc_sum);
        if (lane == 0) atomicAdd(ad.ewc.loss, ad.ewc.lam * ewc_sum);
      }
    } else {
      // whole tile: the only writer of these elements in this launch.  All 16 loads of a (p, j) block are issued before its first store
      // (a load behind a store drains the store queue on gfx9: one vmcnt for both kinds)
#pragma unroll
      for (int p = 0; p < NI; ++p)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float old[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) old[r] = Cb[(long)(j * 32 + (r & 3) + 8 * (r >> 2)) * ldc + p * 32];
#pragma unroll
          for (int r = 0; r < 16; ++r) Cb[(long)(j * 32 + (r & 3) + 8 * (r >> 2)) * ldc + p * 32] = old[r] + acc[p][j][r];
        }
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float s = bsum[j] + __shfl_xor(bsum[j], 32, 64);
        if (half == 0 && (!RAGGED || n0 + wr * 64 + j * 32 + l31 < N)) atomicAdd(P.dbias + n0 + wr * 64 + j * 32 + l31, s);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // the next item's prologue overwrites the buffers
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Host-side planner (no device work).  M[p] % 128 == 0 (whole 64-token reduction tiles, an even number of them, so that every cut of the
// stream-K tail leaves >= 2 tiles on both sides), N[p] % 8 == 0, K[p] % 8 == 0 (anything that is not a multiple of 256 needs the launch's
// `ragged` flag).  Workgroup b of `nwg` (a multiple of 8: XCD = b % 8)
// gets items first[b] .. first[b + 1] - 1.  Returns the number of items written, CLIMB_EINVAL for shapes outside the contract, or
// CLIMB_EUNSUPPORTED when `cap` items do not suffice (at most tiles + 2 nwg + 1 are ever needed; with room for tiles + nwg + 1 only, the plan is
// made without the staggered epilogues of option 22).
// r05, staggered epilogues (climb_set_option 22 = G phase groups, 0 / 1 = off).  Every workgroup owns `rounds` whole tiles plus an equal share of the
// stream-K tail, so all 256 of them reach the optimizer-carrying epilogue of a whole tile at the same moment: 256 x 1.8 MB of p / m / v / shadow
// traffic arrive as one burst (~85 us at the HBM roof), five times per launch, with every matrix pipe idle -- the +0.39 ms the fused AdamW cost over
// the plain launch (r04).  The tail share (~0.2 tile of reduction work per workgroup) is the only freedom the equal-work plan has: phase group
// g = idx % G of an XCD runs the fraction g / (G - 1) of ITS share BEFORE its whole tiles and the rest after them, so the groups reach every epilogue
// ~60 us / (G - 1) apart and each burst is 256 / G workgroups wide, under the other groups' k-loops.  Same items, same sums (the stream-K shares still
// meet through atomics; a share cut in two at an even reduction tile adds one partial item).
static int g_tn_stagger = 0;          // measured r05 (profiles/r05_dw_stagger_ab.txt): the step is 0.03 - 0.13 ms SLOWER with any grouping (inside an XCD the phase groups stop sharing panels in L2; by XCD nothing is gained): off
void climb_tn_set_stagger(int v) { g_tn_stagger = v; }
int climb_tn_get_stagger() { return g_tn_stagger; }
extern "C" int climb_tn_grouped_plan(int nprob, const int* M, const int* N, const int* K, int nwg, int* items_out, int cap, int* first_out) {
  if (nprob <= 0 || nwg <= 0 || (nwg % 8) || !M || !N || !K || !items_out || !first_out) return CLIMB_EINVAL;
  struct Tile { int prob, tn, tk, nkt; };
  struct Piece { Tile x; int kt0, kt1; };
  long ntiles = 0;
  for (int p = 0; p < nprob; ++p) {
    if (M[p] < 128 || (M[p] % 128) || N[p] < 8 || (N[p] % 8) || K[p] < 8 || (K[p] % 8)) return CLIMB_EINVAL;
    ntiles += (long)((N[p] + 255) / 256) * ((K[p] + 255) / 256);
  }
  if (ntiles + nwg + 1 > cap) return CLIMB_EUNSUPPORTED;
  std::vector<Tile> tiles;
  tiles.reserve(ntiles);
  for (int p = 0; p < nprob; ++p) {
    const int nbn = (N[p] + 255) / 256, nbk = (K[p] + 255) / 256;
    for (int i = 0; i < nbn * nbk; ++i) {
      // consecutive tiles (= the CUs of one XCD) walk the narrower operand fastest: the XCD's L2 re-reads the smaller panels
      Tile x;
      x.prob = p;
      x.tn = nbn > nbk ? i / nbk : i % nbn;
      x.tk = nbn > nbk ? i % nbk : i / nbn;
      x.nkt = M[p] / 64;
      tiles.push_back(x);
    }
  }
  const int per_xcd = nwg / 8;
  // per-workgroup item lists, indexed by RANK r = xcd * per_xcd + idx (consecutive ranks share an XCD); blockIdx b <-> rank (b % 8) * per_xcd + b / 8
  const long rounds = ntiles / nwg, tail0 = rounds * nwg;
  long units = 0;
  for (long i = tail0; i < ntiles; ++i) units += tiles[i].nkt;
  long q = (units + nwg - 1) / nwg;
  q += q & 1;                                              // even share: cuts fall on even reduction-tile indices
  // the stream-K shares of the tail, per rank
  std::vector<std::vector<Piece>> share(nwg);
  {
    int r = 0;
    long rem = q;
    for (long i = tail0; i < ntiles; ++i) {
      int pos = 0;
      while (pos < tiles[i].nkt) {
        const int take = (int)((tiles[i].nkt - pos) < rem ? (tiles[i].nkt - pos) : rem);
        share[r].push_back(Piece{tiles[i], pos, pos + take});
        pos += take;
        rem -= take;
        if (rem == 0) { ++r; rem = q; }
      }
    }
  }
  // (a caller that sized its table for the unstaggered plan, tiles + nwg + 1 items, gets that plan: a cut share needs up to nwg more)
  const bool by_xcd = g_tn_stagger >= 100;          // 100 + G: whole XCDs form the phase groups (their 32 workgroups stay in lock step and keep sharing panels in L2)
  const int gs = by_xcd ? g_tn_stagger - 100 : g_tn_stagger;
  const int G = (gs >= 2 && rounds >= 1 && ntiles + 2L * nwg + 1 <= cap) ? gs : 1;
  std::vector<std::vector<Piece>> list(nwg);
  for (int r = 0; r < nwg; ++r) {
    std::vector<Piece> before, after;
    long len = 0;
    for (const Piece& pc : share[r]) len += pc.kt1 - pc.kt0;
    const int g = by_xcd ? (r / per_xcd) % G : (r % per_xcd) % G;
    long want = G > 1 ? (len * g / (G - 1)) & ~1L : 0;     // reduction tiles of the share that run before the whole tiles (even)
    for (const Piece& pc : share[r]) {
      const int n = pc.kt1 - pc.kt0;
      if (want >= n) { before.push_back(pc); want -= n; }
      else if (want >= 2 && n - want >= 2) {               // cut inside this piece: both parts keep >= 2 reduction tiles, even cut point
        before.push_back(Piece{pc.x, pc.kt0, pc.kt0 + (int)want});
        after.push_back(Piece{pc.x, pc.kt0 + (int)want, pc.kt1});
        want = 0;
      } else { after.push_back(pc); want = 0; }
    }
    for (const Piece& pc : before) list[r].push_back(pc);
    for (long j = 0; j < rounds; ++j) list[r].push_back(Piece{tiles[j * nwg + r], 0, tiles[j * nwg + r].nkt});
    for (const Piece& pc : after) list[r].push_back(pc);
  }
  int at = 0;
  std::vector<int> rank_first(nwg + 1);
  for (int b = 0; b < nwg; ++b) {
    const int r = (b % 8) * per_xcd + b / 8;
    first_out[b] = at;
    rank_first[r] = at;
    at += (int)list[r].size();
  }
  first_out[nwg] = at;
  if (at > cap) return CLIMB_EUNSUPPORTED;
  for (int r = 0; r < nwg; ++r)
    for (size_t k = 0; k < list[r].size(); ++k) {
      const Piece& pc = list[r][k];
      int* o = items_out + 8L * (rank_first[r] + (long)k);
      o[0] = pc.x.prob; o[1] = pc.x.tn; o[2] = pc.x.tk; o[3] = pc.kt0; o[4] = pc.kt1; o[5] = (pc.kt0 != 0 || pc.kt1 != pc.x.nkt) ? 1 : 0; o[6] = o[7] = 0;
    }
  return at;
}

// probs: TnGroupProblem[...] in device memory, items / first: the planner's tables copied to device memory (ints).  nwg workgroups
// (what the plan was made for; 256 on MI355X).
extern "C" int climb_gemm_bf16_tn_grouped(const void* probs, const void* items, const void* first, int nwg, int ragged, void* stream) {
  if (!probs || !items || !first || nwg <= 0) return CLIMB_EINVAL;
  constexpr int LDS = 2 * (NTP_A_BYTES + 4 * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  if (ragged)
    hipLaunchKernelGGL(gemm_bf16_tn_grouped_kernel<true>, dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                       (const TnGroupItem*)items, (const int*)first);
  else
    hipLaunchKernelGGL(gemm_bf16_tn_grouped_kernel<false>, dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                       (const TnGroupItem*)items, (const int*)first);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// r06 (split.hip): the grouped launch over split operands.  A problem's A / B name the hi planes of (hi, lo) pairs stacked along the token dimension (the
// lo plane directly behind the hi plane: [2 Mt, .]), its M field = 3 Mt (three products per token tile: 3 Mt / 64 reduction tiles the planner cuts like any reduction) and `reserved` = Mt / 64.
// Whole 256 x 256 tiles only (no ragged problems); dbias += the column sums of A hi + A lo.
extern "C" int climb_gemm_split_tn_grouped(const void* probs, const void* items, const void* first, int nwg, void* stream) {
  if (!probs || !items || !first || nwg <= 0) return CLIMB_EINVAL;
  constexpr int LDS = 2 * (NTP_A_BYTES + 4 * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  hipLaunchKernelGGL((gemm_bf16_tn_grouped_kernel<false, false, false, true>), dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                     (const TnGroupItem*)items, (const int*)first);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// ... and with the optimizer in its epilogue (climb_gemm_bf16_tn_grouped_adamw's contract: opts, adam, grad_dirty): a fused whole tile writes p, m, v and the hi AND lo
// planes of both shadows -- s_lo / st_lo = elements from the hi to the lo plane of the straight / transposed shadow ([2][total] buffers).
extern "C" int climb_gemm_split_tn_grouped_adamw(const void* probs, const void* items, const void* first, int nwg, const void* opts, const float* adam, int grad_dirty,
                                                 long s_lo, long st_lo, void* stream) {
  if (!probs || !items || !first || nwg <= 0 || !opts || !adam || s_lo <= 0 || st_lo <= 0 || s_lo * 2 >= (1L << 32) || st_lo * 2 >= (1L << 32)) return CLIMB_EINVAL;
  constexpr int LDS = 2 * (NTP_A_BYTES + 4 * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  TnAdam ad;
  ad.g = AdamGroup{adam[0], adam[1], adam[2], adam[3], adam[4], adam[5], adam[6], 0.f};
  ad.gscale = adam[7];
  ad.grad_dirty = grad_dirty;
  ad.ewc = TnEwc{nullptr, nullptr, nullptr, nullptr, 0.f, 0};
  ad.s_lo_b = (unsigned)(s_lo * 2);
  ad.st_lo_b = (unsigned)(st_lo * 2);
  hipLaunchKernelGGL((gemm_bf16_tn_grouped_kernel<false, true, false, true>), dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                     (const TnGroupItem*)items, (const int*)first, (const TnGroupOpt*)opts, ad);
  LAUNCH_CHECK();
  return CLIMB_OK;
}

// The same launch with the optimizer in the epilogue of the problems whose TnGroupOpt says `fused` (see above).  opts: device array parallel to
// probs; adam: HOST array of 8 floats { lr, weight decay, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t, gradient scale }; grad_dirty != 0: the
// gradient buffer (the problems' C) is not zero and is added to the tile sums before the update (it is never written for fused problems).
static int tn_grouped_adamw_launch(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam, int grad_dirty,
                                   const TnEwc* ewc, void* stream);
extern "C" int climb_gemm_bf16_tn_grouped_adamw(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam,
                                                int grad_dirty, void* stream) {
  return tn_grouped_adamw_launch(probs, items, first, nwg, ragged, opts, adam, grad_dirty, nullptr, stream);
}
// r05: ... and the EWC term (see TnEwc): `flat` = the fp32 parameter buffer the fused problems' p pointers point into, `star` / `fisher` = theta* / F laid
// out like its encoder range (every fused problem must lie inside it), *loss += lam * sum F (theta - theta*)^2 over the updated elements.
extern "C" int climb_gemm_bf16_tn_grouped_adamw_ewc(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam,
                                                    int grad_dirty, const float* flat, const float* star, const float* fisher, float lam, float* loss, void* stream) {
  if (!flat || !star || !fisher || !loss) return CLIMB_EINVAL;
  const TnEwc e{flat, star, fisher, loss, lam, 0};
  return tn_grouped_adamw_launch(probs, items, first, nwg, ragged, opts, adam, grad_dirty, &e, stream);
}
static int tn_grouped_adamw_launch(const void* probs, const void* items, const void* first, int nwg, int ragged, const void* opts, const float* adam, int grad_dirty,
                                   const TnEwc* ewc, void* stream) {
  if (!probs || !items || !first || nwg <= 0 || !opts || !adam) return CLIMB_EINVAL;
  constexpr int LDS = 2 * (NTP_A_BYTES + 4 * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)gemm_bf16_tn_grouped_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  TnAdam ad;
  ad.g = AdamGroup{adam[0], adam[1], adam[2], adam[3], adam[4], adam[5], adam[6], 0.f};
  ad.gscale = adam[7];
  ad.grad_dirty = grad_dirty;
  ad.ewc = ewc ? *ewc : TnEwc{nullptr, nullptr, nullptr, nullptr, 0.f, 0};
  ad.s_lo_b = ad.st_lo_b = 0;
  if (ewc) {
    if (ragged || grad_dirty) return CLIMB_EUNSUPPORTED;          // (the engine never defers a ragged plan, and folds the term only into a step whose gradient buffer is clean)
    hipLaunchKernelGGL((gemm_bf16_tn_grouped_kernel<false, true, true>), dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                       (const TnGroupItem*)items, (const int*)first, (const TnGroupOpt*)opts, ad);
    LAUNCH_CHECK();
    return CLIMB_OK;
  }
  if (ragged)
    hipLaunchKernelGGL((gemm_bf16_tn_grouped_kernel<true, true>), dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                       (const TnGroupItem*)items, (const int*)first, (const TnGroupOpt*)opts, ad);
  else
    hipLaunchKernelGGL((gemm_bf16_tn_grouped_kernel<false, true>), dim3(nwg), dim3(512), LDS, (hipStream_t)stream, (const TnGroupProblem*)probs,
                       (const TnGroupItem*)items, (const int*)first, (const TnGroupOpt*)opts, ad);
  LAUNCH_CHECK();
  return CLIMB_OK;
}
