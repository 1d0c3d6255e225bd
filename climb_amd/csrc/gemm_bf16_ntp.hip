// bf16 NT GEMM, 256 x (64 NI) x 64 tiles (NI = 4: 256 x 256, NI = 3: 256 x 192), ONE persistent workgroup of 8 waves per CU,
// NI phases per k-tile with counted vmcnt (the "8-phase" structure of the CDNA4 GEMM playbook: 8 phases = 2 k-tiles of 256 x 256).
//
// Why it exists.  The 128 x 128 kernel of gemm_bf16.hip drains vmcnt(0) at a barrier every k-tile (~30 % MFMA utilisation); the
// 192 x 192 three-stage kernel keeps two k-tiles in flight but its 32 x 96 wave tiles read 42 B of LDS per kFLOP and stop at ~840 TF.
// Here a wave owns 64 x (32 NI): 23 - 26 B of LDS per kFLOP, and the phases below keep a whole k-tile of LDS-DMA in flight across
// the barriers.  Measured (r02, M = 12288, random operands): k-loop 1290 - 1380 TF at large K (the 128 x 128 kernel: 930 - 990).
//
// Geometry.  Waves 4 (M) x 2 (N); wave (wr, wc) owns rows wr*64 .. +64, columns wc*32 NI .. +32 NI = [NI n-blocks][2 m-blocks] of
// v_mfma_f32_32x32x16_bf16 (128 / 96 accumulator registers), operands swapped as in gemm_bf16.hip (a lane owns one output row x 4
// consecutive columns per register group: the LDS-turned row-contiguous epilogue of gemm_bf16_nt.h is reused).
//
// LDS: 2 k-tile buffers of { A image [256 rows][64 k] = 32 KB | NI B units of 8 KB }.  B unit p holds the rows of n-block p of BOTH
// wave columns ({wc*32 NI + 32 p + 0..31}), so that in phase p every wave reads unit p and the unit is dead when the phase is over.
// All images have 128-B rows with the 16-B chunks XOR-swizzled by swz(row); they are filled by LDS-DMA (global_load_lds_dwordx4)
// with the swizzle applied on the SOURCE address.  Staging units: A0 / A1 = rows 0..127 / 128..255 (2 DMA instructions per lane
// each), B_p (1 instruction per lane).
//
// Phase p of k-tile T = { ds_read the fragments of n-block p (4 x ds_read_b128; in phase 0 also the A fragments of both m-blocks,
// 8 x, kept for the whole k-tile); issue this phase's staging units; counted vmcnt; s_barrier; 8 MFMAs under s_setprio(1);
// s_barrier }.  The two wave groups (waves 0-3 / 4-7: one of each per SIMD) run staggered by one barrier, so while one group
// issues MFMAs the other issues its ds_reads / DMAs.  With that stagger the placement rules are (derivation in DESIGN.md):
//   RAW  data whose vmcnt wait sits before the first barrier of phase w may be read from phase w + 1 on;
//   WAR  a unit whose last ds_read is in phase q may be restaged from phase q + 2 on.
// The issue schedule (ntp_sched) puts every unit at the earliest phase WAR allows and the wait counts are DERIVED from it at compile
// time (ntp_wait simulates the issue sequence, including the two tail k-tiles and the prologue), so the two cannot drift apart.
// vmcnt is never 0 in the main loop: 7 - 10 DMA instructions (about one k-tile) stay in flight across every barrier.
//
// Persistent: gridDim.x = min(tiles, 256); workgroup b walks tiles b, b + grid, ... (same XCD, XCD-contiguous supertile order), so
// the epilogue's global stores drain under the next tile's k-loop instead of delaying the workgroup's exit (measured: the store burst
// of a 256-CU round is 5 - 13 us of HBM-write time that a one-tile-per-workgroup launch exposes after every round).
#include "gemm_bf16_phase.h"

template <int NI>
struct NtpStage {
  unsigned a_off[4];       // per lane: byte offset (from A) of its 4 DMA sources of the A image, k-tile 0
  unsigned b_off[NI];      // per lane: byte offset (from B) of its DMA source of B unit p
  unsigned wid;            // wave index (uniform)
  // store-wave mode (SW): waves 0-3 stage the WHOLE images -- their own pieces and, at these byte distances, those of wave wid + 4 (A: 8
  // pieces = 64 rows further, B: 4 pieces = the other wave column's 32 rows: same swizzle, so the same per-lane offsets serve both)
  long da, db;
  bool loader;             // this wave issues DMA and waits on vmcnt (SW: waves 0-3 only)
};

// issue the staging units of phase P of the body of k-tile T (D1 / D2: units of k-tile T + 1 / T + 2 are still inside the k-loop)
template <int NI, int P, bool D1, bool D2, int K_ = 0, bool SW = false>
__device__ __forceinline__ void ntp_issue(const NtpStage<NI>& sg, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                          unsigned char* __restrict__ smem, int T) {
  if constexpr (K_ < NTP_MAXI) {
    constexpr int BUF = NTP_A_BYTES + NI * NTP_B_UNIT;
    constexpr int e = ntp_sched(NI, P, K_);
    if constexpr (e >= 0) {
      constexpr int unit = e / 4, d = e % 4;
      if constexpr ((d == 1 && D1) || (d == 2 && D2)) {
        const int t = T + d;
        unsigned char* buf = smem + (t & 1) * BUF;
        const long koff = (long)t * (GB_BK * 2);               // bytes along k: uniform, folds into the scalar base
        if constexpr (unit < 2) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(A) + koff + sg.a_off[unit * 2 + i]),
                                             (lds_void_t*)(buf + unit * (NTP_A_BYTES / 2) + (sg.wid * 2 + i) * 1024), 16, 0, 0);
            if constexpr (SW)
              __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(A) + koff + sg.da + sg.a_off[unit * 2 + i]),
                                               (lds_void_t*)(buf + unit * (NTP_A_BYTES / 2) + (sg.wid * 2 + i + 8) * 1024), 16, 0, 0);
          }
        } else {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(B) + koff + sg.b_off[unit - 2]),
                                           (lds_void_t*)(buf + NTP_A_BYTES + (unit - 2) * NTP_B_UNIT + sg.wid * 1024), 16, 0, 0);
          if constexpr (SW)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(reinterpret_cast<const unsigned char*>(B) + koff + sg.db + sg.b_off[unit - 2]),
                                             (lds_void_t*)(buf + NTP_A_BYTES + (unit - 2) * NTP_B_UNIT + (sg.wid + 4) * 1024), 16, 0, 0);
        }
      }
    }
    ntp_issue<NI, P, D1, D2, K_ + 1, SW>(sg, A, B, smem, T);
  }
}

__device__ __forceinline__ bf16x8 ntp_frag(const unsigned char* __restrict__ p) { return as_bf16x8(*reinterpret_cast<const u32x4*>(p)); }

template <int NI, int P, bool D1, bool D2, int W, bool SW = false>
__device__ __forceinline__ void ntp_phase(f32x16 (&acc)[NI][2], bf16x8 (&a)[2][4], const NtpStage<NI>& sg, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[4],
                                          const unsigned (&boff)[4]) {
  constexpr int BUF = NTP_A_BYTES + NI * NTP_B_UNIT;
  const unsigned char* buf = smem + (T & 1) * BUF;
  bf16x8 b[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) b[ks] = ntp_frag(buf + NTP_A_BYTES + P * NTP_B_UNIT + boff[ks]);
  if constexpr (P == 0) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[jj][ks] = ntp_frag(buf + jj * 4096 + aoff[ks]);
  }
  if constexpr (SW) {
    if (sg.loader) {          // wave-uniform: waves 4-7 issue no loads and never look at vmcnt (it counts their epilogue stores only)
      ntp_issue<NI, P, D1, D2, 0, true>(sg, A, B, smem, T);
      if constexpr (W >= 0) wait_vmcnt<(W < 0 ? 0 : W)>();
    }
  } else {
    ntp_issue<NI, P, D1, D2>(sg, A, B, smem, T);
    if constexpr (W >= 0) wait_vmcnt<(W < 0 ? 0 : W)>();
  }
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
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// one k-tile; MODE 0: main loop (T <= nk - 3), 1: T = nk - 2, 2: T = nk - 1
template <int NI, int MODE, int P = 0, bool SW = false>
__device__ __forceinline__ void ntp_ktile(f32x16 (&acc)[NI][2], bf16x8 (&a)[2][4], const NtpStage<NI>& sg, const bf16_t* __restrict__ A,
                                          const bf16_t* __restrict__ B, unsigned char* __restrict__ smem, int T, const unsigned (&aoff)[4],
                                          const unsigned (&boff)[4]) {
  if constexpr (P < NI) {
    constexpr int W = ntp_wait(NI, 6, MODE == 0 ? 2 : (MODE == 1 ? 4 : 5), P, SW ? 2 : 1, SW ? 4 : 2);
    ntp_phase<NI, P, MODE <= 1, MODE == 0, W, SW>(acc, a, sg, A, B, smem, T, aoff, boff);
    ntp_ktile<NI, MODE, P + 1, SW>(acc, a, sg, A, B, smem, T, aoff, boff);
  }
}
template <int NI, int P = 0, bool SW = false>
__device__ __forceinline__ void ntp_prologue(const NtpStage<NI>& sg, const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                             unsigned char* __restrict__ smem) {
  // what the bodies of k-tiles -2 and -1 would have issued, in the same order
  if constexpr (P < NI) {
    ntp_issue<NI, P, false, true, 0, SW>(sg, A, B, smem, -2);
    ntp_prologue<NI, P + 1, SW>(sg, A, B, smem);
  } else if constexpr (P < 2 * NI) {
    ntp_issue<NI, P - NI, true, true, 0, SW>(sg, A, B, smem, -1);
    ntp_prologue<NI, P + 1, SW>(sg, A, B, smem);
  }
}

// Epilogue of one wave: its 64 x (32 NI) block goes out as sub-blocks of 32 rows x (32 NIE) columns (NIE = 2 for NI = 4: the 16
// residual float4s of a full-width block next to 128 accumulators spill; NIE = 3 for NI = 3), each turned through the wave's private
// NIE*4 KB staging region; the residual / pre-activation operands of sub-block s + 1 are fetched while sub-block s drains.
template <int NI> struct NtpEpi { static constexpr int NIE = NI == 4 ? 2 : NI, PER_ROW = NI / NIE, NSB = 2 * PER_ROW; };
template <int S, typename TO, int EPI, int NI>
__device__ __forceinline__ void ntp_epilogue(const f32x16 (&acc)[NI][2], AuxRegs<EPI, NtpEpi<NI>::NIE * 4> (&ax)[NtpEpi<NI>::NSB],
                                             NtBias<NtpEpi<NI>::NIE> (&bb)[NtpEpi<NI>::PER_ROW], unsigned char* __restrict__ stage, int el, int mw, int nw,
                                             int M, int N, TO* __restrict__ C, long ldc, const float* __restrict__ bias, const void* __restrict__ aux,
                                             long ldaux, bf16_t* __restrict__ aux_out, long ldauxo) {
  using E = NtpEpi<NI>;
  if constexpr (S < E::NSB) {
    constexpr int J = S / E::PER_ROW, H = S % E::PER_ROW;
    if constexpr (S == 0) {
      // no load may sit behind a store (see NtBias in gemm_bf16_nt.h): bias once; the operands of sub-block s + 1 are fetched after the
      // accumulators of sub-block s were staged (registers) and waited for BEFORE sub-block s stores
#pragma unroll
      for (int h = 0; h < E::PER_ROW; ++h) nt_bias_preload<E::NIE>(bb[h], bias, el, nw + h * E::NIE * 32, N);
      nt_aux_prefetch_l<EPI, E::NIE, 1>(ax[0], el, mw, nw, M, N, aux, ldaux, nullptr, 0);
    }
    {
      f32x16 blk[E::NIE];
#pragma unroll
      for (int i = 0; i < E::NIE; ++i) blk[i] = acc[H * E::NIE + i][J];
      nt_epi_stage<E::NIE>(blk, stage, el);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < E::NSB) {
      constexpr int J1 = (S + 1) / E::PER_ROW, H1 = (S + 1) % E::PER_ROW;
      nt_aux_prefetch_l<EPI, E::NIE, 1>(ax[S + 1], el, mw + J1 * 32, nw + H1 * E::NIE * 32, M, N, aux, ldaux, nullptr, 0);
      nt_aux_touch(ax[S + 1]);
    }
    nt_aux_touch(ax[S]);
    __builtin_amdgcn_sched_barrier(0);
    nt_epi_drain<TO, EPI, E::NIE, 0>(ax[S], bb[H], stage, el, mw + J * 32, nw + H * E::NIE * 32, M, N, C, ldc, aux_out, ldauxo);
    __builtin_amdgcn_sched_barrier(0);
    ntp_epilogue<S + 1, TO, EPI, NI>(acc, ax, bb, stage, el, mw, nw, M, N, C, ldc, bias, aux, ldaux, aux_out, ldauxo);
  }
}

// STORE-WAVE MODE (r03, SW).  On gfx9 a wave's loads and stores share one vmcnt, so a wave that has just stored its tile cannot start the
// next tile's counted LDS-DMA waits before the stores have drained: the epilogue's HBM burst ADDS to the k-loop.
// Here the two kinds live in different waves.  Waves 0-3 stage the whole A / B images (their own pieces and those of wave + 4: same per-lane
// offsets, a uniform distance) and are the only ones that wait on vmcnt; waves 4-7 issue no loads in the k-loop and are the only ones that
// store: in the epilogue every wave turns a sub-block of its accumulators through its staging region as before, a raw barrier publishes the
// regions, and wave w + 4 drains its own region and then wave w's (same wave column: the same bias registers serve both).  Waves 4-7 never
// wait for their stores -- they drain under the next tile's k-loop, and the hardware holds back only the next epilogue's store issue if a
// previous tile's are still in flight.  EPI NONE / GELU only (no operand LOADS in the epilogue), whole tiles only (M % 256, N % BN).
template <int S, typename TO, int EPI, int NI>
__device__ __forceinline__ void ntp_epilogue_sw(const f32x16 (&acc)[NI][2], NtBias<NtpEpi<NI>::NIE> (&bb)[NtpEpi<NI>::PER_ROW], unsigned char* __restrict__ smem,
                                                int wid, int el, int m0, int n0, int M, int N, TO* __restrict__ C, long ldc, const float* __restrict__ bias,
                                                bf16_t* __restrict__ aux_out, long ldauxo) {
  using E = NtpEpi<NI>;
  if constexpr (S < E::NSB) {
    constexpr int J = S / E::PER_ROW, H = S % E::PER_ROW;
    const int wc = wid & 1, storer = wid >> 2;
    unsigned char* stage = smem + wid * (E::NIE * 4096);
    if constexpr (S == 0) {
      if (storer) {
#pragma unroll
        for (int h = 0; h < E::PER_ROW; ++h) nt_bias_preload<E::NIE>(bb[h], bias, el, n0 + wc * (32 * NI) + h * E::NIE * 32, N);
      }
    }
    {
      f32x16 blk[E::NIE];
#pragma unroll
      for (int i = 0; i < E::NIE; ++i) blk[i] = acc[H * E::NIE + i][J];
      nt_epi_stage<E::NIE>(blk, stage, el);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                            // every wave's sub-block S is staged
    __builtin_amdgcn_sched_barrier(0);
    if (storer) {
      AuxRegs<EPI, E::NIE * 4> none;
      const int nw = n0 + wc * (32 * NI) + H * E::NIE * 32;
      nt_epi_drain<TO, EPI, E::NIE, 0>(none, bb[H], stage, el, m0 + (wid >> 1) * 64 + J * 32, nw, M, N, C, ldc, aux_out, ldauxo);
      __builtin_amdgcn_sched_barrier(0);
      nt_epi_drain<TO, EPI, E::NIE, 0>(none, bb[H], stage - 4 * (E::NIE * 4096), el, m0 + ((wid - 4) >> 1) * 64 + J * 32, nw, M, N, C, ldc, aux_out, ldauxo);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the drains' LDS reads have returned (their global stores have only been issued)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                            // the staging regions may be rewritten
    __builtin_amdgcn_sched_barrier(0);
    ntp_epilogue_sw<S + 1, TO, EPI, NI>(acc, bb, smem, wid, el, m0, n0, M, N, C, ldc, bias, aux_out, ldauxo);
  }
}

template <typename TO, int EPI, int NI, bool SW = false>
__global__ __launch_bounds__(512) void gemm_bf16_ntp_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                            TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                            const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo, int probe) {
  constexpr int BN = 64 * NI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x (32 KB + NI x 8 KB) = 128 KB / 112 KB
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2, half = lane >> 5, l31 = lane & 31;
  const int nbm = (M + NTP_BM - 1) / NTP_BM, nbn = (N + BN - 1) / BN;
  const int gm = ((probe >> 8) & 255) ? ((probe >> 8) & 255) : ((nbm % 8 == 0) ? nbm / 8 : 8);       // M-tiles per supertile (see nt_tile_id_gm); bits 8..15 of `probe`: A/B override
  // De-phasing (bits 16.. of `probe`, climb_set_option 16; multi-round launches only): every workgroup starts together and every tile costs
  // the same, so the whole chip computes, then the whole chip stores -- an HBM burst with idle matrix pipes, then idle HBM under busy pipes.
  // Workgroup group g = (id / 8) % groups (id % 8 = the XCD: every XCD gets every group) holds back g x dph x 64 clocks ONCE, so that one
  // group's epilogue falls under the others' k-loops for the rest of the launch.
  if constexpr (!SW) {
    const int dph = (probe >> 16) & 1023, ng = 2 << ((probe >> 26) & 3), g = (blockIdx.x >> 3) & (ng - 1);
    if (dph > 0 && g > 0) {
      const long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < (long)g * dph * 64) __builtin_amdgcn_s_sleep(8);
    }
  }
  probe &= 1;
  // fragment read offsets: row (wr*64 | wc*32) + l31 (+ 32 for the second m-block), k-chunk 2 ks + half, swizzled
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cs = ((2 * ks + half) ^ swz(l31)) << 4;       // swz() only looks at row bits 1..3: the same for every block of a wave
    aoff[ks] = (wr * 64 + l31) * 128 + cs;
    boff[ks] = (wc * 32 + l31) * 128 + cs;
  }
  const int nk = K / GB_BK;                                  // >= 2 (launcher)
  for (int id = blockIdx.x; id < nbm * nbn; id += gridDim.x) {
    int tm, tn;
    nt_tile_id_gm(gm, id, nbm, nbn, tm, tn);
    const int m0 = tm * NTP_BM, n0 = tn * BN;
    // the lane id, recomputed per tile from the exec mask (two instructions) instead of kept in a register across the k-loop: with 256 VGPRs
    // allocated the residual instantiation used to SPILL it before the tile loop and reload it (scratch load + vmcnt(0)) in every epilogue
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
    // LDS-DMA plan: a 1 KB piece = 8 rows x 8 chunks; wave w issues pieces 2w, 2w + 1 of A0 and of A1 and piece w of every B unit
    NtpStage<NI> sg;
    sg.wid = wid;
    sg.loader = !SW || grp == 0;
    sg.da = 64 * lda * 2;
    sg.db = 32L * NI * ldb * 2;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + ln, r = u * 128 + (slot >> 3), c = (slot & 7) ^ swz(r);
        int arow = m0 + r;
        arow = arow < M ? arow : M - 1;                     // clamped rows are computed but never stored
        sg.a_off[u * 2 + i] = (unsigned)(((long)arow * lda + c * 8) * 2);
      }
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int slot = wid * 64 + ln, r = slot >> 3, c = (slot & 7) ^ swz(r);       // r = 0..63: wave column r >> 5, row r & 31 of n-block p
      int brow = n0 + (r >> 5) * (32 * NI) + p * 32 + (r & 31);
      brow = brow < N ? brow : N - 1;
      sg.b_off[p] = (unsigned)(((long)brow * ldb + c * 8) * 2);
    }
    f32x16 acc[NI][2];   // [n block][m block]
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[2][4];
    if constexpr (SW) {
      if (grp == 0) {
        ntp_prologue<NI, 0, true>(sg, A, B, smem);
        wait_vmcnt<ntp_wait(NI, 6, -1, NI - 1, 2, 4)>();
      }
    } else {
      ntp_prologue<NI>(sg, A, B, smem);
      wait_vmcnt<ntp_wait(NI, 6, -1, NI - 1)>();             // A0, A1, B_0 of k-tile 0: this wave's pieces have landed
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                            // ... and everybody else's
    if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind group 0 from here on
    __builtin_amdgcn_sched_barrier(0);
    int T = 0;
    for (; T + 2 < nk; ++T) ntp_ktile<NI, 0, 0, SW>(acc, a, sg, A, B, smem, T, aoff, boff);
    ntp_ktile<NI, 1, 0, SW>(acc, a, sg, A, B, smem, T, aoff, boff);
    ntp_ktile<NI, 2, 0, SW>(acc, a, sg, A, B, smem, T + 1, aoff, boff);
    if (grp == 0) __builtin_amdgcn_s_barrier();              // re-align the groups: every wave is past its last fragment read
    __builtin_amdgcn_sched_barrier(0);
    if (probe) {          // measurement aid (climb_set_option 8): the k-loop alone, accumulators kept live, nothing stored
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
    } else {
      // epilogue: the k-tile buffers are free; each wave turns its block through a private staging region inside them.
      // `el` is the lane id behind an opaque move, so that none of the epilogue's address arithmetic is hoisted out of the tile
      // loop and kept in registers across the k-loop (measured: 27 spilled registers and a vmcnt(0) in the k-loop otherwise).
      int el = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(el));
      NtBias<NtpEpi<NI>::NIE> bb[NtpEpi<NI>::PER_ROW];
      if constexpr (SW) {
        ntp_epilogue_sw<0, TO, EPI, NI>(acc, bb, smem, wid, el, m0, n0, M, N, C, ldc, bias, aux_out, ldauxo);
      } else {
        AuxRegs<EPI, NtpEpi<NI>::NIE * 4> ax[NtpEpi<NI>::NSB];
        ntp_epilogue<0, TO, EPI, NI>(acc, ax, bb, smem + wid * (NtpEpi<NI>::NIE * 4096), el, m0 + wr * 64, n0 + wc * (32 * NI), M, N, C, ldc, bias, aux,
                                     ldaux, aux_out, ldauxo);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();       // the staging regions are the next tile's k-tile buffers
    __builtin_amdgcn_sched_barrier(0);
  }
}

static int g_nt256_probe = 0, g_nt256_grid = 256;
// ---------------------------------------------------------------------------------------------------- split-along-K balancing (r03)
// The layer's N = 768 GEMMs tile 12288 x 768 into 192 tiles of 256 x 192: a quarter of the 256 CUs idles for the whole launch.  For the
// long-K ones (down-projection K = 3072, dhn K = 3072, dxn K = 2304: 182 us per layer) this kernel gives every CU 3/4 of a tile's k-loop
// instead: four workgroups (same XCD) share three tiles,
//     member 0: tile0 k-tiles [0, 3u)                      -> finishes tile0 (adds member 1's partial)
//     member 1: tile0 [3u, 4u) first, then tile1 [0, 2u)   -> hands tile0's partial over, finishes tile1 (adds member 2's)
//     member 2: tile2 [0, u) first, then tile1 [2u, 4u)    -> hands over both, finishes nothing
//     member 3: tile2 [u, 4u)                              -> finishes tile2 (adds member 2's first partial)          u = K / 256
// A partial is the raw fp32 accumulator image (96 registers x 512 lanes = 192 KB) written with SYSTEM-scope stores into a slot of the
// registered scratch (climb_set_tn_workspace) and announced by a flag; the finishing workgroup spins on the flag (bounded: a lost partner
// costs a wrong tile, never a hung GPU), adds the partial to its accumulators three 16-register blocks at a time and runs the ordinary
// epilogue.  System scope because nothing guarantees the four workgroups share an L2; the flag is reset by its reader, so a launch always
// starts from zeros (hipGraph replays included).  Every handed-over part is computed BEFORE the part its workgroup finishes, so only the
// tile2 hand-over (both sides end at 3u) can make anybody wait.  Requires exactly 192 tiles of 256 x 192, K % 256 == 0, K >= 512, grid 256.
#define NTSK_SLOT_FLOATS (96 * 512)
__device__ __forceinline__ void ntsk_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ float ntsk_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <typename TO, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_ntsk_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                                             TO* __restrict__ C, long ldc, int M, int N, int K, const float* __restrict__ bias,
                                                             const void* __restrict__ aux, long ldaux, bf16_t* __restrict__ aux_out, long ldauxo,
                                                             float* ws, int* flags) {
  constexpr int NI = 3, BN = 192;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2, half = lane >> 5, l31 = lane & 31;
  const int nbm = M / NTP_BM, nbn = N / BN, gm = nbm / 8;
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int cs = ((2 * ks + half) ^ swz(l31)) << 4;
    aoff[ks] = (wr * 64 + l31) * 128 + cs;
    boff[ks] = (wc * 32 + l31) * 128 + cs;
  }
  const int x = blockIdx.x & 7, idx = blockIdx.x >> 3, q = idx >> 2, mem = idx & 3;
  const int u = K / 256;                                   // a quarter of the tile's k-tiles
  const int slot0 = (q * 8 + x) * 3;                       // this group's three slots (one per tile)
  for (int part = 0; part < 2; ++part) {
    int j, kt0, kt1;
    bool writer;
    if (mem == 0)      { if (part) break; j = 0; kt0 = 0; kt1 = 3 * u; writer = false; }
    else if (mem == 1) { if (!part) { j = 0; kt0 = 3 * u; kt1 = 4 * u; writer = true; } else { j = 1; kt0 = 0; kt1 = 2 * u; writer = false; } }
    else if (mem == 2) { if (!part) { j = 2; kt0 = 0; kt1 = u; writer = true; } else { j = 1; kt0 = 2 * u; kt1 = 4 * u; writer = true; } }
    else               { if (part) break; j = 2; kt0 = u; kt1 = 4 * u; writer = false; }
    int tm, tn;
    nt_tile_id_gm(gm, x + 8 * (3 * q + j), nbm, nbn, tm, tn);
    const int m0 = tm * NTP_BM, n0 = tn * BN;
    const bf16_t* Ak = A + (long)kt0 * GB_BK;              // k-tiles [kt0, kt1) of this tile
    const bf16_t* Bk = B + (long)kt0 * GB_BK;
    const int nk = kt1 - kt0;                              // >= 2
    NtpStage<NI> sg;
    sg.wid = wid;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int slot = (wid * 2 + i) * 64 + lane, r = uu * 128 + (slot >> 3), c = (slot & 7) ^ swz(r);
        sg.a_off[uu * 2 + i] = (unsigned)(((long)(m0 + r) * lda + c * 8) * 2);
      }
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int slot = wid * 64 + lane, r = slot >> 3, c = (slot & 7) ^ swz(r);
      sg.b_off[p] = (unsigned)(((long)(n0 + (r >> 5) * (32 * NI) + p * 32 + (r & 31)) * ldb + c * 8) * 2);
    }
    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    {
      bf16x8 a[2][4];
      ntp_prologue<NI>(sg, Ak, Bk, smem);
      wait_vmcnt<ntp_wait(NI, 6, -1, NI - 1)>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      if (grp == 1) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      int T = 0;
      for (; T + 2 < nk; ++T) ntp_ktile<NI, 0>(acc, a, sg, Ak, Bk, smem, T, aoff, boff);
      ntp_ktile<NI, 1>(acc, a, sg, Ak, Bk, smem, T, aoff, boff);
      ntp_ktile<NI, 2>(acc, a, sg, Ak, Bk, smem, T + 1, aoff, boff);
      if (grp == 0) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    float* slot = ws + (long)(slot0 + j) * NTSK_SLOT_FLOATS + threadIdx.x;
    int* flag = flags + slot0 + j;
    if (writer) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 16; ++r) ntsk_store(slot + ((i * 2 + jj) * 16 + r) * 512, acc[i][jj][r]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's share of the partial is at the coherence point
      __syncthreads();                                       // ... and everybody's
      if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(8);
      }
      __syncthreads();
      // acc += partial, three 16-register blocks in flight (the fragment registers of the k-loop are free by now)
      float t[3][16];
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[b][r] = ntsk_load(slot + (b * 16 + r) * 512);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (b < 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (b == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (b == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b >> 1][b & 1][r] += t[b % 3][r];
        if (b + 3 < 6) {
#pragma unroll
          for (int r = 0; r < 16; ++r) t[b % 3][r] = ntsk_load(slot + ((b + 3) * 16 + r) * 512);
        }
      }
      __syncthreads();                                       // every lane has its share in registers
      if (threadIdx.x == 0) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the next launch starts from zeros
      int el = lane;
      asm volatile("" : "+v"(el));
      NtBias<NtpEpi<NI>::NIE> bb[NtpEpi<NI>::PER_ROW];
      AuxRegs<EPI, NtpEpi<NI>::NIE * 4> ax[NtpEpi<NI>::NSB];
      ntp_epilogue<0, TO, EPI, NI>(acc, ax, bb, smem + wid * (NtpEpi<NI>::NIE * 4096), el, m0 + wr * 64, n0 + wc * (32 * NI), M, N, C, ldc, bias, aux,
                                   ldaux, aux_out, ldauxo);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
}

static float* g_ntsk_ws = nullptr;
static long g_ntsk_bytes = 0;
static bool g_ntsk_flags_clean = false;
// MEASURED (r03, M = 12288, same box, us): dhn (K = 3072) 67.1 -> 70.4, dxn (K = 2304) 48.0 -> 55.4, down + residual 79.5 -> 126 (this
// instantiation spills around the partial add); in the step 10.72 -> 11.4 ms.  The k-loops do get 25 % shorter, but the hand-over costs more:
// 192 KB of system-scope dword stores + the flag + 96 system-scope loads per lane are ~18 us, and one of the three hand-overs of a group is
// always a tie (both sides finish at 3u), so it is fully exposed.  OFF by default; kept (with its race-screen test) as the priced answer to
// "balance the 192-tile GEMMs along K" -- a win needs the partial in LDS before the finishing workgroup's k-loop ends, and 192 KB does not fit.
static int g_ntsk_on = 0;                                    // climb_set_option 14
void climb_ntsk_set_workspace(void* ptr, long bytes) { g_ntsk_ws = (float*)ptr; g_ntsk_bytes = ptr ? bytes : 0; g_ntsk_flags_clean = false; }
void climb_ntsk_enable(int v) { g_ntsk_on = v; }
#define NTSK_FLAGS 192
#define NTSK_BYTES ((long)NTSK_FLAGS * NTSK_SLOT_FLOATS * 4 + 4096)

template <typename TO, int EPI>
static int ntsk_launch_one(hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K, const float* bias,
                           const void* aux, long ldaux, bf16_t* aux_out, long ldauxo) {
  constexpr int LDS = 2 * (NTP_A_BYTES + 3 * NTP_B_UNIT);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_ntsk_kernel<TO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  int* flags = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(g_ntsk_ws) + (long)NTSK_FLAGS * NTSK_SLOT_FLOATS * 4);
  if (!g_ntsk_flags_clean) {                                 // once per registered buffer, in stream order (readers reset their flag afterwards)
    hipError_t e = hipMemsetAsync(flags, 0, 4096, st);
    if (e != hipSuccess) return (int)e;
    g_ntsk_flags_clean = true;
  }
  hipLaunchKernelGGL((gemm_bf16_ntsk_kernel<TO, EPI>), dim3(256), dim3(512), LDS, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo,
                     g_ntsk_ws, flags);
  return CLIMB_OK;
}

// the split-along-K kernel takes: 48 x 4 tiles of 256 x 192 exactly, K % 256 == 0 and K >= 1536 (below that the hand-over costs more than
// the idle quarter of the chip), EPI NONE / RESID, a registered scratch of NTSK_BYTES, persistent grid 256
static int ntsk_try(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                    const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st) {
  if (!g_ntsk_on || g_nt256_grid != 256 || g_nt256_probe) return CLIMB_EUNSUPPORTED;
  if (M != 48 * NTP_BM || N != 4 * 192 || (K % 256) || K < 1536 || g_ntsk_bytes < NTSK_BYTES) return CLIMB_EUNSUPPORTED;
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_NONE) return ntsk_launch_one<bf16_t, EPI_NONE>(st, A, lda, B, ldb, (bf16_t*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo);
  if (c_dtype == CLIMB_DT_F32 && epi == EPI_RESID) return ntsk_launch_one<float, EPI_RESID>(st, A, lda, B, ldb, (float*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo);
  return CLIMB_EUNSUPPORTED;
}


void climb_nt256_set_probe(int v) { g_nt256_probe = v; }      // bit 0: k-loop only; v >> 8: supertile height override (measurement)
void climb_nt256_set_grid(int v) { g_nt256_grid = v; }
int climb_nt256_get_grid() { return g_nt256_grid; }

// MEASURED (r03, M = 12288, same box): QKV 51.6 -> 57.2 us, up-projection + GELU 82.4 -> 93.3 us, step 10.51 -> 10.71 ms.  The stores do
// leave the loading waves' vmcnt, but (i) waves 4-7 now drain two regions each while waves 0-3 idle at the barriers (the k-buffers ARE the
// staging regions: the next tile's DMA cannot start before the drains have read them), and (ii) the k-loop runs slower next to the write
// stream (the r01 probe said the same of a concurrent fill).  OFF by default; bit-exact under the race screens of tests/test_gpu_kernels.py.
static int g_ntp_sw = 0;          // climb_set_option 15: store-wave mode for the multi-round GEMMs without epilogue loads
void climb_ntp_set_sw(int v) { g_ntp_sw = v; }
static int g_ntp_dephase = 0;     // climb_set_option 16: v % 1000 = hold-back unit (x 64 clocks) of the de-phased workgroup groups, v / 1000 = k: 2 << k groups
void climb_ntp_set_dephase(int v) { g_ntp_dephase = v; }

template <typename TO, int EPI, int NI>
static int ntp_launch_one(int nwg, hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K,
                          const float* bias, const void* aux, long ldaux, bf16_t* aux_out, long ldauxo) {
  constexpr int LDS = 2 * (NTP_A_BYTES + NI * NTP_B_UNIT);
  static bool configured = false;      // per instantiation; the attribute is sticky for the process
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_ntp_kernel<TO, EPI, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  if constexpr (EPI == EPI_NONE || EPI == EPI_GELU) {
    // store-wave mode where stores of one tile can hide under the next: more tiles than workgroups, whole tiles only
    const long tiles = (long)((M + NTP_BM - 1) / NTP_BM) * ((N + 64 * NI - 1) / (64 * NI));
    if (g_ntp_sw && !(g_nt256_probe & 1) && tiles > nwg && (M % NTP_BM) == 0 && (N % (64 * NI)) == 0) {
      static bool configured_sw = false;
      if (!configured_sw) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_ntp_kernel<TO, EPI, NI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        configured_sw = true;
      }
      hipLaunchKernelGGL((gemm_bf16_ntp_kernel<TO, EPI, NI, true>), dim3(nwg), dim3(512), LDS, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out,
                         ldauxo, g_nt256_probe);
      return CLIMB_OK;
    }
  }
  int probe = g_nt256_probe & 0xffff;
  if (g_ntp_dephase > 0 && (long)((M + NTP_BM - 1) / NTP_BM) * ((N + 64 * NI - 1) / (64 * NI)) > nwg)
    probe |= ((g_ntp_dephase % 1000) & 1023) << 16 | ((g_ntp_dephase / 1000) & 3) << 26;
  hipLaunchKernelGGL((gemm_bf16_ntp_kernel<TO, EPI, NI>), dim3(nwg), dim3(512), LDS, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out,
                     ldauxo, probe);
  return CLIMB_OK;
}

// bn 256 / 192; c_dtype 0 fp32 / 1 bf16.  Requires K % 64 == 0, K >= 128 and operands below 4 GB (32-bit DMA source offsets).
int climb_nt256_launch(int bn, const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias,
                       int epi, const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, const bf16_t* aux2, long ldaux2, hipStream_t st) {
  if ((bn != 256 && bn != 192) || (K % GB_BK) || K < 2 * GB_BK) return CLIMB_EUNSUPPORTED;
  if (((long)M * lda + K) * 2 >= (1L << 32) || ((long)N * ldb + K) * 2 >= (1L << 32)) return CLIMB_EUNSUPPORTED;
  // 16 residual float4s per lane next to 128 accumulators spill (measured: 80 B/lane of scratch): the residual epilogue only exists
  // for the 192-wide tile, which is also the width the layer's residual GEMMs (N = 768) tile best with
  // (the x GELU' epilogue at 256 columns spills 64 - 72 B/lane as well -- its pre-activation operand next to 128 accumulators -- and goes the same way)
  if (epi == EPI_RESID || epi == EPI_DGELU || epi == EPI_MUL) bn = 192;
  if (bn == 192) {
    const int rc = ntsk_try(A, lda, B, ldb, C, ldc, c_dtype, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo, st);
    if (rc != CLIMB_EUNSUPPORTED) return rc;
  }
  int nwg = ((M + NTP_BM - 1) / NTP_BM) * ((N + bn - 1) / bn);
  if (g_nt256_grid > 0 && nwg > g_nt256_grid) nwg = g_nt256_grid;       // persistent: one workgroup per CU walks the tiles
#define LNTP(TO, E, NI_) return ntp_launch_one<TO, E, NI_>(nwg, st, A, lda, B, ldb, (TO*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo)
#define LNTP_EPI(TO, NI_)                     \
  switch (epi) {                              \
    case EPI_NONE: LNTP(TO, EPI_NONE, NI_);   \
    case EPI_GELU: LNTP(TO, EPI_GELU, NI_);   \
    case EPI_DGELU: LNTP(TO, EPI_DGELU, NI_); \
    default: break;                           \
  }
#define LNTP_EPI256(TO)                     \
  switch (epi) {                            \
    case EPI_NONE: LNTP(TO, EPI_NONE, 4);   \
    case EPI_GELU: LNTP(TO, EPI_GELU, 4);   \
    default: break;                         \
  }
  if (c_dtype == CLIMB_DT_F32 && bn == 256) { LNTP_EPI256(float) }
  if (c_dtype == CLIMB_DT_BF16 && bn == 256) { LNTP_EPI256(bf16_t) if (epi == EPI_GELUD) LNTP(bf16_t, EPI_GELUD, 4); }
#undef LNTP_EPI256
  if (c_dtype == CLIMB_DT_F32 && bn == 192) { LNTP_EPI(float, 3) if (epi == EPI_RESID) LNTP(float, EPI_RESID, 3); }
  if (c_dtype == CLIMB_DT_BF16 && bn == 192) {
    LNTP_EPI(bf16_t, 3)
    if (epi == EPI_RESID) LNTP(bf16_t, EPI_RESID, 3);
    if (epi == EPI_GELUD) LNTP(bf16_t, EPI_GELUD, 3);
    if (epi == EPI_MUL) LNTP(bf16_t, EPI_MUL, 3);
  }
#undef LNTP_EPI
#undef LNTP
  return CLIMB_EUNSUPPORTED;
}
