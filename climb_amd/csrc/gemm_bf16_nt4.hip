// bf16 NT GEMM, 192 x 192 x 64 tiles, ONE persistent workgroup of FOUR waves per CU (one wave per SIMD, up to 512 registers), TWO accumulator
// sets: the epilogue of tile i runs INSIDE the k-loop of tile i + 1 (r04).
//
// Why it exists ("what the r01 - r03 evidence says the next NT kernel is").  In the 8-wave kernels of gemm_bf16_ntp.hip
// half of a K = 768 tile's time is outside the k-loop: the accumulators ARE the epilogue's input, so the k-loop of the next tile cannot start
// before the tile has been turned through LDS and stored, and a wave's global stores share vmcnt with its LDS-DMA loads, so the next tile's
// first counted wait also waits for the whole store burst -- 256 CUs storing in lock step, then 256 CUs computing with HBM idle.  Here
//   * a workgroup's k-tiles form ONE stream across its tiles: the LDS-DMA of the next tile's first k-tiles is issued during the last k-tiles
//     of the current one (three 48 KB stages, two whole k-tiles always in flight), so there is no pipeline drain / refill between tiles;
//   * the last k-step of a tile leaves its result in the second accumulator set; while the next tile accumulates, the finished tile is
//     drained one 32 x 32 block per k-tile: turned through a 4 KB per-wave LDS region (row-major: a lane gets 4 consecutive columns of one row,
//     8 lanes a 128-byte line), bias / residual / GELU / x GELU' applied, stored -- 4 to 8 store instructions per k-tile and wave instead of a
//     burst, each counted vmcnt wait only ever meets stores issued at least a k-tile earlier;
//   * epilogue operands (fp32 residual, 16-bit pre-activation) are fetched one block ahead by loads the compiler does not see (asm), issued at
//     the START of a window, i.e. older than the window's 12 DMA instructions: the window's one wait, vmcnt(12), covers them.
// 192 x 192 because M = 192 B for the benchmark's padded sequences: 12288 x {768, 2304, 3072} is exactly {1, 3, 4} rounds of 256 tiles (the
// 256-row tiles leave a quarter of the chip idle on the five N = 768 GEMMs of a layer), and 3 stages + the turn regions are exactly 160 KB.
//
// Structure.  Waves 2 (M) x 2 (N); wave (wr, wc) owns rows wr*96 .. +96, columns wc*96 .. +96 = [3 m-blocks][3 n-blocks] of
// v_mfma_f32_32x32x16_bf16 (144 accumulator registers per set), operands swapped as in the other NT kernels (a lane owns one output row x 4
// consecutive columns per register group).  Stage s = { A image [192 rows][64 k] | B image [192][64] }, 128-byte rows, 16-byte chunks XOR-
// swizzled by swz(row), filled by LDS-DMA (MUBUF form: per-lane offsets that never change + a scalar offset) with the swizzle on the SOURCE
// address.  A "window" = k-step 3 of k-tile T and k-steps 0 - 2 of k-tile T + 1 = 36 MFMA slots; every slot = one MFMA + its fillers, pinned
// by sched_barrier(0): fragment reads of the NEXT k-step (6 per 9 MFMAs, from the stage of k-tile T + 1), the 12 DMA pieces of k-tile T + 3
// (into the stage k-tile T just left), and the epilogue pieces of one block of the previous tile.  A window ends with
// { lgkmcnt(0); vmcnt(12); s_barrier }: every wave has read k-tile T + 1's stage for the last time, every wave's share of k-tile T + 2 has landed.
// The accumulation order over k is that of the 128 x 128 two-barrier kernel: fp32 results are bit-identical to it (race screens in
// tests/test_gpu_kernels.py).
//
// Takes: M % 192 == 0, N % 192 == 0, (M / 192) % 8 == 0, K % 64 == 0, K >= 640, epilogues NONE / GELU / RESID / DGELU.
#include "gemm_bf16_nt.h"
// A/B aid: the file is compiled twice -- as is (the product), and with -DNT4_SLP_BUILD under other names for climb_set_option(17, 2 | 5).  r04 used the
// second build for the epilogue arithmetic without SLP packing (measured equal or 2 - 5 % slower on the GELU kinds); r05: the second build is the r04
// WINDOW BOUNDARY (tile-walk bookkeeping with its branches between the barrier and the window's first MFMA), the product hides it (see nt4_dma_next).
#ifdef NT4_SLP_BUILD
#define gemm_bf16_nt4_kernel gemm_bf16_nt4slp_kernel
#define climb_nt4_launch climb_nt4slp_launch
#define NT4_OLD_BOUNDARY 1
#else
#define NT4_OLD_BOUNDARY 0
#endif

#define NT4_T 192
#define NT4_OP (NT4_T * 128)             // one operand image of a k-tile: 24 KB
#define NT4_STAGE (2 * NT4_OP)           // 48 KB
#define NT4_TURN (3 * NT4_STAGE)         // 144 KB: the four per-wave turn regions follow
#define NT4_LDS (NT4_TURN + 4 * 4096)    // 160 KB

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct Nt4Lane {            // per-lane constants
  unsigned fo[4];                // fragment read offset of k-step ks inside an operand image, row l31 (the wave's row base and the stage are scalars)
  unsigned vo[4];                // DMA source offsets: A even / odd piece, B even / odd piece
  unsigned wr0;                  // turn region, LDS byte ADDRESS of this lane's float4 of register group 0 (asm ds_write); group g: ^ (g << 5)
  unsigned rd;                   // turn region, byte offset (inside the wave's region) of this lane's row-major float4; + p * 1024
  unsigned vc, vx, vo2, vb;      // row-major lane offsets (bytes) into C / aux / aux_out / bias
};
struct Nt4Walk {            // tile walk of one stream (wave-uniform): tile = (xcd * gm + tml, tn)
  int tml, tn, gm, step, xcd;
  int dm, dn;               // step = dn * gm + dm: the walk's increment without a loop (r05)
  __device__ __forceinline__ void advance() {
#if NT4_OLD_BOUNDARY
    tml += step; while (tml >= gm) { tml -= gm; ++tn; }
#else
    const int t1 = tml + dm;
    const bool c = t1 >= gm;
    tml = c ? t1 - gm : t1;
    tn += dn + (c ? 1 : 0);
#endif
  }
  __device__ __forceinline__ int m0() const { return (xcd * gm + tml) * NT4_T; }
  __device__ __forceinline__ int n0() const { return tn * NT4_T; }
};
struct Nt4Uni {             // wave-uniform state
  __amdgpu_buffer_rsrc_t ra, rb, rc, rx, ro, rbias;
  unsigned pa[6], pb[6];         // byte offset of this wave's DMA piece j inside the tile's A / B rows
  unsigned lda2, ldb2;           // bytes per operand row
  unsigned ldc_b, ldx_b, ldo_b;  // bytes per row of C / aux / aux_out
  unsigned wid, stag;
  int probe;
  // DMA stream
  Nt4Walk dw;
  int dkt, dleft, nk;
  int ksp;                       // r06, split operands: logical k-tiles (nk = 3 ksp: A hi.B hi, A hi.B lo, A lo.B hi of each); unused otherwise
  unsigned alo, blo;             // ... and the byte distance from an operand's hi plane to its lo plane
  unsigned olo;                  // (EPI_GELU_SP / EPI_DGELU_SP) byte distance from aux_out's hi plane to its lo plane
  unsigned curA, curB;           // byte offset of the stream's k-tile: tile rows + k
  unsigned nxtA, nxtB;           // the NEXT window's (computed under this window's MFMAs, committed before its barrier)
  unsigned dma_off, rd_off;      // stage offsets: DMA destination / fragment reads of this window
  unsigned rda, rdb;             // rd_off + this wave's row base inside the A / B image
  // finished ("previous") tile: epilogue addressing
  unsigned cbase, xbase, obase, bbase;
};

__device__ __forceinline__ bf16x8 ntp_frag_(const unsigned char* p) { return as_bf16x8(*reinterpret_cast<const u32x4*>(p)); }

template <int EPI> struct Nt4Aux { };
template <> struct Nt4Aux<EPI_RESID> { f32x4 v[4]; };
template <> struct Nt4Aux<EPI_DGELU_SP> { f32x4 v[4]; };          // (r06) fp32 pre-activation, fetched like the fp32 residual
constexpr bool nt4_aux_f32(int epi) { return epi == EPI_RESID || epi == EPI_DGELU_SP; }
template <> struct Nt4Aux<EPI_DGELU> { u32x2 v[4]; };
template <> struct Nt4Aux<EPI_MUL> { u32x2 v[4]; };

// ---------------------------------------------------------------------------------------------------- loads the compiler does not count
// (a register load hipcc can see is waited for with vmcnt(0) as soon as stores are pending: DESIGN.md section 5, "a load behind a store")
__device__ __forceinline__ void nt4_ld128(f32x4& d, const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void nt4_ld64(u32x2& d, const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(d) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int EPI> __device__ __forceinline__ void nt4_aux_load(Nt4Aux<EPI>&, int, const Nt4Uni&, const Nt4Lane&, unsigned) {}
template <> __device__ __forceinline__ void nt4_aux_load<EPI_RESID>(Nt4Aux<EPI_RESID>& a, int p, const Nt4Uni& u, const Nt4Lane& l, unsigned soff) {
  nt4_ld128(a.v[p], u.rx, l.vx, soff);
}
template <> __device__ __forceinline__ void nt4_aux_load<EPI_DGELU_SP>(Nt4Aux<EPI_DGELU_SP>& a, int p, const Nt4Uni& u, const Nt4Lane& l, unsigned soff) {
  nt4_ld128(a.v[p], u.rx, l.vx, soff);
}
template <> __device__ __forceinline__ void nt4_aux_load<EPI_DGELU>(Nt4Aux<EPI_DGELU>& a, int p, const Nt4Uni& u, const Nt4Lane& l, unsigned soff) {
  nt4_ld64(a.v[p], u.rx, l.vx, soff);
}
template <> __device__ __forceinline__ void nt4_aux_load<EPI_MUL>(Nt4Aux<EPI_MUL>& a, int p, const Nt4Uni& u, const Nt4Lane& l, unsigned soff) {
  nt4_ld64(a.v[p], u.rx, l.vx, soff);
}
template <int EPI> __device__ __forceinline__ u32x2 nt4_auxw(const Nt4Aux<EPI>&, int) { return (u32x2){0u, 0u}; }
template <> __device__ __forceinline__ u32x2 nt4_auxw<EPI_DGELU>(const Nt4Aux<EPI_DGELU>& a, int p) { return a.v[p]; }
// pins the registers of asm loads behind a wait (the compiler must not touch them between load and wait)
template <int EPI> __device__ __forceinline__ void nt4_pin(Nt4Aux<EPI>&) {}
template <> __device__ __forceinline__ void nt4_pin<EPI_RESID>(Nt4Aux<EPI_RESID>& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])::"memory"); }
template <> __device__ __forceinline__ void nt4_pin<EPI_DGELU_SP>(Nt4Aux<EPI_DGELU_SP>& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])::"memory"); }
template <> __device__ __forceinline__ void nt4_pin<EPI_DGELU>(Nt4Aux<EPI_DGELU>& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])::"memory"); }
template <> __device__ __forceinline__ void nt4_pin<EPI_MUL>(Nt4Aux<EPI_MUL>& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])::"memory"); }

// ---------------------------------------------------------------------------------------------------- DMA
constexpr int nt4_dma_piece(int q) {      // slot -> piece (0..5: A, 6..11: B) or -1
  const int s[12] = {4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 33, 35};
  for (int i = 0; i < 12; ++i)
    if (s[i] == q) return i;
  return -1;
}
template <int J>
__device__ __forceinline__ void nt4_dma(const Nt4Uni& u, const Nt4Lane& l, unsigned char* smem) {
  if constexpr (J < 6)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.ra, (lds_void_t*)(smem + u.dma_off + (u.wid * 6 + J) * 1024), 16, l.vo[J & 1], u.curA + u.pa[J], 0, 0);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(u.rb, (lds_void_t*)(smem + u.dma_off + NT4_OP + (u.wid * 6 + (J - 6)) * 1024), 16, l.vo[2 + (J & 1)],
                                             u.curB + u.pb[J - 6], 0, 0);
}
// next k-tile of the DMA stream; past the workgroup's last k-tile it stays there (the same bytes again, into a stage nobody reads any more).
// r05: WHERE this runs matters more than what it costs.  With one wave per SIMD nothing covers the instructions between a window's barrier and its first
// MFMA: the r04 kernel ran this bookkeeping there, a dozen scalar instructions and four taken branches (k-tile wrap, last tile, the walk's while loop), and
// the measurement builds of option 18 put the pure MFMA stream of a launch at 43 cycles per MFMA instead of 32.  Now it is branch-free (selects), computes
// the NEXT window's offsets in a filler position of slot 20 (under the MFMAs), and nt4_window_end commits them BEFORE the barrier.
// r06 SPLIT: the stream's k-tile index d runs over 3 ksp tiles, k-tile d / 3 of A's {hi, hi, lo} and B's {hi, lo, hi} plane for d % 3 = 0, 1, 2: the three
// products of one k-tile follow each other, so the second fetch of an A hi / B hi tile hits the XCD's L2 (phase-major order -- all hi.hi tiles, then
// all hi.lo ... -- fetched every plane from the fabric once per phase: 2 x 124 MB per launch against 45 MB of planes, profiles/r06_bf16x3_*).
template <bool SPLIT = false>
__device__ __forceinline__ void nt4_dma_next(Nt4Uni& u) {
  const int d1 = u.dkt + 1;
  const bool wrap = d1 == u.nk, more = u.dleft > 1, adv = wrap && more;
  u.dkt = wrap ? (more ? 0 : u.nk - 1) : d1;
  u.dleft -= adv ? 1 : 0;
  Nt4Walk w = u.dw;
  w.advance();
  u.dw.tml = adv ? w.tml : u.dw.tml;
  u.dw.tn = adv ? w.tn : u.dw.tn;
  if constexpr (SPLIT) {
    const unsigned kq = (unsigned)u.dkt / 3u, ph = (unsigned)u.dkt - 3u * kq;
    const unsigned kk = kq * 128u;
    u.nxtA = (unsigned)u.dw.m0() * u.lda2 + kk + (ph == 2u ? u.alo : 0u);
    u.nxtB = (unsigned)u.dw.n0() * u.ldb2 + kk + (ph == 1u ? u.blo : 0u);
  } else {
    u.nxtA = (unsigned)u.dw.m0() * u.lda2 + (unsigned)u.dkt * 128u;
    u.nxtB = (unsigned)u.dw.n0() * u.ldb2 + (unsigned)u.dkt * 128u;
  }
}
__device__ __forceinline__ void nt4_dma_commit(Nt4Uni& u) { u.curA = u.nxtA; u.curB = u.nxtB; }
template <bool SPLIT = false>
__device__ __forceinline__ void nt4_dma_advance(Nt4Uni& u) {
#if NT4_OLD_BOUNDARY
  static_assert(!SPLIT, "the A/B build has no split instantiation");
  ++u.dkt;
  if (u.dkt == u.nk) {
    if (u.dleft > 1) { --u.dleft; u.dkt = 0; u.dw.advance(); }
    else u.dkt = u.nk - 1;
  }
  u.curA = (unsigned)u.dw.m0() * u.lda2 + (unsigned)u.dkt * 128u;
  u.curB = (unsigned)u.dw.n0() * u.ldb2 + (unsigned)u.dkt * 128u;
#else
  nt4_dma_next<SPLIT>(u);
  nt4_dma_commit(u);
#endif
}

// ---------------------------------------------------------------------------------------------------- epilogue pieces of one 32 x 32 block
template <typename TO> __device__ __forceinline__ void nt4_store(const f32x4& v, const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ void nt4_store<float>(const f32x4& v, const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
  asm volatile("s_nop 0" ::"v"(v));          // r06: v's registers stay live over one wait state (tools/check_store_hazard.py: the compiler adds none behind a store with an SGPR offset)
}
template <> __device__ __forceinline__ void nt4_store<bf16_t>(const f32x4& v, const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) {
  u32x2 h = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
  __builtin_amdgcn_raw_buffer_store_b64(h, r, voff, soff, 0);
}
template <int G>
__device__ __forceinline__ void nt4_stage_write(const f32x16& acc, const Nt4Lane& l) {
  f32x4 v = {acc[4 * G], acc[4 * G + 1], acc[4 * G + 2], acc[4 * G + 3]};
  asm volatile("ds_write_b128 %0, %1" ::"v"(l.wr0 ^ (unsigned)(G << 5)), "v"(v) : "memory");
}
// The GELU kinds inside the k-loop.  gelu_fast / dgelu_fast / gelu_fast_pair (common.h, the sigmoid form) are chains of dependent operations: one
// element per MFMA slot would issue at the chain's latency (measured with the r04 polynomials: the windows that carried them took 3 x the MFMA time).
// Instead ONE stage of the chain runs per slot for EIGHT elements (two readback pieces), so the eight chains interleave, and the second, independent
// chain of the derivative (q) rides in the stages whose first chain is a transcendental.  The arithmetic per element is exactly that of common.h
// (same operations in the same order: bit-identical to the 8-wave kernels' epilogues).  Stages: GELU 0..7, GELUD 0..8, DGELU 0..10.
struct Nt4Act { float u[8], t[8], p[8], q[8]; };
template <int EPI> struct Nt4ActN { static constexpr int STAGES = EPI == EPI_GELU ? 8 : EPI == EPI_GELUD ? 9 : 11; };
template <int EPI, int STAGE>
__device__ __forceinline__ void nt4_act_stage(Nt4Act& a, f32x4& v0, f32x4& v1, const u32x2& w0, const u32x2& w1) {
  constexpr bool D = EPI == EPI_DGELU;          // pre-activation comes packed (w0, w1) and v is multiplied by the derivative at the end
  constexpr int S = D ? STAGE - 1 : STAGE;      // position in the common chain
#define NT4_ALL _Pragma("unroll") for (int k = 0; k < 8; ++k)
  if constexpr (D && STAGE == 0) {
    a.u[0] = h16lo_to_f32(w0[0]); a.u[1] = h16hi_to_f32(w0[0]); a.u[2] = h16lo_to_f32(w0[1]); a.u[3] = h16hi_to_f32(w0[1]);
    a.u[4] = h16lo_to_f32(w1[0]); a.u[5] = h16hi_to_f32(w1[0]); a.u[6] = h16lo_to_f32(w1[1]); a.u[7] = h16hi_to_f32(w1[1]);
  }
  if constexpr (S == 0) {
    if constexpr (!D) { NT4_ALL a.u[k] = k < 4 ? v0[k & 3] : v1[k & 3]; }
    NT4_ALL a.t[k] = gelu_sig_t(a.u[k]);
  }
  if constexpr (S == 1) { NT4_ALL a.p[k] = fmaf(GELU_SIG_NL2E_C, a.t[k], GELU_SIG_NL2E_B); }
  if constexpr (S == 2) { NT4_ALL a.p[k] = fmaf(a.p[k], a.t[k], GELU_SIG_NL2E_A); }
  if constexpr (S == 3) { NT4_ALL a.p[k] = a.u[k] * a.p[k]; }
  if constexpr (S == 4) {
    NT4_ALL a.p[k] = __builtin_amdgcn_exp2f(a.p[k]);
    if constexpr (EPI != EPI_GELU) { NT4_ALL a.q[k] = fmaf(GELU_SIG_DC, a.t[k], GELU_SIG_DB); }
  }
  if constexpr (S == 5) {
    NT4_ALL a.p[k] = 1.0f + a.p[k];
    if constexpr (EPI != EPI_GELU) { NT4_ALL a.q[k] = fmaf(a.q[k], a.t[k], GELU_SIG_DA); }
  }
  if constexpr (S == 6) {
    NT4_ALL a.p[k] = __builtin_amdgcn_rcpf(a.p[k]);          // s
    if constexpr (EPI != EPI_GELU) { NT4_ALL a.q[k] = a.u[k] * a.q[k]; }
  }
  if constexpr (S == 7) {
    if constexpr (!D) { NT4_ALL { if (k < 4) v0[k & 3] = a.u[k] * a.p[k]; else v1[k & 3] = a.u[k] * a.p[k]; } }
    if constexpr (EPI != EPI_GELU) { NT4_ALL a.t[k] = fmaf(-a.p[k], a.p[k], a.p[k]); }          // s - s^2 (t is dead)
  }
  if constexpr (S == 8 && EPI != EPI_GELU) { NT4_ALL a.q[k] = fmaf(a.q[k], a.t[k], a.p[k]); }   // the derivative
  if constexpr (D && S == 9) { NT4_ALL { if (k < 4) v0[k & 3] = v0[k & 3] * a.q[k]; else v1[k & 3] = v1[k & 3] * a.q[k]; } }
#undef NT4_ALL
}
__device__ __forceinline__ u32x2 nt4_pack4(float a, float b, float c, float d) { return (u32x2){pack_bf16x2(a, b), pack_bf16x2(c, d)}; }

// one row-major float4 (readback p of block (bi, bj) of the finished tile): bias, epilogue, store(s).  SUB: -1 = everything at once.  Inside the
// k-loop the GELU kinds spread their arithmetic over MFMA slots and HOLD the packed result for the next window (stores issued late in a window are
// still unacknowledged at its end, and the window's counted wait then waits for them -- measured with cold operands: +27 us on the up-projection):
// 0 = bias (+ the pre-activation store), 1..4 = element SUB - 1 (4 also packs into `hold`), 5 = store `hold`
// r05: which kinds hold their packed results for the next window's first slots.  r04: the GELU kinds only (their stores would otherwise sit late in the
// window).  The measurement builds say the same of every 16-bit kind: stores of slots 9 - 15 can still be unacknowledged at the window's end, and its
// counted wait, vmcnt(12), cannot tell them from the DMA pieces it is about (loads and stores share the counter and complete out of order relative to
// each other, so it must assume the stores are the ones still pending): +7 us on the QKV product with the drain in the windows against 37 without.
template <typename TO, int EPI> constexpr bool nt4_holds() {
  return EPI == EPI_GELU || EPI == EPI_DGELU || EPI == EPI_GELUD || (!NT4_OLD_BOUNDARY && sizeof(TO) == 2 && (EPI == EPI_NONE || EPI == EPI_MUL));
}
template <typename TO, int EPI, int SUB>
__device__ __forceinline__ void nt4_unit(f32x4& v, u32x2& hold, const f32x4& bias, const Nt4Aux<EPI>& ax, int p, int bi, int bj, const Nt4Uni& u, const Nt4Lane& l) {
  const unsigned rowoff = (unsigned)(bi * 32 + 8 * p);
  const unsigned csoff = u.cbase + rowoff * u.ldc_b + (unsigned)(bj * 32 * (int)sizeof(TO));
  if constexpr (SUB == 5) {                  // the held 16-bit result of an earlier window
    __builtin_amdgcn_raw_buffer_store_b64(hold, u.rc, l.vc, csoff, 0);
  } else if constexpr (EPI == EPI_NONE) {
    v += bias;
    if constexpr (SUB == 7) hold = nt4_pack4(v.x, v.y, v.z, v.w);
    else nt4_store<TO>(v, u.rc, l.vc, csoff);
  } else if constexpr (EPI == EPI_RESID) {
    v += bias;
    v += ax.v[p];
    nt4_store<TO>(v, u.rc, l.vc, csoff);
  } else if constexpr (EPI == EPI_GELU_SP || EPI == EPI_DGELU_SP) {
    // r06, split-operand mode: the activation's (hi, lo) operand planes leave from here (the pass that read the fp32 result back is gone)
    static_assert(sizeof(TO) == 4, "split-mode epilogues have fp32 C");
    const unsigned osoff = u.obase + rowoff * u.ldo_b + (unsigned)(bj * 64);
    if constexpr (EPI == EPI_GELU_SP) {
      v += bias;
      nt4_store<float>(v, u.rc, l.vc, csoff);          // u: the backward evaluates gelu' on it
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = gelu_as(v[k]);
    } else {
      const f32x4 uu = ax.v[p];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= dgelu_as(uu[k]);
    }
    u32x2 h = {pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
    u32x2 lo = {pack_bf16x2(v.x - h16lo_to_f32(h[0]), v.y - h16hi_to_f32(h[0])), pack_bf16x2(v.z - h16lo_to_f32(h[1]), v.w - h16hi_to_f32(h[1]))};
    __builtin_amdgcn_raw_buffer_store_b64(h, u.ro, l.vo2, osoff, 0);
    __builtin_amdgcn_raw_buffer_store_b64(lo, u.ro, l.vo2, osoff + u.olo, 0);
  } else if constexpr (EPI == EPI_MUL) {
    const u32x2 w = ax.v[p];
    v[0] *= h16lo_to_f32(w[0]); v[1] *= h16hi_to_f32(w[0]); v[2] *= h16lo_to_f32(w[1]); v[3] *= h16hi_to_f32(w[1]);
    if constexpr (SUB == 7) hold = nt4_pack4(v.x, v.y, v.z, v.w);
    else nt4_store<TO>(v, u.rc, l.vc, csoff);
  } else {
    static_assert(EPI == EPI_GELU || EPI == EPI_DGELU || EPI == EPI_GELUD, "epilogue");
    static_assert(sizeof(TO) == 2, "the GELU kinds have 16-bit outputs");
    const unsigned osoff = u.obase + rowoff * u.ldo_b + (unsigned)(bj * 64);
    if constexpr (SUB <= 0) {
      v += bias;
      if constexpr (EPI == EPI_GELU) nt4_store<bf16_t>(v, u.ro, l.vo2, osoff);
    }
    if constexpr (SUB < 0) {
      if constexpr (EPI == EPI_GELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gelu_fast(v[k]);
      } else if constexpr (EPI == EPI_GELUD) {
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float y_, d_;
          gelu_fast_pair(v[k], y_, d_);
          v[k] = y_;
          d[k] = d_;
        }
        nt4_store<bf16_t>(d, u.ro, l.vo2, osoff);
      } else {
        const u32x2 w = ax.v[p];
        v[0] *= dgelu_fast(h16lo_to_f32(w[0])); v[1] *= dgelu_fast(h16hi_to_f32(w[0]));
        v[2] *= dgelu_fast(h16lo_to_f32(w[1])); v[3] *= dgelu_fast(h16hi_to_f32(w[1]));
      }
      nt4_store<TO>(v, u.rc, l.vc, csoff);
    }
    if constexpr (SUB == 6) __builtin_amdgcn_raw_buffer_store_b64(hold, u.ro, l.vo2, osoff, 0);          // GELUD: the held derivative
  }
}
// operand loads of block B of the finished tile, piece p
template <int EPI>
__device__ __forceinline__ void nt4_aux_issue(Nt4Aux<EPI>& ax, int p, int bi, int bj, const Nt4Uni& u, const Nt4Lane& l) {
  if constexpr (nt4_aux_f32(EPI)) nt4_aux_load<EPI>(ax, p, u, l, u.xbase + (unsigned)(bi * 32 + 8 * p) * u.ldx_b + (unsigned)(bj * 128));
  if constexpr (EPI == EPI_DGELU || EPI == EPI_MUL) nt4_aux_load<EPI>(ax, p, u, l, u.xbase + (unsigned)(bi * 32 + 8 * p) * u.ldx_b + (unsigned)(bj * 64));
}
__device__ __forceinline__ void nt4_bias_issue(f32x4 (&bias)[3], int j, const Nt4Uni& u, const Nt4Lane& l) { nt4_ld128(bias[j], u.rbias, l.vb, u.bbase + (unsigned)(j * 128)); }
// coordinates of the tile that has just been finished -> epilogue bases
template <typename TO, int EPI>
__device__ __forceinline__ void nt4_set_prev(Nt4Uni& u, int m0, int n0) {
  const unsigned wr = u.wid >> 1, wc = u.wid & 1;
  const unsigned mrow = (unsigned)m0 + wr * 96u, ncol = (unsigned)n0 + wc * 96u;
  u.cbase = mrow * u.ldc_b + ncol * (unsigned)sizeof(TO);
  u.xbase = mrow * u.ldx_b + ncol * (nt4_aux_f32(EPI) ? 4u : 2u);
  u.obase = mrow * u.ldo_b + ncol * 2u;
  u.bbase = ncol * 4u;
}

// ---------------------------------------------------------------------------------------------------- one MFMA slot
// CFG: SW   switch window (k-step 3 finishes a tile INTO accP, k-step 0 starts the next one from zero)
//      EB   block of the finished tile drained in this window (-1: none);  AB: block whose operands are fetched (-1: none);  BL: bias loads
//      DMA  the 12 pieces of k-tile T + 3;  RD: fragment reads (off in the workgroup's very last k-step)
//      SB   (GELU kinds) block whose results, held in packed form since the previous window, are stored in the first slots of this one
//      PR   measurement builds (climb_set_option 18, 16-bit NONE kernel only; results are WRONG with bits 1 / 2): 1 = no DMA inside the windows,
//           2 = no epilogue inside the windows, 4 = the 12 DMA pieces in slots 4 .. 15 (issued as early as the stage is free),
//           8 = no workgroup barrier at the end of a window, 16 = no fragment reads (r05: with 1 | 2 these leave the bare MFMA stream)
template <int PR_, bool SW_, int EB_, int AB_, bool BL_, bool DMA_, bool RD_, int SB_ = -1> struct Nt4Cfg {
  static constexpr bool SW = SW_, BL = BL_ && !(PR_ & 2), DMA = DMA_ && !(PR_ & 1), RD = RD_ && !(PR_ & 16), STAG = (PR_ & 4) != 0, NOBAR = (PR_ & 8) != 0;
  static constexpr int EB = (PR_ & 2) ? -1 : EB_, AB = (PR_ & 2) ? -1 : AB_, SB = (PR_ & 2) ? -1 : SB_;
  static constexpr bool SPLIT = (PR_ & 32) != 0;          // r06: three-phase reduction over (hi, lo) operand planes (not a measurement build)
};
template <typename TO, int EPI, class CFG, int Q>
__device__ __forceinline__ void nt4_slot(f32x16 (&accC)[3][3], f32x16 (&accP)[3][3], bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], f32x4 (&rbk)[4], u32x2 (&hold)[8], Nt4Act& act,
                                         f32x4 (&bias)[3], Nt4Aux<EPI> (&aux)[2], Nt4Uni& u, const Nt4Lane& l, unsigned char* smem) {
  constexpr int st = Q / 9, blk = Q % 9, bi = blk / 3, bj = blk % 3;
  constexpr int par = (st == 0) ? 1 : ((st - 1) & 1);
  if constexpr (CFG::SW && st == 0) accP[bi][bj] = CLIMB_MFMA_H16(fb[par][bj], fa[par][bi], accC[bi][bj], 0, 0, 0);
  else if constexpr (CFG::SW && st == 1) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    accC[bi][bj] = CLIMB_MFMA_H16(fb[par][bj], fa[par][bi], z, 0, 0, 0);
  } else accC[bi][bj] = CLIMB_MFMA_H16(fb[par][bj], fa[par][bi], accC[bi][bj], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // fragment reads of the next k-step (k-step `st` of the k-tile this window reads), in the order the next k-step consumes them
  if constexpr (CFG::RD && blk < 6) {
    if constexpr (blk == 0) fb[par ^ 1][0] = ntp_frag_(smem + (u.rdb + l.fo[st]));
    if constexpr (blk == 1) fa[par ^ 1][0] = ntp_frag_(smem + (u.rda + l.fo[st]));
    if constexpr (blk == 2) fb[par ^ 1][1] = ntp_frag_(smem + (u.rdb + l.fo[st]) + 4096);
    if constexpr (blk == 3) fb[par ^ 1][2] = ntp_frag_(smem + (u.rdb + l.fo[st]) + 8192);
    if constexpr (blk == 4) fa[par ^ 1][1] = ntp_frag_(smem + (u.rda + l.fo[st]) + 4096);
    if constexpr (blk == 5) fa[par ^ 1][2] = ntp_frag_(smem + (u.rda + l.fo[st]) + 8192);
  }
  // switch window: block b of the finished tile leaves the accumulator file (MFMA results are AGPRs) one slot before block b of the next tile
  // starts there -- otherwise both sets are live in the AGPR file at once (288 > 256: scratch spills, measured)
  if constexpr (CFG::SW && CFG::RD && Q >= 8 && Q < 17) asm volatile("" : "+v"(accP[(Q - 8) / 3][(Q - 8) % 3]));
  // operand loads of the block drained in the NEXT window (older than every DMA piece of this window: slots 0..3, first piece in slot 4)
  if constexpr (Q < 4 && CFG::AB >= 0) nt4_aux_issue<EPI>(aux[CFG::AB & 1], Q, CFG::AB / 3, CFG::AB % 3, u, l);
  if constexpr (Q < 3 && CFG::BL) nt4_bias_issue(bias, Q, u, l);
  if constexpr (CFG::DMA && !CFG::STAG && nt4_dma_piece(Q) >= 0) nt4_dma<(nt4_dma_piece(Q) < 0 ? 0 : nt4_dma_piece(Q))>(u, l, smem);
  if constexpr (CFG::DMA && CFG::STAG && Q >= 4 && Q < 16) nt4_dma<(Q >= 4 && Q < 16 ? Q - 4 : 0)>(u, l, smem);      // every piece early in the window
#if !NT4_OLD_BOUNDARY
  if constexpr (CFG::DMA && Q == 20) nt4_dma_next<CFG::SPLIT>(u);          // the next window's stream position (scalar selects, under this slot's MFMA)
#endif
  if constexpr (CFG::EB >= 0) {
    constexpr int ei = CFG::EB / 3, ej = CFG::EB % 3;
    // turn: 4 writes, then 4 row-major reads (LDS operations of one wave execute in order: no wait in between)
    if constexpr (Q == 1) nt4_stage_write<0>(accP[ei][ej], l);
    if constexpr (Q == 2) nt4_stage_write<1>(accP[ei][ej], l);
    if constexpr (Q == 3) nt4_stage_write<2>(accP[ei][ej], l);
    if constexpr (Q == 5) nt4_stage_write<3>(accP[ei][ej], l);
    if constexpr (EPI == EPI_GELU || EPI == EPI_DGELU || EPI == EPI_GELUD) {
      // readback p in slot 5 + p; bias (+ pre-activation store, GELU) of piece p in slot 8 + p; pieces 0, 1: one stage of the activation per slot from
      // 12 on (at most 11 stages), packed into `hold` in 23; pieces 2, 3: from 24 on, packed in 35.  NO result store in this window: held for the
      // next one's first slots
      constexpr int NS = Nt4ActN<EPI>::STAGES;
      if constexpr (Q >= 5 && Q < 9) rbk[Q - 5] = *reinterpret_cast<const f32x4*>(smem + NT4_TURN + u.wid * 4096 + l.rd + (Q - 5) * 1024);
      if constexpr (Q >= 8 && Q < 12) nt4_unit<TO, EPI, 0>(rbk[Q - 8], hold[Q - 8], bias[ej], aux[CFG::EB & 1], Q - 8, ei, ej, u, l);
      if constexpr (Q >= 12 && Q < 12 + NS) nt4_act_stage<EPI, Q - 12>(act, rbk[0], rbk[1], nt4_auxw<EPI>(aux[CFG::EB & 1], 0), nt4_auxw<EPI>(aux[CFG::EB & 1], 1));
      if constexpr (Q >= 24 && Q < 24 + NS) nt4_act_stage<EPI, Q - 24>(act, rbk[2], rbk[3], nt4_auxw<EPI>(aux[CFG::EB & 1], 2), nt4_auxw<EPI>(aux[CFG::EB & 1], 3));
      if constexpr (Q == 23 || Q == 35) {
        constexpr int h = Q == 23 ? 0 : 2;
        hold[h] = nt4_pack4(rbk[h].x, rbk[h].y, rbk[h].z, rbk[h].w);
        hold[h + 1] = nt4_pack4(rbk[h + 1].x, rbk[h + 1].y, rbk[h + 1].z, rbk[h + 1].w);
        if constexpr (EPI == EPI_GELUD) {
          hold[4 + h] = nt4_pack4(act.q[0], act.q[1], act.q[2], act.q[3]);
          hold[5 + h] = nt4_pack4(act.q[4], act.q[5], act.q[6], act.q[7]);
        }
      }
    } else {
      // every store in the first half of the window: readback p in slot 6 + 2 p, bias / residual / store in slot 9 + 2 p
      constexpr int rp = Q == 6 ? 0 : Q == 8 ? 1 : Q == 10 ? 2 : Q == 12 ? 3 : -1;
      constexpr int p = Q == 9 ? 0 : Q == 11 ? 1 : Q == 13 ? 2 : Q == 15 ? 3 : -1;
      if constexpr (rp >= 0) rbk[rp] = *reinterpret_cast<const f32x4*>(smem + NT4_TURN + u.wid * 4096 + l.rd + rp * 1024);
      if constexpr (p >= 0) nt4_unit<TO, EPI, (nt4_holds<TO, EPI>() ? 7 : -1)>(rbk[p], hold[p], bias[ej], aux[CFG::EB & 1], p, ei, ej, u, l);
    }
  }
  if constexpr (CFG::SB >= 0 && Q < 4) nt4_unit<TO, EPI, 5>(rbk[Q], hold[Q], bias[0], aux[0], Q, CFG::SB / 3, CFG::SB % 3, u, l);
  if constexpr (EPI == EPI_GELUD && CFG::SB >= 0 && Q >= 4 && Q < 8) nt4_unit<TO, EPI, 6>(rbk[Q - 4], hold[Q], bias[0], aux[0], Q - 4, CFG::SB / 3, CFG::SB % 3, u, l);
  __builtin_amdgcn_sched_barrier(0);
}
template <typename TO, int EPI, class CFG, int Q, int QE>
__device__ __forceinline__ void nt4_slots(f32x16 (&accC)[3][3], f32x16 (&accP)[3][3], bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], f32x4 (&rbk)[4], u32x2 (&hold)[8], Nt4Act& act,
                                          f32x4 (&bias)[3], Nt4Aux<EPI> (&aux)[2], Nt4Uni& u, const Nt4Lane& l, unsigned char* smem) {
  if constexpr (Q < QE) {
    nt4_slot<TO, EPI, CFG, Q>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
    nt4_slots<TO, EPI, CFG, Q + 1, QE>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
  }
}
// end of a window: this wave has read the stage of k-tile T + 1 for the last time and its share of k-tile T + 2 has landed (only the VM youngest
// loads -- the window's DMA pieces -- may still be in flight; everything older, the asm operand loads included, has returned); then everybody's
__device__ __forceinline__ void nt4_set_rd(Nt4Uni& u, unsigned off) {
  u.rd_off = off;
  u.rda = off + (u.wid >> 1) * (96u * 128u);
  u.rdb = off + NT4_OP + (u.wid & 1) * (96u * 128u);
}
template <int VM, int EPI, bool NOBAR = false>
__device__ __forceinline__ void nt4_window_end(Nt4Uni& u, Nt4Aux<EPI> (&aux)[2], f32x4 (&bias)[3]) {
  __builtin_amdgcn_sched_barrier(0);
#if !NT4_OLD_BOUNDARY
  // everything the next window's first slots need, BEFORE the waits and the barrier (nothing of this window uses these any more: the last DMA piece
  // went out in slot 35, the last fragment read in slot 32); a wave that arrives early does this while it would wait anyway
  nt4_dma_commit(u);
  u.dma_off = u.rd_off;
  nt4_set_rd(u, u.rd_off + NT4_STAGE == 3 * NT4_STAGE ? 0u : u.rd_off + NT4_STAGE);
  asm volatile("" : "+s"(u.curA), "+s"(u.curB), "+s"(u.dma_off), "+s"(u.rda), "+s"(u.rdb));          // materialised here, not after the barrier
  __builtin_amdgcn_sched_barrier(0);
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
  nt4_pin<EPI>(aux[0]);
  nt4_pin<EPI>(aux[1]);
  asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2])::"memory");
  if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#if NT4_OLD_BOUNDARY
  u.dma_off = u.rd_off;
  nt4_set_rd(u, u.rd_off + NT4_STAGE == 3 * NT4_STAGE ? 0u : u.rd_off + NT4_STAGE);
#endif
}
template <typename TO, int EPI, class CFG>
__device__ __forceinline__ void nt4_window(f32x16 (&accC)[3][3], f32x16 (&accP)[3][3], bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], f32x4 (&rbk)[4], u32x2 (&hold)[8], Nt4Act& act,
                                           f32x4 (&bias)[3], Nt4Aux<EPI> (&aux)[2], Nt4Uni& u, const Nt4Lane& l, unsigned char* smem) {
#if NT4_OLD_BOUNDARY
  nt4_dma_advance<CFG::SPLIT>(u);
#endif
  nt4_slots<TO, EPI, CFG, 0, 36>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
  nt4_window_end<12, EPI, CFG::NOBAR>(u, aux, bias);
}

template <typename TO, int EPI, int E0, int E1>
__device__ __forceinline__ void nt4_final(f32x16 (&accP)[3][3], f32x4 (&rbk)[4], u32x2 (&hold)[8], f32x4 (&bias)[3], const Nt4Uni& u, const Nt4Lane& l, unsigned char* smem) {
  Nt4Aux<EPI> ax[E1 - E0];
#pragma unroll
  for (int e = E0; e < E1; ++e)
#pragma unroll
    for (int p = 0; p < 4; ++p) nt4_aux_issue<EPI>(ax[e - E0], p, e / 3, e % 3, u, l);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int e = E0; e < E1; ++e) nt4_pin<EPI>(ax[e - E0]);
  asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2])::"memory");
#pragma unroll
  for (int e = E0; e < E1; ++e) {
    nt4_stage_write<0>(accP[e / 3][e % 3], l); nt4_stage_write<1>(accP[e / 3][e % 3], l);
    nt4_stage_write<2>(accP[e / 3][e % 3], l); nt4_stage_write<3>(accP[e / 3][e % 3], l);
#pragma unroll
    for (int p = 0; p < 4; ++p) rbk[p] = *reinterpret_cast<const f32x4*>(smem + NT4_TURN + u.wid * 4096 + l.rd + p * 1024);
#pragma unroll
    for (int p = 0; p < 4; ++p) nt4_unit<TO, EPI, -1>(rbk[p], hold[p], bias[e % 3], ax[e - E0], p, e / 3, e % 3, u, l);
  }
}

template <typename TO, int EPI, int PROBE = 0>
__global__ __launch_bounds__(256) void gemm_bf16_nt4_kernel(const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K, const float* bias_g,
                                                            const void* aux_g, long ldaux, bf16_t* aux_out, long ldauxo, int probe, unsigned a_lo_b = 0, unsigned b_lo_b = 0, unsigned o_lo_b = 0) {
  constexpr bool SPLIT = (PROBE & 32) != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  Nt4Uni u;
  Nt4Lane l;
  u.wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int nbm = M / NT4_T, nbn = N / NT4_T, nwg = nbm * nbn, G = gridDim.x;
  const int ntw = (nwg - (int)blockIdx.x + G - 1) / G;          // tiles of this workgroup (>= 1)
  u.ksp = K / GB_BK;
  u.nk = SPLIT ? 3 * u.ksp : u.ksp;
  u.alo = a_lo_b;
  u.blo = b_lo_b;
  u.olo = o_lo_b;
  u.probe = probe;          // (r04's run-time vmcnt experiments are gone; the compile-time measurement builds remain)
  u.stag = u.wid % 3u;
  // resources: raw buffers (stride 0), range = 2 GB (the launcher checks sizes); a missing bias reads as zeros through an empty range
  u.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
  u.rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
  u.rc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0x7fffffff, 0x00020000);
  u.rx = __builtin_amdgcn_make_buffer_rsrc((void*)aux_g, 0, 0x7fffffff, 0x00020000);
  u.ro = __builtin_amdgcn_make_buffer_rsrc((void*)aux_out, 0, 0x7fffffff, 0x00020000);
  u.rbias = __builtin_amdgcn_make_buffer_rsrc((void*)bias_g, 0, bias_g ? N * 4 : 0, 0x00020000);
  u.lda2 = (unsigned)lda * 2u;
  u.ldb2 = (unsigned)ldb * 2u;
  u.ldc_b = (unsigned)ldc * (unsigned)sizeof(TO);
  u.ldx_b = (unsigned)ldaux * (nt4_aux_f32(EPI) ? 4u : 2u);
  u.ldo_b = (unsigned)ldauxo * 2u;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    u.pa[j] = (u.wid * 48u + 8u * j) * u.lda2;
    u.pb[j] = (u.wid * 48u + 8u * j) * u.ldb2;
  }
  // lane constants
  {
    const int r8 = lane >> 3, sl = lane & 7;
    const int ce = sl ^ swz(r8), co = sl ^ swz(8 + r8);
    l.vo[0] = (unsigned)r8 * u.lda2 + ce * 16;
    l.vo[1] = (unsigned)r8 * u.lda2 + co * 16;
    l.vo[2] = (unsigned)r8 * u.ldb2 + ce * 16;
    l.vo[3] = (unsigned)r8 * u.ldb2 + co * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) l.fo[ks] = l31 * 128 + (((2 * ks + half) ^ swz(l31)) << 4);
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)smem;
    l.wr0 = base + NT4_TURN + u.wid * 4096 + l31 * 128 + ((half ^ (l31 & 7)) << 4);
    l.rd = r8 * 128 + (((sl ^ r8) & 7) << 4);
    l.vc = (unsigned)r8 * u.ldc_b + sl * 4 * (unsigned)sizeof(TO);
    l.vx = (unsigned)r8 * u.ldx_b + sl * (nt4_aux_f32(EPI) ? 16u : 8u);
    l.vo2 = (unsigned)r8 * u.ldo_b + sl * 8u;
    l.vb = sl * 16u;
  }
  // tile walk: launch position id -> XCD id % 8, whose share is ONE supertile of gm = nbm / 8 M-tiles x all N-tiles, walked M first
  Nt4Walk cw;
  cw.gm = nbm / 8;
  cw.step = G / 8;
  cw.dn = cw.step / cw.gm;
  cw.dm = cw.step - cw.dn * cw.gm;
  cw.xcd = blockIdx.x & 7;
  cw.tn = (blockIdx.x >> 3) / cw.gm;
  cw.tml = (blockIdx.x >> 3) - cw.tn * cw.gm;
  u.dw = cw;
  u.dleft = ntw;
  u.dkt = -1;
  u.dma_off = 0;
  f32x16 accC[3][3], accP[3][3];
  bf16x8 fa[2][3], fb[2][3];
  f32x4 rbk[4], bias[3];
  u32x2 hold[8];          // packed results held for the next window: [0..3] C, [4..7] the derivative (GELUD)
  Nt4Act act;
  Nt4Aux<EPI> aux[2];
#pragma unroll
  for (int j = 0; j < 3; ++j) bias[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // prologue: k-tiles 0, 1, 2 of the stream into stages 0, 1, 2
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    nt4_dma_advance<SPLIT>(u);
    u.dma_off = s * NT4_STAGE;
    nt4_dma<0>(u, l, smem); nt4_dma<1>(u, l, smem); nt4_dma<2>(u, l, smem); nt4_dma<3>(u, l, smem); nt4_dma<4>(u, l, smem); nt4_dma<5>(u, l, smem);
    nt4_dma<6>(u, l, smem); nt4_dma<7>(u, l, smem); nt4_dma<8>(u, l, smem); nt4_dma<9>(u, l, smem); nt4_dma<10>(u, l, smem); nt4_dma<11>(u, l, smem);
  }
#if !NT4_OLD_BOUNDARY
  nt4_dma_next<SPLIT>(u);                                     // the first full window's k-tile (committed by the half window's end below)
#endif
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");          // k-tile 0 is all this half window reads
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  nt4_set_rd(u, 0);
  fb[0][0] = ntp_frag_(smem + (u.rdb + l.fo[0]));
  fa[0][0] = ntp_frag_(smem + (u.rda + l.fo[0]));
  fb[0][1] = ntp_frag_(smem + (u.rdb + l.fo[0]) + 4096);
  fb[0][2] = ntp_frag_(smem + (u.rdb + l.fo[0]) + 8192);
  fa[0][1] = ntp_frag_(smem + (u.rda + l.fo[0]) + 4096);
  fa[0][2] = ntp_frag_(smem + (u.rda + l.fo[0]) + 8192);
  // the second half of a switch window: k-steps 0 - 2 of the first tile's k-tile 0 (no DMA: k-tile 2 is already on its way)
  nt4_slots<TO, EPI, Nt4Cfg<PROBE, true, -1, -1, false, false, true>, 9, 36>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
  nt4_window_end<0, EPI>(u, aux, bias);           // k-tiles 1 and 2 have landed
  u.dma_off = 0;                                  // the stage k-tile 0 leaves at the end of the next window's k-step 3 ... which IS where piece 0 is issued (slot 4 of k-step 3: after this barrier every wave holds its k-step-3 fragments)
  const int nk = u.nk;
  constexpr bool HOLD = nt4_holds<TO, EPI>();      // results held one window (see nt4_unit)
  for (int t = 0;; ++t) {
    int w0 = 0;
    if (t > 0) {
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 0, 1, false, true, true>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 1, 2, false, true, true, HOLD ? 0 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 2, 3, false, true, true, HOLD ? 1 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 3, 4, false, true, true, HOLD ? 2 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 4, 5, false, true, true, HOLD ? 3 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 5, 6, false, true, true, HOLD ? 4 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 6, 7, false, true, true, HOLD ? 5 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 7, 8, false, true, true, HOLD ? 6 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      nt4_window<TO, EPI, Nt4Cfg<PROBE, false, 8, -1, false, true, true, HOLD ? 7 : -1>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      w0 = 9;
      if constexpr (HOLD) {          // the last block's results
        nt4_window<TO, EPI, Nt4Cfg<PROBE, false, -1, -1, false, true, true, 8>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
        w0 = 10;
      }
    }
    for (int w = w0; w + 1 < nk; ++w) nt4_window<TO, EPI, Nt4Cfg<PROBE, false, -1, -1, false, true, true>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
    // the tile is finished by the next k-step 3: its coordinates become the epilogue's
    nt4_set_prev<TO, EPI>(u, cw.m0(), cw.n0());
    if (t + 1 < ntw) {
      nt4_window<TO, EPI, Nt4Cfg<PROBE, true, -1, 0, true, true, true>>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      cw.advance();
    } else {
      nt4_slots<TO, EPI, Nt4Cfg<PROBE, true, -1, -1, false, false, false>, 0, 9>(accC, accP, fa, fb, rbk, hold, act, bias, aux, u, l, smem);
      break;
    }
  }
  // the workgroup's last tile: nothing left to hide under.  Operands first (no store is pending that a load could queue behind), one wait,
  // then the blocks
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) nt4_bias_issue(bias, j, u, l);
#if NT4_OLD_BOUNDARY
  nt4_final<TO, EPI, 0, (EPI == EPI_RESID ? 5 : 9)>(accP, rbk, hold, bias, u, l, smem);
  if constexpr (EPI == EPI_RESID) nt4_final<TO, EPI, 5, 9>(accP, rbk, hold, bias, u, l, smem);
#else
  // r05: ONE round for the fp32 residual too (144 operand registers: the k-loop's fragment and staging registers are dead here, the kernel's peak
  // stays at 256 without scratch) -- the second round's loads used to queue behind the first round's stores (a vmcnt(0) drain in mid-epilogue)
  nt4_final<TO, EPI, 0, 9>(accP, rbk, hold, bias, u, l, smem);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup's LDS allocation
}

#ifndef NT4_SLP_BUILD
int g_nt4 = 1;             // climb_set_option 17: 0 = never; 1 (default) = every epilogue but GELU / GELUD -- inside a step (cold operands) the 8-wave kernel was the
                           // faster one there with the r04 polynomial: 83.1 vs 89.5 us, the step 10.37 vs 10.43 ms; 3 = those too; 4 = GELUD but not GELU;
                           // 2 = everything on the second build (A/B: the r04 window boundary), 5 = like 1 on the second build
int g_nt4_probe = 0;       // climb_set_option 18 (measurement)
int g_nt4_grid = 256;      // follows climb_set_option 9 (CUs left to RCCL)
void climb_nt4_set_probe(int v) { g_nt4_probe = v; }
void climb_nt4_set(int v) { g_nt4 = v; }
void climb_nt4_set_grid(int v) { g_nt4_grid = v; }
int climb_nt4slp_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                        const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st);
#else
extern int g_nt4, g_nt4_probe, g_nt4_grid;
#endif

template <typename TO, int EPI, int PROBE = 0>
static int nt4_launch_one(int nwg, hipStream_t st, const bf16_t* A, long lda, const bf16_t* B, long ldb, TO* C, long ldc, int M, int N, int K, const float* bias,
                          const void* aux, long ldaux, bf16_t* aux_out, long ldauxo) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt4_kernel<TO, EPI, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, NT4_LDS);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  hipLaunchKernelGGL((gemm_bf16_nt4_kernel<TO, EPI, PROBE>), dim3(nwg), dim3(256), NT4_LDS, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo, PROBE ? 0 : g_nt4_probe);
  return CLIMB_OK;
}

int climb_nt4_launch(const bf16_t* A, long lda, const bf16_t* B, long ldb, void* C, long ldc, int c_dtype, int M, int N, int K, const float* bias, int epi,
                     const void* aux, long ldaux, bf16_t* aux_out, long ldauxo, hipStream_t st) {
  if (g_nt4 == 0 || ((g_nt4 == 1 || g_nt4 == 5) && (epi == EPI_GELU || epi == EPI_GELUD)) || (g_nt4 == 4 && epi == EPI_GELU)) return CLIMB_EUNSUPPORTED;
#ifndef NT4_SLP_BUILD
  if (g_nt4 == 2 || g_nt4 == 5) return climb_nt4slp_launch(A, lda, B, ldb, C, ldc, c_dtype, M, N, K, bias, epi, aux, ldaux, aux_out, ldauxo, st);
#endif
  if ((M % NT4_T) || (N % NT4_T) || ((M / NT4_T) % 8) || (K % GB_BK) || K < 10 * GB_BK) return CLIMB_EUNSUPPORTED;
  const long lim = 1L << 31;
  if (((long)M * lda + K) * 2 >= lim || ((long)N * ldb + K) * 2 >= lim || (long)M * ldc * 4 >= lim || (long)M * ldaux * 4 >= lim || (long)M * ldauxo * 2 >= lim)
    return CLIMB_EUNSUPPORTED;
  const int tiles = (M / NT4_T) * (N / NT4_T);
  int grid = g_nt4_grid > 0 ? (g_nt4_grid / 8) * 8 : 256;
  if (grid < 8) grid = 8;
  const int nwg = tiles < grid ? tiles : grid;      // tiles is a multiple of 8
#define L4(TO, E) return nt4_launch_one<TO, E>(nwg, st, A, lda, B, ldb, (TO*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo)
#ifndef NT4_SLP_BUILD
#define L4P(P) return nt4_launch_one<bf16_t, EPI_NONE, P>(nwg, st, A, lda, B, ldb, (bf16_t*)C, ldc, M, N, K, bias, aux, ldaux, aux_out, ldauxo)
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_NONE && g_nt4_probe >= 101 && g_nt4_probe <= 131) {      // measurement builds (option 18 = 100 + bits)
    if (g_nt4_probe == 101) L4P(1);
    if (g_nt4_probe == 102) L4P(2);
    if (g_nt4_probe == 103) L4P(3);
    if (g_nt4_probe == 104) L4P(4);
    if (g_nt4_probe == 111) L4P(11);
    if (g_nt4_probe == 119) L4P(19);
    if (g_nt4_probe == 127) L4P(27);
  }
#undef L4P
#endif
  if (c_dtype == CLIMB_DT_F32 && epi == EPI_RESID) L4(float, EPI_RESID);
  if (c_dtype == CLIMB_DT_F32 && epi == EPI_NONE) L4(float, EPI_NONE);
  if (K < 11 * GB_BK) return CLIMB_EUNSUPPORTED;      // the 16-bit kinds hold their results one window: one more k-tile for those of the ninth block
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_NONE) L4(bf16_t, EPI_NONE);
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_MUL) L4(bf16_t, EPI_MUL);
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_GELU) L4(bf16_t, EPI_GELU);
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_DGELU) L4(bf16_t, EPI_DGELU);
  if (c_dtype == CLIMB_DT_BF16 && epi == EPI_GELUD) L4(bf16_t, EPI_GELUD);
#undef L4
  return CLIMB_EUNSUPPORTED;
}

#ifndef NT4_SLP_BUILD
// r06: the same kernel over split operands (split.hip): A / B name the hi planes, the lo planes lie a_lo / b_lo ELEMENTS behind; fp32 C, epilogues NONE / RESID.
// K = the logical reduction depth (the kernel walks 3 K / 64 k-tiles).
int climb_nt4_split_launch(const bf16_t* A, long lda, long a_lo, const bf16_t* B, long ldb, long b_lo, float* C, long ldc, int M, int N, int K, const float* bias,
                           int epi, const void* aux, long ldaux, hipStream_t st, bf16_t* aux_out, long ldauxo, long o_lo) {
  if (g_nt4 == 0) return CLIMB_EUNSUPPORTED;
  if ((M % NT4_T) || (N % NT4_T) || ((M / NT4_T) % 8) || (K % GB_BK) || 3 * K < 10 * GB_BK) return CLIMB_EUNSUPPORTED;
  const long lim = 1L << 31;
  if (((long)M * lda + K + a_lo) * 2 >= lim || ((long)N * ldb + K + b_lo) * 2 >= lim || (long)M * ldc * 4 >= lim || (long)M * ldaux * 4 >= lim || a_lo < 0 || b_lo < 0 ||
      ((long)M * ldauxo + o_lo) * 2 >= lim || o_lo < 0)
    return CLIMB_EUNSUPPORTED;
  const int tiles = (M / NT4_T) * (N / NT4_T);
  int grid = g_nt4_grid > 0 ? (g_nt4_grid / 8) * 8 : 256;
  if (grid < 8) grid = 8;
  const int nwg = tiles < grid ? tiles : grid;
#define L4S(E)                                                                                                                                              \
  do {                                                                                                                                                      \
    static bool configured = false;                                                                                                                         \
    if (!configured) {                                                                                                                                      \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_nt4_kernel<float, E, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, NT4_LDS);             \
      if (e != hipSuccess) return (int)e;                                                                                                                   \
      configured = true;                                                                                                                                    \
    }                                                                                                                                                       \
    hipLaunchKernelGGL((gemm_bf16_nt4_kernel<float, E, 32>), dim3(nwg), dim3(256), NT4_LDS, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux,          \
                       aux_out, ldauxo, 0, (unsigned)(a_lo * 2), (unsigned)(b_lo * 2), (unsigned)(o_lo * 2));                                                      \
    return CLIMB_OK;                                                                                                                                        \
  } while (0)
  if (epi == EPI_RESID) L4S(EPI_RESID);
  if (epi == EPI_NONE) L4S(EPI_NONE);
  if (epi == EPI_GELU_SP && aux_out && C) L4S(EPI_GELU_SP);
  if (epi == EPI_DGELU_SP && aux_out && aux) L4S(EPI_DGELU_SP);
#undef L4S
  return CLIMB_EUNSUPPORTED;
}
#endif
