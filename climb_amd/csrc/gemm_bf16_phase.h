// The phase schedule shared by the persistent 256-row-tile GEMMs (gemm_bf16_ntp.hip: C = A B^T; gemm_bf16_tnp.hip: C += A^T B):
// which LDS-DMA staging unit is issued in which phase of a k-tile, and -- derived from that at compile time -- the vmcnt of every
// wait, including the two tail k-tiles and the prologue.  See gemm_bf16_ntp.hip for the structure and the hazard rules.
#pragma once
#include "gemm_bf16_nt.h"

#define NTP_BM 256
#define NTP_A_BYTES (NTP_BM * GB_BK * 2)      // 32 KB
#define NTP_B_UNIT (64 * GB_BK * 2)           // 8 KB

// ---------------------------------------------------------------------------------------------------- schedule (compile time)
// unit ids: 0 = A0, 1 = A1, 2 + p = B_p.  ntp_sched(NI, phase, k) = k-th unit issued in that phase of the body of k-tile T, encoded
// unit * 4 + d where the data belongs to k-tile T + d; -1 = none.
#ifndef NTP_VARIANT
#define NTP_VARIANT 0     // measured (r02): variants 0 and 1 are within noise of each other on every shape; 0 issues 2 DMAs per lane per phase
#endif
#define NTP_MAXI 3        // staging units issued per phase, at most
constexpr int ntp_sched(int NI, int p, int k) {
#define U_(unit, d) ((unit) * 4 + (d))
  if (NI == 4) {
    if (NTP_VARIANT == 0) {   // two DMA instructions per lane in every phase; B_0 has 3 phases from issue to its wait, the rest 4 - 5
      const int t[4][NTP_MAXI] = {{U_(2, 1), U_(3, 1), -1}, {U_(4, 1), U_(5, 1), -1}, {U_(0, 2), -1, -1}, {U_(1, 2), -1, -1}};
      return t[p][k];
    }
    // every unit at the EARLIEST phase WAR allows (A and B_0: dead after phase 0 -> phase 2; B_q -> phase q + 2): 4 - 5 phases of slack each
    const int t[4][NTP_MAXI] = {{U_(4, 1), -1, -1}, {U_(5, 1), -1, -1}, {U_(0, 2), U_(2, 2), -1}, {U_(1, 2), U_(3, 2), -1}};
    return t[p][k];
  }
  if (NTP_VARIANT == 0) {
    const int t[3][NTP_MAXI] = {{U_(1, 1), U_(2, 1), -1}, {U_(3, 1), U_(4, 1), -1}, {U_(0, 2), -1, -1}};
    return t[p][k];
  }
  const int t[3][NTP_MAXI] = {{U_(3, 1), -1, -1}, {U_(4, 1), -1, -1}, {U_(0, 2), U_(1, 2), U_(2, 2)}};      // 3 phases of slack for every unit
  return t[p][k];
#undef U_
}
constexpr int ntp_nglds(int unit, int gb = 1, int ga = 2) { return unit < 2 ? ga : gb; }      // DMA instructions per lane: A halves 2 (4 when four waves stage a 256-row image); a B unit 1 (8 waves) or 2 (4 waves)
// is (unit, tile t) read by the phase that FOLLOWS phase p of k-tile T?
constexpr bool ntp_needed_next(int NI, int T, int p, int unit, int t) {
  if (p + 1 < NI) return t == T && unit == 2 + p + 1;
  return t == T + 1 && unit <= 2;             // next k-tile's phase 0: A0, A1, B_0
}
// vmcnt for the wait in phase p of k-tile T (after that phase's own issues) when the k-loop has nk tiles; -1 = nothing to wait for.
// T = -1, p = NI - 1 is the prologue's wait.
constexpr int ntp_wait(int NI, int nk, int T, int p, int gb = 1, int ga = 2) {
  int cum = 0, need_end = -1;
  for (int Tb = -2; Tb <= T; ++Tb)
    for (int pb = 0; pb < NI; ++pb) {
      if (Tb == T && pb > p) break;
      for (int k = 0; k < NTP_MAXI; ++k) {
        const int e = ntp_sched(NI, pb, k);
        if (e < 0) continue;
        const int unit = e / 4, t = Tb + e % 4;
        if (t < 0 || t >= nk) continue;
        cum += ntp_nglds(unit, gb, ga);
        if (ntp_needed_next(NI, T, p, unit, t)) need_end = cum;
      }
    }
  return need_end < 0 ? -1 : cum - need_end;
}
// the hand-derived counts of the 256 x 256 schedule (main loop, second-to-last k-tile, last k-tile, prologue)
#if NTP_VARIANT == 0
static_assert(ntp_wait(4, 6, 2, 0) == 8 && ntp_wait(4, 6, 2, 1) == 9 && ntp_wait(4, 6, 2, 2) == 10 && ntp_wait(4, 6, 2, 3) == 7, "main");
static_assert(ntp_wait(4, 6, 4, 0) == 8 && ntp_wait(4, 6, 4, 1) == 9 && ntp_wait(4, 6, 4, 2) == 8 && ntp_wait(4, 6, 4, 3) == 3, "nk-2");
static_assert(ntp_wait(4, 6, 5, 0) == 2 && ntp_wait(4, 6, 5, 1) == 1 && ntp_wait(4, 6, 5, 2) == 0 && ntp_wait(4, 6, 5, 3) == -1, "nk-1");
static_assert(ntp_wait(4, 6, -1, 3) == 7 && ntp_wait(4, 2, -1, 3) == 7 && ntp_wait(4, 2, 0, 3) == 3 && ntp_wait(4, 2, 1, 0) == 2, "prologue / nk = 2");
static_assert(ntp_wait(3, 6, 2, 0) == 6 && ntp_wait(3, 6, 2, 1) == 7 && ntp_wait(3, 6, 2, 2) == 4, "192 main");
#endif

