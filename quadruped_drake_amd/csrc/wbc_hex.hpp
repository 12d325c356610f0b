// wbc_hex.hpp -- one control tick computed by SIXTEEN lanes (one DPP row) per robot (product math, v4).
//
// Why (profiles/r01/cuts.md, profiles/r02/hex_cuts.md): round 1's 4-lanes-per-robot mapping (retired) gave only N/16 wavefronts
// (256 at N = 4096: one SIMD of four busy per CU) and its per-lane state (three z-space columns, three
// rows of J) does not fit 512 registers -- 1.4 KB/lane of scratch that overflows L2 once two waves
// share a CU.  Here lane h = 4*leg + sub of a 16-lane DPP row:
//   * leg dynamics are replicated on the four sub-lanes of a leg, except the three Newton-Euler passes
//     of the MPTC law (bias, +xi, -xi), which run concurrently on sub-lanes 0/1/2 (same code, other data);
//   * every z-space object is distributed one COLUMN per lane: lane (leg, j<3) owns reduced variable
//     3*leg+j (its column of B = G_b^-1 [W|-X], of the QR factor, its row of J = R^-1); lane (leg, 3)
//     owns the right-hand side / ab0 column, so "matrix | rhs" needs no separate code path;
//   * rows of the least-squares blocks are owned the same way (lane (leg,i) = row 3*leg+i), so every
//     block is an all-pairs product row-vector(6) . column(6) fed by single-instruction
//     v_mov_b64_dpp row_newbcast broadcasts from compile-time lanes;
//   * sums over the four legs are two DPP stages (row_ror:8, row_ror:4), bit-identical on all lanes.
// A wavefront holds 4 robots: N/4 wavefronts, ~1/2 the instructions per wavefront of the quad kernel
// and no scratch.
//
// `Q` is the communication policy: HexDev (wbc_kernels.hip) on the GPU, a 16-fibre lock-step
// emulation in tools/host_tick.cpp for CPU-side validation.
#pragma once
#include <type_traits>
#include <utility>
#include "wbc_tick.hpp"

// host-only diagnostics (tools/host_tick.cpp): how many fast / generic active-set trips a robot ran
#if !defined(__HIPCC__) && defined(WBC_HOST_GI_STATS)
extern int g_gi_fast_trips, g_gi_generic_trips, g_gi_drops, g_gi_force_bail;
extern double* g_gi_dump;   // analysis: [16][NV + 3] per robot = the active set's inputs (own row of J, z, mu_n, inv_s) of every lane
#define WBC_GI_FORCE_BAIL(qc) (g_gi_force_bail == (qc))
#define WBC_GI_STAT(x) do { x; } while (0)
#else
#define WBC_GI_STAT(x) do { } while (0)
#endif
// host instantiations (tests, tools/host_tick.cpp) check the kernel's structural invariants; device code carries none of it
#if !defined(__HIPCC__)
#include <cassert>
#define WBC_HOST_ASSERT(x) assert(x)
#else
#define WBC_HOST_ASSERT(x) do { } while (0)
#endif
#ifndef WBC_GI_FORCE_BAIL
#ifdef WBC_DEV_FORCE_BAIL   // diagnostic device builds: every wavefront leaves the fast path at trip WBC_DEV_FORCE_BAIL
#define WBC_GI_FORCE_BAIL(qc) ((qc) == WBC_DEV_FORCE_BAIL)
#else
#define WBC_GI_FORCE_BAIL(qc) false
#endif
#endif

namespace wbc {

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(integral_constant<int, N-1>{}) -- the index is a
// constant expression inside the body (the fused broadcast-FMA takes its source lane as an immediate)
template <class F, int... I> WBC_HD void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> WBC_HD void static_for(F f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// lane (within the 16-lane row) that owns z-space column / row k = 3*leg + coordinate; the CLF law's 13th
// reduced variable (the slack delta, k = 12) lives on the spare sub-lane 3 of leg 1
enum { HEX_DELTA_LANE = 7 };
WBC_HD constexpr int hex_lane(int k) { return k < 12 ? 4 * (k / 3) + (k % 3) : HEX_DELTA_LANE; }
// Pivot order of the QR factor: slot (whitened coordinate) k belongs to reduced variable hex_piv(k).  The factor is inverted
// explicitly (J = R^-1, one row per lane, forward substitution), and that is only accurate on every scale when R is GRADED --
// the eps-sized pivots of the internal-force directions last.  In the natural order (x, y, z of leg 0, leg 1, ...) a 4-contact
// stand has its six eps-sized pivots interleaved with level-1 sized ones (two point feet span only five wrench directions: y of
// leg 1 already depends on leg 0) and the body-task part of J is contaminated at 1e-12 x cond: 1e-6 .. 1e-5 on the torques of
// saturated stands (profiles/r04/accuracy.md).  Order: the three vertical forces z0 z1 z2 (force z, moments x and y), x0 y0
// (forces x, y), y2 (the yaw moment: left-front and left-hind differ in x) -- six independent wrench columns on any stance --
// then the rest.  Trots keep their structure (one eps-sized pivot, last or followed by swing variables as before).
WBC_HD constexpr int hex_piv(int k) {
  constexpr int order[13] = {2, 5, 8, 0, 1, 7, 3, 4, 6, 9, 10, 11, 12};
  return order[k];
}
WBC_HD constexpr int hex_piv_lane(int k) { return hex_lane(hex_piv(k)); }   // lane that owns the k-th pivot column

// Distributed Householder append: fold P dense rows into the upper-triangular factor.
// Lane hex_lane(c) holds column c: Rcol[12], Acol[P]; the lanes with sub == 3 hold the right-hand side
// as one more column.  Already-pivoted columns are left with O(ulp) residue below the diagonal of R
// (never read) instead of being zeroed: no per-step predication.
// A pivot step costs 2 P + ~45 instructions: the pivot column is never copied -- both the dot product and the
// update read it straight from the pivot lane with the fused broadcast-FMA (Q::fma_bc = v_fmac_f64_dpp).
template <class Q, int P, int NV = NZ>
WBC_HD void hex_qr_append(Q& qo, double* Rcol, double* Acol) {
  qo.template dpp_fence<P>(Acol);
  static_for<NV>([&](auto K) {
    constexpr int k = K;
    constexpr int piv = hex_piv_lane(k);
    // three accumulators, the terms in blocks of <= 15 fused ops (one asm statement each: Q::dot_bc)
    double ta = 0.0, tb = 0.0, tc = 0.0;
    {
      constexpr int C0 = (P >= 15) ? 15 : (P >= 9 ? 9 : (P >= 6 ? 6 : 3));
      static_assert(P % 3 == 0, "row count is a multiple of three");
      qo.template dot_bc<piv, C0>(ta, tb, tc, Acol);
      constexpr int R1 = P - C0;
      if constexpr (R1 > 0) {
        constexpr int C1 = (R1 >= 15) ? 15 : (R1 >= 9 ? 9 : (R1 >= 6 ? 6 : 3));
        qo.template dot_bc<piv, C1>(ta, tb, tc, Acol + C0);
        constexpr int R2 = R1 - C1;
        if constexpr (R2 > 0) {
          constexpr int C2 = (R2 >= 15) ? 15 : (R2 >= 9 ? 9 : (R2 >= 6 ? 6 : 3));
          qo.template dot_bc<piv, C2>(ta, tb, tc, Acol + C0 + C1);
          static_assert(R2 - C2 == 0, "at most three blocks");
        }
      }
    }
    const double t = (ta + tb) + tc;
    const double s2 = qo.bcast16(t, piv);
    const double rkk = qo.bcast16(Rcol[k], piv);
    const double nrm = fast_sqrt(rkk * rkk + s2);   // hardware-seeded root and reciprocal (wbc_tick.hpp)
    const double alpha = (rkk > 0.0) ? -nrm : nrm;
    const double v0 = rkk - alpha;
    const double beta = (s2 > 0.0) ? fast_rcp(nrm * (nrm + fabs(rkk))) : 0.0;  // = 2 / (s2 + v0^2); empty column: no-op
    const double ns = -((v0 * Rcol[k] + t) * beta);
    Rcol[k] += ns * v0;
    static_for<P>([&](auto I) { Acol[I] = qo.template fma_bc<piv>(Acol[I], Acol[I], ns); });
  });
}

// Explicit fused multiply-add: the active set's fast and generic paths must round identically (a robot's result may
// not depend on which path its wavefront took), so nothing there is left to the compiler's contraction choices.
WBC_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

WBC_HD double mk2(double e) { return 1.0 - e; }   // complement of a one-hot mask entry

// 16-lane argmin as ONE fmin butterfly: the 5-bit candidate index rides in the low mantissa bits of the value
// (a 2^-47 relative perturbation of the returned minimum; "no candidate" is the finite HEX_NONE, never inf, so
// that the packed key is never a NaN).  4 x (2 DPP moves + v_min_f64) instead of 4 x ~12 compare/select steps.
constexpr double HEX_NONE = 1e300;
WBC_HD double hex_pack_key(double v, int idx) {
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  b = (b & ~0x1Full) | (unsigned long long)(idx & 31);
  double k;
  __builtin_memcpy(&k, &b, 8);
  return k;
}
WBC_HD int hex_key_index(double k) {
  unsigned long long b;
  __builtin_memcpy(&b, &k, 8);
  return (int)(b & 31ull);
}
// The PICK keys (most violated row / greatest dual gain) are packed coarsely: the low 24 mantissa bits are replaced, i.e. candidates within 6e-8
// of each other count as tied and the LOWER row index wins (the keys that matter are negative: a larger low field is a larger magnitude,
// hence 31 - idx).  Why: on the way into an apex of a friction pyramid -- one x-side and one y-side row of a foot active -- the foot's two
// remaining rows have EXACTLY the same value (2 mu_n f_z) and the same free part, so rounding used to decide the pick, differently on the host and on the
// device and from build to build, and one of the two choices ends in a blocked step: a drop and a re-add, on the device two generic trips for
// the whole wavefront (the slowest wavefront of the headline launch was such a robot; profiles/r05/apex_rule.md).  The pick only needs
// SOME violated row, the exact value of the picked row is fetched from its lane; the blocking-ratio keys (hex_pack_key) stay at 2^-47.
constexpr int HEX_PICK_BITS = 24;
template <int BITS = HEX_PICK_BITS> WBC_HD double hex_pack_pick(double v, int idx) {
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  b = (b & ~((1ull << BITS) - 1ull)) | (unsigned long long)((31 - idx) & 31);
  double k;
  __builtin_memcpy(&k, &b, 8);
  return k;
}
WBC_HD int hex_pick_index(double k) {
  unsigned long long b;
  __builtin_memcpy(&b, &k, 8);
  return 31 - (int)(b & 31ull);
}

// Goldfarb-Idnani on the friction rows, register-resident "constraint-space" form.
//
// Friction row h = 4*leg + r lives on lane h.  Besides its row Jr of J (J J' = H^-1) every lane carries
// Dh = J' n_h (its constraint's image), the running value s_h = n_h' z and, while its row is active,
// the multiplier u_h, its position in the active list and the reciprocal pivot of its column:
//   * d = J' n_p is ONE dynamic row broadcast of Dh from lane p (ds_bpermute), not 12 reductions;
//   * n_p' J2 J2' n_p = |d[q:]|^2, so the primal step length needs no reduction, and |J' n_h| is
//     invariant under the orthogonal updates of J (computed once);
//   * s_h advances by t * Dh[q:] . d[q:] (lane-local): picking the most violated row is one argmin;
//   * J' N_A = [R; 0]: column c of the triangular factor IS Dh[0..c] of the row at position c, so R is never
//     stored; the lane of the active row at position c (pos_h == c) carries row c of W = R^-1 instead, so the dual
//     direction r = W d[0:q] is one lane-local dot product (no back-substitution chain).  Appending a
//     row appends the column [-r/alpha; 1/alpha] to W; dropping one reflects its W row onto the last used slot (below);
//   * the blocking multiplier is a lane-parallel ratio + argmin.
// Nothing lives in LDS and the loop has no divergent inner branches (profiles/r02/hex_cuts.md).
// The optional dense row (index 16: the PC law's Vdot <= 0, the CLF law's CLF row) has its image / value /
// multiplier replicated on all lanes.  NV = 13 for the CLF law (slack delta on lane HEX_DELTA_LANE).
// Optional torque box (TB): lane (leg, j < 3) also owns the two-sided row |tau_(leg,j)| <= tau_max of its own
// joint in a second constraint slot (id 32 + lane): unit normal Tn of the torque-map row, normalised torque
// yt = Tn.z + t0n tracked like s_h, bound bt = tau_max / |T_row| (< 0: slot not eligible).  Only one side of a
// pair can be violated or active at a time; the side is a sign (sig) applied to the slot's image.
// WARM (the persistent rollout kernels; wbc_set_warm_start): `*seed` says, per lane, whether this lane's friction row was active when the robot's PREVIOUS
// tick ended.  Goldfarb-Idnani may add any violated row; a seeded row that is violated is preferred to every other candidate (its pick key is pushed to
// -HEX_SEED: the exact value of a picked row is fetched from its lane anyway), so a closed loop whose active set moves little from tick to tick rebuilds it
// in as many full-step adds as it has rows -- the fast path's compile-time trips -- instead of finding it again through the cold start's detours and drops
// (profiles/r06/warm_start.md: 10 - 13 trips per tick -> the size of the set on saturated closed loops).  Everything else is the same algorithm: same
// tolerance, same steps, same exit; the QP is strictly convex, so the solution does not depend on the order of the adds (outputs agree with the cold start
// to rounding, not bit for bit).  On return `*seed` = this lane's row is active NOW.  With WARM = false (the tick kernels) none of this is compiled.
constexpr double HEX_SEED = 1e295;
template <class Q, bool PC, int NV = NZ, bool TB = false, bool GAIN = false, bool HYB = false, bool WARM = false>
WBC_HD int hex_gi(Q& qo, int h, bool ct, double* Jr, double& z, double mu_n, double inv_s, int* iters_out,
                  double vrow_own = 0.0, double vc = 0.0, double pc_inv = 0.0, const double* Tn = nullptr,
                  double t0n = 0.0, double bt = -1.0, bool deep = true, bool* seed = nullptr) {
  const int sb = h & 3;
  const double seed_bias = (WARM && *seed) ? HEX_SEED : 0.0;   // subtracted from the pick key of this lane's row while it is a violated candidate
  const bool pc = PC && pc_inv > 0.0;
  WBC_GI_STAT(if (g_gi_dump) { double* o = g_gi_dump + h * 16; for (int k = 0; k < NV && k < 13; k++) o[k] = Jr[k]; o[13] = z; o[14] = mu_n; o[15] = ct ? inv_s : 0.0; });
  const double sg = (sb & 1) ? inv_s : -inv_s;   // own row: n_h = sg * e_(leg, sb>>1) + mu_n * e_(leg, 2)
  double Dh[NV], sh_, dnh = 0.0;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    Dh[k] = sg * qo.leg_pairs(Jr[k]) + mu_n * qo.leg_bcast(Jr[k], 2);
    dnh += Dh[k] * Dh[k];
  }
  sh_ = sg * qo.leg_pairs(z) + mu_n * qo.leg_bcast(z, 2);
  const double npl = PC ? -vrow_own * pc_inv : 0.0;   // own entry of the dense row's normal
  // Lane (0, 3) owns no row of J: its Jr[] carries y = Q'b through the reflections (the evaluation after a drop, below), and z there
  // accumulates a meaningless y . d.  That is harmless only while every 16-lane reduction over Jr or z is masked off that lane: the
  // dense row's normal, the torque rows' normals and the tolerance's |z| are zero there by construction -- checked on the host.
  WBC_HOST_ASSERT(h != 3 || ((vrow_own == 0.0 || vrow_own != vrow_own) && (npl == 0.0 || npl != npl) && (z == 0.0 || z != z)));   // (NaN: a malformed instance on its way to the output stage's sentinel)
  if (TB) { for (int c = 0; c < NZ; c++) WBC_HOST_ASSERT(hex_lane(c) != 3); }   // the torque rows read Jr / z of column lanes only
  double Dpc[NV], spc = 0.0, dnpc = 0.0, u_pc = 0.0;
  bool act_pc = false;
  // The dense row (PC: Vdot <= 0, CLF: its decrease condition) is inactive on most robots and most ticks.  LAZY: instead of reflecting its image D_pc through
  // every trip (and building it up front: thirteen 16-lane sums), its VALUE is taken fresh from the current z at every pick (one 16-lane sum) and its image is
  // built from the current rows of J only in a trip that adds it (or, for the evaluation after a drop, while it is active): profiles/r05/lazy_dense.md.
  constexpr bool LAZY = PC && (NV != NZ);   // CLF (13 slots): -4 ... -5 %; PC measured no gain (its trips are mostly fast-path ones) and keeps the reflected image
  auto dense_value = [&]() -> double { return -(qo.sum16(vrow_own * z) + vc) * pc_inv; };
  auto dense_image = [&](double* D) -> double {
    double n2 = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) { D[k] = qo.sum16(Jr[k] * npl); n2 += D[k] * D[k]; }
    return n2;
  };
  if (PC) {
    if constexpr (LAZY) {
      spc = dense_value();
    } else {
#pragma unroll
      for (int k = 0; k < NV; k++) { Dpc[k] = qo.sum16(Jr[k] * npl); dnpc += Dpc[k] * Dpc[k]; }
      spc = -(qo.sum16(vrow_own * z) + vc) * pc_inv;
    }
  }
  double u_h = 0.0, Wr[NV], Wpc[NV];
  bool act_h = false;   // own friction row is in the active set
#pragma unroll
  for (int k = 0; k < NV; k++) { Wr[k] = 0.0; Wpc[k] = 0.0; }
  double Dt[NV], Wt[NV], yt = 0.0, dnt = 0.0, u_t = 0.0, sig_t = 1.0;
  bool act_t = false;
  const bool elig_t = TB && bt >= 0.0;
  if (TB) {
#pragma unroll
    for (int k = 0; k < NV; k++) { Dt[k] = 0.0; Wt[k] = 0.0; }
    yt = t0n;
#pragma unroll
    for (int c = 0; c < NZ; c++) {
      const int src = hex_lane(c);
#pragma unroll
      for (int k = 0; k < NV; k++) Dt[k] += qo.bcast16(Jr[k], src) * Tn[c];
      yt += qo.bcast16(z, src) * Tn[c];
    }
#pragma unroll
    for (int k = 0; k < NV; k++) dnt += Dt[k] * Dt[k];
  }
  int q = 0, iters = 0, status = ST_OK;
  // The APEX rule (profiles/r05/apex_rule.md).  The four friction rows of a foot satisfy n_0 + n_1 = n_2 + n_3 (= 2 mu_n e_z): while three of them
  // are active the fourth is linearly dependent and its value is identically zero -- in floating point a rounding remainder of either sign.  A
  // negative remainder beyond the tolerance used to be picked, found dependent, and cost a drop and an add (on the device: two GENERIC trips and the
  // fast path lost for the whole wavefront) that swap one description of the apex for another; on BASELINE config 2, whose solutions sit on
  // edges and apexes, that was one trip in ten, and three of the six slowest wavefronts of the headline launch.  A row whose three leg-mates are
  // active is therefore not a candidate.  nleg = active friction rows of the own leg (replicated on its four lanes).
  // (Friction-only instantiations -- ID, MPTC, with or without the torque box: the dense-row laws PC / CLF carry their Dpc / Wpc arrays through the same
  // trips and measured 2 % slower with the bookkeeping than they gain -- their picks still get the deterministic tie-break of hex_pack_pick.)
  constexpr bool APEX = !PC;
  // dense-row laws: the keys of rounds 1-4 unchanged (anything else costs the CLF torque-box rollout kernel, at 486 registers, its last ones: 8 B/lane of scratch)
  auto pick_pack = [](double v, int idx) -> double { if constexpr (PC) return hex_pack_key(v, idx); else return hex_pack_pick<HEX_PICK_BITS>(v, idx); };
  auto pick_pack_fine = [](double v, int idx) -> double { if constexpr (PC) return hex_pack_key(v, idx); else return hex_pack_pick<5>(v, idx); };
  auto pick_index = [](double k) -> int { if constexpr (PC) return hex_key_index(k); else return hex_pick_index(k); };
  int nleg = 0;
  auto same_leg = [&](int row) -> int { return (((row ^ h) & 12) == 0) ? 1 : 0; };   // friction row `row` (0..15, robot-uniform) belongs to my leg
  const int maxit = 200;
  bool done = false, need_pick = true, picked = false;
  int p = -1;
  double sp = 0.0, up = 0.0, dnp = 1.0, sgp = 1.0;
  const double INF = __builtin_huge_val();
  // feasibility tolerance from the scale of the unconstrained minimiser (the iterates stay on that scale)
  const double tol = 1e-13 * (1.0 + qo.max16(fabs(z)));
  // the most-violated rule's key of this lane's row: its value -- pushed down by the seed bias while the row is violated (a seeded row that is satisfied is no candidate)
  auto warm_key = [&](double v) -> double { if constexpr (WARM) return (v < -tol) ? v - seed_bias : v; else return v; };
  // ---- Fast path (plain friction rows only): the first trips of a tick are almost always "add the picked row with a
  // full step" on every robot of the wavefront, so the list length q is wave-uniform and equals the trip number.  With q
  // a compile-time constant the position masks and all work on the fixed positions k < q disappear.  The moment any
  // robot of the wavefront needs something else (a partial step = a drop, a dependent row) the trip is abandoned BEFORE
  // it has changed any state and the generic loop below takes over.  Both paths evaluate the same expressions in the
  // same order, so a robot's result does not depend on which path its wavefront took (bit-identical; tests:
  // batch-position invariance).
  // (CLF, 13 slots, through the fast path too: built, bit-identical, measured -1 ... +3 %, profiles/r05/hybrid_pick.md -- not taken)
  constexpr int QF = (!TB && NV == NZ) ? 8 : 0;   // fast-path bodies; ID stands add up to 8-12 rows before the first drop (12 bodies measured slower); PC: see pcv below
  // Which inactive row to add -- Goldfarb-Idnani may take any violated one.  Greatest dual gain s^2 / |free part|^2 (GAIN) needs fewer trips where few rows
  // end up active (one or two feet down: -12 ... -30 % trips), the most violated row needs fewer on saturated stands (four feet: -4 ... -7 % trips and no
  // gain arithmetic: -15 % launch time); three feet: a wash (host emulation by contact count, profiles/r05/hybrid_pick.md).  HYB: the rule is chosen PER
  // ROBOT by its contact count (`deep` = three or four feet down) and a wavefront of deep robots skips the gain arithmetic.  The dense-row laws (PC, CLF)
  // are built that way; the friction-only laws keep their compile-time rule (MPTC: gain, ID: most violated) -- the two votes and the per-pick branch
  // measured +1.5 % on the headline launch and +2 % on the ID stand, the two BASELINE lines, for -2 ... -3 % on their off-design batches.
  const bool use_gain = HYB ? (GAIN && !deep) : GAIN;                          // per robot
  const bool wave_gain_any = HYB ? (GAIN && qo.wave_any(!deep)) : GAIN;        // wave-uniform: somebody needs the gains
  const bool wave_gain_all = HYB ? (GAIN && qo.wave_all(!deep)) : GAIN;        // wave-uniform: every candidate of the wavefront is a violated row
  bool generic = true;   // wave-uniform: the generic loop still has work to do
  // What an abandoned fast trip has already worked out -- pick, the picked row's image, the step's dots, the blocking ratio, the norm --
  // is the front half of a generic trip, evaluated by the same expressions: the first generic trip takes it over instead of
  // repeating the argmin, the crossbar round trip and the dots (~1300 cycles on every wavefront that leaves the fast path; on a trot
  // batch those few wavefronts are the launch's tail: MPTC trot N = 4096 -2 %).  Friction-only laws (ID, MPTC).
  bool handed = false;   // wave-uniform
  int ho_p = -1, ho_hd = -1;
  double ho_d[NV], ho_sp = 0.0, ho_dn = 1.0, ho_d2n = 0.0, ho_zd = 0.0, ho_sd = 0.0, ho_r = 0.0, ho_t1 = 0.0;
  if constexpr (QF > 0) {
    bool stop = false;   // wave-uniform
    static_for<QF>([&](auto QQ) {
      constexpr int qc = QQ;
      if (stop) return;
      // pick (a finished robot offers no candidate)
      int pf;
      {
        double key = HEX_NONE;
        if (!done && ct && !act_h && !(APEX && nleg == 3)) {
          if (GAIN && wave_gain_any) {
            double dd2 = 0.0;
            static_for<NV - qc>([&](auto KK) { constexpr int k = qc + KK; dd2 = fmad(Dh[k], Dh[k], dd2); });
            if constexpr (HYB) {
              const double gk = (dd2 > 1e-22 * dnh) ? -(sh_ * sh_) * fast_rcp(dd2) : -1e290;
              if (!use_gain) key = pick_pack(warm_key(sh_), h);
              else if (sh_ < -tol) key = pick_pack(gk - seed_bias, h);
            } else {
              if (sh_ < -tol) key = pick_pack(((dd2 > 1e-22 * dnh) ? -(sh_ * sh_) * fast_rcp(dd2) : -1e290) - seed_bias, h);
            }
          } else {
            key = pick_pack(warm_key(sh_), h);
          }
        }
        key = qo.min16(key);
        pf = (key < 1e299) ? pick_index(key) : -1;
      }
      // PC law: the dense row (Vdot <= 0) is never added here -- a robot whose dense row is violated sends the wavefront to the
      // generic loop (rare: ~5 % of the robots); while it is inactive the fast trips only carry its image and value along
      const bool pcv = PC && pc && spc < -tol;
      // every wavefront's last trip finds nothing left to repair anywhere: leave before the crossbar round trip
      if (qo.wave_all(done || (pf < 0 && !pcv))) { done = true; stop = true; generic = false; return; }
      // ONE round trip: the picked row's image, value and norm from its lane (own lane when there is no candidate)
      const int pl = (pf >= 0) ? pf : h;
      double d[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) d[k] = qo.bcast16d(Dh[k], pl);
      const double spx = qo.bcast16d(sh_, pl), dnx = qo.bcast16d(dnh, pl);
      // the dense row wins the pick when it is the most violated one (generic loop: spc < sp)
      const bool pcpick = pcv && !(pf >= 0 && !(spc < spx));
      if (!(pf >= 0 && spx < -tol) && !pcpick) done = true;        // nothing (left) to repair on this robot
      // (with the gain pick only violated rows are candidates: a robot with a candidate stays live, the test above was the exit)
      if (!wave_gain_all && qo.wave_all(done)) { stop = true; generic = false; return; }
      double d2n = 0.0, zd = 0.0, sd = 0.0, r_h = 0.0;
      static_for<NV - qc>([&](auto KK) { constexpr int k = qc + KK; d2n = fmad(d[k], d[k], d2n); });
      static_for<NV - qc>([&](auto KK) { constexpr int k = qc + KK; zd = fmad(Jr[k], d[k], zd); sd = fmad(Dh[k], d[k], sd); });
      static_for<qc>([&](auto KK) { r_h = fmad(Wr[KK], d[KK], r_h); });
      r_h = act_h ? r_h : 0.0;
      double t1 = INF;
      bool have_t1 = false;
      if (qc > 0) {
        double key = (act_h && r_h > 0.0) ? hex_pack_key(u_h * fast_rcp(r_h), h) : HEX_NONE;
        key = qo.min16(key);
        have_t1 = key < 1e299;
        t1 = have_t1 ? key : INF;
      }
      const bool dependent = !(d2n > 1e-22 * dnx);
      double nrm, rsn;                                   // |d[qc:]| and its reciprocal from one seed: step length, norm and 1/alpha
      fast_sqrt_rsq(d2n, nrm, rsn);
      const double t2 = -spx * (rsn * rsn);
      const bool full = !dependent && (!have_t1 || !(t1 < t2));
      if (qo.wave_any(!done && (!full || pcpick)) || WBC_GI_FORCE_BAIL(qc)) {   // not a friction add-with-full-step everywhere: generic loop, state untouched
        stop = true;
        if constexpr (!PC) {   // (measured on the PC law, whose dense row sends ~20 % of the wavefronts here: +2 % -- not taken over there)
          handed = true;
          ho_p = pf; ho_sp = spx; ho_dn = dnx; ho_d2n = d2n; ho_zd = zd; ho_sd = sd; ho_r = r_h;
          ho_t1 = t1; ho_hd = have_t1 ? hex_key_index(t1) : -1;
#pragma unroll
          for (int k = 0; k < NV; k++) ho_d[k] = d[k];
        }
        return;
      }
      if (!done) {
        iters++;
        WBC_GI_STAT(if (h == 0) g_gi_fast_trips++);
        u_h = fmad(-t2, r_h, u_h);
        z = fmad(t2, zd, z);
        sh_ = fmad(t2, sd, sh_);
        // one Householder reflection on d[qc:] (H d2 = alpha e_qc), applied to the own row of J and to the image
        const double dq = d[qc];
        const double alpha = (dq > 0.0) ? -nrm : nrm;
        const double ia = (dq > 0.0) ? -rsn : rsn;
        const double beta = fast_rcp(nrm * (nrm + fabs(dq)));   // 2 / (v'v)
        const double vq = dq - alpha;
        const double w = fmad(-alpha, Jr[qc], zd) * beta, wd = fmad(-alpha, Dh[qc], sd) * beta;   // x . v = x . d[qc:] - alpha x_qc
        Jr[qc] = fmad(-w, vq, Jr[qc]);
        Dh[qc] = fmad(-wd, vq, Dh[qc]);
        static_for<NV - qc - 1>([&](auto KK) { constexpr int k = qc + 1 + KK; Jr[k] = fmad(-w, d[k], Jr[k]); Dh[k] = fmad(-wd, d[k], Dh[k]); });
        if (PC) {   // (PC only: the CLF law, whose dense row is built lazily, has no fast path -- QF = 0 for 13 slots)
          double sdpc = 0.0;
          static_for<NV - qc>([&](auto KK) { constexpr int k = qc + KK; sdpc = fmad(Dpc[k], d[k], sdpc); });
          spc = fmad(t2, sdpc, spc);
          const double wp = fmad(-alpha, Dpc[qc], sdpc) * beta;
          Dpc[qc] = fmad(-wp, vq, Dpc[qc]);
          static_for<NV - qc - 1>([&](auto KK) { constexpr int k = qc + 1 + KK; Dpc[k] = fmad(-wp, d[k], Dpc[k]); });
        }
        const bool mine = (h == pf);
        Wr[qc] = mine ? ia : -r_h * ia;
        u_h = mine ? t2 : u_h;
        act_h = act_h || mine;
        if (APEX) nleg += same_leg(pf);
        q = qc + 1;
      }
    });
  }
  // ---- Generic loop.  Position mask kept as doubles (replicated): mk = 1 on the free slots k >= q; the one-hot of the next free
  // slot is eq[k] = mk[k] - mk[k-1].  Products with them replace select chains (d masked, d[q], the reflector).
  // Every trip is ONE Householder reflection per robot, whatever the robot does:
  //   append (full step):   x = d[q:]                      -> alpha e_q,     then q + 1
  //   drop (partial step):  x = W row of the blocking row  -> alpha e_(q-1), then q - 1.
  // The drop: that row of W = R^-1 is, within the used slots, the normal of the span of the REMAINING active images
  // (W R = I), so the reflection leaves every remaining image with a zero in slot q-1: the slot is free again, the list
  // stays a prefix, and there is no position bookkeeping and no per-position Givens sweep (round 2: up to 11 predicated
  // rotations with two 16-lane sums each; the divergent add / drop branches cost a lock-step trip both bodies).
  WBC_GI_TIMERS;
  if (generic) {   // everything below -- the loop and the evaluation after a drop -- lives on this path only: nothing of it is live where the fast path's exit joins
  double mk[NV];
#pragma unroll
  for (int k = 0; k < NV; k++) mk[k] = (k >= q) ? 1.0 : 0.0;
  bool dropped = false, wave_dropped = false;   // this (deep) robot has dropped a row / the wavefront has a deep robot and a drop (wave-uniform)
  const bool wave_deep = qo.wave_any(deep);
  for (int trip = 0; trip < maxit; trip++) {
    WBC_GI_T0();
    double d[NV], dm[NV], d2n = 0.0;
    double zd = 0.0, sd = 0.0, sdpc = 0.0, sdt = 0.0;
    double r_h = 0.0, r_pc = 0.0, r_t = 0.0;
    double t1 = INF;
    int hd = -1;
    const bool takeover = handed && trip == 0;   // wave-uniform
    bool all_done = false;
    if (!takeover) {
      if (!done && need_pick) {
        if constexpr (LAZY) { if (pc && !act_pc) spc = dense_value(); }
        // most violated inactive row: argmin of the tracked values (friction slot: index h, torque slot: 16 + h)
        {
          double key = HEX_NONE;
          if (GAIN && wave_gain_any) {
            // greatest dual gain s^2 / |D_h[q:]|^2 instead of the most violated row: fewer iterations for the worst
            // robots of a trot batch (max 7 -> 6, rows needing >= 5: 88 -> 32 of 4096), more for the 4-contact ID stand
            double dd2 = 0.0;
#pragma unroll
            for (int k = 0; k < NV; k++) dd2 = fmad(mk[k] * Dh[k], Dh[k], dd2);
            // a violated row whose image has no free part (linearly dependent on the active ones) must still be
            // picked -- the dependent-step logic below resolves or reports it -- so it gets the largest finite gain
            if constexpr (HYB) {
              const double gk = (dd2 > 1e-22 * dnh) ? -(sh_ * sh_) * fast_rcp(dd2) : -1e290;
              if (ct && !act_h && !(APEX && nleg == 3)) {
                if (!use_gain) key = pick_pack(warm_key(sh_), h);
                else if (sh_ < -tol) key = pick_pack(gk - seed_bias, h);
              }
            } else {
              if (ct && !act_h && !(APEX && nleg == 3) && sh_ < -tol)
                key = pick_pack(((dd2 > 1e-22 * dnh) ? -(sh_ * sh_) * fast_rcp(dd2) : -1e290) - seed_bias, h);
            }
          } else {
            if (ct && !act_h && !(APEX && nleg == 3)) key = pick_pack(warm_key(sh_), h);
          }
          if (TB) {
            const double st_ = bt - fabs(yt);
            if (elig_t && !act_t && st_ < key) key = pick_pack_fine(st_, 16 + h);   // a torque row's key IS its value downstream (no fetch): 2^-47, as before
          }
          key = qo.min16(key);
          const int ix = pick_index(key);
          sp = key;
          p = (key < 1e299) ? ((ix < 16) ? ix : 16 + ix) : -1;   // torque slot ids are 32 + lane
          if (p < 0) sp = INF;
        }
        picked = true;
      }
      // ONE round trip through the lane crossbar per trip: the image of the (newly or previously) picked row and, for a new
      // pick, its exact value (the key carries index bits, or is the gain), its norm and -- torque rows -- its violated side
      WBC_GI_T(0);   // pick
      const int pl = (p >= 0 && p != 16) ? (p & 15) : h;
      const bool trow = TB && p >= 32;
#pragma unroll
      for (int k = 0; k < NV; k++) d[k] = qo.bcast16d(trow ? Dt[k] : Dh[k], pl);
      const double sp_x = qo.bcast16d(sh_, pl), dn_x = qo.bcast16d(trow ? dnt : dnh, pl);
      double sg_x = 1.0;
      if (TB) sg_x = qo.bcast16d((yt > 0.0) ? -1.0 : 1.0, pl);
      if (picked) {
        picked = false;
        if (p >= 0 && p < 16) sp = sp_x;   // (the friction rows' pick keys are coarse, hex_pack_pick: the exact value comes from the row's lane)
        if (pc && !act_pc && spc < sp) { sp = spc; p = 16; }
        if (!(sp < -tol)) p = -1;
        if (p < 0) {
          done = true;
        } else {
          up = 0.0;
          dnp = dn_x;
          if (PC && !LAZY) dnp = (p == 16) ? dnpc : dnp;
          if (TB) sgp = (p >= 32) ? sg_x : 1.0;   // violated side of a torque row
          need_pick = false;
        }
      }
      WBC_GI_T(1);   // fetch
      all_done = qo.wave_all(done);
    }
    if (all_done) break;
    // from here on the trip is straight-line code: a finished robot runs along with a zero step and a null reflection
    const bool live = !done;
    if (live) iters++;
    WBC_GI_STAT(if (h == 0 && live) g_gi_generic_trips++);
    if (!takeover) {
      if constexpr (LAZY) {
        // a robot that is adding the dense row (a fresh pick, or the same row again after a blocked step: the basis has moved) builds its image now
        if (qo.wave_any(live && p == 16)) {
          const double n2 = dense_image(Dpc);
          dnp = (p == 16) ? n2 : dnp;
        }
      }
#pragma unroll
      for (int k = 0; k < NV; k++) {
        if (TB) d[k] = sgp * d[k];
        if (PC) d[k] = (p == 16) ? Dpc[k] : d[k];
        dm[k] = d[k] * mk[k];
        d2n = fmad(dm[k], dm[k], d2n);
      }
#pragma unroll
      for (int k = 0; k < NV; k++) {
        zd = fmad(Jr[k], dm[k], zd);
        sd = fmad(Dh[k], dm[k], sd);
        if (PC && !LAZY) sdpc = fmad(Dpc[k], dm[k], sdpc);
        if (TB) sdt = fmad(Dt[k], dm[k], sdt);
      }
      // dual step direction r = R^-1 d[0:q]: every active row's lane holds its row of W = R^-1
      // (slots k >= q of every W row are zero by construction, so d needs no masking here)
#pragma unroll
      for (int k = 0; k < NV; k++) {
        r_h = fmad(Wr[k], d[k], r_h);
        if (PC) r_pc = fmad(Wpc[k], d[k], r_pc);
        if (TB) r_t = fmad(Wt[k], d[k], r_t);
      }
      r_h = act_h ? r_h : 0.0;
      if (PC) r_pc = act_pc ? r_pc : 0.0;
      if (TB) r_t = act_t ? r_t : 0.0;
      // blocking multiplier: min over active rows with r > 0 of u / r
      {
        double key = (act_h && r_h > 0.0) ? hex_pack_key(u_h * fast_rcp(r_h), h) : HEX_NONE;
        if (TB) {
          const double c = (act_t && r_t > 0.0) ? u_t * fast_rcp(r_t) : HEX_NONE;
          if (c < key) key = hex_pack_key(c, 16 + h);
        }
        key = qo.min16(key);
        const int ix = hex_key_index(key);
        const bool any = key < 1e299;
        t1 = any ? key : INF;
        hd = any ? ((ix < 16) ? ix : 16 + ix) : -1;
        if (PC) {
          const double c = (act_pc && r_pc > 0.0) ? u_pc * fast_rcp(r_pc) : INF;
          if (c < t1) { t1 = c; hd = 16; }
        }
      }
    } else {
      // the abandoned fast trip's front half (a finished robot's values are never used: zero step, null reflection)
      p = ho_p; sp = ho_sp; dnp = ho_dn; up = 0.0; need_pick = false;   // norm, 1/norm and the full step length follow from d2n and sp below, as on the fast path
#pragma unroll
      for (int k = 0; k < NV; k++) { d[k] = ho_d[k]; dm[k] = d[k] * mk[k]; }
      d2n = ho_d2n; zd = ho_zd; sd = ho_sd; r_h = ho_r; t1 = (ho_hd >= 0) ? ho_t1 : INF; hd = ho_hd;
    }
    const bool have_t1 = hd >= 0;
    const bool dependent = !(d2n > 1e-22 * dnp) || q == NV;
    double nrm_d, rs_d;                           // |d[q:]| and its reciprocal from one seed (as on the fast path)
    fast_sqrt_rsq(d2n, nrm_d, rs_d);
    const double t2 = -sp * (rs_d * rs_d);   // n_p' J2 J2' n_p = |d[q:]|^2
    if (live && dependent && !have_t1) { status = ST_SINGULAR; done = true; }
    const bool go = !done;
    const bool full = go && !dependent && (!have_t1 || !(t1 < t2));
    const bool drop = go && !full;
    const double t = full ? t2 : (drop ? t1 : 0.0);
    u_h = fmad(-t, r_h, u_h);
    if (PC) u_pc = fmad(-t, r_pc, u_pc);
    if (TB) u_t = fmad(-t, r_t, u_t);
    up += t;
    {
      const double tz = (dependent || !go) ? 0.0 : t;
      z = fmad(tz, zd, z);
      sh_ = fmad(tz, sd, sh_);
      if (PC && !LAZY) spc = fmad(tz, sdpc, spc);
      if (TB) yt = fmad(tz, sdt, yt);
      sp = go ? fmad(tz, d2n, sp) : sp;
    }
    WBC_GI_T(2);   // step
    // ---- the trip's reflection: x onto slot q (append) or, after q - 1, onto the freed slot (drop)
    double x[NV], nrm = nrm_d, rsn = rs_d, c_j = zd, c_d = sd, c_p = sdpc, c_t = sdt, c_w = 0.0, c_wp = 0.0, c_wt = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) x[k] = dm[k];
    const bool anyd = qo.wave_any(drop);
    dropped = dropped || (drop && deep);
    wave_dropped = wave_dropped || (anyd && wave_deep);
    if (anyd) {
      WBC_GI_STAT(if (h == 0 && drop) g_gi_drops++);
      double w[NV];
#pragma unroll
      for (int k = 0; k < NV; k++) w[k] = qo.bcast16d((TB && hd >= 32) ? Wt[k] : Wr[k], hd & 15);
      if (PC) {
#pragma unroll
        for (int k = 0; k < NV; k++) w[k] = (hd == 16) ? Wpc[k] : w[k];
      }
      // q - 1 for the dropping robots: mk gains the one-hot of slot q - 1 (= mk[k+1] - mk[k])
      const double dl = drop ? 1.0 : 0.0;
      if (drop) q--;
#pragma unroll
      for (int k = 0; k < NV; k++) mk[k] = fmad(dl, ((k + 1 < NV) ? mk[k + 1] : 1.0) - mk[k], mk[k]);
      double wn = 0.0, wj = 0.0, wdh = 0.0, ww = 0.0, wp1 = 0.0, wp2 = 0.0, wt1 = 0.0, wt2 = 0.0;
#pragma unroll
      for (int k = 0; k < NV; k++) {
        wn = fmad(w[k], w[k], wn);
        wj = fmad(Jr[k], w[k], wj);
        wdh = fmad(Dh[k], w[k], wdh);
        ww = fmad(Wr[k], w[k], ww);
        if (PC) { if (!LAZY) wp1 = fmad(Dpc[k], w[k], wp1); wp2 = fmad(Wpc[k], w[k], wp2); }
        if (TB) { wt1 = fmad(Dt[k], w[k], wt1); wt2 = fmad(Wt[k], w[k], wt2); }
      }
#pragma unroll
      for (int k = 0; k < NV; k++) x[k] = drop ? w[k] : x[k];
      {
        double nw, rw;
        fast_sqrt_rsq(wn, nw, rw);
        nrm = drop ? nw : nrm; rsn = drop ? rw : rsn;
      }
      c_j = drop ? wj : c_j; c_d = drop ? wdh : c_d; c_w = drop ? ww : 0.0;
      if (PC) { c_p = drop ? wp1 : c_p; c_wp = drop ? wp2 : 0.0; }
      if (TB) { c_t = drop ? wt1 : c_t; c_wt = drop ? wt2 : 0.0; }
    }
    WBC_GI_T(3);   // the drop's vector
    double eq[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) eq[k] = (k > 0) ? mk[k] - mk[k - 1] : mk[0];
    double xq = 0.0, xq1 = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k += 2) { xq += eq[k] * x[k]; if (k + 1 < NV) xq1 += eq[k + 1] * x[k + 1]; }
    xq += xq1;
    const double alpha = (xq > 0.0) ? -nrm : nrm;
    const double ia = (xq > 0.0) ? -rsn : rsn;
    const double beta = (full || drop) ? fast_rcp(nrm * (nrm + fabs(xq))) : 0.0;   // 2 / (v'v); null reflection for a robot that rests
    // y . v = y . x - alpha y_q (the dots with x are there already; y_q picked by the one-hot mask)
    double jq = 0.0, dhq = 0.0, dpq = 0.0, dtq = 0.0;
    double hv[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
      hv[k] = fmad(-alpha, eq[k], x[k]);   // entry q: x_q - alpha
      jq = fmad(eq[k], Jr[k], jq);
      dhq = fmad(eq[k], Dh[k], dhq);
      if (PC && !LAZY) dpq = fmad(eq[k], Dpc[k], dpq);
      if (TB) dtq = fmad(eq[k], Dt[k], dtq);
    }
    const double wj_ = fmad(-alpha, jq, c_j) * beta, wd = fmad(-alpha, dhq, c_d) * beta;
    const double wp = PC ? fmad(-alpha, dpq, c_p) * beta : 0.0, wt = TB ? fmad(-alpha, dtq, c_t) * beta : 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      Jr[k] = fmad(-wj_, hv[k], Jr[k]);
      Dh[k] = fmad(-wd, hv[k], Dh[k]);
      if (PC && !LAZY) Dpc[k] = fmad(-wp, hv[k], Dpc[k]);
      if (TB) Dt[k] = fmad(-wt, hv[k], Dt[k]);
    }
    if (anyd) {
      // W' = W H on the rows; the freed slot is zero again in every W row (appends add into it: an appending robot's slot q
      // is still zero here, so clearing it is harmless), the dropped row's own W row is cleared
      double wrq = 0.0, wpq = 0.0, wtq = 0.0;
#pragma unroll
      for (int k = 0; k < NV; k++) {
        wrq = fmad(eq[k], Wr[k], wrq);
        if (PC) wpq = fmad(eq[k], Wpc[k], wpq);
        if (TB) wtq = fmad(eq[k], Wt[k], wtq);
      }
      const double cw = fmad(-alpha, wrq, c_w) * beta;
      const double cwp = PC ? fmad(-alpha, wpq, c_wp) * beta : 0.0, cwt = TB ? fmad(-alpha, wtq, c_wt) * beta : 0.0;
      const bool mined = drop && (h == hd), tmd = TB && drop && (hd == 32 + h), pmd = PC && drop && (hd == 16);
      const double keep = mined ? 0.0 : 1.0, keept = tmd ? 0.0 : 1.0, keepp = pmd ? 0.0 : 1.0;
#pragma unroll
      for (int k = 0; k < NV; k++) {
        double a = fmad(-cw, hv[k], Wr[k]);
        a = fmad(-eq[k], a, a);
        Wr[k] = a * keep;
        if (PC) { double b = fmad(-cwp, hv[k], Wpc[k]); b = fmad(-eq[k], b, b); Wpc[k] = b * keepp; }
        if (TB) { double b = fmad(-cwt, hv[k], Wt[k]); b = fmad(-eq[k], b, b); Wt[k] = b * keept; }
      }
      u_h = mined ? 0.0 : u_h;
      act_h = act_h && !mined;
      if (APEX) nleg -= (drop && hd >= 0 && hd < 16) ? same_leg(hd) : 0;
      if (TB) { u_t = tmd ? 0.0 : u_t; act_t = act_t && !tmd; }
      if (PC) { u_pc = pmd ? 0.0 : u_pc; act_pc = act_pc && !pmd; }
    }
    // append: W' = [W, -r/alpha; 0, 1/alpha]  (rows of inactive lanes are zero, r_h = 0 there; slot q is zero beforehand)
    {
      const bool mine = full && (h == p);
      const double wq = full ? (mine ? ia : -r_h * ia) : 0.0;
#pragma unroll
      for (int k = 0; k < NV; k++) Wr[k] = fmad(eq[k], wq, Wr[k]);
      u_h = mine ? up : u_h;
      act_h = act_h || mine;
      if (APEX) nleg += (full && p >= 0 && p < 16) ? same_leg(p) : 0;
      if (TB) {
        const bool tm = full && (p == 32 + h);
        const double wqt = full ? (tm ? ia : -r_t * ia) : 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) Wt[k] = fmad(eq[k], wqt, Wt[k]);
        u_t = tm ? up : u_t; act_t = act_t || tm; sig_t = tm ? sgp : sig_t;
      }
      if (PC) {
        const bool pm = full && (p == 16);
        const double wqp = full ? (pm ? ia : -r_pc * ia) : 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) Wpc[k] = fmad(eq[k], wqp, Wpc[k]);
        u_pc = pm ? up : u_pc; act_pc = act_pc || pm;
      }
      const double fl = full ? 1.0 : 0.0;
      if (full) { q++; need_pick = true; }
#pragma unroll
      for (int k = 0; k < NV; k++) mk[k] = fmad(-fl, eq[k], mk[k]);
    }
    WBC_GI_T(4);   // reflection
  }
  WBC_GI_TEND();
  // ---- A robot that has DROPPED a row gets its solution from the rotated right-hand side instead of the accumulated steps:
  //     z = J[:, q:] (y[q:] - sum over the active rows of u_a D_a[q:]),   u_a = W_a . y[:q]  (the multipliers, fresh).
  // y = Q'b rides through every reflection in the row slots of lane (0, 3), which owns no row of J.  The first term is the
  // projection in exact arithmetic; the second is the first-order correction for what the active images have LEFT in the free
  // slots (a drop frees its slot through the blocking row's W row, an inverse only to eps x cond: the remaining images keep
  // ~1e-13 of their norm there, and a residual in a body-task slot, where y ~ 300, is amplified by the 1e4 columns of J).
  // Where the digits went was located in round 3 (profiles/r03/truth.md); what closes it was found in round 4
  // (profiles/r04/accuracy.md): on saturated 4-contact stands the accumulated z is off by up to 2e-6 (ID) / 6e-6 (MPTC) of the exact
  // projection on the final active set, the evaluated one alone just as much, this one by 5e-9 -- the level of a fresh QR of the
  // active images.  `deep` = the robot has three or four feet on the ground, i.e. three or six internal-force directions: with two
  // (every trot) there is one, and the robots of a trot batch that drop a row are within 1e-12 of the extended-precision oracle
  // either way (measured on 130 of them).  Robots that never dropped keep the accumulated z; the code runs when any deep robot of
  // the wavefront has dropped, and a robot's result does not depend on its wave-mates.
  if (wave_dropped) {
    // inhomogeneous rows (dense row: n.z = vc pc_inv;  torque row: (sig Tn).z = -bt - sig t0n) put g = sum beta_a W_a into the used slots
    double yk[NV], g[NV];
    if constexpr (LAZY) { if (qo.wave_any(act_pc)) (void)dense_image(Dpc); }   // (lane (0, 3) carries y in its row slots and has no entry in the dense row's normal)
    const double b_pc = (PC && act_pc) ? vc * pc_inv : 0.0, b_t = (TB && act_t) ? -(bt + sig_t * t0n) : 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      yk[k] = qo.bcast16(Jr[k], 3);
      g[k] = 0.0;
      if (TB) g[k] = qo.sum16(b_t * Wt[k]);
      if (PC) g[k] = fmad(b_pc, Wpc[k], g[k]);
    }
    double nu = 0.0, nu_pc = 0.0, nu_t = 0.0, zr = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const double yg = (PC || TB) ? yk[k] - g[k] : yk[k];
      nu = fmad(Wr[k], yg, nu);
      if (PC) nu_pc = fmad(Wpc[k], yg, nu_pc);
      if (TB) nu_t = fmad(Wt[k], yg, nu_t);
    }
    nu = act_h ? nu : 0.0;
    if (PC) nu_pc = act_pc ? nu_pc : 0.0;
    if (TB) nu_t = act_t ? sig_t * nu_t : 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double c = qo.sum16(TB ? fmad(nu_t, Dt[k], nu * Dh[k]) : nu * Dh[k]);
      // LAZY: Dpc holds an image only where dense_image() has just built it (a wavefront with an active dense row); anywhere else it is
      // unwritten storage, and 0 * garbage is not 0 when the garbage is a NaN pattern -- select, never multiply
      if (PC) c = fmad(nu_pc, (!LAZY || act_pc) ? Dpc[k] : 0.0, c);
      const double yh = mk[k] * (yk[k] - c);
      zr = fmad(Jr[k], (PC || TB) ? yh + g[k] : yh, zr);
    }
    z = dropped ? zr : z;
  }
  }   // generic
  z = (h == 3) ? 0.0 : z;   // lane (0, 3): y . d garbage (see the entry check); no 16-lane reduction downstream may pick it up
  *iters_out = iters;
  if (!done && status == ST_OK) status = ST_ITER;
  if constexpr (WARM) *seed = act_h;
  return status;
}

// 6x6 solve without row exchanges for NR right-hand sides held as extra columns.  G_b = M_bb - sum X Jfb is
// the base's 6x6 composite inertia (symmetric positive definite, dominant) minus the light legs' coupling,
// so elimination in natural order is stable; a collapsed pivot is reported through the returned
// "min|pivot| > 1e-12 max|pivot|" (false also for NaN; the caller turns it into status 2) -- a product, not a quotient.
// Row exchanges on register arrays cost ~400 v_cndmask per tick on this kernel.
template <int NR> WBC_HD bool solve6np(double (*Ab)[6 + NR]) {
  double pmin = 0.0, pmax = 0.0;
#pragma unroll
  for (int c = 0; c < 6; c++) {
    const double best = fabs(Ab[c][c]);
    if (c == 0 || best < pmin) pmin = best;
    if (best > pmax) pmax = best;
    const double id = fast_rcp(Ab[c][c]);
#pragma unroll
    for (int j = c + 1; j < 6 + NR; j++) Ab[c][j] *= id;   // row c scaled: unit diagonal
#pragma unroll
    for (int r = c + 1; r < 6; r++) {
      const double f = Ab[r][c];
#pragma unroll
      for (int j = c + 1; j < 6 + NR; j++) Ab[r][j] -= f * Ab[c][j];
    }
  }
#pragma unroll
  for (int c = 5; c >= 0; c--) {
#pragma unroll
    for (int n = 0; n < NR; n++) {
      double sacc = Ab[c][6 + n];
#pragma unroll
      for (int j = c + 1; j < 6; j++) sacc -= Ab[c][j] * Ab[j][6 + n];
      Ab[c][6 + n] = sacc;
    }
  }
  return pmin > 1e-12 * pmax;
}

// Robot-level cold storage ("park"): replicated values that are produced early and consumed late
// (G_b for the MPTC rows) wait in LDS instead of occupying 2 VGPRs each -- a spilled VGPR costs an
// exposed L2 round trip (~0.4 us measured, profiles/r02), an LDS read ~100 cycles.  All 16 lanes
// write the same value to the same address.  Host: a plain array.
enum { PK_GS = 0, PK_ST = 36, PK_N = 36 + 21 };   // G_b (MPTC rows) | state values that sleep through the leg phase
// Lane-private park (lput / lget): what only the output stage needs again (own rows of the torque map, own column of
// B, the metrics ...) leaves the registers for the QR and the active set -- one ds_write_b64 / ds_read_b64 per double
// instead of the two v_accvgpr_write + two v_accvgpr_read the register allocator spends on a value it keeps in an
// AGPR, and ~70 fewer live registers in the two hottest phases (profiles/r02/hex_cuts.md).
enum { LP_YROW = 0, LP_DROW = 6, LP_T0 = 9, LP_BCOL = 10, LP_AB0 = 16, LP_JIC = 22, LP_RF = 25, LP_BC = 28, LP_MET = 31, LP_N = 36 };
struct ParkHost {
  double d[PK_N], l[LP_N];
  void put(int i, double v) { d[i] = v; }
  double get(int i) const { return d[i]; }
  void lput(int i, double v) { l[i] = v; }
  double lget(int i) const { return l[i]; }
};

// Diagnostic builds only (-DWBC_HCUT=k): return after phase k with the live values folded into the
// outputs; timed as whole kernels (tools/build_cuts.sh, tools/run_cuts.sh) -> profiles/r02/hex_cuts.md.
#ifdef WBC_HCUT
#define WBC_HCUT_AT(k, expr)                                                  \
  if (WBC_HCUT == (k)) {                                                      \
    double sink_ = (expr);                                                    \
    if (colv) out_tau(m.act_inv[3 * l + sb], sink_);                          \
    out_met(0, sink_); out_met(1, sink_); out_met(2, 0.0); out_met(3, 0.0);   \
    *iters_out = 0;                                                           \
    return ST_OK;                                                             \
  }
#else
#define WBC_HCUT_AT(k, expr)
#endif

// per-lane pick of element `sb` (0..2) of a replicated triple
WBC_HD double pick3(int sb, double a, double b, double c) { return (sb == 0) ? a : ((sb == 1) ? b : c); }

// The tick.  Every lane of the row calls this with its own Q (lane id h = 4*leg + sub).
// out_tau(row, x): lane (leg, j<3) writes the torque of joint 3*leg+j;  out_met: see the kernel.
template <class Q, int KIND, bool TB, bool WARM = false, class Park, class In, class OutTau, class OutMet>
WBC_HD int hex_tick(const ModelC& m, const ParamsX& P, Q& qo, In in, unsigned mask, double mu, double mass_scale,
                    Park& pk, OutTau out_tau, OutMet out_met, int* iters_out, bool* seed = nullptr) {
  const int h = qo.lane();
  const int l = h >> 2, sb = h & 3;
  const bool ct = (mask >> l) & 1u;
  constexpr bool MP = (KIND == KIND_MPTC || KIND == KIND_PC);   // task-space passivity laws (xi passes, Lambda)
  constexpr int NV = (KIND == KIND_CLF) ? NZ + 1 : NZ;          // reduced variables: z (12) [+ slack delta]
  const bool colv = sb < 3;  // owns a z-space column (else: the right-hand-side / ab0 column, or delta)
  const bool cold = (KIND == KIND_CLF) && h == HEX_DELTA_LANE;   // owns the delta column (CLF law)
  int status = ST_OK;
  bool illc = false;
  WBC_HCUT_AT(0, in(0) + in(20) + in(40) + in(37 + 18 + 9 * l + sb) + mu + mass_scale + m.gravity)
  // ---------------- state (replicated on the 16 lanes)
  double R0[9];
  {
    const double qw = in(0), qx = in(1), qy = in(2), qz = in(3);
    const double qn2 = qw * qw + qx * qx + qy * qy + qz * qz;
    // Drake's RotationMatrix(quaternion) scales by 2 / |q|^2: any non-zero finite quaternion stands for its normalised self; a zero one for nothing
    const double s = 2.0 * fast_rcp(qn2);
    R0[0] = 1.0 - s * (qy * qy + qz * qz); R0[1] = s * (qx * qy - qw * qz); R0[2] = s * (qx * qz + qw * qy);
    R0[3] = s * (qx * qy + qw * qz); R0[4] = 1.0 - s * (qx * qx + qz * qz); R0[5] = s * (qy * qz - qw * qx);
    R0[6] = s * (qx * qz - qw * qy); R0[7] = s * (qy * qz + qw * qx); R0[8] = 1.0 - s * (qx * qx + qy * qy);
  }
  const double p0[3] = {in(4), in(5), in(6)};
  const double w0[3] = {in(19), in(20), in(21)};
  const double v0[3] = {in(22), in(23), in(24)};
  const double gz = m.gravity;
  // mu and the mass scale also enter COMPARISONS, which a NaN (or a negative value) passes silently: an instance whose mu or mass scale is not a positive finite
  // number gets a NaN base mass, which no torque survives (the sentinel of the output stage reports it)
  const double bm = (!(mu > 0.0) | not_finite(mu) | !(mass_scale > 0.0) | not_finite(mass_scale)) ? __builtin_nan("") : m.base_mass * mass_scale;
  double bmc[3], bI[6];
  {
    const double t[3] = {m.base_mc[0] * mass_scale, m.base_mc[1] * mass_scale, m.base_mc[2] * mass_scale};
    rotv(R0, t, bmc);
    rot_inertia(R0, m.base_I, bI);
    for (int i = 0; i < 6; i++) bI[i] *= mass_scale;
  }
  double rpy[3], E[9], rpyd[3];
  {
    // one atan2 evaluation instead of three: sub-lane i evaluates angle i (same routine, other arguments), then the
    // three results are shared inside the quad (f64 atan2 is ~150 instructions)
    const double cp = sqrt(R0[0] * R0[0] + R0[3] * R0[3]), sp = -R0[6];
    {
      const double ang = atan2(pick3(sb, R0[7], -R0[6], R0[3]), pick3(sb, R0[8], cp, R0[0]));
      rpy[0] = qo.leg_bcast(ang, 0); rpy[1] = qo.leg_bcast(ang, 1); rpy[2] = qo.leg_bcast(ang, 2);
    }
    const double icp = fast_rcp(cp);
    const double cy = R0[0] * icp, sy = R0[3] * icp;
    E[0] = cp * cy; E[1] = -sy; E[2] = 0.0; E[3] = cp * sy; E[4] = cy; E[5] = 0.0; E[6] = -sp; E[7] = 0.0; E[8] = 1.0;
    const double Ei[9] = {cy * icp, sy * icp, 0.0, -sy, cy, 0.0, cy * sp * icp, sy * sp * icp, 1.0};
    rotv(Ei, w0, rpyd);
  }
  double xt_b[6], xdt_b[6], xdd_b[6], ades[6];
  {
    double tg_pb[3], tg_pdb[3], tg_pddb[3], tg_rpy[3], tg_rpyd[3], tg_rpydd[3];
    for (int i = 0; i < 3; i++) {
      tg_pb[i] = in(37 + i); tg_pdb[i] = in(40 + i); tg_pddb[i] = in(43 + i);
      tg_rpy[i] = in(46 + i); tg_rpyd[i] = in(49 + i); tg_rpydd[i] = in(52 + i);
    }
    for (int i = 0; i < 3; i++) { xt_b[i] = rpy[i] - tg_rpy[i]; xt_b[3 + i] = p0[i] - tg_pb[i]; }
    if (KIND == KIND_ID) {
      double rpydd_des[3], od[3];
      for (int i = 0; i < 3; i++) {
        ades[3 + i] = tg_pddb[i] - P.Kp_body_p * xt_b[3 + i] - P.Kd_body_p * (v0[i] - tg_pdb[i]);
        rpydd_des[i] = tg_rpydd[i] - P.Kp_body_rpy * xt_b[i] - P.Kd_body_rpy * (rpyd[i] - tg_rpyd[i]);
      }
      rotv(E, rpydd_des, od);
      for (int i = 0; i < 3; i++) { ades[i] = od[i]; xdt_b[i] = 0.0; xdt_b[3 + i] = 0.0; xdd_b[i] = 0.0; xdd_b[3 + i] = 0.0; }
    } else {
      double om_rt[3], xdn[3], xddn[3];
      rotv(E, rpyd, om_rt);      // literal E(rpy) * rpyd round trip of mptc_controller.py:245
      rotv(E, tg_rpyd, xdn);
      rotv(E, tg_rpydd, xddn);
      for (int i = 0; i < 3; i++) {
        xdt_b[i] = om_rt[i] - xdn[i];
        xdt_b[3 + i] = v0[i] - tg_pdb[i];
        xdd_b[i] = xddn[i];
        xdd_b[3 + i] = tg_pddb[i];
        ades[i] = 0.0; ades[3 + i] = 0.0;
      }
    }
  }
  // robot-level values that the leg phase does not touch wait in LDS (the leg phase is the register peak of the tick)
  for (int i = 0; i < 6; i++) { pk.put(PK_ST + i, xt_b[i]); pk.put(PK_ST + 6 + i, xdd_b[i]); pk.put(PK_ST + 12 + i, bI[i]); }
  for (int i = 0; i < 3; i++) pk.put(PK_ST + 18 + i, bmc[i]);
  WBC_STAMP(10);
  WBC_HCUT_AT(1, xt_b[0] + xt_b[4] + xdt_b[1] + xdt_b[5] + xdd_b[2] + ades[1] + bI[3] + bmc[1] + R0[5] + rpyd[0] + E[3])
  // ---------------- own leg (replicated on its four sub-lanes unless noted)
  double rf[3], Jdv[3], pd[3], rd[3], hl[3];
  double X[18], Y[18], Pm[9], Jl[9], Ji[9], Mll6[6];
  double hbN[6];
  double lm = 0.0, lh[3] = {0.0, 0.0, 0.0}, lI[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double Cb_leg[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, Cl[3] = {0.0, 0.0, 0.0}, xi[3] = {0.0, 0.0, 0.0};
  double xt_s[3], xdt_s[3], xdd_s[3];
  double jdxi[3] = {0.0, 0.0, 0.0};  // Jd xi (MPTC swing rows)
  {
    LegKin<double> K;
    LegDyn<double> D;
    double qd[3];
    const double mass3[3] = {m.link[l][0].mass, m.link[l][1].mass, m.link[l][2].mass};
    {
      double sn[3], cs[3], th[3];
      for (int k = 0; k < 3; k++) {
        const int row = m.q_perm[3 * l + k];
        th[k] = in(7 + row);
        qd[k] = in(25 + row);
      }
      {
        // sub-lane k evaluates the sine / cosine of joint k; shared inside the quad
        double so, co;
        wbc_sincos(pick3(sb, th[0], th[1], th[2]), so, co);
        for (int k = 0; k < 3; k++) { sn[k] = qo.leg_bcast(so, k); cs[k] = qo.leg_bcast(co, k); }
        if ((KIND == KIND_ID || KIND == KIND_CLF) && !ct) knee_clamp(sn[2], cs[2]);   // swing legs of the ID-type laws (wbc_tick.hpp)
        if (MP) illc = qo.any16(fabs(sn[2]) < KNEE_ILLCOND);   // a nearly straight knee anywhere: reported as status 3 (wbc_tick.hpp)
      }
      leg_fk_xyy(m, l, R0, sn, cs, K);
    }
    leg_crba(mass3, K, D, lm, lh, lI);
    for (int i = 0; i < 3; i++) rf[i] = K.rf(i);
    for (int k = 0; k < 3; k++) {
      const double d[3] = {rf[0] - K.r(k, 0), rf[1] - K.r(k, 1), rf[2] - K.r(k, 2)};
      const double axv[3] = {K.ax(k, 0), K.ax(k, 1), K.ax(k, 2)};
      double c[3];
      cross(axv, d, c);
      for (int i = 0; i < 3; i++) Jl[3 * i + k] = c[i];
    }
    const double det = inv3_fast(Jl, Ji);
    if (qo.any16(!(fabs(det) > 1e-12))) status = ST_SINGULAR;
    double Mf[9];
    sym_to_full(D.Mll, Mf);
    mm3(Mf, Ji, Pm);  // Pm = Mll Ji
    for (int i = 0; i < 6; i++) Mll6[i] = D.Mll[i];
    // foot velocity relative to the base origin: w0 x rf + Jl qd
    {
      double t[3];
      cross(w0, rf, t);
      for (int i = 0; i < 3; i++) {
        rd[i] = t[i] + Jl[3 * i] * qd[0] + Jl[3 * i + 1] * qd[1] + Jl[3 * i + 2] * qd[2];
        pd[i] = v0[i] + rd[i];
      }
    }
    for (int i = 0; i < 3; i++) {
      const double pf = p0[i] + rf[i];
      const double tp = in(37 + 18 + 9 * l + i), tpd = in(37 + 21 + 9 * l + i), tpdd = in(37 + 24 + 9 * l + i);
      xt_s[i] = ct ? 0.0 : pf - tp;
      xdt_s[i] = ct ? 0.0 : pd[i] - tpd;
      xdd_s[i] = ct ? 0.0 : tpdd;
    }
    if (MP) {
      double Mli[9];
      inv3_fast(Mf, Mli);
      double t[3], jfb[3];
      cross(xdt_b, rf, t);
      for (int i = 0; i < 3; i++) jfb[i] = xdt_b[3 + i] + t[i];
      if (ct) {
        double yx[3];
        for (int i = 0; i < 3; i++) {
          double s = 0.0;
          for (int j = 0; j < 6; j++) s += D.Mbl[3 * j + i] * xdt_b[j];
          yx[i] = s - (Pm[3 * i] * jfb[0] + Pm[3 * i + 1] * jfb[1] + Pm[3 * i + 2] * jfb[2]);
        }
        for (int i = 0; i < 3; i++)
          xi[i] = -(Mli[3 * i] * yx[0] + Mli[3 * i + 1] * yx[1] + Mli[3 * i + 2] * yx[2]) -
                  (Ji[3 * i] * jfb[0] + Ji[3 * i + 1] * jfb[1] + Ji[3 * i + 2] * jfb[2]);
      } else {
        const double y[3] = {xdt_s[0] - jfb[0], xdt_s[1] - jfb[1], xdt_s[2] - jfb[2]};
        rotv(Ji, y, xi);
      }
    }
    // Newton-Euler: sub-lanes 0 and 3 run the bias pass h(v) (with gravity); for the MPTC laws
    // sub-lane 1 runs h(v + xi) and sub-lane 2 h(v - xi) (gravity-free):  C xi = 1/4 [h(v+xi) - h(v-xi)].
    {
      const double c1 = !MP ? 0.0 : ((sb == 1) ? 1.0 : ((sb == 2) ? -1.0 : 0.0));
      const double gq = (!MP || sb == 0 || sb == 3) ? gz : 0.0;
      const double wv[3] = {w0[0] + c1 * xdt_b[0], w0[1] + c1 * xdt_b[1], w0[2] + c1 * xdt_b[2]};
      const double qv[3] = {qd[0] + c1 * xi[0], qd[1] + c1 * xi[1], qd[2] + c1 * xi[2]};
      double hq[3], Nq[3], Fq[3];
      leg_rnea<double, true>(mass3, K, wv, qv, gq, hq, Nq, Fq, &D);
      if (!MP) {
        for (int i = 0; i < 3; i++) { hl[i] = hq[i]; hbN[i] = Nq[i]; hbN[3 + i] = Fq[i]; Jdv[i] = D.Jdv[i]; }
      } else {
        double jx[3];
        for (int i = 0; i < 3; i++) jx[i] = D.Jd[3 * i] * xi[0] + D.Jd[3 * i + 1] * xi[1] + D.Jd[3 * i + 2] * xi[2];
        for (int i = 0; i < 3; i++) {
          hl[i] = qo.leg_bcast(hq[i], 0);
          hbN[i] = qo.leg_bcast(Nq[i], 0);
          hbN[3 + i] = qo.leg_bcast(Fq[i], 0);
          Jdv[i] = qo.leg_bcast(D.Jdv[i], 0);
          jdxi[i] = qo.leg_bcast(jx[i], 0);
          Cl[i] = 0.25 * (qo.leg_bcast(hq[i], 1) - qo.leg_bcast(hq[i], 2));
          Cb_leg[i] = 0.25 * (qo.leg_bcast(Nq[i], 1) - qo.leg_bcast(Nq[i], 2));
          Cb_leg[3 + i] = 0.25 * (qo.leg_bcast(Fq[i], 1) - qo.leg_bcast(Fq[i], 2));
        }
      }
    }
    // X = Mbl Ji, Y = Mbl' - Pm Jfb
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 3; j++)
        X[3 * i + j] = D.Mbl[3 * i] * Ji[j] + D.Mbl[3 * i + 1] * Ji[3 + j] + D.Mbl[3 * i + 2] * Ji[6 + j];
    for (int i = 0; i < 3; i++) {
      const double a0 = Pm[3 * i], a1 = Pm[3 * i + 1], a2 = Pm[3 * i + 2];
      Y[6 * i + 0] = D.Mbl[0 * 3 + i] + (a1 * rf[2] - a2 * rf[1]);
      Y[6 * i + 1] = D.Mbl[1 * 3 + i] + (a2 * rf[0] - a0 * rf[2]);
      Y[6 * i + 2] = D.Mbl[2 * 3 + i] + (a0 * rf[1] - a1 * rf[0]);
      Y[6 * i + 3] = D.Mbl[3 * 3 + i] - a0;
      Y[6 * i + 4] = D.Mbl[4 * 3 + i] - a1;
      Y[6 * i + 5] = D.Mbl[5 * 3 + i] - a2;
    }
  }
  const double bc[3] = {ct ? (-P.Kd_contact * pd[0] - Jdv[0]) : 0.0, ct ? (-P.Kd_contact * pd[1] - Jdv[1]) : 0.0,
                        ct ? (-P.Kd_contact * pd[2] - Jdv[2]) : 0.0};
  // own row of t0 (without the Y ab0 part) and of the torque map's diagonal block D_l
  // (-Jl' for a contact leg, Pm for a swing leg); meaningful on the column lanes (sub < 3)
  double t0_own, Yrow[6], Drow[3], Dcol[3];   // Dcol[i] = D_l[i][sub]: own column of the diagonal block
  {
    double t0l[3];
    for (int i = 0; i < 3; i++) t0l[i] = hl[i] + (Pm[3 * i] * bc[0] + Pm[3 * i + 1] * bc[1] + Pm[3 * i + 2] * bc[2]);
    t0_own = pick3(sb, t0l[0], t0l[1], t0l[2]);
    for (int k = 0; k < 6; k++) Yrow[k] = pick3(sb, Y[k], Y[6 + k], Y[12 + k]);
    for (int j = 0; j < 3; j++) {
      Drow[j] = ct ? -pick3(sb, Jl[3 * j], Jl[3 * j + 1], Jl[3 * j + 2]) : pick3(sb, Pm[j], Pm[3 + j], Pm[6 + j]);
      Dcol[j] = ct ? -pick3(sb, Jl[j], Jl[3 + j], Jl[6 + j]) : pick3(sb, Pm[3 * j], Pm[3 * j + 1], Pm[3 * j + 2]);
    }
  }
  for (int i = 0; i < 6; i++) { xt_b[i] = pk.get(PK_ST + i); xdd_b[i] = pk.get(PK_ST + 6 + i); bI[i] = pk.get(PK_ST + 12 + i); }
  for (int i = 0; i < 3; i++) bmc[i] = pk.get(PK_ST + 18 + i);
  WBC_STAMP(11);
  WBC_HCUT_AT(2, X[0] + X[7] + X[17] + Y[3] + Y[16] + hbN[0] + hbN[5] + lm + lh[1] + lI[3] + Cb_leg[2] + Cl[1] + xi[0] + t0_own + Yrow[2] + Drow[1] + Ji[4] + Mll6[2] + jdxi[0] + xt_s[0] + xdt_s[1] + xt_b[0] + xdt_b[4] + xdd_b[2] + ades[1])
  // ---------------- base: bias wrench, composite inertia -> Gs = G_b, kv
  double Gs[6][6], kv[6];
  {
    double hb[6];
    {
      double t2[3], t3[3], Iw_w[3], t4[3];
      const double g3[3] = {0.0, 0.0, gz};
      cross(w0, bmc, t2);
      cross(w0, t2, t2);
      symv(bI, w0, Iw_w);
      cross(w0, Iw_w, t3);
      cross(bmc, g3, t4);
      for (int i = 0; i < 3; i++) { hb[i] = t3[i] + t4[i]; hb[3 + i] = bm * g3[i] + t2[i]; }
    }
    // every sum over the legs of this phase in two batches (leg-level values are replicated on the sub-lanes: legs_sum_n)
    double Hc[3], Ic[6], Mc;
    {
      double t[16];
      for (int i = 0; i < 6; i++) t[i] = hbN[i] + (ct ? (X[3 * i] * bc[0] + X[3 * i + 1] * bc[1] + X[3 * i + 2] * bc[2]) : 0.0);
      t[6] = lm;
      for (int i = 0; i < 3; i++) t[7 + i] = lh[i];
      for (int i = 0; i < 6; i++) t[10 + i] = lI[i];
      qo.template legs_sum_n<16>(t);
      for (int i = 0; i < 6; i++) kv[i] = hb[i] + t[i];
      Mc = bm + t[6];
      for (int i = 0; i < 3; i++) Hc[i] = bmc[i] + t[7 + i];
      for (int i = 0; i < 6; i++) Ic[i] = bI[i] + t[10 + i];
    }
    double Mbb[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Mbb[i][j] = 0.0;
    Mbb[0][0] = Ic[0]; Mbb[1][1] = Ic[1]; Mbb[2][2] = Ic[2];
    Mbb[0][1] = Mbb[1][0] = Ic[3]; Mbb[0][2] = Mbb[2][0] = Ic[4]; Mbb[1][2] = Mbb[2][1] = Ic[5];
    Mbb[0][4] = -Hc[2]; Mbb[0][5] = Hc[1];
    Mbb[1][3] = Hc[2];  Mbb[1][5] = -Hc[0];
    Mbb[2][3] = -Hc[1]; Mbb[2][4] = Hc[0];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Mbb[3 + j][i] = Mbb[i][3 + j];
    Mbb[3][3] = Mc; Mbb[4][4] = Mc; Mbb[5][5] = Mc;
    {
      double g[36];
      for (int i = 0; i < 6; i++) {
        const double a0 = X[3 * i], a1 = X[3 * i + 1], a2 = X[3 * i + 2];
        g[6 * i + 0] = a1 * rf[2] - a2 * rf[1];
        g[6 * i + 1] = a2 * rf[0] - a0 * rf[2];
        g[6 * i + 2] = a0 * rf[1] - a1 * rf[0];
        g[6 * i + 3] = -a0;
        g[6 * i + 4] = -a1;
        g[6 * i + 5] = -a2;
      }
      qo.template legs_sum_n<36>(g);
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Gs[i][j] = Mbb[i][j] + g[6 * i + j];
    }
    if (MP)
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) pk.put(PK_GS + 6 * i + j, Gs[i][j]);
  }
  // own column of [B | ab0]:  G_b bcol = (W_l or -X_l)[:, sub]  (sub < 3)  |  -kv  (sub == 3)
  double bcol[6], ab0[6];
  {
    double Ab[6][7];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) Ab[i][j] = Gs[i][j];
    // W_l[:, j] = [rf x e_j ; e_j]
    const double wc[6] = {pick3(sb, 0.0, -rf[2], rf[1]), pick3(sb, rf[2], 0.0, -rf[0]), pick3(sb, -rf[1], rf[0], 0.0),
                          pick3(sb, 1.0, 0.0, 0.0), pick3(sb, 0.0, 1.0, 0.0), pick3(sb, 0.0, 0.0, 1.0)};
    for (int i = 0; i < 6; i++) {
      const double xc = pick3(sb, X[3 * i], X[3 * i + 1], X[3 * i + 2]);
      Ab[i][6] = colv ? (ct ? wc[i] : -xc) : -kv[i];
    }
    if (!solve6np<1>(Ab)) status = ST_SINGULAR;
    for (int i = 0; i < 6; i++) {
      bcol[i] = Ab[i][6];
      ab0[i] = qo.leg_bcast(bcol[i], 3);
    }
  }
  WBC_STAMP(12);
  WBC_HCUT_AT(3, bcol[0] + bcol[5] + ab0[2] + Cb_leg[2] + Cl[1] + xi[0] + t0_own + Yrow[2] + Drow[1] + Y[4] + Ji[4] + Mll6[2] + jdxi[0] + xt_s[0] + xdt_s[1] + xt_b[0] + xdt_b[4] + xdd_b[2] + ades[1] + kv[0])
  // ---------------- level-1 rows
  const double eps = P.sq_eps, sw_b = P.sq_w_body, sw_f = P.sq_w_foot;   // formed once on the host (ParamsX)
  double Rcol[NV];
  double met_err = 0.0;
  for (int i = 0; i < 6; i++) met_err += xt_b[i] * xt_b[i];
  met_err += qo.legs_sum(xt_s[0] * xt_s[0] + xt_s[1] * xt_s[1] + xt_s[2] * xt_s[2]);
  double met_V = 0.0, met_Vdot = 0.0;
  double vrow_own = 0.0, vconst = 0.0;  // Vdot = met_Vdot + vconst + sum over column lanes of vrow_own * z
  // CLF law (clf_controller.py:48-234): swing-row data of the own row and the CLF row  g . [z; delta] <= ub
  double clf_dval = 0.0, clf_drhs = 0.0, clf_g = 0.0, clf_ub = 0.0, clf_inv = 0.0, clf_c0 = 0.0, clf_gab0 = 0.0;
  auto init_rcol = [&]()   {
    // diagonal rows: swing leg sqrt(w_foot) (ID) / 0 (MPTC); contact leg eps.  Row 3*leg+sub lives on lane (leg, sub).
    double dval, drhs;
    const double xdd_o = pick3(sb, xdd_s[0], xdd_s[1], xdd_s[2]), xt_o = pick3(sb, xt_s[0], xt_s[1], xt_s[2]),
                 xdt_o = pick3(sb, xdt_s[0], xdt_s[1], xdt_s[2]), jdv_o = pick3(sb, Jdv[0], Jdv[1], Jdv[2]);
    if (ct) { dval = eps; drhs = 0.0; }
    else if (KIND == KIND_ID) {
      const double des = xdd_o - P.Kp_foot * xt_o - P.Kd_foot * xdt_o;
      dval = sw_f; drhs = sw_f * (des - jdv_o);
    } else if (KIND == KIND_CLF) { dval = clf_dval; drhs = clf_drhs; }
    else { dval = 0.0; drhs = 0.0; }
#pragma unroll
    for (int k = 0; k < NZ; k++) {
      const double rk = (KIND == KIND_MPTC || KIND == KIND_PC) ? 0.0 : qo.bcast16(drhs, hex_piv_lane(k));   // task-space laws: no right-hand side on the diagonal rows
      Rcol[k] = colv ? ((hex_piv_lane(k) == h) ? dval : 0.0) : (cold ? 0.0 : rk);   // slot k = variable hex_piv(k)
    }
    if (KIND == KIND_CLF) Rcol[NV - 1] = cold ? sqrt(2.0 * 1000.0) : 0.0;   // w_delta delta^2 = 1/2 (sqrt(2 w) delta)^2  (:73,:206)
  };
  // level-2 rows eps (T z + t0):  T[(l',i)][(l,j)] = Y_l'[i] . B_l[:,j] + delta D_l[i][j].  (Folding them into
  // the level-1 append -- one pass of 12 pivots over 30 rows -- was measured: it spills, profiles/r02.)
  // [B_col | 1 on the right-hand-side lanes]: the rows' constants (t0, c1) ride along as a seventh term of the all-pairs blocks
  double bcol7[7];
  for (int k = 0; k < 6; k++) bcol7[k] = bcol[k];
  bcol7[6] = colv ? 0.0 : 1.0;
  auto level2_rows = [&](double* A2) {
    double Y7[7];
    for (int k = 0; k < 6; k++) Y7[k] = Yrow[k];
    Y7[6] = t0_own;
    qo.template dpp_fence<7>(Y7);
    const double se = colv ? eps : (cold ? 0.0 : -eps);   // the right-hand side carries the opposite sign
    static_for<NZ / 3>([&](auto RR) {
      // three rows at a time, their fused chains interleaved (an inline-asm result is not read for two instructions)
      constexpr int r0 = 3 * RR, r1 = r0 + 1, r2 = r0 + 2;
      constexpr int s0 = hex_lane(r0), s1 = hex_lane(r1), s2 = hex_lane(r2);
      // column lanes: D_l entry + Y_row . B_col ;  rhs lanes: t0_row + Y_row . ab0   (one fused broadcast-FMA per term)
      const bool own = colv && (r0 / 3 == l);   // rows r0..r2 belong to leg RR
      double da = own ? Dcol[0] : 0.0;
      double db = own ? Dcol[1] : 0.0;
      double dc = own ? Dcol[2] : 0.0;
      qo.template rows3_bc7<s0, s1, s2>(da, db, dc, Y7, bcol7);
      A2[r0] = se * da;
      A2[r1] = se * db;
      A2[r2] = se * dc;
    });
  };
  // Level-1 rows (P1 of them) and the 12 level-2 rows are folded in ONE append of P1 + 12 rows: 12 pivot steps
  // instead of 24 (the fused broadcast-FMA needs no copy of the pivot column, so 30 rows fit the registers).
  constexpr int P1 = (KIND == KIND_ID || KIND == KIND_CLF) ? 6 : 18;
  double Acol[P1 + NZ];
  if (KIND == KIND_ID) {
    for (int i = 0; i < 6; i++) Acol[i] = colv ? sw_b * bcol[i] : sw_b * (ades[i] - bcol[i]);
    init_rcol();
  } else if (KIND == KIND_CLF) {
    // ---------------- CLF-QP in task coordinates, unit weights on every task row (see wbc_tick.hpp for the
    // lane-per-robot original; gains are the literals of clf_controller.py:65-73)
    const double Qp_b = 5000.0, Qd_b = 200.0, Qp_f = 200.0, Qd_f = 20.0, rr = 1.0;
    const double pb12 = sqrt(Qp_b * rr), pb22 = sqrt(rr * (Qd_b + 2.0 * pb12)), pb11 = pb12 * pb22 / rr;   // CARE, closed form (:187)
    const double pf12 = sqrt(Qp_f * rr), pf22 = sqrt(rr * (Qd_f + 2.0 * pf12)), pf11 = pf12 * pf22 / rr;
    double V = 0.0, ePFe = 0.0, ub = 0.0, gdot = 0.0, gxdd = 0.0;
    for (int i = 0; i < 6; i++) {
      const double pg = pb12 * xt_b[i] + pb22 * xdt_b[i], gt = 2.0 * pg;
      V += pb11 * xt_b[i] * xt_b[i] + 2.0 * pb12 * xt_b[i] * xdt_b[i] + pb22 * xdt_b[i] * xdt_b[i];
      ePFe += pb11 * xt_b[i] * xdt_b[i] + pb12 * xdt_b[i] * xdt_b[i];
      ub += gt * xdd_b[i];
      gxdd += gt * xdd_b[i];
      const double ystar = xdd_b[i] - pg / rr - gt;
      gdot += gt * bcol[i];   // column lanes: sum_i gt_i B[i][col];  rhs lanes: sum_i gt_i ab0[i]
      Acol[i] = colv ? bcol[i] : (cold ? 0.0 : ystar - bcol[i]);
    }
    // own swing row (lane (leg, sub < 3) of a swing leg)
    const bool swl = colv && !ct;
    const double xt = pick3(sb, xt_s[0], xt_s[1], xt_s[2]), xdt = pick3(sb, xdt_s[0], xdt_s[1], xdt_s[2]),
                 xddn = pick3(sb, xdd_s[0], xdd_s[1], xdd_s[2]), jdv = pick3(sb, Jdv[0], Jdv[1], Jdv[2]);
    const double pgs = pf12 * xt + pf22 * xdt, gts = swl ? 2.0 * pgs : 0.0;
    clf_dval = 1.0;
    clf_drhs = xddn - pgs / rr - jdv - 2.0 * pgs;
    V += qo.sum16(swl ? pf11 * xt * xt + 2.0 * pf12 * xt * xdt + pf22 * xdt * xdt : 0.0);
    ePFe += qo.sum16(swl ? pf11 * xt * xdt + pf12 * xdt * xdt : 0.0);
    const double gs_jx = qo.sum16(gts * (jdv - xddn));   // sum over swing rows of gt (Jdv - xdd_nom)
    ub -= gs_jx;
    const bool any_swing = (mask & 0xFu) != 0xFu;
    const double hb = 0.5 * (pb11 + pb22), db = 0.5 * (pb11 - pb22), evb = hb + sqrt(db * db + pb12 * pb12);
    const double hf = 0.5 * (pf11 + pf22), df = 0.5 * (pf11 - pf22), evf = hf + sqrt(df * df + pf12 * pf12);
    const double qmin = any_swing ? Qd_f : Qd_b, pmax = (any_swing && evf > evb) ? evf : evb;   // gamma = min eig Q / max eig P (:188)
    clf_gab0 = qo.bcast16(gdot, 3);
    clf_ub = ub - (qmin / pmax) * V - 2.0 * ePFe - clf_gab0;
    clf_g = colv ? gdot + gts : (cold ? -1.0 : 0.0);
    clf_inv = 1.0 / sqrt(qo.sum16(clf_g * clf_g));
    clf_c0 = 2.0 * ePFe - gxdd + gs_jx;
    met_V = V;
    init_rcol();
  } else {
    // ---- MPTC in task coordinates (derivation: wbc_tick.hpp / DESIGN.md)
    double MiY[18], Mt_bl[18], Mt_ll[9];
    {
      double Mf[9], Mli[9];
      sym_to_full(Mll6, Mf);
      inv3_fast(Mf, Mli);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 6; j++) MiY[6 * i + j] = Mli[3 * i] * Y[j] + Mli[3 * i + 1] * Y[6 + j] + Mli[3 * i + 2] * Y[12 + j];
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 6; j++) Mt_bl[3 * j + i] = Ji[i] * Y[j] + Ji[3 + i] * Y[6 + j] + Ji[6 + i] * Y[12 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Mt_ll[3 * i + j] = Ji[i] * Pm[j] + Ji[3 + i] * Pm[3 + j] + Ji[6 + i] * Pm[6 + j];
    double LJ_b[6], LJ_s[3];
    {
      double Cb_base[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      for (int sgi = 0; sgi < 2; sgi++) {
        const double sg = sgi ? -0.25 : 0.25, s1 = sgi ? -1.0 : 1.0;
        const double wv[3] = {w0[0] + s1 * xdt_b[0], w0[1] + s1 * xdt_b[1], w0[2] + s1 * xdt_b[2]};
        double t2[3], t3[3], Iw_w[3];
        cross(wv, bmc, t2);
        cross(wv, t2, t2);
        symv(bI, wv, Iw_w);
        cross(wv, Iw_w, t3);
        for (int i = 0; i < 3; i++) { Cb_base[i] += sg * t3[i]; Cb_base[3 + i] += sg * t2[i]; }
      }
      double gl[3], c[3];
      for (int i = 0; i < 3; i++) gl[i] = Ji[i] * Cl[0] + Ji[3 + i] * Cl[1] + Ji[6 + i] * Cl[2];
      cross(rf, gl, c);
      {
        double loc[6];
        for (int j = 0; j < 6; j++) {
          loc[j] = Cb_leg[j] - ((j < 3) ? c[j] : gl[j - 3]);
          if (ct) loc[j] -= MiY[j] * Cl[0] + MiY[6 + j] * Cl[1] + MiY[12 + j] * Cl[2];
        }
        qo.template legs_sum_n<6>(loc);
        for (int j = 0; j < 6; j++) LJ_b[j] = Cb_base[j] + loc[j];
      }
      for (int i = 0; i < 3; i++) LJ_s[i] = ct ? 0.0 : gl[i];
    }
    double s1_s[3];
    {
      double t[3];
      cross(xdt_b, rd, t);
      for (int i = 0; i < 3; i++) s1_s[i] = ct ? 0.0 : xdd_s[i] - Jdv[i] + t[i] + jdxi[i];
    }
    double Ls_s[3], Lx_s[3];
    for (int i = 0; i < 3; i++) {
      double a = Mt_ll[3 * i] * s1_s[0] + Mt_ll[3 * i + 1] * s1_s[1] + Mt_ll[3 * i + 2] * s1_s[2];
      double b = Mt_ll[3 * i] * xdt_s[0] + Mt_ll[3 * i + 1] * xdt_s[1] + Mt_ll[3 * i + 2] * xdt_s[2];
      for (int j = 0; j < 6; j++) { a += Mt_bl[3 * j + i] * xdd_b[j]; b += Mt_bl[3 * j + i] * xdt_b[j]; }
      Ls_s[i] = ct ? 0.0 : a;
      Lx_s[i] = ct ? 0.0 : b;
    }
    double c1_s[3];
    {
      double lv = 0.0, lvd = 0.0;
      for (int i = 0; i < 3; i++) {
        c1_s[i] = LJ_s[i] - Ls_s[i] + P.Kp_foot * xt_s[i] + P.Kd_foot * xdt_s[i];
        lv += 0.5 * P.Kp_foot * xt_s[i] * xt_s[i] + 0.5 * xdt_s[i] * Lx_s[i];
        lvd += -P.Kd_foot * xdt_s[i] * xdt_s[i] + xdt_s[i] * c1_s[i];
      }
      double lvs[2] = {lv, lvd};
      qo.template legs_sum_n<2>(lvs);
      met_V += lvs[0];
      met_Vdot += lvs[1];
    }
    {
      // (Ji Jfb)' Y = [ rf x M ; M ] with M = Ji' Y (= Mt_bl, already formed): column j of the top block is rf x M[:, j]
      // column lanes: vrow_own = [swing] Lx_s[sub] + sum_i lx_i B[i][col];  rhs lanes: the same dot gives vconst
      vrow_own = (colv && !ct) ? pick3(sb, Lx_s[0], Lx_s[1], Lx_s[2]) : 0.0;
      // Lambda_bb = G_b - sum_l C_l is symmetric (a task-space inertia): only the upper triangle is formed
      // (21 instead of 36 leg sums), the lower one is mirrored from the rows already done
      // the 21 + 12 leg sums of this block in one batch
      double Lup[6][6];
      {
        double cs[33];
        int e = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
          for (int j = i; j < 6; j++) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
            double c = (i < 3) ? rf[i1] * Mt_bl[3 * j + i2] - rf[i2] * Mt_bl[3 * j + i1] : Mt_bl[3 * j + (i - 3)];
            if (ct) c += Y[i] * MiY[j] + Y[6 + i] * MiY[6 + j] + Y[12 + i] * MiY[12 + j];
            cs[e++] = c;
          }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
          cs[21 + i] = ct ? 0.0 : Mt_bl[3 * i] * s1_s[0] + Mt_bl[3 * i + 1] * s1_s[1] + Mt_bl[3 * i + 2] * s1_s[2];
          cs[27 + i] = ct ? 0.0 : Mt_bl[3 * i] * xdt_s[0] + Mt_bl[3 * i + 1] * xdt_s[1] + Mt_bl[3 * i + 2] * xdt_s[2];
        }
        qo.template legs_sum_n<33>(cs);
        e = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
#pragma unroll
          for (int j = i; j < 6; j++) { Lup[i][j] = pk.get(PK_GS + 6 * i + j) - cs[e++]; Lup[j][i] = Lup[i][j]; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
          double ls = cs[21 + i], lx = cs[27 + i], lb = 0.0;
#pragma unroll
          for (int j = 0; j < 6; j++) { ls += Lup[i][j] * xdd_b[j]; lx += Lup[i][j] * xdt_b[j]; lb += Lup[i][j] * bcol[j]; }
          const double kp = (i < 3) ? P.Kp_body_rpy : P.Kp_body_p, kd = (i < 3) ? P.Kd_body_rpy : P.Kd_body_p;
          const double c1 = LJ_b[i] - ls + kp * xt_b[i] + kd * xdt_b[i];
          met_V += 0.5 * kp * xt_b[i] * xt_b[i] + 0.5 * xdt_b[i] * lx;
          met_Vdot += -kd * xdt_b[i] * xdt_b[i] + xdt_b[i] * c1;
          vrow_own += lx * bcol[i];
          const double sw_term = ct ? 0.0 : pick3(sb, Mt_bl[3 * i], Mt_bl[3 * i + 1], Mt_bl[3 * i + 2]);
          Acol[i] = colv ? sw_b * (lb + sw_term) : -sw_b * (c1 + lb);
        }
      }
      vconst = qo.leg_bcast(vrow_own, 3);
    }
    // 12 swing rows sqrt(w_foot) [Lambda_sb a_b + Lambda_ss z_leg + c1]; zero for contact legs.
    // Row (lp, i) is owned by lane (lp, i): its Lambda_sb row (6), c1 and Lambda_ss row (3).
    {
      double Msb[6], Msc[3];   // Msc[i] = Lambda_ss[i][sub]: own column of the leg's swing block
      for (int k = 0; k < 6; k++) Msb[k] = ct ? 0.0 : pick3(sb, Mt_bl[3 * k], Mt_bl[3 * k + 1], Mt_bl[3 * k + 2]);
      for (int i = 0; i < 3; i++) Msc[i] = ct ? 0.0 : pick3(sb, Mt_ll[3 * i], Mt_ll[3 * i + 1], Mt_ll[3 * i + 2]);
      double Msb7[7];
      for (int k = 0; k < 6; k++) Msb7[k] = Msb[k];
      Msb7[6] = ct ? 0.0 : pick3(sb, c1_s[0], c1_s[1], c1_s[2]);
      qo.template dpp_fence<7>(Msb7);
      const double sf = colv ? sw_f : -sw_f;
      static_for<NZ / 3>([&](auto RR) {
        constexpr int r0 = 3 * RR, r1 = r0 + 1, r2 = r0 + 2;
        constexpr int s0 = hex_lane(r0), s1 = hex_lane(r1), s2 = hex_lane(r2);
        // column lanes: Lambda_ss entry (row r, own column; own leg only) + Lambda_sb row . B_col ;  rhs lanes: c1_row + Lambda_sb row . ab0
        const bool own = colv && (r0 / 3 == l);
        double da = own ? Msc[0] : 0.0;
        double db = own ? Msc[1] : 0.0;
        double dc = own ? Msc[2] : 0.0;
        qo.template rows3_bc7<s0, s1, s2>(da, db, dc, Msb7, bcol7);
        Acol[6 + r0] = sf * da;
        Acol[6 + r1] = sf * db;
        Acol[6 + r2] = sf * dc;
      });
    }
    init_rcol();
  }
  WBC_STAMP(13);
  WBC_HCUT_AT(4, Acol[0] + Acol[5] + Acol[P1 - 1] + Rcol[0] + met_V + met_Vdot + vrow_own + vconst + bcol[2] + ab0[1] + t0_own + Yrow[2] + Drow[1] + Ji[4])
  level2_rows(Acol + P1);
  // ---------------- what the friction phase itself needs of the end-of-tick data (torque box, PC row) ...
  double Tn[NZ], t0n = 0.0, bt = -1.0;   // torque box: own row of the torque map tau = T z + t0' (t0' = t0_own + Y_row.ab0), normalised
  if (TB) {
    double n2 = 0.0;
#pragma unroll
    for (int c = 0; c < NZ; c++) {
      const int src = hex_lane(c);
      double tc = (c / 3 == l) ? Drow[c % 3] : 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) tc += Yrow[k] * qo.bcast16(bcol[k], src);
      Tn[c] = tc;
      n2 += tc * tc;
    }
    double t0p = t0_own;
#pragma unroll
    for (int k = 0; k < 6; k++) t0p += Yrow[k] * ab0[k];
    const double inrm = (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
    for (int c = 0; c < NZ; c++) Tn[c] *= inrm;
    t0n = t0p * inrm;
    bt = (colv && n2 > 0.0) ? P.tau_max * inrm : -1.0;
  }
  const double pc_vr = colv ? vrow_own : 0.0, pc_c = vconst + met_Vdot;
  double pc_inv = 0.0;
  if (KIND == KIND_PC) {
    const double n2 = qo.sum16(pc_vr * pc_vr);
    pc_inv = (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
  }
  // ... and the rest waits in the lane's LDS slots until the output stage
  {
    for (int k = 0; k < 6; k++) { pk.lput(LP_YROW + k, Yrow[k]); pk.lput(LP_BCOL + k, bcol[k]); pk.lput(LP_AB0 + k, ab0[k]); }
    for (int k = 0; k < 3; k++) {
      pk.lput(LP_DROW + k, Drow[k]);
      pk.lput(LP_JIC + k, pick3(sb, Ji[k], Ji[3 + k], Ji[6 + k]));
      pk.lput(LP_RF + k, rf[k]);
      pk.lput(LP_BC + k, bc[k]);
    }
    pk.lput(LP_T0, t0_own);
    pk.lput(LP_MET + 0, vrow_own); pk.lput(LP_MET + 1, vconst); pk.lput(LP_MET + 2, met_V);
    pk.lput(LP_MET + 3, met_Vdot); pk.lput(LP_MET + 4, met_err);
  }
  // Task-space laws: the swing rows of a contact leg are zero and stay zero under the reflections.  When no robot of the wavefront has
  // more than two swing legs (every trot, every stand) each robot's swing blocks move up into the first two block slots -- selects
  // between registers; blocks move by whole multiples of three rows, so every term keeps its accumulator and its place in the sums:
  // the result has the same bits -- and the append runs over 24 rows instead of 30: MPTC trot N = 4096 26.9 -> 26.3 us, N = 32768 -2.5 %.
  // (A third instantiation over 18 rows for wavefronts where nobody swings was measured: nothing on the MPTC stand, whose launch is
  // the active set's, and +4 % on the trot through the larger code -- not kept.)
  const unsigned ncontact = (mask & 1u) + ((mask >> 1) & 1u) + ((mask >> 2) & 1u) + ((mask >> 3) & 1u);
  if (P1 == 18 && !qo.wave_any(ncontact < 2u)) {
    const bool s0 = !(mask & 1u), s1 = !(mask & 2u), s2 = !(mask & 4u);
    double A24[24];
    for (int i = 0; i < 6; i++) A24[i] = Acol[i];
    for (int i = 0; i < 3; i++) {
      const double b0 = Acol[6 + i], b1 = Acol[9 + i], b2 = Acol[12 + i], b3 = Acol[15 + i];   // a contact leg's block is zero
      const double a23 = s2 ? b2 : b3, a123 = s1 ? b1 : a23;
      A24[6 + i] = s0 ? b0 : a123;                                          // first swing leg's row
      A24[9 + i] = s0 ? a123 : (s1 ? a23 : (s2 ? b3 : 0.0));                // the swing leg after it
    }
    for (int i = 0; i < NZ; i++) A24[12 + i] = Acol[P1 + i];
    hex_qr_append<Q, 24, NV>(qo, Rcol, A24);
  } else {
    hex_qr_append<Q, P1 + NZ, NV>(qo, Rcol, Acol);
  }
  WBC_STAMP(14);
  WBC_HCUT_AT(5, Rcol[0] + Rcol[5] + Rcol[11] + met_V + met_Vdot + vrow_own + vconst + bcol[2] + ab0[1] + t0_own + Yrow[2] + Drow[1] + Ji[4])
  // ---------------- own row of J = R^-1 and unconstrained minimiser
  double z, Jr[NV];
  {
    const double rown = [&] { double x = 0.0;
#pragma unroll
      for (int k = 0; k < NV; k++) x = (hex_piv_lane(k) == h) ? Rcol[k] : x;
      return x; }();
    const double inv_own = fast_rcp(rown);
    double invd[NV];
#pragma unroll
    for (int c = 0; c < NV; c++) invd[c] = qo.bcast16(inv_own, hex_piv_lane(c));
    // pivot range over the column lanes (1/|R_cc|: max <-> min swap)
    const double ainv = fabs(inv_own);
    const double rmax = qo.max16((colv || cold) ? ainv : 0.0), rmin = qo.min16((colv || cold) ? ainv : HEX_NONE);
    // rmin/rmax here are of 1/|R_cc|:  min|R| / max|R| = rmin / rmax;  a zero pivot gives inf/nan -> singular
    if (!(rmin > 1e-13 * rmax) || !(rmax < 1e300)) status = ST_SINGULAR;
    if (status == ST_SINGULAR) {
      // reported, not solved: zero torques AND zero accelerations (include/wbc.h: every step writes vd; a rollout
      // integrates them, so they must be defined); a malformed instance has no position error to report either
      if (colv) out_tau(m.act_inv[3 * l + sb], 0.0);
      out_met(0, 0.0); out_met(1, 0.0); out_met(2, 0.0); out_met(3, 0.0);
      for (int i = 0; i < 6; i++) out_met(4 + i, 0.0);
      if (colv) out_met(4 + 6 + m.q_perm[3 * l + sb], 0.0);
      *iters_out = 0;
      return ST_SINGULAR;
    }
    // my row index rr = the slot of my variable 3*l + sb (column lanes; 12 on the delta lane).  J[rr][c] = (delta_rr,c - sum_{k<c} J[rr][k] R[k][c]) / R[c][c]
    int rr = NZ;
#pragma unroll
    for (int k = 0; k < NZ; k++) rr = (hex_piv_lane(k) == h) ? k : rr;
    double zacc = 0.0;
    qo.template dpp_fence<NV>(Rcol);
    static_for<NV>([&](auto CC) {
      constexpr int c = CC;
      double s = ((colv || cold) && c == rr) ? -1.0 : 0.0;   // the NEGATIVE of the row entry accumulates (no per-term negation)
      double sb2 = 0.0, sc2 = 0.0;                            // three accumulators: no fused op reads the result of the two before it
      static_for<c>([&](auto KK) {
        if constexpr (KK % 3 == 0) s = qo.template fma_bc<hex_piv_lane(c)>(s, Rcol[KK], Jr[KK]);
        else if constexpr (KK % 3 == 1) sb2 = qo.template fma_bc<hex_piv_lane(c)>(sb2, Rcol[KK], Jr[KK]);
        else sc2 = qo.template fma_bc<hex_piv_lane(c)>(sc2, Rcol[KK], Jr[KK]);
      });
      if (c > 2) s += sb2 + sc2; else if (c > 1) s += sb2;
      Jr[c] = -s * invd[c];
      zacc = qo.template fma_bc<3>(zacc, Rcol[c], Jr[c]);   // rhs column after the appends (lane (0,3))
    });
    z = zacc;
    // lane (0, 3) owns no row of J: its row slots carry y = Q'b (the right-hand-side column after the append) through the active
    // set's reflections, for the evaluation of z after a drop (hex_gi)
    // (only a robot with three or four feet down is ever evaluated that way -- `deep` in hex_gi -- so a wavefront of trotting robots skips the copy)
    if (qo.wave_any(((mask & 1u) + ((mask >> 1) & 1u) + ((mask >> 2) & 1u) + ((mask >> 3) & 1u)) >= 3u)) {
#pragma unroll
      for (int c = 0; c < NV; c++) Jr[c] = (h == 3) ? Rcol[c] : Jr[c];
    }
  }
  WBC_STAMP(15);
  WBC_HCUT_AT(6, z + Jr[0] + Jr[5] + Jr[11] + met_V + met_Vdot + vrow_own + vconst + bcol[2] + ab0[1] + t0_own + Yrow[2] + Drow[1] + Ji[4])
  // ---------------- friction rows (+ the optional torque box)
  int iters = 0;
  {
    const bool deep = ((mask & 1u) + ((mask >> 1) & 1u) + ((mask >> 2) & 1u) + ((mask >> 3) & 1u)) >= 3u;   // >= 3 internal-force directions (hex_gi)
    double s, rs;   // sqrt(1 + mu^2) and its reciprocal from one hardware seed
    fast_sqrt_rsq(1.0 + mu * mu, s, rs);
    (void)s;
    int st;
    if (KIND == KIND_PC) {
      st = hex_gi<Q, true, NV, TB, !TB, true, WARM>(qo, h, ct, Jr, z, mu * rs, rs, &iters, pc_vr, pc_c, pc_inv, Tn, t0n, bt, deep, seed);
    } else if (KIND == KIND_CLF) {
      // the CLF row  g . [z; delta] - ub <= 0  is the dense row of the active set (index 16), like PC's Vdot row
      st = hex_gi<Q, true, NV, TB, !TB, true, WARM>(qo, h, ct, Jr, z, mu * rs, rs, &iters, clf_g, -clf_ub, clf_inv, Tn, t0n, bt, deep, seed);
    } else {
      // pick rule of the friction-only laws, fixed at compile time: MPTC greatest dual gain, ID most violated row (gain pivoting for ID: 12.3 vs 11.4 trips)
      st = hex_gi<Q, false, NV, TB, (KIND == KIND_MPTC) && !TB, false, WARM>(qo, h, ct, Jr, z, mu * rs, rs, &iters, 0.0, 0.0, 0.0, Tn, t0n, bt, deep, seed);
    }
    if (st != ST_OK) status = st;
    if (status == ST_OK && illc) status = ST_ILLCOND;
  }
  *iters_out = iters;
  WBC_STAMP(6);
  WBC_HCUT_AT(7, z + (double)iters + met_V + met_Vdot + vrow_own + vconst + bcol[2] + ab0[1] + t0_own + Yrow[2] + Drow[1] + Ji[4])
  // ---------------- outputs: a_b = ab0 + sum B z ;  tau_(l,j) = Y_l[j] a_b + D_l[j] z_l + t0_l[j]
  double jic[3];
  {
    for (int k = 0; k < 6; k++) { Yrow[k] = pk.lget(LP_YROW + k); bcol[k] = pk.lget(LP_BCOL + k); ab0[k] = pk.lget(LP_AB0 + k); }
    for (int k = 0; k < 3; k++) { Drow[k] = pk.lget(LP_DROW + k); jic[k] = pk.lget(LP_JIC + k); rf[k] = pk.lget(LP_RF + k); }
    t0_own = pk.lget(LP_T0);
    vrow_own = pk.lget(LP_MET + 0); vconst = pk.lget(LP_MET + 1); met_V = pk.lget(LP_MET + 2);
    met_Vdot = pk.lget(LP_MET + 3); met_err = pk.lget(LP_MET + 4);
  }
  const double bcp[3] = {pk.lget(LP_BC + 0), pk.lget(LP_BC + 1), pk.lget(LP_BC + 2)};
  const double z0 = qo.leg_bcast(z, 0), z1 = qo.leg_bcast(z, 1), z2 = qo.leg_bcast(z, 2);
  {
    double ab[6];
    for (int i = 0; i < 6; i++) ab[i] = ab0[i] + qo.sum16(colv ? bcol[i] * z : 0.0);
    double s = t0_own;
    for (int k = 0; k < 6; k++) s += Yrow[k] * ab[k];
    s += Drow[0] * z0 + Drow[1] * z1 + Drow[2] * z2;
    // Nothing that is not a number leaves the tick (include/wbc.h "Malformed instances").  A NaN or an inf anywhere in what the law reads -- or finite inputs that
    // overflow on the way -- has reached z and a_b by now, and through them the torque of every lane (0 x inf = NaN: no term drops out); the state-only metrics
    // ride along in one more sum.  mu and the mass scale enter comparisons, which a NaN passes silently: they are tested themselves.  One 16-lane vote.
    const bool bad = not_finite(s) | not_finite(met_V + met_Vdot + met_err);
    if (qo.any16(bad)) status = ST_SINGULAR;
    if (colv) out_tau(m.act_inv[3 * l + sb], (status == ST_SINGULAR) ? 0.0 : s);
    // generalized accelerations of the QP solution (rows 4..21 of out_met); zeros with the zero torques of status 2
    const bool sing = (status == ST_SINGULAR);
    for (int i = 0; i < 6; i++) out_met(4 + i, sing ? 0.0 : ab[i]);
    double t[3], y[3];
    cross(ab, rf, t);
    const double zl[3] = {z0, z1, z2};
    for (int i = 0; i < 3; i++) y[i] = (ct ? bcp[i] : zl[i]) - (ab[3 + i] + t[i]);
    const double qdd = jic[0] * y[0] + jic[1] * y[1] + jic[2] * y[2];
    if (colv) out_met(4 + 6 + m.q_perm[3 * l + sb], sing ? 0.0 : qdd);
  }
  const bool sing = (status == ST_SINGULAR);
  const double errm = sing ? 0.0 : met_err;
  double res = 0.0;
  if (ct) res = fmax(fabs(z0) - mu * z2, fabs(z1) - mu * z2);
  res = fmax(0.0, qo.max16(res));
  // a status-2 instance reports zeros throughout (what is not zero here is finite: the sentinel above has seen V, Vdot's state part and err)
  if (KIND == KIND_CLF) {
    // Vdot = 2 eta'PF eta + 2 eta'PG (J vd + Jdv - xdd_nom)   (clf_controller.py:230)
    const double vd = clf_c0 + clf_gab0 + qo.sum16(colv ? clf_g * z : 0.0);
    out_met(0, sing ? 0.0 : met_V); out_met(1, errm); out_met(2, 0.0); out_met(3, sing ? 0.0 : vd);
  } else if (KIND != KIND_ID) {
    met_Vdot += vconst + qo.sum16(colv ? vrow_own * z : 0.0);
    out_met(0, sing ? 0.0 : met_V); out_met(1, errm); out_met(2, 0.0); out_met(3, sing ? 0.0 : met_Vdot);
  } else {
    out_met(0, 0.0); out_met(1, errm); out_met(2, sing ? 0.0 : res); out_met(3, 0.0);
  }
  return status;
}

}  // namespace wbc
