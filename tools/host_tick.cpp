// host_tick.cpp -- TEST/ANALYSIS TOOL, not product code.
// Instantiates the kernel math (quadruped_drake_amd/csrc/wbc_tick.hpp, wbc_hex.hpp; tools/wbc_scalar_tick.hpp) on the host:
//   * with `double`, so the tick arithmetic can be debugged against the oracle without a GPU;
//   * with an operation-counting scalar, which yields the frozen flops/tick figure that
//     bench.py's roofline uses.
// The shipped library (libwbc_hip.so) never calls this; it has no CPU path.
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>
#include <stdint.h>
#include <string.h>
#include "../quadruped_drake_amd/csrc/wbc_model.hpp"
#include "wbc_scalar_tick.hpp"

struct Cnt {  // counts[0]=add/sub, 1=mul, 2=div, 3=sqrt, 4=trig(sin/cos/atan2), 5=cmp
  double v;
  static thread_local uint64_t c[6];
  Cnt() : v(0) {}
  explicit Cnt(double x) : v(x) {}
};
thread_local uint64_t Cnt::c[6];
inline Cnt operator+(const Cnt& a, const Cnt& b) { Cnt::c[0]++; return Cnt(a.v + b.v); }
inline Cnt operator-(const Cnt& a, const Cnt& b) { Cnt::c[0]++; return Cnt(a.v - b.v); }
inline Cnt operator*(const Cnt& a, const Cnt& b) { Cnt::c[1]++; return Cnt(a.v * b.v); }
inline Cnt operator/(const Cnt& a, const Cnt& b) { Cnt::c[2]++; return Cnt(a.v / b.v); }
inline bool operator<(const Cnt& a, const Cnt& b) { Cnt::c[5]++; return a.v < b.v; }
inline bool operator>(const Cnt& a, const Cnt& b) { Cnt::c[5]++; return a.v > b.v; }
inline bool operator==(const Cnt& a, const Cnt& b) { Cnt::c[5]++; return a.v == b.v; }
inline Cnt sqrt(const Cnt& a) { Cnt::c[3]++; return Cnt(::sqrt(a.v)); }
inline Cnt sin(const Cnt& a) { Cnt::c[4]++; return Cnt(::sin(a.v)); }
inline Cnt cos(const Cnt& a) { Cnt::c[4]++; return Cnt(::cos(a.v)); }
inline Cnt atan2(const Cnt& a, const Cnt& b) { Cnt::c[4]++; return Cnt(::atan2(a.v, b.v)); }

template <class T> static double val(const T& x);
template <> double val<double>(const double& x) { return x; }
template <> double val<Cnt>(const Cnt& x) { return x.v; }

static double* g_vdot = nullptr;  // optional [18][stride] sink for the generalized accelerations (rows 4..21 of out_met)

template <class T>
static int run_one(int kind, const wbc::ModelC& m, const wbc::ParamsC& P, int i, int stride, const double* q,
                   const double* v, const double* tg, unsigned mask, double mu, double ms, double* tau,
                   double* met, int* iters) {
  auto in = [&](int r) -> T {
    if (r < 19) return T(q[(size_t)r * stride + i]);
    if (r < 37) return T(v[(size_t)(r - 19) * stride + i]);
    return T(tg[(size_t)(r - 37) * stride + i]);
  };
  auto ot = [&](int k, T x) { tau[(size_t)k * stride + i] = val<T>(x); };
  auto om = [&](int k, T x) {
    if (k >= 4) { if (g_vdot) g_vdot[(size_t)(k - 4) * stride + i] = val<T>(x); return; }
    if (met) met[(size_t)k * stride + i] = val<T>(x);
  };
  if (kind == wbc::KIND_ID) return wbc::tick<T, wbc::KIND_ID>(m, P, in, mask, T(mu), T(ms), ot, om, iters);
  if (kind == wbc::KIND_PC) return wbc::tick<T, wbc::KIND_PC>(m, P, in, mask, T(mu), T(ms), ot, om, iters);
  if (kind == wbc::KIND_CLF) return wbc::tick<T, wbc::KIND_CLF>(m, P, in, mask, T(mu), T(ms), ot, om, iters);
  return wbc::tick<T, wbc::KIND_MPTC>(m, P, in, mask, T(mu), T(ms), ot, om, iters);
}

#ifndef HOST_TICK_HEX_ONLY   // (the variant builds of tests/host_tick.py::build_variant only call host_hex_batch: a third of the compile time)
extern "C" {

// params12: Kp_body_p, Kd_body_p, Kp_body_rpy, Kd_body_rpy, Kp_foot, Kd_foot, w_body, w_foot, mu,
// Kd_contact, tau_max, eps2 (NULL = the reference defaults)
int host_tick_batch(int kind, const double* flat215, const double* params12, const int* q_perm,
                    const int* act_perm, int n, int stride, const double* q, const double* v,
                    const double* tg, const unsigned char* mask, const double* mu, const double* mass_scale,
                    double* tau, double* met, int* status, int* iters) {
  wbc::ModelC m;
  if (wbc::model_from_flat(flat215, &m)) return -1;
  wbc::model_set_perms(&m, q_perm, act_perm);
  wbc::ParamsC P;
  wbc::params_default(kind, &P);
  if (params12) memcpy(&P, params12, sizeof(double) * 12);
  for (int i = 0; i < n; i++) {
    int it = 0;
    int st = run_one<double>(kind, m, P, i, stride, q, v, tg, mask[i], mu ? mu[i] : P.mu,
                             mass_scale ? mass_scale[i] : 1.0, tau, met, &it);
    if (status) status[i] = st;
    if (iters) iters[i] = it;
  }
  return 0;
}

void host_set_vdot_sink(double* vdot) { g_vdot = vdot; }

// Timing driver for bench.py's cpu_baseline.reduced: `reps` passes over the n instances of the REDUCED algorithm the kernels run (12-variable QP, QR +
// Goldfarb-Idnani: the scalar one-robot instantiation of tools/wbc_scalar_tick.hpp), instances dealt to `nthreads` std::threads in contiguous chunks.
// Outputs of the last pass are kept.  Context beside the literal dense port of oracle/, never the thing measured or shipped.
int host_tick_bench(int kind, const double* flat215, int n, int stride, const double* q, const double* v, const double* tg,
                    const unsigned char* mask, const double* mu, const double* mass_scale, double* tau, int* status, int nthreads, int reps) {
  wbc::ModelC m;
  if (wbc::model_from_flat(flat215, &m) || g_vdot) return -1;
  wbc::model_set_perms(&m, nullptr, nullptr);
  wbc::ParamsC P;
  wbc::params_default(kind, &P);
  if (nthreads < 1) nthreads = 1;
  std::vector<double> met((size_t)4 * stride);
  // chunks of 8 instances (one cache line of every output row) handed out by an atomic counter: the oracle's `schedule(dynamic, 8)`
  std::atomic<long long> next{0};
  const long long chunks_per_pass = (n + 7) / 8, total = chunks_per_pass * reps;
  auto work = [&](int) {
    for (;;) {
      const long long c = next.fetch_add(1);
      if (c >= total) break;
      const int lo = (int)(c % chunks_per_pass) * 8, hi = lo + 8 < n ? lo + 8 : n;
      for (int i = lo; i < hi; i++) {
        int it = 0;
        const int st = run_one<double>(kind, m, P, i, stride, q, v, tg, mask[i], mu ? mu[i] : P.mu, mass_scale ? mass_scale[i] : 1.0, tau, met.data(), &it);
        if (status) status[i] = st;
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return 0;
}

// Mean operation counts per tick over the batch: out[6] = add, mul, div, sqrt, trig, cmp.
int host_tick_count(int kind, const double* flat215, const double* params12, int n, int stride,
                    const double* q, const double* v, const double* tg, const unsigned char* mask,
                    const double* mu, const double* mass_scale, double* out6) {
  wbc::ModelC m;
  if (wbc::model_from_flat(flat215, &m)) return -1;
  wbc::ParamsC P;
  wbc::params_default(kind, &P);
  if (params12) memcpy(&P, params12, sizeof(double) * 12);
  for (int k = 0; k < 6; k++) Cnt::c[k] = 0;
  double tau[12 * 1], met[4];
  for (int i = 0; i < n; i++) {
    int it = 0;
    double t1[12], m1[4];
    // write outputs to a private 1-wide scratch (stride trick: use i=0, own buffers)
    auto in = [&](int r) -> Cnt {
      if (r < 19) return Cnt(q[(size_t)r * stride + i]);
      if (r < 37) return Cnt(v[(size_t)(r - 19) * stride + i]);
      return Cnt(tg[(size_t)(r - 37) * stride + i]);
    };
    auto ot = [&](int k, Cnt x) { t1[k] = x.v; };
    auto om = [&](int k, Cnt x) { if (k < 4) m1[k] = x.v; };
    Cnt muv(mu ? mu[i] : P.mu), msv(mass_scale ? mass_scale[i] : 1.0);
    if (kind == wbc::KIND_ID) wbc::tick<Cnt, wbc::KIND_ID>(m, P, in, mask[i], muv, msv, ot, om, &it);
    else if (kind == wbc::KIND_PC) wbc::tick<Cnt, wbc::KIND_PC>(m, P, in, mask[i], muv, msv, ot, om, &it);
    else if (kind == wbc::KIND_CLF) wbc::tick<Cnt, wbc::KIND_CLF>(m, P, in, mask[i], muv, msv, ot, om, &it);
    else wbc::tick<Cnt, wbc::KIND_MPTC>(m, P, in, mask[i], muv, msv, ot, om, &it);
  }
  (void)tau; (void)met;
  for (int k = 0; k < 6; k++) out6[k] = (double)Cnt::c[k] / n;
  return 0;
}
}

#endif   // HOST_TICK_HEX_ONLY

#include <vector>

// ---------------------------------------------------------------------------------------------
// Host emulation of the 16-lanes-per-robot kernel (wbc_hex.hpp): 16 cooperative fibres (ucontext,
// one OS thread, round-robin) play the 16 lanes of a DPP row; every cross-lane op is
// "publish, barrier, read, barrier".  The reductions use the association of the device butterflies
// (quad_perm xor 1, xor 2, then row_ror:8, row_ror:4), so replicated values are bit-identical.
#include <ucontext.h>
#define WBC_HOST_GI_STATS 1
int g_gi_fast_trips = 0, g_gi_generic_trips = 0, g_gi_drops = 0, g_gi_force_bail = -1;
double* g_gi_dump = nullptr;
static double* g_gi_dump_base = nullptr;
extern "C" void host_gi_dump(double* buf) { g_gi_dump_base = buf; }   // analysis: [n][16][16] inputs of the active set (hex path)
extern "C" void host_gi_force_bail(int qc) { g_gi_force_bail = qc; }   // tests: leave the fast path at trip qc (a wave-mate's drop)
extern "C" void host_gi_stats(int* out, int reset) {
  out[0] = g_gi_fast_trips; out[1] = g_gi_generic_trips; out[2] = g_gi_drops;
  if (reset) g_gi_fast_trips = g_gi_generic_trips = g_gi_drops = 0;
}
#include "../quadruped_drake_amd/csrc/wbc_hex.hpp"

namespace {
struct HexCtx {
  double slot[16];
  int islot[16];
  int count = 0;
  unsigned gen = 0;
  ucontext_t ctx[16], main;
  bool finished[16];
  int cur = 0;
  void (*body)(int) = nullptr;
};
HexCtx* g_hex = nullptr;

void hex_switch_from(int me) {
  HexCtx* c = g_hex;
  for (int d = 1; d <= 16; d++) {
    const int nx = (me + d) % 16;
    if (!c->finished[nx]) {
      if (nx == me) return;
      c->cur = nx;
      swapcontext(&c->ctx[me], &c->ctx[nx]);
      return;
    }
  }
  swapcontext(&c->ctx[me], &c->main);  // everybody finished
}
void hex_barrier(int me) {
  HexCtx* c = g_hex;
  const unsigned g = c->gen;
  if (++c->count == 16) { c->count = 0; c->gen++; return; }
  while (c->gen == g) hex_switch_from(me);
}
void hex_entry(int lane) {
  HexCtx* c = g_hex;
  c->body(lane);
  hex_barrier(lane);  // all lanes leave together
  c->finished[lane] = true;
  hex_switch_from(lane);
}
}  // namespace

struct HexHost {
  int h;
  int lane() const { return h; }
  double xchg(double x, int src) {
    g_hex->slot[h] = x; hex_barrier(h);
    const double r = g_hex->slot[src]; hex_barrier(h);
    return r;
  }
  double bcast16(double x, int src) { return xchg(x, src); }
  template <int SRC> double fma_bc(double acc, double x, double y) { return acc + xchg(x, SRC) * y; }   // device: one v_fmac_f64_dpp
  template <int N> static void dpp_fence(double*) {}                                            // device: hazard fence
  template <int SRC, int N> void dot_bc(double& ta, double& tb, double& tc, const double* a) {
    for (int i = 0; i < N; i++) {
      const double t = xchg(a[i], SRC) * a[i];
      if (i % 3 == 0) ta += t; else if (i % 3 == 1) tb += t; else tc += t;
    }
  }
  template <int L0, int L1, int L2> void rows3_bc(double& d0, double& d1, double& d2, const double* x, const double* y) {
    for (int k = 0; k < 6; k++) { d0 += xchg(x[k], L0) * y[k]; d1 += xchg(x[k], L1) * y[k]; d2 += xchg(x[k], L2) * y[k]; }
  }
  template <int L0, int L1, int L2> void rows3_bc7(double& d0, double& d1, double& d2, const double* x, const double* y) {
    for (int k = 0; k < 7; k++) { d0 += xchg(x[k], L0) * y[k]; d1 += xchg(x[k], L1) * y[k]; d2 += xchg(x[k], L2) * y[k]; }
  }
  double leg_bcast(double x, int s0) { return xchg(x, (h & ~3) | s0); }
  double leg_pairs(double x) { return xchg(x, (h & ~3) | ((h & 3) >> 1)); }   // quad_perm [0,0,1,1]
  double bcast16d(double x, int src) { return xchg(x, src); }                  // dynamic (robot-uniform) source lane
  int bcast16d_i(int x, int src) {
    g_hex->islot[h] = x; hex_barrier(h);
    const int r = g_hex->islot[src]; hex_barrier(h);
    return r;
  }
  static double quad_of(const double* s, int b) { return (s[b] + s[b ^ 1]) + (s[b ^ 2] + s[b ^ 3]); }
  double leg_sum(double x) {
    g_hex->slot[h] = x; hex_barrier(h);
    const double r = quad_of(g_hex->slot, h); hex_barrier(h);
    return r;
  }
  double legs_sum(double x) {
    g_hex->slot[h] = x; hex_barrier(h);
    const double* s = g_hex->slot;
    const double r = (s[h] + s[h ^ 8]) + (s[h ^ 4] + s[h ^ 12]); hex_barrier(h);
    return r;
  }
  template <int N> void legs_sum_n(double* x) {   // device: v_mov_b64_dpp + 3 fused broadcast-FMAs per value (sub-lane 0 of each leg)
    for (int i = 0; i < N; i++) {
      g_hex->slot[h] = x[i]; hex_barrier(h);
      const double* s = g_hex->slot;
      const double r = ((s[0] + s[4]) + s[8]) + s[12]; hex_barrier(h);
      x[i] = r;
    }
  }
  double sum16(double x) {
    g_hex->slot[h] = x; hex_barrier(h);
    const double* s = g_hex->slot;
    const double r = (quad_of(s, h) + quad_of(s, h ^ 8)) + (quad_of(s, h ^ 4) + quad_of(s, h ^ 12)); hex_barrier(h);
    return r;
  }
  double min16(double x) {
    g_hex->slot[h] = x; hex_barrier(h);
    double r = g_hex->slot[0];
    for (int k = 1; k < 16; k++) r = fmin(r, g_hex->slot[k]);
    hex_barrier(h);
    return r;
  }
  double max16(double x) {
    g_hex->slot[h] = x; hex_barrier(h);
    double r = g_hex->slot[0];
    for (int k = 1; k < 16; k++) r = fmax(r, g_hex->slot[k]);
    hex_barrier(h);
    return r;
  }
  bool any16(bool b) {
    g_hex->islot[h] = b; hex_barrier(h);
    bool r = false;
    for (int k = 0; k < 16; k++) r |= (g_hex->islot[k] != 0);
    hex_barrier(h);
    return r;
  }
  void argmin16(double& v, int& i) {
    g_hex->slot[h] = v; g_hex->islot[h] = i; hex_barrier(h);
    double bv = g_hex->slot[0]; int bi = g_hex->islot[0];
    for (int k = 1; k < 16; k++) {
      const double ov = g_hex->slot[k]; const int oi = g_hex->islot[k];
      if (ov < bv || (ov == bv && oi >= 0 && (bi < 0 || oi < bi))) { bv = ov; bi = oi; }
    }
    hex_barrier(h);
    v = bv; i = bi;
  }
  bool wave_all(bool b) { return b; }
  bool wave_any(bool b) { return getenv("WBC_ANY_TRUE") ? true : b; }
  int wave_max_int(int x) { return x; }
};

// warm start of the emulated rollout (host_hex_rollout): the per-lane seed bit of hex_gi<..., WARM> between the ticks of ONE robot
static int g_hex_warm = 0;
static bool g_hex_seed[16];

extern "C" int host_hex_batch(int kind, const double* flat215, const double* params12, const int* q_perm,
                              const int* act_perm, int n, int stride, const double* q, const double* v,
                              const double* tg, const unsigned char* mask, const double* mu,
                              const double* mass_scale, double* tau, double* met, int* status, int* iters) {
  static wbc::ModelC m;
  static wbc::ParamsC P0;
  static wbc::ParamsX P;
  if (wbc::model_from_flat(flat215, &m)) return -1;
  wbc::model_set_perms(&m, q_perm, act_perm);
  wbc::params_default(kind, &P0);
  if (params12) memcpy(&P0, params12, sizeof(double) * 12);
  wbc::params_derive(P0, &P);
  static HexCtx ctx;
  g_hex = &ctx;
  const size_t STK = 1 << 20;
  static std::vector<char> stacks(16 * STK);
  static wbc::ParkHost pk[16];
  struct Args { int kind, i, stride; const double *q, *v, *tg, *mu, *ms; const unsigned char* mask; double *tau, *met; int *status, *iters; };
  static Args A;
  A = Args{kind, 0, stride, q, v, tg, mu, mass_scale, mask, tau, met, status, iters};
  ctx.body = [](int h) {
    const Args& a = A;
    const int i = a.i;
    HexHost qo{h};
    auto in = [&](int r) -> double {
      if (r < 19) return a.q[(size_t)r * a.stride + i];
      if (r < 37) return a.v[(size_t)(r - 19) * a.stride + i];
      return a.tg[(size_t)(r - 37) * a.stride + i];
    };
    auto ot = [&](int k, double x) { a.tau[(size_t)k * a.stride + i] = x; };
    auto om = [&](int k, double x) {
      if (k >= 4) { if (g_vdot && (k >= 10 || h == 0)) g_vdot[(size_t)(k - 4) * a.stride + i] = x; return; }
      if (h == 0 && a.met) a.met[(size_t)k * a.stride + i] = x;
    };
    int it = 0, st;
    const double mui = a.mu ? a.mu[i] : P.mu, msi = a.ms ? a.ms[i] : 1.0;
    const bool tb = P.tau_max < 1e300;
#define HEX_RUN(K) (tb ? wbc::hex_tick<HexHost, K, true>(m, P, qo, in, a.mask[i] & 0xFu, mui, msi, pk[h], ot, om, &it) \
                       : wbc::hex_tick<HexHost, K, false>(m, P, qo, in, a.mask[i] & 0xFu, mui, msi, pk[h], ot, om, &it))
    if (!g_hex_warm) {
      if (a.kind == wbc::KIND_ID) st = HEX_RUN(wbc::KIND_ID);
      else if (a.kind == wbc::KIND_PC) st = HEX_RUN(wbc::KIND_PC);
      else if (a.kind == wbc::KIND_CLF) st = HEX_RUN(wbc::KIND_CLF);
      else st = HEX_RUN(wbc::KIND_MPTC);
    } else {
#ifndef HOST_TICK_HEX_ONLY
      // the rollout kernels' instantiation (WARM): this lane's friction row was active when the robot's previous tick ended
      bool sd = g_hex_seed[h];
#define HEX_RUN_W(K) (tb ? wbc::hex_tick<HexHost, K, true, true>(m, P, qo, in, a.mask[i] & 0xFu, mui, msi, pk[h], ot, om, &it, &sd) \
                         : wbc::hex_tick<HexHost, K, false, true>(m, P, qo, in, a.mask[i] & 0xFu, mui, msi, pk[h], ot, om, &it, &sd))
      if (a.kind == wbc::KIND_ID) st = HEX_RUN_W(wbc::KIND_ID);
      else if (a.kind == wbc::KIND_PC) st = HEX_RUN_W(wbc::KIND_PC);
      else if (a.kind == wbc::KIND_CLF) st = HEX_RUN_W(wbc::KIND_CLF);
      else st = HEX_RUN_W(wbc::KIND_MPTC);
#undef HEX_RUN_W
      g_hex_seed[h] = (st == wbc::ST_OK) && sd;
#else
      st = wbc::ST_SINGULAR;   // (not part of a HOST_TICK_HEX_ONLY build)
#endif
    }
#undef HEX_RUN
    if (h == 0) { if (a.status) a.status[i] = st; if (a.iters) a.iters[i] = it; }
  };
  for (int i = 0; i < n; i++) {
    A.i = i;
    g_gi_dump = g_gi_dump_base ? g_gi_dump_base + (size_t)i * 256 : nullptr;
    ctx.count = 0;
    for (int h = 0; h < 16; h++) {
      ctx.finished[h] = false;
      getcontext(&ctx.ctx[h]);
      ctx.ctx[h].uc_stack.ss_sp = stacks.data() + (size_t)h * STK;
      ctx.ctx[h].uc_stack.ss_size = STK;
      ctx.ctx[h].uc_link = &ctx.main;
      makecontext(&ctx.ctx[h], (void (*)())hex_entry, 1, h);
    }
    ctx.cur = 0;
    swapcontext(&ctx.main, &ctx.ctx[0]);
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Host instantiation of the hinted nearest-sample search of the persistent rollout (wbc_traj_dev.hpp).
#include "../quadruped_drake_amd/csrc/wbc_traj_dev.hpp"
// Closed loop on the host emulation (tests): `steps` ticks of  targets[step] -> 16-lane tick -> semi-implicit Euler (the arithmetic of
// wbc_integrate_kernel / wbc_hex_rollout_kernel) for n robots, one after the other; every robot sees the same target sequence tg_seq[steps][54] and
// masks[steps].  warm != 0: every tick after a robot's first starts its active set from the previous tick's (hex_gi<..., WARM>), as
// wbc_set_warm_start makes the device's rollout kernels do.  q, v: in/out; tau, met, status: the last tick's; iters_sum[n]: active-set trips per robot.
extern "C" int host_hex_rollout(int kind, const double* flat215, const double* params12, int n, int stride, int steps, double dt, int warm,
                                double* q, double* v, const double* tg_seq, const unsigned char* masks, const double* mu, const double* mass_scale,
                                double* tau, double* met, int* status, double* iters_sum, int* status_nonzero_ticks) {
  double* const sink = g_vdot;
  int bad_total = 0;
  for (int i = 0; i < n; i++) {
    double q1[19], v1[18], vd1[18], tau1[12], met1[4];
    for (int r = 0; r < 19; r++) q1[r] = q[(size_t)r * stride + i];
    for (int r = 0; r < 18; r++) v1[r] = v[(size_t)r * stride + i];
    for (int h = 0; h < 16; h++) g_hex_seed[h] = false;
    int st1 = 0;
    double its = 0.0;
    for (int s = 0; s < steps; s++) {
      int it1 = 0;
      for (int r = 0; r < 18; r++) vd1[r] = 0.0;
      g_vdot = vd1;
      g_hex_warm = warm;
      const int rc = host_hex_batch(kind, flat215, params12, nullptr, nullptr, 1, 1, q1, v1, tg_seq + (size_t)s * 54, masks + s, mu ? mu + i : nullptr,
                                    mass_scale ? mass_scale + i : nullptr, tau1, met1, &st1, &it1);
      g_hex_warm = 0;
      g_vdot = sink;
      if (rc) return rc;
      its += it1;
      bad_total += (st1 != 0);
      // v+ = v + dt vd ; quat+ = exp(dt/2 w+) (x) quat, renormalised ; p, joints advance with v+
      double vn[18];
      for (int r = 0; r < 18; r++) { vn[r] = v1[r] + dt * vd1[r]; v1[r] = vn[r]; }
      const double wn = sqrt(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]);
      const double ang = 0.5 * wn * dt;
      double dw = 1.0, dx = 0.0, dy = 0.0, dz = 0.0;
      if (wn > 0.0) { const double sc = sin(ang) / wn; dw = cos(ang); dx = sc * vn[0]; dy = sc * vn[1]; dz = sc * vn[2]; }
      const double w1 = q1[0], x1 = q1[1], y1 = q1[2], z1 = q1[3];
      const double qw = dw * w1 - dx * x1 - dy * y1 - dz * z1, qx = dw * x1 + dx * w1 + dy * z1 - dz * y1;
      const double qy = dw * y1 - dx * z1 + dy * w1 + dz * x1, qz = dw * z1 + dx * y1 - dy * x1 + dz * w1;
      const double inv = 1.0 / sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
      q1[0] = qw * inv; q1[1] = qx * inv; q1[2] = qy * inv; q1[3] = qz * inv;
      for (int r = 0; r < 3; r++) q1[4 + r] += dt * vn[3 + r];
      for (int r = 0; r < 12; r++) q1[7 + r] += dt * vn[6 + r];
    }
    for (int r = 0; r < 19; r++) q[(size_t)r * stride + i] = q1[r];
    for (int r = 0; r < 18; r++) v[(size_t)r * stride + i] = v1[r];
    for (int r = 0; r < 12; r++) tau[(size_t)r * stride + i] = tau1[r];
    if (met) for (int r = 0; r < 4; r++) met[(size_t)r * stride + i] = met1[r];
    if (status) status[i] = st1;
    if (iters_sum) iters_sum[i] = its;
  }
  if (status_nonzero_ticks) *status_nonzero_ticks = bad_total;
  return 0;
}

extern "C" int host_traj_index(const double* ts, int K, double wait_time, double t, int hint) {
  wbc::TrajDev T{K, wait_time, ts, nullptr, nullptr, nullptr, 0};
  return wbc::traj_index(T, t, hint);
}
