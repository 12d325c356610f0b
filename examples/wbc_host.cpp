// wbc_host.cpp -- the north-star host, literally: C++ calling HIP through the thin C ABI of include/wbc.h, one host thread per GPU,
// the robot-instance batch sharded contiguously over the GPUs of the node, and RCCL over xGMI used for ONE thing: gathering the
// end-of-rollout statistics.  No Python and no torch in this process.
//
//   examples/wbc_host --batch batch.bin [--gpus N] [--steps K] [--warmup W] [--ramp-seconds S] [--repeat R]
//
// What it replaces in the reference: the loop of simulate.py:182 (simulator.AdvanceTo) calling
// BasicController.DoSetControlTorques -> ControlLaw once per 5 ms step for ONE robot (controllers/basic_controller.py:286-320,
// controllers/mptc_controller.py:125-310), here for N robots per launch and G GPUs (SURVEY 8b "one host thread per GPU", 8e).
//
// batch.bin is a dump of quadruped_drake_amd.workloads.make_batch (workloads.dump_batch; layout below): the SAME seeded synthetic
// batch bench.py steps, so the two hosts can be compared launch for launch -- tests/test_host_cpp.py holds this binary's torque
// checksum, statistics and kernel time against `bench.py --gpus 1 --config 5 --per-gpu 4096` on the GPU box.
//
// Per GPU thread g of G:   hipSetDevice(g) -> wbc_create -> hipMalloc + upload of its contiguous shard -> clock ramp, W warm-up
// steps -> [barrier] K timed steps (wbc_time_steps: HIP events on the launch stream) -> wbc_stats_pack -> ONE ncclAllGather of
// WBC_NSTAT doubles -> wbc_stats_reduce [barrier].  Rank 0 prints ONE JSON line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/wbc.h"

namespace {

// ---- batch file (little endian): "WBCBATCH" | int32 version = 1, kind, n, has_mu | wbc_model (flat[215], q_perm[12], act_perm[12]) |
//      q[19][n] v[18][n] targets[54][n] (doubles, batch index fastest) | mask[n] (bytes, padded to 8) | [mu[n] mass_scale[n]]
struct Batch {
  int kind = 0, n = 0, has_mu = 0;
  wbc_model model;
  std::vector<double> q, v, tg, mu, ms;
  std::vector<uint8_t> mask;
};

bool read_exact(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }

bool load_batch(const char* path, Batch* b, std::string* err) {
  FILE* f = fopen(path, "rb");
  if (!f) { *err = std::string("cannot open ") + path; return false; }
  char magic[8];
  int32_t hdr[4];
  bool ok = read_exact(f, magic, 8) && memcmp(magic, "WBCBATCH", 8) == 0 && read_exact(f, hdr, sizeof hdr) && hdr[0] == 1;
  if (ok) {
    b->kind = hdr[1]; b->n = hdr[2]; b->has_mu = hdr[3];
    const size_t n = (size_t)b->n;
    b->q.resize(19 * n); b->v.resize(18 * n); b->tg.resize(54 * n); b->mask.resize((n + 7) / 8 * 8);
    ok = b->n > 0 && read_exact(f, b->model.flat, sizeof b->model.flat) && read_exact(f, b->model.q_perm, sizeof b->model.q_perm) &&
         read_exact(f, b->model.act_perm, sizeof b->model.act_perm) && read_exact(f, b->q.data(), 19 * n * 8) &&
         read_exact(f, b->v.data(), 18 * n * 8) && read_exact(f, b->tg.data(), 54 * n * 8) && read_exact(f, b->mask.data(), b->mask.size());
    if (ok && b->has_mu) {
      b->mu.resize(n); b->ms.resize(n);
      ok = read_exact(f, b->mu.data(), n * 8) && read_exact(f, b->ms.data(), n * 8);
    }
  }
  fclose(f);
  if (!ok) *err = std::string("bad batch file ") + path;
  return ok;
}

// contiguous split, the first n % world ranks one instance more (quadruped_drake_amd/stats.py: shard_range)
void shard_range(int n, int rank, int world, int* lo, int* hi) {
  const int base = n / world, rem = n % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

uint64_t fnv1a64(const void* p, size_t bytes) {
  const unsigned char* c = static_cast<const unsigned char*>(p);
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < bytes; i++) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}

class Barrier {   // the threads of this process meet here (C++17: no std::barrier)
 public:
  explicit Barrier(int n) : n_(n) {}
  void wait() {
    std::unique_lock<std::mutex> l(m_);
    const int gen = gen_;
    if (aborted_) return;
    if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
    else cv_.wait(l, [&] { return gen != gen_ || aborted_; });
  }
  void abort() {   // a rank that failed releases the others (they finish their own work and the process reports the failure)
    std::lock_guard<std::mutex> l(m_);
    aborted_ = true;
    cv_.notify_all();
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, count_ = 0, gen_ = 0;
  bool aborted_ = false;
};

struct Options {
  std::string batch;
  int gpus = 0, steps = 200, warmup = 20, repeat = 1;
  double ramp_seconds = 1.0;
};

struct RankResult {
  bool ok = false;
  std::string err;
  int n = 0;
  std::vector<float> kernel_ms;     // per repeat
  std::vector<double> wall_s;       // per repeat: barrier -> K steps -> statistics gathered -> barrier
  double gather_us = 0.0;           // the collective alone (last repeat)
  uint64_t tau_hash = 0;
  double tau_sum = 0.0;
  int status_nonzero = 0;
  wbc_stats reduced;                // identical on every rank (all-gather + the same fold)
  double packed[WBC_NSTAT];
  int num_vgpr = 0, scratch = 0, lds = 0;
};

// What the GPU threads share besides the barrier: the communicators (so that a rank that fails can release peers that are already inside, or about to
// enter, the collective: ncclCommAbort makes a pending ncclAllGather / stream wait return instead of waiting for a rank that will never arrive).
struct Team {
  Barrier bar;
  std::vector<ncclComm_t>* comms;
  std::atomic<bool> failed{false};
  std::once_flag abort_once;
  explicit Team(int n, std::vector<ncclComm_t>* c) : bar(n), comms(c) {}
  void fail() {
    failed.store(true);
    bar.abort();
    std::call_once(abort_once, [&] { for (ncclComm_t c : *comms) (void)ncclCommAbort(c); });
  }
};

// Everything a rank allocates, released on every way out of rank_main (the early returns of the error macros included).
struct RankResources {
  wbc_handle h = nullptr;
  hipStream_t cs = nullptr;
  std::vector<void*> bufs;
  template <class T> hipError_t alloc(T** p, size_t bytes) {
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
    if (e == hipSuccess) bufs.push_back(*p);
    return e;
  }
  ~RankResources() {
    if (cs) (void)hipStreamDestroy(cs);
    for (void* p : bufs) (void)hipFree(p);
    if (h) (void)wbc_destroy(h);
  }
};

#define HOST_FAIL(msg) do { res->err = (msg); team->fail(); return; } while (0)
#define HOST_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) HOST_FAIL(std::string(#x ": ") + hipGetErrorString(e_)); } while (0)
#define HOST_WBC(x) do { if ((x) != 0) HOST_FAIL(std::string(#x ": ") + wbc_last_error()); } while (0)
#define HOST_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) HOST_FAIL(std::string(#x ": ") + ncclGetErrorString(r_)); } while (0)
// a peer has failed: do not enter (or go on after) a collective it will never join
#define HOST_PEERS_OK() do { if (team->failed.load()) { res->err = "stopped: another rank failed"; return; } } while (0)

void rank_main(int rank, int world, const Options& o, const Batch& b, ncclComm_t comm, Team* team, RankResult* res) {
  Barrier* bar = &team->bar;
  int lo, hi;
  shard_range(b.n, rank, world, &lo, &hi);
  const int n = hi - lo, ldh = b.n;
  res->n = n;
  HOST_HIP(hipSetDevice(rank));
  RankResources R;
  wbc_handle& h = R.h;
  HOST_WBC(wbc_create(&b.model, b.kind, nullptr, n, rank, WBC_DEVICE_PTRS, &h));
  double *q, *v, *tg, *mu = nullptr, *ms = nullptr, *tau, *met, *d_send, *d_recv;
  uint8_t* mask;
  int32_t* status;
  const size_t nb = (size_t)n * 8;
  HOST_HIP(R.alloc(&q, 19 * nb)); HOST_HIP(R.alloc(&v, 18 * nb)); HOST_HIP(R.alloc(&tg, 54 * nb));
  HOST_HIP(R.alloc(&mask, n)); HOST_HIP(R.alloc(&tau, 12 * nb)); HOST_HIP(R.alloc(&met, 4 * nb)); HOST_HIP(R.alloc(&status, (size_t)n * 4));
  HOST_HIP(R.alloc(&d_send, WBC_NSTAT * 8)); HOST_HIP(R.alloc(&d_recv, (size_t)world * WBC_NSTAT * 8));
  // the shard's columns lo..hi-1 of every row: host leading dimension = whole batch, device leading dimension = shard
  HOST_HIP(hipMemcpy2D(q, nb, b.q.data() + lo, (size_t)ldh * 8, nb, 19, hipMemcpyHostToDevice));
  HOST_HIP(hipMemcpy2D(v, nb, b.v.data() + lo, (size_t)ldh * 8, nb, 18, hipMemcpyHostToDevice));
  HOST_HIP(hipMemcpy2D(tg, nb, b.tg.data() + lo, (size_t)ldh * 8, nb, 54, hipMemcpyHostToDevice));
  HOST_HIP(hipMemcpy(mask, b.mask.data() + lo, n, hipMemcpyHostToDevice));
  if (b.has_mu) {
    HOST_HIP(R.alloc(&mu, nb)); HOST_HIP(R.alloc(&ms, nb));
    HOST_HIP(hipMemcpy(mu, b.mu.data() + lo, nb, hipMemcpyHostToDevice));
    HOST_HIP(hipMemcpy(ms, b.ms.data() + lo, nb, hipMemcpyHostToDevice));
  }
  hipStream_t& cs = R.cs;   // the collective's stream
  HOST_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  float ms_step = 0.f;
  // clock ramp (the GPU raises its clock over the first second of sustained load) and warm-up, as bench.py does
  const auto t_ramp = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ramp).count() < o.ramp_seconds)
    HOST_WBC(wbc_time_steps(h, 100, n, n, q, v, tg, mask, mu, ms, tau, met, status, &ms_step));
  for (int w = 0; w < o.warmup; w++) HOST_WBC(wbc_step(h, n, n, q, v, tg, mask, mu, ms, tau, met, status));
  HOST_WBC(wbc_sync(h));
  std::vector<double> gathered((size_t)world * WBC_NSTAT);
  {   // warm the collective (RCCL channel set-up) outside the timed region; every rank has come through its set-up, or nobody enters
    HOST_WBC(wbc_stats_pack(h, res->packed));
    bar->wait();
    HOST_PEERS_OK();
    HOST_HIP(hipMemcpyAsync(d_send, res->packed, WBC_NSTAT * 8, hipMemcpyHostToDevice, cs));
    HOST_NCCL(ncclAllGather(d_send, d_recv, WBC_NSTAT, ncclDouble, comm, cs));
    HOST_HIP(hipStreamSynchronize(cs));
  }
  for (int rep = 0; rep < o.repeat; rep++) {
    HOST_WBC(wbc_stats_reset(h));
    HOST_WBC(wbc_sync(h));
    bar->wait();
    const auto t0 = std::chrono::steady_clock::now();
    HOST_WBC(wbc_time_steps(h, o.steps, n, n, q, v, tg, mask, mu, ms, tau, met, status, nullptr));   // K launches + two events, queued
    HOST_WBC(wbc_stats_pack(h, res->packed));                                                        // the one wait
    HOST_WBC(wbc_time_steps_result(h, &ms_step));
    HOST_PEERS_OK();
    const auto tg0 = std::chrono::steady_clock::now();
    HOST_HIP(hipMemcpyAsync(d_send, res->packed, WBC_NSTAT * 8, hipMemcpyHostToDevice, cs));
    HOST_NCCL(ncclAllGather(d_send, d_recv, WBC_NSTAT, ncclDouble, comm, cs));                       // RCCL over xGMI: 176 bytes per rank
    HOST_HIP(hipMemcpyAsync(gathered.data(), d_recv, gathered.size() * 8, hipMemcpyDeviceToHost, cs));
    HOST_HIP(hipStreamSynchronize(cs));
    HOST_PEERS_OK();   // (a collective released by a peer's ncclCommAbort has gathered nothing)
    const auto tg1 = std::chrono::steady_clock::now();
    HOST_WBC(wbc_stats_reduce(gathered.data(), world, &res->reduced));
    bar->wait();
    res->wall_s.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    res->kernel_ms.push_back(ms_step);
    res->gather_us = std::chrono::duration<double>(tg1 - tg0).count() * 1e6;
  }
  // outputs of the last launch: a checksum of the torques (bit-level identity with any other host of the same batch) and the statuses
  std::vector<double> tau_h((size_t)12 * n);
  std::vector<int32_t> st_h(n);
  HOST_HIP(hipMemcpy(tau_h.data(), tau, 12 * nb, hipMemcpyDeviceToHost));
  HOST_HIP(hipMemcpy(st_h.data(), status, (size_t)n * 4, hipMemcpyDeviceToHost));
  res->tau_hash = fnv1a64(tau_h.data(), tau_h.size() * 8);
  for (double x : tau_h) res->tau_sum += x;
  for (int32_t s : st_h) res->status_nonzero += (s != 0);
  int bt = 0;
  HOST_WBC(wbc_kernel_info(h, &res->num_vgpr, &res->scratch, &res->lds, &bt));
  res->ok = true;   // (stream, buffers and handle: ~RankResources)
}

void json_vec(std::string* s, const char* key, const std::vector<double>& x, const char* fmt) {
  char buf[64];
  *s += "\"" + std::string(key) + "\": [";
  for (size_t i = 0; i < x.size(); i++) { snprintf(buf, sizeof buf, fmt, x[i]); *s += (i ? ", " : "") + std::string(buf); }
  *s += "]";
}

double median(std::vector<double> x) {
  for (size_t i = 1; i < x.size(); i++) for (size_t j = i; j > 0 && x[j] < x[j - 1]; j--) std::swap(x[j], x[j - 1]);
  return x.empty() ? 0.0 : (x.size() % 2 ? x[x.size() / 2] : 0.5 * (x[x.size() / 2 - 1] + x[x.size() / 2]));
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
    if (a == "--batch") o.batch = next();
    else if (a == "--gpus") o.gpus = atoi(next());
    else if (a == "--steps") o.steps = atoi(next());
    else if (a == "--warmup") o.warmup = atoi(next());
    else if (a == "--repeat") o.repeat = atoi(next());
    else if (a == "--ramp-seconds") o.ramp_seconds = atof(next());
    else { fprintf(stderr, "usage: wbc_host --batch FILE [--gpus N] [--steps K] [--warmup W] [--ramp-seconds S] [--repeat R]\n"); return 2; }
  }
  if (o.batch.empty() || o.steps <= 0 || o.repeat <= 0) { fprintf(stderr, "wbc_host: --batch FILE is required, steps and repeat positive\n"); return 2; }
  Batch b;
  std::string err;
  if (!load_batch(o.batch.c_str(), &b, &err)) { fprintf(stderr, "wbc_host: %s\n", err.c_str()); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "wbc_host: no GPU visible (the hot path has no CPU fallback)\n"); return 1; }
  const int world = o.gpus > 0 ? o.gpus : ndev;
  if (world > ndev) { fprintf(stderr, "wbc_host: --gpus %d but %d visible\n", world, ndev); return 1; }
  if (b.n < world) { fprintf(stderr, "wbc_host: %d instances cannot be sharded over %d GPUs\n", b.n, world); return 1; }
  // one communicator per GPU, all owned by this process (ncclCommInitAll); each GPU thread drives its own
  std::vector<ncclComm_t> comms(world);
  std::vector<int> devs(world);
  for (int r = 0; r < world; r++) devs[r] = r;
  {
    ncclResult_t r = ncclCommInitAll(comms.data(), world, devs.data());
    if (r != ncclSuccess) { fprintf(stderr, "wbc_host: ncclCommInitAll: %s\n", ncclGetErrorString(r)); return 1; }
  }
  int rccl_version = 0;
  (void)ncclGetVersion(&rccl_version);
  Team team(world, &comms);
  std::vector<RankResult> res(world);
  std::vector<std::thread> th;
  for (int r = 0; r < world; r++) th.emplace_back(rank_main, r, world, std::cref(o), std::cref(b), comms[r], &team, &res[r]);
  for (auto& t : th) t.join();
  if (!team.failed.load())
    for (int r = 0; r < world; r++) (void)ncclCommDestroy(comms[r]);     // (aborted communicators are already gone)
  {
    bool bad = false;
    for (int r = 0; r < world; r++)
      if (!res[r].ok) { fprintf(stderr, "wbc_host: rank %d failed: %s\n", r, res[r].err.c_str()); bad = true; }
    if (bad) return 1;
  }
  // MAX over ranks of the timed region, per repeat; the reported figures are the medians over the repeats
  std::vector<double> wall(o.repeat, 0.0), kms_max(o.repeat, 0.0), per_rank_kms(world), per_rank_ticks(world);
  for (int rep = 0; rep < o.repeat; rep++)
    for (int r = 0; r < world; r++) {
      if (res[r].wall_s[rep] > wall[rep]) wall[rep] = res[r].wall_s[rep];
      if (res[r].kernel_ms[rep] > kms_max[rep]) kms_max[rep] = res[r].kernel_ms[rep];
    }
  for (int r = 0; r < world; r++) {
    std::vector<double> k(res[r].kernel_ms.begin(), res[r].kernel_ms.end());
    per_rank_kms[r] = median(k);
    per_rank_ticks[r] = res[r].packed[0];
  }
  const double wall_med = median(wall), kms = median(kms_max);
  const wbc_stats& st = res[0].reduced;
  int bad = 0;
  for (int r = 0; r < world; r++) bad += res[r].status_nonzero;
  std::string s = "{";
  char buf[512];
  snprintf(buf, sizeof buf, "\"host\": \"examples/wbc_host.cpp: C++ over include/wbc.h, one std::thread per GPU, ncclCommInitAll + one ncclAllGather of %d doubles\", ", WBC_NSTAT);
  s += buf;
  snprintf(buf, sizeof buf, "\"metric\": \"whole-body-QP control ticks/s\", \"value\": %.6e, \"value_kernel_only\": %.6e, \"unit\": \"ticks/s\", \"n_gpus\": %d, \"ranks_seen\": %d, "
           "\"instances\": %d, \"kind\": %d, \"domain_randomised\": %s, \"steps\": %d, \"warmup\": %d, \"repeat\": %d, \"ms_per_step\": %.6f, \"kernel_ms\": %.6f, ",
           (double)b.n * o.steps / wall_med, (double)b.n / (kms * 1e-3), world, world, b.n, b.kind, b.has_mu ? "true" : "false", o.steps, o.warmup, o.repeat,
           wall_med / o.steps * 1e3, kms);
  s += buf;
  json_vec(&s, "per_rank_kernel_ms", per_rank_kms, "%.6f"); s += ", ";
  json_vec(&s, "per_rank_ticks", per_rank_ticks, "%.0f"); s += ", ";
  s += "\"per_rank_tau_fnv1a64\": [";
  for (int r = 0; r < world; r++) { snprintf(buf, sizeof buf, "%s\"%016llx\"", r ? ", " : "", (unsigned long long)res[r].tau_hash); s += buf; }
  s += "], ";
  {
    std::vector<double> ts(world);
    for (int r = 0; r < world; r++) ts[r] = res[r].tau_sum;
    json_vec(&s, "per_rank_tau_sum", ts, "%.17g"); s += ", ";
  }
  snprintf(buf, sizeof buf, "\"status_nonzero\": %d, \"rollout_stats\": {\"ticks\": %.0f, \"status_nonzero\": %.0f, \"iters_sum\": %.0f, \"tau_abs_sum\": %.17g, "
           "\"tau_abs_max\": %.17g, \"err_sum\": %.17g, \"mask_count\": [", bad, st.ticks, st.status_nonzero, st.iters_sum, st.tau_abs_sum, st.tau_abs_max, st.err_sum);
  s += buf;
  for (int k = 0; k < 16; k++) { snprintf(buf, sizeof buf, "%s%.0f", k ? ", " : "", st.mask_count[k]); s += buf; }
  snprintf(buf, sizeof buf, "]}, \"rccl\": {\"version\": %d, \"allgather_us\": %.1f, \"bytes_per_rank\": %d}, \"kernel_info\": {\"num_regs\": %d, \"scratch_bytes_per_lane\": %d, "
           "\"lds_bytes\": %d}, \"wbc_version\": %d}", rccl_version, res[0].gather_us, WBC_NSTAT * 8, res[0].num_vgpr, res[0].scratch, res[0].lds, wbc_version());
  s += buf;
  printf("%s\n", s.c_str());
  return 0;
}
