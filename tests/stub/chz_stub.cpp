// tests/stub/chz_stub.cpp -- TEST INFRASTRUCTURE ONLY.  A CPU stand-in for libchz_hip.so behind the SAME C ABI
// (include/chz_engine.h), for ONE purpose: running the filter.h drop-in's host code (ka9q-radio_amd/csrc/filter_hip.c: job
// numbering, completion signalling, the miss queue with its leader / followers, bank growth, staging buffers, 1000 channel
// pthreads) under ThreadSanitizer / AddressSanitizer where there is no GPU.  The arithmetic is the oracle's (oracle/chz_oracle.c);
// every engine entry point the drop-in calls is an operation on ONE in-order work queue served by a worker thread, so that, as on the
// device, calls return before their work is done and the completion callback runs on a thread that is not the caller's.
// Never built into, linked with or loaded by the product (tests/test_dropin_stub.py builds it into tests/stub/_build/).
#include <atomic>
#include <chrono>
#include <complex>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/chz_engine.h"
#include "../../oracle/chz_oracle.h"

static thread_local char g_err[256];
static int fail(int rc, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
  return rc;
}

struct Bank {
  int P = 0, olen = 0, cap = 0, real = 0, active = 0; bool alive = false;
  std::vector<float> resp;                 // [cap][2P]
  std::vector<int> shift; std::vector<unsigned char> isb, beam_on; std::vector<double> ab;   // [cap], [cap], [cap], [cap][4]
  std::vector<float> out[CHZ_ND];          // [cap][olen * (real ? 1 : 2)]
  double noise_samprate = 0.0; std::vector<double> n0[CHZ_ND];   // estimate_noise() per channel (chz_bank_enable_noise)
  size_t per() const { return (size_t)olen * (real ? 1 : 2); }
};

struct chz_engine {
  int L = 0, M = 0, N = 0, in_type = 0, bins = 0;
  chzo_stream* stream = nullptr;
  std::vector<float> spec[CHZ_ND];
  std::vector<float> pending;              // samples written and not yet transformed (worker only)
  unsigned next_job = 0;                   // the block the stream's history is ready for (worker only)
  std::vector<int> notch_bins; std::vector<double> notch_alpha, notch_state;
  enum { MAX_BANKS = 256 };
  Bank banks[MAX_BANKS];                   // contents touched by the worker only; a new one is published through nbanks
  std::atomic<int> nbanks{0};
  std::thread worker; std::mutex m; std::condition_variable cv, idle;
  std::atomic<bool> failed{false}; int fail_job = -1;     // fault injection (CHZ_STUB_FAIL_JOB)
  std::mutex mm; std::condition_variable mcv; unsigned long long mark_seq = 0, mark_want[8] = {}, mark_done[8] = {};   // chz_input_mark
  std::deque<std::function<void()>> q; bool busy = false, quit = false;
  void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); } cv.notify_one(); }
  void drain() { std::unique_lock<std::mutex> lk(m); idle.wait(lk, [&] { return q.empty() && !busy; }); }
  void loop() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return quit || !q.empty(); });
        if (q.empty()) return;
        f = std::move(q.front()); q.pop_front(); busy = true; }
      f();
      { std::lock_guard<std::mutex> lk(m); busy = false; if (q.empty()) idle.notify_all(); }
    }
  }
};

static std::atomic<int> g_instances{0};     // engines created by this process so far (fault injection below)

extern "C" {

const char* chz_last_error(void) { return g_err; }
int chz_process_exiting(void) { return 0; }
int chz_set_option(const char*, const char*) { return 0; }      /* the stand-in has no dispatch to steer */
// CHZ_STUB_DEVICES=n: the stand-in reports n devices (the drop-in's KA9Q_HIP_DEVICES sharding runs over n independent engines)
int chz_device_count(void) { const char* v = getenv("CHZ_STUB_DEVICES"); const int n = v ? atoi(v) : 1; return n > 0 ? n : 1; }

int chz_engine_create(chz_engine** out, int L, int M, int in_type, int device, const char*, int) {
  if (!out || L < 1 || M < 1 || (in_type != CHZ_REAL && in_type != CHZ_COMPLEX)) return fail(-1, "bad argument");
  if (device < 0 || device >= chz_device_count()) return fail(-2, "device %d out of range (%d devices)", device, chz_device_count());
  chz_engine* e = new chz_engine;
  e->L = L; e->M = M; e->N = L + M - 1; e->in_type = in_type;
  e->stream = chzo_stream_create(L, M, in_type);
  if (!e->stream) { delete e; return fail(-3, "no stream for L=%d M=%d", L, M); }
  e->bins = chzo_stream_bins(e->stream);
  for (auto& s : e->spec) s.assign((size_t)2 * e->bins, 0.f);
  e->worker = std::thread([e] { e->loop(); });
  { const int inst = g_instances.fetch_add(1);
    const char* fj = getenv("CHZ_STUB_FAIL_JOB"); const char* fa = getenv("CHZ_STUB_FAIL_ALWAYS");
    if (fj && (inst == 0 || (fa && fa[0] == '1'))) e->fail_job = atoi(fj) + (inst == 0 ? 0 : 4 * inst); }
  *out = e;
  return 0;
}
void chz_engine_destroy(chz_engine* e) {
  if (!e) return;
  e->drain();
  { std::lock_guard<std::mutex> lk(e->m); e->quit = true; }
  e->cv.notify_all();
  e->worker.join();
  chzo_stream_delete(e->stream);
  delete e;
}
int chz_sync(chz_engine* e) { if (!e) return fail(-1, "null engine"); e->drain(); return 0; }
int chz_slot_sync(chz_engine* e, int) { return chz_sync(e); }
// fault injection for the drop-in's recovery path: env CHZ_STUB_FAIL_JOB=n makes the FIRST engine of the process (every engine
// with CHZ_STUB_FAIL_ALWAYS=1) report a failed device-side check once it has been handed block n, as a notch ticket that ran out does
int chz_engine_check(const chz_engine* e) {
  if (!e) return fail(-1, "null engine");
  if (e->failed.load(std::memory_order_acquire)) return fail(-8, "chz_stub: injected device-side failure");
  return 0;
}
int chz_input_seek(chz_engine* e, unsigned job, const float* history) {
  if (!e) return fail(-1, "null engine");
  e->drain();
  const size_t per = e->in_type == CHZ_REAL ? 1 : 2;
  chzo_stream_delete(e->stream);
  e->stream = chzo_stream_create(e->L, e->M, e->in_type);
  e->pending.clear(); e->next_job = job;
  if (history) {                       // run the history through the stream's overlap: blocks of zeros ending with the M-1 samples
    const long nh = e->M - 1, nblk = (nh + e->L - 1) / e->L;
    std::vector<float> buf((size_t)nblk * e->L * per, 0.f), scratch((size_t)2 * e->bins);
    memcpy(buf.data() + ((size_t)nblk * e->L - (size_t)nh) * per, history, sizeof(float) * (size_t)nh * per);
    for (long b = 0; b < nblk; b++) chzo_stream_push(e->stream, buf.data() + (size_t)b * e->L * per, scratch.data());
  }
  return 0;
}
int chz_input_mark(chz_engine* e, int k) {
  if (!e || k < 0 || k >= 8) return fail(-1, "bad argument");
  const unsigned long long t = ++e->mark_seq;
  e->mark_want[k] = t;
  e->post([e, k, t] { { std::lock_guard<std::mutex> lk(e->mm); e->mark_done[k] = t; } e->mcv.notify_all(); });
  return 0;
}
int chz_input_mark_wait(chz_engine* e, int k) {
  if (!e || k < 0 || k >= 8) return fail(-1, "bad argument");
  std::unique_lock<std::mutex> lk(e->mm);
  e->mcv.wait(lk, [&] { return e->mark_done[k] >= e->mark_want[k]; });
  return 0;
}
int chz_engine_notch_order(chz_engine* e, int) { return e ? 0 : fail(-1, "null engine"); }

int chz_input_write(chz_engine* e, const float* x, long n) {
  if (!e || !x || n < 0) return fail(-1, "bad argument");
  std::vector<float> copy(x, x + (size_t)n * (e->in_type == CHZ_REAL ? 1 : 2));
  e->post([e, copy = std::move(copy)] { e->pending.insert(e->pending.end(), copy.begin(), copy.end()); });
  return 0;
}
int chz_forward(chz_engine* e, unsigned job) {
  if (!e) return fail(-1, "null engine");
  if (e->fail_job >= 0 && (int)job >= e->fail_job) e->post([e] { e->failed.store(true, std::memory_order_release); });
  e->post([e, job] {
    // CHZ_STUB_WEDGE_JOB=n: from block n on the device neither completes anything nor reports anything (a wedged device)
    static const int wedge = [] { const char* v = getenv("CHZ_STUB_WEDGE_JOB"); return v ? atoi(v) : -1; }();
    if (wedge >= 0 && (int)job >= wedge) for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
    static const int delay_ms = [] { const char* v = getenv("CHZ_STUB_FORWARD_DELAY_MS"); return v ? atoi(v) : 0; }();   // a slow device
    if (delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    const size_t need = (size_t)e->L * (e->in_type == CHZ_REAL ? 1 : 2);
    // blocks the caller wrote but never transformed (the drop-in's drop mode): their samples are history for this one, as in
    // the device ring, where block `job` reads the window that ends with ITS samples
    for (; e->next_job != job; e->next_job++) {
      if (e->pending.size() < 2 * need) { fprintf(stderr, "chz_stub: forward(%u) without its input\n", job); abort(); }
      std::vector<float> scratch((size_t)2 * e->bins);
      chzo_stream_push(e->stream, e->pending.data(), scratch.data());
      e->pending.erase(e->pending.begin(), e->pending.begin() + (long)need);
    }
    e->next_job = job + 1;
    if (e->pending.size() < need) { fprintf(stderr, "chz_stub: forward without a block of input\n"); abort(); }
    float* sp = e->spec[job % CHZ_ND].data();
    chzo_stream_push(e->stream, e->pending.data(), sp);
    e->pending.erase(e->pending.begin(), e->pending.begin() + (long)need);
    for (size_t i = 0; i < e->notch_bins.size(); i++) {            // apply_notch_filters, src/filter.c:464-474
      const int b = e->notch_bins[i];
      double& sr = e->notch_state[2 * i]; double& si = e->notch_state[2 * i + 1];
      sr += e->notch_alpha[i] * ((double)sp[2 * b] - sr); si += e->notch_alpha[i] * ((double)sp[2 * b + 1] - si);
      sp[2 * b] = (float)((double)sp[2 * b] - sr); sp[2 * b + 1] = (float)((double)sp[2 * b + 1] - si);
    }
  });
  return 0;
}
int chz_set_notches_alpha(chz_engine* e, const int* bins, const double* alpha, int n) {
  if (!e) return fail(-1, "null engine");
  for (int i = 0; i < n; i++) if (bins[i] < 0 || bins[i] >= e->bins) return fail(-1, "notch bin %d out of range", bins[i]);
  e->drain();
  e->notch_bins.assign(bins, bins + (n > 0 ? n : 0)); e->notch_alpha.assign(alpha, alpha + (n > 0 ? n : 0));
  e->notch_state.assign((size_t)2 * (n > 0 ? n : 0), 0.0);
  return 0;
}
int chz_spectrum_read_async(chz_engine* e, int slot, float* host) {
  if (!e || !host || slot < 0 || slot >= CHZ_ND) return fail(-1, "bad argument");
  e->post([e, slot, host] { memcpy(host, e->spec[slot].data(), sizeof(float) * 2 * (size_t)e->bins); });
  return 0;
}
int chz_host_callback(chz_engine* e, int, void (*fn)(void*), void* arg) {
  if (!e || !fn) return fail(-1, "bad argument");
  // CHZ_STUB_LOSE_CALLBACKS=1: once the engine has failed its stream callbacks are never delivered -- what the runtime does after a sticky device error
  static const bool lose = [] { const char* v = getenv("CHZ_STUB_LOSE_CALLBACKS"); return v && v[0] == '1'; }();
  e->post([e, fn, arg] { if (lose && e->failed.load(std::memory_order_acquire)) return; fn(arg); });
  return 0;
}
// the in-process clique of the drop-in's KA9Q_HIP_EXCHANGE=broadcast: a communicator is just its rank here; the broadcast is a task on
// the root's queue (behind its forward transform) that hands the slot to a task on every other engine's queue (in front of its banks)
struct chz_comm { int rank, world, device; };
// fault injection for the drop-in's exchange ladder: CHZ_STUB_FAIL_COMM=1 -- no clique can be formed; CHZ_STUB_FAIL_BCAST_CALL=n -- the n-th
// broadcast of the process reports an error (as an RCCL call that returns ncclSystemError would)
static std::atomic<int> g_bcast_calls{0};
int chz_comm_create_local(chz_comm** out, int n, const int* devices) {
  if (!out || !devices || n < 1) return fail(-1, "bad argument");
  if (getenv("CHZ_STUB_FAIL_COMM")) return fail(-6, "chz_stub: injected communicator failure (ncclCommInitAll timed out)");
  for (int i = 0; i < n; i++) {
    if (devices[i] < 0 || devices[i] >= chz_device_count()) return fail(-2, "device %d is not visible", devices[i]);
    for (int k = 0; k < i; k++) if (devices[k] == devices[i]) return fail(-2, "device %d is listed twice: an RCCL clique needs one communicator per device", devices[i]);
  }
  for (int i = 0; i < n; i++) out[i] = new chz_comm{i, n, devices[i]};
  return 0;
}
void chz_comm_destroy(chz_comm* c) { delete c; }
int chz_spectrum_broadcast_local(chz_engine* const* engines, chz_comm* const* comms, int n, int slot, int root) {
  if (!engines || !comms || n < 1 || slot < 0 || slot >= CHZ_ND || root < 0 || root >= n) return fail(-1, "bad argument");
  for (int i = 0; i < n; i++) if (!engines[i] || !comms[i] || comms[i]->rank != i || comms[i]->world != n) return fail(-1, "not this clique");
  if (const char* fb = getenv("CHZ_STUB_FAIL_BCAST_CALL")) if (g_bcast_calls.fetch_add(1) + 1 == atoi(fb)) return fail(-6, "chz_stub: injected broadcast failure (ncclSystemError)");
  struct Box { std::mutex m; std::condition_variable cv; bool ready = false; std::vector<float> data; };
  auto box = std::make_shared<Box>();
  chz_engine* r = engines[root];
  r->post([r, slot, box] { { std::lock_guard<std::mutex> lk(box->m); box->data = r->spec[slot]; box->ready = true; } box->cv.notify_all(); });
  for (int i = 0; i < n; i++) {
    if (i == root) continue;
    chz_engine* e = engines[i];
    e->post([e, slot, box] { std::unique_lock<std::mutex> lk(box->m); box->cv.wait(lk, [&] { return box->ready; }); e->spec[slot] = box->data; });
  }
  return 0;
}

int chz_host_alloc(void** p, size_t bytes) { return posix_memalign(p, 64, bytes ? bytes : 64) == 0 ? 0 : fail(-2, "out of memory"); }
void chz_host_free(void* p) { free(p); }
int chz_host_register(void*, size_t) { return 0; }
void chz_host_unregister(void*) {}

static int bank_create(chz_engine* e, int P, int olen, int cap, int real) {
  if (!e || cap < 1 || olen < 1 || olen > P) return fail(-1, "bad bank geometry");
  if ((long long)olen * e->N % e->L != 0 || (long long)olen * e->N / e->L != P) return fail(-1, "P=%d is not olen*N/L", P);
  if (real && (P & 1)) return fail(-3, "real-output channels need an even P");
  static std::mutex create_m;
  std::lock_guard<std::mutex> lk(create_m);
  const int id = e->nbanks.load(std::memory_order_relaxed);
  if (id >= chz_engine::MAX_BANKS) return fail(-2, "the stub holds %d banks", (int)chz_engine::MAX_BANKS);
  Bank& b = e->banks[id]; b.P = P; b.olen = olen; b.cap = cap; b.real = real; b.alive = true;
  b.resp.assign((size_t)cap * 2 * P, 0.f); b.shift.assign((size_t)cap, 0); b.isb.assign((size_t)cap, 0);
  b.beam_on.assign((size_t)cap, 0); b.ab.assign((size_t)cap * 4, 0.0);
  for (auto& o : b.out) o.assign((size_t)cap * b.per(), 0.f);
  e->nbanks.store(id + 1, std::memory_order_release);
  return id;
}
int chz_bank_create(chz_engine* e, int P, int olen, int cap) { return bank_create(e, P, olen, cap, 0); }
int chz_bank_create_real(chz_engine* e, int P, int olen, int cap) { return bank_create(e, P, olen, cap, 1); }
#define BANK(e, id, ch0, n) \
  if (!(e) || (id) < 0 || (id) >= (e)->nbanks.load(std::memory_order_acquire)) return fail(-1, "bad bank"); \
  if ((ch0) < 0 || (n) < 0 || (ch0) + (n) > (e)->banks[(size_t)(id)].cap) return fail(-1, "channel range out of bank capacity")
int chz_bank_destroy(chz_engine* e, int id) {
  BANK(e, id, 0, 0);
  e->post([e, id] { Bank& b = e->banks[(size_t)id]; b.alive = false; b.resp.clear(); b.resp.shrink_to_fit(); for (auto& o : b.out) { o.clear(); o.shrink_to_fit(); } });
  return 0;
}
int chz_bank_set_responses(chz_engine* e, int id, int ch0, int n, const float* resp) {
  BANK(e, id, ch0, n);
  const int P = e->banks[(size_t)id].P;
  std::vector<float> copy(resp, resp + (size_t)n * 2 * P);
  e->post([e, id, ch0, P, copy = std::move(copy)] { memcpy(e->banks[(size_t)id].resp.data() + (size_t)ch0 * 2 * P, copy.data(), sizeof(float) * copy.size()); });
  return 0;
}
int chz_bank_set_shifts(chz_engine* e, int id, int ch0, int n, const int* shifts) {
  BANK(e, id, ch0, n);
  std::vector<int> copy(shifts, shifts + n);
  e->post([e, id, ch0, copy = std::move(copy)] { std::copy(copy.begin(), copy.end(), e->banks[(size_t)id].shift.begin() + ch0); });
  return 0;
}
int chz_bank_set_isb(chz_engine* e, int id, int ch0, int n, const unsigned char* flags) {
  BANK(e, id, ch0, n);
  std::vector<unsigned char> copy(flags, flags + n);
  e->post([e, id, ch0, copy = std::move(copy)] { std::copy(copy.begin(), copy.end(), e->banks[(size_t)id].isb.begin() + ch0); });
  return 0;
}
int chz_bank_set_beam(chz_engine* e, int id, int ch0, int n, const double* ab, const unsigned char* on) {
  BANK(e, id, ch0, n);
  std::vector<double> a(ab, ab + (size_t)4 * n); std::vector<unsigned char> o(on, on + n);
  e->post([e, id, ch0, a = std::move(a), o = std::move(o)] {
    Bank& b = e->banks[(size_t)id];
    std::copy(a.begin(), a.end(), b.ab.begin() + (size_t)4 * ch0); std::copy(o.begin(), o.end(), b.beam_on.begin() + ch0);
  });
  return 0;
}
int chz_bank_set_active(chz_engine* e, int id, int n) {
  BANK(e, id, 0, n);
  e->post([e, id, n] { e->banks[(size_t)id].active = n; });
  return 0;
}
static void run_channels(chz_engine* e, int id, int slot, int ch0, int n) {
  Bank& b = e->banks[(size_t)id];
  if (!b.alive) return;
  for (int c = ch0; c < ch0 + n; c++) {
    float* o = b.out[slot].data() + (size_t)c * b.per();
    const float* r = b.resp.data() + (size_t)c * 2 * b.P;
    if (b.beam_on[(size_t)c])
      chzo_channel_beam(e->spec[slot].data(), e->bins, b.P, b.olen, b.shift[(size_t)c], r, b.ab[4 * c], b.ab[4 * c + 1], b.ab[4 * c + 2], b.ab[4 * c + 3], o);
    else
      chzo_channel(e->spec[slot].data(), e->bins, e->in_type, b.P, b.olen, b.real ? CHZO_REAL : CHZO_COMPLEX, b.shift[(size_t)c], b.isb[(size_t)c], r, o);
    if (b.noise_samprate > 0.0)
      b.n0[slot][(size_t)c] = chzo_estimate_noise(e->spec[slot].data(), e->bins, e->in_type, b.real ? b.P / 2 + 1 : b.P, b.shift[(size_t)c], b.noise_samprate);
  }
}
int chz_bank_execute(chz_engine* e, int id, unsigned job) {
  BANK(e, id, 0, 0);
  e->post([e, id, job] { run_channels(e, id, (int)(job % CHZ_ND), 0, e->banks[(size_t)id].active); });
  return 0;
}
int chz_bank_execute_range(chz_engine* e, int id, unsigned job, int ch0, int n) {
  BANK(e, id, ch0, n);
  e->post([e, id, job, ch0, n] { run_channels(e, id, (int)(job % CHZ_ND), ch0, n); });
  return 0;
}
int chz_bank_read_async(chz_engine* e, int id, int slot, int ch0, int n, float* host) {
  BANK(e, id, ch0, n);
  if (slot < 0 || slot >= CHZ_ND || !host) return fail(-1, "bad argument");
  e->post([e, id, slot, ch0, n, host] {
    Bank& b = e->banks[(size_t)id];
    if (b.alive) memcpy(host, b.out[slot].data() + (size_t)ch0 * b.per(), sizeof(float) * (size_t)n * b.per());
  });
  return 0;
}

int chz_bank_enable_noise(chz_engine* e, int id, double samprate) {
  BANK(e, id, 0, 0);
  if (!(samprate >= 0.0)) return fail(-1, "bad sample rate");
  e->post([e, id, samprate] { Bank& b = e->banks[(size_t)id]; b.noise_samprate = samprate; for (auto& v : b.n0) v.assign((size_t)b.cap, 0.0); });
  return 0;
}
int chz_bank_read_noise_async(chz_engine* e, int id, int slot, int ch0, int n, double* host) {
  BANK(e, id, ch0, n);
  if (slot < 0 || slot >= CHZ_ND || !host) return fail(-1, "bad argument");
  e->post([e, id, slot, ch0, n, host] {
    Bank& b = e->banks[(size_t)id];
    if (b.alive && !b.n0[slot].empty()) memcpy(host, b.n0[slot].data() + ch0, sizeof(double) * (size_t)n);
  });
  return 0;
}

// ---- pooled inline masters (filter2) ----
struct chz_mini { int L, M, N, cap; std::vector<float> resp; std::vector<unsigned char> used; std::mutex m; };
int chz_mini_create(chz_mini** out, int L, int M, int capacity, int) {
  if (!out || L < 1 || M < 1 || capacity < 1) return fail(-1, "bad argument");
  chz_mini* p = new chz_mini; p->L = L; p->M = M; p->N = L + M - 1; p->cap = capacity;
  p->resp.assign((size_t)capacity * 2 * p->N, 0.f); p->used.assign((size_t)capacity, 0);
  *out = p; return 0;
}
void chz_mini_destroy(chz_mini* m) { delete m; }
int chz_mini_capacity(const chz_mini* m) { return m ? m->cap : 0; }
int chz_mini_add(chz_mini* m) {
  if (!m) return fail(-1, "null pool");
  std::lock_guard<std::mutex> lk(m->m);
  for (int i = 0; i < m->cap; i++) if (!m->used[(size_t)i]) { m->used[(size_t)i] = 1; return i; }
  return fail(-7, "pool full");
}
int chz_mini_release(chz_mini* m, int inst) {
  if (!m || inst < 0 || inst >= m->cap) return fail(-1, "bad instance");
  std::lock_guard<std::mutex> lk(m->m); m->used[(size_t)inst] = 0; return 0;
}
int chz_mini_set_response(chz_mini* m, int inst, const float* resp) {
  if (!m || inst < 0 || inst >= m->cap || !resp) return fail(-1, "bad argument");
  std::lock_guard<std::mutex> lk(m->m);
  memcpy(m->resp.data() + (size_t)inst * 2 * m->N, resp, sizeof(float) * 2 * (size_t)m->N); return 0;
}
int chz_mini_execute(chz_mini* m, int n, const int* inst, const float* const* win, const int* shift, const unsigned char* isb, float* const* out) {
  if (!m || n < 0) return fail(-1, "bad argument");
  std::vector<float> spec((size_t)2 * m->N), resp((size_t)2 * m->N);
  for (int i = 0; i < n; i++) {
    { std::lock_guard<std::mutex> lk(m->m); memcpy(resp.data(), m->resp.data() + (size_t)inst[i] * 2 * m->N, sizeof(float) * resp.size()); }
    chzo_forward(win[i], m->N, CHZO_COMPLEX, spec.data());
    chzo_channel(spec.data(), m->N, CHZO_COMPLEX, m->N, m->L, CHZO_COMPLEX, shift ? shift[i] : 0, isb ? isb[i] : 0, resp.data(), out[i]);
  }
  return 0;
}

}  // extern "C"
