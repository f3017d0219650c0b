// tests/stub/fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY.  An in-process stand-in for librccl.so.1 with the ten entry points
// chz_comm.inc binds at run time (CHZ_RCCL_LIB points here in the CPU tier): the "ranks" of a communicator are THREADS of one
// process, each with its own emulated engine, and a collective is a rendezvous plus memcpy between their buffers.  With it the
// engine's real multi-GPU code -- communicator set-up, chz_spectrum_broadcast / _exchange_rows, chz_run_blocks_sharded with its
// two issuing threads and both hand-overs -- runs with world sizes 2 and more where there is no GPU (the emulated runtime executes
// stream work at call time, so "enqueue a collective on a stream" is "take part in it now": every rank calls them in the same order,
// as RCCL itself demands).  Never built into, linked with or loaded by the product.
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <rccl/rccl.h>

namespace {
struct World {
  int n = 0, joined = 0;
  std::mutex m; std::condition_variable cv;
  // one rendezvous slot per collective in flight (they are entered in the same order by every rank)
  long gen = 0; int arrived = 0;
  const void* root_ptr = nullptr;
  // point-to-point mailboxes: [src][dst]
  struct Box { const void* p = nullptr; size_t bytes = 0; bool full = false; };
  std::vector<Box> box;
};
std::mutex g_m;
std::map<std::string, World*> g_worlds;
size_t tsize(ncclDataType_t t) { return (t == ncclFloat64 || t == ncclInt64 || t == ncclUint64) ? 8 : (t == ncclFloat16) ? 2 : (t == ncclInt8 || t == ncclUint8) ? 1 : 4; }
struct Pending { bool send; const void* sp; void* rp; size_t bytes; int peer; };
thread_local std::vector<Pending> t_group; thread_local int t_depth = 0;
}
struct ncclComm { World* w; int rank; };

static ncclResult_t run_p2p(ncclComm* c, const std::vector<Pending>& ops) {
  World* w = c->w;
  std::unique_lock<std::mutex> lk(w->m);
  for (const Pending& o : ops) if (o.send) {                       // publish every send first, then serve the receives
    World::Box& b = w->box[(size_t)c->rank * w->n + o.peer];
    w->cv.wait(lk, [&] { return !b.full; });
    b.p = o.sp; b.bytes = o.bytes; b.full = true;
    w->cv.notify_all();
  }
  for (const Pending& o : ops) if (!o.send) {
    World::Box& b = w->box[(size_t)o.peer * w->n + c->rank];
    w->cv.wait(lk, [&] { return b.full; });
    if (b.bytes != o.bytes) return ncclInvalidArgument;
    memcpy(o.rp, b.p, o.bytes);
    b.full = false;
    w->cv.notify_all();
  }
  // a send is complete once its receiver has copied: wait for the mailboxes this rank filled
  for (const Pending& o : ops) if (o.send) {
    World::Box& b = w->box[(size_t)c->rank * w->n + o.peer];
    w->cv.wait(lk, [&] { return !b.full; });
  }
  return ncclSuccess;
}

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static uint64_t counter = 0;
  std::lock_guard<std::mutex> lk(g_m);
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "fake-rccl-%llu", (unsigned long long)++counter);
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  World* w;
  {
    std::lock_guard<std::mutex> lk(g_m);
    std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) { w = new World; w->n = nranks; w->box.resize((size_t)nranks * nranks); g_worlds[key] = w; }
    else w = it->second;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  std::unique_lock<std::mutex> lk(w->m);
  w->joined++;
  w->cv.notify_all();
  w->cv.wait(lk, [&] { return w->joined >= w->n; });             // like the real thing: returns when every rank has joined
  *comm = new ncclComm{w, rank};
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "fake rccl error"; }
__attribute__((visibility("default"))) ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, void*) {
  World* w = c->w;
  std::unique_lock<std::mutex> lk(w->m);
  const long my = w->gen;
  if (c->rank == root) w->root_ptr = send;
  w->arrived++;
  w->cv.notify_all();
  w->cv.wait(lk, [&] { return w->gen != my || w->arrived >= w->n; });      // everybody (the root included) is here
  if (w->gen == my) {
    if (c->rank != root) memcpy(recv, w->root_ptr, count * tsize(t));
    else if (recv != send) memcpy(recv, send, count * tsize(t));
    w->arrived++;                                                         // second phase: copies done
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->gen != my || w->arrived >= 2 * w->n; });
    if (w->gen == my && w->arrived >= 2 * w->n) { w->gen++; w->arrived = 0; w->root_ptr = nullptr; w->cv.notify_all(); }
  }
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, void*) {
  // control plane only (max of doubles): gather through broadcasts from every rank
  if (t != ncclDouble || op != ncclMax) return ncclInvalidArgument;
  std::vector<double> acc((const double*)send, (const double*)send + count), tmp(count);
  for (int r = 0; r < c->w->n; r++) {
    if (r == c->rank) memcpy(tmp.data(), send, count * 8);
    ncclResult_t e = ncclBroadcast(tmp.data(), tmp.data(), count, ncclDouble, r, c, nullptr);
    if (e != ncclSuccess) return e;
    for (size_t i = 0; i < count; i++) if (tmp[i] > acc[i]) acc[i] = tmp[i];
  }
  memcpy(recv, acc.data(), count * 8);
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void*) {
  t_group.push_back(Pending{true, p, nullptr, count * tsize(t), peer});
  if (t_depth == 0) { auto ops = t_group; t_group.clear(); return run_p2p(c, ops); }
  t_group.back().rp = (void*)c;
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* p, size_t count, ncclDataType_t t, int peer, ncclComm_t c, void*) {
  t_group.push_back(Pending{false, nullptr, p, count * tsize(t), peer});
  if (t_depth == 0) { auto ops = t_group; t_group.clear(); return run_p2p(c, ops); }
  t_group.back().sp = (const void*)c;
  return ncclSuccess;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() {
  if (--t_depth > 0) return ncclSuccess;
  if (t_group.empty()) return ncclSuccess;
  // the communicator rides in the unused pointer of each queued operation
  ncclComm* c = t_group[0].send ? (ncclComm*)t_group[0].rp : (ncclComm*)t_group[0].sp;
  std::vector<Pending> ops = t_group; t_group.clear();
  for (Pending& o : ops) { if (o.send) o.rp = nullptr; else o.sp = nullptr; }
  return run_p2p(c, ops);
}
}
