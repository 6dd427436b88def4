// RCCL behind the C-ABI: the gradient all-reduce of the data-parallel step (reference role: torch DistributedDataParallel's NCCL
// bucket all-reduce, TRAIN:87; SURVEY §8b `vacnic_comm_init` / `vacnic_allreduce_bucket`).
//
// Why not torch.distributed for the data plane.  The step runs on FOUR HIP streams (compute, weight gradients, frozen towers,
// small-token branches) and a process gets four hardware queues.  torch's ProcessGroupNCCL owns an internal stream per device, a
// reducer of one's own adds another: with a fifth / sixth stream alive HIP multiplexes streams onto queues and the step's own
// streams serialise — measured with a ONE-rank communicator, i.e. no bytes on any link: 64.4 -> 69.0 ms per step, and a process
// group that merely ran one broadcast costs the same (profiles/r4_ddp_one_rank_rccl.txt).  Here the collective is an ordinary
// stream-ordered launch on a stream the CALLER names (the weight-gradient stream, where a bucket's writers run), so the process
// keeps its four streams, and — being a C-ABI call like any kernel — a bucket's all-reduce is RECORDED into a launch plan and
// replayed from C++ at its place in the backward pass: no host round trip per bucket.
//
// librccl is resolved with dlopen at vacnic_comm_load (the library the process already uses: torch ships one), never at link
// time: a box without RCCL still loads libvacnic_hip.so and only the comm entry points report VACNIC_UNSUPPORTED.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                // ncclComm_t
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init_rank)(Comm*, int, UniqueId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_bcast)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_destroy)(Comm);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* h = nullptr;
  fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_allreduce allreduce = nullptr; fn_bcast bcast = nullptr;
  fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;
std::vector<Comm> g_comms;
constexpr int NEV = 512;
hipEvent_t g_ev[NEV];
bool g_ev_made[NEV];

constexpr int kSum = 0, kFloat32 = 7, kBfloat16 = 9;     // ncclRedOp_t / ncclDataType_t values of rccl.h

int check_rccl(int r, const char* what) {
  if (r == 0) return VACNIC_OK;
  vacnic_set_error("%s: RCCL error %d (%s)", what, r, g_rccl.errstr ? g_rccl.errstr(r) : "?");
  return VACNIC_HIP_ERROR;
}
Comm get(int64_t h) { return (h >= 0 && h < (int64_t)g_comms.size()) ? g_comms[(size_t)h] : nullptr; }

}  // namespace

extern "C" int vacnic_comm_load(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.h) return VACNIC_OK;
  const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* c : cands) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  VCHECK(h, VACNIC_UNSUPPORTED, "comm_load: librccl not found (%s)", dlerror());
  Rccl r;
  r.h = h;
  r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  r.allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
  r.bcast = (fn_bcast)dlsym(h, "ncclBroadcast");
  r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  VCHECK(r.get_id && r.init_rank && r.allreduce && r.bcast && r.destroy, VACNIC_UNSUPPORTED, "comm_load: librccl lacks an expected symbol");
  g_rccl = r;
  return VACNIC_OK;
}

extern "C" int vacnic_comm_unique_id(void* out128) {
  VCHECK(g_rccl.h, VACNIC_UNSUPPORTED, "comm_unique_id: call vacnic_comm_load first");
  VCHECK(out128, VACNIC_BAD_SHAPE, "comm_unique_id: null buffer");
  return check_rccl(g_rccl.get_id((UniqueId*)out128), "comm_unique_id");
}

// one communicator over `world` ranks (one process per GPU; the calling process's current HIP device is its GPU); id128 = the
// 128 bytes rank 0 obtained from vacnic_comm_unique_id and handed to every rank out of band.  Returns a handle >= 0, or -1.
extern "C" int64_t vacnic_comm_init(const void* id128, int rank, int world) {
  if (!g_rccl.h) { vacnic_set_error("comm_init: call vacnic_comm_load first"); return -1; }
  if (!id128 || world < 1 || rank < 0 || rank >= world) { vacnic_set_error("comm_init: bad rank %d / world %d", rank, world); return -1; }
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  if (check_rccl(g_rccl.init_rank(&c, world, id, rank), "comm_init")) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  g_comms.push_back(c);
  return (int64_t)g_comms.size() - 1;
}

// in-place SUM all-reduce of `count` elements at buf over the communicator, stream-ordered on `stream`; dtype 0 = f32, 1 = bf16
extern "C" int vacnic_allreduce_bucket(int64_t comm, void* buf, int64_t count, int dtype, void* stream) {
  VPLAN_REC(vacnic_allreduce_bucket, comm, buf, count, dtype, stream);
  Comm c = get(comm);
  VCHECK(c && buf && count > 0 && (dtype == 0 || dtype == 1), VACNIC_BAD_SHAPE, "allreduce_bucket: bad communicator / buffer / dtype");
  return check_rccl(g_rccl.allreduce(buf, buf, (size_t)count, dtype == 0 ? kFloat32 : kBfloat16, kSum, c, (hipStream_t)stream), "allreduce_bucket");
}

extern "C" int vacnic_comm_broadcast(int64_t comm, void* buf, int64_t count, int dtype, int root, void* stream) {
  VPLAN_REC(vacnic_comm_broadcast, comm, buf, count, dtype, root, stream);
  Comm c = get(comm);
  VCHECK(c && buf && count > 0 && (dtype == 0 || dtype == 1), VACNIC_BAD_SHAPE, "comm_broadcast: bad communicator / buffer / dtype");
  return check_rccl(g_rccl.bcast(buf, buf, (size_t)count, dtype == 0 ? kFloat32 : kBfloat16, root, c, (hipStream_t)stream), "comm_broadcast");
}

extern "C" int vacnic_comm_destroy(int64_t comm) {
  Comm c = get(comm);
  VCHECK(c, VACNIC_BAD_SHAPE, "comm_destroy: no such communicator");
  std::lock_guard<std::mutex> lk(g_mu);
  g_comms[(size_t)comm] = nullptr;
  return check_rccl(g_rccl.destroy(c), "comm_destroy");
}

// Named events: slot `ev` (0 .. 511) is recorded on one stream and waited for on another — unlike vacnic_stream_fence the two
// halves are separate calls, so the waiter follows ONE point of the producer's stream (a bucket's all-reduce) and not whatever
// that stream has been given since.  Both halves are recordable; a slot is reused every step by the same pair.
extern "C" int vacnic_event_record(int ev, void* stream) {
  VPLAN_REC(vacnic_event_record, ev, stream);
  VCHECK(ev >= 0 && ev < NEV, VACNIC_BAD_SHAPE, "event_record: slot %d outside [0, %d)", ev, NEV);
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ev_made[ev]) {
      if (hipEventCreateWithFlags(&g_ev[ev], hipEventDisableTiming) != hipSuccess) { vacnic_set_error("event_record: hipEventCreate failed"); return VACNIC_HIP_ERROR; }
      g_ev_made[ev] = true;
    }
  }
  if (hipEventRecord(g_ev[ev], (hipStream_t)stream) != hipSuccess) { vacnic_set_error("event_record: %s", hipGetErrorString(hipGetLastError())); return VACNIC_HIP_ERROR; }
  return VACNIC_OK;
}

extern "C" int vacnic_event_wait(int ev, void* stream) {
  VPLAN_REC(vacnic_event_wait, ev, stream);
  VCHECK(ev >= 0 && ev < NEV && g_ev_made[ev], VACNIC_BAD_SHAPE, "event_wait: slot %d was never recorded", ev);
  if (hipStreamWaitEvent((hipStream_t)stream, g_ev[ev], 0) != hipSuccess) { vacnic_set_error("event_wait: %s", hipGetErrorString(hipGetLastError())); return VACNIC_HIP_ERROR; }
  return VACNIC_OK;
}
