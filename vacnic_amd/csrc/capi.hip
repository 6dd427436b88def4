// Error plumbing + version for libvacnic_hip.so (C-ABI never throws/aborts: status + message).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "ok";

namespace vplan { thread_local bool failed = false; }

void vacnic_set_error(const char* fmt, ...) {
  vplan::failed = true;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vacnic_last_error_string(void) { return g_err; }
extern "C" int vacnic_version(void) { return 100; }


// ---------------------------------------------------------------------------------------------------------------------
// Launch plans.  The reference drives every kernel of a step from Python (torch ops, TRAIN:253-383); so does the eager path of
// this library (~1400 C-ABI calls per step, ~40 us of Python + ctypes + autograd each).  A plan is that same call sequence
// RECORDED once — each entry point pushes a closure over its (frozen) arguments while it runs — and replayed from C++:
//     h = vacnic_plan_begin();  ... one step through the ordinary entry points ...;  vacnic_plan_end(h);
//     vacnic_plan_replay(h, 0, vacnic_plan_size(h));            // every later step: one call
// Replay issues exactly the recorded launches on the recorded streams in the recorded order (cross-stream dependencies are
// vacnic_stream_fence calls, recorded like any other), so the GPU-side schedule is the eager one — unlike a hipGraph, whose
// nodes launched no cheaper and overlapped less on this stack.  The caller keeps every buffer at its recorded
// address (a private allocator pool) and refreshes the step's inputs in place; per-step scalars (learning rate, step count,
// dropout counter) already live in device memory.  vacnic_plan_mark() returns the index of the next command: the host splits
// a replay there to interleave work of its own (the RCCL bucket launches of the DDP reducer).
#include <vector>
#include <mutex>
#include <thread>
#include <atomic>

namespace vplan {
struct Plan { std::vector<std::function<int()>> cmds; bool open = false; };
static std::mutex g_mu;                          // plan table (begin / end / destroy / lookup); a recording itself is single-threaded
static std::vector<Plan*> g_plans;
static std::atomic<Plan*> g_rec{nullptr};
static std::thread::id g_rec_tid;                // written before g_rec is published, read only when g_rec is non-null
static int g_paused = 0;                         // vacnic_plan_pause: the recording thread's host-side actions launch without being recorded
thread_local int depth = 0;
// recording is bound to the thread that began it: a C-ABI call from any other thread (an autograd worker, a loader thread) while a
// plan is open launches normally and is NOT frozen into the plan
bool active() { return g_rec.load(std::memory_order_acquire) != nullptr && g_rec_tid == std::this_thread::get_id() && g_paused == 0; }
void push(std::function<int()> f) { g_rec.load(std::memory_order_relaxed)->cmds.push_back(std::move(f)); }
size_t size() { return g_rec.load(std::memory_order_relaxed)->cmds.size(); }
void truncate(size_t n) { auto& c = g_rec.load(std::memory_order_relaxed)->cmds; if (n < c.size()) c.resize(n); }
static Plan* get(int64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (h >= 0 && h < (int64_t)g_plans.size()) ? g_plans[(size_t)h] : nullptr;
}
}  // namespace vplan

extern "C" int64_t vacnic_plan_begin(void) {
  std::lock_guard<std::mutex> lk(vplan::g_mu);
  if (vplan::g_rec.load()) { vacnic_set_error("plan_begin: a plan is already being recorded"); return -1; }
  vplan::Plan* p = new vplan::Plan();
  p->open = true;
  vplan::g_plans.push_back(p);
  vplan::g_rec_tid = std::this_thread::get_id();
  vplan::g_rec.store(p, std::memory_order_release);
  return (int64_t)vplan::g_plans.size() - 1;
}

extern "C" int vacnic_plan_end(int64_t h) {
  vplan::Plan* p = vplan::get(h);
  VCHECK(p && p == vplan::g_rec.load() && vplan::g_rec_tid == std::this_thread::get_id(), VACNIC_BAD_SHAPE,
         "plan_end: plan %ld is not the one this thread is recording", (long)h);
  p->open = false;
  vplan::g_paused = 0;
  vplan::g_rec.store(nullptr, std::memory_order_release);
  return VACNIC_OK;
}

// While a plan is being recorded, work the HOST does at a mark (the DDP reducer's casts and collectives, anything the caller
// repeats itself at every replay) must not be frozen into the plan: pause(1) ... pause(0) around it (nests).
extern "C" int vacnic_plan_pause(int on) {
  VCHECK(vplan::g_rec.load() != nullptr && vplan::g_rec_tid == std::this_thread::get_id(), VACNIC_BAD_SHAPE, "plan_pause: this thread is not recording a plan");
  vplan::g_paused += on ? 1 : -1;
  if (vplan::g_paused < 0) vplan::g_paused = 0;
  return VACNIC_OK;
}

extern "C" int64_t vacnic_plan_size(int64_t h) {
  vplan::Plan* p = vplan::get(h);
  return p ? (int64_t)p->cmds.size() : -1;
}

extern "C" int64_t vacnic_plan_mark(void) {
  return (vplan::g_rec.load() != nullptr && vplan::g_rec_tid == std::this_thread::get_id()) ? (int64_t)vplan::size() : -1;
}

extern "C" int vacnic_plan_replay(int64_t h, int64_t first, int64_t last) {
  vplan::Plan* p = vplan::get(h);
  VCHECK(p && !p->open, VACNIC_BAD_SHAPE, "plan_replay: plan %ld does not exist or is still being recorded", (long)h);
  VCHECK(!vplan::active(), VACNIC_UNSUPPORTED, "plan_replay: not while this thread is recording a plan");
  VCHECK(first >= 0 && first <= last && last <= (int64_t)p->cmds.size(), VACNIC_BAD_SHAPE, "plan_replay: range [%ld, %ld) outside the plan's %ld commands",
         (long)first, (long)last, (long)p->cmds.size());
  for (int64_t i = first; i < last; ++i)
    if (int e = p->cmds[(size_t)i]()) return e;
  return VACNIC_OK;
}

extern "C" int vacnic_plan_destroy(int64_t h) {
  vplan::Plan* p = vplan::get(h);
  VCHECK(p && p != vplan::g_rec.load(), VACNIC_BAD_SHAPE, "plan_destroy: plan %ld does not exist or is being recorded", (long)h);
  std::lock_guard<std::mutex> lk(vplan::g_mu);
  delete p;
  vplan::g_plans[(size_t)h] = nullptr;
  return VACNIC_OK;
}

// dst waits for everything enqueued on src so far (hipEventRecord + hipStreamWaitEvent on an event of the library's ring; the
// pair is issued back to back, so a slot is free again as soon as the call returns).  Recordable: the cross-stream edges of a
// step are part of its plan.
extern "C" int vacnic_stream_fence(void* src, void* dst) {
  VPLAN_REC(vacnic_stream_fence, src, dst);
  if (src == dst) return VACNIC_OK;
  constexpr int RING = 256;
  static hipEvent_t ring[RING];
  static bool made[RING];
  static unsigned next = 0;
  static std::mutex mu;                          // slot choice + record/wait pair: fences may come from more than one host thread
  std::lock_guard<std::mutex> lk(mu);
  const unsigned i = next++ % RING;
  if (!made[i]) {
    if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess) { vacnic_set_error("stream_fence: hipEventCreate failed"); return VACNIC_HIP_ERROR; }
    made[i] = true;
  }
  if (hipEventRecord(ring[i], (hipStream_t)src) != hipSuccess || hipStreamWaitEvent((hipStream_t)dst, ring[i], 0) != hipSuccess) {
    vacnic_set_error("stream_fence: %s", hipGetErrorString(hipGetLastError()));
    return VACNIC_HIP_ERROR;
  }
  return VACNIC_OK;
}
