// Grid-barrier probe for the persistent single-token decoder: what does one phase boundary cost on MI355X
// when G workgroups (one per CU) exchange a few KiB of activations through HBM/L2 between phases?
//   variant F (fences):  plain loads/stores of the payload; release fence (buffer_wbl2 sc1) -> arrive -> spin -> acquire
//                        fence (buffer_inv sc1).  The payload broadcast is served by each XCD's L2.
//   variant S (sc1):     payload through sc1 buffer loads/stores (agent-coherent: write-through / miss-always in the XCD's
//                        L2), no cache maintenance; s_waitcnt vmcnt(0) before the arrive.
// Every phase each thread stores 16 B tagged with (phase, wg, tid) and, after the barrier, reads 3 x 16 B written by three
// OTHER workgroups in that phase and checks the tags (cross-XCD visibility), so the probe is also the correctness test of
// the barrier + visibility recipe.  A spin that exceeds ~0.5 s sets an error flag and every workgroup leaves.
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar_probe.hip -o tools/bin/gridbar_probe && tools/bin/gridbar_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
  __shared__ int bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    bad = 0;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, AGENT);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 50000000ll || __hip_atomic_load(err, __ATOMIC_RELAXED, AGENT)) {   // 100 MHz clock: 0.5 s
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, AGENT);
        bad = 1;
        break;
      }
    }
  }
  __syncthreads();
  return bad == 0;
}

// Two-level arrive + spread release flags.  Workgroup w belongs to group w % NGRP (all on XCD w % 8 when NGRP is a multiple
// of 8); counters and flags sit on their own 128-byte lines.  ph is 1-based and monotonic inside a launch.
//   arrive:  old = grp[g]++ ; the group's last arriver does top++ ; the last of those writes flag[0..NFLAG) = ph
//   wait:    spin on flag[w % NFLAG] >= ph
// NGRP == 1 degenerates to one counter (the arrive serialises on one address) but keeps the spread flags.
template <int NGRP, int NFLAG>
__device__ __forceinline__ bool grid_barrier2(unsigned* bar, unsigned ph, int G, unsigned* err) {
  __shared__ int bad2;
  unsigned* grp = bar; unsigned* top = bar + 32 * 64; unsigned* flag = bar + 32 * 65;
  __syncthreads();
  if (threadIdx.x == 0) {
    bad2 = 0;
    const int w = blockIdx.x, g = w % NGRP;
    const unsigned gsize = (unsigned)((G - g + NGRP - 1) / NGRP);
    const unsigned old = __hip_atomic_fetch_add(grp + 32 * g, 1u, __ATOMIC_RELAXED, AGENT);
    bool last = old + 1 == ph * gsize;
    if (last && NGRP > 1) last = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, AGENT) + 1 == ph * NGRP;
    if (last) {
#pragma unroll
      for (int f = 0; f < NFLAG; ++f) __hip_atomic_store(flag + 32 * f, ph, __ATOMIC_RELAXED, AGENT);
    } else {
      const long long t0 = wall_clock64();
      unsigned* fl = flag + 32 * (w % NFLAG);
      while (__hip_atomic_load(fl, __ATOMIC_RELAXED, AGENT) < ph) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 50000000ll) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, AGENT); bad2 = 1; break; }
      }
    }
  }
  __syncthreads();
  return bad2 == 0;
}

template <int NGRP, int NFLAG, bool PAYLOAD>
__global__ __launch_bounds__(256) void probe2(unsigned* bar, unsigned* err, u32x4* buf, int nph, unsigned* mism, unsigned base) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (unsigned)(2u * G * 256 * 16), 0x00020000);
  unsigned bad = 0;
  for (int ph = 0; ph < nph; ++ph) {
    const unsigned half = (ph & 1) * G * 256;
    if (PAYLOAD) {
      const u32x4 v = (u32x4){(unsigned)ph + base, (unsigned)wg, (unsigned)tid, 0x5a5a0000u + ph};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, (half + wg * 256 + tid) * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!grid_barrier2<NGRP, NFLAG>(bar, (unsigned)ph + 1, G, err)) return;
    if (PAYLOAD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int src = (wg * 7 + 1 + j * 61) % G;
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (half + src * 256 + tid) * 16, 0, 16);
        if (r[0] != (unsigned)ph + base || r[1] != (unsigned)src || r[2] != (unsigned)tid) ++bad;
      }
    }
  }
  if (bad) atomicAdd(mism, bad);
  // exit: the last workgroup out clears the barrier state for the next launch
  __syncthreads();
  if (tid == 0) {
    unsigned* exitc = bar + 32 * 66 + 32 * 64;
    const unsigned old = __hip_atomic_fetch_add(exitc, 1u, __ATOMIC_RELAXED, AGENT);
    if (old == (unsigned)G - 1) {
      for (int i = 0; i < NGRP; ++i) __hip_atomic_store(bar + 32 * i, 0u, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(bar + 32 * 64, 0u, __ATOMIC_RELAXED, AGENT);
      for (int f = 0; f < NFLAG; ++f) __hip_atomic_store(bar + 32 * 65 + 32 * f, 0u, __ATOMIC_RELAXED, AGENT);
      __hip_atomic_store(exitc, 0u, __ATOMIC_RELAXED, AGENT);
    }
  }
}

template <int VARIANT, bool PAYLOAD>
__global__ __launch_bounds__(256) void probe(unsigned* ctr, unsigned* err, u32x4* buf, int nph, unsigned* mism, unsigned base) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (unsigned)(2u * G * 256 * 16), 0x00020000);
  unsigned bad = 0;
  for (int ph = 0; ph < nph; ++ph) {
    const unsigned half = (ph & 1) * G * 256;
    if (PAYLOAD) {
      const u32x4 v = (u32x4){(unsigned)ph + base, (unsigned)wg, (unsigned)tid, 0x5a5a0000u + ph};
      if (VARIANT == 0) buf[half + wg * 256 + tid] = v;
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (half + wg * 256 + tid) * 16, 0, 16);
    }
    if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!grid_barrier(ctr, (unsigned)(ph + 1) * G, err)) return;
    if (VARIANT == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (PAYLOAD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int src = (wg * 7 + 1 + j * 61) % G;
        u32x4 r;
        if (VARIANT == 0) r = buf[half + src * 256 + tid];
        else r = __builtin_amdgcn_raw_buffer_load_b128(rs, (half + src * 256 + tid) * 16, 0, 16);
        if (r[0] != (unsigned)ph + base || r[1] != (unsigned)src || r[2] != (unsigned)tid) ++bad;
      }
    }
  }
  if (bad) atomicAdd(mism, bad);
  // self-reset: the last workgroup through the final arrive puts the counter back to 0 for the next launch
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, AGENT);
    if (old == (unsigned)(nph + 1) * G - 1) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, AGENT);
  }
}


// ---- XCD-local exchange: the 32 workgroups that share one XCD's L2 synchronise among themselves only.  Payload by plain stores
// (the vector L1 is write-through, the data lands in the XCD's L2) or sc1 stores, and sc1 loads; the barrier
// counter / flag of an XCD are touched by that XCD alone.  XCC_ID comes from the hardware register, the rank inside the XCD from a
// monotonic per-XCD counter (old & 31; old >> 5 must equal the launch number).
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf); }

template <bool PAYLOAD, bool PLAIN_STORE>
__global__ __launch_bounds__(256) void probe_xcd(unsigned* bar, unsigned* err, u32x4* buf, int nph, unsigned* mism, unsigned base, unsigned launch_no,
                                                 unsigned* xcd_hist) {
  const int tid = threadIdx.x;
  __shared__ int s_x, s_rank, s_bad;
  if (tid == 0) {
    s_x = xcc_id();
    const unsigned old = __hip_atomic_fetch_add(bar + 32 * (40 + s_x), 1u, __ATOMIC_RELAXED, AGENT);
    s_rank = (int)(old & 31u);
    s_bad = (old >> 5) != launch_no;
    if (s_bad) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, AGENT);
    if (launch_no == 0) atomicAdd(xcd_hist + s_x * 8 + (blockIdx.x & 7), 1u);
  }
  __syncthreads();
  if (s_bad) return;
  const int x = s_x, rank = s_rank;
  unsigned* cnt = bar + 32 * (50 + x);                  // this XCD's arrive counter (monotonic over launches: 32 per phase)
  unsigned* flag = bar + 32 * (60 + x);
  const unsigned ph0 = launch_no * (unsigned)nph;
  u32x4* mybuf = buf + (size_t)x * 2 * 32 * 256;        // [2][32 workgroups][256 threads]
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mybuf, 0, 2u * 32 * 256 * 16, 0x00020000);
  unsigned bad = 0;
  for (int ph = 0; ph < nph; ++ph) {
    const unsigned half = (ph & 1) * 32 * 256;
    if (PAYLOAD) __builtin_amdgcn_raw_buffer_store_b128((u32x4){(unsigned)ph + base, (unsigned)rank, (unsigned)tid, 0u}, rs, (half + rank * 256 + tid) * 16, 0, PLAIN_STORE ? 0 : 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned gph = ph0 + (unsigned)ph + 1u;
      if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, AGENT) + 1 == gph * 32u) {
        __hip_atomic_store(flag, gph, __ATOMIC_RELAXED, AGENT);
      } else {
        const long long t0 = wall_clock64();
        while (__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void*)flag, 0, 4, 0x00020000), 0, 0, 16) < gph) {
          if (wall_clock64() - t0 > 50000000ll) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, AGENT); s_bad = 1; break; }
        }
      }
    }
    __syncthreads();
    if (s_bad) return;
    if (PAYLOAD) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int src = (rank * 7 + 1 + j * 11) & 31;
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (half + src * 256 + tid) * 16, 0, 16);    // sc1 (sc0 alone hits a stale L1 line: the first version of this probe timed out)
        if (r[0] != (unsigned)ph + base || r[1] != (unsigned)src || r[2] != (unsigned)tid) ++bad;
      }
    }
  }
  if (bad) atomicAdd(mism, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VARIANT, bool PAYLOAD>
static void run(const char* name, int G, int nph, unsigned* ctr, unsigned* err, u32x4* buf, unsigned* mism) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned base = 1000;
  for (int rep = 0; rep < 6; ++rep) {
    base += 7777;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<VARIANT, PAYLOAD>), dim3(G), dim3(256), 0, 0, ctr, err, buf, nph, mism, base);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  unsigned h_err = 0, h_m = 0, h_c = 0;
  CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&h_m, mism, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&h_c, ctr, 4, hipMemcpyDeviceToHost));
  printf("%-34s G=%3d  %6.2f us/phase  (launch of %d phases %.1f us)  timeout=%u mismatches=%u counter_after=%u\n", name, G,
         best * 1e3f / nph, nph, best * 1e3f, h_err, h_m, h_c);
  fflush(stdout);
  if (h_err) { printf("barrier timed out: stopping\n"); exit(2); }
}

template <int NGRP, int NFLAG, bool PAYLOAD>
static void run2(const char* name, int G, int nph, unsigned* bar, unsigned* err, u32x4* buf, unsigned* mism) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  unsigned base = 500000;
  for (int rep = 0; rep < 6; ++rep) {
    base += 7777;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe2<NGRP, NFLAG, PAYLOAD>), dim3(G), dim3(256), 0, 0, bar, err, buf, nph, mism, base);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  unsigned h_err = 0, h_m = 0;
  CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&h_m, mism, 4, hipMemcpyDeviceToHost));
  printf("%-34s G=%3d  %6.2f us/phase  NGRP=%d NFLAG=%d payload=%d timeout=%u mismatches=%u\n", name, G, best * 1e3f / nph, NGRP, NFLAG,
         (int)PAYLOAD, h_err, h_m);
  fflush(stdout);
  if (h_err) { printf("barrier timed out: stopping\n"); exit(2); }
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
  unsigned *ctr, *err, *mism; u32x4* buf;
  CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&err, 256)); CK(hipMalloc(&mism, 256));
  CK(hipMalloc(&buf, 2u * 512 * 256 * 16));
  CK(hipMemset(ctr, 0, 256)); CK(hipMemset(err, 0, 256)); CK(hipMemset(mism, 0, 256));
  CK(hipMemset(buf, 0, 2u * 512 * 256 * 16));
  const int nph = 200;
  const int cus = pr.multiProcessorCount;
  for (int G : {64, 128, cus}) {
    if (G > cus) continue;
    run<0, false>("fences, no payload", G, nph, ctr, err, buf, mism);
    run<1, false>("sc1 (no fences), no payload", G, nph, ctr, err, buf, mism);
    run<0, true>("fences + payload", G, nph, ctr, err, buf, mism);
    run<1, true>("sc1 payload (no fences)", G, nph, ctr, err, buf, mism);
  }
  unsigned* bar; CK(hipMalloc(&bar, 32 * 256 * 4)); CK(hipMemset(bar, 0, 32 * 256 * 4));
  for (int G : {128, cus}) {
    if (G > cus) continue;
    run2<1, 1, false>("one counter, one flag", G, nph, bar, err, buf, mism);
    run2<1, 16, false>("one counter, 16 flags", G, nph, bar, err, buf, mism);
    run2<8, 16, false>("8 groups, 16 flags", G, nph, bar, err, buf, mism);
    run2<16, 16, false>("16 groups, 16 flags", G, nph, bar, err, buf, mism);
    run2<16, 32, false>("16 groups, 32 flags", G, nph, bar, err, buf, mism);
    run2<32, 32, false>("32 groups, 32 flags", G, nph, bar, err, buf, mism);
    run2<16, 16, true>("16 groups, 16 flags + payload", G, nph, bar, err, buf, mism);
    run2<32, 32, true>("32 groups, 32 flags + payload", G, nph, bar, err, buf, mism);
  }
  {
    unsigned* hist; CK(hipMalloc(&hist, 64 * 4)); CK(hipMemset(hist, 0, 64 * 4));
    CK(hipMemset(bar, 0, 32 * 256 * 4)); CK(hipMemset(err, 0, 256)); CK(hipMemset(mism, 0, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned launch_no = 0;
    for (int payload = 0; payload < 3; ++payload) {
      float best = 1e9f; unsigned base = 900000;
      for (int rep = 0; rep < 6; ++rep) {
        base += 7777;
        CK(hipEventRecord(e0, 0));
        if (payload == 2) hipLaunchKernelGGL((probe_xcd<true, true>), dim3(256), dim3(256), 0, 0, bar, err, buf, nph, mism, base, launch_no, hist);
        else if (payload) hipLaunchKernelGGL((probe_xcd<true, false>), dim3(256), dim3(256), 0, 0, bar, err, buf, nph, mism, base, launch_no, hist);
        else hipLaunchKernelGGL((probe_xcd<false, false>), dim3(256), dim3(256), 0, 0, bar, err, buf, nph, mism, base, launch_no, hist);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        ++launch_no;
      }
      unsigned h_err = 0, h_m = 0;
      CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_m, mism, 4, hipMemcpyDeviceToHost));
      printf("XCD-local barrier (32 workgroups per XCD, L2-coherent)%s  %6.2f us/phase  err=%u mismatches=%u\n", payload == 2 ? " + payload (plain stores)" : payload ? " + payload (sc1 stores)" : "",
             best * 1e3f / nph, h_err, h_m);
    }
    unsigned hh[64]; CK(hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost));
    printf("workgroups per XCD (first launch), rows = XCC_ID, columns = blockIdx & 7:\n");
    for (int x = 0; x < 8; ++x) { printf("  xcc %d:", x); for (int j = 0; j < 8; ++j) printf(" %3u", hh[x * 8 + j]); printf("\n"); }
  }
  return 0;
}
