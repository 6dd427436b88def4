// LDS-DMA feed-rate probe for the 256x256 GEMM tile (no MFMA, no LDS reads): how fast can one workgroup per CU pull its
// X / W panels L2 -> LDS with `buffer_load_dwordx4 ... lds`, as a function of the ROW WIDTH of one stage?
//   variant A (ROWB = 64):  stage = 256 rows x 64 B of X + 256 rows x 64 B of W   (the 32-wide K stages of the ping-pong loop)
//   variant B (ROWB = 128): stage = 128 rows x 128 B of X + 128 rows x 128 B of W (half tiles along M/N, 64-wide K)
// Same 32 KiB per stage, same 5-slot ring (4 stages in flight), same total bytes, same tile -> workgroup -> XCD mapping as
// gemm_kernel.  A 64-B row segment uses half of every 128-B line it pulls through the CU's vector L1.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/bin/dma_probe && tools/bin/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int OOB = 0x7ffffff0;

template <int ROWB, int NSLOT>
__global__ __launch_bounds__(512) void dma_probe(const char* x, const char* w, unsigned xbytes, unsigned wbytes, int Kbytes, int tiles_n,
                                                 int nt, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, wbytes, 0x00020000);
  constexpr int STAGE = 32768;
  // stage s: ROWB = 64 : k-offset = s * 64 bytes, all 256 rows of both operands
  //          ROWB = 128: k-offset = (s / 2) * 128 bytes, rows [(s & 1) * 128, +128)
  const int nstage = ROWB == 64 ? Kbytes / 64 : Kbytes / 128 * 2;
  auto issue = [&](int s, int slot) {
    char* base = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave * 4 + i;                     // 32 pieces of 1 KiB per stage: 0..15 X, 16..31 W
      const int op = piece >> 4, pp = piece & 15;
      int row, kb;
      if (ROWB == 64) { row = pp * 16 + (lane >> 2); kb = s * 64 + (lane & 3) * 16; }
      else { row = (s & 1) * 128 + pp * 8 + (lane >> 3); kb = (s >> 1) * 128 + (lane & 7) * 16; }
      const int grow = (op == 0 ? tm : tn) * 256 + row;
      const int voff = s < nstage ? (int)((unsigned)grow * (unsigned)Kbytes + (unsigned)kb) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(op == 0 ? xs : ws, LDS_PTR(base + piece * 1024), 16, voff, 0, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s) issue(s, s);
  int slot = NSLOT - 1;
  for (int s = 0; s < nstage; ++s) {
    issue(s + NSLOT - 1, slot);
    slot = slot + 1 == NSLOT ? 0 : slot + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NSLOT - 1)) : "memory");     // stage s landed
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(float*)(smem + (lane & 7) * 4);
}

template <int ROWB, int NSLOT>
float run(const char* x, const char* w, size_t xb, size_t wb, int M, int N, int Kbytes, float* sink, int iters) {
  auto k = dma_probe<ROWB, NSLOT>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * 32768);
  const int tiles_n = N / 256, nt = (M / 256) * tiles_n;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nt), dim3(512), NSLOT * 32768, 0, x, w, (unsigned)xb, (unsigned)wb, Kbytes, tiles_n, nt, sink);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(nt), dim3(512), NSLOT * 32768, 0, x, w, (unsigned)xb, (unsigned)wb, Kbytes, tiles_n, nt, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main() {
  const int M = 16384;
  for (int N : {1024, 4096}) {
    for (int K : {1024, 4096}) {
      const int Kbytes = K * 2;
      const size_t xb = (size_t)M * Kbytes, wb = (size_t)N * Kbytes;
      char *x, *w; float* sink;
      hipMalloc(&x, xb); hipMalloc(&w, wb); hipMalloc(&sink, 65536 * 4);
      hipMemset(x, 1, xb); hipMemset(w, 1, wb);
      const int nt = (M / 256) * (N / 256);
      const double bytes_per_wg = 2.0 * 256 * Kbytes;           // X panel + W panel of one tile
      for (int rep = 0; rep < 2; ++rep) {
        const float a4 = run<64, 4>(x, w, xb, wb, M, N, Kbytes, sink, 20);
        const float b4 = run<128, 4>(x, w, xb, wb, M, N, Kbytes, sink, 20);
        const float a5 = run<64, 5>(x, w, xb, wb, M, N, Kbytes, sink, 20);
        const float b5 = run<128, 5>(x, w, xb, wb, M, N, Kbytes, sink, 20);
        const double rounds = (nt + 255) / 256;
        auto rate = [&](float us) { return bytes_per_wg * rounds / (us * 1e-6) / 1e9; };      // GB/s per CU while a round runs
        printf("M=%d N=%d K=%d tiles=%d | 64-B rows: 4 slots %7.1f us (%5.1f GB/s/CU)  5 slots %7.1f us (%5.1f) | 128-B rows: 4 slots %7.1f us (%5.1f)  5 slots %7.1f us (%5.1f)\n",
               M, N, K, nt, a4, rate(a4), a5, rate(a5), b4, rate(b4), b5, rate(b5));
      }
      hipFree(x); hipFree(w); hipFree(sink);
    }
  }
  return 0;
}
