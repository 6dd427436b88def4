// bf16 MFMA GEMM for gfx950:  out[m][n] = epi(alpha * sum_k X(m,k) W(n,k) + bias[n]).
//
// Replaces every nn.Linear on the VACNIC path (q/k/v/out_proj MFULL:449-452, fc1/fc2 MFULL:581-582,
// _linear_1up/down :587-588, _face_up/down :607-608, ner_map_up/down :594-595, prompt_mlp :1136,
// visual_map :1144, lm_head :1885) and their dgrad / wgrad.
//
// Design (CDNA4): block tile BMxBN with WMxWN waves, each wave a (BM/WM)x(BN/WN) sub-tile of v_mfma_f32_16x16x32_bf16
// tiles.  Configurations (chosen per launch by the host cost model below, or forced by tile_hint):
//   256x256, 8 waves (2x4, 128x64 per wave), 32-wide K stages in a 4-slot LDS ring (128 KiB, one block per CU), PING-PONG
//            K loop: the two waves of every SIMD alternate MFMA and LDS/DMA phases — the large training GEMMs;
//   128x128, 4 waves, 64-wide K tiles, two blocks per CU, software-pipelined loop (barrier in the middle of the tile's MFMAs),
//            register->global epilogue — when that fills the chip better;
//   64x128,  4 waves, 4-deep ring — small latency-bound problems;   256x128 ping-pong (hint 264);
//   skinny   (M <= 8): W-streaming kernel without LDS — the single-token decoder.
// Operands go HBM -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR round trip; the SRD range check zero-fills
// M/N/K edges) with counted vmcnt waits.  The LDS image is lane-linear per wave-instruction,
// so the bank-conflict XOR swizzle is applied to the per-lane SOURCE address and again on the
// fragment read (guide rule 21).  An operand whose reduction index is the strided one in memory
// (dgrad's W, wgrad's dY and X) is staged as [k][row] and read with ds_read_b64_tr_b16 (hardware
// transpose), so no transposed copies are ever materialised.  MFMA A <- W rows, B <- X rows, i.e. the
// wave computes the C^T tile; the fp32 C tile is then staged through LDS so that every global access
// of the epilogue (output, saved pre-activation, activation-backward source, residual) is a 16-byte
// row-contiguous access and split-K atomics are 256 contiguous bytes per wave-instruction.
#include "common.h"

namespace {

constexpr int BK = 64;
constexpr int OOB = 0x7ffffff0;               // voffset beyond any (<2 GiB) buffer -> load returns 0

struct GemmP {
  const bf16_t* x; const bf16_t* w; const float* bias;
  void* out; bf16_t* preact; const bf16_t* dact_src; const bf16_t* residual;
  int M, N, K;
  int ldx, ldw, ldo;
  int act, out_mode, split_k, k_per_split;
  float alpha;
  unsigned x_bytes, w_bytes;
  int tiles_m, tiles_n;
  int debug;            // profiling aid (tile_hint >= 1000): bit0 skip the global stores, bit1 skip the K loop, bit2 skip the epilogue, bit3 return at once
};

// f(k) of the K-strided swizzle: distinct for the 8 k-rows one tr-read half touches.
__device__ __forceinline__ int fk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue this wave's LDS-DMA loads for one operand tile of ROWS rows x BKT k (BKT = 64 or 32).
//   KS=false, BKT=64: tile [ROWS][64 k], 128-B LDS rows, chunk' = chunk ^ (row & 7); one 1-KiB piece = 8 rows
//   KS=false, BKT=32: two 64-B global rows share one 128-B LDS row R = row/2 (chunk = (row&1)*4 + kchunk),
//                     chunk' = chunk ^ (R & 7); one piece = 16 rows
//   KS=true : tile [BKT k][ROWS], 2*ROWS-B rows, chunk' = chunk ^ swz(k); one piece = 512/ROWS k-rows
template <bool KS, int ROWS, int BKT, int NWAVE>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int r0, int R,
                                           int k0, int kend, int ld, int wave, int lane) {
  constexpr int PIECES = ROWS * BKT * 2 / 1024 / NWAVE;      // 1-KiB pieces per wave
  static_assert(PIECES >= 1, "tile too small for this wave count");
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int blk = wave * PIECES + i;
    int voff;
    if (!KS) {
      int row, kch;
      if (BKT == 64) {
        row = blk * 8 + (lane >> 3);
        kch = (lane & 7) ^ (row & 7);
      } else {
        const int Rl = blk * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ (Rl & 7);
        row = 2 * Rl + (lc >> 2);
        kch = lc & 3;
      }
      const int gr = r0 + row, gk = k0 + kch * 8;
      voff = (gr < R && gk < kend) ? (int)(((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u) : OOB;
    } else {
      constexpr int LPR = ROWS / 8;             // lanes (16-B chunks) per k-row
      const int k = blk * (64 / LPR) + lane / LPR;
      const int lc = (lane % LPR) ^ ((fk(k) << 1) & (LPR - 1));   // 64-row tiles have only 8 chunks per k-row
      const int gk = k0 + k, gr = r0 + lc * 8;
      voff = (gk < kend && gr < R) ? (int)(((unsigned)gk * (unsigned)ld + (unsigned)gr) * 2u) : OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + blk * 1024), 16, voff, 0, 0, 0);
  }
}

// Fragment for MFMA 16x16x32: lane l gets element (row = rbase + (l&15), k = kk*32 + 8*(l>>4) + j), j=0..7.
template <bool KS, int ROWS, int BKT>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rbase, int kk, int lane) {
  if (!KS) {
    const int row = rbase + (lane & 15);
    if (BKT == 64) {
      const int chunk = kk * 4 + (lane >> 4);
      return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    } else {
      const int Rl = row >> 1;
      const int chunk = (row & 1) * 4 + (lane >> 4);
      return *(const bf16x8*)(lds_tile + Rl * 128 + ((chunk ^ (Rl & 7)) << 4));
    }
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = (rbase >> 3) + (p >> 1);
    bf16x8 r;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = kk * 32 + 8 * g + 4 * hh + q;
      const int phys = chunk ^ ((fk(k) << 1) & (ROWS / 8 - 1));
      const char* a = lds_tile + k * (ROWS * 2) + phys * 16 + (p & 1) * 8;
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));
      r[4 * hh + 0] = t[0]; r[4 * hh + 1] = t[1]; r[4 * hh + 2] = t[2]; r[4 * hh + 3] = t[3];
    }
    return r;
  }
}

// Epilogue on 8 consecutive outputs of one row (read back from the LDS-staged C tile): bias, saved
// pre-activation, activation or fused activation-backward, residual, then a 16-byte (bf16) /
// 2x16-byte (f32) store or 8 f32 atomics on 32 contiguous bytes.
__device__ __forceinline__ void load8bf(const bf16_t* p, float v[8]) {
  u32x4 r = *(const u32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}
__device__ __forceinline__ void store8bf(bf16_t* p, const float v[8]) {
  *(u32x4*)p = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
}

__device__ __forceinline__ void unpack8bf(u32x4 r, float v[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(r[i] << 16); v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
}

// 8 consecutive, fully in-range, 16-byte-aligned outputs of one row.  Bias and the raw residual / activation-source words
// were loaded by the caller (batched over all of a thread's chunks, so their latency overlaps).
__device__ __forceinline__ void epilogue8_vec(const GemmP& p, float v[8], size_t off, const float bia[8], u32x4 rraw, u32x4 draw) {
  if (p.alpha != 1.0f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] += bia[j];
  if (p.preact) store8bf(p.preact + off, v);
  if (p.dact_src) {
    float d[8];
    unpack8bf(draw, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= act_bwd(p.act, d[j]);
  } else if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_fwd(p.act, v[j]);
  }
  if (p.residual) {
    float d[8];
    unpack8bf(rraw, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += d[j];
  }
  if (p.debug & 1) {
    if (v[0] == 12345.678f) ((bf16_t*)p.out)[off] = 0;
  } else if (p.out_mode == 0) {
    store8bf((bf16_t*)p.out + off, v);
  } else if (p.out_mode == 1) {
    float* o = (float*)p.out + off;
    *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
    *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else {
    float* o = (float*)p.out + off;
    f32x4 a0 = *(f32x4*)o, a1 = *(f32x4*)(o + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { a0[j] += v[j]; a1[j] += v[4 + j]; }
    *(f32x4*)o = a0; *(f32x4*)(o + 4) = a1;
  }
}

__device__ __forceinline__ void epilogue8(const GemmP& p, float v[8], int m, int n, bool add_bias, bool vec_ok) {
  const size_t off = (size_t)m * p.ldo + n;
  const int nv = min(8, p.N - n);
  const bool vec = vec_ok && nv == 8;
  if (p.alpha != 1.0f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
  }
  if (p.bias && add_bias) {
    if (nv == 8) {
      const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);   // arena slots are 16-byte aligned, n % 8 == 0
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
    } else {
      for (int j = 0; j < nv; ++j) v[j] += p.bias[n + j];
    }
  }
  if (p.preact) {
    if (vec) store8bf(p.preact + off, v);
    else for (int j = 0; j < nv; ++j) p.preact[off + j] = f2bf(v[j]);
  }
  if (p.dact_src) {
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (vec) load8bf(p.dact_src + off, d);
    else for (int j = 0; j < nv; ++j) d[j] = bf2f(p.dact_src[off + j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= act_bwd(p.act, d[j]);
  } else if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_fwd(p.act, v[j]);
  }
  if (p.residual) {
    float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (vec) load8bf(p.residual + off, d);
    else for (int j = 0; j < nv; ++j) d[j] = bf2f(p.residual[off + j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += d[j];
  }
  if (p.debug & 1) {
    if (v[0] == 12345.678f) ((bf16_t*)p.out)[off] = 0;      // keeps the values live, never true in practice
  } else if (p.out_mode == 0) {
    bf16_t* o = (bf16_t*)p.out + off;
    if (vec) store8bf(o, v);
    else for (int j = 0; j < nv; ++j) o[j] = f2bf(v[j]);
  } else if (p.out_mode == 1) {
    float* o = (float*)p.out + off;
    if (nv == 8 && (p.ldo & 3) == 0) {
      *(f32x4*)o = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      for (int j = 0; j < nv; ++j) o[j] = v[j];
    }
  } else {
    // accumulate, single K-split: no other workgroup of this launch touches these outputs -> vector read-modify-write
    float* o = (float*)p.out + off;
    if (nv == 8 && (p.ldo & 3) == 0) {
      f32x4 a0 = *(f32x4*)o, a1 = *(f32x4*)(o + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a0[j] += v[j]; a1[j] += v[4 + j]; }
      *(f32x4*)o = a0; *(f32x4*)(o + 4) = a1;
    } else {
      for (int j = 0; j < nv; ++j) o[j] += v[j];
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE, bool XKS, bool WKS>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_kernel(GemmP p) {
  constexpr int NWAVE = WM * WN, NTHR = 64 * NWAVE;
  constexpr int TM = BM / WM, TN = BN / WN;             // per-wave output sub-tile
  constexpr int FA = TM / 16, FB = TN / 16;             // MFMA tiles per wave along m / n
  constexpr int XT = BM * BKT * 2, WT = BN * BKT * 2, STAGE = XT + WT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.debug & 8) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;

  // XCD-aware work order.  Workgroup L of the 1-D grid runs on XCD L % 8, each with its own 4 MiB L2.
  //  * split-K launches (weight gradients): split z = L % nsplit, so one XCD (or nsplit/8 .. 8/nsplit of them) owns a whole
  //    K-slice and every row of dY / X in it is fetched once — all tiles of a slice run concurrently on that XCD
  //    (+2 % on the wgrad GEMMs; with the splits in blockIdx.z every XCD touched every K-slice).
  //  * otherwise each XCD gets a contiguous run of tiles (bijective for any tile count), n fastest so neighbours reuse the
  //    same X panel in their L2.  (Walking 4-column strips inside a run — an 8 x 4 block of tiles in flight instead of
  //    2 x 16 — measured no gain: the 256 MiB memory-side cache already absorbs the W re-reads.)
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x, zsplit = 0;
  if (p.split_k > 1) {
    zsplit = bid % p.split_k;
    bid = bid / p.split_k;
  } else {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = zsplit * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntile = (p.debug & 2) ? 0 : (kend - kbeg + BKT - 1) / BKT;

  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // K-strided operands may read up to round_up(R, 8) columns of a row (host guarantees ld covers it)
  const int RX = XKS ? ((p.M + 7) & ~7) : p.M;
  const int RW = WKS ? ((p.N + 7) & ~7) : p.N;

  f32x4 acc[FB][FA];
#pragma unroll
  for (int b = 0; b < FB; ++b)
#pragma unroll
    for (int a = 0; a < FA; ++a) acc[b][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS: NSTAGE-deep ring, stage s at smem + s*STAGE = {X tile, W tile}.  Tiles t+1 .. t+NSTAGE-1 are in flight
  // while tile t is multiplied; one barrier per K-tile.  Loads are issued unconditionally (a tile past kend is
  // all out-of-range -> zero fill, never read) so the counted vmcnt below is a compile-time constant.
  constexpr int LOADS = (BM + BN) * BKT * 2 / 1024 / NWAVE;   // LDS-DMA instructions per wave per K-tile
  static_assert(LOADS * (NSTAGE - 2) <= 63, "vmcnt immediate");
  constexpr bool PIPED = PIPE && BKT == 64 && NSTAGE == 2;
  constexpr int PRO = PIPED ? 2 : NSTAGE - 1;            // tiles staged before the loop
#pragma unroll
  for (int s = 0; s < PRO; ++s) {
    stage_tile<XKS, BM, BKT, NWAVE>(xs, smem + s * STAGE, m0, RX, kbeg + s * BKT, kend, p.ldx, wave, lane);
    stage_tile<WKS, BN, BKT, NWAVE>(ws, smem + s * STAGE + XT, n0, RW, kbeg + s * BKT, kend, p.ldw, wave, lane);
  }
  wait_vm_lgkm<LOADS * (PRO - 1)>();                     // tile 0 landed
  __builtin_amdgcn_s_barrier();
  int cur = 0, nxt = NSTAGE - 1;
  if constexpr (PIPE && BKT == 32 && NSTAGE == 4) {
    // Ping-pong K loop (8 waves, two per SIMD; 32-wide K stages in a 4-slot ring).  The waves of a workgroup form two
    // groups, A = waves 0..3 and B = waves 4..7 (SIMD partners), that run the same sequence one interval apart:
    //     A:  MEM(0) | COMP(0) | MEM(1) | COMP(1) | ...
    //     B:    -    | MEM(0)  | COMP(0)| MEM(1)  | ...          ('|' = workgroup barrier)
    // MEM(h)  = 12 ds_read_b128 (the fragments of stage h) + this wave's 4 LDS-DMA pieces of stage h+3 + counted wait,
    // COMP(h) = 32 back-to-back MFMAs.  In every interval one wave per SIMD owns the matrix pipe while its partner owns the
    // LDS / vector-memory issue ports, so neither the fragment reads nor the ~60-100-cycle issue cost of an LDS-DMA piece
    // ever stalls the MFMA stream (measured before: MFMA-only loop 1.0 us per 64-K, +0.27 us for the LDS reads, +0.34 us
    // for the DMA issue when both partners do the same thing at the same time).
    // Ring safety: stage j is read by A in interval 2j and by B in interval 2j+1; MEM(j+1) (intervals 2j+2 / 2j+3) refills
    // its slot with stage j+4.  A wave leaves MEM(h) only when its own pieces of stage h+1 have landed (vmcnt(8): stages
    // h+2, h+3 may fly), and a barrier separates that from every later reader.
    static_assert(NWAVE == 8 && LOADS >= 2, "ping-pong loop is written for 8 waves (two per SIMD)");
    // (Issuing part of the DMA pieces in the middle of COMP(h) instead, or staggering the partners' issue points in a 64-wide
    // pipelined loop, measured the same within noise: profiles/r1_gemm_overhead.txt.)
    const bool grp_b = wave >= NWAVE / 2;
    const int nst = ntile;
    bf16x8 xf[FA], wf[FB];
    if (grp_b) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int rd = 0, wr = 3;
    for (int h = 0; h < nst; ++h) {
      const char* xr = smem + rd * STAGE;
      char* xw = smem + wr * STAGE;
      if (!(p.debug & 128) || h == 0) {
#pragma unroll
        for (int a = 0; a < FA; ++a) xf[a] = read_frag<XKS, BM, BKT>(xr, wm * TM + a * 16, 0, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf[b] = read_frag<WKS, BN, BKT>(xr + XT, wn * TN + b * 16, 0, lane);
      }
      if (!(p.debug & 32)) {
        stage_tile<XKS, BM, BKT, NWAVE>(xs, xw, m0, RX, kbeg + (h + 3) * BKT, kend, p.ldx, wave, lane);
        stage_tile<WKS, BN, BKT, NWAVE>(ws, xw + XT, n0, RW, kbeg + (h + 3) * BKT, kend, p.ldw, wave, lane);
      }
      // stage h+1 must have landed (this wave's pieces); newer ones may fly
      wait_vm_lgkm<2 * LOADS>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rd = (rd + 1) & 3;
      wr = (wr + 1) & 3;
    }
    if (!grp_b) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  } else if constexpr (PIPED) {
    // Software-pipelined K loop, two LDS slots.  Per tile t:
    //   read frags (t, kk=1) | MFMA (t, kk=0) | wait tile t+1 landed + BARRIER | issue LDS-DMA of tile t+2 into slot t |
    //   read frags (t+1, kk=0) | MFMA (t, kk=1)
    // - the barrier sits in the middle of the tile's MFMA work: after it the LDS reads of the next tile (and the barrier
    //   skew of the 8 waves) are covered by the 32..64 MFMAs of (t, kk=1);
    // - at the barrier every fragment of tile t is already in registers, so slot t is free: the loads of tile t+2 are
    //   issued right behind it and have a FULL iteration to land (a 64 KiB tile needs ~0.9 us at the per-CU L2->LDS rate
    //   plus latency; a load issued half an iteration before its wait stalls the whole workgroup);
    bf16x8 xf0[FA], wf0[FB], xf1[FA], wf1[FB];
#pragma unroll
    for (int a = 0; a < FA; ++a) xf0[a] = read_frag<XKS, BM, BKT>(smem, wm * TM + a * 16, 0, lane);
#pragma unroll
    for (int b = 0; b < FB; ++b) wf0[b] = read_frag<WKS, BN, BKT>(smem + XT, wn * TN + b * 16, 0, lane);
    for (int t = 0; t < ntile; ++t) {
      char* xcur = smem + cur * STAGE;
      char* wcur = xcur + XT;
#pragma unroll
      for (int a = 0; a < FA; ++a) xf1[a] = read_frag<XKS, BM, BKT>(xcur, wm * TM + a * 16, 1, lane);
#pragma unroll
      for (int b = 0; b < FB; ++b) wf1[b] = read_frag<WKS, BN, BKT>(wcur, wn * TN + b * 16, 1, lane);
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[b], xf0[a], acc[b][a], 0, 0, 0);
      // tile t+1 landed (all of this wave's loads), every LDS read of slot t retired
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage_tile<XKS, BM, BKT, NWAVE>(xs, xcur, m0, RX, kbeg + (t + 2) * BKT, kend, p.ldx, wave, lane);
      stage_tile<WKS, BN, BKT, NWAVE>(ws, wcur, n0, RW, kbeg + (t + 2) * BKT, kend, p.ldw, wave, lane);
      cur ^= 1;
      {
        const char* xn = smem + cur * STAGE;           // tile t+1 (zero-filled past the end: harmless)
#pragma unroll
        for (int a = 0; a < FA; ++a) xf0[a] = read_frag<XKS, BM, BKT>(xn, wm * TM + a * 16, 0, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf0[b] = read_frag<WKS, BN, BKT>(xn + XT, wn * TN + b * 16, 0, lane);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[b], xf1[a], acc[b][a], 0, 0, 0);
    }
  } else {
  bf16x8 xf[FA], wf[FB];
  for (int t = 0; t < ntile; ++t) {
    char* xcur = smem + cur * STAGE;
    char* wcur = xcur + XT;
    if (!(p.debug & 32)) {
      // ring slot `nxt` was last read in iteration t-1 and every wave has passed that iteration's barrier
      char* xnext = smem + nxt * STAGE;
      stage_tile<XKS, BM, BKT, NWAVE>(xs, xnext, m0, RX, kbeg + (t + NSTAGE - 1) * BKT, kend, p.ldx, wave, lane);
      stage_tile<WKS, BN, BKT, NWAVE>(ws, xnext + XT, n0, RW, kbeg + (t + NSTAGE - 1) * BKT, kend, p.ldw, wave, lane);
    }
    if (!(p.debug & 64))
#pragma unroll
    for (int kk = 0; kk < BKT / 32; ++kk) {
      if (!(p.debug & 128) || t == 0) {
#pragma unroll
        for (int a = 0; a < FA; ++a) xf[a] = read_frag<XKS, BM, BKT>(xcur, wm * TM + a * 16, kk, lane);
#pragma unroll
        for (int b = 0; b < FB; ++b) wf[b] = read_frag<WKS, BN, BKT>(wcur, wn * TN + b * 16, kk, lane);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int a = 0; a < FA; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
    }
    // tile t+1 landed (this wave's loads; newer tiles may still fly), LDS reads of this slot retired; then everyone's
    wait_vm_lgkm<LOADS * (NSTAGE - 2)>();
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the (zero-fill) tail loads before LDS is reused
  __builtin_amdgcn_s_barrier();

  if (p.debug & 4) { if (acc[0][0][0] == 12345.678f) ((float*)p.out)[0] = 0.f; return; }
  // ---- direct epilogue for the common plain case (bf16 out, bias + activation only): each lane owns 4 consecutive n of
  // one m per accumulator tile -> bias as one 16-byte load, pack with v_cvt_pk_bf16_f32, one 8-byte store.  No LDS
  // round trip, no barriers; the 32-byte row pieces of the four n-groups are merged by the L2.
  if (BM <= 128 && p.out_mode == 0 && !p.preact && !p.dact_src && !p.residual && (p.ldo & 3) == 0 && !(p.debug & 16)) {
    const int lm_ = lane & 15, ln4_ = (lane >> 4) * 4;
    const bool add_bias_ = p.bias != nullptr;
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      const int m = m0 + wm * TM + a * 16 + lm_;
      if (m >= p.M) continue;
      bf16_t* orow = (bf16_t*)p.out + (size_t)m * p.ldo;
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int n = n0 + wn * TN + b * 16 + ln4_;
        if (n + 4 <= p.N) {
          f32x4 v = acc[b][a];
          if (p.alpha != 1.0f) v *= p.alpha;
          if (add_bias_) v += *(const f32x4*)(p.bias + n);
          if (p.act != VACNIC_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_fwd(p.act, v[j]);
          }
          *(u32x2*)(orow + n) = (u32x2){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        } else if (n < p.N) {
          for (int j = 0; j < 4 && n + j < p.N; ++j) {
            float v = acc[b][a][j] * p.alpha + (add_bias_ ? p.bias[n + j] : 0.f);
            orow[n + j] = f2bf(act_fwd(p.act, v));
          }
        }
      }
    }
    return;
  }
  // ---- epilogue: stage the fp32 C tile through LDS (the operand buffers are free now) in 64-row
  // passes, then every thread handles 8 consecutive n of one row: 16-byte coalesced traffic for the
  // output, the saved pre-activation, the activation-backward source and the residual.
  const int lm = lane & 15, ln4 = (lane >> 4) * 4;
  float* sc = (float*)smem;                 // [64][CLD] f32
  constexpr int CLD = BN + 4;
  constexpr int PASSES = (BM + 63) / 64;
  const bool vec_ok = (p.ldo & 7) == 0;
  const bool add_bias = zsplit == 0;
  float bia[8];
  // A pass stages 64 tile rows: RPWM = 64/WM rows from EACH wave row-group, so that every wave deposits in every pass (the
  // LDS store path has two halves, SIMDs {0,1} and {2,3}; a pass fed by the waves of one wm only ran it at half rate).
  // LDS row r of pass p holds tile row (r / RPWM) * TM + p * RPWM + r % RPWM.
  constexpr int RPWM = 64 / WM;
  static_assert((RPWM % 16 == 0 && TM % RPWM == 0) || BM < 64, "epilogue pass mapping");
  const bool old_map = (p.debug & 256) != 0;      // A/B: one wave row-group per pass (the previous mapping)
  auto tile_row = [&](int pass, int r) { return old_map ? pass * 64 + r : (r / RPWM) * TM + pass * RPWM + r % RPWM; };
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
    for (int a = 0; a < FA; ++a) {
      if (!old_map && (a * 16) / RPWM == pass) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
          *(f32x4*)(sc + (wm * RPWM + (a * 16) % RPWM + lm) * CLD + wn * TN + b * 16 + ln4) = acc[b][a];
      }
      if (old_map && (wm * TM + a * 16) / 64 == pass) {
#pragma unroll
        for (int b = 0; b < FB; ++b)
          *(f32x4*)(sc + (((wm * TM + a * 16) & 63) + lm) * CLD + wn * TN + b * 16 + ln4) = acc[b][a];
      }
    }
    __syncthreads();
    if (p.out_mode == 2 && p.split_k > 1) {
      // split-K accumulate: f32 atomics shaped as 256 contiguous bytes per wave-instruction (one row, 64
      // consecutive columns) — the shape the memory-side atomic units run at full rate on
      constexpr int RPW = 64 / NWAVE;
#pragma unroll 1
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        const int m = m0 + tile_row(pass, row);
        if (m >= p.M) continue;
#pragma unroll
        for (int h = 0; h < BN / 64; ++h) {
          const int n = n0 + h * 64 + lane;
          if (n < p.N) {
            float v = sc[row * CLD + h * 64 + lane] * p.alpha;
            if (p.bias && add_bias) v += p.bias[n];
            atomicAdd((float*)p.out + (size_t)m * p.ldo + n, v);
          }
        }
      }
      __syncthreads();
      continue;
    }
    // every thread owns ONE 8-wide column group (NTHR is a multiple of the chunks per row) and RPT rows of the pass: the LDS
    // reads and the residual / activation-source loads of all RPT chunks are issued before any of them is consumed, and the
    // bias is loaded once per kernel — a chunk-at-a-time loop exposed one L2 round trip per chunk (7-9 us per tile).
    constexpr int CPR = BN / 8;               // 8-wide chunks per row
    constexpr int RPT = 64 * CPR / NTHR;      // chunks per thread per pass
    constexpr int RSTEP = NTHR / CPR;
    static_assert((64 * CPR) % NTHR == 0 && NTHR % CPR == 0, "epilogue chunk mapping");
    const int c8 = (tid % CPR) * 8, rbase = tid / CPR;
    const int n = n0 + c8;
    const bool fast = vec_ok && n + 8 <= p.N && (p.out_mode == 0 || (p.ldo & 3) == 0);
    if (fast) {
      if (pass == 0) {
        if (p.bias && add_bias) {
          const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { bia[j] = b0[j]; bia[4 + j] = b1[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) bia[j] = 0.f;
        }
      }
      float v[RPT][8];
      u32x4 rraw[RPT], draw[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int row = rbase + k * RSTEP;
        const int m = m0 + tile_row(pass, row);
        const f32x4 v0 = *(const f32x4*)(sc + row * CLD + c8), v1 = *(const f32x4*)(sc + row * CLD + c8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[k][j] = v0[j]; v[k][4 + j] = v1[j]; }
        rraw[k] = (u32x4){0, 0, 0, 0}; draw[k] = (u32x4){0, 0, 0, 0};
        if (m < p.M) {
          const size_t off = (size_t)m * p.ldo + n;
          if (p.residual) rraw[k] = *(const u32x4*)(p.residual + off);
          if (p.dact_src) draw[k] = *(const u32x4*)(p.dact_src + off);
        }
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int m = m0 + tile_row(pass, rbase + k * RSTEP);
        if (m < p.M) epilogue8_vec(p, v[k], (size_t)m * p.ldo + n, bia, rraw[k], draw[k]);
      }
    } else if (n < p.N) {
#pragma unroll 1
      for (int k = 0; k < RPT; ++k) {
        const int row = rbase + k * RSTEP;
        const int m = m0 + tile_row(pass, row);
        if (m < p.M) {
          float v[8];
          const f32x4 v0 = *(const f32x4*)(sc + row * CLD + c8), v1 = *(const f32x4*)(sc + row * CLD + c8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
          epilogue8(p, v, m, n, add_bias, vec_ok);
        }
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Skinny-M GEMM (M <= 8): the single-token decoder of caption generation (MFULL:474-501, beams*batch rows) and any other
// GEMM whose X is a handful of rows.  HBM-bound: W [N,K] is streamed exactly once; X (<= 8 x K bf16) is re-read from
// L1/L2.  One wave owns CW consecutive output columns; its 64 lanes split K in 16-byte chunks, accumulate MR x CW partial
// dot products in fp32 and meet in a wave reduction.  No LDS, no barriers.
template <int MR, int CW, int KW>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(GemmP p) {
  // one workgroup per CW output columns: N / CW independent workgroups spread over every CU (N = 1024 -> 256), each limited
  // only by one round trip of its loads.  KW waves split K (long reductions, e.g. fc2 with K = 4096) and meet through LDS.
  const int lane = threadIdx.x & 63;
  const int kw = KW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * CW;
  constexpr int NV = MR * CW;                        // 32 partial dot products per lane
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;
  const int nchunk = p.K >> 3;                       // host guarantees K % 8 == 0
#pragma unroll 4
  for (int ch = lane + 64 * kw; ch < nchunk; ch += 64 * KW) {
    float wv[CW][8];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      if (n0 + c < p.N) load8bf(p.w + (size_t)(n0 + c) * p.ldw + ch * 8, wv[c]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[c][j] = 0.f;
      }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < p.M) {
        float xv[8];
        load8bf(p.x + (size_t)m * p.ldx + ch * 8, xv);
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m * CW + c] += xv[j] * wv[c][j];
      }
    }
  }
  // reduce-scatter butterfly: at offset 32,16,8,4,2 every lane hands the half of its values its partner keeps and adds the
  // half it keeps (16+8+4+2+1 exchanges), then one last exchange at offset 1 -> 32 cross-lane moves for 32 sums instead of 192.
  static_assert(NV == 32, "butterfly written for 32 values per lane");
#define VAC_BFLY(OFF, HALF)                                                      \
  {                                                                              \
    const bool up = (lane & OFF) != 0;                                           \
    _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                           \
      const float send = up ? acc[i] : acc[HALF + i];                            \
      const float keep = up ? acc[HALF + i] : acc[i];                            \
      acc[i] = keep + __shfl_xor(send, OFF, 64);                                 \
    }                                                                            \
  }
  VAC_BFLY(32, 16) VAC_BFLY(16, 8) VAC_BFLY(8, 4) VAC_BFLY(4, 2) VAC_BFLY(2, 1)
#undef VAC_BFLY
  float v = acc[0] + __shfl_xor(acc[0], 1, 64);
  const int idx = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | ((lane >> 1) & 1);
  if constexpr (KW > 1) {
    __shared__ float part[KW][NV];
    if ((lane & 1) == 0) part[kw][idx] = v;
    __syncthreads();
    if (kw != 0) return;
    v = 0.f;
#pragma unroll
    for (int w = 0; w < KW; ++w) v += part[w][idx];
  }
  if ((lane & 1) == 0) {
    const int m = idx / CW, c = idx % CW;
    const int n = n0 + c;
    if (m < p.M && n < p.N) {
      v *= p.alpha;
      if (p.bias) v += p.bias[n];
      v = act_fwd(p.act, v);
      const size_t off = (size_t)m * p.ldo + n;
      if (p.out_mode == 0) ((bf16_t*)p.out)[off] = f2bf(v);
      else if (p.out_mode == 1) ((float*)p.out)[off] = v;
      else ((float*)p.out)[off] += v;
    }
  }
}

template <int BM, int BN, int WM, int WN, int BKT, int NSTAGE, bool PIPE = false>
int launch_gemm(const GemmP& p0, bool xks, bool wks, int zsplits, hipStream_t s) {
  GemmP p = p0;
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n * zsplits), block(64 * WM * WN);
  constexpr size_t lds = NSTAGE * (BM + BN) * BKT * 2;
  static_assert(lds >= 64 * (BN + 4) * 4, "epilogue staging must fit in the operand buffers");
#define VAC_LAUNCH(XK, WK)                                                                            \
  do {                                                                                                \
    auto kern = gemm_kernel<BM, BN, WM, WN, BKT, NSTAGE, PIPE, XK, WK>;                                                  \
    if (lds > 65536) {                                                                                \
      static bool once = false;                                                                       \
      if (!once) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); once = true; } \
    }                                                                                                 \
    hipLaunchKernelGGL(kern, grid, block, lds, s, p);                                                 \
  } while (0)
  if (!xks && !wks) VAC_LAUNCH(false, false);
  else if (!xks && wks) VAC_LAUNCH(false, true);
  else if (xks && wks) VAC_LAUNCH(true, true);
  else VAC_LAUNCH(true, false);
#undef VAC_LAUNCH
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

}  // namespace

// ---- host-side tile selection --------------------------------------------------------------------------------
// Cost model in "MFMA work units" (bm*bn*k elements): a launch needs ceil(tiles / resident slots) rounds; a round
// costs the time of one tile with the CU shared by `per_cu` co-resident workgroups, plus a fixed ~10 us of launch /
// prologue / epilogue (measured: profiles/r1_gemm_overhead.txt), expressed in the same units.
struct TileCfg { int bm, bn, slots, per_cu; double eff; int hint; };
static const TileCfg kCfgs[3] = {
    {256, 256, 256, 1, 1.00, 256},     // 8 waves, 128 KiB LDS, one workgroup per CU
    {128, 128, 512, 2, 0.80, 128},     // 4 waves, 64 KiB LDS, two per CU
    {64, 128, 256, 1, 0.42, 64},       // 4 waves, 4-deep ring (96 KiB): latency-bound small problems
};
static const double kFixedUnits = 27e6;

static double cfg_cost(const TileCfg& c, int64_t M, int64_t N, int64_t kper, int64_t z) {
  const int64_t tiles = ((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * z;
  const int64_t rounds = (tiles + c.slots - 1) / c.slots;
  return (double)rounds * ((double)c.per_cu * c.bm * c.bn * (double)kper / c.eff + kFixedUnits);
}

static int gemm_one(const vacnic_gemm_args* a, int hint, void* stream);

extern "C" int vacnic_gemm_bf16(const vacnic_gemm_args* a, void* stream) {
  VCHECK(a && a->x && a->w && a->out, VACNIC_BAD_SHAPE, "gemm: null operand");
  VCHECK(a->M > 0 && a->N > 0 && a->K > 0, VACNIC_BAD_SHAPE, "gemm: empty problem M=%ld N=%ld K=%ld",
         (long)a->M, (long)a->N, (long)a->K);
  if (a->tile_hint != 0) return gemm_one(a, a->tile_hint, stream);
  if (a->M <= 8 && !a->x_kstrided && !a->w_kstrided && a->split_k <= 1 && !a->preact && !a->dact_src && !a->residual &&
      (a->K & 7) == 0)
    return gemm_one(a, 8, stream);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  int64_t kper = (a->K + split - 1) / split;
  kper = (kper + BK - 1) / BK * BK;
  const int64_t z = (a->K + kper - 1) / kper;
  // best single launch
  int best = 0; double best_cost = 1e300;
  for (int i = 0; i < 3; ++i) {
    if (i == 0 && (a->M < 256 || a->N < 256)) continue;
    const double c = cfg_cost(kCfgs[i], a->M, a->N, kper, z);
    if (c < best_cost) { best_cost = c; best = i; }
  }
  // M-split: rows are independent, so a small row remainder that would open an extra (almost empty) round is cut off
  // and run as its own small launch (e.g. the ViT's M = 32*257 = 8224 = 8192 + 32)
  int split_cfg = -1; int64_t m_main = 0; double split_cost = best_cost;
  for (int i = 0; i < 2; ++i) {
    const int64_t rem = a->M % kCfgs[i].bm;
    if (rem == 0 || a->M - rem < kCfgs[i].bm || (i == 0 && a->N < 256)) continue;
    const double c = cfg_cost(kCfgs[i], a->M - rem, a->N, kper, z) + cfg_cost(kCfgs[2], rem, a->N, kper, z);
    if (c < 0.85 * split_cost) { split_cost = c; split_cfg = i; m_main = a->M - rem; }
  }
  if (split_cfg < 0) return gemm_one(a, kCfgs[best].hint, stream);
  vacnic_gemm_args main_a = *a, tail_a = *a;
  main_a.M = m_main;
  tail_a.M = a->M - m_main;
  const int64_t xoff = a->x_kstrided ? m_main : m_main * a->ldx;
  tail_a.x = (const char*)a->x + xoff * 2;
  const int64_t osz = a->out_mode == 0 ? 2 : 4;
  tail_a.out = (char*)a->out + m_main * a->ldo * osz;
  if (a->preact) tail_a.preact = (char*)a->preact + m_main * a->ldo * 2;
  if (a->dact_src) tail_a.dact_src = (const char*)a->dact_src + m_main * a->ldo * 2;
  if (a->residual) tail_a.residual = (const char*)a->residual + m_main * a->ldo * 2;
  if (int e = gemm_one(&main_a, kCfgs[split_cfg].hint, stream)) return e;
  return gemm_one(&tail_a, kCfgs[2].hint, stream);
}

static int gemm_one(const vacnic_gemm_args* a, int tile_hint, void* stream) {
  VCHECK((a->ldx & 7) == 0 && (a->ldw & 7) == 0, VACNIC_MISALIGNED, "gemm: ldx/ldw must be multiples of 8");
  VCHECK(aligned16(a->x) && aligned16(a->w), VACNIC_MISALIGNED, "gemm: x/w must be 16-byte aligned");
  VCHECK(a->out_mode >= 0 && a->out_mode <= 2, VACNIC_BAD_DTYPE, "gemm: bad out_mode %d", a->out_mode);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  VCHECK(split == 1 || a->out_mode == 2, VACNIC_UNSUPPORTED, "gemm: split_k needs out_mode 2");
  VCHECK(a->bias == nullptr || aligned16(a->bias), VACNIC_MISALIGNED, "gemm: bias must be 16-byte aligned");
  VCHECK(a->ldo >= a->N, VACNIC_BAD_SHAPE, "gemm: ldo < N");
  if (a->x_kstrided) VCHECK(a->ldx >= ((a->M + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldx too small for K-strided X");
  else VCHECK(a->ldx >= ((a->K + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldx < round_up(K, 8) for K-contiguous X");
  if (a->w_kstrided) VCHECK(a->ldw >= ((a->N + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldw too small for K-strided W");
  else VCHECK(a->ldw >= ((a->K + 7) & ~7LL), VACNIC_BAD_SHAPE, "gemm: ldw < round_up(K, 8) for K-contiguous W");

  auto span = [](int64_t rows, int64_t cols, int64_t ld) { return ((rows - 1) * ld + cols) * 2; };
  const int64_t K8 = (a->K + 7) & ~7LL;
  const int64_t xb = a->x_kstrided ? span(a->K, (a->M + 7) & ~7LL, a->ldx) : span(a->M, K8, a->ldx);
  const int64_t wb = a->w_kstrided ? span(a->K, (a->N + 7) & ~7LL, a->ldw) : span(a->N, K8, a->ldw);
  VCHECK(xb < 0x7ffffff0LL && wb < 0x7ffffff0LL, VACNIC_UNSUPPORTED, "gemm: operand larger than 2 GiB");

  GemmP p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = a->bias;
  p.out = a->out; p.preact = (bf16_t*)a->preact; p.dact_src = (const bf16_t*)a->dact_src;
  p.residual = (const bf16_t*)a->residual;
  p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
  p.ldx = (int)a->ldx; p.ldw = (int)a->ldw; p.ldo = (int)a->ldo;
  p.act = a->act; p.out_mode = a->out_mode; p.split_k = split;
  int kps = (int)((a->K + split - 1) / split);
  kps = (kps + BK - 1) / BK * BK;               // multiple of 64: valid for both K-tile depths
  p.k_per_split = kps;
  const int zsplits = (int)((a->K + kps - 1) / kps);
  p.split_k = zsplits;
  p.alpha = a->alpha;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  p.tiles_m = p.tiles_n = 0;
  hipStream_t s = (hipStream_t)stream;
  if (tile_hint == 8) {     // skinny-M kernel (chosen by vacnic_gemm_bf16 for M <= 8, or forced by the caller)
    VCHECK(a->M <= 8 && !a->x_kstrided && !a->w_kstrided && zsplits == 1 && !a->preact && !a->dact_src && !a->residual &&
           (a->K & 7) == 0, VACNIC_UNSUPPORTED, "gemm: the skinny kernel needs M <= 8, K-contiguous operands, K %% 8 == 0 and a plain epilogue");
    constexpr int CW = 4;
    dim3 grid((unsigned)((a->N + CW - 1) / CW));
    if (a->K >= 2048 && a->N <= 8192) hipLaunchKernelGGL((gemm_skinny_kernel<8, CW, 4>), grid, dim3(256), 0, s, p);   // long reduction, few columns: 4 waves split K
    else hipLaunchKernelGGL((gemm_skinny_kernel<8, CW, 1>), grid, dim3(64), 0, s, p);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
  const int force = tile_hint % 1000;
  p.debug = tile_hint / 1000;
  const bool big = force == 256;
  const bool mid = force == 128;
  // A/B baselines kept for the ablations in profiles/: plain K loops and the 64-wide software-pipelined loop
  if (force == 260) return launch_gemm<256, 256, 2, 4, 64, 2, false>(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 261) return launch_gemm<128, 128, 2, 2, 64, 2, false>(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 262) return launch_gemm<256, 256, 2, 4, 64, 2, true>(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  if (force == 264) return launch_gemm<256, 128, 2, 4, 32, 4, true>(p, a->x_kstrided, a->w_kstrided, zsplits, s);   // ping-pong, half-width tile
  if (big) return launch_gemm<256, 256, 2, 4, 32, 4, true>(p, a->x_kstrided, a->w_kstrided, zsplits, s);   // ping-pong loop
  if (mid) return launch_gemm<128, 128, 2, 2, 64, 2, true>(p, a->x_kstrided, a->w_kstrided, zsplits, s);
  return launch_gemm<64, 128, 2, 2, 64, 4>(p, a->x_kstrided, a->w_kstrided, zsplits, s);
}
