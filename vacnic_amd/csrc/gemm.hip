// bf16 MFMA GEMM for gfx950:  out[m][n] = epi(alpha * sum_k X(m,k) W(n,k) + bias[n]).
//
// Replaces every nn.Linear on the VACNIC path (q/k/v/out_proj MFULL:449-452, fc1/fc2 MFULL:581-582,
// _linear_1up/down :587-588, _face_up/down :607-608, ner_map_up/down :594-595, prompt_mlp :1136,
// visual_map :1144, lm_head :1885) and their dgrad / wgrad.
//
// Design (CDNA4): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Operands go HBM -> LDS with `buffer_load_dwordx4 ... lds`
// (LDS-DMA, no VGPR round trip; the SRD range check zero-fills M/N/K edges).  The LDS image is
// lane-linear per wave-instruction, so the bank-conflict XOR swizzle is applied to the per-lane
// SOURCE address and again on the fragment read (guide rule 21).  An operand whose reduction index
// is the strided one in memory (dgrad's W, wgrad's dY and X) is staged as [k][row] and read with
// ds_read_b64_tr_b16 (hardware transpose), so no transposed copies are ever materialised.
// MFMA A <- W rows, B <- X rows, i.e. the wave computes the C^T tile: each lane then owns 4
// consecutive n of one m and stores 8/16 contiguous bytes.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;      // 16 KiB, both layouts
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
};

// f(k) of the K-strided swizzle: distinct for the 8 k-rows one tr-read half touches.
__device__ __forceinline__ int fk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Issue the 4 LDS-DMA loads of this thread for one operand tile.
//   KS=false: tile [128 rows][64 k], 128-B rows, chunk' = chunk ^ (row & 7)
//   KS=true : tile [64 k][128 rows], 256-B rows, chunk' = chunk ^ (fk(k) << 1)
template <bool KS>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int r0, int R,
                                           int k0, int kend, int ld, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = wave * 4 + i;
    int voff;
    if (!KS) {
      const int row = blk * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (row & 7);
      const int gr = r0 + row, gk = k0 + lc * 8;
      voff = (gr < R && gk < kend) ? (int)(((unsigned)gr * (unsigned)ld + (unsigned)gk) * 2u) : OOB;
    } else {
      const int k = blk * 4 + (lane >> 4);
      const int lc = (lane & 15) ^ (fk(k) << 1);
      const int gk = k0 + k, gr = r0 + lc * 8;
      voff = (gk < kend && gr < R) ? (int)(((unsigned)gk * (unsigned)ld + (unsigned)gr) * 2u) : OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + blk * 1024), 16, voff, 0, 0, 0);
  }
}

// Fragment for MFMA 16x16x32: lane l gets element (row = rbase + (l&15), k = kk*32 + 8*(l>>4) + j), j=0..7.
template <bool KS>
__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int rbase, int kk, int lane) {
  if (!KS) {
    const int row = rbase + (lane & 15);
    const int chunk = kk * 4 + (lane >> 4);
    return *(const bf16x8*)(lds_tile + row * 128 + ((chunk ^ (row & 7)) << 4));
  } else {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int chunk = (rbase >> 3) + (p >> 1);
    bf16x8 r;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int k = kk * 32 + 8 * g + 4 * hh + q;
      const int phys = chunk ^ (fk(k) << 1);
      const char* a = lds_tile + k * 256 + phys * 16 + (p & 1) * 8;
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

__device__ __forceinline__ void epilogue8(const GemmP& p, float v[8], int m, int n, bool add_bias, bool vec_ok) {
  const size_t off = (size_t)m * p.ldo + n;
  const int nv = min(8, p.N - n);
  const bool vec = vec_ok && nv == 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] *= p.alpha;
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
  if (p.out_mode == 0) {
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

template <bool XKS, bool WKS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD; give each XCD a contiguous run of tiles
  // (bijective for any tile count), n fastest so neighbours reuse the same X panel in their L2.
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, r = nt & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntile = (kend - kbeg + BK - 1) / BK;

  __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // K-strided operands may read up to round_up(R, 8) columns of a row (host guarantees ld covers it)
  const int RX = XKS ? ((p.M + 7) & ~7) : p.M;
  const int RW = WKS ? ((p.N + 7) & ~7) : p.N;

  f32x4 acc[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[b][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS: buffer b at smem + b*2*TILE_BYTES = {X tile, W tile}
  if (ntile > 0) {
    stage_tile<XKS>(xs, smem, m0, RX, kbeg, kend, p.ldx, wave, lane);
    stage_tile<WKS>(ws, smem + TILE_BYTES, n0, RW, kbeg, kend, p.ldw, wave, lane);
  }
  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    char* xcur = smem + cur * (2 * TILE_BYTES);
    char* wcur = xcur + TILE_BYTES;
    if (t + 1 < ntile) {
      // buffer cur^1 was last read in iteration t-1; every wave has passed that iteration's
      // trailing barrier, so it is free to overwrite.
      char* xnext = smem + (cur ^ 1) * (2 * TILE_BYTES);
      stage_tile<XKS>(xs, xnext, m0, RX, kbeg + (t + 1) * BK, kend, p.ldx, wave, lane);
      stage_tile<WKS>(ws, xnext + TILE_BYTES, n0, RW, kbeg + (t + 1) * BK, kend, p.ldw, wave, lane);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile t landed (this wave's 8 loads)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                         // ... and every other wave's
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xf[a] = read_frag<XKS>(xcur, wm * 64 + a * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) wf[b] = read_frag<WKS>(wcur, wn * 64 + b * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b], xf[a], acc[b][a], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // all reads of buffer cur done
  }

  // ---- epilogue: stage the fp32 C tile through LDS (the operand buffers are free now) in two 64-row
  // passes, then every thread handles 8 consecutive n of one row: 16-byte coalesced traffic for the
  // output, the saved pre-activation, the activation-backward source and the residual.
  const int lm = lane & 15, ln4 = (lane >> 4) * 4;
  float* sc = (float*)smem;                 // [64][CLD] f32 = 33 KiB
  constexpr int CLD = 132;
  const bool vec_ok = (p.ldo & 7) == 0;
  const bool add_bias = blockIdx.z == 0;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (wm == pass) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          *(f32x4*)(sc + (a * 16 + lm) * CLD + wn * 64 + b * 16 + ln4) = acc[b][a];
    }
    __syncthreads();
    if (p.out_mode == 2 && p.split_k > 1) {
      // split-K accumulate: f32 atomics shaped as 256 contiguous bytes per wave-instruction (one row, 64
      // consecutive columns) — the shape the memory-side atomic units run at full rate on
#pragma unroll 1
      for (int rr = 0; rr < 16; ++rr) {
        const int row = wave * 16 + rr;
        const int m = m0 + pass * 64 + row;
        if (m >= p.M) break;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
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
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int chunk = tid + 256 * i;
      const int row = chunk >> 4, c8 = (chunk & 15) * 8;
      const int m = m0 + pass * 64 + row, n = n0 + c8;
      if (m < p.M && n < p.N) {
        float v[8];
        const f32x4 v0 = *(const f32x4*)(sc + row * CLD + c8), v1 = *(const f32x4*)(sc + row * CLD + c8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
        epilogue8(p, v, m, n, add_bias, vec_ok);
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int vacnic_gemm_bf16(const vacnic_gemm_args* a, void* stream) {
  VCHECK(a && a->x && a->w && a->out, VACNIC_BAD_SHAPE, "gemm: null operand");
  VCHECK(a->M > 0 && a->N > 0 && a->K > 0, VACNIC_BAD_SHAPE, "gemm: empty problem M=%ld N=%ld K=%ld",
         (long)a->M, (long)a->N, (long)a->K);
  VCHECK((a->ldx & 7) == 0 && (a->ldw & 7) == 0, VACNIC_MISALIGNED, "gemm: ldx/ldw must be multiples of 8");
  VCHECK(aligned16(a->x) && aligned16(a->w), VACNIC_MISALIGNED, "gemm: x/w must be 16-byte aligned");
  VCHECK(a->out_mode >= 0 && a->out_mode <= 2, VACNIC_BAD_DTYPE, "gemm: bad out_mode %d", a->out_mode);
  const int split = a->split_k < 1 ? 1 : a->split_k;
  VCHECK(split == 1 || a->out_mode == 2, VACNIC_UNSUPPORTED, "gemm: split_k needs out_mode 2");
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
  kps = (kps + BK - 1) / BK * BK;
  p.k_per_split = kps;
  const int zsplits = (int)((a->K + kps - 1) / kps);
  p.alpha = a->alpha;
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
  p.tiles_m = (int)((a->M + BM - 1) / BM); p.tiles_n = (int)((a->N + BN - 1) / BN);
  dim3 grid(p.tiles_m * p.tiles_n, 1, zsplits), block(256);
  const size_t lds = 4 * TILE_BYTES;
  hipStream_t s = (hipStream_t)stream;
  if (!a->x_kstrided && !a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, lds, s, p);
  else if (!a->x_kstrided && a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, lds, s, p);
  else if (a->x_kstrided && a->w_kstrided) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, lds, s, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
