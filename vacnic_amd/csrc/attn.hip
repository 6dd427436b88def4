// Fused attention for gfx950, head_dim 64 — replaces BartAttention.forward's core
// (bmm -> +mask -> softmax(fp32) -> bmm, MFULL:509-548) and nn.MultiheadAttention in the CLIP ViT.
// The [B*H, Tq, Tk] score matrix never touches HBM.
//
// Structure (all three kernels): a workgroup = 4 waves; each wave owns 32 rows of the "outer"
// dimension (queries for fwd / dQ, keys for dK,dV) and all waves sweep 64-row tiles of the other
// one, staged HBM -> LDS by `buffer_load ... lds` (double buffered, zero-filled past the edge).
// MFMA is v_mfma_f32_32x32x16_bf16 with the outer index on the LANE, so softmax state is a
// per-lane scalar and each product's accumulator tile is directly the B operand of the next
// product (no LDS round trip for P / dS).  Operands needed transposed come from the same LDS
// image through ds_read_b64_tr_b16.  One XOR swizzle serves row reads and transposed reads.
//
// Mask semantics follow the reference exactly: masked keys get an ADDITIVE finfo(float32).min
// (_expand_mask MFULL:387-398, _make_causal_mask :373-385), so a fully masked row softmaxes
// uniformly like torch; only tile padding (key >= Tk) uses -inf.
#include "common.h"

namespace {

constexpr float FMIN = -3.4028234663852886e38f;
constexpr int OOB = 0x7ffffff0;
constexpr int TILE_B = 64 * 64 * 2;   // one [64][64] bf16 tile = 8 KiB

__device__ __forceinline__ int swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// stage a [64 rows][64 cols] bf16 tile: 512 16-B chunks, 2 per thread (256 threads)
__device__ __forceinline__ void stage64(__amdgpu_buffer_rsrc_t rsrc, char* tile, int row0, int nrows, int ld,
                                        int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int blk = wave * 2 + i;              // 8 rows per wave-instruction
    const int row = blk * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ swz(row);
    const int gr = row0 + row;
    const int voff = gr < nrows ? (int)(((unsigned)gr * (unsigned)ld + (unsigned)(lc * 8)) * 2u) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(tile + blk * 1024), 16, voff, 0, 0, 0);
  }
}

// row fragment: element j = T[rbase + (l&31)][16*ks + 8*(l>>5) + j]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rbase, int ks, int lane) {
  const int row = rbase + (lane & 31);
  const int chunk = 2 * ks + (lane >> 5);
  return *(const bf16x8*)(tile + row * 128 + ((chunk ^ swz(row)) << 4));
}
// transposed fragment: element j = T[rbase + 8*(j>>2) + 4*(l>>5) + (j&3)][cbase + (l&31)]
// (the k-order an accumulator tile has when reused as the other MFMA operand)
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int rbase, int cbase, int lane) {
  const int g = lane >> 4, i = lane & 15, h = lane >> 5;
  const int col = cbase + 16 * (g & 1) + 4 * (i & 3);
  bf16x8 r;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = rbase + 8 * hh + 4 * h + (i >> 2);
    const char* a = tile + row * 128 + (((col >> 3) ^ swz(row)) << 4) + (col & 7) * 2;
    bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)LDS_PTR(a));
    r[4 * hh + 0] = t[0]; r[4 * hh + 1] = t[1]; r[4 * hh + 2] = t[2]; r[4 * hh + 3] = t[3];
  }
  return r;
}
// registers 8s..8s+7 of a 32x32 accumulator -> bf16 fragment of k-step s
__device__ __forceinline__ bf16x8 pack_acc(const f32x16& x, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (short)f2bf(x[8 * s + j]);
  return r;
}
__device__ __forceinline__ bf16x8 gload8(__amdgpu_buffer_rsrc_t rsrc, int voff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
  return *(bf16x8*)&v;
}
// row index inside a 32x32 accumulator held in register r by lane half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* out; float* lse;
  const bf16_t* dout; const float* delta; float* delta_out; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  const uint8_t* key_mask;
  int B, H, Tq, Tk;
  int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  long bsq, bsk, bsv, bso, bsdq, bsdk, bsdv;
  int causal; float scale;
  // attention-probability dropout (MFULL:546): keep iff byte >= drop_thr (p quantised to 1/256), kept entries scaled by drop_inv
  unsigned drop_thr; float drop_inv; unsigned long long drop_seed; const unsigned long long* drop_seed_dev;
};

// Dropout mask of the probabilities: ONE Philox4x32-10 block per 4 x 4 patch of the [Tq, Tk] matrix of a (batch, head) — 16
// random bytes, byte (qi, ki) of the patch in word qi, bits 8*ki.  A lane that holds 4 consecutive KEYS of one query (forward,
// dQ) takes one word of its patch, a lane that holds 4 consecutive QUERIES of one key (dK/dV) one byte of each word: every
// kernel regenerates the same mask with one block per 4 probabilities, and nothing is stored.
struct DropCtx { unsigned long long seed; unsigned long long bh_base; unsigned pk_per_row; unsigned thr; float inv; };

__device__ __forceinline__ DropCtx drop_ctx(const AttnP& p, int b, int hd) {
  DropCtx d;
  d.seed = p.drop_seed_dev ? p.drop_seed ^ (*p.drop_seed_dev * 0x9E3779B97F4A7C15ull) : p.drop_seed;      // norm.hip's mix_seed
  d.pk_per_row = (unsigned)((p.Tk + 3) >> 2);
  d.bh_base = ((unsigned long long)b * p.H + hd) * (unsigned long long)((p.Tq + 3) >> 2) * d.pk_per_row;
  d.thr = p.drop_thr; d.inv = p.drop_inv;
  return d;
}
__device__ __forceinline__ void drop_patch(const DropCtx& d, int q, int k, uint32_t w[4]) {
  const unsigned long long idx = d.bh_base + (unsigned long long)(q >> 2) * d.pk_per_row + (unsigned)(k >> 2);
  philox4x32((uint32_t)idx, (uint32_t)(idx >> 32), 0x41545444u, 0u, (uint32_t)d.seed, (uint32_t)(d.seed >> 32), w);
}
// The four lanes of a QUAD hold four consecutive queries (forward, dQ) or keys (dK/dV) and, for each of the four row groups g of
// a 32-row accumulator block, need the SAME 4 x 4 patch — one word (their query) or one byte column (their key) of it each.
// So lane i of the quad generates only the patch of group g = i, and the words travel by quad_perm DPP moves: one Philox block
// per 16 probabilities and lane instead of one per 4 (the generator was ~25 VALU instructions per probability — more than the
// softmax itself; attention dropout cost 4 ms of a 70 ms step).  The mask is the same function of (b, h, q, k) as before.
#define VAC_QUAD_BCAST(V_, G_) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(V_), (G_) * 0x55, 0xf, 0xf, true))
// forward / dQ: this lane's query q (q & 3 == lane & 3), 32-key block starting at kblock (+ 4 * lane half): mw[g] = mask word of
// keys kblock + 8 g .. + 3
__device__ __forceinline__ void drop_block_keys(const DropCtx& d, int q, int kblock, int lane, uint32_t mw[4]) {
  const int i = lane & 3;
  uint32_t w[4];
  drop_patch(d, q, kblock + 8 * i, w);
#define VAC_TAKE(G_)                                                                                   \
  {                                                                                                    \
    const uint32_t t0 = VAC_QUAD_BCAST(w[0], G_), t1 = VAC_QUAD_BCAST(w[1], G_);                       \
    const uint32_t t2 = VAC_QUAD_BCAST(w[2], G_), t3 = VAC_QUAD_BCAST(w[3], G_);                       \
    mw[G_] = i == 0 ? t0 : i == 1 ? t1 : i == 2 ? t2 : t3;                                             \
  }
  VAC_TAKE(0) VAC_TAKE(1) VAC_TAKE(2) VAC_TAKE(3)
#undef VAC_TAKE
}
__device__ __forceinline__ void drop_word_factors(const DropCtx& d, uint32_t x, float m[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] = ((x >> (8 * e)) & 0xffu) >= d.thr ? d.inv : 0.f;
}
// dK/dV: this lane's key k (k & 3 == lane & 3), 32-query block starting at qblock (+ 4 * lane half): mb[g] = the mask bytes of
// queries qblock + 8 g .. + 3 for key k, packed like a mask word
__device__ __forceinline__ void drop_block_queries(const DropCtx& d, int qblock, int k, int lane, uint32_t mb[4]) {
  const int i = lane & 3, sh = 8 * i;
  uint32_t w[4];
  drop_patch(d, qblock + 8 * i, k, w);
#define VAC_TAKE(G_)                                                                                   \
  {                                                                                                    \
    const uint32_t t0 = VAC_QUAD_BCAST(w[0], G_), t1 = VAC_QUAD_BCAST(w[1], G_);                       \
    const uint32_t t2 = VAC_QUAD_BCAST(w[2], G_), t3 = VAC_QUAD_BCAST(w[3], G_);                       \
    mb[G_] = ((t0 >> sh) & 0xffu) | (((t1 >> sh) & 0xffu) << 8) | (((t2 >> sh) & 0xffu) << 16) | (((t3 >> sh) & 0xffu) << 24); \
  }
  VAC_TAKE(0) VAC_TAKE(1) VAC_TAKE(2) VAC_TAKE(3)
#undef VAC_TAKE
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const bf16_t* base, int rows, int ld) {
  const unsigned bytes = rows > 0 ? ((unsigned)(rows - 1) * (unsigned)ld + 64u) * 2u : 0u;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// per-key additive bias for the whole key range, in LDS: 0 / FMIN (masked) / -inf (padding), and one word per 64-key tile:
// nonzero when any key of the tile carries a bias.  A tile without one (the common case: most of a padded batch, all of a full
// one) takes the PLAIN softmax path — no bias load / add, the row maximum taken on the raw scores (scale > 0), the subtraction of
// the maximum folded into the scaling FMA: fma + exp2 + add per probability instead of fma + sub + exp2 + add.  The softmax
// loops of these kernels are VALU-bound (forward: 193 vector instructions + 33 exp2 against 16 MFMAs per tile and wave).
// (keys j = tid + 256 i: the 64 lanes of a wave cover exactly tile 4 i + wave, so its flag is one ballot, no atomics)
__device__ __forceinline__ void fill_key_bias(float* kbias, unsigned* dirty, const uint8_t* mask_row, int Tk, int Tk_pad) {
  for (int j = threadIdx.x; j < Tk_pad; j += 256) {
    float b = 0.f;
    if (j >= Tk) b = -INFINITY;
    else if (mask_row && mask_row[j] == 0) b = FMIN;
    kbias[j] = b;
    const unsigned long long any = __builtin_amdgcn_ballot_w64(b != 0.f);
    if ((threadIdx.x & 63) == 0) dirty[j >> 6] = any != 0ull;
  }
}


// XCD-aware workgroup order.  The 1-D grid's workgroup L runs on XCD L % 8 and each XCD has its own L2; the nblk workgroups of
// one (batch, head) all stream the same K/V (forward, dQ) or Q/dO (dK/dV) rows, so they are made consecutive ON ONE XCD —
// with the natural (block, head, batch) order they land on different XCDs and every one of them fetches those rows from HBM
// again (measured at S=512: 302 MB fetched for 100 MB of operands, L2 hit rate 10 %, the kernel HBM-bound at 5.3 TB/s).
__device__ __forceinline__ void xcd_order(int nblk, int H, int nbh, int& blk, int& hd, int& b) {
  const int L = blockIdx.x;
  int bh;
  if ((nbh & 7) == 0) { const int x = L & 7, j = L >> 3; bh = x + 8 * (j / nblk); blk = j % nblk; }
  else { bh = L / nblk; blk = L % nblk; }
  b = bh / H; hd = bh - b * H;
}

// =================================================================================================
// forward:  S^T = K Q^T (keys in registers, query on the lane) -> online softmax -> O^T = V^T P^T
// =================================================================================================
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, hd, qblk;
  xcd_order((p.Tq + 127) >> 7, p.H, p.B * p.H, qblk, hd, b);
  const int q0 = qblk * 128 + wave * 32;
  const int ql = lane & 31, hh = lane >> 5;
  const int Tk_pad = (p.Tk + 63) & ~63;
  float* kbias = (float*)(smem + 4 * TILE_B);
  unsigned* dirty = (unsigned*)(kbias + Tk_pad);

  const bf16_t* qb = p.q + (long)b * p.bsq + hd * 64;
  const bf16_t* kb = p.k + (long)b * p.bsk + hd * 64;
  const bf16_t* vb = p.v + (long)b * p.bsv + hd * 64;
  __amdgpu_buffer_rsrc_t qs = make_rsrc(qb, p.Tq, p.ldq);
  __amdgpu_buffer_rsrc_t ks = make_rsrc(kb, p.Tk, p.ldk);
  __amdgpu_buffer_rsrc_t vs = make_rsrc(vb, p.Tk, p.ldv);

  int ntile = Tk_pad >> 6;
  if (CAUSAL) ntile = min(ntile, (min(p.Tq, qblk * 128 + 128) + 63) >> 6);

  stage64(ks, smem, 0, p.Tk, p.ldk, wave, lane);
  stage64(vs, smem + TILE_B, 0, p.Tk, p.ldv, wave, lane);
  fill_key_bias(kbias, dirty, p.key_mask ? p.key_mask + (long)b * p.Tk : nullptr, p.Tk, Tk_pad);
  // scores live in the log2 domain (v_exp_f32 is 2^x): scale*log2(e) folded into one FMA per score, masks as -FLT_MAX (they
  // absorb any finite score exactly like the reference's additive finfo.min, and two of them overflow to -inf like there)
  const float scale2 = p.scale * 1.4426950408889634f;

  // Q fragments (B operand: element j = Q[q][16*ks + 8*hh + j]) straight from HBM
  bf16x8 qf[4];
  {
    const int qrow = q0 + ql;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      qf[s] = gload8(qs, qrow < p.Tq ? (int)(((unsigned)qrow * (unsigned)p.ldq + 16u * s + 8u * hh) * 2u) : OOB);
  }

  f32x16 o[2];
  o[0] = (f32x16)(0.f); o[1] = (f32x16)(0.f);
  float m_run = -INFINITY, l_run = 0.f;
  const int qidx = q0 + ql;
  DropCtx dctx;
  if (DROP) dctx = drop_ctx(p, b, hd);

  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    char* kt = smem + cur * (2 * TILE_B);
    char* vt = kt + TILE_B;
    if (t + 1 < ntile) {
      char* nk = smem + (cur ^ 1) * (2 * TILE_B);
      stage64(ks, nk, (t + 1) * 64, p.Tk, p.ldk, wave, lane);
      stage64(vs, nk + TILE_B, (t + 1) * 64, p.Tk, p.ldv, wave, lane);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();

    // ---- S^T[key][q] for the two 32-key blocks of this tile
    f32x16 s[2];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      s[kb2] = (f32x16)(0.f);
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4)
        s[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(kt, kb2 * 32, ks4, lane), qf[ks4], s[kb2], 0, 0, 0);
    }
    // ---- scale + masks, tile max, probabilities
    float tmax = -INFINITY, psum = 0.f, alpha;
    if (!CAUSAL && __builtin_amdgcn_readfirstlane((int)dirty[t]) == 0) {
      // plain tile (fill_key_bias): maximum on the raw scores, then one FMA + exp2 per probability
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb2][r]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax * scale2);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);             // m_run = -inf on the first tile -> 0
      const float nm = -m_new;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[kb2][r], scale2, nm));
          s[kb2][r] = e;
          psum += e;
        }
      m_run = m_new;
    } else {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kbase = t * 64 + kb2 * 32 + 8 * g + 4 * hh;
          const f32x4 bias = *(const f32x4*)(kbias + kbase);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = fmaf(s[kb2][4 * g + e], scale2, bias[e]);
            if (CAUSAL && (kbase + e) > qidx) v += FMIN;
            s[kb2][4 * g + e] = v;
            tmax = fmaxf(tmax, v);
          }
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      alpha = __builtin_amdgcn_exp2f(m_run - m_new);             // m_run = -inf on the first tile -> 0
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kb2][r] - m_new);
          s[kb2][r] = e;
          psum += e;
        }
      m_run = m_new;
    }
    l_run = l_run * alpha + psum;
    if (DROP) {          // the normaliser sums the undropped probabilities (softmax first, dropout second: MFULL:534,546)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        uint32_t mw[4];
        drop_block_keys(dctx, qidx, t * 64 + kb2 * 32 + 4 * hh, lane, mw);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float mk[4];
          drop_word_factors(dctx, mw[g], mk);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[kb2][4 * g + e] *= mk[e];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    // ---- O^T[hd][q] += V^T P^T
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = pack_acc(s[kb2], ss);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
          o[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(vt, kb2 * 32 + 16 * ss, hb * 32, lane), pf, o[hb], 0, 0, 0);
      }
    lds_barrier();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  // Output through LDS: the accumulator layout gives every lane 8-byte pieces of ONE query row (row stride ldo), i.e. 512
  // scattered 8-byte writes per wave-instruction — 14 us of a 63 us launch at S=512.  Each wave transposes its 32 x 64 tile
  // in its own 4 KiB of the (now idle) K/V buffers and stores 128-byte row segments with 16-byte lanes.
  {
    char* ot = smem + wave * 4096;                     // [32 rows][128 B], 16-byte chunks XOR-swizzled by the row
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 pk = {pack2bf(o[hb][4 * g] * inv, o[hb][4 * g + 1] * inv),
                          pack2bf(o[hb][4 * g + 2] * inv, o[hb][4 * g + 3] * inv)};
        const int c16 = hb * 4 + g;
        *(u32x2*)(ot + ql * 128 + ((c16 ^ (ql & 7)) << 4) + 8 * hh) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-local: written and read by the same wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (lane >> 3) + 8 * i, c16 = lane & 7;
      const int qrow = q0 + row;
      const u32x4 v = *(const u32x4*)(ot + row * 128 + ((c16 ^ (row & 7)) << 4));
      if (qrow < p.Tq) *(u32x4*)(p.out + (long)b * p.bso + (long)qrow * p.ldo + hd * 64 + c16 * 8) = v;
    }
  }
  if (qidx < p.Tq && p.lse && hh == 0) p.lse[((long)b * p.H + hd) * p.Tq + qidx] = m_run * 0.6931471805599453f + __logf(l_tot);
}

// =================================================================================================
// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
// =================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int H, int Tq, int ldo,
                                                         long bso) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * Tq) return;
  const int b = (int)(row / Tq), q = (int)(row % Tq);
  const bf16_t* orow = o + (long)b * bso + (long)q * ldo;
  const bf16_t* drow = dout + (long)b * bso + (long)q * ldo;
  for (int c = lane; c < H * 8; c += 64) {
    u32x4 a = *(const u32x4*)(orow + c * 8), d = *(const u32x4*)(drow + c * 8);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s += __uint_as_float(a[i] << 16) * __uint_as_float(d[i] << 16);
      s += __uint_as_float(a[i] & 0xffff0000u) * __uint_as_float(d[i] & 0xffff0000u);
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if ((c & 7) == 0) delta[((long)b * H + (c >> 3)) * Tq + q] = s;
  }
}

// =================================================================================================
// dK, dV: wave owns 32 keys (on the lane); sweeps 64-query tiles of Q and dO.
//   S[q][k]  = Q K^T        (A = Q rows from LDS, B = K from registers)
//   dP[q][k] = dO V^T       (A = dO rows from LDS, B = V from registers)
//   dV^T[d][k] += dO^T P    (A = dO^T via tr-read, B = P accumulator)
//   dK^T[d][k] += Q^T dS    (A = Q^T  via tr-read, B = dS accumulator)
// =================================================================================================
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, hd, kblk;
  xcd_order((p.Tk + 127) >> 7, p.H, p.B * p.H, kblk, hd, b);
  const int k0 = kblk * 128 + wave * 32;
  const int kl = lane & 31, hh = lane >> 5;
  const int kidx = k0 + kl;
  // LDS: 2 x {Q tile, dO tile} + 2 x {lse[64], delta[64], -lse log2(e) [64]}
  float* stats = (float*)(smem + 4 * TILE_B);
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;

  const bf16_t* qb = p.q + (long)b * p.bsq + hd * 64;
  const bf16_t* kb = p.k + (long)b * p.bsk + hd * 64;
  const bf16_t* vb = p.v + (long)b * p.bsv + hd * 64;
  const bf16_t* dob = p.dout + (long)b * p.bso + hd * 64;
  __amdgpu_buffer_rsrc_t qs = make_rsrc(qb, p.Tq, p.ldq);
  __amdgpu_buffer_rsrc_t ks = make_rsrc(kb, p.Tk, p.ldk);
  __amdgpu_buffer_rsrc_t vs = make_rsrc(vb, p.Tk, p.ldv);
  __amdgpu_buffer_rsrc_t dos = make_rsrc(dob, p.Tq, p.ldo);
  const float* lse = p.lse + ((long)b * p.H + hd) * p.Tq;
  const float* delta = p.delta + ((long)b * p.H + hd) * p.Tq;

  const int nq_tiles = (p.Tq + 63) >> 6;
  int t_begin = 0;
  if (p.causal) t_begin = min(nq_tiles, (kblk * 128) >> 6);   // queries before the first key see none of them

  // K / V fragments of this wave's keys (B operand: element j = K[k][16*ks + 8*hh + j])
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = gload8(ks, kidx < p.Tk ? (int)(((unsigned)kidx * (unsigned)p.ldk + 16u * s + 8u * hh) * 2u) : OOB);
    vf[s] = gload8(vs, kidx < p.Tk ? (int)(((unsigned)kidx * (unsigned)p.ldv + 16u * s + 8u * hh) * 2u) : OOB);
  }
  float kbias = 0.f;
  if (kidx >= p.Tk) kbias = -INFINITY;
  else if (p.key_mask && p.key_mask[(long)b * p.Tk + kidx] == 0) kbias = FMIN;

  f32x16 dk[2], dv[2];
  dk[0] = (f32x16)(0.f); dk[1] = (f32x16)(0.f); dv[0] = (f32x16)(0.f); dv[1] = (f32x16)(0.f);
  DropCtx dctx;
  if (DROP) dctx = drop_ctx(p, b, hd);

  auto load_stat = [&](int t) -> float {
    float v = 0.f;
    if (tid < 128) {
      const int qi = t * 64 + (tid & 63);
      if (tid < 64) v = qi < p.Tq ? lse[qi] : INFINITY;
      else v = qi < p.Tq ? delta[qi] : 0.f;
    }
    return v;
  };
  auto put_stat = [&](float* st, float sv) {
    if (tid < 128) st[tid] = sv;
    if (tid < 64) st[128 + tid] = -sv * LOG2E;       // (padding queries: lse = +inf -> -inf -> probability 0 on the plain path too)
  };
  // every key of this wave unmasked and in range, no causal mask: the plain path (see fill_key_bias) for every tile
  const bool plain = !p.causal && __builtin_amdgcn_ballot_w64(kbias != 0.f) == 0ull;

  if (t_begin < nq_tiles) {
    const float sv = load_stat(t_begin);
    stage64(qs, smem, t_begin * 64, p.Tq, p.ldq, wave, lane);
    stage64(dos, smem + TILE_B, t_begin * 64, p.Tq, p.ldo, wave, lane);
    put_stat(stats, sv);
  }
  for (int t = t_begin; t < nq_tiles; ++t) {
    const int cur = (t - t_begin) & 1;
    char* qt = smem + cur * (2 * TILE_B);
    char* dot = qt + TILE_B;
    if (t + 1 < nq_tiles) {
      char* nq = smem + (cur ^ 1) * (2 * TILE_B);
      const float sv = load_stat(t + 1);
      stage64(qs, nq, (t + 1) * 64, p.Tq, p.ldq, wave, lane);
      stage64(dos, nq + TILE_B, (t + 1) * 64, p.Tq, p.ldo, wave, lane);
      put_stat(stats + (cur ^ 1) * 192, sv);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    const float* st = stats + cur * 192;

#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
      f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(qt, qb2 * 32, ks4, lane), kf[ks4], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(dot, qb2 * 32, ks4, lane), vf[ks4], dp, 0, 0, 0);
      }
      uint32_t mb[4];
      if (DROP) drop_block_queries(dctx, t * 64 + qb2 * 32 + 4 * hh, kidx, lane, mb);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ql0 = qb2 * 32 + 8 * g + 4 * hh;
        const f32x4 l4 = *(const f32x4*)(st + ql0);
        const f32x4 d4 = *(const f32x4*)(st + 64 + ql0);
        float mq[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_word_factors(dctx, mb[g], mq);
        // (dS carries no softmax scale here: dK = scale * dS^T Q is scaled once, on the way out)
        if (plain) {
          const f32x4 nl4 = *(const f32x4*)(st + 128 + ql0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(s[4 * g + e], scale2, nl4[e]));
            s[4 * g + e] = DROP ? pe * mq[e] : pe;                                   // dV = (P o M)^T dO
            dp[4 * g + e] = pe * ((DROP ? dp[4 * g + e] * mq[e] : dp[4 * g + e]) - d4[e]);    // dS = P o (dP o M - delta)
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = s[4 * g + e] * p.scale + kbias;
            if (p.causal && kidx > (t * 64 + ql0 + e)) v += FMIN;
            const float pe = __expf(v - l4[e]);
            s[4 * g + e] = DROP ? pe * mq[e] : pe;
            dp[4 * g + e] = pe * ((DROP ? dp[4 * g + e] * mq[e] : dp[4 * g + e]) - d4[e]);
          }
        }
      }
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 pf = pack_acc(s, ss), dsf = pack_acc(dp, ss);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
          dv[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(dot, qb2 * 32 + 16 * ss, hb * 32, lane), pf, dv[hb], 0, 0, 0);
          dk[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(qt, qb2 * 32 + 16 * ss, hb * 32, lane), dsf, dk[hb], 0, 0, 0);
        }
      }
    }
    lds_barrier();
  }

  if (kidx < p.Tk) {
    bf16_t* dkrow = p.dk + (long)b * p.bsdk + (long)kidx * p.lddk + hd * 64;
    bf16_t* dvrow = p.dv + (long)b * p.bsdv + (long)kidx * p.lddv + hd * 64;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 a = {pack2bf(dk[hb][4 * g] * p.scale, dk[hb][4 * g + 1] * p.scale), pack2bf(dk[hb][4 * g + 2] * p.scale, dk[hb][4 * g + 3] * p.scale)};
        u32x2 c = {pack2bf(dv[hb][4 * g], dv[hb][4 * g + 1]), pack2bf(dv[hb][4 * g + 2], dv[hb][4 * g + 3])};
        *(u32x2*)(dkrow + hb * 32 + 8 * g + 4 * hh) = a;
        *(u32x2*)(dvrow + hb * 32 + 8 * g + 4 * hh) = c;
      }
  }
}

// =================================================================================================
// dQ: wave owns 32 queries (on the lane); sweeps 64-key tiles of K and V.
//   S^T[k][q]  = K Q^T ; dP^T[k][q] = V dO^T ; dQ^T[d][q] += K^T dS^T (A = K^T via tr-read)
// =================================================================================================
// DELTA_PASS: the same sweep without the dQ product — it only forms delta[q] = sum_k P[q][k] dP[q][k] (with the dropout mask) and
// writes it for the two kernels proper.  delta = rowsum(dO o O) is the same number analytically, but O was rounded to bf16 on its way
// out of the forward and dS = P o (dP - delta) subtracts two nearly equal quantities whenever the value rows share a large common
// component (they do: hidden states of one sequence are 80-99 % parallel): measured on such inputs, dQ is off by 1.1 % with
// rowsum(dO o O_bf16) and by 0.17 % with a delta formed from the same P and dP the kernels use (tools/attn_bwd_error.py;
// full-depth q_proj / k_proj weight gradients 4.7e-2 / 3.5e-2 -> see tests).
template <bool DROP, bool DELTA_PASS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b, hd, qblk;
  xcd_order((p.Tq + 127) >> 7, p.H, p.B * p.H, qblk, hd, b);
  const int q0 = qblk * 128 + wave * 32;
  const int ql = lane & 31, hh = lane >> 5;
  const int qidx = q0 + ql;
  const int Tk_pad = (p.Tk + 63) & ~63;
  float* kbias = (float*)(smem + 4 * TILE_B);
  unsigned* dirty = (unsigned*)(kbias + Tk_pad);
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale2 = p.scale * LOG2E;

  const bf16_t* qb = p.q + (long)b * p.bsq + hd * 64;
  const bf16_t* kb = p.k + (long)b * p.bsk + hd * 64;
  const bf16_t* vb = p.v + (long)b * p.bsv + hd * 64;
  const bf16_t* dob = p.dout + (long)b * p.bso + hd * 64;
  __amdgpu_buffer_rsrc_t qs = make_rsrc(qb, p.Tq, p.ldq);
  __amdgpu_buffer_rsrc_t ks = make_rsrc(kb, p.Tk, p.ldk);
  __amdgpu_buffer_rsrc_t vs = make_rsrc(vb, p.Tk, p.ldv);
  __amdgpu_buffer_rsrc_t dos = make_rsrc(dob, p.Tq, p.ldo);

  int ntile = Tk_pad >> 6;
  if (p.causal) ntile = min(ntile, (min(p.Tq, qblk * 128 + 128) + 63) >> 6);

  stage64(ks, smem, 0, p.Tk, p.ldk, wave, lane);
  stage64(vs, smem + TILE_B, 0, p.Tk, p.ldv, wave, lane);
  fill_key_bias(kbias, dirty, p.key_mask ? p.key_mask + (long)b * p.Tk : nullptr, p.Tk, Tk_pad);

  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = gload8(qs, qidx < p.Tq ? (int)(((unsigned)qidx * (unsigned)p.ldq + 16u * s + 8u * hh) * 2u) : OOB);
    dof[s] = gload8(dos, qidx < p.Tq ? (int)(((unsigned)qidx * (unsigned)p.ldo + 16u * s + 8u * hh) * 2u) : OOB);
  }
  float lse_q = INFINITY, delta_q = 0.f;
  if (qidx < p.Tq) {
    lse_q = p.lse[((long)b * p.H + hd) * p.Tq + qidx];
    if (!DELTA_PASS) delta_q = p.delta[((long)b * p.H + hd) * p.Tq + qidx];
  }
  const float nl2 = -lse_q * LOG2E;                    // (padding queries: -inf -> probability 0)
  float dsum = 0.f;
  f32x16 dq[2];
  dq[0] = (f32x16)(0.f); dq[1] = (f32x16)(0.f);
  DropCtx dctx;
  if (DROP) dctx = drop_ctx(p, b, hd);

  for (int t = 0; t < ntile; ++t) {
    const int cur = t & 1;
    char* kt = smem + cur * (2 * TILE_B);
    char* vt = kt + TILE_B;
    if (t + 1 < ntile) {
      char* nk = smem + (cur ^ 1) * (2 * TILE_B);
      stage64(ks, nk, (t + 1) * 64, p.Tk, p.ldk, wave, lane);
      stage64(vs, nk + TILE_B, (t + 1) * 64, p.Tk, p.ldv, wave, lane);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();
    const bool plain = !p.causal && __builtin_amdgcn_readfirstlane((int)dirty[t]) == 0;      // see fill_key_bias
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      f32x16 s = (f32x16)(0.f), dp = (f32x16)(0.f);
#pragma unroll
      for (int ks4 = 0; ks4 < 4; ++ks4) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(kt, kb2 * 32, ks4, lane), qf[ks4], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row(vt, kb2 * 32, ks4, lane), dof[ks4], dp, 0, 0, 0);
      }
      uint32_t mw[4];
      if (DROP) drop_block_keys(dctx, qidx, t * 64 + kb2 * 32 + 4 * hh, lane, mw);
      // (dS carries no softmax scale here: dQ = scale * dS K is scaled once, on the way out)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int kbase = t * 64 + kb2 * 32 + 8 * g + 4 * hh;
        float mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_word_factors(dctx, mw[g], mk);
        if (plain) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = __builtin_amdgcn_exp2f(fmaf(s[4 * g + e], scale2, nl2));
            const float dpm = DROP ? dp[4 * g + e] * mk[e] : dp[4 * g + e];
            if (DELTA_PASS) dsum += pe * dpm;
            else dp[4 * g + e] = pe * (dpm - delta_q);
          }
        } else {
          const f32x4 bias = *(const f32x4*)(kbias + kbase);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = s[4 * g + e] * p.scale + bias[e];
            if (p.causal && (kbase + e) > qidx) v += FMIN;
            const float pe = __expf(v - lse_q);
            const float dpm = DROP ? dp[4 * g + e] * mk[e] : dp[4 * g + e];
            if (DELTA_PASS) dsum += pe * dpm;                     // (pe is exactly 0 for padding keys: bias = -inf)
            else dp[4 * g + e] = pe * (dpm - delta_q);
          }
        }
      }
      if (!DELTA_PASS) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const bf16x8 dsf = pack_acc(dp, ss);
#pragma unroll
          for (int hb = 0; hb < 2; ++hb)
            dq[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr(kt, kb2 * 32 + 16 * ss, hb * 32, lane), dsf, dq[hb], 0, 0, 0);
        }
      }
    }
    lds_barrier();
  }
  if (DELTA_PASS) {
    dsum += __shfl_xor(dsum, 32, 64);                         // the two lane halves hold different keys of the same query
    if (qidx < p.Tq && hh == 0) p.delta_out[((long)b * p.H + hd) * p.Tq + qidx] = dsum;
    return;
  }
  if (qidx < p.Tq) {
    bf16_t* drow = p.dq + (long)b * p.bsdq + (long)qidx * p.lddq + hd * 64;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 a = {pack2bf(dq[hb][4 * g] * p.scale, dq[hb][4 * g + 1] * p.scale), pack2bf(dq[hb][4 * g + 2] * p.scale, dq[hb][4 * g + 3] * p.scale)};
        *(u32x2*)(drow + hb * 32 + 8 * g + 4 * hh) = a;
      }
  }
}

// p in [0, 1): quantised to 1/256 (the kept probabilities are scaled by 1 / (1 - round(256 p) / 256), so the estimate stays unbiased)
int set_dropout(AttnP& p, float p_drop, uint64_t seed, const uint64_t* seed_dev, const char* who) {
  if (!(p_drop >= 0.f && p_drop < 1.f)) { vacnic_set_error("%s: p_drop must be in [0, 1)", who); return VACNIC_BAD_SHAPE; }
  unsigned thr = (unsigned)(p_drop * 256.f + 0.5f);
  if (thr > 255) thr = 255;
  p.drop_thr = thr;
  p.drop_inv = 256.f / (256.f - (float)thr);
  p.drop_seed = seed; p.drop_seed_dev = (const unsigned long long*)seed_dev;
  return VACNIC_OK;
}

int check_common(int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                 const char* who) {
  if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0) {
    vacnic_set_error("%s: empty problem B=%ld H=%ld Tq=%ld Tk=%ld", who, (long)B, (long)H, (long)Tq, (long)Tk);
    return VACNIC_BAD_SHAPE;
  }
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7)) {
    vacnic_set_error("%s: row strides must be multiples of 8 elements", who);
    return VACNIC_MISALIGNED;
  }
  if (Tk > 8192 || H > 65535 || B > 65535) {
    vacnic_set_error("%s: Tk=%ld / H / B beyond supported range", who, (long)Tk);
    return VACNIC_UNSUPPORTED;
  }
  return VACNIC_OK;
}

}  // namespace


// ---------------------------------------------------------------------------------------------------------------------
// Single-query attention (Tq == 1, no causal flag): the KV-cached decoder step of caption generation (MFULL:474-501).
// One wave per (batch row, head): lanes split the keys for q.k, a wave reduction gives the softmax statistics, the
// probabilities go through LDS, then lanes = (8 key groups) x (8 chunks of 8 dims) accumulate P.V and meet in three
// shuffle steps.  fp32 throughout; same additive finfo.min mask semantics as the tiled kernel.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void attn_decode_kernel(AttnP p) {
  extern __shared__ float probs[];                    // Tk floats, then the per-wave partial results
  const int lane = threadIdx.x & 63, h = blockIdx.x, b = blockIdx.y;
  const int wv = NWV == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // NWV waves split the keys into contiguous ranges (long cross-attention sources) and merge their softmax partials in LDS
  const int per = (p.Tk + NWV - 1) / NWV;
  const int k_lo = wv * per, k_hi = min(p.Tk, k_lo + per);
  const bf16_t* q = p.q + (long)b * p.bsq + h * 64;
  const bf16_t* kb = p.k + (long)b * p.bsk + h * 64;
  const bf16_t* vb = p.v + (long)b * p.bsv + h * 64;
  const uint8_t* km = p.key_mask ? p.key_mask + (long)b * p.Tk : nullptr;
  float qf[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const u32x4 r = *(const u32x4*)(q + c * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) { qf[c * 8 + 2 * i] = __uint_as_float(r[i] << 16); qf[c * 8 + 2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u); }
  }
  float m = -INFINITY;
  // keys in batches of 4 per lane: all 32 16-byte loads of a batch are issued before the first is consumed (with one wave
  // per CU nothing else hides the ~2 us round trip of a load)
  for (int key0 = k_lo + lane; key0 < k_hi; key0 += 256) {
    u32x4 kr[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = key0 + 64 * u;
      const bf16_t* krow = kb + (long)(key < k_hi ? key : key0) * p.ldk;
#pragma unroll
      for (int c = 0; c < 8; ++c) kr[u][c] = *(const u32x4*)(krow + c * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = key0 + 64 * u;
      if (key < k_hi) {
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dot += qf[c * 8 + 2 * i] * __uint_as_float(kr[u][c][i] << 16);
            dot += qf[c * 8 + 2 * i + 1] * __uint_as_float(kr[u][c][i] & 0xffff0000u);
          }
        float sc = dot * p.scale;
        if (km && km[key] == 0) sc += -3.4028234663852886e38f;     // additive finfo(float32).min, as _expand_mask
        probs[key] = sc;
        m = fmaxf(m, sc);
      }
    }
  }
  m = wave_max(m);
  float l = 0.f;
  for (int key = k_lo + lane; key < k_hi; key += 64) {
    const float e = __expf(probs[key] - m);
    probs[key] = e;
    l += e;
  }
  l = wave_sum(l);
  __syncthreads();
  const int kg = lane >> 3, dc = lane & 7;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int key0 = k_lo + kg; key0 < k_hi; key0 += 64) {
    u32x4 vr[8]; float pk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int key = key0 + 8 * u;
      const bool ok = key < k_hi;
      vr[u] = *(const u32x4*)(vb + (long)(ok ? key : key0) * p.ldv + dc * 8);
      pk[u] = ok ? probs[key] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += pk[u] * __uint_as_float(vr[u][i] << 16);
        acc[2 * i + 1] += pk[u] * __uint_as_float(vr[u][i] & 0xffff0000u);
      }
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
  if constexpr (NWV > 1) {
    float* pacc = probs + p.Tk;                        // [NWV][64]
    float* pm = pacc + NWV * 64;                       // [NWV]
    float* pl = pm + NWV;                              // [NWV]
    if (kg == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pacc[wv * 64 + dc * 8 + j] = acc[j];
    }
    if (lane == 0) { pm[wv] = m; pl[wv] = l; }
    __syncthreads();
    if (wv != 0) return;
    float gm = pm[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) gm = fmaxf(gm, pm[w]);
    float gl = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const float sc = pl[w] > 0.f ? __expf(pm[w] - gm) : 0.f;      // an empty key range contributes nothing
      gl += pl[w] * sc;
      if (kg == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += pacc[w * 64 + dc * 8 + j] * sc;
      }
    }
    m = gm; l = gl;
  }
  if (kg == 0) {
    const float inv = 1.f / l;
    bf16_t* o = p.out + (long)b * p.bso + h * 64 + dc * 8;
    *(u32x4*)o = (u32x4){pack2bf(acc[0] * inv, acc[1] * inv), pack2bf(acc[2] * inv, acc[3] * inv),
                         pack2bf(acc[4] * inv, acc[5] * inv), pack2bf(acc[6] * inv, acc[7] * inv)};
  }
  if (p.lse && lane == 0) p.lse[((long)b * p.H + h)] = m + logf(l);
}

extern "C" int vacnic_attn_fwd(const vacnic_attn_fwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_attn_fwd, a, stream);
  VCHECK(a && a->q && a->k && a->v && a->out, VACNIC_BAD_SHAPE, "attn_fwd: null operand");
  if (int e = check_common(a->B, a->H, a->Tq, a->Tk, a->ldq, a->ldk, a->ldv, a->ldo, "attn_fwd")) return e;
  VCHECK(aligned16(a->q) && aligned16(a->k) && aligned16(a->v) && aligned16(a->out) && !(a->bsq & 7) && !(a->bsk & 7) &&
         !(a->bsv & 7) && !(a->bso & 7), VACNIC_MISALIGNED, "attn_fwd: pointers/batch strides must be 16-byte aligned");
  AttnP p = {};
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.out = (bf16_t*)a->out;
  p.lse = a->lse; p.key_mask = a->key_mask;
  p.B = (int)a->B; p.H = (int)a->H; p.Tq = (int)a->Tq; p.Tk = (int)a->Tk;
  p.ldq = (int)a->ldq; p.ldk = (int)a->ldk; p.ldv = (int)a->ldv; p.ldo = (int)a->ldo;
  p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
  p.causal = a->causal; p.scale = a->scale;
  if (int e = set_dropout(p, a->p_drop, a->seed, a->seed_dev, "attn_fwd")) return e;
  if (p.Tq == 1 && !p.causal && p.Tk <= 16384 && !p.drop_thr) {      // decoder step (inference): one wave per (row, head)
    if (p.Tk >= 256) hipLaunchKernelGGL(attn_decode_kernel<4>, dim3(p.H, p.B), dim3(256), (size_t)(p.Tk + 4 * 64 + 8) * 4, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_decode_kernel<1>, dim3(p.H, p.B), dim3(64), (size_t)p.Tk * 4, (hipStream_t)stream, p);
    VLAUNCH_CHECK();
    return VACNIC_OK;
  }
  const int Tk_pad = (p.Tk + 63) & ~63;
  dim3 grid((unsigned)(((p.Tq + 127) / 128) * p.H * p.B));
  const size_t lds = 4 * TILE_B + Tk_pad * 4 + (Tk_pad >> 6) * 4;          // K/V ring + key bias + per-tile flags
  hipStream_t st = (hipStream_t)stream;
  if (p.drop_thr) {
    if (p.causal) hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), lds, st, p);
  } else {
    if (p.causal) hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), lds, st, p);
  }
  VLAUNCH_CHECK();
  return VACNIC_OK;
}

extern "C" int vacnic_attn_bwd(const vacnic_attn_bwd_args* a, void* stream) {
  VPLAN_REC_STRUCT(vacnic_attn_bwd, a, stream);
  VCHECK(a && a->q && a->k && a->v && a->out && a->dout && a->lse && a->delta && a->dq && a->dk && a->dv,
         VACNIC_BAD_SHAPE, "attn_bwd: null operand");
  if (int e = check_common(a->B, a->H, a->Tq, a->Tk, a->ldq, a->ldk, a->ldv, a->ldo, "attn_bwd")) return e;
  VCHECK(!(a->lddq & 7) && !(a->lddk & 7) && !(a->lddv & 7) && !(a->bsq & 7) && !(a->bsk & 7) && !(a->bsv & 7) &&
         !(a->bso & 7) && !(a->bsdq & 7) && !(a->bsdk & 7) && !(a->bsdv & 7), VACNIC_MISALIGNED,
         "attn_bwd: strides must be multiples of 8 elements");
  AttnP p = {};
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v;
  p.dout = (const bf16_t*)a->dout; p.lse = (float*)a->lse; p.delta = a->delta;
  p.dq = (bf16_t*)a->dq; p.dk = (bf16_t*)a->dk; p.dv = (bf16_t*)a->dv; p.key_mask = a->key_mask;
  p.B = (int)a->B; p.H = (int)a->H; p.Tq = (int)a->Tq; p.Tk = (int)a->Tk;
  p.ldq = (int)a->ldq; p.ldk = (int)a->ldk; p.ldv = (int)a->ldv; p.ldo = (int)a->ldo;
  p.lddq = (int)a->lddq; p.lddk = (int)a->lddk; p.lddv = (int)a->lddv;
  p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
  p.bsdq = a->bsdq; p.bsdk = a->bsdk; p.bsdv = a->bsdv;
  p.causal = a->causal; p.scale = a->scale;
  if (int e = set_dropout(p, a->p_drop, a->seed, a->seed_dev, "attn_bwd")) return e;
  hipStream_t s = (hipStream_t)stream;
  const long rows = (long)p.B * p.Tq;
  const int Tk_pad = (p.Tk + 63) & ~63;
  const dim3 gkv((unsigned)(((p.Tk + 127) / 128) * p.H * p.B)), gq((unsigned)(((p.Tq + 127) / 128) * p.H * p.B));
  const size_t lds_q = 4 * TILE_B + Tk_pad * 4 + (Tk_pad >> 6) * 4;
  p.delta_out = a->delta;
  // delta: from the kernels' own P and dP (a sweep of the dQ kernel without its last product) for the decoder-sized problems,
  // where it costs microseconds; rowsum(dO o O) for the long encoder sequences (a_delta_mode: 0 = by size, 1 = always the sweep,
  // 2 = always rowsum(dO o O))
  static int env_mode = -1;
  if (env_mode < 0) { const char* e = getenv("VACNIC_ATTN_DELTA"); env_mode = e ? atoi(e) : 0; }
  const bool sweep = env_mode == 1 || (env_mode == 0 && p.Tq <= 128);
  if (sweep) {
    if (p.drop_thr) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), gq, dim3(256), lds_q, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), gq, dim3(256), lds_q, s, p);
  } else {
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const bf16_t*)a->out,
                       (const bf16_t*)a->dout, a->delta, p.B, p.H, p.Tq, p.ldo, (long)p.bso);
  }
  VLAUNCH_CHECK();
  if (p.drop_thr) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, gkv, dim3(256), 4 * TILE_B + 2 * 192 * 4, s, p);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, gkv, dim3(256), 4 * TILE_B + 2 * 192 * 4, s, p);
  VLAUNCH_CHECK();
  if (p.drop_thr) hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, dim3(256), lds_q, s, p);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, dim3(256), lds_q, s, p);
  VLAUNCH_CHECK();
  return VACNIC_OK;
}
