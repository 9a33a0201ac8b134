// Fused attention core on fp32 operands (gfx950, 3-term split-fp16 MFMA): balanced, DMA-fed.
//
// Reference modules/attention.py:221-246 (scale, key ^ query (^ causal) mask with the -2^32 fill, softmax, .V, optional
// alignments) for every caller that holds Q, K, V as fp32 rows: vnr_op_attention, the training step, and inference calls the
// operand-image kernels (attention3.hip) do not cover.  (Round 1's first-generation kernel, attention.hip, was retired in round 3:
// alignments with Tk > 512 are now a two-pass form -- this kernel without alignments leaves the softmax row statistics, a small
// kernel rebuilds the probabilities from Q, K and those.)
//
// Decomposition.  grid = (ceil(Tq/64), H, B); a workgroup = 4 waves = one 64-row query block.
//   wave w : query tile  qt = w & 1  (rows Q0 + 32*qt .. +31)
//            key half    kh = w >> 1 (keys 32*kh .. 32*kh+31 of EVERY 64-key tile)
// so the two waves that share a query tile split each K/V tile between them: every wave of a workgroup has
// the same MFMA count (tiles * 64), causal blocks differ only by their tile count and are dispatched
// heaviest-first, and 2 workgroups fit a CU (2 waves per SIMD overlap softmax VALU work with MFMAs).
// The partial softmax states (m, l, O) of a wave pair are merged through LDS at the end.
//
// K/V tiles (64 keys x 64 floats each) arrive by LDS-DMA (buffer_load ... lds) into a 2-slot ring; keys beyond
// Tk resolve to out-of-range offsets and are zero-filled by the hardware.  K rows are stored with a 16-byte
// chunk XOR swizzle (source side) so the S^T = K.Q^T operand reads (ds_read_b128 down a column of chunks) are
// conflict-free; V rows are linear (the P.V operand read is 32 consecutive floats per half-wave).
//
// Operand mapping: S^T puts one query per lane (softmax reductions in-lane + one
// cross-half shuffle) and leaves P in A-operand position for O = P.V.
#include "common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace vnr {

namespace {
constexpr unsigned kOob = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ int frow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// exp(x) for x <= 0 on the hardware exp2 unit (v_exp_f32, ~1 ulp): t = x*log2(e) is rounded, the rounding error
// and the low part of log2(e) are folded back to first order: exp(x) = 2^t * (1 + ln2 * e).  Relative error
// ~2e-7 for |x| < 87; below -87 (result < 1.7e-38, i.e. denormal against a softmax denominator >= 1) the result
// is flushed to 0 -- branch free: the softmax calls this 32 times per lane and a divergent fallback costs more
// than the exponentials.
__device__ __forceinline__ float fast_exp(float x) {
  const float L2E = 1.44269502e+00f, L2E_LO = 1.92596299e-08f, LN2 = 6.93147182e-01f;
  const float t = x * L2E;
  float e = __builtin_fmaf(x, L2E, -t);
  e = __builtin_fmaf(x, L2E_LO, e);
  const float r = __builtin_amdgcn_exp2f(t);
  return (x < -87.0f) ? 0.0f : __builtin_fmaf(r, e * LN2, r);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, f16x8& hi, f16x8& lo) {
  const vnr_f8 xs_ = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  vnr_split(xs_, hi, lo);
}
// 3-term split product on the f16 matrix pipe: (ah + al) * (bh + bl) ~= ah*bh + al*bh + ah*bl  (fp32 accumulate)
__device__ __forceinline__ f32x16 mfma3(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  return c;
}
}  // namespace

// NT: 64-key tiles whose logits an alignment launch keeps in registers (2: Tk <= 128, the round-1 form; 4 / 8: Tk <= 256 / 512,
// round 2 -- the stored probabilities of the training step's causal self-attention, which the first-generation kernel produced)
template <bool ALI, int NT = 2>
__global__ void __launch_bounds__(256, 2)
attn2_kernel(const AttnArgs a, int nqb) {
  constexpr int KT = 64;
  constexpr int SLOT = 2 * KT * 256;                 // K tile + V tile, bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [0, 2*SLOT): K/V ring.  The 16 KB K region of slot 0 doubles as the per-wave 32x32 transpose staging (ALI,
  // after every wave finished its S^T blocks) and as the merge scratch (after the last tile); a 2 KB exchange
  // area for row statistics follows the ring.  67.5 KB per workgroup -> two workgroups per CU.
  float* scratch = reinterpret_cast<float*>(smem);
  float* xarea = reinterpret_cast<float*>(smem + 2 * SLOT);

  // XCD-aware work map (speed only): workgroup id w runs on XCD w % 8 and each XCD has a private L2, so all
  // query blocks of one (batch, head) pair -- which re-read the same K/V -- are given ids with equal w % 8.
  // pairs are dealt round-robin to the 8 XCDs; within an XCD the order is query-block-major so that the
  // heaviest causal blocks of every pair still start first.  Falls back to the plain order when the pair count
  // is not a multiple of 8.
  const int npairs = a.B * a.H;
  int wg = blockIdx.x;
  int pair, qidx;
  if ((npairs & 7) == 0) {
    const int xcd = wg & 7, j = wg >> 3, ppx = npairs >> 3;   // ppx pairs per XCD
    qidx = j / ppx;
    pair = (j - qidx * ppx) * 8 + xcd;
  } else {
    qidx = wg / npairs;
    pair = wg - qidx * npairs;
  }
  const int b = pair / a.H, hd = pair - b * a.H;
  const int qb = a.causal ? (nqb - 1 - qidx) : qidx;             // heaviest causal blocks first
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = wave & 1, kh = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  unsigned long long* ts = a.dbg_ts ? a.dbg_ts + (size_t)blockIdx.x * 8 : nullptr;
  auto stamp = [&](int i) { if (ts && tid == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  const int Q0 = qb * 64, q0 = Q0 + qt * 32;
  const int ntiles_all = (a.Tk + KT - 1) / KT;
  const bool wave_active = q0 < a.Tq;
  // per-launch power-of-two operand scales (AttnArgs::qkv_absmax): max |Q|, |K|, |V| -> ~2^10 before the fp16 hi/lo split, so the core
  // has fp32's dynamic range (attention.py:224-246 has no window); the logits and the context are scaled back exactly.  The
  // shifts are clamped to +-60 so that the combined logit factor stays a normal fp32 number.
  float sc_q = 1.f, sc_k = 1.f, sc_v = 1.f, qk_scale = 0.125f, o_scale = 1.f;      // (wave-uniform: scalar registers)
  if (a.qkv_absmax) {
    int sft[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned bits = a.qkv_absmax[2 * i];          // (max, min) pairs as launch_row_range leaves them
      const int e = (int)(bits >> 23) & 0xff;
      int s_ = (e > 0 && e < 255) ? 10 - (e - 127) : 0;
      s_ = s_ > 60 ? 60 : (s_ < -60 ? -60 : s_);
      sft[i] = s_;
    }
    sc_q = __uint_as_float((unsigned)(sft[0] + 127) << 23);
    sc_k = __uint_as_float((unsigned)(sft[1] + 127) << 23);
    sc_v = __uint_as_float((unsigned)(sft[2] + 127) << 23);
    qk_scale = __uint_as_float((unsigned)(-sft[0] - sft[1] - 3 + 127) << 23);       // 2^-3 / (sc_q sc_k)
    o_scale = __uint_as_float((unsigned)(-sft[2] + 127) << 23);
  }

  // ---- descriptors & per-lane DMA offsets --------------------------------------------------------------------
  const unsigned kv_span = (unsigned)((((size_t)(a.Tk - 1)) * a.ldk + (size_t)a.H * 64) * 4);
  const unsigned vv_span = (unsigned)((((size_t)(a.Tk - 1)) * a.ldv + (size_t)a.H * 64) * 4);
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(a.K + (size_t)b * a.k_bs), 0, kv_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(a.V + (size_t)b * a.v_bs), 0, vv_span, 0x00020000);
  // instruction x (0..3) of this wave covers tile rows 4*(wave + 4x) .. +3 ; lane -> row (lane>>4), chunk (lane&15)
  unsigned k_off[4], v_off[4];
  int t_row[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int r = 4 * (wave + 4 * x) + (lane >> 4);
    const int pc = lane & 15;
    t_row[x] = r;
    k_off[x] = (unsigned)(((size_t)r * a.ldk + hd * 64 + 4 * (pc ^ (r & 15))) * 4);
    v_off[x] = (unsigned)(((size_t)r * a.ldv + hd * 64 + 4 * pc) * 4);
  }
  auto issue_tile = [&](int kt, int slot, bool do_k = true, bool do_v = true) {
    char* sb = smem + slot * SLOT;
    const unsigned ks = (unsigned)((size_t)kt * KT * a.ldk * 4), vs = (unsigned)((size_t)kt * KT * a.ldv * 4);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const bool ok = kt * KT + t_row[x] < a.Tk;
      if (do_k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_ptr_t)(sb + (wave + 4 * x) * 1024), 16, ok ? k_off[x] + ks : kOob, 0, 0, 0);
      if (do_v) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_ptr_t)(sb + KT * 256 + (wave + 4 * x) * 1024), 16, ok ? v_off[x] + vs : kOob, 0, 0, 0);
    }
  };

  // The first two tiles are requested before anything depends on the length arrays (their loads would
  // otherwise sit on the critical path: kernarg -> lengths -> first DMA).  A tile that turns out to be
  // skippable is still drained by the vmcnt(0) of the last loop iteration.
  issue_tile(0, 0);
  if (ntiles_all > 1) issue_tile(1, 1);

  // ---- Q fragment for the f16 matrix pipe: lane (i, g = half) keeps d = 16t + 8g .. +7, t = 0..3, as hi/lo fp16 pairs.
  // Every product of QK^T and PV is evaluated as a 3-term hi/lo split (fp32 accumulate): fp32-class accuracy at the
  // f16 MFMA rate; softmax statistics stay in fp32 registers.
  f16x8 qhi[4], qlo[4];
  {
    const int iq = q0 + l31;
    const bool qok = iq < a.Tq;
    const float* qp = a.Q + (size_t)b * a.q_bs + (size_t)(qok ? iq : 0) * a.ldq + hd * 64 + half * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 x0, x1;
      if (qok) { x0 = *reinterpret_cast<const f32x4*>(qp + t * 16); x1 = *reinterpret_cast<const f32x4*>(qp + t * 16 + 4); }
      else { x0[0] = x0[1] = x0[2] = x0[3] = 0.f; x1 = x0; }
      split8(x0 * sc_q, x1 * sc_q, qhi[t], qlo[t]);
    }
  }
  const int qlen = a.q_len ? a.q_len[b] : a.Tq;
  const int klen = a.k_len ? a.k_len[b] : a.Tk;
  int ntiles = ntiles_all;
  {
    int row_hi = Q0 + 64; if (row_hi > a.Tq) row_hi = a.Tq;
    if (!(row_hi > qlen || klen <= 0)) {              // all query rows valid: masked keys have weight exactly 0
      int kmax = klen;
      if (a.causal && row_hi < kmax) kmax = row_hi;
      ntiles = (kmax + KT - 1) / KT;
    }
  }
  const float tau = a.temperature;
  const bool use_tau = tau != 1.0f;
  const int iq = q0 + l31;
  const bool qvalid = iq < qlen;
  // K operand read offsets (bytes) inside a tile: row 32*kh + l31, logical 16-byte chunks 4t + 2*half (+1)
  int k_rd[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    k_rd[t][0] = (32 * kh + l31) * 256 + (((4 * t + 2 * half) ^ (l31 & 15)) << 4);
    k_rd[t][1] = (32 * kh + l31) * 256 + (((4 * t + 2 * half + 1) ^ (l31 & 15)) << 4);
  }

  // S^T block of this wave for tile kt: keys kt*64 + 32*kh + frow(r, half), query l31
  auto qk_block = [&](const char* Ks, int kt, f32x16& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 k0 = *reinterpret_cast<const f32x4*>(Ks + k_rd[t][0]);
      const f32x4 k1 = *reinterpret_cast<const f32x4*>(Ks + k_rd[t][1]);
      f16x8 khi, klo;
      split8(k0 * sc_k, k1 * sc_k, khi, klo);
      st = mfma3(khi, klo, qhi[t], qlo[t], st);
    }
    const int j0 = kt * KT + 32 * kh;                 // first key of this wave's block
    // wave-uniform fast path: every (query, key) of the block is valid and unmasked -> scale only
    const bool plain = (!a.causal || j0 + 31 <= q0) && !use_tau && j0 + 32 <= klen && j0 + 32 <= a.Tk && q0 + 32 <= qlen;
    if (plain) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= qk_scale;   // / sqrt(64) (and the operand scales), exact
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = j0 + frow(r, half);
        float s = st[r] * qk_scale;
        if (use_tau) s = s / tau;
        const bool ok = qvalid && (j < klen) && (!a.causal || j <= iq);
        s = ok ? s : kMaskFill;                       // attention.py:240
        if (j >= a.Tk) s = -INFINITY;                 // key does not exist
        st[r] = s;
      }
    }
  };
  f32x16 O[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
  // O += P.V on the f16 pipe.  k-slot (step t', lane-half g, e) carries key frow(8t'+e, g): P's registers 8t'..8t'+7 are
  // the A operand as they are; the B operand gathers the matching V rows (two runs of 4 consecutive rows).
  auto pv_block = [&](const char* Vs, const f32x16& p) {
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      f32x4 p0, p1, va0, va1, vb0, vb1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p0[e] = p[8 * tp + e]; p1[e] = p[8 * tp + 4 + e];
        const float* r0 = reinterpret_cast<const float*>(Vs) + (32 * kh + frow(8 * tp + e, half)) * 64 + l31;
        const float* r1 = reinterpret_cast<const float*>(Vs) + (32 * kh + frow(8 * tp + 4 + e, half)) * 64 + l31;
        va0[e] = r0[0]; va1[e] = r1[0]; vb0[e] = r0[32]; vb1[e] = r1[32];
      }
      f16x8 phi, plo, vhi, vlo;
      split8(p0, p1, phi, plo);
      split8(va0 * sc_v, va1 * sc_v, vhi, vlo);
      O[0] = mfma3(phi, plo, vhi, vlo, O[0]);
      split8(vb0 * sc_v, vb1 * sc_v, vhi, vlo);
      O[1] = mfma3(phi, plo, vhi, vlo, O[1]);
    }
  };

  float m_run = -INFINITY, l_run = 0.f;
  stamp(1);

  if (!ALI) {
    // ---- online softmax over this wave's half of every tile ------------------------------------------------------
    for (int kt = 0; kt < ntiles; ++kt) {
      const int slot = kt & 1;
      // kt = 0: the two prologue tiles and the Q fragment are all older than anything else -> drain;
      // later: tile kt landed, tile kt+1 (8 DMA instructions of this wave) may stay in flight
      if (kt > 0 && kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* Ks = smem + slot * SLOT;
      const char* Vs = Ks + KT * 256;
      if (wave_active) {
        f32x16 st;
        qk_block(Ks, kt, st);
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        // finite floor: a wave whose key half holds no existing key yet (all -inf) must produce p = 0, not NaN
        const float m_new = fmaxf(fmaxf(m_run, mt), -3.0e38f);
        const float alpha = fast_exp(m_run - m_new);  // 0 on the first tile
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = fast_exp(st[r] - m_new); st[r] = p; ps += p; }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        if (kt > 0 && __any(m_new != m_run)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ar = __shfl(alpha, frow(r, half), 64);
            O[0][r] *= ar;
            O[1][r] *= ar;
          }
        }
        m_run = m_new;
        pv_block(Vs, st);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                   // everyone is done with this slot
      asm volatile("" ::: "memory");
      if (kt + 2 < ntiles) issue_tile(kt + 2, slot);
    }
  } else if (NT > 2) {
    // ---- alignments requested, Tk <= 64 NT (round 2): the logits of EVERY tile stay in registers (16 per tile and wave) -----------
    // Phase 1 walks the K tiles through the ring (the V regions of the two slots already hold V tiles 0 and 1 from the prologue),
    // phase 2 is the softmax over the whole row (statistics of the two key halves exchanged through LDS), phase 3 walks the V
    // tiles, and the probabilities leave at the very end: no load is ever issued behind a store (vmcnt counts both, in order).
    f32x16 st[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt < ntiles) {                                // (workgroup-uniform)
        if (kt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // prologue tiles, the Q fragment
        else if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K tile kt landed, K tile kt + 1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wave_active) qk_block(smem + (kt & 1) * SLOT, kt, st[kt]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // everyone is done with this K slot
        asm volatile("" ::: "memory");
        if (kt + 2 < ntiles) issue_tile(kt + 2, kt & 1, true, false);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[t][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float* xch = xarea;                               // [2 rounds][4 waves][32]
    if (half == 0) xch[wave * 32 + l31] = mt;
    __syncthreads();
    m_run = fmaxf(fmaxf(mt, xch[(wave ^ 2) * 32 + l31]), -3.0e38f);
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float p = fast_exp(st[t][r] - m_run); st[t][r] = p; ps += p; }
    ps += __shfl_xor(ps, 32, 64);
    if (half == 0) xch[128 + wave * 32 + l31] = ps;
    __syncthreads();
    l_run = ps + xch[128 + (wave ^ 2) * 32 + l31];
    {
      const float linv = 1.0f / l_run;                // softmax, attention.py:242 (one true division per row, then products)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] *= linv;
    }
    // phase 3: O += P.V over the V tiles (tiles 0 and 1 are resident; tile kt + 2 replaces tile kt)
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (kt < ntiles) {
        if (kt >= 2) {
          if (kt + 1 < ntiles) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        if (wave_active) pv_block(smem + (kt & 1) * SLOT + KT * 256, st[kt]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // everyone is done with this V slot
        asm volatile("" ::: "memory");
        if (kt + 2 < ntiles) issue_tile(kt + 2, kt & 1, false, true);
      }
    }
    // the probabilities: every 32 x 32 block through a wave-private LDS transpose (the K region of slot 0 is free now),
    // 128-byte row pieces out; tiles this workgroup skipped carry weight exactly 0
    if (wave_active) {
      float* Pw = scratch + wave * 1024;              // [32 queries][32 keys], XOR-swizzled columns
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < ntiles_all) {
          const bool have = t < ntiles;
#pragma unroll
          for (int r = 0; r < 16; ++r) Pw[l31 * 32 + (frow(r, half) ^ l31)] = have ? st[t][r] : 0.f;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int rr = 8 * x + (lane >> 3), kc = lane & 7;
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = Pw[rr * 32 + ((4 * kc + e) ^ rr)];
            const int qrow = q0 + rr, key = t * KT + 32 * kh + 4 * kc;
            if (qrow < a.Tq) {
              float* dst = a.ali + (((size_t)b * a.H + hd) * a.Tq + qrow) * a.Tk + key;
              if (key + 3 < a.Tk && !(a.Tk & 3)) { f32x4 o4 = {v4[0], v4[1], v4[2], v4[3]}; *reinterpret_cast<f32x4*>(dst) = o4; }
              else
#pragma unroll
                for (int e = 0; e < 4; ++e) if (key + e < a.Tk) dst[e] = v4[e];
            }
          }
        }
      }
    }
    m_run = 0.f;                                       // partials already share one scale: merge = plain sum
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    // ---- alignments requested (Tk <= 128: at most two tiles, logits stay in registers) ------------------------------
    f32x16 st0, st1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st0[r] = -INFINITY; st1[r] = -INFINITY; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave_active) {
      qk_block(smem, 0, st0);
      if (ntiles > 1) qk_block(smem + SLOT, 1, st1);
    }
    // row maximum over ALL keys: this wave's part, then the partner's (kh ^ 1) through LDS
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, fmaxf(st0[r], st1[r]));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float* xch = xarea;                               // [2 rounds][4 waves][32]
    if (half == 0) xch[wave * 32 + l31] = mt;
    __syncthreads();
    m_run = fmaxf(fmaxf(mt, xch[(wave ^ 2) * 32 + l31]), -3.0e38f);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st0[r] = fast_exp(st0[r] - m_run); st1[r] = fast_exp(st1[r] - m_run);
      ps += st0[r] + st1[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    if (half == 0) xch[128 + wave * 32 + l31] = ps;
    __syncthreads();
    l_run = ps + xch[128 + (wave ^ 2) * 32 + l31];
    {
      const float linv = 1.0f / l_run;                // softmax, attention.py:242 (one true division per row, then products)
#pragma unroll
      for (int r = 0; r < 16; ++r) { st0[r] *= linv; st1[r] *= linv; }
    }
    if (wave_active) {
      // alignment rows: transpose each 32x32 block through LDS, store 128-byte row pieces with 16-byte lanes
      float* Pw = scratch + wave * 1024;              // [32 queries][32 keys], XOR-swizzled columns
      for (int t = 0; t < ntiles_all; ++t) {
        const f32x16& p = t == 0 ? st0 : st1;
        const bool have = t < ntiles;                 // skipped tiles: weight exactly 0
#pragma unroll
        for (int r = 0; r < 16; ++r) Pw[l31 * 32 + (frow(r, half) ^ l31)] = have ? p[r] : 0.f;
        // lane -> row rr = 8*x + (lane>>3), key chunk kc = lane&7 (4 keys)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int rr = 8 * x + (lane >> 3), kc = lane & 7;
          float v4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = Pw[rr * 32 + ((4 * kc + e) ^ rr)];
          const int qrow = q0 + rr, key = t * KT + 32 * kh + 4 * kc;
          if (qrow < a.Tq) {
            float* dst = a.ali + (((size_t)b * a.H + hd) * a.Tq + qrow) * a.Tk + key;
            if (key + 3 < a.Tk && !(a.Tk & 3)) { f32x4 o4 = {v4[0], v4[1], v4[2], v4[3]}; __builtin_nontemporal_store(o4, reinterpret_cast<f32x4*>(dst)); }
            else
#pragma unroll
              for (int e = 0; e < 4; ++e) if (key + e < a.Tk) dst[e] = v4[e];
          }
        }
      }
      pv_block(smem + KT * 256, st0);
      if (ntiles > 1) pv_block(smem + SLOT + KT * 256, st1);
    }
    m_run = 0.f;                                       // partials already share one scale: merge = plain sum
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }

  stamp(2);
  // ---- merge the wave pair (same query tile, other key half): each wave finalises dv block nb = kh ---------------------
  // give = the block the partner finalises; layout [wave][16 regs][64 lanes]
  float* mg = scratch;                                 // 4 waves x 1024 floats
  float* ml = xarea + 256;                             // m,l of every wave: [4][2][32]
  {
    const int give = 1 - kh;
#pragma unroll
    for (int r = 0; r < 16; ++r) mg[wave * 1024 + r * 64 + lane] = O[give][r];
    if (half == 0) { ml[wave * 64 + l31] = m_run; ml[wave * 64 + 32 + l31] = l_run; }
  }
  __syncthreads();
  if (wave_active) {
    const int pw = wave ^ 2;
    const float m_o = ml[pw * 64 + l31], l_o = ml[pw * 64 + 32 + l31];
    float sa = 1.f, sb = 1.f, linv = 1.f;
    if (!ALI) {
      const float m_f = fmaxf(m_run, m_o);
      sa = fast_exp(m_run - m_f);                      // floor - finite -> 0 when this half saw no key
      sb = fast_exp(m_o - m_f);
      linv = 1.0f / (l_run * sa + l_o * sb);
      // row statistics for the recomputing backward pass of the training step (one wave of the pair, one lane half)
      if (a.row_max && kh == 0 && half == 0 && q0 + l31 < a.Tq) {
        const size_t ri = ((size_t)b * a.H + hd) * a.Tq + q0 + l31;
        a.row_max[ri] = m_f;
        a.row_linv[ri] = linv;
      }
    }
    const int nb = kh;
    float* ob = a.ctx + (size_t)b * a.o_bs + hd * 64 + nb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int src = frow(r, half);                   // per-query factors live in lane = query index
      const float fa = __shfl(sa * linv, src, 64) * o_scale, fb = __shfl(sb * linv, src, 64) * o_scale;
      const float o = O[nb][r] * fa + mg[pw * 1024 + r * 64 + lane] * fb;
      const int row = q0 + src;
      if (row < a.Tq) __builtin_nontemporal_store(o, ob + (size_t)row * a.ldo);   // nt: streams out during the kernel (see common.h)
    }
  }
  stamp(3);
  if (ts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(4); if (tid == 0) { ts[5] = ntiles; unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); ts[7] = xcc; } }
}

// Alignments of a call with more than 512 keys (no BASELINE configuration has one): P[b][h][q][:] = exp(s - max) / sum rebuilt from
// Q, K and the row statistics the alignment-free pass left, with the masking of qk_block above.  One wave per query row, plain fp32
// dot products (a fallback shape: 1e-6 of the 3-term products of the main kernel).
__global__ void __launch_bounds__(256)
attn_probs_kernel(const AttnArgs a) {
  __shared__ float qs[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave, hd = blockIdx.y, b = blockIdx.z;
  if (q >= a.Tq) return;
  const int qlen = a.q_len ? a.q_len[b] : a.Tq, klen = a.k_len ? a.k_len[b] : a.Tk;
  qs[wave][lane] = a.Q[(size_t)b * a.q_bs + (size_t)q * a.ldq + hd * 64 + lane];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (wave-private row: the wave's own lanes are its only readers)
  const size_t ri = ((size_t)b * a.H + hd) * a.Tq + q;
  const float m = a.row_max[ri], linv = a.row_linv[ri];
  const bool use_tau = a.temperature != 1.0f;
  float* out = a.ali + ri * a.Tk;
  for (int j = lane; j < a.Tk; j += 64) {
    const float* kp = a.K + (size_t)b * a.k_bs + (size_t)j * a.ldk + hd * 64;
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
      const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + d);
      dot += (kv[0] * qs[wave][d] + kv[1] * qs[wave][d + 1]) + (kv[2] * qs[wave][d + 2] + kv[3] * qs[wave][d + 3]);
    }
    float sv = dot * 0.125f;
    if (use_tau) sv = sv / a.temperature;
    const bool ok = q < qlen && j < klen && (!a.causal || j <= q);
    sv = ok ? sv : kMaskFill;                                   // attention.py:240
    out[j] = fast_exp(sv - m) * linv;
  }
}

// operands inside the 2 GiB offset range of the K / V buffer descriptors
bool attention2_supported(const AttnArgs& a) {
  const size_t lim = (size_t)1 << 31;
  if (((size_t)a.Tk * a.ldk + 256) * 4 >= lim || ((size_t)a.Tk * a.ldv + 256) * 4 >= lim) return false;
  return true;
}

hipError_t launch_attention2(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  static const char* ts_path = getenv("VNR_ATTN_TS");
  const int nqb = (a.Tq + 63) / 64;
  unsigned long long* dts = nullptr;
  const size_t nts = (size_t)nqb * a.H * a.B * 8;
  if (ts_path) { if (hipMalloc((void**)&dts, nts * 8) != hipSuccess) return hipErrorOutOfMemory; (void)hipMemset(dts, 0, nts * 8); a.dbg_ts = dts; }
  struct Dump { const char* p; unsigned long long* d; size_t n; hipStream_t s; const AttnArgs* a; int nqb;
    ~Dump() { if (!p) return; (void)hipStreamSynchronize(s); std::vector<unsigned long long> h(n); (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); (void)hipFree(d);
      FILE* f = fopen(p, "ab"); if (f) { int hdr[8] = {a->B, a->H, a->Tq, a->Tk, a->causal, a->ali ? 1 : 0, (int)(n / 8), nqb}; fwrite(hdr, 4, 8, f); fwrite(h.data(), 8, n, f); fclose(f); } } } dump{ts_path, dts, nts, s, &a, nqb};
  const size_t lds = 2 * (2 * 64 * 256) + (256 + 256) * sizeof(float);
  dim3 grid(nqb * a.H * a.B);
  if (a.ali && a.Tk > 256 && a.Tk <= 384) {          // (six / seven tiles of logits in registers)
    auto k = attn2_kernel<true, 6>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, dim3(256), lds, s, a, nqb);
  } else if (a.ali && a.Tk > 384 && a.Tk <= 448) {
    auto k = attn2_kernel<true, 7>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, dim3(256), lds, s, a, nqb);
  } else if (a.ali && a.Tk > 128) {
    auto k = attn2_kernel<true, 4>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, dim3(256), lds, s, a, nqb);
  } else if (a.ali) {
    auto k = attn2_kernel<true>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, dim3(256), lds, s, a, nqb);
  } else {
    auto k = attn2_kernel<false>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, dim3(256), lds, s, a, nqb);
  }
  return hipGetLastError();
}

// entry point of every attention call on fp32 operands (common.h)
hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return hipErrorInvalidValue;
  if ((a.ldq & 3) || (a.ldk & 3) || (a.ldv & 3)) return hipErrorInvalidValue;
  if (!attention2_supported(a)) return hipErrorInvalidValue;
  if (a.ali && a.Tk > 448) {                           // two passes: context + row statistics, then the probabilities (round 6: from 449 keys --
                                                       // the eight-tile register form of attn2_kernel spilled 8 VGPRs and is gone)
    if (!a.row_max || !a.row_linv) return hipErrorInvalidValue;
    AttnArgs f = a;
    f.ali = nullptr;
    const hipError_t e = launch_attention2(f, s);
    if (e != hipSuccess) return e;
    vnr_launch(attn_probs_kernel, dim3((a.Tq + 3) / 4, a.H, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  return launch_attention2(a, s);
}

}  // namespace vnr
