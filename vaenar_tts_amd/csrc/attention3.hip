// Attention core on producer-split operand images (gfx950, split-fp16 MFMA 32x32x16): attn3_kernel<ALI> for non-causal calls with
// Tk <= 128 (cross-attention over the text, alignments optional), attn3g_kernel for any Tk and the causal mask (further below).
//
// Same contract as attention.hip / attention2.hip (reference modules/attention.py:221-246: scale, key ^ query mask with the
// -2**32+1 fill, softmax, . V, optional alignments) for the cross-attention of a CrossAttentionBLK (attention.py:445-447),
// whose memory is the text (Tk = T_text <= 128 at every BASELINE configuration).
//
// Why a third kernel.  attention2 is bound by VALU issue, not by HBM: per wave ~1400 vector instructions against 48 MFMAs,
// a third of them spent splitting fp32 K / V / Q into the fp16 hi/lo pairs the f16 matrix pipe wants -- work repeated by every
// workgroup that re-reads the same K/V (14x for the decoder cross-attention).  Here the PRODUCERS of Q, K, V (the chain kernel's
// query stage, the K|V panel GEMM) store the operands already split, in the layouts of common.h "attention operand images":
//   * the images are operand-major: every Q / K / V operand is ONE fully coalesced 1 KiB wave load straight from global memory
//     into MFMA operand position (no LDS, no conversion, no gather: V's k-slot order matches P's register order);
//   * so the kernel needs no K/V staging, no DMA ring, no barrier before the first MFMA.
//
// Decomposition.  A workgroup = 4 waves = ONE 32-query tile of one (batch, head); wave w owns keys 32w .. 32w+31 (Tk <= 128 => at
// most 4 key blocks): 13 x B x H workgroups at Tq = 400 instead of 7 x B x H -- finer grains, all co-resident (4 per CU).
//   S^T = K.Q^T (lane = query: softmax statistics are in-lane + one cross-half shuffle, P is already the A operand of P.V);
//   the four waves exchange (row max, row sum) once through LDS: p = 2^(t - m_w) locally, then one factor 2^(m_w - M) / L;
//   alignments leave as 128-byte row pieces after a wave-private 32x32 LDS transpose (16-byte lanes, XOR-swizzled chunks);
//   the partial O^T = V^T.P^T tiles (a lane owns a query; computed from the unnormalised p BEFORE the statistics barrier, scaled
//   afterwards) are summed through LDS as [query][64 d] rows, each wave storing 8 rows as 256-byte pieces.
// Logits are kept in the log2 domain (scale 1/8 . log2(e) / temperature folded into one multiply, exp = v_exp_f32): one
// rounding of t = s.log2(e) differs from the reference's exp(s - max) by <= 2^-24 . |t| relative -- below 3e-6 for |s| < 40.
#include "common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace vnr {

namespace {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int frow3(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ f32x16 mfma3x(const h8& ah, const h8& al, const h8& bh, const h8& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
  return c;
}
__device__ __forceinline__ void split8x(const float* x, h8& hi, h8& lo) {
  { const vnr_f8 xs_ = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]}; vnr_split(xs_, hi, lo); }
}
constexpr float kLog2e = 1.44269504088896340736f;
constexpr int kXchg = 8192;                         // bytes of exchange space per wave
}  // namespace

// Persistent form (round 2): a workgroup owns a CHUNK of consecutive query tiles of one (batch, head) -- `nchunk` chunks per pair,
// grid = B.H.nchunk -- instead of one tile.  K and V of the wave's key block are loaded ONCE and stay in registers; the Q tile of
// the next iteration is requested right after the S^T MFMAs of the current one (its registers are dead by then), so from the
// second tile on no operand load is exposed, and the alignment / context stores of tile i drain under the MFMAs of tile i+1
// instead of all 19.7 MB leaving at the same moment at the end of 832 simultaneous one-tile workgroups.
// LDS hazards across iterations: the exchange space is only written after barrier 1 of an iteration, which every wave reaches
// after its merge reads of the previous iteration -- no extra barrier, no double buffer.
// SKEL (measurement only, VNR_ATTN3_SKEL; bench.py: roofline_cross_attention.floor): the two halves of the kernel apart, on the SAME grid, the
// same loads and the same store addresses -- 1 = traffic only (operand images read, alignments and context written; no MFMA, no
// softmax, no LDS transposes: what the memory system alone needs for this access pattern), 2 = all the arithmetic and LDS traffic
// but no global store.  The slower of the two is the floor of THIS decomposition; results are meaningless in both.
template <bool ALI, int SKEL = 0>
__global__ void __launch_bounds__(256, 2)
attn3_kernel(const Attn3Args a, int nqt, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stats = reinterpret_cast<float*>(smem + 4 * kXchg);      // [4 waves][m | l][32]

  // XCD-aware work map (as attention2): all chunks of one (batch, head) share K/V -> same id mod 8 = same L2
  const int npairs = a.B * a.H;
  const int wg = blockIdx.x;
  int pair, chunk;
  if ((npairs & 7) == 0) {
    const int xcd = wg & 7, j = wg >> 3, ppx = npairs >> 3;
    chunk = j / ppx;
    pair = (j - chunk * ppx) * 8 + xcd;
  } else {
    chunk = wg / npairs;
    pair = wg - chunk * npairs;
  }
  const int qt_begin = (int)(((long long)chunk * nqt) / nchunk), qt_end = (int)(((long long)(chunk + 1) * nqt) / nchunk);
  if (qt_begin >= qt_end) return;                     // (workgroup-uniform)
  unsigned long long* ts = a.dbg_ts ? a.dbg_ts + (size_t)blockIdx.x * 32 : nullptr;      // measurement only (VNR_ATTN3_TS)
  auto stamp = [&](int i) { if (ts && threadIdx.x == 0 && i < 32) ts[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  const int b = pair / a.H, hd = pair - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int kb0 = 32 * wave;
  const bool active = kb0 < a.Tk;                     // wave-uniform: this wave's key block exists
  const bool partial = kb0 + 32 > a.Tk;               // ... but not all of it

  // ---- operand loads: every one a fully coalesced 1 KiB wave read of an operand-major image tile (common.h).  K / V go straight
  //      to registers, once.  The Q tile travels through LDS: each wave fetches a quarter of the NEXT tile (2 KiB) right after its
  //      S^T MFMAs, parks it in LDS after the statistics barrier and every wave reads the whole tile back at the top of the next
  //      iteration -- so the only vector-memory wait inside the loop sits BEFORE the iteration's stores are issued, and the
  //      alignment / context stores are never waited for (vmcnt counts stores too: a register prefetch consumed at the loop top
  //      would drain them every iteration).
  h8 qhi[4], qlo[4], khi[4], klo[4], vhi[2][2], vlo[2][2];
  const int ttq = (a.Tq + 31) >> 5, ttk = (a.Tk + 31) >> 5;
  const char* qbase = a.Qi + (size_t)(b * a.H + hd) * ttq * kAoiTile + wave * 2048 + lane * 16;
  char* qlds = smem + 4 * kXchg + 4 * 64 * sizeof(float);        // [8 KiB] the current Q tile, image layout
  h8 qn0, qn1;                                                     // this wave's quarter of the next Q tile
  auto fetch_q = [&](int qidx) {
    const char* qp = qbase + (size_t)qidx * kAoiTile;
    qn0 = *reinterpret_cast<const h8*>(qp);
    qn1 = *reinterpret_cast<const h8*>(qp + 1024);
  };
  auto park_q = [&]() {
    *reinterpret_cast<h8*>(qlds + wave * 2048 + lane * 16) = qn0;
    *reinterpret_cast<h8*>(qlds + wave * 2048 + 1024 + lane * 16) = qn1;
  };
  fetch_q(qt_begin);
  {
    const size_t kt = ((size_t)(b * a.H + hd) * ttk + (active ? wave : 0)) * kAoiTile + lane * 16;
    const char* kp = a.Ki + kt;
    const char* vp = a.Vi + kt;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      khi[t] = *reinterpret_cast<const h8*>(kp + 1024 * t);
      klo[t] = *reinterpret_cast<const h8*>(kp + 4096 + 1024 * t);
    }
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        vhi[tp][nb] = *reinterpret_cast<const h8*>(vp + (tp * 2 + nb) * 1024);
        vlo[tp][nb] = *reinterpret_cast<const h8*>(vp + 4096 + (tp * 2 + nb) * 1024);
      }
  }
  const int qlen = a.q_len ? a.q_len[b] : a.Tq;
  const int klen = a.k_len ? a.k_len[b] : a.Tk;
  __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): Q quarter, K and V are in registers; nothing is outstanding at the loop top
  stamp(1);
  if (active && partial) {                            // positions past Tk hold whatever the workspace held: force zeros (once)
#pragma unroll
    for (int tp = 0; tp < 2; ++tp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = kb0 + 16 * tp + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (key >= a.Tk) { vhi[tp][0][e] = vhi[tp][1][e] = vlo[tp][0][e] = vlo[tp][1][e] = (_Float16)0.f; }
      }
  }
  park_q();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const float c = (a.temperature != 1.0f) ? 0.125f * kLog2e / a.temperature : 0.125f * kLog2e;
  char* xb = smem + wave * kXchg;                                  // this wave's exchange space
  stamp(2);

#pragma unroll 1
  for (int qidx = qt_begin; qidx < qt_end; ++qidx) {
    const int q0 = qidx * 32;
    const bool more = qidx + 1 < qt_end;
    const int sb = 4 + 6 * (qidx - qt_begin);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qhi[t] = *reinterpret_cast<const h8*>(qlds + 1024 * t + lane * 16);
      qlo[t] = *reinterpret_cast<const h8*>(qlds + 4096 + 1024 * t + lane * 16);
    }
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    if (SKEL == 1) {                                  // every loaded register is consumed (no load may be optimised away), nothing else happens
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float u = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) u += (float)khi[t][r & 7] + (float)klo[t][r & 7] + (float)qhi[t][r & 7] + (float)qlo[t][r & 7];
        st[r] = u;
      }
    } else if (active) {
#pragma unroll
      for (int t = 0; t < 4; ++t) st = mfma3x(khi[t], klo[t], qhi[t], qlo[t], st);
    }
    stamp(sb);
    if (more) fetch_q(qidx + 1);                      // lands under this tile's softmax and P.V MFMAs

    // ---- logits (log2 domain) and masks ----------------------------------------------------------------------------------
    const int iq = q0 + l31;
    if (SKEL == 1) {
    } else if (!partial && kb0 + 32 <= klen && q0 + 32 <= qlen) {          // wave-uniform: nothing masked in this block
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= c;
    } else {
      const bool qvalid = iq < qlen;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kb0 + frow3(r, half);
        float s = st[r] * c;
        s = (qvalid && j < klen) ? s : kMaskFill * kLog2e;          // attention.py:240
        if (j >= a.Tk) s = -INFINITY;                                 // key does not exist
        st[r] = s;
      }
    }
    float mt = -INFINITY, m_w = 0.f, ls = 1.f;
    if (SKEL != 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    m_w = fmaxf(mt, -3.0e38f);                            // finite floor: a wave without keys gives p = 0, not NaN
    ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(st[r] - m_w); st[r] = p; ls += p; }
    ls += __shfl_xor(ls, 32, 64);
    if (half == 0) { stats[wave * 64 + l31] = m_w; stats[wave * 64 + 32 + l31] = ls; }
    }

    // ---- O^T partial = V^T.P^T over this wave's 32 keys, from the UNNORMALISED p (relative to this wave's maximum): issued
    //      before the statistics barrier so that the exchange latency hides under the MFMAs.  Transposed accumulation (as in
    //      attn3g_kernel): a lane owns one query, so the factor 2^(m_w - M) / L is an in-lane product afterwards.
    f32x16 O[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
    if (SKEL == 1) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nb][r] = (float)vhi[r >> 3][nb][r & 7] + (float)vlo[r >> 3][nb][r & 7];
    } else if (active) {
#pragma unroll
      for (int tp = 0; tp < 2; ++tp) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = st[8 * tp + e];
        h8 phi, plo;
        split8x(pv, phi, plo);
        O[0] = mfma3x(vhi[tp][0], vlo[tp][0], phi, plo, O[0]);
        O[1] = mfma3x(vhi[tp][1], vlo[tp][1], phi, plo, O[1]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // every wave has read the current Q tile (its S^T MFMAs precede the barrier): park the next one.  The wait below is the
    // only vmcnt wait of the iteration and precedes the iteration's stores
    stamp(sb + 1);
    if (more) { __builtin_amdgcn_s_waitcnt(0x0F70); park_q(); }
    stamp(sb + 2);
    float f = 1.f;
    if (SKEL != 1) {
      float mw[4], lw[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) { mw[w] = stats[w * 64 + l31]; lw[w] = stats[w * 64 + 32 + l31]; }
      const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
      float L = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) L += lw[w] * __builtin_amdgcn_exp2f(mw[w] - M);
      f = __builtin_amdgcn_exp2f(m_w - M) * (1.0f / L);              // softmax, attention.py:242 (one division per row)
    }

    if (ALI && active) {
      // alignment rows: normalise, 32x32 transpose through LDS (chunk = 4 keys, XOR-swizzled by row), 128-byte row pieces out
      float* xw = reinterpret_cast<float*>(xb);
      if (SKEL != 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 p4 = {st[4 * j] * f, st[4 * j + 1] * f, st[4 * j + 2] * f, st[4 * j + 3] * f};
        *reinterpret_cast<f32x4*>(xw + l31 * 32 + (((2 * j + half) ^ (l31 & 7)) << 2)) = p4;
      }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const int rr = 8 * x + (lane >> 3), kc = lane & 7;
        f32x4 v4;
        if (SKEL == 1) v4 = f32x4{st[4 * x], st[4 * x + 1], st[4 * x + 2], st[4 * x + 3]};
        else v4 = *reinterpret_cast<const f32x4*>(xw + rr * 32 + ((kc ^ (rr & 7)) << 2));
        const int qrow = q0 + rr, key = kb0 + 4 * kc;
        if (qrow < a.Tq && (SKEL != 2 || a.Tq < 0)) {
          float* dst = a.ali + (((size_t)b * a.H + hd) * a.Tq + qrow) * a.Tk + key;
          if (key + 3 < a.Tk && !(a.Tk & 3)) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(dst));
          else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (key + e < a.Tk) dst[e] = v4[e];
        }
      }
    }
    stamp(sb + 3);
    // ---- sum the four partial tiles: rows [query l31][16-byte chunk = nb*8 + 2j + half, XOR-swizzled by row], already scaled ---
    if (SKEL != 1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const f32x4 o4 = {O[nb][4 * jj] * f, O[nb][4 * jj + 1] * f, O[nb][4 * jj + 2] * f, O[nb][4 * jj + 3] * f};
        *reinterpret_cast<f32x4*>(xb + l31 * 256 + (((nb * 8 + 2 * jj + half) ^ (l31 & 15)) << 4)) = o4;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(sb + 4);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int rr = 8 * wave + 4 * u + (lane >> 4), ch = lane & 15;   // query row of the tile, 16-byte chunk (4 channels)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (SKEL == 1) acc = f32x4{O[u][0], O[u][1], O[u][2], O[u][3]};
      else
#pragma unroll
      for (int w = 0; w < 4; ++w) acc += *reinterpret_cast<const f32x4*>(smem + w * kXchg + rr * 256 + ((ch ^ (rr & 15)) << 4));
      const int row = q0 + rr;
      if (row < a.Tq && (SKEL != 2 || a.Tq < 0)) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.ctx + (size_t)b * a.o_bs + (size_t)row * a.ldo + hd * 64 + 4 * ch));
    }
    stamp(sb + 5);
  }
  stamp(30);
  if (ts && threadIdx.x == 0) { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); ts[31] = ((unsigned long long)(xcc & 15) << 32) | (unsigned)(qt_end - qt_begin); }
}

// ---- general kernel: any Tk, optional causal mask, online softmax over the key blocks of a wave ----------------------------
// Same workgroup shape (4 waves = one 32-query tile); wave w walks key blocks w, w+4, w+8, ... with the usual running
// (max, sum) rescaling.  The output is accumulated TRANSPOSED, O^T = V^T.P^T: the V image tile is the A operand as it is, P in
// its S^T accumulator layout is the B operand as it is, and a lane owns one QUERY -- the per-block rescale and the final
// normalisation are in-lane products (no cross-lane broadcast of alpha), and the partial tiles go to LDS as [query][64 d] rows
// so that the merge pass stores 256-byte row pieces.  Blocks that cannot contribute are skipped when every query row of the
// tile is valid (keys >= k_len have weight exactly 0; causal: keys beyond the tile's last row); tiles with padded query rows
// walk every block (their rows are uniform over ALL Tk keys).  The next block's K tile is requested right after
// the S^T MFMAs and its V tile right after the O^T MFMAs, so a block's loads fly under the previous block's softmax.
__global__ void __launch_bounds__(256, 2)
attn3g_kernel(const Attn3Args a, int nqt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stats = reinterpret_cast<float*>(smem + 4 * kXchg);      // [4 waves][m | l][32]
  const int npairs = a.B * a.H;
  const int wg = blockIdx.x;
  int pair, qidx;
  if ((npairs & 7) == 0) {
    const int xcd = wg & 7, j = wg >> 3, ppx = npairs >> 3;
    qidx = j / ppx;
    pair = (j - qidx * ppx) * 8 + xcd;
  } else {
    qidx = wg / npairs;
    pair = wg - qidx * npairs;
  }
  if (a.causal) qidx = nqt - 1 - qidx;                 // heaviest tiles first
  const int b = pair / a.H, hd = pair - b * a.H;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int q0 = qidx * 32;
  const int ttq = (a.Tq + 31) >> 5, ttk = (a.Tk + 31) >> 5;
  const char* kbase = a.Ki + (size_t)(b * a.H + hd) * ttk * kAoiTile + lane * 16;
  const char* vbase = a.Vi + (size_t)(b * a.H + hd) * ttk * kAoiTile + lane * 16;

  h8 qhi[4], qlo[4], khi[4], klo[4], vhi[2][2], vlo[2][2];
  auto load_k = [&](int kb) {
    const char* kp = kbase + (size_t)kb * kAoiTile;
#pragma unroll
    for (int t = 0; t < 4; ++t) { khi[t] = *reinterpret_cast<const h8*>(kp + 1024 * t); klo[t] = *reinterpret_cast<const h8*>(kp + 4096 + 1024 * t); }
  };
  auto load_v = [&](int kb) {
    const char* vp = vbase + (size_t)kb * kAoiTile;
#pragma unroll
    for (int x = 0; x < 4; ++x) { vhi[x >> 1][x & 1] = *reinterpret_cast<const h8*>(vp + 1024 * x); vlo[x >> 1][x & 1] = *reinterpret_cast<const h8*>(vp + 4096 + 1024 * x); }
  };
  {
    const char* qp = a.Qi + ((size_t)(b * a.H + hd) * ttq + qidx) * kAoiTile + lane * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) { qhi[t] = *reinterpret_cast<const h8*>(qp + 1024 * t); qlo[t] = *reinterpret_cast<const h8*>(qp + 4096 + 1024 * t); }
  }
  // first block of this wave is requested before anything depends on the length arrays (block `wave` exists whenever
  // wave < ttk; a block that turns out to be skippable costs one wasted tile read)
  const bool any0 = wave < ttk;
  load_k(any0 ? wave : 0);
  load_v(any0 ? wave : 0);
  const int qlen = a.q_len ? a.q_len[b] : a.Tq;
  const int klen = a.k_len ? a.k_len[b] : a.Tk;
  int nkb = ttk;
  {
    int row_hi = q0 + 32; if (row_hi > a.Tq) row_hi = a.Tq;
    if (row_hi <= qlen && klen > 0) {                  // every query row valid: masked keys have weight exactly 0
      int kmax = klen;
      if (a.causal && row_hi < kmax) kmax = row_hi;
      nkb = (kmax + 31) >> 5;
    }
  }
  const float c = (a.temperature != 1.0f) ? 0.125f * kLog2e / a.temperature : 0.125f * kLog2e;
  const int iq = q0 + l31;
  const bool qvalid = iq < qlen;

  f32x16 O[2];                                         // O^T: lane = query l31, register r of block nb = channel 32 nb + frow3(r, half)
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = wave; kb < nkb; kb += 4) {
    const int kb0 = 32 * kb;
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) st = mfma3x(khi[t], klo[t], qhi[t], qlo[t], st);
    const bool more = kb + 4 < nkb;
    if (more) load_k(kb + 4);                          // flies under this block's softmax and O^T MFMAs
    const bool partial = kb0 + 32 > a.Tk;
    if (!partial && kb0 + 32 <= klen && q0 + 32 <= qlen && (!a.causal || kb0 + 31 <= q0)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] *= c;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kb0 + frow3(r, half);
        float s = st[r] * c;
        s = (qvalid && j < klen && (!a.causal || j <= iq)) ? s : kMaskFill * kLog2e;   // attention.py:240
        if (j >= a.Tk) s = -INFINITY;
        st[r] = s;
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(fmaxf(m_run, mt), -3.0e38f);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 0 on the first block
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(st[r] - m_new); st[r] = p; ps += p; }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    if (kb != wave && __any(m_new != m_run)) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[nb][r] *= alpha;
    }
    m_run = m_new;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      if (partial) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int key = kb0 + 16 * tp + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (key >= a.Tk) { vhi[tp][0][e] = vhi[tp][1][e] = vlo[tp][0][e] = vlo[tp][1][e] = (_Float16)0.f; }
        }
      }
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = st[8 * tp + e];
      h8 phi, plo;
      split8x(pv, phi, plo);
      O[0] = mfma3x(vhi[tp][0], vlo[tp][0], phi, plo, O[0]);
      O[1] = mfma3x(vhi[tp][1], vlo[tp][1], phi, plo, O[1]);
    }
    if (more) load_v(kb + 4);
  }

  // ---- merge: partial tiles as [query l31][16-byte chunk = nb*8 + 2j + half, XOR-swizzled by row] + (m, l) per query ---------
  char* xw = smem + wave * kXchg;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 o4 = {O[nb][4 * j], O[nb][4 * j + 1], O[nb][4 * j + 2], O[nb][4 * j + 3]};
      *reinterpret_cast<f32x4*>(xw + l31 * 256 + (((nb * 8 + 2 * j + half) ^ (l31 & 15)) << 4)) = o4;
    }
  if (half == 0) { stats[wave * 64 + l31] = fmaxf(m_run, -3.0e38f); stats[wave * 64 + 32 + l31] = l_run; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rr = 8 * wave + 4 * u + (lane >> 4), ch = lane & 15;   // query row of the tile, 16-byte chunk (4 channels)
    float mw[4], lw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = stats[w * 64 + rr]; lw[w] = stats[w * 64 + 32 + rr]; }
    const float M = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3]));
    float L = 0.f, f[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { f[w] = __builtin_amdgcn_exp2f(mw[w] - M); L += lw[w] * f[w]; }
    const float linv = 1.0f / L;                       // softmax denominator, attention.py:242
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += *reinterpret_cast<const f32x4*>(smem + w * kXchg + rr * 256 + ((ch ^ (rr & 15)) << 4)) * (f[w] * linv);
    const int row = q0 + rr;
    if (row < a.Tq) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.ctx + (size_t)b * a.o_bs + (size_t)row * a.ldo + hd * 64 + 4 * ch));
  }
}

// fp32 [rows][cols] -> operand images, one thread per 4 columns
__global__ void __launch_bounds__(256) aoi_convert_kernel(const float* src, int ld, int rows, int cols, const AoiDesc d) {
  const int q4 = cols >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * q4) return;
  const int row = (int)(i / q4), col = 4 * (int)(i - (size_t)row * q4);
  const float4 x = *reinterpret_cast<const float4*>(src + (size_t)row * ld + col);
  const float v[4] = {x.x, x.y, x.z, x.w};
  aoi_store4(d, row, col, v);
}

hipError_t launch_aoi_convert(const float* src, int ld, int rows, int cols, const AoiDesc& d, hipStream_t s) {
  if ((cols & 63) || (ld & 3) || d.mode == 0 || d.T <= 0 || d.TT != (d.T + 31) / 32) return hipErrorInvalidValue;
  const size_t n = (size_t)rows * (cols >> 2);
  vnr_launch(aoi_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld, rows, cols, d);
  return hipGetLastError();
}

hipError_t launch_attention3(const Attn3Args& a, hipStream_t s) {
  if (a.Tk <= 0 || a.Tq <= 0 || !a.Qi || !a.Ki || !a.Vi || !a.ctx || (a.ldo & 3)) return hipErrorInvalidValue;
  if (a.ali && (a.Tk > 128 || a.causal)) return hipErrorInvalidValue;     // alignments: single-round kernel only
  const int nqt = (a.Tq + 31) / 32;
  const size_t lds = 4 * kXchg + 4 * 64 * sizeof(float) + kAoiTile;      // exchange space, statistics, the Q tile of attn3_kernel
  dim3 grid(nqt * a.H * a.B);
  static const bool force_general = getenv("VNR_ATTN3_GENERAL") != nullptr;   // A/B switch: looped kernel even where the single-round one applies
  // chunks of query tiles per (batch, head) for the persistent single-round kernel: about two workgroups per CU (512), never
  // more chunks than tiles.  VNR_ATTN3_CHUNKS overrides (0 = one tile per workgroup, the round-1 decomposition)
  static const char* env_chunks = getenv("VNR_ATTN3_CHUNKS");
  int nchunk = nqt;
  if (!env_chunks || atoi(env_chunks) > 0) {
    const int want = env_chunks ? atoi(env_chunks) : (512 + a.B * a.H - 1) / (a.B * a.H);
    nchunk = want < nqt ? (want < 1 ? 1 : want) : nqt;
  }
  dim3 grid1(nchunk * a.H * a.B);
  static const char* ts_path = getenv("VNR_ATTN3_TS");             // measurement only: per-workgroup timeline of the alignment kernel
  if (a.ali && ts_path) {
    Attn3Args aa = a;
    const size_t n = (size_t)grid1.x * 32;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemset(d, 0, n * 8);
    aa.dbg_ts = d;
    vnr_launch(attn3_kernel<true>, grid1, dim3(256), lds, s, aa, nqt, nchunk);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hbuf(n);
    (void)hipMemcpy(hbuf.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FILE* f = fopen(ts_path, "ab");
    if (f) { int hdr[4] = {a.B * a.H, a.Tq, nchunk, (int)grid1.x}; fwrite(hdr, 4, 4, f); fwrite(hbuf.data(), 8, n, f); fclose(f); }
    return hipGetLastError();
  }
  static const int skel = getenv("VNR_ATTN3_SKEL") ? atoi(getenv("VNR_ATTN3_SKEL")) : 0;      // measurement only (see attn3_kernel): WRONG results
  if (a.ali && skel == 1) vnr_launch(attn3_kernel<true, 1>, grid1, dim3(256), lds, s, a, nqt, nchunk);
  else if (a.ali && skel == 2) vnr_launch(attn3_kernel<true, 2>, grid1, dim3(256), lds, s, a, nqt, nchunk);
  else if (a.ali) vnr_launch(attn3_kernel<true>, grid1, dim3(256), lds, s, a, nqt, nchunk);
  else if (!force_general && a.Tk <= 128 && !a.causal) vnr_launch(attn3_kernel<false>, grid1, dim3(256), lds, s, a, nqt, nchunk);
  else vnr_launch(attn3g_kernel, grid, dim3(256), lds, s, a, nqt);
  return hipGetLastError();
}

}  // namespace vnr
