// Row-panel chain kernel (gfx950, split-fp16 MFMA): a workgroup owns 32 activation rows and runs a short PROGRAM of
// dense stages on them without leaving the CU.  Replaces chains of the reference's per-row ops inside
// CrossAttentionBLK (modules/attention.py:440-452) and FFN (modules/utils.py:48-53):
//     LN(att_proj(concat(x, ctx)) + x)  ->  query projection                                   (2 stages)
//     LN(att_proj(concat(y, ctx)) + y)  ->  FFN dense1+relu -> dense2 + residual -> LN  ->  next Q|K|V / flow heads
// Only weights are streamed (each wave owns 32 output columns and prefetches exactly its own weight operands into
// registers, 4 k-tiles deep, from an operand-major image: no LDS, no workgroup barrier inside a stage); activations stay
// in two LDS panels [32][<=256] kept in the
// split form the f16 matrix pipe consumes (per 32-k tile: 32 x fp16 hi | 32 x fp16 lo), so every product is the
// 3-term hi/lo split with fp32 accumulation (fp32-class accuracy, see gemm2.hip).  FFN hidden activations never
// touch HBM: the hidden layer is produced and consumed in chunks of <= 256 columns.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace vnr {

namespace {
constexpr unsigned kOob3 = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds3_t;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// LDS map of a workgroup with RT row tiles (ROWS = 32 RT activation rows):
//   [0, 2048 RT)                LayerNorm exchange scratch [2][ROWS][8] fp32
//   [.., + 2 or 3 x ROWS KiB)   the activation panels: ROWS rows x 8 k-tiles x 128 B (three for 32-row workgroups)
//   then                        bias of every stage [nstages][256] fp32, then (gamma | beta) [256 + 256] fp32 of every LayerNorm stage
template <int RT> struct ChainLds {
  static constexpr int ROWS = 32 * RT;
  static constexpr int PANEL_BYTES = ROWS * 1024;
  static constexpr int NPANELS = RT == 1 ? 3 : 2;        // (the third panel: hidden FFN chunks alternate between panels 1 and 2)
  static constexpr int P_OFF = 2048 * RT;
  static constexpr int PRM_OFF = P_OFF + NPANELS * PANEL_BYTES;
};

__device__ __forceinline__ float act3(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_TANH) return fast_tanhf(v);
  return v;
}
// workgroup barrier that orders LDS traffic only: global stores stay in flight (a __syncthreads() would drain vmcnt
// and expose the full store latency at every stage boundary)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// byte offset of 16-byte chunk `c` (0..7) of k-tile `kt` in row `r` of a panel (chunk index XOR-swizzled by row)
__device__ __forceinline__ int panel_off(int r, int kt, int c) { return r * 1024 + ((((kt << 3) + c) ^ (r & 15)) << 4); }
}  // namespace

// RT = 1: 32-row panels (M/32 workgroups).  RT = 2: 64-row panels (M/64 workgroups): every weight operand fetched from L2 feeds
// two MFMA sets, which halves the weight stream per row -- the stream through the CU's vector-memory return path (64 B/clk) is
// what bounds this kernel -- at the price of half as many workgroups: the same launch time when one batch owns the GPU, but
// half the CUs stay free for the kernels of other batches in flight (bench.py --streams).
template <int RT>
__global__ void __launch_bounds__(512)
panel_chain_kernel(const ChainArgs g) {
  using L = ChainLds<RT>;
  constexpr int ROWS = L::ROWS, PANEL_BYTES = L::PANEL_BYTES, P_OFF = L::P_OFF, PRM_OFF = L::PRM_OFF;
  // weight k-tiles kept in flight per wave (register prefetch depth): a k-tile of the 64-row variant carries twice the MFMAs, so
  // half the depth covers the same time (and the accumulators need the registers)
  constexpr int PF = RT == 2 ? 2 : 4;
  static_assert(PF % 2 == 0, "the activation operand double buffer alternates with the slot index: the depth must be even");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // (the L2-warming prefetch workgroups of chain_prefetch.h serve the 4-wave kernel only: measured +-0 for this one, and its stage loop is
  //  better off without the per-stage progress word -- launch_panel_chain gives this kernel no prefetch blocks)
  const int m0 = blockIdx.x * ROWS;
  auto panel_ptr = [&](int i) -> char* { return smem + P_OFF + i * PANEL_BYTES; };
  float* scratch = reinterpret_cast<float*>(smem);
  float* prm = reinterpret_cast<float*>(smem + PRM_OFF);       // [stage][256] bias, then the LayerNorm (gamma | beta) slots
  unsigned long long* ts = g.dbg_ts ? g.dbg_ts + (size_t)blockIdx.x * 128 : nullptr;
  auto stamp = [&](int i) { if (ts && tid == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  auto wstamp = [&](int si, int i) { if (ts && si == g.dbg_stage && lane == 0) ts[64 + wave * 8 + i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  // the second-dispatched half of the workgroup (waves 4..7) loses the per-SIMD issue arbitration against its older partner
  // on every phase (it leaves each k-loop ~1 kcyc later: profiles/r02_chain_rows32_timeline.txt); one static priority bump
  // for that half evens the pair out (MI355X_MICROARCH.md, "static priority for the younger half")
  if (g.prio_mode == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  // ---- the weight stream --------------------------------------------------------------------------------------------
  // Every stage's weights sit in an operand-major image: block (column block, k-tile) = 4 x 1 KiB pieces, piece
  // i = 2*t + part (t = k16 step, part 0 = hi / 1 = lo); lane l reads bytes [16 l, 16 l + 16) of a piece, i.e. exactly
  // its MFMA operand slot, so every wave instruction is one fully coalesced 1 KiB read straight into registers.
  // The k-tiles of ALL stages form one flat sequence per wave; PF tiles are always in flight and slot u is refilled
  // right after the MFMAs that consumed it -- across stage boundaries too.  Every refill is issued unconditionally
  // (tiles a wave does not need, the padding of a stage to a multiple of PF, and the tail after the last stage use an
  // out-of-range buffer offset: no memory traffic, zeros returned), so the number of loads between a refill and its use
  // is a compile-time constant and the compiler's s_waitcnt is vmcnt(4*(PF-1)), never a full drain.
  h16x8 wreg[PF][4];
  int fs = 0, fk = 0, fnk = 0, fpad = 0;
  unsigned fvoff = kOob3;
  __amdgpu_buffer_rsrc_t frs;
  auto open_stage = [&](int s_) {
    const ChainStage& st = g.st[s_];
    frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(st.w), 0, 0x40000000, 0x00020000);
    fnk = st.nk;
    fpad = (st.nk + PF - 1) / PF * PF;
    fvoff = (32 * wave < st.n) ? (unsigned)((wave * st.kt_total + st.kt0) * 4096 + lane * 16) : kOob3;
  };
  auto fetch = [&](int u) {
    const unsigned vo = (fk < fnk) ? fvoff : kOob3;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wreg[u][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, vo, fk * 4096 + i * 1024, 0));
    if (++fk == fpad) {
      fk = 0;
      if (fs + 1 < g.nstages) { ++fs; open_stage(fs); } else { fnk = 0; }      // past the end: dummy (out-of-range) refills
    }
  };
  open_stage(0);
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(u);

  // ---- epilogue parameters of the whole program -> LDS by LDS-DMA (no registers; read back with ds_read at the epilogues, so
  //      no vector-memory wait is ever needed there): stage s is copied by wave s mod 8 -- its bias (1 KiB) and, for a
  //      LayerNorm stage, gamma | beta (2 KiB, contiguous in the packed block) into the slot the host assigned (lds_ln)
  if (g.prm) {
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.prm), 0, (unsigned)g.nstages * 3072u, 0x00020000);
    for (int s_ = wave; s_ < g.nstages; s_ += 8) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + PRM_OFF + s_ * 1024), 16, (unsigned)(s_ * 3072 + lane * 16), 0, 0, 0);
      const int lo = g.st[s_].lds_ln;
      if (lo >= 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + PRM_OFF + lo * 4), 16, (unsigned)(s_ * 3072 + 1024 + lane * 16), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + PRM_OFF + lo * 4 + 1024), 16, (unsigned)(s_ * 3072 + 2048 + lane * 16), 0, 0, 0);
      }
    }
  } else {
    // no packed block (training step): every stage's parameters through their own pointers; a descriptor of exactly n floats zero-fills
    // the slot's padding, one of zero records gives the all-zero bias of a stage without one
    for (int s_ = wave; s_ < g.nstages; s_ += 8) {
      const float* bp = g.st[s_].bias; const float* gp = g.st[s_].gamma; const float* ep = g.st[s_].beta;
      const unsigned nb = (unsigned)g.st[s_].n * 4u;
      const bool skip = g.st[s_].acc_mode == 1 || g.st[s_].acc_mode == 2;            // (no epilogue of their own: the slot must still be zero)
      const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp ? bp : g.in0), 0, (bp && !skip) ? nb : 0u, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds3_t)(smem + PRM_OFF + s_ * 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
      const int lo = g.st[s_].lds_ln;
      if (lo >= 0) {
        const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp), 0, nb, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep ? ep : gp), 0, ep ? nb : 0u, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds3_t)(smem + PRM_OFF + lo * 4), 16, (unsigned)(lane * 16), 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsE, (lds3_t)(smem + PRM_OFF + lo * 4 + 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
      }
    }
  }
  // ---- input panels (fp32 rows in HBM -> split fp16 panel): all reads issued first, rows beyond M read as zeros --------
  {
    const int q4 = g.D >> 2;                                       // float4 per row (<= 64)
    float4 x[2][4 * RT];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const float* src = pi == 0 ? g.in0 : g.in1;
      const int ld = pi == 0 ? g.ld0 : g.ld1;
#pragma unroll
      for (int it = 0; it < 4 * RT; ++it) {
        const int e = tid + 512 * it, r = e / q4, j = e - r * q4;
        x[pi][it] = (src && r < ROWS && m0 + r < g.M) ? *reinterpret_cast<const float4*>(src + (size_t)(m0 + r) * ld + 4 * j)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      if (!(pi == 0 ? g.in0 : g.in1)) continue;
#pragma unroll
      for (int it = 0; it < 4 * RT; ++it) {
        const int e = tid + 512 * it, r = e / q4, j = e - r * q4;
        if (r >= ROWS) continue;
        const int kt = j >> 3, p = (j & 7) * 4;                    // position inside the 32-k tile
        h16x4 hi, lo;
        const float xv[4] = {x[pi][it].x, x[pi][it].y, x[pi][it].z, x[pi][it].w};
        { const vnr_f4 xs_ = {xv[0], xv[1], xv[2], xv[3]}; vnr_split(xs_, hi, lo); }
        *reinterpret_cast<h16x4*>(panel_ptr(pi) + panel_off(r, kt, p >> 3) + (p & 4) * 2) = hi;
        *reinterpret_cast<h16x4*>(panel_ptr(pi) + panel_off(r, kt, 4 + (p >> 3)) + (p & 4) * 2) = lo;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // parameter DMA landed (the head weight tiles too: issued first; issuing them
                                                                // behind the panel rows and leaving them in flight measured the same, round 5)
  lds_barrier();
  stamp(1);

  f32x16 accF[RT];                                      // persistent accumulators of the FFN second layer
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accF[rt][r] = 0.f;

#pragma unroll 1
  for (int si = 0; si < g.nstages; ++si) {
    // Every lane-dependent address of a stage is re-derived from an opaque copy of the lane id: otherwise the compiler hoists dozens of
    // loop-invariant offsets out of the stage loop, runs out of the 256 VGPRs and SPILLS them -- and a scratch reload in an epilogue
    // is a vector-memory load like any other: its s_waitcnt drains the whole weight prefetch stream (measured: +15 % on every chain
    // launch when one more epilogue variant pushed the kernel from 4 to 33 spilled registers).
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    const int half = lane_s >> 5, l31 = lane_s & 31;
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));
    if (RT == 1 && g.att_stage > 0 && si == g.att_stage) {
      stamp(60);                                                       // attention phase begins (the previous stage's barrier is behind)
      // ================= fused cross-attention of this panel (see ChainArgs::att_stage) ===================================
      // wave w: head = w >> 1, key blocks 2 (w & 1) and 2 (w & 1) + 1 (32 keys each, Tk <= 128).  Per block, as attn3_kernel:
      // S^T = K.Q^T (lane = query row: softmax statistics in-lane + one half swap), logits in the log2 domain, online softmax
      // over the wave's two blocks, O^T += V^T.P^T.  The two waves of a head are merged through LDS; the context replaces the
      // queries in panel 1 (split fp16, like any stage output).  A flat 32-row panel can straddle two batch elements (T is not
      // a multiple of 32): then the whole computation runs once per element and every lane keeps the pass of its own row.
      typedef _Float16 h8a __attribute__((ext_vector_type(8)));
      const int head = wave >> 1, kp = wave & 1;
      const int H = g.D >> 6, TTk = (g.att_Tk + 31) >> 5;
      const int row = m0 + l31;                                       // this lane's query row (global)
      int b_lo = m0 / g.att_Tq, b_hi = (m0 + 31 < g.M ? m0 + 31 : g.M - 1) / g.att_Tq;
      b_lo = __builtin_amdgcn_readfirstlane(b_lo); b_hi = __builtin_amdgcn_readfirstlane(b_hi);
      char* P1 = panel_ptr(1);
      char* Pc = panel_ptr(g.att_ali ? 2 : 1);                        // the context: in place of the queries, or -- when the alignments are
                                                                      // wanted, whose pass below multiplies K and Q once more -- in panel 2
      float* xs = reinterpret_cast<float*>(smem + g.att_lds);        // merge scratch: [4 heads][32 rows][64 + 2] floats
      const float c2 = (g.att_temp != 1.0f) ? 0.125f * 1.44269504088896340736f / g.att_temp : 0.125f * 1.44269504088896340736f;
      for (int bb = b_lo; bb <= b_hi; ++bb) {
        const int qlen = g.att_qlen ? g.att_qlen[bb] : g.att_Tq, klen = g.att_klen ? g.att_klen[bb] : g.att_Tk;
        const int tq = row - bb * g.att_Tq;                           // position of this lane's row inside element bb
        const bool mine = row < g.M && tq >= 0 && tq < g.att_Tq;
        const bool qvalid = mine && tq < qlen;
        f32x16 O[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
        for (int x = 0; x < 2; ++x) {
          const int kb = 2 * kp + x;
          if (32 * kb >= g.att_Tk) break;                             // (wave-uniform)
          const char* kt = g.att_K + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
          const char* vt = g.att_V + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
          h8a khi[4], klo[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) { khi[t] = *reinterpret_cast<const h8a*>(kt + 1024 * t); klo[t] = *reinterpret_cast<const h8a*>(kt + 4096 + 1024 * t); }
          f32x16 sacc;
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const h8a qh = *reinterpret_cast<const h8a*>(P1 + panel_off(l31, 2 * head + (t >> 1), 2 * (t & 1) + half));
            const h8a ql = *reinterpret_cast<const h8a*>(P1 + panel_off(l31, 2 * head + (t >> 1), 4 + 2 * (t & 1) + half));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi[t], qh, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo[t], qh, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi[t], ql, sacc, 0, 0, 0);
          }
          h8a vhi[2][2], vlo[2][2];
#pragma unroll
          for (int y = 0; y < 4; ++y) { vhi[y >> 1][y & 1] = *reinterpret_cast<const h8a*>(vt + 1024 * y); vlo[y >> 1][y & 1] = *reinterpret_cast<const h8a*>(vt + 4096 + 1024 * y); }
          const int kb0 = 32 * kb;
          float mt = -INFINITY;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = kb0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float sv = sacc[r] * c2;
            sv = (qvalid && j < klen) ? sv : kMaskFill * 1.44269504088896340736f;     // attention.py:240
            if (j >= g.att_Tk) sv = -INFINITY;                                             // key does not exist
            sacc[r] = sv;
            mt = fmaxf(mt, sv);
          }
          mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
          const float m_new = fmaxf(fmaxf(m_run, mt), -3.0e38f);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                      // 0 on the first block
          float ps = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float pr = __builtin_amdgcn_exp2f(sacc[r] - m_new); sacc[r] = pr; ps += pr; }
          ps += __shfl_xor(ps, 32, 64);
          l_run = l_run * alpha + ps;
          if (x > 0) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int r = 0; r < 16; ++r) O[nb][r] *= alpha;
          }
          m_run = m_new;
#pragma unroll
          for (int tp = 0; tp < 2; ++tp) {
            if (kb0 + 32 > g.att_Tk) {                                 // positions past Tk hold whatever the workspace held
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int key = kb0 + 16 * tp + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (key >= g.att_Tk) { vhi[tp][0][e] = vhi[tp][1][e] = vlo[tp][0][e] = vlo[tp][1][e] = (_Float16)0.f; }
              }
            }
            h8a phi, plo;
            { const vnr_f8 xs_ = {sacc[8 * tp + 0], sacc[8 * tp + 1], sacc[8 * tp + 2], sacc[8 * tp + 3], sacc[8 * tp + 4], sacc[8 * tp + 5], sacc[8 * tp + 6], sacc[8 * tp + 7]}; vnr_split(xs_, phi, plo); }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              O[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][nb], phi, O[nb], 0, 0, 0);
              O[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo[tp][nb], phi, O[nb], 0, 0, 0);
              O[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][nb], plo, O[nb], 0, 0, 0);
            }
          }
        }
        // merge the two waves of this head: kp = 1 parks (m, l, O) in LDS, kp = 0 combines.  O^T layout: lane = query l31,
        // register r of block nb = channel 32 nb + frow(r, half)
        float* xh = xs + head * (32 * 66);
        const float mfin = fmaxf(m_run, -3.0e38f);
        if (kp == 1) {
          if (half == 0) { xh[l31 * 66 + 64] = mfin; xh[l31 * 66 + 65] = l_run; }
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) xh[l31 * 66 + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * half] = O[nb][r];
        }
        lds_barrier();
        if (kp == 0) {
          const float m1 = xh[l31 * 66 + 64], l1 = xh[l31 * 66 + 65];
          const float M = fmaxf(mfin, m1);
          const float f0 = __builtin_amdgcn_exp2f(mfin - M), f1 = __builtin_amdgcn_exp2f(m1 - M);
          const float linv = 1.0f / (l_run * f0 + l1 * f1);                                // softmax denominator, attention.py:242
          if (g.att_ali && half == 0) { xh[l31 * 66 + 64] = M; xh[l31 * 66 + 65] = linv; }   // for the alignment pass of both waves (read behind the barrier)
          // the context replaces the queries of this lane's row in panel 1 (tiles 2 head, 2 head + 1: read by this head's two
          // waves only, and both are past their S^T products of this pass: the barrier above).  Rows of the other batch element of a
          // straddling panel keep their queries for the next pass.
          if (mine) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                // (vector-typed conversions: ONE node for the high halves -- behind a product like this one the compiler may otherwise fuse the
                //  multiply into the conversion on one side of the split only, gemm3c.hip: split_hi_lo)
                typedef float f32x4c __attribute__((ext_vector_type(4)));
                f32x4c xv4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int r = 4 * q + e;
                  xv4[e] = (O[nb][r] * f0 + xh[l31 * 66 + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * half] * f1) * linv;
                }
                const h16x4 hi = __builtin_convertvector(xv4, h16x4);
                const h16x4 lo = __builtin_convertvector(xv4 - __builtin_convertvector(hi, f32x4c), h16x4);
                const int pcol = 8 * q + 4 * half;                     // column inside the 32-channel tile 2 head + nb
                *reinterpret_cast<h16x4*>(Pc + panel_off(l31, 2 * head + nb, pcol >> 3) + (pcol & 4) * 2) = hi;
                *reinterpret_cast<h16x4*>(Pc + panel_off(l31, 2 * head + nb, 4 + (pcol >> 3)) + (pcol & 4) * 2) = lo;
              }
          }
        }
        lds_barrier();                                                 // scratch free for the next pass; the context of this pass is in place
        if (g.att_ali) {                                               // (workgroup-uniform)
          // alignments = softmax(logits) of this head's rows (attention.py:242-246): every wave normalises its own two key blocks with
          // the head's final (max, 1 / sum) and writes them as 128-byte row pieces after a wave-private 32 x 32 transpose through the
          // merge scratch (free now) -- lane-per-query registers would leave 32-byte pieces, which the memory system takes at a
          // tenth of the rate (profiles/r02_store_bw_probe.txt)
          const float M_fin = xh[l31 * 66 + 64], linv_fin = xh[l31 * 66 + 65];      // left there by the kp = 0 wave (below the merge)
          lds_barrier();                                               // both waves of the head hold them: the region may be overwritten
          float* tb = xs + wave * (32 * 33);
#pragma unroll 1
          for (int x = 0; x < 2; ++x) {
            const int kb = 2 * kp + x;
            if (32 * kb >= g.att_Tk) break;                            // (wave-uniform)
            // S^T of the block once more (K tile from L1 / L2, Q still in panel 1): holding both blocks' probabilities through P.V and
            // the merge cost 32 registers the kernel does not have (20 spilled)
            const char* kt = g.att_K + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const h8a kh = *reinterpret_cast<const h8a*>(kt + 1024 * t), kl = *reinterpret_cast<const h8a*>(kt + 4096 + 1024 * t);
              const h8a qh = *reinterpret_cast<const h8a*>(P1 + panel_off(l31, 2 * head + (t >> 1), 2 * (t & 1) + half));
              const h8a ql = *reinterpret_cast<const h8a*>(P1 + panel_off(l31, 2 * head + (t >> 1), 4 + 2 * (t & 1) + half));
              sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh, sacc, 0, 0, 0);
              sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh, sacc, 0, 0, 0);
              sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql, sacc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int j = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
              float sv = sacc[r] * c2;
              sv = (qvalid && j < klen) ? sv : kMaskFill * 1.44269504088896340736f;         // attention.py:240
              if (j >= g.att_Tk) sv = -INFINITY;
              tb[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half] = __builtin_amdgcn_exp2f(sv - M_fin) * linv_fin;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int rr = 8 * it + (lane_s >> 3), k4 = (lane_s & 7) * 4;
              const float4 pv = make_float4(tb[rr * 33 + k4], tb[rr * 33 + k4 + 1], tb[rr * 33 + k4 + 2], tb[rr * 33 + k4 + 3]);
              const int rg = m0 + rr, tqr = rg - bb * g.att_Tq, key0 = 32 * kb + k4;
              if (rg < g.M && tqr >= 0 && tqr < g.att_Tq && key0 < g.att_Tk)
                *reinterpret_cast<float4*>(g.att_ali + (((size_t)(bb * H + head) * g.att_Tq + tqr) * g.att_Tk + key0)) = pv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          lds_barrier();                                               // the scratch is the next pass's merge area again
        }
      }
      stamp(61);                                                       // attention (and alignment) phase ends
    }
    const ChainStage st = g.st[si];                      // by value: one scalar burst from the kernarg segment per stage
    const bool wave_on = 32 * wave < st.n;               // this wave owns output columns 32w .. 32w+31
    const char* const Ap0 = panel_ptr(st.a0);
    const char* const Ap1 = panel_ptr(st.a1);
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    // A stage that holds only V columns of a Q|K|V panel (out_fmt 4) is computed UN-transposed (operands swapped: lane <->
    // output column, registers <-> the 32 rows in k-slot order): 8 consecutive registers are then exactly one 16-byte unit of
    // the V image (common.h), stored with fully coalesced 1 KiB wave writes instead of 2-byte scatters.  Needs the 16-row
    // half panels to coincide with half tiles: rows per batch element % 16 == 0.
    const bool vswap = st.out_fmt == 4 && st.out && st.aoi_c0 >= 2 * st.aoi_D && !((st.aoi_c0 - 2 * st.aoi_D) & 31) && !(st.aoi_T & 15) &&
                       st.acc_mode == 0 && st.dst < 0 && st.res < 0 && !st.gamma && !st.pe && st.act == ACT_IDENTITY;
    wstamp(si, 0);                                       // stage descriptor in registers
    const int npad = (st.nk + PF - 1) / PF * PF;
    // activation operands are read ONE k-tile ahead (registers a[cur] / a[nxt]) so the LDS latency of tile kt+1 hides
    // under the MFMAs of tile kt; reads past the stage's last tile are clamped (harmless re-read)
    h16x8 afr[2][RT][4];
    auto read_a = [&](int kt, int set) {
      const int kc = kt < st.nk ? kt : st.nk - 1;
      const char* Ap = (kc < st.asw) ? Ap0 : Ap1;
      const int akt = (kc < st.asw) ? kc + st.akt0 : kc - st.asw;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          afr[set][rt][2 * t] = *reinterpret_cast<const h16x8*>(Ap + panel_off(32 * rt + l31, akt, 2 * t + half));
          afr[set][rt][2 * t + 1] = *reinterpret_cast<const h16x8*>(Ap + panel_off(32 * rt + l31, akt, 4 + 2 * t + half));
        }
    };
    // The k-loop exists three times, selected ONCE per stage by wave-uniform conditions, so that its body is straight-line
    // code: with the selection inside the loop (round 1) the accumulator became a phi of three paths and the compiler copied it
    // (s_nop 11 + sixteen v_mov_b64) between the MFMA groups of consecutive k-tiles.  (Measured: the copies were NOT what
    // bounds the loop -- 1.45 ms per step before and after; neither is the dependent accumulator chain: two alternating
    // accumulators per wave changed nothing either.)  Padding k-tiles (kt >= nk) are not skipped either: their weight operands arrive
    // as zeros from the out-of-range refill, the MFMAs add exact zeros.
    auto kloop = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;      // 0: D^T (lane <-> activation row), 1: D (V image stage), 2: idle wave
      if (MODE != 2) read_a(0, 0);
#pragma unroll 1
      for (int kb = 0; kb < npad; kb += PF) {
        if (g.prio_mode >= 2) {            // experiment: the two waves of a SIMD take turns at the issue arbiter
          const int turn = g.prio_mode == 2 ? (kb / PF) : si;
          if ((turn ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (MODE != 2) {
            read_a(kb + u + 1, (u + 1) & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int rt = 0; rt < RT; ++rt) {
                if (MODE == 0) {
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t], afr[u & 1][rt][2 * t], acc[rt], 0, 0, 0);
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t], afr[u & 1][rt][2 * t + 1], acc[rt], 0, 0, 0);
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t + 1], afr[u & 1][rt][2 * t], acc[rt], 0, 0, 0);
                } else {
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[u & 1][rt][2 * t], wreg[u][2 * t], acc[rt], 0, 0, 0);
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[u & 1][rt][2 * t + 1], wreg[u][2 * t], acc[rt], 0, 0, 0);
                  acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[u & 1][rt][2 * t], wreg[u][2 * t + 1], acc[rt], 0, 0, 0);
                }
              }
          }
          fetch(u);                                                         // refill this slot PF tiles ahead (flat sequence)
        }
      }
    };
    if (!wave_on) kloop(std::integral_constant<int, 2>{});
    else if (!vswap) kloop(std::integral_constant<int, 0>{});
    else kloop(std::integral_constant<int, 1>{});
    stamp(2 + 2 * si);
    wstamp(si, 1);
    if (g.range_flag && wave_on) {                       // overflow sentinel (common.h: range_note): a row with a split operand out of range is all NaN
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float probe = acc[rt][0];                        // D^T layout: this lane's row
        if (vswap) {                                     // un-transposed stage: the 16 registers are 16 rows
#pragma unroll
          for (int r = 1; r < 16; ++r) probe += acc[rt][r];
        }
        range_note(g.range_flag, probe);
      }
    }
    // ---- FFN second layer: accumulate over hidden chunks (modes 1,2: no epilogue yet) ----------------------------------
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      if (st.acc_mode == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accF[rt][r] = acc[rt][r];
      } else if (st.acc_mode >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accF[rt][r] += acc[rt][r];
      }
    }
    if (st.acc_mode == 1 || st.acc_mode == 2) { if (st.sync_after) lds_barrier(); stamp(3 + 2 * si); continue; }   // (a barrier only if the hidden panel is rewritten next)
    // ---- fast path: a full-width hidden stage h = relu(x.W + b) -> other panel (FFN dense1 chunks, utils.py:49): no residual,
    //      no LayerNorm, no HBM output, no column masks -- a third of the stages of a block chain
    if (st.acc_mode == 0 && st.act == ACT_RELU && st.n == 256 && !st.pe && st.res < 0 && !st.gamma && !st.out && st.dst >= 0 &&
        st.dst != st.a0 && !(st.asw < st.nk && st.dst == st.a1)) {
      const float* sp = prm + si * 256;
      char* Dp = panel_ptr(st.dst);
      const int kt = wave;                                              // this wave's 32 columns = k-tile `wave` of the destination
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int prow = 32 * rt + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bi = *reinterpret_cast<const float4*>(sp + 32 * wave + 8 * q + 4 * half);
          const float x[4] = {fmaxf(acc[rt][4 * q] * st.scale + bi.x, 0.f), fmaxf(acc[rt][4 * q + 1] * st.scale + bi.y, 0.f),
                              fmaxf(acc[rt][4 * q + 2] * st.scale + bi.z, 0.f), fmaxf(acc[rt][4 * q + 3] * st.scale + bi.w, 0.f)};
          h16x4 hi, lo;
          { const vnr_f4 xs_ = {x[0], x[1], x[2], x[3]}; vnr_split(xs_, hi, lo); }
          const int p = 8 * q + 4 * half;
          *reinterpret_cast<h16x4*>(Dp + panel_off(prow, kt, p >> 3) + (p & 4) * 2) = hi;
          *reinterpret_cast<h16x4*>(Dp + panel_off(prow, kt, 4 + (p >> 3)) + (p & 4) * 2) = lo;
        }
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    if (vswap) {
      if (wave_on) {
        const int cv = st.aoi_c0 - 2 * st.aoi_D + 32 * wave + l31;          // V column of this lane: head cv >> 6, channel cv & 63
        const float bv = (prm + si * 256)[32 * wave + l31];                // (zero padded when the stage has no bias)
        const int Hh = st.aoi_D >> 6, TT = (st.aoi_T + 31) >> 5;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int tp = 0; tp < 2; ++tp) {
            const int R = m0 + 32 * rt + 16 * tp;
            if (R >= g.M) continue;
            const int bb = R / st.aoi_T, tt = R - bb * st.aoi_T;
            h16x8 hi, lo;
            {
              vnr_f8 xs_;
#pragma unroll
              for (int e = 0; e < 8; ++e) xs_[e] = acc[rt][8 * tp + e] * st.scale + bv;
              vnr_split(xs_, hi, lo);
            }
            char* pdst = reinterpret_cast<char*>(st.out) + 2 * st.aoi_img_bytes + ((size_t)(bb * Hh + (cv >> 6)) * TT + (tt >> 5)) * kAoiTile +
                         ((tt >> 4) & 1) * 2048 + ((cv >> 5) & 1) * 1024 + ((half * 32 + l31) << 4);
            *reinterpret_cast<h16x8*>(pdst) = hi;
            *reinterpret_cast<h16x8*>(pdst + 4096) = lo;
          }
      }
      if (st.sync_after) lds_barrier();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    if (st.acc_mode == 3) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] = accF[rt][r];
    }

    // ---- epilogue: v = act(acc*scale + bias) (+ residual panel) ; optional LayerNorm over the row ------------------------
    // lane (row 32 rt + l31, half) holds columns n = 32*wave + 8q + 4*half + e  (register 4q + e)
    float v[RT][16];
    const float* sp = prm + si * 256;
    bool cok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * wave + 8 * q + 4 * half;
      cok[q] = wave_on && col < st.n;                                   // (n is a multiple of 4)
      const float4 bi = *reinterpret_cast<const float4*>(sp + col);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        v[rt][4 * q + 0] = acc[rt][4 * q + 0] * st.scale + bi.x;
        v[rt][4 * q + 1] = acc[rt][4 * q + 1] * st.scale + bi.y;
        v[rt][4 * q + 2] = acc[rt][4 * q + 2] * st.scale + bi.z;
        v[rt][4 * q + 3] = acc[rt][4 * q + 3] * st.scale + bi.w;
      }
    }
    if (st.act == ACT_RELU) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[rt][r] = fmaxf(v[rt][r], 0.f);
    } else if (st.act == ACT_TANH) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[rt][r] = fast_tanhf(v[rt][r]);
    }
    wstamp(si, 4);                                       // accumulators drained, bias / activation applied
    if (g.cpl_stage > 0 && si == g.cpl_stage) {
      // ================= fused affine coupling (see ChainArgs::cpl_stage; the arithmetic of misc.hip: coupling_fwd_kernel) ============
      // v = [log_scale (hc columns) | shift (hc columns)].  The active waves park v in LDS ([ROWS][2 hc + 4] floats) and are done with their
      // accumulators; after the barrier ALL threads share the elementwise work, one 4-column piece of zp and one of the conditioning
      // half per thread and trip: zp' = sigmoid(log_scale + 2) * zp + shift goes back to z and, like the conditioning half, into the
      // destination panel in split form (what any stage output looks like).
      const int cst = st.n + 4;
      float* cx = reinterpret_cast<float*>(smem + g.cpl_lds);
      if (wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(cx + (32 * rt + l31) * cst + 32 * wave + 8 * q + 4 * half) =
                make_float4(v[rt][4 * q], v[rt][4 * q + 1], v[rt][4 * q + 2], v[rt][4 * q + 3]);
      }
      lds_barrier();
      {
        // hc = 64 (checked by the launcher): thread -> (row tid >> 4, 4-column piece tid & 15), first the transformed half, then the
        // conditioning half
        char* Dp = panel_ptr(st.dst);
        const int c = (tid_s & 15) * 4;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int prow = 32 * rt + (tid_s >> 4), row = m0 + prow;
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            const int zoff = which ? g.cpl_cond_off : g.cpl_zp_off;
            float* zp = g.cpl_z + (size_t)row * g.cpl_ld + zoff + c;
            const float4 zo = row < g.M ? *reinterpret_cast<const float4*>(zp) : make_float4(0.f, 0.f, 0.f, 0.f);
            float o[4] = {zo.x, zo.y, zo.z, zo.w};
            if (which == 0) {
              const float4 ls = *reinterpret_cast<const float4*>(cx + prow * cst + c), sh = *reinterpret_cast<const float4*>(cx + prow * cst + 64 + c);
              const float lv[4] = {ls.x, ls.y, ls.z, ls.w}, sv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float scale = 1.0f / (1.0f + expf(-(lv[e] + 2.0f)));                    // tf.math.sigmoid(log_scale + 2), flow.py:231
                o[e] = scale * o[e] + sv[e];                                                  // _affine, flow.py:216
              }
              if (row < g.M) out_store4(zp, o[0], o[1], o[2], o[3]);
            }
            h16x4 hi, lo;
            { const vnr_f4 xs_ = {o[0], o[1], o[2], o[3]}; vnr_split(xs_, hi, lo); }
            const int kt = (zoff + c) >> 5, p = (zoff + c) & 31;
            *reinterpret_cast<h16x4*>(Dp + panel_off(prow, kt, p >> 3) + (p & 4) * 2) = hi;
            *reinterpret_cast<h16x4*>(Dp + panel_off(prow, kt, 4 + (p >> 3)) + (p & 4) * 2) = lo;
          }
        }
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = m0 + 32 * rt + l31, prow = 32 * rt + l31;
      if (st.pe) {                                                      // + pos_weight * PE[t] (encoder.py:85, transform.py:51)
        const float* pr = st.pe + (size_t)(row % st.pe_T) * st.n;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * wave + 8 * q + 4 * half;
          if (cok[q] && row < g.M) {
            const float4 p4 = *reinterpret_cast<const float4*>(pr + col);
            v[rt][4 * q] += st.pe_w * p4.x; v[rt][4 * q + 1] += st.pe_w * p4.y; v[rt][4 * q + 2] += st.pe_w * p4.z; v[rt][4 * q + 3] += st.pe_w * p4.w;
          }
        }
      }
      if (st.res >= 0) {                                                // residual = hi + lo of the panel entry (22 bits)
        const char* Rp = panel_ptr(st.res);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * wave + 8 * q + 4 * half, kt = col >> 5, p = col & 31;
          const h16x4 rh = *reinterpret_cast<const h16x4*>(Rp + panel_off(prow, kt, p >> 3) + (p & 4) * 2);
          const h16x4 rl = *reinterpret_cast<const h16x4*>(Rp + panel_off(prow, kt, 4 + (p >> 3)) + (p & 4) * 2);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[rt][4 * q + e] += (float)rh[e] + (float)rl[e];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (!cok[q]) v[rt][4 * q] = v[rt][4 * q + 1] = v[rt][4 * q + 2] = v[rt][4 * q + 3] = 0.f;
    }
    if (st.gamma && st.out_pre) {                                       // training: x + Dense(.) before the normalisation
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = m0 + 32 * rt + l31;
        if (row < g.M) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (cok[q]) out_store4(st.out_pre + (size_t)row * st.ldo + 32 * wave + 8 * q + 4 * half, v[rt][4 * q], v[rt][4 * q + 1], v[rt][4 * q + 2], v[rt][4 * q + 3]);
        }
      }
    }
    if (st.gamma) {
      // LayerNormalization (eps 1e-3).  ONE exchange: every wave reduces its own <= 32 columns of a row to (sum, M2 about its own
      // mean) -- two in-lane passes and two half-swaps, no LDS -- and the eight partials are merged exactly (Chan et al.):
      // var.n = sum_w [M2_w + c_w (mean_w - mean)^2].  (Round 1 exchanged the mean first and the centred squares second: two
      // barriers and two LDS round trips per LayerNorm stage.)
      int cw = st.n - 32 * wave; cw = cw < 0 ? 0 : (cw > 32 ? 32 : cw);          // valid columns of this wave
      const float rcw = cw > 0 ? 1.f / (float)cw : 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += v[rt][r];
        s1 += __shfl_xor(s1, 32, 64);
        const float mw = s1 * rcw;
        float m2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = cok[q] ? v[rt][4 * q + e] - mw : 0.f;
            m2 += d * d;
          }
        m2 += __shfl_xor(m2, 32, 64);
        if (half == 0) { scratch[(32 * rt + l31) * 8 + wave] = s1; scratch[ROWS * 8 + (32 * rt + l31) * 8 + wave] = m2; }
      }
      lds_barrier();
      const float* lnp = prm + st.lds_ln;                               // gamma [256] | beta [256]
      const float rn = 1.f / (float)st.n;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 sa = *reinterpret_cast<const f32x4*>(scratch + (32 * rt + l31) * 8), sb = *reinterpret_cast<const f32x4*>(scratch + (32 * rt + l31) * 8 + 4);
        const f32x4 ma = *reinterpret_cast<const f32x4*>(scratch + ROWS * 8 + (32 * rt + l31) * 8), mb = *reinterpret_cast<const f32x4*>(scratch + ROWS * 8 + (32 * rt + l31) * 8 + 4);
        const float sw[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
        const float mq[8] = {ma[0], ma[1], ma[2], ma[3], mb[0], mb[1], mb[2], mb[3]};
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += sw[w];
        const float mu = tot * rn;
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          int c = st.n - 32 * w; c = c < 0 ? 0 : (c > 32 ? 32 : c);
          const float dm = c > 0 ? sw[w] / (float)c - mu : 0.f;
          var += mq[w] + (float)c * dm * dm;
        }
        const float rstd = 1.0f / sqrtf(var * rn + kLnEps);
        if (st.out_stats && wave == 0 && half == 0 && m0 + 32 * rt + l31 < g.M)
          *reinterpret_cast<float2*>(st.out_stats + 2 * (size_t)(m0 + 32 * rt + l31)) = make_float2(mu, rstd);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * wave + 8 * q + 4 * half;
          const float4 ga = *reinterpret_cast<const float4*>(lnp + col), be = *reinterpret_cast<const float4*>(lnp + 256 + col);
          v[rt][4 * q + 0] = (v[rt][4 * q + 0] - mu) * rstd * ga.x + be.x;
          v[rt][4 * q + 1] = (v[rt][4 * q + 1] - mu) * rstd * ga.y + be.y;
          v[rt][4 * q + 2] = (v[rt][4 * q + 2] - mu) * rstd * ga.z + be.z;
          v[rt][4 * q + 3] = (v[rt][4 * q + 3] - mu) * rstd * ga.w + be.w;
        }
      }
    } else if (st.dst >= 0 && (st.dst == st.a0 || (st.asw < st.nk && st.dst == st.a1))) {
      lds_barrier();                                                  // in-place stage: every wave is done reading the source panel
    }
    wstamp(si, 2);
    // ---- outputs: HBM (fp32, 16-byte row pieces) and/or destination panel (split fp16) -----------------------------------
    // Attention operand images (common.h): a wave's 32 columns lie inside one head, so everything but (q, half) is either
    // per-lane-per-stage (the row's tile) or wave-uniform (image, head, channel base).  Q/K-type columns take the fast
    // path below; V-type columns of a stage that could not run un-transposed fall back to the generic scatter.
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = m0 + 32 * rt + l31, prow = 32 * rt + l31;
      char* img_row = nullptr;                           // Q/K-type: address of this lane's 16-byte unit for t = 0, g = 0
      bool img_generic = false;
      if (st.out && st.out_fmt != 0 && wave_on && row < g.M) {
        const int Dd = st.out_fmt == 1 ? st.n : st.aoi_D, wc0 = (st.out_fmt == 1 ? 0 : st.aoi_c0) + 32 * wave;
        const int which = __builtin_amdgcn_readfirstlane(wc0 / Dd), cw = wc0 - which * Dd;
        if (which == 2) img_generic = true;
        else {
          const int TT = (st.aoi_T + 31) >> 5, bb = row / st.aoi_T, tt = row - bb * st.aoi_T;
          img_row = reinterpret_cast<char*>(st.out) + (size_t)which * st.aoi_img_bytes +
                    ((size_t)(bb * (Dd >> 6) + (cw >> 6)) * TT + (tt >> 5)) * kAoiTile + ((cw & 63) >> 4) * 1024 + ((tt & 31) << 4) + half * 8;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * wave + 8 * q + 4 * half;
        if (!cok[q]) continue;
        if (st.out && row < g.M) {
          if (img_row) {                                   // channel d = (cw & 63) + 8q + 4 half: t = d >> 4, g = q & 1
            h16x4 hi, lo;
            { const vnr_f4 xs_ = {v[rt][4 * q + 0], v[rt][4 * q + 1], v[rt][4 * q + 2], v[rt][4 * q + 3]}; vnr_split(xs_, hi, lo); }
            char* pd = img_row + (q >> 1) * 1024 + (q & 1) * 512;
            *reinterpret_cast<h16x4*>(pd) = hi;
            *reinterpret_cast<h16x4*>(pd + 4096) = lo;
          } else if (img_generic) {
            AoiDesc ad; ad.mode = 4; ad.D = st.aoi_D; ad.T = st.aoi_T; ad.TT = (st.aoi_T + 31) >> 5; ad.blk_bytes = st.aoi_img_bytes;
            ad.qk = reinterpret_cast<char*>(st.out); ad.vt = ad.qk + 2 * st.aoi_img_bytes;
            aoi_store4(ad, row, st.aoi_c0 + col, &v[rt][4 * q]);
          }
          else out_store4(st.out + (size_t)row * st.ldo + col, v[rt][4 * q], v[rt][4 * q + 1], v[rt][4 * q + 2], v[rt][4 * q + 3]);
        }
        if (st.dst >= 0) {
          h16x4 hi, lo;
          { const vnr_f4 xs_ = {v[rt][4 * q + 0], v[rt][4 * q + 1], v[rt][4 * q + 2], v[rt][4 * q + 3]}; vnr_split(xs_, hi, lo); }
          const int kt = col >> 5, p = col & 31;
          *reinterpret_cast<h16x4*>(panel_ptr(st.dst) + panel_off(prow, kt, p >> 3) + (p & 4) * 2) = hi;
          *reinterpret_cast<h16x4*>(panel_ptr(st.dst) + panel_off(prow, kt, 4 + (p >> 3)) + (p & 4) * 2) = lo;
        }
      }
    }
    wstamp(si, 5);                                                    // outputs issued
    // Stage boundaries are where this kernel loses its time: a barrier re-aligns the eight waves, and in phase they all block on the
    // address unit together and then multiply together (tools/probes/chain_round_probe.hip: 527 clk per k-tile round free-running,
    // 843 with a boundary every 8 rounds).  So the barrier is placed only where the launcher's hazard analysis needs one.
    if (st.sync_after) lds_barrier();                                 // panels are complete / free before the next stage
    stamp(3 + 2 * si);
    wstamp(si, 3);
  }
}

template <int RT>
static hipError_t launch_chain_rt(const ChainArgs& g, int lds, hipStream_t s) {
  static int attr_set[kMaxDevices] = {0};
  opt_in_dynamic_lds((const void*)panel_chain_kernel<RT>, lds, attr_set);
  constexpr int ROWS = 32 * RT;
  const int wgs = (g.M + ROWS - 1) / ROWS;
  static const char* ts_path = getenv("VNR_CHAIN_TS");
  if (ts_path) {
    ChainArgs gg = g;
    static const char* ts_stage = getenv("VNR_CHAIN_TS_STAGE");       // stage whose per-wave stamps are taken (default 1)
    gg.dbg_stage = ts_stage ? atoi(ts_stage) : 1;
    const size_t n = (size_t)wgs * 128;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d, 0, n * 8, s);
    gg.dbg_ts = d;
    vnr_launch(panel_chain_kernel<RT>, dim3(wgs), dim3(512), lds, s, gg);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hbuf(n);
    (void)hipMemcpy(hbuf.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FILE* f = fopen(ts_path, "ab");
    if (f) { int hdr[4] = {g.M, g.D | (g.att_stage > 0 ? (g.att_stage << 20) : 0) | (g.att_ali ? (1 << 16) : 0), g.nstages, (int)(n / 128)}; /* D <= 256: flags above bit 15 */ fwrite(hdr, 4, 4, f); fwrite(hbuf.data(), 8, n, f); fclose(f); }
    return hipGetLastError();
  }
  vnr_launch(panel_chain_kernel<RT>, dim3(wgs), dim3(512), lds, s, g);
  return hipGetLastError();
}

hipError_t launch_panel_chain(const ChainArgs& g_in, hipStream_t s) {
  ChainArgs g = g_in;
  static const int prio_mode = getenv("VNR_CHAIN_PRIO") ? atoi(getenv("VNR_CHAIN_PRIO")) : 1;
  g.prio_mode = prio_mode;
  if (g.M <= 0 || g.D <= 0 || (g.D & 31) || g.D > 256 || g.nstages <= 0 || g.nstages > kMaxChainStages) return hipErrorInvalidValue;
  int nln = 0;
  for (int i = 0; i < g.nstages; ++i) {
    ChainStage& st = g.st[i];
    if (st.n <= 0 || st.n > 256 || (st.n & 3) || st.nk <= 0 || st.nk > 16 || st.akt0 < 0 || st.akt0 + (st.asw < st.nk ? st.asw : st.nk) > 8 || (st.pe && st.pe_T <= 0) || st.asw <= 0 || (st.asw < st.nk && st.nk - st.asw > 8) || (st.asw < st.nk ? st.asw : st.nk) > 8 || !st.w) return hipErrorInvalidValue;
    if (st.out && st.out_fmt == 0 && (st.ldo & 3)) return hipErrorInvalidValue;
    if (st.out && st.out_fmt == 1 && ((st.n & 63) || st.aoi_T <= 0)) return hipErrorInvalidValue;
    if (st.out && st.out_fmt == 4 && ((st.aoi_D & 63) || st.aoi_T <= 0 || st.aoi_D <= 0 || (st.aoi_c0 & 3) || st.aoi_c0 + st.n > 3 * st.aoi_D)) return hipErrorInvalidValue;
    if (st.out_fmt != 0 && st.out_fmt != 1 && st.out_fmt != 4) return hipErrorInvalidValue;
    // LDS slot (float offset from the parameter area) of this stage's gamma | beta: behind the bias table
    st.lds_ln = (st.gamma && st.acc_mode != 1 && st.acc_mode != 2) ? g.nstages * 256 + 512 * nln++ : -1;
  }
  // where a stage boundary needs its workgroup barrier: (i) after a stage that wrote a panel (its columns come from all waves),
  // (ii) in front of a stage whose epilogue overwrites a panel that an earlier stage read since the last barrier (a slower wave
  // may still be in that stage's k-loop).  Stages that only read (FFN second layers accumulating in registers, the tails that go
  // to HBM) need none, and with the hidden chunks alternating between panels 1 and 2 neither does the step from one chunk to the next.
  {
    const int npanels = g.rows64 ? 2 : 3;
    bool rd[3] = {false, false, false};
    for (int i = 0; i < g.nstages; ++i) {
      ChainStage& st = g.st[i];
      if (st.a0 < 0 || st.a0 >= npanels || st.a1 < 0 || st.a1 >= npanels || st.dst >= npanels || st.res >= npanels) return hipErrorInvalidValue;
      if (g.att_stage > 0 && i == g.att_stage) {         // the fused attention: reads and rewrites panel 1 between its own barriers
        if (rd[1] && i > 0) g.st[i - 1].sync_after = 1;
        rd[0] = rd[1] = rd[2] = false;
      }
      if (st.dst >= 0 && rd[st.dst] && i > 0) { g.st[i - 1].sync_after = 1; rd[0] = rd[1] = rd[2] = false; }
      rd[st.a0] = true;
      if (st.asw < st.nk) rd[st.a1] = true;
      if (st.res >= 0) rd[st.res] = true;
      st.sync_after = 0;
      if (st.dst >= 0) { st.sync_after = 1; rd[0] = rd[1] = rd[2] = false; }
    }
    static const bool all_sync = getenv("VNR_CHAIN_ALL_BARRIERS") != nullptr;      // A/B switch: a barrier after every stage (round 1 .. early round 2)
    if (all_sync) for (int i = 0; i < g.nstages; ++i) g.st[i].sync_after = 1;
  }
  const int prm_bytes = (g.nstages * 256 + nln * 512) * 4;
  if (g.cpl_stage > 0) {                                // fused coupling: the head pair of a flow step, z rows 16-byte addressable
    if (g.cpl_stage >= g.nstages || g.rows64) return hipErrorInvalidValue;
    const ChainStage& cs = g.st[g.cpl_stage];
    if (cs.n != 128 || cs.dst < 0 || cs.out || cs.gamma || cs.pe || cs.res >= 0 || cs.acc_mode != 0 || cs.act != ACT_IDENTITY || !g.cpl_z ||
        (g.cpl_ld & 3) || (g.cpl_zp_off & 31) || (g.cpl_cond_off & 31) || g.cpl_zp_off + cs.n / 2 > 256 || g.cpl_cond_off + cs.n / 2 > 256)
      return hipErrorInvalidValue;
  }
  if (g.rows64) { if (g.att_stage > 0) return hipErrorInvalidValue; return launch_chain_rt<2>(g, ChainLds<2>::PRM_OFF + prm_bytes, s); }
  int lds = ChainLds<1>::PRM_OFF + prm_bytes;
  if (g.att_stage > 0) {                                // merge scratch of the fused cross-attention: [D / 64 heads][32 rows][66 floats]
    if (g.att_stage >= g.nstages || g.D != 256 || !g.att_K || !g.att_V || g.att_Tk <= 0 || g.att_Tk > 128 || g.att_Tq <= 0) return hipErrorInvalidValue;
    if (g.att_ali && ((g.att_Tk & 3) || ((size_t)g.att_ali & 15))) return hipErrorInvalidValue;
    g.att_lds = (lds + 15) & ~15;
    lds = g.att_lds + (g.D >> 6) * 32 * 66 * 4;
  }
  if (g.cpl_stage > 0) {                                // head exchange [32 rows][2 hc + 4] floats: the attention's merge scratch is free by then
    const int need = 32 * (g.st[g.cpl_stage].n + 4) * 4;
    if (g.att_stage > 0 && g.cpl_stage > g.att_stage && need <= (g.D >> 6) * 32 * 66 * 4) g.cpl_lds = g.att_lds;
    else { g.cpl_lds = (lds + 15) & ~15; lds = g.cpl_lds + need; }
  }
  {
    // (debugging aid, VNR_CHAIN_W8_MASK: 1 = launches with the fused attention, 2 = launches without it, 4 = launches with the fused
    //  coupling, 8 = launches with alignments run on the 8-wave kernel)
    static const int w8mask = getenv("VNR_CHAIN_W8_MASK") ? atoi(getenv("VNR_CHAIN_W8_MASK")) : 0;
    if (w8mask && (((w8mask & 1) && g.att_stage > 0) || ((w8mask & 2) && g.att_stage <= 0) || ((w8mask & 4) && g.cpl_stage > 0) || ((w8mask & 8) && g.att_ali))) g.waves4 = 0;
  }
  if (g.waves4) {                                       // gemm3c.hip takes REGULAR programs only (its single k-loop instance): see there
    for (int i = 0; i < g.nstages && g.waves4; ++i) {
      const ChainStage& st = g.st[i];
      const bool ok = (st.nk == 4 && st.akt0 == 0 && st.asw >= st.nk) || (!(st.nk & 7) && !(st.akt0 & 1) && (st.asw >= st.nk || !(st.asw & 7)));
      if (!ok) g.waves4 = 0;
      // (the 4-wave kernel serves inference programs: no pre-normalisation / statistics outputs of the training chains; its image
      //  stores address the three Q | K | V images through ONE 2 GiB buffer descriptor)
      if (st.out_pre || st.out_stats || (st.out && st.out_fmt != 0 && 3 * st.aoi_img_bytes > 0x7f000000ll)) g.waves4 = 0;
      if (st.kt_total >= 1024 || st.kt0 >= 1024 || st.kt_total < 0 || st.kt0 < 0) g.waves4 = 0;      // (its fetch table packs them into 10 bits each)
    }
    if (g.D != 256 && g.D != 128) g.waves4 = 0;
  }
  if (g.waves4) {                                       // gemm3c.hip turns the V-type stages of a Q|K|V tail through LDS: wave-private 2 x 32 x 33 floats
    bool anyv = false;
    int firstv = -1;
    for (int i = 0; i < g.nstages; ++i) {
      const ChainStage& st = g.st[i];
      const bool isv = st.out_fmt == 4 && st.out && st.aoi_c0 >= 2 * st.aoi_D && !((st.aoi_c0 - 2 * st.aoi_D) & 31) && !(st.aoi_T & 15) && st.acc_mode == 0 &&
                       st.dst < 0 && st.res < 0 && !st.gamma && !st.pe && st.act == ACT_IDENTITY;
      if (isv && firstv < 0) firstv = i;
      anyv = anyv || isv;
    }
    g.vt_lds = 0;
    if (anyv) {
      const int need = 4 * 2 * 32 * 33 * 4;
      if (g.att_stage > 0 && firstv > g.att_stage) g.vt_lds = g.att_lds;       // (the attention phase is over by then; its scratch is the same size)
      else if (g.cpl_stage > 0 && firstv > g.cpl_stage && g.cpl_lds != g.att_lds) {   // ... or the coupling's exchange area, grown to fit (a panel barrier lies between)
        g.vt_lds = g.cpl_lds;
        if (lds < g.cpl_lds + need) lds = g.cpl_lds + need;
      }
      else { g.vt_lds = (lds + 15) & ~15; lds = g.vt_lds + need; }
    }
  }
  {
    // prefetch workgroups on the CUs the row panels leave idle: whole groups of 8 (one per XCD), at most 7 per XCD
    static int cus[kMaxDevices] = {0};
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < kMaxDevices && !cus[dev]) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cus[dev] = pr.multiProcessorCount; }
    const int ncu = (dev >= 0 && dev < kMaxDevices && cus[dev] > 0) ? cus[dev] : 256;
    const int rows = g.rows64 ? 64 : 32;
    // segmented panels: the 4-wave kernel only, whole batch elements, and not when they would cost another round of workgroups
    if (!g.waves4 || g.rows64 || g.seg_T <= 0 || g.M % g.seg_T || !(g.seg_T & 31) ||
        (chain_workers(g, rows) + ncu - 1) / ncu > ((g.M + rows - 1) / rows + ncu - 1) / ncu) g.seg_T = 0;
    const int wgs = chain_workers(g, rows);
    int idle = ncu - wgs;
    idle = idle < 0 ? 0 : (idle > 56 ? 56 : idle);
    g.pf_wgs = (g.pf_progress && g.waves4 && !g.rows64) ? (idle & ~7) : 0;     // (the 4-wave kernel only)
    if (!g.pf_wgs) g.pf_progress = nullptr;
  }
  if (lds > 160 * 1024 && g.waves4) {                  // the 4-wave kernel's extra scratch does not fit: the 8-wave kernel takes the program
    ChainArgs g8 = g_in;
    g8.waves4 = 0;
    return launch_panel_chain(g8, s);
  }
  if (lds > 160 * 1024) {
    if (getenv("VNR_CHAIN_DEBUG")) fprintf(stderr, "panel chain: %d bytes of LDS (stages %d, D %d, attention %d, coupling %d, waves4 %d)\n", lds, g.nstages, g.D, g.att_stage, g.cpl_stage, g.waves4);
    return hipErrorInvalidValue;
  }
  if (g.waves4) return launch_chain4(g, lds, s);
  return launch_chain_rt<1>(g, lds, s);
}

}  // namespace vnr
