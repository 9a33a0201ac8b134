// Backward row-panel chains of a CrossAttentionBLK (training step, round 3) -- the mirror image of gemm3.hip's forward chains.
//
// Between the attention cores the backward pass of a block (reference: tape.gradient of modules/attention.py:440-452 and
// modules/utils.py:48-53, train.py:136) is per-row work again:
//   segment C'  d(block output) -> LayerNorm3 backward -> dense2 data gradient -> relu' -> dense1 data gradient (+ the residual) ->
//               LayerNorm2 backward -> att_proj2 data gradient: d(y) (+ the residual) and d(cross context)
//   segment B'  d(y) -> LayerNorm1 backward -> att_proj1 data gradient: d(x) (+ the residual) and d(self context)
// Unfused that was, per block, 6 data-gradient GEMMs of K = 256 (eight k-tiles per workgroup: launch, prologue and epilogue bound),
// 3 LayerNorm-backward launches, 4 bias-gradient column-sum passes and 3 residual adds -- about 20 launches and 0.4 ms at M = 12800.
// Here a workgroup owns 32 RT rows and walks the segment without leaving the CU, with the machinery of panel_chain_kernel: the
// gradient panel lives in LDS in split-fp16 form, every wave streams the weight operands of its own 32 output columns from an
// operand-major image (of the kernel AS STORED, [K][N]: the data gradient dX = dY.W^T reads W with its roles swapped), products are
// the 3-term hi/lo split with fp32 accumulation.  What the rest of the step needs leaves the kernel as by-products: dv (the gradient
// in front of each LayerNorm = the dY of the Dense before it) and dh go to HBM for the kernel-gradient GEMMs on the side stream
// together with their abs-max words; bias / gamma / beta gradients are column sums reduced across the 32 lanes of a row group, left as
// one row of partials per workgroup and summed by a small second kernel (BwdChainArgs::partial).
//
// Gradients follow the loss scale (KL weight 1e-5 puts whole branches at 1e-9): everything inside the kernel is held multiplied by
// the power of two that maps ~max |incoming gradient| (a word left by the producer) to 2^6; results are scaled back exactly.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace vnr {

namespace {
constexpr unsigned kOobB = 0x80000000u;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <int RT> struct BLds {
  static constexpr int ROWS = 32 * RT;
  static constexpr int PANEL_BYTES = ROWS * 1024;
  static constexpr int P_OFF = 2048 * RT;               // [0, P_OFF): two exchange arrays [ROWS][8] fp32
  static constexpr int TOTAL = P_OFF + 2 * PANEL_BYTES;
};
__device__ __forceinline__ void lds_barrier_b() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ int panel_off_b(int r, int kt, int c) { return r * 1024 + ((((kt << 3) + c) ^ (r & 15)) << 4); }
}  // namespace

template <int RT>
__global__ void __launch_bounds__(512)
bwd_chain_kernel(const BwdChainArgs g) {
  using L = BLds<RT>;
  constexpr int ROWS = L::ROWS, PANEL_BYTES = L::PANEL_BYTES, P_OFF = L::P_OFF;
  constexpr int PF = RT == 2 ? 2 : 4;                   // weight k-tiles in flight per wave (as panel_chain_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * ROWS;
  auto panel_ptr = [&](int i) -> char* { return smem + P_OFF + i * PANEL_BYTES; };
  float* scr = reinterpret_cast<float*>(smem);          // [2][ROWS][8]

  // ---- scale of the gradients inside the kernel ---------------------------------------------------------------------------------
  float s0 = 1.f, inv0 = 1.f;
  {
    unsigned bits = g.amax_in ? *g.amax_in : 0u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < g.npre && g.amax_pre[i]) { const unsigned b2 = *g.amax_pre[i]; bits = b2 > bits ? b2 : bits; }      // (non-negative floats: integer order)
    const int e = (int)(bits >> 23) & 0xff;
    if (e > 0 && e < 255) {
      int sft = 6 - (e - 127);
      if (sft > 120) sft = 120; if (sft < -120) sft = -120;
      s0 = __uint_as_float((unsigned)(sft + 127) << 23);
      inv0 = __uint_as_float((unsigned)(-sft + 127) << 23);
    }
  }
  const float wsc = 1.f / 256.f;                        // the weight images are pre-scaled by 256 (train.inc)

  // ---- the weight stream: the GEMM stages of the segment as one flat sequence of k-tiles (eight per stage) ------------------------
  const int nch = g.seg == 0 ? g.F >> 8 : 0;
  const int npre = g.npre;
  const int nstages = npre + 2 * nch + 2;
  h16x8 wreg[PF][4];
  int fs = 0, fk = 0;
  unsigned fvoff = kOobB;
  __amdgpu_buffer_rsrc_t frs;
  auto open_stage = [&](int s_) {
    const void* w; int kt_total, kt0, cb0;
    if (s_ < npre) { w = s_ == 0 ? g.pre_w[0] : s_ == 1 ? g.pre_w[1] : g.pre_w[2]; kt_total = 8; kt0 = 0; cb0 = 0; }                 // head: d(output of a reader of the LayerNorm output) . W^T
    else if ((s_ -= npre) < 2 * nch) {
      const int ch = s_ >> 1;
      if (!(s_ & 1)) { w = g.w2r; kt_total = 8; kt0 = 0; cb0 = 8 * ch; }              // d(hidden chunk) = dv3 . W2^T[:, chunk]
      else { w = g.w1r; kt_total = g.F >> 5; kt0 = 8 * ch; cb0 = 0; }                  // d(o) += d(hidden chunk) . W1^T[chunk, :]
    } else { w = g.pr; kt_total = 8; kt0 = 0; cb0 = (s_ - 2 * nch) * 8; }              // att_proj data gradient, halves 0 / 1
    frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 0x40000000, 0x00020000);
    fvoff = (unsigned)(((cb0 + wave) * kt_total + kt0) * 4096 + lane * 16);
  };
  auto fetch = [&](int u) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wreg[u][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, fvoff, fk * 4096 + i * 1024, 0));
    if (++fk == 8) {
      fk = 0;
      if (fs + 1 < nstages) { ++fs; open_stage(fs); } else { fvoff = kOobB; }           // past the end: dummy refills (zeros, no traffic)
    }
  };
  open_stage(0);
#pragma unroll
  for (int u = 0; u < PF; ++u) fetch(u);

  // ---- one GEMM stage: acc = A(panel) . W (this wave's 32 columns), eight k-tiles ---------------------------------------------------
  f32x16 acc[RT];
  auto kloop = [&](const char* Ap) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    h16x8 afr[2][RT][4];
    auto read_a = [&](int kt, int set) {
      const int kc = kt < 8 ? kt : 7;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          afr[set][rt][2 * t] = *reinterpret_cast<const h16x8*>(Ap + panel_off_b(32 * rt + l31, kc, 2 * t + half));
          afr[set][rt][2 * t + 1] = *reinterpret_cast<const h16x8*>(Ap + panel_off_b(32 * rt + l31, kc, 4 + 2 * t + half));
        }
    };
    read_a(0, 0);
#pragma unroll 1
    for (int kb = 0; kb < 8; kb += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        read_a(kb + u + 1, (u + 1) & 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t], afr[u & 1][rt][2 * t], acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t], afr[u & 1][rt][2 * t + 1], acc[rt], 0, 0, 0);
            acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][2 * t + 1], afr[u & 1][rt][2 * t], acc[rt], 0, 0, 0);
          }
        fetch(u);
      }
    }
  };

  // ---- helpers of the epilogues.  Lane (row 32 rt + l31, half) of wave w holds columns 32 w + 8 q + 4 half + e (register 4 q + e) -------
  auto store_panel = [&](int pi, const float (&x)[RT][16]) {            // split fp16 hi | lo into panel pi (k-tile = this wave's index)
    char* Dp = panel_ptr(pi);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h16x4 hi, lo;
        { const vnr_f4 xs_ = {x[rt][4 * q + 0], x[rt][4 * q + 1], x[rt][4 * q + 2], x[rt][4 * q + 3]}; vnr_split(xs_, hi, lo); }
        const int p = 8 * q + 4 * half;
        *reinterpret_cast<h16x4*>(Dp + panel_off_b(32 * rt + l31, wave, p >> 3) + (p & 4) * 2) = hi;
        *reinterpret_cast<h16x4*>(Dp + panel_off_b(32 * rt + l31, wave, 4 + (p >> 3)) + (p & 4) * 2) = lo;
      }
  };
  auto load_panel = [&](int pi, float (&x)[RT][16]) {                   // this lane's own entries of panel pi (hi + lo: 22 bits)
    const char* Sp = panel_ptr(pi);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = 8 * q + 4 * half;
        const h16x4 rh = *reinterpret_cast<const h16x4*>(Sp + panel_off_b(32 * rt + l31, wave, p >> 3) + (p & 4) * 2);
        const h16x4 rl = *reinterpret_cast<const h16x4*>(Sp + panel_off_b(32 * rt + l31, wave, 4 + (p >> 3)) + (p & 4) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[rt][4 * q + e] = (float)rh[e] + (float)rl[e];
      }
  };
  // rows of a [M][ld] fp32 matrix, this lane's 16 columns of column window c0 (rows beyond M read as 0)
  auto load_rows = [&](const float* src, int ld, int c0, float (&x)[RT][16]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = m0 + 32 * rt + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < g.M) v4 = *reinterpret_cast<const float4*>(src + (size_t)row * ld + c0 + 32 * wave + 8 * q + 4 * half);
        x[rt][4 * q] = v4.x; x[rt][4 * q + 1] = v4.y; x[rt][4 * q + 2] = v4.z; x[rt][4 * q + 3] = v4.w;
      }
    }
  };
  // x * mul -> HBM rows (assign, or add to what is there); the lane's running max |x * mul| is kept in `mx` (one atomicMax per
  // workgroup and word at the very end: 1600 waves hammering one word per store was measurable)
  auto store_rows = [&](float* dst, int ld, int c0, const float (&x)[RT][16], float mul, bool add, float& mx) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = m0 + 32 * rt + l31;
      if (row >= g.M) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4* p = reinterpret_cast<float4*>(dst + (size_t)row * ld + c0 + 32 * wave + 8 * q + 4 * half);
        float4 o = make_float4(x[rt][4 * q] * mul, x[rt][4 * q + 1] * mul, x[rt][4 * q + 2] * mul, x[rt][4 * q + 3] * mul);
        if (add) { const float4 old = *p; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *p = o;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      }
    }
  };
  float mxA = 0.f, mxH = 0.f, mxB = 0.f, mx1 = 0.f, mx0 = 0.f;
  // column sums over the workgroup's rows of a per-lane array (already summed over the row tiles): 32 lanes of a half -> lane 0 of it,
  // which stores them into this workgroup's row of the partial-sum matrix at column offset `off`
  const int pcols = g.seg == 0 ? 6 * 256 + g.F : 3 * 256;
  float* prow = g.partial + (size_t)blockIdx.x * pcols;
  auto colsum_flush = [&](const float (&c)[16], int off, float mul) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = c[4 * q + e];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
        x[e] = v * mul;
      }
      if (l31 == 0) *reinterpret_cast<float4*>(prow + off + 32 * wave + 8 * q + 4 * half) = make_float4(x[0], x[1], x[2], x[3]);
    }
  };
  // LayerNormalization backward (eps folded into the saved rstd): dy (scaled, per lane) -> dx (scaled) in place; the row's mean and
  // 1/std come from the forward chain (ChainStage::out_stats); dgamma / dbeta column sums are flushed here
  auto ln_bwd = [&](float (&dy)[RT][16], const float* v, const float* stats, const float* gamma, int off_dgamma, int off_dbeta) {
    float ga[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 g4 = *reinterpret_cast<const float4*>(gamma + 32 * wave + 8 * q + 4 * half);
      ga[4 * q] = g4.x; ga[4 * q + 1] = g4.y; ga[4 * q + 2] = g4.z; ga[4 * q + 3] = g4.w;
    }
    float xh[RT][16], rs[RT];
    load_rows(v, 256, 0, xh);
    float cg[16], cb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cg[i] = 0.f; cb[i] = 0.f; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = m0 + 32 * rt + l31;
      float2 ms = make_float2(0.f, 0.f);
      if (row < g.M) ms = *reinterpret_cast<const float2*>(stats + 2 * (size_t)row);
      rs[rt] = ms.y;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float x = (xh[rt][i] - ms.x) * ms.y;
        const float d = dy[rt][i];
        const float gv = d * ga[i];
        xh[rt][i] = x; dy[rt][i] = gv;
        s1 += gv; s2 += gv * x;
        cg[i] += d * x; cb[i] += d;
      }
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (half == 0) { scr[(32 * rt + l31) * 8 + wave] = s1; scr[ROWS * 8 + (32 * rt + l31) * 8 + wave] = s2; }
    }
    lds_barrier_b();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(scr + (32 * rt + l31) * 8), a1 = *reinterpret_cast<const f32x4*>(scr + (32 * rt + l31) * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(scr + ROWS * 8 + (32 * rt + l31) * 8), b1 = *reinterpret_cast<const f32x4*>(scr + ROWS * 8 + (32 * rt + l31) * 8 + 4);
      const float S1 = ((a0[0] + a0[1]) + (a0[2] + a0[3]) + (a1[0] + a1[1]) + (a1[2] + a1[3])) * (1.f / 256.f);
      const float S2 = ((b0[0] + b0[1]) + (b0[2] + b0[3]) + (b1[0] + b1[1]) + (b1[2] + b1[3])) * (1.f / 256.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) dy[rt][i] = rs[rt] * (dy[rt][i] - S1 - xh[rt][i] * S2);
    }
    colsum_flush(cg, off_dgamma, inv0);
    colsum_flush(cb, off_dbeta, inv0);
    lds_barrier_b();                                                    // the exchange arrays are free again
  };
  auto bias_sum = [&](const float (&x)[RT][16], int off) {            // bias gradient: column sums of a gradient about to leave
    float c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { c[i] = 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) c[i] += x[rt][i]; }
    colsum_flush(c, off, inv0);
  };

  // ================= head: incoming gradient -> LayerNorm backward -> dvA ================================================================
  float d[RT][16];
  if (npre > 0) {
    f32x16 accP[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accP[rt][r] = 0.f;
#pragma unroll 1
    for (int i = 0; i < npre; ++i) {
      load_rows(i == 0 ? g.pre_src[0] : i == 1 ? g.pre_src[1] : g.pre_src[2], 256, 0, d);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 16; ++j) d[rt][j] *= s0;
      store_panel(1, d);
      lds_barrier_b();
      kloop(panel_ptr(1));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accP[rt][r] += acc[rt][r];
      lds_barrier_b();                                                   // panel 1 may be rewritten
    }
    load_rows(g.dy, g.ld_dy, 0, d);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) d[rt][i] = d[rt][i] * s0 + accP[rt][i] * wsc;
  } else {
    load_rows(g.dy, g.ld_dy, 0, d);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) d[rt][i] *= s0;
  }
  ln_bwd(d, g.vA, g.stA, g.gA, 0, 256);
  store_rows(g.dvA, 256, 0, d, inv0, false, mxA);
  bias_sum(d, 512);
  store_panel(0, d);
  lds_barrier_b();

  if (g.seg == 0) {
    // ================= FFN: hidden chunks of 256 columns ==============================================================================
    f32x16 accF[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accF[rt][r] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
      float hm[RT][16];
      load_rows(g.hdn, g.F, 256 * ch, hm);                               // (requested before the k-loop: 52 MB of hidden activations per launch)
      kloop(panel_ptr(0));                                               // d(hidden chunk) before the activation's derivative
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 16; ++i) hm[rt][i] = hm[rt][i] > 0.f ? acc[rt][i] * wsc : 0.f;      // relu': utils.py:49
      store_rows(g.dh, g.F, 256 * ch, hm, inv0, false, mxH);
      bias_sum(hm, 768 + 256 * ch);
      store_panel(1, hm);                                                // (every wave left the previous chunk's second k-loop: barrier below)
      lds_barrier_b();
      kloop(panel_ptr(1));                                               // d(o) += d(hidden chunk) . W1^T
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accF[rt][r] += acc[rt][r];
      lds_barrier_b();                                                   // panel 1 may be rewritten
    }
    // ================= d(o) = FFN path + the residual dv3 -> LayerNorm2 backward -> dv2 ===============================================
    load_panel(0, d);                                                    // dv3 (this lane's entries)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 16; ++i) d[rt][i] += accF[rt][i] * wsc;
    ln_bwd(d, g.vB, g.stB, g.gB, 768 + g.F, 1024 + g.F);
    store_rows(g.dvB, 256, 0, d, inv0, false, mxB);
    bias_sum(d, 1280 + g.F);
    store_panel(0, d);                                                   // (in place: every lane rewrites exactly the entries it read)
    lds_barrier_b();
  }
  // ================= att_proj data gradient: first half + the residual, second half ======================================================
  kloop(panel_ptr(0));
  load_panel(0, d);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) d[rt][i] += acc[rt][i] * wsc;
  store_rows(g.out0, 256, 0, d, inv0, g.acc0 != 0, mx0);
  kloop(panel_ptr(0));
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) d[rt][i] = acc[rt][i] * wsc;
  store_rows(g.out1, 256, 0, d, inv0, false, mx1);
  // ---- abs-max by-products: lanes -> wave -> workgroup (through the exchange array), one atomicMax per word ---------------------------
  {
    float m4[5] = {mxA, mxH, mxB, mx1, mx0};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m4[i] = fmaxf(m4[i], __shfl_xor(m4[i], o, 64));
    }
    lds_barrier_b();                                                     // (the exchange arrays are idle: every LayerNorm stage is behind us)
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) scr[wave * 8 + i] = m4[i];
    }
    lds_barrier_b();
    if (tid < 5) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) m = fmaxf(m, scr[w * 8 + tid]);
      unsigned* dst = tid == 0 ? g.amaxA : tid == 1 ? g.amax_dh : tid == 2 ? g.amaxB : tid == 3 ? g.amax_out1 : g.amax_out0;
      if (dst) amax_publish(dst, m);
    }
  }
}

// second kernel of a backward-chain launch: the per-workgroup column-sum partials -> the gradients (one thread per column; every
// destination is written by this launch only, so a plain read-add-write)
struct ColFinishArgs { const float* partial; int nwg, pcols, nseg; float* dst[8]; int off[9]; };
__global__ void __launch_bounds__(256) bwd_chain_colsum_kernel(const ColFinishArgs a) {
  __shared__ float part[8][32];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;                  // 32 columns x 8 row groups per workgroup
  const int j = blockIdx.x * 32 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < a.pcols) {
    int w = rg;
    for (; w + 24 < a.nwg; w += 32) {
      s0 += a.partial[(size_t)w * a.pcols + j]; s1 += a.partial[(size_t)(w + 8) * a.pcols + j];
      s2 += a.partial[(size_t)(w + 16) * a.pcols + j]; s3 += a.partial[(size_t)(w + 24) * a.pcols + j];
    }
    for (; w < a.nwg; w += 8) s0 += a.partial[(size_t)w * a.pcols + j];
  }
  part[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && j < a.pcols) {
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) tot += part[r][c];
    int sg = 0;
    while (sg + 1 < a.nseg && j >= a.off[sg + 1]) ++sg;
    a.dst[sg][j - a.off[sg]] += tot;
  }
}

// finish_stream: where the small kernel that adds up the per-workgroup partial column sums (bias / gamma / beta gradients) runs; the
// caller orders it after `s` (the training step puts it on its kernel-gradient stream: it feeds nothing but gradients)
hipError_t launch_bwd_chain(const BwdChainArgs& g, int rows64, hipStream_t s, hipStream_t finish_stream, int part) {
  if (g.M <= 0 || (g.seg != 0 && g.seg != 1) || !g.dy || !g.vA || !g.stA || !g.gA || !g.dvA || !g.dgA || !g.dbA || !g.dbiasA || !g.pr ||
      !g.out0 || !g.out1 || (g.ld_dy & 3) || !g.partial || g.npre < 0 || g.npre > 3)
    return hipErrorInvalidValue;
  for (int i = 0; i < g.npre; ++i) if (!g.pre_src[i] || !g.pre_w[i]) return hipErrorInvalidValue;
  if (g.seg == 0 && (!g.w2r || !g.w1r || g.F <= 0 || (g.F & 255) || !g.hdn || !g.dh || !g.dbias1 || !g.vB || !g.stB || !g.gB || !g.dvB || !g.dgB ||
                     !g.dbB || !g.dbiasB))
    return hipErrorInvalidValue;
  const int rows = rows64 ? 64 : 32, nwg = (g.M + rows - 1) / rows;
  if (part != 2) {                                          // part: 0 both kernels, 1 the chain only, 2 the finish only
    if (rows64) {
      static int done[kMaxDevices] = {0};
      opt_in_dynamic_lds((const void*)bwd_chain_kernel<2>, BLds<2>::TOTAL, done);
      vnr_launch(bwd_chain_kernel<2>, dim3(nwg), dim3(512), BLds<2>::TOTAL, s, g);
    } else {
      static int done[kMaxDevices] = {0};
      opt_in_dynamic_lds((const void*)bwd_chain_kernel<1>, BLds<1>::TOTAL, done);
      vnr_launch(bwd_chain_kernel<1>, dim3(nwg), dim3(512), BLds<1>::TOTAL, s, g);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || part == 1) return e;
  }
  ColFinishArgs c;
  c.partial = g.partial; c.nwg = nwg; c.pcols = bwd_chain_pcols(g.seg, g.F);
  if (g.seg == 0) {
    float* d[7] = {g.dgA, g.dbA, g.dbiasA, g.dbias1, g.dgB, g.dbB, g.dbiasB};
    const int o[8] = {0, 256, 512, 768, 768 + g.F, 1024 + g.F, 1280 + g.F, 1536 + g.F};
    c.nseg = 7;
    for (int i = 0; i < 7; ++i) { c.dst[i] = d[i]; c.off[i] = o[i]; }
    c.off[7] = o[7];
  } else {
    float* d[3] = {g.dgA, g.dbA, g.dbiasA};
    c.nseg = 3;
    for (int i = 0; i < 3; ++i) { c.dst[i] = d[i]; c.off[i] = 256 * i; }
    c.off[3] = 768;
  }
  vnr_launch(bwd_chain_colsum_kernel, dim3((c.pcols + 31) / 32), dim3(256), 0, finish_stream ? finish_stream : s, c);
  return hipGetLastError();
}

}  // namespace vnr
