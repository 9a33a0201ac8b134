// Row-panel chain kernel, second generation (round 5): ONE wave per SIMD.  Same programs, same stage descriptors, same LDS panels
// and the same arithmetic as panel_chain_kernel<1> (gemm3.hip) -- the per-row dense chains of a CrossAttentionBLK
// (modules/attention.py:440-452), its FFN (modules/utils.py:48-53), the fused cross-attention (attention.py:224-246), the flow
// coupling (modules/flow.py:216-239) and the pre-chains behind it -- but a workgroup is 4 waves of 64 output columns instead of
// 8 waves of 32:
//   * a wave owns a SIMD and its whole 512-entry register file: EIGHT k-tiles of weight operands (a complete K = 256 stage,
//     64 one-KiB loads) are in flight per wave instead of four half-sized ones, so the next stage's weights stream in THROUGH the
//     epilogue / LayerNorm / attention phases of the current one;
//   * the A operands (activations from the LDS panel) are read once per k-tile for both column tiles of the wave: half the LDS reads;
//   * consecutive MFMAs go to alternating accumulators with the refill loads between them: the k-loop free-runs at the
//     vector-memory path's 512 clk per k-tile round (tools/probes/chain_b_probe.hip, profiles/r05_chain_b_probe.txt: a stage of
//     8 rounds + a short epilogue 5.7 kcyc against 7.1 for the 8-wave layout, whose younger wave of every SIMD pair finishes each
//     k-loop ~2 kcyc after the older one);
//   * the fused cross-attention needs no merge: a wave IS a head and walks all key blocks itself (no LDS exchange, no barrier
//     inside the phase).
// Selected by ChainArgs::waves4 (engine option "chain_waves4", default on; 32-row panels only).
#include "common.h"
#include "chain_prefetch.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace vnr {

namespace {
constexpr unsigned kOob3 = 0x80000000u;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds3_t;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// LDS map (identical to ChainLds<1> of gemm3.hip, so launch_panel_chain's size arithmetic serves both kernels):
//   [0, 2048)              LayerNorm exchange scratch [2][32][4] fp32 (half used)
//   [2048, + 3 x 32 KiB)   the activation panels: 32 rows x 8 k-tiles x 128 B
//   then                   bias of every stage [nstages][256] fp32, then (gamma | beta) [256 + 256] fp32 of every LayerNorm stage
constexpr int kRows = 32, kPanelBytes = 32768, kPOff = 2048, kPrmOff = kPOff + 3 * kPanelBytes;
constexpr int kNW = 4;              // waves per workgroup
constexpr int kDepth = 8;           // weight k-tiles in flight per wave = one trip of the k-loop

__device__ __forceinline__ void lds_barrier4() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ int panel_off4(int r, int kt, int c) { return r * 1024 + ((((kt << 3) + c) ^ (r & 15)) << 4); }
// x = hi + lo in fp16, the low part derived from the STORED high bits.  Written "h = (_Float16)x; hi = h; lo = (_Float16)(x - (float)h)"
// behind "x = O * linv" the compiler (fp contraction, hipcc's default) evaluated h twice: fused into the multiply for the value it
// subtracted (v_fma_mixlo_f16: ONE rounding of the exact product) and as v_cvt_pk_f16_f32 of the rounded fp32 product for the value it
// stored (TWO roundings).  The two differ by one fp16 ulp whenever the fp32 rounding crosses an fp16 tie -- about one element in ten
// thousand of the attention context came out hi + lo = x +- 2^-9, which cost this kernel 1e-4 at the mel until the last day of round 5
// (profiles/r05_experiments.txt r05i, profiles/r05_cvt_pk_probe.txt).  Vector-typed conversions give the compiler ONE node for the high
// part: whatever instruction it picks, the stored bits and the subtracted ones are the same.  (An empty asm on the packed register does
// it too, but pins registers: two spilled.)
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_hi_lo(const float (&x)[4], h16x4& hi, h16x4& lo) {
  const f32x4v xv = {x[0], x[1], x[2], x[3]};
  hi = __builtin_convertvector(xv, h16x4);               // ONE conversion node: the stored bits and the ones subtracted are the same
  lo = __builtin_convertvector(xv - __builtin_convertvector(hi, f32x4v), h16x4);
}
__device__ __forceinline__ void split_hi_lo(const float (&x)[8], h16x8& hi, h16x8& lo) {
  const f32x8v xv = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
  hi = __builtin_convertvector(xv, h16x8);
  lo = __builtin_convertvector(xv - __builtin_convertvector(hi, f32x8v), h16x8);
}
// split-fp16 store of 4 consecutive columns [p, p + 4) (p % 4 == 0) of k-tile kt, row r
__device__ __forceinline__ void panel_put4(char* P, int r, int kt, int p, const float* x) {
  h16x4 hi, lo;
  const float xx[4] = {x[0], x[1], x[2], x[3]};
  split_hi_lo(xx, hi, lo);
  *reinterpret_cast<h16x4*>(P + panel_off4(r, kt, p >> 3) + (p & 4) * 2) = hi;
  *reinterpret_cast<h16x4*>(P + panel_off4(r, kt, 4 + (p >> 3)) + (p & 4) * 2) = lo;
}
}  // namespace

__global__ void __launch_bounds__(256)
panel_chain4_kernel(const ChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int nworkers = chain_workers(g, kRows);            // L2 warming (chain_prefetch.h)
    if ((int)blockIdx.x >= nworkers) { chain_prefetch_role(g, ((int)blockIdx.x - nworkers) >> 3, (g.pf_wgs + 7) >> 3, g.dbg_ts ? g.dbg_ts + (size_t)blockIdx.x * 128 : nullptr); return; }
  }
  // rows [m0, mend) of this workgroup.  Segmented launches (ChainArgs::seg_T): the panels of a batch element start at its first row,
  // so no panel straddles two elements (the fused attention would run once per element: 8 of 200 workgroups of an S1 block launch
  // -- T = 400 = 12.5 panels -- took 14 kcyc longer than the rest, and a launch lasts as long as its slowest workgroup)
  int m0 = blockIdx.x * kRows, mend = g.M;
  if (g.seg_T > 0) {
    const int ppb = (g.seg_T + kRows - 1) / kRows, b = (int)blockIdx.x / ppb;
    m0 = b * g.seg_T + ((int)blockIdx.x - b * ppb) * kRows;
    mend = (b + 1) * g.seg_T;
  }
  auto panel_ptr = [&](int i) -> char* { return smem + kPOff + i * kPanelBytes; };
  float* scratch = reinterpret_cast<float*>(smem);
  float* prm = reinterpret_cast<float*>(smem + kPrmOff);
  unsigned long long* ts = g.dbg_ts ? g.dbg_ts + (size_t)blockIdx.x * 128 : nullptr;
  auto stamp = [&](int i) { if (ts && tid == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  auto wstamp = [&](int si, int i) { if (ts && si == g.dbg_stage && lane == 0) ts[64 + wave * 8 + i] = __builtin_amdgcn_s_memtime(); };
  // (measurement, VNR_CHAIN_TS_SLOTS=1: dbg_stage carries 0x100 -- the per-wave stamps are then the ends of the 8 k-loop slots of the LAST trip
  //  of that stage, [32 + wave] the start of its k-loop: tools/r05_slot_ts.py)
  auto sstamp = [&](int si, int i) { if (ts && (si | 0x100) == g.dbg_stage && lane == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (ts && tid == 0) ts[62] = chain_xcc_id();            // (measurement: s_memtime counts per XCD)

  // ---- the weight stream --------------------------------------------------------------------------------------------
  // Operand-major images as in gemm3.hip: block (32-column block, k-tile) = 4 x 1 KiB pieces (piece 2t + part: k16 step t, hi / lo);
  // a wave owns column blocks 2 wave and 2 wave + 1.  The k-tiles of all stages form one flat sequence of TRIPS of 8 k-tiles
  // (a stage is padded to whole trips; padding tiles, tiles of column blocks a stage does not have, and the tail after the last
  // stage are out-of-range buffer reads: zeros, no traffic).  Slot u of the register ring always holds k-tile u of the trip being
  // multiplied; each of its 8 pieces is re-requested for the NEXT trip right behind the last MFMA that read it -- the fetch cursor
  // (fs, fk) therefore runs exactly one trip ahead of the multiplier, across stage boundaries, and every wait is a static vmcnt.
  h16x8 wreg[kDepth][2][4];
  int fs = 0, fk = 0, fnk = 0, fpad = 0;
  unsigned fvoff0 = kOob3;                               // offset of this lane in column block 2 wave of the fetched stage (or out of range)
  int fjd = 0;                                           // + this (a scalar) = column block 2 wave + 1; fj1: that block exists
  bool fj1 = false;
  __amdgpu_buffer_rsrc_t frs;
  auto set_stage = [&](const void* w, int nk, int n, int ktt, int k0) {
    frs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, 0x40000000, 0x00020000);
    fnk = nk;
    fpad = (nk + kDepth - 1) / kDepth * kDepth;
    fvoff0 = (64 * wave < n) ? (unsigned)(((2 * wave) * ktt + k0) * 4096 + lane * 16) : kOob3;
    fjd = ktt * 4096;
    fj1 = 64 * wave + 32 < n;
  };
  // The fetch cursor opens a stage from a table held in three REGISTERS -- lane s = stage s: the image address as a distance from stage
  // 0's image (two dwords) and nk | n / 4 | kt_total | kt0 packed into one -- with three v_readlane.  Through the kernel-argument segment
  // it was two dependent rounds of scalar loads behind the last k-tile of every trip: ~1 kcyc that one wave per SIMD cannot hide
  // (per-slot stamps, profiles/r05_experiments.txt).  Only the head of the stream (prologue) still opens stages that way.
  unsigned ft_lo = 0, ft_hi = 0, ft_pk = 0;
  {
    // (through the kernel-argument POINTER: g.st[lane] with a lane-varying index makes the compiler copy all of `g` to scratch)
    typedef const __attribute__((address_space(4))) char* kbytes_t;
    const int sl = lane < g.nstages ? lane : 0;
    kbytes_t sp_ = (kbytes_t)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainArgs, st) + (size_t)sl * sizeof(ChainStage);
    const unsigned long long wd = *reinterpret_cast<const __attribute__((address_space(4))) unsigned long long*>(sp_ + offsetof(ChainStage, w)) -
                                  reinterpret_cast<unsigned long long>(g.st[0].w);
    const unsigned nk_ = *reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(sp_ + offsetof(ChainStage, nk));
    const unsigned n_ = *reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(sp_ + offsetof(ChainStage, n));
    const unsigned ktt_ = *reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(sp_ + offsetof(ChainStage, kt_total));
    const unsigned k0_ = *reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(sp_ + offsetof(ChainStage, kt0));
    ft_lo = (unsigned)wd; ft_hi = (unsigned)(wd >> 32);
    ft_pk = nk_ | ((n_ >> 2) << 5) | (ktt_ << 12) | (k0_ << 22);          // nk <= 16, n / 4 <= 64, kt_total, kt0 < 1024 (launch_panel_chain checks)
  }
  auto open_stage_k = [&](int s_) {
    const ChainStage& st = g.st[s_];
    set_stage(st.w, st.nk, st.n, st.kt_total, st.kt0);
  };
  auto open_stage = [&](int s_) {
    const long long wd = (long long)((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ft_lo, s_) |
                                     ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ft_hi, s_) << 32));
    const unsigned pk = (unsigned)__builtin_amdgcn_readlane((int)ft_pk, s_);
    set_stage(static_cast<const char*>(g.st[0].w) + wd, (int)(pk & 31u), (int)(((pk >> 5) & 127u) << 2), (int)((pk >> 12) & 1023u), (int)(pk >> 22));
  };
  // Validity of a fetched k-tile (kt < nk of the fetched stage) is decided once per TRIP for three slot groups -- a stage has at least
  // two k-tiles: slots 0-1 always carry one, slots 2-3 iff more than 2 remain, slots 4-7 iff more than 4 -- so the loop body holds one
  // select per piece (on a scalar condition); the piece index rides in the instruction's immediate offset, the k-tile in the scalar offset.  (The validity cannot live in
  // the descriptor -- zero records for an invalid group: the hardware checks offset >= num_records - scalar offset, which wraps.)
  int fleft = 0;                                         // fnk - fk of the fetched trip (a scalar: the six selected offsets as registers
                                                         // were the ones the allocator spilled across the stage loop)
  int fsoff = 0;                                         // fk * 4096
  auto trip_offsets = [&]() {
    fleft = fnk - fk;
    fsoff = fk * 4096;
  };
  auto piece = [&](int u, int j, int i) {
    const bool ok = fleft > (u < 2 ? 0 : (u < 4 ? 2 : 4));
    const unsigned v = (ok && (j == 0 || fj1)) ? fvoff0 + (j ? (unsigned)fjd : 0u) : kOob3;      // (one register for both column blocks)
    wreg[u][j][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(frs, v + i * 1024, fsoff + u * 4096, 0));
  };
  auto advance = [&]() {
    fk += kDepth;
    if (fk >= fpad) {
      fk = 0;
      if (fs + 1 < g.nstages) { ++fs; open_stage(fs); } else { fnk = 0; }
    }
    trip_offsets();
  };
  // ---- input panels (fp32 rows in HBM -> split fp16 panel): all reads issued first, rows beyond M read as zeros --------
  {
    const int q4 = g.D >> 2;                                       // float4 per row (<= 64)
    float4 x[2][8];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      const float* src = pi == 0 ? g.in0 : g.in1;
      const int ld = pi == 0 ? g.ld0 : g.ld1;
      // buffer reads: a missing source, rows beyond M and slots beyond the panel are out-of-range offsets (zeros, no branch per read)
      const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src ? src + (size_t)m0 * ld : g.in0), 0,
                                                                            src ? 0x40000000u : 0u, 0x00020000);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int e = tid + 256 * it, r = e / q4, j = e - r * q4;
        const unsigned off = (r < kRows && m0 + r < mend) ? (unsigned)((r * ld + 4 * j) * 4) : kOob3;
        x[pi][it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
      }
    }
    stamp(56);
    // The head of the weight stream goes out BEHIND the rows (memory returns in order), and in two halves around the conversion: k-tiles
    // 0-3 travel while the rows are split into their panels, k-tiles 4-7 and the parameter DMA follow.  (All 64 loads of a wave first:
    // the conversion then waited for the whole head -- 8.9 kcyc of issue back-pressure at launch start -- before it touched a row.)
    open_stage_k(0);
    trip_offsets();
#pragma unroll
    for (int u = 0; u < kDepth / 2; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) piece(u, j, i);
    stamp(59);
    // What the rows do not overwrite starts as zeros: panel 2, a panel without a source, and the k-tiles beyond D of the others -- a short
    // stage (nk = 4) multiplies tiles 4-7 of its source rows with zero weights, and 0 x NaN from whatever the LDS held would poison the
    // accumulators.  (Disjoint from what the conversion writes: no barrier between the two.)
    {
      const int dt = g.D >> 5;                                      // k-tiles the rows fill
#pragma unroll
      for (int pi = 0; pi < 3; ++pi) {
        const bool whole = pi == 2 || !(pi == 0 ? g.in0 : g.in1);
        if (!whole && dt == 8) continue;
        char* P = panel_ptr(pi);
        // a row is 8 k-tiles x 8 chunks of 16 bytes (XOR-swizzled within the row): chunk index c of row r holds tile (c ^ (r & 15)) >> 3
        for (int o = tid; o < kRows * 64; o += 256) {
          const int r = o >> 6, c = o & 63;
          if (whole || (((c ^ (r & 15)) >> 3) >= dt)) *reinterpret_cast<float4*>(P + r * 1024 + c * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
      if (!(pi == 0 ? g.in0 : g.in1)) continue;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int e = tid + 256 * it, r = e / q4, j = e - r * q4;
        if (r >= kRows) continue;
        const float xv[4] = {x[pi][it].x, x[pi][it].y, x[pi][it].z, x[pi][it].w};
        panel_put4(panel_ptr(pi), r, j >> 3, (j & 7) * 4, xv);
      }
    }
    stamp(58);
#pragma unroll
    for (int u = kDepth / 2; u < kDepth; ++u)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) piece(u, j, i);
    // (the head of the stream: stage 1 is opened through the kernel arguments -- written out, not a flag of advance(): two paths to
    //  &g.st[fs] that meet make the compiler copy all of `g` to scratch)
    fk += kDepth;
    if (fk >= fpad) {
      fk = 0;
      if (g.nstages > 1) { fs = 1; open_stage_k(1); } else { fnk = 0; }
    }
    trip_offsets();
    // ---- epilogue parameters of the whole program -> LDS by LDS-DMA (stage s by wave s mod 4) ----------------------------------
    if (g.prm) {
      const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.prm), 0, (unsigned)g.nstages * 3072u, 0x00020000);
      for (int s_ = wave; s_ < g.nstages; s_ += kNW) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + kPrmOff + s_ * 1024), 16, (unsigned)(s_ * 3072 + lane * 16), 0, 0, 0);
        const int lo = g.st[s_].lds_ln;
        if (lo >= 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + kPrmOff + lo * 4), 16, (unsigned)(s_ * 3072 + 1024 + lane * 16), 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds3_t)(smem + kPrmOff + lo * 4 + 1024), 16, (unsigned)(s_ * 3072 + 2048 + lane * 16), 0, 0, 0);
        }
      }
    } else {
      for (int s_ = wave; s_ < g.nstages; s_ += kNW) {
        const float* bp = g.st[s_].bias; const float* gp = g.st[s_].gamma; const float* ep = g.st[s_].beta;
        const unsigned nb = (unsigned)g.st[s_].n * 4u;
        const bool skip = g.st[s_].acc_mode == 1 || g.st[s_].acc_mode == 2;
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bp ? bp : g.in0), 0, (bp && !skip) ? nb : 0u, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds3_t)(smem + kPrmOff + s_ * 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
        const int lo = g.st[s_].lds_ln;
        if (lo >= 0) {
          const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp), 0, nb, 0x00020000);
          const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ep ? ep : gp), 0, ep ? nb : 0u, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds3_t)(smem + kPrmOff + lo * 4), 16, (unsigned)(lane * 16), 0, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsE, (lds3_t)(smem + kPrmOff + lo * 4 + 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
        }
      }
    }

  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the first trip of weights and the parameter DMA landed
  lds_barrier4();
  stamp(1);

  f32x16 accF[2];                                       // persistent accumulators of the FFN second layer
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) accF[j][r] = 0.f;

#pragma unroll 1
  for (int si = 0; si < g.nstages; ++si) {
    // every lane-dependent address of a stage is re-derived from an opaque copy of the lane id (gemm3.hip: otherwise the compiler
    // hoists dozens of loop-invariant offsets out of the stage loop and spills them)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    const int half = lane_s >> 5, l31 = lane_s & 31;
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));
    if (tid == 0) chain_publish_stage(g, si);               // (chain_prefetch.h)
    if (ts && tid == 0) ts[96 + si] = __builtin_amdgcn_s_memrealtime();      // (measurement: 100 MHz, the same clock on every CU)
    if (g.att_stage > 0 && si == g.att_stage) {
      stamp(60);                                                       // attention phase begins
      // ================= fused cross-attention of this panel (ChainArgs::att_stage) ========================================
      // wave = head (D = 256: four heads).  Per 32-key block, as attn3_kernel: S^T = K.Q^T (lane = query row: softmax statistics
      // in-lane + one half swap), logits in the log2 domain, online softmax over the blocks, O^T += V^T.P^T.  The context replaces
      // this head's queries in panel 1 (tiles 2 head, 2 head + 1: nobody else reads them) -- or goes to panel 2 when the
      // alignments are wanted, whose pass multiplies K and Q once more.  A flat 32-row panel can straddle two batch elements:
      // then the computation runs once per element and every lane keeps the pass of its own row.
      const int head = wave;
      const int H = g.D >> 6, TTk = (g.att_Tk + 31) >> 5;
      const int row = m0 + l31;
      int b_lo = m0 / g.att_Tq, b_hi = (m0 + 31 < mend ? m0 + 31 : mend - 1) / g.att_Tq;
      b_lo = __builtin_amdgcn_readfirstlane(b_lo); b_hi = __builtin_amdgcn_readfirstlane(b_hi);
      char* P1 = panel_ptr(1);
      char* Pc = panel_ptr(g.att_ali ? 2 : 1);
      float* xs = reinterpret_cast<float*>(smem + g.att_lds);        // transpose scratch of the alignment pass: [4 waves][32][33] floats
      const float c2 = (g.att_temp != 1.0f) ? 0.125f * 1.44269504088896340736f / g.att_temp : 0.125f * 1.44269504088896340736f;
      for (int bb = b_lo; bb <= b_hi; ++bb) {
        const int qlen = g.att_qlen ? g.att_qlen[bb] : g.att_Tq, klen = g.att_klen ? g.att_klen[bb] : g.att_Tk;
        const int tq = row - bb * g.att_Tq;
        const bool mine = row < mend && tq >= 0 && tq < g.att_Tq;
        const bool qvalid = mine && tq < qlen;
        f32x16 O[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < TTk; ++kb) {
          const char* kt = g.att_K + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
          const char* vt = g.att_V + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
          h16x8 khi[4], klo[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) { khi[t] = *reinterpret_cast<const h16x8*>(kt + 1024 * t); klo[t] = *reinterpret_cast<const h16x8*>(kt + 4096 + 1024 * t); }
          f32x16 sacc;
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const h16x8 qh = *reinterpret_cast<const h16x8*>(P1 + panel_off4(l31, 2 * head + (t >> 1), 2 * (t & 1) + half));
            const h16x8 ql = *reinterpret_cast<const h16x8*>(P1 + panel_off4(l31, 2 * head + (t >> 1), 4 + 2 * (t & 1) + half));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi[t], qh, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo[t], qh, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi[t], ql, sacc, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);               // (the K registers are dead: V lands in them)
          h16x8 vhi[2][2], vlo[2][2];
#pragma unroll
          for (int y = 0; y < 4; ++y) { vhi[y >> 1][y & 1] = *reinterpret_cast<const h16x8*>(vt + 1024 * y); vlo[y >> 1][y & 1] = *reinterpret_cast<const h16x8*>(vt + 4096 + 1024 * y); }
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
          if (kb > 0) {
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
            h16x8 phi, plo;
            float pv8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) pv8[e] = sacc[8 * tp + e];
            split_hi_lo(pv8, phi, plo);
            O[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][0], phi, O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][1], phi, O[1], 0, 0, 0);
            O[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo[tp][0], phi, O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo[tp][1], phi, O[1], 0, 0, 0);
            O[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][0], plo, O[0], 0, 0, 0);
            O[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi[tp][1], plo, O[1], 0, 0, 0);
          }
        }
        const float M_fin = fmaxf(m_run, -3.0e38f);
        const float linv = 1.0f / l_run;                                                   // softmax denominator, attention.py:242
        // the context: O^T layout -- lane = query l31, register r of block nb = channel 32 nb + (r & 3) + 8 (r >> 2) + 4 half
        if (mine) {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float xv[4] = {O[nb][4 * q] * linv, O[nb][4 * q + 1] * linv, O[nb][4 * q + 2] * linv, O[nb][4 * q + 3] * linv};
              panel_put4(Pc, l31, 2 * head + nb, 8 * q + 4 * half, xv);
            }
        }
        if (g.att_ali) {                                               // (workgroup-uniform)
          // alignments = softmax(logits) of this head's rows (attention.py:242-246): S^T of every block once more (K tile from L1 / L2,
          // Q still in panel 1), normalised with the final (max, 1 / sum), written as 128-byte row pieces after a wave-private
          // 32 x 32 transpose through LDS (lane-per-query registers would leave 32-byte pieces: a tenth of the store rate)
          float* tb = xs + wave * (32 * 33);
#pragma unroll 1
          for (int kb = 0; kb < TTk; ++kb) {
            const char* kt = g.att_K + ((size_t)(bb * H + head) * TTk + kb) * kAoiTile + lane_s * 16;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const h16x8 kh = *reinterpret_cast<const h16x8*>(kt + 1024 * t), kl = *reinterpret_cast<const h16x8*>(kt + 4096 + 1024 * t);
              const h16x8 qh = *reinterpret_cast<const h16x8*>(P1 + panel_off4(l31, 2 * head + (t >> 1), 2 * (t & 1) + half));
              const h16x8 ql = *reinterpret_cast<const h16x8*>(P1 + panel_off4(l31, 2 * head + (t >> 1), 4 + 2 * (t & 1) + half));
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
              tb[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * half] = __builtin_amdgcn_exp2f(sv - M_fin) * linv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int rr = 8 * it + (lane_s >> 3), k4 = (lane_s & 7) * 4;
              const float4 pv = make_float4(tb[rr * 33 + k4], tb[rr * 33 + k4 + 1], tb[rr * 33 + k4 + 2], tb[rr * 33 + k4 + 3]);
              const int rg = m0 + rr, tqr = rg - bb * g.att_Tq, key0 = 32 * kb + k4;
              if (rg < mend && tqr >= 0 && tqr < g.att_Tq && key0 < g.att_Tk)
                *reinterpret_cast<float4*>(g.att_ali + (((size_t)(bb * H + head) * g.att_Tq + tqr) * g.att_Tk + key0)) = pv;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
        }
      }
      lds_barrier4();                                                  // the context panel is complete for every wave
      stamp(61);                                                       // attention (and alignment) phase ends
    }
    const ChainStage st = g.st[si];                      // by value: one scalar burst from the kernarg segment per stage
    const bool wave_on = 64 * wave < st.n;               // this wave owns output columns 64w .. 64w+63 (column blocks 2w, 2w+1)
    const char* const Ap0 = panel_ptr(st.a0);
    const char* const Ap1 = panel_ptr(st.a1);
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // a stage that holds only V columns of a Q|K|V panel: its V-image units are 8 consecutive KEYS of one channel (common.h), i.e. the
    // transpose of what a lane holds; the 8-wave kernel multiplies such a stage with its operands swapped, here (one k-loop instance)
    // the epilogue turns each 32 x 32 block through LDS and stores the same coalesced 1 KiB pieces
    const bool vstage = st.out_fmt == 4 && st.out && st.aoi_c0 >= 2 * st.aoi_D && !((st.aoi_c0 - 2 * st.aoi_D) & 31) && !(st.aoi_T & 15) &&
                       st.acc_mode == 0 && st.dst < 0 && st.res < 0 && !st.gamma && !st.pe && st.act == ACT_IDENTITY;
    wstamp(si, 0);
    const int npad = (st.nk + kDepth - 1) / kDepth * kDepth;
    h16x8 afr[2][4];                                     // activation operands, ONE k-tile ahead: [set][2 t + (hi | lo)]
    // One trip = 8 slots.  Waves without columns and padding k-tiles multiply the zeros their out-of-range refills returned.
    // Consecutive MFMAs alternate between the two accumulators; the refill of a piece follows the last MFMA that read it.
    // With one wave per SIMD nothing hides the loop's scalar / vector bookkeeping (about five issue slots fit beside an MFMA), so the
    // A operands are addressed through eight per-trip row pointers (even / odd tile x four 16-byte chunks, the XOR swizzle folded in)
    // plus immediate offsets: no per-tile arithmetic at all.  ONE instance of the loop: a second one (a general form for irregular
    // stages was tried) makes every wreg element a phi of the two and the kernel spills ~500 registers.
    auto mfma_slot = [&](int u) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const h16x8 ah = afr[u & 1][2 * t], al = afr[u & 1][2 * t + 1];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][0][2 * t], ah, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][1][2 * t], ah, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][0][2 * t], al, acc[0], 0, 0, 0);
        piece(u, 0, 2 * t);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][1][2 * t], al, acc[1], 0, 0, 0);
        piece(u, 1, 2 * t);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][0][2 * t + 1], ah, acc[0], 0, 0, 0);
        piece(u, 0, 2 * t + 1);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[u][1][2 * t + 1], ah, acc[1], 0, 0, 0);
        piece(u, 1, 2 * t + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // launch_panel_chain admits only REGULAR programs to this kernel: nk = 4 (then the stage reads tiles 0-3 of its panel: K = 128, first
    // tile 0) or a multiple of 8 with an even first tile and the panel switch (asw) on a trip boundary.  Slot u of a trip then reads
    // tile t0 + u of ONE panel; for nk = 4 the slots 4-7 read tiles 4-7 of the same rows -- finite stale data (the panels are
    // zero-filled at kernel start) against the zeros of their out-of-range weight refills.
    {
      const int x15 = l31 & 15;
      int arow[2][4];                                    // LDS byte addresses of this lane's chunks in tile pair 0 of the trip's source
      auto trip_rows = [&](int kb) {
        const int kc = kb < st.nk ? kb : 0;
        const char* Ap = (kc < st.asw) ? Ap0 : Ap1;
        const int t0 = (kc < st.asw) ? kc + st.akt0 : kc - st.asw;
        const int base = (int)(size_t)(Ap - smem) + (t0 >> 1) * 256 + l31 * 1024;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int idx = 0; idx < 4; ++idx) {
            const int c = ((idx & 1) ? 4 : 0) + 2 * (idx >> 1) + half;
            arow[par][idx] = base + ((((par << 3) | c) ^ x15) << 4);
          }
      };
      auto read_r = [&](int u, int set) {                // k-tile u of the current trip rows
#pragma unroll
        for (int idx = 0; idx < 4; ++idx) afr[set][idx] = *reinterpret_cast<const h16x8*>(smem + arow[u & 1][idx] + (u >> 1) * 256);
      };
      sstamp(si, 32 + wave);
      trip_rows(0);
      read_r(0, 0);
#pragma unroll 1
      for (int kb = 0; kb < npad; kb += kDepth) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
          if (u + 1 < kDepth) read_r(u + 1, (u + 1) & 1);
          else { trip_rows(kb + kDepth); read_r(0, 0); }   // the next trip's first tile (after the last trip: a harmless re-read)
          mfma_slot(u);
          sstamp(si, 64 + wave * 8 + u);
        }
        advance();
      }
    }
    stamp(2 + 2 * si);
    wstamp(si, 1);
    // (overflow sentinel, common.h: range_note -- lane = row, and a row with a split operand out of the fp16 range is all NaN.  The probe
    //  sits in every epilogue path, on the first value the path derives from its accumulators anyway: read here, straight from the
    //  accumulator file, it cost the kernel its first spilled register.  TWO paths carry none -- the Q | K image tails and the plain
    //  Dense stages of a pre-chain: there the probe's branch spilled a register too.  What they would see is seen elsewhere: the tails
    //  multiply a LayerNorm output, |y| <= |gamma| sqrt(D) + |beta| whatever the input -- a property of the variables that
    //  vnr_finalize_weights checks (engine.hip: ln_static_bound) --; the pre-chain stages multiply z and the pre-projection's x, which
    //  leave as fp32 rows and are split AGAIN by the next launch's panel load in front of a probed stage (att_proj1 reads x and adds it
    //  as the residual), and a NaN they hand on reaches that probe as well)
    // ---- FFN second layer: accumulate over hidden chunks (modes 1, 2: no epilogue yet) ---------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (st.acc_mode == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accF[j][r] = acc[j][r];
      } else if (st.acc_mode >= 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accF[j][r] += acc[j][r];
      }
    }
    if (st.acc_mode == 1 || st.acc_mode == 2) { if (st.sync_after) lds_barrier4(); stamp(3 + 2 * si); continue; }
    // ---- fast path: a full-width stage h = act(x.W + b) -> other panel (FFN dense1 chunks, utils.py:49; the query projection) -----
    if (st.acc_mode == 0 && (st.act == ACT_RELU || st.act == ACT_IDENTITY) && st.n == 256 && !st.pe && st.res < 0 && !st.gamma && !st.out &&
        st.dst >= 0 && st.dst != st.a0 && !(st.asw < st.nk && st.dst == st.a1) && !(g.cpl_stage > 0 && si == g.cpl_stage)) {
      const float floor_v = st.act == ACT_RELU ? 0.f : -__builtin_inff();       // (identity: the query projection in front of the attention)
      const float* sp = prm + si * 256;
      char* Dp = panel_ptr(st.dst);
      // With ONE wave per SIMD nothing hides an LDS round trip: left to itself the compiler emits read bias -> wait -> 25 dependent VALU
      // on eight recycled registers -> two stores, eight times over (3.0 kcyc for this epilogue against 1.9 for the 8-wave kernel's).
      // So: all eight bias reads first, the arithmetic of both column blocks on their own registers, the sixteen stores last.
      float4 bi[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bi[j][q] = *reinterpret_cast<const float4*>(sp + 32 * (2 * wave + j) + 8 * q + 4 * half);
      __builtin_amdgcn_sched_barrier(0);
      h16x4 xh[2][4], xl[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bq[4] = {bi[j][q].x, bi[j][q].y, bi[j][q].z, bi[j][q].w};
          float x4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x4[e] = acc[j][4 * q + e] * st.scale + bq[e];
          if (j == 0 && q == 0) range_note(g.range_flag, x4[0]);          // (before the ReLU: fmaxf(NaN, 0) = 0 would heal the row)
#pragma unroll
          for (int e = 0; e < 4; ++e) x4[e] = fmaxf(x4[e], floor_v);
          split_hi_lo(x4, xh[j][q], xl[j][q]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kt = 2 * wave + j, p = 8 * q + 4 * half;            // this column block = k-tile kt of the destination
          *reinterpret_cast<h16x4*>(Dp + panel_off4(l31, kt, p >> 3) + (p & 4) * 2) = xh[j][q];
          *reinterpret_cast<h16x4*>(Dp + panel_off4(l31, kt, 4 + (p >> 3)) + (p & 4) * 2) = xl[j][q];
        }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    if (vstage) {
      if (wave_on) {
        // both 32 x 32 blocks of the wave at once (two wave-private [32 channels][33] float buffers): all writes, ONE wait, all reads
        const int Hh = st.aoi_D >> 6, TT = (st.aoi_T + 31) >> 5;
        float* tb = reinterpret_cast<float*>(smem + g.vt_lds) + wave * (2 * 32 * 33);
        const float* bp = prm + si * 256 + 64 * wave;                   // (zero padded when the stage has no bias)
        const bool j1 = 64 * wave + 32 < st.n;                          // (wave-uniform; block 0 exists: wave_on)
        float4 bi[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) bi[j][q] = *reinterpret_cast<const float4*>(bp + 32 * j + 8 * q + 4 * half);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float bq[4] = {bi[j][q].x, bi[j][q].y, bi[j][q].z, bi[j][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float tv = acc[j][4 * q + e] * st.scale + bq[e];
              if (j == 0 && q == 0 && e == 0) range_note(g.range_flag, tv);
              tb[j * (32 * 33) + (8 * q + 4 * half + e) * 33 + l31] = tv;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float x[2][2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int tp = 0; tp < 2; ++tp)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[j][tp][e] = tb[j * (32 * 33) + l31 * 33 + 16 * tp + (e & 3) + 4 * half + 8 * (e >> 2)];   // k-slot (tp, g = half, e) -> row of the panel
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (j == 1 && !j1) continue;
          const int cv = st.aoi_c0 - 2 * st.aoi_D + 32 * (2 * wave + j) + l31;   // V column of this lane: head cv >> 6, channel cv & 63
#pragma unroll
          for (int tp = 0; tp < 2; ++tp) {
            const int R = m0 + 16 * tp;
            if (R >= mend) continue;
            const int bb = R / st.aoi_T, tt = R - bb * st.aoi_T;
            h16x8 hi, lo;
            float x8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x8[e] = x[j][tp][e];
            split_hi_lo(x8, hi, lo);
            char* pdst = reinterpret_cast<char*>(st.out) + 2 * st.aoi_img_bytes + ((size_t)(bb * Hh + (cv >> 6)) * TT + (tt >> 5)) * kAoiTile +
                         ((tt >> 4) & 1) * 2048 + ((cv >> 5) & 1) * 1024 + ((half * 32 + l31) << 4);
            *reinterpret_cast<h16x8*>(pdst) = hi;
            *reinterpret_cast<h16x8*>(pdst + 4096) = lo;
          }
        }
      }
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    if (st.acc_mode == 3) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = accF[j][r];
    }

    // ---- fast path: a full-width stage whose output is a Q or K operand image (the Q | K | V tails of a block: attention.py:436-440) ------
    if (st.out && st.out_fmt != 0 && st.n == 256 && st.dst < 0 && st.res < 0 && !st.gamma && !st.pe && st.act == ACT_IDENTITY && st.acc_mode == 0 &&
        ((st.out_fmt == 1 ? 0 : st.aoi_c0) + 255) / (st.out_fmt == 1 ? st.n : st.aoi_D) < 2 && !(g.cpl_stage > 0 && si == g.cpl_stage)) {
      const float* sp = prm + si * 256;
      const int row = m0 + l31;
      const int Dd = st.out_fmt == 1 ? st.n : st.aoi_D, TT = (st.aoi_T + 31) >> 5;
      const int bb = row / st.aoi_T, tt = row - bb * st.aoi_T;
      const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(st.out, 0, 0x7ffffff0u, 0x00020000);
      float4 bi[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) bi[j][q] = *reinterpret_cast<const float4*>(sp + 32 * (2 * wave + j) + 8 * q + 4 * half);
      __builtin_amdgcn_sched_barrier(0);
      h16x4 xh[2][4], xl[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bq[4] = {bi[j][q].x, bi[j][q].y, bi[j][q].z, bi[j][q].w};
          float x4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x4[e] = acc[j][4 * q + e] * st.scale + bq[e];
          split_hi_lo(x4, xh[j][q], xl[j][q]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // this lane's 8-byte unit for channel d = (cw & 63) + 8q + 4 half of image `which`: t = d >> 4, g = q & 1 (common.h: AoiDesc)
        const int wc0 = (st.out_fmt == 1 ? 0 : st.aoi_c0) + 32 * (2 * wave + j);
        const int which = __builtin_amdgcn_readfirstlane(wc0 / Dd), cw = wc0 - which * Dd;
        const unsigned io = (unsigned)((long long)which * st.aoi_img_bytes + ((long long)(bb * (Dd >> 6) + (cw >> 6)) * TT + (tt >> 5)) * kAoiTile) +
                            (unsigned)(((cw & 63) >> 4) * 1024 + ((tt & 31) << 4) + half * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned o = row < mend ? io + (unsigned)((q >> 1) * 1024 + (q & 1) * 512) : kOob3;     // (rows beyond the panel: dropped by the hardware)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, xh[j][q]), rsI, o, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, xl[j][q]), rsI, o == kOob3 ? kOob3 : o + 4096u, 0, 0);
        }
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }

    // ---- fast path: a plain Dense stage of whole 32-column blocks -- y = x.W + b (+ pos_weight * PE[t]) -> HBM rows and / or another panel
    // (the folded ActNorm o InvertibleLinear and the pre_projection of a flow step: flow.py:149-166, transform.py:45-51) -----------------
    if (!st.gamma && st.res < 0 && st.act == ACT_IDENTITY && st.acc_mode == 0 && !(st.n & 31) && (st.out ? st.out_fmt == 0 : st.dst >= 0) &&
        !(st.dst >= 0 && (st.dst == st.a0 || (st.asw < st.nk && st.dst == st.a1))) && !(g.cpl_stage > 0 && si == g.cpl_stage) && wave_on) {
      const float* sp = prm + si * 256;
      const int row = m0 + l31;
      const bool rok = row < mend;
      const bool j1 = 64 * wave + 32 < st.n;                            // (wave-uniform; block 0 exists: wave_on)
      const float* prow = st.pe ? st.pe + (size_t)(row % st.pe_T) * st.n + 64 * wave + 4 * half : nullptr;
      float* orow = st.out ? st.out + (size_t)row * st.ldo + 64 * wave + 4 * half : nullptr;
      char* Dp = panel_ptr(st.dst >= 0 ? st.dst : 0);
      // one 32-column block at a time (both at once: 24 spilled registers): PE reads first -- the slowest of the stage --, then bias,
      // values, split, the eight panel stores, the four row stores
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j == 1 && !j1) continue;
        float4 p4[4];
        if (st.pe) {
#pragma unroll
          for (int q = 0; q < 4; ++q) p4[q] = rok ? *reinterpret_cast<const float4*>(prow + 32 * j + 8 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 bi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bi[q] = *reinterpret_cast<const float4*>(sp + 32 * (2 * wave + j) + 8 * q + 4 * half);
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[4 * q + 0] = acc[j][4 * q + 0] * st.scale + bi[q].x;
          v[4 * q + 1] = acc[j][4 * q + 1] * st.scale + bi[q].y;
          v[4 * q + 2] = acc[j][4 * q + 2] * st.scale + bi[q].z;
          v[4 * q + 3] = acc[j][4 * q + 3] * st.scale + bi[q].w;
        }
        if (st.pe) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { v[4 * q] += st.pe_w * p4[q].x; v[4 * q + 1] += st.pe_w * p4[q].y; v[4 * q + 2] += st.pe_w * p4[q].z; v[4 * q + 3] += st.pe_w * p4[q].w; }
        }
        if (st.dst >= 0) {
          h16x4 xh[4], xl[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x4[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            split_hi_lo(x4, xh[q], xl[q]);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int po = panel_off4(l31, 2 * wave + j, q) + 8 * half;
            *reinterpret_cast<h16x4*>(Dp + po) = xh[q];
            *reinterpret_cast<h16x4*>(Dp + (po ^ 64)) = xl[q];
          }
        }
        if (st.out && rok) {
#pragma unroll
          for (int q = 0; q < 4; ++q) out_store4(orow + 32 * j + 8 * q, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    // (same, for a wave without columns in this stage: nothing to do but the barrier)
    // ---- fast path: the full-width residual + LayerNorm stage (att_proj1 / att_proj2 / FFN dense2 of a block: attention.py:449-452,
    // utils.py:50-53), straight-line: every LDS read of a phase is issued before the arithmetic that needs it (see the hidden path) ------
    if (st.gamma && st.n == 256 && !st.pe && st.res >= 0 && !st.out_pre && !st.out_stats && st.out_fmt == 0 && st.act == ACT_IDENTITY &&
        st.dst >= 0 && !(g.cpl_stage > 0 && si == g.cpl_stage)) {
      const float* sp = prm + si * 256;
      const float* lnp = prm + st.lds_ln;                               // gamma [256] | beta [256]
      const char* Rp = panel_ptr(st.res);
      char* Dp = panel_ptr(st.dst);
      int po[2][4];                                                     // this lane's 8-byte piece of (block j, group q): hi chunk; lo = +4 chunks
      float4 bi[2][4];
      h16x4 rh[2][4], rl[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          po[j][q] = panel_off4(l31, 2 * wave + j, q) + 8 * half;
          bi[j][q] = *reinterpret_cast<const float4*>(sp + 32 * (2 * wave + j) + 8 * q + 4 * half);
          rh[j][q] = *reinterpret_cast<const h16x4*>(Rp + po[j][q]);
          rl[j][q] = *reinterpret_cast<const h16x4*>(Rp + (po[j][q] ^ 64));
        }
      __builtin_amdgcn_sched_barrier(0);
      float v[2][16];
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float bq[4] = {bi[j][q].x, bi[j][q].y, bi[j][q].z, bi[j][q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[j][4 * q + e] * st.scale + bq[e];
            x += (float)rh[j][q][e] + (float)rl[j][q][e];               // residual = hi + lo of the panel entry (22 bits)
            v[j][4 * q + e] = x;
            ps[e] += x;
          }
        }
      range_note(g.range_flag, v[0][0]);
      wstamp(si, 6);
      // gamma / beta do not depend on the statistics: their reads travel while the row sums are exchanged
      float4 ga[2][4], be[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * (2 * wave + j) + 8 * q + 4 * half;
          ga[j][q] = *reinterpret_cast<const float4*>(lnp + col);
          be[j][q] = *reinterpret_cast<const float4*>(lnp + 256 + col);
        }
      float s1 = (ps[0] + ps[1]) + (ps[2] + ps[3]);
      s1 += __shfl_xor(s1, 32, 64);
      const float mw = s1 * (1.f / 64.f);
      float pm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = v[j][r] - mw; pm[r & 3] += d * d; }
      float m2 = (pm[0] + pm[1]) + (pm[2] + pm[3]);
      m2 += __shfl_xor(m2, 32, 64);
      wstamp(si, 7);
      if (half == 0) { scratch[l31 * 4 + wave] = s1; scratch[kRows * 4 + l31 * 4 + wave] = m2; }
      lds_barrier4();                                                   // (also: every wave is done reading the source / residual panels)
      const f32x4 sa = *reinterpret_cast<const f32x4*>(scratch + l31 * 4);
      const f32x4 ma = *reinterpret_cast<const f32x4*>(scratch + kRows * 4 + l31 * 4);
      const float mu = (sa[0] + sa[1] + sa[2] + sa[3]) * (1.f / 256.f);
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < kNW; ++w) { const float dm = sa[w] * (1.f / 64.f) - mu; var += ma[w] + 64.f * dm * dm; }      // Chan et al.: exact merge
      const float rstd = 1.0f / sqrtf(var * (1.f / 256.f) + kLnEps);
      h16x4 xh[2][4], xl[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gq[4] = {ga[j][q].x, ga[j][q].y, ga[j][q].z, ga[j][q].w}, eq[4] = {be[j][q].x, be[j][q].y, be[j][q].z, be[j][q].w};
          float x4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { x4[e] = (v[j][4 * q + e] - mu) * rstd * gq[e] + eq[e]; v[j][4 * q + e] = x4[e]; }
          split_hi_lo(x4, xh[j][q], xl[j][q]);
        }
      wstamp(si, 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<h16x4*>(Dp + po[j][q]) = xh[j][q];
          *reinterpret_cast<h16x4*>(Dp + (po[j][q] ^ 64)) = xl[j][q];
        }
      if (st.out && m0 + l31 < mend) {
        float* orow = st.out + (size_t)(m0 + l31) * st.ldo + 64 * wave + 4 * half;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) out_store4(orow + 32 * j + 8 * q, v[j][4 * q], v[j][4 * q + 1], v[j][4 * q + 2], v[j][4 * q + 3]);
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }

    // ---- epilogue: v = act(acc*scale + bias) (+ residual panel) ; optional LayerNorm over the row ------------------------
    // lane (row l31, half) holds columns n = 32 (2 wave + j) + 8q + 4 half + e  (register 4q + e of block j)
    float v[2][16];
    const float* sp = prm + si * 256;
    bool cok[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * (2 * wave + j) + 8 * q + 4 * half;
        cok[j][q] = col < st.n;                                         // (n is a multiple of 4)
        const float4 bi = *reinterpret_cast<const float4*>(sp + col);
        v[j][4 * q + 0] = acc[j][4 * q + 0] * st.scale + bi.x;
        v[j][4 * q + 1] = acc[j][4 * q + 1] * st.scale + bi.y;
        v[j][4 * q + 2] = acc[j][4 * q + 2] * st.scale + bi.z;
        v[j][4 * q + 3] = acc[j][4 * q + 3] * st.scale + bi.w;
      }
    range_note(g.range_flag, v[0][0]);
    if (st.act == ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[j][r] = fmaxf(v[j][r], 0.f);
    } else if (st.act == ACT_TANH) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[j][r] = fast_tanhf(v[j][r]);
    }
    wstamp(si, 4);
    if (g.cpl_stage > 0 && si == g.cpl_stage) {
      // ================= fused affine coupling (ChainArgs::cpl_stage; arithmetic of misc.hip: coupling_fwd_kernel) ==============
      // v = [log_scale (hc columns) | shift (hc columns)], hc = 64: waves 0 and 1 park it in LDS ([32][2 hc + 4] floats); behind the
      // barrier all 256 threads share the elementwise work (thread -> row tid >> 4 of a 16-row half, 4-column piece tid & 15).
      const int cst = st.n + 4;
      float* cx = reinterpret_cast<float*>(smem + g.cpl_lds);
      if (wave_on) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (cok[j][q])
              *reinterpret_cast<float4*>(cx + l31 * cst + 32 * (2 * wave + j) + 8 * q + 4 * half) =
                  make_float4(v[j][4 * q], v[j][4 * q + 1], v[j][4 * q + 2], v[j][4 * q + 3]);
      }
      // the four pieces of z this thread works on are requested BEFORE the exchange barrier (they do not depend on the heads): one memory
      // round trip under the barrier instead of four behind it
      const int c = (tid_s & 15) * 4;
      float4 zo[2][2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const int row = m0 + 16 * rr + (tid_s >> 4);
          zo[rr][which] = row < mend ? *reinterpret_cast<const float4*>(g.cpl_z + (size_t)row * g.cpl_ld + (which ? g.cpl_cond_off : g.cpl_zp_off) + c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      lds_barrier4();
      {
        char* Dp = panel_ptr(st.dst);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int prow = 16 * rr + (tid_s >> 4), row = m0 + prow;
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            const int zoff = which ? g.cpl_cond_off : g.cpl_zp_off;
            float o[4] = {zo[rr][which].x, zo[rr][which].y, zo[rr][which].z, zo[rr][which].w};
            if (which == 0) {
              const float4 ls = *reinterpret_cast<const float4*>(cx + prow * cst + c), sh = *reinterpret_cast<const float4*>(cx + prow * cst + 64 + c);
              const float lv[4] = {ls.x, ls.y, ls.z, ls.w}, sv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float scale = 1.0f / (1.0f + expf(-(lv[e] + 2.0f)));                    // tf.math.sigmoid(log_scale + 2), flow.py:231
                o[e] = scale * o[e] + sv[e];                                                  // _affine, flow.py:216
              }
              if (row < mend) out_store4(g.cpl_z + (size_t)row * g.cpl_ld + zoff + c, o[0], o[1], o[2], o[3]);
            }
            panel_put4(Dp, prow, (zoff + c) >> 5, (zoff + c) & 31, o);
          }
        }
      }
      wstamp(si, 5);
      if (st.sync_after) lds_barrier4();
      stamp(3 + 2 * si); wstamp(si, 3);
      continue;
    }
    {
      const int row = m0 + l31;
      if (st.pe) {                                                      // + pos_weight * PE[t] (encoder.py:85, transform.py:51)
        const float* pr = st.pe + (size_t)(row % st.pe_T) * st.n;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * (2 * wave + j) + 8 * q + 4 * half;
            if (cok[j][q] && row < mend) {
              const float4 p4 = *reinterpret_cast<const float4*>(pr + col);
              v[j][4 * q] += st.pe_w * p4.x; v[j][4 * q + 1] += st.pe_w * p4.y; v[j][4 * q + 2] += st.pe_w * p4.z; v[j][4 * q + 3] += st.pe_w * p4.w;
            }
          }
      }
      if (st.res >= 0) {                                                // residual = hi + lo of the panel entry (22 bits)
        const char* Rp = panel_ptr(st.res);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kt = 2 * wave + j, p = 8 * q + 4 * half;
            const h16x4 rh = *reinterpret_cast<const h16x4*>(Rp + panel_off4(l31, kt, p >> 3) + (p & 4) * 2);
            const h16x4 rl = *reinterpret_cast<const h16x4*>(Rp + panel_off4(l31, kt, 4 + (p >> 3)) + (p & 4) * 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][4 * q + e] += (float)rh[e] + (float)rl[e];
          }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (!cok[j][q]) v[j][4 * q] = v[j][4 * q + 1] = v[j][4 * q + 2] = v[j][4 * q + 3] = 0.f;
    }
    wstamp(si, 6);                                       // (finer epilogue stamps, measurement) residual / PE added
    if (st.gamma && st.out_pre) {                                       // training: x + Dense(.) before the normalisation
      const int row = m0 + l31;
      if (row < mend) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (cok[j][q]) out_store4(st.out_pre + (size_t)row * st.ldo + 32 * (2 * wave + j) + 8 * q + 4 * half, v[j][4 * q], v[j][4 * q + 1], v[j][4 * q + 2], v[j][4 * q + 3]);
      }
    }
    if (st.gamma) {
      // LayerNormalization (eps 1e-3), ONE exchange: every wave reduces its own <= 64 columns of a row to (sum, M2 about its own
      // mean) and the four partials are merged exactly (Chan et al.): var.n = sum_w [M2_w + c_w (mean_w - mean)^2]
      int cw = st.n - 64 * wave; cw = cw < 0 ? 0 : (cw > 64 ? 64 : cw);          // valid columns of this wave
      const float rcw = cw > 0 ? 1.f / (float)cw : 0.f;
      float s1 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += v[j][r];
      s1 += __shfl_xor(s1, 32, 64);
      const float mw = s1 * rcw;
      float m2 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = cok[j][q] ? v[j][4 * q + e] - mw : 0.f;
            m2 += d * d;
          }
      m2 += __shfl_xor(m2, 32, 64);
      wstamp(si, 7);                                     // row statistics of this wave's columns ready
      if (half == 0) { scratch[l31 * 4 + wave] = s1; scratch[kRows * 4 + l31 * 4 + wave] = m2; }
      lds_barrier4();
      const float* lnp = prm + st.lds_ln;                               // gamma [256] | beta [256]
      const float rn = 1.f / (float)st.n;
      const f32x4 sa = *reinterpret_cast<const f32x4*>(scratch + l31 * 4);
      const f32x4 ma = *reinterpret_cast<const f32x4*>(scratch + kRows * 4 + l31 * 4);
      const float tot = sa[0] + sa[1] + sa[2] + sa[3];
      const float mu = tot * rn;
      float var = 0.f;
#pragma unroll
      for (int w = 0; w < kNW; ++w) {
        int c = st.n - 64 * w; c = c < 0 ? 0 : (c > 64 ? 64 : c);
        const float dm = c > 0 ? sa[w] / (float)c - mu : 0.f;
        var += ma[w] + (float)c * dm * dm;
      }
      const float rstd = 1.0f / sqrtf(var * rn + kLnEps);
      if (st.out_stats && wave == 0 && half == 0 && m0 + l31 < mend)
        *reinterpret_cast<float2*>(st.out_stats + 2 * (size_t)(m0 + l31)) = make_float2(mu, rstd);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = 32 * (2 * wave + j) + 8 * q + 4 * half;
          const float4 ga = *reinterpret_cast<const float4*>(lnp + col), be = *reinterpret_cast<const float4*>(lnp + 256 + col);
          v[j][4 * q + 0] = (v[j][4 * q + 0] - mu) * rstd * ga.x + be.x;
          v[j][4 * q + 1] = (v[j][4 * q + 1] - mu) * rstd * ga.y + be.y;
          v[j][4 * q + 2] = (v[j][4 * q + 2] - mu) * rstd * ga.z + be.z;
          v[j][4 * q + 3] = (v[j][4 * q + 3] - mu) * rstd * ga.w + be.w;
        }
    } else if (st.dst >= 0 && (st.dst == st.a0 || (st.asw < st.nk && st.dst == st.a1))) {
      lds_barrier4();                                                 // in-place stage: every wave is done reading the source panel
    }
    wstamp(si, 2);
    // ---- outputs: HBM (fp32, 16-byte row pieces) and/or destination panel (split fp16) -----------------------------------
    {
      const int row = m0 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = 32 * (2 * wave + j);                             // first column of this block
        if (cb >= st.n) continue;                                       // (wave-uniform)
        char* img_row = nullptr;                         // Q/K-type: address of this lane's 16-byte unit for t = 0, g = 0
        bool img_generic = false;
        if (st.out && st.out_fmt != 0 && row < mend) {
          const int Dd = st.out_fmt == 1 ? st.n : st.aoi_D, wc0 = (st.out_fmt == 1 ? 0 : st.aoi_c0) + cb;
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
          const int col = cb + 8 * q + 4 * half;
          if (!cok[j][q]) continue;
          if (st.out && row < mend) {
            if (img_row) {                                   // channel d = (cw & 63) + 8q + 4 half: t = d >> 4, g = q & 1
              h16x4 hi, lo;
              float x4[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) x4[e] = v[j][4 * q + e];
              split_hi_lo(x4, hi, lo);
              char* pd = img_row + (q >> 1) * 1024 + (q & 1) * 512;
              *reinterpret_cast<h16x4*>(pd) = hi;
              *reinterpret_cast<h16x4*>(pd + 4096) = lo;
            } else if (img_generic) {
              AoiDesc ad; ad.mode = 4; ad.D = st.aoi_D; ad.T = st.aoi_T; ad.TT = (st.aoi_T + 31) >> 5; ad.blk_bytes = st.aoi_img_bytes;
              ad.qk = reinterpret_cast<char*>(st.out); ad.vt = ad.qk + 2 * st.aoi_img_bytes;
              aoi_store4(ad, row, st.aoi_c0 + col, &v[j][4 * q]);
            }
            else out_store4(st.out + (size_t)row * st.ldo + col, v[j][4 * q], v[j][4 * q + 1], v[j][4 * q + 2], v[j][4 * q + 3]);
          }
          if (st.dst >= 0) panel_put4(panel_ptr(st.dst), l31, col >> 5, col & 31, &v[j][4 * q]);
        }
      }
    }
    wstamp(si, 5);
    if (st.sync_after) lds_barrier4();                                // panels are complete / free before the next stage
    stamp(3 + 2 * si);
    wstamp(si, 3);
  }
}

hipError_t launch_chain4(const ChainArgs& g, int lds, hipStream_t s) {
  static int attr_set[kMaxDevices] = {0};
  opt_in_dynamic_lds((const void*)panel_chain4_kernel, lds, attr_set);
  const int wgs = chain_workers(g, kRows);
  static const char* ts_path = getenv("VNR_CHAIN_TS");
  if (ts_path) {
    ChainArgs gg = g;
    static const char* ts_stage = getenv("VNR_CHAIN_TS_STAGE");       // stage whose per-wave stamps are taken (default 1)
    gg.dbg_stage = ts_stage ? atoi(ts_stage) : 1;
    if (getenv("VNR_CHAIN_TS_SLOTS")) gg.dbg_stage |= 0x100;
    const size_t n = (size_t)(wgs + gg.pf_wgs) * 128;       // (the prefetch workgroups stamp too)
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemsetAsync(d, 0, n * 8, s);
    gg.dbg_ts = d;
    vnr_launch(panel_chain4_kernel, dim3(wgs + gg.pf_wgs), dim3(256), lds, s, gg);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> hbuf(n);
    (void)hipMemcpy(hbuf.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FILE* f = fopen(ts_path, "ab");
    if (f) { int hdr[4] = {g.M, g.D | (g.att_stage > 0 ? (g.att_stage << 20) : 0) | (g.att_ali ? (1 << 16) : 0), g.nstages, (int)(n / 128)}; /* D <= 256: flags above bit 15 */ fwrite(hdr, 4, 4, f); fwrite(hbuf.data(), 8, n, f); fclose(f); }
    return hipGetLastError();
  }
  vnr_launch(panel_chain4_kernel, dim3(wgs + g.pf_wgs), dim3(256), lds, s, g);
  return hipGetLastError();
}

}  // namespace vnr
