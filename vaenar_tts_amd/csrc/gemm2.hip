// The tiled GEMM of the path: LDS-DMA (buffer_load ... lds) multi-stage ring, fp32 MFMA or the 3-term split-fp16 products.
//
// One kernel template covers every dense contraction that is not a row-panel chain (gemm3.hip):
//   tf.keras.layers.Dense                       (reference modules/attention.py:154-159,401,427,432,
//                                                utils.py:44-45, decoder.py:164,174,179, transform.py:12-17,36)
//   tf.concat([x, ctx], -1) -> Dense            (attention.py:410-412,440-449) as two K panels
//   tf.keras.layers.Conv1D(k, 'same') + act + BN (utils.py:76-85) as an implicit GEMM over taps
// with fused epilogues (bias / activation / folded BatchNorm / positional term / residual / LayerNorm for row panels / attention
// operand images / split rows).  (Round 1's register-staged first generation, gemm.hip, was retired in round 3: this kernel now
// also takes A panels that lie further apart than one 2 GiB buffer descriptor.)  The global->LDS traffic never passes through VGPRs and is
// issued NSTAGE-1 k-tiles ahead with counted s_waitcnt vmcnt(N) and a raw s_barrier, so a single wave per
// SIMD keeps the matrix pipe busy (the S1 GEMMs are mid-sized: 200..1800 workgroups, i.e. about one
// workgroup per CU, where the register-staged kernel exposes the L2 latency of every k-tile).
//
// LDS image: k-tile rows of 32 floats (128 B) UNPADDED -- one DMA wave-instruction writes 8 rows x 128 B
// lane-linearly -- with a 16-byte-chunk XOR swizzle applied on the SOURCE side (lane l fetches logical
// chunk (l&7) ^ ((row>>1)&7) of its row) and the same XOR on the ds_read_b128 side: every 16-lane read
// group then touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// Out-of-range rows / k / conv taps are mapped to a byte offset beyond the buffer descriptor's
// num_records, for which the hardware returns zeros.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace vnr {

namespace {

constexpr unsigned kOobOffset = 0x80000000u;   // beyond any descriptor we build (< 2 GiB)

__device__ __forceinline__ float act2(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_TANH) return fast_tanhf(v);
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// (x = hi + lo with hi = fp16(x), lo = fp16(x - hi), 22 significant bits carried by two fp16 values: common.h, vnr_split)

}  // namespace

// SPLIT: 0 = fp32 MFMA; 1 = split-fp16 weights, fp32 activations split in registers; 2 = split-fp16 weights AND activations
// already stored as split rows (GemmArgs::a_split): the k-loop is ds_read + MFMA only
// LW > 0 (round 3): LW extra LOADER waves issue every LDS-DMA piece of the ring; the WM x WN multiplier waves only read fragments
// and multiply.  One 1-KiB DMA piece costs the issuing wave 60..185 cycles (MI355X_MICROARCH.md, per-instruction constants) -- with
// 64x64 tiles every multiplier wave paid four of them per k-tile beside 192 cycles of MFMA, and with one wave per SIMD nothing ran
// meanwhile.  A loader wave shares its SIMD with one multiplier wave and takes those stalls (and the conv tap walker's address
// arithmetic) off the matrix pipe's critical path.
// (round 6) the loader-wave variants ask for FOUR waves per SIMD, i.e. two co-resident 8-wave workgroups per CU: with fp32 activations
// (SPLIT = 1) the kernel took 140 registers -- occupancy 3 -- so a launch of 384 or 512 workgroups (the encoder's Q|K|V and FFN dense1
// products) ran as two rounds of 9 us on 256 CUs although its 64 KB rings fit twice into a CU's LDS
template <int BM, int BN, int WM, int WN, int NSTAGE, int MODE, bool LN, int SPLIT, int LW = 0>
__global__ void __launch_bounds__((WM * WN + LW) * 64, (LW > 0 && NSTAGE <= 4) ? 4 : 1)
gemm2_kernel(const GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int NW = WM * WN;
  constexpr int DW = LW ? LW : NW;                                   // waves that issue the DMA
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int AQ = BM / 8 / DW, BQ = BN / 8 / DW, LPW = AQ + BQ;   // DMA instructions per issuing wave per stage
  static_assert(BM % (8 * DW) == 0 && BN % (8 * DW) == 0, "stage rows must split evenly over the issuing waves");
  static_assert(!LW || (SPLIT && !LN), "loader waves: split-fp16 path without the LayerNorm epilogue (it synchronises the whole workgroup)");
  static_assert(NSTAGE >= 3 && NSTAGE <= 6, "ring depth");
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // XCD-aware tile order (speed only), bijective
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  if (g.nslice && !(tiles_n & 7)) {
    // N-sliced raster (tiles_n % 8 == 0): workgroup ids go round-robin over the 8 XCDs, so XCD x takes the column tiles
    // [x c, (x + 1) c), c = tiles_n / 8, and walks them row tile by row tile -- its slice of the weight panel (c BN rows of K) stays in
    // ITS L2 for the whole launch and every activation row tile is fetched once per XCD.  The default raster gives an XCD whole row
    // tiles, i.e. the ENTIRE weight panel once per tiles_n workgroups: for the 14-block cross K | V projection (N = 7168 -> 12.6 MB
    // of split weights against 4 MB of L2) that was 407 MB of fetches for 19 MB of operands (profiles/r03_hbm_traffic_pmc.txt).
    const int c = tiles_n >> 3, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tm = idx / c;
    tn = xcd * c + (idx - tm * c);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long* ts = g.dbg_ts ? g.dbg_ts + (size_t)blockIdx.x * 8 : nullptr;
  auto stamp = [&](int i) { if (ts && tid == 0) ts[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  if (ts && tid == 0) { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); ts[6] = hwid; ts[7] = xcc; }
  const bool is_loader = LW && wave >= NW;
  const int dwave = LW ? (is_loader ? wave - NW : 0) : wave;       // index among the issuing waves
  const int wm = is_loader ? 0 : wave / WN, wn = is_loader ? 0 : wave - wm * WN;
  const int half = lane >> 5, l31 = lane & 31;

  // ---- buffer descriptors (wave-uniform: built from kernel arguments only) ---------------------------
  // ONE descriptor covers both A panels when their span fits the 2 GiB offset range (base = the lower of the two pointers):
  // the panel switch is then an offset, not a different SRD.
  const size_t a1_span = ((size_t)(g.M - 1) * g.lda1 + (MODE == 1 ? g.conv_C : g.K1)) * 4;
  const size_t a2_span = g.A2 ? ((size_t)(g.M - 1) * g.lda2 + (g.K - g.K1)) * 4 : 0;
  const char* abase = (g.A2 && (const char*)g.A2 < (const char*)g.A1) ? (const char*)g.A2 : (const char*)g.A1;
  const size_t delta1 = (size_t)((const char*)g.A1 - abase);
  const size_t delta2 = g.A2 ? (size_t)((const char*)g.A2 - abase) : 0;
  const size_t a_end = (delta1 + a1_span > delta2 + a2_span) ? delta1 + a1_span : delta2 + a2_span;
  // (two panels further apart than the 2 GiB offset range -- different arena chunks of a long run -- get a descriptor each; the
  //  select is wave-uniform scalar work per DMA instruction)
  const bool far = g.A2 && a_end >= ((size_t)1 << 31);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(far ? (const char*)g.A1 : abase), 0, (unsigned)(far ? a1_span : a_end), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)g.A2, 0, (unsigned)a2_span, 0x00020000);   // used when far
  const unsigned d1 = far ? 0u : (unsigned)delta1, d2 = far ? 0u : (unsigned)delta2;
  // SPLIT: the weight panel is the pre-split fp16 image [N][ceil(K/32)][hi x32 | lo x32] (128 bytes per k-tile,
  // zero padded), i.e. the same bytes-per-row geometry as fp32 with K rounded up to 32.
  const int b_row_bytes = SPLIT ? ((g.K + 31) >> 5) * 128 : g.ldw * 4;
  const unsigned b_bytes = SPLIT ? (unsigned)((size_t)g.N * b_row_bytes)
                                 : (unsigned)(((size_t)(g.N - 1) * g.ldw + g.K) * 4);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(SPLIT ? (void*)g.Wsplit : (void*)g.Wt, 0, b_bytes, 0x00020000);

  // ---- per-lane DMA source descriptions ------------------------------------------------------------------
  // instruction x of this wave covers stage rows 8*(wave + NW*x) .. +7; lane -> row (lane>>3), chunk (lane&7)
  unsigned a_off1[AQ], a_off2[AQ];   // plain: byte offset of (row, logical chunk) at k0 = 0, or OOB
  int a_c4[AQ];                      // 4 * logical chunk (k offset inside the tile)
  int cv_t[AQ], cv_row[AQ], cv_cc[AQ], cv_j[AQ];   // conv: time index, b*T, channel offset, tap
  bool a_ok[AQ];
#pragma unroll
  for (int x = 0; x < AQ; ++x) {
    const int r = 8 * (dwave + DW * x) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int m = m0 + r;
    a_ok[x] = m < g.M;
    a_c4[x] = 4 * c;
    if (MODE == 0) {
      a_off1[x] = a_ok[x] ? d1 + (unsigned)(((size_t)m * g.lda1 + 4 * c) * 4) : kOobOffset;
      a_off2[x] = (a_ok[x] && g.A2) ? d2 + (unsigned)(((size_t)m * g.lda2 + 4 * c) * 4) : kOobOffset;
    } else {
      const int mm = a_ok[x] ? m : 0;
      const int b = mm / g.conv_T;
      cv_t[x] = mm - b * g.conv_T;
      cv_row[x] = b * g.conv_T;
      int cc = 4 * c, j = 0;
      while (cc >= g.conv_C) { cc -= g.conv_C; ++j; }
      cv_cc[x] = cc; cv_j[x] = j;
      a_off1[x] = 0; a_off2[x] = 0;
    }
  }
  unsigned b_off[BQ];
  int b_c4[BQ];
#pragma unroll
  for (int x = 0; x < BQ; ++x) {
    const int r = 8 * (dwave + DW * x) + (lane >> 3);     // row inside the B tile
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int n = n0 + r;
    b_c4[x] = 4 * c;
    b_off[x] = (n < g.N) ? (unsigned)((size_t)n * b_row_bytes + 16 * c) : kOobOffset;
  }

  // DMA instruction d (0 <= d < LPW) of k-tile `kt` into ring slot `slot`; d < AQ are A rows, the rest B rows.
  // Tiles are issued strictly in order (the conv tap walker is incremental).  Splitting the tile's DMA into
  // single instructions lets the main loop drop one into each MFMA issue gap.
  auto issue_one = [&](int kt, int slot, int d) {
    const int k0 = kt * 32;
    char* sbase = smem + slot * STAGE_BYTES;
    if (d < AQ) {
      const int x = d;
      lds_ptr_t dst = (lds_ptr_t)(sbase + (dwave + DW * x) * 1024);
      if (MODE == 0) {
        const bool second = k0 >= g.K1;                  // tile-uniform: K1 % 32 == 0 (checked by the launcher)
        const int lim = second ? g.K : g.K1;
        const int soff = (second ? (k0 - g.K1) : k0) * 4;
        unsigned off = second ? a_off2[x] : a_off1[x];
        if (k0 + a_c4[x] >= lim) off = kOobOffset;
        if (far && second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, dst, 16, off, soff, 0, 0);     // (wave-uniform branch, never taken in the usual case)
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, off, soff, 0, 0);
      } else {
        const int tt = cv_t[x] + cv_j[x] - (g.taps >> 1);
        const bool ok = a_ok[x] && cv_j[x] < g.taps && tt >= 0 && tt < g.conv_T;
        const unsigned off = ok ? (unsigned)(((size_t)(cv_row[x] + tt) * g.lda1 + cv_cc[x]) * 4) : kOobOffset;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, off, 0, 0, 0);
        cv_cc[x] += 32;                                  // advance to the next k-tile
        while (cv_cc[x] >= g.conv_C) { cv_cc[x] -= g.conv_C; ++cv_j[x]; }
      }
    } else {
      const int x = d - AQ;
      unsigned off = b_off[x];
      if (!SPLIT && k0 + b_c4[x] >= g.K) off = kOobOffset;   // (the split image is zero padded to whole k-tiles)
      if (SPLIT && k0 >= g.K) off = kOobOffset;
      lds_ptr_t dst = (lds_ptr_t)(sbase + BM * 128 + (dwave + DW * x) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, off, k0 * 4, 0, 0);
    }
  };
  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int d = 0; d < LPW; ++d) issue_one(kt, slot, d);
  };

  // ---- LDS read offsets (bytes): row l31 of a 32-row block, logical chunk 2*c8 + half -----------------------
  const int swz = (l31 >> 1) & 7;
  int rd_off[4];
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) rd_off[c8] = l31 * 128 + (((2 * c8 + half) ^ swz) << 4);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- epilogue parameters are fetched NOW so that their latency hides under the main loop ------------------
  const bool vec_ok = !(g.N & 3) && !(g.ldc & 3) && (!g.residual || !(g.ldr & 3));
  // (big tiles -- four accumulator blocks per wave -- fetch them AFTER the loop instead: 96 more live registers through the loop
  // spilled, and their one exposed latency is small beside 40 k-tiles)
  constexpr bool PRM_LATE = MI * NI >= 4;
  float4 p_bias[NI][4], p_sc[NI][4], p_sh[NI][4];     // per-lane column groups: col = .. + j*32 + 8q + 4*half
  // (round 6) the loader-wave variants fetch the folded BatchNorm scale / shift AFTER the loop: 32 registers less through the k-loop,
  // which is what lets two of their 8-wave workgroups share a CU (4 waves per SIMD = 128 registers); only the three prenet
  // convolutions pay the one exposed latency
  constexpr bool AFFINE_LATE = LW > 0 && !LN;
  auto load_params = [&](bool want_bias, bool want_affine) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = (LN ? 0 : n0) + wn * TN + j * 32 + 8 * q + 4 * half;
        if (want_bias) p_bias[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (want_affine) { p_sc[j][q] = make_float4(1.f, 1.f, 1.f, 1.f); p_sh[j][q] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (vec_ok && col < g.N && !is_loader) {
          if (want_bias && g.bias) p_bias[j][q] = *reinterpret_cast<const float4*>(g.bias + col);
          if (want_affine) {
            if (LN) {
              p_sc[j][q] = *reinterpret_cast<const float4*>(g.ln_gamma + col);
              p_sh[j][q] = *reinterpret_cast<const float4*>(g.ln_beta + col);
            } else if (g.bn_scale) {
              p_sc[j][q] = *reinterpret_cast<const float4*>(g.bn_scale + col);
              p_sh[j][q] = *reinterpret_cast<const float4*>(g.bn_shift + col);
            }
          }
        }
      }
  };
  if (!PRM_LATE) load_params(true, !AFFINE_LATE);

  // Software pipeline over k-tiles (32 k each = four 8-k groups G0..G3):
  //   * the DMA of tile kt+NSTAGE-1 is issued in EVERY iteration, one instruction per MFMA issue gap of G0
  //     (tiles past the end resolve to out-of-range offsets = zero fill into a free slot), so the body is
  //     branch-free and the wait is a constant;
  //   * MFMA operand fragments are double-buffered one group ahead;
  //   * the "tile kt+1 has landed" wait + barrier sits between G2 and G3 of tile kt, so the first two
  //     fragment groups of tile kt+1 are fetched under G3 and the next iteration starts on the matrix pipe.
  // Slot reuse: tile kt+NSTAGE-1 overwrites the slot of tile kt-1, whose LDS reads every wave completed
  // (lgkmcnt(0)) before it passed the barrier of iteration kt-1.
  const int nk = (g.K + 31) >> 5;
  float a_inv = 1.f;                                  // inverse of the gradient pre-scale of A (split path, see below)
  // a tile made of V-type image columns only (and nothing but a scale in the epilogue) is computed un-transposed: see mfmaS
  bool vtile = false;
  if (SPLIT && !LN && g.aoi.mode >= 2 && !(g.aoi.T & 15) && !(g.aoi.D & 63) && !g.bias && g.act == ACT_IDENTITY && !g.bn_scale && !g.pe &&
      !g.residual && !(g.M & 15)) {
    if (g.aoi.mode == 2) vtile = true;
    else if (g.aoi.mode == 3) vtile = (n0 % (2 * g.aoi.D)) >= g.aoi.D && ((n0 + BN - 1) % (2 * g.aoi.D)) >= g.aoi.D && (n0 / (2 * g.aoi.D)) == ((n0 + BN - 1) / (2 * g.aoi.D));
    else if (g.aoi.mode == 4) vtile = n0 >= 2 * g.aoi.D;
    if (n0 + BN > g.N) vtile = false;                 // (ragged last tile: generic path)
  }
  if (!LW || is_loader) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s, s);
  }
  // LayerNorm row panels: the residual tile [32][N] is fetched NOW by LDS-DMA into a staging region behind the
  // ring (16-byte chunks XOR-swizzled by row so the row-per-lane ds_read_b128 of the epilogue is conflict-free);
  // it lands under the main loop.  The same region later holds the normalised rows, which leave as whole
  // contiguous rows (1 KiB per wave instruction) instead of 64 partial cache lines per instruction.
  constexpr int RES_INSTR = LN ? (32 * BN / 4 / 64) / NW : 0;      // residual DMA instructions per wave (BN-wide rows)
  const bool staged = LN && vec_ok && (g.N & 63) == 0;
  float* Rst = reinterpret_cast<float*>(smem + NSTAGE * STAGE_BYTES);
  if (LN && staged && g.residual) {
    const int cpr = g.N >> 2;                                     // 16-byte chunks per row
    const unsigned r_bytes = (unsigned)(((size_t)(g.M - 1) * g.ldr + g.N) * 4);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)g.residual, 0, r_bytes, 0x00020000);
#pragma unroll
    for (int x = 0; x < RES_INSTR; ++x) {
      const int gidx = (wave + NW * x) * 64 + lane;                // linear chunk index inside the [32][N] tile
      const int r = gidx / cpr, pc = gidx - r * cpr;
      const int c = pc ^ (r & 15);
      const unsigned off = (r < 32 && m0 + r < g.M) ? (unsigned)(((size_t)(m0 + r) * g.ldr + 4 * c) * 4) : kOobOffset;
      if ((wave + NW * x) * 64 < 32 * cpr)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (lds_ptr_t)((char*)Rst + (wave + NW * x) * 1024), 16, off, 0, 0, 0);
    }
  }
  if constexpr (SPLIT) {
    // ---- split-fp16 pipeline: every fp32 product a*w is evaluated as hi_a*hi_w + lo_a*hi_w + hi_a*lo_w on the
    // fp16 matrix pipe (v_mfma_f32_32x32x16_f16, fp32 accumulate); the dropped lo*lo term is 2^-22 relative.
    // A k-tile (32 k) = two k16 steps; lane-half g owns k = 16t + 8g .. +7 of step t:
    //   A (fp32 in LDS): 16-byte chunks 4t+2g, 4t+2g+1 -> split in registers
    //   B (pre-split)  : hi chunk 2t+g, lo chunk 4+2t+g
    // A may be a gradient (training step): its magnitude follows the loss scale and can sit far below the fp16 range, so it
    // is pre-scaled by the power of two that maps the launch-wide max |A| to ~2^10 and the result is scaled back exactly.
    float a_sc = 1.f;
    if (g.a_absmax) {
      unsigned bits = *g.a_absmax;
      if (g.a_absmax2) bits = max(bits, *g.a_absmax2);
      if (g.a_absmax3) bits = max(bits, *g.a_absmax3);
      const int e = (int)(bits >> 23) & 0xff;
      if (e > 0 && e < 255) {
        int sft = 10 - (e - 127);
        if (sft > 126) sft = 126; if (sft < -126) sft = -126;
        a_sc = __uint_as_float((unsigned)(sft + 127) << 23);
        a_inv = __uint_as_float((unsigned)(-sft + 127) << 23);
      }
    }
    int rdA[2][2], rdB[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      rdA[t][0] = l31 * 128 + (((4 * t + 2 * half) ^ swz) << 4);
      rdA[t][1] = l31 * 128 + (((4 * t + 2 * half + 1) ^ swz) << 4);
      rdB[t][0] = l31 * 128 + (((2 * t + half) ^ swz) << 4);
      rdB[t][1] = l31 * 128 + (((4 + 2 * t + half) ^ swz) << 4);
    }
    f16x8 ahi[2][MI], alo[2][MI], bhi[2][NI], blo[2][NI];
    auto fragS = [&](const char* As, const char* Bs, int t, int set) {
      if constexpr (SPLIT == 2) {                          // A rows arrive pre-split: same chunk geometry as the weight image
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ahi[set][i] = *reinterpret_cast<const f16x8*>(As + i * 4096 + rdB[t][0]);
          alo[set][i] = *reinterpret_cast<const f16x8*>(As + i * 4096 + rdB[t][1]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          bhi[set][j] = *reinterpret_cast<const f16x8*>(Bs + j * 4096 + rdB[t][0]);
          blo[set][j] = *reinterpret_cast<const f16x8*>(Bs + j * 4096 + rdB[t][1]);
        }
      } else {
      f32x4 x0[MI], x1[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        x0[i] = *reinterpret_cast<const f32x4*>(As + i * 4096 + rdA[t][0]);
        x1[i] = *reinterpret_cast<const f32x4*>(As + i * 4096 + rdA[t][1]);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        bhi[set][j] = *reinterpret_cast<const f16x8*>(Bs + j * 4096 + rdB[t][0]);
        blo[set][j] = *reinterpret_cast<const f16x8*>(Bs + j * 4096 + rdB[t][1]);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const f32x4 s0 = x0[i] * a_sc, s1 = x1[i] * a_sc;
        const vnr_f8 xs_ = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        vnr_split(xs_, ahi[set][i], alo[set][i]);
      }
      }
    };
    // VT (tile-uniform, chosen once per workgroup): the tile holds only V-type columns of an attention operand image -> the
    // product is formed UN-transposed (operands swapped: lane <-> output column, registers <-> rows in k-slot order), so that 8
    // consecutive accumulator registers are one 16-byte unit of the V image and the epilogue writes coalesced 1 KiB pieces
    // instead of sixteen 2-byte scatters per 4 values (the cross K|V panel GEMM spent most of its 81 us there).
    auto mfmaS = [&](auto vt_tag, int set, int kt_dma, int ns, bool with_dma) {
      constexpr bool VT = decltype(vt_tag)::value;
      int d = 0;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[set][i], bhi[set][j], acc[i][j], 0, 0, 0);   // D
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[set][j], ahi[set][i], acc[i][j], 0, 0, 0);      // D^T
          if (with_dma && d < LPW) { issue_one(kt_dma, ns, d); ++d; }
          if (VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[set][i], bhi[set][j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[set][j], alo[set][i], acc[i][j], 0, 0, 0);
          if (with_dma && d < LPW) { issue_one(kt_dma, ns, d); ++d; }
          if (VT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[set][i], blo[set][j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo[set][j], ahi[set][i], acc[i][j], 0, 0, 0);
          if (with_dma && d < LPW) { issue_one(kt_dma, ns, d); ++d; }
        }
      if (with_dma)
#pragma unroll
        for (; d < LPW; ++d) issue_one(kt_dma, ns, d);
    };
    stamp(1);
    if (LW && is_loader) {
      // ---- loader wave: the whole DMA stream of this workgroup's ring.  Same barrier sequence as the multipliers (one before the
      // loop, one per k-tile): at barrier kt the tile kt+1 has landed (counted vmcnt wait) and every multiplier has finished its
      // reads of tile kt-1 (lgkmcnt(0) on their side), whose slot the next iteration refills.
      wait_vmcnt<(NSTAGE - 2) * LPW>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        int ns = slot + NSTAGE - 1; if (ns >= NSTAGE) ns -= NSTAGE;
        int nx = slot + 1; if (nx >= NSTAGE) nx -= NSTAGE;
        issue(kt + NSTAGE - 1, ns);
        wait_vmcnt<(NSTAGE - 2) * LPW>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = nx;
      }
      wait_vmcnt<0>();                                  // the dummy tail tiles must not land in LDS after the workgroup is gone
      return;                                           // (the epilogue of a non-LayerNorm tile has no workgroup barrier)
    }
    wait_vmcnt<(NSTAGE - 2) * LPW>();                  // (with a staged residual in flight this also drains tile 1: conservative)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    fragS(smem + wm * TM * 128, smem + BM * 128 + wn * TN * 128, 0, 0);
    auto kloopS = [&](auto vt_tag) {
      int slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        const char* As = smem + slot * STAGE_BYTES + wm * TM * 128;
        const char* Bs = smem + slot * STAGE_BYTES + BM * 128 + wn * TN * 128;
        int ns = slot + NSTAGE - 1; if (ns >= NSTAGE) ns -= NSTAGE;
        int nx = slot + 1; if (nx >= NSTAGE) nx -= NSTAGE;
        __builtin_amdgcn_sched_barrier(0);
        fragS(As, Bs, 1, 1);                               // second k16 step of this tile
        mfmaS(vt_tag, 0, kt + NSTAGE - 1, ns, LW == 0);    // first step + DMA of tile kt+NSTAGE-1 (the loader waves' job when LW > 0)
        __builtin_amdgcn_sched_barrier(0);
        if (LW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTAGE - 2) * LPW) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        mfmaS(vt_tag, 1, 0, 0, false);
        fragS(smem + nx * STAGE_BYTES + wm * TM * 128, smem + nx * STAGE_BYTES + BM * 128 + wn * TN * 128, 0, 0);
        slot = nx;
      }
    };
    if (vtile) kloopS(std::true_type{}); else kloopS(std::false_type{});
  } else {
  f32x4 fa[2][MI], fb[2][NI];
    auto frag = [&](const char* As, const char* Bs, int c8, int set) {
  #pragma unroll
      for (int i = 0; i < MI; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(As + i * 4096 + rd_off[c8]);
  #pragma unroll
      for (int j = 0; j < NI; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(Bs + j * 4096 + rd_off[c8]);
    };
    auto mfma_group = [&](int set, int kt_dma, int ns, bool with_dma) {
      int d = 0;
  #pragma unroll
      for (int s = 0; s < 4; ++s)
  #pragma unroll
        for (int i = 0; i < MI; ++i)
  #pragma unroll
          for (int j = 0; j < NI; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][s], fa[set][i][s], acc[i][j], 0, 0, 0);   // D^T: lane <-> row m
            if (with_dma && d < LPW) { issue_one(kt_dma, ns, d); ++d; }
          }
      if (with_dma)
  #pragma unroll
        for (; d < LPW; ++d) issue_one(kt_dma, ns, d);       // more DMA instructions than MFMAs in one group
    };
    stamp(1);
    wait_vmcnt<(NSTAGE - 2) * LPW>();                   // tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const char* As = smem + wm * TM * 128;
      const char* Bs = smem + BM * 128 + wn * TN * 128;
      frag(As, Bs, 0, 0);
      frag(As, Bs, 1, 1);
    }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const char* As = smem + slot * STAGE_BYTES + wm * TM * 128;
      const char* Bs = smem + slot * STAGE_BYTES + BM * 128 + wn * TN * 128;
      int ns = slot + NSTAGE - 1; if (ns >= NSTAGE) ns -= NSTAGE;   // slot of tile kt-1 == slot of tile kt+NSTAGE-1
      int nx = slot + 1; if (nx >= NSTAGE) nx -= NSTAGE;             // slot of tile kt+1
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(0, kt + NSTAGE - 1, ns, true);          // G0 + DMA of tile kt+NSTAGE-1
      frag(As, Bs, 2, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(1, 0, 0, false);                        // G1
      frag(As, Bs, 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_group(0, 0, 0, false);                        // G2
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTAGE - 2) * LPW) : "memory");   // tile kt+1 landed; my reads of tile kt done
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        const char* An = smem + nx * STAGE_BYTES + wm * TM * 128;
        const char* Bn = smem + nx * STAGE_BYTES + BM * 128 + wn * TN * 128;
        frag(An, Bn, 0, 0);                              // first group of tile kt+1 (set 0 is free after G2)
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(1, 0, 0, false);                      // G3
        frag(An, Bn, 1, 1);
      }
      slot = nx;
    }
}
  if constexpr (SPLIT != 0) {
    // overflow sentinel (common.h: range_note): an A element that left the fp16 range made every accumulator of its row NaN
    if (g.range_flag) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float probe = acc[i][0][0];                   // D^T layout: this lane's row
        if (vtile) {                                  // un-transposed layout: the 16 registers are 16 rows of the block
#pragma unroll
          for (int r = 1; r < 16; ++r) probe += acc[i][0][r];
        }
        range_note(g.range_flag, probe);
      }
    }
  }
  if (PRM_LATE) load_params(true, true);
  else if (AFFINE_LATE) load_params(false, true);
  wait_vmcnt<0>();                                    // drain the dummy tail tiles before LDS is reused / exit

  // Result layout (operands swapped, D^T): lane (l31, half) owns output ROW m = .. + l31 and, per 32x32 block,
  // the 4-column groups n = .. + 8*q + 4*half + {0,1,2,3}, q = 0..3 (register 4q + e): 16-byte stores, and a
  // row's LayerNorm statistics are an in-lane sum plus one cross-half shuffle.
  stamp(3);
  const float ascale = SPLIT ? g.acc_scale * a_inv : 1.0f;
  if (!LN && vtile) {
    // accumulator layout D: lane (l31, half) <-> column n of the 32-column block, register r <-> row (r & 3) + 8 (r >> 2) + 4 half
    // of the 32-row block; registers 8 tp .. 8 tp + 7 are the eight k-slots of 16-key half tile tp (common.h, V tile)
    const int Hh = g.aoi.D >> 6;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * TN + j * 32 + l31;
        int blk = 0, cc = n;
        if (g.aoi.mode == 3) { blk = n / (2 * g.aoi.D); cc = n - blk * 2 * g.aoi.D - g.aoi.D; }
        else if (g.aoi.mode == 4) cc = n - 2 * g.aoi.D;
        const int head = cc >> 6, dch = cc & 63;
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
          const int R = m0 + wm * TM + i * 32 + 16 * tp;
          if (R >= g.M) continue;
          const int bb = R / g.aoi.T, tt = R - bb * g.aoi.T;
          f16x8 hi, lo;
          {
            vnr_f8 xs_;
#pragma unroll
            for (int e = 0; e < 8; ++e) xs_[e] = acc[i][j][8 * tp + e] * ascale;
            vnr_split(xs_, hi, lo);
          }
          char* pd = g.aoi.vt + (size_t)blk * g.aoi.blk_bytes + ((size_t)(bb * Hh + head) * g.aoi.TT + (tt >> 5)) * kAoiTile +
                     ((tt >> 4) & 1) * 2048 + ((dch >> 5) & 1) * 1024 + ((half * 32 + (dch & 31)) << 4);
          *reinterpret_cast<f16x8*>(pd) = hi;
          *reinterpret_cast<f16x8*>(pd + 4096) = lo;
        }
      }
  } else
  if (!LN) {
    if (vec_ok) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = m0 + wm * TM + i * 32 + l31;
        const bool rok = row < g.M;
        const float* perow = g.pe ? g.pe + (size_t)(row % g.pe_T) * g.N : nullptr;
        float4 res[NI][4], pev[NI][4];
        // batch every dependent load of this row block first ...
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * TN + j * 32 + 8 * q + 4 * half;
            res[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            pev[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok && col < g.N) {
              if (g.residual) res[j][q] = *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col);
              if (perow) pev[j][q] = *reinterpret_cast<const float4*>(perow + col);
            }
          }
        // ... then compute and store
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = n0 + wn * TN + j * 32 + 8 * q + 4 * half;
            if (!rok || col >= g.N) continue;
            const float4 bi = p_bias[j][q], sc = p_sc[j][q], sh = p_sh[j][q];
            float v[4] = {acc[i][j][4 * q] * ascale + bi.x, acc[i][j][4 * q + 1] * ascale + bi.y, acc[i][j][4 * q + 2] * ascale + bi.z, acc[i][j][4 * q + 3] * ascale + bi.w};
            if (g.bn_scale && g.bn_first) { v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act2(v[e], g.act);
            if (g.bn_scale && !g.bn_first) { v[0] = v[0] * sc.x + sh.x; v[1] = v[1] * sc.y + sh.y; v[2] = v[2] * sc.z + sh.z; v[3] = v[3] * sc.w + sh.w; }
            const float4 pe4 = pev[j][q], r4 = res[j][q];
            v[0] += g.pe_w * pe4.x + r4.x; v[1] += g.pe_w * pe4.y + r4.y; v[2] += g.pe_w * pe4.z + r4.z; v[3] += g.pe_w * pe4.w + r4.w;
            if (g.aoi.mode) aoi_store4(g.aoi, row, col, v);      // attention operand image instead of fp32
            else if (g.c_split) {                                // split rows: 4 x fp16 hi at its place in the 32-channel tile, lo 64 bytes on
              typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
              h4_t hi, lo;
              { const vnr_f4 xs_ = {v[0], v[1], v[2], v[3]}; vnr_split(xs_, hi, lo); }
              char* pc = reinterpret_cast<char*>(g.C) + (size_t)row * g.ldc * 4 + (col >> 5) * 128 + (col & 31) * 2;
              *reinterpret_cast<h4_t*>(pc) = hi;
              *reinterpret_cast<h4_t*>(pc + 64) = lo;
            }
            else out_store4(g.C + (size_t)row * g.ldc + col, v[0], v[1], v[2], v[3]);
          }
      }
    } else {   // generic (unaligned N / strides): scalar path
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = m0 + wm * TM + i * 32 + l31;
        if (row >= g.M) continue;
        const float* perow = g.pe ? g.pe + (size_t)(row % g.pe_T) * g.N : nullptr;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int c = n0 + wn * TN + j * 32 + 8 * q + 4 * half + e;
              if (c >= g.N) continue;
              float t = acc[i][j][4 * q + e] * ascale + (g.bias ? g.bias[c] : 0.f);
              if (g.bn_scale && g.bn_first) t = t * g.bn_scale[c] + g.bn_shift[c];
              t = act2(t, g.act);
              if (g.bn_scale && !g.bn_first) t = t * g.bn_scale[c] + g.bn_shift[c];
              if (perow) t += g.pe_w * perow[c];
              if (g.residual) t += g.residual[(size_t)row * g.ldr + c];
              g.C[(size_t)row * g.ldc + c] = t;
            }
      }
    }
  } else {
    // ---- row-panel epilogue: v = act(acc + bias) + residual ; LayerNorm over the full row (BN >= N) ----------
    static_assert(!LN || (WM == 1 && MI == 1), "LayerNorm epilogue: one 32-row panel per workgroup");
    __syncthreads();                                  // all waves are done with the ring -> reuse it
    float* red = reinterpret_cast<float*>(smem);      // [2][32][NW]
    const int cpr = g.N >> 2;
    const int row = m0 + l31;
    const bool rok = row < g.M;
    float v[NI][16];
    float s1 = 0.f;
    if (vec_ok) {
      float4 res[NI][4];
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn * TN + j * 32 + 8 * q + 4 * half;
          if (staged)      // swizzled LDS copy (DMA'd at kernel start; zeros for rows >= M)
            res[j][q] = (g.residual && col < g.N) ? *reinterpret_cast<const float4*>(Rst + (size_t)l31 * g.N + 4 * ((col >> 2) ^ (l31 & 15)))
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
          else
            res[j][q] = (rok && g.residual && col < g.N) ? *reinterpret_cast<const float4*>(g.residual + (size_t)row * g.ldr + col)
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn * TN + j * 32 + 8 * q + 4 * half;
          const bool ok = rok && col < g.N;
          const float4 bi = p_bias[j][q], r4 = res[j][q];
          v[j][4 * q + 0] = ok ? act2(acc[0][j][4 * q + 0] * ascale + bi.x, g.act) + r4.x : 0.f;
          v[j][4 * q + 1] = ok ? act2(acc[0][j][4 * q + 1] * ascale + bi.y, g.act) + r4.y : 0.f;
          v[j][4 * q + 2] = ok ? act2(acc[0][j][4 * q + 2] * ascale + bi.z, g.act) + r4.z : 0.f;
          v[j][4 * q + 3] = ok ? act2(acc[0][j][4 * q + 3] * ascale + bi.w, g.act) + r4.w : 0.f;
          s1 += (v[j][4 * q] + v[j][4 * q + 1]) + (v[j][4 * q + 2] + v[j][4 * q + 3]);
        }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int col = wn * TN + j * 32 + 8 * q + 4 * half + e;
            float t = 0.f;
            if (rok && col < g.N) {
              t = act2(acc[0][j][4 * q + e] * ascale + (g.bias ? g.bias[col] : 0.f), g.act);
              if (g.residual) t += g.residual[(size_t)row * g.ldr + col];
            }
            v[j][4 * q + e] = t;
            s1 += t;
          }
    }
    s1 += __shfl_xor(s1, 32, 64);
    if (half == 0) red[l31 * NW + wave] = s1;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mean += red[l31 * NW + w];
    mean *= 1.f / (float)g.N;
    float s2 = 0.f;                                   // centred (population) variance, as tf.nn.moments
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = wn * TN + j * 32 + 8 * q + 4 * half + e;
          const float d = (col < g.N) ? v[j][4 * q + e] - mean : 0.f;
          s2 += d * d;
        }
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) red[32 * NW + l31 * NW + wave] = s2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) var += red[32 * NW + l31 * NW + w];
    const float rstd = 1.0f / sqrtf(var * (1.f / (float)g.N) + kLnEps);
    if (staged) {
      // normalised rows go back to the staging tile (same swizzle), then leave as whole rows
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn * TN + j * 32 + 8 * q + 4 * half;
          if (col >= g.N) continue;
          const float4 ga = p_sc[j][q], be = p_sh[j][q];
          *reinterpret_cast<float4*>(Rst + (size_t)l31 * g.N + 4 * ((col >> 2) ^ (l31 & 15))) =
              make_float4((v[j][4 * q + 0] - mean) * rstd * ga.x + be.x, (v[j][4 * q + 1] - mean) * rstd * ga.y + be.y,
                          (v[j][4 * q + 2] - mean) * rstd * ga.z + be.z, (v[j][4 * q + 3] - mean) * rstd * ga.w + be.w);
        }
      __syncthreads();
#pragma unroll
      for (int x = 0; x < RES_INSTR; ++x) {
        const int gidx = (wave + NW * x) * 64 + lane;
        const int r = gidx / cpr, c = gidx - r * cpr;               // logical chunk c of row r
        if (r < 32 && m0 + r < g.M) {
          const float4 o4 = *reinterpret_cast<const float4*>(Rst + (size_t)r * g.N + 4 * (c ^ (r & 15)));
          out_store4(g.C + (size_t)(m0 + r) * g.ldc + 4 * c, o4.x, o4.y, o4.z, o4.w);
        }
      }
    } else if (rok) {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = wn * TN + j * 32 + 8 * q + 4 * half;
          if (col >= g.N) continue;
          if (vec_ok) {
            const float4 ga = p_sc[j][q], be = p_sh[j][q];
            out_store4(g.C + (size_t)row * g.ldc + col,
                      (v[j][4 * q + 0] - mean) * rstd * ga.x + be.x, (v[j][4 * q + 1] - mean) * rstd * ga.y + be.y,
                      (v[j][4 * q + 2] - mean) * rstd * ga.z + be.z, (v[j][4 * q + 3] - mean) * rstd * ga.w + be.w);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.N)
                g.C[(size_t)row * g.ldc + col + e] = (v[j][4 * q + e] - mean) * rstd * g.ln_gamma[col + e] + g.ln_beta[col + e];
          }
        }
    }
  }
  stamp(4);                                           // stores issued
  if (ts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(5); }   // stores complete
}

// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NSTAGE, bool LN, int SPLIT = 0, int LW = 0>
static hipError_t launch2(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = LN ? 1 : (g.N + BN - 1) / BN;
  const size_t lds = (size_t)NSTAGE * (BM + BN) * 128 + (LN ? (size_t)32 * BN * 4 : 0);
  const dim3 grid(tiles_m * tiles_n), block((WM * WN + LW) * 64);
  static const char* ts_path = getenv("VNR_GEMM_TS");
  if (ts_path) {   // measurement only: synchronous launch + dump of the per-workgroup stamps
    GemmArgs gg = g;
    const size_t n = (size_t)grid.x * 8;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemset(d, 0, n * 8);
    gg.dbg_ts = d;
    const int mode = g.taps > 0 ? 1 : 0;
    if (mode) { auto k = gemm2_kernel<BM, BN, WM, WN, NSTAGE, 1, false, SPLIT, LW>;
      if (lds > 48 * 1024) { static int done[kMaxDevices] = {0}; opt_in_dynamic_lds((const void*)k, (int)lds, done); }
      vnr_launch(k, grid, block, lds, s, gg, tiles_m, tiles_n); }
    else { auto k = gemm2_kernel<BM, BN, WM, WN, NSTAGE, 0, LN, SPLIT, LW>;
      if (lds > 48 * 1024) { static int done[kMaxDevices] = {0}; opt_in_dynamic_lds((const void*)k, (int)lds, done); }
      vnr_launch(k, grid, block, lds, s, gg, tiles_m, tiles_n); }
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    FILE* f = fopen(ts_path, "ab");
    if (f) { int hdr[8] = {g.M, g.N, g.K, BM, BN, NSTAGE, (int)grid.x, LN ? 1 : 0}; fwrite(hdr, 4, 8, f); fwrite(h.data(), 8, n, f); fclose(f); }
    return hipGetLastError();
  }
  if (g.taps > 0) {
    if (LN) return hipErrorInvalidValue;
    auto k = gemm2_kernel<BM, BN, WM, WN, NSTAGE, 1, false, SPLIT, LW>;
    if (lds > 48 * 1024) { static int done[kMaxDevices] = {0}; opt_in_dynamic_lds((const void*)k, (int)lds, done); }
    vnr_launch(k, grid, block, lds, s, g, tiles_m, tiles_n);
  } else {
    auto k = gemm2_kernel<BM, BN, WM, WN, NSTAGE, 0, LN, SPLIT, LW>;
    if (lds > 48 * 1024) { static int done[kMaxDevices] = {0}; opt_in_dynamic_lds((const void*)k, (int)lds, done); }
    vnr_launch(k, grid, block, lds, s, g, tiles_m, tiles_n);
  }
  return hipGetLastError();
}

// true when the kernel can take this problem (operands inside the 2 GiB offset range of a buffer descriptor, epilogue combinations)
bool gemm2_supported(const GemmArgs& g) {
  if (g.A2 && (g.K1 & 31)) return false;                                   // panel switch must be tile-uniform
  const size_t lim = (size_t)1 << 31;
  if (((size_t)g.M * g.lda1 + g.K) * 4 >= lim) return false;
  if (g.A2 && ((size_t)g.M * g.lda2 + g.K) * 4 >= lim) return false;       // (each panel behind its own descriptor when they lie far apart)
  if (((size_t)g.N * g.ldw + g.K) * 4 >= lim) return false;
  if (g.ln_gamma && (g.N > 256 || g.taps > 0 || g.bn_scale || g.pe)) return false;
  if (g.aoi.mode && (g.ln_gamma || (g.N & 3) || (g.ldc & 3) || (g.residual && (g.ldr & 3)))) return false;   // image stores: 4-column groups
  if ((g.a_split || g.c_split) && (!g.Wsplit || g.ln_gamma || g.aoi.mode || (g.c_split && ((g.N & 31) || (g.ldc & 31))))) return false;
  return true;
}

hipError_t launch_gemm2(const GemmArgs& g_in, hipStream_t s) {
  static const int force_tile = getenv("VNR_GEMM_TILE") ? atoi(getenv("VNR_GEMM_TILE")) : -1;
  GemmArgs g = g_in;
  {
    // N-sliced raster (see the kernel) when the split weight panel is larger than the activation panel and than half an XCD's L2, and
    // the 64-column tiles divide evenly over the 8 XCDs.  VNR_GEMM_NSLICE=0 restores the row-major raster everywhere (A/B switch).
    static const int ns_on = getenv("VNR_GEMM_NSLICE") ? atoi(getenv("VNR_GEMM_NSLICE")) : 1;
    const size_t wbytes = (size_t)g.N * g.K * 4, abytes = (size_t)g.M * g.K * 4;
    const bool wide = (g.taps > 0 && g.M >= 8192 && g.N >= 128) || (g.wide_tiles && g.N >= 128);       // (the launches that take 128-column tiles)
    const int tn64 = (g.N + 63) / 64;
    g.nslice = (ns_on && g.Wsplit && !g.ln_gamma && !wide && !(tn64 & 7) && wbytes > abytes && wbytes > ((size_t)2 << 20)) ? 1 : 0;
  }
  static const int st = getenv("VNR_GEMM_STAGES") ? atoi(getenv("VNR_GEMM_STAGES")) : 0;   // measurement knob
  if (g.Wsplit) {   // split-fp16 variant (engine decides per call; weights were pre-split at finalize)
    if (g.ln_gamma) return g.N <= 128 ? launch2<32, 128, 1, 2, 3, true, 1>(g, s) : launch2<32, 256, 1, 4, 3, true, 1>(g, s);
    static const int stile = getenv("VNR_SPLIT_TILE") ? atoi(getenv("VNR_SPLIT_TILE")) : -1;   // measurement knob
    static const int sstages = getenv("VNR_SPLIT_STAGES") ? atoi(getenv("VNR_SPLIT_STAGES")) : -1;   // measurement knob
    int t = stile, ns = sstages;
    // 64x64 tiles (3 workgroups per CU) are fastest or tied on every S1 shape except the long-K PostNet convolutions
    // (M = 12800, K = 5*256): there the 64x128 tile halves the activation re-reads (53 -> 40 us per layer, tools/sweep_split_tiles.sh).
    // With several batches in flight (wide_tiles) the 64x128 tile wins everywhere: +4 % aggregate throughput (tools/ab_tiles.sh)
    if (t < 0) t = ((g.taps > 0 && g.M >= 8192 && g.N >= 128) || (g.wide_tiles && g.N >= 128)) ? 1 : 2;
    if ((t == 3 || t == 4) && (g.M < 8192 || g.N < 128)) t = (g.taps > 0 && g.M >= 8192 && g.N >= 128) ? 1 : 2;   // (the 8-wave 128x128 experiment: large-M launches only)
    // 64 x 256 tiles (8 waves of 32 x 64; round 4 experiment, VNR_SPLIT_TILE=5 / VNR_GEMM_ROWTILE=1): one workgroup per 64 rows takes ALL 256
    // columns -- the PostNet convolutions (M = 12800, N = 256) become 200 workgroups on 256 CUs in ONE round instead of 400 in 1.6
    static const int rowtile = getenv("VNR_GEMM_ROWTILE") ? atoi(getenv("VNR_GEMM_ROWTILE")) : 0;
    if ((t == 5 || rowtile) && g.taps > 0 && g.M >= 8192 && g.N == 256 && !g.wide_tiles) {
      if (g.a_split) { if (!((g.K & 31) || (g.K1 & 31) || (g.conv_C & 31) || g.a_absmax)) return launch2<64, 256, 2, 4, 3, false, 2>(g, s); }
      else return launch2<64, 256, 2, 4, 3, false, 1>(g, s);
    }
    if (t == 5) t = 1;
    // Ring depth by how many workgroups share a CU.  A k-tile of a 64x64 workgroup is 16 KB and arrives ~2 kcyc after its DMA was
    // issued: with three stages (two tiles in flight) ONE workgroup per CU streams 32 KB per latency -- a quarter of what the
    // CU's vector-memory path delivers (64 B/clk) -- so its k-loop runs at the memory latency, not at any bandwidth.  Co-resident
    // workgroups hide that for each other (3 x 48 KB of LDS), a launch of <= 256 workgroups has nobody to hide behind: it gets
    // six stages (96 KB), two per CU get four (2 x 64 KB).  VNR_SPLIT_STAGES pins the depth for A/B runs.
    static const int ncu = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    if (ns < 0 && g.wide_tiles) ns = 3;                   // several batches in flight (bench.py --streams / batches_in_flight): co-resident workgroups of the
                                                          // other batches hide the latency; deeper rings and loader waves only cost residency there
    if (ns < 0) {
      const int bm = t == 0 ? 128 : 64, bn = t == 2 ? 64 : 128;
      const long wgs = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
      ns = (t == 2) ? (wgs <= ncu ? 6 : wgs <= 2 * ncu ? 4 : 3) : (t == 1 && wgs <= ncu) ? 4 : 3;
    }
    // Loader-wave variant (LW = 4, see the kernel) of the 64x64 tile: only for launches of at most two workgroups per CU.  Measured
    // (profiles/r03_experiments.txt): the k-loops of the encoder's M = 2048 GEMMs run 10-25 % shorter (conv 58 -> 51 kcyc, N = 1024
    // 17.9 -> 12.9), but with three or more co-resident 4-wave workgroups per CU -- every training GEMM, the cross K|V panel -- the
    // neighbours already hide each other's DMA stalls and the 8-wave workgroups only lower the residency (T1 step 33.8 -> 35.6 ms);
    // 128x128 tiles with loader waves reach 61 % MFMA issue in the loop but four accumulator blocks per wave make the epilogue
    // (tanh + BatchNorm + split stores by 4 of 8 waves) as long as the loop saved.  VNR_GEMM_LW=0 restores the round-2 kernels.
    static const int lw_on = getenv("VNR_GEMM_LW") ? atoi(getenv("VNR_GEMM_LW")) : 1;
    if (lw_on && !g.no_loader_waves && !g.wide_tiles && t == 2 && stile < 0 && (!g.a_split || (!(g.K & 31) && !(g.K1 & 31) && !(g.taps > 0 && (g.conv_C & 31)) && !g.a_absmax))) {
      const long w64 = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
      if (w64 <= 2 * ncu) {
        const int ns64 = sstages > 0 ? sstages : (w64 <= ncu ? 6 : 4);
        if (g.a_split) return ns64 >= 6 ? launch2<64, 64, 2, 2, 6, false, 2, 4>(g, s) : launch2<64, 64, 2, 2, 4, false, 2, 4>(g, s);
        return ns64 >= 6 ? launch2<64, 64, 2, 2, 6, false, 1, 4>(g, s) : launch2<64, 64, 2, 2, 4, false, 1, 4>(g, s);
      }
    }
    if (g.a_split && t != 3 && t != 4) {                                   // activations arrive as split rows (64x64 / 64x128 tiles)
      if ((g.K & 31) || (g.K1 & 31) || (g.taps > 0 && (g.conv_C & 31)) || g.a_absmax) return hipErrorInvalidValue;
      if (t == 1) return ns >= 4 ? launch2<64, 128, 2, 2, 4, false, 2>(g, s) : launch2<64, 128, 2, 2, 3, false, 2>(g, s);
      return ns >= 6 ? launch2<64, 64, 2, 2, 6, false, 2>(g, s) : ns >= 4 ? launch2<64, 64, 2, 2, 4, false, 2>(g, s) : launch2<64, 64, 2, 2, 3, false, 2>(g, s);
    }
    if (t == 3) return g.a_split ? launch2<128, 128, 2, 4, 3, false, 2>(g, s) : launch2<128, 128, 2, 4, 3, false, 1>(g, s);   // experiment: 128x128, 8 waves of 64x32
    if (t == 4) return g.a_split ? launch2<128, 128, 4, 2, 3, false, 2>(g, s) : launch2<128, 128, 4, 2, 3, false, 1>(g, s);   // experiment: 128x128, 8 waves of 32x64
    if (t == 0) return ns >= 5 ? launch2<128, 128, 2, 2, 5, false, 1>(g, s) : ns == 4 ? launch2<128, 128, 2, 2, 4, false, 1>(g, s) : launch2<128, 128, 2, 2, 3, false, 1>(g, s);
    if (t == 1) return ns >= 4 ? launch2<64, 128, 2, 2, 4, false, 1>(g, s) : launch2<64, 128, 2, 2, 3, false, 1>(g, s);
    return ns >= 6 ? launch2<64, 64, 2, 2, 6, false, 1>(g, s) : ns >= 4 ? launch2<64, 64, 2, 2, 4, false, 1>(g, s) : launch2<64, 64, 2, 2, 3, false, 1>(g, s);
  }
  if (g.ln_gamma) {
    (void)st;
    if (g.N <= 128) return launch2<32, 128, 1, 2, 3, true>(g, s);
    return launch2<32, 256, 1, 4, 3, true>(g, s);
  }
  // 64x64 tiles (3 workgroups per CU) measured fastest or tied on every S1 shape (profiles/r01_*); the larger
  // tiles stay selectable for experiments (VNR_GEMM_TILE = 0: 128x128, 1: 64x128).
  int best = 2;
  if (force_tile >= 0) best = force_tile;
  switch (best) {
    case 0: return launch2<128, 128, 2, 2, 3, false>(g, s);
    case 1: return launch2<64, 128, 2, 2, 3, false>(g, s);
    default: return st == 4 ? launch2<64, 64, 2, 2, 4, false>(g, s) : launch2<64, 64, 2, 2, 3, false>(g, s);
  }
}

// entry point of every tiled GEMM / convolution of the engine (common.h)
hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipErrorInvalidValue;
  if ((g.K & 3) || (g.K1 & 3) || (g.lda1 & 3) || (g.A2 && (g.lda2 & 3)) || (g.ldw & 3)) return hipErrorInvalidValue;
  if (g.taps > 0 && ((g.conv_C & 3) || g.K != g.taps * g.conv_C)) return hipErrorInvalidValue;
  if (!gemm2_supported(g)) return hipErrorInvalidValue;
  static const char* log_path = getenv("VNR_GEMM_LOG");           // measurement only: one line per launch (shape and variant)
  if (log_path) {
    if (FILE* f = fopen(log_path, "a")) {
      fprintf(f, "M %d N %d K %d taps %d split %d a_split %d c_split %d grad %d act %d ln %d res %d A2 %d\n", g.M, g.N, g.K, g.taps, g.Wsplit ? 1 : 0, g.a_split,
              g.c_split, g.a_absmax ? 1 : 0, g.act, g.ln_gamma ? 1 : 0, g.residual ? 1 : 0, g.A2 ? 1 : 0);
      fclose(f);
    }
  }
  return launch_gemm2(g, s);
}

}  // namespace vnr
