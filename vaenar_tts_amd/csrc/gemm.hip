// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32 fma chains at the
// 157 TFLOP/s matrix rate).  One kernel template covers every dense contraction of the
// VAENAR-TTS path:
//   tf.keras.layers.Dense                       (reference modules/attention.py:154-159,401,427,432,
//                                                utils.py:44-45, decoder.py:164,174,179, transform.py:12-17,36)
//   tf.concat([x, ctx], -1) -> Dense            (attention.py:410-412,440-449) as two K panels
//   tf.keras.layers.Conv1D(k, 'same') + act + BN (utils.py:76-85) as an implicit GEMM over taps
//   tf.keras.layers.Embedding -> Conv1D          (encoder.py:81) as a row gather in the A loader
// Layout: activations [M][K] row-major; weights pre-transposed to Wt[N][K] so both MFMA operands
// are read from LDS as k-contiguous 16-byte vectors.  K is consumed in a permuted order
// (lane-half h owns k = 8c+4h+s) which is legal because both operands use the same permutation.
#include "common.h"
#include <stdlib.h>

namespace vnr {

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;   // 36 floats: 16-B aligned rows, conflict-free ds_read_b128

template <int MODE>
struct ARow {   // per-thread description of one A row handled by this thread
  const float* p1;   // plain: A1 + row*lda1 ; conv: unused
  const float* p2;   // plain: A2 + m*lda2
  int t;             // conv: time index of the row inside its sequence
  int rowbase;       // conv: b*T
  bool valid;
};

template <int MODE>
__device__ __forceinline__ float4 load_a4(const GemmArgs& g, const ARow<MODE>& r, int k) {
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!r.valid || k >= g.K) return z;
  if (MODE == 0) {
    if (k < g.K1) return *reinterpret_cast<const float4*>(r.p1 + k);
    return *reinterpret_cast<const float4*>(r.p2 + (k - g.K1));
  } else {
    const int j = k / g.conv_C;
    const int c = k - j * g.conv_C;
    const int tt = r.t + j - (g.taps >> 1);
    if (tt < 0 || tt >= g.conv_T) return z;
    int row = r.rowbase + tt;
    if (g.gather_ids) row = g.gather_ids[row];
    return *reinterpret_cast<const float4*>(g.A1 + (size_t)row * g.lda1 + c);
  }
}

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ void __launch_bounds__(WM* WN * 64)
gemm_kernel(const GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;      // wave tile
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int NLA = BM * (BK / 4) / NT, NLB = BN * (BK / 4) / NT;
  static_assert(MI >= 1 && NI >= 1 && NLA >= 1 && NLB >= 1, "tile too small for the block");
  constexpr int TILE_FLOATS = (BM + BN) * LDS_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware tile order (speed only): block b runs on XCD b % 8; give each XCD a contiguous
  // run of tiles so that neighbours sharing an A row-panel hit the same L2.
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int half = lane >> 5, l31 = lane & 31;

  // ---- per-thread load descriptors -----------------------------------------------------------
  ARow<MODE> arow[NLA];
  int a_kq[NLA], a_lds[NLA];
#pragma unroll
  for (int x = 0; x < NLA; ++x) {
    const int e = tid + x * NT, row = e >> 3, kq = e & 7;
    const int m = m0 + row;
    a_kq[x] = kq * 4;
    a_lds[x] = row * LDS_STRIDE + kq * 4;
    arow[x].valid = m < g.M;
    const int mm = arow[x].valid ? m : 0;
    if (MODE == 0) {
      const int src = g.gather_ids ? g.gather_ids[mm] : mm;
      arow[x].p1 = g.A1 + (size_t)src * g.lda1;
      arow[x].p2 = g.A2 ? g.A2 + (size_t)mm * g.lda2 : nullptr;
      arow[x].t = 0; arow[x].rowbase = 0;
    } else {
      const int b = mm / g.conv_T;
      arow[x].t = mm - b * g.conv_T;
      arow[x].rowbase = b * g.conv_T;
      arow[x].p1 = nullptr; arow[x].p2 = nullptr;
    }
  }
  const float* b_ptr[NLB];
  int b_kq[NLB], b_lds[NLB];
  bool b_valid[NLB];
#pragma unroll
  for (int x = 0; x < NLB; ++x) {
    const int e = tid + x * NT, row = e >> 3, kq = e & 7;
    const int n = n0 + row;
    b_kq[x] = kq * 4;
    b_lds[x] = (BM + row) * LDS_STRIDE + kq * 4;
    b_valid[x] = n < g.N;
    b_ptr[x] = g.Wt + (size_t)(b_valid[x] ? n : 0) * g.ldw;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NLA], rb[NLB];
  auto g2r = [&](int k0) {
#pragma unroll
    for (int x = 0; x < NLA; ++x) ra[x] = load_a4<MODE>(g, arow[x], k0 + a_kq[x]);
#pragma unroll
    for (int x = 0; x < NLB; ++x) {
      const int k = k0 + b_kq[x];
      rb[x] = (b_valid[x] && k < g.K) ? *reinterpret_cast<const float4*>(b_ptr[x] + k)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto r2s = [&](float* buf) {
#pragma unroll
    for (int x = 0; x < NLA; ++x) *reinterpret_cast<float4*>(buf + a_lds[x]) = ra[x];
#pragma unroll
    for (int x = 0; x < NLB; ++x) *reinterpret_cast<float4*>(buf + b_lds[x]) = rb[x];
  };

  const int nk = (g.K + BK - 1) / BK;
  g2r(0);
  r2s(smem);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const float* As = smem + (kt & 1) * TILE_FLOATS;
    const float* Bs = As + BM * LDS_STRIDE;
    if (kt + 1 < nk) g2r((kt + 1) * BK);           // issue next tile's global loads early
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(As + (wm * TM + i * 32 + l31) * LDS_STRIDE + c * 8 + half * 4);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bs + (wn * TN + j * 32 + l31) * LDS_STRIDE + c * 8 + half * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) r2s(smem + ((kt + 1) & 1) * TILE_FLOATS);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn * TN + j * 32 + l31;
    const bool cok = col < g.N;
    const float bias = (cok && g.bias) ? g.bias[col] : 0.f;
    const float bns = (cok && g.bn_scale) ? g.bn_scale[col] : 1.f;
    const float bnb = (cok && g.bn_shift) ? g.bn_shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (cok && row < g.M) {
          float v = acc[i][j][r] + bias;
          if (g.bn_scale && g.bn_first) v = v * bns + bnb;
          v = apply_act(v, g.act);
          if (g.bn_scale && !g.bn_first) v = v * bns + bnb;
          if (g.pe) v += g.pe_w * g.pe[(size_t)(row % g.pe_T) * g.N + col];
          if (g.residual) v += g.residual[(size_t)row * g.ldr + col];
          g.C[(size_t)row * g.ldc + col] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-panel variant with a fused LayerNorm epilogue: one workgroup owns BM = 32 complete output
// rows (N <= NW*64 columns, NW waves side by side), so the row statistics are reduced with
// wavefront shuffles + one LDS exchange instead of a second pass over HBM
// (LayerNormalization after att_proj / FFN: attention.py:412-413,443,450, utils.py:51-52).
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ void __launch_bounds__(NW * 64)
gemm_ln_kernel(const GemmArgs g) {
  constexpr int BM = 32, BN = NW * 64, NT = NW * 64;
  constexpr int NLA = (BM * (BK / 4) + NT - 1) / NT;   // 256 float4 of A per tile
  constexpr int NLB = BN * (BK / 4) / NT;              // = 8
  constexpr int TILE_FLOATS = (BM + BN) * LDS_STRIDE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* red = smem + 2 * TILE_FLOATS;                 // [2][32][NW] row partial sums

  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;

  const float *ap1[NLA], *ap2[NLA];
  int a_lds[NLA], a_k[NLA];
  bool a_valid[NLA], a_loader[NLA];
#pragma unroll
  for (int x = 0; x < NLA; ++x) {
    const int e = tid + x * NT, row = (e >> 3) & 31, kq = e & 7;
    a_loader[x] = e < BM * (BK / 4);
    a_k[x] = kq * 4;
    a_lds[x] = row * LDS_STRIDE + kq * 4;
    const int am = m0 + row;
    a_valid[x] = a_loader[x] && am < g.M;
    ap1[x] = g.A1 + (size_t)(a_valid[x] ? am : 0) * g.lda1;
    ap2[x] = g.A2 ? g.A2 + (size_t)(a_valid[x] ? am : 0) * g.lda2 : nullptr;
  }
  const float* b_ptr[NLB];
  int b_lds[NLB], b_kq[NLB];
  bool b_valid[NLB];
#pragma unroll
  for (int x = 0; x < NLB; ++x) {
    const int e = tid + x * NT, row = e >> 3, kq = e & 7;
    b_kq[x] = kq * 4;
    b_lds[x] = (BM + row) * LDS_STRIDE + kq * 4;
    b_valid[x] = row < g.N;
    b_ptr[x] = g.Wt + (size_t)(b_valid[x] ? row : 0) * g.ldw;
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  float4 ra[NLA], rb[NLB];
  auto g2r = [&](int k0) {
#pragma unroll
    for (int x = 0; x < NLA; ++x) {
      const int k = k0 + a_k[x];
      ra[x] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_valid[x] && k < g.K)
        ra[x] = (k < g.K1) ? *reinterpret_cast<const float4*>(ap1[x] + k)
                           : *reinterpret_cast<const float4*>(ap2[x] + (k - g.K1));
    }
#pragma unroll
    for (int x = 0; x < NLB; ++x) {
      const int kk = k0 + b_kq[x];
      rb[x] = (b_valid[x] && kk < g.K) ? *reinterpret_cast<const float4*>(b_ptr[x] + kk)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto r2s = [&](float* buf) {
#pragma unroll
    for (int x = 0; x < NLA; ++x)
      if (a_loader[x]) *reinterpret_cast<float4*>(buf + a_lds[x]) = ra[x];
#pragma unroll
    for (int x = 0; x < NLB; ++x) *reinterpret_cast<float4*>(buf + b_lds[x]) = rb[x];
  };
  const int nk = (g.K + BK - 1) / BK;
  g2r(0);
  r2s(smem);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const float* As = smem + (kt & 1) * TILE_FLOATS;
    const float* Bs = As + BM * LDS_STRIDE;
    if (kt + 1 < nk) g2r((kt + 1) * BK);
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(As + l31 * LDS_STRIDE + c * 8 + half * 4);
      f32x4 b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(Bs + (wave * 64 + j * 32 + l31) * LDS_STRIDE + c * 8 + half * 4);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[j][s], acc[j], 0, 0, 0);
    }
    if (kt + 1 < nk) r2s(smem + ((kt + 1) & 1) * TILE_FLOATS);
    __syncthreads();
  }

  // epilogue: v = act(acc + bias) + residual ; LayerNorm over the full row
  float v[2][16];
  float gam[2], bet[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wave * 64 + j * 32 + l31;
    const bool cok = col < g.N;
    const float bias = (cok && g.bias) ? g.bias[col] : 0.f;
    gam[j] = cok ? g.ln_gamma[col] : 0.f;
    bet[j] = cok ? g.ln_beta[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      float t = 0.f;
      if (cok && row < g.M) {
        t = apply_act(acc[j][r] + bias, g.act);
        if (g.residual) t += g.residual[(size_t)row * g.ldr + col];
      }
      v[j][r] = t;
    }
  }
  const float inv_n = 1.f / (float)g.N;
  float mean[16], rstd[16];
  // pass 1: mean
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s = v[0][r] + v[1][r];
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);   // stays inside the 32-lane half
    if (l31 == 0) red[((r & 3) + 8 * (r >> 2) + 4 * half) * NW + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[lr * NW + w];
    mean[r] = s * inv_n;
  }
  // pass 2: centred (population) variance, as tf.nn.moments computes it
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 64 + j * 32 + l31;
      const float d = (col < g.N) ? v[j][r] - mean[r] : 0.f;
      s += d * d;
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l31 == 0) red[32 * NW + ((r & 3) + 8 * (r >> 2) + 4 * half) * NW + wave] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lr = (r & 3) + 8 * (r >> 2) + 4 * half;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[32 * NW + lr * NW + w];
    rstd[r] = 1.0f / sqrtf(s * inv_n + kLnEps);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wave * 64 + j * 32 + l31;
    if (col >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row < g.M) g.C[(size_t)row * g.ldc + col] = (v[j][r] - mean[r]) * rstd[r] * gam[j] + bet[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const size_t lds = 2 * (size_t)(BM + BN) * LDS_STRIDE * sizeof(float);
  const dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
  if (g.taps > 0) {
    auto k = gemm_kernel<BM, BN, WM, WN, 1>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, block, lds, s, g, tiles_m, tiles_n);
  } else {
    auto k = gemm_kernel<BM, BN, WM, WN, 0>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    vnr_launch(k, grid, block, lds, s, g, tiles_m, tiles_n);
  }
  return hipGetLastError();
}

template <int NW>
static hipError_t launch_ln_cfg(const GemmArgs& g, hipStream_t s) {
  const size_t lds = (2 * (size_t)(32 + NW * 64) * LDS_STRIDE + 2 * 32 * NW) * sizeof(float);
  auto k = gemm_ln_kernel<NW>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  vnr_launch(k, dim3((g.M + 31) / 32), dim3(NW * 64), lds, s, g);
  return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return hipErrorInvalidValue;
  if ((g.K & 3) || (g.K1 & 3) || (g.lda1 & 3) || (g.A2 && (g.lda2 & 3)) || (g.ldw & 3))
    return hipErrorInvalidValue;
  if (g.taps > 0 && ((g.conv_C & 3) || g.K != g.taps * g.conv_C)) return hipErrorInvalidValue;

  static const bool force_v1 = getenv("VNR_GEMM_V1") != nullptr;   // A/B switch for measurements
  if (!force_v1 && gemm2_supported(g)) return launch_gemm2(g, s);
  if (g.aoi.mode || g.a_split || g.c_split) return hipErrorInvalidValue;       // operand-image / split-row formats exist in gemm2 only

  if (g.ln_gamma) {
    if (g.taps > 0 || g.gather_ids || g.bn_scale || g.pe || g.N > 256) return hipErrorInvalidValue;
    if (g.N <= 128) return launch_ln_cfg<2>(g, s);
    return launch_ln_cfg<4>(g, s);
  }

  // Tile choice: MFMA-bound kernel, one wave per SIMD; cost ~ (workgroups per CU, rounded up)
  // x (32x32 MFMA blocks per wave).  Prefer the larger tile on ties (fewer LDS/L2 bytes per flop).
  struct Cand { int bm, bn, blocks; };
  const Cand cands[4] = {{128, 128, 4}, {64, 128, 2}, {128, 64, 2}, {64, 64, 1}};
  int best = 0; long best_cost = -1;
  for (int i = 0; i < 4; ++i) {
    const long tiles = (long)((g.M + cands[i].bm - 1) / cands[i].bm) * ((g.N + cands[i].bn - 1) / cands[i].bn);
    const long cost = ((tiles + 255) / 256) * cands[i].blocks;
    if (best_cost < 0 || cost < best_cost) { best = i; best_cost = cost; }
  }
  switch (best) {
    case 0: return launch_cfg<128, 128, 2, 2>(g, s);
    case 1: return launch_cfg<64, 128, 2, 2>(g, s);
    case 2: return launch_cfg<128, 64, 2, 2>(g, s);
    default: return launch_cfg<64, 64, 2, 2>(g, s);
  }
}

}  // namespace vnr
