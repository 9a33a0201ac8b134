// Fused multi-head attention core for gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces MultiHeadScaledProductAttention.call after the Q/K/V projections
// (reference modules/attention.py:221-246): split heads, QK^T / sqrt(64) / temperature,
// key AND query length mask (AND causal band), masked fill with -2^32, softmax, .V, merge heads.
// The reference materialises logits, the tiled boolean mask, the paddings tensor and the softmax
// output in memory; here nothing but Q, K, V, the context and (only when asked for) the
// alignments touches HBM.
//
// Work split: grid (ceil(Tq/128), H, B); a workgroup is 4 waves, each wave owns 32 query rows;
// K/V tiles of KT keys are staged once per workgroup in LDS and shared by the 4 waves.
//
// Operand mapping ("swapped" QK^T): the wave computes S^T = K.Q^T, so in the MFMA result layout
//   lane (i = lane&31, h = lane>>5) holds query i and keys  jb*32 + (r&3) + 8*(r>>2) + 4*h, r<16.
// Row-softmax reductions (over keys) are therefore in-lane plus ONE cross-half shuffle, and the
// probabilities are already in A-operand position for O = P.V when the k-steps of that MFMA walk
// keys in the same (jb, r, h) order -- V is read from LDS with the matching row index.
//
// Semantics kept from the reference (SURVEY.md section 8 "must-reproduce" 2):
//   * masked logits are exactly -4294967296.0f, so a fully masked row (padded query) becomes a
//     uniform distribution over ALL Tk keys -- key tiles are skipped only for workgroups/waves
//     whose query rows are all valid (skipped keys then have weight exactly 0);
//   * keys >= Tk do not exist (weight 0, never counted in the uniform case).
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace vnr {

constexpr int KS = 68;   // K tile row stride (floats): conflict-free ds_read_b128 along d
constexpr int VS = 64;   // V tile row stride: 32 consecutive lanes read 32 consecutive floats

__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int KT, bool WRITE_ALI>
__global__ void __launch_bounds__(256)
attn_kernel(const AttnArgs a) {
  constexpr int NJB = KT / 32;            // 32-key blocks per tile
  constexpr int NLD = KT * 16 / 256;      // float4 per thread per matrix per tile
  constexpr int PS = KT + 1;              // staging row stride for the alignment transpose
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;
  float* Vs = Ks + KT * KS;
  float* Ps = Vs + KT * VS;               // [4 waves][32][PS], only when WRITE_ALI

  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int Q0 = blockIdx.x * 128, q0 = Q0 + wave * 32;
  const int qlen = a.q_len ? a.q_len[b] : a.Tq;
  const int klen = a.k_len ? a.k_len[b] : a.Tk;
  const int ntiles_all = (a.Tk + KT - 1) / KT;

  auto tiles_for = [&](int row_lo, int row_hi) -> int {
    if (row_lo >= a.Tq) return 0;
    if (row_hi > a.Tq) row_hi = a.Tq;
    if (row_hi > qlen || klen <= 0) return ntiles_all;   // some padded query: uniform over all keys
    int kmax = klen;
    if (a.causal && row_hi < kmax) kmax = row_hi;        // keys j <= i < row_hi
    return (kmax + KT - 1) / KT;
  };
  const int wg_tiles = tiles_for(Q0, Q0 + 128);
  const int wv_tiles = tiles_for(q0, q0 + 32);

  const float* Kb = a.K + (size_t)b * a.k_bs + hd * 64;
  const float* Vb = a.V + (size_t)b * a.v_bs + hd * 64;

  // ---- Q fragment: lane (i,h) keeps Q[q0+i][8c+4h .. +3], c = 0..7 --------------------------
  f32x4 qf[8];
  {
    const int iq = q0 + l31;
    const bool qok = iq < a.Tq;
    const float* qp = a.Q + (size_t)b * a.q_bs + (size_t)(qok ? iq : 0) * a.ldq + hd * 64 + half * 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (qok) qf[c] = *reinterpret_cast<const f32x4*>(qp + c * 8);
      else { qf[c][0] = 0.f; qf[c][1] = 0.f; qf[c][2] = 0.f; qf[c][3] = 0.f; }
    }
  }
  const float tau = a.temperature;
  const bool use_tau = a.temperature != 1.0f;

  float4 rk[NLD], rv[NLD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int x = 0; x < NLD; ++x) {
      const int e = tid + x * 256, row = e >> 4, c4 = (e & 15) * 4;
      const int key = kt * KT + row;
      if (key < a.Tk) {
        rk[x] = *reinterpret_cast<const float4*>(Kb + (size_t)key * a.ldk + c4);
        rv[x] = *reinterpret_cast<const float4*>(Vb + (size_t)key * a.ldv + c4);
      } else {
        rk[x] = make_float4(0.f, 0.f, 0.f, 0.f);
        rv[x] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int x = 0; x < NLD; ++x) {
      const int e = tid + x * 256, row = e >> 4, c4 = (e & 15) * 4;
      *reinterpret_cast<float4*>(Ks + row * KS + c4) = rk[x];
      *reinterpret_cast<float4*>(Vs + row * VS + c4) = rv[x];
    }
  };

  f32x16 st[NJB];
  // S^T tile = K_tile . Q^T, then scale and mask exactly like attention.py:227-241
  auto qk_tile = [&](int kt) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[jb][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (jb * 32 + l31) * KS + c * 8 + half * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          st[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[c][s], st[jb], 0, 0, 0);
      }
    }
    const int iq = q0 + l31;
    const bool qvalid = iq < qlen;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kt * KT + jb * 32 + frag_row(r, half);
        float s = st[jb][r] * 0.125f;                 // / sqrt(64), exact
        if (use_tau) s = s / tau;                     // / temperature
        const bool ok = qvalid && (j < klen) && (!a.causal || j <= iq);
        s = ok ? s : kMaskFill;
        if (j >= a.Tk) s = -INFINITY;
        st[jb][r] = s;
      }
  };
  auto tile_max = [&]() -> float {
    float m = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, st[jb][r]);
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };

  f32x16 O[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[nb][r] = 0.f;
  // O += P . V over this tile; P (st) is the A operand as-is, V rows follow the (jb, r, half) walk
  auto pv_tile = [&]() {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vrow = Vs + (jb * 32 + frag_row(r, half)) * VS + l31;
        const float v0 = vrow[0], v1 = vrow[32];
        O[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[jb][r], v0, O[0], 0, 0, 0);
        O[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[jb][r], v1, O[1], 0, 0, 0);
      }
  };

  float m_run = -INFINITY, l_run = 0.f;

  if (!WRITE_ALI) {
    // ---------------- single pass, online softmax ----------------------------------------------
    if (wg_tiles > 0) { load_tile(0); store_tile(); }
    __syncthreads();
    for (int kt = 0; kt < wg_tiles; ++kt) {
      if (kt + 1 < wg_tiles) load_tile(kt + 1);          // prefetch into registers
      if (kt < wv_tiles) {
        qk_tile(kt);
        const float m_new = fmaxf(m_run, tile_max());
        const float alpha = expf(m_run - m_new);          // 0 on the first tile
        float ps = 0.f;
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = expf(st[jb][r] - m_new);
            st[jb][r] = p;
            ps += p;
          }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (kt > 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ar = __shfl(alpha, frag_row(r, half), 64);
            O[0][r] *= ar;
            O[1][r] *= ar;
          }
        }
        pv_tile();
      }
      __syncthreads();
      if (kt + 1 < wg_tiles) { store_tile(); __syncthreads(); }
    }
    if (wv_tiles > 0) {
      const float linv = 1.0f / l_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float li = __shfl(linv, frag_row(r, half), 64);
        O[0][r] *= li;
        O[1][r] *= li;
      }
    }
  } else {
    // ---------------- two passes: statistics, then normalised P (stored) and P.V -----------------
    for (int kt = 0; kt < wg_tiles; ++kt) {
      load_tile(kt); store_tile();
      __syncthreads();
      if (kt < wv_tiles) {
        qk_tile(kt);
        const float m_new = fmaxf(m_run, tile_max());
        float ps = 0.f;
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
          for (int r = 0; r < 16; ++r) ps += expf(st[jb][r] - m_new);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * expf(m_run - m_new) + ps;
        m_run = m_new;
      }
      if (wg_tiles > 1) __syncthreads();
    }
    float* Pw = Ps + wave * 32 * PS;
    for (int kt = 0; kt < wg_tiles; ++kt) {
      if (wg_tiles > 1) { load_tile(kt); store_tile(); __syncthreads(); }
      if (kt < wv_tiles) {
        if (wg_tiles > 1) qk_tile(kt);        // single tile: st still holds the masked logits
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = expf(st[jb][r] - m_run) / l_run;      // softmax, attention.py:242
            st[jb][r] = p;
            Pw[l31 * PS + jb * 32 + frag_row(r, half)] = p;
          }
        // transposed, coalesced store of the alignment rows of this wave
        for (int rr = 0; rr < 32; ++rr) {
          const int iq = q0 + rr;
          if (iq >= a.Tq) break;
          float* arow = a.ali + (((size_t)b * a.H + hd) * a.Tq + iq) * a.Tk + kt * KT;
#pragma unroll
          for (int x = 0; x < KT / 64; ++x) {
            const int j = lane + x * 64;
            if (kt * KT + j < a.Tk) arow[j] = Pw[rr * PS + j];
          }
        }
        pv_tile();
      }
      if (wg_tiles > 1) __syncthreads();
    }
    // keys of skipped tiles have weight exactly 0
    for (int kt = wv_tiles; kt < ntiles_all; ++kt) {
      for (int rr = 0; rr < 32; ++rr) {
        const int iq = q0 + rr;
        if (iq >= a.Tq) break;
        float* arow = a.ali + (((size_t)b * a.H + hd) * a.Tq + iq) * a.Tk + kt * KT;
#pragma unroll
        for (int x = 0; x < KT / 64; ++x) {
          const int j = lane + x * 64;
          if (kt * KT + j < a.Tk) arow[j] = 0.f;
        }
      }
    }
  }

  // ---- context store: lane holds dv = nb*32 + l31 for queries frag_row(r, half) ----------------
  if (wv_tiles > 0) {
    float* ob = a.ctx + (size_t)b * a.o_bs + hd * 64 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int iq = q0 + frag_row(r, half);
      if (iq < a.Tq) {
        ob[(size_t)iq * a.ldo] = O[0][r];
        ob[(size_t)iq * a.ldo + 32] = O[1][r];
      }
    }
  }
}

template <int KT, bool ALI>
static hipError_t launch_attn_cfg(const AttnArgs& a, hipStream_t s) {
  size_t lds = (size_t)KT * (KS + VS) * sizeof(float);
  if (ALI) lds += (size_t)4 * 32 * (KT + 1) * sizeof(float);
  auto k = attn_kernel<KT, ALI>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  dim3 grid((a.Tq + 127) / 128, a.H, a.B);
  vnr_launch(k, grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.H <= 0 || a.Tq <= 0 || a.Tk <= 0) return hipErrorInvalidValue;
  if ((a.ldq & 3) || (a.ldk & 3) || (a.ldv & 3)) return hipErrorInvalidValue;
  static const bool force_v1 = getenv("VNR_ATTN_V1") != nullptr;   // A/B switch for measurements
  if (!force_v1 && attention2_supported(a)) return launch_attention2(a, s);
  if (a.ali) return launch_attn_cfg<128, true>(a, s);
  if (a.causal) return launch_attn_cfg<64, false>(a, s);
  return launch_attn_cfg<128, false>(a, s);
}

}  // namespace vnr
