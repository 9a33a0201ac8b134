// Small HBM-bound kernels of the VAENAR-TTS path (gfx950): LayerNorm, positional-encoding table,
// flow coupling, length predictor, masked reductions and weight preparation.
// Wavefront = 64 lanes everywhere; row reductions use wave shuffles.
#include "common.h"
#include <stdlib.h>
#include <math.h>

namespace vnr {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- tf.keras.layers.LayerNormalization(): one wave per row, two-pass statistics in registers ---
// (attention.py:402,428,433; utils.py:46).  dim % 4 == 0, dim <= 64*4*VPT.
template <int VPT>   // float4 per lane
__global__ void __launch_bounds__(256)
layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                  const float* __restrict__ beta, int rows, int dim, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * dim;
  float4 v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (lane + i * 64) * 4;
    v[i] = (c < dim) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < dim) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + kLnEps);
  float* yr = y + (size_t)row * dim;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < dim) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      *reinterpret_cast<float4*>(yr + c) = o;
    }
  }
}

hipError_t launch_layer_norm(const float* x, const float* gamma, const float* beta, int rows,
                             int dim, float* y, hipStream_t s) {
  if (rows <= 0 || dim <= 0 || (dim & 3) || dim > 2048) return hipErrorInvalidValue;
  const dim3 grid((rows + 3) / 4), block(256);
  if (dim <= 256) vnr_launch(layer_norm_kernel<1>, grid, block, 0, s, x, gamma, beta, rows, dim, y);
  else if (dim <= 512) vnr_launch(layer_norm_kernel<2>, grid, block, 0, s, x, gamma, beta, rows, dim, y);
  else if (dim <= 1024) vnr_launch(layer_norm_kernel<4>, grid, block, 0, s, x, gamma, beta, rows, dim, y);
  else vnr_launch(layer_norm_kernel<8>, grid, block, 0, s, x, gamma, beta, rows, dim, y);
  return hipGetLastError();
}

// ---- PositionalEncoding.positional_encoding (utils.py:333-355) ----------------------------------
// The reference evaluates every stage in float32; each stage here is the correctly rounded
// float32 value (transcendentals evaluated in double, then rounded).
__global__ void pe_kernel(int T, int dim, float step, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= T * dim) return;
  const int t = idx / dim, d = idx - t * dim;
  const float pos = (float)t * step;                                  // utils.py:342
  const float df = (float)d;
  const bool even = (d & 1) == 0;                                     // utils.py:351-352
  const float e = (even ? df : df - 1.0f) / (float)dim;               // utils.py:353-354
  const float w = (float)pow(10000.0, (double)e);
  const float arg = pos / w;
  out[idx] = even ? (float)sin((double)arg) : (float)cos((double)arg);
}

hipError_t launch_positional_encoding(int T, int dim, float step, float* out, hipStream_t s) {
  if (T <= 0 || dim <= 0) return hipErrorInvalidValue;
  const int n = T * dim;
  vnr_launch(pe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, T, dim, step, out);
  return hipGetLastError();
}

// ---- weight layout: out[c*ldo + r] = in[r*cols + c] ----------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ in, int rows, int cols,
                                 float* __restrict__ out, int ldo) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = by + i, c = bx + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = bx + i, r = by + threadIdx.x;
    if (c < cols && r < rows) out[(size_t)c * ldo + r] = tile[threadIdx.x][i];
  }
}

hipError_t launch_transpose(const float* in, int rows, int cols, float* out, int ldo, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return hipErrorInvalidValue;
  vnr_launch(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s,
                     in, rows, cols, out, ldo);
  return hipGetLastError();
}

// ---- TransformerCoupling._forward tail (flow.py:231-237) -----------------------------------------
// heads [M, 2*half] = log_scale | shift ; zp = z[:, zp_off : zp_off+half] updated in place;
// row_logdet[m] = sum_c log(sigmoid(log_scale + 2)).  One wave per row.
__global__ void __launch_bounds__(256)
coupling_fwd_kernel(const float* __restrict__ heads, float* __restrict__ z, int M, int half,
                    int ldz, int zp_off, float* __restrict__ row_logdet) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* hr = heads + (size_t)row * 2 * half;
  float* zr = z + (size_t)row * ldz + zp_off;
  float acc = 0.f;
  for (int c = lane; c < half; c += 64) {
    const float ls = hr[c], sh = hr[half + c];
    const float scale = 1.0f / (1.0f + expf(-(ls + 2.0f)));            // tf.math.sigmoid(log_scale + 2)
    zr[c] = scale * zr[c] + sh;                                        // _affine, flow.py:216
    acc += logf(scale);
  }
  acc = wave_sum(acc);
  if (lane == 0 && row_logdet) row_logdet[row] = acc;
}

hipError_t launch_coupling_fwd(const float* heads, float* z, int M, int half, int zp_off,
                               float* row_logdet, hipStream_t s) {
  vnr_launch(coupling_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, heads, z, M, half,
                     2 * half, zp_off, row_logdet);
  return hipGetLastError();
}

// ---- out[b] (+)= scale * sum_{t < len[b]} rows[b*T + t]; fixed summation order, fp64 accumulate ---
__global__ void __launch_bounds__(64)
masked_row_reduce_kernel(const float* __restrict__ rows, const int32_t* __restrict__ len, int T,
                         float scale, float* __restrict__ out, int accumulate) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = len ? min(len[b], T) : T;
  double acc = 0.0;
  for (int t = lane; t < n; t += 64) acc += (double)rows[(size_t)b * T + t];
  acc = wave_sum_d(acc);
  if (lane == 0) {
    const float v = scale * (float)acc;
    out[b] = accumulate ? out[b] + v : v;
  }
}

// ---- TransformerCoupling._backward tail (flow.py:246-255): zp <- (zp - shift) / (sigmoid(ls+2) + 1e-12) ---------
__global__ void __launch_bounds__(256)
coupling_bwd_kernel(const float* __restrict__ heads, float* __restrict__ z, int M, int half, int ldz,
                    int zp_off, float* __restrict__ row_logdet) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* hr = heads + (size_t)row * 2 * half;
  float* zr = z + (size_t)row * ldz + zp_off;
  float acc = 0.f;
  for (int c = lane; c < half; c += 64) {
    const float ls = hr[c], sh = hr[half + c];
    const float scale = 1.0f / (1.0f + expf(-(ls + 2.0f)));
    zr[c] = (zr[c] - sh) / (scale + 1e-12f);                            // _inverse_affine, flow.py:220
    acc += logf(scale);
  }
  acc = wave_sum(acc);
  if (lane == 0 && row_logdet) row_logdet[row] = acc;
}
hipError_t launch_coupling_bwd(const float* heads, float* z, int M, int half, int zp_off,
                               float* row_logdet, hipStream_t s) {
  vnr_launch(coupling_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, heads, z, M, half,
                     2 * half, zp_off, row_logdet);
  return hipGetLastError();
}

// ---- BasePosterior.reparameterize + log_probability rows (posterior.py:21-39, 42-72), n_sample = 1 ------------------
// z = eps * exp(0.5 * logvar) + mu ; row_lp[m] = -0.5 * (C*log(2pi) + sum_c (logvar + eps^2)).  One wave per row.
__global__ void __launch_bounds__(256)
reparam_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
               int M, int C, float* __restrict__ z, float* __restrict__ row_lp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const size_t o = (size_t)row * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float lv = logvar[o + c], e = eps ? eps[o + c] : 0.f;
    z[o + c] = e * expf(0.5f * lv) + mu[o + c];
    acc += lv + e * e;
  }
  acc = wave_sum(acc);
  if (lane == 0) row_lp[row] = -0.5f * ((float)C * 1.8378770664093453f + acc);
}
hipError_t launch_reparam(const float* mu, const float* logvar, const float* eps, int M, int C, float* z,
                          float* row_lp, hipStream_t s) {
  vnr_launch(reparam_kernel, dim3((M + 3) / 4), dim3(256), 0, s, mu, logvar, eps, M, C, z, row_lp);
  return hipGetLastError();
}

// ---- rows of VAENAR._compute_l2_loss (models.py:78): row[b*T + t] = mean_c (rec - tgt)^2 ------------------------------
__global__ void __launch_bounds__(256)
sqerr_rows_kernel(const float* __restrict__ rec, int rec_T, const float* __restrict__ tgt, int T, int B, int C,
                  float* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int b = row / T, t = row - b * T;
  const float* r = rec + ((size_t)b * rec_T + t) * C;
  const float* g = tgt + (size_t)row * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = r[c] - g[c]; acc += d * d; }
  acc = wave_sum(acc);
  if (lane == 0) rows[row] = acc / (float)C;
}
hipError_t launch_sqerr_rows(const float* rec, int rec_T, const float* tgt, int T, int B, int C, float* rows,
                             hipStream_t s) {
  vnr_launch(sqerr_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, rec, rec_T, tgt, T, B, C, rows);
  return hipGetLastError();
}

// ---- per-utterance scalars of the ELBO (models.py:89-103, 184-196) ------------------------------------------------------
// l2[b] = (sum_out[b] + sum_init[b]) / len[b] ; length[b] = (log pred - log len)^2 ; kl[b] = post_lp - prior_lp
__global__ void elbo_scalars_kernel(const float* sum_out, const float* sum_init, const int32_t* mel_len,
                                    const float* pred_len, const float* post_lp, const float* prior_lp, int B,
                                    float* l2, float* length_l2, float* kl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float n = (float)mel_len[b];
  l2[b] = sum_out[b] / n + sum_init[b] / n;
  const float d = logf(pred_len[b]) - logf(n);
  length_l2[b] = d * d;
  kl[b] = post_lp[b] - prior_lp[b];
}
hipError_t launch_elbo_scalars(const float* sum_out, const float* sum_init, const int32_t* mel_len,
                               const float* pred_len, const float* post_lp, const float* prior_lp, int B,
                               float* l2, float* length_l2, float* kl, hipStream_t s) {
  vnr_launch(elbo_scalars_kernel, dim3((B + 63) / 64), dim3(64), 0, s, sum_out, sum_init, mel_len,
                     pred_len, post_lp, prior_lp, B, l2, length_l2, kl);
  return hipGetLastError();
}

// ---- max |x| of a buffer (bits of the non-negative float compare like unsigned ints) ---------------------------------
__global__ void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) amax_publish(out, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}
hipError_t launch_absmax(const float* x, size_t n, unsigned* out, hipStream_t s) {
  const unsigned blocks = (unsigned)((n + 255) / 256 > 512 ? 512 : (n + 255) / 256);
  vnr_launch(absmax_kernel, dim3(blocks), dim3(256), 0, s, x, n, out);
  return hipGetLastError();
}

// ---- weight image of the split-fp16 GEMM path: w*scale = hi + lo (two fp16), k-tiles of 32 ------------------------------
__global__ void split_weights_kernel(const float* __restrict__ Wt, int N, int K, float scale, _Float16* __restrict__ out) {
  const int KT = (K + 31) >> 5;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * KT * 32) return;
  const int p = idx & 31; const size_t nk = idx >> 5; const int kt = nk % KT; const size_t n = nk / KT;
  const int k = kt * 32 + p;
  const float w = k < K ? Wt[n * K + k] * scale : 0.f;
  _Float16 hi, lo;
  vnr_split(w, hi, lo);
  out[nk * 64 + p] = hi;
  out[nk * 64 + 32 + p] = lo;
}
hipError_t launch_split_weights(const float* Wt, int N, int K, float scale, void* out, hipStream_t s) {
  const size_t n = (size_t)N * ((K + 31) >> 5) * 32;
  vnr_launch(split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wt, N, K, scale, (_Float16*)out);
  return hipGetLastError();
}

// ---- operand-major split image for the row-panel chain kernel (gemm3.hip) ------------------------------------------------------
// out[((nb*KT + kt)*4 + 2*t + part)*512 + lane*8 + e] = part(w[32 nb + (lane&31)][32 kt + 16 t + 8 (lane>>5) + e] * scale)
__global__ void opmajor_weights_kernel(const float* __restrict__ Wt, int N, int K, float scale, _Float16* __restrict__ out) {
  const int KT = (K + 31) >> 5, NB = (N + 31) >> 5;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (nb, kt, t, lane)
  if (idx >= (size_t)NB * KT * 2 * 64) return;
  const int lane = idx & 63, t = (idx >> 6) & 1;
  const size_t blk = idx >> 7; const int kt = blk % KT; const size_t nb = blk / KT;
  const int n = (int)nb * 32 + (lane & 31), k0 = kt * 32 + 16 * t + 8 * (lane >> 5);
  _Float16* hi = out + ((blk * 4 + 2 * t) * 512) + lane * 8;
  _Float16* lo = hi + 512;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float w = (n < N && k0 + e < K) ? Wt[(size_t)n * K + k0 + e] * scale : 0.f;
    _Float16 h, l;
    vnr_split(w, h, l);
    hi[e] = h; lo[e] = l;
  }
}
hipError_t launch_opmajor_weights(const float* Wt, int N, int K, float scale, void* out, hipStream_t s) {
  const size_t n = (size_t)((N + 31) / 32) * ((K + 31) / 32) * 128;
  vnr_launch(opmajor_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wt, N, K, scale, (_Float16*)out);
  return hipGetLastError();
}

// ---- tf.keras.layers.Embedding (encoder.py:81): out[r, :] = table[ids[r], :]; one wave per row, float4 lanes ------
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, int rows, int dim,
                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = table + (size_t)ids[row] * dim;
  float* dst = out + (size_t)row * dim;
  for (int c = lane * 4; c < dim; c += 256) *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
}
// Embedding gather straight into "split rows" (common.h, GemmArgs::a_split): lane handles 4 channels of one row
__global__ void __launch_bounds__(256)
gather_rows_split_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, int rows, int dim, char* __restrict__ out) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* src = table + (size_t)ids[row] * dim;
  char* dst = out + (size_t)row * dim * 4;
  for (int c = lane * 4; c < dim; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + c);
    const float x[4] = {v.x, v.y, v.z, v.w};
    h4 hi, lo;
    { const vnr_f4 xs_ = {x[0], x[1], x[2], x[3]}; vnr_split(xs_, hi, lo); }
    char* p = dst + (c >> 5) * 128 + (c & 31) * 2;
    *reinterpret_cast<h4*>(p) = hi;
    *reinterpret_cast<h4*>(p + 64) = lo;
  }
}
hipError_t launch_gather_rows_split(const float* table, const int32_t* ids, int rows, int dim, float* out, hipStream_t s) {
  if (dim & 31) return hipErrorInvalidValue;
  vnr_launch(gather_rows_split_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, table, ids, rows, dim, reinterpret_cast<char*>(out));
  return hipGetLastError();
}
hipError_t launch_gather_rows(const float* table, const int32_t* ids, int rows, int dim, float* out, hipStream_t s) {
  if (dim & 3) return hipErrorInvalidValue;
  vnr_launch(gather_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, table, ids, rows, dim, out);
  return hipGetLastError();
}

hipError_t launch_masked_row_reduce(const float* rows, const int32_t* len, int B, int T, float scale,
                                    float* out, int accumulate, hipStream_t s) {
  vnr_launch(masked_row_reduce_kernel, dim3(B), dim3(64), 0, s, rows, len, T, scale, out,
                     accumulate);
  return hipGetLastError();
}

// ---- DenseLengthPredictor.call (length_predictor.py:35-42) ----------------------------------------
// One workgroup (4 waves) per utterance; a wave takes rows t = wave, wave+4, ...; the dot product
// x[b,t,:].w is a wave reduction; exp in fp32 like the reference; the final masked sum over time is
// accumulated in fp64 (it feeds an int32 truncation, inference.py:135).
__global__ void __launch_bounds__(256)
length_predictor_kernel(const float* __restrict__ x, const float* __restrict__ w,
                        const float* __restrict__ bias, const int32_t* __restrict__ len, int T,
                        int D, int act, float* __restrict__ out) {
  __shared__ double part[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = len ? min(len[b], T) : T;
  double acc = 0.0;
  for (int t = wave; t < n; t += 4) {
    const float* xr = x + ((size_t)b * T + t) * D;
    float d = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + c);
      const float4 wv = *reinterpret_cast<const float4*>(w + c);
      d += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
    }
    d = wave_sum(d) + bias[0];
    if (act == ACT_RELU) d = fmaxf(d, 0.f);
    else if (act == ACT_TANH) d = tanhf(d);
    acc += (double)expf(d);
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (float)((part[0] + part[1]) + (part[2] + part[3]));
}

hipError_t launch_length_predictor(const float* x, const float* w, const float* bias,
                                   const int32_t* len, int B, int T, int D, int act, float* out,
                                   hipStream_t s) {
  if (D & 3) return hipErrorInvalidValue;
  vnr_launch(length_predictor_kernel, dim3(B), dim3(256), 0, s, x, w, bias, len, T, D, act, out);
  return hipGetLastError();
}

// ---- BasePrior._initial_sample log-probability (prior.py:36-41) -----------------------------------
__global__ void __launch_bounds__(256)
gauss_logprob_kernel(const float* __restrict__ eps, const int32_t* __restrict__ len, int T, int C,
                     float* __restrict__ out) {
  __shared__ double part[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = len ? min(len[b], T) : T;
  const float log2pi = 1.8378770664093453f;
  double acc = 0.0;
  const size_t total = (size_t)n * C;
  const float* e = eps ? eps + (size_t)b * T * C : nullptr;
  for (size_t i = threadIdx.x; i < total; i += 256) {
    const float v = e ? e[i] : 0.f;
    acc += (double)(-0.5f * (log2pi + v * v));
  }
  acc = wave_sum_d(acc);
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (float)((part[0] + part[1]) + (part[2] + part[3]));
}

hipError_t launch_gauss_logprob(const float* eps, const int32_t* len, int B, int T, int C, float* out,
                                hipStream_t s) {
  vnr_launch(gauss_logprob_kernel, dim3(B), dim3(256), 0, s, eps, len, T, C, out);
  return hipGetLastError();
}

__global__ void axpy_len_kernel(float* y, const int32_t* len, float alpha, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) y[b] += alpha * (float)len[b];
}
hipError_t launch_axpy_len(float* y, const int32_t* len, float alpha, int B, hipStream_t s) {
  vnr_launch(axpy_len_kernel, dim3((B + 63) / 64), dim3(64), 0, s, y, len, alpha, B);
  return hipGetLastError();
}

// ---- ActNorm o InvertibleLinear folding (flow.py:166-175 then 123-135) ----------------------------
// (z*exp(ls) + b).W == z.(diag(exp(ls)) W) + b.W  ->  Wt_out[n][k] = exp(ls[k]) W[k][n], b_out = b.W
__global__ void fold_actnorm_linear_kernel(const float* __restrict__ ls, const float* __restrict__ bias,
                                           const float* __restrict__ W, int C, float* __restrict__ Wt,
                                           float* __restrict__ bout) {
  const int n = blockIdx.x;            // output channel
  double acc = 0.0;
  for (int k = threadIdx.x; k < C; k += blockDim.x) {
    const float w = W[(size_t)k * C + n];
    Wt[(size_t)n * C + k] = expf(ls[k]) * w;
    acc += (double)bias[k] * (double)w;
  }
  acc = wave_sum_d(acc);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += part[i];
    bout[n] = (float)t;
  }
}
hipError_t launch_fold_actnorm_linear(const float* log_scale, const float* bias, const float* W, int C,
                                      float* Wt_out, float* b_out, hipStream_t s) {
  vnr_launch(fold_actnorm_linear_kernel, dim3(C), dim3(128), 0, s, log_scale, bias, W, C,
                     Wt_out, b_out);
  return hipGetLastError();
}

// ---- BatchNormalization inference affine (tf.nn.batch_normalization): -----------------------------
// inv = gamma * rsqrt(var + eps); y = x*inv + (beta - mean*inv)
__global__ void bn_affine_kernel(const float* gamma, const float* beta, const float* mean,
                                 const float* var, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = gamma[c] * (1.0f / sqrtf(var[c] + kBnEps));
  scale[c] = inv;
  shift[c] = beta[c] - mean[c] * inv;
}
hipError_t launch_bn_affine(const float* gamma, const float* beta, const float* mean, const float* var,
                            int C, float* scale, float* shift, hipStream_t s) {
  vnr_launch(bn_affine_kernel, dim3((C + 127) / 128), dim3(128), 0, s, gamma, beta, mean, var,
                     C, scale, shift);
  return hipGetLastError();
}

// ---- training-mode statistics and dropout ------------------------------------------------------------
// Column sums of x[M][C] (row stride ld) in float64: out[c] += sum_m f(x[m][c]) with f(v) = v, or (v - mean[c])^2
// when `mean` is given (second pass of a two-pass variance).  Used by BatchNormalization(training=True)
// (tf.nn.moments over axes (0,1), padded frames included: utils.py:79-83) and ActNormFlow.init (flow.py:189-196).
__global__ void col_sum_kernel(const float* x, int M, int C, int ld, const double* mean, double* out, unsigned* amax, float* fout, double* det) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;                       // 4 row groups per block
  double acc = 0.0;
  float mx = 0.f;
  if (c < C) {
    const double mu = mean ? mean[c] : 0.0;
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + rg;
    for (; m + 3 * step < M; m += 4 * step) {            // four independent loads in flight per thread
      const float x0 = x[(size_t)m * ld + c], x1 = x[(size_t)(m + step) * ld + c], x2 = x[(size_t)(m + 2 * step) * ld + c],
                  x3 = x[(size_t)(m + 3 * step) * ld + c];
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x0), fabsf(x1))), fmaxf(fabsf(x2), fabsf(x3)));
      if (mean) acc += ((double)x0 - mu) * ((double)x0 - mu) + ((double)x1 - mu) * ((double)x1 - mu) + ((double)x2 - mu) * ((double)x2 - mu) +
                       ((double)x3 - mu) * ((double)x3 - mu);
      else acc += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
    }
    for (; m < M; m += step) {
      const float xv = x[(size_t)m * ld + c];
      const double v = (double)xv;
      mx = fmaxf(mx, fabsf(xv));
      acc += mean ? (v - mu) * (v - mu) : v;
    }
  }
  __shared__ double part[4][64];
  part[rg][threadIdx.x & 63] = acc;
  __shared__ float wmx[4];
  if (amax) {                                            // by-product: max |x| of the block (scale of the split-fp16 gradient GEMM),
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));      // one candidate per workgroup (same-address atomics serialise)
    if ((threadIdx.x & 63) == 0) wmx[rg] = mx;
  }
  __syncthreads();
  if (amax && threadIdx.x == 0) amax_publish(amax, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
  if (rg == 0 && c < C) {
    const double t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (det) det[(size_t)blockIdx.y * C + c] = t;        // deterministic mode: this row group's partial, added in order by the finish kernel
    else if (fout) atomicAdd(&fout[c], (float)t);        // straight into a float32 gradient (<= 128 block partials per column)
    else atomicAdd(&out[c], t);
  }
}
// Same sums with 16-byte loads: a lane owns 4 consecutive columns (C, ld multiples of 4, x 16-byte aligned), a wave covers 256
// columns of a row, the block's 4 waves take rows m, m+1, m+2, m+3 of a stride-(4 gridDim.y) walk with two loads in flight.
// yact != null: x is a gradient dy and yact the OUTPUT of an activation (act_bwd_kernel's job folded into this pass: dy *= act'(y) is
// written back in place before it enters the sums) -- one launch and one pass over dy less per activated Dense / Conv1D layer.
__global__ void __launch_bounds__(256) col_sum4_kernel(float* x, int M, int C, int ld, const double* mean, double* out, unsigned* amax, float* fout, const float* yact, int act,
                                double* out2, double* det) {      // out2 != null (mean == null): also out2[c] += sum_m x^2 -- both BatchNorm sums in ONE pass
  const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
  const int rg = threadIdx.x >> 6;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  double acc2[4] = {0.0, 0.0, 0.0, 0.0};
  float mx = 0.f;
  if (c < C) {
    double mu[4] = {0.0, 0.0, 0.0, 0.0};
    if (mean) { mu[0] = mean[c]; mu[1] = mean[c + 1]; mu[2] = mean[c + 2]; mu[3] = mean[c + 3]; }
    const int step = gridDim.y * 4;
    int m = blockIdx.y * 4 + rg;
    auto add = [&](const float4& v) {
      const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mx = fmaxf(mx, fabsf(xv[e]));
        const double d = (double)xv[e] - mu[e];
        acc[e] += mean ? d * d : (double)xv[e];
        if (out2) acc2[e] += (double)xv[e] * (double)xv[e];
      }
    };
    auto dact = [&](float4& v, const float4& y) {
      if (act == ACT_RELU) { v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f; }
      else if (act == ACT_TANH) { v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y; v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w; }
    };
    if (yact) {                                                // (the activation output is a contiguous [M][C] block)
      // (a lane walks ~25 rows of a T1-sized gradient: with two rows in flight the kernel was latency-bound at 0.9 TB/s; four rows = 8 loads)
      for (; m + 3 * step < M; m += 4 * step) {
        float4* p[4]; float4 v[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { p[u] = reinterpret_cast<float4*>(x + (size_t)(m + u * step) * ld + c); v[u] = *p[u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = *reinterpret_cast<const float4*>(yact + (size_t)(m + u * step) * C + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) { dact(v[u], y[u]); *p[u] = v[u]; add(v[u]); }
      }
      for (; m + step < M; m += 2 * step) {
        float4* p0 = reinterpret_cast<float4*>(x + (size_t)m * ld + c);
        float4* p1 = reinterpret_cast<float4*>(x + (size_t)(m + step) * ld + c);
        float4 v0 = *p0, v1 = *p1;
        const float4 y0 = *reinterpret_cast<const float4*>(yact + (size_t)m * C + c), y1 = *reinterpret_cast<const float4*>(yact + (size_t)(m + step) * C + c);
        dact(v0, y0); dact(v1, y1);
        *p0 = v0; *p1 = v1;
        add(v0); add(v1);
      }
      for (; m < M; m += step) {
        float4* p0 = reinterpret_cast<float4*>(x + (size_t)m * ld + c);
        float4 v0 = *p0;
        dact(v0, *reinterpret_cast<const float4*>(yact + (size_t)m * C + c));
        *p0 = v0;
        add(v0);
      }
    } else {
    for (; m + 7 * step < M; m += 8 * step) {                  // eight rows in flight (see above)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(x + (size_t)(m + u * step) * ld + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) add(v[u]);
    }
    for (; m + step < M; m += 2 * step) {
      const float4 v0 = *reinterpret_cast<const float4*>(x + (size_t)m * ld + c);
      const float4 v1 = *reinterpret_cast<const float4*>(x + (size_t)(m + step) * ld + c);
      add(v0); add(v1);
    }
    for (; m < M; m += step) add(*reinterpret_cast<const float4*>(x + (size_t)m * ld + c));
    }
  }
  __shared__ double part[4][256];
#pragma unroll
  for (int e = 0; e < 4; ++e) part[rg][(threadIdx.x & 63) * 4 + e] = acc[e];
  __shared__ float wmx[4];
  if (amax) {
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) wmx[rg] = mx;
  }
  __syncthreads();
  if (amax && threadIdx.x == 0) amax_publish(amax, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
    const double t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    if (det) det[(size_t)blockIdx.y * C + cc] = t;
    else if (fout) atomicAdd(&fout[cc], (float)t);
    else atomicAdd(&out[cc], t);
  }
  if (out2) {                                                // (workgroup-uniform)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) part[rg][(threadIdx.x & 63) * 4 + e] = acc2[e];
    __syncthreads();
    if (cc < C) {
      const double t2 = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
      if (det) det[((size_t)gridDim.y + blockIdx.y) * C + cc] = t2;
      else atomicAdd(&out2[cc], t2);
    }
  }
}
static bool col_sum4_ok(const float* x, int C, int ld) {
  static const bool v1 = getenv("VNR_COLSUM_V1") != nullptr;           // A/B switch
  return !v1 && !(C & 3) && !(ld & 3) && !((size_t)x & 15);
}
static hipError_t launch_col_sum_any(const float* x, int M, int C, int ld, const double* mean, double* out, unsigned* amax, float* fout, hipStream_t s,
                                     const float* yact = nullptr, int act = ACT_IDENTITY, double* out2 = nullptr) {
  int rb = (M + 127) / 128; if (rb > 128) rb = 128; if (rb < 1) rb = 1;
  if (col_sum4_ok(x, C, ld) && (!yact || !((size_t)yact & 15))) {
    // every workgroup ends in one atomic per column: 256 groups contending for the same 256 words cost more than the longer
    // per-thread loops, 64 groups of a 256-column matrix leave three quarters of the CUs without work -- about 256 workgroups
    // per launch, between 32 and 128 row groups (VNR_COLSUM_GROUPS pins the count)
    static const int forced = getenv("VNR_COLSUM_GROUPS") ? atoi(getenv("VNR_COLSUM_GROUPS")) : 0;
    const int cb = (C + 255) / 256;
    int rb4 = forced > 0 ? forced : 256 / cb; if (!forced) { if (rb4 > 128) rb4 = 128; if (rb4 < 32) rb4 = 32; }
    const int rmax = (M + 63) / 64; if (rb4 > rmax) rb4 = rmax; if (rb4 < 1) rb4 = 1;
    double* det = static_cast<double*>(det_scratch(s, (size_t)2 * rb4 * C * sizeof(double)));
    vnr_launch(col_sum4_kernel, dim3(cb, rb4), dim3(256), 0, s, const_cast<float*>(x), M, C, ld, mean, out, amax, fout, yact, act, out2, det);
    if (det) {
      hipError_t e = fout ? launch_det_finish_df(det, rb4, (size_t)C, fout, s) : launch_det_finish_dd(det, rb4, (size_t)C, out, s);
      if (e == hipSuccess && out2) e = launch_det_finish_dd(det + (size_t)rb4 * C, rb4, (size_t)C, out2, s);
      return e;
    }
  } else {
    if (yact) return hipErrorInvalidValue;
    if (out2) return hipErrorNotSupported;
    double* det = static_cast<double*>(det_scratch(s, (size_t)rb * C * sizeof(double)));
    vnr_launch(col_sum_kernel, dim3((C + 63) / 64, rb), dim3(256), 0, s, x, M, C, ld, mean, out, amax, fout, det);
    if (det) return fout ? launch_det_finish_df(det, rb, (size_t)C, fout, s) : launch_det_finish_dd(det, rb, (size_t)C, out, s);
  }
  return hipGetLastError();
}
hipError_t launch_col_sum_amax(const float* x, int M, int C, int ld, const double* mean, double* out, unsigned* amax, hipStream_t s) {
  return launch_col_sum_any(x, M, C, ld, mean, out, amax, nullptr, s);
}
// grad[c] += sum_m x[m][c]  (bias gradients), optional abs-max by-product
hipError_t launch_col_sum_grad(const float* x, int M, int C, int ld, float* grad, unsigned* amax, hipStream_t s) {
  return launch_col_sum_any(x, M, C, ld, nullptr, nullptr, amax, grad, s);
}
// bias gradient of an ACTIVATED layer: dy *= act'(y) in place (y = the layer's output, contiguous [M][C]), then as above.
// Returns hipErrorNotSupported when the 16-byte kernel cannot take the operands (the caller then runs the two passes).
hipError_t launch_col_sum_grad_act(float* dy, const float* y, int act, int M, int C, int ld, float* grad, unsigned* amax, hipStream_t s) {
  if (act == ACT_IDENTITY) return launch_col_sum_any(dy, M, C, ld, nullptr, nullptr, amax, grad, s);
  if (!col_sum4_ok(dy, C, ld) || ((size_t)y & 15)) return hipErrorNotSupported;
  return launch_col_sum_any(dy, M, C, ld, nullptr, nullptr, amax, grad, s, y, act);
}
// sum[c] += sum_m x[m][c] and sumsq[c] += sum_m x[m][c]^2 in one pass (float64); hipErrorNotSupported when the 16-byte kernel cannot take x
hipError_t launch_col_sum2(const float* x, int M, int C, int ld, double* sum, double* sumsq, hipStream_t s) {
  return launch_col_sum_any(x, M, C, ld, nullptr, sum, nullptr, nullptr, s, nullptr, ACT_IDENTITY, sumsq);
}
hipError_t launch_col_sum(const float* x, int M, int C, int ld, const double* mean, double* out, hipStream_t s) {
  return launch_col_sum_amax(x, M, C, ld, mean, out, nullptr, s);
}
__global__ void scale_d_kernel(double* v, int n, double f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] *= f;
}
hipError_t launch_scale_d(double* v, int n, double f, hipStream_t s) {
  vnr_launch(scale_d_kernel, dim3((n + 127) / 128), dim3(128), 0, s, v, n, f);
  return hipGetLastError();
}
// BatchNormalization(training=True) (Keras, TF 2.2, non-fused path for rank-3 inputs): normalise with the batch
// mean / population variance, update the moving statistics with momentum 0.99.
//   scale = gamma * rsqrt(var + eps), shift = beta - mean * scale;  moving = moving * 0.99 + batch * 0.01
// raw: mean / sq hold sum x and sum x^2 (launch_col_sum2); they are rewritten as the mean and the centred sum of squares, which is
// what the backward pass reads (float64: sum x^2 - M mean^2 keeps ~1e-13 of the variance for activations of unit order)
__global__ void bn_train_finish_kernel(double* mean, double* sq, int M, int C, const float* gamma,
                                       const float* beta, float momentum, float* moving_mean, float* moving_var,
                                       float* scale, float* shift, int raw, const unsigned* skip_moving) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (raw) {
    const double m = mean[c] / (double)M;
    double ss = sq[c] - (double)M * m * m;
    if (ss < 0.0) ss = 0.0;
    mean[c] = m; sq[c] = ss;
  }
  const float mu = (float)mean[c], var = (float)(sq[c] / (double)M);
  const float inv = gamma[c] * (1.0f / sqrtf(var + kBnEps));
  scale[c] = inv;
  shift[c] = beta[c] - mu * inv;
  // (the overflow sentinel of the split path has tripped in this call: its statistics may come from non-finite rows, the call will be
  //  repeated on exact fp32 -- the variables must see ONE update, that of the repeat)
  if (skip_moving && *reinterpret_cast<const volatile unsigned*>(skip_moving)) return;
  moving_mean[c] = moving_mean[c] * momentum + mu * (1.0f - momentum);
  moving_var[c] = moving_var[c] * momentum + var * (1.0f - momentum);
}
hipError_t launch_bn_train_finish(const double* mean, const double* sq, int M, int C, const float* gamma, const float* beta,
                                  float momentum, float* moving_mean, float* moving_var, float* scale, float* shift, hipStream_t s, int raw,
                                  const unsigned* skip_moving) {
  vnr_launch(bn_train_finish_kernel, dim3((C + 127) / 128), dim3(128), 0, s, const_cast<double*>(mean), const_cast<double*>(sq), M, C, gamma, beta, momentum,
                     moving_mean, moving_var, scale, shift, raw, skip_moving);
  return hipGetLastError();
}
// ActNormFlow.init (flow.py:189-196): log_scale = log(1 / (std + 1e-8)), bias = -mean / (std + 1e-8) from the
// statistics of ALL rows (padding included); also emits exp(log_scale) for the forward that follows.
__global__ void actnorm_init_finish_kernel(const double* mean, const double* sq, int M, int C, float* log_scale, float* bias,
                                           float* scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mu = (float)mean[c], sd = sqrtf((float)(sq[c] / (double)M));
  const float ls = logf(1.0f / (sd + 1e-8f));
  log_scale[c] = ls;
  bias[c] = -mu / (sd + 1e-8f);
  scale[c] = expf(ls);
}
hipError_t launch_actnorm_init_finish(const double* mean, const double* sq, int M, int C, float* log_scale, float* bias,
                                      float* scale, hipStream_t s) {
  vnr_launch(actnorm_init_finish_kernel, dim3((C + 127) / 128), dim3(128), 0, s, mean, sq, M, C, log_scale, bias, scale);
  return hipGetLastError();
}
// Counter-based dropout mask: keep element i of site `key` iff mix32(i * 0x9E3779B1 + key) >= rate * 2^32
// (murmur3 finaliser).  oracle/vaenar_numpy.py:dropout_keep is the bit-identical NumPy statement, so training-mode
// parity runs with dropout ON.  tf.keras.layers.Dropout: y = x * keep / (1 - rate).
__device__ __forceinline__ unsigned mix32(unsigned k) {
  k ^= k >> 16; k *= 0x85EBCA6Bu; k ^= k >> 13; k *= 0xC2B2AE35u; k ^= k >> 16;
  return k;
}
// y[m][c] = dropout( x[m][c] * scale[c] + shift[c] + pe_w * pe[(m % T)][c] )   (each term optional)
__global__ void rowop_kernel(const float* x, int M, int C, const float* scale, const float* shift, const float* pe, int T,
                             float pe_w, float rate, unsigned key, float* y) {
  const size_t n = (size_t)M * C;
  const unsigned thresh = rate > 0.f ? (unsigned)fminf(rate * 4294967296.0f, 4294967040.0f) : 0u;
  const float keep_scale = rate > 0.f ? 1.0f / (1.0f - rate) : 1.0f;
  if (!(C & 3) && n < 0xffffffffull && !(((size_t)x | (size_t)y | (size_t)scale | (size_t)shift | (size_t)pe) & 15)) {
    // four consecutive elements of one row per thread, 32-bit index arithmetic (the element-wise form below spends its time in a 64-bit
    // division per element); the dropout mask stays a function of the flat element index, as the oracle restates it
    const unsigned n4 = (unsigned)(n >> 2), uC = (unsigned)C, uT = (unsigned)T;
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
      const unsigned i = q * 4u, m = i / uC, c = i - m * uC;
      float4 v = *reinterpret_cast<const float4*>(x + i);
      if (scale) { const float4 a = *reinterpret_cast<const float4*>(scale + c); v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w; }
      if (shift) { const float4 a = *reinterpret_cast<const float4*>(shift + c); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
      if (pe) { const float4 a = *reinterpret_cast<const float4*>(pe + (size_t)(m % uT) * uC + c); v.x += pe_w * a.x; v.y += pe_w * a.y; v.z += pe_w * a.z; v.w += pe_w * a.w; }
      if (rate > 0.f) {
        v.x = (mix32(i * 0x9E3779B1u + key) >= thresh) ? v.x * keep_scale : 0.f;
        v.y = (mix32((i + 1u) * 0x9E3779B1u + key) >= thresh) ? v.y * keep_scale : 0.f;
        v.z = (mix32((i + 2u) * 0x9E3779B1u + key) >= thresh) ? v.z * keep_scale : 0.f;
        v.w = (mix32((i + 3u) * 0x9E3779B1u + key) >= thresh) ? v.w * keep_scale : 0.f;
      }
      *reinterpret_cast<float4*>(y + i) = v;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
    float v = x[i];
    if (scale) v *= scale[c];
    if (shift) v += shift[c];
    if (pe) v += pe_w * pe[(size_t)(m % T) * C + c];
    if (rate > 0.f) v = (mix32((unsigned)i * 0x9E3779B1u + key) >= thresh) ? v * keep_scale : 0.f;
    y[i] = v;
  }
}
hipError_t launch_rowop(const float* x, int M, int C, const float* scale, const float* shift, const float* pe, int T, float pe_w,
                        float rate, unsigned key, float* y, hipStream_t s) {
  const size_t n = (size_t)M * C;
  int blocks = (int)((n + 1023) / 1024); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;      // (4 elements per thread on the 16-byte path)
  vnr_launch(rowop_kernel, dim3(blocks), dim3(256), 0, s, x, M, C, scale, shift, pe, T > 0 ? T : 1, pe_w, rate, key, y);
  return hipGetLastError();
}


// ---- tf.random.normal on the device (prior.py:35, posterior.py:35): Philox-4x32-10 counter-based generator -------------------
// Element block j (4 normals: elements 4j .. 4j+3) uses counter (lo32(j + offset), hi32(j + offset), 0, 0) and key (lo32(seed),
// hi32(seed)); ten rounds with the published multipliers / Weyl constants; the four 32-bit outputs x0..x3 make two Box-Muller
// pairs: u = ((x >> 8) + 0.5) 2^-24 in (0, 1), r = sqrt(-2 ln u_a), theta = 2 pi u_b -> r cos(theta), r sin(theta).  Same
// statement in oracle/vaenar_numpy.py (philox_normal).  TensorFlow's own Philox stream layout is not reproduced (its op-level
// seeding is unavailable without TensorFlow): parity runs inject eps, this generator serves temperature > 0 runs.
// ---- n_sample > 1 (models.py:146-178): the reference tiles the text encoding, targets and lengths n_sample times (sample index inner)
// dst[(b * ns + s) * n + i] = src[b * n + i]   (n 4-byte words per batch element; floats and int32 lengths alike)
__global__ void __launch_bounds__(256) tile_rows_kernel(const uint32_t* __restrict__ src, size_t n, int B, int ns, uint32_t* __restrict__ dst) {
  const size_t total = (size_t)B * ns * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / n, b = row / ns;
    dst[i] = src[b * n + (i - row * n)];
  }
}
hipError_t launch_tile_rows(const void* src, size_t n_words, int B, int ns, void* dst, hipStream_t s) {
  const size_t total = (size_t)B * ns * n_words;
  if (!total) return hipSuccess;
  size_t blocks = (total + 1023) / 1024; if (blocks > 4096) blocks = 4096;
  vnr_launch(tile_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint32_t*>(src), n_words, B, ns, static_cast<uint32_t*>(dst));
  return hipGetLastError();
}
// the gradient of that tiling: dst[b * n + i] (+)= scale * sum_s src[(b * ns + s) * n + i], samples added in index order (deterministic)
__global__ void __launch_bounds__(256) tile_sum_kernel(const float* __restrict__ src, size_t n, int B, int ns, float scale, int accumulate, float* __restrict__ dst) {
  const size_t total = (size_t)B * n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / n, c = i - b * n;
    float acc = 0.f;
    for (int k = 0; k < ns; ++k) acc += src[(b * ns + k) * n + c];
    dst[i] = accumulate ? dst[i] + scale * acc : scale * acc;
  }
}
hipError_t launch_tile_sum(const float* src, size_t n, int B, int ns, float scale, int accumulate, float* dst, hipStream_t s) {
  const size_t total = (size_t)B * n;
  if (!total) return hipSuccess;
  size_t blocks = (total + 1023) / 1024; if (blocks > 4096) blocks = 4096;
  vnr_launch(tile_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n, B, ns, scale, accumulate, dst);
  return hipGetLastError();
}

// ---- BasePosterior.reparameterize / log_probability with nsamples (posterior.py:21-72): row m' = (b * ns + s) * T + t reads mu / logvar
// row b * T + t.  samples = eps * exp(0.5 logvar) + mu (NULL eps = zeros, the `random=False` branch); row_lp[m'] = -0.5 (C log 2pi +
// sum_c (logvar + n^2)) with n = eps when given, else (z - mu) / (exp(0.5 logvar) + epsilon) (posterior.py:59-61).  One wave per row.
__global__ void __launch_bounds__(256)
posterior_rows_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps, const float* __restrict__ zin,
                      int Mt, int ns, int T, int C, float epsilon, float* __restrict__ z_out, float* __restrict__ row_lp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= Mt) return;
  const int bs = row / T, t = row - bs * T, b = bs / ns;
  const size_t o = (size_t)row * C, so = ((size_t)b * T + t) * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float lv = logvar[so + c], m = mu[so + c];
    float nrm;
    if (zin && !eps) nrm = (zin[o + c] - m) / (expf(0.5f * lv) + epsilon);
    else nrm = eps ? eps[o + c] : 0.f;
    if (z_out) z_out[o + c] = nrm * expf(0.5f * lv) + m;
    acc += lv + nrm * nrm;
  }
  acc = wave_sum(acc);
  if (lane == 0 && row_lp) row_lp[row] = -0.5f * ((float)C * 1.8378770664093453f + acc);
}
hipError_t launch_posterior_rows(const float* mu, const float* logvar, const float* eps, const float* zin, int B, int ns, int T, int C,
                                 float epsilon, float* z_out, float* row_lp, hipStream_t s) {
  const int Mt = B * ns * T;
  vnr_launch(posterior_rows_kernel, dim3((Mt + 3) / 4), dim3(256), 0, s, mu, logvar, eps, zin, Mt, ns, T, C, epsilon, z_out, row_lp);
  return hipGetLastError();
}
// out[b] = mean_s x[b * ns + s]   (the means over the samples of models.py:79-83,90)
__global__ void group_mean_kernel(const float* x, int B, int ns, float* out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.f;
  for (int k = 0; k < ns; ++k) acc += x[b * ns + k];
  out[b] = acc / (float)ns;
}
hipError_t launch_group_mean(const float* x, int B, int ns, float* out, hipStream_t s) {
  vnr_launch(group_mean_kernel, dim3((B + 63) / 64), dim3(64), 0, s, x, B, ns, out);
  return hipGetLastError();
}

namespace {
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}
__global__ void __launch_bounds__(256) philox_normal_kernel(float* out, size_t n, unsigned long long seed, unsigned long long offset, float stddev) {
  const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (4 * j >= n) return;
  const unsigned long long ctr = (unsigned long long)j + offset;
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  const float k24 = 1.0f / 16777216.0f;
  const float ua = ((float)(c0 >> 8) + 0.5f) * k24, ub = ((float)(c1 >> 8) + 0.5f) * k24;
  const float uc = ((float)(c2 >> 8) + 0.5f) * k24, ud = ((float)(c3 >> 8) + 0.5f) * k24;
  const float r0 = sqrtf(-2.0f * logf(ua)), r1 = sqrtf(-2.0f * logf(uc));
  float s0, q0, s1, q1;
  sincosf(6.28318530717958647692f * ub, &s0, &q0);
  sincosf(6.28318530717958647692f * ud, &s1, &q1);
  const float z[4] = {stddev * r0 * q0, stddev * r0 * s0, stddev * r1 * q1, stddev * r1 * s1};
  if (4 * j + 3 < n && !(reinterpret_cast<uintptr_t>(out) & 15)) *reinterpret_cast<float4*>(out + 4 * j) = make_float4(z[0], z[1], z[2], z[3]);
  else
    for (int e = 0; e < 4; ++e) if (4 * j + e < n) out[4 * j + e] = z[e];
}
}  // namespace
hipError_t launch_philox_normal(float* out, size_t n, unsigned long long seed, unsigned long long offset, float stddev, hipStream_t s) {
  if (!n) return hipSuccess;
  const size_t blocks = ((n + 3) / 4 + 255) / 256;
  vnr_launch(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, n, seed, offset, stddev);
  return hipGetLastError();
}


// ---- range survey of an activation matrix (engine.hip: range guard of the split-fp16 path) ------------------------------------------
// out[0] = bits of the largest row maximum max_r max_c |x[r][c]| (atomicMax), out[1] = bits of the smallest NON-ZERO row maximum
// (atomicMin; the caller initialises it to 0x7f800000).  Non-negative floats order like their bit patterns; a NaN counts as +inf.
namespace {
__global__ void __launch_bounds__(256) row_range_kernel(const float* x, long long ld, int rows, int cols, unsigned* out, int T, long long bs) {
  // Round 6: at most 128 workgroups whose waves stride over the rows (16-byte loads), ONE atomic pair per workgroup.  (One wave and one atomic per row --
  // the round-5 form -- queued 6400 atomics on one word whenever the word still held its initial value: 62 us per call at S1 size,
  // three calls per attention core on the exact mode.)
  __shared__ float wmax[4], wmin[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gmax = 0.f, gmin = INFINITY;                     // largest / smallest non-zero row maximum seen by this wave
  const bool vec = !(cols & 3) && !(ld & 3) && !(bs & 3) && !((size_t)x & 15);
  // four rows per trip: their loads are all issued before the first reduction (a wave walks ~12 rows at S1 size; one dependent
  // load -> shuffle chain per row took 10 us per call)
  const int stride = gridDim.x * 4;
  for (int row0 = blockIdx.x * 4 + wave; row0 < rows; row0 += 4 * stride) {
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec && cols <= 256) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + u * stride;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && 4 * lane < cols) {
          const int bb = T > 0 ? row / T : 0;
          v[u] = *reinterpret_cast<const float4*>(x + (size_t)bb * bs + (size_t)(row - bb * T) * ld + 4 * lane);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        m[u] = (v[u].x != v[u].x || v[u].y != v[u].y || v[u].z != v[u].z || v[u].w != v[u].w) ? INFINITY : a;
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + u * stride;
        if (row >= rows) continue;
        // T > 0: a batch of matrices -- row r is row r % T of element r / T, elements bs floats apart (the attention operands of one call)
        const int bb = T > 0 ? row / T : 0;
        const float* p = x + (size_t)bb * bs + (size_t)(row - bb * T) * ld;
        for (int c = lane; c < cols; c += 64) { const float v = p[c]; m[u] = fmaxf(m[u], (v != v) ? INFINITY : fabsf(v)); }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float mm = m[u];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
      gmax = fmaxf(gmax, mm);
      if (mm != 0.f && row0 + u * stride < rows) gmin = fminf(gmin, mm);
    }
  }
  if (lane == 0) { wmax[wave] = gmax; wmin[wave] = gmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float mn = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
    const unsigned b = __float_as_uint(mx);
    if (b > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, b);
    const unsigned bn = __float_as_uint(mn);
    if (mn != INFINITY && bn < __hip_atomic_load(out + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(out + 1, bn);
  }
}
// a table of small device-to-device copies in ONE launch (engine.hip: the BatchNormalization moving statistics are saved in front of a
// call that may have to repeat itself, and put back before the repeat)
__global__ void __launch_bounds__(256) copy_batch_kernel(const CopyJob* jobs) {
  const CopyJob j = jobs[blockIdx.y];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < j.n; i += (long long)gridDim.x * blockDim.x) j.dst[i] = j.src[i];
}
// true (word 0 of flag set) when any element of x[0, n) is NaN or infinite: the training step's last look at the flat gradient on the
// exact-fp32 fallback, whose products carry no sentinel of their own (engine.hip, "range sentinel")
__global__ void __launch_bounds__(256) finite_check_kernel(const float* x, size_t n, unsigned* flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    bad |= !(fabsf(v) < INFINITY);
  }
  if (bad) *reinterpret_cast<volatile unsigned*>(flag) = 1u;
}
}  // namespace
hipError_t launch_row_range(const float* x, long long ld, int rows, int cols, unsigned* out, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const unsigned wgs = (unsigned)((rows + 3) / 4);
  vnr_launch(row_range_kernel, dim3(wgs < 128u ? wgs : 128u), dim3(256), 0, s, x, ld, rows, cols, out, 0, (long long)0);
  return hipGetLastError();
}
// the same over a batch of B matrices of T rows each, `bs` floats apart: ONE record for the whole operand
hipError_t launch_row_range_batched(const float* x, long long ld, int T, long long bs, int B, int cols, unsigned* out, hipStream_t s) {
  if (T <= 0 || B <= 0 || cols <= 0) return hipSuccess;
  const long long rows = (long long)T * B;
  if (rows > 0x7fffffffLL) return hipErrorInvalidValue;
  const unsigned wgs = (unsigned)((rows + 3) / 4);
  vnr_launch(row_range_kernel, dim3(wgs < 128u ? wgs : 128u), dim3(256), 0, s, x, ld, (int)rows, cols, out, T, bs);
  return hipGetLastError();
}
hipError_t launch_copy_batch(const CopyJob* jobs, int njobs, hipStream_t s) {
  if (njobs <= 0) return hipSuccess;
  vnr_launch(copy_batch_kernel, dim3(4, njobs), dim3(256), 0, s, jobs);
  return hipGetLastError();
}
hipError_t launch_finite_check(const float* x, size_t n, unsigned* flag, hipStream_t s) {
  if (!n) return hipSuccess;
  vnr_launch(finite_check_kernel, dim3(1024), dim3(256), 0, s, x, n, flag);
  return hipGetLastError();
}

}  // namespace vnr
