// Griffin-Lim vocoder step that follows the text->mel path, gfx950.
//
// Replaces reference audio/audio.py: inv_mel_spectrogram :81-84 = _denormalize :206-216 -> +ref_level_db -> _db_to_amp :189-191 ->
// _mel_to_linear :166-174 (pinv(mel basis) . mel, floor 1e-10) -> ** power -> _griffin_lim :95-102 (istft of S.e^{j.phase0}, then
// griffin_lim_iters times: phase = angle(stft(y)); y = istft(S.e^{j.phase})) with librosa 0.8.0's stft / istft semantics
// (periodic Hann of win_length zero-padded symmetrically to n_fft, center=True reflect padding, overlap-add divided by the window
// sum of squares where it exceeds float32 tiny, n_fft/2 samples cropped at both ends) -- restated in oracle/audio_numpy.py.
//
// One fused kernel per Griffin-Lim iteration, one workgroup per (utterance, pair of frames):
//   * the time signal is never materialised between iterations.  An iteration keeps only the windowed inverse-FFT frames
//     fr[b][t][win_length] (the padded window is zero outside its middle win_length samples, so a frame touches only those);
//     frame t of the next iteration rebuilds the win_length samples it needs by GATHERING the <= ceil(win/hop)+1 overlapping
//     frames of the previous iteration, summing them in frame order (deterministic: no float atomics) and dividing by the window
//     sum of squares computed in the same loop; center=True's reflect padding is an index reflection in that gather;
//   * forward FFT (n_fft = 2048 = 8.8.8.4: Stockham passes with 8 points per thread in registers, 4 LDS round trips; fp32,
//     twiddles from a table computed in float64), phase
//     normalisation X/|X| (angle(0) = 0 -> 1), scaling by the target magnitude, Hermitian extension, inverse FFT, window -> fr.
// HBM traffic per iteration: S once (4.1 KB per frame) + the frame buffers (4 KB written, ~5 x 4 KB gathered, mostly L2 hits):
// bound by the LDS round trips and barriers of the two FFTs, not by HBM.
#include "common.h"
#include <math.h>
#include <vector>

namespace vnr {

namespace {
constexpr int kNfft = 2048, kLog = 11, kHalf = kNfft / 2, kBins = kHalf + 1;
struct VocArgs {
  const float* S;            // [B][T][kBins] target magnitudes
  const float* ang0;         // first pass: initial phases [B][T][kBins] (radians)
  const float* fr_prev;      // [B][T][win] windowed inverse-FFT frames of the previous pass (null on the first pass)
  float* fr_next;            // [B][T][win]
  const int32_t* frames;     // [B] frames per utterance or null (= T)
  const float2* tw;          // exp(-2 pi j m / kNfft), m < kHalf
  const float* window;       // periodic Hann [win]
  int B, T, hop, win;
};

// LDS index padding: one float2 of padding per 32 (256 bytes).  The autosort writes of the first two radix-8 passes stride by
// 8 p float2 across lanes (p = 1, 8): unpadded that is an 8-way bank conflict on every store.
__device__ __forceinline__ int pidx(int i) { return i + (i >> 5); }
constexpr int kPadded = kNfft + kNfft / 32;
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// twiddle exp(-/+ 2 pi j idx / kNfft) for idx in [0, kNfft) from the half-circle table (idx >= kNfft/2: negate)
template <bool INV>
__device__ __forceinline__ float2 twid(const float2* tw, int idx) {
  float2 w = tw[idx & (kHalf - 1)];
  if (idx & kHalf) { w.x = -w.x; w.y = -w.y; }
  if (INV) w.y = -w.y;
  return w;
}
#define VNR_BF(a, b) { const float2 t_ = a; a = make_float2(t_.x + b.x, t_.y + b.y); b = make_float2(t_.x - b.x, t_.y - b.y); }
// multiply by -j (forward) / +j (inverse)
template <bool INV> __device__ __forceinline__ float2 mul_mj(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }
// 8-point DFT in registers (decimation in frequency); results leave in bit-reversed order: X[k] = u[rev3(k)]
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&u)[8]) {
  const float h = 0.70710678118654752440f;
  VNR_BF(u[0], u[4]); VNR_BF(u[1], u[5]); VNR_BF(u[2], u[6]); VNR_BF(u[3], u[7]);
  // twiddles w8^1, w8^2, w8^3 on the lower half
  { const float2 a = u[5]; u[5] = INV ? make_float2(h * (a.x - a.y), h * (a.x + a.y)) : make_float2(h * (a.x + a.y), h * (a.y - a.x)); }
  u[6] = mul_mj<INV>(u[6]);
  { const float2 a = u[7]; u[7] = INV ? make_float2(-h * (a.x + a.y), h * (a.x - a.y)) : make_float2(h * (a.y - a.x), -h * (a.x + a.y)); }
  VNR_BF(u[0], u[2]); VNR_BF(u[1], u[3]); VNR_BF(u[4], u[6]); VNR_BF(u[5], u[7]);
  u[3] = mul_mj<INV>(u[3]); u[7] = mul_mj<INV>(u[7]);
  VNR_BF(u[0], u[1]); VNR_BF(u[2], u[3]); VNR_BF(u[4], u[5]); VNR_BF(u[6], u[7]);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2 (&u)[4]) {   // bit-reversed outputs: X[0] = u0, X[2] = u1, X[1] = u2, X[3] = u3
  VNR_BF(u[0], u[2]); VNR_BF(u[1], u[3]);
  u[3] = mul_mj<INV>(u[3]);
  VNR_BF(u[0], u[1]); VNR_BF(u[2], u[3]);
}

// in-LDS Stockham FFT of kNfft = 8.8.8.4 complex points by 256 threads: three radix-8 passes (8 points per thread in registers)
// and one radix-4 pass (two butterflies per thread): 4 LDS round trips and barriers instead of the 11 of a radix-2 schedule.
// Pass with radix R after sub-transforms of length p: thread i (< N/R): k = i mod p; inputs x[i + r N/R] times w^(r k) with
// w = exp(-2 pi j / (p R)); outputs y[(i - k) R + k + r' p].  Data in `x`, scratch `y`; returns the buffer holding the result
// (natural order).  INV: conjugate twiddles (unnormalised inverse).
template <bool INV>
__device__ void fft2048(float2* x, const float2* tw, int tid) {
  // in place: every pass reads its points into registers, the workgroup synchronises, then the results are written to their
  // autosort positions of the SAME buffer (16 KB instead of a ping-pong pair: 5 instead of 2 workgroups per CU)
  int p = 1;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int i = tid, k = i & (p - 1), j = ((i - k) << 3) + k;
    float2 u[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = x[pidx(i + r * (kNfft / 8))];
    if (pass > 0) {
      const int step = k * (kNfft / 8 / p);            // r k N / (p R)
#pragma unroll
      for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], twid<INV>(tw, r * step));
    }
    dft8<INV>(u);
    __syncthreads();
    const int rev[8] = {0, 4, 2, 6, 1, 5, 3, 7};
#pragma unroll
    for (int r = 0; r < 8; ++r) x[pidx(j + r * p)] = u[rev[r]];
    __syncthreads();
    p <<= 3;
  }
  // p = 512: radix-4 pass, 512 butterflies (two per thread)
  float2 v[2][4];
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int i = tid + 256 * h2, k = i & (p - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[h2][r] = x[pidx(i + r * (kNfft / 4))];
    const int step = k * (kNfft / 4 / p);
#pragma unroll
    for (int r = 1; r < 4; ++r) v[h2][r] = cmul(v[h2][r], twid<INV>(tw, r * step));
    dft4<INV>(v[h2]);
  }
  __syncthreads();
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    const int i = tid + 256 * h2, k = i & (p - 1), j = ((i - k) << 2) + k;
    x[pidx(j)] = v[h2][0]; x[pidx(j + p)] = v[h2][2]; x[pidx(j + 2 * p)] = v[h2][1]; x[pidx(j + 3 * p)] = v[h2][3];
  }
  __syncthreads();
}

// sample n of the overlap-added, normalised signal in istft coordinates (0 <= n < n_fft + hop (nf - 1)), rebuilt from the frames.
// At most kMaxOv frames overlap a sample (host checks ceil(win / hop) <= kMaxOv); the loop has a fixed trip count and predicated
// loads so that all of a sample's reads are in flight together (a data-dependent loop serialised ~5 L2 round trips per sample).
constexpr int kMaxOv = 8;
template <int NOV>                                   // NOV >= ceil(win / hop): 4 for both reference configurations (1024/256, 800/200)
__device__ __forceinline__ float ola_sample(const float* fr, const float* wsq, int n, int nf, int hop, int win, int lpad) {
  const int t_hi = (n - lpad) / hop;                 // newest frame whose window support starts at or before n (n >= lpad always)
  float v[NOV], w2[NOV];
#pragma unroll
  for (int j = 0; j < NOV; ++j) {
    const int t = t_hi - j, i = n - t * hop - lpad;  // i >= 0 by construction
    const bool ok = t >= 0 && t < nf && i < win;
    v[j] = ok ? fr[(size_t)t * win + i] : 0.f;
    w2[j] = ok ? wsq[i] : 0.f;
  }
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int j = NOV - 1; j >= 0; --j) { num += v[j]; den += w2[j]; }     // frame order (oldest first), like the overlap-add loop
  return den > 1.17549435e-38f ? num / den : num;   // librosa.istft: divide where window_sumsquare > tiny(float32)
}

// One workgroup = TWO consecutive frames (2f, 2f+1) of one utterance, carried through ONE complex FFT pair: z = x0 + j x1 has
// Z[k] = X0[k] + j X1[k], so X0[k] = (Z[k] + conj Z[N-k]) / 2 and X1[k] = (Z[k] - conj Z[N-k]) / (2j); after the phase step the
// two Hermitian spectra are packed as W = Y0 + j Y1 and one inverse transform returns y0 in the real and y1 in the imaginary part.
// Halves the butterfly work per frame (the kernel is bound by vector issue).
template <bool FIRST, int NOV>
__global__ void __launch_bounds__(256) gl_frame_kernel(const VocArgs a) {
  __shared__ float2 buf[kPadded];                      // padded indexing: pidx()
  __shared__ float2 tws[kHalf];
  extern __shared__ float wins[];                     // window [win] | window^2 [win]
  float* wsqs = wins + a.win;
  const int tid = threadIdx.x;
  const int f0 = 2 * blockIdx.x, b = blockIdx.y;
  const int nf = a.frames ? a.frames[b] : a.T;
  if (f0 >= nf) return;
  const bool two = f0 + 1 < nf;                        // the second frame of the pair exists
  const int lpad = (kNfft - a.win) / 2;
  // the target magnitudes (and the initial phases) are requested first: their HBM round trip overlaps everything up to the
  // phase step.  Thread tid owns bins tid + 256 u, u < 5 (bin 1024 = Nyquist: thread 0, u = 4)
  float sm[2][5], th0[2][5];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const size_t row = ((size_t)b * a.T + f0 + g) * kBins;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int k = tid + 256 * u;
      const bool ok = k < kBins && (g == 0 || two);
      sm[g][u] = ok ? fabsf(a.S[row + k]) : 0.f;
      th0[g][u] = (FIRST && ok) ? a.ang0[row + k] : 0.f;
    }
  }
  for (int i = tid; i < kHalf; i += 256) tws[i] = a.tw[i];
  for (int i = tid; i < 2 * a.win; i += 256) wins[i] = a.window[i];
  if (!FIRST) {
    __syncthreads();
    // analysis frames: padded[f hop + i], i in the window's support; padded = reflect-pad(y, n_fft/2), y = istft signal cropped by n_fft/2
    const int L = a.hop * (nf - 1);
    const float* frp = a.fr_prev + (size_t)b * a.T * a.win;
#pragma unroll
    for (int u = 0; u < kNfft / 256; ++u) {
      const int i = tid + 256 * u;
      float v[2] = {0.f, 0.f};
      const int iw = i - lpad;
      if (iw >= 0 && iw < a.win) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (g == 1 && !two) continue;
          int q = (f0 + g) * a.hop + i - kHalf;      // index into y
          if (q < 0) q = -q;
          if (q >= L) q = 2 * (L - 1) - q;
          q = q < 0 ? 0 : q;
          v[g] = wins[iw] * ola_sample<NOV>(frp, wsqs, q + kHalf, nf, a.hop, a.win, lpad);
        }
      }
      buf[pidx(i)] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    fft2048<false>(buf, tws, tid);
  }
  // phase: unit = X / |X| (angle(0) = 0 -> 1); spectrum of the next signal = S . unit, Hermitian; DC and Nyquist are real (irfft
  // ignores their imaginary parts).  Bin k and its mirror N - k are both owned by the thread that owns k: read, synchronise, write.
  float2 wk[5], wm[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int k = tid + 256 * u;
    float2 u0 = make_float2(1.f, 0.f), u1 = make_float2(1.f, 0.f);
    if (k < kBins) {
      if (FIRST) {
        float sn, cs;
        sincosf(th0[0][u], &sn, &cs); u0 = make_float2(cs, sn);
        sincosf(th0[1][u], &sn, &cs); u1 = make_float2(cs, sn);
      } else {
        const float2 zk = buf[pidx(k)], zm = buf[pidx((kNfft - k) & (kNfft - 1))];
        const float2 x0 = make_float2(zk.x + zm.x, zk.y - zm.y);          // 2 X0[k]
        const float2 x1 = make_float2(zk.y + zm.y, zm.x - zk.x);          // 2 X1[k] = -j (Z[k] - conj Z[N-k])
        const float m0 = sqrtf(x0.x * x0.x + x0.y * x0.y), m1 = sqrtf(x1.x * x1.x + x1.y * x1.y);
        if (m0 > 0.f) u0 = make_float2(x0.x / m0, x0.y / m0);
        if (m1 > 0.f) u1 = make_float2(x1.x / m1, x1.y / m1);
      }
    }
    float2 y0 = make_float2(sm[0][u] * u0.x, sm[0][u] * u0.y), y1 = make_float2(sm[1][u] * u1.x, sm[1][u] * u1.y);
    if (k == 0 || k == kHalf) { y0.y = 0.f; y1.y = 0.f; }
    wk[u] = make_float2(y0.x - y1.y, y0.y + y1.x);                         // W[k]   = Y0[k] + j Y1[k]
    wm[u] = make_float2(y0.x + y1.y, y1.x - y0.y);                         // W[N-k] = conj Y0[k] + j conj Y1[k]
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int k = tid + 256 * u;
    if (k < kBins) {
      buf[pidx(k)] = wk[u];
      if (k > 0 && k < kHalf) buf[pidx(kNfft - k)] = wm[u];
    }
  }
  __syncthreads();
  fft2048<true>(buf, tws, tid);
  float* frn = a.fr_next + ((size_t)b * a.T + f0) * a.win;
  for (int i = tid; i < a.win; i += 256) {
    const float2 y = buf[pidx(i + lpad)];
    frn[i] = wins[i] * (y.x * (1.0f / kNfft));
    if (two) frn[a.win + i] = wins[i] * (y.y * (1.0f / kNfft));
  }
}

// final signal: wav[b][q] = overlap-added, normalised signal cropped by n_fft/2 (istft center=True), q < hop (nf - 1); 0 beyond
template <int NOV>
__global__ void __launch_bounds__(256) gl_final_kernel(const float* fr, const float* window, const int32_t* frames, int T, int hop, int win,
                                                         int Lmax, float* wav) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Lmax) return;
  const int nf = frames ? frames[b] : T;
  const int L = hop * (nf - 1);
  float v = 0.f;
  if (q < L) v = ola_sample<NOV>(fr + (size_t)b * T * win, window + win, q + kHalf, nf, hop, win, (kNfft - win) / 2);   // window | window^2
  wav[(size_t)b * Lmax + q] = v;
}

// S[b][t][k] = max(1e-10, sum_m invT[m][k] . 10^((denorm(mel[b][t][m]) + ref) / 20)) ^ power
__global__ void __launch_bounds__(256) mel_to_linear_kernel(const float* mel, const float* invT, int n_mels, int n_freq, float min_db, float ref_db,
                                                              float max_abs, int symmetric, float power, float* S) {
  extern __shared__ float amp[];
  const size_t bt = blockIdx.x;
  for (int m = threadIdx.x; m < n_mels; m += 256) {
    const float x = mel[bt * n_mels + m];
    float d;
    if (symmetric) d = (fminf(fmaxf(x, -max_abs), max_abs) + max_abs) * (-min_db) / (2.f * max_abs) + min_db;   // audio.py:207-211
    else d = fminf(fmaxf(x, 0.f), max_abs) * (-min_db) / max_abs + min_db;                                        // audio.py:212-216
    amp[m] = exp10f((d + ref_db) * 0.05f);                                                                        // audio.py:189-191
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_freq; k += 256) {
    float acc = 0.f;
    for (int m = 0; m < n_mels; ++m) acc += invT[(size_t)m * n_freq + k] * amp[m];
    acc = fmaxf(1e-10f, acc);                                                                                      // audio.py:174
    S[bt * n_freq + k] = (power == 1.5f) ? acc * sqrtf(acc) : powf(acc, power);
  }
}

__global__ void __launch_bounds__(256) uniform_angles_kernel(float* ang, size_t n, unsigned long long seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);        // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  ang[i] = 6.28318530717958647692f * ((float)(z >> 40) * (1.0f / 16777216.0f));    // 2 pi rand()  (audio.py:96)
}

}  // namespace

hipError_t launch_mel_to_linear(const float* mel, const float* invT, int BT, int n_mels, int n_freq, float min_db, float ref_db, float max_abs,
                                int symmetric, float power, float* S, hipStream_t s) {
  if (BT <= 0 || n_mels <= 0 || n_freq <= 0) return hipErrorInvalidValue;
  vnr_launch(mel_to_linear_kernel, dim3(BT), dim3(256), (unsigned)(n_mels * sizeof(float)), s, mel, invT, n_mels, n_freq, min_db, ref_db, max_abs,
             symmetric, power, S);
  return hipGetLastError();
}

hipError_t launch_uniform_angles(float* ang, size_t n, unsigned long long seed, hipStream_t s) {
  vnr_launch(uniform_angles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ang, n, seed);
  return hipGetLastError();
}

// twiddle table exp(-2 pi j m / 2048), m < 1024, and the periodic Hann window: computed in float64 on the host
void voc_tables(int win, std::vector<float>& tw, std::vector<float>& window) {
  tw.resize(2 * kHalf); window.resize(2 * win);      // window | window^2 (as the kernels square it: in fp32)
  for (int m = 0; m < kHalf; ++m) { const double th = -2.0 * M_PI * m / kNfft; tw[2 * m] = (float)cos(th); tw[2 * m + 1] = (float)sin(th); }
  for (int n = 0; n < win; ++n) { window[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / win)); window[win + n] = window[n] * window[n]; }
}

hipError_t launch_gl_pass(const float* S, const float* ang0, const float* fr_prev, float* fr_next, const int32_t* frames, const float* tw,
                          const float* window, int B, int T, int hop, int win, hipStream_t s) {
  VocArgs a;
  a.S = S; a.ang0 = ang0; a.fr_prev = fr_prev; a.fr_next = fr_next; a.frames = frames; a.tw = reinterpret_cast<const float2*>(tw); a.window = window;
  a.B = B; a.T = T; a.hop = hop; a.win = win;
  const unsigned lds = (unsigned)(2 * win * sizeof(float));
  const bool ov4 = (win + hop - 1) / hop <= 4;
  if (ang0) vnr_launch(gl_frame_kernel<true, 4>, dim3((T + 1) / 2, B), dim3(256), lds, s, a);      // (the first pass gathers nothing)
  else if (ov4) vnr_launch(gl_frame_kernel<false, 4>, dim3((T + 1) / 2, B), dim3(256), lds, s, a);
  else vnr_launch(gl_frame_kernel<false, kMaxOv>, dim3((T + 1) / 2, B), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_gl_final(const float* fr, const float* window, const int32_t* frames, int B, int T, int hop, int win, float* wav, hipStream_t s) {
  const int Lmax = hop * (T - 1);
  if (Lmax <= 0) return hipErrorInvalidValue;
  if ((win + hop - 1) / hop <= 4) vnr_launch(gl_final_kernel<4>, dim3((Lmax + 255) / 256, B), dim3(256), 0, s, fr, window, frames, T, hop, win, Lmax, wav);
  else vnr_launch(gl_final_kernel<kMaxOv>, dim3((Lmax + 255) / 256, B), dim3(256), 0, s, fr, window, frames, T, hop, win, Lmax, wav);
  return hipGetLastError();
}

}  // namespace vnr
