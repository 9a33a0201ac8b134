// Griffin-Lim vocoder step that follows the text->mel path, gfx950.
//
// Replaces reference audio/audio.py: inv_mel_spectrogram :81-84 = _denormalize :206-216 -> +ref_level_db -> _db_to_amp :189-191 ->
// _mel_to_linear :166-174 (pinv(mel basis) . mel, floor 1e-10) -> ** power -> _griffin_lim :95-102 (istft of S.e^{j.phase0}, then
// griffin_lim_iters times: phase = angle(stft(y)); y = istft(S.e^{j.phase})) with librosa 0.8.0's stft / istft semantics
// (periodic Hann of win_length zero-padded symmetrically to n_fft, center=True reflect padding, overlap-add divided by the window
// sum of squares where it exceeds float32 tiny, n_fft/2 samples cropped at both ends) -- restated in oracle/audio_numpy.py.
//
// One fused kernel per Griffin-Lim iteration, one workgroup per (utterance, frame):
//   * the time signal is never materialised between iterations.  An iteration keeps only the windowed inverse-FFT frames
//     fr[b][t][win_length] (the padded window is zero outside its middle win_length samples, so a frame touches only those);
//     frame t of the next iteration rebuilds the win_length samples it needs by GATHERING the <= ceil(win/hop)+1 overlapping
//     frames of the previous iteration, summing them in frame order (deterministic: no float atomics) and dividing by the window
//     sum of squares computed in the same loop; center=True's reflect padding is an index reflection in that gather;
//   * forward FFT (n_fft = 2048, radix-2 Stockham in LDS, fp32, twiddles from a table computed in float64), phase
//     normalisation X/|X| (angle(0) = 0 -> 1), scaling by the target magnitude, Hermitian extension, inverse FFT, window -> fr.
// HBM traffic per iteration: S once (4.1 KB per frame) + the frame buffers (4 KB written, ~5 x 4 KB gathered, mostly L2 hits):
// bound by LDS / barrier latency of the 22 butterfly stages, not by HBM.
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

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-LDS Stockham radix-2 FFT of kNfft complex points by 256 threads; data in `x`, scratch `y`; returns the buffer holding the
// result (natural order).  INV: conjugate twiddles (unnormalised inverse).
template <bool INV>
__device__ float2* fft2048(float2* x, float2* y, const float2* tw, int tid) {
#pragma unroll 1
  for (int s = 0; s < kLog; ++s) {
    const int p = 1 << s;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + 256 * u;
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      float2 w = tw[k << (kLog - 1 - s)];
      if (INV) w.y = -w.y;
      const float2 u0 = x[i], u1 = cmul(x[i + kHalf], w);
      y[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      y[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
  }
  return x;
}

// sample n of the overlap-added, normalised signal in istft coordinates (0 <= n < n_fft + hop (nf - 1)), rebuilt from the frames.
// At most kMaxOv frames overlap a sample (host checks ceil(win / hop) <= kMaxOv); the loop has a fixed trip count and predicated
// loads so that all of a sample's reads are in flight together (a data-dependent loop serialised ~5 L2 round trips per sample).
constexpr int kMaxOv = 8;
__device__ __forceinline__ float ola_sample(const float* fr, const float* wsq, int n, int nf, int hop, int win, int lpad) {
  const int t_hi = (n - lpad) / hop;                 // newest frame whose window support starts at or before n (n >= lpad always)
  float v[kMaxOv], w2[kMaxOv];
#pragma unroll
  for (int j = 0; j < kMaxOv; ++j) {
    const int t = t_hi - j, i = n - t * hop - lpad;  // i >= 0 by construction
    const bool ok = t >= 0 && t < nf && i < win;
    v[j] = ok ? fr[(size_t)t * win + i] : 0.f;
    w2[j] = ok ? wsq[i] : 0.f;
  }
  float num = 0.f, den = 0.f;
#pragma unroll
  for (int j = kMaxOv - 1; j >= 0; --j) { num += v[j]; den += w2[j]; }     // frame order (oldest first), like the overlap-add loop
  return den > 1.17549435e-38f ? num / den : num;   // librosa.istft: divide where window_sumsquare > tiny(float32)
}

template <bool FIRST>
__global__ void __launch_bounds__(256) gl_frame_kernel(const VocArgs a) {
  __shared__ float2 bufA[kNfft];
  __shared__ float2 bufB[kNfft];
  __shared__ float2 tws[kHalf];
  __shared__ float wins[kNfft], wsqs[kNfft];          // window and its square (win <= n_fft entries used)
  const int tid = threadIdx.x;
  const int f = blockIdx.x, b = blockIdx.y;
  const int nf = a.frames ? a.frames[b] : a.T;
  if (f >= nf) return;
  const int lpad = (kNfft - a.win) / 2;
  for (int i = tid; i < kHalf; i += 256) tws[i] = a.tw[i];
  for (int i = tid; i < a.win; i += 256) { const float w = a.window[i]; wins[i] = w; wsqs[i] = w * w; }
  if (!FIRST) __syncthreads();
  float2* X;
  if (!FIRST) {
    // analysis frame: padded[f hop + i], i in the window's support; padded = reflect-pad(y, n_fft/2), y = istft signal cropped by n_fft/2
    const int L = a.hop * (nf - 1);
    const float* frp = a.fr_prev + (size_t)b * a.T * a.win;
#pragma unroll
    for (int u = 0; u < kNfft / 256; ++u) {
      const int i = tid + 256 * u;
      float v = 0.f;
      const int iw = i - lpad;
      if (iw >= 0 && iw < a.win) {
        int q = f * a.hop + i - kHalf;               // index into y
        if (q < 0) q = -q;
        if (q >= L) q = 2 * (L - 1) - q;
        q = q < 0 ? 0 : q;
        v = wins[iw] * ola_sample(frp, wsqs, q + kHalf, nf, a.hop, a.win, lpad);
      }
      bufA[i] = make_float2(v, 0.f);
    }
    __syncthreads();
    X = fft2048<false>(bufA, bufB, tws, tid);
  } else {
    __syncthreads();
    X = bufA;
  }
  float2* Y = (X == bufA) ? bufB : bufA;
  // phase: unit = X / |X| (angle(0) = 0 -> 1); spectrum of the next signal = S . unit, Hermitian; DC and Nyquist are real (irfft
  // ignores their imaginary parts)
  const float* Sp = a.S + ((size_t)b * a.T + f) * kBins;
  for (int k = tid; k < kBins; k += 256) {
    float2 unit;
    if (FIRST) {
      const float th = a.ang0[((size_t)b * a.T + f) * kBins + k];
      float sn, cs; sincosf(th, &sn, &cs);
      unit = make_float2(cs, sn);
    } else {
      const float2 x = X[k];
      const float mag = sqrtf(x.x * x.x + x.y * x.y);
      unit = mag > 0.f ? make_float2(x.x / mag, x.y / mag) : make_float2(1.f, 0.f);
    }
    const float s = fabsf(Sp[k]);
    float2 y = make_float2(s * unit.x, s * unit.y);
    if (k == 0 || k == kHalf) y.y = 0.f;
    Y[k] = y;
    if (k > 0 && k < kHalf) Y[kNfft - k] = make_float2(y.x, -y.y);
  }
  __syncthreads();
  float2* other = (Y == bufA) ? bufB : bufA;
  const float2* R = fft2048<true>(Y, other, tws, tid);
  float* frn = a.fr_next + ((size_t)b * a.T + f) * a.win;
  for (int i = tid; i < a.win; i += 256) frn[i] = wins[i] * (R[i + lpad].x * (1.0f / kNfft));
}

// final signal: wav[b][q] = overlap-added, normalised signal cropped by n_fft/2 (istft center=True), q < hop (nf - 1); 0 beyond
__global__ void __launch_bounds__(256) gl_final_kernel(const float* fr, const float* window, const int32_t* frames, int T, int hop, int win,
                                                         int Lmax, float* wav) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Lmax) return;
  const int nf = frames ? frames[b] : T;
  const int L = hop * (nf - 1);
  float v = 0.f;
  if (q < L) v = ola_sample(fr + (size_t)b * T * win, window + win, q + kHalf, nf, hop, win, (kNfft - win) / 2);   // window | window^2
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
  if (ang0) vnr_launch(gl_frame_kernel<true>, dim3(T, B), dim3(256), 0, s, a);
  else vnr_launch(gl_frame_kernel<false>, dim3(T, B), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_gl_final(const float* fr, const float* window, const int32_t* frames, int B, int T, int hop, int win, float* wav, hipStream_t s) {
  const int Lmax = hop * (T - 1);
  if (Lmax <= 0) return hipErrorInvalidValue;
  vnr_launch(gl_final_kernel, dim3((Lmax + 255) / 256, B), dim3(256), 0, s, fr, window, frames, T, hop, win, Lmax, wav);
  return hipGetLastError();
}

}  // namespace vnr
