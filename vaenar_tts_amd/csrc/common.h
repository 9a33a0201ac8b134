// Shared declarations of the gfx950 kernels behind libvaenar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hip/hip_ext.h>
#include <map>
#include <tuple>
#include <type_traits>
#include <vector>
#include <utility>

namespace vnr {

// ---- instrumented launch -------------------------------------------------------------------------------------------------
// Every kernel of the library is launched through vnr_launch.  When the engine's profiler has armed the slot (ProfScope in
// engine.hip) the launch goes through hipExtLaunchKernel with a start and a stop event attached to the DISPATCH ITSELF: their
// difference is the kernel's own begin..end on the GPU -- the quantity rocprofv3 --kernel-trace reports -- instead of an event
// pair recorded around the launch, which adds ~2.5 us of packet processing to every kernel.
struct ProfSlot { hipEvent_t e0 = nullptr, e1 = nullptr; bool armed = false, used = false; };
inline thread_local ProfSlot g_prof_slot;
#if defined(__HIPCC__)
#ifdef __HIPCC__
// tanh for the epilogues (PostNet: tf.nn.tanh, modules/utils.py:104): (e - 1) / (e + 1) with e = 2^(2 log2(e) |x|) on the
// transcendental unit (v_exp_f32, v_rcp_f32: 1 ulp each) and the odd Taylor polynomial below |x| = 1/8, where the quotient would
// cancel.  Absolute error <= 1.2e-7 everywhere (libm's tanhf: 6e-8) -- the 64x128 PostNet tiles spent as long in ocml's tanhf
// (~40 instructions per value, 32 values per lane) as in half their k-loop.
__device__ __forceinline__ float fast_tanhf(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(fminf(ax, 20.f) * 2.88539008177792681472f);
  const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  const float x2 = ax * ax;
  const float small = ax * (1.f + x2 * (-0.333333333333f + x2 * (0.133333333333f + x2 * -0.0539682539683f)));
  return (x != x) ? x : copysignf(ax < 0.125f ? small : big, x);       // NaN propagates like tf.nn.tanh (fminf would have clamped it to +-1)
}
#endif

// Opt-in to more than 48 KiB of dynamic LDS.  The attribute belongs to the CURRENT DEVICE's copy of the function, so the
// "already done" record is kept per device (a second engine handle on another device of the same process must opt in too).
// `done` is the call site's own `static int done[kMaxDevices]` (largest size granted so far per device).
constexpr int kMaxDevices = 32;
inline void opt_in_dynamic_lds(const void* fn, int lds, int (&done)[kMaxDevices]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); return; }
  if (done[dev] < lds) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); done[dev] = lds; }
}

template <typename... KArgs, typename Tuple, size_t... I>
inline void vnr_launch_ext(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t s, Tuple& t, std::index_sequence<I...>) {
  void* ptrs[] = {static_cast<void*>(&std::get<I>(t))..., nullptr};
  // a failed launch leaves its error for the caller's hipGetLastError(), like the plain launch
  g_prof_slot.used = hipExtLaunchKernel(reinterpret_cast<const void*>(kernel), grid, block, ptrs, lds, s, g_prof_slot.e0, g_prof_slot.e1, 0) == hipSuccess;
}
template <typename... KArgs, typename... Args>
inline void vnr_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t s, Args&&... args) {
  if (g_prof_slot.armed && !g_prof_slot.used) {
    std::tuple<std::remove_cv_t<KArgs>...> t{static_cast<std::remove_cv_t<KArgs>>(args)...};
    vnr_launch_ext(kernel, grid, block, lds, s, t, std::index_sequence_for<KArgs...>{});
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, s, static_cast<KArgs>(args)...);
  }
}
#endif

// ---- deterministic accumulation (engine option "deterministic": TF_DETERMINISTIC_OPS=1 of the reference's train.py:17-32) --------
// Every kernel of the training step that ends in float / double atomics onto shared words (column sums over row groups, the row splits
// of the kernel-gradient GEMMs ...) can instead leave one PARTIAL per contributing workgroup in a scratch buffer (plain stores); a small
// finish kernel then adds the partials in index order -- a fixed summation order, so two identical steps give identical bits.  The
// launchers look at g_det (set by vnr_train_step around the step); the scratch is per stream and is reused by consecutive launches
// of that stream (the finish kernel of launch i precedes the producer of launch i + 1 in stream order).
struct DetState {
  std::map<hipStream_t, std::pair<void*, size_t>> bufs;
  std::map<hipStream_t, unsigned*> tickets;                // per stream: [2][kDetTickets] arrival / departure counters of the in-kernel ordered finish (gemm_tn3_kernel), all zero between launches
  std::vector<void*> retired;                              // outgrown buffers: possibly still read by kernels in flight; freed once the step has synchronised
  bool alloc_failed = false;                               // a scratch allocation failed: the launcher fell back to float atomics -- the step reports it
  void release_retired() { for (void* p : retired) (void)hipFree(p); retired.clear(); }
  void release() { for (auto& kv : bufs) (void)hipFree(kv.second.first); for (auto& kv : tickets) (void)hipFree(kv.second); release_retired(); bufs.clear(); tickets.clear(); }
};
constexpr int kDetTickets = 1024;
inline thread_local DetState* g_det = nullptr;
// the training step on its exact-fp32 fallback (engine.hip: vnr_context::train_fp32): the kernel-gradient GEMMs take the fp32 MFMA kernel
inline thread_local bool g_train_exact = false;
inline void* det_scratch(hipStream_t s, size_t bytes) {
  if (!g_det) return nullptr;
  auto& b = g_det->bufs[s];
  if (b.second < bytes) {
    void* p = nullptr;
    const size_t cap = bytes < ((size_t)16 << 20) ? ((size_t)16 << 20) : bytes + bytes / 2;
    if (hipMalloc(&p, cap) != hipSuccess) { g_det->alloc_failed = true; return nullptr; }      // (vnr_train_step turns this into VNR_ERR_NOMEM)
    if (b.first) g_det->retired.push_back(b.first);
    b = {p, cap};
  }
  return b.first;
}
// the counters of the in-kernel ordered finish for launches on stream s (zeroed once; every launch leaves them at zero), or null
inline unsigned* det_tickets(hipStream_t s) {
  if (!g_det) return nullptr;
  auto it = g_det->tickets.find(s);
  if (it != g_det->tickets.end()) return it->second;
  unsigned* p = nullptr;
  if (hipMalloc((void**)&p, 2 * kDetTickets * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, 2 * kDetTickets * sizeof(unsigned)) != hipSuccess) return nullptr;
  g_det->tickets[s] = p;
  return p;
}
// out[i] += sum_{p < nparts} part[p * n + i], partials added in index order (i < n)
hipError_t launch_det_finish_dd(const double* part, int nparts, size_t n, double* out, hipStream_t s);
hipError_t launch_det_finish_df(const double* part, int nparts, size_t n, float* out, hipStream_t s);
hipError_t launch_det_finish_ff(const float* part, int nparts, size_t n, float* out, hipStream_t s);
// C[k * ldc + n] += sum_{p < nparts} part[(p * K + k) * N + n]
hipError_t launch_det_finish_2d(const float* part, int nparts, int K, int N, float* C, int ldc, hipStream_t s);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kMaskFill = -4294967296.0f;   // fp32(-2**32 + 1), attention.py:240
constexpr float kLnEps = 1e-3f;               // Keras LayerNormalization default
constexpr float kBnEps = 1e-3f;               // Keras BatchNormalization default

enum { ACT_IDENTITY = 0, ACT_RELU = 1, ACT_TANH = 2 };

// ---- attention operand images (consumed by attention3.hip, produced by the GEMM / chain epilogues) -------------------
// The f16 matrix pipe wants Q, K, V as fp16 hi/lo pairs (3-term split, gemm2.hip).  Splitting them inside the attention
// kernel costs more VALU issue slots than the MFMAs themselves and is repeated by every workgroup that re-reads the same
// K/V, so the PRODUCER of Q / K / V stores the split form once -- and stores it OPERAND-MAJOR, so that every operand
// load of the attention kernel is one fully coalesced 1 KiB wave read (lane l gets bytes [16 l, 16 l + 16)):
//   image = [batch][head][tile of 32 rows (queries / keys)] x 8 KiB;  rows beyond T inside the last tile are never written.
//   Q / K tile: [hi | lo][t = 0..3][lane = 32 g + (row & 31)][8 x fp16: channels d = 16 t + 8 g + 0..7]
//               (lane (row, g) of a 32x32x16 MFMA holds k-slots 8g..8g+7 of step t);
//   V tile:     [hi | lo][tp = 0..1][nb = 0..1][lane = 32 g + (d & 31)][8 x fp16: k-slots e = 0..7], d = 32 nb + (d & 31),
//               k-slot (tp, g, e) = key 16 tp + (e & 3) + 4 g + 8 (e >> 2) of the tile -- the order in which the S^T
//               accumulator of attention3 already holds P, so the B operand of P.V needs no gather.
// mode 0: plain fp32 output.  mode 1: every column is Q/K-type (image at `qk`).  mode 2: every column is V-type (image at
// `vt`).  mode 3: a cross K|V panel -- column blocks of 2*D: the first D columns (K) go to image `qk` + blk * blk_bytes, the
// last D columns (V) to `vt` + blk * blk_bytes, blk = column / (2*D).  mode 4: a self-attention Q|K|V panel of 3*D columns:
// Q to `qk`, K to `qk` + blk_bytes, V to `vt`.
struct AoiDesc {
  int mode = 0;
  int D = 0;                // H * 64
  int T = 1;                // rows per batch element
  int TT = 1;               // tiles per batch element = ceil(T / 32)
  char* qk = nullptr;       // Q/K-type images [block][B][H][TT][8 KiB]
  char* vt = nullptr;       // V-type images   [block][B][H][TT][8 KiB]
  long long blk_bytes = 0;  // B * H * TT * 8192
};
constexpr int kAoiTile = 8192;
#if defined(__HIPCC__)
// ---- x = hi + lo in fp16: THE split of the 3-term products, one statement for every kernel (round 6) ------------------------------------
// The low part must be derived from the STORED high bits.  Written per element -- `h = (_Float16)x; hi = h; lo = (_Float16)(x - (float)h)`
// -- behind a multiply or an fma, the compiler may evaluate `h` twice: fused into the arithmetic for the value it subtracts
// (v_fma_mixlo_f16 / v_fma_mixhi_f16: ONE rounding of the exact result) and as v_cvt of the rounded fp32 value for the value it stores
// (TWO roundings); the two differ by one fp16 ulp whenever the fp32 rounding crosses an fp16 tie (round 5: 1.3e-4 at the mel, every
// test green; profiles/r05_experiments.txt r05i).  Vector-typed conversions give the compiler ONE node for the high part, and it lowers
// them to v_cvt_pk_f16_f32.  tests/test_isa_split_sites.py disassembles every built object and fails on any v_fma_mix{lo,hi}_f16.
typedef float vnr_f2 __attribute__((ext_vector_type(2)));
typedef float vnr_f4 __attribute__((ext_vector_type(4)));
typedef float vnr_f8 __attribute__((ext_vector_type(8)));
typedef _Float16 vnr_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 vnr_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 vnr_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void vnr_split(const vnr_f4 x, vnr_h4& hi, vnr_h4& lo) {
  hi = __builtin_convertvector(x, vnr_h4);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, vnr_f4), vnr_h4);
}
__device__ __forceinline__ void vnr_split(const vnr_f8 x, vnr_h8& hi, vnr_h8& lo) {
  hi = __builtin_convertvector(x, vnr_h8);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, vnr_f8), vnr_h8);
}
__device__ __forceinline__ void vnr_split(const float x, _Float16& hi, _Float16& lo) {      // one value: the same conversion on a pair
  const vnr_f2 xv = {x, x};
  const vnr_h2 h = __builtin_convertvector(xv, vnr_h2);
  const vnr_h2 l = __builtin_convertvector(xv - __builtin_convertvector(h, vnr_f2), vnr_h2);
  hi = h[0]; lo = l[0];
}
typedef _Float16 aoi_h4 __attribute__((ext_vector_type(4)));
// 4 consecutive output columns [col, col+4) of row `row` (col % 4 == 0)
__device__ __forceinline__ void aoi_store4(const AoiDesc& a, int row, int col, const float* v) {
  aoi_h4 hi, lo;
  { const vnr_f4 xs_ = {v[0], v[1], v[2], v[3]}; vnr_split(xs_, hi, lo); }
  int c = col, blk = 0;
  bool is_v = a.mode == 2;
  if (a.mode == 3) {
    blk = col / (2 * a.D);
    c = col - blk * 2 * a.D;
    if (c >= a.D) { is_v = true; c -= a.D; }
  } else if (a.mode == 4) {
    blk = col / a.D;
    c = col - blk * a.D;
    if (blk == 2) { is_v = true; blk = 0; }
  }
  const int b = row / a.T, t = row - b * a.T, r = t & 31;
  const int head = c >> 6, d = c & 63, H = a.D >> 6;
  char* base = (is_v ? a.vt : a.qk) + (size_t)blk * a.blk_bytes + ((size_t)(b * H + head) * a.TT + (t >> 5)) * kAoiTile;
  if (!is_v) {
    char* p = base + (d >> 4) * 1024 + ((((d >> 3) & 1) * 32 + r) << 4) + (d & 7) * 2;
    *reinterpret_cast<aoi_h4*>(p) = hi;
    *reinterpret_cast<aoi_h4*>(p + 4096) = lo;
  } else {
    const int o = r & 15, pos = (o & 3) | ((o & 4) << 1) | ((o & 8) >> 1);     // k-slot order: bits 2 and 3 swapped
    char* p = base + ((r >> 4) & 1) * 2048 + (d >> 5) * 1024 + ((((pos >> 3) * 32) + (d & 31)) << 4) + (pos & 7) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {          // d + e stays inside the same 32-channel block (d % 4 == 0)
      *reinterpret_cast<_Float16*>(p + 16 * e) = hi[e];
      *reinterpret_cast<_Float16*>(p + 4096 + 16 * e) = lo[e];
    }
  }
}
#endif

// C[M,N] = epilogue( A[M,K] . W[K,N] ), W given transposed as Wt[N][K] (k contiguous).
// A is assembled on the fly:
//   plain   : k <  K1 -> A1[m*lda1 + k] ; k >= K1 -> A2[m*lda2 + k-K1]     (tf.concat, K8)
//   conv    : taps > 0, K = taps*conv_C, k = j*conv_C + c ->
//             A1[(b*T + t + j - taps/2)*lda1 + c] or 0 outside [0,T)        ('same' Conv1D, K2)
// Epilogue order: +bias -> act -> *bn_scale+bn_shift -> +pe_w*pe[m % pe_T] -> +residual
//                 -> (optional, separate pass) LayerNorm.
struct GemmArgs {
  const float* A1 = nullptr; int lda1 = 0; int K1 = 0;
  const float* A2 = nullptr; int lda2 = 0;
  const float* Wt = nullptr; int ldw = 0;
  const void* Wsplit = nullptr;     // optional pre-split fp16 image of Wt (split-fp16 MFMA path), see gemm2.hip
  float acc_scale = 1.f;            // 2^-s when the split image was pre-scaled by 2^s
  const unsigned* a_absmax = nullptr;   // split path, A is a gradient: device word with the bits of max |A|; A is pre-scaled to ~2^10 (gemm2.hip)
  const unsigned* a_absmax2 = nullptr; const unsigned* a_absmax3 = nullptr;   // more words of the same kind (A = column blocks with one word each): the largest counts
  const float* bias = nullptr;
  int act = ACT_IDENTITY;
  const float* bn_scale = nullptr; const float* bn_shift = nullptr;
  int bn_first = 0;                 // 1: BN before the activation (Conv1D bn_before_act=True)
  const float* pe = nullptr; int pe_T = 1; float pe_w = 0.f;
  const float* residual = nullptr; int ldr = 0;
  float* C = nullptr; int ldc = 0;
  int M = 0, N = 0, K = 0;
  int taps = 0, conv_T = 1, conv_C = 0;
  // fused LayerNorm epilogue (row-panel kernel, requires N <= 256; wider rows use launch_layer_norm)
  const float* ln_gamma = nullptr; const float* ln_beta = nullptr;
  unsigned long long* dbg_ts = nullptr;   // measurement-only: per-workgroup s_memtime stamps [tiles][8]
  AoiDesc aoi;                      // mode != 0: C is written as an attention operand image (attention3.hip) instead of fp32
  int wide_tiles = 0;               // split path: prefer 64x128 workgroup tiles (fewer, denser workgroups; engine option "gemm_wide_tiles")
  int no_loader_waves = 0;          // the training step: its GEMMs share the GPU with the kernel-gradient stream, where 8-wave workgroups only cost residency
  // "split rows": an activation matrix [M][C] (C % 32 == 0) stored, per row and per 32-channel tile, as 32 x fp16 hi | 32 x fp16 lo
  // (x = hi + lo) -- the SAME 128 bytes per (row, tile) as fp32, so every address of the LDS-DMA (plain and conv tap walker) is
  // unchanged, but the consumer's k-loop needs no fp32 -> (hi, lo) conversion (which outweighed the MFMAs: the same element is
  // converted once per tap and per column tile, 40x for an encoder convolution).  Split path (Wsplit) only.
  int nslice = 0;                   // tile raster: every XCD owns a slice of the N columns (all row tiles of tiles_n / 8 column tiles) instead of
                                    // a run of row tiles -- for a weight panel that does not fit one XCD's L2 (set by launch_gemm2, see gemm2.hip)
  int a_split = 0;                  // A1 (and A2) are split rows
  int c_split = 0;                  // C is written as split rows (ldc = C columns, bytes per row = 4 * ldc)
  unsigned* range_flag = nullptr;   // overflow sentinel of the split path (see range_note below); null = not watched
};

struct AttnArgs {
  const float* Q; int ldq;          // [B,Tq,H*64], row stride ldq
  const float* K; int ldk;          // [B,Tk,H*64]
  const float* V; int ldv;
  const int32_t* q_len;             // [B] or null
  const int32_t* k_len;             // [B] or null
  float* ctx; int ldo;              // [B,Tq,H*64]
  float* ali;                       // [B,H,Tq,Tk] or null
  int B, H, Tq, Tk;
  int causal;
  float temperature;
  // batch strides in floats (rows * ld by default); allow K/V shared panels
  long long q_bs, k_bs, v_bs, o_bs;
  unsigned long long* dbg_ts = nullptr;   // measurement-only: per-workgroup s_memtime stamps [wgs][8]
  // softmax row statistics for a recomputing backward pass (training step, calls without alignments): [B][H][Tq] each,
  // P_ij = exp(s_ij - row_max[i]) * row_linv[i] with s the masked, scaled logit exactly as the kernel formed it; or null
  float* row_max = nullptr;
  float* row_linv = nullptr;
  // per-launch power-of-two operand scales (attention2.hip): device words [3][2] whose first element holds the bits of max |Q|, max |K|, max |V| of this
  // call (launch_row_range_batched); the kernel maps each maximum to ~2^10 before the fp16 hi/lo split and scales the logits / the context back
  // exactly -- fp32 dynamic range for the attention core (exact-fp32 mode of the engine, the training step's fallback).  Null: unscaled.
  const unsigned* qkv_absmax = nullptr;
};

// ---- row-panel chain kernel (gemm3.hip) ---------------------------------------------------------------------------
constexpr int kMaxChainStages = 20;
// Plain 16-byte output store.  (Non-temporal stores were measured: they help for the attention outputs -- the
// end-of-kernel L2 write-back shrinks -- but cost 10 % of the step on GEMM / chain outputs, whose consumers then
// miss the cache.)
#if defined(__HIPCC__)
// The overflow sentinel of the split-fp16 path.  An activation with |x| >= 65520 splits into hi = +-inf, lo = x - hi = -+inf; in the
// consuming product the terms hi.w_hi and lo.w_hi are infinities of opposite sign (or inf.0), so EVERY accumulator of that row comes out
// NaN, whatever the weights -- and an activation function may heal it on the way out (fmaxf(NaN, 0) = 0: a ReLU stage would hand on
// zeros).  Each split product therefore looks at ONE accumulator per row right behind its k-loop and raises the handle's flag word
// (host-pinned: the engine reads it at its synchronisation points, engine.hip "range sentinel").  Rare branch, no state kept.
__device__ __forceinline__ void range_note(unsigned* flag, float acc_elem) {
  const bool bad = !(__builtin_fabsf(acc_elem) < __builtin_inff());
  // wave-uniform branch + SCALAR store (s_store_dword exists on the gfx9 family; tools/probes/sstore_probe.hip): the raise needs no vector
  // register -- with a vector store the one-wave-per-SIMD chain kernel (512 registers in use) spilled.  `flag` is never null (an unwatched
  // launch gets the handle's scrap word): a null test in front would keep the pointer live across the kernel's stage loop.
  if (__builtin_amdgcn_ballot_w64(bad)) {
    const unsigned one = 1u;
    asm volatile("s_store_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(one), "s"(flag) : "memory");
  }
}
__device__ __forceinline__ void out_store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
#endif

struct ChainStage {
  const void* w;            // operand-major split image, positioned at this stage's first 32-column block
  int kt_total;             // k-tiles per column block (block stride = kt_total * 4096 bytes)
  int kt0, nk;              // first k-tile and number of k-tiles consumed by this stage
  int n;                    // output columns (<= 256)
  int a0, a1, asw;          // activation panel a0 for k-tiles [0, asw), a1 for [asw, nk) (re-based to its tile 0)
  int akt0;                 // first k-tile read from panel a0 (a column offset of 32*akt0 into the panel)
  const float* pe; int pe_T; float pe_w;   // + pe_w * pe[(row % pe_T)][0..n) after the activation (positional term) or null
  const float* bias;        // [n] or null
  int act;
  int res;                  // residual panel index or -1
  const float* gamma; const float* beta;   // LayerNorm or null
  int acc_mode;             // 0: plain stage; 1/2/3: FFN second layer over hidden chunks (start / continue / finish+epilogue)
  float* out; int ldo;      // HBM output [M, n] or null
  float* out_pre;           // LayerNorm stages: the value BEFORE normalisation (x + Dense(.)), fp32 [M, n] with row stride ldo, or null --
                            // what the training step's LayerNorm backward needs (train.inc: xblk_chain)
  float* out_stats;         // LayerNorm stages: (mean, 1 / sqrt(var + eps)) of every row, [M][2] fp32, or null (the fused backward chain)
  int out_fmt;              // 0: fp32; 1: `out` is a Q-type attention operand image (AoiDesc mode 1, D = n, n % 64 == 0);
                            // 4: this stage holds columns [aoi_c0, aoi_c0 + n) of a Q|K|V panel of 3*aoi_D columns whose three
                            //    images (aoi_img_bytes each) start at `out` (AoiDesc mode 4)
  int aoi_T;                // out_fmt 1/4: rows per batch element
  int aoi_D, aoi_c0;
  long long aoi_img_bytes;
  int dst;                  // destination panel or -1
  float scale;              // 2^-s of the pre-scaled weight image
  int lds_ln;               // (set by launch_panel_chain) LDS slot of gamma | beta for a LayerNorm stage, -1 otherwise
  int sync_after;           // (set by launch_panel_chain) workgroup barrier at the end of the stage: only where a later stage needs it
};
struct ChainArgs {
  const float* in0; int ld0;   // panel 0 <- in0[M, D]
  const float* in1; int ld1;   // panel 1 <- in1[M, D] (or null)
  int M, D, nstages;
  int rows64;                  // 1: 64-row panels (M/64 workgroups), 0: 32-row panels -- see gemm3.hip
  const float* prm;            // [nstages][bias | gamma | beta][256] fp32: the program's epilogue parameters, zero padded; null = the
                               // kernel fetches bias / gamma / beta of every stage through their own pointers (training: the values
                               // change every step, a cached packed copy would be stale)
  // Fused cross-attention (attention.py:445-447; round 2): before stage `att_stage` runs (<= 0 = none), the workgroup computes the
  // cross-attention of its 32 rows itself -- queries = panel 1 (left there by the query-projection stage), K / V = the block's
  // operand images (one 8 KiB tile per (batch, head, 32 keys), Tk <= 128) -- and writes the context into panel 1: chain B,
  // the attention kernel and chain C of a CrossAttentionBLK become ONE launch (no alignments: prior / posterior blocks).
  int att_stage;
  const char* att_K; const char* att_V;        // image bases of this block: [B][H][ceil(Tk/32)][8 KiB]
  const int32_t* att_qlen; const int32_t* att_klen;
  int att_Tq, att_Tk, att_B;                   // rows per batch element, keys, batch
  float att_temp;
  int att_lds;                                 // (set by launch_panel_chain) byte offset of the merge scratch in LDS
  float* att_ali;                              // alignments [B][H][Tq][Tk] fp32 (attention.py:242-246 `alignments`) or null: the decoder's blocks
                                               // (decoder.py:188-192) leave them from the fused launch too (round 4; att_Tk % 4 == 0)
  // Fused flow coupling (flow.py:223-239; round 4): stage `cpl_stage` (<= 0 = none) is the log_scale | shift head pair of a
  // TransformerCoupling (n = 2 hc columns: log_scale, then shift).  Its epilogue applies the affine coupling to this panel's rows of z
  // instead of storing the heads: zp' = sigmoid(log_scale + 2) * zp + shift on columns [cpl_zp_off, + hc) of cpl_z (row stride cpl_ld),
  // written back in place, and leaves the whole new z (conditioning half | zp') in panel `st.dst` in split form -- the stages behind it
  // are then the NEXT flow step's ActNorm o InvertibleLinear, pre_projection and first Q|K|V (or the decoder's pre_projection): the
  // coupling kernel and the next step's pre-chain launch disappear.
  int cpl_stage;
  float* cpl_z; int cpl_ld, cpl_zp_off, cpl_cond_off;
  int cpl_lds;                                 // (set by launch_panel_chain) byte offset of the shift exchange scratch in LDS
  unsigned long long* dbg_ts;   // measurement only: [wgs][128] s_memtime stamps (start, panels, loop/epilogue per stage; [64 + 8 wave + i]: stage dbg_stage per wave)
  int dbg_stage;
  int seg_T;                    // > 0 (4-wave kernel, M % seg_T == 0): rows per batch element; every element's panels start at its first row
                                // (ceil(seg_T / 32) workgroups per element, the last one short) instead of M / 32 flat panels
  unsigned* pf_progress;        // L2 warming (chain_prefetch.h): device words [8 XCDs][16] the workers publish their stage in; null = no prefetch workgroups
  unsigned pf_epoch;            // this launch's number on that array (monotonic per engine handle)
  int pf_wgs;                   // (set by launch_panel_chain) prefetch workgroups appended to the grid
  int vt_lds;                   // (set by launch_panel_chain, waves4 only) byte offset of the V-stage transpose scratch [4 waves][32][33] fp32 in LDS
  int waves4;                   // 1: the one-wave-per-SIMD kernel (gemm3c.hip: 4 waves x 64 columns, 8 k-tiles in flight; 32-row panels only), 0: panel_chain_kernel
  unsigned* range_flag;         // overflow sentinel of the split path (range_note); NEVER null for a chain launch (the engine always passes its word:
                                // the 512-register kernel has no room for the null test)
  int prio_mode;                // experiment switch (VNR_CHAIN_PRIO): 0 none, 1 static bump for waves 4..7 (default), 2 alternating per k-tile group, 3 per stage
  ChainStage st[kMaxChainStages];
};
// worker workgroups of a chain launch (panels of `rows` rows)
__attribute__((always_inline)) inline __host__ __device__ int chain_workers(const ChainArgs& g, int rows) {
  return g.seg_T > 0 ? (g.M / g.seg_T) * ((g.seg_T + rows - 1) / rows) : (g.M + rows - 1) / rows;
}
hipError_t launch_panel_chain(const ChainArgs& g, hipStream_t s);
hipError_t launch_chain4(const ChainArgs& g, int lds, hipStream_t s);      // gemm3c.hip; called by launch_panel_chain (which validates the program and lays out the LDS)

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s);
bool gemm2_supported(const GemmArgs& g);          // LDS-DMA ring kernel (gemm2.hip) can take it
hipError_t launch_gemm2(const GemmArgs& g, hipStream_t s);
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
bool attention2_supported(const AttnArgs& a);      // balanced DMA-fed kernel (attention2.hip) can take it
hipError_t launch_attention2(const AttnArgs& a, hipStream_t s);
// attention core on producer-split operand images (attention3.hip); alignments only for non-causal calls with Tk <= 128
struct Attn3Args {
  const char* Qi;                              // Q image [B][H][ceil(Tq/32)][8 KiB]
  const char* Ki; const char* Vi;              // K / V images [B][H][ceil(Tk/32)][8 KiB]
  const int32_t* q_len; const int32_t* k_len;  // [B] or null
  float* ctx; int ldo; long long o_bs;         // [B,Tq,H*64] fp32
  float* ali;                                  // [B,H,Tq,Tk] or null
  int B, H, Tq, Tk;
  float temperature;
  int causal = 0;
  unsigned long long* dbg_ts = nullptr;        // measurement only (VNR_ATTN3_TS): [wgs][32] s_memtime stamps of attn3_kernel
};
hipError_t launch_attention3(const Attn3Args& a, hipStream_t s);
// fp32 [rows][cols] -> operand images (tests, op-level entry; the engine's producers write the images directly)
hipError_t launch_aoi_convert(const float* src, int ld, int rows, int cols, const AoiDesc& d, hipStream_t s);
// Griffin-Lim vocoder step after the path (vocoder.hip)
hipError_t launch_mel_to_linear(const float* mel, const float* invT, int BT, int n_mels, int n_freq, float min_db, float ref_db, float max_abs,
                                int symmetric, float power, float* S, hipStream_t s);
hipError_t launch_uniform_angles(float* ang, size_t n, unsigned long long seed, hipStream_t s);
hipError_t launch_gl_pass(const float* S, const float* ang0, const float* fr_prev, float* fr_next, const int32_t* frames, const float* tw,
                          const float* window, int B, int T, int hop, int win, hipStream_t s);
void voc_tables(int win, std::vector<float>& tw, std::vector<float>& window);     // host: twiddles exp(-2 pi j m / 2048) and periodic Hann, from float64
hipError_t launch_gl_final(const float* fr, const float* window, const int32_t* frames, int B, int T, int hop, int win, float* wav, hipStream_t s);
// tf.random.normal replacement: Philox-4x32-10 + Box-Muller, out[i] ~ N(0, stddev^2) (misc.hip)
hipError_t launch_philox_normal(float* out, size_t n, unsigned long long seed, unsigned long long offset, float stddev, hipStream_t s);
hipError_t launch_layer_norm(const float* x, const float* gamma, const float* beta, int rows,
                             int dim, float* y, hipStream_t s);
hipError_t launch_positional_encoding(int T, int dim, float step, float* out, hipStream_t s);
hipError_t launch_transpose(const float* in, int rows, int cols, float* out, int ldo,
                            hipStream_t s);   // out[c*ldo + r] = in[r*cols + c]
// zp <- sigmoid(ls+2)*zp + shift on one half of z; rowsum[m] = sum_c log(scale) (flow.py:223-239)
hipError_t launch_coupling_fwd(const float* heads /*[M,2*half]: log_scale | shift*/, float* z,
                               int M, int half, int zp_off, float* row_logdet, hipStream_t s);
// Wt [N][K] fp32 -> [N][ceil(K/32)][hi x32 | lo x32] fp16 of (w * scale), zero padded
hipError_t launch_split_weights(const float* Wt, int N, int K, float scale, void* out, hipStream_t s);
hipError_t launch_absmax(const float* x, size_t n, unsigned* out, hipStream_t s);   // *out = max(*out, bits(max|x|))
// Wt [N][K] fp32 -> operand-major split image [ceil(N/32)][ceil(K/32)][2 steps][hi|lo][64 lanes][8 fp16] (gemm3.hip)
hipError_t launch_opmajor_weights(const float* Wt, int N, int K, float scale, void* out, hipStream_t s);
hipError_t launch_gather_rows(const float* table, const int32_t* ids, int rows, int dim, float* out, hipStream_t s);
hipError_t launch_gather_rows_split(const float* table, const int32_t* ids, int rows, int dim, float* out, hipStream_t s);   // out = split rows (dim % 32 == 0)
hipError_t launch_coupling_bwd(const float* heads, float* z, int M, int half, int zp_off,
                               float* row_logdet, hipStream_t s);
hipError_t launch_reparam(const float* mu, const float* logvar, const float* eps, int M, int C, float* z,
                          float* row_lp, hipStream_t s);
hipError_t launch_sqerr_rows(const float* rec, int rec_T, const float* tgt, int T, int B, int C, float* rows,
                             hipStream_t s);
// n_sample > 1 (models.py:146-178): dst[(b * ns + s) * n + i] = src[b * n + i] (4-byte words), and the gradient of that tiling
hipError_t launch_tile_rows(const void* src, size_t n_words, int B, int ns, void* dst, hipStream_t s);
hipError_t launch_tile_sum(const float* src, size_t n, int B, int ns, float scale, int accumulate, float* dst, hipStream_t s);
// BasePosterior.reparameterize / log_probability rows with nsamples (posterior.py:21-72); see misc.hip
hipError_t launch_posterior_rows(const float* mu, const float* logvar, const float* eps, const float* zin, int B, int ns, int T, int C,
                                 float epsilon, float* z_out, float* row_lp, hipStream_t s);
hipError_t launch_group_mean(const float* x, int B, int ns, float* out, hipStream_t s);
hipError_t launch_elbo_scalars(const float* sum_out, const float* sum_init, const int32_t* mel_len,
                               const float* pred_len, const float* post_lp, const float* prior_lp, int B,
                               float* l2, float* length_l2, float* kl, hipStream_t s);
// out[b] = sum_{t < len[b]} rows[b*T + t]   (deterministic order)
hipError_t launch_masked_row_reduce(const float* rows, const int32_t* len, int B, int T,
                                    float scale, float* out, int accumulate, hipStream_t s);
// DenseLengthPredictor: out[b] = sum_{t<len} exp(act(x[b,t,:].w + bias))
hipError_t launch_length_predictor(const float* x, const float* w, const float* bias,
                                   const int32_t* len, int B, int T, int D, int act, float* out,
                                   hipStream_t s);
// out[b] = sum_{t<len[b]} sum_c -0.5*(log(2pi) + eps^2)     (prior.py:36-41)
hipError_t launch_gauss_logprob(const float* eps, const int32_t* len, int B, int T, int C,
                                float* out, hipStream_t s);
// y[b] = a[b] + alpha * float(len[b])
hipError_t launch_axpy_len(float* y, const int32_t* len, float alpha, int B, hipStream_t s);
// folded ActNorm o InvertibleLinear: Wt_out[n][k] = exp(ls[k]) * W[k][n]; b_out[n] = sum_k b[k] W[k][n]
hipError_t launch_fold_actnorm_linear(const float* log_scale, const float* bias, const float* W,
                                      int C, float* Wt_out, float* b_out, hipStream_t s);
// BN inference affine: scale = gamma * rsqrt(var + eps); shift = beta - mean * scale
hipError_t launch_bn_affine(const float* gamma, const float* beta, const float* mean,
                            const float* var, int C, float* scale, float* shift, hipStream_t s);

// training-mode statistics / dropout (misc.hip)
hipError_t launch_col_sum(const float* x, int M, int C, int ld, const double* mean, double* out, hipStream_t s);
hipError_t launch_col_sum2(const float* x, int M, int C, int ld, double* sum, double* sumsq, hipStream_t s);
hipError_t launch_col_sum_amax(const float* x, int M, int C, int ld, const double* mean, double* out, unsigned* amax, hipStream_t s);
hipError_t launch_col_sum_grad(const float* x, int M, int C, int ld, float* grad, unsigned* amax, hipStream_t s);
hipError_t launch_col_sum_grad_act(float* dy, const float* y, int act, int M, int C, int ld, float* grad, unsigned* amax, hipStream_t s);
hipError_t launch_scale_d(double* v, int n, double f, hipStream_t s);
hipError_t launch_bn_train_finish(const double* mean, const double* sq, int M, int C, const float* gamma, const float* beta,
                                  float momentum, float* moving_mean, float* moving_var, float* scale, float* shift, hipStream_t s, int raw = 0,
                                  const unsigned* skip_moving = nullptr);      // skip_moving: the handle's sentinel word -- no moving update once it is set
hipError_t launch_actnorm_init_finish(const double* mean, const double* sq, int M, int C, float* log_scale, float* bias,
                                      float* scale, hipStream_t s);
hipError_t launch_rowop(const float* x, int M, int C, const float* scale, const float* shift, const float* pe, int T, float pe_w,
                        float rate, unsigned key, float* y, hipStream_t s);

// training step: backward / optimizer kernels (train_kernels.hip)
hipError_t launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int K, int N, int T, int shift, hipStream_t s);
hipError_t launch_absmax2d(const float* x, int ld, int rows, int cols, unsigned* out, hipStream_t s);
// out[0] = max over rows of the row maximum of |x|, out[1] = min over rows with a non-zero maximum (float bits; misc.hip)
hipError_t launch_row_range(const float* x, long long ld, int rows, int cols, unsigned* out, hipStream_t s);
hipError_t launch_row_range_batched(const float* x, long long ld, int T, long long bs, int B, int cols, unsigned* out, hipStream_t s);
hipError_t launch_finite_check(const float* x, size_t n, unsigned* flag, hipStream_t s);
struct CopyJob { const float* src; float* dst; long long n; };
hipError_t launch_copy_batch(const CopyJob* jobs, int njobs, hipStream_t s);      // jobs: device table
#if defined(__HIPCC__)
// Publish a candidate maximum (bits of a non-negative float; the word is zero on entry and only grows).  Same-address atomics serialise at
// ~15-20 ns each: one per wave was 1024-4096 per attention-backward launch, 6-25 us of a 40-90 us kernel (profiles/r03_experiments.txt).
// A device-scope look at the word first keeps all but the first few candidates of a launch away from the atomic unit (a stale value only
// costs an unnecessary atomic).
__device__ __forceinline__ void amax_publish(unsigned* dst, float m) {
  const unsigned bits = __float_as_uint(m);
  if (m > 0.f && bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
}
#endif
struct WordList { const unsigned* p[64]; int n; };
hipError_t launch_max_words(const WordList& w, unsigned* out, hipStream_t s);      // *out = max(*out, max_i *w.p[i])
hipError_t launch_gemm_tn_scaled(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int K, int N, int T,
                                 int shift, const unsigned* b_absmax, hipStream_t s);
// several of them as one launch (train_kernels.hip: gemm_tn3_group_kernel; at most kTnGroupMax jobs, each with the geometry and the bits of its own launch)
struct TnCall { const float* A; int lda; const float* B; int ldb; float* C; int ldc; int M, K, N, T, shift; const unsigned* b_absmax; };
constexpr int kTnGroupMax = 8;
hipError_t launch_gemm_tn_group(const TnCall* calls, int n, hipStream_t s);
// recomputing ("flash-style") backward: no stored probabilities, no dS in HBM -- P is rebuilt from Q, K and the row statistics
// the forward call left in row_max / row_linv (AttnArgs), dS lives in registers.  rowdot_ws: scratch [B][H][Tq].
hipError_t launch_attention_bwd_recompute(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                                          const float* dO, int lddo, const float* row_max, const float* row_linv, float* rowdot_ws,
                                          float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv, const int32_t* q_len,
                                          const int32_t* k_len, int B, int H, int Tq, int Tk, int causal, float temperature,
                                          unsigned* amax_slot, hipStream_t s);
hipError_t launch_attention_bwd(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, const float* O, int ldo,
                                const float* dO, int lddo, const float* P, float* dS, float* dQ, int lddq, float* dK, int lddk,
                                float* dV, int lddv, const int32_t* q_len, const int32_t* k_len, int B, int H, int Tq, int Tk,
                                int causal, float temperature, unsigned* amax_slot, hipStream_t s, unsigned* amax_dq = nullptr,
                                unsigned* amax_dk = nullptr, unsigned* amax_dv = nullptr, int amax_ready = 0);   // amax_d*: optional by-products (MFMA kernels only)
hipError_t launch_ln_bwd(const float* v, const float* dy, const float* gamma, int rows, int D, float* dv, float* dgamma, float* dbeta, hipStream_t s);
bool launch_ln_bwd_acc(const float* v, const float* dy, const float* gamma, int rows, int D, float* dst, int lddst, float* dgamma,
                       float* dbeta, hipStream_t s, hipError_t* err);
hipError_t launch_act_bwd(float* d, const float* y, size_t n, int act, hipStream_t s);
hipError_t launch_axpby2d(const float* x, int ldx, float a, float* y, int ldy, int rows, int cols, int accumulate, hipStream_t s);
hipError_t launch_add_d2f(float* g, const double* sum, int n, float a, hipStream_t s);
hipError_t launch_bn_bwd(const float* d, const float* x, const double* mean, const double* sq, const float* gamma, int M, int C,
                         double* s1, double* s2, float* dx, float* dgamma, float* dbeta, hipStream_t s);
hipError_t launch_embed_bwd(const float* d, const int32_t* ids, int M, int C, float* dE, hipStream_t s, int V = 0);   // V: rows of the table (deterministic mode's chunked form)
hipError_t launch_pe_weight_bwd(const float* d, const float* pe, int M, int C, int T, float* out, hipStream_t s);
hipError_t launch_coupling_inv(const float* heads, float* z, int M, int half, int zp_off, float* zp_in, float* rowld, hipStream_t s);
hipError_t launch_coupling_inv_bwd(const float* heads, const float* zp_in, float* dz, const float* g_b, const int32_t* len, int M,
                                   int T, int half, int zp_off, float* dheads, hipStream_t s);
hipError_t launch_actnorm_inv_bwd(const float* x, float* dy, const float* ls, const float* bias, int M, int C, double* s_b, double* s_ls, hipStream_t s);
hipError_t launch_gauss_bwd(const float* eps, const float* g_b, const int32_t* len, int M, int T, int C, float* d, hipStream_t s);
// the same direction of inverse = True flows: the _forward passes (flow.py:223-239, 166-175) -- train_kernels.hip
hipError_t launch_coupling_fwd_bwd(const float* heads, const float* zp_in, float* dz, const float* g_b, const int32_t* len, int M,
                                   int T, int half, int zp_off, float* dheads, hipStream_t s);
hipError_t launch_actnorm_fwd_bwd(const float* x, float* dy, const float* ls, int M, int C, double* s_b, double* s_ls, hipStream_t s);
hipError_t launch_reparam_bwd(const float* dz, const float* eps, const float* logvar, const float* gpost, const int32_t* len, int M,
                              int T, int C, float* dmu, float* dlogvar, hipStream_t s);
hipError_t launch_l2_bwd(const float* rec, int Tr, const float* tgt, int Tm, const int32_t* len, int B, int C, float seed, float* d, hipStream_t s);
hipError_t launch_length_loss(const float* x, const float* w, const float* bias, const int32_t* text_len, const int32_t* mel_len,
                              int B, int T, int D, float seed, float* pred, float* ll, float* dw, float* db, hipStream_t s);
hipError_t launch_adam(float* const* w, const float* const* g, float* const* m, float* const* v, const int64_t* n, int ntensors,
                       float lr_t, float b1, float b2, float eps, hipStream_t s, const unsigned* skip = nullptr);
hipError_t launch_conv_flip(const float* W, int k, int cin, int cout, float* Wb, hipStream_t s);
hipError_t launch_invert(const float* W, int C, float* Winv, float* WinvT, float* logabsdet, hipStream_t s);
hipError_t launch_invert_batch(const float* const* W, float* const* Winv, float* const* WinvT, float* const* lad, int n, int C, hipStream_t s);
hipError_t launch_actnorm_inv_params(const float* ls, const float* bias, int C, float* sc, float* sh, float* lssum, hipStream_t s);
hipError_t launch_axpy_len_dev(float* y, const int32_t* len, const float* alpha, float sign, int B, hipStream_t s);
hipError_t launch_actnorm_fwd_params(const float* ls, int C, float* sc, float* lssum, hipStream_t s);      // sc = exp(ls), *lssum = sum(ls)
hipError_t launch_train_seeds(const float* sum_out, const float* sum_init, const int32_t* mel_len, const float* ll, const float* post_lp,
                              const float* prior_lp, const int32_t* red_len, int B, int Bl, float kw, float lw, float* g_post, float* g_prior,
                              float* cg, float* scalars, hipStream_t s, int part = 0);
hipError_t launch_axpy_dev(float* y, const float* x, const float* cg, float alpha, int n, hipStream_t s);
struct TransposeJobHost { const float* in; float* out; int rows, cols; };      // same layout as the device-side job record
hipError_t launch_transpose_batch(const void* jobs_device, int njobs, hipStream_t s);
struct SplitJobHost { const float* src; int rows, cols; char* dst; int dst_kt = 0; };   // same layout as the device-side job record (dst_kt: k-tiles per row of the
                                                                                       // destination image when the job fills a column band of a wider one; 0 = its own)
hipError_t launch_split_batch(const void* jobs_device, int njobs, float scale, hipStream_t s);
// ---- backward row-panel chains of a CrossAttentionBLK (gemm3b.hip; training step) ---------------------------------------------------
// Every matrix is fp32 row-major with 256 columns unless a leading dimension is given; `*r` are operand-major split images (scale 256)
// of Dense kernels AS STORED [K][N]; column-sum outputs (dg*, db*, dbias*) receive float atomics; amax* are optional device words
// that receive the bits of max |value written| (atomicMax, zero on entry).
struct BwdChainArgs {
  int M;                                  // rows
  int seg;                                // 0: LN3' -> dense2' -> relu' -> dense1' -> LN2' -> att_proj2' ; 1: LN1' -> att_proj1'
  const unsigned* amax_in;                // bits of ~max |dy| (any value within a factor 2^8: it only centres the fp16 range)
  const float* dy; int ld_dy;             // gradient of the segment's LayerNorm output (block output / y)
  // head: dy += sum_i pre_src[i] . W_i^T -- data gradients of Dense layers that READ the LayerNorm output (the next block's self Q, K, V
  // projections of the block output; the cross query projection of y): pre_src[i] = d(their output) [M,256], pre_w[i] = image of their
  // kernel [256][256], amax_pre[i] = bits of max |pre_src[i]| or null
  int npre; const float* pre_src[3]; const void* pre_w[3]; const unsigned* amax_pre[3];
  const float* vA; const float* stA; const float* gA;        // head LayerNorm: its input, (mean, rstd) per row [M][2], gamma
  float* dvA; float* dgA; float* dbA; float* dbiasA; unsigned* amaxA;   // d(input), dgamma, dbeta, bias gradient of the Dense in front
  const void* w2r; const void* w1r; int F;                   // seg 0: dense2 [F][256], dense1 [256][F] images, hidden width
  const float* hdn; float* dh; float* dbias1; unsigned* amax_dh;        // hidden activations [M][F] and their gradient (before dense1)
  const float* vB; const float* stB; const float* gB;        // seg 0: LayerNorm2
  float* dvB; float* dgB; float* dbB; float* dbiasB; unsigned* amaxB;
  const void* pr;                         // att_proj kernel image [512][256]
  float* out0; int acc0;                  // d . W^T[:, 0:256] + d (the residual): gradient of the projection's first input; acc0: add to what is there
  unsigned* amax_out0;                    // by-product: max |out0 as written|
  float* out1; unsigned* amax_out1;       // d . W^T[:, 256:512]: gradient of the attention context
  // Column sums leave the kernel as one row of per-workgroup partials, partial[workgroup][pcols] (plain stores), in the order
  // dgA | dbA | dbiasA (256 each) and, for seg 0, dbias1 (F) | dgB | dbB | dbiasB; launch_bwd_chain adds a small second kernel that sums
  // the rows and adds the totals to the gradients.  (Float atomics from 200 workgroups onto the same 256 words -- the first version --
  // serialised in the L2: 0.2 ms of the launch.)  Scratch of ceil(M / rows per workgroup) * pcols floats.
  float* partial;
};
hipError_t launch_bwd_chain(const BwdChainArgs& g, int rows64, hipStream_t s, hipStream_t finish_stream = nullptr, int part = 0);
inline int bwd_chain_pcols(int seg, int F) { return seg == 0 ? 6 * 256 + F : 3 * 256; }
struct OpmJobHost { const float* src; int N, K; char* dst; };          // same layout as the kernel's job record
hipError_t launch_opmajor_batch(const void* jobs_device, int njobs, float scale, hipStream_t s);

}  // namespace vnr
