// What would wave specialisation buy the chain kernel's k-loop?  Two skeletons of panel_chain_kernel<1>'s inner loop (no epilogues,
// no stage switches), 200 workgroups of 512 threads, 32 activation rows in LDS as split-fp16 k-tiles (4 KiB per k-tile), weights as
// operand-major 1 KiB pieces (4 per 32-column tile per k-tile) streamed from a 2 MiB L2-resident image:
//   A  "as built":   every wave owns 32 output columns: per k-tile 4 ds_read_b128 (activations) + 6 MFMA 32x32x16 + 4 weight loads
//                    straight into registers, 4 k-tiles in flight per wave.
//   B  "specialised": waves 0..3 multiply 64 columns each (per k-tile 4 + 8 ds_read_b128, 12 MFMAs), waves 4..7 DMA the k-tile's
//                    32 KiB of weights into an LDS ring (depth 2 or 3), one workgroup barrier per k-tile.
// Prints s_memtime clocks per k-tile round (median over workgroups).  Build: hipcc --offload-arch=gfx950 -O3 chain_round_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_t;
constexpr int NK = 64;                                   // k-tile rounds per launch
constexpr int IMG_TILES = 64;                            // distinct k-tiles in the weight image (x 8 column tiles x 4 KiB = 2 MiB)

__global__ void __launch_bounds__(512) kA(const char* w, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [8 k-tiles][4 KiB] activation operands
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8 * 4096 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  h8 wr[4][4];
  f16v acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int fk = 0;
  auto fetch = [&](int u) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wr[u][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), ((fk % IMG_TILES) * 8 + wave) * 4096 + i * 1024, 0));
    ++fk;
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) fetch(u);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int kb = 0; kb < NK; kb += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const char* ap = smem + ((kb + u) & 7) * 4096 + lane * 16;
      h8 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t], a[2 * t], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t], a[2 * t + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t + 1], a[2 * t], acc, 0, 0, 0);
      }
      fetch(u);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc[0] == 1234.5f) sink[0] = acc[1];
}

// A with the chain kernel's stage structure: every 8 rounds an "epilogue" (the wave's 32 x 32 outputs converted to split fp16 and
// written into the activation panel, 16 ds_write_b64) and the LDS barrier of a stage boundary; FLAGS bit 0: static priority for the
// younger half (as built), bit 1: a stage descriptor fetched from the kernarg segment by value at every boundary (s_load burst)
struct Desc { int v[38]; };
struct Descs { Desc d[8]; };
template <int FLAGS>
__global__ void __launch_bounds__(512) kA2(const char* w, unsigned long long* out, float* sink, const Descs ds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8 * 4096 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  if ((FLAGS & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  h8 wr[4][4];
  int fk = 0;
  auto fetch = [&](int u) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wr[u][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), ((fk % IMG_TILES) * 8 + wave) * 4096 + i * 1024, 0));
    ++fk;
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) fetch(u);
  float keep = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int stage = 0; stage < NK / 8; ++stage) {
    int extra = 0;
    if (FLAGS & 2) { const Desc dsc = ds.d[stage & 7]; extra = dsc.v[3] + dsc.v[37]; }
    f16v acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < 8; kb += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const char* ap = smem + ((kb + u + ((FLAGS & 8) ? wave : 0)) & 7) * 4096 + lane * 16;
        h8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t], a[2 * t], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t], a[2 * t + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * t + 1], a[2 * t], acc, 0, 0, 0);
        }
        fetch(u);
      }
    }
    // epilogue: relu, split, into k-tile `wave` of the panel (what a hidden FFN stage does)
    if (!(FLAGS & 4)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // everybody has read the panel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float x = fmaxf(acc[4 * q + e] + (float)extra, 0.f); const _Float16 h = (_Float16)x; hi[e] = h; lo[e] = (_Float16)(x - (float)h); }
      *reinterpret_cast<h4*>(smem + wave * 4096 + (q * 64 + lane) * 8 % 2048) = hi;
      *reinterpret_cast<h4*>(smem + wave * 4096 + 2048 + (q * 64 + lane) * 8 % 2048) = lo;
    }
    keep += acc[0];
    if (!(FLAGS & 4)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (keep == 1234.5f) sink[0] = keep;
}

// A2 with the two barriers of a stage boundary replaced by per-tile flags: wave w writes k-tile w of the next stage's input and then
// publishes flag[w] = stage + 1; in the next stage it walks the k-tiles in the rotated order w, w + 1, ... and, before reading tile t,
// waits until flag[t] > stage (checked one round ahead, so the flag read's latency hides under the products).  The write-after-read
// hazard on the panel is covered by alternating between two panels (a wave can only be one stage ahead of the slowest one).
__global__ void __launch_bounds__(512) kA3(const char* w, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 panels][8 k-tiles][4 KiB] | flags [8]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2 * 8 * 4096 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  volatile int* flags = reinterpret_cast<volatile int*>(smem + 2 * 8 * 4096);
  if (threadIdx.x < 8) flags[threadIdx.x] = 0;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  h8 wr[4][4];
  int fk = 0;
  auto fetch = [&](int u) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wr[u][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), ((fk % IMG_TILES) * 8 + wave) * 4096 + i * 1024, 0));
    ++fk;
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) fetch(u);
  float keep = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int stage = 0; stage < NK / 8; ++stage) {
    const char* pin = smem + (stage & 1) * 32768;
    char* pout = smem + ((stage + 1) & 1) * 32768;
    f16v acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int fl = stage;                                      // own tile: already there
#pragma unroll 1
    for (int kb = 0; kb < 8; kb += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = (wave + kb + u) & 7, tn = (wave + kb + u + 1) & 7;
        while (fl < stage) { __builtin_amdgcn_s_sleep(1); fl = flags[t]; }       // tile t published?  (stage 0: the initial fill)
        const char* ap = pin + t * 4096 + lane * 16;
        h8 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
        fl = (kb + u + 1 < 8) ? flags[tn] : stage;       // next round's flag, read one round ahead
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * tt], a[2 * tt], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * tt], a[2 * tt + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][2 * tt + 1], a[2 * tt], acc, 0, 0, 0);
        }
        fetch(u);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float x = fmaxf(acc[4 * q + e], 0.f); const _Float16 h = (_Float16)x; hi[e] = h; lo[e] = (_Float16)(x - (float)h); }
      *reinterpret_cast<h4*>(pout + wave * 4096 + (q * 64 + lane) * 8 % 2048) = hi;
      *reinterpret_cast<h4*>(pout + wave * 4096 + 2048 + (q * 64 + lane) * 8 % 2048) = lo;
    }
    keep += acc[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) flags[wave] = stage + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (keep == 1234.5f) sink[0] = keep;
}

template <int DEPTH>
__global__ void __launch_bounds__(512) kB(const char* w, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [8 k-tiles][4 KiB] activations | ring [DEPTH][32 KiB]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8 * 4096 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  char* ring = smem + 8 * 4096;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave >= 4) {
    // loader wave lw: pieces lw, lw + 4, ... of the 32 one-KiB pieces of a k-tile
    const int lw = wave - 4;
    auto dma = [&](int kt) {
      char* st = ring + (kt % DEPTH) * 32768;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int pc = 4 * x + lw;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)(st + pc * 1024), 16, (unsigned)(lane * 16), (kt % IMG_TILES) * 32768 + pc * 1024, 0, 0);
      }
    };
    for (int kt = 0; kt < DEPTH - 1; ++kt) dma(kt);
#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      // tile kt must be visible after this barrier: everything but the newest DEPTH - 2 tiles has landed
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      barrier();                                         // multipliers start tile kt; the stage of tile kt - 1 is free
      if (kt + DEPTH - 1 < NK) dma(kt + DEPTH - 1); else asm volatile("s_nop 0");
    }
    barrier();
  } else {
    f16v acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      barrier();
      const char* ap = smem + (kt & 7) * 4096 + lane * 16;
      const char* wp = ring + (kt % DEPTH) * 32768 + (2 * wave) * 4096 + lane * 16;
      h8 a[4], b[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) b[j][i] = *reinterpret_cast<const h8*>(wp + j * 4096 + 1024 * i);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][2 * t], a[2 * t], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][2 * t], a[2 * t + 1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j][2 * t + 1], a[2 * t], acc[j], 0, 0, 0);
        }
    }
    barrier();
    if (acc[0][0] + acc[1][0] == 1234.5f) sink[0] = acc[0][1];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
  char* w; unsigned long long* out; float* sink;
  const size_t wbytes = (size_t)IMG_TILES * 32768;
  hipMalloc(&w, wbytes); hipMemset(w, 0, wbytes); hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 16);
  hipFuncSetAttribute((const void*)kB<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kB<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  auto report = [&](const char* name, int wgs) {
    std::vector<unsigned long long> h(wgs * 8); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> per; for (int b = 0; b < wgs; ++b) { unsigned long long m = 0; for (int v = 0; v < 8; ++v) m = std::max(m, h[b * 8 + v]); per.push_back((double)m / NK); }
    std::sort(per.begin(), per.end());
    printf("%-58s %4d workgroups: %.0f clk per k-tile round (median), %.0f (slowest workgroup)\n", name, wgs, per[per.size() / 2], per.back());
  };
  for (int wgs : {200, 8}) {
    for (int rep = 0; rep < 3; ++rep) { kA<<<wgs, 512, 8 * 4096>>>(w, out, sink); hipDeviceSynchronize(); }
    report("A  as built (every wave loads and multiplies)", wgs);
    Descs dd; for (auto& d : dd.d) for (int& v : d.v) v = 0;
    for (int rep = 0; rep < 3; ++rep) { kA2<0><<<wgs, 512, 8 * 4096>>>(w, out, sink, dd); hipDeviceSynchronize(); }
    report("A2 + stage boundaries every 8 rounds (epilogue + 2 barriers)", wgs);
    for (int rep = 0; rep < 3; ++rep) { kA2<1><<<wgs, 512, 8 * 4096>>>(w, out, sink, dd); hipDeviceSynchronize(); }
    report("A2 + s_setprio 1 for waves 4..7", wgs);
    for (int rep = 0; rep < 3; ++rep) { kA2<3><<<wgs, 512, 8 * 4096>>>(w, out, sink, dd); hipDeviceSynchronize(); }
    report("A2 + s_setprio + stage descriptor from the kernarg segment", wgs);
    for (int rep = 0; rep < 3; ++rep) { kA2<4><<<wgs, 512, 8 * 4096>>>(w, out, sink, dd); hipDeviceSynchronize(); }
    report("A2 epilogues but NO barriers (hazards ignored)", wgs);
    for (int rep = 0; rep < 3; ++rep) { kA3<<<wgs, 512, 2 * 8 * 4096 + 64>>>(w, out, sink); hipDeviceSynchronize(); }
    report("A3 boundaries synchronised by per-tile flags, rotated k order", wgs);
    for (int rep = 0; rep < 3; ++rep) { kB<2><<<wgs, 512, 8 * 4096 + 2 * 32768>>>(w, out, sink); hipDeviceSynchronize(); }
    report("B  specialised, LDS ring depth 2 (96 KiB of LDS)", wgs);
    for (int rep = 0; rep < 3; ++rep) { kB<3><<<wgs, 512, 8 * 4096 + 3 * 32768>>>(w, out, sink); hipDeviceSynchronize(); }
    report("B  specialised, LDS ring depth 3 (128 KiB of LDS)", wgs);
  }
  return 0;
}
