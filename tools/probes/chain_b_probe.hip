// Round-5 probe for the next chain-kernel generation: does ONE wave per SIMD (4 waves x 64 output columns, up to 512 registers,
// a whole K = 256 stage of weight operands in flight) run the chain kernel's stage sequence faster than the as-built layout
// (8 waves x 32 columns, 4 k-tiles in flight)?  Skeleton of panel_chain_kernel<1>: 32 activation rows in LDS as split-fp16
// k-tiles (4 KiB per k-tile), weights as operand-major 1 KiB pieces streamed from an L2-resident image, a stage = 8 k-tile rounds,
// then an epilogue (outputs converted to split fp16 and written into the other panel, plus EPI dependent LDS round trips that
// stand for the LayerNorm exchanges / image stores of the real epilogues) between two workgroup barriers.
//   NW = 8, DEPTH = 4: as built.      NW = 4, DEPTH = 4 / 6 / 8: one wave per SIMD.
// Prints clocks per stage split into k-loop and epilogue (s_memtime, wave 0), median over workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 chain_b_probe.hip -o chain_b_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int NSTAGE = 16;                               // stages per launch (8 rounds each)
constexpr int IMG_TILES = 64;                            // distinct k-tiles in the weight image (x 8 column tiles x 4 KiB = 2 MiB)

template <int NW, int DEPTH, int EPI, int FLAGS>
__global__ void __launch_bounds__(NW * 64) kS(const char* w, unsigned long long* out, float* sink) {
  constexpr int CT = 8 / NW;                             // 32-column tiles per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 panels][8 k-tiles][4 KiB] | scratch 4 KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2 * 8 * 4096 / 4 + 1024; i += NW * 64) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  float* scratch = reinterpret_cast<float*>(smem + 2 * 8 * 4096);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  h8 wr[DEPTH][CT][4];
  int fk = 0;
  auto fetch = [&](int u) {
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        wr[u][j][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), ((fk % IMG_TILES) * 8 + wave * CT + j) * 4096 + i * 1024, 0));
    ++fk;
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) fetch(u);
  float keep = 0.f;
  unsigned long long t_loop = 0, t_epi = 0;
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int stage = 0; stage < NSTAGE; ++stage) {
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    const char* pin = smem + (stage & 1) * 32768;
    char* pout = smem + ((stage + 1) & 1) * 32768;
    f16v acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const h8*>(pin + lane * 16 + 1024 * i);
    static_assert(8 % DEPTH == 0 || DEPTH == 6, "");
    constexpr int TRIPS = DEPTH == 6 ? 1 : 8 / DEPTH;          // (DEPTH 6: 6 rounds per stage -- same per-round accounting)
    constexpr int ROUNDS = TRIPS * DEPTH;
#pragma unroll 1
    for (int kb = 0; kb < ROUNDS; kb += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const char* ap = pin + ((kb + u + 1) & 7) * 4096 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[(u + 1) & 1][i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t], a[u & 1][2 * t], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t], a[u & 1][2 * t + 1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t + 1], a[u & 1][2 * t], acc[j], 0, 0, 0);
          }
        fetch(u);
      }
    }
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    if (FLAGS & 1) {
      barrier();                                         // everybody has read the input panel
      // "LayerNorm-like" exchange chain: EPI dependent LDS round trips
      float x = acc[0][0];
#pragma unroll 1
      for (int e = 0; e < EPI; ++e) {
        scratch[(lane + 64 * wave + e) & 1023] = x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = x * 1.0001f + scratch[(lane * 7 + e + 64 * wave) & 1023];
      }
      keep += x;
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float xv = fmaxf(acc[j][4 * q + e], 0.f); const _Float16 h = (_Float16)xv; hi[e] = h; lo[e] = (_Float16)(xv - (float)h); }
          *reinterpret_cast<h4*>(pout + (wave * CT + j) * 4096 + (q * 64 + lane) * 8 % 2048) = hi;
          *reinterpret_cast<h4*>(pout + (wave * CT + j) * 4096 + 2048 + (q * 64 + lane) * 8 % 2048) = lo;
        }
      barrier();
    } else {
#pragma unroll
      for (int j = 0; j < CT; ++j) keep += acc[j][0];
    }
    const unsigned long long tc = __builtin_amdgcn_s_memtime();
    t_loop += tb - ta; t_epi += tc - tb;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 3] = t1 - t0; out[(blockIdx.x * 8 + wave) * 3 + 1] = t_loop; out[(blockIdx.x * 8 + wave) * 3 + 2] = t_epi; }
  if (keep == 1234.5f) sink[0] = keep;
}


// Second generation of the 4-wave skeleton: the instruction order of the k-loop is PINNED (sched_barrier after every group: the
// compiler's own schedules of kS<4, ...> range from 545 to 856 clk per round for the same loop), and the refills of the last DEFER
// slots of a stage are not issued in the k-loop -- where every load instruction blocks its wave at the address unit's queue -- but
// in the epilogue, where the vector-memory path is otherwise idle: SPREAD = 0 all at once behind the k-loop, SPREAD = 1 one piece
// per dependent LDS round trip of the epilogue.
template <int EPI, int DEFER, int SPREAD, int ORDER = 0, int SWZ = 0, int LAYOUT = 0>
__global__ void __launch_bounds__(256) kP(const char* w, unsigned long long* out, float* sink) {
  constexpr int NW = 4, CT = 2, DEPTH = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 2 * 8 * 4096 / 4 + 1024; i += NW * 64) reinterpret_cast<float*>(smem)[i] = 0.001f * i;
  __syncthreads();
  float* scratch = reinterpret_cast<float*>(smem + 2 * 8 * 4096);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 0x7fffffff, 0x00020000);
  h8 wr[DEPTH][CT][4];
  int fk = 0;                                            // k-tile index of the NEXT stage's slot 0 (advanced per stage)
  auto piece = [&](int u, int j, int i) {
    // LAYOUT 0: [k-tile][column block] -- the 8 column blocks of a k-tile round are 32 KiB of contiguous memory.
    // LAYOUT 1: the engine's operand-major image, [stage][column block][k-tile]: a wave's blocks are 32 KiB apart, the waves 64 KiB
    const int kt = (fk + u) % IMG_TILES;
    const int tile = LAYOUT == 0 ? kt * 8 + wave * CT + j : (kt >> 3) * 64 + (wave * CT + j) * 8 + (kt & 7);
    wr[u][j][i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), tile * 4096 + i * 1024, 0));
  };
#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) piece(u, j, i);
  fk += 8;
  float keep = 0.f;
  unsigned long long t_loop = 0, t_epi = 0;
  auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
  int nkc = 8, aswc = 8;
  asm volatile("" : "+s"(nkc), "+s"(aswc));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int stage = 0; stage < NSTAGE; ++stage) {
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    const char* pin = smem + (stage & 1) * 32768;
    char* pout = smem + ((stage + 1) & 1) * 32768;
    f16v acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    h8 a[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = *reinterpret_cast<const h8*>(pin + lane * 16 + 1024 * i);
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      if (SWZ) {
        // the real kernel's panel addressing: row l31 * 1024 + (((kt * 8 + c) ^ (l31 & 15)) << 4), tile clamped / panel selected by scalars
        const int l31 = lane & 31, half = lane >> 5;
        int kt = u + 1 + (stage & 1); kt = kt < nkc ? kt : nkc - 1;
        const char* Ap = (kt < aswc) ? pin : pout;
        const int akt = (kt < aswc) ? kt + (stage & 2) : kt - aswc;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[(u + 1) & 1][2 * t] = *reinterpret_cast<const h8*>(Ap + (l31 * 1024 + (((((akt << 3) + 2 * t + half) ^ (l31 & 15)) << 4)) & 32767));
          a[(u + 1) & 1][2 * t + 1] = *reinterpret_cast<const h8*>(Ap + (l31 * 1024 + (((((akt << 3) + 4 + 2 * t + half) ^ (l31 & 15)) << 4)) & 32767));
        }
      } else {
      const char* ap = pin + ((u + 1) & 7) * 4096 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[(u + 1) & 1][i] = *reinterpret_cast<const h8*>(ap + 1024 * i);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ORDER == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t], a[u & 1][2 * t], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t], a[u & 1][2 * t + 1], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, j, 2 * t);
          __builtin_amdgcn_sched_barrier(0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][j][2 * t + 1], a[u & 1][2 * t], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, j, 2 * t + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // consecutive MFMAs on DIFFERENT accumulators; a refill sits behind the MFMA pair that freed its register
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][0][2 * t], a[u & 1][2 * t], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][1][2 * t], a[u & 1][2 * t], acc[1], 0, 0, 0);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][0][2 * t], a[u & 1][2 * t + 1], acc[0], 0, 0, 0);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, 0, 2 * t);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][1][2 * t], a[u & 1][2 * t + 1], acc[1], 0, 0, 0);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, 1, 2 * t);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][0][2 * t + 1], a[u & 1][2 * t], acc[0], 0, 0, 0);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, 0, 2 * t + 1);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[u][1][2 * t + 1], a[u & 1][2 * t], acc[1], 0, 0, 0);
          if (ORDER == 1) __builtin_amdgcn_sched_barrier(0);
          if (u < DEPTH - DEFER) piece(u, 1, 2 * t + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    const unsigned long long tb = __builtin_amdgcn_s_memtime();
    if (SPREAD == 0) {
#pragma unroll
      for (int u = DEPTH - DEFER; u < DEPTH; ++u)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) piece(u, j, i);
      __builtin_amdgcn_sched_barrier(0);
    }
    barrier();
    float x = acc[0][0];
#pragma unroll
    for (int e = 0; e < EPI; ++e) {
      scratch[(lane + 64 * wave + e) & 1023] = x;
      if (SPREAD == 1 && e < DEFER * 8) { __builtin_amdgcn_sched_barrier(0); piece(DEPTH - DEFER + e / 8, (e >> 2) & 1, e & 3); __builtin_amdgcn_sched_barrier(0); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      x = x * 1.0001f + scratch[(lane * 7 + e + 64 * wave) & 1023];
    }
    if (SPREAD == 1) {
#pragma unroll
      for (int e = EPI; e < DEFER * 8; ++e) piece(DEPTH - DEFER + e / 8, (e >> 2) & 1, e & 3);
    }
    keep += x;
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float xv = fmaxf(acc[j][4 * q + e], 0.f); const _Float16 h = (_Float16)xv; hi[e] = h; lo[e] = (_Float16)(xv - (float)h); }
        *reinterpret_cast<h4*>(pout + (wave * CT + j) * 4096 + (q * 64 + lane) * 8 % 2048) = hi;
        *reinterpret_cast<h4*>(pout + (wave * CT + j) * 4096 + 2048 + (q * 64 + lane) * 8 % 2048) = lo;
      }
    barrier();
    fk += 8;
    const unsigned long long tc = __builtin_amdgcn_s_memtime();
    t_loop += tb - ta; t_epi += tc - tb;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 3] = t1 - t0; out[(blockIdx.x * 8 + wave) * 3 + 1] = t_loop; out[(blockIdx.x * 8 + wave) * 3 + 2] = t_epi; }
  if (keep == 1234.5f) sink[0] = keep;
}

template <int EPI, int DEFER, int SPREAD, int ORDER = 0, int SWZ = 0, int LAYOUT = 0>
static void runP(const char* name, const char* w, unsigned long long* out, float* sink) {
  const int lds = 2 * 8 * 4096 + 4096;
  hipFuncSetAttribute((const void*)kP<EPI, DEFER, SPREAD, ORDER, SWZ, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int wgs : {200}) {
    for (int rep = 0; rep < 3; ++rep) { kP<EPI, DEFER, SPREAD, ORDER, SWZ, LAYOUT><<<wgs, 256, lds>>>(w, out, sink); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(wgs * 8 * 3); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> tot, lp, ep;
    for (int b = 0; b < wgs; ++b) {
      unsigned long long m = 0; for (int v = 0; v < 4; ++v) m = std::max(m, h[(b * 8 + v) * 3]);
      tot.push_back((double)m / NSTAGE); lp.push_back((double)h[b * 8 * 3 + 1] / NSTAGE); ep.push_back((double)h[b * 8 * 3 + 2] / NSTAGE);
    }
    std::sort(tot.begin(), tot.end()); std::sort(lp.begin(), lp.end()); std::sort(ep.begin(), ep.end());
    printf("%-54s %3d wgs: %6.0f clk per stage (median; slowest %6.0f) = k-loop %6.0f (%4.0f per round) + epilogue %6.0f\n", name, wgs,
           tot[tot.size() / 2], tot.back(), lp[lp.size() / 2], lp[lp.size() / 2] / 8, ep[ep.size() / 2]);
  }
}

template <int NW, int DEPTH, int EPI, int FLAGS>
static void run(const char* name, const char* w, unsigned long long* out, float* sink) {
  const int lds = 2 * 8 * 4096 + 4096;
  hipFuncSetAttribute((const void*)kS<NW, DEPTH, EPI, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int wgs : {200, 8}) {
    for (int rep = 0; rep < 3; ++rep) { kS<NW, DEPTH, EPI, FLAGS><<<wgs, NW * 64, lds>>>(w, out, sink); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(wgs * 8 * 3); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> tot, lp, ep;
    for (int b = 0; b < wgs; ++b) {
      unsigned long long m = 0; for (int v = 0; v < NW; ++v) m = std::max(m, h[(b * 8 + v) * 3]);
      tot.push_back((double)m / NSTAGE); lp.push_back((double)h[b * 8 * 3 + 1] / NSTAGE); ep.push_back((double)h[b * 8 * 3 + 2] / NSTAGE);
    }
    std::sort(tot.begin(), tot.end()); std::sort(lp.begin(), lp.end()); std::sort(ep.begin(), ep.end());
    const int rounds = DEPTH == 6 ? 6 : 8;
    printf("%-54s %3d wgs: %6.0f clk per stage (median; slowest %6.0f) = k-loop %6.0f (%4.0f per round) + epilogue %6.0f\n", name, wgs,
           tot[tot.size() / 2], tot.back(), lp[lp.size() / 2], lp[lp.size() / 2] / rounds, ep[ep.size() / 2]);
  }
}

int g_rand = 0;
int main(int argc, char** argv) {
  char* w; unsigned long long* out; float* sink;
  const size_t wbytes = (size_t)IMG_TILES * 32768;
  hipMalloc(&w, wbytes); hipMemset(w, 0, wbytes);
  if (argc > 1) {                                       // random fp16 weights in [-1, 1): the data the matrix pipe toggles on
    std::vector<_Float16> hw(wbytes / 2); unsigned x = 12345u;
    for (auto& v : hw) { x = x * 1664525u + 1013904223u; v = (_Float16)(((int)(x >> 8) % 2001 - 1000) * 0.001f); }
    hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice); g_rand = 1;
    printf("== random weights and activations\n");
  } hipMalloc(&out, 256 * 8 * 3 * 8); hipMalloc(&sink, 16);
  printf("-- free-running (no epilogue, no barrier)\n");
  run<8, 4, 0, 0>("8 waves x 32 cols, 4 tiles in flight (as built)", w, out, sink);
  run<4, 4, 0, 0>("4 waves x 64 cols, 4 tiles in flight", w, out, sink);
  run<4, 8, 0, 0>("4 waves x 64 cols, 8 tiles in flight", w, out, sink);
  printf("-- stage boundaries, short epilogue (split stores only)\n");
  run<8, 4, 0, 1>("8 waves x 32 cols, 4 tiles in flight (as built)", w, out, sink);
  run<4, 4, 0, 1>("4 waves x 64 cols, 4 tiles in flight", w, out, sink);
  run<4, 6, 0, 1>("4 waves x 64 cols, 6 tiles in flight (6-round stages)", w, out, sink);
  run<4, 8, 0, 1>("4 waves x 64 cols, 8 tiles in flight", w, out, sink);
  printf("-- stage boundaries, epilogue with 12 dependent LDS round trips\n");
  run<8, 4, 12, 1>("8 waves x 32 cols, 4 tiles in flight (as built)", w, out, sink);
  run<4, 4, 12, 1>("4 waves x 64 cols, 4 tiles in flight", w, out, sink);
  run<4, 8, 12, 1>("4 waves x 64 cols, 8 tiles in flight", w, out, sink);
  printf("-- stage boundaries, epilogue with 32 dependent LDS round trips\n");
  run<8, 4, 32, 1>("8 waves x 32 cols, 4 tiles in flight (as built)", w, out, sink);
  run<4, 4, 32, 1>("4 waves x 64 cols, 4 tiles in flight", w, out, sink);
  run<4, 8, 32, 1>("4 waves x 64 cols, 8 tiles in flight", w, out, sink);
  printf("-- 4 waves x 64 cols, 8 tiles in flight, PINNED k-loop order; DEFER = refills of the last slots issued in the epilogue\n");
  runP<0, 0, 0>("EPI  0, all refills in the loop", w, out, sink);
  runP<0, 2, 0>("EPI  0, 2 slots deferred, burst behind the loop", w, out, sink);
  runP<0, 4, 0>("EPI  0, 4 slots deferred, burst behind the loop", w, out, sink);
  runP<12, 0, 0>("EPI 12, all refills in the loop", w, out, sink);
  runP<12, 2, 0>("EPI 12, 2 slots deferred, burst", w, out, sink);
  runP<12, 4, 0>("EPI 12, 4 slots deferred, burst", w, out, sink);
  runP<12, 2, 1>("EPI 12, 2 slots deferred, spread over the round trips", w, out, sink);
  runP<12, 4, 1>("EPI 12, 4 slots deferred, spread over the round trips", w, out, sink);
  runP<32, 0, 0>("EPI 32, all refills in the loop", w, out, sink);
  runP<32, 4, 0>("EPI 32, 4 slots deferred, burst", w, out, sink);
  runP<32, 4, 1>("EPI 32, 4 slots deferred, spread over the round trips", w, out, sink);
  runP<32, 6, 1>("EPI 32, 6 slots deferred, spread over the round trips", w, out, sink);
  printf("-- the same, consecutive MFMAs on different accumulators (ORDER 1 pinned, ORDER 2 left to the compiler inside a k16 step)\n");
  runP<0, 0, 0, 1>("EPI  0, ORDER 1", w, out, sink);
  runP<0, 0, 0, 2>("EPI  0, ORDER 2", w, out, sink);
  runP<12, 0, 0, 1>("EPI 12, ORDER 1", w, out, sink);
  runP<12, 0, 0, 2>("EPI 12, ORDER 2", w, out, sink);
  runP<32, 0, 0, 1>("EPI 32, ORDER 1", w, out, sink);
  runP<32, 0, 0, 2>("EPI 32, ORDER 2", w, out, sink);
  runP<0, 4, 0, 1>("EPI  0, ORDER 1, 4 slots deferred (burst): loop floor", w, out, sink);
  printf("-- ORDER 2 with the real kernel's per-tile A addressing (swizzle arithmetic on the VALU, clamps / panel select on the SALU)\n");
  runP<0, 0, 0, 2, 1>("EPI  0, ORDER 2, swizzled A addresses", w, out, sink);
  runP<12, 0, 0, 2, 1>("EPI 12, ORDER 2, swizzled A addresses", w, out, sink);
  printf("-- ORDER 2, weight image in the ENGINE's layout [column block][k-tile] (a k-tile round reads 8 blocks 32 KiB apart) against [k-tile][column block]\n");
  runP<0, 0, 0, 2, 0, 0>("EPI  0, [k-tile][column block] (contiguous round)", w, out, sink);
  runP<0, 0, 0, 2, 0, 1>("EPI  0, [column block][k-tile] (engine layout)", w, out, sink);
  runP<12, 0, 0, 2, 0, 1>("EPI 12, [column block][k-tile] (engine layout)", w, out, sink);
  return 0;
}
