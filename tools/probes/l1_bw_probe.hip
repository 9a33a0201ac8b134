// Per-CU vector-memory return bandwidth probe (gfx950): 8 waves per workgroup, one workgroup per CU, every wave issues
// buffer_load_dwordx4 (1 KiB per wave instruction) back to back, `depth` loads in flight, from (a) one L1-resident 4 KiB tile,
// (b) a 256 KiB image shared by all workgroups (L2-resident), (c) the same with 512 threads / 1024 threads.
// Prints bytes per clock per CU (s_memtime ticks).  Build: hipcc --offload-arch=gfx950 -O3 l1_bw_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DEPTH>
__global__ void __launch_bounds__(1024) k(const float* w, int span_tiles, int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 0x7fffffff, 0x00020000);
  f4 r[DEPTH];
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  int t = wave;                                    // tile index (4 KiB per tile = 4 pieces of 1 KiB)
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { r[d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), (t % span_tiles) * 4096 + (d & 3) * 1024, 0)); if ((d & 3) == 3) t += 8; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc += r[d];
      r[d] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16), (t % span_tiles) * 4096 + (d & 3) * 1024, 0));
      if ((d & 3) == 3) t += 8;
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc += r[d];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (acc[0] == 123.456f) sink[0] = acc[1];
}
int main() {
  const size_t bytes = 64u << 20;
  float* w; unsigned long long* out; float* sink;
  hipMalloc(&w, bytes); hipMemset(w, 0, bytes); hipMalloc(&out, 4096 * 16 * 8); hipMalloc(&sink, 16);
  const int iters = 200;
  struct Case { const char* name; int span; int threads; int wgs; } cases[] = {
    {"L1-resident 32 KiB, 8 waves, 256 WGs", 8, 512, 256}, {"L1-resident, 8 waves, 1 WG", 8, 512, 1},
    {"L2 256 KiB image, 8 waves, 200 WGs", 64, 512, 200}, {"L2 1 MiB image, 8 waves, 200 WGs", 256, 512, 200},
    {"L2 1 MiB image, 16 waves, 200 WGs", 256, 1024, 200}, {"L2 1 MiB image, 4 waves, 200 WGs", 256, 256, 200},
    {"L2 1 MiB image, 8 waves, 100 WGs", 256, 512, 100}, {"MALL 32 MiB image, 8 waves, 200 WGs", 8192, 512, 200}};
  for (auto& c : cases) for (int depth : {4, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (depth == 4) k<4><<<c.wgs, c.threads>>>(w, c.span, iters * 4, out, sink); else k<16><<<c.wgs, c.threads>>>(w, c.span, iters, out, sink);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(c.wgs * 16); hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = c.threads / 64; double sum = 0; unsigned long long mx = 0;
    for (int b = 0; b < c.wgs; ++b) for (int v = 0; v < nw; ++v) { sum += h[b * 16 + v]; if (h[b * 16 + v] > mx) mx = h[b * 16 + v]; }
    const double loads = (double)(depth == 4 ? iters * 4 * 4 : iters * 16) + depth;
    const double avg = sum / (c.wgs * nw);
    printf("%-40s depth %2d: %.1f B/clk/CU (avg wave), %.1f (slowest wave); %.0f clk per 1 KiB load per wave\n", c.name, depth,
           loads * 1024 * nw / avg, loads * 1024 * nw / (double)mx, avg / loads);
  }
  return 0;
}
