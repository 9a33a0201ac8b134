// Per-CU vector-memory STORE bandwidth probe (gfx950): every wave issues global_store_dwordx4 (1 KiB per wave instruction, each lane
// 16 contiguous bytes, a wave's 64 lanes either one contiguous KiB or 8 rows x 128 B / 32 rows x 32 B pieces like the attention
// kernels) back to back over its own region.  Prints bytes per clock per CU (s_memtime ticks) for several waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(1024) k(float* out, int iters, int pattern, unsigned long long* tsout) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t gw = (size_t)blockIdx.x * nw + wave;
  // pattern 0: contiguous KiB; 1: 8 rows x 128 B (row stride 512 B); 2: 32 rows x 32 B (row stride 512 B)
  size_t loff;
  if (pattern == 0) loff = lane * 4;
  else if (pattern == 1) loff = (size_t)(lane >> 3) * 128 + (lane & 7) * 4;
  else loff = (size_t)(lane >> 1) * 128 + (lane & 1) * 4;
  float* base = out + gw * (size_t)iters * 4096 + loff;       // 16 KiB of floats per iteration group keeps regions disjoint
  f4 v = {1.f, 2.f, 3.f, (float)lane};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    float* p = base + (size_t)i * 4096;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p)); else *reinterpret_cast<f4*>(p) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) tsout[gw] = t1 - t0;
}
int main() {
  const int iters = 64;
  const size_t maxw = 256 * 16;
  float* out; unsigned long long* ts;
  hipMalloc(&out, maxw * iters * 4096 * 4); hipMalloc(&ts, maxw * 8);
  for (int wgs : {256, 32, 8}) for (int threads : {64, 256, 512, 1024}) for (int pattern : {0, 1, 2}) for (int nt : {1, 0}) {
    if (wgs != 256 && (pattern == 2 || threads == 512)) continue;
    const int nw = threads / 64;
    for (int rep = 0; rep < 2; ++rep) {
      if (nt) k<true><<<wgs, threads>>>(out, iters, pattern, ts); else k<false><<<wgs, threads>>>(out, iters, pattern, ts);
      hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(wgs * nw); hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto x : h) sum += x;
    const double avg = sum / h.size();
    printf("%3d CUs, %2d waves/CU, pattern %d (%s), %s: %.1f B/clk/CU; %.0f clk per 1 KiB store per wave; whole grid %.2f TB/s at 2.4 GHz\n", wgs, nw, pattern,
           pattern == 0 ? "1 KiB contiguous" : pattern == 1 ? "8 x 128 B rows" : "32 x 32 B rows", nt ? "nontemporal" : "plain",
           (double)iters * 1024 * nw / avg, avg / iters, (double)iters * 1024 * nw / avg * wgs * 2.4e9 / 1e12);
  }
  return 0;
}
