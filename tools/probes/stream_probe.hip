// Probe: the floor for a small HBM-bound launch with the decoder cross-attention's traffic
// (read Q 6.55 MB + K,V 4.19 MB, write ctx 6.55 MB + alignments 13.11 MB) and nothing else.
// Variants: plain / non-temporal stores, grid sizes, bytes per thread.  Times are per launch, from HIP events
// around a train of back-to-back launches on one stream (what rocprofv3 --kernel-trace sees plus the inter-launch gap).
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe.bin stream_probe.hip && ./stream_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each thread: reads nr f32x4 (grid-strided), writes nw f32x4 (grid-strided)
template <bool NT>
__global__ void __launch_bounds__(256) rw_kernel(const f32x4* __restrict__ in, size_t n_in, f32x4* __restrict__ out, size_t n_out) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = tid; i < n_in; i += nthr) acc += in[i];
  for (size_t i = tid; i < n_out; i += nthr) {
    f32x4 v = acc; v[0] += (float)i;
    if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
__global__ void empty_kernel() {}

int main() {
  const size_t rd = 6553600 + 4194304, wr = 6553600 + 13107200;     // bytes
  f32x4 *in, *out;
  hipMalloc((void**)&in, rd); hipMalloc((void**)&out, wr);
  hipMemset(in, 0, rd); hipMemset(out, 0, wr);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 200;
  auto time_it = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 10; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = 1e3 * ms / iters;
    printf("%-44s %8.2f us/launch  %8.1f GB/s  frac of 8 TB/s %.3f\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
  };
  time_it("empty kernel (launch train gap)", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, 0);
  for (int grid : {448, 896, 2048, 4096, 16384}) {
    char nm[96];
    snprintf(nm, sizeof nm, "read 10.7 MB + write 19.7 MB, grid %d plain", grid);
    time_it(nm, [&] { hipLaunchKernelGGL(rw_kernel<false>, dim3(grid), dim3(256), 0, 0, in, rd / 16, out, wr / 16); }, (double)(rd + wr));
    snprintf(nm, sizeof nm, "read 10.7 MB + write 19.7 MB, grid %d nt", grid);
    time_it(nm, [&] { hipLaunchKernelGGL(rw_kernel<true>, dim3(grid), dim3(256), 0, 0, in, rd / 16, out, wr / 16); }, (double)(rd + wr));
  }
  time_it("write 19.7 MB only, grid 2048 nt", [&] { hipLaunchKernelGGL(rw_kernel<true>, dim3(2048), dim3(256), 0, 0, in, (size_t)0, out, wr / 16); }, (double)wr);
  time_it("write 19.7 MB only, grid 2048 plain", [&] { hipLaunchKernelGGL(rw_kernel<false>, dim3(2048), dim3(256), 0, 0, in, (size_t)0, out, wr / 16); }, (double)wr);
  time_it("read 10.7 MB only, grid 2048", [&] { hipLaunchKernelGGL(rw_kernel<true>, dim3(2048), dim3(256), 0, 0, in, rd / 16, out, (size_t)256); }, (double)rd);
  time_it("hipMemcpyAsync D2D 15 MB (30 MB traffic)", [&] { hipMemcpyAsync(out, out + (15u << 20) / 16, 15u << 20, hipMemcpyDeviceToDevice, 0); }, 2.0 * (15u << 20));
  return 0;
}
