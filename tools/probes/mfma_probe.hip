// Standalone probe: what limits the fp32 MFMA main loop on MI355X?  (built & run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// V0/V1: pure MFMA, NACC independent accumulators, ITERS*16 MFMAs per wave
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16 / NACC; ++s)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
  }
  float s = 0; for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V2: MFMA fed by ds_read_b128 from LDS (like the GEMM inner loop), wave tile MIxNI blocks, no global traffic
template <int MI, int NI>
__global__ void __launch_bounds__(256) k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  f32x16 acc[MI][NI];
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(sm + i * 1024 + l31 * 32 + (((2 * c8 + half) ^ ((l31 >> 1) & 7)) << 2));
#pragma unroll
      for (int j = 0; j < NI; ++j) b[j] = *(const f32x4*)(sm + 4096 + j * 1024 + l31 * 32 + (((2 * c8 + half) ^ ((l31 >> 1) & 7)) << 2));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
  }
  float s = 0; for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 256;   // x16 MFMAs (k_mfma) ; x (16*MI*NI) MFMAs (k_lds)
  for (int grid : {256, 512, 1024, 2048}) {
    auto rep = [&](const char* name, float ms, double mfmas_per_wave) {
      double flops = (double)grid * 4 * mfmas_per_wave * 4096.0;
      printf("%-28s grid %5d  %8.1f us  %7.1f TFLOP/s\n", name, grid, ms * 1e3, flops / (ms * 1e-3) / 1e12);
    };
    rep("mfma 1 acc", timeit([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, out, iters); }, 5), iters * 16.0);
    rep("mfma 4 acc", timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters); }, 5), iters * 16.0);
    rep("lds-fed 1x1", timeit([&] { hipLaunchKernelGGL((k_lds<1, 1>), dim3(grid), dim3(256), 0, 0, out, iters); }, 5), iters * 16.0);
    rep("lds-fed 1x2", timeit([&] { hipLaunchKernelGGL((k_lds<1, 2>), dim3(grid), dim3(256), 0, 0, out, iters); }, 5), iters * 32.0);
    rep("lds-fed 2x2", timeit([&] { hipLaunchKernelGGL((k_lds<2, 2>), dim3(grid), dim3(256), 0, 0, out, iters); }, 5), iters * 64.0);
  }
  return 0;
}
