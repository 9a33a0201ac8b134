// Round 6 probe: does a SCALAR store (s_store_dword ... glc + s_dcache_wb) from a gfx950 kernel land in host-pinned memory (hipHostMalloc)
// and in device memory, visible to the host after a stream synchronisation?  (csrc/common.h: range_note uses it so that raising the range
// sentinel needs no vector register.)   hipcc --offload-arch=gfx950 -o tools/probes/bin/sstore_probe tools/probes/sstore_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* flag_h, unsigned* flag_d, const float* x, int which) {
  const float v = x[threadIdx.x + blockIdx.x * blockDim.x];
  const bool bad = !(__builtin_fabsf(v) < __builtin_inff());
  if (__builtin_amdgcn_ballot_w64(bad)) {
    const unsigned one = 1u + blockIdx.x;
    unsigned* f = which ? flag_d : flag_h;
    asm volatile("s_store_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(one), "s"(f) : "memory");
  }
}
int main() {
  unsigned *fh = nullptr, *fd = nullptr; float* x = nullptr;
  hipHostMalloc((void**)&fh, 64, hipHostMallocDefault); hipMalloc((void**)&fd, 64); hipMalloc((void**)&x, 4096 * 4);
  float hx[4096]; for (int i = 0; i < 4096; ++i) hx[i] = 1.f;
  for (int trial = 0; trial < 4; ++trial) {
    const int which = trial & 1, poison = trial >> 1;
    fh[0] = 0; hipMemset(fd, 0, 64);
    hx[3000] = poison ? __builtin_nanf("") : 1.f;
    hipMemcpy(x, hx, sizeof hx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, 0, fh, fd, x, which);
    hipError_t e = hipDeviceSynchronize();
    unsigned d = 0; hipMemcpy(&d, fd, 4, hipMemcpyDeviceToHost);
    printf("target %s poison %d: host word %u device word %u (%s)\n", which ? "device" : "host-pinned", poison, fh[0], d, hipGetErrorString(e));
  }
  return 0;
}
