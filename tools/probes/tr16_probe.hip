#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __fp16 f4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(const short* in, short* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = in[i];
  __syncthreads();
  // lane l supplies the address of 4 contiguous 16-bit elements: row (l&15)>>2 of group l>>4, cols 4*(l&3)
  const int l = threadIdx.x, q = l >> 4, jr = (l & 15) >> 2, c = l & 3;
  const short* p = lds + (4 * q + jr) * pitch + 4 * c;
  s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  std::vector<short> h(8192); for (int i = 0; i < 8192; ++i) h[i] = (short)i;
  short *d, *o; hipMalloc(&d, 16384); hipMalloc(&o, 512);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  for (int pitch : {16, 160}) {
    k<<<1, 64>>>(d, o, pitch); short r[256]; hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    printf("pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" r%d c%d", r[l*4+j] / pitch, r[l*4+j] % pitch); printf("\n"); }
  }
  return 0;
}
