// round 5: do v_cvt_f16_f32 and v_cvt_pk_f16_f32 (gfx950) round every fp32 input alike?  All 2^32 bit patterns.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt_pk_probe tools/probes/cvt_pk_probe.hip ; run: /tmp/cvt_pk_probe
// (profiles/r05_experiments.txt r05i: the hi / lo split of csrc/gemm3c.hip stored halves from one and subtracted halves from the other)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void scan(unsigned long long* count, unsigned* examples) {
  const unsigned long long n = 1ull << 32;
  unsigned long long local = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned bits = (unsigned)i;
    float x = __uint_as_float(bits);
    unsigned a, b;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(x));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(b) : "v"(x));
    a &= 0xffffu; const unsigned b0 = b & 0xffffu, b1 = b >> 16;
    const bool nan_a = (a & 0x7c00u) == 0x7c00u && (a & 0x3ffu);
    const bool nan_b = (b0 & 0x7c00u) == 0x7c00u && (b0 & 0x3ffu);
    if ((a != b0 && !(nan_a && nan_b)) || b0 != b1) {
      ++local;
      const unsigned long long slot = atomicAdd(count + 1, 1ull);
      if (slot < 16) { examples[3 * slot] = bits; examples[3 * slot + 1] = a; examples[3 * slot + 2] = b; }
    }
  }
  atomicAdd(count, local);
}
static float h2f(unsigned h) {
  const int s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  float v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m + 1024), e - 25));
  return s ? -v : v;
}
int main() {
  unsigned long long* c; unsigned* ex;
  hipMalloc(&c, 16); hipMalloc(&ex, 16 * 12); hipMemset(c, 0, 16); hipMemset(ex, 0, 16 * 12);
  scan<<<1024, 256>>>(c, ex);
  unsigned long long hc[2]; unsigned hex[48];
  hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); hipMemcpy(hex, ex, 16 * 12, hipMemcpyDeviceToHost);
  printf("fp32 inputs for which v_cvt_f16_f32 and v_cvt_pk_f16_f32 give different fp16 bits: %llu of 4294967296\n", hc[0]);
  for (int i = 0; i < 16 && i < (int)hc[1]; ++i) {
    float x; unsigned b = hex[3 * i]; memcpy(&x, &b, 4);
    printf("  x = %.9g (0x%08x): v_cvt_f16_f32 -> 0x%04x = %.9g   v_cvt_pk_f16_f32 -> 0x%04x = %.9g\n", x, b, hex[3 * i + 1], h2f(hex[3 * i + 1]), hex[3 * i + 2] & 0xffff, h2f(hex[3 * i + 2] & 0xffff));
  }
  return 0;
}
