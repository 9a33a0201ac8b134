// Probe: 64x64-tile DMA-ring GEMM skeleton at the ffn1 shape; ablate MFMA / DMA / epilogue / barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// flags: 1 = do MFMA, 2 = do DMA, 4 = do epilogue store, 8 = barrier
template <int FLAGS, int NSTAGE, int BM, int BN>
__global__ void __launch_bounds__(256) ring(const float* A, const float* W, float* C, int M, int N, int K, int tiles_n) {
  constexpr int NW = 4, AQ = BM / 8 / NW, BQ = BN / 8 / NW, LPW = AQ + BQ, STAGE = (BM + BN) * 128;
  constexpr int MI = BM / 64, NI = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x, tm = bid / tiles_n, tn = bid - tm * tiles_n, m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((size_t)N * K * 4), 0x00020000);
  unsigned aoff[AQ], boff[BQ];
  for (int x = 0; x < AQ; ++x) { int r = 8 * (wave + NW * x) + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7); aoff[x] = ((m0 + r) * K + 4 * c) * 4; }
  for (int x = 0; x < BQ; ++x) { int r = 8 * (wave + NW * x) + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7); boff[x] = ((n0 + r) * K + 4 * c) * 4; }
  auto issue = [&](int kt, int slot) {
    if (!(FLAGS & 2)) return;
    char* sb = smem + slot * STAGE;
#pragma unroll
    for (int x = 0; x < AQ; ++x) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(sb + (wave + NW * x) * 1024), 16, aoff[x], kt * 128, 0, 0);
#pragma unroll
    for (int x = 0; x < BQ; ++x) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(sb + BM * 128 + (wave + NW * x) * 1024), 16, boff[x], kt * 128, 0, 0);
  };
  int rd[4];
  for (int c8 = 0; c8 < 4; ++c8) rd[c8] = l31 * 128 + (((2 * c8 + half) ^ ((l31 >> 1) & 7)) << 4);
  f32x16 acc[MI][NI];
  for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / 32;
  for (int s = 0; s < NSTAGE - 1; ++s) if (s < nk) issue(s, s);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (FLAGS & 2) {
      const int younger = min(nk - 1, kt + NSTAGE - 2) - kt;
      if (NSTAGE == 4 && younger >= 2) wait_vmcnt<2 * LPW>(); else if (younger >= 1) wait_vmcnt<LPW>(); else wait_vmcnt<0>();
    }
    if (FLAGS & 8) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NSTAGE - 1 < nk) { int ns = slot + NSTAGE - 1; if (ns >= NSTAGE) ns -= NSTAGE; issue(kt + NSTAGE - 1, ns); }
    if (FLAGS & 1) {
      const char* As = smem + slot * STAGE + wm * (BM / 2) * 128;
      const char* Bs = smem + slot * STAGE + BM * 128 + wn * (BN / 2) * 128;
#pragma unroll
      for (int c8 = 0; c8 < 4; ++c8) {
        f32x4 a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(As + i * 4096 + rd[c8]);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *(const f32x4*)(Bs + j * 4096 + rd[c8]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    }
    if (++slot == NSTAGE) slot = 0;
  }
  if (FLAGS & 4) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          C[(size_t)row * N + col] = acc[i][j][r];
        }
      }
  } else {
    float s = 0; for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.f) C[0] = s;
  }
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

template <int FLAGS, int NSTAGE, int BM, int BN>
void run(const char* name, const float* A, const float* W, float* C, int M, int N, int K) {
  const int tiles_n = N / BN, grid = (M / BM) * tiles_n;
  const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
  (void)hipFuncSetAttribute((const void*)&ring<FLAGS, NSTAGE, BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring<FLAGS, NSTAGE, BM, BN>), dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, tiles_n);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((ring<FLAGS, NSTAGE, BM, BN>), dim3(grid), dim3(256), lds, 0, A, W, C, M, N, K, tiles_n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-34s tile %3dx%3d st%d grid %5d  %8.2f us  %7.1f TFLOP/s-equiv\n", name, BM, BN, NSTAGE, grid, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
}

int main() {
  const int M = 6400, N = 1024, K = 256;
  float *A, *W, *C;
  (void)hipMalloc(&A, (size_t)M * 1024 * 4); (void)hipMalloc(&W, (size_t)1024 * 1024 * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemset(A, 0, (size_t)M * 1024 * 4); (void)hipMemset(W, 0, (size_t)1024 * 1024 * 4);
  printf("ffn1 shape M=%d N=%d K=%d\n", M, N, K);
  run<15, 4, 64, 64>("full", A, W, C, M, N, K);
  run<15, 3, 64, 64>("full", A, W, C, M, N, K);
  run<14, 4, 64, 64>("no mfma", A, W, C, M, N, K);
  run<13, 4, 64, 64>("no dma", A, W, C, M, N, K);
  run<11, 4, 64, 64>("no epilogue store", A, W, C, M, N, K);
  run<7, 4, 64, 64>("no barrier (racy)", A, W, C, M, N, K);
  run<9, 4, 64, 64>("mfma+barrier only", A, W, C, M, N, K);
  run<1, 4, 64, 64>("mfma only", A, W, C, M, N, K);
  run<15, 3, 128, 128>("full", A, W, C, M, N, K);
  run<13, 3, 128, 128>("no dma", A, W, C, M, N, K);
  run<14, 3, 128, 128>("no mfma", A, W, C, M, N, K);
  run<11, 3, 128, 128>("no epilogue store", A, W, C, M, N, K);
  run<1, 3, 128, 128>("mfma only", A, W, C, M, N, K);
  printf("att_proj shape M=6400 N=256 K=512\n");
  run<15, 4, 64, 64>("full", A, W, C, 6400, 256, 512);
  run<15, 3, 128, 128>("full", A, W, C, 6400, 256, 512);
  run<1, 4, 64, 64>("mfma only", A, W, C, 6400, 256, 512);
  printf("square M=N=K=1024*4 (A reused as 4096x1024 -> K=1024)\n");
  run<15, 3, 128, 128>("full", A, W, C, 4096, 1024, 1024);
  run<1, 3, 128, 128>("mfma only", A, W, C, 4096, 1024, 1024);
  return 0;
}
