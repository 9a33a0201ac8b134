// L2 warming for the row-panel chain kernels (gemm3.hip, gemm3c.hip) -- round 5.
//
// Every worker workgroup of a chain launch streams the SAME weight images (4.2 MB per block launch) and the 25 workers of an XCD walk
// them in near lock step: a line is fetched from the Infinity Cache / HBM once per XCD (first touch) and served from that XCD's L2 to
// the 24 workgroups behind.  The look-ahead of the whole XCD is therefore ONE workgroup's register prefetch (128 - 256 KiB), and at
// a first-touch latency of 2 - 3 us that caps the XCD's unique-byte rate below the 64 B/clk every one of its CUs consumes
// (measured: the same launches with L2-resident weights run 7 % (8-wave kernel) / 14 % (4-wave kernel) faster,
// profiles/r05_experiments.txt).  A launch occupies 200 of 256 CUs (6400 rows / 32), so the 56 idle CUs -- seven per XCD -- run
// PREFETCH workgroups: they walk the launch's weight images a few stages ahead of the workers with plain 16-byte loads whose
// data is dropped: the lines are then L2 hits when the workers ask.  Pacing (a 4 MiB L2 must not be flooded 4 MB ahead): the workers
// publish the stage they have reached in one word per XCD (an atomic max of epoch * 32 + stage: no reset between launches), the
// prefetchers stay kAhead stages in front of it.  Nothing waits for a prefetcher, ever; a prefetcher that sees no progress for
// ~100 us gives up (workers not resident: other kernels hold the CUs).  Placement (block b -> XCD b % 8) is only used for speed: the
// XCD is read from the hardware register on both sides.
#pragma once
#include "common.h"

namespace vnr {

constexpr int kPrefetchAhead = 4;                          // stages (~256 KiB each) a prefetcher may run in front of its XCD's workers

__device__ __forceinline__ unsigned chain_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x & 7u;
}
// worker side: stage `si` begins (one lane of the workgroup)
__device__ __forceinline__ void chain_publish_stage(const ChainArgs& g, int si) {
  if (!g.pf_progress) return;
  unsigned* w = g.pf_progress + 16 * chain_xcc_id();       // one word per XCD, 64 bytes apart
  const unsigned v = g.pf_epoch * 32u + (unsigned)si;
  if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(w, v);     // (the word only grows: look first)
}
// prefetcher side: the whole life of a prefetch workgroup (blockDim.x threads, rank `p` of `np` among its XCD's prefetchers)
__device__ __forceinline__ void chain_prefetch_role(const ChainArgs& g, int p, int np) {
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const unsigned* w = g.pf_progress + 16 * chain_xcc_id();
  const unsigned base = g.pf_epoch * 32u;
  float sink = 0.f;
  for (int s = 0; s < g.nstages; ++s) {
    // wait until the workers of this XCD are within kPrefetchAhead stages (they publish base + stage; an older epoch reads as "not started")
    if (s > kPrefetchAhead) {
      int spins = 0;
      while (true) {
        const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= base + (unsigned)(s - kPrefetchAhead) && v < base + 32u) break;
        if (v >= base + 32u) return;                         // a later launch owns the word already: this one is over
        if (++spins > 4000) return;                          // ~100 us without progress: give up (never a hang)
        __builtin_amdgcn_s_sleep(32);
      }
    }
    const ChainStage& st = g.st[s];
    const int ncb = (st.n + 31) >> 5;
    const char* img = static_cast<const char*>(st.w);
    // the stage's image: ncb column blocks x nk k-tiles of 4 KiB; 16 bytes per thread and trip, prefetcher p takes every np-th chunk
    const long long nchunk = (long long)ncb * st.nk * 256;   // 16-byte units
    const long long step = (long long)np * nthreads;
    for (long long c0 = (long long)p * nthreads + tid; c0 < nchunk; c0 += 8 * step) {
      float4 v[8];                                           // eight independent loads in flight per thread
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long c = c0 + q * step;
        const long long cc = c < nchunk ? c : c0;            // (the tail re-reads its first chunk)
        const int tile = (int)(cc >> 8), inb = (int)(cc & 255);
        const int cb = tile / st.nk, kt = tile - cb * st.nk;
        v[q] = *reinterpret_cast<const float4*>(img + ((size_t)cb * st.kt_total + st.kt0 + kt) * 4096 + (size_t)inb * 16);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) sink += v[q].x;
    }
  }
  if (sink == 1234.5678f && g.pf_progress) g.pf_progress[127] = 1u;       // (keeps the loads alive; never true in practice, and harmless)
}

}  // namespace vnr
