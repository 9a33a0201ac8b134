// L2 warming for the row-panel chain kernels (gemm3.hip, gemm3c.hip) -- round 5.
//
// Every worker workgroup of a chain launch streams the SAME weight images (4.2 MB per block launch) and the 25 workers of an XCD walk
// them in near lock step: a line is fetched from the Infinity Cache / HBM once per XCD (first touch) and served from that XCD's L2 to
// the 24 workgroups behind.  The look-ahead of the whole XCD is therefore ONE workgroup's register prefetch (128 - 256 KiB), and at
// a first-touch latency of 2 - 3 us that caps the XCD's unique-byte rate below the 64 B/clk every one of its CUs consumes
// (measured: the same launches with L2-resident weights run 7 % (8-wave kernel) / 14 % (4-wave kernel) faster,
// profiles/r05_experiments.txt).  A launch occupies 200 of 256 CUs (6400 rows / 32), so the 56 idle CUs -- seven per XCD -- run
// PREFETCH workgroups: they walk the launch's weight images a few stages ahead of the workers with one 4-byte load per cache line whose
// data is dropped: the lines are then L2 hits when the workers ask.  Pacing (a 4 MiB L2 must not be flooded 4 MB ahead): the workers
// publish the stage they have reached in one word per XCD (an atomic max of epoch * 32 + stage: no reset between launches), the
// prefetchers stay kAhead stages in front of it.  No worker waits for a prefetcher, ever -- but the LAUNCH ends with its last workgroup, so a
// prefetcher that sees no progress for ~0.5 ms gives up (workers not resident: other kernels hold the CUs).  Placement (block b -> XCD b % 8) is only used for speed: the
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
// worker side: stage `si` begins (one lane of the workgroup).  Fire and forget: NO read of the word first -- a device-scope load is a
// round trip to memory (~2 us), and the first version's "look before the atomic" made wave 0 wait for it in front of every k-loop.
// (An XCD-local form was tried -- plain store through the write-through vector cache, polling loads with sc0: workers and prefetchers
// of a word share one L2 -- and dropped: a spinning prefetcher can keep hitting its CU's vector cache, never sees the store, and the
// launch then lasts until its give-up timeout: 58 ms per step instead of 2.4, profiles/r05_experiments.txt.)
__device__ __forceinline__ void chain_publish_stage(const ChainArgs& g, int si) {
  if (!g.pf_progress || g.prio_mode >= 20) return;         // (measurement only, VNR_CHAIN_PRIO >= 20: nobody publishes)
  unsigned* w = g.pf_progress + 16 * chain_xcc_id();       // one word per XCD, 64 bytes apart
  const unsigned v = g.pf_epoch * 32u + (unsigned)si;
  if (g.prio_mode == 5) { if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) atomicMax(w, v); return; }   // (measurement: look first)
  if (g.prio_mode != 6 && (((int)blockIdx.x >> 3) & 7)) return;      // every 8th worker of an XCD publishes ((measurement, 6: all of them)
  (void)__hip_atomic_fetch_max(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned chain_read_progress(const ChainArgs&, const unsigned* w) {
  return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// prefetcher side: the whole life of a prefetch workgroup (blockDim.x threads, rank `p` of `np` among its XCD's prefetchers)
__device__ __forceinline__ void chain_prefetch_role(const ChainArgs& g, int p, int np, unsigned long long* ts = nullptr) {
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const unsigned* w = g.pf_progress + 16 * chain_xcc_id();
  const unsigned base = g.pf_epoch * 32u;
  const int ahead = (g.prio_mode >= 10 && g.prio_mode < 30) ? g.prio_mode - 10 : kPrefetchAhead;      // (measurement only: VNR_CHAIN_PRIO = 10 + stages ahead)
  float sink = 0.f;
  if (ts && tid == 0) { ts[62] = chain_xcc_id(); ts[63] = 1000 + p; }     // (measurement: a prefetcher's row of the stamp file)
  for (int s = 0; s < g.nstages; ++s) {
    // wait until the workers of this XCD are within kPrefetchAhead stages (they publish base + stage; an older epoch reads as "not started")
    if (s > ahead) {
      int spins = 0;
      while (true) {
        const unsigned v = chain_read_progress(g, w);
        if (v >= base + (unsigned)(s - ahead) && v < base + 32u) break;
        if (v >= base + 32u) return;                         // a later launch owns the word already: this one is over
        if (++spins > 40) break;                             // ~0.1 ms without progress (a poll is a 2 - 3 us round trip; no publisher on this XCD?): go on unpaced
        __builtin_amdgcn_s_sleep(32);
      }
    }
    if (ts && tid == 0) ts[32 + s] = __builtin_amdgcn_s_memrealtime();   // stage s released to this prefetcher
    const ChainStage& st = g.st[s];
    const int ncb = (st.n + 31) >> 5;
    const char* img = static_cast<const char*>(st.w);
    // the stage's image: ncb column blocks x nk k-tiles of 4 KiB.  ONE 4-byte load per cache line pulls the line into the L2 (a lane
    // per line: a wave's load covers 64 lines = 8 KiB); prefetcher p takes every np-th group of lines.  (First version: every 16-byte
    // chunk was read -- 8 loads per line; the seven prefetchers of an XCD then moved ~45 B/clk against the ~37 B/clk of unique bytes
    // the workers consume, and the workers' k-loops jittered between the L2-hit and the miss rate.)
    const int lsh = g.prio_mode == 7 ? 6 : 7;                // (measurement only, VNR_CHAIN_PRIO=7: one load per 64 bytes)
    const int lpt = 4096 >> lsh;                             // lines per 4 KiB tile
    const long long nline = (long long)ncb * st.nk * lpt;
    const long long step = (long long)np * nthreads;
    for (long long c0 = (long long)p * nthreads + tid; c0 < nline; c0 += 8 * step) {
      float v[8];                                            // eight independent loads in flight per thread
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long c = c0 + q * step;
        const long long cc = c < nline ? c : c0;             // (the tail re-reads its first line)
        const int tile = (int)(cc / lpt), inb = (int)(cc - (long long)tile * lpt);
        const int cb = tile / st.nk, kt = tile - cb * st.nk;
        v[q] = *reinterpret_cast<const float*>(img + ((size_t)cb * st.kt_total + st.kt0 + kt) * 4096 + ((size_t)inb << lsh));
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) sink += v[q];
    }
    if (ts && tid == 0) ts[96 + s] = __builtin_amdgcn_s_memrealtime();   // ... and its lines requested AND returned (thread 0's)
  }
  if (sink == 1234.5678f && g.pf_progress) g.pf_progress[127] = 1u;       // (keeps the loads alive; never true in practice, and harmless)
}

}  // namespace vnr
