#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP operators at the S1 shapes (through the C ABI).
usage: python tools/bench_ops.py [gemm|attn|all] [--iters N]
Times come from HIP events recorded around each launch on the engine's stream."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vaenar_tts_amd import _lib  # noqa: E402
from vaenar_tts_amd.configs import tiny_hps  # noqa: E402

# (name, M, K1, K2, N, ln, act)
GEMMS = [
    ("xblk qkv", 6400, 256, 0, 768, 0, None),
    ("xblk att_proj+LN", 6400, 256, 256, 256, 1, None),
    ("xblk q", 6400, 256, 0, 256, 0, None),
    ("xblk ffn1", 6400, 256, 0, 1024, 0, "relu"),
    ("xblk ffn2+LN", 6400, 1024, 0, 256, 1, None),
    ("flow fold", 6400, 128, 0, 128, 0, None),
    ("flow pre_proj", 6400, 64, 0, 256, 0, None),
    ("flow heads", 6400, 256, 0, 128, 0, None),
    ("kv all", 2048, 512, 0, 7168, 0, None),
    ("enc qkv", 2048, 512, 0, 768, 0, None),
    ("enc att_proj", 2048, 512, 256, 512, 0, None),
    ("enc ffn1", 2048, 512, 0, 1024, 0, "relu"),
    ("enc ffn2", 2048, 1024, 0, 512, 0, None),
    ("dec out_proj", 6400, 256, 0, 160, 0, None),
    ("dec residual", 12800, 256, 0, 80, 0, None),
    ("square 4096", 4096, 4096, 0, 4096, 0, None),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--split", type=int, default=0, help="use the split-fp16 GEMM kernel")
    ap.add_argument("--presplit", type=int, default=0, help="cross attention through attention3 (operands converted to images first; only the attention3 launch is timed)")
    args = ap.parse_args()
    eng = _lib.Engine(tiny_hps(), 0)
    eng.set_option('op_dense_split', args.split)
    eng.set_option('op_attn_presplit', args.presplit)
    r = np.random.Generator(np.random.PCG64(0))
    if args.what in ("gemm", "all"):
        print("%-20s %6s %5s %5s %9s %9s %7s" % ("gemm", "M", "K", "N", "avg_us", "TFLOP/s", "frac"))
        for name, M, K1, K2, N, ln, act in GEMMS:
            K = K1 + K2
            a1 = eng.to_device(r.standard_normal((M, K1)).astype(np.float32))
            a2 = eng.to_device(r.standard_normal((M, max(K2, 4))).astype(np.float32))
            w = eng.to_device((r.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
            b = eng.to_device(r.standard_normal(N).astype(np.float32))
            res = eng.to_device(r.standard_normal((M, N)).astype(np.float32))
            g = eng.to_device(np.ones(N, np.float32))
            out = eng.empty((M, N))
            d = _lib.vnr_dense_desc()
            d.d_a1, d.lda1, d.k1 = a1.ptr, K1, K1
            if K2:
                d.d_a2, d.lda2, d.k2 = a2.ptr, K2, K2
            d.d_w, d.d_bias, d.activation = w.ptr, b.ptr, _lib.ACT[act]
            if ln:
                d.d_residual, d.ldr, d.d_ln_gamma, d.d_ln_beta = res.ptr, N, g.ptr, b.ptr
            d.d_c, d.ldc, d.m, d.n = out.ptr, N, M, N
            for _ in range(3):
                _lib.check(eng.lib.vnr_op_dense(eng.handle, C.byref(d)), eng.handle)
            eng.profile(True); eng.profile_reset()
            for _ in range(args.iters):
                _lib.check(eng.lib.vnr_op_dense(eng.handle, C.byref(d)), eng.handle)
            p = eng.profile_get("gemm"); eng.profile(False); eng.profile_reset()
            us = 1e3 * p["ms"] / p["launches"]
            tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
            print("%-20s %6d %5d %5d %9.2f %9.2f %7.3f" % (name, M, K, N, us, tf, tf / 157.3))
    if args.what in ("attn", "all"):
        print("%-24s %9s %9s %9s" % ("attention", "avg_us", "GB/s", "TFLOP/s"))
        for name, B, H, Tq, Tk, causal, ali in [("self causal 400", 16, 4, 400, 400, 1, 0),
                                                 ("cross 400x128", 16, 4, 400, 128, 0, 0),
                                                 ("cross 400x128 +ali", 16, 4, 400, 128, 0, 1),
                                                 ("enc self 128", 16, 4, 128, 128, 0, 0)]:
            D = 64 * H
            q = eng.to_device(r.standard_normal((B, Tq, D)).astype(np.float32))
            k = eng.to_device(r.standard_normal((B, Tk, D)).astype(np.float32))
            v = eng.to_device(r.standard_normal((B, Tk, D)).astype(np.float32))
            ctx = eng.empty((B, Tq, D))
            al = eng.empty((B, H, Tq, Tk)) if ali else None
            call = lambda: _lib.check(eng.lib.vnr_op_attention(
                eng.handle, q.ptr, D, k.ptr, D, v.ptr, D, None, None, B, H, Tq, Tk, causal, 1.0, ctx.ptr, D,
                None if al is None else al.ptr), eng.handle)
            for _ in range(3):
                call()
            cls = "attn_self" if causal else ("attn_cross_ali" if ali else "attn_cross")
            eng.profile(True); eng.profile_reset()
            for _ in range(args.iters):
                call()
            p = eng.profile_get(cls); eng.profile(False); eng.profile_reset()
            us = 1e3 * p["ms"] / p["launches"]
            print("%-24s %9.2f %9.1f %9.2f" % (name, us, p["bytes"] / p["launches"] / (us * 1e-6) / 1e9,
                                               p["flops"] / p["launches"] / (us * 1e-6) / 1e12))
    eng.close()


if __name__ == "__main__":
    main()
