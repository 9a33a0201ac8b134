#!/usr/bin/env python3
"""bench.py -- mel-frames/sec of VAENAR.inference on MI355X (BASELINE.json metric).

A "step" is one pass of the text->mel hot path (text encoder -> flow prior sample -> decoder,
models.py:199-210) over one synthetic batch S1 = (B=16, T_text=128, T_mel=800, 80 bins, rf=2) whose
inputs (token ids, lengths, prior noise) are already resident in HBM.  With N GPUs every rank runs
its own S1 batch (weak scaling, utterances are independent -> no data-path collective; the only
cross-rank traffic is the timing barrier / max, done over gloo).

Output: ONE JSON line on rank 0 (see the contract in the task description) with, besides the
throughput, a `roofline` object for the dominant kernel (the fp32-MFMA GEMM family), a
`roofline_cross_attention` object for the decoder cross-attention core (HBM-bound kernel named by
the north star), a `cpu_baseline` (the NumPy oracle, fp32, timed on this box's host cores) and a
`parity` object (max-abs mel error of the measured GPU output vs that CPU run).
"""
import argparse
import json
import os
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

S1 = dict(B=int(os.environ.get("VNR_BENCH_B", "16")), T_text=128, T_mel=800, rf=2)   # (VNR_BENCH_B: tuning experiments only)
ALG_GFLOP_S1 = 343.2           # algorithmic FLOPs of one S1 inference batch
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (no sparsity)
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the extra training-step measurement")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--exact-fp32", action="store_true", help="disable the split-fp16 GEMM path (exact fp32 MFMA everywhere)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (A/B switches), repeatable")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent B=16 batches kept in flight per GPU, one engine handle (= one HIP stream) each; the K timed "
                         "steps are dealt round-robin to them.  One S1 batch leaves CUs idle (200 row panels on 256 CUs, single "
                         "waves of attention workgroups); further streams fill them (measured 1: 4.86 M, 2: 6.11 M, 3: 6.63 M, 4: 6.03 M frames/s).  1 = the strictly sequential schedule")
    args = ap.parse_args()

    from vaenar_tts_amd import dist as vdist
    rank, local_rank, world = vdist.init("gloo")      # control plane only; no data-path collective
    barrier = vdist.barrier if world > 1 else (lambda: None)

    from vaenar_tts_amd import _lib
    from vaenar_tts_amd.configs import LJHPS
    from vaenar_tts_amd.models import VAENAR
    from vaenar_tts_amd.synthetic import make_batch
    from vaenar_tts_amd.weights import init_weights

    hps = LJHPS
    ndev = _lib.device_count()
    if ndev <= 0:
        raise SystemExit("bench.py needs an AMD GPU (libvaenar_hip has no CPU fallback)")
    device = local_rank % ndev
    weights = init_weights(hps, seed=1234, mode="synthetic", include_posterior=False)
    B, Tt, Tm, rf = S1["B"], S1["T_text"], S1["T_mel"], S1["rf"]
    Tz = (Tm + rf - 1) // rf
    nstreams = max(1, args.streams)
    lanes = []                                           # one engine handle (own stream, workspace, weights copy) per batch in flight
    for si in range(nstreams):
        m = VAENAR(hps, device=device, weights=weights)
        if args.exact_fp32:
            m.engine.set_option("split_fp16", 0)
        if nstreams > 1:
            # 64-row panels in the chain kernel: half the workgroups, half the weight stream per row.  One batch alone is 10 %
            # slower with them (2.92 vs 2.63 ms) but the CUs they leave free take the other batches' kernels: 7.5 vs 6.6 M frames/s
            m.engine.set_option("chain_rows64", 1)
            m.engine.set_option("gemm_wide_tiles", 1)       # same idea for the GEMM kernel: 64x128 tiles (+4 % with 3 streams)
        for kv in args.opt:                              # A/B switches, e.g. --opt attn_presplit_self=0
            name, val = kv.split("=")
            m.engine.set_option(name, int(val))
        bt = make_batch(B, Tt, Tm, ragged=False, seed=1234 + rank + 1000 * si, temperature=1.0)
        # inputs resident in HBM before the timed region (mel lengths stay on the host: only their max decides launch shapes)
        lanes.append({"model": m, "batch": bt, "ids": m.engine.to_device(bt["ids"], np.int32),
                      "tl": m.engine.to_device(bt["text_lengths"], np.int32), "eps": m.engine.to_device(bt["eps"], np.float32)})
    model, eng, batch = lanes[0]["model"], lanes[0]["model"].engine, lanes[0]["batch"]
    d_ids, d_tl, d_ml, d_eps = lanes[0]["ids"], lanes[0]["tl"], batch["mel_lengths"], lanes[0]["eps"]

    def step(i=0):
        ln = lanes[i % nstreams]
        return ln["model"].inference(ln["ids"], ln["batch"]["mel_lengths"], ln["tl"], reduction_factor=rf, eps=ln["eps"],
                                     return_alignments=True)

    def sync_all():
        for ln in lanes:
            ln["model"].engine.synchronize()

    def timed(nsteps, use_lanes):
        sync_all()
        barrier()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i if use_lanes else 0)
        sync_all()
        barrier()
        return vdist.max_over_ranks(time.perf_counter() - t0)

    for i in range(max(args.warmup, 1) * nstreams):
        step(i)
    mel, ali = step(0)
    dt = timed(args.steps, True)                         # EXACTLY K steps, dealt round-robin to the streams
    dt_single = timed(args.steps, False) if nstreams > 1 else dt   # the same K steps strictly one after another (reported beside)

    ms_per_step = 1e3 * dt / args.steps
    frames_per_step = B * Tm * world
    value = frames_per_step / (dt / args.steps)

    out = {
        "metric": "mel-frames/sec", "value": value, "unit": "mel-frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.exact_fp32 else "f32 (fp32 in/out/accumulate; GEMM and attention products are evaluated as a "
                                               "3-term fp16 hi/lo split on the f16 matrix pipe, 22 bits per operand)",
        "data": "synthetic",
        "config": {"workload": "S1 VAENAR.inference: B=16 per GPU, T_text=128, T_mel=800, 80-bin, rf=2, "
                               "LJHPS architecture, random-init weights, prior noise temperature 1.0, "
                               "decoder alignments returned", "global_batch": B * world,
                   "parallelism": "batch-sharded x%d (no collective)" % world,
                   "batches_in_flight_per_gpu": nstreams},
        "single_stream": {"ms_per_step": 1e3 * dt_single / args.steps, "value": frames_per_step / (dt_single / args.steps),
                          "note": "the same K steps issued strictly one after another on one stream (batches_in_flight 1)"},
    }

    if rank == 0:
        # ---- per-kernel roofline: HIP events around every launch on the engine's stream -----------
        eng.profile(True)
        eng.profile_reset()
        launches0 = eng.launch_count()
        for _ in range(args.profile_steps):
            step()
        eng.synchronize()
        launches = (eng.launch_count() - launches0) // max(1, args.profile_steps)
        prof = {c: eng.profile_get(c) for c in ("gemm", "attn_self", "attn_cross", "attn_cross_ali",
                                                "layer_norm", "misc")}
        eng.profile(False)
        eng.profile_reset()
        traffic = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as fh:
                traffic = json.load(fh)        # PMC-derived bytes per launch, collected by separate rocprofv3 passes
        except Exception:
            pass
        g = prof["gemm"]
        gemm_tflops = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        out["roofline"] = {
            "kernel": "gemm2_kernel family (LDS-DMA ring; Dense/concat/conv/LN epilogues) + panel_chain_kernel (row-panel "
                      "chains of the attention blocks); 3-term split-fp16 MFMA 32x32x16" if not args.exact_fp32 else
                      "gemm2_kernel family (fp32 MFMA 32x32x2, LDS-DMA ring; Dense/concat/conv/LN epilogues)",
            "bound": "mfma", "achieved": gemm_tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": gemm_tflops / PEAK_FP32_MFMA_TFLOPS,
            "peak_note": "achieved = ALGORITHMIC fp32 FLOPs (2*M*N*K) / GEMM kernel time; peak = fp32 matrix peak, the "
                         "ceiling of the fp32 contract.  Split launches execute 3 f16 MFMA FLOPs per algorithmic FLOP: "
                         "against the dense f16 MFMA peak the fraction is frac_f16_peak.",
            "frac_f16_peak": (gemm_tflops * (1.0 if args.exact_fp32 else 3.0)) / PEAK_F16_MFMA_TFLOPS,
            "traffic": traffic.get("gemm_bytes_per_launch"),
            "traffic_source": traffic.get("source"),
            "measured": "one batch in flight (the profiled pass after the timed region runs on a single stream); same as "
                        "`rocprofv3 --kernel-trace --stats -- python3 bench.py --streams 1 ...` in profiles/",
            "launches_per_step": g["launches"] // max(1, args.profile_steps),
            "avg_launch_us": 1e3 * g["ms"] / max(1, g["launches"]),
            "flops_per_step": g["flops"] / max(1, args.profile_steps),
        }
        a = prof["attn_cross_ali"]
        if a["launches"]:
            gbps = a["bytes"] / (a["ms"] * 1e-3) / 1e9
            out["roofline_cross_attention"] = {
                "kernel": "attn3_kernel<true> (decoder cross-attention core on producer-split operand images, alignments stored)",
                "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": gbps / PEAK_HBM_GBPS, "traffic": traffic.get("cross_attention_ali_bytes_per_launch"),
                "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
                "avg_launch_us": 1e3 * a["ms"] / a["launches"],
            }
        out["end_to_end"] = {
            "algorithmic_gflop_per_step": ALG_GFLOP_S1,
            "achieved_tflops": ALG_GFLOP_S1 * 1e9 / (ms_per_step * 1e-3) / 1e12,
            "frac_of_fp32_mfma_peak": ALG_GFLOP_S1 * 1e9 / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "kernel_launches_per_step": launches,
            "kernel_ms_per_step": {c: p["ms"] / max(1, args.profile_steps) for c, p in prof.items()},
        }
        out["device"] = eng.device_info()
        # ---- host-to-host latency of ONE call (ids on the host -> mel on the host), one batch in flight ----
        lat = []
        for _ in range(20):
            t1 = time.perf_counter()
            m1, _a1 = model.inference(batch["ids"], batch["mel_lengths"], batch["text_lengths"], reduction_factor=rf, eps=d_eps,
                                      return_alignments=False)
            m1.numpy()                                       # device -> host copy ends the call
            lat.append(1e3 * (time.perf_counter() - t1))
        lat.sort()
        out["latency_host_to_host_ms"] = {"min": lat[0], "median": lat[len(lat) // 2], "p95": lat[int(0.95 * (len(lat) - 1))],
                                          "note": "one S1 batch: token ids and lengths uploaded, 4.1 MB of mels downloaded, alignments not "
                                                  "requested; PCIe-inclusive, never `value`"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline: the NumPy oracle (fp32) on the same S1 batch, host cores of this box ----
        from oracle.vaenar_numpy import Oracle
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
        except Exception:
            threads = os.cpu_count() or 1
        orc = Oracle(hps, weights, np.float32)
        ref = None
        t_cpu = []
        for _ in range(max(1, args.cpu_iters - 1)):
            t1 = time.perf_counter()
            ref, _ = orc.inference(batch["ids"], batch["mel_lengths"], batch["text_lengths"], rf, batch["eps"])
            t_cpu.append(time.perf_counter() - t1)
        best_np = min(t_cpu)
        # second stand-in: the torch-CPU restatement in fp32 (oneDNN / MKL GEMMs, all host cores) -- the
        # closest thing in class to the reference's TF2 + MKL-DNN CPU path.  The FASTER of the two is the baseline.
        best_t, t_threads = None, None
        try:
            import torch
            from oracle.vaenar_torch import TorchOracle
            torc = TorchOracle(hps, weights, torch.float32)
            # the GEMMs of one S1 batch are too small for every core of a large host: sweep the thread count and keep the
            # best (16 threads on the 256-CPU GPU box; more threads are slower)
            cand = sorted({n for n in (8, 16, 32, min(32, torch.get_num_threads())) if n <= (os.cpu_count() or 1)})
            for n in cand:
                torch.set_num_threads(n)
                for _ in range(2):
                    t1 = time.perf_counter()
                    torc.inference(batch["ids"], batch["mel_lengths"], batch["text_lengths"], rf, batch["eps"])
                    d = time.perf_counter() - t1
                    if best_t is None or d < best_t:
                        best_t, t_threads = d, n
        except Exception:
            pass
        use_torch = best_t is not None and best_t < best_np
        best = best_t if use_torch else best_np
        out["cpu_baseline"] = {
            "value": B * Tm / best, "unit": "mel-frames/s", "cores": int(t_threads if use_torch else threads), "kind": "port",
            "sample": "the full S1 batch (16 x 800 frames), best run of %s; the reference's TF2-CPU path cannot run here (no "
                      "TensorFlow)" % ("oracle/vaenar_torch.py in fp32 (torch CPU: oneDNN/MKL, %d threads)" % t_threads if use_torch
                                       else "oracle/vaenar_numpy.py in fp32 (NumPy/OpenBLAS)"),
            "seconds_per_batch": best, "host_cpus": os.cpu_count(),
            "candidates_mel_frames_per_s": {"numpy_fp32_oracle": B * Tm / best_np,
                                            "torch_cpu_fp32_restatement": (B * Tm / best_t) if best_t else None},
        }
        got = mel.numpy()
        out["parity"] = {"max_abs_mel_err": float(np.abs(got - ref).max()),
                         "against": "oracle fp32 on the same batch", "tolerance": 1e-3}
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]

    if rank == 0 and world == 1 and not args.no_train:
        # ---- the training step (BASELINE config 3: train.py step, ELBO fwd+bwd+Adam, B=32, no all-reduce) -- reported
        #      beside the headline metric; never part of `value` ------------------------------------------------------
        try:
            eng.close()
            tw = init_weights(hps, seed=1234, mode="synthetic", include_posterior=True)
            tm = VAENAR(hps, device=device, weights=tw)
            TB, trf = 32, 2
            tb = make_batch(TB, Tt, Tm, ragged=False, seed=99)
            r = np.random.Generator(np.random.PCG64(7))
            t_mels = tm.engine.to_device(r.standard_normal((TB, Tm, hps.Audio.num_mels)).astype(np.float32), np.float32)
            t_eps = tm.engine.to_device(r.standard_normal((TB, (Tm + trf - 1) // trf, hps.Common.latent_dim)).astype(np.float32), np.float32)
            t_ids = tm.engine.to_device(tb["ids"], np.int32)
            res = None
            for i in range(2):
                res = tm.train_step(t_ids, t_mels, tb["text_lengths"], tb["mel_lengths"], 1e-5, trf, eps=t_eps, dropout_seed=i)
            n0 = tm.engine.launch_count()
            t1 = time.perf_counter()
            nst = 4
            for i in range(nst):
                res = tm.train_step(t_ids, t_mels, tb["text_lengths"], tb["mel_lengths"], 1e-5, trf, eps=t_eps, dropout_seed=2 + i)
            tdt = (time.perf_counter() - t1) / nst
            out["training"] = {
                "workload": "T1 train_step (train.py:127-138): training-mode ELBO forward + backward of all 501 variables + Adam, "
                            "B=32, T_text=128, T_mel=800, rf=2, LJHPS, exact fp32, 1 GPU, no gradient all-reduce",
                "ms_per_step": 1e3 * tdt, "mel_frames_per_s": TB * Tm / tdt, "steps": nst,
                "kernel_launches_per_step": (tm.engine.launch_count() - n0) // nst,
                "approx_tflops": 3.0 * ALG_GFLOP_S1 * (TB / B) * 1e9 / tdt / 1e12,
                "loss": res[0], "mel_l2": res[1], "kl": res[2], "length_l2": res[3],
            }
            tm.engine.close()
        except Exception as e:                        # never let the extra block take the headline line down
            out["training"] = {"error": repr(e)}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as tdist
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
