"""Kernel-level parity: every HIP operator, called through the C ABI, against the NumPy oracle
(float64) on the same seeded inputs.  Tolerances are stated per test; fp32 MFMA is an exact
fp32 fma chain, so differences are fp32 round-off of the contraction length."""
import ctypes as C

import numpy as np
import pytest

from oracle import vaenar_numpy as O
from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import tiny_hps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(tiny_hps(), 0)
    yield e
    e.close()


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def dense_gpu(eng, a1, w, a2=None, bias=None, act=None, residual=None, ln=None, pe=None, pe_w=0.0):
    m = a1.shape[0]
    n = w.shape[1]
    d = _lib.vnr_dense_desc()
    keep = []

    def dev(x):
        if x is None:
            return None
        t = eng.to_device(np.asarray(x, np.float32)); keep.append(t); return t.ptr
    d.d_a1, d.lda1, d.k1 = dev(a1), a1.shape[1], a1.shape[1]
    if a2 is not None:
        d.d_a2, d.lda2, d.k2 = dev(a2), a2.shape[1], a2.shape[1]
    d.d_w, d.d_bias, d.activation = dev(w), dev(bias), _lib.ACT[act]
    if residual is not None:
        d.d_residual, d.ldr = dev(residual), n
    if ln is not None:
        d.d_ln_gamma, d.d_ln_beta = dev(ln[0]), dev(ln[1])
    if pe is not None:
        d.d_pe, d.pe_T, d.pe_weight = dev(pe), pe.shape[0], pe_w
    out = eng.empty((m, n))
    d.d_c, d.ldc, d.m, d.n = out.ptr, n, m, n
    _lib.check(eng.lib.vnr_op_dense(eng.handle, C.byref(d)), eng.handle)
    return out.numpy()


@pytest.mark.parametrize("m,k,n", [(1, 4, 1), (37, 64, 80), (200, 512, 256), (300, 100, 513), (6400, 256, 1024),
                                   (2048, 512, 768), (129, 1024, 160)])
def test_dense_plain(eng, m, k, n):
    r = rng(m * 7 + n)
    a, w, b = r.standard_normal((m, k)), r.standard_normal((k, n)) / np.sqrt(k), r.standard_normal(n)
    got = dense_gpu(eng, a, w, bias=b, act="relu")
    ref = O.dense(a.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64),
                  b.astype(np.float32).astype(np.float64), "relu")
    np.testing.assert_allclose(got, ref, atol=2e-5 * np.sqrt(k / 64 + 1), rtol=1e-5)


@pytest.mark.parametrize("m,k1,k2,n", [(70, 96, 128, 96), (400, 256, 256, 256), (333, 512, 256, 512)])
def test_dense_concat_residual_layernorm(eng, m, k1, k2, n):
    """LN(x + Dense(concat(x, ctx))) -- attention.py:410-413 / 440-443 -- fused (n<=256) and two-pass."""
    r = rng(m + k1)
    a1, a2 = r.standard_normal((m, k1)), r.standard_normal((m, k2))
    w, b = r.standard_normal((k1 + k2, n)) / np.sqrt(k1 + k2), r.standard_normal(n) * 0.1
    res = r.standard_normal((m, n))
    g, be = 1 + 0.1 * r.standard_normal(n), 0.1 * r.standard_normal(n)
    f = lambda x: np.asarray(x, np.float32).astype(np.float64)
    got = dense_gpu(eng, a1, w, a2=a2, bias=b, residual=res, ln=(g, be))
    ref = O.layer_norm(f(res) + O.dense(np.concatenate([f(a1), f(a2)], -1), f(w), f(b)), f(g), f(be))
    np.testing.assert_allclose(got, ref, atol=3e-5, rtol=1e-5)


def test_dense_concat_panels_further_apart_than_one_descriptor(eng):
    """concat(x, ctx) -> Dense with the two row panels more than 2 GiB apart (two arena chunks of a long run): round 3 retired the
    first-generation GEMM that used to take this case; the DMA kernel now gives each panel its own buffer descriptor."""
    r = rng(77)
    m, k1, k2, n = 333, 256, 256, 256
    a1, a2 = r.standard_normal((m, k1)), r.standard_normal((m, k2))
    w, b = r.standard_normal((k1 + k2, n)) / np.sqrt(k1 + k2), r.standard_normal(n) * 0.1
    spacer = eng.empty((9 << 26,))                       # one 2.25 GiB allocation: x at its start, ctx 2.2 GiB further on
    d1 = _lib.DeviceArray(eng, a1.shape, np.float32, ptr=int(spacer.ptr), owner=spacer).copy_from(a1.astype(np.float32))
    d2 = _lib.DeviceArray(eng, a2.shape, np.float32, ptr=int(spacer.ptr) + (2252 << 20), owner=spacer).copy_from(a2.astype(np.float32))
    assert int(d2.ptr) - int(d1.ptr) >= (1 << 31)
    dw, db = eng.to_device(w.astype(np.float32)), eng.to_device(b.astype(np.float32))
    out = eng.empty((m, n))
    d = _lib.vnr_dense_desc()
    d.d_a1, d.lda1, d.k1, d.d_a2, d.lda2, d.k2 = d1.ptr, k1, k1, d2.ptr, k2, k2
    d.d_w, d.d_bias, d.activation = dw.ptr, db.ptr, _lib.ACT[None]
    d.d_c, d.ldc, d.m, d.n = out.ptr, n, m, n
    for opt in (0, 1):                                   # exact fp32 MFMA and the split-fp16 kernel
        eng.set_option("op_dense_split", opt)
        _lib.check(eng.lib.vnr_op_dense(eng.handle, C.byref(d)), eng.handle)
        f = lambda x: np.asarray(x, np.float32).astype(np.float64)
        ref = O.dense(np.concatenate([f(a1), f(a2)], -1), f(w), f(b))
        np.testing.assert_allclose(out.numpy(), ref, atol=3e-5, rtol=1e-5)
    eng.set_option("op_dense_split", 0)
    del spacer


def test_dense_pe_epilogue(eng):
    r = rng(5)
    T, B, k, n = 13, 3, 64, 128
    a, w, b = r.standard_normal((B * T, k)), r.standard_normal((k, n)) / 8, r.standard_normal(n)
    pe = O.positional_encoding(T, n, 1.0)
    got = dense_gpu(eng, a, w, bias=b, pe=pe, pe_w=1.25)
    f = lambda x: np.asarray(x, np.float32).astype(np.float64)
    ref = O.dense(f(a), f(w), f(b)) + np.float64(np.float32(1.25)) * np.tile(pe.astype(np.float64), (B, 1))
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,T,cin,cout,k,act,bn_first", [(2, 9, 16, 48, 5, "tanh", 0), (3, 40, 96, 96, 5, "relu", 0),
                                                          (1, 3, 80, 256, 5, "identity", 0), (2, 17, 48, 48, 3, "relu", 1),
                                                          (16, 128, 512, 512, 5, "relu", 0)])
def test_conv1d_bn(eng, B, T, cin, cout, k, act, bn_first):
    """Conv1D('same') + bias -> act -> BN(moving stats) (utils.py:76-85), incl. zero padding at the
    tensor edges and sequences shorter than the kernel."""
    r = rng(B * 100 + T)
    x = r.standard_normal((B, T, cin))
    w = r.standard_normal((k, cin, cout)) / np.sqrt(k * cin)
    b, g, be = r.standard_normal(cout) * 0.1, 1 + 0.1 * r.standard_normal(cout), 0.1 * r.standard_normal(cout)
    mu, var = 0.1 * r.standard_normal(cout), r.uniform(0.5, 1.5, cout)
    dv = lambda a: eng.to_device(np.asarray(a, np.float32))
    xs = [dv(t) for t in (x, w, b, g, be, mu, var)]
    y = eng.empty((B, T, cout))
    _lib.check(eng.lib.vnr_op_conv1d_bn(eng.handle, xs[0].ptr, B, T, cin, xs[1].ptr, k, cout, xs[2].ptr, _lib.ACT[act],
                                        bn_first, xs[3].ptr, xs[4].ptr, xs[5].ptr, xs[6].ptr, y.ptr), eng.handle)
    f = lambda a: np.asarray(a, np.float32).astype(np.float64)
    c = O.conv1d_same(f(x), f(w), f(b))
    if bn_first:
        ref = O.act(O.batch_norm_infer(c, f(g), f(be), f(mu), f(var)), act)
    else:
        ref = O.batch_norm_infer(O.act(c, act), f(g), f(be), f(mu), f(var))
    np.testing.assert_allclose(y.numpy(), ref, atol=3e-5, rtol=1e-5)


def attention_ref(q, k, v, ql, kl, H, causal, tau):
    """attention.py:221-246 on projected tensors, float64."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dh = D // H
    qh = q.reshape(B, Tq, H, dh).transpose(0, 2, 1, 3)
    kh = k.reshape(B, Tk, H, dh).transpose(0, 2, 1, 3)
    vh = v.reshape(B, Tk, H, dh).transpose(0, 2, 1, 3)
    lg = qh @ kh.transpose(0, 1, 3, 2) / np.sqrt(float(dh)) / tau
    mask = O.sequence_mask(kl, Tk)[:, None, :] & O.sequence_mask(ql, Tq)[:, :, None]
    if causal:
        mask = mask & np.tril(np.ones((Tq, Tk), bool))[None]
    lg = np.where(mask[:, None], lg, np.float64(O.MASK_FILL))
    ali = O.softmax_last(lg)
    ctx = (ali @ vh).transpose(0, 2, 1, 3).reshape(B, Tq, D)
    return ctx, ali


@pytest.mark.parametrize("B,H,Tq,Tk,causal,want_ali,tau,ragged", [
    (2, 2, 7, 7, 0, 0, 1.0, True),        # tiny, padded queries -> uniform rows
    (2, 2, 45, 45, 1, 0, 1.0, True),      # causal, ragged
    (3, 4, 400, 400, 1, 0, 1.0, True),    # S1-shaped causal self attention, multi-tile + skipping
    (3, 4, 400, 128, 0, 1, 1.0, True),    # S1-shaped decoder cross attention with alignments
    (2, 4, 130, 200, 0, 1, 1.0, True),    # Tk > 128 with alignments (two-pass path)
    (3, 4, 400, 400, 1, 1, 1.0, True),    # the training step's causal self-attention with stored probabilities (Tk <= 512: 8-tile form)
    (2, 2, 70, 300, 0, 1, 0.8, True),     # 5 tiles, partial last tile (Tk % 64 != 0), temperature
    (1, 2, 40, 520, 0, 1, 1.0, True),     # Tk > 512 with alignments: two passes (context + row statistics, then the probabilities)
    (2, 2, 70, 520, 0, 1, 0.9, True),     # ... ragged: padded query rows exactly uniform, temperature
    (2, 2, 530, 530, 1, 1, 1.0, True),    # ... causal self-attention beyond 512 keys
    (2, 1, 33, 129, 0, 0, 0.7, True),     # odd sizes, temperature != 1
    (1, 2, 64, 64, 1, 1, 1.0, False),     # causal with alignments requested
    (2, 4, 128, 128, 0, 0, 1.0, False),   # encoder-shaped, full lengths
])
def test_attention(eng, B, H, Tq, Tk, causal, want_ali, tau, ragged):
    r = rng(Tq * 3 + Tk)
    D = 64 * H
    q, k, v = r.standard_normal((B, Tq, D)), r.standard_normal((B, Tk, D)), r.standard_normal((B, Tk, D))
    q *= 1.5   # sharper softmax
    if ragged:
        ql = np.maximum(1, Tq - np.arange(B) * max(1, Tq // 3)).astype(np.int32)
        kl = np.maximum(1, Tk - np.arange(B) * max(1, Tk // 4)).astype(np.int32)
    else:
        ql, kl = np.full(B, Tq, np.int32), np.full(B, Tk, np.int32)
    if causal:
        kl = ql.copy()   # self attention: memory_lengths = query_lengths (attention.py:437-439)
    dq, dk, dv_ = (eng.to_device(t.astype(np.float32)) for t in (q, k, v))
    dql, dkl = eng.to_device(ql), eng.to_device(kl)
    ctx = eng.empty((B, Tq, D))
    ali = eng.empty((B, H, Tq, Tk)) if want_ali else None
    _lib.check(eng.lib.vnr_op_attention(eng.handle, dq.ptr, D, dk.ptr, D, dv_.ptr, D, dql.ptr, dkl.ptr, B, H, Tq, Tk,
                                        causal, tau, ctx.ptr, D, None if ali is None else ali.ptr), eng.handle)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    rctx, rali = attention_ref(f(q), f(k), f(v), ql, kl, H, causal, np.float64(np.float32(tau)))
    np.testing.assert_allclose(ctx.numpy(), rctx, atol=2e-5, rtol=1e-5)
    if want_ali:
        got = ali.numpy()
        np.testing.assert_allclose(got, rali, atol=2e-6, rtol=1e-5)
        # fully masked (padded-query) rows are exactly uniform over all Tk keys
        for b in range(B):
            if ql[b] < Tq:
                assert np.all(got[b, :, ql[b]:, :] == np.float32(1.0) / np.float32(Tk))


@pytest.mark.parametrize("B,H,Tq,Tk,causal,tau,ragged", [
    (2, 2, 7, 7, 0, 1.0, True),           # tiny, padded queries -> uniform rows
    (2, 2, 45, 45, 1, 1.0, True),         # causal, ragged, partial blocks
    (3, 4, 400, 400, 1, 1.0, True),       # S1-shaped causal self attention: 13 key blocks over 4 waves, block skipping
    (3, 4, 400, 400, 1, 1.0, False),      # ... full lengths (fast path everywhere below the diagonal)
    (2, 4, 130, 200, 0, 1.0, True),       # Tk > 128, non-causal: two rounds for some waves
    (2, 1, 33, 129, 0, 0.7, True),        # odd sizes, temperature != 1, a block holding one key
    (2, 4, 128, 128, 0, 1.0, False),      # encoder-shaped, full lengths
    (1, 2, 300, 520, 0, 1.0, True),       # 17 key blocks: up to five rounds per wave
])
def test_attention_presplit_general(eng, B, H, Tq, Tk, causal, tau, ragged):
    """attention3.hip general kernel (any Tk, causal, online softmax over a wave's key blocks, O^T accumulation)."""
    r = rng(Tq * 7 + Tk)
    D = 64 * H
    q, k, v = 1.5 * r.standard_normal((B, Tq, D)), r.standard_normal((B, Tk, D)), r.standard_normal((B, Tk, D))
    if ragged:
        ql = np.maximum(1, Tq - np.arange(B) * max(1, Tq // 3)).astype(np.int32)
        kl = np.maximum(1, Tk - np.arange(B) * max(1, Tk // 4)).astype(np.int32)
    else:
        ql, kl = np.full(B, Tq, np.int32), np.full(B, Tk, np.int32)
    if causal:
        kl = ql.copy()
    dq, dk, dv_ = (eng.to_device(t.astype(np.float32)) for t in (q, k, v))
    dql, dkl = eng.to_device(ql), eng.to_device(kl)
    ctx = eng.empty((B, Tq, D))
    eng.set_option("op_attn_presplit", 1)
    try:
        _lib.check(eng.lib.vnr_op_attention(eng.handle, dq.ptr, D, dk.ptr, D, dv_.ptr, D, dql.ptr, dkl.ptr, B, H, Tq, Tk,
                                            causal, tau, ctx.ptr, D, None), eng.handle)
    finally:
        eng.set_option("op_attn_presplit", 0)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    rctx, _ = attention_ref(f(q), f(k), f(v), ql, kl, H, causal, np.float64(np.float32(tau)))
    np.testing.assert_allclose(ctx.numpy(), rctx, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("B,H,Tq,Tk,want_ali,tau,ragged", [
    (3, 4, 400, 128, 1, 1.0, True),       # S1-shaped decoder cross attention with alignments
    (3, 4, 400, 128, 0, 1.0, False),      # prior cross attention, nothing masked (fast path)
    (2, 2, 7, 7, 1, 1.0, True),           # tiny: one partial key block, padded queries -> uniform rows
    (2, 1, 33, 100, 1, 0.7, True),        # odd sizes (Tk % 4 == 0, partial last block), temperature != 1
    (2, 4, 130, 65, 1, 1.0, True),        # Tk % 4 != 0: scalar alignment stores, third block holds one key
    (1, 2, 64, 5, 0, 1.0, False),         # a single short key block
    (2, 4, 96, 128, 0, 1.0, True),        # ragged without alignments
])
def test_attention_presplit(eng, B, H, Tq, Tk, want_ali, tau, ragged):
    """attention3.hip (cross attention, Tk <= 128, operands as producer-split images) through vnr_op_attention with the
    option "op_attn_presplit": same reference and tolerances as test_attention."""
    r = rng(Tq * 5 + Tk)
    D = 64 * H
    q, k, v = 1.5 * r.standard_normal((B, Tq, D)), r.standard_normal((B, Tk, D)), r.standard_normal((B, Tk, D))
    if ragged:
        ql = np.maximum(1, Tq - np.arange(B) * max(1, Tq // 3)).astype(np.int32)
        kl = np.maximum(1, Tk - np.arange(B) * max(1, Tk // 4)).astype(np.int32)
    else:
        ql, kl = np.full(B, Tq, np.int32), np.full(B, Tk, np.int32)
    dq, dk, dv_ = (eng.to_device(t.astype(np.float32)) for t in (q, k, v))
    dql, dkl = eng.to_device(ql), eng.to_device(kl)
    ctx = eng.empty((B, Tq, D))
    ali = eng.empty((B, H, Tq, Tk)) if want_ali else None
    eng.set_option("op_attn_presplit", 1)
    try:
        _lib.check(eng.lib.vnr_op_attention(eng.handle, dq.ptr, D, dk.ptr, D, dv_.ptr, D, dql.ptr, dkl.ptr, B, H, Tq, Tk,
                                            0, tau, ctx.ptr, D, None if ali is None else ali.ptr), eng.handle)
    finally:
        eng.set_option("op_attn_presplit", 0)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    rctx, rali = attention_ref(f(q), f(k), f(v), ql, kl, H, 0, np.float64(np.float32(tau)))
    np.testing.assert_allclose(ctx.numpy(), rctx, atol=2e-5, rtol=1e-5)
    if want_ali:
        got = ali.numpy()
        np.testing.assert_allclose(got, rali, atol=2e-6, rtol=1e-5)
        for b in range(B):
            if ql[b] < Tq:
                assert np.all(got[b, :, ql[b]:, :] == np.float32(1.0) / np.float32(Tk))


@pytest.mark.parametrize("rows,dim", [(5, 96), (6400, 256), (2048, 512), (3, 1024)])
def test_layer_norm(eng, rows, dim):
    r = rng(rows + dim)
    x, g, b = 3 * r.standard_normal((rows, dim)) + 1.0, 1 + 0.1 * r.standard_normal(dim), r.standard_normal(dim)
    dx, dg, db = (eng.to_device(t.astype(np.float32)) for t in (x, g, b))
    y = eng.empty((rows, dim))
    _lib.check(eng.lib.vnr_op_layer_norm(eng.handle, dx.ptr, dg.ptr, db.ptr, rows, dim, y.ptr), eng.handle)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(y.numpy(), O.layer_norm(f(x), f(g), f(b)), atol=5e-6, rtol=1e-5)


@pytest.mark.parametrize("T,dim,step", [(128, 512, 5.59 / 2), (400, 256, 1.0), (7, 96, 5.59 / 5)])
def test_positional_encoding(eng, T, dim, step):
    """utils.py:333-355.  Arguments reach ~360 rad in fp32 (ulp 3e-5): the table is compared with the
    oracle's correctly-rounded fp32 staging; tolerance = one ulp of the largest argument."""
    out = eng.empty((T, dim))
    _lib.check(eng.lib.vnr_op_positional_encoding(eng.handle, T, dim, float(np.float32(step)), out.ptr), eng.handle)
    ref = O.positional_encoding(T, dim, np.float32(step))
    np.testing.assert_allclose(out.numpy(), ref, atol=4e-5, rtol=0)


def _gemm_class_launches(eng, fn):
    """(split-path launches, exact-fp32 launches) of the tiled GEMM kernel while fn() runs."""
    eng.profile(True); eng.profile_reset()
    try:
        out = fn()
        eng.synchronize()
        n_split, n_exact = eng.profile_get("gemm")["launches"], eng.profile_get("gemm_fp32")["launches"]
    finally:
        eng.profile(False)
    return out, n_split, n_exact


@pytest.mark.parametrize("rows", ["window", "wide"])
@pytest.mark.parametrize("m,k1,k2,n,ln", [(200, 512, 0, 256, 0), (6400, 256, 0, 1024, 0), (333, 256, 256, 256, 1),
                                          (129, 1024, 0, 160, 0), (400, 96, 128, 96, 1), (64, 100, 0, 513, 0)])
def test_dense_split_fp16(eng, m, k1, k2, n, ln, rows):
    """The split-fp16 GEMM path (hi*hi + lo*hi + hi*lo on the fp16 matrix pipe, fp32 accumulate) against float64, judged per row RELATIVE
    TO THAT ROW'S OWN PRODUCT SCALE -- fp32 round-off class, 1e-5 -- not relative to max(scale, 1) as rounds 1-4 did (a row of magnitude
    1e-3 could be wrong by 3 % and pass).  "window": row magnitudes 0.05 ... 300, inside the split path's activation window [2^-6, 2^15)
    (include/vaenar_hip.h, "Arithmetic contract"): the split kernel must run and meet the criterion.  "wide": rows from 1e-5 to 1e5 in
    one matrix -- fp32's range, not fp16's: the operator's per-row range check must send the call to the exact kernel (an inf or a
    3 % error is a failure), same criterion."""
    r = rng(m + n + k1)
    mags = [0.05, 1.0, 30.0, 300.0] if rows == "window" else [1e-5, 1e-3, 1.0, 30.0, 1e5]
    a1 = r.standard_normal((m, k1)) * r.choice(mags, size=(m, 1))                   # mixed row magnitudes
    a2 = (r.standard_normal((m, k2)) * (np.abs(a1).max(1, keepdims=True) / 4)) if k2 else None
    w = r.standard_normal((k1 + k2, n)) / np.sqrt(k1 + k2)
    b = r.standard_normal(n) * (0.0 if not ln else 1.0)                               # (a bias of order 1 would hide the small rows)
    res = r.standard_normal((m, n)) if ln else None
    g, be = 1 + 0.1 * r.standard_normal(n), 0.1 * r.standard_normal(n)
    f = lambda x: np.asarray(x, np.float32).astype(np.float64)
    eng.set_option("op_dense_split", 1)
    try:
        got, n_split, n_exact = _gemm_class_launches(
            eng, lambda: dense_gpu(eng, a1, w, a2=a2, bias=b, residual=res, ln=(g, be) if ln else None))
    finally:
        eng.set_option("op_dense_split", 0)
    assert (n_split, n_exact) == ((1, 0) if rows == "window" else (0, 1)), (n_split, n_exact)
    assert np.isfinite(got).all()
    x = f(a1) if a2 is None else np.concatenate([f(a1), f(a2)], -1)
    ref = O.dense(x, f(w), f(b))
    if ln:
        # LayerNorm(res + x.w): rows whose product is far below the residual are decided by the residual; the criterion applies to the
        # normalised output (order 1) with the product's own error amplified by at most 1 / sigma ~ 1
        ref = O.layer_norm(f(res) + ref, f(g), f(be))
        worst = (np.abs(got - ref) / np.maximum(1.0, np.abs(x).max(1, keepdims=True))).max()
        assert worst < 1e-5, worst
        return
    pscale = np.sqrt((x * x).sum(1, keepdims=True) / (k1 + k2))                        # the row's product scale: |x.w| ~ rms(x) for unit columns
    worst = (np.abs(got - ref) / pscale).max()
    print(f"{rows}: worst error / row product scale {worst:.3e}")
    assert worst < 1e-5, worst


@pytest.mark.parametrize("B,T,K,N,shift,scale", [
    (32, 400, 256, 256, 0, 1.0),        # Dense of a decoder block at T1 size: the 128 x 128 transposing-read kernel
    (16, 100, 512, 256, 0, 3e-9),       # gradient of loss-scale magnitude (kl_weight 1e-5): the pre-scaling by max |dy|
    (4, 200, 256, 384, -2, 1.0),        # Conv1D tap left of centre: rows that would cross an utterance boundary drop out
    (4, 200, 128, 1024, 1, 1.0),        # tap right of centre, wide output
    (3, 97, 260, 132, 0, 1.0),          # ragged tile edges (K, N multiples of 4 only), M not a multiple of 32
    (2, 46, 80, 256, 0, 1.0),           # narrow input: second-generation kernel (K < 128)
    (5, 77, 256, 3, 0, 1.0),            # N = 3: second-generation kernel
])
def test_kernel_gradient_gemm(eng, B, T, K, N, shift, scale):
    """train.py:136 tape.gradient w.r.t. a Dense kernel / one Conv1D tap: dW = sum_m x[m + shift]^T dy[m] inside each utterance.
    Transpose-detecting inputs (K != N, random), compared with float64; 22-bit operands -> 2e-6 of the largest entry."""
    r = rng(B * T + K + N)
    M = B * T
    x = r.standard_normal((M, K)).astype(np.float32)
    dy = (scale * r.standard_normal((M, N))).astype(np.float32)
    dx, ddy = eng.to_device(x), eng.to_device(dy)
    dw = eng.empty((K, N))
    _lib.check(eng.lib.vnr_op_kernel_grad(eng.handle, dx.ptr, K, ddy.ptr, N, M, K, N, T, shift, dw.ptr), eng.handle)
    x3 = x.astype(np.float64).reshape(B, T, K)
    xs = np.zeros_like(x3)
    if shift >= 0:
        xs[:, :T - shift] = x3[:, shift:]
    else:
        xs[:, -shift:] = x3[:, :T + shift]
    ref = np.einsum('btk,btn->kn', xs, dy.astype(np.float64).reshape(B, T, N))
    np.testing.assert_allclose(dw.numpy(), ref, atol=2e-6 * np.abs(ref).max(), rtol=0)
