"""CPU ORACLE (test infrastructure, NOT product code) for the VAENAR-TTS text->mel path.

A NumPy restatement of the reference algorithm (thuhcsi/VAENAR-TTS, TF 2.2 /
Keras).  Every function cites the reference file:line it follows.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product (``vaenar_tts_amd``) never does.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures and it
cannot be executed here or on the GPU box (it imports TensorFlow 2.2, which is
not installed and not installable offline).  The
arithmetic lives in third-party TensorFlow 2.2.0 (environment.yml:119-122);
what is restated here is its documented semantics at the
reference's own call sites.  The oracle is cross-checked by (i) analytic
known-answer tests (tests/test_oracle_kat.py) and (ii) fixtures produced by
importing the reference's own ``modules/*.py`` over a ``tensorflow``-named shim
(oracle/tf_shim, this container only) -- see oracle/make_golden.py.

All tensors are ``[batch, time, channels]``; ``dtype`` selects float64 (the
specification) or float32 (the timed CPU baseline).
"""
import math

import numpy as np

MASK_FILL = np.float32(-2.0 ** 32 + 1)   # attention.py:240 -> fp32 -4294967296.0
LN_EPS = 1e-3                            # Keras LayerNormalization default
BN_EPS = 1e-3                            # Keras BatchNormalization default
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- #
# primitives (TF semantics)
# --------------------------------------------------------------------------- #
def sequence_mask(lengths, maxlen=None):
    """tf.sequence_mask: mask[b,t] = t < lengths[b] (attention.py:196,202)."""
    lengths = np.asarray(lengths)
    if maxlen is None:
        maxlen = int(lengths.max())
    return np.arange(maxlen)[None, :] < lengths[:, None]


def positional_encoding(length, dim, step=1.0):
    """PositionalEncoding.positional_encoding (utils.py:333-355).

    The reference evaluates this in float32 op by op; each stage below is
    rounded to float32 so the table is the correctly-rounded float32 result.
    Even channel d: sin(p / 10000^(d/D)); odd d: cos(p / 10000^((d-1)/D));
    p = t * step.
    """
    f32 = np.float32
    pos = (np.arange(length, dtype=f32) * f32(step)).astype(f32)[:, None]        # :340-345
    d = np.arange(dim, dtype=f32)[None, :]                                        # :346-350
    e_even = (d / f32(dim)).astype(f32)
    e_odd = ((d - f32(1)) / f32(dim)).astype(f32)
    w_even = np.power(10000.0, e_even.astype(np.float64)).astype(f32)
    w_odd = np.power(10000.0, e_odd.astype(np.float64)).astype(f32)
    a_even = (pos / w_even).astype(f32)
    a_odd = (pos / w_odd).astype(f32)
    even = (np.arange(dim) % 2 == 0)[None, :]                                     # :351-352
    pe = np.where(even, np.sin(a_even.astype(np.float64)), np.cos(a_odd.astype(np.float64)))
    return pe.astype(f32)


def dense(x, kernel, bias=None, activation=None):
    """tf.keras.layers.Dense on the last axis: act(x @ W + b), W [in, out]."""
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return act(y, activation)


def act(x, name):
    if name is None or name == "identity":
        return x
    if name == "relu":
        return np.maximum(x, 0)
    if name == "tanh":
        return np.tanh(x)
    raise ValueError(name)


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """tf.keras.layers.LayerNormalization(): last axis, population variance."""
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def conv1d_same(x, kernel, bias):
    """tf.keras.layers.Conv1D(padding='same', stride 1): cross-correlation,
    y[b,t,o] = b_o + sum_j sum_c x[b, t+j-k//2, c] W[j,c,o], zeros outside [0,T)."""
    k, cin, cout = kernel.shape
    B, T, _ = x.shape
    left = (k - 1) // 2
    xp = np.zeros((B, T + k - 1, cin), dtype=x.dtype)
    xp[:, left:left + T] = x
    cols = np.concatenate([xp[:, j:j + T] for j in range(k)], axis=-1)   # [B,T,k*cin]
    return cols @ kernel.reshape(k * cin, cout) + bias


def batch_norm_infer(x, gamma, beta, mean, var, eps=BN_EPS):
    """BatchNormalization, inference: (x-mean)*gamma/sqrt(var+eps)+beta."""
    inv = gamma / np.sqrt(var + eps)
    return x * inv + (beta - mean * inv)


def batch_norm_train(x, gamma, beta, eps=BN_EPS):
    """BatchNormalization, training: batch statistics over axes (0,1), padding
    included; returns (y, batch_mean, batch_var)."""
    mean = x.mean((0, 1))
    var = ((x - mean) ** 2).mean((0, 1))
    return (x - mean) / np.sqrt(var + eps) * gamma + beta, mean, var


# ---- counter-based dropout masks (training-mode parity) --------------------------------------------
# The reference draws its masks from TensorFlow's stateful RNG (tf.keras.layers.Dropout), which cannot be reproduced.
# The engine uses a counter-based hash instead (vaenar_tts_amd/csrc/misc.hip: rowop_kernel / mix32, engine.hip:
# site_key); the statements below are bit-identical to it so training-mode parity runs with dropout ON.
DROPOUT_SITES = {"text_encoder/pe_dropout": 8, "posterior/prenet/dropout1": 16, "posterior/prenet/dropout2": 17,
                 "posterior/pe_dropout": 18}
for _i in range(8):
    DROPOUT_SITES["text_encoder/prenet/conv_stack/%d/dropout" % _i] = _i
    DROPOUT_SITES["decoder/postnet/conv_stack/%d/dropout" % _i] = 32 + _i


def _mix32(k):
    """murmur3 finaliser on uint32 arrays (wrap-around arithmetic)."""
    k = np.asarray(k, np.uint32).copy()
    with np.errstate(over="ignore"):
        k ^= k >> np.uint32(16); k *= np.uint32(0x85EBCA6B)
        k ^= k >> np.uint32(13); k *= np.uint32(0xC2B2AE35)
        k ^= k >> np.uint32(16)
    return k


def dropout_site_key(seed, site):
    with np.errstate(over="ignore"):
        return _mix32(np.uint32(seed & 0xFFFFFFFF) ^ (np.uint32(site + 1) * np.uint32(0x9E3779B9)))


def dropout_keep(shape, rate, seed, site):
    """Boolean keep-mask of one Dropout call: element i (row-major) is kept iff
    mix32(i * 0x9E3779B1 + key(seed, site)) >= rate * 2^32."""
    n = int(np.prod(shape))
    rate32 = np.float32(rate)
    thresh = np.uint32(min(np.float32(rate32 * np.float32(4294967296.0)), np.float32(4294967040.0)))
    with np.errstate(over="ignore"):
        k = _mix32(np.arange(n, dtype=np.uint32) * np.uint32(0x9E3779B1) + dropout_site_key(seed, site))
    return (k >= thresh).reshape(shape)


def philox4x32_10(counter, key):
    """Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter [n,4] uint32, key [2]
    uint32 -> [n,4] uint32.  Pinned by the Random123 known-answer vectors (tests/test_oracle_kat.py)."""
    c = np.array(counter, dtype=np.uint64).reshape(-1, 4) & np.uint64(0xFFFFFFFF)
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    M0, M1, W0, W1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        c = np.stack([(p1 >> np.uint64(32)) ^ c[:, 1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[:, 3] ^ k1, p0 & mask], 1)
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return c.astype(np.uint32)


def philox_normal(n, seed, offset=0, stddev=1.0):
    """The device generator behind vnr_random_normal (csrc/misc.hip philox_normal_kernel), restated: element block j =
    elements 4j..4j+3 <- counter (j + offset, 0, 0), key = seed; Box-Muller on ((x >> 8) + 0.5) 2^-24.  float64 arithmetic,
    rounded to fp32 at the end (the kernel works in fp32: agreement to a few 1e-6)."""
    nb = (int(n) + 3) // 4
    ctr = np.arange(nb, dtype=np.uint64) + np.uint64(offset)
    cnt = np.stack([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.zeros(nb, np.uint64), np.zeros(nb, np.uint64)], 1)
    x = philox4x32_10(cnt, (int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)).astype(np.float64)
    u = (np.floor(x / 256.0).astype(np.float32).astype(np.float64) + 0.5) / 16777216.0
    u = u.astype(np.float32).astype(np.float64)                       # the kernel forms u in fp32 (exact: 24-bit integers + 0.5)
    r0, r1 = np.sqrt(-2.0 * np.log(u[:, 0])), np.sqrt(-2.0 * np.log(u[:, 2]))
    t0, t1 = 2.0 * np.pi * u[:, 1], 2.0 * np.pi * u[:, 3]
    z = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], 1).reshape(-1)[:int(n)]
    return (stddev * z).astype(np.float32)


def softmax_last(x):
    """tf.math.softmax: exp(x-max)/sum."""
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------- #
# the model
# --------------------------------------------------------------------------- #
class Oracle:
    """Restatement of models.VAENAR and modules.* driven by a flat weight dict."""

    def __init__(self, hps, weights, dtype=np.float64):
        self.hps = hps
        self.dtype = dtype
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
        self.dropout_masks = None      # {site: keep_mask*scale} injected for training parity
        self.dropout_seed = None       # or: counter-based masks, bit-identical to the engine's (dropout_keep)
        self.update_moving_stats = False   # BatchNormalization(training=True) also assigns moving_mean / moving_variance
        self.last = {}                 # intermediates of the last call (for tests)

    # -- helpers ---------------------------------------------------------------
    def _g(self, path):
        return self.w[path]

    def _pe(self, T, D, step):
        return positional_encoding(T, D, step).astype(self.dtype)

    def _drop(self, x, site, training):
        """Dropout with injectable mask (training only); identity otherwise."""
        if training and self.dropout_masks is not None and site in self.dropout_masks:
            return x * self.dropout_masks[site].astype(self.dtype)
        if training and self.dropout_seed is not None:
            rate = self._drop_rate(site)
            if rate > 0:
                keep = dropout_keep(x.shape, rate, self.dropout_seed, DROPOUT_SITES[site])
                scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))      # tf.keras Dropout: x / (1 - rate)
                return np.where(keep, x * self.dtype(scale), self.dtype(0))
        return x

    def _drop_rate(self, site):
        h = self.hps
        if site.startswith("text_encoder/prenet"):
            return h.Encoder.Transformer.pre_drop_rate
        if site == "text_encoder/pe_dropout":
            return h.Encoder.Transformer.pos_drop_rate
        if site.startswith("decoder/postnet"):
            return h.Decoder.Transformer.post_drop_rate
        if site.startswith("posterior/prenet"):
            return h.Posterior.Transformer.pre_drop_rate
        if site == "posterior/pe_dropout":
            return h.Posterior.Transformer.pos_drop_rate
        raise KeyError(site)

    # -- attention.py ----------------------------------------------------------
    def mha(self, p, inputs, memory, memory_lengths, query_lengths, causality, num_head,
            temperature):
        """MultiHeadScaledProductAttention.call (attention.py:217-246)."""
        q = inputs @ self._g(f"{p}/query_layer/kernel")                   # :218
        k = memory @ self._g(f"{p}/key_layer/kernel")                     # :219
        v = memory @ self._g(f"{p}/value_layer/kernel")                   # :220
        B, Tq, D = q.shape
        Tk = k.shape[1]
        dh = D // num_head
        qh = q.reshape(B, Tq, num_head, dh).transpose(0, 2, 1, 3)         # _split_head :163
        kh = k.reshape(B, Tk, num_head, dh).transpose(0, 2, 1, 3)
        vh = v.reshape(B, Tk, num_head, dh).transpose(0, 2, 1, 3)
        logits = qh @ kh.transpose(0, 1, 3, 2)                            # :224-226
        logits = logits / self.dtype(math.sqrt(float(dh)))                # :227-228
        logits = logits / self.dtype(temperature)                         # :229
        ml = np.full(B, Tk) if memory_lengths is None else np.asarray(memory_lengths)
        ql = np.full(B, Tq) if query_lengths is None else np.asarray(query_lengths)
        mask = sequence_mask(ml, Tk)[:, None, :] & sequence_mask(ql, Tq)[:, :, None]   # :192-209
        if causality:
            mask = mask & np.tril(np.ones((Tq, Tk), dtype=bool))[None]    # :212-215
        logits = np.where(mask[:, None], logits, self.dtype(MASK_FILL))   # :240-241
        ali = softmax_last(logits)                                        # :242
        ctx = ali @ vh                                                    # :243
        ctx = ctx.transpose(0, 2, 1, 3).reshape(B, Tq, D)                 # _merge_head :179
        return ctx, ali

    def ffn(self, p, x):
        """FFN.call (utils.py:48-53)."""
        h = dense(x, self._g(f"{p}/dense1/kernel"), self._g(f"{p}/dense1/bias"), "relu")
        o = dense(h, self._g(f"{p}/dense2/kernel"), self._g(f"{p}/dense2/bias"))
        return layer_norm(o + x, self._g(f"{p}/layer_norm/gamma"), self._g(f"{p}/layer_norm/beta"))

    def self_attention_blk(self, p, x, lengths, heads, temperature):
        """SelfAttentionBLK.call (attention.py:405-415)."""
        att, ali = self.mha(f"{p}/attention", x, x, lengths, lengths, False, heads, temperature)
        ctx = np.concatenate([x, att], -1)                                # :410
        proj = dense(ctx, self._g(f"{p}/att_proj/kernel"), self._g(f"{p}/att_proj/bias"))
        y = layer_norm(x + proj, self._g(f"{p}/layer_norm/gamma"), self._g(f"{p}/layer_norm/beta"))
        return self.ffn(f"{p}/ffn", y), ali

    def cross_attention_blk(self, p, x, memory, query_lengths, memory_lengths, heads, temperature):
        """CrossAttentionBLK.call (attention.py:436-452)."""
        sa, _ = self.mha(f"{p}/self_attention", x, x, query_lengths, query_lengths, True,
                         heads, temperature)                              # :437-439
        ctx = np.concatenate([x, sa], -1)                                 # :440
        y = dense(ctx, self._g(f"{p}/att_proj1/kernel"), self._g(f"{p}/att_proj1/bias"))
        y = layer_norm(y + x, self._g(f"{p}/layer_norm1/gamma"), self._g(f"{p}/layer_norm1/beta"))
        ca, cross_ali = self.mha(f"{p}/cross_attention", y, memory, memory_lengths,
                                 query_lengths, False, heads, temperature)  # :444-446
        ctx = np.concatenate([y, ca], -1)                                 # :447
        o = dense(ctx, self._g(f"{p}/att_proj2/kernel"), self._g(f"{p}/att_proj2/bias"))
        o = layer_norm(o + y, self._g(f"{p}/layer_norm2/gamma"), self._g(f"{p}/layer_norm2/beta"))
        return self.ffn(f"{p}/ffn", o), cross_ali

    # -- utils.py Conv1D / ConvPreNet / PostNet / PreNet -------------------------
    def conv_bn(self, p, x, activation, training, bn_before_act=False):
        """Conv1D.call (utils.py:76-85): conv -> act -> BN (bn_before_act=False)."""
        y = conv1d_same(x, self._g(f"{p}/conv1d/kernel"), self._g(f"{p}/conv1d/bias"))
        g, b = self._g(f"{p}/bn/gamma"), self._g(f"{p}/bn/beta")

        def bn(t):
            if training:
                out, mean, var = batch_norm_train(t, g, b)
                self.last[f"{p}/bn/batch_mean"] = mean
                self.last[f"{p}/bn/batch_var"] = var
                if self.update_moving_stats:      # Keras momentum 0.99 (population variance, non-fused rank-3 path)
                    mom = self.dtype(np.float32(0.99)); om = self.dtype(np.float32(1.0) - np.float32(0.99))
                    self.w[f"{p}/bn/moving_mean"] = self._g(f"{p}/bn/moving_mean") * mom + mean * om
                    self.w[f"{p}/bn/moving_variance"] = self._g(f"{p}/bn/moving_variance") * mom + var * om
                return out
            return batch_norm_infer(t, g, b, self._g(f"{p}/bn/moving_mean"),
                                    self._g(f"{p}/bn/moving_variance"))
        if bn_before_act:
            y = act(bn(y), activation)
        else:
            y = bn(act(y, activation))
        return self._drop(y, f"{p}/dropout", training)

    # -- encoder.py ------------------------------------------------------------
    def text_encoder(self, ids, lengths, pos_step=1.0, training=False):
        """TransformerEncoder.call (encoder.py:79-93)."""
        e = self.hps.Encoder.Transformer
        x = self._g("text_encoder/emb_layer/embeddings")[np.asarray(ids)]          # :81
        for i in range(e.n_conv):                                                  # utils.py:33-38
            x = self.conv_bn(f"text_encoder/prenet/conv_stack/{i}", x, e.pre_activation,
                             training, e.bn_before_act)
        x = dense(x, self._g("text_encoder/prenet/projection/kernel"),
                  self._g("text_encoder/prenet/projection/bias"))
        self.last["prenet_outs"] = x
        T, D = x.shape[1], x.shape[2]
        x = x + self._g("text_encoder/pos_weight") * self._pe(T, D, pos_step)      # :85-86
        x = self._drop(x, "text_encoder/pe_dropout", training)
        for i in range(e.n_blk):                                                   # :89-92
            x, _ = self.self_attention_blk(f"text_encoder/self_attentions/{i}", x, lengths,
                                           e.attention_heads, e.attention_temperature)
        return x

    # -- length_predictor.py ---------------------------------------------------
    def length_predictor(self, x, lengths):
        """DenseLengthPredictor.call (length_predictor.py:35-42)."""
        proj = dense(x, self._g("length_predictor/projection/kernel"),
                     self._g("length_predictor/projection/bias"),
                     self.hps.LengthPredictor.Dense.activation)
        mask = sequence_mask(lengths, x.shape[1])[:, :, None].astype(self.dtype)
        return (np.exp(proj) * mask).sum((1, 2))

    # -- transform.py ----------------------------------------------------------
    def transformer_transform(self, p, z_half, cond, cond_lengths, target_lengths):
        """TransformerTransform.call (transform.py:46-59)."""
        r = self.hps.Prior.Transformer
        x = dense(z_half, self._g(f"{p}/pre_projection/kernel"), self._g(f"{p}/pre_projection/bias"))
        T, D = x.shape[1], x.shape[2]
        x = x + self._g(f"{p}/pos_weight") * self._pe(T, D, 1.0)                   # :51-52
        for b in range(r.n_transformer_blk):                                       # :53-56
            x, _ = self.cross_attention_blk(f"{p}/attentions/{b}", x, cond, target_lengths,
                                            cond_lengths, r.attention_heads, r.temperature)
        log_scale = dense(x, self._g(f"{p}/log_scale_proj/kernel"), self._g(f"{p}/log_scale_proj/bias"))
        shift = dense(x, self._g(f"{p}/shift_proj/kernel"), self._g(f"{p}/shift_proj/bias"))
        return log_scale, shift

    # -- flow.py ---------------------------------------------------------------
    def actnorm_forward(self, p, z, lengths):
        """ActNormFlow._forward (flow.py:166-175)."""
        ls = self._g(f"{p}/log_scale")
        out = z * np.exp(ls) + self._g(f"{p}/bias")
        return out, np.asarray(lengths, self.dtype) * ls.sum()

    def actnorm_backward(self, p, z, lengths, epsilon=1e-8):
        """ActNormFlow._backward (flow.py:177-187)."""
        ls = self._g(f"{p}/log_scale")
        out = (z - self._g(f"{p}/bias")) / (np.exp(ls) + self.dtype(epsilon))
        return out, -np.asarray(lengths, self.dtype) * ls.sum()

    def actnorm_init(self, p, z, lengths, init_scale=1.0, epsilon=1e-8):
        """ActNormFlow.init (flow.py:189-196): statistics over ALL rows incl. padding."""
        C = z.shape[-1]
        flat = z.reshape(-1, C)
        mean, std = flat.mean(0), flat.std(0)
        self.w[f"{p}/log_scale"] = np.log(init_scale / (std + self.dtype(epsilon)))
        self.w[f"{p}/bias"] = -mean / (std + self.dtype(epsilon))
        return self.actnorm_forward(p, z, lengths)

    def invlinear_forward(self, p, z, lengths):
        """InvertibleLinearFlow._forward (flow.py:123-135)."""
        W = self._g(f"{p}/weight")
        logdet = self.dtype(np.float32(np.linalg.slogdet(W.astype(np.float64))[1]))
        return z @ W, np.asarray(lengths, self.dtype) * logdet

    def invlinear_backward(self, p, z, lengths):
        """InvertibleLinearFlow._backward (flow.py:137-150)."""
        W = self._g(f"{p}/weight")
        logdet = self.dtype(np.float32(
            np.linalg.slogdet(np.linalg.inv(W.astype(np.float64)))[1]))
        return z @ np.linalg.inv(W), np.asarray(lengths, self.dtype) * logdet

    def coupling(self, p, upper, z, cond, z_lengths, cond_lengths, backward=False):
        """TransformerCoupling._forward / _backward (flow.py:223-257)."""
        half = z.shape[-1] // 2
        lower_pt, upper_pt = z[..., :half], z[..., half:]                          # _split :212
        zc, zp = (lower_pt, upper_pt) if upper else (upper_pt, lower_pt)           # :228,246
        log_scale, shift = self.transformer_transform(f"{p}/net", zc, cond, cond_lengths, z_lengths)
        scale = sigmoid(log_scale + 2.0)                                           # :231
        if backward:
            zp = (zp - shift) / (scale + self.dtype(1e-12))                        # :220
        else:
            zp = scale * zp + shift                                                # :216
        mask = sequence_mask(z_lengths, z.shape[1])[:, :, None].astype(self.dtype)
        logdet = (np.log(scale) * mask).sum((1, 2))                                # :237
        if backward:
            logdet = -logdet                                                       # :255
        out = np.concatenate([zc, zp], -1) if upper else np.concatenate([zp, zc], -1)   # :238
        return out, logdet

    # -- prior.py --------------------------------------------------------------
    def initial_sample(self, lengths, eps):
        """BasePrior._initial_sample (prior.py:26-42) with injected epsilon
        (already multiplied by the temperature; shape [B, max(lengths), C])."""
        lengths = np.asarray(lengths)
        assert eps.shape[1] == int(lengths.max())
        eps = eps.astype(self.dtype)
        logp = -0.5 * (LOG_2PI + eps ** 2)
        mask = sequence_mask(lengths)[:, :, None].astype(self.dtype)
        return eps, (mask * logp).sum((1, 2))

    def _inverse_flows(self):
        """Prior.Transformer.inverse (prior.py:81,88-99): every flow of the prior is built with this flag, and BaseFlow.call / fwd_pass /
        bwd_pass (flow.py:36-47,76-113) swap _forward and _backward when it is set."""
        return bool(getattr(self.hps.Prior.Transformer, "inverse", False))

    def prior_sample(self, lengths, cond, cond_lengths, eps):
        """TransformerPrior.sample (prior.py:154-169) = TransformerPrior.call (prior.py:101-117): actnorm(z), linear(z) are BaseFlow.call,
        the coupling runs fwd_pass -- with inverse=True flows all three are the _backward passes, in the same step order."""
        inv = self._inverse_flows()
        z, logp = self.initial_sample(lengths, eps)
        for s in range(self.hps.Prior.Transformer.n_blk):
            p = f"prior/glow/{s}"
            z, ld = (self.actnorm_backward if inv else self.actnorm_forward)(f"{p}/0", z, lengths); logp = logp - ld
            z, ld = (self.invlinear_backward if inv else self.invlinear_forward)(f"{p}/1", z, lengths); logp = logp - ld
            z, ld = self.coupling(f"{p}/2", s % 2 == 0, z, cond, lengths, cond_lengths, backward=inv)
            logp = logp - ld
            self.last[f"prior_z_{s}"] = z
        return z, logp

    def prior_init(self, lengths, cond, cond_lengths, eps):
        """TransformerPrior.init (prior.py:171-186): data-dependent ActNorm init.  actnorm.init and coupling.init are called directly
        (flow.py:189-196: statistics, then _forward; flow.py:259-262: _forward) whatever `inverse` says; linear(z) is BaseFlow.call."""
        inv = self._inverse_flows()
        z, logp = self.initial_sample(lengths, eps)
        for s in range(self.hps.Prior.Transformer.n_blk):
            p = f"prior/glow/{s}"
            z, ld = self.actnorm_init(f"{p}/0", z, lengths); logp = logp - ld
            z, ld = (self.invlinear_backward if inv else self.invlinear_forward)(f"{p}/1", z, lengths); logp = logp - ld
            z, ld = self.coupling(f"{p}/2", s % 2 == 0, z, cond, lengths, cond_lengths)
            logp = logp - ld
        return z, logp

    def prior_log_probability(self, z, cond, z_lengths, cond_lengths):
        """TransformerPrior.log_probability (prior.py:119-152): bwd_pass of every flow, steps reversed -- _backward passes, or with
        inverse=True flows the _forward passes."""
        inv = self._inverse_flows()
        eps = z
        accum = np.zeros(z.shape[0], self.dtype)
        for s in reversed(range(self.hps.Prior.Transformer.n_blk)):
            p = f"prior/glow/{s}"
            eps, ld = self.coupling(f"{p}/2", s % 2 == 0, eps, cond, z_lengths, cond_lengths,
                                    backward=not inv); accum = accum + ld
            eps, ld = (self.invlinear_forward if inv else self.invlinear_backward)(f"{p}/1", eps, z_lengths); accum = accum + ld
            eps, ld = (self.actnorm_forward if inv else self.actnorm_backward)(f"{p}/0", eps, z_lengths); accum = accum + ld
        logp = -0.5 * (LOG_2PI + eps ** 2)
        mask = sequence_mask(z_lengths, z.shape[1])[:, :, None].astype(self.dtype)
        return (mask * logp).sum((1, 2)) + accum

    # -- decoder.py ------------------------------------------------------------
    def decoder(self, z, text_embd, z_lengths, text_lengths, reduction_factor=2, training=False):
        """TransformerDecoder.call (decoder.py:181-199)."""
        d = self.hps.Decoder.Transformer
        out_dim = self.hps.Common.output_dim
        B, T, _ = z.shape
        x = dense(z, self._g("decoder/pre_projection/kernel"), self._g("decoder/pre_projection/bias"))
        alignments = {}
        for b in range(d.nblk):                                                    # :188-192
            x, ali = self.cross_attention_blk(f"decoder/attentions/{b}", x, text_embd, z_lengths,
                                              text_lengths, d.attention_heads,
                                              d.attention_temperature)
            alignments[f"decoder-attention-{b}"] = ali
        full = dense(x, self._g("decoder/out_projection/kernel"), self._g("decoder/out_projection/bias"))
        initial = full[:, :, :reduction_factor * out_dim]                          # :193
        initial = initial.reshape(B, T * reduction_factor, out_dim)                # :194-195
        r = initial                                                                # PostNet utils.py:111-115
        for i in range(d.post_n_conv):
            a = "tanh" if i < d.post_n_conv - 1 else "identity"                    # utils.py:103
            r = self.conv_bn(f"decoder/postnet/conv_stack/{i}", r, a, training)
        r = dense(r, self._g("decoder/residual_projection/kernel"),
                  self._g("decoder/residual_projection/bias"))
        return initial, r + initial, alignments                                    # :198-199

    # -- posterior.py ----------------------------------------------------------
    def posterior(self, mels, text_embd, text_lengths, target_lengths, training=False):
        """TransformerPosterior.call (posterior.py:115-130). Returns the two heads in
        the reference's *return order* (mu_projection output, logvar_projection output)."""
        q = self.hps.Posterior.Transformer
        x = dense(mels, self._g("posterior/prenet/dense1/kernel"),
                  self._g("posterior/prenet/dense1/bias"), q.pre_activation)       # utils.py:13-18
        x = self._drop(x, "posterior/prenet/dropout1", training)
        x = dense(x, self._g("posterior/prenet/dense2/kernel"),
                  self._g("posterior/prenet/dense2/bias"), q.pre_activation)
        x = self._drop(x, "posterior/prenet/dropout2", training)
        T, D = x.shape[1], x.shape[2]
        x = x + self._g("posterior/pos_weight") * self._pe(T, D, 1.0)              # :120-121
        x = self._drop(x, "posterior/pe_dropout", training)
        for b in range(q.nblk):
            x, _ = self.cross_attention_blk(f"posterior/attentions/{b}", x, text_embd,
                                            target_lengths, text_lengths, q.attention_heads,
                                            q.temperature)
        mu = dense(x, self._g("posterior/mu_projection/kernel"), self._g("posterior/mu_projection/bias"))
        logvar = dense(x, self._g("posterior/logvar_projection/kernel"),
                       self._g("posterior/logvar_projection/bias"))
        return mu, logvar

    def reparameterize(self, mu, logvar, eps):
        """BasePosterior.reparameterize (posterior.py:21-39); eps [B, n, T, C] injected."""
        std = np.exp(0.5 * logvar)
        return eps * std[:, None] + mu[:, None], eps

    def posterior_log_probability(self, mu, logvar, eps=None, seq_lengths=None, z=None, epsilon=1e-8):
        """BasePosterior.log_probability (posterior.py:42-72): eps given, or z given (then the noise is rebuilt as
        (z - mu) / (std + epsilon), :59-61); seq_lengths None = every frame (:66-68).  [B, nsamples]."""
        dim = mu.shape[2]
        if eps is None:
            eps = (z - mu[:, None]) / (np.exp(0.5 * logvar)[:, None] + epsilon)
        tl = -0.5 * (dim * LOG_2PI + (logvar[:, None] + eps ** 2.0).sum(3))
        mask = (sequence_mask(seq_lengths, mu.shape[1])[:, None, :].astype(self.dtype) if seq_lengths is not None
                else np.ones((mu.shape[0], 1, mu.shape[1]), self.dtype))
        return (mask * tl).sum(2)

    # -- models.py -------------------------------------------------------------
    def inference(self, ids, mel_lengths, text_lengths, reduction_factor=2, eps=None):
        """VAENAR.inference (models.py:199-210); eps = temperature * N(0,1) injected
        (prior.sample default temperature 1.0, prior.py:154)."""
        mel_lengths = np.asarray(mel_lengths)
        reduced = (mel_lengths + reduction_factor - 1) // reduction_factor
        pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(reduction_factor)
        text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=False)
        if eps is None:
            eps = np.zeros((len(reduced), int(reduced.max()), self.hps.Common.latent_dim))
        z, logp = self.prior_sample(reduced, text_embd, text_lengths, eps)
        initial, mel, ali = self.decoder(z, text_embd, reduced, text_lengths, reduction_factor)
        self.last.update(text_embd=text_embd, z=z, prior_logprobs=logp, initial=initial)
        return mel, ali

    def test_step(self, ids, text_lengths, eps_fn=None):
        """test_step of inference.py:128-143 (length predicted, +80 frames head-room).
        eps_fn(B, Tz, C) -> injected temperature*noise; None -> zeros (temperature 0,
        the reference default inference.py:95)."""
        rf = self.hps.Common.final_reduction_factor
        pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(rf)
        text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=False)
        pred = self.length_predictor(text_embd, text_lengths)                      # :133-134
        pred_ml = pred.astype(np.float32).astype(np.int32)                         # :135 trunc
        reduced = (pred_ml + 80 + rf - 1) // rf                                    # :136-137
        B, Tz, C = len(reduced), int(reduced.max()), self.hps.Common.latent_dim
        eps = np.zeros((B, Tz, C)) if eps_fn is None else eps_fn(B, Tz, C)
        z, _ = self.prior_sample(reduced, text_embd, text_lengths, eps)
        _, mel, ali = self.decoder(z, text_embd, reduced, text_lengths, rf)
        self.last.update(text_embd=text_embd, pred_float=pred, z=z)
        return mel, pred_ml + 80, ali

    @staticmethod
    def l2_loss(rec, tgt, lengths, n_sample=1, reduce=True):
        """VAENAR._compute_l2_loss (models.py:67-86)."""
        B, T, D = rec.shape
        r = rec.reshape(-1, n_sample, T, D)
        t = tgt.reshape(-1, n_sample, T, D)
        mask = sequence_mask(lengths, T).astype(rec.dtype).reshape(-1, n_sample, T)
        lens = np.asarray(lengths).reshape(-1, n_sample).astype(rec.dtype)
        l2 = ((((r - t) ** 2).mean(-1) * mask).sum(-1) / lens).mean(-1)
        return l2.mean() if reduce else l2

    def call(self, ids, mel_targets, mel_lengths, text_lengths, reduction_factor=2,
             training=False, reduce_loss=True, eps=None):
        """VAENAR.call (models.py:105-197).  eps [B, n_sample, Tz, C] injected (n_sample = hps.Train.num_samples, models.py:13);
        decoded outputs and alignments have batch * n_sample rows, sample index inner (models.py:146-178)."""
        ns = int(self.hps.Train.num_samples)
        rf = reduction_factor
        mel_lengths = np.asarray(mel_lengths)
        text_lengths = np.asarray(text_lengths)
        mel_targets = np.asarray(mel_targets, self.dtype)
        B, Tm, _ = mel_targets.shape
        reduced_mels = mel_targets[:, ::rf, :]                                     # :123
        reduced_lens = (mel_lengths + rf - 1) // rf                                # :125
        pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(rf)
        text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=training)
        pred = self.length_predictor(text_embd, text_lengths)                      # :132-133
        lsq = (np.log(pred) - np.log(mel_lengths.astype(self.dtype))) ** 2         # :97-103
        length_loss = lsq.mean() if reduce_loss else lsq
        # quirk 1 (models.py:136 vs posterior.py:130): first head is *used as* logvar
        logvar, mu = self.posterior(reduced_mels, text_embd, text_lengths, reduced_lens, training)
        Tz = reduced_mels.shape[1]
        if eps is None:
            eps = np.zeros((B, ns, Tz, self.hps.Common.latent_dim), self.dtype)
        eps = np.asarray(eps, self.dtype).reshape(B, ns, Tz, -1)
        samples, eps = self.reparameterize(mu, logvar, eps)                        # :141
        post_lp = self.posterior_log_probability(mu, logvar, eps, reduced_lens)    # :143-144  [B, ns]
        zs = samples.reshape(B * ns, Tz, -1)                                       # :146-148
        rep = lambda a: np.repeat(np.asarray(a), ns, axis=0)                       # noqa: E731  tile(expand_dims(x, 1), [1, ns, ...]) -> reshape (:150-178)
        b_text, b_tgt, b_ml, b_rl, b_tl = rep(text_embd), rep(mel_targets), rep(mel_lengths), rep(reduced_lens), rep(text_lengths)
        initial, outs, ali = self.decoder(zs, b_text, b_rl, b_tl, rf, training)
        initial, outs = initial[:, :Tm], outs[:, :Tm]                              # :182-183
        l2 = self.l2_loss(outs, b_tgt, b_ml, ns, reduce_loss) + \
            self.l2_loss(initial, b_tgt, b_ml, ns, reduce_loss)                    # :184-188
        prior_lp = self.prior_log_probability(zs, b_text, b_rl, b_tl)
        kl = (post_lp - prior_lp.reshape(B, ns)).mean(1)                           # :89-95
        kl = kl.mean() if reduce_loss else kl
        self.last.update(text_embd=text_embd, mu=mu, logvar=logvar, samples=zs,
                         post_lp=post_lp, prior_lp=prior_lp, initial=initial)
        return outs, l2, kl, length_loss, ali

    def init(self, ids, mel_lengths, text_lengths, eps):
        """VAENAR.init (models.py:212-226); BN runs in training mode (batch stats)."""
        rf = self.hps.Common.max_reduction_factor
        reduced = (np.asarray(mel_lengths) + rf - 1) // rf
        pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(rf)
        text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=True)
        z, _ = self.prior_init(reduced, text_embd, text_lengths, eps)
        _, mel, _ = self.decoder(z, text_embd, reduced, text_lengths, rf, training=True)
        return mel
