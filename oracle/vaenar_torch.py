"""Autograd restatement of the training step (TEST INFRASTRUCTURE ONLY -- never imported by the product).

oracle/vaenar_numpy.py is the float64 specification of the forward pass; this file restates the SAME functions on
torch CPU float64 tensors so that torch.autograd supplies the gradients of
    loss = mel_l2 + kl_weight * max(kl, 0) + length_weight * length_l2          (train.py:130-136)
with respect to every trainable variable, and the Keras Adam update (train.py:116-117).  The forward of this file is
pinned against vaenar_numpy.py (tests/test_oracle_torch.py: agreement to 1e-9), which in turn is pinned against the
reference's own Python on the tensorflow shim (tests/golden/refshim_*).  PARITY UNPINNED against real TensorFlow
kernels, like the NumPy oracle (the reference has no tests or fixtures and cannot run here).

Every function cites the reference lines it follows (same citations as vaenar_numpy.py).
"""
import math

import numpy as np
import torch

from . import vaenar_numpy as vn

F64 = torch.float64
_DT = [torch.float64]          # working dtype (float64 = specification; float32 only for the timed CPU baseline)


def _t(a):
    return torch.as_tensor(np.asarray(a, np.float64), dtype=_DT[0])


# ReLU sites.  A hidden unit whose pre-activation lies within float32 rounding of zero has no well-defined mask for an fp32 implementation
# (it follows the last bits of the forward pass), and one flipped mask changes that layer's kernel gradients by ~4e-3 of their maximum
# and everything upstream by ~1e-3.  So that tests can tell such a flip from a defect, the oracle (i) records the pre-activations of
# every ReLU site it evaluates, keyed by the path of the bias that feeds the site, and (ii) can be run with chosen units' masks
# inverted (`relu_flips` = {site: [flat indices]}): the gradient an implementation with THAT mask computes (oracle/kinks.py).
_RELU = {"pre": None, "flips": None}


def dense(x, kernel, bias=None, activation=None, site=None):
    y = x @ kernel
    if bias is not None:
        y = y + bias
    return act(y, activation, site)


def act(x, name, site=None):
    if name is None or name == "identity":
        return x
    if name == "relu":
        if site is not None and _RELU["pre"] is not None:
            _RELU["pre"][site] = x.detach()
            flips = (_RELU["flips"] or {}).get(site)
            if flips:
                mask = (x.detach() > 0).reshape(-1).clone()
                idx = torch.as_tensor(list(flips), dtype=torch.long)
                mask[idx] = ~mask[idx]
                return x * mask.reshape(x.shape).to(x.dtype)
        return torch.relu(x)
    if name == "tanh":
        return torch.tanh(x)
    raise ValueError(name)


def layer_norm(x, gamma, beta, eps=vn.LN_EPS):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def conv1d_same(x, kernel, bias):
    """utils.py:61-66 Conv1D('same'): cross-correlation, zero padding (k-1)//2 left."""
    k, cin, cout = kernel.shape
    B, T, _ = x.shape
    left = (k - 1) // 2
    xp = torch.zeros((B, T + k - 1, cin), dtype=x.dtype)
    xp = torch.cat([xp[:, :left], x, xp[:, left + T:]], 1)
    cols = torch.cat([xp[:, j:j + T] for j in range(k)], -1)
    return cols @ kernel.reshape(k * cin, cout) + bias


class TorchOracle:
    def __init__(self, hps, weights, dtype=torch.float64, grad=None):
        """``grad``: track gradients (default: only in float64, the specification; float32 + grad = the quick full-size digest
        of tests/test_gpu_round2.py)."""
        from vaenar_tts_amd.weights import is_trainable
        self.hps = hps
        _DT[0] = dtype
        self.w = {}
        want_grad = (dtype == torch.float64) if grad is None else bool(grad)
        for k, v in weights.items():
            t = _t(v).clone()
            t.requires_grad_(is_trainable(k) and want_grad)
            self.w[k] = t
        self.dropout_seed = None
        self.update_moving_stats = True
        self.last = {}

    def _g(self, p):
        return self.w[p]

    def _pe(self, T, D, step):
        return _t(vn.positional_encoding(T, D, step))

    def _drop(self, x, site, training):
        if training and self.dropout_seed is not None:
            rate = vn.Oracle._drop_rate(self, site)
            if rate > 0:
                keep = vn.dropout_keep(tuple(x.shape), rate, self.dropout_seed, vn.DROPOUT_SITES[site])
                scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))
                return x * _t(keep.astype(np.float64) * scale)
        return x

    # attention.py:217-246
    def mha(self, p, inputs, memory, memory_lengths, query_lengths, causality, num_head, temperature):
        q = inputs @ self._g(f"{p}/query_layer/kernel")
        k = memory @ self._g(f"{p}/key_layer/kernel")
        v = memory @ self._g(f"{p}/value_layer/kernel")
        B, Tq, D = q.shape
        Tk = k.shape[1]
        dh = D // num_head
        qh = q.reshape(B, Tq, num_head, dh).permute(0, 2, 1, 3)
        kh = k.reshape(B, Tk, num_head, dh).permute(0, 2, 1, 3)
        vh = v.reshape(B, Tk, num_head, dh).permute(0, 2, 1, 3)
        logits = qh @ kh.transpose(2, 3)
        logits = logits / math.sqrt(float(dh))
        logits = logits / float(temperature)
        ml = np.full(B, Tk) if memory_lengths is None else np.asarray(memory_lengths)
        ql = np.full(B, Tq) if query_lengths is None else np.asarray(query_lengths)
        mask = vn.sequence_mask(ml, Tk)[:, None, :] & vn.sequence_mask(ql, Tq)[:, :, None]
        if causality:
            mask = mask & np.tril(np.ones((Tq, Tk), dtype=bool))[None]
        mask = torch.as_tensor(mask[:, None])
        logits = torch.where(mask, logits, torch.tensor(float(vn.MASK_FILL), dtype=logits.dtype))
        ali = torch.softmax(logits, -1)
        ctx = (ali @ vh).permute(0, 2, 1, 3).reshape(B, Tq, D)
        return ctx, ali

    def ffn(self, p, x):
        # distance of the closest hidden unit to the ReLU kink: a unit within float32 rounding of zero has no well-defined mask for an
        # fp32 implementation (tests relax the comparison of exactly that layer's gradients, see test_gpu_train.py)
        pre = x @ self._g(f"{p}/dense1/kernel") + self._g(f"{p}/dense1/bias")
        self.last.setdefault("relu_margin", {})[p] = float(pre.detach().abs().min())
        h = dense(x, self._g(f"{p}/dense1/kernel"), self._g(f"{p}/dense1/bias"), "relu", site=f"{p}/dense1/bias")
        o = dense(h, self._g(f"{p}/dense2/kernel"), self._g(f"{p}/dense2/bias"))
        return layer_norm(o + x, self._g(f"{p}/layer_norm/gamma"), self._g(f"{p}/layer_norm/beta"))

    def self_attention_blk(self, p, x, lengths, heads, temperature):
        att, ali = self.mha(f"{p}/attention", x, x, lengths, lengths, False, heads, temperature)
        proj = dense(torch.cat([x, att], -1), self._g(f"{p}/att_proj/kernel"), self._g(f"{p}/att_proj/bias"))
        y = layer_norm(x + proj, self._g(f"{p}/layer_norm/gamma"), self._g(f"{p}/layer_norm/beta"))
        return self.ffn(f"{p}/ffn", y), ali

    def cross_attention_blk(self, p, x, memory, query_lengths, memory_lengths, heads, temperature):
        sa, _ = self.mha(f"{p}/self_attention", x, x, query_lengths, query_lengths, True, heads, temperature)
        y = dense(torch.cat([x, sa], -1), self._g(f"{p}/att_proj1/kernel"), self._g(f"{p}/att_proj1/bias"))
        y = layer_norm(y + x, self._g(f"{p}/layer_norm1/gamma"), self._g(f"{p}/layer_norm1/beta"))
        ca, cross_ali = self.mha(f"{p}/cross_attention", y, memory, memory_lengths, query_lengths, False, heads,
                                 temperature)
        o = dense(torch.cat([y, ca], -1), self._g(f"{p}/att_proj2/kernel"), self._g(f"{p}/att_proj2/bias"))
        o = layer_norm(o + y, self._g(f"{p}/layer_norm2/gamma"), self._g(f"{p}/layer_norm2/beta"))
        return self.ffn(f"{p}/ffn", o), cross_ali

    # utils.py:76-85
    def conv_bn(self, p, x, activation, training, bn_before_act=False):
        assert not bn_before_act
        y = act(conv1d_same(x, self._g(f"{p}/conv1d/kernel"), self._g(f"{p}/conv1d/bias")), activation, site=f"{p}/conv1d/bias")
        g, b = self._g(f"{p}/bn/gamma"), self._g(f"{p}/bn/beta")
        if training:
            mean = y.mean((0, 1))
            var = ((y - mean) ** 2).mean((0, 1))
            out = (y - mean) / torch.sqrt(var + vn.BN_EPS) * g + b
            if self.update_moving_stats:
                mom = float(np.float32(0.99)); om = float(np.float32(1.0) - np.float32(0.99))
                with torch.no_grad():
                    self.w[f"{p}/bn/moving_mean"] = self._g(f"{p}/bn/moving_mean") * mom + mean.detach() * om
                    self.w[f"{p}/bn/moving_variance"] = self._g(f"{p}/bn/moving_variance") * mom + var.detach() * om
        else:
            inv = g / torch.sqrt(self._g(f"{p}/bn/moving_variance") + vn.BN_EPS)
            out = y * inv + (b - self._g(f"{p}/bn/moving_mean") * inv)
        return self._drop(out, f"{p}/dropout", training)

    # encoder.py:79-93
    def text_encoder(self, ids, lengths, pos_step=1.0, training=False):
        e = self.hps.Encoder.Transformer
        x = self._g("text_encoder/emb_layer/embeddings")[torch.as_tensor(np.asarray(ids, np.int64))]
        for i in range(e.n_conv):
            x = self.conv_bn(f"text_encoder/prenet/conv_stack/{i}", x, e.pre_activation, training, e.bn_before_act)
        x = dense(x, self._g("text_encoder/prenet/projection/kernel"), self._g("text_encoder/prenet/projection/bias"))
        T, D = x.shape[1], x.shape[2]
        x = x + self._g("text_encoder/pos_weight") * self._pe(T, D, pos_step)
        x = self._drop(x, "text_encoder/pe_dropout", training)
        for i in range(e.n_blk):
            x, _ = self.self_attention_blk(f"text_encoder/self_attentions/{i}", x, lengths, e.attention_heads,
                                           e.attention_temperature)
        return x

    # length_predictor.py:35-42
    def length_predictor(self, x, lengths):
        proj = dense(x, self._g("length_predictor/projection/kernel"), self._g("length_predictor/projection/bias"),
                     self.hps.LengthPredictor.Dense.activation)
        mask = _t(vn.sequence_mask(lengths, x.shape[1])[:, :, None])
        return (torch.exp(proj) * mask).sum((1, 2))

    # transform.py:46-59
    def transformer_transform(self, p, z_half, cond, cond_lengths, target_lengths):
        r = self.hps.Prior.Transformer
        x = dense(z_half, self._g(f"{p}/pre_projection/kernel"), self._g(f"{p}/pre_projection/bias"))
        T, D = x.shape[1], x.shape[2]
        x = x + self._g(f"{p}/pos_weight") * self._pe(T, D, 1.0)
        for b in range(r.n_transformer_blk):
            x, _ = self.cross_attention_blk(f"{p}/attentions/{b}", x, cond, target_lengths, cond_lengths,
                                            r.attention_heads, r.temperature)
        return (dense(x, self._g(f"{p}/log_scale_proj/kernel"), self._g(f"{p}/log_scale_proj/bias")),
                dense(x, self._g(f"{p}/shift_proj/kernel"), self._g(f"{p}/shift_proj/bias")))

    # flow.py:177-187, 137-150, 241-257 (backward direction = prior.log_probability)
    def actnorm_backward(self, p, z, lengths):
        ls = self._g(f"{p}/log_scale")
        return (z - self._g(f"{p}/bias")) / (torch.exp(ls) + 1e-8), -_t(lengths) * ls.sum()

    def invlinear_backward(self, p, z, lengths):
        W = self._g(f"{p}/weight")
        Winv = torch.linalg.inv(W)
        return z @ Winv, _t(lengths) * torch.linalg.slogdet(Winv)[1]

    def coupling_backward(self, p, upper, z, cond, z_lengths, cond_lengths):
        half = z.shape[-1] // 2
        lower_pt, upper_pt = z[..., :half], z[..., half:]
        zc, zp = (lower_pt, upper_pt) if upper else (upper_pt, lower_pt)
        log_scale, shift = self.transformer_transform(f"{p}/net", zc, cond, cond_lengths, z_lengths)
        scale = torch.sigmoid(log_scale + 2.0)
        zp = (zp - shift) / (scale + 1e-12)
        mask = _t(vn.sequence_mask(z_lengths, z.shape[1])[:, :, None])
        logdet = -(torch.log(scale) * mask).sum((1, 2))
        out = torch.cat([zc, zp], -1) if upper else torch.cat([zp, zc], -1)
        return out, logdet

    # flow.py:166-175, 123-135, 223-239: the _forward passes -- what bwd_pass runs when the flows were built with inverse = True
    def actnorm_forward(self, p, z, lengths):
        ls = self._g(f"{p}/log_scale")
        return z * torch.exp(ls) + self._g(f"{p}/bias"), _t(lengths) * ls.sum()

    def invlinear_forward(self, p, z, lengths):
        W = self._g(f"{p}/weight")
        return z @ W, _t(lengths) * torch.linalg.slogdet(W)[1]

    def coupling_forward(self, p, upper, z, cond, z_lengths, cond_lengths):
        half = z.shape[-1] // 2
        lower_pt, upper_pt = z[..., :half], z[..., half:]
        zc, zp = (lower_pt, upper_pt) if upper else (upper_pt, lower_pt)
        log_scale, shift = self.transformer_transform(f"{p}/net", zc, cond, cond_lengths, z_lengths)
        scale = torch.sigmoid(log_scale + 2.0)
        zp = scale * zp + shift
        mask = _t(vn.sequence_mask(z_lengths, z.shape[1])[:, :, None])
        logdet = (torch.log(scale) * mask).sum((1, 2))
        out = torch.cat([zc, zp], -1) if upper else torch.cat([zp, zc], -1)
        return out, logdet

    # prior.py:119-152; BaseFlow.bwd_pass (flow.py:91-113) runs _backward, or _forward when Prior.Transformer.inverse is set (prior.py:81,88-99)
    def prior_log_probability(self, z, cond, z_lengths, cond_lengths):
        inv = bool(getattr(self.hps.Prior.Transformer, "inverse", False))
        eps = z
        accum = torch.zeros(z.shape[0], dtype=z.dtype)
        for s in reversed(range(self.hps.Prior.Transformer.n_blk)):
            p = f"prior/glow/{s}"
            eps, ld = (self.coupling_forward if inv else self.coupling_backward)(f"{p}/2", s % 2 == 0, eps, cond, z_lengths, cond_lengths); accum = accum + ld
            eps, ld = (self.invlinear_forward if inv else self.invlinear_backward)(f"{p}/1", eps, z_lengths); accum = accum + ld
            eps, ld = (self.actnorm_forward if inv else self.actnorm_backward)(f"{p}/0", eps, z_lengths); accum = accum + ld
        logp = -0.5 * (vn.LOG_2PI + eps ** 2)
        mask = _t(vn.sequence_mask(z_lengths, z.shape[1])[:, :, None])
        return (mask * logp).sum((1, 2)) + accum

    # flow.py:166-175, 123-135, 223-239 (forward direction = prior.sample) and models.py:199-210
    def prior_sample(self, lengths, cond, cond_lengths, eps):
        z = eps
        if bool(getattr(self.hps.Prior.Transformer, "inverse", False)):      # BaseFlow.call / fwd_pass swap to the _backward passes (flow.py:36-47,76-90)
            for s in range(self.hps.Prior.Transformer.n_blk):
                p = f"prior/glow/{s}"
                z, _ = self.actnorm_backward(f"{p}/0", z, lengths)
                z, _ = self.invlinear_backward(f"{p}/1", z, lengths)
                z, _ = self.coupling_backward(f"{p}/2", s % 2 == 0, z, cond, lengths, cond_lengths)
            return z
        for s in range(self.hps.Prior.Transformer.n_blk):
            p = f"prior/glow/{s}"
            z = z * torch.exp(self._g(f"{p}/0/log_scale")) + self._g(f"{p}/0/bias")
            z = z @ self._g(f"{p}/1/weight")
            half = z.shape[-1] // 2
            lower_pt, upper_pt = z[..., :half], z[..., half:]
            upper = s % 2 == 0
            zc, zp = (lower_pt, upper_pt) if upper else (upper_pt, lower_pt)
            log_scale, shift = self.transformer_transform(f"{p}/2/net", zc, cond, cond_lengths, lengths)
            zp = torch.sigmoid(log_scale + 2.0) * zp + shift
            z = torch.cat([zc, zp], -1) if upper else torch.cat([zp, zc], -1)
        return z

    def inference(self, ids, mel_lengths, text_lengths, reduction_factor=2, eps=None):
        """VAENAR.inference (models.py:199-210) -- used as the timed CPU baseline (float32, all cores, oneDNN/MKL)."""
        with torch.no_grad():
            reduced = (np.asarray(mel_lengths) + reduction_factor - 1) // reduction_factor
            pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(reduction_factor)
            text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=False)
            z = self.prior_sample(reduced, text_embd, text_lengths, _t(eps))
            _, mel = self.decoder(z, text_embd, reduced, text_lengths, reduction_factor, False)
            return mel

    # decoder.py:181-199
    def decoder(self, z, text_embd, z_lengths, text_lengths, reduction_factor=2, training=False):
        d = self.hps.Decoder.Transformer
        out_dim = self.hps.Common.output_dim
        B, T, _ = z.shape
        x = dense(z, self._g("decoder/pre_projection/kernel"), self._g("decoder/pre_projection/bias"))
        for b in range(d.nblk):
            x, _ = self.cross_attention_blk(f"decoder/attentions/{b}", x, text_embd, z_lengths, text_lengths,
                                            d.attention_heads, d.attention_temperature)
        full = dense(x, self._g("decoder/out_projection/kernel"), self._g("decoder/out_projection/bias"))
        initial = full[:, :, :reduction_factor * out_dim].reshape(B, T * reduction_factor, out_dim)
        r = initial
        for i in range(d.post_n_conv):
            r = self.conv_bn(f"decoder/postnet/conv_stack/{i}", r, "tanh" if i < d.post_n_conv - 1 else "identity", training)
        r = dense(r, self._g("decoder/residual_projection/kernel"), self._g("decoder/residual_projection/bias"))
        return initial, r + initial

    # posterior.py:115-130
    def posterior(self, mels, text_embd, text_lengths, target_lengths, training=False):
        q = self.hps.Posterior.Transformer
        x = dense(mels, self._g("posterior/prenet/dense1/kernel"), self._g("posterior/prenet/dense1/bias"), q.pre_activation,
                  site="posterior/prenet/dense1/bias")
        x = self._drop(x, "posterior/prenet/dropout1", training)
        x = dense(x, self._g("posterior/prenet/dense2/kernel"), self._g("posterior/prenet/dense2/bias"), q.pre_activation,
                  site="posterior/prenet/dense2/bias")
        x = self._drop(x, "posterior/prenet/dropout2", training)
        T, D = x.shape[1], x.shape[2]
        x = x + self._g("posterior/pos_weight") * self._pe(T, D, 1.0)
        x = self._drop(x, "posterior/pe_dropout", training)
        for b in range(q.nblk):
            x, _ = self.cross_attention_blk(f"posterior/attentions/{b}", x, text_embd, target_lengths, text_lengths,
                                            q.attention_heads, q.temperature)
        return (dense(x, self._g("posterior/mu_projection/kernel"), self._g("posterior/mu_projection/bias")),
                dense(x, self._g("posterior/logvar_projection/kernel"), self._g("posterior/logvar_projection/bias")))

    @staticmethod
    def l2_loss(rec, tgt, lengths):
        """models.py:67-86 with n_sample = 1, reduce=False."""
        T = rec.shape[1]
        mask = _t(vn.sequence_mask(lengths, T))
        return (((rec - tgt) ** 2).mean(-1) * mask).sum(-1) / _t(lengths)

    # models.py:105-197 (n_sample = 1), per-utterance losses
    def call(self, ids, mel_targets, mel_lengths, text_lengths, reduction_factor=2, training=False, eps=None):
        rf = reduction_factor
        mel_lengths = np.asarray(mel_lengths)
        mel_targets = _t(mel_targets)
        B, Tm, _ = mel_targets.shape
        reduced_mels = mel_targets[:, ::rf, :]
        reduced_lens = (mel_lengths + rf - 1) // rf
        pos_step = np.float32(self.hps.Common.mel_text_len_ratio) / np.float32(rf)
        text_embd = self.text_encoder(ids, text_lengths, pos_step=pos_step, training=training)
        pred = self.length_predictor(text_embd.detach(), text_lengths)             # tf.stop_gradient, models.py:133
        length_loss = (torch.log(pred) - torch.log(_t(mel_lengths))) ** 2
        logvar, mu = self.posterior(reduced_mels, text_embd, text_lengths, reduced_lens, training)   # head swap :136
        Tz = reduced_mels.shape[1]
        eps = _t(eps).reshape(B, Tz, -1)
        zs = eps * torch.exp(0.5 * logvar) + mu                                     # posterior.py:21-39
        dim = mu.shape[2]
        tl = -0.5 * (dim * vn.LOG_2PI + (logvar + eps ** 2).sum(2))                 # posterior.py:42-72
        post_lp = (_t(vn.sequence_mask(reduced_lens, Tz)) * tl).sum(1)
        initial, outs = self.decoder(zs, text_embd, reduced_lens, text_lengths, rf, training)
        initial, outs = initial[:, :Tm], outs[:, :Tm]
        l2 = self.l2_loss(outs, mel_targets, mel_lengths) + self.l2_loss(initial, mel_targets, mel_lengths)
        prior_lp = self.prior_log_probability(zs, text_embd, reduced_lens, text_lengths)
        kl = post_lp - prior_lp
        self.last.update(text_embd=text_embd, mu=mu, logvar=logvar, samples=zs, post_lp=post_lp, prior_lp=prior_lp)
        return outs, l2, kl, length_loss

    # train.py:127-138
    def train_loss(self, ids, mel_targets, mel_lengths, text_lengths, reduction_factor, eps, kl_weight=1e-5,
                   length_weight=1.0, dropout_seed=0):
        self.dropout_seed = dropout_seed
        outs, l2, kl, ll = self.call(ids, mel_targets, mel_lengths, text_lengths, reduction_factor, True, eps)
        mel_l2, kl_m, len_l2 = l2.mean(), kl.mean(), ll.mean()
        loss = mel_l2 + kl_weight * torch.clamp(kl_m, min=0.0) + length_weight * len_l2
        return loss, mel_l2, kl_m, len_l2

    def gradients(self, *args, relu_flips=None, **kwargs):
        """{path: dloss/dvariable} for every trainable variable (zeros where the graph does not reach).  self.last["relu_pre"] =
        {bias path of the site: pre-activations} of every ReLU the forward evaluated; relu_flips = {site: [flat indices]} inverts the
        masks of those units (see _RELU above)."""
        for t in self.w.values():
            t.grad = None
        _RELU["pre"], _RELU["flips"] = {}, relu_flips
        try:
            loss, mel_l2, kl, len_l2 = self.train_loss(*args, **kwargs)
            self.last["relu_pre"] = _RELU["pre"]
        finally:
            _RELU["pre"], _RELU["flips"] = None, None
        loss.backward()
        g = {k: (t.grad.numpy().copy() if t.grad is not None else np.zeros(tuple(t.shape)))
             for k, t in self.w.items() if t.requires_grad}
        return g, dict(loss=float(loss.detach()), mel_l2=float(mel_l2.detach()), kl=float(kl.detach()), length_l2=float(len_l2.detach()))


def adam_step(weights, grads, m, v, step, lr=1.25e-4, beta1=0.9, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam (non-amsgrad) as used by train.py:116-117: lr_t = lr * sqrt(1-b2^t) / (1-b1^t);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t * m / (sqrt(v) + eps).  NumPy arrays in place; step >= 1."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    for k, g in grads.items():
        m[k] = beta1 * m[k] + (1.0 - beta1) * g
        v[k] = beta2 * v[k] + (1.0 - beta2) * g * g
        weights[k] = weights[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)
