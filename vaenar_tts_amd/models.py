"""models.VAENAR mirror (/root/reference/models/models.py:9-226) on the MI355X engine.

Same constructor reads of ``hps`` and the same public attributes the reference scripts reach
into (inference.py:129-142: text_encoder, length_predictor, prior, decoder, mel_text_len_ratio).
Returned tensors are device arrays exposing ``.numpy()``.
"""
import numpy as np

from ._lib import Engine, check
from .modules import (TransformerEncoder, TransformerDecoder, TransformerPrior, TransformerPosterior,
                      DenseLengthPredictor)


class VAENAR:
    def __init__(self, hps, name='VAENAR', device=0, weights=None, **kwargs):
        self.name = name
        self.hps = hps
        self.engine = Engine(hps, device)
        eng = self.engine
        self.n_sample = hps.Train.num_samples                                   # models.py:13
        self.mel_text_len_ratio = hps.Common.mel_text_len_ratio                 # models.py:14
        self.max_reduction_factor = hps.Common.max_reduction_factor             # models.py:15
        e = hps.Encoder.Transformer
        self.text_encoder = TransformerEncoder(                                 # models.py:16-30
            vocab_size=e.vocab_size, embd_dim=e.embd_dim, pre_nconv=e.n_conv, pre_hidden=e.pre_hidden,
            pre_conv_kernel=e.conv_kernel, pre_activation=e.pre_activation,
            prenet_drop_rate=e.pre_drop_rate, bn_before_act=e.bn_before_act,
            pos_drop_rate=e.pos_drop_rate, nblk=e.n_blk, attention_dim=e.attention_dim,
            attention_heads=e.attention_heads, attention_temperature=e.attention_temperature,
            ffn_hidden=e.ffn_hidden, engine=eng)
        d = hps.Decoder.Transformer
        self.decoder = TransformerDecoder(                                      # models.py:31-43
            nblk=d.nblk, attention_dim=d.attention_dim, attention_heads=d.attention_heads,
            temperature=d.attention_temperature, ffn_hidden=d.ffn_hidden, post_n_conv=d.post_n_conv,
            post_conv_filters=d.post_conv_filters, post_conv_kernel=d.post_conv_kernel,
            post_drop_rate=d.post_drop_rate, out_dim=hps.Common.output_dim,
            max_reduction_factor=hps.Common.max_reduction_factor, name='transformer_decoder', engine=eng)
        self.length_predictor = DenseLengthPredictor(                           # models.py:44-45
            activation=hps.LengthPredictor.Dense.activation, engine=eng)
        q = hps.Posterior.Transformer
        self.posterior = TransformerPosterior(                                  # models.py:46-56
            pre_hidden=q.pre_hidden, pos_drop_rate=q.pos_drop_rate, pre_drop_rate=q.pre_drop_rate,
            pre_activation=q.pre_activation, nblk=q.nblk, attention_dim=q.attention_dim,
            attention_heads=q.attention_heads, temperature=q.temperature, ffn_hidden=q.ffn_hidden,
            latent_dim=hps.Common.latent_dim, engine=eng)
        p = hps.Prior.Transformer
        self.prior = TransformerPrior(                                          # models.py:57-65
            n_blk=p.n_blk, channels=hps.Common.latent_dim, n_transformer_blk=p.n_transformer_blk,
            attention_dim=p.attention_dim, attention_heads=p.attention_heads,
            temperature=p.temperature, ffn_hidden=p.ffn_hidden, inverse=p.inverse, engine=eng)
        self.posterior.noise = self.prior           # one device noise stream for both tf.random.normal sites (prior.py:35, posterior.py:35)
        self._len_cache = {}
        if weights is not None:
            self.load_weights(weights)

    # weights (replaces tf.train.Checkpoint(model=model).restore, inference.py:122-123) -------------
    def load_weights(self, weights):
        """``weights``: {path: ndarray} (vaenar_tts_amd.weights), a path to an ``.npz`` of it, or the prefix of a TensorFlow
        checkpoint written by the reference's train.py (``<prefix>.index`` + ``.data-*``, read by tf_checkpoint.py)."""
        if isinstance(weights, str):
            import os
            if os.path.exists(weights + ".index"):
                from .tf_checkpoint import load_model_weights
                weights = load_model_weights(weights, self.hps, strict=False)
            else:
                from .weights import load_npz
                weights = load_npz(weights)
        self.engine.load_weights(weights)

    # models.py:199-210 ----------------------------------------------------------------------------
    def inference(self, inputs, mel_lengths, text_lengths=None, reduction_factor=2, eps=None,
                  temperature=1.0, return_alignments=True, fused=True):
        """VAENAR.inference: (predicted_mel [B, Tz*rf, out_dim], {decoder-attention-i: alignments}).
        ``eps``: injected prior noise (already times temperature) for parity runs; default: drawn
        like prior.sample's tf.random.normal with temperature 1.0 (prior.py:154).
        ``fused=True`` runs the whole path as ONE library call on one stream (vnr_inference);
        ``fused=False`` goes module by module exactly like models.py:201-209."""
        eng = self.engine
        rf = int(reduction_factor)
        ml = mel_lengths.numpy() if hasattr(mel_lengths, "numpy") and not isinstance(mel_lengths, np.ndarray) \
            else np.asarray(mel_lengths)
        reduced = ((ml.astype(np.int64) + rf - 1) // rf).astype(np.int32)       # models.py:200
        pos_step = np.float32(self.mel_text_len_ratio) / np.float32(rf)          # models.py:201
        if not fused:
            text_embd = self.text_encoder(inputs, text_lengths, pos_step=pos_step, training=False)
            z, _ = self.prior.sample(reduced, text_embd, text_lengths, training=False,
                                     temperature=temperature, eps=eps)
            _, mel, ali = self.decoder(inputs=z, text_embd=text_embd, z_lengths=reduced,
                                       text_lengths=text_lengths, training=False, reduction_factor=rf,
                                       return_alignments=return_alignments)
            return mel, ali
        ids = eng.asarray(inputs, np.int32)
        B, Tt = ids.shape
        Tz = int(reduced.max())
        tl = eng.asarray(np.full(B, Tt, np.int32) if text_lengths is None else text_lengths, np.int32)
        # the reduced lengths live on the device; identical length vectors reuse the resident copy so a
        # steady-state call issues no host->device copy (and therefore no stream synchronisation)
        key = (reduced.tobytes(), rf)
        rl = self._len_cache.get(key)
        if rl is None:
            if len(self._len_cache) > 64:
                self._len_cache.clear()
            rl = self._len_cache[key] = eng.to_device(reduced, np.int32)
        C = self.hps.Common.latent_dim
        if eps is None and float(temperature) != 0.0:
            eps = self.prior.draw((B, Tz, C), float(temperature))        # on the device (prior.py:35)
        eps_d = None if eps is None else eng.asarray(eps, np.float32)
        if eps_d is not None:
            assert eps_d.shape == (B, Tz, C), (eps_d.shape, (B, Tz, C))
        dec = self.decoder
        mel = eng.empty((B, Tz * rf, dec.out_dim))
        ali = eng.empty((dec.nblk, B, dec.heads, Tz, Tt)) if return_alignments else None
        eng.call("vnr_inference", ids.ptr, tl.ptr, rl.ptr, B, Tt, Tz, rf, float(pos_step),
                                    None if eps_d is None else eps_d.ptr, mel.ptr,
                                    None if ali is None else ali.ptr, None)
        alignments = {}
        if ali is not None:
            n = B * dec.heads * Tz * Tt
            for i, nm in enumerate(dec.block_names):
                alignments[nm] = ali.view(i * n, (B, dec.heads, Tz, Tt))
        return mel, alignments

    def __call__(self, inputs, mel_targets, mel_lengths, text_lengths=None, reduction_factor=2,
                 training=None, reduce_loss=None, eps=None, return_alignments=True, dropout_seed=0):
        """VAENAR.call (models.py:105-197), forward.  training=False is the dev_step of train.py:148-155;
        training=True is the forward half of train_step (train.py:130-134): Dropout active (counter-based masks
        keyed by ``dropout_seed``; oracle/vaenar_numpy.py reproduces them bit for bit), BatchNormalization on batch
        statistics with the moving statistics updated in the engine's weight store.
        Returns (decoded_outs [B,Tm,out_dim], l2_loss, kl_divergence, length_loss, dec_alignments); with
        reduce_loss the three losses are means over the batch (models.py:84,92,101), otherwise [B] vectors.
        ``eps`` [B,n_sample,Tz,C] (or [B,Tz,C] when n_sample = 1) replaces tf.random.normal of posterior.reparameterize
        (posterior.py:35); default: drawn on the device (prior.draw).  With ``self.n_sample`` > 1 (hps.Train.num_samples) the
        outputs follow models.py:146-197: decoded_outs and the alignments have batch * n_sample rows (sample index inner), the
        per-utterance l2 / kl are means over the samples.  (The backward pass and Adam: ``train_step``.)"""
        eng = self.engine
        ns = int(self.n_sample)                                                 # models.py:13 (hps.Train.num_samples)
        rf = int(reduction_factor)
        ids = eng.asarray(inputs, np.int32)
        B, Tt = ids.shape
        mel = eng.asarray(mel_targets, np.float32)
        Tm = mel.shape[1]
        ml_h = mel_lengths.numpy() if hasattr(mel_lengths, "numpy") and not isinstance(mel_lengths, np.ndarray) \
            else np.asarray(mel_lengths)
        ml = eng.to_device(ml_h.astype(np.int32), np.int32)
        rl = eng.to_device(((ml_h.astype(np.int64) + rf - 1) // rf).astype(np.int32), np.int32)     # models.py:125
        tl = eng.asarray(np.full(B, Tt, np.int32) if text_lengths is None else text_lengths, np.int32)
        Tz = (Tm + rf - 1) // rf
        C = self.hps.Common.latent_dim
        # posterior.reparameterize draws eps [batch, n_sample, time, dim] (posterior.py:35): everything after the posterior runs on
        # batch * n_sample rows (sample index inner, models.py:146-178) -- the engine tiles text encoding, targets and lengths itself
        if eps is None:
            eps = self.prior.draw((B, ns, Tz, C))                          # on the device (posterior.py:35 / prior.py:35)
        if hasattr(eps, "ptr"):
            eps_d = eps
            assert eps_d.size == B * ns * Tz * C, (eps_d.shape, (B, ns, Tz, C))
        else:
            eps_d = eng.asarray(np.asarray(eps, np.float32).reshape(B, ns, Tz, C), np.float32)
        pos_step = np.float32(self.mel_text_len_ratio) / np.float32(rf)          # models.py:128
        dec = self.decoder
        outs = eng.empty((B * ns, Tm, dec.out_dim))
        l2, kl, ll = eng.empty((B,)), eng.empty((B,)), eng.empty((B,))
        ali = eng.empty((dec.nblk, B * ns, dec.heads, Tz, Tt)) if return_alignments else None
        aux = eng.empty((3, B) if ns == 1 else (B + 2 * B * ns,))
        eng.set_option("n_sample", ns)
        eng.set_option("training", 1 if training else 0)
        try:
            if training:
                eng.set_option("dropout_seed", int(dropout_seed) & 0x7FFFFFFF)
            eng.call("vnr_elbo_fwd", ids.ptr, tl.ptr, mel.ptr, ml.ptr, rl.ptr, B, Tt, Tm, rf, float(pos_step),
                                       eps_d.ptr, outs.ptr, l2.ptr, kl.ptr, ll.ptr, None if ali is None else ali.ptr,
                                       aux.ptr)
        finally:                       # a failing call must not leave the handle in training mode (the next inference would run
            eng.set_option("training", 0)      # with Dropout and batch statistics)
            eng.set_option("n_sample", 1)
        self.last_aux = aux           # predicted lengths [B] | posterior log-probs [B * n_sample] | prior log-probs [B * n_sample]
        alignments = {}
        if ali is not None:
            n = B * ns * dec.heads * Tz * Tt
            for i, nm in enumerate(dec.block_names):
                alignments[nm] = ali.view(i * n, (B * ns, dec.heads, Tz, Tt))
        if reduce_loss:
            l2, kl, ll = (np.float32(t.numpy().mean(dtype=np.float32)) for t in (l2, kl, ll))
        return outs, l2, kl, ll, alignments

    call = __call__

    def init(self, text_inputs, mel_lengths, text_lengths=None, eps=None, dropout_seed=0):
        """VAENAR.init (models.py:212-226; init_step of train.py:176-179): text encoder with training=True ->
        TransformerPrior.init (every ActNorm takes log_scale / bias from the statistics of its input, flow.py:189-196)
        -> decoder at max_reduction_factor.  Returns the predicted mel; the ActNorm variables and the BN moving
        statistics in the engine's weight store are updated (read them with ``get_weights``).
        ``eps`` [B,Tz,C] replaces tf.random.normal of prior._initial_sample (prior.py:31)."""
        eng = self.engine
        rf = int(self.max_reduction_factor)
        ids = eng.asarray(text_inputs, np.int32)
        B, Tt = ids.shape
        ml_h = mel_lengths.numpy() if hasattr(mel_lengths, "numpy") and not isinstance(mel_lengths, np.ndarray) \
            else np.asarray(mel_lengths)
        red = ((ml_h.astype(np.int64) + rf - 1) // rf).astype(np.int32)          # models.py:213
        Tz = int(red.max())
        rl = eng.to_device(red, np.int32)
        tl = eng.asarray(np.full(B, Tt, np.int32) if text_lengths is None else text_lengths, np.int32)
        C = self.hps.Common.latent_dim
        if eps is None:
            eps = self.prior.draw((B, Tz, C))                              # on the device (posterior.py:35 / prior.py:35)
        eps_d = eng.asarray(np.asarray(eps, np.float32).reshape(B, Tz, C) if not hasattr(eps, "ptr") else eps, np.float32)
        pos_step = np.float32(self.mel_text_len_ratio) / np.float32(rf)          # models.py:214
        mel = eng.empty((B, Tz * rf, self.decoder.out_dim))
        eng.set_option("dropout_seed", int(dropout_seed) & 0x7FFFFFFF)
        eng.call("vnr_init", ids.ptr, tl.ptr, rl.ptr, B, Tt, Tz, float(pos_step), eps_d.ptr, mel.ptr, record=False)
        self._len_cache = {}
        return mel

    def train_step(self, texts, mels, t_lengths, m_lengths, kl_weight, reduction_factor, eps=None, dropout_seed=0,
                   learning_rate=None, apply_update=True):
        """train_step of train.py:127-138: returns (loss, mel_l2, kl_divergence, length_l2) as floats after one
        Adam update (``apply_update=False``: gradients only, see ``gradients``).  Hyper-parameters from
        hps.Train (learning_rate, length_weight); Adam beta_1 0.9, beta_2 0.999, epsilon 1e-7 (train.py:116-117)."""
        eng = self.engine
        rf = int(reduction_factor)
        ids = eng.asarray(texts, np.int32)
        B, Tt = ids.shape
        mel = eng.asarray(mels, np.float32)
        Tm = mel.shape[1]
        ml_h = np.asarray(m_lengths.numpy() if hasattr(m_lengths, "numpy") and not isinstance(m_lengths, np.ndarray) else m_lengths)
        ml = eng.to_device(ml_h.astype(np.int32), np.int32)
        rl = eng.to_device(((ml_h.astype(np.int64) + rf - 1) // rf).astype(np.int32), np.int32)
        tl = eng.asarray(np.full(B, Tt, np.int32) if t_lengths is None else t_lengths, np.int32)
        Tz = (Tm + rf - 1) // rf
        C = self.hps.Common.latent_dim
        ns = int(self.n_sample)
        if eps is None:
            eps = self.prior.draw((B, ns, Tz, C))                          # on the device (posterior.py:35 / prior.py:35)
        eps_d = eng.asarray(np.asarray(eps, np.float32).reshape(B, ns, Tz, C) if not hasattr(eps, "ptr") else eps, np.float32)
        assert eps_d.size == B * ns * Tz * C, (eps_d.shape, (B, ns, Tz, C))
        pos_step = np.float32(self.mel_text_len_ratio) / np.float32(rf)
        tr = self.hps.Train
        lr = tr.learning_rate if learning_rate is None else learning_rate
        scal = np.zeros(4, np.float32)
        eng.set_option("dropout_seed", int(dropout_seed) & 0x7FFFFFFF)
        eng.set_option("n_sample", ns)
        try:
            eng.call("vnr_train_step", ids.ptr, tl.ptr, mel.ptr, ml.ptr, rl.ptr, B, Tt, Tm, rf, float(pos_step),
                                         eps_d.ptr, float(kl_weight), float(tr.length_weight), float(lr), 0.9, 0.999, 1e-7,
                                         1 if apply_update else 0, scal.ctypes.data, record=False)
        finally:
            eng.set_option("n_sample", 1)
        self._len_cache = {}
        return float(scal[3]), float(scal[0]), float(scal[1]), float(scal[2])

    def gradients(self, paths=None):
        """{path: d loss / d variable} of the last train_step (tape.gradient, train.py:136)."""
        from .weights import weight_spec, is_trainable
        spec = weight_spec(self.hps)
        out = {}
        for k in (paths or [p for p in spec if is_trainable(p)]):
            sh = spec[k]
            out[k] = self.engine.get_gradient(k, sh if len(sh) else (1,)).reshape(sh)
        return out

    # optimizer state: the `optimizer` part of tf.train.Checkpoint(step, optimizer, model) (train.py:246-255) ------------
    def get_optimizer_state(self):
        """({path: Adam m}, {path: Adam v}, iterations) of every trainable variable (zeros / 0 before the first step)."""
        from .weights import weight_spec, is_trainable
        spec = weight_spec(self.hps)
        m, v = {}, {}
        for k in (p for p in spec if is_trainable(p)):
            sh = spec[k]
            m[k] = self.engine.get_optimizer_slot(k, "m", sh if len(sh) else (1,)).reshape(sh)
            v[k] = self.engine.get_optimizer_slot(k, "v", sh if len(sh) else (1,)).reshape(sh)
        return m, v, self.engine.get_optimizer_step()

    def set_optimizer_state(self, m, v, iterations):
        """Restore Adam's slots and iteration counter (after ``load_weights``): the next train_step continues the
        interrupted run -- same moments, same bias correction."""
        for k, a in m.items():
            self.engine.set_optimizer_slot(k, "m", a)
        for k, a in v.items():
            self.engine.set_optimizer_slot(k, "v", a)
        self.engine.set_optimizer_step(int(iterations))

    def save_checkpoint(self, prefix, step=0, save_counter=1):
        """``manager.save()`` of train.py:262,301: TensorFlow tensor bundle ``<prefix>.index`` / ``.data-*`` with the model
        variables, Adam's slots and counters, the epoch counter ``step`` (tf_checkpoint.save_training_checkpoint)."""
        from .tf_checkpoint import save_training_checkpoint
        m, v, it = self.get_optimizer_state()
        save_training_checkpoint(prefix, self.get_weights(), m, v, iterations=it, step=step, save_counter=save_counter,
                                 learning_rate=self.hps.Train.learning_rate)
        return prefix

    def restore_checkpoint(self, prefix, strict=True):
        """``checkpoint.restore(manager.latest_checkpoint)`` of train.py:249: variables, optimizer slots and counters.
        Returns the stored epoch counter ``step`` (train.py:252).  The MODEL variables must all be there with the shapes of
        this configuration (a bundle of another config / dataset, or a truncated one, raises instead of leaving variables at
        their random initial values; ``strict=False`` downgrades that to a warning).  Bundles without optimizer entries restore
        the model only -- and say so, because Adam then restarts its moments and bias correction."""
        from .tf_checkpoint import load_training_checkpoint, check_training_checkpoint
        ck = load_training_checkpoint(prefix, self.hps, strict=False)
        weights, opt = check_training_checkpoint(ck, self.hps, prefix, strict)
        self.engine.load_weights(weights)
        if opt is not None:
            self.set_optimizer_state(*opt)
        self._len_cache = {}
        return ck["step"] or 0

    # model.trainable_variables (train.py:136-137) ---------------------------------------------------------------------------
    @property
    def trainable_variables(self):
        """The list `tape.gradient(loss, model.trainable_variables)` / `optimizer.apply_gradients(zip(...))` iterate
        (train.py:136-137), in Keras order: views with ``.name``, ``.shape``, ``.numpy()``, ``.assign()`` (and ``.gradient()``
        after a ``train_step``).  The BatchNormalization moving statistics are not in it (``non_trainable_variables``)."""
        from .variables import model_variables
        return model_variables(self.engine, self.hps, True, None, self.engine.has_posterior())

    @property
    def variables(self):
        from .variables import model_variables
        return model_variables(self.engine, self.hps, False, None, self.engine.has_posterior())

    @property
    def non_trainable_variables(self):
        return [v for v in self.variables if not v.trainable]

    trainable_weights = trainable_variables
    weights = variables

    def get_weights(self, paths=None):
        """{path: ndarray} read back from the engine (after init / training-mode forwards)."""
        from .weights import weight_spec
        spec = weight_spec(self.hps)
        have = self.engine._weights_loaded
        return {k: eng_get(self.engine, k, spec[k]) for k in (paths or [p for p in spec if not have or p in have])}


def eng_get(engine, path, shape):
    return engine.get_weight(path, shape if len(shape) else (1,)).reshape(shape)
