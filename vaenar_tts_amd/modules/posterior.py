"""TransformerPosterior mirror (/root/reference/modules/posterior.py:10-138)."""
import numpy as np

from ._base import EngineModule, check


class TransformerPosterior(EngineModule):
    var_prefix = "posterior"

    def __init__(self, pre_hidden, pre_drop_rate, pre_activation, pos_drop_rate, nblk, attention_dim,
                 attention_heads, temperature, ffn_hidden, latent_dim, name='TransformerPosterior',
                 engine=None):
        super().__init__(name, engine)
        self.latent_dim = latent_dim
        self.noise = None            # the model's device noise stream (TransformerPrior.draw: tf.random.normal's role), set by VAENAR

    def __call__(self, inputs, src_enc, src_lengths=None, target_lengths=None, training=None, dropout_seed=None):
        """posterior.py:115-130 -> (mu_projection output, logvar_projection output, None).
        NB models.py:136 unpacks this as ``logvar, mu, _``."""
        e = self.engine
        x = self._f32(inputs)
        mem = self._f32(src_enc)
        B, Tz, _ = x.shape
        Tt = mem.shape[1]
        sl = self._i32(src_lengths, B, Tt)
        tl = self._i32(target_lengths, B, Tz)
        mu = e.empty((B, Tz, self.latent_dim))
        logvar = e.empty((B, Tz, self.latent_dim))
        with self._training(training, dropout_seed):
            e.call("vnr_posterior_fwd", x.ptr, mem.ptr, sl.ptr, tl.ptr, B, Tz, Tt, mu.ptr, logvar.ptr)
        return mu, logvar, None

    call = __call__

    # BasePosterior.reparameterize (posterior.py:21-39) ------------------------------------------------------------------
    def reparameterize(self, mu, logvar, nsamples=1, random=True, eps=None):
        """samples, eps -- both [batch, nsamples, max_time, dim] (device).  ``eps`` injects the noise (parity runs); otherwise it
        is tf.random.normal's counterpart drawn on the device (``random=True``) or zeros (``random=False``, posterior.py:37)."""
        e = self.engine
        m, lv = self._f32(mu), self._f32(logvar)
        B, T, C = m.shape
        ns = int(nsamples)
        assert lv.shape == m.shape and C == self.latent_dim, (m.shape, lv.shape)
        if eps is not None:
            eps_d = self._f32(eps)
            assert eps_d.size == B * ns * T * C, (eps_d.shape, (B, ns, T, C))
            if eps_d.shape != (B, ns, T, C):
                eps_d = eps_d.view(0, (B, ns, T, C))
        elif random:
            eps_d = self._draw((B, ns, T, C))
        else:
            eps_d = e.zeros((B, ns, T, C))
        samples = e.empty((B, ns, T, C))
        e.call("vnr_posterior_reparameterize", m.ptr, lv.ptr, eps_d.ptr, B, ns, T, samples.ptr)
        return samples, eps_d

    # BasePosterior.log_probability (posterior.py:42-72) -----------------------------------------------------------------
    def log_probability(self, mu, logvar, z=None, eps=None, seq_lengths=None, epsilon=1e-8):
        """[batch, nsamples] log-probabilities (device) of samples ``z`` or of their noises ``eps`` (both [batch, nsamples,
        max_time, dim]): ``eps`` wins when both are given (posterior.py:59-61)."""
        e = self.engine
        m, lv = self._f32(mu), self._f32(logvar)
        B, T, C = m.shape
        src = eps if eps is not None else z
        if src is None:
            raise ValueError("log_probability needs z or eps")
        sd = self._f32(src)
        assert len(sd.shape) == 4 and sd.shape[0] == B and sd.shape[2:] == (T, C), (sd.shape, m.shape)
        ns = sd.shape[1]
        lens = None if seq_lengths is None else self._i32(seq_lengths)
        out = e.empty((B, ns))
        e.call("vnr_posterior_log_probability", m.ptr, lv.ptr, None if eps is not None else sd.ptr,
                                                  sd.ptr if eps is not None else None, self._ptr(lens), B, ns, T,
                                                  float(epsilon), out.ptr)
        return out

    # TransformerPosterior.sample (posterior.py:132-138) -----------------------------------------------------------------
    def sample(self, inputs, src_enc, input_lengths, src_lengths, nsamples=1, random=True, training=None, eps=None,
               dropout_seed=None):
        """(samples [batch, nsamples, tgt_max_time, dim], log-probabilities [batch, nsamples]) -- the documented contract of
        BasePosterior.sample (posterior.py:74-87): call -> reparameterize -> log_probability of the drawn noise over the valid
        frames.  The reference's own body (posterior.py:134-138) cannot execute and nothing calls it (VAENAR.call uses
        reparameterize / log_probability directly, models.py:141-144): it passes its arguments positionally, so
        ``input_lengths`` (int32 [batch]) lands in log_probability's ``eps`` slot and ``eps ** 2.`` raises, and ``self.call``
        receives the two length vectors swapped.  This method implements what the docstring there specifies."""
        mu, logvar, _ = self(inputs, src_enc, src_lengths=src_lengths, target_lengths=input_lengths, training=training,
                             dropout_seed=dropout_seed)
        samples, eps_d = self.reparameterize(mu, logvar, nsamples, random, eps=eps)
        log_probs = self.log_probability(mu, logvar, eps=eps_d, seq_lengths=input_lengths)
        return samples, log_probs

    def _draw(self, shape):
        if self.noise is not None:
            return self.noise.draw(shape)
        return self.engine.random_normal(shape, 0, 0, 1.0)
