"""TransformerPosterior mirror (/root/reference/modules/posterior.py:21-72,90-138)."""
from ._base import EngineModule, check


class TransformerPosterior(EngineModule):
    def __init__(self, pre_hidden, pre_drop_rate, pre_activation, pos_drop_rate, nblk, attention_dim,
                 attention_heads, temperature, ffn_hidden, latent_dim, name='TransformerPosterior',
                 engine=None):
        super().__init__(name, engine)
        self.latent_dim = latent_dim

    def __call__(self, inputs, src_enc, src_lengths=None, target_lengths=None, training=None, dropout_seed=None):
        """posterior.py:115-130 -> (mu_projection output, logvar_projection output, None).
        NB models.py:136 unpacks this as ``logvar, mu, _``."""
        self._set_training(training, dropout_seed)
        e = self.engine
        x = self._f32(inputs)
        mem = self._f32(src_enc)
        B, Tz, _ = x.shape
        Tt = mem.shape[1]
        sl = self._i32(src_lengths, B, Tt)
        tl = self._i32(target_lengths, B, Tz)
        mu = e.empty((B, Tz, self.latent_dim))
        logvar = e.empty((B, Tz, self.latent_dim))
        check(e.lib.vnr_posterior_fwd(e.handle, x.ptr, mem.ptr, sl.ptr, tl.ptr, B, Tz, Tt, mu.ptr, logvar.ptr),
              e.handle)
        self._set_training(False)
        return mu, logvar, None

    call = __call__
