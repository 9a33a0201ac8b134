"""TransformerPrior mirror (/root/reference/modules/prior.py:79-186)."""
import numpy as np

from ._base import EngineModule, check


class TransformerPrior(EngineModule):
    var_prefix = "prior"

    def __init__(self, n_blk, channels, n_transformer_blk, attention_dim, attention_heads,
                 temperature, ffn_hidden, inverse=False, name='GlowPrior', engine=None, **kwargs):
        super().__init__(name, engine)
        # inverse=True (prior.py:88-99: every flow is built with the flag; BaseFlow.call / fwd_pass / bwd_pass, flow.py:36-113, then swap
        # _forward and _backward): sample / call / init / log_probability, VAENAR.inference, the ELBO forward and (round 6) the training
        # step run that way (engine option "prior_inverse").  LJHPS / DataBakerHPS use inverse=False (hparams.py:344,462).
        self.inverse = bool(inverse)
        if self.inverse:
            self.engine.set_option("prior_inverse", 1)
        self.channels = channels
        self.noise_seed = 0          # seed / running offset of the device generator (vnr_random_normal): tf.random.normal's role
        self.noise_offset = 0

    def seed(self, seed):
        """Re-seed the device noise stream (tf.random.set_seed's role for prior.py:35 / posterior.py:35)."""
        self.noise_seed, self.noise_offset = int(seed), 0

    def draw(self, shape, stddev=1.0):
        """tf.random.normal(shape, stddev=stddev) ON THE DEVICE: a Philox-4x32 stream keyed by ``noise_seed``; successive
        draws take disjoint counter ranges.  Nothing crosses PCIe."""
        n = int(np.prod(shape))
        a = self.engine.random_normal(shape, self.noise_seed, self.noise_offset, stddev)
        self.noise_offset += (n + 3) // 4
        return a

    def sample(self, targets_lengths, condition_inputs, condition_lengths=None, training=None,
               temperature=1.0, eps=None, return_logprobs=True):
        """prior.py:154-169.  ``eps`` (already scaled by the temperature, [B, max(len), C]) replaces
        tf.random.normal (prior.py:35) when given; otherwise it is drawn on the device (``draw``; temperature 0 -> exact zeros,
        the inference.py:95 default -- no noise buffer at all)."""
        # (training: the flows hold no Dropout / BatchNormalization -- prior.py:161-169 forwards the flag to layers that ignore it)
        e = self.engine
        B, Tz, Tt, zl, cond, tl, eps_d = self._flow_args(targets_lengths, condition_inputs, condition_lengths, temperature, eps)
        z = e.empty((B, Tz, self.channels))
        logp = e.empty((B,)) if return_logprobs else None
        e.call("vnr_prior_sample", zl.ptr, cond.ptr, tl.ptr, B, Tz, Tt, self._ptr(eps_d), z.ptr,
                                     self._ptr(logp))
        return z, logp

    def _flow_args(self, targets_lengths, condition_inputs, condition_lengths, temperature, eps):
        e = self.engine
        lens_h = targets_lengths.numpy() if hasattr(targets_lengths, "numpy") and not isinstance(
            targets_lengths, np.ndarray) else np.asarray(targets_lengths)
        lens_h = lens_h.astype(np.int32)
        B, Tz = len(lens_h), int(lens_h.max())                  # prior.py:33-34
        cond = self._f32(condition_inputs)
        Tt = cond.shape[1]
        zl = e.asarray(targets_lengths if not isinstance(targets_lengths, np.ndarray) else lens_h, np.int32)
        tl = self._i32(condition_lengths, B, Tt)
        if eps is None and float(temperature) != 0.0:
            eps = self.draw((B, Tz, self.channels), float(temperature))
        eps_d = None if eps is None else self._f32(eps)
        if eps_d is not None:
            assert eps_d.shape == (B, Tz, self.channels), (eps_d.shape, (B, Tz, self.channels))
        return B, Tz, Tt, zl, cond, tl, eps_d

    def __call__(self, inputs, targets_lengths, condition_lengths, training=None, temperature=1.0, eps=None):
        """TransformerPrior.call (prior.py:101-117): ``inputs`` are the condition inputs; the same arithmetic as ``sample`` (every flow
        through BaseFlow.call / fwd_pass, flow.py:36-47,76-91) with the arguments in this order."""
        return self.sample(targets_lengths, inputs, condition_lengths, training=training, temperature=temperature, eps=eps)

    call = __call__

    def init(self, conditions, targets_lengths, condition_lengths, training=None, eps=None):
        """TransformerPrior.init (prior.py:171-186): data-dependent initialisation -- every ActNormFlow takes log_scale / bias
        from the statistics of its input (flow.py:189-196), written to the engine's weight store (``model.trainable_variables``
        / ``get_weights`` read them back).  Returns (z, logprobs).  ``eps`` [B, max(len), C] replaces tf.random.normal of
        _initial_sample (prior.py:35, temperature 1)."""
        e = self.engine
        B, Tz, Tt, zl, cond, tl, eps_d = self._flow_args(targets_lengths, conditions, condition_lengths, 1.0, eps)
        z = e.empty((B, Tz, self.channels))
        logp = e.empty((B,))
        e.call("vnr_prior_init", zl.ptr, cond.ptr, tl.ptr, B, Tz, Tt, self._ptr(eps_d), z.ptr, logp.ptr, record=False)
        return z, logp

    def log_probability(self, z, condition_inputs, z_lengths=None, condition_lengths=None, training=None):
        """prior.py:119-152: log p(z | text) by running the flow backwards; [B] (device)."""
        e = self.engine                 # (training: no training-dependent layer inside the flows, see sample)
        zd = self._f32(z)
        cond = self._f32(condition_inputs)
        B, Tz, _ = zd.shape
        Tt = cond.shape[1]
        zl = self._i32(z_lengths, B, Tz)
        tl = self._i32(condition_lengths, B, Tt)
        out = e.empty((B,))
        e.call("vnr_prior_log_probability", zd.ptr, cond.ptr, zl.ptr, tl.ptr, B, Tz, Tt, out.ptr)
        return out
