"""TransformerDecoder mirror (/root/reference/modules/decoder.py:156-199)."""
from ._base import EngineModule, check


class TransformerDecoder(EngineModule):
    var_prefix = "decoder"

    def __init__(self, nblk, attention_dim, attention_heads, temperature, ffn_hidden, post_n_conv,
                 post_conv_filters, post_conv_kernel, post_drop_rate, out_dim, max_reduction_factor,
                 name='TransformerDecoder', engine=None):
        super().__init__(name, engine)
        self.nblk, self.heads = nblk, attention_heads
        self.out_dim, self.max_reduction_factor = out_dim, max_reduction_factor
        # block names as in decoder.py:172 -> keys of the returned alignment dict (:192)
        self.block_names = ['decoder-attention-{}'.format(i) for i in range(nblk)]

    def __call__(self, inputs, text_embd, z_lengths=None, text_lengths=None, reduction_factor=2,
                 training=None, return_alignments=True, dropout_seed=None):
        """decoder.py:181-199 -> (initial_outs, outputs, {name: alignments [B,H,Tz,Tt]})."""
        e = self.engine
        z = self._f32(inputs)
        mem = self._f32(text_embd)
        B, Tz, _ = z.shape
        Tt = mem.shape[1]
        zl = self._i32(z_lengths, B, Tz)
        tl = self._i32(text_lengths, B, Tt)
        rf = int(reduction_factor)
        initial = e.empty((B, Tz * rf, self.out_dim))
        outputs = e.empty((B, Tz * rf, self.out_dim))
        ali = e.empty((self.nblk, B, self.heads, Tz, Tt)) if return_alignments else None
        with self._training(training, dropout_seed):
            e.call("vnr_decoder_fwd", z.ptr, mem.ptr, zl.ptr, tl.ptr, B, Tz, Tt, rf, initial.ptr,
                                        outputs.ptr, self._ptr(ali))
        alignments = {}
        if ali is not None:
            n = B * self.heads * Tz * Tt
            for i, name in enumerate(self.block_names):
                alignments[name] = ali.view(i * n, (B, self.heads, Tz, Tt))
        return initial, outputs, alignments

    call = __call__
