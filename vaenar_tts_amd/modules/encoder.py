"""TransformerEncoder mirror (/root/reference/modules/encoder.py:58-93)."""
from ._base import EngineModule, check


class TransformerEncoder(EngineModule):
    var_prefix = "text_encoder"

    def __init__(self, vocab_size, embd_dim, pre_nconv, pre_hidden, pre_conv_kernel,
                 prenet_drop_rate, pre_activation, bn_before_act, pos_drop_rate, nblk,
                 attention_dim, attention_heads, attention_temperature, ffn_hidden,
                 name='TextEncoder', engine=None, **kwargs):
        super().__init__(name, engine)
        self.vocab_size, self.embd_dim, self.pre_hidden = vocab_size, embd_dim, pre_hidden

    def __call__(self, inputs, input_lengths=None, pos_step=1.0, training=None, dropout_seed=None):
        """encoder.py:79-93: ids [B,T] -> text encoding [B,T,pre_hidden] (device)."""
        e = self.engine
        ids = e.asarray(inputs, 'int32')
        B, T = ids.shape
        lens = self._i32(input_lengths, B, T)
        out = e.empty((B, T, self.pre_hidden))
        with self._training(training, dropout_seed):
            e.call("vnr_text_encoder_fwd", ids.ptr, lens.ptr, B, T, float(pos_step), out.ptr)
        return out

    call = __call__
