"""Frozen hyper-parameters of the VAENAR-TTS text->mel path.

Mirrors the attribute tree the reference model constructor reads
(/root/reference/models/models.py:10-65 reads hps.Encoder.Transformer.*,
hps.Decoder.Transformer.*, hps.Posterior.Transformer.*, hps.Prior.Transformer.*,
hps.LengthPredictor.Dense.*, hps.Common.*, hps.Train.num_samples,
hps.Audio.num_mels).  Values are the ones of LJHPS / DataBakerHPS
(/root/reference/configs/hparams.py:233-348 and :351-474); activations are
named by strings ('relu', 'tanh', 'identity') because there is no TensorFlow
here.  Only the values the text->mel path needs are kept.
"""
import copy


class _NS:
    """Plain attribute namespace (class-attribute style like the reference)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __repr__(self):
        return "NS(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))

    def to_dict(self):
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to_dict() if isinstance(v, _NS) else v
        return out


def _audio(sample_rate, frame_length_sample, frame_shift_sample, min_level_db):
    # hparams.py:266-282 (LJSpeech) / :384-400 (DataBaker): read by audio/audio.py (the vocoder step after the path)
    return _NS(num_mels=80, num_freq=1025, min_mel_freq=0., max_mel_freq=8000., sample_rate=sample_rate,
               frame_length_sample=frame_length_sample, frame_shift_sample=frame_shift_sample, n_mfcc=13, preemphasize=0.97,
               min_level_db=min_level_db, ref_level_db=20.0, max_abs_value=1, symmetric_specs=False, griffin_lim_iters=60,
               power=1.5, center=True)


def _base(vocab_size, mel_text_len_ratio, audio, characters):
    return _NS(
        Texts=_NS(pad='_', bos='^', eos='~', characters=characters),              # hparams.py:260-264 / :378-382
        Train=_NS(
            random_seed=123456, epochs=2000, train_batch_size=32, test_batch_size=8,
            num_samples=1, length_weight=1.0, kl_weight=1.0, kl_weight_init=1e-5,
            kl_weight_increase_epoch=1, kl_weight_end=1e-5, learning_rate=1.25e-4,
            reduction_factors=[5, 4, 3, 2], reduce_interval=[0, 200, 400, 600],
            shuffle_buffer=128, shuffle=True, test_interval=50),
        Dataset=_NS(buffer_size=65536, num_parallel_reads=64, pad_factor=0),             # hparams.py:253-258
        Audio=audio,
        Common=_NS(latent_dim=128, output_dim=80, final_reduction_factor=2,
                   max_reduction_factor=5, mel_text_len_ratio=mel_text_len_ratio),
        Encoder=_NS(Transformer=_NS(
            vocab_size=vocab_size, embd_dim=512, n_conv=3, pre_hidden=512, conv_kernel=5,
            pre_activation='relu', pre_drop_rate=0.1, pos_drop_rate=0.1,
            bn_before_act=False, n_blk=4, attention_dim=256, attention_heads=4,
            attention_temperature=1.0, ffn_hidden=1024)),
        Decoder=_NS(Transformer=_NS(
            nblk=2, attention_dim=256, attention_heads=4, ffn_hidden=1024,
            attention_temperature=1.0, post_n_conv=5, post_conv_filters=256,
            post_conv_kernel=5, post_drop_rate=0.2)),
        Posterior=_NS(Transformer=_NS(
            pre_hidden=256, pos_drop_rate=0.2, pre_drop_rate=0.5, pre_activation='relu',
            nblk=2, attention_dim=256, attention_heads=4, temperature=1.0,
            ffn_hidden=1024)),
        Prior=_NS(Transformer=_NS(
            n_blk=6, n_transformer_blk=2, attention_dim=256, attention_heads=4,
            temperature=1.0, ffn_hidden=1024, inverse=False)),
        LengthPredictor=_NS(Dense=_NS(activation='identity')),
    )


# /root/reference/configs/hparams.py:233-348
LJHPS = _base(vocab_size=43, mel_text_len_ratio=5.59, audio=_audio(22050, 1024, 256, -100.0),
              characters='_^~abcdefghijklmnopqrstuvwxyz!\'\"(),-.:;? []')
# /root/reference/configs/hparams.py:351-474 (differs in vocab 39 :411, ratio 4.21 :407)
DataBakerHPS = _base(vocab_size=39, mel_text_len_ratio=4.21, audio=_audio(16000, 800, 200, -115.0),
                    characters='_^~abcdefghijklmnopqrstuvwxyz12345,./- ')


def tiny_hps():
    """Reduced-width configuration for fast CPU/GPU parity tests.

    Same topology as LJHPS; widths shrunk.  The per-head width stays 64 because
    the HIP attention kernel is specialised for d_head = 64 (the only value the
    reference configurations use: 256 / 4, hparams.py:300-302).
    """
    h = copy.deepcopy(LJHPS)
    h.Common.latent_dim = 32
    h.Common.output_dim = 16
    h.Audio.num_mels = 16
    e = h.Encoder.Transformer
    e.vocab_size, e.embd_dim, e.pre_hidden, e.n_blk = 43, 96, 96, 2
    e.attention_dim, e.attention_heads, e.ffn_hidden = 128, 2, 160
    d = h.Decoder.Transformer
    d.attention_dim, d.attention_heads, d.ffn_hidden = 128, 2, 160
    d.post_conv_filters = 48
    p = h.Posterior.Transformer
    p.pre_hidden, p.attention_dim, p.attention_heads, p.ffn_hidden = 128, 128, 2, 160
    r = h.Prior.Transformer
    r.n_blk, r.attention_dim, r.attention_heads, r.ffn_hidden = 3, 128, 2, 160
    return h
