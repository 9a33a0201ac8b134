"""Synthetic utterance batches for parity tests and bench.py.

Token ids follow the LJSpeech symbol table layout of the reference
(configs/hparams.py:263-267: 0 = pad '_', 1 = bos '^', 2 = eos '~', 3.. = characters):
row b = [1, U{3..V-1} x (len-2), 2, 0-pad ...].
"""
import numpy as np


def make_batch(B, T_text, T_mel, vocab_size=43, latent_dim=128, reduction_factor=2,
               ragged=False, seed=1234, temperature=0.0, text_step=4, mel_step=24):
    """Returns dict(ids [B,T_text] i32, text_lengths [B] i32, mel_lengths [B] i32,
    eps [B, ceil(max_mel/rf), latent] f32 = temperature * N(0,1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if ragged:
        text_lengths = np.maximum(3, T_text - text_step * np.arange(B)).astype(np.int32)
        mel_lengths = np.maximum(reduction_factor, T_mel - mel_step * np.arange(B)).astype(np.int32)
    else:
        text_lengths = np.full(B, T_text, np.int32)
        mel_lengths = np.full(B, T_mel, np.int32)
    ids = np.zeros((B, T_text), np.int32)
    for b in range(B):
        n = int(text_lengths[b])
        ids[b, 0] = 1
        ids[b, 1:n - 1] = rng.integers(3, vocab_size, n - 2)
        ids[b, n - 1] = 2
    Tz = int((int(mel_lengths.max()) + reduction_factor - 1) // reduction_factor)
    noise = rng.standard_normal((B, Tz, latent_dim)).astype(np.float32)
    eps = (np.float32(temperature) * noise).astype(np.float32)
    return dict(ids=ids, text_lengths=text_lengths, mel_lengths=mel_lengths, eps=eps)
