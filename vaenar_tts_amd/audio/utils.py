"""`TestUtils` -- reference audio/utils.py:10-40 (write_mels, synthesize_and_save_wavs) without matplotlib plotting."""
import os

import numpy as np

from .audio import Audio


class TestUtils:
    __test__ = False          # (not a pytest class)

    def __init__(self, hps, save_dir, engine=None, device=0):
        self.prcocessor = Audio(hps.Audio, engine=engine, device=device)      # (attribute name as in audio/utils.py:12)
        self.hps = hps
        self.save_dir = save_dir

    def write_mels(self, step, mel_batch, mel_lengths, ids, prefix=''):       # audio/utils.py:16-22
        for i in range(mel_batch.shape[0]):
            mel = mel_batch[i][:mel_lengths[i], :]
            idx = ids[i].decode('utf-8') if type(ids[i]) is bytes else ids[i]
            np.save(os.path.join(self.save_dir, '{}-{}-{}.npy'.format(prefix, idx, step)), mel)

    def synthesize_and_save_wavs(self, step, mel_batch, mel_lengths, ids, prefix='', seed=0):   # audio/utils.py:24-40
        """The reference starts one Python thread per utterance (librosa on the CPU); here the whole batch goes through one
        Griffin-Lim call on the GPU, then de-emphasis and the int16 wav file per utterance as in the reference."""
        lens = [int(x) for x in mel_lengths]
        wavs = self.prcocessor.inv_mel_spectrogram_batch(np.asarray(mel_batch, np.float32), lens, seed=seed)
        for i, wav_arr in enumerate(wavs):
            idx = ids[i].decode('utf-8') if type(ids[i]) is bytes else ids[i]
            wav_arr = self.prcocessor.inv_preemphasize(wav_arr)
            self.prcocessor.save_wav(wav_arr, os.path.join(self.save_dir, '{}-{}-{}.wav'.format(prefix, idx, step)))
        print('All wavs for {} are synthesized!'.format(prefix))
