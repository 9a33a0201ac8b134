"""`TestUtils` -- reference audio/utils.py:10-116: write_mels, synthesize_and_save_wavs (Griffin-Lim on the GPU) and, since round 6, the
figures the harnesses draw from `test_step`'s by-products: predicted mel spectrograms and the decoder's attention alignments (matplotlib,
host only; imported on first use -- the text -> mel path itself never needs it)."""
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

    # ---- figures (audio/utils.py:42-116; called from train.py:319-323 and inference.py:160-164) ------------------------------------------
    @staticmethod
    def _pyplot():
        try:
            import matplotlib
            matplotlib.use('agg')
            import matplotlib.pyplot as plt
            return plt
        except ImportError as e:                      # the figures are an extra of the harnesses, not of the path
            raise RuntimeError("drawing spectrograms / alignments needs matplotlib (%s)" % e)

    def _name(self, prefix, idx, step, ext='.pdf'):
        idx = idx.decode('utf-8') if type(idx) is bytes else idx
        return os.path.join(self.save_dir, '{}-{}-{}{}'.format(prefix, idx, step, ext))

    def _ids_to_symbols(self, id_list):              # audio/utils.py:60-68: hps.Texts.characters, position = id
        table = list(self.hps.Texts.characters)
        return [table[int(i)] if 0 <= int(i) < len(table) else '?' for i in id_list]

    def draw_melspectrograms(self, step, mel_batch, mel_lengths, ids, prefix=''):          # audio/utils.py:42-58
        """One `<prefix>-<id>-<step>.pdf` per utterance: the first `mel_lengths[i]` frames, frequency upwards.  (The reference fans
        the utterances out to a multiprocessing pool; a test batch is a handful of figures: drawn in turn.)"""
        plt = self._pyplot()
        for i in range(len(ids)):
            n = int(mel_lengths[i])
            fig = plt.figure()
            plt.imshow(np.asarray(mel_batch[i])[:n, :].T, aspect='auto', origin='lower')
            plt.tight_layout()
            fig.savefig(self._name(prefix, ids[i], step))
            plt.close(fig)

    def multi_draw_attention_alignments(self, batch_ali, batch_texts, text_lengths, mel_lengths, step, ids, prefix='posterior'):
        """audio/utils.py:70-116.  `batch_ali` [B, T_dec, T_text] -> one panel per utterance; [B, heads, T_dec, T_text] -> one panel per
        head on a 2 x heads/2 grid.  The text axis carries the input symbols as tick labels, decoder time runs upwards.  (The
        multi-head branch of the reference shows every decoder row, padded ones included; so does this.)"""
        plt = self._pyplot()
        batch_ali = np.asarray(batch_ali)
        if batch_ali.ndim not in (3, 4):
            raise ValueError("alignments must be [B, T_dec, T_text] or [B, heads, T_dec, T_text], got %r" % (batch_ali.shape,))
        for i in range(len(ids)):
            tlen, mlen = int(text_lengths[i]), int(mel_lengths[i])
            symbols = self._ids_to_symbols(np.asarray(batch_texts[i])[:tlen])
            if batch_ali.ndim == 3:
                fig, ax = plt.subplots()
                ax.set_xticks(np.arange(tlen)); ax.set_xticklabels(symbols, fontsize=3)
                ax.imshow(batch_ali[i, :mlen, :tlen], aspect='auto', origin='lower')
            else:
                heads = batch_ali.shape[1]
                fig = plt.figure()
                for j in range(heads):
                    ax = fig.add_subplot(2, max(1, (heads + 1) // 2), j + 1)
                    ax.set_xticks(np.arange(tlen)); ax.set_xticklabels(symbols, fontsize=2)
                    ax.imshow(batch_ali[i, j, :, :tlen], aspect='auto', origin='lower')
            plt.tight_layout()
            fig.savefig(self._name(prefix, ids[i], step))
            plt.close(fig)
        print('Attentions for {} are plotted'.format(prefix))
