"""`Audio` -- the inverse (mel -> waveform) half of reference audio/audio.py:11-246 on the HIP engine.

Same class name, constructor argument (``hps.Audio``) and method names as the reference for what follows the text->mel path:
``inv_mel_spectrogram`` (audio.py:81-84), ``_griffin_lim`` (:95-102), ``_stft_parameters`` (:153-160), ``_build_mel_basis``
(:176-183), ``inv_preemphasize`` (:232-246), ``save_wav`` (:18-21).  The analysis direction (load_wav, melspectrogram, mfcc, ...)
is data preparation and stays out of scope.

Mel -> linear (pinv of the mel filterbank) and the Griffin-Lim iterations run in libvaenar_hip.so (csrc/vocoder.hip) through
``vnr_voc_mel_to_linear`` / ``vnr_voc_griffin_lim``; there is no CPU fallback.  The filterbank itself is librosa 0.8.0's
``filters.mel(..., htk=False, norm='slaney')`` (environment.yml:65), rebuilt here because librosa is not a dependency of this
package; its pseudo-inverse is taken once on the host.  De-emphasis is the reference's own one-pole ``scipy.signal.lfilter``.
"""
import ctypes as C

import numpy as np

from .. import _lib


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) of librosa 0.8.0 (Slaney scale, slaney area norm) -> [n_mels, 1+n_fft/2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    return weights * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


class Audio:
    def __init__(self, audio_hparams, engine=None, device=0):
        self.hps = audio_hparams
        if engine is None:                                   # a handle of its own (the vocoder entry points use no model weights)
            from ..configs import tiny_hps
            engine = _lib.Engine(tiny_hps(), device)
        self.engine = engine
        self._inv_basis_t = None

    # ---- parameters ---------------------------------------------------------------------------------------------------
    def _stft_parameters(self):                              # audio.py:153-160
        return (self.hps.num_freq - 1) * 2, self.hps.frame_shift_sample, self.hps.frame_length_sample

    def _build_mel_basis(self):                              # audio.py:176-183
        n_fft = (self.hps.num_freq - 1) * 2
        return mel_filterbank(self.hps.sample_rate, n_fft, self.hps.num_mels, self.hps.min_mel_freq, self.hps.max_mel_freq)

    def _inv_mel_basis_t(self):
        if self._inv_basis_t is None:                        # audio.py:167: np.linalg.pinv(mel basis), kept transposed on the device
            inv = np.linalg.pinv(self._build_mel_basis())    # [num_freq, num_mels]
            self._inv_basis_t = self.engine.to_device(np.ascontiguousarray(inv.T), np.float32)
        return self._inv_basis_t

    # ---- mel -> waveform ----------------------------------------------------------------------------------------------
    def linear_from_mel_batch(self, mels):
        """[B, T, num_mels] (host or device) -> device [B, T, num_freq]: S ** power of audio.py:81-84."""
        h, e = self.hps, self.engine
        d_mel = e.asarray(mels, np.float32) if isinstance(mels, _lib.DeviceArray) else e.to_device(mels, np.float32)
        B, T, M = d_mel.shape
        assert M == h.num_mels
        S = e.empty((B, T, h.num_freq))
        e.call("vnr_voc_mel_to_linear", d_mel.ptr, self._inv_mel_basis_t().ptr, B, T, M, h.num_freq, float(h.min_level_db),
               float(h.ref_level_db), float(h.max_abs_value), int(bool(h.symmetric_specs)), float(h.power), S.ptr)
        return S

    def _griffin_lim_batch(self, S, frames=None, init_angles=None, seed=0, n_iters=None):
        """S device [B, T, num_freq] -> device [B, hop*(T-1)].  frames [B] (<= T) for ragged batches; init_angles [B,T,num_freq]
        radians (audio.py:96 draws 2 pi rand(), unseeded) or None = drawn on the device from `seed`."""
        e = self.engine
        n_fft, hop, win = self._stft_parameters()
        B, T, F = S.shape
        assert F == n_fft // 2 + 1
        n_iters = self.hps.griffin_lim_iters if n_iters is None else int(n_iters)
        d_fr = None if frames is None else e.to_device(np.asarray(frames, np.int32), np.int32)
        d_ang = None if init_angles is None else e.to_device(init_angles, np.float32)
        wav = e.empty((B, hop * (T - 1)))
        e.call("vnr_voc_griffin_lim", S.ptr, None if d_ang is None else d_ang.ptr, C.c_uint64(int(seed)),
               None if d_fr is None else d_fr.ptr, B, T, n_fft, hop, win, n_iters, wav.ptr)
        return wav

    def inv_mel_spectrogram_batch(self, mels, lengths=None, init_angles=None, seed=0, n_iters=None):
        """mels [B, T, num_mels] -> list of B float32 waveforms (utterance b has hop * (lengths[b] - 1) samples)."""
        S = self.linear_from_mel_batch(mels)
        wav = self._griffin_lim_batch(S, lengths, init_angles, seed, n_iters).numpy()
        hop = self.hps.frame_shift_sample
        T = S.shape[1]
        lens = [T] * S.shape[0] if lengths is None else [int(x) for x in lengths]
        return [wav[b, :hop * (lens[b] - 1)] for b in range(S.shape[0])]

    def inv_mel_spectrogram(self, mel_spectrogram, init_angles=None, seed=0):
        """audio.py:81-84: mel_spectrogram [num_mels, T] (the reference passes mel.T, audio/utils.py:26) -> waveform [hop*(T-1)].
        init_angles [num_freq, T] radians reproduces a given np.random.rand draw (angles = 2 pi rand)."""
        mel = np.asarray(mel_spectrogram, np.float32)
        assert mel.ndim == 2 and mel.shape[0] == self.hps.num_mels
        ang = None if init_angles is None else np.ascontiguousarray(np.asarray(init_angles, np.float32).T)[None]
        return self.inv_mel_spectrogram_batch(np.ascontiguousarray(mel.T)[None], None, ang, seed)[0]

    def _griffin_lim(self, S, init_angles=None, seed=0):     # audio.py:95-102: S [num_freq, T] magnitudes (already ** power)
        S = np.ascontiguousarray(np.asarray(S, np.float32).T)[None]
        ang = None if init_angles is None else np.ascontiguousarray(np.asarray(init_angles, np.float32).T)[None]
        return self._griffin_lim_batch(self.engine.to_device(S, np.float32), None, ang, seed).numpy()[0]

    # ---- after Griffin-Lim --------------------------------------------------------------------------------------------
    def inv_preemphasize(self, x):                           # audio.py:232-246
        if self.hps.preemphasize is None:
            return x
        from scipy import signal
        return signal.lfilter([1], [1, -self.hps.preemphasize], x)

    def save_wav(self, wav, path):                           # audio.py:18-21
        from scipy.io import wavfile
        wav = np.asarray(wav) * (32767 / max(0.01, np.max(np.abs(wav))))
        wavfile.write(path, self.hps.sample_rate, wav.astype(np.int16))
