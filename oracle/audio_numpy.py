"""TEST INFRASTRUCTURE -- CPU restatement (NumPy float64 / complex128) of the vocoder step that follows the text->mel path:
mel -> linear magnitude -> Griffin-Lim -> de-emphasis.

Reference: /root/reference/audio/audio.py
    inv_mel_spectrogram :81-84   _griffin_lim :95-102   _stft :104-127   _istft :129-151   _stft_parameters :153-160
    _mel_to_linear :166-174      _build_mel_basis :176-183   _db_to_amp :189-191   _denormalize :206-216
    inv_preemphasize :232-246    save_wav :18-21
and /root/reference/audio/utils.py:24-30 (synthesize_and_save_wavs: inv_mel_spectrogram(mel.T) -> inv_preemphasize -> save_wav).

The arithmetic of stft / istft / filters.mel lives in third-party librosa 0.8.0 (environment.yml:65), which is NOT installed here
and cannot be installed offline: this file restates librosa 0.8.0's published algorithm --
    stft:  periodic Hann window of win_length, zero-padded symmetrically to n_fft (util.pad_center); center=True pads the signal by
           n_fft//2 on both sides with np.pad(mode='reflect'); frame t = padded[t*hop : t*hop + n_fft]; rfft(window * frame);
    istft: irfft of every column, times the same padded window, overlap-added at t*hop into a buffer of n_fft + hop*(n_frames-1)
           samples; divided by the window sum-of-squares (filters.window_sumsquare, norm=None) wherever that exceeds
           util.tiny (smallest normal number of the dtype); center=True crops n_fft//2 samples from both ends;
    filters.mel: Slaney mel scale (htk=False: linear below 1 kHz at 200/3 Hz per mel, log above with step log(6.4)/27), triangular
           weights on the rfft bin centres, 'slaney' area normalisation 2 / (f[i+2] - f[i]).
PARITY UNPINNED against the real librosa; pinned by known-answer tests instead (tests/test_audio_oracle.py:
scipy.signal.stft / istft on the same padded window, perfect reconstruction, Parseval, the published mel-scale break point).

Griffin-Lim starts from np.random.rand phases in the reference (audio.py:96, unseeded): every function here takes the initial
phases as an argument so that the GPU path can be compared on identical inputs.
"""
import numpy as np


def hann_periodic(win_length):
    """scipy.signal.get_window('hann', win_length, fftbins=True)."""
    n = np.arange(win_length, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)


def padded_window(n_fft, win_length):
    """librosa.util.pad_center(get_window('hann', win_length), n_fft)."""
    w = np.zeros(n_fft, np.float64)
    lpad = (n_fft - win_length) // 2
    w[lpad:lpad + win_length] = hann_periodic(win_length)
    return w


def stft(y, n_fft, hop_length, win_length, center=True):
    """librosa.stft(y, n_fft, hop_length, win_length, window='hann', center, pad_mode='reflect') -> [1 + n_fft/2, n_frames]."""
    y = np.asarray(y, np.float64)
    w = padded_window(n_fft, win_length)
    if center:
        y = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.fft.rfft(w[:, None] * y[idx], axis=0)


def window_sumsquare(n_frames, n_fft, hop_length, win_length):
    """librosa.filters.window_sumsquare('hann', n_frames, hop_length, win_length, n_fft, norm=None)."""
    x = np.zeros(n_fft + hop_length * (n_frames - 1), np.float64)
    wsq = padded_window(n_fft, win_length) ** 2
    for t in range(n_frames):
        x[t * hop_length:t * hop_length + n_fft] += wsq
    return x


def istft(D, hop_length, win_length, center=True, tiny=np.finfo(np.float32).tiny):
    """librosa.istft(D, hop_length, win_length, window='hann', center) -> [hop*(n_frames-1)] (center=True)."""
    D = np.asarray(D, np.complex128)
    n_fft = 2 * (D.shape[0] - 1)
    n_frames = D.shape[1]
    w = padded_window(n_fft, win_length)
    y = np.zeros(n_fft + hop_length * (n_frames - 1), np.float64)
    frames = w[:, None] * np.fft.irfft(D, n=n_fft, axis=0)
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + n_fft] += frames[:, t]
    wss = window_sumsquare(n_frames, n_fft, hop_length, win_length)
    nz = wss > tiny
    y[nz] /= wss[nz]
    if center:
        y = y[n_fft // 2:-(n_fft // 2)]
    return y


def hz_to_mel(f):
    """librosa.hz_to_mel(htk=False) (Slaney)."""
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') -> [n_mels, 1 + n_fft/2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


class AudioOracle:
    """Restatement of reference audio/audio.py `Audio` (inverse direction only)."""

    def __init__(self, hps):
        self.hps = hps

    def _stft_parameters(self):                                            # audio.py:153-160
        return (self.hps.num_freq - 1) * 2, self.hps.frame_shift_sample, self.hps.frame_length_sample

    def _build_mel_basis(self):                                            # audio.py:176-183
        n_fft = (self.hps.num_freq - 1) * 2
        return mel_basis(self.hps.sample_rate, n_fft, self.hps.num_mels, self.hps.min_mel_freq, self.hps.max_mel_freq)

    def _denormalize(self, S):                                             # audio.py:206-216
        h = self.hps
        if h.symmetric_specs:
            return (np.clip(S, -h.max_abs_value, h.max_abs_value) + h.max_abs_value) * (-h.min_level_db) / (2 * h.max_abs_value) + h.min_level_db
        return np.clip(S, 0, h.max_abs_value) * (-h.min_level_db) / h.max_abs_value + h.min_level_db

    @staticmethod
    def _db_to_amp(x):                                                     # audio.py:189-191
        return np.power(10.0, x * 0.05)

    def _mel_to_linear(self, mel):                                         # audio.py:166-174  (mel [n_mels, T])
        inv = np.linalg.pinv(self._build_mel_basis())
        return np.maximum(1e-10, inv @ mel)

    def linear_from_mel(self, mel):
        """S ** power of inv_mel_spectrogram (audio.py:81-84): the magnitudes Griffin-Lim starts from, [num_freq, T]."""
        mel = np.asarray(mel, np.float64)
        S = self._mel_to_linear(self._db_to_amp(self._denormalize(mel) + self.hps.ref_level_db))
        return S ** self.hps.power

    def griffin_lim(self, S, init_angles, n_iters=None):                   # audio.py:95-102; init_angles = 2*pi*rand(*S.shape)
        n_fft, hop, win = self._stft_parameters()
        n_iters = self.hps.griffin_lim_iters if n_iters is None else n_iters
        S = np.abs(S).astype(np.complex128)
        y = istft(S * np.exp(1j * init_angles), hop, win, self.hps.center)
        for _ in range(n_iters):
            ang = np.exp(1j * np.angle(stft(y, n_fft, hop, win, self.hps.center)))
            y = istft(S * ang, hop, win, self.hps.center)
        return y

    def inv_mel_spectrogram(self, mel, init_angles, n_iters=None):         # audio.py:81-84 (mel [n_mels, T])
        return self.griffin_lim(self.linear_from_mel(mel), init_angles, n_iters)

    def inv_preemphasize(self, x):                                         # audio.py:232-246: lfilter([1], [1, -a], x)
        a = self.hps.preemphasize
        if a is None:
            return x
        y = np.empty_like(np.asarray(x, np.float64))
        acc = 0.0
        for i, v in enumerate(x):
            acc = v + a * acc
            y[i] = acc
        return y

    @staticmethod
    def to_int16(wav):                                                     # audio.py:18-21 (save_wav's scaling and cast)
        wav = np.asarray(wav, np.float64) * (32767 / max(0.01, np.max(np.abs(wav))))
        return wav.astype(np.int16)
