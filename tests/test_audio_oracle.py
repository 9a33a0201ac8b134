"""CPU tests of the vocoder oracle (oracle/audio_numpy.py, a restatement of librosa 0.8.0's stft / istft / filters.mel and of
reference audio/audio.py:81-246).  librosa is not installed here (parity unpinned); the restatement is pinned by
  * the values printed in librosa's own documentation for hz_to_mel / mel_to_hz / filters.mel,
  * scipy.signal.stft / istft (an independent implementation) driven with the same padded window,
  * analytic properties: perfect reconstruction under the COLA window, Parseval, a pure tone's bin, Griffin-Lim consistency,
  * scipy.signal.lfilter -- the reference's own de-emphasis call (audio.py:236)."""
import numpy as np
import pytest
import scipy.signal as ss

from oracle import audio_numpy as A
from vaenar_tts_amd.audio.audio import mel_filterbank
from vaenar_tts_amd.configs import LJHPS, DataBakerHPS


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def test_mel_scale_known_answers():
    # librosa docs: hz_to_mel(60) = 0.9; hz_to_mel([110, 220, 440]) = [1.65, 3.3, 6.6]; mel_to_hz(3) = 200.;
    # mel_to_hz([1,2,3,4,5]) = [66.667, 133.333, 200., 266.667, 333.333]
    assert abs(float(A.hz_to_mel(60)) - 0.9) < 1e-12
    np.testing.assert_allclose(A.hz_to_mel([110, 220, 440]), [1.65, 3.3, 6.6], rtol=1e-12)
    np.testing.assert_allclose(A.mel_to_hz([1, 2, 3, 4, 5]), [66.66666667, 133.33333333, 200.0, 266.66666667, 333.33333333], rtol=1e-9)
    # the Slaney break point: linear below 1 kHz (15 mel), log above with 27 steps per factor 6.4
    assert abs(float(A.hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(A.hz_to_mel(6400.0)) - 42.0) < 1e-9
    np.testing.assert_allclose(A.mel_to_hz(A.hz_to_mel([30.0, 999.0, 1000.0, 4321.0, 7600.0])), [30.0, 999.0, 1000.0, 4321.0, 7600.0], rtol=1e-12)


def test_mel_filterbank_known_answers():
    # librosa docs, filters.mel(sr=22050, n_fft=2048): first row [0., 0.016, ..., 0., 0.], second row [0., 0.009, ...] (3 decimals, column 3)
    mb = A.mel_basis(22050, 2048, 128, 0.0, 11025.0)
    assert mb.shape == (128, 1025)
    assert mb[0, 0] == 0.0 and round(float(mb[0, 1]), 3) == 0.016 and mb[0, -1] == 0.0
    assert round(float(mb[1, 3]), 3) == 0.010 and mb[1, 0] == 0.0 and mb[1, 1] == 0.0
    # every filter is a non-negative triangle; slaney norm: equal area 1 in Hz -> sum * bin width ~ 1 for filters wide enough
    ref = A.mel_basis(22050, 2048, 80, 0.0, 8000.0)
    assert (ref >= 0).all() and ((ref > 0).sum(1) >= 2).all()
    np.testing.assert_allclose(ref[40:].sum(1) * (22050 / 2048), 1.0, rtol=0.03)
    # the product's own builder (vaenar_tts_amd/audio/audio.py) is the same function
    for hps in (LJHPS, DataBakerHPS):
        a = hps.Audio
        np.testing.assert_allclose(mel_filterbank(a.sample_rate, 2048, a.num_mels, a.min_mel_freq, a.max_mel_freq),
                                   A.mel_basis(a.sample_rate, 2048, a.num_mels, a.min_mel_freq, a.max_mel_freq), rtol=0, atol=1e-15)


@pytest.mark.parametrize("n_fft,hop,win", [(2048, 256, 1024), (2048, 200, 800), (512, 128, 512)])
def test_stft_against_scipy_and_reconstruction(n_fft, hop, win):
    y = rng(n_fft + hop).standard_normal(hop * 37)
    D = A.stft(y, n_fft, hop, win)
    assert D.shape == (n_fft // 2 + 1, 1 + len(y) // hop)
    # scipy's ShortTime FFT on the reflect-padded signal with the same zero-padded periodic Hann
    w = A.padded_window(n_fft, win)
    np.testing.assert_allclose(w[(n_fft - win) // 2:(n_fft - win) // 2 + win], ss.get_window("hann", win, fftbins=True), atol=1e-15)
    _, _, Z = ss.stft(np.pad(y, n_fft // 2, mode="reflect"), window=w, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft,
                      boundary=None, padded=False)
    np.testing.assert_allclose(Z * w.sum(), D, atol=1e-10)
    # istft(stft(y)) == y on hop * (frames - 1) samples (window sum-of-squares normalisation)
    y2 = A.istft(D, hop, win)
    assert len(y2) == hop * (D.shape[1] - 1)
    np.testing.assert_allclose(y2, y[:len(y2)], atol=1e-12)
    # scipy's istft inverts the same matrix to the same signal (its own normalisation)
    _, y3 = ss.istft(Z, window=w, nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=False)
    np.testing.assert_allclose(y3[n_fft // 2:n_fft // 2 + len(y2)], y2, atol=1e-10)


def test_stft_pure_tone_and_reflect_padding():
    n_fft, hop, win = 2048, 256, 1024
    k = 100
    y = np.cos(2 * np.pi * k * np.arange(hop * 40) / n_fft)
    D = np.abs(A.stft(y, n_fft, hop, win))
    assert (D[:, 5:-5].argmax(0) == k).all()                      # interior frames peak at bin k
    np.testing.assert_allclose(D[k, 5:-5], 0.5 * A.padded_window(n_fft, win).sum(), rtol=1e-9)
    # frame 0 is centred on sample 0: it sees y[1024:0:-1] mirrored (np.pad reflect: the edge sample is not repeated)
    x = rng(3).standard_normal(hop * 20)
    f0 = np.concatenate([x[n_fft // 2:0:-1], x[:n_fft // 2]])
    np.testing.assert_allclose(A.stft(x, n_fft, hop, win)[:, 0], np.fft.rfft(A.padded_window(n_fft, win) * f0), atol=1e-12)


def test_window_sumsquare_and_istft_edges():
    n_fft, hop, win = 2048, 256, 1024
    wss = A.window_sumsquare(12, n_fft, hop, win)
    assert len(wss) == n_fft + hop * 11
    np.testing.assert_allclose(wss[n_fft // 2 + win:-(n_fft // 2 + win)], 1.5, atol=1e-12)   # Hann^2 at 75 % overlap sums to 3/2
    assert (wss[:(n_fft - win) // 2] == 0).all()                  # nothing reaches the zero-padded flanks of the first frame


def test_griffin_lim_restatement():
    o = A.AudioOracle(LJHPS.Audio)
    r = rng(7)
    mel = r.uniform(0.0, 1.0, (80, 40))
    S = o.linear_from_mel(mel)
    assert S.shape == (1025, 40) and (S >= (1e-10) ** 1.5).all()
    # audio.py:81-84 written out
    db = np.clip(mel, 0, 1) * 100.0 - 100.0 + 20.0
    lin = np.maximum(1e-10, np.linalg.pinv(o._build_mel_basis()) @ (10.0 ** (db * 0.05)))
    np.testing.assert_allclose(S, lin ** 1.5, rtol=1e-12)
    ang = 2 * np.pi * r.random(S.shape)
    y0 = o.griffin_lim(S, ang, 0)
    np.testing.assert_allclose(y0, A.istft(S * np.exp(1j * ang), 256, 1024), atol=1e-12)
    # a CONSISTENT spectrogram (the STFT of a real signal) is a fixed point of the iteration
    sig = r.standard_normal(256 * 39)
    D = A.stft(sig, 2048, 256, 1024)
    y = o.griffin_lim(np.abs(D), np.angle(D), 3)
    np.testing.assert_allclose(y, sig[:len(y)], atol=1e-9)
    # spectral convergence does not increase over the iterations (Griffin & Lim 1984)
    errs = []
    for it in (0, 2, 8):
        yi = o.griffin_lim(S, ang, it)
        errs.append(np.linalg.norm(np.abs(A.stft(yi, 2048, 256, 1024)) - S) / np.linalg.norm(S))
    assert errs[0] > errs[1] > errs[2]


def test_deemphasis_and_int16():
    o = A.AudioOracle(LJHPS.Audio)
    x = rng(9).standard_normal(500)
    np.testing.assert_allclose(o.inv_preemphasize(x), ss.lfilter([1], [1, -0.97], x), atol=1e-12)   # audio.py:236
    w = o.to_int16(np.array([0.0, 0.5, -1.0, 0.25]))
    assert w.dtype == np.int16 and list(w) == [0, 16383, -32767, 8191]
    assert list(o.to_int16(np.array([0.001, -0.002]))) == [3276, -6553]       # the 0.01 floor of audio.py:19


def test_oracle_reproduces_the_committed_fixture():
    """tests/golden/audio_lj.npz (oracle/make_audio_golden.py): the restatement is frozen against its own committed outputs."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_lj.npz"))
    o = A.AudioOracle(LJHPS.Audio)
    for b, n in enumerate(z["lengths"]):
        S = o.linear_from_mel(z["mels"][b, :n].T.astype(np.float64))
        np.testing.assert_allclose([S.sum(), (S ** 2).sum()], z["S_sum_%d" % b], rtol=1e-12)
        a = z["init_angles"][b, :n].T.astype(np.float64)
        for it in (0, 2, 5):
            np.testing.assert_allclose(o.griffin_lim(S, a, it), z["wav%d_it%d" % (b, it)], atol=1e-5 * np.abs(z["wav%d_it%d" % (b, it)]).max())


@pytest.mark.parametrize("n_fft,hop,win", [(2048, 256, 1024), (2048, 200, 800)])
def test_stft_istft_against_torch(n_fft, hop, win):
    """torch.stft / torch.istft implement librosa's conventions (centred zero-padded window, reflect padding, window-envelope
    normalisation) independently of this repository: a third implementation next to scipy's."""
    import torch
    y = rng(hop + win).standard_normal(hop * 41)
    w = torch.hann_window(win, periodic=True, dtype=torch.float64)
    Z = torch.stft(torch.from_numpy(y), n_fft, hop_length=hop, win_length=win, window=w, center=True, pad_mode="reflect",
                   return_complex=True)
    D = A.stft(y, n_fft, hop, win)
    np.testing.assert_allclose(D, Z.numpy(), atol=1e-10)
    # inverse of an arbitrary (inconsistent) spectrogram: the quantity Griffin-Lim needs
    r = rng(7)
    X = np.abs(D) * np.exp(2j * np.pi * r.random(D.shape))
    X[0] = X[0].real; X[-1] = X[-1].real                      # (both drop the imaginary part of DC / Nyquist)
    ref = torch.istft(torch.from_numpy(X), n_fft, hop_length=hop, win_length=win, window=w, center=True).numpy()
    got = A.istft(X, hop, win)
    assert got.shape == ref.shape == (hop * (D.shape[1] - 1),)
    np.testing.assert_allclose(got, ref, atol=1e-10)
