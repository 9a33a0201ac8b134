"""GPU parity of the vocoder step (csrc/vocoder.hip through the C ABI and the `Audio` mirror) against oracle/audio_numpy.py.

Tolerances.  The HIP path computes in fp32 (as librosa does: complex64 STFT, float32 ISTFT); the oracle in float64.  One
istft(S e^{j phase}) agrees to ~1e-6 of the signal's peak.  Griffin-Lim iterates x -> P_S(stft(istft(x))): a phase at a bin
whose magnitude is ~0 is ill-conditioned, so trajectories separate slowly; the tests bound the distance after a few
iterations relative to the signal peak and check the size-independent properties (consistent spectrogram = fixed point, spectral
convergence) on the full 60 iterations."""
import numpy as np
import pytest

from oracle import audio_numpy as A
from vaenar_tts_amd import _lib
from vaenar_tts_amd.audio import Audio, TestUtils
from vaenar_tts_amd.configs import LJHPS, DataBakerHPS, tiny_hps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(tiny_hps(), 0)
    yield e
    e.close()


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("hps", [LJHPS, DataBakerHPS], ids=["ljspeech", "databaker"])
def test_mel_to_linear(eng, hps):
    au = Audio(hps.Audio, engine=eng)
    o = A.AudioOracle(hps.Audio)
    mel = rng(1).uniform(-0.1, 1.1, (3, 37, 80)).astype(np.float32)      # includes values the clip of _denormalize cuts
    got = au.linear_from_mel_batch(mel).numpy()
    for b in range(3):
        ref = o.linear_from_mel(mel[b].T.astype(np.float64)).T
        # fp32 accumulation of 80 terms of mixed sign (pinv has negative entries): error relative to the frame's scale
        np.testing.assert_allclose(got[b], ref, rtol=2e-4, atol=2e-6 * ref.max())


@pytest.mark.parametrize("hps,T", [(LJHPS, 40), (DataBakerHPS, 33)], ids=["ljspeech", "databaker"])
def test_first_istft_and_few_iterations(eng, hps, T):
    au = Audio(hps.Audio, engine=eng)
    o = A.AudioOracle(hps.Audio)
    r = rng(T)
    mel = r.uniform(0.0, 1.0, (80, T))
    ang = 2 * np.pi * r.random((1025, T))
    S = o.linear_from_mel(mel)
    for iters, tol in ((0, 2e-6), (1, 2e-5), (3, 1e-4)):
        ref = o.griffin_lim(S, ang, iters)
        d_S = eng.to_device(np.ascontiguousarray(S.T)[None], np.float32)
        got = au._griffin_lim_batch(d_S, None, np.ascontiguousarray(ang.T)[None], n_iters=iters).numpy()[0]
        assert got.shape == ref.shape == (hps.Audio.frame_shift_sample * (T - 1),)
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (iters, np.abs(got - ref).max() / np.abs(ref).max())


def test_eight_fold_overlap(eng):
    """hop = win / 8: the general gather (up to 8 overlapping frames per sample) instead of the 4-frame one of the two reference
    configurations."""
    import copy
    hps = copy.deepcopy(LJHPS.Audio)
    hps.frame_shift_sample = 128
    au = Audio(hps, engine=eng)
    o = A.AudioOracle(hps)
    r = rng(21)
    T = 45
    S = o.linear_from_mel(r.uniform(0.0, 1.0, (80, T)))
    ang = 2 * np.pi * r.random((1025, T))
    ref = o.griffin_lim(S, ang, 2)
    d_S = eng.to_device(np.ascontiguousarray(S.T)[None], np.float32)
    got = au._griffin_lim_batch(d_S, None, np.ascontiguousarray(ang.T)[None], n_iters=2).numpy()[0]
    assert got.shape == ref.shape == (128 * (T - 1),)
    assert np.abs(got - ref).max() <= 5e-5 * np.abs(ref).max()


def test_ragged_batch_matches_single_utterances(eng):
    au = Audio(LJHPS.Audio, engine=eng)
    r = rng(5)
    T, lens = 48, [48, 31, 20]
    mels = r.uniform(0.0, 1.0, (3, T, 80)).astype(np.float32)
    ang = (2 * np.pi * r.random((3, T, 1025))).astype(np.float32)
    batch = au.inv_mel_spectrogram_batch(mels, lens, init_angles=ang, n_iters=4)
    for b, n in enumerate(lens):
        single = au.inv_mel_spectrogram_batch(mels[b:b + 1, :n], None, init_angles=ang[b:b + 1, :n], n_iters=4)[0]
        assert batch[b].shape == (256 * (n - 1),)
        np.testing.assert_array_equal(batch[b], single)           # frames past an utterance's length never contribute


def test_consistent_spectrogram_is_a_fixed_point_and_convergence(eng):
    au = Audio(LJHPS.Audio, engine=eng)
    r = rng(11)
    sig = r.standard_normal(256 * 59)
    D = A.stft(sig, 2048, 256, 1024)                               # [1025, 60]
    y = au._griffin_lim(np.abs(D), init_angles=np.angle(D))        # the full 60 iterations of hps.griffin_lim_iters
    np.testing.assert_allclose(y, sig[:len(y)], atol=2e-4)
    # random magnitudes: the spectral distance of stft(y) to S does not increase with the iteration count
    S = A.AudioOracle(LJHPS.Audio).linear_from_mel(r.uniform(0, 1, (80, 50)))
    ang = 2 * np.pi * r.random(S.shape)
    d_S = eng.to_device(np.ascontiguousarray(S.T)[None], np.float32)
    errs = []
    for it in (0, 5, 60):
        yi = au._griffin_lim_batch(d_S, None, np.ascontiguousarray(ang.T)[None], n_iters=it).numpy()[0].astype(np.float64)
        errs.append(np.linalg.norm(np.abs(A.stft(yi, 2048, 256, 1024)) - S) / np.linalg.norm(S))
    assert errs[0] > errs[1] > errs[2], errs


def test_reference_call_surface_and_wav_files(eng, tmp_path):
    """audio/utils.py:24-40: mel -> inv_mel_spectrogram(mel.T) -> inv_preemphasize -> save_wav, per utterance of a batch."""
    from scipy.io import wavfile
    tu = TestUtils(LJHPS, str(tmp_path), engine=eng)
    r = rng(2)
    mels = r.uniform(0.2, 0.8, (2, 30, 80)).astype(np.float32)
    tu.write_mels(7, mels, [30, 22], [b"a", "b"], prefix="prior")
    assert np.load(tmp_path / "prior-b-7.npy").shape == (22, 80)
    tu.synthesize_and_save_wavs(7, mels, [30, 22], [b"a", "b"], prefix="prior", seed=3)
    sr, w = wavfile.read(tmp_path / "prior-a-7.wav")
    assert sr == 22050 and w.dtype == np.int16 and w.shape == (256 * 29,) and np.abs(w).max() in (32766, 32767)    # scaled to full range, truncated (audio.py:19-20)
    # the 2-D entry point of the reference (mel.T) with a given phase draw agrees with the oracle after the full chain
    au = tu.prcocessor
    ang = 2 * np.pi * r.random((1025, 22))
    o = A.AudioOracle(LJHPS.Audio)
    au.hps.griffin_lim_iters, keep = 2, au.hps.griffin_lim_iters
    try:
        got = au.inv_preemphasize(au.inv_mel_spectrogram(mels[1, :22].T, init_angles=ang))
        ref = o.inv_preemphasize(o.inv_mel_spectrogram(mels[1, :22].T.astype(np.float64), ang, 2))
    finally:
        au.hps.griffin_lim_iters = keep
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()


def test_against_the_committed_fixture(eng):
    """tests/golden/audio_lj.npz: ragged batch, given phase draw; waveforms after 0 / 2 / 5 iterations and after de-emphasis."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "audio_lj.npz"))
    au = Audio(LJHPS.Audio, engine=eng)
    lens = [int(x) for x in z["lengths"]]
    for it, tol in ((0, 2e-6), (2, 5e-5), (5, 3e-4)):
        wavs = au.inv_mel_spectrogram_batch(z["mels"], lens, init_angles=z["init_angles"], n_iters=it)
        for b in range(len(lens)):
            ref = z["wav%d_it%d" % (b, it)]
            assert wavs[b].shape == ref.shape
            assert np.abs(wavs[b] - ref).max() <= tol * np.abs(ref).max(), (it, b, np.abs(wavs[b] - ref).max() / np.abs(ref).max())
        if it == 5:
            for b in range(len(lens)):
                ref = z["wav%d_deemph" % b]
                assert np.abs(au.inv_preemphasize(wavs[b]) - ref).max() <= 1e-3 * np.abs(ref).max()


def test_bad_arguments_fail_loudly(eng):
    S = eng.zeros((1, 8, 513))
    wav = eng.empty((1, 128 * 7))
    rc = eng.lib.vnr_voc_griffin_lim(eng.handle, S.ptr, None, 0, None, 1, 8, 1024, 128, 512, 1, wav.ptr)
    assert rc != 0 and b"2048" in eng.lib.vnr_last_error(eng.handle)
    S = eng.zeros((1, 4, 1025))
    rc = eng.lib.vnr_voc_griffin_lim(eng.handle, S.ptr, None, 0, None, 1, 4, 2048, 256, 1024, 1, wav.ptr)
    assert rc != 0                                                 # hop * (frames - 1) <= n_fft / 2: reflect padding undefined
