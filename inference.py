#!/usr/bin/env python3
"""Counterpart of the reference harness /root/reference/inference.py (inference_test, :84-168) on the
MI355X engine: same flags (:85-99), same call order in test_step (:128-143), one warm-up call then
timing of test_step only (:145-155), RTF print (:165-168), mels written as
``prior-{fid}-{step}.npy`` float32 [pred_len, 80] (audio/utils.py:16-22).

Differences forced by the environment: no TensorFlow -> weights come from an ``.npz`` of the
object-graph variable tree (``--ckpt_path``; ``synthetic`` = seeded random init) and batches come from
an ``.npz`` with ``ids`` [N,T] / ``text_lengths`` [N] (``--data_dir``; ``synthetic`` = seeded random
utterances) instead of TFRecords.  With torchrun (WORLD_SIZE > 1) the utterances are sharded over
the ranks -- one engine per GPU, no collective on the data path.
"""
import argparse
import os
import time

import numpy as np

from vaenar_tts_amd import dist as vdist
from vaenar_tts_amd.configs import DataBakerHPS, LJHPS
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import init_weights


def inference_test():
    parser = argparse.ArgumentParser('Inference parameters parser')
    parser.add_argument('--dataset', type=str, choices=['ljspeech', 'databaker'], default='ljspeech')
    parser.add_argument('--data_dir', type=str, default='synthetic', help=".npz with ids/text_lengths, or 'synthetic'")
    parser.add_argument('--ckpt_path', type=str, default='synthetic-0',
                        help="TensorFlow checkpoint prefix (ckpt-N, as written by the reference's train.py), .npz weight tree, or "
                             "'synthetic-<step>'")
    parser.add_argument('--test_dir', type=str, default='gpurun_out/test_dir')
    parser.add_argument('--batch_size', type=int, default=1)
    parser.add_argument('--temperature', type=float, default=0.)
    parser.add_argument('--write_mels', type=int, default=1)
    parser.add_argument('--draw_alignments', type=int, default=0, help='one figure per utterance and decoder block (inference.py:160-164)')
    parser.add_argument('--write_wavs', type=int, default=0,
                        help='also run the Griffin-Lim vocoder step on the GPU (audio/utils.py:24-40) and write prior-{fid}-{step}.wav')
    parser.add_argument('--num_utterances', type=int, default=8, help='synthetic data only')
    parser.add_argument('--seed', type=int, default=1234)
    args = parser.parse_args()
    rank, local_rank, world = vdist.init()
    ckpt_step = args.ckpt_path.split('-')[-1]                                     # inference.py:102
    os.makedirs(args.test_dir, exist_ok=True)
    hparams = {'ljspeech': LJHPS, 'databaker': DataBakerHPS}[args.dataset]        # inference.py:107
    rf = hparams.Common.final_reduction_factor

    if args.data_dir == 'synthetic':
        data = make_batch(args.num_utterances, 96, 2 * rf, vocab_size=hparams.Encoder.Transformer.vocab_size,
                          ragged=True, seed=args.seed, text_step=5)
    else:
        with np.load(args.data_dir) as z:
            data = {k: z[k] for k in z.files}
    n_total = len(data['text_lengths'])
    lo, hi = vdist.shard_bounds(n_total, rank, world)
    fids = np.arange(lo, hi)

    model = VAENAR(hparams, device=local_rank)                                    # inference.py:121
    if args.ckpt_path.startswith('synthetic'):
        model.load_weights(init_weights(hparams, seed=args.seed, mode='synthetic', include_posterior=False))
    else:
        model.load_weights(args.ckpt_path)                                        # replaces Checkpoint.restore :122-123
    model.prior.seed(args.seed + rank)                                            # device noise stream (prior.py:35) per rank
    tester = None
    if args.write_wavs:
        from vaenar_tts_amd.audio import TestUtils
        tester = TestUtils(hparams, args.test_dir, engine=model.engine)          # inference.py:112

    def test_step(t, t_l):                                                        # inference.py:128-143
        text_pos_step = np.float32(model.mel_text_len_ratio) / np.float32(rf)
        text_embd = model.text_encoder(t, t_l, pos_step=text_pos_step, training=False)
        predicted_lengths = model.length_predictor(text_embd, t_l, training=False)
        predicted_m_l = predicted_lengths.numpy().astype(np.int32)               # tf.cast(float, int32) :135
        reduced_pred_ml = (predicted_m_l + 80 + rf - 1) // rf                    # :136-137
        # temperature > 0: the initial noise is drawn on the device (vnr_random_normal), nothing is uploaded
        prior_latents, _ = model.prior.sample(reduced_pred_ml, text_embd, t_l, training=False,
                                              temperature=args.temperature, return_logprobs=False)
        _, prior_dec_outs, prior_dec_alignments = model.decoder(
            prior_latents, text_embd, reduced_pred_ml, t_l, training=False, reduction_factor=rf)
        return prior_dec_outs, predicted_m_l + 80, prior_dec_alignments

    def batches():
        for s in range(lo, hi, args.batch_size):
            e = min(hi, s + args.batch_size)
            tl = data['text_lengths'][s:e].astype(np.int32)
            yield np.arange(s, e), data['ids'][s:e, :int(tl.max())].astype(np.int32), tl

    for _, texts, t_lengths in batches():                                         # warm-up, inference.py:145-147
        test_step(texts, t_lengths)
        break
    time_consumed, durations = 0., 0.
    for ids, texts, t_lengths in batches():
        time_begin = time.time()
        prior_outs, pred_m_lens, prior_ali = test_step(texts, t_lengths)
        outs = prior_outs.numpy()                                                 # device -> host ends the step
        time_end = time.time()
        time_consumed += time_end - time_begin
        durations += np.sum(pred_m_lens) * hparams.Audio.frame_shift_sample / hparams.Audio.sample_rate   # inference.py:155
        if args.write_mels:                                                       # audio/utils.py:16-22
            for i, fid in enumerate(ids):
                np.save(os.path.join(args.test_dir, 'prior-{}-{}.npy'.format(fid, ckpt_step)),
                        outs[i, :pred_m_lens[i]].astype(np.float32))
        if tester is not None:                                                    # audio/utils.py:24-40
            tester.synthesize_and_save_wavs(ckpt_step, outs, np.minimum(pred_m_lens, outs.shape[1]), list(ids), prefix='prior',
                                            seed=args.seed)
        if args.draw_alignments:                                                  # inference.py:160-164
            from vaenar_tts_amd.audio.utils import TestUtils
            drawer = tester if tester is not None else TestUtils(hparams, args.test_dir, engine=model.engine)
            for k in prior_ali:
                drawer.multi_draw_attention_alignments(prior_ali[k].numpy(), texts, t_lengths, (np.asarray(pred_m_lens) + rf - 1) // rf,
                                                       ckpt_step, list(ids), 'prior-{}'.format(k))
    time_consumed = vdist.max_over_ranks(time_consumed)
    durations = float(np.sum(vdist.gather_to_rank0(np.array([durations])))) if world > 1 else durations
    if rank == 0:
        print('Total time consumed is {} Secs,total synthesis duration is {} Secs,Average RTF is {}.'.format(
            time_consumed, durations, time_consumed / max(durations, 1e-9)))


def synthesize_from_text(argv):
    """Counterpart of reference inference.py:14-83: a text file (one sentence per line) -> character ids (english cleaners /
    pinyin syllables) -> test_step -> Griffin-Lim -> ``test-{i}-{step}.wav`` (alignment plots are not drawn)."""
    from vaenar_tts_amd import texts as vtexts
    from vaenar_tts_amd.audio import TestUtils
    parser = argparse.ArgumentParser('Synthesis parameters parser')
    parser.add_argument('--dataset', type=str, choices=['ljspeech', 'databaker'], default='ljspeech')
    parser.add_argument('--text', type=str, required=True, help='text file with one sentence per line (databaker: TONE3 pinyin syllables)')
    parser.add_argument('--ckpt_path', type=str, default='synthetic-0')
    parser.add_argument('--test_dir', type=str, default='gpurun_out/test_dir')
    parser.add_argument('--temperature', type=float, default=0.)
    parser.add_argument('--seed', type=int, default=1234)
    args = parser.parse_args(argv)
    ckpt_step = args.ckpt_path.split('-')[-1]
    os.makedirs(args.test_dir, exist_ok=True)
    hparams = {'ljspeech': LJHPS, 'databaker': DataBakerHPS}[args.dataset]
    rf = hparams.Common.final_reduction_factor
    to_array = vtexts.text_to_array if args.dataset == 'ljspeech' else vtexts.pinyin_to_array
    with open(args.text, 'r') as f:
        arrays = [to_array(line.strip(), hparams) for line in f if line.strip()]
    text_batch, text_lens = vtexts.pad_batch(arrays)                              # inference.py:43-53
    model = VAENAR(hparams, device=0)
    if args.ckpt_path.startswith('synthetic'):
        model.load_weights(init_weights(hparams, seed=args.seed, mode='synthetic', include_posterior=False))
    else:
        model.load_weights(args.ckpt_path)
    tester = TestUtils(hparams, args.test_dir, engine=model.engine)
    text_pos_step = np.float32(model.mel_text_len_ratio) / np.float32(rf)         # inference.py:58-72
    text_embd = model.text_encoder(text_batch, text_lens, pos_step=text_pos_step, training=False)
    predicted_m_l = model.length_predictor(text_embd, text_lens, training=False).numpy().astype(np.int32)
    reduced_pred_ml = (predicted_m_l + 80 + rf - 1) // rf
    model.prior.seed(args.seed)
    prior_latents, _ = model.prior.sample(reduced_pred_ml, text_embd, text_lens, training=False, temperature=args.temperature,
                                          return_logprobs=False)
    _, prediction, _ = model.decoder(prior_latents, text_embd, reduced_pred_ml, text_lens, training=False)
    pred_lens = predicted_m_l + 80
    outs = prediction.numpy()
    ids = [str(i) for i in range(len(text_lens))]
    tester.synthesize_and_save_wavs(ckpt_step, outs, np.minimum(pred_lens, outs.shape[1]), ids, prefix='test', seed=args.seed)


if __name__ == '__main__':
    import sys
    if '--text' in sys.argv:
        synthesize_from_text(sys.argv[1:])
    else:
        inference_test()
