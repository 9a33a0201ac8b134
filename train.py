#!/usr/bin/env python3
"""Counterpart of the reference harness /root/reference/train.py on the MI355X engine: the same flow --
init_step on the first batch (:176-179, :262-267), one train_step at max_reduction_factor (:263-267), then per epoch the
KL-weight schedule (:230-234, :270-271), the reduction-factor schedule (:236-243), train_one_epoch (:181-204),
dev_one_epoch (:207-225) and a checkpoint (:300-303) -- with the model calls of :121-179 going to
vaenar_tts_amd.models.VAENAR (training-mode forward, backward and Adam on the GPU: vnr_train_step).

Differences forced by the environment: no TensorFlow -> batches are synthetic (``--data_dir synthetic``: seeded random
utterances of ``--t_text`` x ``--t_mel``) or a directory of the reference's ``{train,dev}-*.tfrecords`` files, read without
TensorFlow by vaenar_tts_amd/tf_record_utils.py.  Checkpoints are what the reference's tf.train.Checkpoint(step, optimizer, model) +
CheckpointManager(max_to_keep=20) write (:246-249): TensorFlow tensor bundles ``ckpt-<save_counter>.index`` / ``.data-*`` holding
the variables, Adam's slots and counters and the epoch counter, plus the ``checkpoint`` state file ``latest_checkpoint`` reads
(vaenar_tts_amd/tf_checkpoint.py, no TensorFlow); a restart restores all three, so Adam resumes with its moments and bias correction.
With torchrun (WORLD_SIZE > 1) the run is data-parallel: the batch is sharded by utterance over the ranks, every rank
holds a replica, and the flat gradient is all-reduced with RCCL over xGMI inside vnr_train_step (one exchange per step);
the host control plane (vaenar_tts_amd/dist.py: TCP to rank 0, standard library) only carries the RCCL id, barriers and the averaged log scalars.
"""
import argparse
import os
import time

import numpy as np

from vaenar_tts_amd import dist as vdist
from vaenar_tts_amd.configs import DataBakerHPS, LJHPS, tiny_hps
from vaenar_tts_amd.models import VAENAR
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.tf_checkpoint import CheckpointManager
from vaenar_tts_amd.weights import init_weights


def synthetic_batches(hps, n_batches, batch_size, t_text, t_mel, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    out = []
    for i in range(n_batches):
        b = make_batch(batch_size, t_text, t_mel, vocab_size=hps.Encoder.Transformer.vocab_size,
                       latent_dim=hps.Common.latent_dim, ragged=True, seed=seed + i, text_step=1, mel_step=3)
        b["mels"] = r.standard_normal((batch_size, t_mel, hps.Audio.num_mels)).astype(np.float32)
        out.append(b)
    return out


NOISE_STRIDE = 1 << 26          # Philox counter blocks (4 normals each) reserved per training iteration: 2.7e8 draws


def noise_schedule(seed, world, rank, it):
    """(noise_seed, noise_offset, dropout_seed) of Adam iteration ``it`` (1-based) on ``rank``: the reparameterisation noise
    (posterior.py:35) is one Philox stream per rank keyed by seed * world + rank, iteration ``it`` owns the counter blocks
    [it, it + 1) * NOISE_STRIDE (4 normals each), the Dropout masks are keyed by (seed + 7919 it) * world + rank.  Pure function of
    (seed, world, rank, it): a run restarted from a checkpoint at iteration k draws for k + 1 exactly what the uninterrupted run
    would, shards never share noise, and no iteration replays an earlier one."""
    return seed * world + rank, it * NOISE_STRIDE, (seed + 7919 * it) * world + rank


def get_reduction_factor(hps, ep):                          # train.py:236-243
    intervals, rfs = hps.Train.reduce_interval, hps.Train.reduction_factors
    i = 0
    while i < len(intervals) and intervals[i] <= ep:
        i += 1
    return rfs[i - 1 if i > 0 else 0]


def main():
    ap = argparse.ArgumentParser('Training parameters parser')
    ap.add_argument('--dataset', type=str, choices=['ljspeech', 'databaker', 'tiny'], default='ljspeech')
    ap.add_argument('--data_dir', type=str, default='synthetic')
    ap.add_argument('--model_dir', type=str, default='gpurun_out/model_dir')
    ap.add_argument('--log_dir', type=str, default='gpurun_out/log_dir')
    ap.add_argument('--test_dir', type=str, default=None, help='directory to save test results (default <log_dir>/test)')
    ap.add_argument('--test_interval', type=int, default=None, help='epochs between test syntheses (default hps.Train.test_interval = 50)')
    ap.add_argument('--epochs', type=int, default=2, help='the reference trains hps.Train.epochs = 2000')
    ap.add_argument('--steps_per_epoch', type=int, default=4, help='synthetic data only')
    ap.add_argument('--batch_size', type=int, default=None, help='GLOBAL batch (default hps.Train.train_batch_size)')
    ap.add_argument('--t_text', type=int, default=128)
    ap.add_argument('--t_mel', type=int, default=800)
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--deterministic', type=int, default=1,
                    help="1 (default, like the reference's TF_DETERMINISTIC_OPS=1 + pinned seeds, train.py:17-32): every gradient is "
                         "accumulated in a fixed order, two runs with the same seed give the same bits; 0: float atomics (about 10 %% faster)")
    args = ap.parse_args()
    hps = {'ljspeech': LJHPS, 'databaker': DataBakerHPS, 'tiny': tiny_hps()}[args.dataset]
    rank, local_rank, world = vdist.init()
    seed = hps.Train.random_seed if args.seed is None else args.seed
    os.makedirs(args.model_dir, exist_ok=True)
    test_dir = args.test_dir or os.path.join(args.log_dir, 'test')
    test_interval = args.test_interval or hps.Train.test_interval
    gb = args.batch_size or hps.Train.train_batch_size
    assert gb % world == 0, "global batch must divide over the ranks"

    # 1. data (train.py:61-111): every rank builds the same global batches and keeps its shard
    if args.data_dir == 'synthetic':
        train = synthetic_batches(hps, args.steps_per_epoch, gb, args.t_text, args.t_mel, seed)
        dev = synthetic_batches(hps, 1, gb, args.t_text, args.t_mel, seed + 1000)
        test = synthetic_batches(hps, 1, hps.Train.test_batch_size, args.t_text, args.t_mel, seed + 2000)
        for b in test:
            b["fids"] = ["synthetic%02d" % i for i in range(len(b["text_lengths"]))]
    else:                                                     # the reference's TFRecord files (tf_record_utils.py)
        from vaenar_tts_amd.tf_record_utils import TFRecordWriter
        rec = TFRecordWriter(save_dir=args.data_dir)

        def load(mode, bs, shuffle):
            out = []
            for _f, texts, mels, tl, ml in rec.create_dataset(
                    hps.Dataset.buffer_size, hps.Dataset.num_parallel_reads, hps.Dataset.pad_factor, bs, hps.Audio.num_mels,
                    hps.Train.shuffle_buffer, shuffle, rec.get_tfrecords_list(mode), seed=seed):
                if len(tl) == bs or mode == 'test':           # the data-parallel shards need full batches
                    out.append({"fids": _f, "ids": texts, "mels": mels, "text_lengths": tl, "mel_lengths": ml})
            return out
        train, dev = load('train', gb, hps.Train.shuffle), load('dev', gb, False)
        test = load('test', hps.Train.test_batch_size, False) if rec.get_tfrecords_list('test') else []      # train.py:102-110
    train = [vdist.shard_batch(b, rank, world) for b in train]
    dev = [vdist.shard_batch(b, rank, world) for b in dev]

    # 2. model + optimizer (train.py:114-117) and the checkpoint manager (train.py:246-255)
    manager = CheckpointManager(args.model_dir, max_to_keep=20)
    latest = manager.latest_checkpoint                        # the state file's entry (numeric order of the files without one)
    model = VAENAR(hps, device=local_rank, weights=init_weights(hps, seed=seed, mode='reference'))
    model.engine.set_option("deterministic", 1 if args.deterministic else 0)
    step = 0
    if latest:                                                # every rank restores variables, Adam slots and counters itself
        step = model.restore_checkpoint(latest)
        if rank == 0:
            print("Restored from {}".format(latest))
    elif rank == 0:
        print("Initializing from scratch.")
    if world > 1:                                             # data-parallel: bind the RCCL communicator
        uid = vdist.broadcast_bytes(model.engine.comm_unique_id() if rank == 0 else None)
        model.engine.comm_init(world, rank, uid)
    kw_init, kw_end, kw_epochs = hps.Train.kl_weight_init, hps.Train.kl_weight_end, hps.Train.kl_weight_increase_epoch
    kw_step = (kw_end - kw_init) / kw_epochs                 # train.py:230-234

    def save():                                               # manager.save(): rank 0 writes (BatchNorm moving statistics are rank 0's)
        path = None
        if rank == 0:
            path = manager.save(lambda prefix, n: model.save_checkpoint(prefix, step=step, save_counter=n))
        vdist.barrier()
        return path

    if not latest:                                            # train.py:256-267
        b = train[0]
        model.prior.seed(seed * world + rank)                 # (the initial step draws from counter range 0)
        model.init(text_inputs=b["ids"], mel_lengths=b["mel_lengths"], text_lengths=b["text_lengths"], dropout_seed=seed)
        if world > 1:
            model.engine.comm_broadcast_weights()             # ActNorm init and BN statistics of rank 0 everywhere
        path = save()
        if rank == 0:
            print("Initial checkpoint for step {}: {}".format(step, path))
        out = model.train_step(b["ids"], b["mels"], b["text_lengths"], b["mel_lengths"], kw_init, hps.Common.max_reduction_factor,
                               dropout_seed=seed + rank)
        if rank == 0:
            print('Initial step: total {:.6f}, mel-l2 {:.6f}, kl {:.3f}, len-l2 {:.3f}'.format(*out))

    it = model.engine.get_optimizer_step()                    # Adam iterations so far: dropout seeds continue after a restart
    # reparameterisation noise (posterior.py:35, tf.random.normal): one device stream per rank, keyed by (seed, rank); iteration
    # `it` owns the counter range [it * NOISE_STRIDE, (it + 1) * NOISE_STRIDE), so the shards of a data-parallel step see
    # independent noise and a restarted run continues the stream instead of replaying it from offset 0
    model.prior.seed(seed * world + rank)
    for epoch in range(step + 1, args.epochs + 1):            # train.py:269-306
        kw = kw_init + kw_step * epoch if epoch <= kw_epochs else kw_end
        rf = get_reduction_factor(hps, epoch)
        if rank == 0:
            print('Training Epoch {}, kl weight is {}, reduction factor is {}...'.format(epoch, kw, rf))
        t0 = time.time()
        acc = np.zeros(4)
        for s, b in enumerate(train):                         # train_one_epoch, train.py:181-204
            ts = time.time()
            it += 1
            model.prior.noise_seed, model.prior.noise_offset, dseed = noise_schedule(seed, world, rank, it)
            out = model.train_step(b["ids"], b["mels"], b["text_lengths"], b["mel_lengths"], kw, rf, dropout_seed=dseed)
            out = [vdist.mean_over_ranks(x) for x in out]
            acc += out
            if rank == 0:
                print('Step {}: total {:.6f}, mel-l2 {:.6f}, kl {:.3f}, len-l2 {:.3f}, time {:.3f}'.format(s, *out, time.time() - ts))
        acc /= len(train)
        frames = sum(int(b["mel_lengths"].sum()) for b in train) * world
        if rank == 0:
            print('\nTraining Epoch {} finished in {:.3f} Secs ({:.0f} mel-frames/s)'.format(epoch, time.time() - t0, frames / (time.time() - t0)))
        dacc = np.zeros(4)
        for b in dev:                                         # dev_one_epoch, train.py:207-225 (training=False)
            _, l2, kl, ll, _ = model(b["ids"], b["mels"], b["mel_lengths"], b["text_lengths"], reduction_factor=rf,
                                     training=False, reduce_loss=True, return_alignments=False)
            kl = float(kl)
            dacc += [vdist.mean_over_ranks(float(l2) + kw * kl + hps.Train.length_weight * float(ll)),
                     vdist.mean_over_ranks(float(l2)), vdist.mean_over_ranks(kl), vdist.mean_over_ranks(float(ll))]
        dacc /= len(dev)
        if rank == 0:
            print('Epoch {}:  train-total {}, train-mel-l2 {}, train-kl {},train-len-l2 {}, dev-total {}, dev-l2 {}, dev-kl {}, '
                  'dev-len-l2 {}'.format(epoch, *acc, *dacc))
        path = save()                                         # train.py:300-303: save, THEN step += 1 (the stored counter lags by one)
        if rank == 0:
            print("Saved checkpoint for epoch {}: {}".format(step, path))
        step += 1
        if epoch % test_interval == 0 and rank == 0 and test:     # train.py:308-325: test_step -> wavs (+ the predicted mels) of ONE test batch
            print('Testing ...')
            from vaenar_tts_amd.audio.utils import TestUtils
            os.makedirs(test_dir, exist_ok=True)
            tester = TestUtils(hps, test_dir, engine=model.engine)
            tb = test[0]
            mel, _ali = model.inference(tb["ids"], tb["mel_lengths"], tb["text_lengths"], reduction_factor=rf)      # test_step, train.py:163-171
            mel = mel.numpy()
            ml = np.minimum(np.asarray(tb["mel_lengths"]), mel.shape[1])
            try:
                tester.synthesize_and_save_wavs(epoch, mel, ml, tb["fids"], 'test')
            except Exception as e:                             # (the reference swallows everything here, train.py:317-318)
                print('Something wrong with the generated waveform! ({})'.format(e))
            tester.write_mels(epoch, mel, ml, tb["fids"], 'test')
            try:                                               # train.py:319-323: the figures of the test batch (matplotlib, host only)
                tester.draw_melspectrograms(epoch, mel, ml, tb["fids"], 'test')
                for k in _ali:
                    tester.multi_draw_attention_alignments(_ali[k].numpy(), tb["ids"], tb["text_lengths"], (ml + rf - 1) // rf, epoch,
                                                           tb["fids"], 'test-{}'.format(k))
            except RuntimeError as e:
                print('figures skipped: {}'.format(e))
            print('test finished, check {} for the results'.format(test_dir))
        vdist.barrier()
    vdist.barrier()
    model.engine.close()


if __name__ == '__main__':
    main()
