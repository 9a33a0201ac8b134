"""The N>1 path on CPU: world_size-2 gloo processes run the batch-sharding / barrier / max-over-ranks /
gather logic bench.py and inference.py use (the per-rank compute is replaced by the oracle on a
tiny model so the shards' results can be checked against the unsharded run)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle.vaenar_numpy import Oracle
    from vaenar_tts_amd import dist
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.synthetic import make_batch
    from vaenar_tts_amd.weights import init_weights
    r, lr, w = dist.init("gloo")
    assert (r, w) == (rank, world)
    hps = tiny_hps()
    weights = init_weights(hps, seed=5)
    batch = make_batch(5, 9, 24, latent_dim=hps.Common.latent_dim, ragged=False, temperature=1.0)
    shard = dist.shard_batch(batch, rank, world)
    assert len(shard["text_lengths"]) == (3 if rank == 0 else 2)        # 5 utterances over 2 ranks
    dist.barrier()
    mel, _ = Oracle(hps, weights, np.float64).inference(shard["ids"], shard["mel_lengths"], shard["text_lengths"], 2,
                                                        shard["eps"])
    t = dist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    full = dist.gather_to_rank0(mel)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), full)
    dist.barrier()


def test_batch_sharded_inference_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle.vaenar_numpy import Oracle
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.synthetic import make_batch
    from vaenar_tts_amd.weights import init_weights
    hps = tiny_hps()
    batch = make_batch(5, 9, 24, latent_dim=hps.Common.latent_dim, ragged=False, temperature=1.0)
    ref, _ = Oracle(hps, init_weights(hps, seed=5), np.float64).inference(batch["ids"], batch["mel_lengths"],
                                                                          batch["text_lengths"], 2, batch["eps"])
    got = np.load(tmp_path / "gathered.npy")
    np.testing.assert_allclose(got, ref, atol=1e-12)       # utterances are independent: sharding changes nothing


def test_shard_bounds_cover_everything():
    from vaenar_tts_amd.dist import shard_bounds
    for n in (1, 5, 16, 128):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
