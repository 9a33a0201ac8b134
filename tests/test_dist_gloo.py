"""The N>1 path on CPU: world_size-2 processes run the batch-sharding / barrier / max-over-ranks / gather logic bench.py,
inference.py and train.py use (the per-rank compute is replaced by the oracle on a tiny model so the shards' results can be
checked against the unsharded run).  The product's control plane (vaenar_tts_amd/dist.py) is standard-library TCP;
torch.distributed with the gloo backend runs BESIDE it in these tests as the checker: every collective of the control plane must
return what gloo returns."""
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
    import torch
    import torch.distributed as tdist
    r, lr, w = dist.init()
    assert (r, w) == (rank, world) and dist.is_initialized()
    tdist.init_process_group("gloo", rank=rank, world_size=world)           # the checker
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
    x = torch.tensor([1.0 + rank, 0.1 * (rank + 1)], dtype=torch.float64)
    tdist.all_reduce(x[:1], op=tdist.ReduceOp.MAX)
    tdist.all_reduce(x[1:], op=tdist.ReduceOp.SUM)
    assert t == float(x[0]) and abs(dist.mean_over_ranks(0.1 * (rank + 1)) - float(x[1]) / world) < 1e-15
    full = dist.gather_to_rank0(mel)
    box = [None] * world if rank == 0 else None
    tdist.gather_object(mel, box, dst=0)
    if rank == 0:
        assert np.array_equal(full, np.concatenate(box, 0))
        np.save(os.path.join(out_dir, "gathered.npy"), full)
    else:
        assert full is None
    blob = dist.broadcast_bytes(bytes(range(7, 135)) if rank == 0 else None)
    ref = [bytes(range(7, 135)) if rank == 0 else None]
    tdist.broadcast_object_list(ref, src=0)
    assert blob == ref[0]
    dist.barrier()
    tdist.barrier()
    dist.shutdown()
    assert not dist.is_initialized() and dist.max_over_ranks(3.0) == 3.0      # back to the single-process behaviour
    tdist.destroy_process_group()


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


def _dp_case(databaker=False):
    """Global batch [a, b, a, b]: the two contiguous shards hold the same two utterances, so the per-replica BatchNormalization
    batch statistics (not synchronised) EQUAL the global-batch statistics and the average of the shard
    gradients must equal the global-batch gradient exactly -- for every variable, not just in direction."""
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.synthetic import make_batch
    from vaenar_tts_amd.weights import init_weights
    hps = tiny_hps()
    if databaker:
        # BASELINE config 5's model (reference configs/hparams.py:351-474): what sets DataBakerHPS apart on the text -> mel path is the
        # vocabulary (39 symbols, :411) and the mel / text length ratio (4.21, :407: the encoder's positional step); widths as tiny_hps
        from vaenar_tts_amd.configs import DataBakerHPS
        hps.Encoder.Transformer.vocab_size = DataBakerHPS.Encoder.Transformer.vocab_size
        hps.Common.mel_text_len_ratio = DataBakerHPS.Common.mel_text_len_ratio
        assert hps.Encoder.Transformer.vocab_size == 39 and abs(hps.Common.mel_text_len_ratio - 4.21) < 1e-12
    w = init_weights(hps, seed=5)
    half = make_batch(2, 7, 16, vocab_size=hps.Encoder.Transformer.vocab_size, latent_dim=hps.Common.latent_dim, ragged=True, text_step=2, mel_step=4)
    r = np.random.Generator(np.random.PCG64(1))
    half["mels"] = r.standard_normal((2, 16, hps.Audio.num_mels))
    half["eps"] = r.standard_normal((2, 8, hps.Common.latent_dim))
    batch = {k: (np.concatenate([v, v], 0) if isinstance(v, np.ndarray) and v.shape[:1] == (2,) else v) for k, v in half.items()}
    return hps, w, batch


def _train_worker(rank, world, port, out_dir):
    """Data-parallel training, host side: the RCCL id travels as bytes, every rank keeps its shard of the global batch, and
    the rank-averaged gradient equals the gradient of the global batch (the device all-reduce of vnr_train_step is replaced
    by a gloo all-reduce of the autograd oracle's gradients: sum, then 1 / world -- the same arithmetic)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as tdist
    from oracle.vaenar_torch import TorchOracle
    from vaenar_tts_amd import dist
    dist.init()
    tdist.init_process_group("gloo", rank=rank, world_size=world)           # stands in for the device all-reduce (RCCL) below
    uid = dist.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128))
    hps, w, batch = _dp_case(databaker=bool(os.environ.get("VNR_TEST_DATABAKER")))
    sh = dist.shard_batch(batch, rank, world)
    assert sh["mels"].shape[0] == 2
    o = TorchOracle(hps, w)
    o.update_moving_stats = False
    g, sc = o.gradients(sh["ids"], sh["mels"], sh["mel_lengths"], sh["text_lengths"], 2, sh["eps"], kl_weight=1.0, dropout_seed=None)
    # the flat bucket of vnr_train_step: every trainable variable, sorted by path, one all-reduce (sum) then 1 / world
    names = sorted(g)
    flat = torch.tensor(np.concatenate([g[k].reshape(-1) for k in names]))
    tdist.all_reduce(flat)
    flat /= world
    assert abs(dist.mean_over_ranks(float(rank)) - 0.5) < 1e-12
    if rank == 0:
        np.save(os.path.join(out_dir, "avg_grad.npy"), flat.numpy())
    dist.barrier()
    dist.shutdown()
    tdist.destroy_process_group()


@pytest.mark.parametrize("databaker", [False, True], ids=["lj-shaped", "databaker-shaped"])
def test_data_parallel_gradient_average_world2(tmp_path, databaker, monkeypatch):
    """Rank shards + the flat-bucket average reproduce the 1-rank gradient of the global batch for every variable; since round 6 also for
    the DataBakerHPS shape of BASELINE config 5 (vocabulary 39, length ratio 4.21 -- reference datasets/datasets.py:179-192 is the
    only rank / size vestige of the reference, its train.py has no distribution strategy)."""
    world = 2
    if databaker:
        monkeypatch.setenv("VNR_TEST_DATABAKER", "1")
    else:
        monkeypatch.delenv("VNR_TEST_DATABAKER", raising=False)
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle.vaenar_torch import TorchOracle
    hps, w, batch = _dp_case(databaker)
    o = TorchOracle(hps, w)
    o.update_moving_stats = False
    g, _ = o.gradients(batch["ids"], batch["mels"], batch["mel_lengths"], batch["text_lengths"], 2, batch["eps"], kl_weight=1.0,
                       dropout_seed=None)
    got = np.load(tmp_path / "avg_grad.npy")
    ref = np.concatenate([g[k].reshape(-1) for k in sorted(g)])
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())


def _late_worker(rank, world, port, out_dir):
    """Ranks that start seconds apart, with a foreign listener squatting on the first candidate port."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    from vaenar_tts_amd import dist
    if rank == 0:
        time.sleep(1.5)                      # the other ranks poll until rank 0 listens
    dist.init(timeout=60.0)
    assert dist.max_over_ranks(float(rank)) == float(world - 1)
    got = dist.gather_to_rank0(np.full((rank + 1, 2), rank, np.int32))
    if rank == 0:
        assert got.shape == (sum(range(1, world + 1)), 2) and got[0, 0] == 0 and got[-1, 0] == world - 1
    dist.barrier()
    dist.shutdown()


def test_control_plane_world3_with_a_foreign_listener_and_late_rank0():
    port = _free_port()
    squat = socket.socket(); squat.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        squat.bind(("127.0.0.1", port + 1)); squat.listen(4)      # not ours: no handshake reply -> the ranks move to the next candidate
    except OSError:
        squat = None
    try:
        mp.spawn(_late_worker, args=(3, port, ""), nprocs=3, join=True)
    finally:
        if squat is not None:
            squat.close()


def test_rank0_bind_address_choice(monkeypatch):
    """rank 0 binds MASTER_ADDR's interface only when that is an address of this host the other ranks can reach; a name
    that resolves to a loopback alias (Debian's 127.0.1.1) or to nothing local falls back to every interface; VNR_CTL_BIND overrides."""
    sys.path.insert(0, ROOT)
    from vaenar_tts_amd import dist
    monkeypatch.delenv("VNR_CTL_BIND", raising=False)
    assert dist._bind_address("127.0.0.1") == "127.0.0.1" and dist._bind_address("localhost") == "127.0.0.1"
    assert dist._bind_address("no-such-host.invalid") == ""
    assert dist._bind_address("10.255.255.1") == ""                  # not an address of a local interface
    monkeypatch.setattr(socket, "getaddrinfo", lambda *a, **k: [(socket.AF_INET, socket.SOCK_STREAM, 6, "", ("127.0.1.1", 0))])
    assert dist._bind_address("node0") == ""                          # the loopback alias of a real host name
    monkeypatch.setenv("VNR_CTL_BIND", "192.0.2.7")
    assert dist._bind_address("node0") == "192.0.2.7"
