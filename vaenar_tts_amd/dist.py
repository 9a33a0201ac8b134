"""Multi-GPU plumbing of the batch-sharded path (one process per GPU) -- standard library only.

Utterances are independent, so a global batch is split into contiguous per-rank shards and there is NO
data-path collective in inference; training exchanges gradients with RCCL inside the engine (vnr_train_step).  What the host side
needs between ranks is a control plane: a barrier, the max / mean of a scalar, gathering small host arrays on rank 0 and
broadcasting the 128-byte RCCL unique id.  That is a star of TCP connections to rank 0, keyed on the launcher's
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's environment; `bench.py --gpus N` sets the same variables itself).
No PyTorch here: torch.distributed (gloo) is only the checker of tests/test_dist_gloo.py.

torchrun's own rendezvous store already listens on MASTER_PORT, so rank 0 listens on the first free port of
MASTER_PORT + 1 ... + 32 (VNR_RDZV_PORT overrides the base) and every connection starts with a handshake that carries a job key
(hash of MASTER_ADDR, MASTER_PORT, WORLD_SIZE): a foreign listener on a candidate port is skipped, not trusted.
"""
import hashlib
import io
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b"VNRCTL1\0"
_CANDIDATES = 32
_state = {"rank": 0, "world": 1, "hub": None, "peers": None, "sock": None}


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def _job_key(addr, port, world):
    # (address, port and world size are not secret: the launcher's run id -- torchrun exports TORCHELASTIC_RUN_ID -- or VNR_JOB_SECRET is
    #  mixed in when present, so that a host that merely knows the rendezvous cannot join as a rank)
    secret = os.environ.get("VNR_JOB_SECRET") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    return hashlib.sha256(("%s:%d:%d:%s" % (addr, port, world, secret)).encode()).digest()[:16]


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


_MAX_MESSAGE = 1 << 31           # a gathered mel batch is tens of MB; anything beyond 2 GiB is a corrupt or hostile length word


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_MESSAGE:
        raise ConnectionError("control plane: message length %d exceeds the %d-byte cap" % (n, _MAX_MESSAGE))
    return _recv_exact(sock, n)


def _exchange_timeout():
    """Seconds an exchange waits for a peer (VNR_CTL_TIMEOUT, default 1800: longer than any step or checkpoint write; 0 = forever)."""
    try:
        return float(os.environ.get("VNR_CTL_TIMEOUT", "1800"))
    except ValueError:
        return 1800.0


def _bind_address(addr):
    """The interface rank 0 listens on.  MASTER_ADDR's own interface when the name resolves to an address of THIS host that the other
    ranks can reach (loopback only when the job itself is on loopback); every interface otherwise -- a name that /etc/hosts maps to
    127.0.1.1 on node 0 (Debian's default), a virtual / NAT address that is not on a local interface: binding those would lock the
    remote ranks out.  The handshake's job key authenticates peers either way.  VNR_CTL_BIND overrides ("" = all)."""
    forced = os.environ.get("VNR_CTL_BIND")
    if forced is not None:
        return forced
    if addr in ("localhost", "127.0.0.1"):
        return "127.0.0.1"
    try:
        infos = socket.getaddrinfo(addr, None, socket.AF_INET, socket.SOCK_STREAM)
    except OSError:
        return ""
    for info in infos:
        ip = info[4][0]
        if ip.startswith("127."):
            continue                      # a loopback alias of a real host name: the other nodes cannot reach it
        probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        try:
            probe.bind((ip, 0))           # succeeds only for an address of a local interface
            return ip
        except OSError:
            pass
        finally:
            probe.close()
    return ""


def init(backend=None, timeout=300.0):
    """Join the control plane.  Returns (rank, local_rank, world).  `backend` is accepted for the callers' old signature."""
    rank, local_rank, world = env_rank()
    if world <= 1 or _state["sock"] is not None or _state["peers"] is not None:
        return rank, local_rank, world
    addr = os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.setdefault("MASTER_PORT", "29511"))
    base = int(os.environ.get("VNR_RDZV_PORT", mport + 1))
    key = _job_key(addr, mport, world)
    deadline = time.monotonic() + timeout
    _state.update(rank=rank, world=world)
    if rank == 0:
        srv = None
        bind_addr, last_err = _bind_address(addr), None
        for p in range(base, base + _CANDIDATES):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((bind_addr, p))
                s.listen(world + 8)
                srv = s
                break
            except OSError as e:
                last_err = e
                s.close()
        if srv is None:
            raise RuntimeError("control plane: could not listen on %r, ports [%d, %d): %s" % (bind_addr or "*", base, base + _CANDIDATES, last_err))
        peers = {}
        srv.settimeout(1.0)
        while len(peers) < world - 1:
            if time.monotonic() > deadline:
                raise TimeoutError("control plane: %d of %d ranks joined within %.0f s" % (len(peers) + 1, world, timeout))
            try:
                c, _ = srv.accept()
            except socket.timeout:
                continue
            try:
                c.settimeout(3.0)
                hello = _recv_exact(c, len(_MAGIC) + 16 + 4)
                r = struct.unpack("<i", hello[-4:])[0]
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):-4] != key or not (0 < r < world) or r in peers:
                    c.close()
                    continue
                c.sendall(_MAGIC + key)
                c.settimeout(None)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[r] = c
            except (OSError, ConnectionError, struct.error):
                c.close()
        srv.close()
        _state.update(hub=True, peers=peers)
    else:
        sock = None
        while sock is None:
            if time.monotonic() > deadline:
                raise TimeoutError("control plane: rank %d could not reach rank 0 at %s:%d+ within %.0f s" % (rank, addr, base, timeout))
            for p in range(base, base + _CANDIDATES):
                try:
                    s = socket.create_connection((addr, p), timeout=2.0)
                except OSError:
                    continue
                try:
                    s.settimeout(3.0)
                    s.sendall(_MAGIC + key + struct.pack("<i", rank))
                    if _recv_exact(s, len(_MAGIC) + 16) == _MAGIC + key:
                        s.settimeout(None)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        sock = s
                        break
                    s.close()
                except (OSError, ConnectionError):
                    s.close()
            if sock is None:
                time.sleep(0.05)
        _state.update(hub=False, sock=sock)
    return rank, local_rank, world


def is_initialized():
    return _state["peers"] is not None or _state["sock"] is not None


def shutdown():
    """Leave the control plane (after a final barrier by the caller)."""
    if _state["peers"]:
        for c in _state["peers"].values():
            c.close()
    if _state["sock"] is not None:
        _state["sock"].close()
    _state.update(hub=None, peers=None, sock=None, rank=0, world=1)


def _exchange(payload, combine):
    """Every rank contributes `payload` (bytes); rank 0 runs combine([payload of rank 0, 1, ...]) -> {rank: reply bytes} or one reply
    for all; every rank returns its reply."""
    if not is_initialized():
        out = combine([payload])
        return out[0] if isinstance(out, dict) else out
    tmo = _exchange_timeout() or None
    if _state["hub"]:
        peers = _state["peers"]
        parts = [payload]
        for r in range(1, _state["world"]):
            peers[r].settimeout(tmo)
            try:
                parts.append(_recv(peers[r]))
            except socket.timeout:
                raise TimeoutError("control plane: rank %d did not reach the exchange within %.0f s (VNR_CTL_TIMEOUT)" % (r, tmo))
        out = combine(parts)
        for r in range(1, _state["world"]):
            _send(peers[r], out[r] if isinstance(out, dict) else out)
        return out[0] if isinstance(out, dict) else out
    _state["sock"].settimeout(tmo)
    _send(_state["sock"], payload)
    try:
        return _recv(_state["sock"])
    except socket.timeout:
        raise TimeoutError("control plane: rank %d got no answer from rank 0 within %.0f s (a rank is missing from the exchange; "
                           "VNR_CTL_TIMEOUT)" % (_state["rank"], tmo))


def shard_bounds(n, rank, world):
    """Contiguous, balanced [lo, hi) of n utterances for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(batch, rank, world):
    """Slice every per-utterance array of a batch dict (leading dim = utterance)."""
    n = len(batch["text_lengths"])
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in batch.items()}


def barrier():
    _exchange(b"", lambda parts: b"")


def _reduce(x, fn):
    def combine(parts):
        return struct.pack("<d", fn([struct.unpack("<d", p)[0] for p in parts]))
    return struct.unpack("<d", _exchange(struct.pack("<d", float(x)), combine))[0]


def max_over_ranks(x):
    return _reduce(x, max)


def mean_over_ranks(x):
    return _reduce(x, lambda v: sum(v) / len(v))      # rank order: every rank gets the same float


def gather_to_rank0(array):
    """Concatenate per-rank host arrays (equal trailing dims) on rank 0; None elsewhere."""
    if not is_initialized():
        return array
    buf = io.BytesIO()
    np.save(buf, np.ascontiguousarray(array), allow_pickle=False)
    got = {}

    def combine(parts):
        got["all"] = np.concatenate([np.load(io.BytesIO(p), allow_pickle=False) for p in parts], 0)
        return b""
    _exchange(buf.getvalue(), combine)
    return got.get("all")


def broadcast_bytes(data, src=0):
    """Control-plane broadcast of a short byte string (the 128-byte RCCL unique id of data-parallel training)."""
    if not is_initialized():
        return data
    me = _state["rank"]
    return _exchange(bytes(data) if me == src else b"", lambda parts: parts[src])
