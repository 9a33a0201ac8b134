"""The reference's on-disk formats without TensorFlow (F3 TFRecord input, F2 checkpoints): known-answer
vectors of the published formats, round trips, and the padded-batch / batch-shuffle semantics of create_dataset."""
import os
import struct

import numpy as np
import pytest

from vaenar_tts_amd import tf_record_utils as tfr
from vaenar_tts_amd._lib import crc32c


def test_crc32c_known_answers():
    assert crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert crc32c(bytes(32)) == 0x8A9136AA and crc32c(b"\xff" * 32) == 0x62A8AB43     # RFC 3720 B.4
    assert crc32c(b"6789", crc32c(b"12345")) == 0xE3069283          # incremental
    assert tfr.masked_crc32c(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def test_example_wire_format_known_answer():
    # tf.train.Example(features=Features(feature={'a': Feature(int64_list=Int64List(value=[1]))})).SerializeToString()
    entry = tfr._ld(1, tfr._ld(1, b"a") + tfr._ld(2, tfr.TFRecordWriter._int64_feature(1)))
    assert tfr._ld(1, entry) == b"\n\x0c\n\n\n\x01a\x12\x05\x1a\x03\n\x01\x01"
    # tf.io.serialize_tensor(tf.constant([1, 2], tf.int64))
    assert tfr.serialize_tensor(np.array([1, 2], np.int64)) == b"\x08\t\x12\x04\x12\x02\x08\x02\"\x10" + struct.pack("<qq", 1, 2)
    # a scalar float tensor stored in float_val form parses too: TensorProto{dtype: DT_FLOAT, tensor_shape: {}, float_val: 2.5}
    assert tfr.parse_tensor(b"\x08\x01\x12\x00\x2d" + struct.pack("<f", 2.5)) == np.float32(2.5)


def test_tfrecord_round_trip_and_dataset(tmp_path):
    r = np.random.Generator(np.random.PCG64(0))
    data_dir, save_dir = tmp_path / "data", tmp_path / "rec"
    for d in (data_dir / "texts", data_dir / "mels", save_dir):
        os.makedirs(d)
    fids = ["LJ%03d" % i for i in range(7)]
    items = {}
    for i, fid in enumerate(fids):
        text = r.integers(1, 43, 5 + i).astype(np.int64)            # the reference stores int64 text and float64 mels
        mel = r.standard_normal((20 + 3 * i, 80))
        np.save(data_dir / "texts" / (fid + ".npy"), text); np.save(data_dir / "mels" / (fid + ".npy"), mel)
        items[fid] = (text, mel)
    for mode, ids in (("train", fids[:5]), ("dev", fids[5:6]), ("test", fids[6:])):
        (data_dir / (mode + ".txt")).write_text("\n".join(ids) + "\n")
    w = tfr.TFRecordWriter(train_split=2, data_dir=str(data_dir), save_dir=str(save_dir))
    w.write_all()
    files = w.get_tfrecords_list("train")
    assert [os.path.basename(f) for f in files] == ["train-0.tfrecords", "train-1.tfrecords"]
    # framing: length, masked crc of the length, payload, masked crc of the payload
    raw = open(files[0], "rb").read()
    (n,) = struct.unpack_from("<Q", raw, 0)
    assert struct.unpack_from("<I", raw, 8)[0] == tfr.masked_crc32c(raw[:8])
    assert struct.unpack_from("<I", raw, 12 + n)[0] == tfr.masked_crc32c(raw[12:12 + n])
    # a flipped payload byte is detected
    bad = bytearray(raw); bad[20] ^= 1
    (tmp_path / "bad.tfrecords").write_bytes(bytes(bad))
    with pytest.raises(IOError):
        list(tfr.TFRecordWriter.read_records(str(tmp_path / "bad.tfrecords")))
    # parse_example: dtypes and values of tf_record_utils.py:108-124
    seen = {}
    for f in files:
        for rec in tfr.TFRecordWriter.read_records(f):
            fid, text, mel, tl, ml = w.parse_example(rec)
            seen[fid.decode()] = (text, mel, tl, ml)
    assert sorted(seen) == fids[:5]
    for fid, (text, mel, tl, ml) in seen.items():
        assert text.dtype == np.int32 and mel.dtype == np.float32 and tl.dtype == np.int32
        np.testing.assert_array_equal(text, items[fid][0]); np.testing.assert_allclose(mel, items[fid][1].astype(np.float32))
        assert tl == len(items[fid][0]) and ml == items[fid][1].shape[0]
    # create_dataset: padded batches (zeros), optional batch-level shuffle
    batches = list(w.create_dataset(0, 1, 0, 2, 80, 4, False, files))
    assert [len(b[0]) for b in batches] == [2, 2, 1]
    fb, texts, mels, tls, mls = batches[0]
    assert texts.shape == (2, tls.max()) and mels.shape == (2, mls.max(), 80)
    short = int(np.argmin(tls))
    assert (texts[short, tls[short]:] == 0).all() and (mels[short, mls[short]:] == 0).all()
    sh = list(w.create_dataset(0, 1, 0, 2, 80, 4, True, files, seed=3))
    assert sorted(tuple(b[0]) for b in sh) == sorted(tuple(b[0]) for b in batches)       # whole batches move, contents do not
    # pad_factor: frames padded up to a multiple (pre_pad :96-106)
    w.pad_factor = 8
    assert w.pre_pad(np.ones((21, 80), np.float32)).shape == (24, 80)


# ---- TensorFlow checkpoints (tensor bundles), SURVEY section 8f F2 -----------------------------------------------------------
def _hand_table(path, key, value):
    """An SSTable with ONE data block holding ONE entry, assembled here byte by byte from the leveldb format description
    (independent of vaenar_tts_amd.tf_checkpoint's writer)."""
    from vaenar_tts_amd.tf_record_utils import _varint as vi

    def mask(c):
        return ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff

    def block(entries):
        b = b"".join(vi(0) + vi(len(k)) + vi(len(v)) + k + v for k, v in entries)     # no prefix sharing
        b += b"".join(struct.pack("<I", 0) for _ in range(1)) + struct.pack("<I", 1)   # one restart at offset 0
        return b

    out = bytearray()

    def emit(b):
        off = len(out)
        out.extend(b + b"\x00" + struct.pack("<I", mask(crc32c(b + b"\x00"))))
        return off, len(b)
    d = emit(block([(key, value)]))
    m = emit(block([]))
    i = emit(block([(key, vi(d[0]) + vi(d[1]))]))
    footer = vi(m[0]) + vi(m[1]) + vi(i[0]) + vi(i[1])
    out.extend(footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57))
    open(path, "wb").write(bytes(out))


def test_checkpoint_index_hand_assembled(tmp_path):
    from vaenar_tts_amd import tf_checkpoint as ck
    p = str(tmp_path / "t.index")
    _hand_table(p, b"model/x/.ATTRIBUTES/VARIABLE_VALUE", b"\x08\x01")
    assert ck.read_index(p) == {b"model/x/.ATTRIBUTES/VARIABLE_VALUE": b"\x08\x01"}
    raw = bytearray(open(p, "rb").read()); raw[3] ^= 0x40
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        ck.read_index(p)


def test_checkpoint_round_trip_and_model_mapping(tmp_path):
    from vaenar_tts_amd import tf_checkpoint as ck
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.weights import init_weights, weight_spec
    hps = tiny_hps()
    w = init_weights(hps, seed=3)
    prefix = str(tmp_path / "ckpt-7")
    extra = {"optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE": np.array(7, np.int64),
             "step/.ATTRIBUTES/VARIABLE_VALUE": np.array(7, np.int64),
             "model/decoder/pre_projection/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE":
                 np.zeros(weight_spec(hps)["decoder/pre_projection/kernel"], np.float32)}
    tensors = {"model/%s%s" % (p, ck.SUFFIX): a for p, a in w.items()}
    tensors.update(extra)
    ck.write_checkpoint(prefix, tensors)
    names = {k for k, _, _ in ck.list_variables(prefix)}
    assert names == set(tensors) and len(names) > 300               # many 4 KB index blocks, restart points every 16 keys
    back = ck.read_checkpoint(prefix)
    assert back["step/.ATTRIBUTES/VARIABLE_VALUE"] == 7 and back["step/.ATTRIBUTES/VARIABLE_VALUE"].dtype == np.int64
    got = ck.load_model_weights(prefix, hps)
    assert list(got) == list(weight_spec(hps))                      # the variable tree, optimizer slots and counters dropped
    for p in w:
        np.testing.assert_array_equal(got[p], w[p])
    # a flipped byte in the data shard is caught by the per-tensor checksum
    dpath = prefix + ".data-00000-of-00001"
    raw = bytearray(open(dpath, "rb").read()); raw[100] ^= 1
    open(dpath, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        ck.read_checkpoint(prefix)
    # strict loading reports what is missing
    ck.save_model_weights(str(tmp_path / "partial"), {k: v for k, v in w.items() if not k.startswith("posterior/")})
    with pytest.raises(KeyError):
        ck.load_model_weights(str(tmp_path / "partial"), hps)
    assert "posterior/pos_weight" not in ck.load_model_weights(str(tmp_path / "partial"), hps, include_posterior=False)


def test_object_graph_of_saved_model_weights(tmp_path):
    """save_model_weights writes the _CHECKPOINTABLE_OBJECT_GRAPH entry object-based restore walks: every variable is reachable
    from node 0 through children whose local names spell its attribute path, and carries its checkpoint key."""
    from vaenar_tts_amd import tf_checkpoint as ck
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.weights import init_weights
    hps = tiny_hps()
    w = init_weights(hps, seed=3, mode="synthetic")
    prefix = str(tmp_path / "ckpt-7")
    ck.save_model_weights(prefix, w)
    got = ck.read_checkpoint(prefix, with_strings=True)
    nodes = ck.parse_object_graph(got[ck.OBJECT_GRAPH_KEY])
    assert nodes[0][0] == [(1, "model")] and nodes[0][1] == []
    for path in w:
        nid = 0
        for part in ["model"] + path.split("/"):
            nxt = [c for c, name in nodes[nid][0] if name == part]
            assert len(nxt) == 1, (path, part)
            nid = nxt[0]
        assert nodes[nid][1] == [("VARIABLE_VALUE", "model/" + path, "model/" + path + ck.SUFFIX)]
        np.testing.assert_array_equal(got["model/" + path + ck.SUFFIX], w[path])
    assert sum(len(a) for _, a in nodes) == len(w)                       # exactly one attribute per variable
    # known answer of the proto encoding for a two-variable tree: node0{child 1 'model'}, node1{child 2 'a', child 4 'b'},
    # node2{child 3 '0'}, node3{attr}, node4{attr}
    g = ck.object_graph_proto(["a/0", "b"], root="model")
    n = ck.parse_object_graph(g)
    assert [c for c, _ in n] == [[(1, "model")], [(2, "a"), (4, "b")], [(3, "0")], [], []]
    assert g[:11] == bytes([0x0a, 0x0b, 0x0a, 0x09, 0x08, 0x01, 0x12, 0x05]) + b"mod"
    # the string tensor's on-disk layout: varint length, masked CRC-32C of the uint32 length, bytes
    raw = open(prefix + ".data-00000-of-00001", "rb").read()
    e = ck._parse_entry(ck.read_index(prefix + ".index")[ck.OBJECT_GRAPH_KEY.encode()])
    blob = raw[e["offset"]:e["offset"] + e["size"]]
    ln, pos = ck._read_varint(blob, 0)
    assert ln == len(got[ck.OBJECT_GRAPH_KEY]) and blob[pos + 4:] == got[ck.OBJECT_GRAPH_KEY]
    assert blob[pos:pos + 4] == struct.pack("<I", ck._mask(ck.crc32c(struct.pack("<I", ln))))
    # and load_model_weights is unaffected by the extra entry
    w2 = ck.load_model_weights(prefix, hps)
    assert set(w2) == set(w)


# ---- the training checkpoint (train.py:246-255): model + Adam slots + counters, CheckpointManager -----------------------------
def _tiny_state():
    from vaenar_tts_amd.configs import tiny_hps
    from vaenar_tts_amd.weights import init_weights, is_trainable
    hps = tiny_hps()
    w = {k: np.asarray(v, np.float32) for k, v in init_weights(hps, seed=1).items()}
    r = np.random.default_rng(0)
    m = {k: r.standard_normal(a.shape).astype(np.float32) for k, a in w.items() if is_trainable(k)}
    v = {k: np.abs(r.standard_normal(a.shape)).astype(np.float32) for k, a in w.items() if is_trainable(k)}
    return hps, w, m, v


def test_training_checkpoint_round_trip_and_keys(tmp_path):
    from vaenar_tts_amd import tf_checkpoint as tc
    hps, w, m, v = _tiny_state()
    prefix = str(tmp_path / "ckpt-3")
    tc.save_training_checkpoint(prefix, w, m, v, iterations=1234567890123, step=41, save_counter=3, learning_rate=1.25e-4)
    keys = {k: (tuple(s), d) for k, s, d in tc.list_variables(prefix)}
    # the names TF2's object-based saver gives tf.train.Checkpoint(step, optimizer, model) with a Keras Adam
    assert keys["step/.ATTRIBUTES/VARIABLE_VALUE"] == ((), np.int64) and keys["save_counter/.ATTRIBUTES/VARIABLE_VALUE"] == ((), np.int64)
    assert keys["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] == ((), np.int64)
    for name in ("learning_rate", "beta_1", "beta_2", "decay"):
        assert keys["optimizer/%s/.ATTRIBUTES/VARIABLE_VALUE" % name] == ((), np.float32)
    k = "model/decoder/pre_projection/kernel"
    assert keys[k + "/.ATTRIBUTES/VARIABLE_VALUE"][0] == w["decoder/pre_projection/kernel"].shape
    for slot in ("m", "v"):
        assert keys["%s/.OPTIMIZER_SLOT/optimizer/%s/.ATTRIBUTES/VARIABLE_VALUE" % (k, slot)][0] == w["decoder/pre_projection/kernel"].shape
    assert not any("moving_mean/.OPTIMIZER_SLOT" in x for x in keys)            # BN statistics are not optimised
    back = tc.load_training_checkpoint(prefix, hps)
    assert back["iterations"] == 1234567890123 and back["step"] == 41 and back["save_counter"] == 3
    assert all(np.array_equal(back["weights"][p], w[p]) for p in w)
    assert all(np.array_equal(back["m"][p], m[p]) and np.array_equal(back["v"][p], v[p]) for p in m)
    # the model-only reader (inference.py:122-123 + expect_partial) skips the optimizer entries
    only = tc.load_model_weights(prefix, hps)
    assert set(only) == set(w) and all(np.array_equal(only[p], w[p]) for p in w)
    # object graph: every slot variable hangs off the optimizer node and points at its original variable
    g = tc.parse_object_graph(tc.read_checkpoint(prefix, with_strings=True)[tc.OBJECT_GRAPH_KEY], with_slots=True)
    root_children = dict((name, nid) for nid, name in g[0][0])
    assert {"model", "optimizer", "step", "save_counter"} <= set(root_children)
    slots = g[root_children["optimizer"]][2]
    assert len(slots) == 2 * len(m)
    for orig, slot, nid in slots[:20]:
        ok = g[orig][1][0][2]                       # checkpoint key of the original variable
        assert g[nid][1][0][2] == ok[:-len(tc.SUFFIX)] + "/.OPTIMIZER_SLOT/optimizer/" + slot + tc.SUFFIX


def test_checkpoint_manager_numeric_order_and_max_to_keep(tmp_path):
    """a lexicographic sort resumes from ckpt-9 when ckpt-10..ckpt-89 exist.  The manager follows the
    `checkpoint` state file like tf.train.CheckpointManager.latest_checkpoint, and falls back to the NUMERIC counter."""
    from vaenar_tts_amd import tf_checkpoint as tc
    hps, w, m, v = _tiny_state()
    small = {k: w[k] for k in list(w)[:3]}
    mgr = tc.CheckpointManager(str(tmp_path), max_to_keep=4)
    for i in range(1, 13):
        p = mgr.save(lambda prefix, n: tc.save_training_checkpoint(prefix, small, {}, {}, iterations=i, step=i - 1, save_counter=n))
        assert os.path.basename(p) == "ckpt-%d" % i
    names = sorted(f for f in os.listdir(tmp_path) if f.endswith(".index"))
    assert names == ["ckpt-10.index", "ckpt-11.index", "ckpt-12.index", "ckpt-9.index"]          # 4 kept; note the lexicographic trap
    state = open(tmp_path / "checkpoint").read()
    assert state.startswith('model_checkpoint_path: "ckpt-12"') and state.count("all_model_checkpoint_paths") == 4
    assert os.path.basename(tc.CheckpointManager(str(tmp_path)).latest_checkpoint) == "ckpt-12"
    os.remove(tmp_path / "checkpoint")                                                            # no state file: numeric order
    fresh = tc.CheckpointManager(str(tmp_path))
    assert os.path.basename(fresh.latest_checkpoint) == "ckpt-12" and fresh.next_counter() == 13
    assert tc.load_training_checkpoint(fresh.latest_checkpoint)["step"] == 11
    assert tc.CheckpointManager(str(tmp_path / "empty")).latest_checkpoint is None


def test_bundle_written_atomically_and_truncated_leftovers_are_skipped(tmp_path):
    """bundles are written under a temporary prefix and renamed into place (index last), and a restart that finds
    no state file skips any `ckpt-N` whose table or data shard is not whole instead of dying in restore."""
    from vaenar_tts_amd import tf_checkpoint as tc
    hps, w, m, v = _tiny_state()
    small = {k: w[k] for k in list(w)[:3]}
    mgr = tc.CheckpointManager(str(tmp_path), max_to_keep=4)
    for i in (1, 2):
        mgr.save(lambda prefix, n: tc.save_training_checkpoint(prefix, small, {}, {}, iterations=i, step=i - 1, save_counter=n))
    assert not [f for f in os.listdir(tmp_path) if ".tmp" in f]                  # nothing temporary is left behind
    assert tc.bundle_is_complete(str(tmp_path / "ckpt-2"))
    # a crash during the FIRST save of a run used to leave a truncated ckpt-3.index and no state file
    idx = open(tmp_path / "ckpt-2.index", "rb").read()
    open(tmp_path / "ckpt-3.index", "wb").write(idx[:len(idx) // 2])
    open(tmp_path / "ckpt-3.data-00000-of-00001", "wb").write(b"\0" * 10)
    # ... and a whole index whose data shard is short
    open(tmp_path / "ckpt-4.index", "wb").write(idx)
    open(tmp_path / "ckpt-4.data-00000-of-00001", "wb").write(b"\0" * 10)
    os.remove(tmp_path / "checkpoint")
    assert not tc.bundle_is_complete(str(tmp_path / "ckpt-3")) and not tc.bundle_is_complete(str(tmp_path / "ckpt-4"))
    fresh = tc.CheckpointManager(str(tmp_path))
    assert os.path.basename(fresh.latest_checkpoint) == "ckpt-2"
    assert tc.load_training_checkpoint(fresh.latest_checkpoint)["step"] == 1


def test_restore_refuses_a_bundle_that_does_not_match(tmp_path):
    """restore_checkpoint used strict=False and dropped the missing / mis-shaped lists."""
    import pytest
    import warnings
    from vaenar_tts_amd import tf_checkpoint as tc
    hps, w, m, v = _tiny_state()
    good = str(tmp_path / "ckpt-1")
    tc.save_training_checkpoint(good, w, m, v, iterations=5, step=2, save_counter=1)
    weights, opt = tc.check_training_checkpoint(tc.load_training_checkpoint(good), hps, good)
    assert set(weights) == set(w) and opt is not None and opt[2] == 5 and set(opt[0]) == set(m)
    k0 = "decoder/pre_projection/kernel"
    # (i) a variable missing, (ii) a variable of another shape
    part = str(tmp_path / "ckpt-2")
    tc.save_training_checkpoint(part, {k: a for k, a in w.items() if k != k0}, {}, {}, iterations=5, step=2, save_counter=2)
    with pytest.raises(KeyError, match="missing"):
        tc.check_training_checkpoint(tc.load_training_checkpoint(part), hps, part)
    w2 = dict(w); w2[k0] = np.zeros((3, 5), np.float32)
    other = str(tmp_path / "ckpt-3")
    tc.save_training_checkpoint(other, w2, {}, {}, iterations=5, step=2, save_counter=3)
    with pytest.raises(KeyError, match="mis-shaped"):
        tc.check_training_checkpoint(tc.load_training_checkpoint(other), hps, other)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        weights, opt = tc.check_training_checkpoint(tc.load_training_checkpoint(other), hps, other, strict=False)
    assert k0 not in weights and opt is None and any("mis-shaped" in str(r.message) for r in rec)
    # (iii) only the `m` slots: never half an optimizer
    half = str(tmp_path / "ckpt-4")
    tc.save_training_checkpoint(half, w, m, {}, iterations=5, step=2, save_counter=4)
    with pytest.raises(KeyError, match="incomplete optimizer state"):
        tc.check_training_checkpoint(tc.load_training_checkpoint(half), hps, half)
    # (iv) a model-only bundle restores, and says that Adam restarts
    mo = str(tmp_path / "ckpt-5")
    tc.save_model_weights(mo, w)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        weights, opt = tc.check_training_checkpoint(tc.load_training_checkpoint(mo), hps, mo)
    assert opt is None and set(weights) == set(w) and any("no optimizer slots" in str(r.message) for r in rec)


def test_multi_shard_bundles_and_stale_state_entries(tmp_path):
    """`bundle_is_complete` honours BundleHeaderProto.num_shards / BundleEntryProto.shard_id (TensorFlow writes
    several shards for large models); the `checkpoint` state file's entries are checked like the fallback listing; temporary files
    of a save that died are removed; re-saving a prefix never leaves the old index pointing at new data."""
    import glob
    import struct
    from vaenar_tts_amd import tf_checkpoint as tc
    from vaenar_tts_amd.tf_record_utils import _field, _ld, _varint
    d = str(tmp_path)
    # a hand-made two-shard bundle: header num_shards = 2, one float32 tensor per shard
    a, b = np.arange(6, dtype=np.float32), np.arange(4, dtype=np.float32) + 10
    def entry(arr, shard):
        raw = arr.tobytes()
        shape = _ld(2, _field(1, 0, _varint(arr.shape[0])))
        e = _field(1, 0, _varint(1)) + _ld(2, shape)               # DT_FLOAT
        if shard:
            e += _field(3, 0, _varint(shard))
        return e + _field(5, 0, _varint(len(raw))) + _field(6, 5, struct.pack("<I", tc._mask(tc.crc32c(raw))))
    prefix = os.path.join(d, "two")
    items = [(b"", _field(1, 0, _varint(2)) + _ld(3, _field(1, 0, _varint(1)))), (b"a", entry(a, 0)), (b"b", entry(b, 1))]
    tc._write_table(prefix + ".index", items)
    open(prefix + ".data-00000-of-00002", "wb").write(a.tobytes())
    assert not tc.bundle_is_complete(prefix)                          # second shard missing
    open(prefix + ".data-00001-of-00002", "wb").write(b.tobytes()[:8])
    assert not tc.bundle_is_complete(prefix)                          # second shard truncated
    open(prefix + ".data-00001-of-00002", "wb").write(b.tobytes())
    assert tc.bundle_is_complete(prefix)
    got = tc.read_checkpoint(prefix)
    assert np.array_equal(got["a"], a) and np.array_equal(got["b"], b)
    # state file naming a truncated bundle + a leftover temporary file
    mdir = os.path.join(d, "m"); os.makedirs(mdir)
    tc.write_checkpoint(os.path.join(mdir, "ckpt-1"), {"x": a})
    tc.write_checkpoint(os.path.join(mdir, "ckpt-2"), {"x": b})
    with open(os.path.join(mdir, "ckpt-2.data-00000-of-00001"), "wb") as f:
        f.write(b"\0")                                                # a save that died inside the data file
    import subprocess, sys
    dead = subprocess.Popen([sys.executable, "-c", "pass"]); dead.wait()             # a pid that no longer exists: its temporaries are leftovers
    open(os.path.join(mdir, "ckpt-3.tmp%d.index" % dead.pid), "wb").write(b"junk")
    live = os.path.join(mdir, "ckpt-4.tmp%d.index" % os.getpid())                     # a LIVE writer's file (another rank saving right now) stays
    open(live, "wb").write(b"in flight")
    open(os.path.join(mdir, "checkpoint"), "w").write('model_checkpoint_path: "ckpt-2"\nall_model_checkpoint_paths: "ckpt-1"\nall_model_checkpoint_paths: "ckpt-2"\n')
    m = tc.CheckpointManager(mdir)
    assert m.latest_checkpoint.endswith("ckpt-1") and glob.glob(os.path.join(mdir, "*.tmp*")) == [live]
    os.remove(live)
    # re-saving an existing prefix: whole again afterwards, and the index is replaced only behind the data
    tc.write_checkpoint(os.path.join(mdir, "ckpt-1"), {"x": b})
    assert tc.bundle_is_complete(os.path.join(mdir, "ckpt-1")) and np.array_equal(tc.read_checkpoint(os.path.join(mdir, "ckpt-1"))["x"], b)


# ---- cross-check against a third-party protobuf implementation (google.protobuf; tests/tf_protos.py builds TensorFlow's messages from
#      the published field numbers) -- VERDICT round 4 #6: the hand-rolled codecs were pinned by their own round trips and a few KATs only --------
def _rand_items(n=4, seed=3):
    r = np.random.Generator(np.random.PCG64(seed))
    return [("LJ%03d-%04d" % (i, 7 * i), r.integers(1, 43, 3 + 2 * i).astype(np.int64), r.standard_normal((5 + 3 * i, 80))) for i in range(n)]


def test_example_codec_against_the_protobuf_library():
    """tf.train.Example / tf.io.serialize_tensor both ways: the library's bytes parse with tf_record_utils, tf_record_utils' bytes
    parse with the library; the serialized tensors inside agree byte for byte."""
    from tf_protos import P, example, tensor_proto
    w = tfr.TFRecordWriter()
    for fid, text, mel in _rand_items():
        lib_bytes = example(fid, text, mel).SerializeToString(deterministic=True)
        got = w.parse_example(lib_bytes)                                   # library -> ours
        assert got[0] == fid.encode() and got[3] == len(text) and got[4] == mel.shape[0]
        assert np.array_equal(got[1], text.astype(np.int32)) and np.array_equal(got[2], mel.astype(np.float32))
        ours = tfr.TFRecordWriter.serialize_example(fid, text, mel, len(text), mel.shape[0])
        ex = P["Example"].FromString(ours)                                 # ours -> library
        f = ex.features.feature
        assert sorted(f) == ["fid", "mel", "mel_len", "text", "text_len"]
        assert f["fid"].bytes_list.value[0] == fid.encode() and f["text_len"].int64_list.value[0] == len(text)
        assert f["mel_len"].int64_list.value[0] == mel.shape[0]
        t = P["TensorProto"].FromString(f["mel"].bytes_list.value[0])
        assert t.dtype == 2 and [d.size for d in t.tensor_shape.dim] == list(mel.shape)
        assert np.array_equal(np.frombuffer(t.tensor_content, "<f8").reshape(mel.shape), mel)
        t = P["TensorProto"].FromString(f["text"].bytes_list.value[0])
        assert t.dtype == 9 and np.array_equal(np.frombuffer(t.tensor_content, "<i8"), text)
        # (no byte equality for the Example itself: protobuf does not define the order of map entries -- TensorFlow's C++ writer emits
        #  them in hash order -- which is why both directions are PARSED; the map-free TensorProto below is compared byte for byte)
        assert tfr.serialize_tensor(mel) == tensor_proto(mel).SerializeToString()
    # value-list forms a TensorFlow writer may also produce: int64_val / float_val / double_val instead of tensor_content
    t = P["TensorProto"](); t.dtype = 9; t.tensor_shape.dim.add().size = 3; t.int64_val.extend([5, -2, 1 << 40])
    assert np.array_equal(tfr.parse_tensor(t.SerializeToString()), np.array([5, -2, 1 << 40], np.int64))
    t = P["TensorProto"](); t.dtype = 2; t.tensor_shape.dim.add().size = 2; t.double_val.extend([0.5, -1.25])
    assert np.array_equal(tfr.parse_tensor(t.SerializeToString()), np.array([0.5, -1.25]))


def test_bundle_entry_and_header_against_the_protobuf_library(tmp_path):
    """tensor_bundle.proto: every BundleEntryProto / the BundleHeaderProto our writer puts into the index parse with the library to the
    values written, and entries the library encodes parse with our reader."""
    from tf_protos import P
    from vaenar_tts_amd import tf_checkpoint as ck
    r = np.random.Generator(np.random.PCG64(1))
    tensors = {"model/a/.ATTRIBUTES/VARIABLE_VALUE": r.standard_normal((3, 5)).astype(np.float32),
               "model/b/c/.ATTRIBUTES/VARIABLE_VALUE": r.standard_normal((7,)).astype(np.float32),
               "step/.ATTRIBUTES/VARIABLE_VALUE": np.int64(12)}
    prefix = str(tmp_path / "ckpt-1")
    ck.write_checkpoint(prefix, tensors)
    index = dict(ck.read_index(prefix + ".index"))
    hdr = P["BundleHeaderProto"].FromString(index[b""])
    assert hdr.num_shards == 1 and hdr.endianness == 0 and hdr.version.producer == 1
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    for key, arr in tensors.items():
        e = P["BundleEntryProto"].FromString(index[key.encode()])
        assert e.dtype == {np.dtype("float32"): 1, np.dtype("int64"): 9}[np.asarray(arr).dtype]
        assert [d.size for d in e.shape.dim] == list(np.shape(arr)) and e.shard_id == 0 and e.size == np.asarray(arr).nbytes
        raw = data[e.offset:e.offset + e.size]
        assert np.array_equal(np.frombuffer(raw, np.asarray(arr).dtype).reshape(np.shape(arr)), arr)
        assert e.crc32c == ck._mask(crc32c(raw))
        # library -> ours
        mine = ck._parse_entry(e.SerializeToString())
        assert (mine["dtype"], mine["shape"], mine["shard_id"], mine["offset"], mine["size"], mine["crc32c"]) == \
               (e.dtype, list(np.shape(arr)), 0, e.offset, e.size, e.crc32c)
    # and the whole bundle still reads back through the reader
    back = ck.read_checkpoint(prefix)
    assert all(np.array_equal(back[k], v) for k, v in tensors.items())


def test_object_graph_against_the_protobuf_library():
    """trackable_object_graph.proto: the graph our writer emits parses with the library into the node / child / attribute / slot
    structure tf.train.Checkpoint(step, optimizer, model) expects, and a graph the library encodes parses with our reader."""
    from tf_protos import P
    from vaenar_tts_amd import tf_checkpoint as ck
    paths = ["decoder/pre_projection/kernel", "decoder/pre_projection/bias", "prior/glow/0/1/weight"]
    buf = ck.object_graph_proto(paths, extra=("step", "optimizer/iter"), slots=[(paths[0], "m"), (paths[0], "v")])
    g = P["TrackableObjectGraph"].FromString(buf)
    names = {}

    def walk(nid, prefix):
        for ch in g.nodes[nid].children:
            names[ch.node_id] = prefix + [ch.local_name]
            walk(ch.node_id, prefix + [ch.local_name])
    walk(0, [])
    keys = {a.checkpoint_key for n in g.nodes for a in n.attributes}
    for p in paths:
        assert "model/%s/.ATTRIBUTES/VARIABLE_VALUE" % p in keys
        assert any("/".join(v) == "model/" + p for v in names.values())
    assert "step/.ATTRIBUTES/VARIABLE_VALUE" in keys and "optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE" in keys
    opt = next(i for i, v in names.items() if v == ["optimizer"])
    slots = [(s.original_variable_node_id, s.slot_name, s.slot_variable_node_id) for s in g.nodes[opt].slot_variables]
    assert sorted(s[1] for s in slots) == ["m", "v"]
    for orig, slot, nid in slots:
        assert "/".join(names[orig]) == "model/" + paths[0]
        assert g.nodes[nid].attributes[0].checkpoint_key == "model/%s/.OPTIMIZER_SLOT/optimizer/%s/.ATTRIBUTES/VARIABLE_VALUE" % (paths[0], slot)
    assert all(a.name == "VARIABLE_VALUE" for n in g.nodes for a in n.attributes)
    # library -> ours: re-encode with the library (its own field order / lengths) and read with parse_object_graph
    mine = ck.parse_object_graph(g.SerializeToString(), with_slots=True)
    assert len(mine) == len(g.nodes)
    for (children, attrs, sl), n in zip(mine, g.nodes):
        assert children == [(c.node_id, c.local_name) for c in n.children]
        assert attrs == [(a.name, a.full_name, a.checkpoint_key) for a in n.attributes]
        assert sl == [(s.original_variable_node_id, s.slot_name, s.slot_variable_node_id) for s in n.slot_variables]


def write_tfrecords_with_protobuf_library(save_dir, items_by_mode, train_split=2):
    """{train,dev}-*.tfrecords as the reference's writer lays them out (datasets/tf_record_utils.py:84-94), every Example encoded by the
    protobuf LIBRARY (tests/tf_protos.py); only the record framing (length, masked CRC-32C) is this repository's."""
    from tf_protos import example
    os.makedirs(save_dir, exist_ok=True)
    for mode, items in items_by_mode.items():
        parts = [items[i::train_split] for i in range(train_split)] if mode == "train" else [items]
        for i, part in enumerate(parts):
            tfr.TFRecordWriter.write_records(os.path.join(save_dir, "%s-%d.tfrecords" % (mode, i)),
                                             (example(fid, text, mel).SerializeToString() for fid, text, mel in part))


def test_tfrecords_written_by_the_protobuf_library_feed_create_dataset(tmp_path):
    items = _rand_items(6, seed=9)
    write_tfrecords_with_protobuf_library(str(tmp_path), {"train": items[:4], "dev": items[4:]})
    rec = tfr.TFRecordWriter(save_dir=str(tmp_path))
    got = list(rec.create_dataset(16, 2, 2, 2, 80, 8, False, rec.get_tfrecords_list("train"), seed=1))
    assert sum(len(b[3]) for b in got) == 4
    by_fid = {fid.encode(): (text, mel) for fid, text, mel in items}
    for fids, texts, mels, tl, ml in got:
        for j, fid in enumerate(fids):
            text, mel = by_fid[bytes(fid)]
            assert tl[j] == len(text) and ml[j] == mel.shape[0]
            assert np.array_equal(texts[j, :tl[j]], text.astype(np.int32)) and np.array_equal(mels[j, :ml[j]], mel.astype(np.float32))
