"""TFRecord input of the reference (``/root/reference/datasets/tf_record_utils.py``) without TensorFlow.

Same class and method names (``TFRecordWriter.serialize_example / write / parse_example / create_dataset /
get_tfrecords_list``), same on-disk bytes:

* record framing of ``tf.io.TFRecordWriter`` (tf_record_utils.py:77-83): ``uint64 length | uint32 masked_crc32c(length) |
  data | uint32 masked_crc32c(data)``, little endian, ``masked = ((crc >> 15) | (crc << 17)) + 0xa282ead8``;
* ``tf.train.Example`` protobuf wire format (:35-53): ``Example{1: Features{1: map<string, Feature>}}``,
  ``Feature{1: BytesList{1: bytes*} | 2: FloatList{1: packed float} | 3: Int64List{1: packed varint}}``;
* ``tf.io.serialize_tensor`` payloads (:46-47): ``TensorProto{1: dtype, 2: TensorShapeProto{2: Dim{1: size}*},
  4: tensor_content}`` -- the reference stores ``text`` as int64 and ``mel`` as float64 (:119-120) and casts on read (:124);
* ``create_dataset`` (:126-142): parse -> ``padded_batch`` (zeros up to the longest of the batch) -> shuffle of BATCHES
  through a buffer of ``shuffle_buffer`` batches.  TensorFlow's shuffle order itself cannot be
  reproduced (its RNG); a seeded NumPy generator drives the same buffer algorithm.

The checksums come from the library's host routine ``vnr_crc32c`` (pure C, no GPU).  PARITY UNPINNED against files written
by real TensorFlow (none exist in this environment): the format statements above are the published ones, pinned here by
known-answer vectors (tests/test_tf_formats.py).
"""
import os
import struct

import numpy as np

from ._lib import crc32c

_MASK_DELTA = 0xa282ead8
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_INT64: np.int64}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items()}


def masked_crc32c(data):
    crc = crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


# ---- protobuf wire format (the handful of messages of this path) ---------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]; pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _ld(num, payload):                      # length-delimited field
    return _field(num, 2, _varint(len(payload)) + payload)


def _parse(buf):
    """Yield (field number, wire type, value) of one message; value = int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wire == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wire == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wire)
        yield num, wire, v


def serialize_tensor(array):
    """tf.io.serialize_tensor: TensorProto with dtype, shape and raw little-endian content."""
    a = np.asarray(array, order="C")                   # (ascontiguousarray would promote 0-d to 1-d)
    dt = _DT_OF_NP[a.dtype]
    shape = b"".join(_ld(2, _field(1, 0, _varint(d))) for d in a.shape)
    return _field(1, 0, _varint(dt)) + _ld(2, shape) + _ld(4, a.astype(a.dtype.newbyteorder("<")).tobytes())


def parse_tensor(buf, out_type=None):
    """tf.io.parse_tensor (tensor_content form; also the repeated-value forms for scalars / short tensors)."""
    dt, shape, content, vals = None, [], None, []
    for num, wire, v in _parse(buf):
        if num == 1:
            dt = v
        elif num == 2:
            shape = [next(iter(val for n2, _, val in _parse(dim) if n2 == 1), 0) for n1, _, dim in _parse(v) if n1 == 2]
        elif num == 4:
            content = v
        elif num in (5, 6, 7, 10):            # float_val / double_val / int_val / int64_val (packed or not)
            if wire == 2:
                fmt = {5: "<f", 6: "<d"}.get(num)
                if fmt:
                    vals += [x[0] for x in struct.iter_unpack(fmt, v)]
                else:
                    p = 0
                    while p < len(v):
                        x, p = _read_varint(v, p); vals.append(x - (1 << 64) if x >> 63 else x)
            else:
                vals.append(struct.unpack("<f", struct.pack("<I", v))[0] if num == 5 else
                            struct.unpack("<d", struct.pack("<Q", v))[0] if num == 6 else (v - (1 << 64) if v >> 63 else v))
    np_dt = _NP_OF_DT[dt]
    if out_type is not None and np.dtype(out_type) != np.dtype(np_dt):
        raise ValueError("tensor is %s, expected %s" % (np.dtype(np_dt), np.dtype(out_type)))
    if content is not None:
        return np.frombuffer(content, dtype=np.dtype(np_dt).newbyteorder("<")).reshape(shape).astype(np_dt)
    n = int(np.prod(shape)) if shape else 1
    a = np.asarray(vals if len(vals) != 1 else vals * n, dtype=np_dt)
    return a.reshape(shape)


def _feature(kind, payload):
    return _ld(kind, _ld(1, payload) if kind == 1 else _ld(1, payload))


class TFRecordWriter:
    """Mirror of datasets/tf_record_utils.py:TFRecordWriter (writer AND reader, as in the reference)."""

    def __init__(self, train_split=None, data_dir=None, save_dir=None):
        self.train_split = train_split
        self.data_dir = data_dir
        self.save_dir = save_dir
        self.train_ids_file = os.path.join(self.data_dir, 'train.txt') if data_dir is not None else None
        self.dev_ids_file = os.path.join(self.data_dir, 'dev.txt') if data_dir is not None else None
        self.test_ids_file = os.path.join(self.data_dir, 'test.txt') if data_dir is not None else None
        self.pad_factor = 0

    # -- tf.train.Feature helpers (:17-31) -------------------------------------------------------------------------
    @staticmethod
    def _bytes_feature(value):
        return _ld(1, _ld(1, bytes(value)))                                   # Feature.bytes_list.value[0]

    @staticmethod
    def _float_feature(value):
        return _ld(2, _ld(1, struct.pack("<f", float(value))))                # packed FloatList

    @staticmethod
    def _int64_feature(value):
        return _ld(3, _ld(1, _varint(int(value))))                            # packed Int64List

    @staticmethod
    def serialize_example(fid, text, mel, text_len, mel_len):
        """tf_record_utils.py:33-53 -> bytes of the tf.train.Example (map entries in the reference's key order)."""
        feature = [
            ('fid', TFRecordWriter._bytes_feature(fid.encode('utf-8'))),
            ('text', TFRecordWriter._bytes_feature(serialize_tensor(np.asarray(text)))),
            ('mel', TFRecordWriter._bytes_feature(serialize_tensor(np.asarray(mel)))),
            ('text_len', TFRecordWriter._int64_feature(text_len)),
            ('mel_len', TFRecordWriter._int64_feature(mel_len)),
        ]
        # protobuf serialises map fields in key order in deterministic mode; TF's python API emits sorted keys
        entries = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v)) for k, v in sorted(feature))
        return _ld(1, entries)                                                # Example.features

    # -- record framing --------------------------------------------------------------------------------------------
    @staticmethod
    def write_records(path, payloads):
        with open(path, "wb") as f:
            for data in payloads:
                hdr = struct.pack("<Q", len(data))
                f.write(hdr + struct.pack("<I", masked_crc32c(hdr)) + data + struct.pack("<I", masked_crc32c(data)))

    @staticmethod
    def read_records(path, check=True):
        with open(path, "rb") as f:
            while True:
                hdr = f.read(8)
                if not hdr:
                    return
                if len(hdr) != 8:
                    raise IOError("truncated TFRecord header in %s" % path)
                (n,) = struct.unpack("<Q", hdr)
                (c1,) = struct.unpack("<I", f.read(4))
                data = f.read(n)
                (c2,) = struct.unpack("<I", f.read(4))
                if check and (c1 != masked_crc32c(hdr) or c2 != masked_crc32c(data)):
                    raise IOError("TFRecord checksum mismatch in %s" % path)
                yield data

    # -- dataset files (:55-94) ------------------------------------------------------------------------------------
    def _parse_fids(self, mode='train'):
        fids_f = {'train': self.train_ids_file, 'dev': self.dev_ids_file, 'test': self.test_ids_file}[mode]
        with open(fids_f, 'r', encoding='utf-8') as f:
            return [line.strip() for line in f]

    def _get_features(self, fid):
        text = np.load(os.path.join(self.data_dir, 'texts', '{}.npy'.format(fid)))
        mel = np.load(os.path.join(self.data_dir, 'mels', '{}.npy'.format(fid)))
        return text, mel, len(text), mel.shape[0]

    def write(self, mode='train'):
        fids = self._parse_fids(mode)
        splited = [fids[i::self.train_split] for i in range(self.train_split)] if mode == 'train' else [fids]
        for i, ids in enumerate(splited):
            path = os.path.join(self.save_dir, '{}-{}.tfrecords'.format(mode, i))
            self.write_records(path, (self.serialize_example(fid, *self._get_features(fid)) for fid in ids))

    def write_all(self):
        self.write('train'); self.write('dev'); self.write('test')

    # -- reading (:96-142) -----------------------------------------------------------------------------------------
    def pre_pad(self, inputs):
        n = inputs.shape[0]
        if self.pad_factor in (0, 1) or n % self.pad_factor == 0:
            return inputs
        return np.concatenate([inputs, np.zeros((self.pad_factor - n % self.pad_factor,) + inputs.shape[1:], inputs.dtype)], 0)

    def parse_example(self, serialized_example):
        """-> (fid bytes, text int32 [T], mel float32 [T_mel, num_mels], text_len int32, mel_len int32)  (:108-124)"""
        feats = {}
        for num, _, features in _parse(serialized_example):
            if num != 1:
                continue
            for n1, _, entry in _parse(features):
                if n1 != 1:
                    continue
                key, feat = None, None
                for n2, _, v in _parse(entry):
                    if n2 == 1:
                        key = v.decode()
                    elif n2 == 2:
                        feat = v
                feats[key] = feat

        def bytes_of(feat):
            for kind, _, lst in _parse(feat):
                if kind == 1:
                    return next(v for n, _, v in _parse(lst) if n == 1)
            raise ValueError("not a bytes feature")

        def int_of(feat):
            for kind, _, lst in _parse(feat):
                if kind == 3:
                    for n, wire, v in _parse(lst):
                        if n == 1:
                            return _read_varint(v, 0)[0] if wire == 2 else v
            raise ValueError("not an int64 feature")

        fid = bytes_of(feats['fid'])
        text = parse_tensor(bytes_of(feats['text']), np.int64)
        mel = self.pre_pad(parse_tensor(bytes_of(feats['mel']), np.float64))
        return (fid, text.astype(np.int32), mel.astype(np.float32), np.int32(int_of(feats['text_len'])),
                np.int32(int_of(feats['mel_len'])))

    def create_dataset(self, buffer_size, num_parallel_reads, pad_factor, batch_size, num_mels, shuffle_buffer, shuffle,
                       tfrecord_files, seed=1):
        """Generator of (fids, texts [B,T], mels [B,T_mel,num_mels], text_lens [B], mel_lens [B]) batches (:126-142).
        ``buffer_size`` / ``num_parallel_reads`` are I/O tuning knobs of tf.data and have no effect here."""
        self.pad_factor = pad_factor

        def batches():
            cur = []
            for path in tfrecord_files:
                for rec in self.read_records(path):
                    cur.append(self.parse_example(rec))
                    if len(cur) == batch_size:
                        yield self._pad_batch(cur, num_mels); cur = []
            if cur:
                yield self._pad_batch(cur, num_mels)

        if not shuffle:
            return batches()

        def shuffled():                      # tf.data shuffle: fill a buffer, emit a random element, refill
            rng = np.random.Generator(np.random.PCG64(seed))
            buf = []
            for b in batches():
                buf.append(b)
                if len(buf) > shuffle_buffer:
                    yield buf.pop(int(rng.integers(len(buf))))
            while buf:
                yield buf.pop(int(rng.integers(len(buf))))
        return shuffled()

    @staticmethod
    def _pad_batch(items, num_mels):
        B = len(items)
        Tt = max(len(it[1]) for it in items); Tm = max(it[2].shape[0] for it in items)
        texts = np.zeros((B, Tt), np.int32); mels = np.zeros((B, Tm, num_mels), np.float32)
        for i, (_, t, m, _, _) in enumerate(items):
            texts[i, :len(t)] = t; mels[i, :m.shape[0]] = m
        return ([it[0] for it in items], texts, mels, np.asarray([it[3] for it in items], np.int32),
                np.asarray([it[4] for it in items], np.int32))

    def get_tfrecords_list(self, mode='train'):
        assert self.save_dir is not None
        return sorted(os.path.join(self.save_dir, f) for f in os.listdir(self.save_dir)
                      if f.startswith(mode) and f.endswith('.tfrecords'))
