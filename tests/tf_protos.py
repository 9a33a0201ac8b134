"""Test support: the TensorFlow messages behind the reference's on-disk formats, built at import time with the protobuf LIBRARY
(google.protobuf, a third-party encoder / decoder) from the published .proto field numbers -- an implementation of the wire format
that is not this repository's hand-rolled codec (vaenar_tts_amd/tf_record_utils.py, tf_checkpoint.py), so the two can check each other.

  tensorflow/core/example/feature.proto, example.proto      BytesList / FloatList / Int64List / Feature / Features / Example
  tensorflow/core/framework/tensor_shape.proto, tensor.proto, types.proto      TensorShapeProto, TensorProto (DT_FLOAT 1, DT_DOUBLE 2,
                                                                               DT_INT32 3, DT_STRING 7, DT_INT64 9)
  tensorflow/core/framework/versions.proto                  VersionDef
  tensorflow/core/protobuf/tensor_bundle.proto              BundleHeaderProto, BundleEntryProto
  tensorflow/core/protobuf/trackable_object_graph.proto     TrackableObjectGraph

No TensorFlow is installed (and none is needed): the field numbers below ARE the format."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto
_L = {"opt": _T.LABEL_OPTIONAL, "rep": _T.LABEL_REPEATED}
_K = {"int32": _T.TYPE_INT32, "int64": _T.TYPE_INT64, "float": _T.TYPE_FLOAT, "double": _T.TYPE_DOUBLE, "bytes": _T.TYPE_BYTES,
      "string": _T.TYPE_STRING, "bool": _T.TYPE_BOOL, "fixed32": _T.TYPE_FIXED32, "msg": _T.TYPE_MESSAGE}


def _msg(parent, name, fields, oneof=None, map_entry=False):
    m = parent.message_type.add() if isinstance(parent, descriptor_pb2.FileDescriptorProto) else parent.nested_type.add()
    m.name = name
    if oneof:
        m.oneof_decl.add().name = oneof
    for spec in fields:
        fname, num, label, kind = spec[:4]
        f = m.field.add()
        f.name, f.number, f.label, f.type = fname, num, _L[label], _K[kind]
        if kind == "msg":
            f.type_name = spec[4]
        if label == "rep" and kind in ("int32", "int64", "float", "double", "bool"):
            f.options.packed = True                        # proto3 default; TensorFlow's files are proto3
        if oneof and len(spec) > 5 and spec[5] == "oneof":
            f.oneof_index = 0
    if map_entry:
        m.options.map_entry = True
    return m


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "vnr_test_tf_formats.proto", "tensorflow", "proto3"
    _msg(fd, "BytesList", [("value", 1, "rep", "bytes")])
    _msg(fd, "FloatList", [("value", 1, "rep", "float")])
    _msg(fd, "Int64List", [("value", 1, "rep", "int64")])
    _msg(fd, "Feature", [("bytes_list", 1, "opt", "msg", ".tensorflow.BytesList", "oneof"),
                         ("float_list", 2, "opt", "msg", ".tensorflow.FloatList", "oneof"),
                         ("int64_list", 3, "opt", "msg", ".tensorflow.Int64List", "oneof")], oneof="kind")
    feats = _msg(fd, "Features", [("feature", 1, "rep", "msg", ".tensorflow.Features.FeatureEntry")])
    _msg(feats, "FeatureEntry", [("key", 1, "opt", "string"), ("value", 2, "opt", "msg", ".tensorflow.Feature")], map_entry=True)
    _msg(fd, "Example", [("features", 1, "opt", "msg", ".tensorflow.Features")])
    shape = _msg(fd, "TensorShapeProto", [("dim", 2, "rep", "msg", ".tensorflow.TensorShapeProto.Dim"), ("unknown_rank", 3, "opt", "bool")])
    _msg(shape, "Dim", [("size", 1, "opt", "int64"), ("name", 2, "opt", "string")])
    _msg(fd, "TensorProto", [("dtype", 1, "opt", "int32"), ("tensor_shape", 2, "opt", "msg", ".tensorflow.TensorShapeProto"),
                             ("version_number", 3, "opt", "int32"), ("tensor_content", 4, "opt", "bytes"),
                             ("float_val", 5, "rep", "float"), ("double_val", 6, "rep", "double"), ("int_val", 7, "rep", "int32"),
                             ("string_val", 8, "rep", "bytes"), ("int64_val", 10, "rep", "int64"), ("bool_val", 11, "rep", "bool")])
    _msg(fd, "VersionDef", [("producer", 1, "opt", "int32"), ("min_consumer", 2, "opt", "int32"), ("bad_consumers", 3, "rep", "int32")])
    _msg(fd, "BundleHeaderProto", [("num_shards", 1, "opt", "int32"), ("endianness", 2, "opt", "int32"),
                                   ("version", 3, "opt", "msg", ".tensorflow.VersionDef")])
    _msg(fd, "BundleEntryProto", [("dtype", 1, "opt", "int32"), ("shape", 2, "opt", "msg", ".tensorflow.TensorShapeProto"),
                                  ("shard_id", 3, "opt", "int32"), ("offset", 4, "opt", "int64"), ("size", 5, "opt", "int64"),
                                  ("crc32c", 6, "opt", "fixed32")])
    tog = _msg(fd, "TrackableObjectGraph", [("nodes", 1, "rep", "msg", ".tensorflow.TrackableObjectGraph.TrackableObject")])
    obj = _msg(tog, "TrackableObject", [("children", 1, "rep", "msg", ".tensorflow.TrackableObjectGraph.TrackableObject.ObjectReference"),
                                        ("attributes", 2, "rep", "msg", ".tensorflow.TrackableObjectGraph.TrackableObject.SerializedTensor"),
                                        ("slot_variables", 3, "rep", "msg", ".tensorflow.TrackableObjectGraph.TrackableObject.SlotVariableReference")])
    _msg(obj, "ObjectReference", [("node_id", 1, "opt", "int32"), ("local_name", 2, "opt", "string")])
    _msg(obj, "SerializedTensor", [("name", 1, "opt", "string"), ("full_name", 2, "opt", "string"), ("checkpoint_key", 3, "opt", "string"),
                                   ("optional_restore", 4, "opt", "bool")])
    _msg(obj, "SlotVariableReference", [("original_variable_node_id", 1, "opt", "int32"), ("slot_name", 2, "opt", "string"),
                                        ("slot_variable_node_id", 3, "opt", "int32")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorflow." + n))
    return {n: get(n) for n in ("BytesList", "FloatList", "Int64List", "Feature", "Features", "Example", "TensorShapeProto", "TensorProto",
                                "VersionDef", "BundleHeaderProto", "BundleEntryProto", "TrackableObjectGraph")}


P = _build()
DT = {"float32": 1, "float64": 2, "int32": 3, "int64": 9}


def tensor_proto(array):
    """What tf.io.serialize_tensor writes for a numeric ndarray: dtype, shape, raw little-endian tensor_content."""
    import numpy as np
    a = np.ascontiguousarray(array)
    t = P["TensorProto"]()
    t.dtype = DT[a.dtype.name]
    for d in a.shape:
        t.tensor_shape.dim.add().size = int(d)
    t.tensor_content = a.astype(a.dtype.newbyteorder("<")).tobytes()
    return t


def example(fid, text, mel):
    """The Example datasets/tf_record_utils.py:35-53 writes: fid (bytes), text / mel (serialized tensors: int64 / float64), their lengths."""
    ex = P["Example"]()
    f = ex.features.feature
    f["fid"].bytes_list.value.append(fid.encode())
    f["text"].bytes_list.value.append(tensor_proto(text).SerializeToString())
    f["mel"].bytes_list.value.append(tensor_proto(mel).SerializeToString())
    f["text_len"].int64_list.value.append(int(text.shape[0]))
    f["mel_len"].int64_list.value.append(int(mel.shape[0]))
    return ex
