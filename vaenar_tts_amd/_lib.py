"""ctypes binding of libvaenar_hip.so (C ABI: include/vaenar_hip.h).

There is no CPU fallback: if the shared library is missing, or no AMD GPU is
visible when an engine is created, an exception is raised.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvaenar_hip.so")
ABI_VERSION = 6
VNR_ERR_RANGE = -6

ACT = {"identity": 0, None: 0, "relu": 1, "tanh": 2}


class VnrError(RuntimeError):
    pass


class VnrRangeError(VnrError):
    """VNR_ERR_RANGE: the range sentinel of the split-fp16 path tripped (include/vaenar_hip.h, "Arithmetic contract"): what was computed
    since the previous synchronisation point is invalid and the modules involved now run on exact fp32.  ``Engine`` answers it by
    issuing the pending calls again (``Engine._replay``); user code sees it only when the exact path is non-finite too."""


class vnr_config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("latent_dim", C.c_int32), ("output_dim", C.c_int32), ("max_reduction_factor", C.c_int32),
        ("num_mels", C.c_int32),
        ("enc_vocab_size", C.c_int32), ("enc_embd_dim", C.c_int32), ("enc_n_conv", C.c_int32),
        ("enc_pre_hidden", C.c_int32), ("enc_conv_kernel", C.c_int32),
        ("enc_pre_activation", C.c_int32), ("enc_bn_before_act", C.c_int32), ("enc_n_blk", C.c_int32),
        ("enc_attention_dim", C.c_int32), ("enc_attention_heads", C.c_int32),
        ("enc_ffn_hidden", C.c_int32), ("enc_attention_temperature", C.c_float),
        ("dec_nblk", C.c_int32), ("dec_attention_dim", C.c_int32), ("dec_attention_heads", C.c_int32),
        ("dec_ffn_hidden", C.c_int32), ("dec_post_n_conv", C.c_int32),
        ("dec_post_conv_filters", C.c_int32), ("dec_post_conv_kernel", C.c_int32),
        ("dec_attention_temperature", C.c_float),
        ("prior_n_blk", C.c_int32), ("prior_n_transformer_blk", C.c_int32),
        ("prior_attention_dim", C.c_int32), ("prior_attention_heads", C.c_int32),
        ("prior_ffn_hidden", C.c_int32), ("prior_temperature", C.c_float),
        ("post_pre_hidden", C.c_int32), ("post_pre_activation", C.c_int32), ("post_nblk", C.c_int32),
        ("post_attention_dim", C.c_int32), ("post_attention_heads", C.c_int32),
        ("post_ffn_hidden", C.c_int32), ("post_temperature", C.c_float),
        ("lenpred_activation", C.c_int32),
        ("enc_pre_drop_rate", C.c_float), ("enc_pos_drop_rate", C.c_float), ("dec_post_drop_rate", C.c_float),
        ("post_pre_drop_rate", C.c_float), ("post_pos_drop_rate", C.c_float),
    ]


class vnr_dense_desc(C.Structure):
    _fields_ = [
        ("d_a1", C.c_void_p), ("lda1", C.c_int32), ("k1", C.c_int32),
        ("d_a2", C.c_void_p), ("lda2", C.c_int32), ("k2", C.c_int32),
        ("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("activation", C.c_int32),
        ("d_residual", C.c_void_p), ("ldr", C.c_int32),
        ("d_ln_gamma", C.c_void_p), ("d_ln_beta", C.c_void_p),
        ("d_pe", C.c_void_p), ("pe_T", C.c_int32), ("pe_weight", C.c_float),
        ("d_c", C.c_void_p), ("ldc", C.c_int32), ("m", C.c_int32), ("n", C.c_int32),
    ]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_pd, _pi64 = C.POINTER(C.c_double), C.POINTER(C.c_int64)

# name -> argtypes; every function returns int (vnr_status) unless noted
PROTOTYPES = {
    "vnr_abi_version": [],
    "vnr_device_count": [C.POINTER(C.c_int)],
    "vnr_create": [C.POINTER(vnr_config), _i, C.POINTER(_vp)],
    "vnr_destroy": [_vp],
    "vnr_device_info": [_vp, C.c_char_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "vnr_malloc": [_vp, _sz, C.POINTER(_vp)],
    "vnr_free": [_vp, _vp],
    "vnr_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "vnr_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "vnr_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "vnr_memset": [_vp, _vp, _i, _sz],
    "vnr_synchronize": [_vp],
    "vnr_set_weight": [_vp, C.c_char_p, _vp, _pi64, _i],
    "vnr_get_weight": [_vp, C.c_char_p, _vp, C.c_int64],
    "vnr_finalize_weights": [_vp],
    "vnr_text_encoder_fwd": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "vnr_length_predictor_fwd": [_vp, _vp, _vp, _i, _i, _vp],
    "vnr_prior_sample": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "vnr_decoder_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "vnr_posterior_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "vnr_inference": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "vnr_prior_log_probability": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "vnr_posterior_reparameterize": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "vnr_posterior_log_probability": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    "vnr_prior_init": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "vnr_elbo_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "vnr_op_dense": [_vp, C.POINTER(vnr_dense_desc)],
    "vnr_op_conv1d_bn": [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "vnr_op_attention": [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp],
    "vnr_op_layer_norm": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "vnr_op_kernel_grad": [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "vnr_op_positional_encoding": [_vp, _i, _i, _f, _vp],
    "vnr_random_normal": [_vp, C.c_uint64, C.c_uint64, _f, _vp, _sz],
    "vnr_voc_mel_to_linear": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _i, _f, _vp],
    "vnr_voc_griffin_lim": [_vp, _vp, _vp, C.c_uint64, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "vnr_set_option": [_vp, C.c_char_p, _i],
    "vnr_init": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp],
    "vnr_train_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _f, _f, _f, _f, _f, _f, _i, _vp],
    "vnr_get_gradient": [_vp, C.c_char_p, _vp, C.c_int64],
    "vnr_get_optimizer_slot": [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64],
    "vnr_set_optimizer_slot": [_vp, C.c_char_p, C.c_char_p, _vp, C.c_int64],
    "vnr_get_optimizer_step": [_vp, _pi64],
    "vnr_set_optimizer_step": [_vp, C.c_int64],
    "vnr_comm_unique_id": [_vp, C.c_char_p],
    "vnr_comm_init": [_vp, _i, _i, C.c_char_p],
    "vnr_comm_broadcast_weights": [_vp],
    "vnr_comm_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "vnr_range_info": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int64)],
    "vnr_range_sentinel": [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "vnr_comm_destroy": [_vp],
    "vnr_profile_enable": [_vp, _i],
    "vnr_profile_reset": [_vp],
    "vnr_profile_get": [_vp, C.c_char_p, _pd, _pi64, _pd, _pd],
    "vnr_launch_count": [_vp, _pi64],
}

_lib = None
HW_QUEUES = None        # {value, set_by, hip_possibly_initialised_before} once load() has run


def load():
    """Load libvaenar_hip.so and declare every prototype; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share one serialise.  Every engine handle
    # owns a stream (several handles = several batches in flight) and a training handle three more (kernel gradients, decoder branch,
    # RCCL): with the default, a fourth handle already lands on the queue of another one -- three batches in flight took 2.2 ms per
    # S1 batch instead of 1.6 as soon as a fourth engine existed in the process (profiles/r04_experiments.txt).  Read by the HIP runtime
    # when it initialises, i.e. at the first engine; a value the caller has set wins.
    global HW_QUEUES
    preset = os.environ.get("GPU_MAX_HW_QUEUES")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # The setting only counts if the HIP runtime has not initialised yet.  A profiler's preloaded library (rocprofv3 --pmc ...) or another
    # HIP user in the process initialises it BEFORE this line runs: the collection scripts therefore export the variable themselves
    # (tools/collect_profiles_r06.sh), and this is said out loud instead of letting a profiled run use another queue mapping silently.
    early = [m for m in ("torch",) if m in sys.modules and getattr(sys.modules[m], "cuda", None) is not None
             and sys.modules[m].cuda.is_initialized()]
    if os.environ.get("LD_PRELOAD", "").find("rocprof") >= 0 or os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_LIBRARY_CTOR"):
        early.append("a profiler's preloaded library")
    HW_QUEUES = {"value": os.environ["GPU_MAX_HW_QUEUES"], "set_by": "caller" if preset is not None else "vaenar_tts_amd._lib.load()",
                 "hip_possibly_initialised_before": early}
    if preset is None and early:
        import warnings
        warnings.warn("GPU_MAX_HW_QUEUES was not set in the environment and %s may have initialised HIP already: the engine's streams may share "
                      "hardware queues (export GPU_MAX_HW_QUEUES=8 in front of the command)" % ", ".join(early))
    if not os.path.exists(LIB_PATH):
        raise VnrError(
            "libvaenar_hip.so not found at %s -- build it with `python __graft_entry__.py` or "
            "`make -C vaenar_tts_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.argtypes = args
        fn.restype = C.c_int
    lib.vnr_last_error.argtypes = [_vp]
    lib.vnr_last_error.restype = C.c_char_p
    lib.vnr_crc32c.argtypes = [C.c_uint32, _vp, _sz]
    lib.vnr_crc32c.restype = C.c_uint32
    if lib.vnr_abi_version() != ABI_VERSION:
        raise VnrError("libvaenar_hip.so ABI version %d != binding %d" % (lib.vnr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != 0:
        msg = load().vnr_last_error(handle)
        raise (VnrRangeError if rc == VNR_ERR_RANGE else VnrError)("libvaenar_hip error %d: %s" % (rc, (msg or b"?").decode()))


def config_from_hps(hps):
    e, d, p, q = hps.Encoder.Transformer, hps.Decoder.Transformer, hps.Prior.Transformer, hps.Posterior.Transformer
    c = vnr_config()
    c.abi_version = ABI_VERSION
    c.latent_dim, c.output_dim = hps.Common.latent_dim, hps.Common.output_dim
    c.max_reduction_factor, c.num_mels = hps.Common.max_reduction_factor, hps.Audio.num_mels
    c.enc_vocab_size, c.enc_embd_dim, c.enc_n_conv = e.vocab_size, e.embd_dim, e.n_conv
    c.enc_pre_hidden, c.enc_conv_kernel = e.pre_hidden, e.conv_kernel
    c.enc_pre_activation, c.enc_bn_before_act = ACT[e.pre_activation], int(bool(e.bn_before_act))
    c.enc_n_blk, c.enc_attention_dim, c.enc_attention_heads = e.n_blk, e.attention_dim, e.attention_heads
    c.enc_ffn_hidden, c.enc_attention_temperature = e.ffn_hidden, e.attention_temperature
    c.dec_nblk, c.dec_attention_dim, c.dec_attention_heads = d.nblk, d.attention_dim, d.attention_heads
    c.dec_ffn_hidden, c.dec_post_n_conv = d.ffn_hidden, d.post_n_conv
    c.dec_post_conv_filters, c.dec_post_conv_kernel = d.post_conv_filters, d.post_conv_kernel
    c.dec_attention_temperature = d.attention_temperature
    c.prior_n_blk, c.prior_n_transformer_blk = p.n_blk, p.n_transformer_blk
    c.prior_attention_dim, c.prior_attention_heads = p.attention_dim, p.attention_heads
    c.prior_ffn_hidden, c.prior_temperature = p.ffn_hidden, p.temperature
    c.post_pre_hidden, c.post_pre_activation, c.post_nblk = q.pre_hidden, ACT[q.pre_activation], q.nblk
    c.post_attention_dim, c.post_attention_heads = q.attention_dim, q.attention_heads
    c.post_ffn_hidden, c.post_temperature = q.ffn_hidden, q.temperature
    c.lenpred_activation = ACT[hps.LengthPredictor.Dense.activation]
    c.enc_pre_drop_rate, c.enc_pos_drop_rate = e.pre_drop_rate, e.pos_drop_rate
    c.dec_post_drop_rate = d.post_drop_rate
    c.post_pre_drop_rate, c.post_pos_drop_rate = q.pre_drop_rate, q.pos_drop_rate
    return c


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object -- host routine of the library (no GPU needed)."""
    buf = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return int(load().vnr_crc32c(C.c_uint32(crc), C.c_char_p(bytes(buf)), len(buf)))


def device_count():
    n = C.c_int(0)
    rc = load().vnr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class DeviceArray:
    """A typed view of device memory owned by one engine.  ``.numpy()`` copies to the host
    (the reference's tensors expose the same method, inference.py:156)."""

    __slots__ = ("engine", "ptr", "shape", "dtype", "_owner", "nbytes")

    def __init__(self, engine, shape, dtype=np.float32, ptr=None, owner=None):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if ptr is None:
            self.ptr = engine._take(max(self.nbytes, 4))
            self._owner = True
        else:
            self.ptr = ptr
            self._owner = owner        # keeps the parent allocation alive for views

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            self.engine._checkpoint("vnr_memcpy_d2h", out.ctypes.data, self.ptr, self.nbytes)
        return out

    def copy_from(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.shape == self.shape, (host.shape, self.shape)
        if self.nbytes:
            self.engine._checkpoint("vnr_memcpy_h2d", self.ptr, host.ctypes.data, self.nbytes)
        return self

    def view(self, offset_elems, shape):
        return DeviceArray(self.engine, shape, self.dtype, self.ptr + offset_elems * self.dtype.itemsize,
                           owner=self)

    def __del__(self):
        if getattr(self, "_owner", None) is True and self.engine is not None and self.engine.handle:
            try:
                self.engine._give(max(self.nbytes, 4), self.ptr)
            except Exception:
                pass


class Engine:
    """One HIP engine = one device = one stream (vnr_handle)."""

    def __init__(self, hps, device=0):
        self.lib = load()
        self.hps = hps
        self.handle = None
        h = C.c_void_p()
        cfg = config_from_hps(hps)
        check(self.lib.vnr_create(C.byref(cfg), int(device), C.byref(h)))
        self.handle = h.value
        self.device = device
        self._weights_loaded = set()
        # Freed device buffers are kept per size and handed out again: the steady state of a serving / training loop then
        # issues no hipMalloc / hipFree (hipFree synchronises the device) and sees the same pointers every step.  Reuse is
        # safe because everything runs in order on the engine's single stream.
        self._pool = {}
        self._pool_bytes = 0
        self._pool_cap = 8 << 30
        # Range sentinel (include/vaenar_hip.h, "Arithmetic contract of the split path"): every compute call issued since the last clean
        # synchronisation point, as (entry point, arguments).  A checkpoint that answers VNR_ERR_RANGE has moved the modules involved to
        # exact fp32; the calls are then issued again IN ORDER -- their inputs are untouched (new host data reaches the device only through
        # vnr_memcpy_h2d, which is a checkpoint itself) and every entry point writes its outputs in full, so the device ends in the state
        # the original sequence was meant to produce -- and the checkpoint is repeated.  Only raw pointers are recorded: pooled buffers may
        # be reused meanwhile (in-order replay reproduces the same reuse), but none goes back to the driver while calls are pending.
        self._log = []
        self._log_cap = 256
        self._training_mode = False
        self.range_replays = 0

    # -- calls ------------------------------------------------------------------
    def call(self, name, *args, record=True):
        """One library entry point on this handle.  Compute calls of the inference / evaluation kind are recorded for the range
        sentinel's replay; calls that change variables (training mode, vnr_init, vnr_train_step: ``record=False``) check and repeat
        themselves inside the library.  A call that FAILS with VNR_ERR_RANGE has not run yet (the library looks at the sentinel before
        it starts): the pending calls are replayed and it is issued once more."""
        fn = getattr(self.lib, name)
        try:
            check(fn(self.handle, *args), self.handle)
        except VnrRangeError:
            self._replay()
            check(fn(self.handle, *args), self.handle)
            self._log.clear()          # that call's own checkpoint was clean
        if record and not self._training_mode:
            self._log.append((name, args))
            if len(self._log) >= self._log_cap:
                self.synchronize()

    def _pending(self):
        """True while a compute call waits for its checkpoint (option settings alone do not count)."""
        return any(name != "vnr_set_option" for name, _ in self._log)

    def _replay(self):
        log, self._log = self._log, []
        self.range_replays += 1
        for name, args in log:
            check(getattr(self.lib, name)(self.handle, *args), self.handle)
        self._log = log                 # pending until a checkpoint comes back clean

    def _checkpoint(self, name, *args):
        """vnr_synchronize / vnr_memcpy_d2h / vnr_memcpy_h2d: the synchronisation points at which the library reads the sentinel."""
        fn = getattr(self.lib, name)
        try:
            check(fn(self.handle, *args), self.handle)
        except VnrRangeError:
            self._replay()
            check(fn(self.handle, *args), self.handle)
        self._log.clear()

    def _take(self, nbytes):
        lst = self._pool.get(nbytes)
        if lst:
            self._pool_bytes -= nbytes
            return lst.pop()
        p = C.c_void_p()
        check(self.lib.vnr_malloc(self.handle, nbytes, C.byref(p)), self.handle)
        return p.value

    def _give(self, nbytes, ptr):
        if self._pool_bytes + nbytes <= self._pool_cap or self._pending():      # (nothing goes back to the driver while calls are pending a checkpoint)
            self._pool.setdefault(nbytes, []).append(ptr)
            self._pool_bytes += nbytes
        else:
            self.lib.vnr_free(self.handle, ptr)

    # -- memory -----------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        self.call("vnr_memset", a.ptr, 0, max(a.nbytes, 1))
        return a

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.shape, host.dtype).copy_from(host)

    def asarray(self, x, dtype):
        """Accept a DeviceArray (checked) or anything numpy can convert."""
        if isinstance(x, DeviceArray):
            assert x.dtype == np.dtype(dtype), (x.dtype, dtype)
            return x
        if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
            x = x.numpy()
        return self.to_device(np.asarray(x), dtype)

    def random_normal(self, shape, seed, offset=0, stddev=1.0):
        """tf.random.normal(shape, stddev=stddev) drawn on the device (Philox-4x32-10, vnr_random_normal)."""
        a = DeviceArray(self, shape, np.float32)
        self.call("vnr_random_normal", int(seed) & (2 ** 64 - 1), int(offset), float(stddev), a.ptr, a.size)
        return a

    def synchronize(self):
        self._checkpoint("vnr_synchronize")

    def device_info(self):
        buf = C.create_string_buffer(128)
        cu, wf = C.c_int(0), C.c_int(0)
        check(self.lib.vnr_device_info(self.handle, buf, 128, C.byref(cu), C.byref(wf)), self.handle)
        return dict(name=buf.value.decode(), compute_units=cu.value, wavefront=wf.value)

    # -- weights ----------------------------------------------------------------
    def set_weight(self, path, array):
        if self._pending():
            self.synchronize()          # pending calls belong to the OLD variables: their checkpoint (and replay) comes first
        a = np.ascontiguousarray(array, dtype=np.float32)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        check(self.lib.vnr_set_weight(self.handle, path.encode(), a.ctypes.data, shape, a.ndim), self.handle)
        self._weights_loaded.add(path)

    def has_posterior(self):
        """False for an inference-only weight set (no posterior/ variables were uploaded)."""
        return not self._weights_loaded or any(p.startswith("posterior/") for p in self._weights_loaded)

    def get_weight(self, path, shape):
        out = np.empty(shape, np.float32)
        self._checkpoint("vnr_get_weight", path.encode(), out.ctypes.data, out.size)      # (a device -> host copy: a checkpoint of the range sentinel)
        return out

    # -- data-parallel training (RCCL) ----------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(self.lib.vnr_comm_unique_id(self.handle, buf), self.handle)
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        assert len(unique_id) == 128
        check(self.lib.vnr_comm_init(self.handle, int(nranks), int(rank), C.create_string_buffer(unique_id, 128)), self.handle)

    def comm_broadcast_weights(self):
        check(self.lib.vnr_comm_broadcast_weights(self.handle), self.handle)

    def comm_info(self):
        """(nranks, rank) of the bound communicator as RCCL reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int(0), C.c_int(0)
        check(self.lib.vnr_comm_info(self.handle, C.byref(n), C.byref(r)), self.handle)
        return n.value, r.value

    def comm_destroy(self):
        check(self.lib.vnr_comm_destroy(self.handle), self.handle)

    def range_info(self):
        """The range guard of the split-fp16 path (include/vaenar_hip.h, "Arithmetic contract"): per-module states
        (0 not surveyed, 1 in window = split path, 2 exact fp32 forced), the smallest / largest tensor maximum of the last survey and
        the number of surveys run."""
        st = (C.c_int * 4)()
        lo, hi, n = C.c_float(0), C.c_float(0), C.c_int64(0)
        check(self.lib.vnr_range_info(self.handle, st, C.byref(lo), C.byref(hi), C.byref(n)), self.handle)
        trips, tf32, pend = C.c_int64(0), C.c_int(0), C.c_int(0)
        check(self.lib.vnr_range_sentinel(self.handle, C.byref(trips), C.byref(tf32), C.byref(pend)), self.handle)
        return {"encoder": st[0], "prior": st[1], "decoder": st[2], "posterior": st[3], "lo": lo.value, "hi": hi.value, "surveys": n.value,
                "sentinel_trips": trips.value, "train_fp32": tf32.value, "pending_modules": pend.value, "replays": self.range_replays}

    def get_gradient(self, path, shape):
        out = np.empty(shape, np.float32)
        self._checkpoint("vnr_get_gradient", path.encode(), out.ctypes.data, out.size)
        return out

    # -- optimizer state (train.py:246-255: Checkpoint(step, optimizer, model)) ----------------
    def get_optimizer_slot(self, path, slot, shape):
        out = np.empty(shape, np.float32)
        self._checkpoint("vnr_get_optimizer_slot", path.encode(), slot.encode(), out.ctypes.data, out.size)
        return out

    def set_optimizer_slot(self, path, slot, array):
        a = np.ascontiguousarray(array, dtype=np.float32)
        check(self.lib.vnr_set_optimizer_slot(self.handle, path.encode(), slot.encode(), a.ctypes.data, a.size), self.handle)

    def get_optimizer_step(self):
        n = C.c_int64(0)
        check(self.lib.vnr_get_optimizer_step(self.handle, C.byref(n)), self.handle)
        return n.value

    def set_optimizer_step(self, iterations):
        check(self.lib.vnr_set_optimizer_step(self.handle, int(iterations)), self.handle)

    def load_weights(self, weights):
        for k, v in weights.items():
            self.set_weight(k, v)
        self.finalize()

    def finalize(self):
        check(self.lib.vnr_finalize_weights(self.handle), self.handle)

    def set_option(self, name, value):
        check(self.lib.vnr_set_option(self.handle, name.encode(), int(value)), self.handle)
        if name == "training":
            self._training_mode = bool(int(value))
        elif not self._training_mode and not name.startswith("range_"):
            # options are part of the sequence a replay repeats -- also the ones set in FRONT of the first pending call (VAENAR.call sets
            # n_sample, issues vnr_elbo_fwd and sets it back: a replay that starts at the call would run it with the restored value)
            self._log.append(("vnr_set_option", (name.encode(), int(value))))

    # -- instrumentation ----------------------------------------------------------
    def profile(self, on):
        check(self.lib.vnr_profile_enable(self.handle, int(bool(on))), self.handle)

    def profile_reset(self):
        check(self.lib.vnr_profile_reset(self.handle), self.handle)

    def profile_get(self, cls):
        ms, n, fl, by = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        check(self.lib.vnr_profile_get(self.handle, cls.encode(), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)),
              self.handle)
        return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)

    def launch_count(self):
        n = C.c_int64(0)
        check(self.lib.vnr_launch_count(self.handle, C.byref(n)), self.handle)
        return n.value

    def close(self):
        if self.handle:
            for lst in self._pool.values():
                for ptr in lst:
                    self.lib.vnr_free(self.handle, ptr)
            self._pool.clear(); self._pool_bytes = 0
            self.lib.vnr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
