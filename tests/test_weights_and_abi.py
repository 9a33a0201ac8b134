"""Host logic (no GPU): weights contract, synthetic inputs, and the C-ABI surface of libvaenar_hip.so."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import DataBakerHPS, LJHPS, tiny_hps
from vaenar_tts_amd.synthetic import make_batch
from vaenar_tts_amd.weights import count_params, init_weights, load_npz, save_npz, weight_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vaenar_hip.h")


def test_weight_tree_matches_reference_parameter_counts():
    """SURVEY.md section 8 'weights contract': 501 variables, per-module counts."""
    s = weight_spec(LJHPS)
    assert len(s) == 501
    assert count_params(s) == 34731497 and count_params(s, trainable_only=True) == 34725865
    sub = lambda p: sum(int(np.prod(v)) for k, v in s.items() if k.startswith(p))
    assert sub("text_encoder") == 11580929 and sub("prior") == 16165638 and sub("decoder") == 4204000
    assert sub("posterior") == 2780417 and sub("length_predictor") == 513 and sub("prior/glow/0/") == 2694273
    assert count_params(weight_spec(LJHPS, include_posterior=False)) == 31951080
    assert weight_spec(DataBakerHPS)["text_encoder/emb_layer/embeddings"] == (39, 512)


def test_init_is_deterministic_and_roundtrips(tmp_path):
    a, b = init_weights(tiny_hps(), seed=3), init_weights(tiny_hps(), seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    ref = init_weights(tiny_hps(), seed=3, mode="reference")
    assert not ref["prior/glow/0/2/net/log_scale_proj/kernel"].any()          # zero-init heads
    W = ref["prior/glow/2/1/weight"].astype(np.float64)
    np.testing.assert_allclose(W @ W.T, np.eye(len(W)), atol=1e-5)            # qr(randn) orthonormal (flow.py:120)
    save_npz(tmp_path / "w.npz", a)
    c = load_npz(tmp_path / "w.npz")
    assert list(c) == list(a) and all(np.array_equal(a[k], c[k]) for k in a)


def test_synthetic_batch_layout():
    b = make_batch(16, 128, 800, ragged=True)
    assert b["text_lengths"].tolist() == [128 - 4 * i for i in range(16)]
    assert b["mel_lengths"].tolist() == [800 - 24 * i for i in range(16)]
    assert b["eps"].shape == (16, 400, 128) and not b["eps"].any()           # temperature 0 -> zeros
    for i in range(16):
        n = b["text_lengths"][i]
        assert b["ids"][i, 0] == 1 and b["ids"][i, n - 1] == 2 and not b["ids"][i, n:].any()
        assert b["ids"][i, 1:n - 1].min() >= 3 and b["ids"][i, 1:n - 1].max() < 43


def test_library_exports_every_declared_symbol():
    """libvaenar_hip.so loads without a GPU and exports every function include/vaenar_hip.h declares."""
    lib = _lib.load()
    declared = set(re.findall(r"\b(vnr_[a-z0-9_]+)\s*\(", open(HEADER).read()))
    declared -= {"vnr_status", "vnr_config", "vnr_dense_desc", "vnr_context", "vnr_handle"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    extra = {"vnr_last_error", "vnr_crc32c"}          # non-int return types, bound separately in _lib.load
    assert declared == set(_lib.PROTOTYPES) | extra, declared ^ (set(_lib.PROTOTYPES) | extra)
    assert lib.vnr_abi_version() == _lib.ABI_VERSION


def test_ctypes_structs_match_the_header_layout():
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "vaenar_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(vnr_config), offsetof(vnr_config, enc_attention_temperature),
         offsetof(vnr_config, lenpred_activation), sizeof(vnr_dense_desc), offsetof(vnr_dense_desc, d_pe),
         offsetof(vnr_dense_desc, n));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).split()
    got = [int(x) for x in out]
    exp = [ctypes.sizeof(_lib.vnr_config), _lib.vnr_config.enc_attention_temperature.offset,
           _lib.vnr_config.lenpred_activation.offset, ctypes.sizeof(_lib.vnr_dense_desc),
           _lib.vnr_dense_desc.d_pe.offset, _lib.vnr_dense_desc.n.offset]
    assert got == exp


def test_engine_creation_fails_loudly_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.VnrError, match="no HIP device"):
        _lib.Engine(LJHPS, 0)


def test_config_struct_carries_the_hparams():
    c = _lib.config_from_hps(LJHPS)
    assert (c.latent_dim, c.output_dim, c.enc_vocab_size, c.prior_n_blk, c.dec_post_n_conv) == (128, 80, 43, 6, 5)
    assert c.enc_pre_activation == 1 and c.lenpred_activation == 0 and c.enc_bn_before_act == 0


def test_training_noise_schedule_is_restart_safe():
    """train.py: the reparameterisation-noise stream and the dropout seeds are a pure function of (seed, world, rank, iteration) --
    a run restarted from a checkpoint continues the stream instead of replaying it, shards never share noise."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vnr_train_script", os.path.join(ROOT, "train.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    seen = {}
    for world in (1, 2, 8):
        for rank in range(world):
            for it in range(1, 6):
                ns, off, ds = tr.noise_schedule(1234, world, rank, it)
                assert (ns, off, ds) == tr.noise_schedule(1234, world, rank, it)            # no hidden state: a restart at `it` draws the same
                assert off == it * tr.NOISE_STRIDE and off + 32 * 1 * 400 * 128 // 4 <= (it + 1) * tr.NOISE_STRIDE   # a T1 shard's draw fits its range
                seen.setdefault(world, set()).add((ns, off))
                if rank:
                    assert ns != tr.noise_schedule(1234, world, 0, it)[0] and ds != tr.noise_schedule(1234, world, 0, it)[2]
        assert len(seen[world]) == world * 5                                              # every (rank, iteration) has its own counter range
    assert tr.NOISE_STRIDE * 4 >= 32 * 4 * 400 * 128                                      # room for n_sample = 4 at T1 size


def test_build_record_shows_spill_free_chain_kernels():
    """__graft_entry__.build() records the registers / spills hipcc reports for the hot kernels (BUILD_RECORD.json, tracked).  The
    row-panel chain kernels must not spill vector registers: a scratch reload inside a stage epilogue drains the weight prefetch
    stream (33 spilled VGPRs cost 15 % of every chain launch).  Checked when the record belongs to these sources."""
    import json
    import bench
    path = os.path.join(ROOT, "vaenar_tts_amd", "BUILD_RECORD.json")
    rec = json.load(open(path))
    assert rec["abi_version"] == _lib.ABI_VERSION and rec["arch"] == "gfx950"
    if rec["kernel_source_digest"] != bench.kernel_source_digest():
        pytest.skip("BUILD_RECORD.json is from other kernel sources (run __graft_entry__.build())")
    res = rec["kernel_resources"]
    for k in ("panel_chain_kernel<1>", "panel_chain_kernel<2>", "bwd_chain_kernel<1>", "bwd_chain_kernel<2>"):
        assert res[k]["vgpr_spill"] == 0 and res[k]["scratch_bytes_per_lane"] == 0, (k, res[k])
        assert res[k]["waves_per_simd"] >= 2, (k, res[k])
    k = "panel_chain4_kernel"                            # the one-wave-per-SIMD generation (gemm3c.hip): the whole register file, still no spill
    assert res[k]["vgpr_spill"] == 0 and res[k]["scratch_bytes_per_lane"] == 0 and res[k]["waves_per_simd"] == 1, (k, res[k])
