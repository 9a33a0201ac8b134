/*
 * vaenar_hip.h -- C ABI of libvaenar_hip.so: the MI355X (gfx950) execution engine of the
 * VAENAR-TTS text->mel path.
 *
 * The reference (thuhcsi/VAENAR-TTS) has no native/FFI interface: its operator API for this
 * path is the Python call surface models.VAENAR / modules.* executed by TensorFlow ops.  Each
 * entry point below therefore names the reference *Python* interface it replaces (file:line
 * under /root/reference).  The Python mirror of that surface lives in vaenar_tts_amd/ and
 * binds these symbols with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative vnr_status; the message is available
 *     from vnr_last_error(handle) (or vnr_last_error(NULL) when no handle exists yet);
 *   - no exception crosses the ABI, no torch/TF types appear in a signature;
 *   - pointers named d_* are DEVICE pointers obtained from vnr_malloc (plain hipMalloc memory
 *     on the handle's device); pointers named h_* / host are caller-owned host memory;
 *   - tensors are fp32 row-major [batch, time, channels]; lengths are int32 [batch];
 *   - one handle = one device = one HIP stream; a handle is not thread-safe, different
 *     handles are independent (one per GPU process for data parallelism);
 *   - all module calls are asynchronous on the handle's stream; vnr_memcpy_d2h and
 *     vnr_synchronize are the synchronisation points -- and the checkpoints of the split path's range sentinel: they return
 *     VNR_ERR_RANGE when an activation left the fp16 range since the previous one ("Arithmetic contract of the split path").
 */
#ifndef VAENAR_HIP_H
#define VAENAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNR_ABI_VERSION 6

typedef struct vnr_context *vnr_handle;

typedef enum vnr_status {
  VNR_OK = 0,
  VNR_ERR_ARG = -1,        /* bad argument / unsupported shape */
  VNR_ERR_HIP = -2,        /* a HIP runtime call failed */
  VNR_ERR_WEIGHT = -3,     /* unknown weight path, shape mismatch, or weights not finalized */
  VNR_ERR_NOMEM = -4,
  VNR_ERR_STATE = -5,
  VNR_ERR_RANGE = -6       /* the range sentinel of the split-fp16 path tripped ("Arithmetic contract" below): the results computed since the
                            * previous synchronisation point are invalid, the modules involved have moved to exact fp32 -- issue those calls again */
} vnr_status;

enum { VNR_ACT_IDENTITY = 0, VNR_ACT_RELU = 1, VNR_ACT_TANH = 2 };

/* Hyper-parameters read by the reference constructor models/models.py:10-65
 * (values: configs/hparams.py:233-348).  The engine requires head width 64
 * (attention_dim / heads), odd conv kernels, channel counts divisible by 4 and enc_pre_hidden
 * divisible by 32 (vnr_create says which one it is when a configuration does not fit). */
typedef struct vnr_config {
  int32_t abi_version;                 /* = VNR_ABI_VERSION */
  /* Common (hparams.py:284-289) */
  int32_t latent_dim, output_dim, max_reduction_factor, num_mels;
  /* Encoder.Transformer (hparams.py:291-306) */
  int32_t enc_vocab_size, enc_embd_dim, enc_n_conv, enc_pre_hidden, enc_conv_kernel,
          enc_pre_activation, enc_bn_before_act, enc_n_blk, enc_attention_dim,
          enc_attention_heads, enc_ffn_hidden;
  float   enc_attention_temperature;
  /* Decoder.Transformer (hparams.py:308-321) */
  int32_t dec_nblk, dec_attention_dim, dec_attention_heads, dec_ffn_hidden, dec_post_n_conv,
          dec_post_conv_filters, dec_post_conv_kernel;
  float   dec_attention_temperature;
  /* Prior.Transformer (hparams.py:336-344) */
  int32_t prior_n_blk, prior_n_transformer_blk, prior_attention_dim, prior_attention_heads,
          prior_ffn_hidden;
  float   prior_temperature;
  /* Posterior.Transformer (hparams.py:323-334) */
  int32_t post_pre_hidden, post_pre_activation, post_nblk, post_attention_dim,
          post_attention_heads, post_ffn_hidden;
  float   post_temperature;
  /* LengthPredictor.Dense (hparams.py:346-348) */
  int32_t lenpred_activation;
  /* Dropout rates, active only with option "training" (hparams.py:299-300,321,326-327): encoder prenet convs,
   * encoder positional dropout, PostNet convs, posterior PreNet, posterior positional dropout */
  float   enc_pre_drop_rate, enc_pos_drop_rate, dec_post_drop_rate, post_pre_drop_rate, post_pos_drop_rate;
} vnr_config;

/* ---- lifecycle ------------------------------------------------------------------------ */
/* replaces VAENAR.__init__ (models/models.py:10-65): builds the engine for one device. */
int vnr_create(const vnr_config *cfg, int device, vnr_handle *out);
int vnr_destroy(vnr_handle h);
const char *vnr_last_error(vnr_handle h);
int vnr_abi_version(void);
/* CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 to start): the checksum of TFRecord framing
 * (datasets/tf_record_utils.py:77-83 via tf.io.TFRecordWriter) and of tensor-bundle checkpoints (train.py:246-249).
 * Pure host code, no GPU needed. */
uint32_t vnr_crc32c(uint32_t crc, const void *data, size_t n);
int vnr_device_count(int *count);
/* name, CU count and wavefront size of the handle's device (name buffer >= 64 bytes) */
int vnr_device_info(vnr_handle h, char *name, int name_len, int *compute_units, int *wavefront);

/* ---- device memory (plumbing) ---------------------------------------------------------- */
int vnr_malloc(vnr_handle h, size_t bytes, void **d_ptr);
int vnr_free(vnr_handle h, void *d_ptr);
int vnr_memcpy_h2d(vnr_handle h, void *d_dst, const void *h_src, size_t bytes);
int vnr_memcpy_d2h(vnr_handle h, void *h_dst, const void *d_src, size_t bytes); /* syncs */
int vnr_memcpy_d2d(vnr_handle h, void *d_dst, const void *d_src, size_t bytes);
int vnr_memset(vnr_handle h, void *d_dst, int value, size_t bytes);
int vnr_synchronize(vnr_handle h);

/* ---- weights ---------------------------------------------------------------------------- */
/* replaces tf.train.Checkpoint(model=model).restore (inference.py:122-123): one call per
 * variable of the object-graph tree, path e.g. "decoder/attentions/0/att_proj1/kernel"
 * (vaenar_tts_amd/weights.py lists all paths).  The data is copied. */
int vnr_set_weight(vnr_handle h, const char *path, const float *host, const int64_t *shape,
                   int ndim);
/* (Updating an EXISTING variable with the same shape on a finalized engine -- tf.Variable.assign on one of
 * model.trainable_variables -- needs no vnr_finalize_weights: the packed panels are rebuilt by the next module call.) */
/* reads a variable back (model.trainable_variables / checkpoint save, train.py:246-255) */
int vnr_get_weight(vnr_handle h, const char *path, float *host, int64_t count);
/* packs the weights for the kernels (transposed [out][in] panels, fused QKV / K|V panels,
 * ActNorm o InvertibleLinear folding (flow.py:166-175 + 123-135), BN inference affine
 * (utils.py:76-85)).  Must be called after the last vnr_set_weight and again after any
 * weight update. */
int vnr_finalize_weights(vnr_handle h);

/* ---- modules: the reference call surface -------------------------------------------------- */
/* TransformerEncoder.call (modules/encoder.py:79-93), training=False.
 * d_ids [B,T] int32, d_lengths [B] int32 -> d_out [B,T,enc_pre_hidden]. */
int vnr_text_encoder_fwd(vnr_handle h, const int32_t *d_ids, const int32_t *d_lengths, int B,
                         int T, float pos_step, float *d_out);
/* DenseLengthPredictor.call (modules/length_predictor.py:35-42): d_out [B] float. */
int vnr_length_predictor_fwd(vnr_handle h, const float *d_text_embd, const int32_t *d_lengths,
                             int B, int T, float *d_out);
/* TransformerPrior.sample (modules/prior.py:154-169).  d_eps [B,Tz,latent] is the initial
 * noise already multiplied by the temperature (prior.py:35); NULL = zeros (temperature 0,
 * inference.py:95).  d_z [B,Tz,latent]; d_logprobs [B] may be NULL. */
int vnr_prior_sample(vnr_handle h, const int32_t *d_z_lengths, const float *d_text_embd,
                     const int32_t *d_text_lengths, int B, int Tz, int Tt, const float *d_eps,
                     float *d_z, float *d_logprobs);
/* TransformerDecoder.call (modules/decoder.py:181-199), training=False.
 * d_initial, d_outputs [B, Tz*rf, output_dim]; d_alignments NULL or
 * [dec_nblk][B, heads, Tz, Tt] (the dict decoder-attention-{i}, decoder.py:172,192). */
int vnr_decoder_fwd(vnr_handle h, const float *d_z, const float *d_text_embd,
                    const int32_t *d_z_lengths, const int32_t *d_text_lengths, int B, int Tz,
                    int Tt, int reduction_factor, float *d_initial, float *d_outputs,
                    float *d_alignments);
/* TransformerPosterior.call (modules/posterior.py:115-130), training=False (no dropout).
 * d_mels [B,Tz,num_mels] -> d_mu, d_logvar [B,Tz,latent] in the reference's RETURN order
 * (mu_projection output first). */
int vnr_posterior_fwd(vnr_handle h, const float *d_mels, const float *d_text_embd,
                      const int32_t *d_text_lengths, const int32_t *d_target_lengths, int B,
                      int Tz, int Tt, float *d_mu, float *d_logvar);
/* VAENAR.inference (models/models.py:199-210): encoder -> prior.sample -> decoder on one
 * stream, no host synchronisation.  d_reduced_lengths = ceil(mel_lengths / rf) (models.py:200),
 * Tz = max of them.  d_mel [B, Tz*rf, output_dim]; d_alignments as vnr_decoder_fwd;
 * d_text_embd_out NULL or [B,Tt,enc_pre_hidden]. */
int vnr_inference(vnr_handle h, const int32_t *d_ids, const int32_t *d_text_lengths,
                  const int32_t *d_reduced_lengths, int B, int Tt, int Tz, int reduction_factor,
                  float pos_step, const float *d_eps, float *d_mel, float *d_alignments,
                  float *d_text_embd_out);

/* TransformerPrior.log_probability (modules/prior.py:119-152), training=False: runs the flow backwards
 * (coupling._backward flow.py:241-257, linear._backward :137-150, actnorm._backward :177-187).
 * d_z [B,Tz,latent] -> d_logprobs [B]. */
int vnr_prior_log_probability(vnr_handle h, const float *d_z, const float *d_text_embd,
                              const int32_t *d_z_lengths, const int32_t *d_text_lengths, int B, int Tz,
                              int Tt, float *d_logprobs);
/* BasePosterior.reparameterize (modules/posterior.py:21-39): d_samples [B, nsamples, T, latent] = d_eps * exp(0.5 * logvar) + mu with
 * d_mu, d_logvar [B, T, latent]; d_eps [B, nsamples, T, latent] replaces tf.random.normal (:35; draw it with vnr_random_normal) and
 * NULL is the `random=False` branch (:37, zeros). */
int vnr_posterior_reparameterize(vnr_handle h, const float *d_mu, const float *d_logvar, const float *d_eps, int B,
                                 int nsamples, int T, float *d_samples);
/* BasePosterior.log_probability (modules/posterior.py:42-72): d_logprobs [B, nsamples] = sum_{t < len} -0.5 (latent * log(2 pi) +
 * sum_c (logvar + n^2)); n = d_eps when given (the `eps is not None` branch, :59), else (d_z - mu) / (exp(0.5 logvar) + epsilon)
 * (:60-61).  d_z / d_eps [B, nsamples, T, latent]; d_lengths [B] or NULL = every frame (:66-68). */
int vnr_posterior_log_probability(vnr_handle h, const float *d_mu, const float *d_logvar, const float *d_z, const float *d_eps,
                                  const int32_t *d_lengths, int B, int nsamples, int T, float epsilon, float *d_logprobs);
/* TransformerPrior.init (modules/prior.py:171-186): the flow run forwards like sample(), but every ActNormFlow first sets its
 * log_scale / bias from the statistics of its input (flow.py:189-196).  The variables in the weight store are updated and every
 * packed panel rebuilt.  d_eps [B,Tz,latent] the initial noise (NULL = zeros); d_z [B,Tz,latent]; d_logprobs [B] or NULL. */
int vnr_prior_init(vnr_handle h, const int32_t *d_z_lengths, const float *d_text_embd, const int32_t *d_text_lengths, int B,
                   int Tz, int Tt, const float *d_eps, float *d_z, float *d_logprobs);
/* VAENAR.call (models/models.py:105-197) forward, reduce_loss=False.  Option "n_sample" (default 1) is hps.Train.num_samples
 * (models.py:13): the encoder and the posterior run on the B utterances, then text encoding, targets and lengths are tiled n_sample
 * times (sample index inner, models.py:149-178) and the decoder and prior.log_probability run on B * n_sample rows; d_eps is then
 * [B, n_sample, Tz, latent], d_outs [B * n_sample, Tm, output_dim], d_alignments [dec_nblk][B * n_sample, heads, Tz, Tt], the
 * per-utterance d_l2 / d_kl [B] are means over the samples (models.py:79-83,90), d_aux = predicted lengths [B] | posterior
 * log-probs [B * n_sample] | prior log-probs [B * n_sample].  With option "training" = 1 it is
 * the training-mode forward of train.py:130-134: Dropout active (counter-based masks keyed by option "dropout_seed"),
 * BatchNormalization on batch statistics with the moving statistics updated in the weight store.  Default (training=0):
 * encoder -> length predictor -> posterior -> reparameterize (d_eps [B,Tz,latent] or NULL = zeros) ->
 * posterior log-prob -> decoder -> masked L2 of outputs and initial outputs -> prior.log_probability.
 * Tz = ceil(Tm / rf); d_reduced_lengths = ceil(mel_lengths / rf).
 * Outputs: d_outs [B,Tm,output_dim] (cropped, models.py:183); per-utterance d_l2, d_kl, d_length_l2 [B]
 * (the reduce_loss=True scalars are their means, models.py:84,92,101); d_alignments as vnr_decoder_fwd or NULL;
 * d_aux NULL or [3*B] = predicted lengths | posterior log-probs | prior log-probs. */
int vnr_elbo_fwd(vnr_handle h, const int32_t *d_ids, const int32_t *d_text_lengths,
                 const float *d_mel_targets, const int32_t *d_mel_lengths,
                 const int32_t *d_reduced_lengths, int B, int Tt, int Tm, int reduction_factor,
                 float pos_step, const float *d_eps, float *d_outs, float *d_l2, float *d_kl,
                 float *d_length_l2, float *d_alignments, float *d_aux);

/* ---- single operators (kernel-level parity tests and micro-benchmarks) ---------------------- */
/* tf.keras.layers.Dense on [M,K] (+ optional second input panel = tf.concat on the last axis,
 * attention.py:410-412,440-449): C = epilogue(A1.W[0:K1] + A2.W[K1:K] + bias).
 * d_w is the Keras kernel [K,N] row-major.  Epilogue order: +bias -> activation ->
 * *bn_scale+bn_shift -> +pe_weight*pe[m % pe_T] -> +residual -> LayerNorm(gamma,beta,eps 1e-3).
 * Any optional pointer may be NULL.  Shape limits (checked, VNR_ERR_ARG with a message): k1, k2 and the row strides multiples of 4
 * floats; with a second panel (k2 > 0) k1 a multiple of 32; every operand below 2 GiB. */
typedef struct vnr_dense_desc {
  const float *d_a1; int32_t lda1; int32_t k1;
  const float *d_a2; int32_t lda2; int32_t k2;
  const float *d_w;                       /* [k1+k2, n] */
  const float *d_bias;                    /* [n] */
  int32_t activation;
  const float *d_residual; int32_t ldr;   /* [m, n] */
  const float *d_ln_gamma, *d_ln_beta;    /* [n] */
  const float *d_pe; int32_t pe_T; float pe_weight;   /* [pe_T, n] */
  float *d_c; int32_t ldc;
  int32_t m, n;
} vnr_dense_desc;
int vnr_op_dense(vnr_handle h, const vnr_dense_desc *desc);
/* modules/utils.py Conv1D.call (:76-85) at inference: Conv1D(k, 'same') + bias -> activation
 * -> BatchNormalization(moving stats) (bn_before_act=0) or BN -> activation (=1).
 * d_x [B,T,Cin], d_kernel [k,Cin,Cout] (Keras), bn vectors [Cout] -> d_y [B,T,Cout]. */
int vnr_op_conv1d_bn(vnr_handle h, const float *d_x, int B, int T, int Cin, const float *d_kernel,
                     int k, int Cout, const float *d_bias, int activation, int bn_before_act,
                     const float *d_gamma, const float *d_beta, const float *d_mean,
                     const float *d_var, float *d_y);
/* MultiHeadScaledProductAttention.call core (modules/attention.py:224-246) on projected
 * Q [B,Tq,H*64], K/V [B,Tk,H*64] (row strides ldq/ldk/ldv floats): masked softmax(QK^T /
 * sqrt(64) / temperature) V with key AND query length masks (NULL = full) and optional causal
 * mask; masked logits are filled with -2^32 (attention.py:240).  d_ctx [B,Tq,H*64] (stride ldo);
 * d_alignments NULL or [B,H,Tq,Tk]. */
int vnr_op_attention(vnr_handle h, const float *d_q, int ldq, const float *d_k, int ldk,
                     const float *d_v, int ldv, const int32_t *d_q_lengths,
                     const int32_t *d_k_lengths, int B, int H, int Tq, int Tk, int causal,
                     float temperature, float *d_ctx, int ldo, float *d_alignments);
/* tf.keras.layers.LayerNormalization() over the last axis (eps 1e-3): rows x dim. */
int vnr_op_layer_norm(vnr_handle h, const float *d_x, const float *d_gamma, const float *d_beta,
                      int rows, int dim, float *d_y);
/* The kernel gradient tape.gradient (train.py:136) produces for one Dense kernel or one tap of a Conv1D kernel (modules/utils.py:
 * 33-38, 76-85): d_dw [K,N] = sum over rows m of x[m + shift]^T . dy[m]; a row whose shifted partner falls outside its own
 * utterance (T rows per utterance, T <= 0: one utterance of M rows) contributes nothing ('same' padding).  d_x rows of ldx floats,
 * d_dy rows of lddy floats.  This is the operation behind every weight gradient inside vnr_train_step (split-fp16 MFMA, dy
 * pre-scaled by its maximum); exposed for the op-level parity tests.  Synchronises the stream. */
int vnr_op_kernel_grad(vnr_handle h, const float *d_x, int ldx, const float *d_dy, int lddy, int M, int K,
                       int N, int T, int shift, float *d_dw);
/* tf.random.normal(shape, mean=0, stddev) of BasePrior._initial_sample (modules/prior.py:35, stddev = temperature) and
 * BasePosterior.reparameterize (modules/posterior.py:35): n floats ~ N(0, stddev^2) written on the device by a counter-based
 * Philox-4x32-10 generator + Box-Muller (element block j = elements 4j..4j+3 <- counter j + offset, key = seed), so a
 * temperature > 0 run uploads no noise.  Deterministic in (seed, offset); disjoint offset ranges give independent streams. */
int vnr_random_normal(vnr_handle h, uint64_t seed, uint64_t offset, float stddev, float *d_out, size_t n);
/* PositionalEncoding.positional_encoding (modules/utils.py:333-355): d_out [T,dim]. */
int vnr_op_positional_encoding(vnr_handle h, int T, int dim, float step, float *d_out);

/* ---- vocoder step after the path: reference audio/audio.py ---------------------------------
 * Audio.inv_mel_spectrogram up to the Griffin-Lim input (audio.py:81-84): _denormalize (:206-216) -> + ref_level_db ->
 * _db_to_amp (:189-191) -> _mel_to_linear (:166-174: pinv(mel basis) . mel, floored at 1e-10) -> ** power.
 * d_mel [B,T,n_mels] (the path's mel output layout), d_inv_basis_t [n_mels, n_freq] = pinv(mel basis) transposed (the
 * caller computes it once on the host), d_S [B,T,n_freq]. */
int vnr_voc_mel_to_linear(vnr_handle h, const float *d_mel, const float *d_inv_basis_t, int B, int T,
                          int n_mels, int n_freq, float min_level_db, float ref_level_db,
                          float max_abs_value, int symmetric_specs, float power, float *d_S);
/* Audio._griffin_lim (audio.py:95-102) with librosa 0.8.0 stft / istft semantics (window 'hann' of `win` samples padded
 * to n_fft, center=True / reflect): y = istft(S e^{j phase0}); `iters` times: phase = angle(stft(y)), y = istft(S e^{j
 * phase}).  d_S [B,T,n_fft/2+1] magnitudes; d_init_angles [B,T,n_fft/2+1] radians (the reference draws 2 pi rand(), unseeded)
 * or NULL = drawn on the device from `seed`; d_frames [B] frames per utterance (<= T) or NULL = T;
 * d_wav [B, hop*(T-1)] float (samples past an utterance's own hop*(frames-1) are 0).  n_fft must be 2048. */
int vnr_voc_griffin_lim(vnr_handle h, const float *d_S, const float *d_init_angles, uint64_t seed,
                        const int32_t *d_frames, int B, int T, int n_fft, int hop, int win, int iters,
                        float *d_wav);

/* Engine options.  "split_fp16" (default 1): Dense/Conv GEMMs outside the text encoder evaluate every fp32
 * product as hi*hi + lo*hi + hi*lo on the fp16 matrix pipe (fp32 accumulate; 22 significant bits per operand,
 * measured mel error vs the float64 oracle ~3e-6, same as exact fp32 MFMA); 0 = exact fp32 MFMA everywhere.
 * "split_encoder" (default 1): the text encoder (which feeds the integer frame-count predictor) uses the split path as
 * well; 0 keeps that chain on exact fp32 MFMA.  "op_dense_split" (default 0): vnr_op_dense uses the split kernel.
 * "chain" (default 1): fused row-panel chains of the attention blocks.  "attn_presplit" / "attn_presplit_self" (default 1):
 * the producers of Q, K, V write fp16 hi/lo operand-major image tiles and the attention core loads them directly
 * (cross-attention with T_text <= 128 / every self-attention); 0 = fp32 Q, K, V and the in-kernel split.
 * "op_attn_presplit" (default 0): vnr_op_attention converts its fp32 operands to images and takes that kernel (tests).
 * "late_dec_kv" (default 1): vnr_inference computes the decoder's cross K|V right before the decoder.
 * "prior_inverse" (default 0): Prior.Transformer.inverse = True (/root/reference/modules/prior.py:81-99: every flow of the prior is built
 *    with the flag, and BaseFlow.call / fwd_pass / bwd_pass, flow.py:36-113, swap _forward and _backward).  vnr_prior_sample,
 *    vnr_prior_log_probability, vnr_prior_init, vnr_inference, vnr_elbo_fwd and vnr_init then follow that dispatch (one launch per
 *    operation; neither LJHPS nor DataBakerHPS sets it), and so does vnr_train_step (round 6: log_probability through the _forward passes
 *    of coupling, InvertibleLinear and ActNorm, flow.py:223-239,123-135,166-175, with their gradients).
 * "chain_rows64" (default 0): 64-row panels in the chain kernel -- half the workgroups, half the weight stream per row; slower
 * for one batch alone, faster in aggregate when several handles keep batches in flight on one GPU (bench.py --streams).
 * "gemm_wide_tiles" (default 0): the same trade for the tiled GEMM kernel (64x128 workgroup tiles wherever N >= 128).
 * "split_rows" (default 1): the conv stacks (encoder prenet, PostNet) hand their activations from layer to layer as pre-split
 * fp16 hi|lo rows (same bytes as fp32) so that the consumers' k-loops run without fp32 -> (hi, lo) conversions.
 * "fuse_xattn" (default 1): a CrossAttentionBLK (attention.py:436-452) whose alignments are not requested runs its query
 * projection, the cross-attention over the text and everything after it as ONE chain launch (the workgroup attends for its own
 * 32 rows); 0 = three launches (chain, attention kernel, chain).  Blocks whose alignments are returned are never fused.
 * "attn_bwd_recompute" (default 0): vnr_train_step keeps no attention probabilities -- the forward leaves the softmax row
 * statistics and the backward kernels rebuild P from Q and K (2 x 82 MB less workspace per causal self-attention at B = 32,
 * T = 400); measured slower than the stored form on the T1 step (35.3 vs 32.1 ms), hence off.
 * "train_chain" (default 1): vnr_train_step runs the forward of every CrossAttentionBLK (attention.py:436-452) over the latent
 * frames as two row-panel chain launches that also store every intermediate the backward pass needs (0 = one GEMM / LayerNorm
 * launch per layer; 2 / 3 = 64- / 32-row panels forced, 1 picks 64-row panels from 192 panels up).  "train_chain_bwd" (default 1,
 * needs train_chain): the backward of those blocks between their attention cores -- LayerNorm', dense2', relu', dense1',
 * LayerNorm', att_proj' with the bias / gamma / beta gradients and the abs-max words of the kernel-gradient GEMMs as by-products --
 * as two backward-chain launches per block (0 = one launch per operation).
 * "n_sample" (default 1): hps.Train.num_samples for vnr_elbo_fwd / vnr_train_step (models.py:13,141-178), see vnr_elbo_fwd.
 * "deterministic" (default 0): the reference pins TF_DETERMINISTIC_OPS=1 and every seed (train.py:17-32); with 1 vnr_train_step
 * accumulates every gradient in a fixed order (no float atomics): two identical steps give bit-identical gradients and variables.
 * "chain_waves4" (default 1 since round 5): 32-row chain launches of the inference path on the one-wave-per-SIMD kernel
 * (csrc/gemm3c.hip: 4 waves x 64 columns, 8 k-tiles of weight operands in flight per wave) instead of the 8-wave kernel of
 * csrc/gemm3.hip (0); same programs, the same split products in another accumulation order (S1 mel against the float64 oracle: 3.5e-6 / 3.3e-6).  Programs the 4-wave kernel does not
 * take (irregular k ranges, scratch beyond 160 KB of LDS, the training chains) run on the 8-wave kernel either way.
 * "chain_prefetch" (default 1, with chain_waves4): workgroups on the CUs a chain launch leaves idle walk its weight images a few
 * stages ahead of the workers of their XCD (L2 warming, csrc/chain_prefetch.h).  "chain_segments" (default 1, with chain_waves4):
 * the 32-row panels of a launch with the fused cross-attention start at utterance boundaries (ceil(T / 32) workgroups per
 * utterance) so that no workgroup attends for two utterances.
 * "kv_overlap" (default 0; a round-6 experiment that measured 1 % SLOWER): vnr_inference computes the prior's cross K | V projection on a
 * second stream beside the first flow step's pre-chain and self-attention (which do not read it); the first launch that reads it waits
 * for it.  Same results.
 * "range_guard" (default 1): see "Arithmetic contract of the split path" below; setting it (to either value) forgets the surveys.
 * "range_sentinel" (default 1; 0 = the split products are not watched: measurement only) and "train_fp32" (0 / 1: the training step on
 * exact fp32 MFMA; set by a sentinel trip inside vnr_train_step): same section. */
int vnr_set_option(vnr_handle h, const char *name, int value);

/* Arithmetic contract of the split path (ABI version 5; the sentinel and the range-free exact mode: version 6).
 *
 * TensorFlow evaluates every Dense / Conv1D / matmul of the path in fp32 (/root/reference/modules/attention.py:217-246,
 * modules/utils.py:44-53,76-85).  With "split_fp16" = 1 a product a*w is evaluated here as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo with
 * a = a_hi + a_lo in fp16 and fp32 accumulation.  Weight panels are split once, pre-scaled by a per-panel power of two, so ANY fp32
 * kernel magnitude keeps 22 bits.  Activations are split UNSCALED: the representation a_hi + a_lo carries 22 significant bits while
 * |a| >= 2^-3, an absolute resolution of 2^-25 below that (fp16 subnormals), and becomes inf above 65504.  Stated per tensor: if the
 * largest magnitude of a GEMM input tensor is m, the error of every product row is bounded by 2^-22 (m >= 2^-3) resp.
 * 2^-25 / m (m < 2^-3) RELATIVE TO m times the kernel column's 1-norm -- fp32 round-off class while m lies in [2^-6, 2^15).
 *
 * "range_guard" = 1 enforces that window instead of assuming it.  Activation magnitudes are a property of the WEIGHTS (the inputs of
 * the path are token ids, lengths and unit-variance noise), so the first call of vnr_text_encoder_fwd / vnr_prior_sample /
 * vnr_prior_log_probability / vnr_decoder_fwd / vnr_posterior_fwd / vnr_inference / vnr_elbo_fwd (outside training mode) after
 * the weights of a module changed runs that call on exact fp32 MFMA with an abs-max survey of the input tensor of every Dense /
 * Conv1D and of every attention operand (Q, K, V):
 *   - all maxima in [2^-6, 2^15): the module is marked "in window"; the call is repeated on the split path (so it returns what
 *     every later call returns) and later calls cost nothing extra;
 *   - a Dense / Conv1D input outside: the modules of that call run on exact fp32 MFMA from then on (fp32 dynamic range, about half
 *     the speed) until their weights change;
 *   - an attention operand outside (either side): likewise -- since ABI version 6 the attention cores of the exact mode take per-launch
 *     power-of-two operand scales from the maxima of Q, K and V (csrc/attention2.hip), so the exact mode has NO window: fp32's dynamic
 *     range throughout, like the reference's tf.matmul (round 5 refused such weights with VNR_ERR_STATE).
 * All-zero tensors (the noise at temperature 0) are ignored.  vnr_op_dense with "op_dense_split" applies the same window per ROW of
 * its inputs and falls back to the exact kernel.
 *
 * The survey is a SAMPLE (one call's inputs); the guarantee is the RANGE SENTINEL ("range_sentinel", default 1; ABI version 6).  What can
 * go wrong on the split path after the survey is an activation beyond fp16's 65504 -- a longer text, caller tensors (mels, z, injected
 * noise) of another magnitude, a temperature above 1.  Its split is (+-inf, -+inf); in the consuming product hi*w_hi and lo*w_hi are
 * infinities of opposite sign, so EVERY accumulator of that row is NaN whatever the weights, and an activation function may heal it
 * on the way out (fmaxf(NaN, 0) = 0).  Every split product (csrc/gemm2.hip, gemm3.hip, gemm3c.hip) therefore looks at one accumulator per
 * row right behind its k-loop and raises a host-visible word.  The host reads the word wherever it synchronises with the stream:
 *   - vnr_synchronize, vnr_memcpy_d2h, vnr_memcpy_h2d (before its copy) return VNR_ERR_RANGE when it is raised: NOTHING computed since
 *     the previous such call may be trusted (in particular not the bytes vnr_memcpy_d2h just copied).  The modules that ran on the split
 *     path since then are in state 2 (exact fp32) from that moment; the caller issues the same calls again -- their inputs are
 *     untouched, every entry point writes its outputs in full -- and gets fp32-exact results.  The Python layer does this by itself
 *     (vaenar_tts_amd/_lib.py: Engine._replay): a caller of `.numpy()` never sees the error, only a slower call.
 *   - entry points that change variables cannot be repeated by the caller and check synchronously: training-mode forwards
 *     ("training" = 1), vnr_init and vnr_prior_init repeat themselves once on exact fp32 before they return; vnr_train_step copies the
 *     word behind its backward pass, takes the max over the ranks of the communicator, predicates Adam (and every BatchNormalization
 *     moving update) on it, and when it is raised repeats the step on exact fp32 MFMA with scaled attention cores and fp32
 *     kernel-gradient GEMMs -- the handle's later steps stay there ("train_fp32" = 1; the option resets it).  On that path the products
 *     carry no sentinel (the attention backward runs on the plain fp32 kernels of round 1 there: nothing is split); one pass over the
 *     flat gradient looks for non-finite values instead -- inputs or variables that hold NaN / inf end in VNR_ERR_RANGE with the
 *     variables untouched, never in a silent update.
 * Below the window nothing is detected at run time: a tensor that sinks under 2^-6 on a later call keeps 2^-25 of ABSOLUTE resolution
 * (fp16 subnormals) -- graceful, not wrong by orders of magnitude; set "range_guard" again for a fresh survey.
 *
 * vnr_range_sentinel: trips = checkpoints that found the word raised so far; train_fp32 = 1 once the training step runs on its exact path;
 * pending_modules = bit mask (bit 0 encoder .. 3 posterior) of modules whose split-path results have not passed a checkpoint yet.
 * vnr_range_info: states4[0..3] = encoder, prior, decoder, posterior: 0 not surveyed, 1 in window (split path), 2 exact fp32 forced;
 * lo / hi = smallest / largest non-zero tensor maximum of the last survey; surveys = number of surveys run.  Any pointer may be NULL. */
int vnr_range_info(vnr_handle h, int *states4, float *lo, float *hi, int64_t *surveys);
int vnr_range_sentinel(vnr_handle h, int64_t *trips, int *train_fp32, int *pending_modules);
/* "training" (default 0) and "dropout_seed": the reference's `training=` argument (modules call signatures) for
 * vnr_text_encoder_fwd / vnr_posterior_fwd / vnr_decoder_fwd / vnr_elbo_fwd: Dropout layers (encoder.py:87,
 * utils.py:15,17,84, posterior.py:122) draw counter-based masks; BatchNormalization (utils.py:79-83) normalises with the
 * batch mean / population variance over all B*T rows and updates moving_mean / moving_variance (momentum 0.99). */

/* train_step (train.py:127-138): VAENAR.call(training=True) under the tape, loss = mel_l2 + kl_weight * max(kl, 0) +
 * length_weight * length_l2 (train.py:135; reduce_loss=True means over the batch), gradients w.r.t. every trainable
 * variable (the length predictor sees stop_gradient(text_embd), models.py:133), then tf.keras.optimizers.Adam
 * (train.py:116-117: lr_t = lr sqrt(1-b2^t)/(1-b1^t), w -= lr_t m / (sqrt(v) + epsilon)).  Dropout masks follow option
 * "dropout_seed"; d_eps [B,Tz,latent] ([B,n_sample,Tz,latent] with option "n_sample") is the reparameterisation noise.  apply_update = 0 computes the gradients only
 * (vnr_get_gradient).  h_scalars (HOST, 4 floats or NULL) = mel_l2, kl, length_l2, loss -- the tuple train_step returns.
 * After an update the inference panels are rebuilt lazily by the next inference-mode call.  Synchronises. */
int vnr_train_step(vnr_handle h, const int32_t *d_ids, const int32_t *d_text_lengths,
                   const float *d_mel_targets, const int32_t *d_mel_lengths,
                   const int32_t *d_reduced_lengths, int B, int Tt, int Tm, int rf, float pos_step,
                   const float *d_eps, float kl_weight, float length_weight, float learning_rate,
                   float beta1, float beta2, float epsilon, int apply_update, float *h_scalars);
/* Data-parallel training (the reference trains on one device, its train.py has no distribution
 * strategy to mirror): one process + one handle per GPU, the batch sharded by utterance, ONE exchange per step -- an RCCL
 * all-reduce (sum, then 1/N) of the flat fp32 gradient (34.7 M values) over xGMI between backward and Adam, issued by
 * vnr_train_step on the handle's stream once a communicator is bound.  vnr_comm_unique_id: rank 0 creates the 128-byte
 * id, the host control plane (vaenar_tts_amd/dist.py, gloo) broadcasts it; vnr_comm_init: collective over all ranks;
 * vnr_comm_broadcast_weights: every rank takes rank 0's variables (after vnr_init) and re-packs.  librccl.so is bound
 * with dlopen at the first call, so inference-only processes never load it. */
int vnr_comm_unique_id(vnr_handle h, char *id128);
int vnr_comm_init(vnr_handle h, int nranks, int rank, const char *id128);
int vnr_comm_broadcast_weights(vnr_handle h);
/* rank count and this rank as RCCL itself reports them for the bound communicator (ncclCommCount / ncclCommUserRank) */
int vnr_comm_info(vnr_handle h, int *nranks, int *rank);
int vnr_comm_destroy(vnr_handle h);

/* d loss / d variable of the last vnr_train_step (n floats, layout of the variable) -- tape.gradient (train.py:136). */
int vnr_get_gradient(vnr_handle h, const char *path, float *host, int64_t n);

/* Optimizer state -- the reference checkpoints tf.train.Checkpoint(step, optimizer, model) and restores all three
 * (train.py:246-255): Adam's first / second moment of a trainable variable (slot "m" / "v", layout of the variable; zero
 * until a step has run) and Adam's iteration counter (`optimizer.iterations`, which sets the bias correction of the next step).
 * Restore order: vnr_set_weight (all) -> vnr_finalize_weights -> vnr_set_optimizer_slot / _step; updating an EXISTING
 * variable with vnr_set_weight keeps the optimizer state, adding or resizing one discards it. */
int vnr_get_optimizer_slot(vnr_handle h, const char *path, const char *slot, float *host, int64_t n);
int vnr_set_optimizer_slot(vnr_handle h, const char *path, const char *slot, const float *host, int64_t n);
int vnr_get_optimizer_step(vnr_handle h, int64_t *iterations);
int vnr_set_optimizer_step(vnr_handle h, int64_t iterations);

/* VAENAR.init (models/models.py:212-226) <- train.py:176-179 init_step: text encoder (training=True) ->
 * TransformerPrior.init (prior.py:171-186: every ActNormFlow sets log_scale / bias from the statistics of its input,
 * flow.py:189-196) -> decoder at max_reduction_factor.  d_reduced_lengths = ceil(mel_lengths / max_rf), Tz = their
 * padded maximum, d_eps [B,Tz,latent] the initial noise (NULL = zeros), d_mel NULL or [B, Tz*max_rf, output_dim].
 * The ActNorm variables and BN moving statistics in the weight store are updated and all packed panels rebuilt. */
int vnr_init(vnr_handle h, const int32_t *d_ids, const int32_t *d_text_lengths,
             const int32_t *d_reduced_lengths, int B, int Tt, int Tz, float pos_step,
             const float *d_eps, float *d_mel);


/* ---- instrumentation ------------------------------------------------------------------------ */
/* When enabled every kernel launch is bracketed by HIP events on the handle's stream and
 * accumulated per kernel class ("gemm": tiled GEMM launches on the 3-term split-fp16 path, "gemm_fp32": tiled GEMM launches
 * on exact fp32 MFMA, "chain": row-panel chain launches (split), "attn_self", "attn_cross", "attn_cross_ali", "layer_norm",
 * "misc").  vnr_profile_get synchronises.  flops/bytes are the ALGORITHMIC counts of the
 * launches (2*M*N*K per GEMM; Q+K+V+ctx(+alignments) bytes per attention). */
int vnr_profile_enable(vnr_handle h, int on);
int vnr_profile_reset(vnr_handle h);
int vnr_profile_get(vnr_handle h, const char *kernel_class, double *total_ms, int64_t *launches,
                    double *flops, double *bytes);
/* number of kernel launches issued since creation (all classes) */
int vnr_launch_count(vnr_handle h, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* VAENAR_HIP_H */
