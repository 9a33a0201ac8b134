// libvaenar_hip.so host side: context, weight store and packing, module orchestration on one HIP
// stream, and the extern "C" ABI declared in include/vaenar_hip.h.
#include "../../include/vaenar_hip.h"
#include "common.h"
#include <chrono>

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <memory>
#include <map>
#include <tuple>
#include <utility>
#include <string>
#include <unordered_map>
#include <vector>

using namespace vnr;

namespace {

thread_local std::string g_last_error;

struct Tensor {
  float* d = nullptr;
  std::vector<int64_t> shape;
  int64_t n = 0;
  float scalar = 0.f;   // host copy for rank-0 variables (pos_weight)
};

struct XBlk {   // CrossAttentionBLK (attention.py:418-452), packed
  const float *qkv_wt, *proj1_wt, *proj1_b, *ln1_g, *ln1_b, *q_wt, *proj2_wt, *proj2_b, *ln2_g, *ln2_b,
      *ffn1_wt, *ffn1_b, *ffn2_wt, *ffn2_b, *ffn_g, *ffn_b;
  int kv_col;        // column offset of this block's K|V inside its cross-K/V panel output
  int D, F;
};
struct SBlk {   // SelfAttentionBLK (attention.py:392-415), packed
  const float *qkv_wt, *proj_wt, *proj_b, *ln_g, *ln_b, *ffn1_wt, *ffn1_b, *ffn2_wt, *ffn2_b, *ffn_g, *ffn_b;
};
struct ConvL {  // Conv1D + BN (utils.py:56-85), packed
  const float *wt, *bias, *bn_scale, *bn_shift;
  int cin, cout, k;
  float *gamma, *beta, *moving_mean, *moving_var;   // the BN variables themselves (training mode: batch statistics + moving update)
  float drop_rate; int site;                        // Dropout after BN (utils.py:84), active when training
};
struct FlowStep {
  const float *fold_wt, *fold_b;         // ActNorm o InvertibleLinear
  float *an_log_scale, *an_bias;         // the ActNorm variables themselves (data-dependent init writes them, flow.py:189-196)
  const float* lin_w;                    // InvertibleLinear weight [C][C] as stored
  double lin_logdet;                     // log|det W| cast to fp32 (flow.py:127-129)
  double logdet_per_frame;               // sum(log_scale) + log|det W|  (flow.py:168,127-129)
  const float *inv_wt, *inv_b;           // (InvertibleLinear o ActNorm)^-1 : eps = (z.inv(W) - b) / (exp(ls) + 1e-8)
  double inv_logdet_per_frame;           // -sum(log_scale) + log|det inv(W)|  (flow.py:181,141-145)
  const float *pre_wt, *pre_b;           // transform pre_projection
  const float* pre_wt_x = nullptr;       // the same kernel as a [Dp][C] panel over the WHOLE latent row (zero columns for the half that is not the conditioning
                                         // one): a K = C stage that reads tiles 0 .. C/32 - 1 of the z panel -- what gemm3c.hip's regular k-loop takes
  float pos_weight;
  const float *heads_wt, *heads_b;       // log_scale_proj | shift_proj
  std::vector<XBlk> blks;
};

struct ProfRec { int cls; hipEvent_t e0, e1; double flops, bytes; };
// "gemm": tiled GEMM launches on the split-fp16 path (3 f16 MFMA FLOPs per algorithmic FLOP); "gemm_fp32": tiled GEMM launches on
// exact fp32 MFMA; "chain": panel_chain_kernel launches (always split)
// "chain_ali": chain launches whose fused cross-attention also writes the alignments (the decoder's blocks since round 4)
// training step only: "gemm_tn" kernel-gradient GEMMs (split), "bwd_chain" backward row-panel chains (split), "attn_bwd" attention backward (split)
enum { CLS_GEMM = 0, CLS_ATTN_SELF, CLS_ATTN_CROSS, CLS_ATTN_CROSS_ALI, CLS_LN, CLS_MISC, CLS_GEMM_F32, CLS_CHAIN, CLS_CHAIN_ALI, CLS_GEMM_TN, CLS_BWD_CHAIN,
       CLS_ATTN_BWD, CLS_COUNT };
const char* kClsNames[CLS_COUNT] = {"gemm", "attn_self", "attn_cross", "attn_cross_ali", "layer_norm", "misc", "gemm_fp32", "chain", "chain_ali", "gemm_tn",
                                    "bwd_chain", "attn_bwd"};

// Dropout sites (one mask stream per tf.keras.layers.Dropout instance of the path): encoder.py:70,87; utils.py:73,84
// (prenet / postnet convolutions); utils.py:11-17 (posterior PreNet, two uses of one layer); posterior.py:99,122
enum { SITE_ENC_CONV = 0, SITE_ENC_PE = 8, SITE_POST_PRENET1 = 16, SITE_POST_PRENET2 = 17, SITE_POST_PE = 18, SITE_POSTNET_CONV = 32 };
inline unsigned host_mix32(unsigned k) { k ^= k >> 16; k *= 0x85EBCA6Bu; k ^= k >> 13; k *= 0xC2B2AE35u; k ^= k >> 16; return k; }
// per-site key of the counter-based mask (oracle/vaenar_numpy.py: dropout_site_key is the same statement)
inline unsigned site_key(unsigned seed, int site) { return host_mix32(seed ^ ((unsigned)(site + 1) * 0x9E3779B9u)); }

}  // namespace

namespace { struct TrainState; }

struct vnr_context {
  vnr_config cfg;
  TrainState* train = nullptr;   // optimizer state, gradient buffers, transposed kernels (train.inc); built on the first training step
  bool packed_stale = false;
  bool derived_fresh = false;    // training: the transposed / flipped / split copies of the kernels (train.inc) match the weight store
  // data-parallel training: RCCL communicator over xGMI (one process per GPU), bound at run time with dlopen so that an
  // inference-only process never loads librccl
  void* rccl_lib = nullptr; ncclComm_t comm = nullptr; int comm_size = 1, comm_rank = 0;     // an optimizer step changed the variables: inference panels are rebuilt lazily (check_ready)
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  std::unordered_map<std::string, Tensor> w;
  std::vector<float*> packed_allocs;
  bool finalized = false;

  // packed model
  const float* emb = nullptr;
  float enc_pos_weight = 1.f;
  std::vector<ConvL> enc_convs, post_convs;
  const float *enc_proj_wt = nullptr, *enc_proj_b = nullptr;
  std::vector<SBlk> enc_blks;
  const float *lp_w = nullptr, *lp_b = nullptr;
  std::vector<FlowStep> flow;
  const float* prior_kv_wt = nullptr; int prior_kv_n = 0;     // all prior cross K|V panels
  const float* dec_kv_wt = nullptr; int dec_kv_n = 0;         // contiguous after the prior panels
  const float *dec_pre_wt = nullptr, *dec_pre_b = nullptr, *dec_out_wt = nullptr, *dec_out_b = nullptr,
              *dec_res_wt = nullptr, *dec_res_b = nullptr;
  std::vector<XBlk> dec_blks;
  bool has_posterior = false;
  const float *post_d1_wt = nullptr, *post_d1_b = nullptr, *post_d2_wt = nullptr, *post_d2_b = nullptr,
              *post_mu_wt = nullptr, *post_mu_b = nullptr, *post_lv_wt = nullptr, *post_lv_b = nullptr,
              *post_kv_wt = nullptr;
  int post_kv_n = 0;
  float post_pos_weight = 1.f;
  std::vector<XBlk> post_blks;

  // split-fp16 weight images (gemm2.hip SPLIT path): one per packed fp32 panel, keyed by the panel's base pointer
  struct SplitPanel { int N, K; void* img; float acc_scale; void* opm; };   // opm: operand-major image (gemm3.hip)
  std::map<const float*, SplitPanel> split_panels;
  std::map<std::vector<const void*>, float*> chain_prm;   // packed epilogue parameters of a chain program, keyed by its parameter pointers
  std::vector<void*> split_allocs;
  bool split_enabled = true;     // engine option "split_fp16"
  bool chain_enabled = true;     // engine option "chain": fused row-panel chains (gemm3.hip)
  bool op_attn_presplit = false; // engine option "op_attn_presplit": vnr_op_attention takes the attention3 path (tests / micro-benchmarks)
  bool in_train_step = false;    // set by vnr_train_step around its launches (GemmArgs::no_loader_waves)
  bool gemm_wide_tiles = false;  // engine option "gemm_wide_tiles": 64x128 tiles for every split GEMM with N >= 128 (see chain_rows64)
  bool chain_rows64 = false;     // engine option "chain_rows64": 64-row panels in the chain kernel (half the workgroups, half the weight stream per row)
  bool prior_inverse = false;    // engine option "prior_inverse": Prior.Transformer.inverse = True (prior_inverse_body); inference / evaluation / init only
  bool late_dec_kv = true;       // engine option "late_dec_kv": vnr_inference computes the decoder's cross K|V right before the decoder
  int train_chain_bwd = 1;           // engine option "train_chain_bwd": the backward of the same blocks as two backward-chain launches (gemm3b.hip)
  int train_chain = 1;               // engine option "train_chain" (train.inc: xblk_chain): 0 off, 1 auto, 2 / 3 force 64- / 32-row panels
  bool attn_bwd_recompute = false;   // engine option "attn_bwd_recompute" (train.inc: attn)
  bool chain_segments = true;    // engine option "chain_segments": panels of a fused-attention chain launch start at batch-element boundaries (ChainArgs::seg_T; 4-wave kernel)
  bool chain_prefetch = true;    // engine option "chain_prefetch": prefetch workgroups on the CUs a chain launch leaves idle warm the XCDs' L2 ahead of the workers (chain_prefetch.h)
  unsigned* chain_progress = nullptr; unsigned chain_epoch = 0;   // their pacing words [8 XCDs][16] and the launch counter
  bool chain_waves4 = true;      // engine option "chain_waves4": 32-row chain launches on the one-wave-per-SIMD kernel (gemm3c.hip); 0 = the 8-wave kernel of rounds 1-4
  bool fuse_xattn = true;        // engine option "fuse_xattn": chain B + cross-attention + chain C of a block as ONE launch when no alignments are requested
  bool split_rows = true;        // engine option "split_rows": conv stacks pass their activations as pre-split fp16 hi|lo rows (no conversion in the k-loops)
  bool aoi_self = true;          // engine option "attn_presplit_self": the same for the causal self-attention Q|K|V
  bool aoi_enabled = true;       // engine option "attn_presplit": cross-attention on producer-split operands (attention3.hip)
  // cross K|V panels of the current call that were written as attention operand images (cleared with the workspace)
  struct KvAoi { const float* base; int n; int D; int B; int Tt; AoiDesc d; };
  std::vector<KvAoi> kv_aoi;
  bool split_encoder = true;     // engine option "split_encoder": the text encoder uses the split path too (measured as accurate as exact fp32: profiles/r01_split_accuracy.txt)
  bool op_dense_split = false;   // engine option "op_dense_split" (kernel-level tests of the split path)
  bool training = false;         // engine option "training": Dropout active, BatchNormalization on batch statistics (+ moving update)
  int n_sample = 1;              // engine option "n_sample": hps.Train.num_samples of VAENAR.call (models.py:13,141-178) for vnr_elbo_fwd / vnr_train_step
  bool deterministic = false;    // engine option "deterministic": TF_DETERMINISTIC_OPS=1 of train.py:17-32 -- no float atomics in the training step
  DetState det;                  // its scratch: per-stream partial buffers (common.h)
  unsigned drop_seed = 0;        // engine option "dropout_seed"
  bool split_scope = false;      // set by the module bodies: never inside the encoder -> length predictor chain
  // ---- range guard of the split-fp16 path (include/vaenar_hip.h, "Arithmetic contract") ------------------------------------------
  // The fp16 hi/lo split of an ACTIVATION is unscaled: a tensor whose magnitudes leave [2^-6, 2^15) loses the 22-bit contract
  // (|a| > 65504 would even become inf).  Activation ranges are a property of the weights (the path's inputs are token ids and
  // unit noise), so the FIRST inference-type call of a module after the weights changed runs on exact fp32 MFMA with an abs-max
  // survey of every GEMM input / output; in-window modules then run split (the call is repeated on the split path so that it
  // returns what later calls return), out-of-window modules stay on exact fp32 until the weights change again.
  bool range_guard = true;       // engine option "range_guard"
  int range_state[4] = {0, 0, 0, 0};   // per module (encoder, prior, decoder, posterior): 0 unknown, 1 in window, 2 exact fp32 forced
  bool split_suspended = false;  // exact fp32 for the duration of a guarded call (survey, or a module in state 2)
  bool surveying = false;
  unsigned* survey_words = nullptr; int survey_n = 0;        // device [kSurveyMax][2]: (max, min non-zero) row maximum of each surveyed matrix
  std::vector<int> survey_kind;                              // 0: input of a Dense / Conv1D product, 1: operand of an attention core
  float range_lo = 0.f, range_hi = 0.f;                      // last survey: smallest / largest tensor maximum
  int64_t range_surveys = 0;                                 // surveys run so far (tests)
  // ---- range sentinel (round 6): the guarantee behind the survey's sample -------------------------------------------------------------
  // The survey looks at ONE call's inputs.  A later call can still drive an activation past fp16's 65504 (a longer text, larger caller
  // tensors -- mels, z, injected noise --, a temperature above 1): its split is then (+-inf, -+inf), every accumulator of that row in the
  // consuming product is NaN, and an activation may heal it silently (fmaxf(NaN, 0) = 0).  Every split product (gemm2 / gemm3 / gemm3c)
  // therefore probes one accumulator per row behind its k-loop and raises THIS word (common.h: range_note; host-pinned memory the
  // kernels write through the unified address space).  The host looks at it wherever it synchronises with the stream anyway
  // (vnr_synchronize, vnr_memcpy_d2h / h2d: range_checkpoint): a raised word fails that call with VNR_ERR_RANGE -- nothing computed since
  // the previous checkpoint may be trusted --, moves the modules that ran since then to exact fp32 (state 2: fp32 MFMA products, attention
  // cores with per-launch operand scales, i.e. no window at all) and the caller re-issues its calls (vaenar_tts_amd/_lib.py does that by
  // itself: Engine._replay).  Calls that change variables (training-mode forwards, vnr_init, vnr_train_step) check synchronously and repeat
  // themselves on exact fp32 before they return; their BatchNormalization / Adam updates are predicated on the word.
  unsigned* range_flag = nullptr;      // host-pinned sentinel word
  bool range_sentinel = true;          // engine option "range_sentinel" (0: the products are not watched; measurement only)
  unsigned mods_pending = 0;           // modules that ran on the split path since the last checkpoint
  int64_t range_trips = 0;             // checkpoints that found the word raised
  bool train_fp32 = false;             // the training step runs on exact fp32 MFMA with scaled attention cores (VNR_TRAIN_FP32, or a trip inside a step)
  unsigned* d_step_flag = nullptr;     // device copy of the word the training step's Adam launch is predicated on (all-reduced over the ranks)
  // BatchNormalization moving statistics: saved in front of a call that may repeat itself after a sentinel trip, put back before the repeat
  // (the layers in FRONT of the trip have already taken their update -- predicating the update on the word protects only the ones behind it)
  CopyJob* bn_save_jobs = nullptr; CopyJob* bn_restore_jobs = nullptr; float* bn_backup = nullptr; int bn_njobs = 0; size_t bn_table_gen = (size_t)-1;
  size_t w_generation = 0;             // bumped whenever a variable is (re)allocated
  unsigned* amax_words = nullptr; int amax_next = -1;     // exact mode: the attention cores' abs-max words, 8 per call, cleared by ONE memset per top-level call (-1: not yet)
  // (round 6) vnr_inference: the prior's cross K | V projection (one GEMM of the text encoding, 3072 workgroups, ~63 us) runs on a SECOND
  // stream beside the first flow step's pre-chain and self-attention, which do not read it; the first launch that does (a chain launch
  // with the fused cross-attention, or a cross-attention core) waits for `kv_wait`.  Engine option "kv_overlap" (default 0: measured slower; off while the
  // dispatch-event profile is on: its per-class sums are meant to add up to the wall time).
  bool kv_overlap = false;             // (measured: 2.254 against 2.228 ms per S1 step with the overlap -- the GEMM's workgroups on the CUs the chain launch leaves idle
                                       //  slow the chain down by more than the projection's own time they hide; profiles/r06_experiments.txt r06e)
  hipStream_t kv_stream = nullptr; hipEvent_t kv_fork = nullptr, kv_done = nullptr; hipEvent_t kv_wait = nullptr;
  std::vector<std::pair<const float*, std::pair<int, int>>> panel_registry;   // (base, (N, K)) recorded while packing

  // workspace arena (chunks; bump allocation, reset at every top-level call)
  struct Chunk { char* p; size_t cap, off; };
  std::vector<Chunk> chunks;
  // positional-encoding tables, keyed by (T, dim, step bits)
  std::map<std::tuple<int, int, uint32_t>, float*> pe_cache;
  std::map<int, float*> voc_tables;     // win_length -> [twiddles 2048 floats | Hann window win floats] (vocoder.hip)

  // instrumentation
  bool profiling = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
  int64_t launches = 0;
};

namespace {

int fail(vnr_handle h, int code, const std::string& msg) {
  g_last_error = msg;
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                          \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return fail(h, VNR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
  } while (0)
#define TRY(expr)                      \
  do {                                 \
    int _rc = (expr);                  \
    if (_rc != VNR_OK) return _rc;     \
  } while (0)

// ---- arena --------------------------------------------------------------------------------------
void ws_reset(vnr_handle h) { for (auto& c : h->chunks) c.off = 0; h->kv_aoi.clear(); h->amax_next = -1; }

float* ws_alloc(vnr_handle h, size_t nfloats) {
  size_t bytes = ((nfloats * sizeof(float)) + 255) & ~(size_t)255;
  for (auto& c : h->chunks)
    if (c.off + bytes <= c.cap) { float* p = (float*)(c.p + c.off); c.off += bytes; return p; }
  size_t cap = bytes > ((size_t)64 << 20) ? bytes : ((size_t)64 << 20);
  char* p = nullptr;
  if (hipMalloc((void**)&p, cap) != hipSuccess) { h->err = "workspace hipMalloc failed"; return nullptr; }
  h->chunks.push_back({p, cap, bytes});
  return (float*)p;
}
#define WS(var, n)                                                            \
  float* var = ws_alloc(h, (size_t)(n));                                      \
  if (!var) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed")

// ---- instrumented launches ------------------------------------------------------------------------
hipEvent_t get_event(vnr_handle h) {
  if (!h->event_pool.empty()) { hipEvent_t e = h->event_pool.back(); h->event_pool.pop_back(); return e; }
  hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
  vnr_handle h; ProfRec rec; bool on;
  // the two events ride on the kernel's own dispatch packet (vnr_launch, common.h); a scope whose body launched nothing
  // through vnr_launch records them back to back (zero duration)
  ProfScope(vnr_handle h_, int cls, double flops, double bytes) : h(h_), on(h_->profiling) {
    h->launches++;
    if (on) { rec = {cls, get_event(h), get_event(h), flops, bytes}; g_prof_slot.e0 = rec.e0; g_prof_slot.e1 = rec.e1; g_prof_slot.used = false; g_prof_slot.armed = true; }
  }
  ~ProfScope() {
    if (!on) return;
    if (!g_prof_slot.used) { hipEventRecord(rec.e0, h->stream); hipEventRecord(rec.e1, h->stream); }
    g_prof_slot.armed = false;
    h->prof.push_back(rec);
  }
};

constexpr int kSurveyMax = 2048;
constexpr float kRangeLo = 0.015625f, kRangeHi = 32768.f;      // [2^-6, 2^15)
inline bool split_active(vnr_handle h) { return h->split_enabled && !h->split_suspended; }
// survey: one (max, min non-zero row max) record per matrix
// (T > 0: a batch of B matrices of T rows, bs floats apart -- the operand of one attention call is ONE record whatever the batch; round 5
//  took one per batch element and ran out of records at B = 17, silently: ADVICE round 5)
int survey_matrix(vnr_handle h, const float* x, long long ld, int rows, int cols, int kind, int T = 0, long long bs = 0, int B = 1) {
  if (!x || rows <= 0 || cols <= 0) return VNR_OK;
  if (h->survey_n >= kSurveyMax) return fail(h, VNR_ERR_STATE, "range survey: more than " + std::to_string(kSurveyMax) + " tensors in one call (raise kSurveyMax)");
  const hipError_t e = T > 0 ? launch_row_range_batched(x, ld, T, bs, B, cols, h->survey_words + 2 * h->survey_n, h->stream)
                             : launch_row_range(x, ld, rows, cols, h->survey_words + 2 * h->survey_n, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("range survey: ") + hipGetErrorString(e));
  h->survey_n++;
  h->survey_kind.push_back(kind);
  return VNR_OK;
}

int run_gemm(vnr_handle h, const GemmArgs& g_in) {
  GemmArgs g = g_in;
  if (h->surveying) {
    TRY(survey_matrix(h, g.A1, g.lda1, g.M, g.taps > 0 ? g.conv_C : (g.A2 ? g.K1 : g.K), 0));
    if (g.A2) TRY(survey_matrix(h, g.A2, g.lda2, g.M, g.K - g.K1, 0));
  }
  if (split_active(h) && h->split_scope && g.M >= 64) {
    // the fp32 panel that contains g.Wt (sub-panels start on a row boundary and keep the row length K)
    auto it = h->split_panels.upper_bound(g.Wt);
    if (it != h->split_panels.begin()) {
      --it;
      const auto& sp = it->second;
      const ptrdiff_t off = g.Wt - it->first;
      if (off >= 0 && off < (ptrdiff_t)sp.N * sp.K && off % sp.K == 0 && g.ldw == sp.K && g.K == sp.K &&
          off / sp.K + g.N <= sp.N) {
        g.Wsplit = (const char*)sp.img + (size_t)(off / sp.K) * ((sp.K + 31) / 32) * 128;
        g.acc_scale = sp.acc_scale;
      }
    }
  }
  g.wide_tiles = h->gemm_wide_tiles ? 1 : 0;
  g.no_loader_waves = h->in_train_step ? 1 : 0;
  g.range_flag = (g.Wsplit && h->range_sentinel) ? h->range_flag : nullptr;
  if ((g.a_split || g.c_split) && !g.Wsplit) return fail(h, VNR_ERR_STATE, "split-row activations need the split-fp16 weight image of the layer");
  ProfScope ps(h, g.Wsplit ? CLS_GEMM : CLS_GEMM_F32, 2.0 * g.M * (double)g.N * g.K, 0.0);
  hipError_t e = launch_gemm(g, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("gemm launch: ") + hipGetErrorString(e) +
                                   " (M=" + std::to_string(g.M) + " N=" + std::to_string(g.N) + " K=" + std::to_string(g.K) + ")");
  return VNR_OK;
}
// split image of the fp32 panel rows starting at Wt (row length K): pointer to its first row, k-tiles per row, 2^-s
struct SplitRef { const void* img = nullptr; int kt_total = 0; float scale = 1.f; const char* opm = nullptr; };
bool split_lookup(vnr_handle h, const float* Wt, int K, int N, SplitRef& out) {
  auto it = h->split_panels.upper_bound(Wt);
  if (it == h->split_panels.begin()) return false;
  --it;
  const auto& sp = it->second;
  const ptrdiff_t off = Wt - it->first;
  if (off < 0 || off >= (ptrdiff_t)sp.N * sp.K || off % sp.K != 0 || K != sp.K || off / sp.K + N > sp.N) return false;
  out.kt_total = (sp.K + 31) / 32;
  out.img = (const char*)sp.img + (size_t)(off / sp.K) * out.kt_total * 128;
  out.scale = sp.acc_scale;
  out.opm = ((off / sp.K) % 32 == 0) ? (const char*)sp.opm + (size_t)(off / sp.K / 32) * out.kt_total * 4096 : nullptr;
  return true;
}
struct Tail { const float* wt; int n; const float* bias; float* out; int ldo; int qkv_T = 0, qkv_B = 0; };   // qkv_T > 0: a Q|K|V panel written as operand images (rows per batch element, batch)   // extra Dense(D -> n) on the block output

struct PreStage { const float* wt; int K, N; int src, akt0; const float* bias; const float* pe; int pe_T; float pe_w;
                  float* out; int ldo; int dst; int qkv_T = 0, qkv_B = 0; };
// What follows the log_scale | shift heads of a flow step inside the SAME chain launch (ChainArgs::cpl_stage, gemm3.hip): the affine
// coupling on z (in place) and the next pre-chain -- the next flow step's ActNorm o InvertibleLinear, pre_projection (+PE) and first
// Q|K|V, or the decoder's pre_projection and first Q|K|V.  The coupled z sits in panel 1; `stages` name chain panels (0 .. 2).
struct PostChain { float* z; int ld, zp_off, cond_off; std::vector<PreStage> stages; };

// The chain kernel reads the epilogue parameters (bias | gamma | beta, 256 floats each, per stage) of its whole program
// as ONE contiguous block: it is assembled once per distinct program (device-to-device copies, stream ordered) and cached.
int chain_params(vnr_handle h, ChainArgs& g) {
  std::vector<const void*> key;
  key.reserve(4 * g.nstages);
  for (int i = 0; i < g.nstages; ++i) {
    key.push_back(g.st[i].bias); key.push_back(g.st[i].gamma); key.push_back(g.st[i].beta);
    key.push_back((const void*)(uintptr_t)(unsigned)(g.st[i].n * 4 + g.st[i].acc_mode));
  }
  auto it = h->chain_prm.find(key);
  if (it == h->chain_prm.end()) {
    float* d = nullptr;
    const size_t bytes = (size_t)g.nstages * 768 * sizeof(float);
    HIP_TRY(h, hipMalloc((void**)&d, bytes));
    h->split_allocs.push_back(d);
    HIP_TRY(h, hipMemsetAsync(d, 0, bytes, h->stream));
    for (int i = 0; i < g.nstages; ++i) {
      const ChainStage& st = g.st[i];
      if (st.acc_mode == 1 || st.acc_mode == 2) continue;
      const float* src[3] = {st.bias, st.gamma, st.beta};
      for (int a = 0; a < 3; ++a)
        if (src[a]) HIP_TRY(h, hipMemcpyAsync(d + (size_t)i * 768 + a * 256, src[a], (size_t)st.n * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    }
    it = h->chain_prm.emplace(std::move(key), d).first;
  }
  g.prm = it->second;
  return VNR_OK;
}

// the first reader of the cross K | V projection joins the stream it was computed on (vnr_context::kv_overlap)
int kv_join(vnr_handle h) {
  if (h->kv_wait) { HIP_TRY(h, hipStreamWaitEvent(h->stream, h->kv_wait, 0)); h->kv_wait = nullptr; }
  return VNR_OK;
}
int run_chain(vnr_handle h, ChainArgs& g, double flops) {
  if (g.att_stage > 0) TRY(kv_join(h));
  TRY(chain_params(h, g));
  g.rows64 = h->chain_rows64 ? 1 : 0;
  g.waves4 = (h->chain_waves4 && !h->chain_rows64) ? 1 : 0;
  // (measured: the 4-wave kernel's k-loops wait on first-touch L2 misses, -13 % with the prefetchers; the 8-wave kernel's do not, +-0 %;
  //  64-row panels = several batches in flight: the idle CUs belong to them)
  g.pf_progress = (h->chain_prefetch && g.waves4 && !h->chain_rows64) ? h->chain_progress : nullptr;
  // the pacing words hold launch epoch x 32 + stage and only ever grow (atomic max, never reset): before epoch x 32 wraps the 32-bit word
  // -- 2^27 chain launches, a few hours of back-to-back inference -- the words are cleared in stream order and the count starts over
  // (after the wrap every prefetcher would have seen "a later launch owns the word" and left, for the rest of the process)
  if (h->chain_epoch >= (1u << 27) - 2u) {
    if (h->chain_progress) HIP_TRY(h, hipMemsetAsync(h->chain_progress, 0, 1024, h->stream));
    h->chain_epoch = 0;
  }
  g.pf_epoch = ++h->chain_epoch;
  g.range_flag = (h->range_sentinel && h->range_flag) ? h->range_flag : h->d_step_flag + 4;      // (never null: common.h range_note; an unwatched launch raises a scrap word)
  // bytes of a launch that also writes alignments: the attention core's own traffic as SURVEY D3 counts it (Q + K, V + context +
  // alignments) -- Q and the context never reach HBM here, the figure is what a stand-alone core would move
  const double ali_bytes = (g.att_stage > 0 && g.att_ali)
      ? 4.0 * ((double)g.M * g.D * 2 + (double)g.att_B * g.att_Tk * g.D * 2) + 4.0 * (double)g.M * (g.D / 64) * g.att_Tk : 0.0;
  ProfScope ps(h, ali_bytes > 0 ? CLS_CHAIN_ALI : CLS_CHAIN, flops, ali_bytes);
  hipError_t e = launch_panel_chain(g, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("panel chain launch: ") + hipGetErrorString(e));
  return VNR_OK;
}

int run_attention(vnr_handle h, const AttnArgs& a_in, bool cross) {
  AttnArgs a = a_in;
  if (cross) TRY(kv_join(h));
  if (h->surveying) {                                     // the attention cores of the split path take Q, K and V unscaled
    TRY(survey_matrix(h, a.Q, a.ldq, a.Tq * a.B, a.H * 64, 1, a.Tq, a.q_bs, a.B));
    TRY(survey_matrix(h, a.K, a.ldk, a.Tk * a.B, a.H * 64, 1, a.Tk, a.k_bs, a.B));
    TRY(survey_matrix(h, a.V, a.ldv, a.Tk * a.B, a.H * 64, 1, a.Tk, a.v_bs, a.B));
  }
  if (!split_active(h) || (h->in_train_step && h->train_fp32)) {
    // exact-fp32 mode: the core still multiplies fp16 hi/lo pairs, so its operands get per-launch power-of-two scales from their
    // maxima (attention2.hip: AttnArgs::qkv_absmax) -- fp32's dynamic range, like the reference's tf.matmul (attention.py:224-246)
    constexpr int kAmaxWords = 4096;
    unsigned* words = nullptr;
    if (!h->amax_words && hipMalloc((void**)&h->amax_words, kAmaxWords * sizeof(unsigned)) != hipSuccess) h->amax_words = nullptr;
    if (h->amax_words && h->amax_next < 0) { HIP_TRY(h, hipMemsetAsync(h->amax_words, 0, kAmaxWords * sizeof(unsigned), h->stream)); h->amax_next = 0; }
    if (h->amax_words && h->amax_next + 8 <= kAmaxWords) { words = h->amax_words + h->amax_next; h->amax_next += 8; }
    else {
      words = reinterpret_cast<unsigned*>(ws_alloc(h, 8));
      if (!words) return fail(h, VNR_ERR_NOMEM, "attention operand scales");
      HIP_TRY(h, hipMemsetAsync(words, 0, 8 * sizeof(unsigned), h->stream));
    }
    hipError_t e1 = launch_row_range_batched(a.Q, a.ldq, a.Tq, a.q_bs, a.B, a.H * 64, words, h->stream);
    if (e1 == hipSuccess) e1 = launch_row_range_batched(a.K, a.ldk, a.Tk, a.k_bs, a.B, a.H * 64, words + 2, h->stream);
    if (e1 == hipSuccess) e1 = launch_row_range_batched(a.V, a.ldv, a.Tk, a.v_bs, a.B, a.H * 64, words + 4, h->stream);
    if (e1 != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("attention operand scales: ") + hipGetErrorString(e1));
    h->launches += 3;
    a.qkv_absmax = words;
  }
  if (a.ali && a.Tk > 448 && !a.row_max) {             // two-pass alignment form (attention2.hip): scratch for the row statistics
    const size_t n = (size_t)a.B * a.H * a.Tq;
    a.row_max = ws_alloc(h, n); a.row_linv = ws_alloc(h, n);
    if (!a.row_max || !a.row_linv) return fail(h, VNR_ERR_NOMEM, "attention row statistics");
  }
  const double io = 4.0 * ((double)a.B * a.Tq * a.H * 64 * 2 + (double)a.B * a.Tk * a.H * 64 * 2) +
                    (a.ali ? 4.0 * (double)a.B * a.H * a.Tq * a.Tk : 0.0);
  const double fl = 4.0 * (double)a.B * a.H * a.Tq * (double)a.Tk * 64;
  ProfScope ps(h, cross ? (a.ali ? CLS_ATTN_CROSS_ALI : CLS_ATTN_CROSS) : CLS_ATTN_SELF, fl, io);
  hipError_t e = launch_attention(a, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
  return VNR_OK;
}
int run_attention3(vnr_handle h, const Attn3Args& a, bool cross) {
  if (cross) TRY(kv_join(h));
  const double io = 4.0 * ((double)a.B * a.Tq * a.H * 64 * 2 + (double)a.B * a.Tk * a.H * 64 * 2) +
                    (a.ali ? 4.0 * (double)a.B * a.H * a.Tq * a.Tk : 0.0);
  const double fl = 4.0 * (double)a.B * a.H * a.Tq * (double)a.Tk * 64;
  ProfScope ps(h, cross ? (a.ali ? CLS_ATTN_CROSS_ALI : CLS_ATTN_CROSS) : CLS_ATTN_SELF, fl, io);
  hipError_t e = launch_attention3(a, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("attention3 launch: ") + hipGetErrorString(e));
  return VNR_OK;
}
int run_ln(vnr_handle h, const float* x, const float* g, const float* b, int rows, int dim, float* y) {
  ProfScope ps(h, CLS_LN, 0.0, 8.0 * rows * (double)dim);
  hipError_t e = launch_layer_norm(x, g, b, rows, dim, y, h->stream);
  if (e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string("layer_norm launch: ") + hipGetErrorString(e));
  return VNR_OK;
}
#define RUN_MISC(h, call)                                                                       \
  do {                                                                                          \
    ProfScope _ps(h, CLS_MISC, 0.0, 0.0);                                                       \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess) return fail(h, VNR_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
  } while (0)

// ---- weights --------------------------------------------------------------------------------------
const Tensor* find_w(vnr_handle h, const std::string& path) {
  auto it = h->w.find(path);
  return it == h->w.end() ? nullptr : &it->second;
}

struct Packer {
  vnr_handle h;
  int rc = VNR_OK;
  std::string missing;
  const float* raw(const std::string& path, std::initializer_list<int64_t> shape) {
    const Tensor* t = find_w(h, path);
    if (!t) { if (rc == VNR_OK) { rc = VNR_ERR_WEIGHT; missing = "missing weight " + path; } return nullptr; }
    if (std::vector<int64_t>(shape) != t->shape) {
      if (rc == VNR_OK) { rc = VNR_ERR_WEIGHT; missing = "shape mismatch for " + path; }
      return nullptr;
    }
    return t->d;
  }
  float scalar(const std::string& path) {
    const Tensor* t = find_w(h, path);
    if (!t || t->n != 1) { if (rc == VNR_OK) { rc = VNR_ERR_WEIGHT; missing = "missing scalar " + path; } return 0.f; }
    return t->scalar;
  }
  float* alloc(size_t n) {
    float* p = nullptr;
    if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { if (rc == VNR_OK) { rc = VNR_ERR_NOMEM; missing = "hipMalloc packed"; } return nullptr; }
    h->packed_allocs.push_back(p);
    return p;
  }
  // Keras kernel [K,N] -> Wt [N][K] written at row offset n_off of a panel with row length K
  void transpose_into(const float* src, int K, int N, float* panel, int n_off) {
    if (!src || !panel || rc != VNR_OK) return;
    if (launch_transpose(src, K, N, panel + (size_t)n_off * K, K, h->stream) != hipSuccess) { rc = VNR_ERR_HIP; missing = "transpose launch"; }
  }
  void reg(const float* base, int N, int K) { if (base) h->panel_registry.push_back({base, {N, K}}); }
  const float* wt(const std::string& path, int K, int N) {
    const float* src = raw(path, {K, N});
    float* p = alloc((size_t)K * N);
    transpose_into(src, K, N, p, 0);
    reg(p, N, K);
    return p;
  }
  void copy_into(const float* src, float* dst, size_t n) {
    if (!src || !dst || rc != VNR_OK) return;
    if (hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, h->stream) != hipSuccess) { rc = VNR_ERR_HIP; missing = "d2d copy"; }
  }
};

void pack_xblk(Packer& P, const std::string& p, int D, int mem, int F, float* kv_panel, int kv_row, XBlk& o) {
  o.D = D; o.F = F;
  float* qkv = P.alloc((size_t)3 * D * D);
  P.transpose_into(P.raw(p + "/self_attention/query_layer/kernel", {D, D}), D, D, qkv, 0);
  P.transpose_into(P.raw(p + "/self_attention/key_layer/kernel", {D, D}), D, D, qkv, D);
  P.transpose_into(P.raw(p + "/self_attention/value_layer/kernel", {D, D}), D, D, qkv, 2 * D);
  o.qkv_wt = qkv;
  P.reg(qkv, 3 * D, D);
  o.proj1_wt = P.wt(p + "/att_proj1/kernel", 2 * D, D);
  o.proj1_b = P.raw(p + "/att_proj1/bias", {D});
  o.ln1_g = P.raw(p + "/layer_norm1/gamma", {D});
  o.ln1_b = P.raw(p + "/layer_norm1/beta", {D});
  o.q_wt = P.wt(p + "/cross_attention/query_layer/kernel", D, D);
  P.transpose_into(P.raw(p + "/cross_attention/key_layer/kernel", {mem, D}), mem, D, kv_panel, kv_row);
  P.transpose_into(P.raw(p + "/cross_attention/value_layer/kernel", {mem, D}), mem, D, kv_panel, kv_row + D);
  o.kv_col = kv_row;
  o.proj2_wt = P.wt(p + "/att_proj2/kernel", 2 * D, D);
  o.proj2_b = P.raw(p + "/att_proj2/bias", {D});
  o.ln2_g = P.raw(p + "/layer_norm2/gamma", {D});
  o.ln2_b = P.raw(p + "/layer_norm2/beta", {D});
  o.ffn1_wt = P.wt(p + "/ffn/dense1/kernel", D, F);
  o.ffn1_b = P.raw(p + "/ffn/dense1/bias", {F});
  o.ffn2_wt = P.wt(p + "/ffn/dense2/kernel", F, D);
  o.ffn2_b = P.raw(p + "/ffn/dense2/bias", {D});
  o.ffn_g = P.raw(p + "/ffn/layer_norm/gamma", {D});
  o.ffn_b = P.raw(p + "/ffn/layer_norm/beta", {D});
}

void pack_conv(Packer& P, const std::string& p, int k, int cin, int cout, ConvL& o) {
  o.k = k; o.cin = cin; o.cout = cout;
  const float* src = P.raw(p + "/conv1d/kernel", {k, cin, cout});
  float* wt = P.alloc((size_t)k * cin * cout);
  P.transpose_into(src, k * cin, cout, wt, 0);      // [k*cin, cout] -> [cout][k*cin]
  P.reg(wt, cout, k * cin);
  o.wt = wt;
  o.bias = P.raw(p + "/conv1d/bias", {cout});
  float* sc = P.alloc(cout); float* sh = P.alloc(cout);
  const float* g = P.raw(p + "/bn/gamma", {cout});
  const float* b = P.raw(p + "/bn/beta", {cout});
  const float* m = P.raw(p + "/bn/moving_mean", {cout});
  const float* v = P.raw(p + "/bn/moving_variance", {cout});
  o.gamma = const_cast<float*>(g); o.beta = const_cast<float*>(b); o.moving_mean = const_cast<float*>(m); o.moving_var = const_cast<float*>(v);
  if (P.rc == VNR_OK && launch_bn_affine(g, b, m, v, cout, sc, sh, P.h->stream) != hipSuccess) { P.rc = VNR_ERR_HIP; P.missing = "bn_affine launch"; }
  o.bn_scale = sc; o.bn_shift = sh;
}

// log|det W| of a CxC matrix in float64 (LU with partial pivoting) -- tf.linalg.slogdet(float64(W))[1]
double slogdet_abs(std::vector<double> a, int n) {
  double acc = 0.0;
  for (int c = 0; c < n; ++c) {
    int piv = c; double best = fabs(a[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r) { double v = fabs(a[(size_t)r * n + c]); if (v > best) { best = v; piv = r; } }
    if (best == 0.0) return -INFINITY;
    if (piv != c) for (int k = 0; k < n; ++k) std::swap(a[(size_t)c * n + k], a[(size_t)piv * n + k]);
    const double d = a[(size_t)c * n + c];
    acc += log(fabs(d));
    for (int r = c + 1; r < n; ++r) {
      const double f = a[(size_t)r * n + c] / d;
      if (f != 0.0) for (int k = c + 1; k < n; ++k) a[(size_t)r * n + k] -= f * a[(size_t)c * n + k];
    }
  }
  return acc;
}

// inverse of an n x n matrix in float64 (Gauss-Jordan with partial pivoting); false when singular
bool invert_matrix(std::vector<double> a, int n, std::vector<double>& inv) {
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int piv = c; double best = fabs(a[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r) { const double v = fabs(a[(size_t)r * n + c]); if (v > best) { best = v; piv = r; } }
    if (best == 0.0) return false;
    if (piv != c) for (int k = 0; k < n; ++k) { std::swap(a[(size_t)c * n + k], a[(size_t)piv * n + k]); std::swap(inv[(size_t)c * n + k], inv[(size_t)piv * n + k]); }
    const double d = 1.0 / a[(size_t)c * n + c];
    for (int k = 0; k < n; ++k) { a[(size_t)c * n + k] *= d; inv[(size_t)c * n + k] *= d; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = a[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int k = 0; k < n; ++k) { a[(size_t)r * n + k] -= f * a[(size_t)c * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)c * n + k]; }
    }
  }
  return true;
}

int get_pe(vnr_handle h, int T, int dim, float step, const float** out) {
  uint32_t bits; memcpy(&bits, &step, 4);
  auto key = std::make_tuple(T, dim, bits);
  auto it = h->pe_cache.find(key);
  if (it != h->pe_cache.end()) { *out = it->second; return VNR_OK; }
  float* p = nullptr;
  HIP_TRY(h, hipMalloc((void**)&p, (size_t)T * dim * sizeof(float)));
  RUN_MISC(h, launch_positional_encoding(T, dim, step, p, h->stream));
  h->pe_cache[key] = p;
  *out = p;
  return VNR_OK;
}

// self-attention Q|K|V as attention operand images (three images of img_bytes each instead of the fp32 [M, 3D] panel)
bool self_aoi_on(vnr_handle h, int D, int heads) {
  return h->aoi_enabled && h->aoi_self && split_active(h) && D > 0 && D == heads * 64;      // (the exact mode's cores take fp32 operands and scale them: run_attention)
}
long long aoi_img_bytes(int B, int T, int D) { return (long long)B * (D / 64) * ((T + 31) / 32) * kAoiTile; }
size_t qkv_floats(bool aoi, int B, int T, int D) { return aoi ? (size_t)(3 * aoi_img_bytes(B, T, D) / 4) : (size_t)B * T * 3 * D; }
void set_qkv_aoi(AoiDesc& d, float* base, int B, int T, int D) {
  d.mode = 4; d.D = D; d.T = T; d.TT = (T + 31) / 32; d.blk_bytes = aoi_img_bytes(B, T, D);
  d.qk = reinterpret_cast<char*>(base); d.vt = d.qk + 2 * d.blk_bytes;
}
// the operand images of the cross K|V that run_xblk is about to read at kv + col (null when that panel is plain fp32)
const vnr_context::KvAoi* find_kv_aoi(vnr_handle h, const float* kv, int col, int D, int B, int Tt, int* blk) {
  for (const auto& r : h->kv_aoi) {
    const ptrdiff_t off = (kv + col) - r.base;
    if (off < 0 || off >= r.n || r.D != D || r.B != B || r.Tt != Tt || off % (2 * D)) continue;
    *blk = (int)(off / (2 * D));
    return &r;
  }
  return nullptr;
}


// ---- module bodies ----------------------------------------------------------------------------------
// CrossAttentionBLK.call (attention.py:436-452).  x [M,D] -> out [M,D]; kv = cross K|V panel output
// [B*Tt, kv_ld] of the memory; ali (optional) [B,H,Tq,Tt].
// qkv_buf [M,3D]: the fused self-attention Q|K|V of x; when qkv_ready it was already produced by the previous
// block's chain tail.  tails: extra Dense(D -> n) layers applied to the block OUTPUT inside the same launch (the next
// block's Q|K|V, the flow heads, the decoder out-projection, the posterior heads).
int run_xblk(vnr_handle h, const XBlk& k, const float* x, float* out, const float* kv, int kv_ld,
             const int32_t* q_len, const int32_t* m_len, int B, int Tq, int Tt, int heads, float tau,
             float* ali, float* qkv, bool qkv_ready, const std::vector<Tail>& tails, bool self_aoi,
             const PostChain* post = nullptr, bool* post_done = nullptr) {
  if (post_done) *post_done = false;
  const int M = B * Tq, D = k.D, F = k.F;
  // cross K|V written as attention operand images (run_kv): the query is produced as an image too and attention3 runs
  int kv_blk = 0;
  const vnr_context::KvAoi* ka = find_kv_aoi(h, kv, k.kv_col, D, B, Tt, &kv_blk);
  WS(sa, (size_t)M * D); WS(y, (size_t)M * D); WS(q, (size_t)B * ((Tq + 31) / 32 * 32) * D); WS(ca, (size_t)M * D);
  GemmArgs g;
  // self attention: fused Q|K|V projection (no bias, attention.py:154-159)
  if (!qkv_ready) {
    g = GemmArgs(); g.A1 = x; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = k.qkv_wt; g.ldw = D; g.C = qkv; g.ldc = 3 * D; g.M = M; g.N = 3 * D;
    if (self_aoi) set_qkv_aoi(g.aoi, qkv, B, Tq, D);
    TRY(run_gemm(h, g));
  }
  if (self_aoi) {                                        // Q|K|V arrived as operand images: attention3 general kernel
    const long long ib = aoi_img_bytes(B, Tq, D);
    Attn3Args t;
    t.Qi = reinterpret_cast<const char*>(qkv); t.Ki = t.Qi + ib; t.Vi = t.Qi + 2 * ib;
    t.q_len = q_len; t.k_len = q_len; t.ctx = sa; t.ldo = D; t.o_bs = (long long)Tq * D; t.ali = nullptr;
    t.B = B; t.H = heads; t.Tq = Tq; t.Tk = Tq; t.temperature = tau; t.causal = 1;
    TRY(run_attention3(h, t, false));
  }
  AttnArgs a;
  a.Q = qkv; a.ldq = 3 * D; a.K = qkv + D; a.ldk = 3 * D; a.V = qkv + 2 * D; a.ldv = 3 * D;
  a.q_len = q_len; a.k_len = q_len; a.ctx = sa; a.ldo = D; a.ali = nullptr; a.B = B; a.H = heads; a.Tq = Tq; a.Tk = Tq;
  a.causal = 1; a.temperature = tau;
  a.q_bs = (long long)Tq * 3 * D; a.k_bs = a.q_bs; a.v_bs = a.q_bs; a.o_bs = (long long)Tq * D;
  if (!self_aoi) TRY(run_attention(h, a, false));

  // ---- fused row-panel chains (gemm3.hip) when the split images exist and the widths fit one 256-column panel --------
  SplitRef r_p1, r_q, r_p2, r_f1, r_f2;
  bool chain = split_active(h) && h->split_scope && h->chain_enabled && D <= 256 && !(D & 31) && !(F & 31) &&
               split_lookup(h, k.proj1_wt, 2 * D, D, r_p1) && split_lookup(h, k.q_wt, D, D, r_q) &&
               split_lookup(h, k.proj2_wt, 2 * D, D, r_p2) && split_lookup(h, k.ffn1_wt, D, F, r_f1) &&
               split_lookup(h, k.ffn2_wt, F, D, r_f2) && r_p1.opm && r_q.opm && r_p2.opm && r_f1.opm && r_f2.opm;
  std::vector<SplitRef> r_t(tails.size());
  int tail_stages = 0;
  for (size_t i = 0; chain && i < tails.size(); ++i) {
    chain = split_lookup(h, tails[i].wt, D, tails[i].n, r_t[i]) && r_t[i].opm && !(tails[i].n & 3) && !(tails[i].ldo & 3);
    tail_stages += (tails[i].n + 255) / 256;
  }
  const int PT = D / 32, nchunks = (F + 255) / 256;
  if (chain && 1 + 2 * nchunks + tail_stages > kMaxChainStages) chain = false;
  // the coupling + next pre-chain behind the heads: needs the three panels of a 32-row workgroup, one tail of <= 256 columns that is
  // the log_scale | shift pair, and split images of every appended layer
  std::vector<SplitRef> r_p;
  int post_stages = 0;
  bool post_ok = chain && post && !h->chain_rows64 && tails.size() == 1 && !(tails[0].n & 63) && tails[0].n <= 256 && tails[0].qkv_T == 0;
  static const bool no_post = getenv("VNR_NO_POSTCHAIN") != nullptr;      // A/B switch: coupling kernel + pre-chain launch per flow step
  if (no_post) post_ok = false;
  if (post_ok) {
    r_p.resize(post->stages.size());
    for (size_t i = 0; post_ok && i < post->stages.size(); ++i) {
      const PreStage& ps = post->stages[i];
      post_ok = !(ps.K & 31) && ps.K <= 256 && !(ps.N & 3) && !(ps.out && (ps.ldo & 3)) && !(ps.N > 256 && ps.dst >= 0) &&
                split_lookup(h, ps.wt, ps.K, ps.N, r_p[i]) && r_p[i].opm;
      post_stages += (ps.N + 255) / 256;
    }
  }

  // one launch for the whole block after the self-attention (chain B -> cross-attention -> chain C): the query projection leaves
  // q in LDS, the workgroup attends over the text K/V images itself and continues with the context in place (gemm3.hip,
  // ChainArgs::att_stage).  Needs the operand images of the memory, no alignment output, 32-row panels, four 64-wide heads.
  // (round 4: also when the alignments ARE requested -- the decoder's blocks, decoder.py:188-192: the workgroup writes its rows of them)
  static const bool no_fuse_ali = getenv("VNR_NO_FUSE_ALI") != nullptr;      // A/B switch: chain B, attn3_kernel<true>, chain C as three launches
  const bool fused = chain && ka && (!ali || (!(Tt & 3) && !no_fuse_ali)) && h->fuse_xattn && !h->chain_rows64 && D == 256 && heads * 64 == D && Tt <= 128 &&
                     3 + 2 * nchunks + tail_stages <= kMaxChainStages;
  if (post_ok && (fused ? 3 : 1) + 2 * nchunks + tail_stages + post_stages > kMaxChainStages) post_ok = false;
  ChainArgs cb; memset(&cb, 0, sizeof(cb));
  if (chain) {
    // chain B: y = LN1(att_proj1(concat(x, sa)) + x) ; q = y . Wq
    ChainArgs& c = cb;
    c.in0 = x; c.ld0 = D; c.in1 = sa; c.ld1 = D; c.M = M; c.D = D; c.nstages = 2;
    ChainStage& s0 = c.st[0];
    s0.w = r_p1.opm; s0.kt_total = r_p1.kt_total; s0.kt0 = 0; s0.nk = 2 * PT; s0.n = D; s0.a0 = 0; s0.a1 = 1; s0.asw = PT; s0.bias = k.proj1_b;
    s0.act = ACT_IDENTITY; s0.res = 0; s0.gamma = k.ln1_g; s0.beta = k.ln1_b; s0.acc_mode = 0; s0.out = fused ? nullptr : y; s0.ldo = D; s0.dst = 0; s0.scale = r_p1.scale;
    ChainStage& s1 = c.st[1];
    s1.w = r_q.opm; s1.kt_total = r_q.kt_total; s1.kt0 = 0; s1.nk = PT; s1.n = D; s1.a0 = 0; s1.a1 = 0; s1.asw = PT; s1.bias = nullptr; s1.act = ACT_IDENTITY;
    s1.res = -1; s1.gamma = nullptr; s1.beta = nullptr; s1.acc_mode = 0; s1.scale = r_q.scale;
    if (fused) { s1.out = nullptr; s1.ldo = 0; s1.dst = 1; s1.out_fmt = 0; }
    else { s1.out = q; s1.ldo = D; s1.dst = -1; s1.out_fmt = ka ? 1 : 0; s1.aoi_T = Tq; }
    if (!fused) TRY(run_chain(h, c, 2.0 * M * D * (2.0 * D + D)));
  } else {
    g = GemmArgs(); g.A1 = x; g.lda1 = D; g.K1 = D; g.A2 = sa; g.lda2 = D; g.K = 2 * D; g.Wt = k.proj1_wt; g.ldw = 2 * D;
    g.bias = k.proj1_b; g.residual = x; g.ldr = D; g.ln_gamma = k.ln1_g; g.ln_beta = k.ln1_b; g.C = y; g.ldc = D; g.M = M; g.N = D;
    if (D > 256) { g.ln_gamma = nullptr; g.ln_beta = nullptr; TRY(run_gemm(h, g)); TRY(run_ln(h, y, k.ln1_g, k.ln1_b, M, D, y)); }
    else TRY(run_gemm(h, g));
    g = GemmArgs(); g.A1 = y; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = k.q_wt; g.ldw = D; g.C = q; g.ldc = D; g.M = M; g.N = D;
    if (ka) { g.aoi.mode = 1; g.aoi.D = D; g.aoi.T = Tq; g.aoi.TT = (Tq + 31) / 32; g.aoi.qk = reinterpret_cast<char*>(q); }
    TRY(run_gemm(h, g));
  }
  // cross attention
  if (fused) {
    // (inside the chain launch below)
  } else if (ka) {
    Attn3Args t;
    t.Qi = reinterpret_cast<const char*>(q);
    t.Ki = ka->d.qk + (size_t)kv_blk * ka->d.blk_bytes; t.Vi = ka->d.vt + (size_t)kv_blk * ka->d.blk_bytes;
    t.q_len = q_len; t.k_len = m_len; t.ctx = ca; t.ldo = D; t.o_bs = (long long)Tq * D; t.ali = ali;
    t.B = B; t.H = heads; t.Tq = Tq; t.Tk = Tt; t.temperature = tau;
    if (heads * 64 != D) return fail(h, VNR_ERR_ARG, "attention3: heads * 64 != attention width");
    TRY(run_attention3(h, t, true));
  } else {
  a.Q = q; a.ldq = D; a.K = kv + k.kv_col; a.ldk = kv_ld; a.V = kv + k.kv_col + D; a.ldv = kv_ld;
  a.q_len = q_len; a.k_len = m_len; a.ctx = ca; a.ldo = D; a.ali = ali; a.Tq = Tq; a.Tk = Tt; a.causal = 0;
  a.q_bs = (long long)Tq * D; a.k_bs = (long long)Tt * kv_ld; a.v_bs = a.k_bs; a.o_bs = (long long)Tq * D;
  TRY(run_attention(h, a, true));
  }

  if (chain) {
    // chain C: o = LN2(att_proj2(concat(y, ca)) + y) ; out = LN(dense2(relu(dense1(o))) + o) ; tails on out
    ChainArgs c; memset(&c, 0, sizeof(c));
    c.in0 = y; c.ld0 = D; c.in1 = ca; c.ld1 = D; c.M = M; c.D = D;
    int n = 0;
    if (fused) {                                           // chain B's two stages first, then the attention, then chain C
      c = cb;
      n = 2;
      c.att_stage = 2;
      c.att_K = ka->d.qk + (size_t)kv_blk * ka->d.blk_bytes; c.att_V = ka->d.vt + (size_t)kv_blk * ka->d.blk_bytes;
      c.att_qlen = q_len; c.att_klen = m_len; c.att_Tq = Tq; c.att_Tk = Tt; c.att_B = B; c.att_temp = tau;
      c.att_ali = ali;                                     // (then the context is left in panel 2: the alignment pass needs the queries once more)
      if (h->chain_segments && (Tq & 31)) c.seg_T = Tq;    // no panel straddles two batch elements: the attention phase runs once in every workgroup
    }
    ChainStage* s = &c.st[n++];
    s->w = r_p2.opm; s->kt_total = r_p2.kt_total; s->kt0 = 0; s->nk = 2 * PT; s->n = D; s->a0 = 0; s->a1 = (fused && ali) ? 2 : 1; s->asw = PT; s->bias = k.proj2_b; s->act = ACT_IDENTITY;
    s->res = 0; s->gamma = k.ln2_g; s->beta = k.ln2_b; s->acc_mode = 0; s->out = nullptr; s->ldo = 0; s->dst = 0; s->scale = r_p2.scale;
    for (int ch = 0; ch < nchunks; ++ch) {                 // hidden columns [256*ch, 256*ch + w)
      const int c0 = ch * 256, w = (F - c0 < 256) ? F - c0 : 256;
      const int hp = (!h->chain_rows64 && (ch & 1)) ? 2 : 1;   // hidden chunks alternate between two panels (32-row workgroups have three):
                                                           // chunk ch + 1 can be written while slower waves still read chunk ch
      s = &c.st[n++];                                      // h_ch = relu(o . W1[:, chunk] + b1[chunk])  -> panel hp
      s->w = r_f1.opm + (size_t)(c0 / 32) * r_f1.kt_total * 4096; s->kt_total = r_f1.kt_total; s->kt0 = 0; s->nk = PT; s->n = w;
      s->a0 = 0; s->a1 = 0; s->asw = PT; s->bias = k.ffn1_b + c0; s->act = ACT_RELU; s->res = -1; s->gamma = nullptr; s->beta = nullptr; s->acc_mode = 0;
      s->out = nullptr; s->ldo = 0; s->dst = hp; s->scale = r_f1.scale;
      s = &c.st[n++];                                      // acc += h_ch . W2[chunk, :]
      const bool last = ch == nchunks - 1;
      s->w = r_f2.opm; s->kt_total = r_f2.kt_total; s->kt0 = c0 / 32; s->nk = w / 32; s->n = D; s->a0 = hp; s->a1 = hp; s->asw = w / 32;
      s->bias = last ? k.ffn2_b : nullptr; s->act = ACT_IDENTITY; s->res = last ? 0 : -1; s->gamma = last ? k.ffn_g : nullptr; s->beta = last ? k.ffn_b : nullptr;
      s->acc_mode = nchunks == 1 ? 0 : (ch == 0 ? 1 : (last ? 3 : 2));
      s->out = last ? out : nullptr; s->ldo = D; s->dst = last ? 0 : -1; s->scale = r_f2.scale;
    }
    double fl = 2.0 * M * D * (2.0 * D) + 4.0 * M * (double)D * F;
    if (fused) fl += 2.0 * M * D * (2.0 * D + D) + 4.0 * (double)B * heads * Tq * (double)Tt * 64;     // chain B + the attention products
    for (size_t i = 0; i < tails.size(); ++i)
      for (int c0 = 0; c0 < tails[i].n; c0 += 256) {
        const int w = (tails[i].n - c0 < 256) ? tails[i].n - c0 : 256;
        s = &c.st[n++];
        s->w = r_t[i].opm + (size_t)(c0 / 32) * r_t[i].kt_total * 4096; s->kt_total = r_t[i].kt_total; s->kt0 = 0; s->nk = PT; s->n = w;
        s->a0 = 0; s->a1 = 0; s->asw = PT; s->bias = tails[i].bias ? tails[i].bias + c0 : nullptr; s->act = ACT_IDENTITY; s->res = -1; s->gamma = nullptr; s->beta = nullptr;
        s->acc_mode = 0; s->out = tails[i].out + c0; s->ldo = tails[i].ldo; s->dst = -1; s->scale = r_t[i].scale;
        if (tails[i].qkv_T > 0) { s->out = tails[i].out; s->out_fmt = 4; s->aoi_T = tails[i].qkv_T; s->aoi_D = tails[i].n / 3; s->aoi_c0 = c0;
                                  s->aoi_img_bytes = aoi_img_bytes(tails[i].qkv_B, tails[i].qkv_T, tails[i].n / 3); }
        fl += 2.0 * M * D * (double)w;
      }
    if (post_ok) {
      // the heads stay on the CU: coupling in their epilogue, the coupled z in panel 1, then the appended pre-chain
      ChainStage& hs = c.st[n - 1];
      hs.out = nullptr; hs.ldo = 0; hs.dst = 1;
      c.cpl_stage = n - 1; c.cpl_z = post->z; c.cpl_ld = post->ld; c.cpl_zp_off = post->zp_off; c.cpl_cond_off = post->cond_off;
      for (size_t i = 0; i < post->stages.size(); ++i) {
        const PreStage& ps = post->stages[i];
        for (int c0 = 0; c0 < ps.N; c0 += 256) {
          s = &c.st[n++];
          s->w = r_p[i].opm + (size_t)(c0 / 32) * r_p[i].kt_total * 4096; s->kt_total = r_p[i].kt_total; s->kt0 = 0; s->nk = ps.K / 32;
          s->n = ps.N - c0 < 256 ? ps.N - c0 : 256; s->a0 = ps.src; s->a1 = ps.src; s->asw = s->nk; s->akt0 = ps.akt0;
          s->bias = ps.bias ? ps.bias + c0 : nullptr; s->act = ACT_IDENTITY; s->res = -1; s->acc_mode = 0;
          s->pe = ps.pe; s->pe_T = ps.pe_T; s->pe_w = ps.pe_w;
          s->out = ps.out ? ps.out + c0 : nullptr; s->ldo = ps.ldo; s->dst = ps.dst; s->scale = r_p[i].scale;
          if (ps.out && ps.qkv_T > 0) { s->out = ps.out; s->out_fmt = 4; s->aoi_T = ps.qkv_T; s->aoi_D = ps.N / 3; s->aoi_c0 = c0; s->aoi_img_bytes = aoi_img_bytes(ps.qkv_B, ps.qkv_T, ps.N / 3); }
          fl += 2.0 * M * (double)ps.K * s->n;
        }
      }
      if (post_done) *post_done = true;
    }
    c.nstages = n;
    TRY(run_chain(h, c, fl));
    return VNR_OK;
  }
  WS(o, (size_t)M * D); WS(hid, (size_t)M * F);
  // LN2(att_proj2(concat(y, ca)) + y)
  g = GemmArgs(); g.A1 = y; g.lda1 = D; g.K1 = D; g.A2 = ca; g.lda2 = D; g.K = 2 * D; g.Wt = k.proj2_wt; g.ldw = 2 * D;
  g.bias = k.proj2_b; g.residual = y; g.ldr = D; g.ln_gamma = k.ln2_g; g.ln_beta = k.ln2_b; g.C = o; g.ldc = D; g.M = M; g.N = D;
  if (D > 256) { g.ln_gamma = nullptr; g.ln_beta = nullptr; TRY(run_gemm(h, g)); TRY(run_ln(h, o, k.ln2_g, k.ln2_b, M, D, o)); }
  else TRY(run_gemm(h, g));
  // FFN (utils.py:48-53)
  g = GemmArgs(); g.A1 = o; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = k.ffn1_wt; g.ldw = D; g.bias = k.ffn1_b; g.act = ACT_RELU;
  g.C = hid; g.ldc = F; g.M = M; g.N = F;
  TRY(run_gemm(h, g));
  g = GemmArgs(); g.A1 = hid; g.lda1 = F; g.K1 = F; g.K = F; g.Wt = k.ffn2_wt; g.ldw = F; g.bias = k.ffn2_b; g.residual = o; g.ldr = D;
  g.ln_gamma = k.ffn_g; g.ln_beta = k.ffn_b; g.C = out; g.ldc = D; g.M = M; g.N = D;
  if (D > 256) { g.ln_gamma = nullptr; g.ln_beta = nullptr; TRY(run_gemm(h, g)); TRY(run_ln(h, out, k.ffn_g, k.ffn_b, M, D, out)); }
  else TRY(run_gemm(h, g));
  for (const Tail& t : tails) {      // unfused tails
    g = GemmArgs(); g.A1 = out; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = t.wt; g.ldw = D; g.bias = t.bias; g.C = t.out; g.ldc = t.ldo; g.M = M; g.N = t.n;
    if (t.qkv_T > 0) set_qkv_aoi(g.aoi, t.out, t.qkv_B, t.qkv_T, t.n / 3);
    TRY(run_gemm(h, g));
  }
  return VNR_OK;
}

// A short chain of Dense layers on one row panel (gemm3.hip) in front of a block stack: e.g. folded ActNorm o
// InvertibleLinear -> pre_projection (+PE) on the conditioning half -> Q|K|V of the first block.  Every stage reads a
// panel (optionally from a k-tile offset: a column window) and may write HBM and/or a panel; wide outputs are cut into
// 256-column stages.  *done = false (nothing launched) when a split image is missing or a shape does not fit.
int run_prechain(vnr_handle h, const float* in, int ld_in, int Cin, int M, const std::vector<PreStage>& stages, bool* done) {
  *done = false;
  static const bool off = getenv("VNR_NO_PRECHAIN") != nullptr;       // A/B switch
  if (off || !(split_active(h) && h->split_scope && h->chain_enabled) || Cin > 256 || (Cin & 31)) return VNR_OK;
  ChainArgs c; memset(&c, 0, sizeof(c));
  c.in0 = in; c.ld0 = ld_in; c.in1 = nullptr; c.M = M; c.D = Cin;
  int n = 0;
  double fl = 0.0;
  for (const PreStage& p : stages) {
    SplitRef r;
    if ((p.K & 31) || p.K > 256 || (p.N & 3) || (p.out && (p.ldo & 3)) || !split_lookup(h, p.wt, p.K, p.N, r) || !r.opm) return VNR_OK;
    if (p.N > 256 && p.dst >= 0) return VNR_OK;
    for (int c0 = 0; c0 < p.N; c0 += 256) {
      if (n >= kMaxChainStages) return VNR_OK;
      ChainStage& s = c.st[n++];
      s.w = r.opm + (size_t)(c0 / 32) * r.kt_total * 4096; s.kt_total = r.kt_total; s.kt0 = 0; s.nk = p.K / 32;
      s.n = p.N - c0 < 256 ? p.N - c0 : 256; s.a0 = p.src; s.a1 = p.src; s.asw = s.nk; s.akt0 = p.akt0;
      s.bias = p.bias ? p.bias + c0 : nullptr; s.act = ACT_IDENTITY; s.res = -1; s.acc_mode = 0;
      s.pe = p.pe; s.pe_T = p.pe_T; s.pe_w = p.pe_w;
      s.out = p.out ? p.out + c0 : nullptr; s.ldo = p.ldo; s.dst = p.dst; s.scale = r.scale;
      if (p.out && p.qkv_T > 0) { s.out = p.out; s.out_fmt = 4; s.aoi_T = p.qkv_T; s.aoi_D = p.N / 3; s.aoi_c0 = c0; s.aoi_img_bytes = aoi_img_bytes(p.qkv_B, p.qkv_T, p.N / 3); }
      fl += 2.0 * M * (double)p.K * s.n;
    }
  }
  c.nstages = n;
  TRY(run_chain(h, c, fl));
  *done = true;
  return VNR_OK;
}

// a stack of CrossAttentionBLKs (transform.py:53-56, decoder.py:188-192, posterior.py:124-127): block i's chain produces block
// i+1's Q|K|V; `tails` are applied to the last block's output.  Returns the buffer holding the stack output.
int run_xstack(vnr_handle h, const std::vector<XBlk>& blks, float* xa, float* xb, const float* kv, int kv_ld,
               const int32_t* q_len, const int32_t* m_len, int B, int Tq, int Tt, int heads, float tau, float* ali_base,
               size_t ali_stride, const std::vector<Tail>& tails, float** result, float* qkv_pre = nullptr,
               const PostChain* post = nullptr, bool* post_done = nullptr) {
  if (post_done) *post_done = false;
  const int M = B * Tq;
  float* xc = xa; float* xn = xb;
  if (blks.empty()) { *result = xc; return VNR_OK; }
  const int D = blks[0].D;
  const bool saoi = self_aoi_on(h, D, heads);            // (the callers' pre-chains decide with the same predicate)
  float* qkv0 = qkv_pre;                                 // Q|K|V of the first block already computed by a pre-chain
  if (!qkv0) { qkv0 = ws_alloc(h, qkv_floats(saoi, B, Tq, D)); if (!qkv0) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed"); }
  WS(qkv1, qkv_floats(saoi, B, Tq, D));
  (void)M;
  float* qc = qkv0; float* qn = qkv1;
  bool ready = qkv_pre != nullptr;
  for (size_t b = 0; b < blks.size(); ++b) {
    std::vector<Tail> t;
    if (b + 1 < blks.size()) t.push_back({blks[b + 1].qkv_wt, 3 * D, nullptr, qn, 3 * D, saoi ? Tq : 0, B});
    else t = tails;
    const bool last = b + 1 == blks.size();
    TRY(run_xblk(h, blks[b], xc, xn, kv, kv_ld, q_len, m_len, B, Tq, Tt, heads, tau,
                 ali_base ? ali_base + b * ali_stride : nullptr, qc, ready, t, saoi, last ? post : nullptr, last ? post_done : nullptr));
    ready = b + 1 < blks.size();      // (also true on the unfused path: the tails loop computed it)
    std::swap(xc, xn); std::swap(qc, qn);
  }
  *result = xc;
  return VNR_OK;
}

// a_split / c_split: the input / output activation is in "split rows" (common.h, GemmArgs::a_split) -- inference mode only
int run_conv(vnr_handle h, const ConvL& c, const float* x, const int32_t* gather, int B, int T, int act,
             int bn_first, float* y, int a_split = 0, int c_split = 0) {
  GemmArgs g;
  g.a_split = a_split; g.c_split = c_split;
  g.A1 = x; g.lda1 = c.cin; g.K1 = c.k * c.cin; g.K = c.k * c.cin; g.Wt = c.wt; g.ldw = c.k * c.cin;
  g.bias = c.bias; g.act = act; g.bn_scale = c.bn_scale; g.bn_shift = c.bn_shift; g.bn_first = bn_first;
  g.C = y; g.ldc = c.cout; g.M = B * T; g.N = c.cout; g.taps = c.k; g.conv_T = T; g.conv_C = c.cin;
  if (gather) return fail(h, VNR_ERR_ARG, "run_conv: the row gather is a separate kernel (launch_gather_rows)");
  if (!h->training) return run_gemm(h, g);
  // training=True (utils.py:76-85): conv -> act -> BatchNormalization on BATCH statistics (all B*T rows, padding
  // included; moving statistics updated, momentum 0.99) -> Dropout
  if (bn_first) return fail(h, VNR_ERR_ARG, "training mode supports bn_before_act=False only");
  g.bn_scale = nullptr; g.bn_shift = nullptr;
  TRY(run_gemm(h, g));
  const int M = B * T, C = c.cout;
  WS(stat, (size_t)4 * C + 2 * C);                       // 2*C doubles (mean, centred squares) + scale/shift floats
  double* mean = reinterpret_cast<double*>(stat);
  double* sq = mean + C;
  float* sc = stat + 4 * C; float* sh = sc + C;
  HIP_TRY(h, hipMemsetAsync(stat, 0, (size_t)4 * C * sizeof(float), h->stream));
  RUN_MISC(h, launch_col_sum(y, M, C, C, nullptr, mean, h->stream));
  RUN_MISC(h, launch_scale_d(mean, C, 1.0 / (double)M, h->stream));
  RUN_MISC(h, launch_col_sum(y, M, C, C, mean, sq, h->stream));
  RUN_MISC(h, launch_bn_train_finish(mean, sq, M, C, c.gamma, c.beta, 0.99f, c.moving_mean, c.moving_var, sc, sh, h->stream, 0, h->range_flag));
  RUN_MISC(h, launch_rowop(y, M, C, sc, sh, nullptr, 1, 0.f, c.drop_rate, site_key(h->drop_seed, c.site), y, h->stream));
  return VNR_OK;
}
// A conv stack can hand its activations from layer to layer as split rows when the split-fp16 path is on for this scope, every
// layer has its weight image and the channel counts are whole 32-channel tiles (option "split_rows", default 1)
bool split_rows_ok(vnr_handle h, const std::vector<ConvL>& convs, int M) {
  if (!h->split_rows || h->training || !split_active(h) || !h->split_scope || convs.empty() || M < 64) return false;      // (run_gemm takes the split path from 64 rows)
  for (const ConvL& c : convs) {
    SplitRef r;
    if ((c.cout & 31) || !split_lookup(h, c.wt, c.k * c.cin, c.cout, r)) return false;
  }
  return true;
}
// the moving statistics changed (training-mode forward): refresh the folded inference affine of every BN
int refresh_bn_affine(vnr_handle h) {
  for (auto* v : {&h->enc_convs, &h->post_convs})
    for (const ConvL& c : *v)
      RUN_MISC(h, launch_bn_affine(c.gamma, c.beta, c.moving_mean, c.moving_var, c.cout, const_cast<float*>(c.bn_scale),
                                   const_cast<float*>(c.bn_shift), h->stream));
  return VNR_OK;
}

// cross-attention K|V of the memory for a group of blocks: one GEMM over a stacked panel
// Cross K|V of the memory for a run of blocks (one GEMM).  D > 0: every block of the panel has attention width D = H*64 and the
// consumers are run_xblk cross-attentions; when Tt <= 128 the panel is then written as attention operand images (side buffers;
// common.h) instead of fp32 and registered in h->kv_aoi so that run_xblk takes attention3.
int run_kv(vnr_handle h, const float* text_embd, int B, int Tt, int mem, const float* panel, int n, float* out, int D) {
  GemmArgs g;
  g.A1 = text_embd; g.lda1 = mem; g.K1 = mem; g.K = mem; g.Wt = panel; g.ldw = mem; g.C = out; g.ldc = n; g.M = B * Tt; g.N = n;
  if (h->aoi_enabled && split_active(h) && D > 0 && !(D & 63) && Tt <= 128 && n % (2 * D) == 0 && gemm2_supported(g)) {
    const int nblk = n / (2 * D), TT = (Tt + 31) / 32;
    const size_t blk_bytes = (size_t)B * (D / 64) * TT * kAoiTile;
    WS(img, 2 * nblk * blk_bytes / 4);
    g.aoi.mode = 3; g.aoi.D = D; g.aoi.T = Tt; g.aoi.TT = TT; g.aoi.blk_bytes = (long long)blk_bytes;
    g.aoi.qk = reinterpret_cast<char*>(img); g.aoi.vt = g.aoi.qk + nblk * blk_bytes;
    h->kv_aoi.push_back({out, n, D, B, Tt, g.aoi});
  }
  return run_gemm(h, g);
}
int encoder_body_impl(vnr_handle h, const int32_t* ids, const int32_t* lens, int B, int T, float pos_step, float* out);
// The encoder feeds the length predictor, whose float sum is truncated to an integer frame count (inference.py:135).
// Option "split_encoder" = 0 keeps this chain on the exact fp32 MFMA path; the split path measured equally accurate
// (profiles/r01_split_accuracy.txt) and is the default.
int encoder_body(vnr_handle h, const int32_t* ids, const int32_t* lens, int B, int T, float pos_step, float* out) {
  const bool saved = h->split_scope;
  h->split_scope = h->split_encoder;
  const int rc = encoder_body_impl(h, ids, lens, B, T, pos_step, out);
  h->split_scope = saved;
  return rc;
}
int encoder_body_impl(vnr_handle h, const int32_t* ids, const int32_t* lens, int B, int T, float pos_step, float* out) {
  const vnr_config& c = h->cfg;
  const int M = B * T, Dm = c.enc_pre_hidden, A = c.enc_attention_dim, F = c.enc_ffn_hidden;
  WS(xa, (size_t)M * Dm); WS(xb, (size_t)M * Dm);
  float* cur = xa; float* nxt = xb;
  // Embedding (encoder.py:81): a 4 MB row gather, then the conv stack (utils.py:33-38) on the DMA GEMM kernel
  if (c.enc_embd_dim > Dm) return fail(h, VNR_ERR_ARG, "embd_dim larger than pre_hidden is not supported");
  // split rows through the conv stack (embedding -> 3 convolutions -> projection): no fp32 -> (hi, lo) conversion in any k-loop
  SplitRef rproj;
  const bool sr = split_rows_ok(h, h->enc_convs, M) && !(c.enc_embd_dim & 31) && !(Dm & 31) && split_lookup(h, h->enc_proj_wt, Dm, Dm, rproj);
  if (sr) RUN_MISC(h, launch_gather_rows_split(h->emb, ids, M, c.enc_embd_dim, nxt, h->stream));
  else RUN_MISC(h, launch_gather_rows(h->emb, ids, M, c.enc_embd_dim, nxt, h->stream));
  std::swap(cur, nxt);
  for (size_t i = 0; i < h->enc_convs.size(); ++i) {
    TRY(run_conv(h, h->enc_convs[i], cur, nullptr, B, T, c.enc_pre_activation, c.enc_bn_before_act, nxt, sr, sr));
    std::swap(cur, nxt);
  }
  const float* pe = nullptr;
  TRY(get_pe(h, T, Dm, pos_step, &pe));
  GemmArgs g;
  g.A1 = cur; g.lda1 = Dm; g.K1 = Dm; g.K = Dm; g.Wt = h->enc_proj_wt; g.ldw = Dm; g.bias = h->enc_proj_b;
  g.pe = pe; g.pe_T = T; g.pe_w = h->enc_pos_weight; g.C = nxt; g.ldc = Dm; g.M = M; g.N = Dm;
  g.a_split = sr;
  TRY(run_gemm(h, g));                                       // prenet.projection + pos_weight*PE (encoder.py:85-86)
  if (h->training && c.enc_pos_drop_rate > 0.f)              // pe_dropout (encoder.py:87)
    RUN_MISC(h, launch_rowop(nxt, M, Dm, nullptr, nullptr, nullptr, 1, 0.f, c.enc_pos_drop_rate, site_key(h->drop_seed, SITE_ENC_PE), nxt, h->stream));
  std::swap(cur, nxt);
  const bool eaoi = self_aoi_on(h, A, c.enc_attention_heads);       // Q|K|V as attention operand images (attention3.hip)
  WS(qkv, qkv_floats(eaoi, B, T, A)); WS(att, (size_t)M * A); WS(y, (size_t)M * Dm); WS(hid, (size_t)M * F);
  for (size_t i = 0; i < h->enc_blks.size(); ++i) {
    const SBlk& k = h->enc_blks[i];
    g = GemmArgs(); g.A1 = cur; g.lda1 = Dm; g.K1 = Dm; g.K = Dm; g.Wt = k.qkv_wt; g.ldw = Dm; g.C = qkv; g.ldc = 3 * A; g.M = M; g.N = 3 * A;
    if (eaoi) set_qkv_aoi(g.aoi, qkv, B, T, A);
    TRY(run_gemm(h, g));
    if (eaoi) {
      const long long ib = aoi_img_bytes(B, T, A);
      Attn3Args t3;
      t3.Qi = reinterpret_cast<const char*>(qkv); t3.Ki = t3.Qi + ib; t3.Vi = t3.Qi + 2 * ib;
      t3.q_len = lens; t3.k_len = lens; t3.ctx = att; t3.ldo = A; t3.o_bs = (long long)T * A; t3.ali = nullptr;
      t3.B = B; t3.H = c.enc_attention_heads; t3.Tq = T; t3.Tk = T; t3.temperature = c.enc_attention_temperature; t3.causal = 0;
      TRY(run_attention3(h, t3, false));
    }
    AttnArgs a;
    a.Q = qkv; a.ldq = 3 * A; a.K = qkv + A; a.ldk = 3 * A; a.V = qkv + 2 * A; a.ldv = 3 * A;
    a.q_len = lens; a.k_len = lens; a.ctx = att; a.ldo = A; a.ali = nullptr; a.B = B; a.H = c.enc_attention_heads;
    a.Tq = T; a.Tk = T; a.causal = 0; a.temperature = c.enc_attention_temperature;
    a.q_bs = (long long)T * 3 * A; a.k_bs = a.q_bs; a.v_bs = a.q_bs; a.o_bs = (long long)T * A;
    if (!eaoi) TRY(run_attention(h, a, false));
    // LN(x + att_proj(concat(x, att)))  (attention.py:410-413)
    g = GemmArgs(); g.A1 = cur; g.lda1 = Dm; g.K1 = Dm; g.A2 = att; g.lda2 = A; g.K = Dm + A; g.Wt = k.proj_wt; g.ldw = Dm + A;
    g.bias = k.proj_b; g.residual = cur; g.ldr = Dm; g.C = y; g.ldc = Dm; g.M = M; g.N = Dm;
    if (Dm <= 256) { g.ln_gamma = k.ln_g; g.ln_beta = k.ln_b; TRY(run_gemm(h, g)); }
    else { TRY(run_gemm(h, g)); TRY(run_ln(h, y, k.ln_g, k.ln_b, M, Dm, y)); }
    g = GemmArgs(); g.A1 = y; g.lda1 = Dm; g.K1 = Dm; g.K = Dm; g.Wt = k.ffn1_wt; g.ldw = Dm; g.bias = k.ffn1_b; g.act = ACT_RELU;
    g.C = hid; g.ldc = F; g.M = M; g.N = F;
    TRY(run_gemm(h, g));
    float* dst = (i + 1 == h->enc_blks.size()) ? out : nxt;
    g = GemmArgs(); g.A1 = hid; g.lda1 = F; g.K1 = F; g.K = F; g.Wt = k.ffn2_wt; g.ldw = F; g.bias = k.ffn2_b; g.residual = y; g.ldr = Dm;
    g.C = dst; g.ldc = Dm; g.M = M; g.N = Dm;
    if (Dm <= 256) { g.ln_gamma = k.ffn_g; g.ln_beta = k.ffn_b; TRY(run_gemm(h, g)); }
    else { TRY(run_gemm(h, g)); TRY(run_ln(h, dst, k.ffn_g, k.ffn_b, M, Dm, dst)); }
    std::swap(cur, nxt);
  }
  if (h->enc_blks.empty()) HIP_TRY(h, hipMemcpyAsync(out, cur, (size_t)M * Dm * 4, hipMemcpyDeviceToDevice, h->stream));
  return VNR_OK;
}

// TransformerPrior.sample (prior.py:154-169); kv = prior cross K|V panel output [B*Tt, kv_ld].
// final_post / final_done: stages to run behind the LAST flow step's coupling inside its chain launch (the decoder's pre-chain in
// vnr_inference; the coupled z is in panel 1) and whether that happened.
// ---- Prior.Transformer.inverse = True (prior.py:81-99; engine option "prior_inverse") ------------------------------------------------
// Every flow of the prior is then built with inverse = True and BaseFlow.call / fwd_pass / bwd_pass (flow.py:36-47,76-113) swap
// _forward and _backward:
//   mode 0  sample / call (prior.py:101-117,154-169): steps 0 .. n-1, each ActNorm._backward -> InvertibleLinear._backward ->
//           coupling._backward, logprobs -= logdet;
//   mode 1  log_probability (prior.py:119-152): steps n-1 .. 0, each coupling._forward -> InvertibleLinear._forward -> ActNorm._forward,
//           accumulated logdet + the Gaussian of the result;
//   mode 2  init (prior.py:171-186): ActNorm.init (statistics, then _forward -- called directly, flow.py:189-196) ->
//           InvertibleLinear._backward (BaseFlow.call) -> coupling.init = _forward (flow.py:259-262).
// Neither LJHPS nor DataBakerHPS sets the flag (hparams.py:344,462): one unfused launch per operation, inv(W) by the in-LDS float64
// Gauss-Jordan kernel of the training step; the coupling networks run on the same block launches as everywhere else.
int prior_inverse_body(vnr_handle h, int mode, const int32_t* z_len, const int32_t* t_len, const float* kv, int kv_ld, int B, int Tz, int Tt,
                       const float* zin, float* z_out, float* logprobs) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, C = c.latent_dim, half = C / 2, D = c.prior_attention_dim;
  const int nsteps = (int)h->flow.size();
  if (nsteps > 8 || C > 128) return fail(h, VNR_ERR_ARG, "inverse flows: at most 8 flow steps of at most 128 channels");
  WS(za, (size_t)M * C); WS(zb, (size_t)M * C); WS(xa, (size_t)M * D); WS(xb, (size_t)M * D); WS(heads, (size_t)M * C);
  WS(stat, (size_t)4 * C + C); WS(wt, (size_t)C * C); WS(rowld, (size_t)M); WS(prm, (size_t)64 + 2 * C); WS(base, (size_t)B);
  float* sc = prm + 64; float* sh = prm + 64 + C; float* lssum = prm;
  const float* Wsrc[8]; float* Winv[8]; float* WinvT[8]; float* lad[8];
  for (int s = 0; s < nsteps; ++s) {
    Wsrc[s] = h->flow[s].lin_w; Winv[s] = ws_alloc(h, (size_t)C * C); WinvT[s] = ws_alloc(h, (size_t)C * C); lad[s] = ws_alloc(h, 64);
    if (!Winv[s] || !WinvT[s] || !lad[s]) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed");
  }
  if (mode != 1) RUN_MISC(h, launch_invert_batch(Wsrc, Winv, WinvT, lad, nsteps, C, h->stream));
  if (zin) HIP_TRY(h, hipMemcpyAsync(za, zin, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  else HIP_TRY(h, hipMemsetAsync(za, 0, (size_t)M * C * 4, h->stream));
  if (logprobs) {
    if (mode == 1) HIP_TRY(h, hipMemsetAsync(logprobs, 0, (size_t)B * 4, h->stream));
    else RUN_MISC(h, launch_gauss_logprob(za, z_len, B, Tz, C, logprobs, h->stream));                  // _initial_sample (prior.py:36-41)
  }
  const float* pe = nullptr;
  TRY(get_pe(h, Tz, D, 1.0f, &pe));
  float* zc = za; float* zn = zb;
  auto dense_cc = [&](const float* Wt_nk) -> int {          // zn = zc . W on the exact fp32 path, Wt_nk = W^T [N][K]
    GemmArgs g;
    g.A1 = zc; g.lda1 = C; g.K1 = C; g.K = C; g.Wt = Wt_nk; g.ldw = C; g.C = zn; g.ldc = C; g.M = M; g.N = C;
    const bool saved = h->split_scope; h->split_scope = false;
    const int rc = run_gemm(h, g);
    h->split_scope = saved;
    if (rc == VNR_OK) std::swap(zc, zn);
    return rc;
  };
  auto coupling = [&](int s, bool backward, float sign) -> int {     // on zc, in place; logprobs += sign * sum_valid log(scale)
    const FlowStep& f = h->flow[s];
    const bool upper = (s % 2) == 0;
    const int cond_off = upper ? 0 : half, zp_off = upper ? half : 0;
    GemmArgs g;
    g.A1 = zc + cond_off; g.lda1 = C; g.K1 = half; g.K = half; g.Wt = f.pre_wt; g.ldw = half; g.bias = f.pre_b;
    g.pe = pe; g.pe_T = Tz; g.pe_w = f.pos_weight; g.C = xa; g.ldc = D; g.M = M; g.N = D;
    TRY(run_gemm(h, g));
    float* xc = nullptr;
    TRY(run_xstack(h, f.blks, xa, xb, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.prior_attention_heads, c.prior_temperature,
                   nullptr, 0, {Tail{f.heads_wt, C, f.heads_b, heads, C}}, &xc));
    if (f.blks.empty()) {
      g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = f.heads_wt; g.ldw = D; g.bias = f.heads_b; g.C = heads; g.ldc = C; g.M = M; g.N = C;
      TRY(run_gemm(h, g));
    }
    if (backward) RUN_MISC(h, launch_coupling_bwd(heads, zc, M, half, zp_off, rowld, h->stream));
    else RUN_MISC(h, launch_coupling_fwd(heads, zc, M, half, zp_off, rowld, h->stream));
    if (logprobs) RUN_MISC(h, launch_masked_row_reduce(rowld, z_len, B, Tz, sign, logprobs, 1, h->stream));
    return VNR_OK;
  };
  for (int i = 0; i < nsteps; ++i) {
    const int s = mode == 1 ? nsteps - 1 - i : i;
    const FlowStep& f = h->flow[s];
    if (mode == 0) {
      // ActNorm._backward (flow.py:177-187): (z - b) / (exp(ls) + 1e-8), logdet = -len * sum(ls)
      RUN_MISC(h, launch_actnorm_inv_params(f.an_log_scale, f.an_bias, C, sc, sh, lssum, h->stream));
      RUN_MISC(h, launch_rowop(zc, M, C, sc, sh, nullptr, 1, 0.f, 0.f, 0u, zc, h->stream));
      if (logprobs) RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lssum, 1.f, B, h->stream));
      // InvertibleLinear._backward (flow.py:137-150): z . inv(W), logdet = len * log|det inv(W)|
      TRY(dense_cc(WinvT[s]));
      if (logprobs) RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lad[s], -1.f, B, h->stream));
      TRY(coupling(s, true, 1.0f));                          // logdet = -sum log(scale): logprobs -= logdet
    } else if (mode == 1) {
      TRY(coupling(s, false, 1.0f));                         // logdet = +sum log(scale), accumulated
      RUN_MISC(h, launch_transpose(f.lin_w, C, C, wt, C, h->stream));
      TRY(dense_cc(wt));                                     // InvertibleLinear._forward: z . W, logdet = len * log|det W|
      if (logprobs) RUN_MISC(h, launch_axpy_len(logprobs, z_len, (float)f.lin_logdet, B, h->stream));
      // ActNorm._forward (flow.py:166-175): z * exp(ls) + b, logdet = len * sum(ls)
      RUN_MISC(h, launch_actnorm_fwd_params(f.an_log_scale, C, sc, lssum, h->stream));
      RUN_MISC(h, launch_rowop(zc, M, C, sc, f.an_bias, nullptr, 1, 0.f, 0.f, 0u, zc, h->stream));
      if (logprobs) RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lssum, 1.f, B, h->stream));
    } else {
      // ActNorm.init: per-channel mean / population std over ALL B*Tz rows (padding included), then the forward (flow.py:189-196)
      double* mean = reinterpret_cast<double*>(stat);
      double* sq = mean + C;
      float* isc = stat + 4 * C;
      HIP_TRY(h, hipMemsetAsync(stat, 0, (size_t)4 * C * sizeof(float), h->stream));
      RUN_MISC(h, launch_col_sum(zc, M, C, C, nullptr, mean, h->stream));
      RUN_MISC(h, launch_scale_d(mean, C, 1.0 / (double)M, h->stream));
      RUN_MISC(h, launch_col_sum(zc, M, C, C, mean, sq, h->stream));
      RUN_MISC(h, launch_actnorm_init_finish(mean, sq, M, C, f.an_log_scale, f.an_bias, isc, h->stream));
      RUN_MISC(h, launch_rowop(zc, M, C, isc, f.an_bias, nullptr, 1, 0.f, 0.f, 0u, zc, h->stream));
      if (logprobs) {
        RUN_MISC(h, launch_actnorm_fwd_params(f.an_log_scale, C, sc, lssum, h->stream));
        RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lssum, -1.f, B, h->stream));
      }
      TRY(dense_cc(WinvT[s]));                               // linear(z) is BaseFlow.call -> _backward
      if (logprobs) RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lad[s], -1.f, B, h->stream));
      TRY(coupling(s, false, -1.0f));                        // coupling.init = _forward: logprobs -= sum log(scale)
    }
  }
  if (mode == 1 && logprobs) {
    RUN_MISC(h, launch_gauss_logprob(zc, z_len, B, Tz, C, base, h->stream));     // -0.5 (log 2pi + eps^2), masked (prior.py:147-150)
    RUN_MISC(h, launch_masked_row_reduce(base, nullptr, B, 1, 1.0f, logprobs, 1, h->stream));
  }
  if (z_out) HIP_TRY(h, hipMemcpyAsync(z_out, zc, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  return VNR_OK;
}

int prior_body(vnr_handle h, const int32_t* z_len, const int32_t* t_len, const float* kv, int kv_ld, int B,
               int Tz, int Tt, const float* eps, float* z_out, float* logprobs, const std::vector<PreStage>* final_post = nullptr,
               bool* final_done = nullptr) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, C = c.latent_dim, half = C / 2, D = c.prior_attention_dim;
  if (final_done) *final_done = false;
  if (h->prior_inverse) return prior_inverse_body(h, 0, z_len, t_len, kv, kv_ld, B, Tz, Tt, eps, z_out, logprobs);
  WS(za, (size_t)M * C); WS(xa0, (size_t)M * D); WS(xa1, (size_t)M * D); WS(xb, (size_t)M * D); WS(heads, (size_t)M * C); WS(rowld, (size_t)M);
  float* xas[2] = {xa0, xa1};
  // (round 6: the first flow step READS the caller's noise in place -- every step writes its result to za / zb / z_out, never to its
  //  input -- instead of a 4 MB copy in front of it: one 5 us launch less per inference)
  const bool eps_in_place = eps && eps != z_out;         // (a caller that samples INTO its noise buffer keeps the copy)
  if (!eps) HIP_TRY(h, hipMemsetAsync(za, 0, (size_t)M * C * 4, h->stream));
  else if (!eps_in_place) HIP_TRY(h, hipMemcpyAsync(za, eps, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  if (logprobs) RUN_MISC(h, launch_gauss_logprob(eps, z_len, B, Tz, C, logprobs, h->stream));
  const float* pe = nullptr;
  TRY(get_pe(h, Tz, D, 1.0f, &pe));                       // transform.py:51
  WS(zb, (size_t)M * C);
  float* zc = eps_in_place ? const_cast<float*>(eps) : za;
  const int nsteps = (int)h->flow.size();
  const bool saoi = self_aoi_on(h, D, c.prior_attention_heads);
  // the pre-chain of flow step s on chain panels (p_in -> p_mid -> p_out): (ActNorm o InvertibleLinear) -> pre_projection +
  // pos_weight * PE on the conditioning half -> Q|K|V of the first block
  auto pre_stages = [&](int s, float* dst, float* xa, float* qkv_pre, int p_in, int p_mid, int p_out) {
    const FlowStep& f = h->flow[s];
    const int cond_off = (s % 2) == 0 ? 0 : half;
    return std::vector<PreStage>{
        PreStage{f.fold_wt, C, C, p_in, 0, f.fold_b, nullptr, 1, 0.f, dst, C, p_mid},
        (h->chain_waves4 && f.pre_wt_x && C == 128) ? PreStage{f.pre_wt_x, C, D, p_mid, 0, f.pre_b, pe, Tz, f.pos_weight, xa, D, p_out}
                                                    : PreStage{f.pre_wt, half, D, p_mid, cond_off / 32, f.pre_b, pe, Tz, f.pos_weight, xa, D, p_out},
        PreStage{f.blks[0].qkv_wt, D, 3 * D, p_out, 0, nullptr, nullptr, 1, 0.f, qkv_pre, 3 * D, -1, saoi ? Tz : 0, B}};
  };
  auto step_dst = [&](int s, float* cur) { return (s == nsteps - 1) ? z_out : (cur == za ? zb : za); };
  auto chainable = [&](int s) { return !h->flow[s].blks.empty() && !(half & 31) && h->flow[s].blks[0].D == D; };
  bool pre_done = false;                                  // this step's pre-chain ran at the end of the previous step's last chain launch
  float* qkv_pre = nullptr;
  float* dst = nullptr;
  for (int s = 0; s < nsteps; ++s) {
    const FlowStep& f = h->flow[s];
    if (!pre_done) dst = step_dst(s, zc);
    float* xa = xas[s & 1];
    const bool upper = (s % 2) == 0;                      // prior.py:85-87
    const int cond_off = upper ? 0 : half, zp_off = upper ? half : 0;   // flow.py:227-228
    // one row-panel chain: (ActNorm o InvertibleLinear) -> pre_projection + pos_weight*PE on the conditioning half -> Q|K|V
    // of the first block (three launches otherwise)
    bool fused = pre_done;
    if (!pre_done) {
      qkv_pre = nullptr;
      if (chainable(s)) {
        qkv_pre = ws_alloc(h, qkv_floats(saoi, B, Tz, D));
        if (!qkv_pre) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed");
        TRY(run_prechain(h, zc, C, C, M, pre_stages(s, dst, xa, qkv_pre, 0, 1, 0), &fused));
      }
    }
    GemmArgs g;
    if (!fused) {
      qkv_pre = nullptr;
      // actnorm o invertible linear
      g.A1 = zc; g.lda1 = C; g.K1 = C; g.K = C; g.Wt = f.fold_wt; g.ldw = C; g.bias = f.fold_b; g.C = dst; g.ldc = C; g.M = M; g.N = C;
      TRY(run_gemm(h, g));
      g = GemmArgs(); g.A1 = dst + cond_off; g.lda1 = C; g.K1 = half; g.K = half; g.Wt = f.pre_wt; g.ldw = half; g.bias = f.pre_b;
      g.pe = pe; g.pe_T = Tz; g.pe_w = f.pos_weight; g.C = xa; g.ldc = D; g.M = M; g.N = D;
      TRY(run_gemm(h, g));
    }
    if (logprobs) RUN_MISC(h, launch_axpy_len(logprobs, z_len, (float)(-f.logdet_per_frame), B, h->stream));
    // behind the heads of this step's last block, in the same chain launch: the coupling on dst and the NEXT pre-chain (the next flow
    // step's, or the caller's after the last step).  Not when the log-determinants are wanted (the coupling kernel leaves them).
    PostChain post; post.z = dst; post.ld = C; post.zp_off = zp_off; post.cond_off = cond_off;
    float* next_dst = nullptr; float* next_qkv = nullptr;
    bool want_post = !logprobs && chainable(s) && !(C & 63);
    if (want_post && s + 1 < nsteps) {
      if (chainable(s + 1)) {
        next_dst = step_dst(s + 1, dst);
        next_qkv = ws_alloc(h, qkv_floats(saoi, B, Tz, D));
        if (!next_qkv) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed");
        post.stages = pre_stages(s + 1, next_dst, xas[(s + 1) & 1], next_qkv, 1, 2, 0);
      }
    } else if (want_post && final_post) post.stages = *final_post;
    float* xc = nullptr;        // the log_scale | shift heads ride on the last block's chain (tail)
    bool post_done = false;
    TRY(run_xstack(h, f.blks, xa, xb, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.prior_attention_heads, c.prior_temperature,
                   nullptr, 0, {Tail{f.heads_wt, C, f.heads_b, heads, C}}, &xc, qkv_pre, want_post ? &post : nullptr, &post_done));
    if (f.blks.empty()) {
      g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = f.heads_wt; g.ldw = D; g.bias = f.heads_b;
      g.C = heads; g.ldc = C; g.M = M; g.N = C;
      TRY(run_gemm(h, g));
    }
    if (!post_done) {
      RUN_MISC(h, launch_coupling_fwd(heads, dst, M, half, zp_off, logprobs ? rowld : nullptr, h->stream));
      if (logprobs) RUN_MISC(h, launch_masked_row_reduce(rowld, z_len, B, Tz, -1.0f, logprobs, 1, h->stream));
    }
    zc = dst;
    pre_done = post_done && !post.stages.empty() && s + 1 < nsteps;
    if (pre_done) { dst = next_dst; qkv_pre = next_qkv; }
    if (post_done && s + 1 == nsteps && final_post && !post.stages.empty() && final_done) *final_done = true;
  }
  if (nsteps == 0) HIP_TRY(h, hipMemcpyAsync(z_out, za, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  return VNR_OK;
}

// TransformerPrior.init (prior.py:171-186): like sample(), but every ActNorm first sets its variables from the statistics
// of its input (ActNormFlow.init, flow.py:189-196) -- written straight into the weight store.  Runs unfolded (the folded
// panels are rebuilt by the caller afterwards).
int prior_init_body(vnr_handle h, const int32_t* z_len, const int32_t* t_len, const float* kv, int kv_ld, int B,
                    int Tz, int Tt, const float* eps, float* z_out, float* logprobs = nullptr) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, C = c.latent_dim, half = C / 2, D = c.prior_attention_dim;
  if (h->prior_inverse) return prior_inverse_body(h, 2, z_len, t_len, kv, kv_ld, B, Tz, Tt, eps, z_out, logprobs);
  WS(za, (size_t)M * C); WS(zb, (size_t)M * C); WS(xa, (size_t)M * D); WS(xb, (size_t)M * D); WS(heads, (size_t)M * C);
  WS(stat, (size_t)4 * C + C); WS(wt, (size_t)C * C); WS(rowld, (size_t)M); WS(lsum, (size_t)64 + 2 * C);
  if (logprobs) RUN_MISC(h, launch_gauss_logprob(eps, z_len, B, Tz, C, logprobs, h->stream));      // _initial_sample (prior.py:36-41)
  double* mean = reinterpret_cast<double*>(stat);
  double* sq = mean + C;
  float* sc = stat + 4 * C;
  if (eps) HIP_TRY(h, hipMemcpyAsync(za, eps, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  else HIP_TRY(h, hipMemsetAsync(za, 0, (size_t)M * C * 4, h->stream));
  const float* pe = nullptr;
  TRY(get_pe(h, Tz, D, 1.0f, &pe));
  float* zc = za; float* zn = zb;
  const int nsteps = (int)h->flow.size();
  for (int s = 0; s < nsteps; ++s) {
    const FlowStep& f = h->flow[s];
    // ActNorm.init: per-channel mean / population std over ALL B*Tz rows (padding included), then the forward
    HIP_TRY(h, hipMemsetAsync(stat, 0, (size_t)4 * C * sizeof(float), h->stream));
    RUN_MISC(h, launch_col_sum(zc, M, C, C, nullptr, mean, h->stream));
    RUN_MISC(h, launch_scale_d(mean, C, 1.0 / (double)M, h->stream));
    RUN_MISC(h, launch_col_sum(zc, M, C, C, mean, sq, h->stream));
    RUN_MISC(h, launch_actnorm_init_finish(mean, sq, M, C, f.an_log_scale, f.an_bias, sc, h->stream));
    RUN_MISC(h, launch_rowop(zc, M, C, sc, f.an_bias, nullptr, 1, 0.f, 0.f, 0u, zc, h->stream));       // z * exp(ls) + b
    if (logprobs) {        // logprobs -= len * sum(log_scale) (the NEW variables, flow.py:166-175) + len * log|det W| (flow.py:127-135)
      RUN_MISC(h, launch_actnorm_inv_params(f.an_log_scale, f.an_bias, C, lsum + 64, lsum + 64 + C, lsum, h->stream));
      RUN_MISC(h, launch_axpy_len_dev(logprobs, z_len, lsum, -1.f, B, h->stream));
      RUN_MISC(h, launch_axpy_len(logprobs, z_len, (float)(-f.lin_logdet), B, h->stream));
    }
    // InvertibleLinear forward z . W on the exact fp32 path (flow.py:123-135)
    RUN_MISC(h, launch_transpose(f.lin_w, C, C, wt, C, h->stream));
    GemmArgs g;
    g.A1 = zc; g.lda1 = C; g.K1 = C; g.K = C; g.Wt = wt; g.ldw = C; g.C = zn; g.ldc = C; g.M = M; g.N = C;
    const bool saved = h->split_scope; h->split_scope = false;
    const int rc = run_gemm(h, g);
    h->split_scope = saved;
    TRY(rc);
    float* dst = zn;
    const bool upper = (s % 2) == 0;
    const int cond_off = upper ? 0 : half, zp_off = upper ? half : 0;
    g = GemmArgs(); g.A1 = dst + cond_off; g.lda1 = C; g.K1 = half; g.K = half; g.Wt = f.pre_wt; g.ldw = half; g.bias = f.pre_b;
    g.pe = pe; g.pe_T = Tz; g.pe_w = f.pos_weight; g.C = xa; g.ldc = D; g.M = M; g.N = D;
    TRY(run_gemm(h, g));
    float* xc = nullptr;
    TRY(run_xstack(h, f.blks, xa, xb, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.prior_attention_heads, c.prior_temperature,
                   nullptr, 0, {Tail{f.heads_wt, C, f.heads_b, heads, C}}, &xc));
    if (f.blks.empty()) {
      g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = f.heads_wt; g.ldw = D; g.bias = f.heads_b;
      g.C = heads; g.ldc = C; g.M = M; g.N = C;
      TRY(run_gemm(h, g));
    }
    RUN_MISC(h, launch_coupling_fwd(heads, dst, M, half, zp_off, logprobs ? rowld : nullptr, h->stream));
    if (logprobs) RUN_MISC(h, launch_masked_row_reduce(rowld, z_len, B, Tz, -1.0f, logprobs, 1, h->stream));
    std::swap(zc, zn);
  }
  HIP_TRY(h, hipMemcpyAsync(z_out, zc, (size_t)M * C * 4, hipMemcpyDeviceToDevice, h->stream));
  return VNR_OK;
}

// TransformerDecoder.call (decoder.py:181-199); kv = decoder cross K|V panel output
// the decoder's pre-chain (pre_projection -> Q|K|V of the first block) on chain panels p_in -> p_mid
std::vector<PreStage> decoder_pre_stages(vnr_handle h, int B, int Tz, float* xa, float* qkv_pre, int p_in, int p_mid) {
  const vnr_config& c = h->cfg;
  const int C = c.latent_dim, D = c.dec_attention_dim;
  const bool saoi = self_aoi_on(h, D, c.dec_attention_heads);
  return {PreStage{h->dec_pre_wt, C, D, p_in, 0, h->dec_pre_b, nullptr, 1, 0.f, xa, D, p_mid},
          PreStage{h->dec_blks[0].qkv_wt, D, 3 * D, p_mid, 0, nullptr, nullptr, 1, 0.f, qkv_pre, 3 * D, -1, saoi ? Tz : 0, B}};
}
// pre_xa / pre_qkv: the pre-chain already ran (behind the prior's last coupling, vnr_inference) into these buffers
int decoder_body(vnr_handle h, const float* z, const float* kv, int kv_ld, const int32_t* z_len,
                 const int32_t* t_len, int B, int Tz, int Tt, int rf, float* initial, float* outputs,
                 float* alignments, float* pre_xa = nullptr, float* pre_qkv = nullptr) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, C = c.latent_dim, D = c.dec_attention_dim, od = c.output_dim;
  if (rf < 1 || rf > c.max_reduction_factor) return fail(h, VNR_ERR_ARG, "reduction_factor out of range");
  float* xa = pre_xa;
  if (!xa) { xa = ws_alloc(h, (size_t)M * D); if (!xa) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed"); }
  WS(xb, (size_t)M * D);
  GemmArgs g;
  // pre_projection -> Q|K|V of the first block as one row-panel chain (two launches otherwise)
  bool fused = pre_xa && pre_qkv;
  float* qkv_pre = pre_qkv;
  if (!fused && !h->dec_blks.empty() && h->dec_blks[0].D == D) {
    const bool saoi = self_aoi_on(h, D, c.dec_attention_heads);
    qkv_pre = ws_alloc(h, qkv_floats(saoi, B, Tz, D));
    if (!qkv_pre) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed");
    TRY(run_prechain(h, z, C, C, M, decoder_pre_stages(h, B, Tz, xa, qkv_pre, 0, 1), &fused));
  }
  if (!fused) {
    qkv_pre = nullptr;
    g.A1 = z; g.lda1 = C; g.K1 = C; g.K = C; g.Wt = h->dec_pre_wt; g.ldw = C; g.bias = h->dec_pre_b; g.C = xa; g.ldc = D; g.M = M; g.N = D;
    TRY(run_gemm(h, g));
  }
  const size_t ali_sz = (size_t)B * c.dec_attention_heads * Tz * Tt;
  // out_projection[:, :, :rf*out_dim] -> reshape [B, Tz*rf, out_dim] (decoder.py:193-195): only the live
  // columns are computed; the [M, rf*od] result IS the reshaped tensor.  It rides on the last block's chain.
  float* init = initial;
  if (!init) { WS(tmp, (size_t)M * rf * od); init = tmp; }
  float* xc = nullptr;
  TRY(run_xstack(h, h->dec_blks, xa, xb, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.dec_attention_heads, c.dec_attention_temperature,
                 alignments, ali_sz, {Tail{h->dec_out_wt, rf * od, h->dec_out_b, init, rf * od}}, &xc, qkv_pre));
  if (h->dec_blks.empty()) {
    g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = h->dec_out_wt; g.ldw = D; g.bias = h->dec_out_b;
    g.C = init; g.ldc = rf * od; g.M = M; g.N = rf * od;
    TRY(run_gemm(h, g));
  }
  const int Tm = Tz * rf, Mm = B * Tm, Fp = c.dec_post_conv_filters;
  WS(pa, (size_t)Mm * Fp); WS(pb, (size_t)Mm * Fp);
  const float* cur = init; float* nxt = pa;
  const int nconv = (int)h->post_convs.size();
  // split rows between the PostNet layers (the first layer reads the fp32 initial outputs: 80 channels are not whole tiles)
  SplitRef rres;
  const bool sr = nconv > 0 && split_rows_ok(h, h->post_convs, Mm) && split_lookup(h, h->dec_res_wt, Fp, od, rres);
  for (int i = 0; i < nconv; ++i) {   // PostNet (utils.py:98-115): tanh x (n-1), identity; BN after act
    TRY(run_conv(h, h->post_convs[i], cur, nullptr, B, Tm, i < nconv - 1 ? ACT_TANH : ACT_IDENTITY, 0, nxt, sr && i > 0, sr));
    cur = nxt; nxt = (nxt == pa) ? pb : pa;
  }
  const int kres = nconv > 0 ? Fp : od;
  g = GemmArgs(); g.A1 = cur; g.lda1 = kres; g.K1 = kres; g.K = kres; g.Wt = h->dec_res_wt; g.ldw = kres; g.bias = h->dec_res_b;
  g.residual = init; g.ldr = od; g.C = outputs; g.ldc = od; g.M = Mm; g.N = od;
  g.a_split = sr;
  TRY(run_gemm(h, g));               // residual_projection + initial_outs (decoder.py:197-198)
  return VNR_OK;
}

int posterior_body(vnr_handle h, const float* mels, const float* kv, int kv_ld, const int32_t* t_len,
                   const int32_t* z_len, int B, int Tz, int Tt, float* mu, float* logvar) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, P = c.post_pre_hidden, C = c.latent_dim;
  WS(xa, (size_t)M * P); WS(xb, (size_t)M * P);
  GemmArgs g;   // PreNet (utils.py:13-18)
  g.A1 = mels; g.lda1 = c.num_mels; g.K1 = c.num_mels; g.K = c.num_mels; g.Wt = h->post_d1_wt; g.ldw = c.num_mels; g.bias = h->post_d1_b;
  g.act = c.post_pre_activation; g.C = xa; g.ldc = P; g.M = M; g.N = P;
  TRY(run_gemm(h, g));
  const float* pe = nullptr;
  TRY(get_pe(h, Tz, P, 1.0f, &pe));
  g = GemmArgs(); g.A1 = xa; g.lda1 = P; g.K1 = P; g.K = P; g.Wt = h->post_d2_wt; g.ldw = P; g.bias = h->post_d2_b; g.act = c.post_pre_activation;
  g.C = xb; g.ldc = P; g.M = M; g.N = P;
  if (!h->training) {
    g.pe = pe; g.pe_T = Tz; g.pe_w = h->post_pos_weight;
    TRY(run_gemm(h, g));             // + pos_weight * PE (posterior.py:120-121)
  } else {
    // training=True: the PreNet's Dropout(0.5) is active after both layers (Keras call context, utils.py:15,17), then
    // + pos_weight * PE and pe_dropout (posterior.py:121-122)
    RUN_MISC(h, launch_rowop(xa, M, P, nullptr, nullptr, nullptr, 1, 0.f, c.post_pre_drop_rate, site_key(h->drop_seed, SITE_POST_PRENET1), xa, h->stream));
    TRY(run_gemm(h, g));
    RUN_MISC(h, launch_rowop(xb, M, P, nullptr, nullptr, nullptr, 1, 0.f, c.post_pre_drop_rate, site_key(h->drop_seed, SITE_POST_PRENET2), xb, h->stream));
    RUN_MISC(h, launch_rowop(xb, M, P, nullptr, nullptr, pe, Tz, h->post_pos_weight, c.post_pos_drop_rate, site_key(h->drop_seed, SITE_POST_PE), xb, h->stream));
  }
  const int D = c.post_attention_dim;
  float* xc = nullptr;
  TRY(run_xstack(h, h->post_blks, xb, xa, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.post_attention_heads, c.post_temperature,
                 nullptr, 0, {Tail{h->post_mu_wt, C, h->post_mu_b, mu, C}, Tail{h->post_lv_wt, C, h->post_lv_b, logvar, C}}, &xc));
  if (h->post_blks.empty()) {
    g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = h->post_mu_wt; g.ldw = D; g.bias = h->post_mu_b; g.C = mu; g.ldc = C; g.M = M; g.N = C;
    TRY(run_gemm(h, g));
    g.Wt = h->post_lv_wt; g.bias = h->post_lv_b; g.C = logvar;
    TRY(run_gemm(h, g));
  }
  return VNR_OK;
}

// TransformerPrior.log_probability (prior.py:119-152); kv = prior cross K|V panel output.  z is consumed (overwritten).
int prior_logprob_body(vnr_handle h, float* z, const int32_t* z_len, const int32_t* t_len, const float* kv, int kv_ld,
                       int B, int Tz, int Tt, float* logprobs) {
  const vnr_config& c = h->cfg;
  const int M = B * Tz, C = c.latent_dim, half = C / 2, D = c.prior_attention_dim;
  if (h->prior_inverse) return prior_inverse_body(h, 1, z_len, t_len, kv, kv_ld, B, Tz, Tt, z, nullptr, logprobs);
  WS(xa, (size_t)M * D); WS(xb, (size_t)M * D); WS(heads, (size_t)M * C); WS(rowld, (size_t)M); WS(zb, (size_t)M * C);
  HIP_TRY(h, hipMemsetAsync(logprobs, 0, (size_t)B * 4, h->stream));
  const float* pe = nullptr;
  TRY(get_pe(h, Tz, D, 1.0f, &pe));
  float* zc = z; float* zn = zb;
  for (int s = (int)h->flow.size() - 1; s >= 0; --s) {
    const FlowStep& f = h->flow[s];
    const bool upper = (s % 2) == 0;
    const int cond_off = upper ? 0 : half, zp_off = upper ? half : 0;
    GemmArgs g;   // coupling net on the conditioning half (flow.py:245-248)
    g.A1 = zc + cond_off; g.lda1 = C; g.K1 = half; g.K = half; g.Wt = f.pre_wt; g.ldw = half; g.bias = f.pre_b;
    g.pe = pe; g.pe_T = Tz; g.pe_w = f.pos_weight; g.C = xa; g.ldc = D; g.M = M; g.N = D;
    TRY(run_gemm(h, g));
    float* xc = nullptr;
    TRY(run_xstack(h, f.blks, xa, xb, kv, kv_ld, z_len, t_len, B, Tz, Tt, c.prior_attention_heads, c.prior_temperature,
                   nullptr, 0, {Tail{f.heads_wt, C, f.heads_b, heads, C}}, &xc));
    if (f.blks.empty()) {
      g = GemmArgs(); g.A1 = xc; g.lda1 = D; g.K1 = D; g.K = D; g.Wt = f.heads_wt; g.ldw = D; g.bias = f.heads_b; g.C = heads; g.ldc = C; g.M = M; g.N = C;
      TRY(run_gemm(h, g));
    }
    RUN_MISC(h, launch_coupling_bwd(heads, zc, M, half, zp_off, rowld, h->stream));
    RUN_MISC(h, launch_masked_row_reduce(rowld, z_len, B, Tz, -1.0f, logprobs, 1, h->stream));    // logdet = -sum log(scale)
    g = GemmArgs(); g.A1 = zc; g.lda1 = C; g.K1 = C; g.K = C; g.Wt = f.inv_wt; g.ldw = C; g.bias = f.inv_b; g.C = zn; g.ldc = C; g.M = M; g.N = C;
    TRY(run_gemm(h, g));                                     // linear.bwd then actnorm.bwd, folded
    RUN_MISC(h, launch_axpy_len(logprobs, z_len, (float)f.inv_logdet_per_frame, B, h->stream));
    std::swap(zc, zn);
  }
  WS(base, (size_t)B);
  RUN_MISC(h, launch_gauss_logprob(zc, z_len, B, Tz, C, base, h->stream));   // -0.5 (log 2pi + eps^2), masked (prior.py:147-150)
  RUN_MISC(h, launch_masked_row_reduce(base, nullptr, B, 1, 1.0f, logprobs, 1, h->stream));
  return VNR_OK;
}

int check_ready(vnr_handle h) {
  if (!h) return fail(nullptr, VNR_ERR_ARG, "null handle");
  if (!h->finalized) return fail(h, VNR_ERR_WEIGHT, "weights not finalized: call vnr_finalize_weights first");
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->packed_stale) { h->packed_stale = false; TRY(vnr_finalize_weights(h)); }
  h->split_scope = true;          // module bodies other than the encoder may use the split-fp16 GEMM path
  return VNR_OK;
}

// ---- range guard (see vnr_context::range_guard) ------------------------------------------------------------------------------------
enum { MOD_ENC = 1, MOD_PRIOR = 2, MOD_DEC = 4, MOD_POST = 8 };
int survey_begin(vnr_handle h) {
  if (!h->survey_words) HIP_TRY(h, hipMalloc((void**)&h->survey_words, (size_t)kSurveyMax * 2 * sizeof(unsigned)));
  std::vector<unsigned> init((size_t)kSurveyMax * 2);
  for (int i = 0; i < kSurveyMax; ++i) { init[2 * i] = 0u; init[2 * i + 1] = 0x7f800000u; }
  HIP_TRY(h, hipMemcpyAsync(h->survey_words, init.data(), init.size() * sizeof(unsigned), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));            // (the host vector goes out of scope)
  h->survey_n = 0; h->survey_kind.clear();
  h->surveying = true;
  return VNR_OK;
}
// verdict: 1 every surveyed tensor maximum inside [kRangeLo, kRangeHi), 2 a Dense / Conv1D input or an attention operand outside
// (exact fp32 products and scaled attention cores from now on)
int survey_end(vnr_handle h, int* verdict) {
  h->surveying = false;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  std::vector<unsigned> w((size_t)h->survey_n * 2);
  if (h->survey_n) HIP_TRY(h, hipMemcpy(w.data(), h->survey_words, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
  float lo = INFINITY, hi = 0.f; int v = 1;
  for (int i = 0; i < h->survey_n; ++i) {
    float mx; memcpy(&mx, &w[2 * i], 4);
    if (mx == 0.f) continue;                               // an all-zero matrix (temperature 0: the noise) splits exactly
    lo = fminf(lo, mx); hi = fmaxf(hi, mx);
    if (!(mx >= kRangeLo && mx < kRangeHi)) v = 2;         // (an attention operand too: round 6 gave the exact mode's cores per-launch operand scales)
  }
  h->range_lo = (lo == INFINITY) ? 0.f : lo; h->range_hi = hi;
  h->range_surveys++;
  *verdict = v;
  return VNR_OK;
}
// The one range statement that holds for EVERY input: a LayerNormalization output obeys |y_k| <= |gamma_k| sqrt(n) + |beta_k|
// (the normalised row has 2-norm <= sqrt(n)).  Two epilogue paths of the one-wave-per-SIMD chain kernel multiply such outputs without
// a sentinel probe of their own (gemm3c.hip) -- so a module whose LayerNorm variables would allow an output beyond fp16's range is kept
// on exact fp32 from the start (state 2), whatever the survey samples.  Runs inside vnr_finalize_weights (a few hundred small copies).
int ln_static_bound(vnr_handle h) {
  std::vector<float> ga, be;
  for (auto& kv : h->w) {
    const std::string& p = kv.first;
    const size_t cut = p.rfind("/gamma");
    if (cut == std::string::npos || cut + 6 != p.size()) continue;
    const size_t prev = p.rfind('/', cut - 1);
    if (p.compare(prev == std::string::npos ? 0 : prev + 1, 10, "layer_norm") != 0) continue;      // (not the BatchNormalization gammas)
    const Tensor& g = kv.second;
    const Tensor* b = find_w(h, p.substr(0, cut) + "/beta");
    if (!g.d || g.n <= 0) continue;
    ga.resize((size_t)g.n); be.assign((size_t)g.n, 0.f);
    HIP_TRY(h, hipMemcpy(ga.data(), g.d, (size_t)g.n * sizeof(float), hipMemcpyDeviceToHost));
    if (b && b->d && b->n == g.n) HIP_TRY(h, hipMemcpy(be.data(), b->d, (size_t)g.n * sizeof(float), hipMemcpyDeviceToHost));
    double bound = 0.0; bool finite = true;
    const double rn = sqrt((double)g.n);
    for (int64_t i = 0; i < g.n; ++i) {
      const double v = fabs((double)ga[(size_t)i]) * rn + fabs((double)be[(size_t)i]);
      if (!(v < INFINITY)) finite = false;
      if (v > bound) bound = v;
    }
    if (finite && bound < 65504.0) continue;
    const int m = p.compare(0, 13, "text_encoder/") == 0 ? 0 : p.compare(0, 6, "prior/") == 0 ? 1 : p.compare(0, 8, "decoder/") == 0 ? 2 :
                  p.compare(0, 10, "posterior/") == 0 ? 3 : -1;
    if (m >= 0) h->range_state[m] = 2;
  }
  return VNR_OK;
}

// ---- range sentinel: checkpoints (see vnr_context::range_flag) --------------------------------------------------------------------
// Called behind a stream synchronisation.  Raised word: everything since the previous checkpoint is suspect -> VNR_ERR_RANGE, the
// modules that ran on the split path since then move to exact fp32 (they stay there until their weights change).
int range_checkpoint(vnr_handle h) {
  if (!h->range_flag) return VNR_OK;
  const unsigned mods = h->mods_pending;
  h->mods_pending = 0;
  if (!*reinterpret_cast<volatile unsigned*>(h->range_flag)) return VNR_OK;
  *reinterpret_cast<volatile unsigned*>(h->range_flag) = 0u;
  h->range_trips++;
  for (int m = 0; m < 4; ++m) if (mods & (1u << m)) h->range_state[m] = 2;
  return fail(h, VNR_ERR_RANGE, "an activation left the fp16 range (|x| >= 65504) inside a split-fp16 product since the last synchronisation point: the "
              "results computed since then are invalid.  The modules involved now run on exact fp32 (products on fp32 MFMA, attention cores with "
              "per-launch operand scales): issue those calls again");
}
// has the word been raised since it was last cleared?  (synchronous users: the stream is idle)  Clears it.
bool range_tripped(vnr_handle h) {
  if (!h->range_flag || !*reinterpret_cast<volatile unsigned*>(h->range_flag)) return false;
  *reinterpret_cast<volatile unsigned*>(h->range_flag) = 0u;
  h->range_trips++;
  return true;
}
int bn_backup_prepare(vnr_handle h) {
  if (h->bn_table_gen == h->w_generation) return VNR_OK;
  if (h->bn_save_jobs) { hipFree(h->bn_save_jobs); h->bn_save_jobs = nullptr; }
  if (h->bn_restore_jobs) { hipFree(h->bn_restore_jobs); h->bn_restore_jobs = nullptr; }
  if (h->bn_backup) { hipFree(h->bn_backup); h->bn_backup = nullptr; }
  std::vector<CopyJob> save, restore;
  auto is_moving = [](const std::string& p) {
    return (p.size() > 12 && p.compare(p.size() - 12, 12, "/moving_mean") == 0) || (p.size() > 16 && p.compare(p.size() - 16, 16, "/moving_variance") == 0);
  };
  size_t total = 0;
  for (auto& kv : h->w) if (is_moving(kv.first) && kv.second.d && kv.second.n > 0) total += (size_t)kv.second.n;
  h->bn_njobs = 0;
  if (total) {
    HIP_TRY(h, hipMalloc((void**)&h->bn_backup, total * sizeof(float)));
    size_t off = 0;
    for (auto& kv : h->w) {
      if (!is_moving(kv.first) || !kv.second.d || kv.second.n <= 0) continue;
      save.push_back({kv.second.d, h->bn_backup + off, (long long)kv.second.n});
      restore.push_back({h->bn_backup + off, kv.second.d, (long long)kv.second.n});
      off += (size_t)kv.second.n;
    }
    HIP_TRY(h, hipMalloc((void**)&h->bn_save_jobs, save.size() * sizeof(CopyJob)));
    HIP_TRY(h, hipMalloc((void**)&h->bn_restore_jobs, restore.size() * sizeof(CopyJob)));
    HIP_TRY(h, hipMemcpy(h->bn_save_jobs, save.data(), save.size() * sizeof(CopyJob), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->bn_restore_jobs, restore.data(), restore.size() * sizeof(CopyJob), hipMemcpyHostToDevice));
    h->bn_njobs = (int)save.size();
  }
  h->bn_table_gen = h->w_generation;
  return VNR_OK;
}
int bn_backup_save(vnr_handle h) {
  TRY(bn_backup_prepare(h));
  if (!h->bn_njobs) return VNR_OK;
  h->launches++;
  const hipError_t e = launch_copy_batch(h->bn_save_jobs, h->bn_njobs, h->stream);
  return e == hipSuccess ? VNR_OK : fail(h, VNR_ERR_HIP, std::string("saving the moving statistics: ") + hipGetErrorString(e));
}
int bn_backup_restore(vnr_handle h) {
  if (!h->bn_njobs || h->bn_table_gen != h->w_generation) return VNR_OK;
  h->launches++;
  const hipError_t e = launch_copy_batch(h->bn_restore_jobs, h->bn_njobs, h->stream);
  return e == hipSuccess ? VNR_OK : fail(h, VNR_ERR_HIP, std::string("restoring the moving statistics: ") + hipGetErrorString(e));
}
// A call that changes variables (Dropout / batch-statistics forwards with their moving update, vnr_init): it cannot be replayed by the
// caller, so it checks itself -- synchronise, look at the word, and on a trip run once more on exact fp32 (the BatchNormalization moving
// update of the first pass was predicated on the word: the variables see one update).
template <class F> int range_checked_sync(vnr_handle h, F body) {
  if (!h->range_sentinel || !h->range_flag || !h->split_enabled) return body();
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (range_checkpoint(h) != VNR_OK) { /* an earlier asynchronous call tripped: reported at ITS checkpoint below */ return VNR_ERR_RANGE; }
  TRY(bn_backup_save(h));
  int rc = body();
  if (rc != VNR_OK) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (!range_tripped(h)) return VNR_OK;
  TRY(bn_backup_restore(h));          // the layers in front of the trip have taken their moving update already: the repeat starts from the saved statistics
  h->split_suspended = true;
  rc = body();
  h->split_suspended = false;
  if (rc != VNR_OK) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (range_tripped(h)) return fail(h, VNR_ERR_RANGE, "non-finite values on the exact-fp32 path as well: the inputs or the variables hold NaN / inf");
  return VNR_OK;
}
template <class F> int range_guarded(vnr_handle h, unsigned mods, F body) {
  if (!h) return body();
  TRY(check_ready(h));                                     // (a stale pack is rebuilt here: that resets the states before they are read)
  h->split_suspended = false;
  if (h->in_train_step) return body();                     // (vnr_train_step checks and repeats itself)
  if (h->training) return range_checked_sync(h, body);     // a forward that updates variables: synchronous check, no survey
  if (!h->split_enabled) return body();
  bool any_exact = false, any_unknown = false;
  for (int m = 0; m < 4; ++m) if (mods & (1u << m)) { any_exact |= h->range_state[m] == 2; any_unknown |= h->range_state[m] == 0; }
  // "range_guard" = 0 switches the SURVEY off; a state 2 that the sentinel or the static LayerNorm bound has set is honoured either way
  if (any_exact && (!any_unknown || !h->range_guard)) { h->split_suspended = true; const int rc = body(); h->split_suspended = false; return rc; }
  if (!h->range_guard || (!any_exact && !any_unknown)) { h->mods_pending |= mods; return body(); }
  TRY(survey_begin(h));
  h->split_suspended = true;
  const int rc = body();
  h->split_suspended = false;
  int verdict = 1;
  const int rc2 = survey_end(h, &verdict);
  if (rc != VNR_OK) return rc;
  if (rc2 != VNR_OK) return rc2;
  for (int m = 0; m < 4; ++m) if (mods & (1u << m)) h->range_state[m] = (any_exact || verdict == 2) ? 2 : 1;
  if (verdict == 1 && !any_exact) { h->mods_pending |= mods; return body(); }   // in window: the call returns what every later call returns (split path)
  return VNR_OK;                                           // exact fp32 results stand
}

#include "train.inc"

}  // namespace

// =====================================================================================================
extern "C" {

int vnr_abi_version(void) { return VNR_ABI_VERSION; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host: the checksum of the reference's on-disk
// formats (TFRecord framing datasets/tf_record_utils.py:77-83, tensor-bundle checkpoints train.py:246-249).  Host only.
uint32_t vnr_crc32c(uint32_t crc, const void* data, size_t n) {
  static uint32_t table[8][256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xff];
    ready = true;
  }
  const unsigned char* p = static_cast<const unsigned char*>(data);
  uint32_t c = ~crc;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xff] ^ table[6][(lo >> 8) & 0xff] ^ table[5][(lo >> 16) & 0xff] ^ table[4][lo >> 24] ^
        table[3][hi & 0xff] ^ table[2][(hi >> 8) & 0xff] ^ table[1][(hi >> 16) & 0xff] ^ table[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = (c >> 8) ^ table[0][(c ^ *p++) & 0xff];
  return ~c;
}

const char* vnr_last_error(vnr_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int vnr_device_count(int* count) {
  if (!count) return fail(nullptr, VNR_ERR_ARG, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(nullptr, VNR_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *count = n;
  return VNR_OK;
}

int vnr_create(const vnr_config* cfg, int device, vnr_handle* out) {
  if (!cfg || !out) return fail(nullptr, VNR_ERR_ARG, "null argument");
  if (cfg->abi_version != VNR_ABI_VERSION) return fail(nullptr, VNR_ERR_ARG, "vnr_config.abi_version mismatch");
  auto bad = [&](const char* m) { return fail(nullptr, VNR_ERR_ARG, std::string("unsupported configuration: ") + m); };
  if (cfg->enc_attention_dim != 64 * cfg->enc_attention_heads || cfg->dec_attention_dim != 64 * cfg->dec_attention_heads ||
      cfg->prior_attention_dim != 64 * cfg->prior_attention_heads || cfg->post_attention_dim != 64 * cfg->post_attention_heads)
    return bad("attention head width must be 64");
  if ((cfg->latent_dim & 7) || (cfg->output_dim & 3) || (cfg->num_mels & 3) || (cfg->enc_embd_dim & 3) || (cfg->enc_pre_hidden & 3) ||
      (cfg->enc_ffn_hidden & 3) || (cfg->dec_ffn_hidden & 3) || (cfg->prior_ffn_hidden & 3) || (cfg->post_ffn_hidden & 3) ||
      (cfg->dec_post_conv_filters & 3) || (cfg->post_pre_hidden & 3))
    return bad("channel counts must be multiples of 4 (latent_dim of 8)");
  if (cfg->enc_n_conv < 1) return bad("the encoder prenet needs at least one conv layer");
  // tf.concat([x, ctx], -1) -> Dense (attention.py:410-412,440-449) runs as two K panels behind one descriptor: the panel switch must
  // fall on a 32-column k-tile boundary (gemm2.hip; the register-staged kernel that took any multiple of 4 was retired in round 3)
  if (cfg->enc_pre_hidden & 31) return bad("enc_pre_hidden must be a multiple of 32 (first panel of the encoder blocks' concat -> att_proj)");
  if (!(cfg->enc_conv_kernel & 1) || !(cfg->dec_post_conv_kernel & 1)) return bad("conv kernels must be odd");
  if (cfg->post_pre_hidden != cfg->post_attention_dim) return bad("posterior pre_hidden must equal attention_dim");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, VNR_ERR_HIP, "no HIP device available (libvaenar_hip needs an AMD GPU)");
  if (device < 0 || device >= n) return fail(nullptr, VNR_ERR_ARG, "device index out of range");
  vnr_handle h = new vnr_context();
  h->cfg = *cfg;
  h->device = device;
  if (const char* e = getenv("VNR_CHAIN_ROWS64")) h->chain_rows64 = atoi(e) != 0;
  if (const char* e = getenv("VNR_CHAIN_WAVES4")) h->chain_waves4 = atoi(e) != 0;
  if (const char* e = getenv("VNR_CHAIN_SEGMENTS")) h->chain_segments = atoi(e) != 0;
  if (const char* e = getenv("VNR_CHAIN_PREFETCH")) h->chain_prefetch = atoi(e) != 0;      // test / measurement override of the option's default
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(nullptr, VNR_ERR_HIP, "hipSetDevice / hipStreamCreate failed");
  }
  if (hipMalloc((void**)&h->chain_progress, 1024) != hipSuccess || hipMemset(h->chain_progress, 0, 1024) != hipSuccess) {
    if (h->chain_progress) hipFree(h->chain_progress);
    hipStreamDestroy(h->stream);
    delete h;
    return fail(nullptr, VNR_ERR_NOMEM, "hipMalloc of the chain kernels' pacing words failed");
  }
  // the range sentinel's word: host-pinned (the kernels store through the unified address space, the host reads it behind a
  // synchronisation without a copy) + the device word the training step's Adam launch reads
  if (hipHostMalloc((void**)&h->range_flag, 64, hipHostMallocDefault) != hipSuccess) h->range_flag = nullptr;
  if (h->range_flag) *h->range_flag = 0u;
  if (hipMalloc((void**)&h->d_step_flag, 64) != hipSuccess || hipMemset(h->d_step_flag, 0, 64) != hipSuccess) {
    if (h->range_flag) hipHostFree(h->range_flag);
    hipFree(h->chain_progress);
    hipStreamDestroy(h->stream);
    delete h;
    return fail(nullptr, VNR_ERR_NOMEM, "hipMalloc of the range sentinel's word failed");
  }
  if (getenv("VNR_TRAIN_FP32")) h->train_fp32 = true;       // A/B switch: exact fp32 MFMA GEMMs in the training step
  *out = h;
  return VNR_OK;
}

int vnr_destroy(vnr_handle h) {
  if (!h) return VNR_OK;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->comm) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  train_free(h);
  h->det.release();
  for (auto& kv : h->w) hipFree(kv.second.d);
  for (auto p : h->packed_allocs) hipFree(p);
  for (auto& c : h->chunks) hipFree(c.p);
  for (auto& kv : h->pe_cache) hipFree(kv.second);
  for (auto& kv : h->voc_tables) hipFree(kv.second);
  if (h->survey_words) hipFree(h->survey_words);
  if (h->chain_progress) hipFree(h->chain_progress);
  if (h->range_flag) hipHostFree(h->range_flag);
  if (h->d_step_flag) hipFree(h->d_step_flag);
  if (h->amax_words) hipFree(h->amax_words);
  if (h->kv_stream) { hipStreamSynchronize(h->kv_stream); hipStreamDestroy(h->kv_stream); }
  if (h->kv_fork) hipEventDestroy(h->kv_fork);
  if (h->kv_done) hipEventDestroy(h->kv_done);
  if (h->bn_save_jobs) hipFree(h->bn_save_jobs);
  if (h->bn_restore_jobs) hipFree(h->bn_restore_jobs);
  if (h->bn_backup) hipFree(h->bn_backup);
  for (auto& r : h->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  for (auto e : h->event_pool) hipEventDestroy(e);
  hipStreamDestroy(h->stream);
  delete h;
  return VNR_OK;
}

int vnr_device_info(vnr_handle h, char* name, int name_len, int* compute_units, int* wavefront) {
  if (!h) return fail(nullptr, VNR_ERR_ARG, "null handle");
  hipDeviceProp_t p;
  HIP_TRY(h, hipGetDeviceProperties(&p, h->device));
  if (name && name_len > 0) { snprintf(name, name_len, "%s (%s)", p.name[0] ? p.name : "AMD GPU", p.gcnArchName); }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (wavefront) *wavefront = p.warpSize;
  return VNR_OK;
}

int vnr_malloc(vnr_handle h, size_t bytes, void** d_ptr) {
  if (!h || !d_ptr) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMalloc(d_ptr, bytes ? bytes : 4));
  return VNR_OK;
}
int vnr_free(vnr_handle h, void* d_ptr) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  if (!d_ptr) return VNR_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipFree(d_ptr));
  return VNR_OK;
}
int vnr_memcpy_h2d(vnr_handle h, void* d, const void* s, size_t n) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  // a checkpoint of the range sentinel BEFORE the copy: what a flagged call read must still be there when the caller repeats it
  if (h->mods_pending || (h->range_flag && *reinterpret_cast<volatile unsigned*>(h->range_flag))) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    TRY(range_checkpoint(h));
  }
  HIP_TRY(h, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));   // pageable host memory may be reused by the caller
  return VNR_OK;
}
int vnr_memcpy_d2h(vnr_handle h, void* dst, const void* s, size_t n) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(dst, s, n, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return range_checkpoint(h);                     // (range sentinel: the bytes just copied are not to be trusted when the word is raised)
}
int vnr_memcpy_d2d(vnr_handle h, void* dst, const void* s, size_t n) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(dst, s, n, hipMemcpyDeviceToDevice, h->stream));
  return VNR_OK;
}
int vnr_memset(vnr_handle h, void* d, int value, size_t n) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemsetAsync(d, value, n, h->stream));
  return VNR_OK;
}
int vnr_synchronize(vnr_handle h) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return range_checkpoint(h);
}

int vnr_set_weight(vnr_handle h, const char* path, const float* host, const int64_t* shape, int ndim) {
  if (!h || !path || !host || ndim < 0 || ndim > 4 || (ndim > 0 && !shape)) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  int64_t n = 1;
  std::vector<int64_t> shp;
  for (int i = 0; i < ndim; ++i) { if (shape[i] <= 0) return fail(h, VNR_ERR_ARG, "bad shape"); n *= shape[i]; shp.push_back(shape[i]); }
  // optimizer state and gradient tables point into the weight store: they survive an in-place update of an existing variable
  // (a checkpoint restore), but not a new or resized one
  bool in_place = false;
  { auto it = h->w.find(path); in_place = it != h->w.end() && it->second.d && it->second.n == n && it->second.shape == shp; }
  if (h->train && !in_place) train_free(h);
  Tensor& t = h->w[path];
  if (t.d && t.n != n) { hipFree(t.d); t.d = nullptr; }
  if (!t.d) h->w_generation++;
  if (!t.d) HIP_TRY(h, hipMalloc((void**)&t.d, (size_t)n * sizeof(float)));
  t.n = n; t.shape = shp;
  if (n == 1) t.scalar = host[0];
  HIP_TRY(h, hipMemcpyAsync(t.d, host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // a variable of the SAME shape updated in place on a finalized engine (tf.Variable.assign, model.trainable_variables): every
  // pointer of the packed model stays valid, the packed panels are rebuilt lazily by the next call (check_ready), like after an
  // optimizer step.  A new or resized variable needs an explicit vnr_finalize_weights.
  if (in_place && h->finalized) h->packed_stale = true;
  else h->finalized = false;
  h->derived_fresh = false;
  return VNR_OK;
}

int vnr_get_weight(vnr_handle h, const char* path, float* host, int64_t count) {
  if (!h || !path || !host) return fail(h, VNR_ERR_ARG, "bad argument");
  const Tensor* t = find_w(h, path);
  if (!t) return fail(h, VNR_ERR_WEIGHT, std::string("unknown weight ") + path);
  if (count != t->n) return fail(h, VNR_ERR_WEIGHT, std::string("element count mismatch for ") + path);
  return vnr_memcpy_d2h(h, host, t->d, (size_t)count * sizeof(float));
}

int vnr_finalize_weights(vnr_handle h) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->packed_stale = false;
  for (int m = 0; m < 4; ++m) h->range_state[m] = 0;      // activation ranges are a property of the weights: surveyed again on the next call
  TRY(ln_static_bound(h));                                // ... except where the variables alone already say "outside"
  for (auto& kv : h->w) if (kv.second.n == 1 && kv.second.d) HIP_TRY(h, hipMemcpy(&kv.second.scalar, kv.second.d, sizeof(float), hipMemcpyDeviceToHost));
  for (auto p : h->packed_allocs) hipFree(p);
  h->packed_allocs.clear();
  for (auto p : h->split_allocs) hipFree(p);
  h->split_allocs.clear(); h->split_panels.clear(); h->panel_registry.clear(); h->chain_prm.clear();
  h->enc_convs.clear(); h->post_convs.clear(); h->enc_blks.clear(); h->flow.clear(); h->dec_blks.clear(); h->post_blks.clear();
  const vnr_config& c = h->cfg;
  Packer P{h};
  const int Dm = c.enc_pre_hidden, A = c.enc_attention_dim;
  // ---- text encoder --------------------------------------------------------------------------------
  h->emb = P.raw("text_encoder/emb_layer/embeddings", {c.enc_vocab_size, c.enc_embd_dim});
  h->enc_pos_weight = P.scalar("text_encoder/pos_weight");
  int cin = c.enc_embd_dim;
  for (int i = 0; i < c.enc_n_conv; ++i) {
    ConvL L; pack_conv(P, "text_encoder/prenet/conv_stack/" + std::to_string(i), c.enc_conv_kernel, cin, Dm, L);
    L.drop_rate = c.enc_pre_drop_rate; L.site = SITE_ENC_CONV + i;
    h->enc_convs.push_back(L); cin = Dm;
  }
  h->enc_proj_wt = P.wt("text_encoder/prenet/projection/kernel", Dm, Dm);
  h->enc_proj_b = P.raw("text_encoder/prenet/projection/bias", {Dm});
  for (int i = 0; i < c.enc_n_blk; ++i) {
    const std::string p = "text_encoder/self_attentions/" + std::to_string(i);
    SBlk k;
    float* qkv = P.alloc((size_t)3 * A * Dm);
    P.transpose_into(P.raw(p + "/attention/query_layer/kernel", {Dm, A}), Dm, A, qkv, 0);
    P.transpose_into(P.raw(p + "/attention/key_layer/kernel", {Dm, A}), Dm, A, qkv, A);
    P.transpose_into(P.raw(p + "/attention/value_layer/kernel", {Dm, A}), Dm, A, qkv, 2 * A);
    k.qkv_wt = qkv;
    P.reg(qkv, 3 * A, Dm);
    k.proj_wt = P.wt(p + "/att_proj/kernel", Dm + A, Dm);
    k.proj_b = P.raw(p + "/att_proj/bias", {Dm});
    k.ln_g = P.raw(p + "/layer_norm/gamma", {Dm}); k.ln_b = P.raw(p + "/layer_norm/beta", {Dm});
    k.ffn1_wt = P.wt(p + "/ffn/dense1/kernel", Dm, c.enc_ffn_hidden); k.ffn1_b = P.raw(p + "/ffn/dense1/bias", {c.enc_ffn_hidden});
    k.ffn2_wt = P.wt(p + "/ffn/dense2/kernel", c.enc_ffn_hidden, Dm); k.ffn2_b = P.raw(p + "/ffn/dense2/bias", {Dm});
    k.ffn_g = P.raw(p + "/ffn/layer_norm/gamma", {Dm}); k.ffn_b = P.raw(p + "/ffn/layer_norm/beta", {Dm});
    h->enc_blks.push_back(k);
  }
  h->lp_w = P.raw("length_predictor/projection/kernel", {Dm, 1});
  h->lp_b = P.raw("length_predictor/projection/bias", {1});
  // ---- prior + decoder: one stacked cross-attention K|V panel (prior blocks first, decoder after) ------
  const int C = c.latent_dim, half = C / 2, Dp = c.prior_attention_dim, Dd = c.dec_attention_dim;
  h->prior_kv_n = c.prior_n_blk * c.prior_n_transformer_blk * 2 * Dp;
  h->dec_kv_n = c.dec_nblk * 2 * Dd;
  float* kv_panel = P.alloc((size_t)(h->prior_kv_n + h->dec_kv_n) * Dm);
  P.reg(kv_panel, h->prior_kv_n + h->dec_kv_n, Dm);
  h->prior_kv_wt = kv_panel;
  h->dec_kv_wt = kv_panel ? kv_panel + (size_t)h->prior_kv_n * Dm : nullptr;
  std::vector<double> Wh((size_t)C * C);
  std::vector<float> Wf((size_t)C * C), lsf(C);
  for (int s = 0; s < c.prior_n_blk; ++s) {
    const std::string p = "prior/glow/" + std::to_string(s);
    FlowStep f;
    const float* ls = P.raw(p + "/0/log_scale", {C});
    const float* ab = P.raw(p + "/0/bias", {C});
    const float* W = P.raw(p + "/1/weight", {C, C});
    float* fw = P.alloc((size_t)C * C); float* fb = P.alloc(C);
    if (P.rc == VNR_OK) {
      if (launch_fold_actnorm_linear(ls, ab, W, C, fw, fb, h->stream) != hipSuccess) { P.rc = VNR_ERR_HIP; P.missing = "fold launch"; }
      if (hipMemcpy(Wf.data(), W, (size_t)C * C * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(lsf.data(), ls, (size_t)C * 4, hipMemcpyDeviceToHost) != hipSuccess) { P.rc = VNR_ERR_HIP; P.missing = "d2h for slogdet"; }
      for (size_t i = 0; i < Wf.size(); ++i) Wh[i] = (double)Wf[i];
      float lssum = 0.f;   // tf.reduce_sum(log_scale) in fp32 (flow.py:168)
      for (int i = 0; i < C; ++i) lssum += lsf[i];
      f.lin_logdet = (double)(float)slogdet_abs(Wh, C);
      f.logdet_per_frame = (double)lssum + f.lin_logdet;   // cast to fp32, flow.py:127-129
      // backward direction (prior.log_probability): eps = (z . inv(W) - b) / (exp(ls) + 1e-8)  (flow.py:137-150, 177-187)
      std::vector<double> Winv;
      std::vector<float> abf(C), wti((size_t)C * C), bi(C);
      if (hipMemcpy(abf.data(), ab, (size_t)C * 4, hipMemcpyDeviceToHost) != hipSuccess) { P.rc = VNR_ERR_HIP; P.missing = "d2h actnorm bias"; }
      if (!invert_matrix(Wh, C, Winv)) { P.rc = VNR_ERR_WEIGHT; P.missing = "singular invertible-linear weight " + p + "/1/weight"; }
      else {
        for (int n = 0; n < C; ++n) {
          const double d = 1.0 / ((double)expf(lsf[n]) + 1e-8);
          for (int k = 0; k < C; ++k) wti[(size_t)n * C + k] = (float)((double)(float)Winv[(size_t)k * C + n] * d);   // inv(W) rounded to fp32 like tf.linalg.inv
          bi[n] = (float)(-(double)abf[n] * d);
        }
        float* iw = P.alloc((size_t)C * C); float* ib = P.alloc(C);
        if (iw && ib && (hipMemcpy(iw, wti.data(), (size_t)C * C * 4, hipMemcpyHostToDevice) != hipSuccess ||
                         hipMemcpy(ib, bi.data(), (size_t)C * 4, hipMemcpyHostToDevice) != hipSuccess)) { P.rc = VNR_ERR_HIP; P.missing = "h2d inverse fold"; }
        f.inv_wt = iw; f.inv_b = ib;
        P.reg(iw, C, C);
        f.inv_logdet_per_frame = -(double)lssum + (double)(float)slogdet_abs(Winv, C);
      }
    }
    f.fold_wt = fw; f.fold_b = fb;
    f.an_log_scale = const_cast<float*>(ls); f.an_bias = const_cast<float*>(ab); f.lin_w = W;
    P.reg(fw, C, C);
    f.pos_weight = P.scalar(p + "/2/net/pos_weight");
    f.pre_wt = P.wt(p + "/2/net/pre_projection/kernel", half, Dp);
    f.pre_b = P.raw(p + "/2/net/pre_projection/bias", {Dp});
    {
      float* px = P.alloc((size_t)Dp * C);
      const int cond_off = (s % 2) == 0 ? 0 : half;         // prior.py:85-87, flow.py:227-228
      if (px && f.pre_wt && P.rc == VNR_OK &&
          (hipMemsetAsync(px, 0, (size_t)Dp * C * sizeof(float), h->stream) != hipSuccess ||
           hipMemcpy2DAsync(px + cond_off, (size_t)C * sizeof(float), f.pre_wt, (size_t)half * sizeof(float), (size_t)half * sizeof(float), Dp,
                            hipMemcpyDeviceToDevice, h->stream) != hipSuccess)) { P.rc = VNR_ERR_HIP; P.missing = "expanded pre_projection panel"; }
      f.pre_wt_x = px;
      P.reg(px, Dp, C);
    }
    float* hw = P.alloc((size_t)C * Dp); float* hb = P.alloc(C);
    P.transpose_into(P.raw(p + "/2/net/log_scale_proj/kernel", {Dp, half}), Dp, half, hw, 0);
    P.transpose_into(P.raw(p + "/2/net/shift_proj/kernel", {Dp, half}), Dp, half, hw, half);
    P.copy_into(P.raw(p + "/2/net/log_scale_proj/bias", {half}), hb, half);
    P.copy_into(P.raw(p + "/2/net/shift_proj/bias", {half}), hb ? hb + half : nullptr, half);
    f.heads_wt = hw; f.heads_b = hb;
    P.reg(hw, C, Dp);
    for (int b = 0; b < c.prior_n_transformer_blk; ++b) {
      XBlk k;
      pack_xblk(P, p + "/2/net/attentions/" + std::to_string(b), Dp, Dm, c.prior_ffn_hidden, kv_panel,
                (s * c.prior_n_transformer_blk + b) * 2 * Dp, k);
      f.blks.push_back(k);
    }
    h->flow.push_back(f);
  }
  h->dec_pre_wt = P.wt("decoder/pre_projection/kernel", C, Dd);
  h->dec_pre_b = P.raw("decoder/pre_projection/bias", {Dd});
  for (int b = 0; b < c.dec_nblk; ++b) {
    XBlk k;
    pack_xblk(P, "decoder/attentions/" + std::to_string(b), Dd, Dm, c.dec_ffn_hidden, kv_panel, h->prior_kv_n + b * 2 * Dd, k);
    k.kv_col = b * 2 * Dd;   // relative to the decoder panel; vnr_inference adds prior_kv_n
    h->dec_blks.push_back(k);
  }
  const int od = c.output_dim, nout = od * c.max_reduction_factor;
  h->dec_out_wt = P.wt("decoder/out_projection/kernel", Dd, nout);
  h->dec_out_b = P.raw("decoder/out_projection/bias", {nout});
  cin = od;
  for (int i = 0; i < c.dec_post_n_conv; ++i) {
    ConvL L; pack_conv(P, "decoder/postnet/conv_stack/" + std::to_string(i), c.dec_post_conv_kernel, cin, c.dec_post_conv_filters, L);
    L.drop_rate = c.dec_post_drop_rate; L.site = SITE_POSTNET_CONV + i;
    h->post_convs.push_back(L); cin = c.dec_post_conv_filters;
  }
  h->dec_res_wt = P.wt("decoder/residual_projection/kernel", cin, od);
  h->dec_res_b = P.raw("decoder/residual_projection/bias", {od});
  // ---- posterior (optional: only present for training / ELBO evaluation) ---------------------------------
  h->has_posterior = find_w(h, "posterior/pos_weight") != nullptr;
  if (h->has_posterior) {
    const int Pq = c.post_pre_hidden, Dq = c.post_attention_dim;
    h->post_pos_weight = P.scalar("posterior/pos_weight");
    h->post_d1_wt = P.wt("posterior/prenet/dense1/kernel", c.num_mels, Pq); h->post_d1_b = P.raw("posterior/prenet/dense1/bias", {Pq});
    h->post_d2_wt = P.wt("posterior/prenet/dense2/kernel", Pq, Pq); h->post_d2_b = P.raw("posterior/prenet/dense2/bias", {Pq});
    h->post_kv_n = c.post_nblk * 2 * Dq;
    float* pkv = P.alloc((size_t)h->post_kv_n * Dm);
    h->post_kv_wt = pkv;
    P.reg(pkv, h->post_kv_n, Dm);
    for (int b = 0; b < c.post_nblk; ++b) {
      XBlk k; pack_xblk(P, "posterior/attentions/" + std::to_string(b), Dq, Dm, c.post_ffn_hidden, pkv, b * 2 * Dq, k);
      h->post_blks.push_back(k);
    }
    h->post_mu_wt = P.wt("posterior/mu_projection/kernel", Dq, C); h->post_mu_b = P.raw("posterior/mu_projection/bias", {C});
    h->post_lv_wt = P.wt("posterior/logvar_projection/kernel", Dq, C); h->post_lv_b = P.raw("posterior/logvar_projection/bias", {C});
  }
  if (P.rc != VNR_OK) return fail(h, P.rc, P.missing);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // ---- split-fp16 images: w * 2^s = hi + lo (fp16 pair), s per panel so that max|w| * 2^s <= 2^14 ----------------------
  {
    unsigned* dmax = nullptr;
    HIP_TRY(h, hipMalloc((void**)&dmax, h->panel_registry.size() * sizeof(unsigned) + 4));
    HIP_TRY(h, hipMemsetAsync(dmax, 0, h->panel_registry.size() * sizeof(unsigned) + 4, h->stream));
    for (size_t i = 0; i < h->panel_registry.size(); ++i) {
      auto& e = h->panel_registry[i];
      HIP_TRY(h, launch_absmax(e.first, (size_t)e.second.first * e.second.second, dmax + i, h->stream));
    }
    std::vector<unsigned> hmax(h->panel_registry.size() + 1);
    HIP_TRY(h, hipMemcpyAsync(hmax.data(), dmax, h->panel_registry.size() * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(dmax));
    for (size_t i = 0; i < h->panel_registry.size(); ++i) {
      auto& e = h->panel_registry[i];
      const int N = e.second.first, K = e.second.second;
      float mx; memcpy(&mx, &hmax[i], 4);
      int sexp = 8;
      if (mx > 0.f && std::isfinite(mx)) { sexp = (int)floor(log2(16384.0 / (double)mx)); if (sexp > 24) sexp = 24; if (sexp < -24) sexp = -24; }
      void* img = nullptr;
      HIP_TRY(h, hipMalloc(&img, (size_t)N * ((K + 31) / 32) * 128));
      h->split_allocs.push_back(img);
      HIP_TRY(h, launch_split_weights(e.first, N, K, (float)ldexp(1.0, sexp), img, h->stream));
      void* opm = nullptr;
      HIP_TRY(h, hipMalloc(&opm, (size_t)((N + 31) / 32) * ((K + 31) / 32) * 4096));
      h->split_allocs.push_back(opm);
      HIP_TRY(h, launch_opmajor_weights(e.first, N, K, (float)ldexp(1.0, sexp), opm, h->stream));
      h->split_panels[e.first] = {N, K, img, (float)ldexp(1.0, -sexp), opm};
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  h->finalized = true;
  return VNR_OK;
}

// ---- modules ------------------------------------------------------------------------------------------
static int vnr_text_encoder_fwd_impl(vnr_handle h, const int32_t* d_ids, const int32_t* d_lengths, int B, int T, float pos_step, float* d_out) {
  TRY(check_ready(h));
  if (!d_ids || !d_out || B <= 0 || T <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  TRY(encoder_body(h, d_ids, d_lengths, B, T, pos_step, d_out));
  return h->training ? refresh_bn_affine(h) : VNR_OK;
}
int vnr_text_encoder_fwd(vnr_handle h, const int32_t* d_ids, const int32_t* d_lengths, int B, int T, float pos_step, float* d_out) {
  return range_guarded(h, MOD_ENC, [&] { return vnr_text_encoder_fwd_impl(h, d_ids, d_lengths, B, T, pos_step, d_out); });
}

int vnr_length_predictor_fwd(vnr_handle h, const float* d_text_embd, const int32_t* d_lengths, int B, int T, float* d_out) {
  TRY(check_ready(h));
  if (!d_text_embd || !d_out || B <= 0 || T <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  RUN_MISC(h, launch_length_predictor(d_text_embd, h->lp_w, h->lp_b, d_lengths, B, T, h->cfg.enc_pre_hidden,
                                      h->cfg.lenpred_activation, d_out, h->stream));
  return VNR_OK;
}

static int vnr_prior_sample_impl(vnr_handle h, const int32_t* d_z_lengths, const float* d_text_embd, const int32_t* d_text_lengths,
                     int B, int Tz, int Tt, const float* d_eps, float* d_z, float* d_logprobs) {
  TRY(check_ready(h));
  if (!d_z_lengths || !d_text_embd || !d_z || B <= 0 || Tz <= 0 || Tt <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  WS(kv, (size_t)B * Tt * h->prior_kv_n);
  TRY(run_kv(h, d_text_embd, B, Tt, h->cfg.enc_pre_hidden, h->prior_kv_wt, h->prior_kv_n, kv, h->cfg.prior_attention_dim));
  return prior_body(h, d_z_lengths, d_text_lengths, kv, h->prior_kv_n, B, Tz, Tt, d_eps, d_z, d_logprobs);
}
int vnr_prior_sample(vnr_handle h, const int32_t* d_z_lengths, const float* d_text_embd, const int32_t* d_text_lengths,
                     int B, int Tz, int Tt, const float* d_eps, float* d_z, float* d_logprobs) {
  return range_guarded(h, MOD_PRIOR, [&] { return vnr_prior_sample_impl(h, d_z_lengths, d_text_embd, d_text_lengths, B, Tz, Tt, d_eps, d_z, d_logprobs); });
}

static int vnr_decoder_fwd_impl(vnr_handle h, const float* d_z, const float* d_text_embd, const int32_t* d_z_lengths,
                    const int32_t* d_text_lengths, int B, int Tz, int Tt, int reduction_factor, float* d_initial,
                    float* d_outputs, float* d_alignments) {
  TRY(check_ready(h));
  if (!d_z || !d_text_embd || !d_outputs || B <= 0 || Tz <= 0 || Tt <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  WS(kv, (size_t)B * Tt * h->dec_kv_n);
  TRY(run_kv(h, d_text_embd, B, Tt, h->cfg.enc_pre_hidden, h->dec_kv_wt, h->dec_kv_n, kv, h->cfg.dec_attention_dim));
  TRY(decoder_body(h, d_z, kv, h->dec_kv_n, d_z_lengths, d_text_lengths, B, Tz, Tt, reduction_factor, d_initial,
                   d_outputs, d_alignments));
  return h->training ? refresh_bn_affine(h) : VNR_OK;
}
int vnr_decoder_fwd(vnr_handle h, const float* d_z, const float* d_text_embd, const int32_t* d_z_lengths,
                    const int32_t* d_text_lengths, int B, int Tz, int Tt, int reduction_factor, float* d_initial,
                    float* d_outputs, float* d_alignments) {
  return range_guarded(h, MOD_DEC, [&] { return vnr_decoder_fwd_impl(h, d_z, d_text_embd, d_z_lengths, d_text_lengths, B, Tz, Tt, reduction_factor, d_initial, d_outputs, d_alignments); });
}

static int vnr_posterior_fwd_impl(vnr_handle h, const float* d_mels, const float* d_text_embd, const int32_t* d_text_lengths,
                      const int32_t* d_target_lengths, int B, int Tz, int Tt, float* d_mu, float* d_logvar) {
  TRY(check_ready(h));
  if (!h->has_posterior) return fail(h, VNR_ERR_WEIGHT, "posterior weights were not loaded");
  if (!d_mels || !d_text_embd || !d_mu || !d_logvar || B <= 0 || Tz <= 0 || Tt <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  WS(kv, (size_t)B * Tt * h->post_kv_n);
  TRY(run_kv(h, d_text_embd, B, Tt, h->cfg.enc_pre_hidden, h->post_kv_wt, h->post_kv_n, kv, h->cfg.post_attention_dim));
  return posterior_body(h, d_mels, kv, h->post_kv_n, d_text_lengths, d_target_lengths, B, Tz, Tt, d_mu, d_logvar);
}
int vnr_posterior_fwd(vnr_handle h, const float* d_mels, const float* d_text_embd, const int32_t* d_text_lengths,
                      const int32_t* d_target_lengths, int B, int Tz, int Tt, float* d_mu, float* d_logvar) {
  return range_guarded(h, MOD_POST, [&] { return vnr_posterior_fwd_impl(h, d_mels, d_text_embd, d_text_lengths, d_target_lengths, B, Tz, Tt, d_mu, d_logvar); });
}

static int vnr_inference_impl(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const int32_t* d_reduced_lengths,
                  int B, int Tt, int Tz, int reduction_factor, float pos_step, const float* d_eps, float* d_mel,
                  float* d_alignments, float* d_text_embd_out) {
  TRY(check_ready(h));
  if (!d_ids || !d_reduced_lengths || !d_mel || B <= 0 || Tt <= 0 || Tz <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  const int Dm = h->cfg.enc_pre_hidden, C = h->cfg.latent_dim;
  float* text_embd = d_text_embd_out;
  if (!text_embd) { WS(t, (size_t)B * Tt * Dm); text_embd = t; }
  TRY(encoder_body(h, d_ids, d_text_lengths, B, Tt, pos_step, text_embd));
  WS(z, (size_t)B * Tz * C);
  if (h->late_dec_kv) {
    // the decoder's cross K|V are produced right before the decoder instead of ~2 ms earlier with the prior's: the decoder
    // cross-attention (the HBM-bound kernel of the path) then finds them in L2 / Infinity Cache (12.5 -> ? us per launch)
    WS(kvp, (size_t)B * Tt * h->prior_kv_n);
    if (h->kv_overlap && !h->profiling && !h->surveying && !h->flow.empty()) {
      // fork: the projection on the side stream behind the encoder; the main stream goes on with the first pre-chain + self-attention
      if (!h->kv_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->kv_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->kv_fork, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->kv_done, hipEventDisableTiming));
      }
      HIP_TRY(h, hipEventRecord(h->kv_fork, h->stream));
      HIP_TRY(h, hipStreamWaitEvent(h->kv_stream, h->kv_fork, 0));
      hipStream_t main_s = h->stream;
      h->stream = h->kv_stream;
      const int rc_kv = run_kv(h, text_embd, B, Tt, Dm, h->prior_kv_wt, h->prior_kv_n, kvp, h->cfg.prior_attention_dim);
      const hipError_t e_rec = hipEventRecord(h->kv_done, h->kv_stream);
      h->stream = main_s;
      TRY(rc_kv);
      HIP_TRY(h, e_rec);
      h->kv_wait = h->kv_done;
    } else {
      TRY(run_kv(h, text_embd, B, Tt, Dm, h->prior_kv_wt, h->prior_kv_n, kvp, h->cfg.prior_attention_dim));
    }
    // the decoder's pre-chain rides behind the last flow step's coupling (same chain launch) when the shapes allow
    const int Dd = h->cfg.dec_attention_dim;
    std::vector<PreStage> dpre;
    float *dxa = nullptr, *dqkv = nullptr;
    if (!h->dec_blks.empty() && h->dec_blks[0].D == Dd && C <= 256 && !(C & 31)) {
      dxa = ws_alloc(h, (size_t)B * Tz * Dd);
      dqkv = ws_alloc(h, qkv_floats(self_aoi_on(h, Dd, h->cfg.dec_attention_heads), B, Tz, Dd));
      if (!dxa || !dqkv) return fail(h, VNR_ERR_NOMEM, "workspace allocation failed");
      dpre = decoder_pre_stages(h, B, Tz, dxa, dqkv, 1, 0);
    }
    bool dec_pre_done = false;
    const int rc_prior = prior_body(h, d_reduced_lengths, d_text_lengths, kvp, h->prior_kv_n, B, Tz, Tt, d_eps, z, nullptr, dpre.empty() ? nullptr : &dpre, &dec_pre_done);
    TRY(kv_join(h));                                       // (nothing read the projection, or the prior failed: the side stream joins here at the latest)
    TRY(rc_prior);
    WS(kvd, (size_t)B * Tt * h->dec_kv_n);
    TRY(run_kv(h, text_embd, B, Tt, Dm, h->dec_kv_wt, h->dec_kv_n, kvd, h->cfg.dec_attention_dim));
    return decoder_body(h, z, kvd, h->dec_kv_n, d_reduced_lengths, d_text_lengths, B, Tz, Tt, reduction_factor, nullptr, d_mel, d_alignments,
                        dec_pre_done ? dxa : nullptr, dec_pre_done ? dqkv : nullptr);
  }
  // every cross-attention K|V of the memory (all prior blocks + decoder blocks) in one GEMM
  const int kv_n = h->prior_kv_n + h->dec_kv_n;
  WS(kv, (size_t)B * Tt * kv_n);
  TRY(run_kv(h, text_embd, B, Tt, Dm, h->prior_kv_wt, kv_n, kv, (h->cfg.prior_attention_dim == h->cfg.dec_attention_dim ? h->cfg.prior_attention_dim : 0)));
  TRY(prior_body(h, d_reduced_lengths, d_text_lengths, kv, kv_n, B, Tz, Tt, d_eps, z, nullptr));
  return decoder_body(h, z, kv + h->prior_kv_n, kv_n, d_reduced_lengths, d_text_lengths, B, Tz, Tt, reduction_factor,
                      nullptr, d_mel, d_alignments);
}
int vnr_inference(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const int32_t* d_reduced_lengths,
                  int B, int Tt, int Tz, int reduction_factor, float pos_step, const float* d_eps, float* d_mel,
                  float* d_alignments, float* d_text_embd_out) {
  return range_guarded(h, MOD_ENC | MOD_PRIOR | MOD_DEC, [&] { return vnr_inference_impl(h, d_ids, d_text_lengths, d_reduced_lengths, B, Tt, Tz, reduction_factor, pos_step, d_eps, d_mel, d_alignments, d_text_embd_out); });
}

static int vnr_prior_log_probability_impl(vnr_handle h, const float* d_z, const float* d_text_embd, const int32_t* d_z_lengths,
                              const int32_t* d_text_lengths, int B, int Tz, int Tt, float* d_logprobs) {
  TRY(check_ready(h));
  if (!d_z || !d_text_embd || !d_z_lengths || !d_logprobs || B <= 0 || Tz <= 0 || Tt <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  WS(kv, (size_t)B * Tt * h->prior_kv_n);
  TRY(run_kv(h, d_text_embd, B, Tt, h->cfg.enc_pre_hidden, h->prior_kv_wt, h->prior_kv_n, kv, h->cfg.prior_attention_dim));
  WS(zc, (size_t)B * Tz * h->cfg.latent_dim);
  HIP_TRY(h, hipMemcpyAsync(zc, d_z, (size_t)B * Tz * h->cfg.latent_dim * 4, hipMemcpyDeviceToDevice, h->stream));
  return prior_logprob_body(h, zc, d_z_lengths, d_text_lengths, kv, h->prior_kv_n, B, Tz, Tt, d_logprobs);
}
int vnr_prior_log_probability(vnr_handle h, const float* d_z, const float* d_text_embd, const int32_t* d_z_lengths,
                              const int32_t* d_text_lengths, int B, int Tz, int Tt, float* d_logprobs) {
  return range_guarded(h, MOD_PRIOR, [&] { return vnr_prior_log_probability_impl(h, d_z, d_text_embd, d_z_lengths, d_text_lengths, B, Tz, Tt, d_logprobs); });
}

// BasePosterior.reparameterize (posterior.py:21-39): samples [B, ns, T, C] = eps * exp(0.5 logvar) + mu; d_eps [B, ns, T, C] is the
// noise (the caller draws it with vnr_random_normal) or NULL = zeros (`random=False`).
int vnr_posterior_reparameterize(vnr_handle h, const float* d_mu, const float* d_logvar, const float* d_eps, int B, int nsamples, int T,
                                 float* d_samples) {
  if (!h || !d_mu || !d_logvar || !d_samples || B <= 0 || nsamples <= 0 || T <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  RUN_MISC(h, launch_posterior_rows(d_mu, d_logvar, d_eps, nullptr, B, nsamples, T, h->cfg.latent_dim, 0.f, d_samples, nullptr, h->stream));
  return VNR_OK;
}
// BasePosterior.log_probability (posterior.py:42-72): d_logprobs [B, ns] = sum_{t < len} -0.5 (C log 2pi + sum_c (logvar + n^2)), n = eps
// when given, else (z - mu) / (exp(0.5 logvar) + epsilon); d_lengths NULL = every frame.
int vnr_posterior_log_probability(vnr_handle h, const float* d_mu, const float* d_logvar, const float* d_z, const float* d_eps,
                                  const int32_t* d_lengths, int B, int nsamples, int T, float epsilon, float* d_logprobs) {
  if (!h || !d_mu || !d_logvar || !d_logprobs || (!d_z && !d_eps) || B <= 0 || nsamples <= 0 || T <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  ws_reset(h);
  const int Bt = B * nsamples;
  WS(rowlp, (size_t)Bt * T);
  RUN_MISC(h, launch_posterior_rows(d_mu, d_logvar, d_eps, d_z, B, nsamples, T, h->cfg.latent_dim, epsilon, nullptr, rowlp, h->stream));
  const int32_t* len = d_lengths;
  if (d_lengths && nsamples > 1) {
    WS(lt, (size_t)Bt);
    RUN_MISC(h, launch_tile_rows(d_lengths, 1, B, nsamples, lt, h->stream));
    len = reinterpret_cast<const int32_t*>(lt);
  }
  RUN_MISC(h, launch_masked_row_reduce(rowlp, len, Bt, T, 1.0f, d_logprobs, 0, h->stream));
  return VNR_OK;
}
// TransformerPrior.init (prior.py:171-186): like sample(), but every ActNormFlow first takes log_scale / bias from the statistics of its
// input (flow.py:189-196).  The variables in the weight store are updated and all packed panels rebuilt.  d_logprobs [B] may be NULL.
int vnr_prior_init(vnr_handle h, const int32_t* d_z_lengths, const float* d_text_embd, const int32_t* d_text_lengths, int B, int Tz, int Tt,
                   const float* d_eps, float* d_z, float* d_logprobs) {
  TRY(check_ready(h));
  if (!d_z_lengths || !d_text_embd || !d_z || B <= 0 || Tz <= 0 || Tt <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  h->derived_fresh = false;
  TRY(range_checked_sync(h, [&]() -> int {                // (range sentinel: the ActNorm variables are ASSIGNED -- a repeat on exact fp32 overwrites them)
    ws_reset(h);
    WS(kv, (size_t)B * Tt * h->prior_kv_n);
    TRY(run_kv(h, d_text_embd, B, Tt, h->cfg.enc_pre_hidden, h->prior_kv_wt, h->prior_kv_n, kv, h->cfg.prior_attention_dim));
    return prior_init_body(h, d_z_lengths, d_text_lengths, kv, h->prior_kv_n, B, Tz, Tt, d_eps, d_z, d_logprobs);
  }));
  return vnr_finalize_weights(h);
}

static int vnr_elbo_fwd_impl(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const float* d_mel_targets,
                 const int32_t* d_mel_lengths, const int32_t* d_reduced_lengths, int B, int Tt, int Tm, int rf,
                 float pos_step, const float* d_eps, float* d_outs, float* d_l2, float* d_kl, float* d_length_l2,
                 float* d_alignments, float* d_aux) {
  TRY(check_ready(h));
  if (!h->has_posterior) return fail(h, VNR_ERR_WEIGHT, "posterior weights were not loaded");
  if (!d_ids || !d_mel_targets || !d_mel_lengths || !d_reduced_lengths || !d_outs || !d_l2 || !d_kl || !d_length_l2 ||
      B <= 0 || Tt <= 0 || Tm <= 0 || rf < 1)
    return fail(h, VNR_ERR_ARG, "bad argument");
  ws_reset(h);
  const vnr_config& c = h->cfg;
  const int Dm = c.enc_pre_hidden, C = c.latent_dim, od = c.output_dim;
  const int Tz = (Tm + rf - 1) / rf;                                   // mel_targets[:, ::rf, :] (models.py:123)
  const int ns = h->n_sample, Bt = B * ns;                             // models.py:141-178: everything after the posterior runs on batch * n_sample rows
  if (c.num_mels != od) return fail(h, VNR_ERR_ARG, "num_mels must equal output_dim for the L2 loss");
  WS(text_embd, (size_t)B * Tt * Dm);
  TRY(encoder_body(h, d_ids, d_text_lengths, B, Tt, pos_step, text_embd));
  WS(pred, (size_t)Bt);
  RUN_MISC(h, launch_length_predictor(text_embd, h->lp_w, h->lp_b, d_text_lengths, B, Tt, Dm, c.lenpred_activation, pred, h->stream));
  // reduced mels: frames 0, rf, 2rf, ... (strided copy, rows of num_mels floats)
  WS(rmel, (size_t)B * Tz * c.num_mels);
  for (int b = 0; b < B; ++b)
    HIP_TRY(h, hipMemcpy2DAsync(rmel + (size_t)b * Tz * c.num_mels, (size_t)c.num_mels * 4, d_mel_targets + (size_t)b * Tm * c.num_mels,
                                (size_t)rf * c.num_mels * 4, (size_t)c.num_mels * 4, Tz, hipMemcpyDeviceToDevice, h->stream));
  // posterior (first head is USED as logvar, second as mu: models.py:136 vs posterior.py:130)
  WS(kvp, (size_t)B * Tt * h->post_kv_n);
  TRY(run_kv(h, text_embd, B, Tt, Dm, h->post_kv_wt, h->post_kv_n, kvp, h->cfg.post_attention_dim));
  WS(head1, (size_t)B * Tz * C); WS(head2, (size_t)B * Tz * C);
  TRY(posterior_body(h, rmel, kvp, h->post_kv_n, d_text_lengths, d_reduced_lengths, B, Tz, Tt, head1, head2));
  const float* logvar = head1; const float* mu = head2;
  // n_sample > 1: text encoding, targets and lengths tiled n_sample times, sample index inner (models.py:149-178)
  const float* text_t = text_embd; const float* mel_t = d_mel_targets;
  const int32_t *tl_t = d_text_lengths, *rl_t = d_reduced_lengths, *ml_t = d_mel_lengths;
  if (ns > 1) {
    WS(tt, (size_t)Bt * Tt * Dm); WS(mt, (size_t)Bt * Tm * od); WS(lens, (size_t)3 * Bt + 64); WS(pt, (size_t)Bt);
    RUN_MISC(h, launch_tile_rows(text_embd, (size_t)Tt * Dm, B, ns, tt, h->stream));
    RUN_MISC(h, launch_tile_rows(d_mel_targets, (size_t)Tm * od, B, ns, mt, h->stream));
    int32_t* li = reinterpret_cast<int32_t*>(lens);
    if (d_text_lengths) RUN_MISC(h, launch_tile_rows(d_text_lengths, 1, B, ns, li, h->stream));
    RUN_MISC(h, launch_tile_rows(d_reduced_lengths, 1, B, ns, li + Bt, h->stream));
    RUN_MISC(h, launch_tile_rows(d_mel_lengths, 1, B, ns, li + 2 * Bt, h->stream));
    RUN_MISC(h, launch_tile_rows(pred, 1, B, ns, pt, h->stream));
    text_t = tt; mel_t = mt; tl_t = d_text_lengths ? li : nullptr; rl_t = li + Bt; ml_t = li + 2 * Bt; pred = pt;
  }
  const int kv_ld = h->prior_kv_n + h->dec_kv_n;
  WS(kv, (size_t)Bt * Tt * kv_ld);
  TRY(run_kv(h, text_t, Bt, Tt, Dm, h->prior_kv_wt, kv_ld, kv, (h->cfg.prior_attention_dim == h->cfg.dec_attention_dim ? h->cfg.prior_attention_dim : 0)));
  WS(z, (size_t)Bt * Tz * C); WS(rowlp, (size_t)Bt * Tz); WS(post_lp, (size_t)Bt);
  if (ns > 1) RUN_MISC(h, launch_posterior_rows(mu, logvar, d_eps, nullptr, B, ns, Tz, C, 0.f, z, rowlp, h->stream));
  else RUN_MISC(h, launch_reparam(mu, logvar, d_eps, B * Tz, C, z, rowlp, h->stream));
  RUN_MISC(h, launch_masked_row_reduce(rowlp, rl_t, Bt, Tz, 1.0f, post_lp, 0, h->stream));
  // decoder on the samples, cropped to the target length (models.py:179-183)
  WS(dinit, (size_t)Bt * Tz * rf * od); WS(douts, (size_t)Bt * Tz * rf * od);
  TRY(decoder_body(h, z, kv + h->prior_kv_n, kv_ld, rl_t, tl_t, Bt, Tz, Tt, rf, dinit, douts, d_alignments));
  HIP_TRY(h, hipMemcpy2DAsync(d_outs, (size_t)Tm * od * 4, douts, (size_t)Tz * rf * od * 4, (size_t)Tm * od * 4, Bt, hipMemcpyDeviceToDevice, h->stream));
  WS(rows, (size_t)Bt * Tm); WS(sum_out, (size_t)Bt); WS(sum_init, (size_t)Bt);
  RUN_MISC(h, launch_sqerr_rows(douts, Tz * rf, mel_t, Tm, Bt, od, rows, h->stream));
  RUN_MISC(h, launch_masked_row_reduce(rows, ml_t, Bt, Tm, 1.0f, sum_out, 0, h->stream));
  RUN_MISC(h, launch_sqerr_rows(dinit, Tz * rf, mel_t, Tm, Bt, od, rows, h->stream));
  RUN_MISC(h, launch_masked_row_reduce(rows, ml_t, Bt, Tm, 1.0f, sum_init, 0, h->stream));
  // prior log-probability of the samples (z is consumed)
  WS(prior_lp, (size_t)Bt);
  TRY(prior_logprob_body(h, z, rl_t, tl_t, kv, kv_ld, Bt, Tz, Tt, prior_lp));
  if (ns > 1) {          // per-utterance terms = means over the samples (models.py:79-83,90)
    WS(l2t, (size_t)Bt); WS(llt, (size_t)Bt); WS(klt, (size_t)Bt);
    RUN_MISC(h, launch_elbo_scalars(sum_out, sum_init, ml_t, pred, post_lp, prior_lp, Bt, l2t, llt, klt, h->stream));
    RUN_MISC(h, launch_group_mean(l2t, B, ns, d_l2, h->stream));
    RUN_MISC(h, launch_group_mean(llt, B, ns, d_length_l2, h->stream));
    RUN_MISC(h, launch_group_mean(klt, B, ns, d_kl, h->stream));
  } else {
    RUN_MISC(h, launch_elbo_scalars(sum_out, sum_init, d_mel_lengths, pred, post_lp, prior_lp, B, d_l2, d_length_l2, d_kl, h->stream));
  }
  if (d_aux) {   // [pred_lengths (B) | posterior_logprobs (B * n_sample) | prior_logprobs (B * n_sample)] (diagnostics / tests)
    HIP_TRY(h, hipMemcpy2DAsync(d_aux, 4, pred, (size_t)ns * 4, 4, B, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_aux + B, post_lp, (size_t)Bt * 4, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_aux + B + (size_t)Bt, prior_lp, (size_t)Bt * 4, hipMemcpyDeviceToDevice, h->stream));
  }
  if (h->training) TRY(refresh_bn_affine(h));     // the moving statistics moved
  return VNR_OK;
}
int vnr_elbo_fwd(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const float* d_mel_targets,
                 const int32_t* d_mel_lengths, const int32_t* d_reduced_lengths, int B, int Tt, int Tm, int rf,
                 float pos_step, const float* d_eps, float* d_outs, float* d_l2, float* d_kl, float* d_length_l2,
                 float* d_alignments, float* d_aux) {
  return range_guarded(h, MOD_ENC | MOD_PRIOR | MOD_DEC | MOD_POST, [&] {
    return vnr_elbo_fwd_impl(h, d_ids, d_text_lengths, d_mel_targets, d_mel_lengths, d_reduced_lengths, B, Tt, Tm, rf, pos_step, d_eps, d_outs, d_l2, d_kl,
                             d_length_l2, d_alignments, d_aux);
  });
}

// train_step (train.py:127-138): training-mode forward, gradients of mel_l2 + kl_weight * max(kl, 0) + length_weight *
// length_l2 w.r.t. every trainable variable, Keras Adam (apply_update = 0: gradients only, read with vnr_get_gradient).
int vnr_train_step(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const float* d_mel_targets,
                   const int32_t* d_mel_lengths, const int32_t* d_reduced_lengths, int B, int Tt, int Tm, int rf, float pos_step,
                   const float* d_eps, float kl_weight, float length_weight, float learning_rate, float beta1, float beta2,
                   float epsilon, int apply_update, float* h_scalars) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  if (!h->finalized) return fail(h, VNR_ERR_WEIGHT, "weights not finalized: call vnr_finalize_weights first");
  HIP_TRY(h, hipSetDevice(h->device));
  if (!h->has_posterior) return fail(h, VNR_ERR_WEIGHT, "posterior weights were not loaded");
  if (!d_ids || !d_text_lengths || !d_mel_targets || !d_mel_lengths || !d_reduced_lengths || !d_eps || B <= 0 || Tt <= 0 || Tm <= 0 || rf < 1)
    return fail(h, VNR_ERR_ARG, "bad argument");
  if (h->cfg.num_mels != h->cfg.output_dim) return fail(h, VNR_ERR_ARG, "num_mels must equal output_dim for the L2 loss");
  if (rf > h->cfg.max_reduction_factor) return fail(h, VNR_ERR_ARG, "reduction_factor out of range");
  // range sentinel: whatever asynchronous calls are still pending get their checkpoint first (their verdict is theirs, not this step's)
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  TRY(range_checkpoint(h));
  const bool saved_split = h->split_scope, saved_training = h->training;
  int rc = VNR_OK;
  TRY(bn_backup_save(h));
  for (int attempt = 0; attempt < 2; ++attempt) {
    ws_reset(h);
    h->split_scope = false;            // exact fp32 GEMMs throughout the training step
    h->training = true;
    h->in_train_step = true;
    g_det = h->deterministic ? &h->det : nullptr;          // kernels that would end in float atomics leave ordered partials instead (common.h: DetState)
    g_train_exact = h->train_fp32;
    if (h->range_flag) h->range_flag[1] = 0u;
    rc = train_step_impl(h, d_ids, d_text_lengths, d_mel_targets, d_mel_lengths, d_reduced_lengths, B, Tt, Tm, rf, pos_step, d_eps,
                         kl_weight, length_weight, learning_rate, beta1, beta2, epsilon, apply_update, h_scalars);
    g_det = nullptr; g_train_exact = false;
    h->split_scope = saved_split; h->training = saved_training; h->in_train_step = false;
    if (rc != VNR_OK || !h->range_flag || !h->range_sentinel) { if (h->range_flag) *h->range_flag = 0u; break; }
    // The step's verdict (train_step_impl copied the word behind the backward pass, all-reduced it over the ranks and predicated Adam and
    // the BatchNormalization moving updates on it).  Raised: the variables, Adam's moments and the moving statistics are as before the
    // step; the step is repeated ONCE on exact fp32 MFMA with scaled attention cores and fp32 kernel-gradient GEMMs, and the handle
    // stays on that path (its weights produce activations the split path cannot carry).
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool tripped = h->range_flag[1] != 0u;
    *h->range_flag = 0u; h->range_flag[1] = 0u;
    HIP_TRY(h, hipMemsetAsync(h->d_step_flag, 0, sizeof(unsigned), h->stream));
    if (!tripped) break;
    h->range_trips++;
    TRY(bn_backup_restore(h));          // (the layers in front of the trip took their moving update in the first attempt)
    if (apply_update) train_step_rollback(h);
    if (h->train_fp32) {
      rc = fail(h, VNR_ERR_RANGE, "vnr_train_step: non-finite gradients on the exact-fp32 path as well (inputs or variables hold NaN / inf): the "
                                  "variables were NOT updated");
      break;
    }
    h->train_fp32 = true;
    h->derived_fresh = false;
  }
  if (h->det.alloc_failed) {                               // never lose bit-reproducibility silently under memory pressure
    h->det.alloc_failed = false;
    return fail(h, VNR_ERR_NOMEM, "deterministic mode: a scratch buffer for ordered partial sums could not be allocated -- that launch fell back to "
                                  "float atomics, so this step is NOT bit-reproducible (free device memory or set deterministic = 0)");
  }
  if (rc == VNR_OK && !h->packed_stale) TRY(refresh_bn_affine(h));      // moving statistics moved
  return rc;
}

// ---- data-parallel training: RCCL communicator (one process per GPU; the 128-byte id travels over the host control plane) ----
int vnr_comm_unique_id(vnr_handle h, char* id128) {
  if (!h || !id128) return fail(h, VNR_ERR_ARG, "null argument");
  TRY(rccl_load(h));
  ncclUniqueId id;
  RCCL_TRY(h, g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return VNR_OK;
}
int vnr_comm_init(vnr_handle h, int nranks, int rank, const char* id128) {
  if (!h || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(h, VNR_ERR_ARG, "bad communicator arguments");
  if (h->comm) return fail(h, VNR_ERR_STATE, "communicator already initialised");
  TRY(rccl_load(h));
  HIP_TRY(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  RCCL_TRY(h, g_rccl.CommInitRank(&h->comm, nranks, id, rank));
  h->comm_size = nranks; h->comm_rank = rank;
  return VNR_OK;
}
// every rank starts from rank 0's variables (incl. BN moving statistics and the ActNorm init of vnr_init)
int vnr_comm_broadcast_weights(vnr_handle h) {
  if (!h || !h->comm) return fail(h, VNR_ERR_STATE, "communicator not initialised");
  HIP_TRY(h, hipSetDevice(h->device));
  std::vector<std::string> names;
  for (auto& kv : h->w) names.push_back(kv.first);
  std::sort(names.begin(), names.end());          // identical order on every rank
  for (auto& nm : names) {
    Tensor& t = h->w[nm];
    RCCL_TRY(h, g_rccl.Broadcast(t.d, t.d, (size_t)t.n, ncclFloat, 0, h->comm, h->stream));
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->derived_fresh = false;
  for (auto& kv : h->w) if (kv.second.n == 1) HIP_TRY(h, hipMemcpy(&kv.second.scalar, kv.second.d, sizeof(float), hipMemcpyDeviceToHost));
  return vnr_finalize_weights(h);
}
// rank count and rank of the bound communicator AS RCCL REPORTS THEM (ncclCommCount / ncclCommUserRank), not the launcher's environment
int vnr_comm_info(vnr_handle h, int* nranks, int* rank) {
  if (!h || !nranks || !rank) return fail(h, VNR_ERR_ARG, "null argument");
  if (!h->comm) return fail(h, VNR_ERR_STATE, "communicator not initialised");
  if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(h, VNR_ERR_STATE, "librccl.so lacks ncclCommCount / ncclCommUserRank");
  RCCL_TRY(h, g_rccl.CommCount(h->comm, nranks));
  RCCL_TRY(h, g_rccl.CommUserRank(h->comm, rank));
  return VNR_OK;
}
int vnr_comm_destroy(vnr_handle h) {
  if (!h) return VNR_OK;
  if (h->comm) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; h->comm_size = 1; h->comm_rank = 0; }
  return VNR_OK;
}

// gradient of the last vnr_train_step w.r.t. variable `path` (n floats, synchronises)
int vnr_get_gradient(vnr_handle h, const char* path, float* host, int64_t n) {
  if (!h || !path || !host) return fail(h, VNR_ERR_ARG, "null argument");
  if (!h->train) return fail(h, VNR_ERR_STATE, "no training step has run");
  auto it = h->train->index.find(path);
  if (it == h->train->index.end()) return fail(h, VNR_ERR_WEIGHT, std::string("not a trainable variable: ") + path);
  if (n != h->train->n[it->second]) return fail(h, VNR_ERR_ARG, std::string("size mismatch for ") + path);
  return vnr_memcpy_d2h(h, host, h->train->g[it->second], (size_t)n * sizeof(float));
}

// ---- optimizer state (train.py:246-249: tf.train.Checkpoint(step, optimizer, model) saves / restores Adam's slots) --------
static int opt_slot(vnr_handle h, const char* path, const char* slot, float** p, int64_t n) {
  if (!h || !path || !slot) return fail(h, VNR_ERR_ARG, "null argument");
  if (!h->finalized) return fail(h, VNR_ERR_WEIGHT, "weights not finalized: call vnr_finalize_weights first");
  HIP_TRY(h, hipSetDevice(h->device));
  TRY(train_prepare(h));
  auto it = h->train->index.find(path);
  if (it == h->train->index.end()) return fail(h, VNR_ERR_WEIGHT, std::string("not a trainable variable: ") + path);
  if (n != h->train->n[it->second]) return fail(h, VNR_ERR_ARG, std::string("size mismatch for ") + path);
  if (!strcmp(slot, "m")) *p = h->train->m[it->second];
  else if (!strcmp(slot, "v")) *p = h->train->v[it->second];
  else return fail(h, VNR_ERR_ARG, std::string("unknown optimizer slot ") + slot + " (Adam has m and v)");
  return VNR_OK;
}
int vnr_get_optimizer_slot(vnr_handle h, const char* path, const char* slot, float* host, int64_t n) {
  float* p = nullptr;
  if (!host) return fail(h, VNR_ERR_ARG, "null argument");
  TRY(opt_slot(h, path, slot, &p, n));
  return vnr_memcpy_d2h(h, host, p, (size_t)n * sizeof(float));
}
int vnr_set_optimizer_slot(vnr_handle h, const char* path, const char* slot, const float* host, int64_t n) {
  float* p = nullptr;
  if (!host) return fail(h, VNR_ERR_ARG, "null argument");
  TRY(opt_slot(h, path, slot, &p, n));
  return vnr_memcpy_h2d(h, p, host, (size_t)n * sizeof(float));
}
int vnr_get_optimizer_step(vnr_handle h, int64_t* iterations) {
  if (!h || !iterations) return fail(h, VNR_ERR_ARG, "null argument");
  *iterations = h->train ? (int64_t)h->train->step : 0;
  return VNR_OK;
}
int vnr_set_optimizer_step(vnr_handle h, int64_t iterations) {
  if (!h || iterations < 0) return fail(h, VNR_ERR_ARG, "bad argument");
  if (!h->finalized) return fail(h, VNR_ERR_WEIGHT, "weights not finalized: call vnr_finalize_weights first");
  HIP_TRY(h, hipSetDevice(h->device));
  TRY(train_prepare(h));
  h->train->step = (long)iterations;
  return VNR_OK;
}

// VAENAR.init (models.py:212-226): encoder(training=True) -> prior.init -> decoder(training=True, rf = max).
static int init_impl(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const int32_t* d_reduced_lengths, int B,
                     int Tt, int Tz, float pos_step, const float* d_eps, float* d_mel) {
  const vnr_config& c = h->cfg;
  const int Dm = c.enc_pre_hidden, C = c.latent_dim, rf = c.max_reduction_factor;
  WS(text_embd, (size_t)B * Tt * Dm);
  TRY(encoder_body(h, d_ids, d_text_lengths, B, Tt, pos_step, text_embd));
  const int kv_ld = h->prior_kv_n + h->dec_kv_n;
  WS(kv, (size_t)B * Tt * kv_ld);
  TRY(run_kv(h, text_embd, B, Tt, Dm, h->prior_kv_wt, kv_ld, kv, (h->cfg.prior_attention_dim == h->cfg.dec_attention_dim ? h->cfg.prior_attention_dim : 0)));
  WS(z, (size_t)B * Tz * C);
  TRY(prior_init_body(h, d_reduced_lengths, d_text_lengths, kv, kv_ld, B, Tz, Tt, d_eps, z));
  float* mel = d_mel;
  if (!mel) { WS(tmp, (size_t)B * Tz * rf * c.output_dim); mel = tmp; }
  return decoder_body(h, z, kv + h->prior_kv_n, kv_ld, d_reduced_lengths, d_text_lengths, B, Tz, Tt, rf, nullptr, mel, nullptr);
}
int vnr_init(vnr_handle h, const int32_t* d_ids, const int32_t* d_text_lengths, const int32_t* d_reduced_lengths, int B,
             int Tt, int Tz, float pos_step, const float* d_eps, float* d_mel) {
  TRY(check_ready(h));
  if (!d_ids || !d_reduced_lengths || B <= 0 || Tt <= 0 || Tz <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  const bool saved = h->training;
  h->training = true;
  // (range sentinel: ActNormFlow.init ASSIGNS its variables from the statistics of its input -- a repeat on exact fp32 overwrites them)
  const int rc = range_checked_sync(h, [&] { ws_reset(h); return init_impl(h, d_ids, d_text_lengths, d_reduced_lengths, B, Tt, Tz, pos_step, d_eps, d_mel); });
  h->training = saved;
  h->derived_fresh = false;
  TRY(rc);
  // ActNorm variables and BN moving statistics changed: rebuild every folded / packed panel from the weight store
  return vnr_finalize_weights(h);
}

// ---- single operators ---------------------------------------------------------------------------------------
int vnr_op_dense(vnr_handle h, const vnr_dense_desc* d) {
  if (!h || !d) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  const int K = d->k1 + d->k2;
  if (!d->d_a1 || !d->d_w || !d->d_c || d->m <= 0 || d->n <= 0 || K <= 0) return fail(h, VNR_ERR_ARG, "bad dense descriptor");
  if (d->k2 > 0 && (d->k1 & 31)) return fail(h, VNR_ERR_ARG, "vnr_op_dense: with a second input panel (k2 > 0) k1 must be a multiple of 32");
  if ((d->k1 & 3) || (d->k2 & 3) || (d->lda1 & 3) || (d->k2 > 0 && (d->lda2 & 3)))
    return fail(h, VNR_ERR_ARG, "vnr_op_dense: k1, k2 and the row strides must be multiples of 4 floats (16-byte rows)");
  if (((size_t)d->m * d->lda1 + K) * 4 >= ((size_t)1 << 31) || ((size_t)d->n * K + K) * 4 >= ((size_t)1 << 31))
    return fail(h, VNR_ERR_ARG, "vnr_op_dense: an operand spans 2 GiB or more (one buffer descriptor per operand)");
  ws_reset(h);
  WS(wt, (size_t)K * d->n);
  RUN_MISC(h, launch_transpose(d->d_w, K, d->n, wt, K, h->stream));
  GemmArgs g;
  g.A1 = d->d_a1; g.lda1 = d->lda1; g.K1 = d->k1; g.A2 = d->k2 > 0 ? d->d_a2 : nullptr; g.lda2 = d->lda2; g.K = K;
  g.Wt = wt; g.ldw = K; g.bias = d->d_bias; g.act = d->activation; g.residual = d->d_residual; g.ldr = d->ldr;
  g.pe = d->d_pe; g.pe_T = d->pe_T > 0 ? d->pe_T : 1; g.pe_w = d->pe_weight; g.C = d->d_c; g.ldc = d->ldc; g.M = d->m; g.N = d->n;
  h->split_scope = false;                  // the operator entry point picks the path explicitly:
  bool split_ok = h->op_dense_split;
  if (split_ok && h->range_guard) {          // the split path's range contract, per ROW at the operator level: a row maximum outside the window -> exact fp32
    TRY(survey_begin(h));
    h->surveying = false;
    TRY(survey_matrix(h, g.A1, g.lda1, g.M, g.A2 ? g.K1 : g.K, 0));
    if (g.A2) TRY(survey_matrix(h, g.A2, g.lda2, g.M, g.K - g.K1, 0));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned w[4] = {0u, 0x7f800000u, 0u, 0x7f800000u};
    HIP_TRY(h, hipMemcpy(w, h->survey_words, (size_t)h->survey_n * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->survey_n; ++i) {
      float mx, mn; memcpy(&mx, &w[2 * i], 4); memcpy(&mn, &w[2 * i + 1], 4);
      if (mx != 0.f && !(mx < kRangeHi && mn >= kRangeLo)) split_ok = false;
    }
  }
  if (split_ok) {                           // option "op_dense_split": exercise the split-fp16 kernel on a temporary image
    WS(img, (size_t)d->n * ((K + 31) / 32) * 32);
    RUN_MISC(h, launch_split_weights(wt, d->n, K, 256.f, img, h->stream));
    g.Wsplit = img; g.acc_scale = 1.f / 256.f;
  }
  if (d->d_ln_gamma && d->d_ln_beta) {
    if (d->n <= 256 && !d->d_pe) { g.ln_gamma = d->d_ln_gamma; g.ln_beta = d->d_ln_beta; return run_gemm(h, g); }
    if (d->ldc != d->n) return fail(h, VNR_ERR_ARG, "LayerNorm epilogue needs a dense output (ldc == n)");
    TRY(run_gemm(h, g));
    return run_ln(h, d->d_c, d->d_ln_gamma, d->d_ln_beta, d->m, d->n, d->d_c);
  }
  return run_gemm(h, g);
}

int vnr_op_conv1d_bn(vnr_handle h, const float* d_x, int B, int T, int Cin, const float* d_kernel, int k, int Cout,
                     const float* d_bias, int activation, int bn_before_act, const float* d_gamma, const float* d_beta,
                     const float* d_mean, const float* d_var, float* d_y) {
  if (!h || !d_x || !d_kernel || !d_y || !(k & 1)) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  ws_reset(h);
  WS(wt, (size_t)k * Cin * Cout); WS(sc, Cout); WS(sh, Cout);
  RUN_MISC(h, launch_transpose(d_kernel, k * Cin, Cout, wt, k * Cin, h->stream));
  ConvL L; L.wt = wt; L.bias = d_bias; L.bn_scale = nullptr; L.bn_shift = nullptr; L.k = k; L.cin = Cin; L.cout = Cout;
  if (d_gamma) {
    RUN_MISC(h, launch_bn_affine(d_gamma, d_beta, d_mean, d_var, Cout, sc, sh, h->stream));
    L.bn_scale = sc; L.bn_shift = sh;
  }
  return run_conv(h, L, d_x, nullptr, B, T, activation, bn_before_act, d_y);
}

int vnr_op_attention(vnr_handle h, const float* d_q, int ldq, const float* d_k, int ldk, const float* d_v, int ldv,
                     const int32_t* d_q_lengths, const int32_t* d_k_lengths, int B, int H, int Tq, int Tk, int causal,
                     float temperature, float* d_ctx, int ldo, float* d_alignments) {
  if (!h || !d_q || !d_k || !d_v || !d_ctx) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  AttnArgs a;
  a.Q = d_q; a.ldq = ldq; a.K = d_k; a.ldk = ldk; a.V = d_v; a.ldv = ldv; a.q_len = d_q_lengths; a.k_len = d_k_lengths;
  a.ctx = d_ctx; a.ldo = ldo; a.ali = d_alignments; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.causal = causal; a.temperature = temperature;
  a.q_bs = (long long)Tq * ldq; a.k_bs = (long long)Tk * ldk; a.v_bs = (long long)Tk * ldv; a.o_bs = (long long)Tq * ldo;
  if (h->op_attn_presplit && (!d_alignments || (!causal && Tk <= 128))) {
    // option "op_attn_presplit": the attention3 path -- fp32 operands are first rewritten as attention operand images
    // (inside the engine the producers write them directly)
    ws_reset(h);
    const int D = H * 64, ttq = (Tq + 31) / 32, ttk = (Tk + 31) / 32;
    WS(qi, (size_t)B * H * ttq * kAoiTile / 4); WS(ki, (size_t)B * H * ttk * kAoiTile / 4); WS(vi, (size_t)B * H * ttk * kAoiTile / 4);
    AoiDesc dq; dq.mode = 1; dq.D = D; dq.T = Tq; dq.TT = ttq; dq.qk = reinterpret_cast<char*>(qi);
    AoiDesc dk; dk.mode = 1; dk.D = D; dk.T = Tk; dk.TT = ttk; dk.qk = reinterpret_cast<char*>(ki);
    AoiDesc dv; dv.mode = 2; dv.D = D; dv.T = Tk; dv.TT = ttk; dv.vt = reinterpret_cast<char*>(vi);
    RUN_MISC(h, launch_aoi_convert(d_q, ldq, B * Tq, D, dq, h->stream));
    RUN_MISC(h, launch_aoi_convert(d_k, ldk, B * Tk, D, dk, h->stream));
    RUN_MISC(h, launch_aoi_convert(d_v, ldv, B * Tk, D, dv, h->stream));
    Attn3Args t;
    t.Qi = dq.qk; t.Ki = dk.qk; t.Vi = dv.vt; t.q_len = d_q_lengths; t.k_len = d_k_lengths; t.ctx = d_ctx; t.ldo = ldo; t.o_bs = a.o_bs; t.ali = d_alignments;
    t.B = B; t.H = H; t.Tq = Tq; t.Tk = Tk; t.temperature = temperature; t.causal = causal;
    return run_attention3(h, t, !causal);
  }
  return run_attention(h, a, !causal);
}

int vnr_op_layer_norm(vnr_handle h, const float* d_x, const float* d_gamma, const float* d_beta, int rows, int dim, float* d_y) {
  if (!h || !d_x || !d_gamma || !d_beta || !d_y) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  return run_ln(h, d_x, d_gamma, d_beta, rows, dim, d_y);
}

// kernel gradient of one Dense / one Conv1D tap as tape.gradient produces it (train.py:136): dW[K][N] = sum_m x[m + shift]^T . dy[m],
// rows that would cross an utterance boundary (T rows per utterance) skipped -- the op behind every weight gradient of vnr_train_step
int vnr_op_kernel_grad(vnr_handle h, const float* d_x, int ldx, const float* d_dy, int lddy, int M, int K, int N, int T, int shift, float* d_dw) {
  if (!h || !d_x || !d_dy || !d_dw || M <= 0 || K <= 0 || N <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  unsigned* amax = nullptr;
  HIP_TRY(h, hipMalloc(&amax, sizeof(unsigned)));
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(unsigned), h->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_dw, 0, (size_t)K * N * sizeof(float), h->stream);
  if (e == hipSuccess) e = launch_absmax2d(d_dy, lddy, M, N, amax, h->stream);
  if (e == hipSuccess) e = launch_gemm_tn_scaled(d_x, ldx, d_dy, lddy, d_dw, N, M, K, N, T > 0 ? T : M, shift, amax, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(amax);                                    // on every path: the scratch word must not leak when a launch fails
  HIP_TRY(h, e);
  return VNR_OK;
}

// tf.random.normal(shape, stddev) on the device (prior.py:35, posterior.py:35)
int vnr_random_normal(vnr_handle h, uint64_t seed, uint64_t offset, float stddev, float* d_out, size_t n) {
  if (!h || !d_out) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  RUN_MISC(h, launch_philox_normal(d_out, n, seed, offset, stddev, h->stream));
  return VNR_OK;
}

int vnr_op_positional_encoding(vnr_handle h, int T, int dim, float step, float* d_out) {
  if (!h || !d_out) return fail(h, VNR_ERR_ARG, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  RUN_MISC(h, launch_positional_encoding(T, dim, step, d_out, h->stream));
  return VNR_OK;
}

// ---- vocoder step after the path (reference audio/audio.py:81-102; vocoder.hip) ---------------------------------------------
int vnr_voc_mel_to_linear(vnr_handle h, const float* d_mel, const float* d_inv_basis_t, int B, int T, int n_mels, int n_freq,
                          float min_level_db, float ref_level_db, float max_abs_value, int symmetric_specs, float power, float* d_S) {
  if (!h || !d_mel || !d_inv_basis_t || !d_S || B <= 0 || T <= 0 || n_mels <= 0 || n_freq <= 0) return fail(h, VNR_ERR_ARG, "bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  RUN_MISC(h, launch_mel_to_linear(d_mel, d_inv_basis_t, B * T, n_mels, n_freq, min_level_db, ref_level_db, max_abs_value, symmetric_specs,
                                   power, d_S, h->stream));
  return VNR_OK;
}

int vnr_voc_griffin_lim(vnr_handle h, const float* d_S, const float* d_init_angles, uint64_t seed, const int32_t* d_frames, int B, int T,
                        int n_fft, int hop, int win, int iters, float* d_wav) {
  if (!h || !d_S || !d_wav || B <= 0 || iters < 0) return fail(h, VNR_ERR_ARG, "bad argument");
  if (n_fft != 2048) return fail(h, VNR_ERR_ARG, "griffin_lim: the FFT kernel is built for n_fft = 2048 (num_freq 1025, hparams.py:268)");
  if (win <= 0 || win > n_fft || hop <= 0 || hop > win) return fail(h, VNR_ERR_ARG, "griffin_lim: need 0 < hop <= win <= n_fft");
  if ((win + hop - 1) / hop > 8) return fail(h, VNR_ERR_ARG, "griffin_lim: more than 8 overlapping frames per sample (win / hop > 8)");
  if ((long long)hop * (T - 1) <= n_fft / 2) return fail(h, VNR_ERR_ARG, "griffin_lim: hop * (frames - 1) must exceed n_fft / 2 (reflect padding)");
  HIP_TRY(h, hipSetDevice(h->device));
  ws_reset(h);
  float* tab = nullptr;
  auto it = h->voc_tables.find(win);
  if (it != h->voc_tables.end()) tab = it->second;
  else {
    std::vector<float> tw, window;
    voc_tables(win, tw, window);
    HIP_TRY(h, hipMalloc((void**)&tab, (tw.size() + window.size()) * sizeof(float)));
    HIP_TRY(h, hipMemcpy(tab, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(tab + tw.size(), window.data(), window.size() * sizeof(float), hipMemcpyHostToDevice));
    h->voc_tables[win] = tab;
  }
  const float* tw = tab; const float* window = tab + 2048;
  const size_t nfr = (size_t)B * T * win;
  WS(fr0, nfr); WS(fr1, nfr);
  const float* ang = d_init_angles;
  if (!ang) {                                            // audio.py:96: 2 pi rand(*S.shape) (the reference does not seed it)
    const size_t n = (size_t)B * T * (n_fft / 2 + 1);
    WS(a0, n);
    RUN_MISC(h, launch_uniform_angles(a0, n, seed, h->stream));
    ang = a0;
  }
  RUN_MISC(h, launch_gl_pass(d_S, ang, nullptr, fr0, d_frames, tw, window, B, T, hop, win, h->stream));       // y = istft(S e^{j phase0})
  float* cur = fr0; float* nxt = fr1;
  for (int i = 0; i < iters; ++i) {                      // phase = angle(stft(y)); y = istft(S e^{j phase})  (audio.py:99-101)
    RUN_MISC(h, launch_gl_pass(d_S, nullptr, cur, nxt, d_frames, tw, window, B, T, hop, win, h->stream));
    std::swap(cur, nxt);
  }
  RUN_MISC(h, launch_gl_final(cur, window, d_frames, B, T, hop, win, d_wav, h->stream));
  return VNR_OK;
}

int vnr_set_option(vnr_handle h, const char* name, int value) {
  if (!h || !name) return fail(h, VNR_ERR_ARG, "null argument");
  if (!strcmp(name, "split_fp16")) { h->split_enabled = value != 0; return VNR_OK; }
  if (!strcmp(name, "chain")) { h->chain_enabled = value != 0; return VNR_OK; }
  if (!strcmp(name, "attn_presplit")) { h->aoi_enabled = value != 0; return VNR_OK; }
  if (!strcmp(name, "gemm_wide_tiles")) { h->gemm_wide_tiles = value != 0; return VNR_OK; }
  if (!strcmp(name, "chain_rows64")) { h->chain_rows64 = value != 0; return VNR_OK; }
  if (!strcmp(name, "late_dec_kv")) { h->late_dec_kv = value != 0; return VNR_OK; }
  if (!strcmp(name, "prior_inverse")) { h->prior_inverse = value != 0; return VNR_OK; }
  if (!strcmp(name, "split_rows")) { h->split_rows = value != 0; return VNR_OK; }
  if (!strcmp(name, "fuse_xattn")) { h->fuse_xattn = value != 0; return VNR_OK; }
  if (!strcmp(name, "chain_waves4")) { h->chain_waves4 = value != 0; return VNR_OK; }
  if (!strcmp(name, "chain_prefetch")) { h->chain_prefetch = value != 0; return VNR_OK; }
  if (!strcmp(name, "chain_segments")) { h->chain_segments = value != 0; return VNR_OK; }
  if (!strcmp(name, "attn_bwd_recompute")) { h->attn_bwd_recompute = value != 0; return VNR_OK; }
  if (!strcmp(name, "train_chain_bwd")) { h->train_chain_bwd = value != 0; return VNR_OK; }
  if (!strcmp(name, "train_chain")) { if (value < 0 || value > 3) return fail(h, VNR_ERR_ARG, "train_chain: 0..3"); h->train_chain = value; return VNR_OK; }
  if (!strcmp(name, "attn_presplit_self")) { h->aoi_self = value != 0; return VNR_OK; }
  if (!strcmp(name, "op_attn_presplit")) { h->op_attn_presplit = value != 0; return VNR_OK; }
  if (!strcmp(name, "split_encoder")) { h->split_encoder = value != 0; return VNR_OK; }
  if (!strcmp(name, "op_dense_split")) { h->op_dense_split = value != 0; return VNR_OK; }
  if (!strcmp(name, "training")) { h->training = value != 0; return VNR_OK; }
  if (!strcmp(name, "dropout_seed")) { h->drop_seed = (unsigned)value; return VNR_OK; }
  if (!strcmp(name, "n_sample")) { if (value < 1 || value > 64) return fail(h, VNR_ERR_ARG, "n_sample: 1..64"); h->n_sample = value; return VNR_OK; }
  if (!strcmp(name, "deterministic")) { h->deterministic = value != 0; return VNR_OK; }
  if (!strcmp(name, "range_guard")) { h->range_guard = value != 0; for (int m = 0; m < 4; ++m) h->range_state[m] = 0; return VNR_OK; }
  if (!strcmp(name, "range_sentinel")) { h->range_sentinel = value != 0; return VNR_OK; }
  if (!strcmp(name, "kv_overlap")) { h->kv_overlap = value != 0; return VNR_OK; }
  if (!strcmp(name, "train_fp32")) { h->train_fp32 = value != 0; h->derived_fresh = false; return VNR_OK; }
  return fail(h, VNR_ERR_ARG, std::string("unknown option ") + name);
}

int vnr_range_info(vnr_handle h, int* states4, float* lo, float* hi, int64_t* surveys) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  if (states4) for (int m = 0; m < 4; ++m) states4[m] = h->range_state[m];
  if (lo) *lo = h->range_lo;
  if (hi) *hi = h->range_hi;
  if (surveys) *surveys = h->range_surveys;
  return VNR_OK;
}
// the range sentinel (vnr_context::range_flag): checkpoints that found the word raised so far, whether the training step has moved to its
// exact-fp32 path, and the modules (bit 0 encoder .. bit 3 posterior) whose split-path results still wait for a checkpoint
int vnr_range_sentinel(vnr_handle h, int64_t* trips, int* train_fp32, int* pending_modules) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  if (trips) *trips = h->range_trips;
  if (train_fp32) *train_fp32 = h->train_fp32 ? 1 : 0;
  if (pending_modules) *pending_modules = (int)h->mods_pending;
  return VNR_OK;
}

// ---- instrumentation -------------------------------------------------------------------------------------------
int vnr_profile_enable(vnr_handle h, int on) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  h->profiling = on != 0;
  return VNR_OK;
}
int vnr_profile_reset(vnr_handle h) {
  if (!h) return fail(h, VNR_ERR_ARG, "null handle");
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (auto& r : h->prof) { h->event_pool.push_back(r.e0); h->event_pool.push_back(r.e1); }
  h->prof.clear();
  return VNR_OK;
}
int vnr_profile_get(vnr_handle h, const char* kernel_class, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (!h || !kernel_class) return fail(h, VNR_ERR_ARG, "null argument");
  int cls = -1;
  for (int i = 0; i < CLS_COUNT; ++i) if (!strcmp(kernel_class, kClsNames[i])) cls = i;
  if (cls < 0) return fail(h, VNR_ERR_ARG, std::string("unknown kernel class ") + kernel_class);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  double ms = 0, fl = 0, by = 0; int64_t n = 0;
  for (auto& r : h->prof) {
    if (r.cls != cls) continue;
    float t = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&t, r.e0, r.e1));
    ms += t; fl += r.flops; by += r.bytes; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  return VNR_OK;
}
int vnr_launch_count(vnr_handle h, int64_t* count) {
  if (!h || !count) return fail(h, VNR_ERR_ARG, "null argument");
  *count = h->launches;
  return VNR_OK;
}

}  // extern "C"
