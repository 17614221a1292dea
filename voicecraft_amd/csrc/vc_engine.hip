// vc_engine.hip — host side of libvcengine.so: the C ABI of include/vc_engine.h, HBM arena
// management, weight packing, prefill scheduling, the decode loop (eager or hipGraph replay) and
// output assembly.  No torch types anywhere; the caller hands raw device pointers and a stream.
//
// Control flow restated from VoiceCraft.inference_tts / inference_tts_batch / inference
// (models/voicecraft.py:908-1153, :1156-1439, :561-906); see DESIGN.md §2 for the mapping.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "vc_common.h"

namespace {

std::string g_create_err;

struct RawTensor {
  std::vector<int64_t> shape;
  float* dev = nullptr;   // fp32 staging copy in HBM
  long numel = 0;
};

struct Layer {
  uint4 *Wqkv = nullptr, *Wo = nullptr, *W1 = nullptr, *W2 = nullptr;
  uint4 *Wo8 = nullptr, *W28 = nullptr;                                 // Wo / W2 once more in 8-channel tiles (finished-row producers, vc_gemm.hip)
  uint4 *Wqkv16 = nullptr;                                              // Wqkv . gamma in 16-channel tiles (prefill and wide-decode passes, option "qkv16")
  uint4 *Wqkv8 = nullptr;                                               // Wqkv . gamma in 8-channel tiles (one-row paired QKV kernel, option "qkv_p8")
  float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;   // bqkv / b1 hold the FOLDED biases (W beta + b)
  float *wg_qkv = nullptr, *wg_1 = nullptr;                             // row sums of the folded weights W . gamma
  void *kc = nullptr, *vc = nullptr;   // KV cache of this layer: WT [max_seqs][H][S_max][hd]
};

struct Plan { int n_tiles, KT, ksplit, nchunk; };

}  // namespace

struct vc_engine {
  vc_model_cfg cfg{};
  int device = 0;
  int dtype = -1;
  bool finalized = false;
  std::string err;
  std::map<std::string, RawTensor> raw;
  std::vector<void*> allocs;

  int d = 0, H = 0, hd = 0, L = 0, K = 0, V = 0, P = 0, S_max = 0, B_max = 0;
  int esz = 2;

  std::vector<Layer> layers;
  uint4 *Wh1 = nullptr, *Wh2 = nullptr;
  float *bh1 = nullptr, *bh2 = nullptr, *wg_h1 = nullptr;               // bh1: folded with the final LayerNorm
  long wh2_group_stride = 0;
  float *text_emb = nullptr, *audio_emb = nullptr, *mask_emb = nullptr, *pe = nullptr;
  float alpha_text = 1.f, alpha_audio = 1.f;
  Plan p_qkv{}, p_o{}, p_f1{}, p_f2{}, p_h1{}, p_h2{};
  Plan p_qkv16;                         // the same matrix on 16-channel tiles (Layer.Wqkv16)

  // activations / scratch
  float *emb = nullptr;                 // prefill rows [emb_cap][d] (one or several prompts back to back)
  int emb_cap = 0;
  int *pre_row_seq = nullptr, *pre_row_pos = nullptr;
  float *hA = nullptr, *hB = nullptr, *q = nullptr, *parts = nullptr, *att_o = nullptr, *att_ml = nullptr;
  void *act = nullptr, *hh = nullptr, *xn = nullptr;
  float *logits = nullptr;              // [B_max][K][V]
  float *dec_h = nullptr;               // [max(VC_ROWS, max_seqs)][d]
  int NS = VC_ROWS;                     // rows the per-sequence buffers are sized for
  int *dec_row_seq = nullptr, *dec_row_pos = nullptr, *logit_row = nullptr;
  SeqState *st = nullptr, *st_fin = nullptr;      // st_fin: final states by slot of sequences a re-pack moved out of the step (repack_k)
  long long* dbg_ts = nullptr;
  int *one = nullptr;                   // device word holding 1: the "always active" flag of prefill launches
  int *share_len = nullptr;             // device word: text positions shared with sequence 0 (AttnArgs.share_len), 0 outside such calls
  int *step_ctr = nullptr;              // SampleArgs.step_ctr
  int *n_active = nullptr, *samp = nullptr, *cond = nullptr, *amax = nullptr, *gen = nullptr, *err_flag = nullptr;
  int gen_cap = 0;
  // pinned host staging
  SeqState *h_st = nullptr, *h_st2 = nullptr;
  int *h_flag = nullptr;                // [0] error/poll word, [1] staging, [8] "sequences still active" written by the device
  SampleDyn *h_dyn = nullptr, *d_dyn = nullptr;   // per-call sampler values (vc_common.h)
  // captured decode steps, kept across calls: key = (sequences, rows per sequence, best-of-N)
  // ... and by the option state they were captured under (vc_set_option): toggling an option back and forth reuses the captures
  std::map<std::pair<std::tuple<int, int, int>, std::string>, hipGraphExec_t> graphs;
  std::string opt_state;                // canonical text of every option that shapes a decode step
  int steps_per_graph = 8;              // VC_GRAPH_STEPS: decode steps captured into one graph launch
  hipEvent_t ev_pace[2]{};

  hipStream_t own_stream = nullptr;     // used when the caller passes the null stream (not capturable)
  // Non-temporal weight loads of the decode kernels, per matrix: bit 0 QKV, 1 out-projection, 2 FFN-up, 3 FFN-down, 4 heads-1, 5 heads-2.
  // Option "nt" (VC_NT).  Until round 4 the compiled QKV / out-projection / heads-2 kernels carried NO such load whatever this said (the
  // compiler merged the kernel's two load arms and dropped the hint): 28 reproduces that mix, 63 = every matrix (default), 0 = none.
  int nt_decode = 63;
  // option "nt", second value: the decode attention's K/V loads carry the hint too - 0 never, 1 always, 2 (default) from two rows per step up
  // (in-process A/Bs, profiles/r04d_bench_*attn_nt*: one row +0.3 % +- 0.09 - there the launch is latency-bound and carries the prefetch
  // role -, 8 rows -4.8 % +- 0.05, 32 rows -6.5 % +- 0.05: several caches stream 58-230 MB per layer through L2 otherwise)
  int attn_nt = 2;
  static constexpr int ln_split_rows = 3;      // slab-form passes with at least this many rows run LayerNorm as its own launch (was an option through round 5)
  // finished-row form of decode passes of 2..fr_rows rows (forward_rows_fr): 0 = off.  VC_FINISHED_ROWS / option "finished_rows"
  int fr_rows = VC_ROWS;
  // (weight tiles per workgroup of the finished-row consumers (QKV, FFN-up): two from 5 rows up - in-process A/Bs of round 4,
  // profiles/r04b_bench_batch*.json.log: 8 rows -3.4 % +- 0.1 with two, 4 rows +0.5 % +- 0.05 - lnw_two(); an option through round 5)
  // option "tile_attn": prefill attention kernel - 1 = tile_attn_k (16 query rows per wave, keys split over the waves), 2 =
  // tile_attn64_k (64 query rows per workgroup, transposed score product, P in registers; bf16 / head_dim 128 only, prompts are then
  // laid out on 64-row boundaries)
  // Default 2 when the call's longest prompt has at least 768 rows ("tile_attn" = "k[,min_rows]"): measured on one layer of giga830M (profiles/r04o_attn64_probe.log)
  // 512 / 800 / 2048 causal rows: 14.3 / 24.6 / 86.7 us (first kernel) against 15.5 / 21.0 / 52.5 us (second).
  int tile_attn = 2, tile_attn_min_rows = 768;
  static constexpr int fr_split_rows = VC_FR_MAX_ROWS;   // finished-row passes of more rows run the attention unsplit (it normalises itself, plain out-projection prologue; unsplit from 5 rows measured +1.7..+4.5 %, r04j)
  // option "fr_one" (round 5): ONE-row steps - the FFN down-projection finishes its row (row_gemm_fr1_k: 8-channel tiles over the whole K,
  // two k-tiles per MFMA fragment, residual + bias added) instead of leaving 4 split-K slabs, so the next layer's QKV projection (and
  // heads-1) reads one 8 KB row instead of h + bias + 4 slabs = 40 KB in each of its 512 workgroups; 0 = off
  int fr_one = 1;
  // option "fr_pair" (round 5): the FFN down-projection of 2..8-row steps with two k-tiles per MFMA fragment (rows_gemm_frp_k) instead
  // of half-filled 8-channel fragments (rows_gemm_fr_k)
  int fr_pair = 1;
  // option "qkv_p8" (round 5): one-row steps behind a finished row (fr_one) run the QKV projection on 8-channel tiles with two k-tiles
  // per MFMA fragment (row_gemm_fr1_k<PRO_LN, EPI_QKV>: every lane's 16 bytes are weights) instead of 12-channel tiles; 2 (round 6): steps of
  // 2..8 finished rows too, behind the producers' centred copy (rows_gemm_qp_k)
  int qkv_p8 = 2;
  // option "qkv16" (round 5): prefill passes and wide decode passes (17..64 rows) run the QKV projection on a 16-channel image of the
  // folded matrix (every A lane of every MFMA a weight) instead of the 12-channel tiles one-row steps were tuned on; + 6 d^2 bytes per
  // layer in bf16 (0.4 GB at giga830M).  Packed for engines that can take wide steps (max_seqs > 16) or with VC_QKV16=1 at creation;
  // otherwise the option stays off and prefill passes read the 12-channel tiles.
  int qkv16 = 1;
  // option "wide_heads" (round 5): decode steps of 17..64 rows run the prediction heads once on the weight-stationary kernel of those
  // steps (one LayerNorm launch + two rows_gemm_mt_k launches) instead of once per 16 rows on the rows-GEMM
  int wide_heads = 1;
  // option "wide_gemm" (round 6): the linear layers of 17..64-row steps on rows_gemm_wd_k (vc_gemm_wd.hip: every row tile of the step in
  // flight at once, X fragments straight from L2 into registers, one barrier per launch) instead of the weight-stationary
  // rows_gemm_mt_k, which walks the row tiles one after the other; 0 = that kernel (two weight tiles per workgroup)
  int wide_gemm = 1;
  // option "wd_stage": 1 (default) = that kernel takes X as whole cache lines through a wave-private LDS stage (rows_gemm_wds_k: 64 rows FFN-up
  // 13.5 -> 9.7 us, the 64-row step -9.5 % +- 0.01, 32 rows -2.7 %; profiles/r06c_*), 0 = MFMA fragments straight from L2 (16 half lines per request)
  int wd_stage = 1;
  // option "shrink" (round 6): a multi-utterance call whose sequences retire at different steps re-packs the live ones onto the rows of a
  // narrower step (the next power of two >= the live count: 64 -> 32 -> 16 -> 8 -> 4 -> 2 -> 1) instead of keeping its launch form until
  // the longest sequence ends; 0 = the fixed width of rounds 1-5.  Results do not depend on it: everything per sequence is indexed by its slot.
  int shrink = 1;
  int cur_rows = 0;                     // rows per step the last decode loop ended on (tts_run reads the states back accordingly)
  int h_parts = 0;                      // split-K slabs the last pass left pending on hB (what the heads' LayerNorm has to sum)
  // option "attn_fast": decode attention with the wave's maximum taken before any exponential (no online rescaling inside a wave) and,
  // in bf16 mode, hardware exp2 (v_exp_f32) instead of expf
  int attn_fast = 1;
  // option "hq" (round 6): the finished-row producers of 2..16-row steps also write their rows CENTRED and in the compute dtype (hqA / hqB =
  // WT(h - c), c = the row's mean as the previous LayerNorm found it: row_mu ping-pong), and the LayerNorm-folding consumers read that copy
  // (PRO_LNQ: half the bytes of the fp32 row in bf16 mode, no conversion) - the fold is exact for any centring constant
  int hq = 1;
  void *hqA = nullptr, *hqB = nullptr;  // centred copies of hA / hB, WT [VC_ROWS][d]
  float* row_mu[2] = {nullptr, nullptr};       // [VC_ROWS] each: ping-pong of the rows' means
  float* mu_zero = nullptr;             // [VC_ROWS] zeros: GemmArgs.row_mu of every launch that centres nothing
  int mu_cur = 0;                       // which row_mu buffer holds the means of the rows in hB (run_heads16 reads it behind forward_rows_fr)
  bool hq_h = false;                    // hqB holds the centred copy of the finished rows in hB
  // option "att_p16" (round 6; bf16 mode): finished-row passes of 2..8 rows hand the attention's split partials to the out-projection as
  // bf16 (131 -> 66 KB per out-projection workgroup at 8 rows; the merged row is rounded to bf16 for the MFMA anyway)
  int att_p16 = 1;
  static constexpr int attn_blocks_multi = 512, attn_blocks_one = 256;   // attention workgroups aimed at (several rows / one row); 256 -> 64 at one row measured +1.1..+1.8 % (r05f)
  int prefill_rows_per_pass = VC_MAX_ROWS;   // VC_PREFILL_ROWS=16 falls back to the decode kernels for the prompt
  hipEvent_t ev[3]{};
  float ms[3]{0, 0, 0};
  double host_ms[8]{};                  // host wall clock of the last call's phases (vc_debug_read "host_ms")
  double bytes_total = 0;               // HBM bytes owned by the engine
  bool finished_rows_h = false;         // run_heads16: hB holds the finished residual (no slabs, no pending bias)
  // training objective (vc_eval_forward), allocated on first use
  int *ce_tgt = nullptr, *ce_hit = nullptr;
  float *ce_nll = nullptr;
  double *ce_sum = nullptr;
  long long *ce_hits = nullptr, *ce_cnt = nullptr;
};

namespace {

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int fail(vc_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_err = buf;
  return code;
}

#define HIPCHK(e, call)                                                                  \
  do {                                                                                   \
    hipError_t _err = (call);                                                            \
    if (_err != hipSuccess)                                                              \
      return fail(e, VC_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_err), \
                  __FILE__, __LINE__);                                                   \
  } while (0)

template <typename T>
int dalloc(vc_engine* e, T** p, size_t n) {
  void* v = nullptr;
  hipError_t err = hipMalloc(&v, std::max<size_t>(n * sizeof(T), 16));
  if (err != hipSuccess) return fail(e, VC_EHIP, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(err));
  e->allocs.push_back(v);
  e->bytes_total += (double)(n * sizeof(T));
  *p = reinterpret_cast<T*>(v);
  return VC_OK;
}

Plan make_plan(int N, int Kdim, int dtype, bool allow_split, const char* env_override, int th = 16) {
  Plan p;
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16;
  p.n_tiles = (N + th - 1) / th;
  p.KT = Kdim / KW;
  p.ksplit = 1;
  if (allow_split) {
    // grow the grid towards ~2 blocks per CU while every wave still streams >= 8 KiB per chunk
    while (p.n_tiles * p.ksplit < 512 && p.ksplit < VC_MAX_KSPLIT && (p.KT / (p.ksplit * 2)) >= 32 &&
           (p.KT % (p.ksplit * 2 * 8)) == 0)
      p.ksplit *= 2;
    if (env_override) {
      const char* v = getenv(env_override);
      if (v) {
        const int ks = atoi(v);
        if (ks >= 1 && ks <= VC_MAX_KSPLIT && p.KT % (ks * 8) == 0) p.ksplit = ks;
      }
    }
  }
  const int per = p.KT / p.ksplit;
  int ktw = 16;
  while (ktw > 2 && per % (4 * ktw) != 0) ktw >>= 1;
  p.nchunk = per / (4 * ktw);
  return p;
}

const RawTensor* find_raw(vc_engine* e, const std::string& key) {
  auto it = e->raw.find(key);
  return it == e->raw.end() ? nullptr : &it->second;
}

int need(vc_engine* e, const std::string& key, std::initializer_list<int64_t> shape, const RawTensor** out) {
  const RawTensor* t = find_raw(e, key);
  if (!t) return fail(e, VC_EMISSING, "weight '%s' was never loaded", key.c_str());
  std::vector<int64_t> want(shape);
  if (t->shape != want) {
    std::string got, exp;
    for (auto v : t->shape) got += std::to_string(v) + ",";
    for (auto v : want) exp += std::to_string(v) + ",";
    return fail(e, VC_EINVAL, "weight '%s' has shape [%s] but the config implies [%s]", key.c_str(), got.c_str(), exp.c_str());
  }
  *out = t;
  return VC_OK;
}

int pack_matrix(vc_engine* e, const std::string& key, int N, int Kdim, uint4** out, int th = 16,
                const float* colscale = nullptr) {
  const RawTensor* t;
  int rc = need(e, key, {N, Kdim}, &t);
  if (rc) return rc;
  const int KW = e->dtype == VC_DTYPE_BF16 ? 32 : 16;
  const long units = (long)((N + th - 1) / th) * (Kdim / KW) * 4 * th;
  rc = dalloc(e, out, (size_t)units);
  if (rc) return rc;
  HIPCHK(e, vc_launch_pack(t->dev, colscale, *out, N, Kdim, e->dtype, th, 0));
  return VC_OK;
}

// A linear layer behind a LayerNorm (vc_gemm.hip, "LayerNorm fold"): packs W . gamma, and produces
// wg = rowsum(W . gamma) and the folded bias cb = W beta + bias.
int pack_folded(vc_engine* e, const std::string& wkey, const std::string& bkey, const std::string& ln_prefix,
                int N, int Kdim, uint4** Wout, float** wg, float** cb, int th = 16) {
  const RawTensor *tw, *tb, *tg, *tbe;
  int rc;
  if ((rc = need(e, wkey, {N, Kdim}, &tw))) return rc;
  if ((rc = need(e, bkey, {N}, &tb))) return rc;
  if ((rc = need(e, ln_prefix + "weight", {Kdim}, &tg))) return rc;
  if ((rc = need(e, ln_prefix + "bias", {Kdim}, &tbe))) return rc;
  if ((rc = pack_matrix(e, wkey, N, Kdim, Wout, th, tg->dev))) return rc;
  if ((rc = dalloc(e, wg, (size_t)N))) return rc;
  if ((rc = dalloc(e, cb, (size_t)N))) return rc;
  HIPCHK(e, vc_launch_fold_vecs(tw->dev, tg->dev, tbe->dev, tb->dev, *wg, *cb, N, Kdim, e->dtype, 0));
  return VC_OK;
}

int keep_vec(vc_engine* e, const std::string& key, int n, float** out) {
  const RawTensor* t;
  int rc = need(e, key, {n}, &t);
  if (rc) return rc;
  rc = dalloc(e, out, (size_t)n);
  if (rc) return rc;
  HIPCHK(e, hipMemcpy(*out, t->dev, (size_t)n * 4, hipMemcpyDeviceToDevice));
  return VC_OK;
}

// ---------------------------------------------------------------- one forward pass over <= 16 rows
struct RowSrc {
  const float* h_in;
  const int* row_seq;
  const int* row_pos;
  int n_rows;       // rows carried
  int nsplit;
  const int* n_active;
  int nt;           // force the streaming-load policy (microbenchmarks)
  int tiled;        // 1: every 16-row tile holds consecutive positions of ONE sequence (prefill_batch): MFMA attention
};

// The consumers of 2..16 finished rows fold the producers' centred copy of the rows (PRO_LNQ) when a row of it is whole 1 KB requests of a
// wave (64 lanes x 16 bytes: d a multiple of 512 in bf16, of 256 in fp32); other widths keep the fp32 rows (PRO_LNW).
bool hq_on(const vc_engine* e) {
  const int bytes = e->d * (e->dtype == VC_DTYPE_BF16 ? 2 : 4);
  return e->hq != 0 && bytes % 1024 == 0 && bytes <= 8192;
}

// Option "qkv_p8" = 2: behind a centred copy, the QKV projection of 2..8 finished rows runs on the 8-channel image with two k-tiles per
// fragment (rows_gemm_qp_k: every lane of every weight request a weight) instead of the 12-channel tiles of rows_gemm_k.
// Measured (profiles/r06i_ab_qp.log, per step): giga830M 2 / 4 / 6 / 8 rows -2.0 / -1.1 / -0.3 / +0.5 %, giga330M 8 rows -1.3 %: at d >= 2048 the form
// stops at 6 rows (the 8-row launch itself: 7.42 us against 7.33 us of the 12-channel tiles).
bool qp_on(const vc_engine* e, const Layer& ly, int rows) {
  return e->qkv_p8 >= 2 && hq_on(e) && ly.Wqkv8 != nullptr && rows <= (e->d >= 2048 ? 6 : 8) && vc_gemm_qp_ok(rows, 3 * e->d, e->d, e->dtype) != 0;
}

GemmArgs base_args(vc_engine* e, const RowSrc& rs, const Plan& p, int N, int Kdim) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.N = N; g.K = Kdim; g.n_tiles = p.n_tiles; g.KT = p.KT; g.nchunk = p.nchunk;
  g.r_lds = std::min(rs.n_rows, VC_ROWS);
  g.rows_cap = VC_SLAB_ROWS;
  g.row_seq = rs.row_seq; g.row_pos = rs.row_pos; g.n_rows = rs.n_rows;
  g.n_active = rs.n_active ? rs.n_active : e->one;
  g.nt = (rs.n_active != nullptr || rs.nt) ? 1 : 0;   // decode steps (and the kernel microbenchmarks) stream once; per matrix: nt_bit
  g.d = e->d; g.H = e->H; g.hd = e->hd; g.S_max = e->S_max;
  g.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7;            // head_dim is 32, 64 or 128 (vc_create)
  g.cache_seq_stride = (long)e->H * e->S_max * e->hd;
  g.dbg_ts = e->dbg_ts;
  g.row_mu = e->mu_zero;
  g.wd_stage = e->wd_stage;
  return g;
}

// ONE-row steps whose FFN down-projection finishes the row (option "fr_one" bit 0): the launcher's own statement of what it takes
// fr_one != 0: wherever the kernel can run.  (The first build - FFN-down alone - gained 2.1 % at giga830M and LOST 1.5 % at giga330M,
// whose d / 8 = 128 workgroups fill half the chip; together with the paired QKV projection it takes along the form gains there
// too: -0.5 % +- 0.1 / -0.9 % +- 0.09 on two boxes, profiles/r05e_ab_330M.log, r05f_ab_330M.log.)
inline bool fd_one(const vc_engine* e, int rows) {
  if (rows != 1 || e->fr_one == 0) return false;
  return !e->layers.empty() && e->layers[0].W28 != nullptr && vc_gemm_fr1_ok(e->d, 4 * e->d, e->dtype, VC_FR_WAVES) != 0;
}

inline int attn_nt_for(const vc_engine* e, int rows) { return e->attn_nt == 1 || (e->attn_nt == 2 && rows >= 2); }
enum { NT_QKV = 1, NT_O = 2, NT_F1 = 4, NT_F2 = 8, NT_H1 = 16, NT_H2 = 32 };
inline void nt_bit(const vc_engine* e, GemmArgs& g, int bit) { if (!(e->nt_decode & bit)) g.nt = 0; }

int attn_nsplit(vc_engine* e, int rows) {
  // 8-wave blocks: one decode row is covered by ~256 of them; several rows get ~512 (two per CU), which
  // halves the positions each block walks while the merge in the out-projection stays <= 4 partials
  int ns = (rows > 1 ? e->attn_blocks_multi : e->attn_blocks_one) / std::max(1, rows * e->H);   // (fewer splits = a cheaper merge in the out-projection)
  return std::max(1, std::min(ns, VC_MAX_NSPLIT));
}

// Rows a pass may carry in the finished-row form: X of the FFN down-projection (rows x 4d elements) has to fit the LDS
// of one workgroup, and the out-projection merges rows x nsplit <= 16 attention partials per thread in one batch.
int fr_max_rows(const vc_engine* e) {
  // the FFN down-projection decides (X = rows x 4d elements): one piece up to 8 rows in bf16 at d = 2048 (4 in the exact mode), two
  // halves beyond while a wave can hold its fragments of both; vc_gemm_fr_form is the launcher's own statement of that
  int r = std::min(e->fr_rows, VC_ROWS);
  while (r >= 2 && vc_gemm_fr_form(r, e->d, 4 * e->d, e->dtype, PRO_PLAIN, 1) == 0) --r;
  return r >= 2 ? r : 0;
}
bool lnw_two(const vc_engine*, int rows) { return rows >= 5; }
// splits of the decode attention in the finished-row form: the out-projection merges rows x splits <= 16 partials per thread in
// one batch of loads; from 9 rows up the attention is unsplit (rows x heads >= 144 workgroups) and normalises itself
int fr_nsplit(vc_engine* e, int rows) {
  if (rows > std::min(e->fr_split_rows, VC_FR_MAX_ROWS)) return 1;
  int ns = std::min(attn_nsplit(e, rows), 16 / rows);
  int p = 1;
  while (p * 2 <= ns) p *= 2;
  return std::max(2, p);
}

// The same pass in the finished-row form (2..fr_max_rows rows): five launches per layer, no split-K slabs, no LayerNorm
// launch.  The residual stream is a whole row at every launch boundary: QKV and FFN-up fold the LayerNorm of finished rows
// (PRO_LNW, one wave per row), the out-projection and the FFN down-projection own 8 output channels over the whole K and
// add residual + bias in their epilogue (rows_gemm_fr_k).  Leaves the finished rows in hB.
int forward_rows_fr(vc_engine* e, const RowSrc& rs, hipStream_t s) {
  const int d = e->d;
  // option "hq": the producers leave a centred copy of their rows in the compute dtype and the consumers fold THAT (PRO_LNQ).  `cur` names
  // the row_mu buffer with the latest means: a consumer reads its producer's constant there and leaves the row's new mean in the other one.
  const bool hq = hq_on(e);
  int cur = 0;
  for (int l = 0; l < e->L; ++l) {
    Layer& ly = e->layers[l];
    const float* h_res = (l == 0) ? rs.h_in : e->hB;      // residual entering the layer
    {  // q,k,v = Wqkv LN1(h) + b ; K/V go straight into the cache
      GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
      g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.wg = ly.wg_qkv;
      g.h_in = h_res;
      g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
      g.mt = rs.n_rows > VC_FR_MAX_ROWS ? 4 : lnw_two(e, rs.n_rows) ? 3 : 0;
      if (hq) g.row_mu_out = e->row_mu[cur ^ 1];
      if (hq && l > 0) {       // the previous layer's FFN down-projection left hqB = WT(hB - row_mu[cur])
        g.x_in = e->hqB; g.x_ld = d; g.row_mu = e->row_mu[cur];
        if (qp_on(e, ly, rs.n_rows)) { g.Wp = ly.Wqkv8; HIPCHK(e, vc_launch_gemm_qp(g, e->dtype, s)); }
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNQ, EPI_QKV, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNW, EPI_QKV, 1, 1, s));
      }
      cur ^= 1;
    }
    {
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc;
      a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
      a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = rs.nsplit;
      a.scale = 1.0f / sqrtf((float)e->hd);
      a.row_seq = rs.row_seq; a.row_pos = rs.row_pos; a.n_rows = rs.n_rows;
      a.n_active = rs.n_active ? rs.n_active : e->one; a.dbg_ts = e->dbg_ts;
      a.att_o = e->att_o; a.att_ml = e->att_ml; a.share_len = e->share_len;
      a.nt = (rs.n_active != nullptr || rs.nt) ? attn_nt_for(e, rs.n_rows) : 0;
      a.fast = e->attn_fast;
      a.part16 = (e->att_p16 && e->dtype == VC_DTYPE_BF16 && rs.nsplit > 1) ? 1 : 0;
      if (rs.nsplit == 1) a.x_out = e->xn;        // unsplit (9..16 rows): the workgroup saw every position and normalises itself
      HIPCHK(e, vc_launch_attn(a, e->dtype, rs.n_rows, s));
    }
    {  // h' = h + bo + Wo merge(attention partials of all heads)
      GemmArgs g = base_args(e, rs, e->p_o, d, d);
      g.Wp = ly.Wo8; g.bias = ly.bo;
      g.h_in = h_res; g.h_out = e->hA;
      if (hq) { g.hq_out = e->hqA; g.row_mu = e->row_mu[cur]; }      // centred on the mean the QKV consumer found for h
      if (rs.nsplit == 1) {
        g.x_in = e->xn; g.x_ld = d;
        HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_PLAIN, s));
      } else {
        g.att_o = e->att_o; g.att_ml = e->att_ml; g.nsplit = rs.nsplit;
        g.att_p16 = (e->att_p16 && e->dtype == VC_DTYPE_BF16) ? 1 : 0;
        HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_ATT, s));
      }
    }
    {  // a = relu(W1 LN2(h') + b1)
      GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
      g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.wg = ly.wg_1;
      g.h_in = e->hA;
      g.out = e->act; g.out_ld = 4 * d;
      g.mt = rs.n_rows > VC_FR_MAX_ROWS ? 4 : lnw_two(e, rs.n_rows) ? 3 : 0;
      if (hq) {
        g.x_in = e->hqA; g.x_ld = d; g.row_mu = e->row_mu[cur]; g.row_mu_out = e->row_mu[cur ^ 1];
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNQ, EPI_RELU, 1, 1, s));
        cur ^= 1;
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNW, EPI_RELU, 1, 1, s));
      }
    }
    {  // h'' = h' + b2 + W2 a
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W28; g.bias = ly.b2;
      g.x_in = e->act; g.x_ld = 4 * d;
      g.h_in = e->hA; g.h_out = e->hB;
      if (hq) { g.hq_out = e->hqB; g.row_mu = e->row_mu[cur]; }      // centred on the mean the FFN-up consumer found for h'
      if (e->fr_pair && vc_gemm_frp_ok(rs.n_rows, d, 4 * d, e->dtype)) HIPCHK(e, vc_launch_gemm_frp(g, e->dtype, s));      // 2..8 rows: two k-tiles per fragment
      else HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_PLAIN, s));
    }
  }
  e->hq_h = hq;
  e->mu_cur = cur;
  return VC_OK;
}

// One pass of up to 16 rows through every layer (decode step, 3-row span switch, short prompts).
int forward_rows(vc_engine* e, const RowSrc& rs, hipStream_t s) {
  const int d = e->d;
  e->finished_rows_h = false;
  e->hq_h = false;
  if (rs.n_rows >= 2 && rs.n_rows <= fr_max_rows(e)) {
    RowSrc fr = rs;
    fr.nsplit = fr_nsplit(e, rs.n_rows);
    e->finished_rows_h = true;            // tells run_heads16 (the caller's next step) that hB holds finished rows
    return forward_rows_fr(e, fr, s);
  }
  // The in-GEMM LayerNorm prologue walks the rows one after the other (each a dependent round trip
  // in every one of the 384-512 workgroups): from ln_split_rows rows on, a per-row LayerNorm launch
  // plus the plain prologue is cheaper (measured: 8 rows 20 us -> ~12 us per GEMM).
  const bool split_ln = rs.n_rows >= e->ln_split_rows;
  // one row, option fr_one: the FFN down-projection writes the FINISHED residual row into hB (row_gemm_fr1_k), so the row entering a
  // layer is whole: the QKV prologue reads it alone and writes nothing, the FFN-up prologue adds the out-projection's two slabs to
  // it and leaves h' in hA for the down-projection's epilogue
  const bool fd = fd_one(e, rs.n_rows);
  e->finished_rows_h = fd;
  e->h_parts = e->p_f2.ksplit;
  for (int l = 0; l < e->L; ++l) {
    Layer& ly = e->layers[l];
    const float* h_res = (l == 0) ? rs.h_in : e->hB;
    {  // x = LN1(h); q,k,v = Wqkv x + b ; K/V go straight into the cache
      GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
      g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv;
      g.h_in = h_res;
      g.h_out = fd ? nullptr : e->hA;
      g.parts = e->parts;
      g.n_parts = (l == 0 || fd) ? 0 : e->p_f2.ksplit;
      g.prev_bias = (l == 0) ? ly.bo : e->layers[l - 1].b2;    // (any readable [d] vector when unused)
      g.has_prev_bias = (l == 0 || fd) ? 0 : 1;
      g.wg = ly.wg_qkv;
      g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
      if (fd && e->qkv_p8 && ly.Wqkv8 && !split_ln) {
        // the row entering the layer is finished (layer 0: the sampler's dec_h row): 8-channel tiles, two k-tiles per fragment
        g.Wp = ly.Wqkv8;
        HIPCHK(e, vc_launch_gemm_fr1(g, e->dtype, PRO_LN, EPI_QKV, s));
      } else if (split_ln) {   // several rows: LayerNorm once per row, then the plain-prologue GEMM
        g.x_out = e->xn;
        HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
        g.x_in = e->xn; g.x_ld = d;
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LN, EPI_QKV, 1, 1, s));
      }
    }
    {  //                                                                 
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc;
      a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
      a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = rs.nsplit;
      a.scale = 1.0f / sqrtf((float)e->hd);
      a.row_seq = rs.row_seq; a.row_pos = rs.row_pos; a.n_rows = rs.n_rows;
      a.n_active = rs.n_active ? rs.n_active : e->one; a.dbg_ts = e->dbg_ts;
      a.att_o = e->att_o; a.att_ml = e->att_ml; a.share_len = e->share_len;
      a.nt = (rs.n_active != nullptr || rs.nt) ? attn_nt_for(e, rs.n_rows) : 0;
      a.fast = e->attn_fast;
      HIPCHK(e, vc_launch_attn(a, e->dtype, rs.n_rows, s));
    }
    {  // out-projection of the merged attention output -> split-K partial slabs
      GemmArgs g = base_args(e, rs, e->p_o, d, d);
      g.Wp = ly.Wo; nt_bit(e, g, NT_O);
      g.att_o = e->att_o; g.att_ml = e->att_ml; g.nsplit = rs.nsplit;
      g.part_out = e->parts;
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_ATT, EPI_PART, e->p_o.ksplit, 1, s));
    }
    {  // h' = h + attn + bo ; a = relu(W1 LN2(h') + b1)                  
      GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
      g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1;
      g.h_in = fd ? h_res : e->hA; g.h_out = fd ? e->hA : e->hB;
      g.parts = e->parts; g.n_parts = e->p_o.ksplit; g.prev_bias = ly.bo; g.has_prev_bias = 1;
      g.wg = ly.wg_1;
      g.out = e->act; g.out_ld = 4 * d;
      if (split_ln) {
        g.x_out = e->xn;
        HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
        g.x_in = e->xn; g.x_ld = d;
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_RELU, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LN, EPI_RELU, 1, 1, s));
      }
    }
    if (fd) {  // h'' = h' + b2 + W2 a: the finished row (one row, row_gemm_fr1_k)
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W28; g.bias = ly.b2;
      g.x_in = e->act; g.x_ld = 4 * d;
      g.h_in = e->hA; g.h_out = e->hB;
      HIPCHK(e, vc_launch_gemm_fr1(g, e->dtype, PRO_PLAIN, EPI_RES, s));
    } else {  // W2 a -> partial slabs, folded into the next LayerNorm prologue together with b2
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W2; nt_bit(e, g, NT_F2);
      g.x_in = e->act; g.x_ld = 4 * d;
      g.part_out = e->parts;
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_f2.ksplit, 1, s));
    }
  }
  return VC_OK;
}

// final LayerNorm + the K prediction heads (voicecraft.py:181-185, :1084-1086) for n rows;
// row r reads hidden row gather[r] (or in_row0 + r) and writes logits row (out_row0 + r).
int run_heads16(vc_engine* e, const int* gather, int n, int in_row0, int out_row0, const int* n_active, hipStream_t s) {
  if (gather && n > 1) return fail(e, VC_EINVAL, "internal: a gathered head pass carries one row");
  RowSrc rs{};
  rs.n_rows = n; rs.n_active = n_active;
  {  //                                                                   
    GemmArgs g = base_args(e, rs, e->p_h1, e->K * e->P, e->d);
    g.Wp = e->Wh1; nt_bit(e, g, NT_H1); g.bias = e->bh1;
    g.h_in = e->hB + (size_t)in_row0 * e->d; g.h_out = nullptr;
    g.parts = e->parts + (size_t)in_row0 * e->d; g.n_parts = e->h_parts; g.prev_bias = e->layers[e->L - 1].b2; g.has_prev_bias = 1;
    if (e->finished_rows_h) { g.n_parts = 0; g.has_prev_bias = 0; }       // the finished-row form left the whole residual in hB
    g.wg = e->wg_h1; g.gather_rows = gather;
    g.out = (char*)e->hh + (size_t)out_row0 * e->K * e->P * e->esz; g.out_ld = e->K * e->P;
    if (e->finished_rows_h && !gather && n >= 2 && n <= VC_FR_MAX_ROWS) {     // finished rows: LayerNorm fold per wave, no extra launch
      if (e->hq_h && in_row0 == 0) {      // ... on the last FFN down-projection's centred copy
        g.x_in = e->hqB; g.x_ld = e->d; g.row_mu = e->row_mu[e->mu_cur];
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNQ, EPI_GELU, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LNW, EPI_GELU, 1, 1, s));
      }
    } else if (!gather && n >= e->ln_split_rows) {
      g.x_out = e->xn;
      HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
      g.x_in = e->xn; g.x_ld = e->d;
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_GELU, 1, 1, s));
    } else {
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LN, EPI_GELU, 1, 1, s));
    }
  }
  {  //                                                                          
    GemmArgs g = base_args(e, rs, e->p_h2, e->V, e->P);
    g.Wp = e->Wh2; nt_bit(e, g, NT_H2); g.bias = e->bh2;
    g.w_group_stride = e->wh2_group_stride; g.bias_group_stride = e->V;
    g.x_in = (char*)e->hh + (size_t)out_row0 * e->K * e->P * e->esz; g.x_ld = e->K * e->P; g.x_group_stride = e->P;
    g.out = e->logits + (size_t)out_row0 * e->K * e->V;
    HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_LOGITS, 1, e->K, s));
  }
  return VC_OK;
}
// 17..64 rows (wide decode steps): the final LayerNorm once per row (ln_rows_k), then both head matrices on the weight-stationary
// kernel of those steps (rows_gemm_mt_k: every weight read ONCE per step) instead of once per 16 rows - at 64 rows the 16-row passes
// streamed the 33.6 MB of head weights four times (option "wide_heads").
bool use_wd(const vc_engine* e, int rows);
int run_heads_wide(vc_engine* e, int n, const int* n_active, hipStream_t s) {
  RowSrc rs{};
  rs.n_rows = n; rs.n_active = n_active;
  const bool wd = use_wd(e, n);
  {
    GemmArgs g = base_args(e, rs, e->p_h1, e->K * e->P, e->d);
    g.Wp = e->Wh1; nt_bit(e, g, NT_H1); g.bias = e->bh1;
    g.h_in = e->hB; g.h_out = nullptr;
    g.parts = e->parts; g.n_parts = e->h_parts; g.prev_bias = e->layers[e->L - 1].b2; g.has_prev_bias = 1;
    g.wg = e->wg_h1;
    g.out = e->hh; g.out_ld = e->K * e->P;
    g.x_out = e->xn;
    HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
    g.x_in = e->xn; g.x_ld = e->d; g.mt = 2;
    if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_GELU, 1, 1, s));
    else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_GELU, 1, 1, s));
  }
  {
    GemmArgs g = base_args(e, rs, e->p_h2, e->V, e->P);
    g.Wp = e->Wh2; nt_bit(e, g, NT_H2); g.bias = e->bh2;
    g.w_group_stride = e->wh2_group_stride; g.bias_group_stride = e->V;
    g.x_in = e->hh; g.x_ld = e->K * e->P; g.x_group_stride = e->P;
    g.out = e->logits; g.mt = 2;
    if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_LOGITS, 1, e->K, s));
    else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_LOGITS, 1, e->K, s));
  }
  return VC_OK;
}
int run_heads(vc_engine* e, const int* gather, int n, int out_row0, const int* n_active, hipStream_t s) {
  if (e->wide_heads && !gather && out_row0 == 0 && n > VC_ROWS && n <= VC_MAX_SEQS && n_active != nullptr && !e->finished_rows_h)
    return run_heads_wide(e, n, n_active, s);
  for (int r0 = 0; r0 < n; r0 += VC_ROWS) {       // wide batches: the heads run 16 rows at a time
    int rc = run_heads16(e, gather, std::min(VC_ROWS, n - r0), r0, out_row0 + r0, n_active, s);
    if (rc) return rc;
  }
  return VC_OK;
}

// Prefill pass over up to VC_MAX_ROWS prompt rows: LayerNorm once per row (ln_rows_k), then the
// multi-tile rows-GEMM (weights streamed once per pass).  Same buffers and slabs as the decode pass.
// Which image of the QKV matrix a many-row pass reads (option "qkv16").  The 16-channel image has 3/4 of the 12-channel image's tiles,
// and the block GEMM picks its workgroup tile from the tile count (vc_gemm_pf.hip launch_blk_e: 128-channel tiles from 240 workgroups):
// where the smaller count would push a pass back to the 64-channel form the 12-channel image stays (giga830M, 385..512 rows: 24.0 us
// against 34.0, profiles/r05j_qkv16_probe.log); everywhere else the full fragments win or tie (240 rows -6.7 %, 800 rows -1.7 %,
// 1 280 rows -26.7 % - there the pass also leaves the 256 x 256 kernel, whose 160 workgroups fill 5/8 of the chip -, 32-row decode
// steps -1.8 % per step).
bool use_qkv16(const vc_engine* e, int rows, int mtv) {
  if (!e->qkv16 || e->layers.empty() || !e->layers[0].Wqkv16) return false;
  if (mtv == 2) return true;                          // wide decode passes: a weight stream, fewer instructions per byte
  const long blk = (rows + 127) / 128;
  const long w12 = (long)(e->p_qkv.n_tiles / 8) * blk, w16 = (long)(e->p_qkv16.n_tiles / 8) * blk;
  return !(w12 >= 240 && w16 < 240);
}

// K slices of a split-K producer on the wide-decode kernel (rows_gemm_wd_k: two 16-channel tiles per workgroup, the slice split over
// 8 waves): the fewest slices that put a workgroup on every CU, among those the kernel has a form for; 0 = none.
int wd_ksplit(const vc_engine* e, int N, int Kdim) {
  const int n_wg = (N / 16 + 1) / 2;
  int best = 0;
  for (int ks = 1; ks <= VC_MAX_KSPLIT; ks *= 2) {
    if (!vc_gemm_wd_kpw(Kdim, e->dtype, ks)) continue;
    best = ks;
    if (n_wg * ks >= 256) break;
  }
  return best;
}
// Does a pass of `rows` rows run its linear layers on rows_gemm_wd_k?  (Every matrix of the layer must have a form - the QKV projection
// needs the 16-channel image.)
bool use_wd(const vc_engine* e, int rows) {
  if (!e->wide_gemm || rows <= VC_ROWS || rows > VC_MAX_SEQS || e->layers.empty() || !e->layers[0].Wqkv16 || !e->qkv16) return false;
  const int d = e->d;
  return vc_gemm_wd_kpw(d, e->dtype, 1) && wd_ksplit(e, d, d) && wd_ksplit(e, d, 4 * d) && vc_gemm_wd_kpw(e->P, e->dtype, 1);
}

int prefill_rows(vc_engine* e, const RowSrc& rs, hipStream_t s) {
  const int d = e->d;
  e->finished_rows_h = false;           // this pass leaves h + split-K slabs
  e->hq_h = false;
  // prefill passes run on the block GEMM (mt = 1); decode passes of 17..64 rows (n_active set) are still weight
  // streams: they take the weight-stationary multi-tile kernel (mt = 2), which reads every weight once at the decode rate
  const int mtv = (rs.n_active != nullptr && rs.n_rows <= VC_MAX_SEQS && !getenv("VC_WIDE_BLK")) ? 2 : 1;
  const bool wd = mtv == 2 && use_wd(e, rs.n_rows);                 // wide decode step on rows_gemm_wd_k
  const int ks_o = wd ? wd_ksplit(e, d, d) : e->p_o.ksplit, ks_f2 = wd ? wd_ksplit(e, d, 4 * d) : e->p_f2.ksplit;
  e->h_parts = ks_f2;
  for (int l = 0; l < e->L; ++l) {
    Layer& ly = e->layers[l];
    {
      GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
      g.h_in = (l == 0) ? rs.h_in : e->hB; g.h_out = e->hA; g.parts = e->parts;
      g.n_parts = (l == 0) ? 0 : ks_f2;
      g.prev_bias = (l == 0) ? ly.bo : e->layers[l - 1].b2; g.has_prev_bias = (l == 0) ? 0 : 1;
      g.x_out = e->xn;
      HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
      g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.x_in = e->xn; g.x_ld = d; g.mt = mtv;
      g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
      if (use_qkv16(e, rs.n_rows, mtv)) {     // every A lane a weight: the 16-channel image
        g.Wp = ly.Wqkv16; g.n_tiles = e->p_qkv16.n_tiles; g.KT = e->p_qkv16.KT; g.nchunk = e->p_qkv16.nchunk;
        if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_QKV16, 1, 1, s));
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV16, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV, 1, 1, s));
      }
    }
    {
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc; a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
      a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = rs.nsplit; a.scale = 1.0f / sqrtf((float)e->hd);
      a.row_seq = rs.row_seq; a.row_pos = rs.row_pos; a.n_rows = rs.n_rows; a.att_o = e->att_o; a.att_ml = e->att_ml;
      a.n_active = rs.n_active ? rs.n_active : e->one; a.dbg_ts = e->dbg_ts; a.share_len = e->share_len;
      a.nt = rs.n_active != nullptr ? attn_nt_for(e, rs.n_rows) : 0;      // wide decode passes stream their K/V once, prefill passes re-read it
      a.fast = e->attn_fast;
      if (rs.nsplit == 1) a.x_out = e->xn;    // xn is free between the QKV GEMM and the FFN LayerNorm
      if (rs.tiled == 2 && rs.nsplit == 1) HIPCHK(e, vc_launch_tile_attn64(a, s));
      else if (rs.tiled && rs.nsplit == 1) HIPCHK(e, vc_launch_tile_attn(a, e->dtype, s));
      else HIPCHK(e, vc_launch_attn(a, e->dtype, rs.n_rows, s));
    }
    {
      GemmArgs g = base_args(e, rs, e->p_o, d, d);
      g.Wp = ly.Wo; nt_bit(e, g, NT_O); g.part_out = e->parts; g.mt = mtv;
      if (rs.nsplit == 1) {
        g.x_in = e->xn; g.x_ld = d;
        if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_PART, ks_o, 1, s));
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_o.ksplit, 1, s));
      } else {
        g.att_o = e->att_o; g.att_ml = e->att_ml; g.nsplit = rs.nsplit;
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_ATT, EPI_PART, e->p_o.ksplit, 1, s));
      }
    }
    {
      GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
      g.h_in = e->hA; g.h_out = e->hB; g.parts = e->parts; g.n_parts = ks_o; g.prev_bias = ly.bo; g.has_prev_bias = 1;
      g.x_out = e->xn;
      HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
      g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.x_in = e->xn; g.x_ld = d; g.out = e->act; g.out_ld = 4 * d; g.mt = mtv;
      if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_RELU, 1, 1, s));
      else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_RELU, 1, 1, s));
    }
    {
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W2; nt_bit(e, g, NT_F2); g.x_in = e->act; g.x_ld = 4 * d; g.part_out = e->parts; g.mt = mtv;
      if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_PART, ks_f2, 1, s));
      else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_f2.ksplit, 1, s));
    }
  }
  return VC_OK;
}

// Prefill of one or several prompts as ONE row stream: the prompts' rows are laid back to back in `emb`
// (each row carries its own (sequence, position)), pushed through the decoder in passes of up to
// prefill_rows_per_pass rows - so the weights are streamed once per pass for all sequences together, not
// once per sequence - and the heads run on the last row of each prompt as soon as its pass is through.
// pas[i] describes prompt i (x, y, segments); slots[i] is its sequence slot.
int prefill_batch(vc_engine* e, std::vector<PromptArgs>& pas, const std::vector<int>& slots, hipStream_t s) {
  const int chunk = e->prefill_rows_per_pass;
  size_t i0 = 0;
  while (i0 < pas.size()) {
    // ---- a group of prompts that fits the row arena
    // (every prompt starts on a multiple of 16 rows, so that a 16-row tile never mixes sequences: the MFMA
    // attention of the prefill works on such tiles; the padding rows are inactive, row_pos = -1)
    size_t i1 = i0;
    int R = 0;
    // (tile_attn64_k works on 64-row blocks of one sequence: prompts then start on multiples of 64 rows)
    // ... and only when the call holds a LONG prompt: the kernel's gain grows with the sequence (800 rows -15 %, 2048 rows -39 %
    // of the attention launch; 512 rows +8 %), while the 64-row layout costs up to 48 padding rows per prompt in every GEMM of the
    // pass - config 5's eight 231-row prompts would pay 6.7 % more GEMM rows for no attention gain (measured neutral: prefill 5.2 ms
    // either way, profiles/r04p_default_attn64.log), so they stay on the first kernel and 16-row boundaries
    long longest = 0;
    for (const PromptArgs& pa : pas) longest = std::max<long>(longest, pa.Lx + pa.n_cols - pa.skip);
    // (... and only when the passes are cut on 64-row boundaries: with a `prefill_rows` option that is not a multiple of 64 a block of
    // a later pass would straddle one prompt's tail padding and the next prompt's first rows - ADVICE r04)
    const bool attn64 = e->tile_attn == 2 && e->dtype == VC_DTYPE_BF16 && e->hd == 128 && longest >= e->tile_attn_min_rows && chunk % 64 == 0 &&
                        !getenv("VC_NO_TILE_ATTN");
    const int al = attn64 ? 64 : 16;
    auto rows_of = [al](const PromptArgs& pa) { return ((pa.Lx + pa.n_cols - pa.skip) + al - 1) & ~(al - 1); };
    while (i1 < pas.size() && (i1 == i0 || R + rows_of(pas[i1]) <= e->emb_cap)) {
      R += rows_of(pas[i1]);
      ++i1;
    }
    if (R > e->emb_cap) return fail(e, VC_ECAP, "a prompt of %d rows does not fit the prefill arena of %d rows", R, e->emb_cap);
    HIPCHK(e, hipMemsetAsync(e->pre_row_pos, 0xFF, (size_t)R * sizeof(int), s));      // -1 everywhere
    HIPCHK(e, hipMemsetAsync(e->pre_row_seq, 0, (size_t)R * sizeof(int), s));
    HIPCHK(e, hipMemsetAsync(e->emb, 0, (size_t)R * e->d * sizeof(float), s));
    std::vector<int> last(i1 - i0);
    int row0 = 0;
    for (size_t i = i0; i < i1; ++i) {
      PromptArgs& pa = pas[i];
      const int rows = pa.Lx + pa.n_cols - pa.skip;
      pa.seq = slots[i]; pa.row0 = row0;
      pa.emb = e->emb; pa.row_seq = e->pre_row_seq; pa.row_pos = e->pre_row_pos; pa.err = e->err_flag;
      last[i - i0] = row0 + rows - 1;
      pa.logit_row = e->logit_row + slots[i];
      pa.logit_row_val = last[i - i0] % chunk;            // index of the prompt's last row inside its pass
      HIPCHK(e, vc_launch_prompt(pa, s));
      row0 += (rows + al - 1) & ~(al - 1);
    }
    for (int r0 = 0; r0 < R; r0 += chunk) {
      RowSrc rs{};
      rs.h_in = e->emb + (size_t)r0 * e->d;
      rs.row_seq = e->pre_row_seq + r0; rs.row_pos = e->pre_row_pos + r0;
      rs.n_rows = std::min(chunk, R - r0);
      int rc;
      if (rs.n_rows > VC_ROWS) { rs.nsplit = 1; rs.tiled = getenv("VC_NO_TILE_ATTN") ? 0 : (attn64 ? 2 : 1); rc = prefill_rows(e, rs, s); }
      else { rs.nsplit = attn_nsplit(e, rs.n_rows); rc = forward_rows(e, rs, s); }
      if (rc) return rc;
      for (size_t i = i0; i < i1; ++i)
        if (last[i - i0] >= r0 && last[i - i0] < r0 + rs.n_rows)
          if ((rc = run_heads(e, e->logit_row + slots[i], 1, slots[i], nullptr, s))) return rc;
    }
    i0 = i1;
  }
  return VC_OK;
}

void fill_prompt_common(vc_engine* e, PromptArgs& pa, const int64_t* x, int Lx, const int64_t* y, int T) {
  memset(&pa, 0, sizeof pa);
  pa.x = x; pa.y = y; pa.Lx = Lx; pa.T = T; pa.K = e->K; pa.d = e->d; pa.V = e->V;
  pa.empty_token = e->cfg.empty_token;
  pa.text_emb = e->text_emb; pa.audio_emb = e->audio_emb; pa.mask_emb = e->mask_emb; pa.pe = e->pe;
  pa.alpha_text = e->alpha_text; pa.alpha_audio = e->alpha_audio;
  pa.text_rows = e->cfg.text_rows;
}

// Engine-constant sampler arguments; the per-call values go through e->h_dyn -> e->d_dyn (push_sample_dyn).
SampleArgs make_sample_args(vc_engine* e, int B, int rps) {
  SampleArgs a;
  memset(&a, 0, sizeof a);
  a.logits = e->logits; a.B = B; a.K = e->K; a.V = e->V; a.d = e->d;
  a.empty_token = e->cfg.empty_token; a.gen_stride = e->gen_cap; a.dyn = e->d_dyn;
  a.st = e->st; a.n_active = e->n_active; a.host_active = e->h_flag + 8; a.host_live = e->h_flag + 9; a.step_ctr = e->step_ctr; a.graph_steps = std::max(1, e->steps_per_graph); a.samp = e->samp;
  a.gen = e->gen;
  a.rps = rps; a.dec_h = e->dec_h; a.row_seq = e->dec_row_seq; a.row_pos = e->dec_row_pos;
  a.logit_row = e->logit_row;
  a.audio_emb = e->audio_emb; a.mask_emb = e->mask_emb; a.pe = e->pe; a.alpha_audio = e->alpha_audio;
  a.max_positions = e->S_max;
  return a;
}

int push_sample_dyn(vc_engine* e, const vc_sample_cfg* sc, const int64_t* forced, int n_forced, float* logits_out,
                    int logit_steps, int n_seq, hipStream_t s) {
  SampleDyn& d = *e->h_dyn;
  memset(&d, 0, sizeof d);
  d.top_k = sc->top_k; d.top_p = sc->top_p; d.temperature = sc->temperature;
  d.stop_repetition = sc->stop_repetition;
  if (sc->n_silence < 0 || sc->n_silence > VC_MAX_SILENCE)
    return fail(e, VC_EINVAL, "n_silence %d outside [0,%d]", sc->n_silence, VC_MAX_SILENCE);
  d.n_silence = sc->n_silence;
  for (int i = 0; i < d.n_silence; ++i) d.silence[i] = sc->silence_tokens[i];
  d.seed = sc->seed;
  d.forced = forced; d.n_forced = forced ? n_forced : 0; d.forced_mode = sc->forced_mode;
  d.logits_out = logits_out; d.logit_steps = logits_out ? logit_steps : 0;
  d.max_steps = e->gen_cap;                               // rows of the gen buffer per sequence
  d.n_seq = n_seq;                                        // sequences of the call (row stride of forced / logits_out)
  d.dbg_ts = getenv("VC_SAMPLER_TS") ? e->dbg_ts : nullptr;
  HIPCHK(e, hipMemcpyAsync(e->d_dyn, e->h_dyn, sizeof(SampleDyn), hipMemcpyHostToDevice, s));
  return VC_OK;
}

int decode_step(vc_engine* e, const SampleArgs& sa, int B, int rps, bool grouped, hipStream_t s) {
  RowSrc rs{};
  rs.h_in = e->dec_h; rs.row_seq = e->dec_row_seq; rs.row_pos = e->dec_row_pos;
  rs.n_rows = B * rps; rs.nsplit = attn_nsplit(e, B * rps); rs.n_active = e->n_active;
  int rc;
  if (rs.n_rows > VC_ROWS) {             // more than one MFMA row tile: the step runs on the block GEMM (per-row LayerNorm launch,
    rs.nsplit = 1;                // attention one workgroup per (row, head), no split partials)
    rc = prefill_rows(e, rs, s);
  } else {
    rc = forward_rows(e, rs, s);
  }
  if (rc) return rc;
  // rps == 1: logit_row[b] == b (vc_tokens.hip advance_phase); the 3-row span switch is single-sequence
  rc = run_heads(e, rps == 1 ? nullptr : e->logit_row, B, 0, e->n_active, s);
  if (rc) return rc;
  HIPCHK(e, vc_launch_sample(sa, grouped, s));
  return VC_OK;
}

// The decode loop: every step is the same launch sequence (all step-dependent values live in HBM, the
// per-call sampler values behind a pointer), so `steps_per_graph` steps are captured ONCE per (sequences,
// rows per sequence, best-of-N) into a hipGraphExec that is kept for the life of the engine and replayed.
// The host never touches the stream inside the loop: the device clears a pinned host word when the last
// sequence retires, and the host paces itself one graph behind the GPU with events, so a queued graph is
// always waiting when the running one ends (measured before: ~8 us of idle GPU per graph launch and ~100 us
// per blocking poll).  Steps replayed after the last sequence retired are no-ops (*n_active == 0).
void refresh_opt_state(vc_engine* e);

int decode_loop(vc_engine* e, const SampleArgs& sa0, int B0, int rps, bool grouped, const vc_sample_cfg* sc,
                int max_steps, int* steps_run, hipStream_t s, bool precapture_only = false) {
  const int G = std::max(1, e->steps_per_graph);
  const double t0 = now_ms();
  if (precapture_only) e->host_ms[1] = e->host_ms[2] = 0;
  // Rows per step.  A multi-utterance call starts with one row per sequence; when few enough sequences are left (option "shrink") the
  // live ones are re-packed onto the rows of a narrower step - the next power of two >= the live count - and the loop goes on with
  // that width's captured graph (graphs are keyed by width).  B / sa = the width in force.
  int B = B0;
  SampleArgs sa = sa0;
  const bool can_shrink = e->shrink && !grouped && rps == 1 && B0 > 1;
  auto width_for = [](int live) { int p = 1; while (p < live) p *= 2; return p; };
  // One captured graph per (rows per step, option state), kept for the life of the engine.
  auto exec_for = [&](hipGraphExec_t* out) -> int {
    const auto key = std::make_pair(std::make_tuple(B, rps, grouped ? 1 : 0), e->opt_state);
    auto it = e->graphs.find(key);
    int rc = VC_OK;
    if (it != e->graphs.end()) {
      *out = it->second;
    } else {
      const double tc = now_ms();
      hipGraph_t graph = nullptr;
      hipError_t be = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      if (be != hipSuccess) rc = fail(e, VC_EHIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(be));
      for (int i = 0; i < G && rc == VC_OK; ++i) rc = decode_step(e, sa, B, rps, grouped, s);
      if (be == hipSuccess) {
        hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc == VC_OK && ce != hipSuccess) rc = fail(e, VC_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
      }
      e->host_ms[1] += now_ms() - tc;
      if (rc == VC_OK) {
        const double ti = now_ms();
        hipError_t ie = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) rc = fail(e, VC_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        else e->graphs[key] = *out;
        e->host_ms[2] += now_ms() - ti;
      }
      if (graph) (void)hipGraphDestroy(graph);
    }
    return rc;
  };
  if (precapture_only) {       // every graph this call can need, captured before the caller starts its decode timer
    if (!sc->use_graph) return VC_OK;
    hipGraphExec_t exec = nullptr;
    int rc0 = exec_for(&exec);
    if (can_shrink)            // ... and the narrower widths a shrinking batch passes through
      for (int w = width_for(B0) / 2; w >= 1 && rc0 == VC_OK; w /= 2) {
        if (w >= B0) continue;
        B = w; sa = make_sample_args(e, w, rps);
        rc0 = exec_for(&exec);
      }
    return rc0;
  }
  const double t1 = now_ms();
  HIPCHK(e, hipMemsetAsync(e->step_ctr, 0, sizeof(int), s));      // the loop's first step is step 0 of batch 0 (SampleArgs.host_live slots)
  volatile int* live = e->h_flag + 8;
  volatile int* live_n = e->h_flag + 9;                   // two slots: sequences still live as the LAST step of a batch of G steps found them
  int launched = 0, rc = VC_OK, batch = 0;
  while (launched < max_steps && rc == VC_OK) {
    if (batch >= 2) {   // pace: at most two batches in flight; the older one must have ended before a third is queued
      hipError_t we = hipEventSynchronize(e->ev_pace[batch & 1]);
      if (we != hipSuccess) { rc = fail(e, VC_EHIP, "pacing event: %s", hipGetErrorString(we)); break; }
      if (*live <= 0) break;
      if (can_shrink) {
        // the slot of the batch whose end has just been waited for (launched two iterations ago): the batch still in flight writes
        // the other one, so the width schedule is the same in every run of the same call - and with it, in bf16, every sampled token
        const int n_live = live_n[batch & 1];
        const int w = n_live >= 1 ? width_for(n_live) : B;
        if (w < B) {
          RepackArgs ra;
          memset(&ra, 0, sizeof ra);
          ra.st = e->st; ra.st_fin = e->st_fin; ra.dec_h = e->dec_h; ra.row_seq = e->dec_row_seq; ra.row_pos = e->dec_row_pos;
          ra.logit_row = e->logit_row; ra.err = e->err_flag; ra.B_old = B; ra.B_new = w; ra.d = e->d;
          hipError_t re = vc_launch_repack(ra, s);
          if (re != hipSuccess) { rc = fail(e, VC_EHIP, "re-pack launch: %s", hipGetErrorString(re)); break; }
          B = w; sa = make_sample_args(e, w, rps);
          e->host_ms[6] += 1;                             // re-packs of this call
        }
      }
    }
    if (sc->use_graph) {
      hipGraphExec_t exec = nullptr;
      rc = exec_for(&exec);
      if (rc) break;
      hipError_t le = hipGraphLaunch(exec, s);
      if (le != hipSuccess) rc = fail(e, VC_EHIP, "hipGraphLaunch failed: %s", hipGetErrorString(le));
    } else {
      for (int i = 0; i < G && rc == VC_OK; ++i) rc = decode_step(e, sa, B, rps, grouped, s);
    }
    if (rc) break;
    launched += G;
    hipError_t re = hipEventRecord(e->ev_pace[batch & 1], s);
    if (re != hipSuccess) { rc = fail(e, VC_EHIP, "hipEventRecord: %s", hipGetErrorString(re)); break; }
    ++batch;
  }
  e->host_ms[3] = now_ms() - t1;
  e->host_ms[4] = 0;
  e->host_ms[5] = launched;                               // decode steps LAUNCHED (a multiple of steps_per_graph; the tail are no-ops)
  e->cur_rows = B;
  if (steps_run) *steps_run = launched;
  return rc;
}

SeqState init_state(vc_engine* e, int Lx, int n_cols, bool tts, int n_spans) {
  SeqState st;
  memset(&st, 0, sizeof st);
  st.Lx = Lx; st.y_len = n_cols; st.prev_token = -1; st.n_spans = n_spans; st.kept = 1; st.group = -1;
  const vc_model_cfg& c = e->cfg;
  if (tts) {
    st.cap_len = Lx * (c.encodec_sr / 5);                 // voicecraft.py:1042
    st.min_gen = c.encodec_sr / 5;                        // :1024
    st.term_token = c.eos > 0 ? c.eos : c.eog;            // :938
    st.kill_token = c.eos > 0 ? c.eog : -1;               // :1091-1093
  } else {
    st.cap_len = Lx * 10;                                 // :751
    st.min_gen = -1;
    st.term_token = c.eog;
    st.kill_token = c.eos > 0 ? c.eos : -1;               // :816-818
  }
  return st;
}

int check_ready(vc_engine* e) {
  if (!e) return VC_EINVAL;
  if (!e->finalized) return fail(e, VC_ESTATE, "weights are not finalized (call vc_finalize_weights)");
  hipError_t err = hipSetDevice(e->device);
  if (err != hipSuccess) return fail(e, VC_EHIP, "hipSetDevice: %s", hipGetErrorString(err));
  return VC_OK;
}

int check_err_flag(vc_engine* e, hipStream_t s) {
  HIPCHK(e, hipMemcpyAsync(e->h_flag, e->err_flag, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  if (const int bits = *e->h_flag) {
    (void)hipMemsetAsync(e->err_flag, 0, sizeof(int), s);   // already on the error path
    if (bits & 2)
      return fail(e, VC_EINVAL, "shared_text_prefix: a sequence's text differs from sequence 0's inside the shared prefix");
    if (bits & 4)
      return fail(e, VC_ESTATE, "internal: a re-pack found more live sequences than the narrower step has rows");
    return fail(e, VC_EINVAL, "token id out of range in x or y (text rows %d, audio vocab %d)", e->cfg.text_rows, e->V);
  }
  return VC_OK;
}

// One option by name (vc_set_option, and the VC_* environment variables at creation).
int apply_option(vc_engine* e, const std::string& name, const char* value) {
  int v0 = 0, v1 = 0, v2 = 0;
  const int n = sscanf(value ? value : "", "%d,%d,%d", &v0, &v1, &v2);
  if (n < 1) return fail(e, VC_EINVAL, "option '%s': '%s' is not a number list", name.c_str(), value ? value : "(null)");
  if (name == "graph_steps") { e->steps_per_graph = std::max(1, std::min(64, v0));
  } else if (name == "tile_attn") {
    e->tile_attn = v0 == 2 ? 2 : 1;
    if (n >= 2) e->tile_attn_min_rows = std::max(64, v1);
  } else if (name == "finished_rows" || name == "fr_one") {
    if (v0 > 0 && !e->layers.empty() && !e->layers[0].W28)
      return fail(e, VC_ESTATE, "option '%s': this engine was created with VC_FINISHED_ROWS=0 and VC_FR_ONE=0 and holds no 8-channel weight images", name.c_str());
    if (name == "fr_one") e->fr_one = std::max(0, std::min(v0, 2));
    else e->fr_rows = std::max(0, std::min(v0, VC_ROWS));
  } else if (name == "fr_pair") { e->fr_pair = v0 ? 1 : 0;
  } else if (name == "qkv_p8") {
    if (v0 && !e->layers.empty() && !e->layers[0].Wqkv8 && vc_gemm_fr1_ok(3 * e->d, e->d, e->dtype, 4))
      return fail(e, VC_ESTATE, "option 'qkv_p8': this engine was created with VC_QKV_P8=0 or VC_FR_ONE=0 and holds no 8-channel image of the QKV matrix");
    e->qkv_p8 = std::max(0, std::min(v0, 2));      // 2: steps of 2..8 finished rows as well (rows_gemm_qp_k)
  } else if (name == "wide_heads") { e->wide_heads = v0 ? 1 : 0;
  } else if (name == "wide_gemm") { e->wide_gemm = v0 ? 1 : 0;
  } else if (name == "shrink") { e->shrink = v0 ? 1 : 0;
  } else if (name == "wd_stage") { e->wd_stage = v0 ? 1 : 0;
  } else if (name == "qkv16") {
    if (v0 && !e->layers.empty() && !e->layers[0].Wqkv16)
      return fail(e, VC_ESTATE, "option 'qkv16': this engine holds no 16-channel image of the QKV matrix (packed for max_seqs > 16, or with VC_QKV16=1 at creation)");
    e->qkv16 = v0 ? 1 : 0;
  } else if (name == "attn_fast") { e->attn_fast = v0 ? 1 : 0;
  } else if (name == "att_p16") { e->att_p16 = v0 ? 1 : 0;
  } else if (name == "hq") { e->hq = v0 ? 1 : 0;
  } else if (name == "nt") {      // "mask[,kv]": the weight matrices streamed with the hint, and the decode attention's K/V loads (0 never, 1 always, 2 from two rows up)
    e->nt_decode = v0 & 63;
    if (n >= 2) e->attn_nt = std::max(0, std::min(v1, 2));
  } else if (name == "prefill_rows") { e->prefill_rows_per_pass = std::max(VC_ROWS, std::min(VC_MAX_ROWS, v0 / VC_ROWS * VC_ROWS));   // 16: decode kernels only
  } else {
    return fail(e, VC_EINVAL, "unknown option '%s'", name.c_str());
  }
  return VC_OK;
}

void refresh_opt_state(vc_engine* e) {
  char buf[256];
  snprintf(buf, sizeof buf, "g=%d|nt=%d,%d|fr=%d,%d,%d,%d|ta=%d,%d|r1=%d,%d,%d|q16=%d,%d,%d,%d|sh=%d",
           e->steps_per_graph, e->nt_decode, e->attn_nt, e->fr_rows, e->fr_pair, e->att_p16, e->hq, e->tile_attn, e->tile_attn_min_rows,
           e->fr_one, e->attn_fast, e->qkv_p8, e->qkv16, e->wide_heads, e->wide_gemm, e->wd_stage, e->shrink);
  e->opt_state = buf;
}

}  // namespace

// =====================================================================================  C ABI
// Host-only: how a decode pass of `rows` rows would be launched for this model shape and compute dtype (no engine, no GPU).
extern "C" int vc_debug_plan(const vc_model_cfg* c, int compute_dtype, int rows, int32_t out[16]) {
  if (!c || !out || rows < 1 || rows > VC_MAX_SEQS || c->d_model <= 0 || c->nhead <= 0 || c->d_model % c->nhead) return VC_EINVAL;
  if (compute_dtype != VC_DTYPE_BF16 && compute_dtype != VC_DTYPE_F32) return VC_EINVAL;
  vc_engine e;
  e.cfg = *c; e.d = c->d_model; e.H = c->nhead; e.hd = c->d_model / c->nhead; e.dtype = compute_dtype;
  const int d = e.d, frmax = fr_max_rows(&e);
  const bool fr = rows >= 2 && rows <= VC_ROWS && rows <= frmax;
  const int ns = fr ? fr_nsplit(&e, rows) : attn_nsplit(&e, rows);
  out[0] = frmax;
  out[1] = rows > VC_ROWS ? 2 : fr ? 1 : 0;                                   // 0 slab form (rows-GEMM), 1 finished-row form, 2 wide decode
  out[2] = rows > VC_ROWS ? 1 : ns;                                           // attention splits
  out[3] = fr ? (rows > VC_FR_MAX_ROWS ? 4 : lnw_two(&e, rows) ? 3 : 0) : -1; // consumers: GemmArgs.mt (0 one tile, 3 two tiles, 4 two tiles x two rows per wave)
  out[4] = fr ? vc_gemm_fr_form(rows, d, d, compute_dtype, ns == 1 ? PRO_PLAIN : PRO_ATT, ns) : -1;    // out-projection producer form
  out[5] = fr ? vc_gemm_fr_form(rows, d, 4 * d, compute_dtype, PRO_PLAIN, 1) : -1;                      // FFN-down producer form
  out[6] = fr && rows <= VC_FR_MAX_ROWS ? 1 : 0;                              // heads-1 folds finished rows itself (else LayerNorm launch)
  out[7] = (3 * d / VC_TH_QKV) % 2 == 0 && (4 * d / 16) % 2 == 0 ? 1 : 0;     // the consumers' tile counts are even (two tiles per workgroup)
  // the default forms of rounds 5-6, from the launchers' own predicates
  for (int i = 8; i < 16; ++i) out[i] = -1;
  if (rows == 1) {
    out[8] = vc_gemm_fr1_ok(d, 4 * d, compute_dtype, VC_FR_WAVES) ? 1 : 0;                       // fr_one: FFN-down finishes its row
    out[9] = out[8] && vc_gemm_fr1_ok(3 * d, d, compute_dtype, 4) ? 1 : 0;                       // qkv_p8: the paired QKV projection behind it
  }
  if (fr && rows <= VC_FR_MAX_ROWS) out[10] = vc_gemm_frp_ok(rows, d, 4 * d, compute_dtype) ? 1 : 0;   // fr_pair
  if (fr && rows <= VC_FR_MAX_ROWS)       // qkv_p8 = 2: layers 1.. run the QKV projection on the 8-channel image (rows_gemm_qp_k) - the image's packing rule, qp_on's row rule
    out[9] = (hq_on(&e) && vc_gemm_fr1_ok(3 * d, d, compute_dtype, 4) && rows <= (d >= 2048 ? 6 : 8) && vc_gemm_qp_ok(rows, 3 * d, d, compute_dtype)) ? 2 : 0;
  if (rows > VC_ROWS) {
    e.P = c->head_hidden;
    const int ko = wd_ksplit(&e, d, d), kf = wd_ksplit(&e, d, 4 * d);
    out[12] = ko; out[13] = kf;
    out[14] = vc_gemm_wd_kpw(d, compute_dtype, 1); out[15] = vc_gemm_wd_kpw(c->head_hidden, compute_dtype, 1);
    out[11] = (ko && kf && out[14] && out[15]) ? 1 : 0;
  }
  return VC_OK;
}

extern "C" int vc_set_option(vc_engine* e, const char* name, const char* value) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (!name || !value) return fail(e, VC_EINVAL, "null argument to vc_set_option");
  rc = apply_option(e, name, value);
  refresh_opt_state(e);
  return rc;
}

extern "C" const char* vc_version(void) { return "vcengine 0.1 (gfx950)"; }

extern "C" const char* vc_last_error(const vc_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

extern "C" int vc_create(const vc_model_cfg* c, int hip_device, vc_engine** out) {
  if (!c || !out) return fail(nullptr, VC_EINVAL, "null argument");
  *out = nullptr;
  if (c->d_model <= 0 || c->d_model % 256 || c->d_model > 2048)
    return fail(nullptr, VC_EINVAL, "d_model %d must be a multiple of 256 and <= 2048", c->d_model);
  if (c->nhead <= 0 || c->d_model % c->nhead) return fail(nullptr, VC_EINVAL, "nhead %d does not divide d_model", c->nhead);
  const int hd = c->d_model / c->nhead;
  if (hd != 32 && hd != 64 && hd != 128)   // the attention kernel spreads a cached row over 4, 8, 16 or 32 lanes
    return fail(nullptr, VC_EINVAL, "head_dim %d must be 32, 64 or 128", hd);
  if ((hd & (hd - 1)) != 0) return fail(nullptr, VC_EINVAL, "head_dim %d must be a power of two", hd);
  if (c->n_codebooks < 1 || c->n_codebooks > VC_MAX_CODEBOOKS) return fail(nullptr, VC_EINVAL, "n_codebooks %d unsupported", c->n_codebooks);
  const int V = c->audio_vocab_size + c->n_special;
  if (V > 64 * VC_VPL) return fail(nullptr, VC_EINVAL, "audio vocabulary %d too large (max %d)", V, 64 * VC_VPL);
  if (c->head_hidden % 256) return fail(nullptr, VC_EINVAL, "head_hidden %d must be a multiple of 256", c->head_hidden);
  if (c->empty_token != c->audio_vocab_size || c->eog != c->audio_vocab_size + 1 ||
      c->audio_pad_token != c->audio_vocab_size + 2)   // voicecraft.py:132-134
    return fail(nullptr, VC_EINVAL, "special tokens must be empty=V, eog=V+1, pad=V+2");
  if (c->eos >= V || c->eog >= V) return fail(nullptr, VC_EINVAL, "eos/eog outside the vocabulary");
  if (c->max_seqs < 1 || c->max_seqs > VC_MAX_SEQS) return fail(nullptr, VC_EINVAL, "max_seqs must be in [1,%d]", VC_MAX_SEQS);
  if (c->max_positions < 32) return fail(nullptr, VC_EINVAL, "max_positions too small");
  if (c->max_n_spans < 1 || c->max_n_spans > VC_MAX_SPANS) return fail(nullptr, VC_EINVAL, "max_n_spans unsupported");
  hipError_t err = hipSetDevice(hip_device);
  if (err != hipSuccess) return fail(nullptr, VC_EHIP, "hipSetDevice(%d): %s", hip_device, hipGetErrorString(err));
  vc_engine* e = new vc_engine();
  e->cfg = *c; e->device = hip_device;
  e->d = c->d_model; e->H = c->nhead; e->hd = hd; e->L = c->num_layers; e->K = c->n_codebooks;
  e->V = V; e->P = c->head_hidden; e->S_max = c->max_positions; e->B_max = c->max_seqs;
  e->NS = std::max(VC_ROWS, c->max_seqs);
  *out = e;
  return VC_OK;
}

extern "C" void vc_destroy(vc_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (auto& kv : e->raw) if (kv.second.dev) (void)hipFree(kv.second.dev);
  for (void* p : e->allocs) (void)hipFree(p);
  if (e->h_st) (void)hipHostFree(e->h_st);
  if (e->h_st2) (void)hipHostFree(e->h_st2);
  if (e->h_flag) (void)hipHostFree(e->h_flag);
  if (e->h_dyn) (void)hipHostFree(e->h_dyn);
  for (auto& kv : e->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  for (auto& ev : e->ev_pace) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

extern "C" int vc_load_tensor(vc_engine* e, const char* key, const void* data, int on_device, int dtype,
                              const int64_t* shape, int ndim) {
  if (!e || !key || !data || ndim < 0 || ndim > 4) return fail(e, VC_EINVAL, "bad argument to vc_load_tensor");
  if (e->finalized) return fail(e, VC_ESTATE, "weights already finalized");
  if (dtype != VC_DTYPE_F32) return VC_OK;   // eog / eos buffers (int64) carry no information we need
  HIPCHK(e, hipSetDevice(e->device));
  RawTensor t;
  t.numel = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); t.numel *= shape[i]; }
  if (t.numel <= 0) return fail(e, VC_EINVAL, "empty tensor '%s'", key);
  auto it = e->raw.find(key);
  if (it != e->raw.end()) { (void)hipFree(it->second.dev); e->raw.erase(it); }
  HIPCHK(e, hipMalloc((void**)&t.dev, (size_t)t.numel * 4));
  HIPCHK(e, hipMemcpy(t.dev, data, (size_t)t.numel * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  e->raw[key] = t;
  return VC_OK;
}

extern "C" int vc_finalize_weights(vc_engine* e, int compute_dtype) {
  if (!e) return VC_EINVAL;
  if (e->finalized) return fail(e, VC_ESTATE, "already finalized");
  if (compute_dtype != VC_DTYPE_BF16 && compute_dtype != VC_DTYPE_F32) return fail(e, VC_EINVAL, "compute dtype must be bf16 or f32");
  HIPCHK(e, hipSetDevice(e->device));
  e->dtype = compute_dtype;
  e->esz = compute_dtype == VC_DTYPE_BF16 ? 2 : 4;
  const int d = e->d, L = e->L, K = e->K, V = e->V, P = e->P;
  int rc;
  const RawTensor* t;
  // ---- embeddings (kept fp32: a handful of row gathers per step)
  if ((rc = need(e, "text_embedding.word_embeddings.weight", {e->cfg.text_rows, d}, &t))) return rc;
  if ((rc = dalloc(e, &e->text_emb, (size_t)t->numel))) return rc;
  HIPCHK(e, hipMemcpy(e->text_emb, t->dev, (size_t)t->numel * 4, hipMemcpyDeviceToDevice));
  if ((rc = dalloc(e, &e->audio_emb, (size_t)K * V * d))) return rc;
  for (int k = 0; k < K; ++k) {
    if ((rc = need(e, "audio_embedding." + std::to_string(k) + ".word_embeddings.weight", {V, d}, &t))) return rc;
    HIPCHK(e, hipMemcpy(e->audio_emb + (size_t)k * V * d, t->dev, (size_t)V * d * 4, hipMemcpyDeviceToDevice));
  }
  if ((rc = need(e, "mask_embedding", {e->cfg.max_n_spans, d}, &t))) return rc;
  if ((rc = dalloc(e, &e->mask_emb, (size_t)t->numel))) return rc;
  HIPCHK(e, hipMemcpy(e->mask_emb, t->dev, (size_t)t->numel * 4, hipMemcpyDeviceToDevice));
  if ((rc = need(e, "text_positional_embedding.alpha", {1}, &t))) return rc;
  HIPCHK(e, hipMemcpy(&e->alpha_text, t->dev, 4, hipMemcpyDeviceToHost));
  if ((rc = need(e, "audio_positional_embedding.alpha", {1}, &t))) return rc;
  HIPCHK(e, hipMemcpy(&e->alpha_audio, t->dev, 4, hipMemcpyDeviceToHost));
  // ---- sinusoidal table (embedding.py:69-92).  "pe" may be supplied by the loader (torch's own
  // sin/cos); otherwise it is computed here in fp32 with the same formula.
  if ((rc = dalloc(e, &e->pe, (size_t)e->S_max * d))) return rc;
  if (const RawTensor* pt = find_raw(e, "pe")) {
    if (pt->shape.size() != 2 || pt->shape[0] < e->S_max || pt->shape[1] != d)
      return fail(e, VC_EINVAL, "'pe' must be [>=%d][%d]", e->S_max, d);
    HIPCHK(e, hipMemcpy(e->pe, pt->dev, (size_t)e->S_max * d * 4, hipMemcpyDeviceToDevice));
  } else {
    std::vector<float> pe((size_t)e->S_max * d);
    for (int i = 0; i < d; i += 2) {
      const float div = expf((float)i * -(logf(10000.0f) / (float)d));
      for (int p = 0; p < e->S_max; ++p) {
        const float a = (float)p * div;
        pe[(size_t)p * d + i] = sinf(a);
        pe[(size_t)p * d + i + 1] = cosf(a);
      }
    }
    HIPCHK(e, hipMemcpy(e->pe, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
  }
  // ---- decoder layers
  const char* env_fr = getenv("VC_FINISHED_ROWS");
  const char* env_f1 = getenv("VC_FR_ONE");
  const bool want_fr8 = !(env_fr && atoi(env_fr) == 0 && env_f1 && atoi(env_f1) == 0);
  // the 16-channel image of the QKV matrix serves wide decode steps (17..64 rows) and prefill passes: packed for engines that can
  // take such steps (max_seqs > 16) or on request (VC_QKV16=1); a narrower engine runs its prefill on the 12-channel tiles (-0.4 GB)
  const char* env_q16 = getenv("VC_QKV16");
  const bool want_qkv16 = env_q16 ? atoi(env_q16) != 0 : e->B_max > VC_ROWS;
  if (!want_qkv16) e->qkv16 = 0;
  e->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    const std::string pre = "decoder.layers." + std::to_string(l) + ".";
    Layer& ly = e->layers[l];
    if ((rc = pack_folded(e, pre + "self_attn.in_proj_weight", pre + "self_attn.in_proj_bias", pre + "norm1.", 3 * d, d,
                          &ly.Wqkv, &ly.wg_qkv, &ly.bqkv, VC_TH_QKV))) return rc;
    if (want_qkv16) {
      const RawTensor* tg1;
      if ((rc = need(e, pre + "norm1.weight", {d}, &tg1))) return rc;
      if ((rc = pack_matrix(e, pre + "self_attn.in_proj_weight", 3 * d, d, &ly.Wqkv16, 16, tg1->dev))) return rc;
    }
    if ((rc = pack_matrix(e, pre + "self_attn.out_proj.weight", d, d, &ly.Wo))) return rc;
    if ((rc = keep_vec(e, pre + "self_attn.out_proj.bias", d, &ly.bo))) return rc;
    if ((rc = pack_folded(e, pre + "linear1.weight", pre + "linear1.bias", pre + "norm2.", 4 * d, d,
                          &ly.W1, &ly.wg_1, &ly.b1))) return rc;
    if ((rc = pack_matrix(e, pre + "linear2.weight", d, 4 * d, &ly.W2))) return rc;
    // the second, 8-channel-tile image of the two row-producing matrices (finished-row forms): + 10 d^2 elements per layer = + 0.67 GB
    // at giga830M in bf16.  An engine created with BOTH forms preset off (VC_FINISHED_ROWS=0 and VC_FR_ONE=0) does not pack them,
    // and refuses to switch the forms on later (apply_option).
    if (want_fr8) {
      if ((rc = pack_matrix(e, pre + "self_attn.out_proj.weight", d, d, &ly.Wo8, VC_TH_RES))) return rc;
      if ((rc = pack_matrix(e, pre + "linear2.weight", d, 4 * d, &ly.W28, VC_TH_RES))) return rc;
      // (the one-row paired QKV kernel reads the folded matrix in 8-channel tiles: only where its form can run, i.e. behind fr_one)
      // (not packed when either of the two options it serves is preset off - VC_QKV_P8=0 / VC_FR_ONE=0 -: 0.4 GB at giga830M)
      const char* env_p8 = getenv("VC_QKV_P8");
      const bool want_p8 = !(env_p8 && atoi(env_p8) == 0) && !(env_f1 && atoi(env_f1) == 0);
      if (want_p8 && vc_gemm_fr1_ok(3 * d, d, e->dtype, 4)) {
        const RawTensor* tg1;
        if ((rc = need(e, pre + "norm1.weight", {d}, &tg1))) return rc;
        if ((rc = pack_matrix(e, pre + "self_attn.in_proj_weight", 3 * d, d, &ly.Wqkv8, VC_TH_RES, tg1->dev))) return rc;
      }
    }
    if ((rc = keep_vec(e, pre + "linear2.bias", d, &ly.b2))) return rc;
    const size_t cache_bytes = (size_t)e->B_max * e->H * e->S_max * e->hd * e->esz;
    char* kc; char* vc;
    if ((rc = dalloc(e, &kc, cache_bytes))) return rc;
    if ((rc = dalloc(e, &vc, cache_bytes))) return rc;
    ly.kc = kc; ly.vc = vc;
    // free the staging copies of this layer right away (3.3 GB for the 830M shape otherwise)
    for (const char* k2 : {"self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"}) {
      auto it = e->raw.find(pre + k2);
      if (it != e->raw.end()) { HIPCHK(e, hipDeviceSynchronize()); (void)hipFree(it->second.dev); e->raw.erase(it); }
    }
  }
  // ---- heads: first linears concatenated to one [K*P][d] matrix, second ones one group each
  {
    // (folded with the final LayerNorm decoder.norm, transformer.py:484-485)
    float *cat, *bcat;
    const RawTensor *tg, *tbe;
    if ((rc = need(e, "decoder.norm.weight", {d}, &tg))) return rc;
    if ((rc = need(e, "decoder.norm.bias", {d}, &tbe))) return rc;
    HIPCHK(e, hipMalloc((void**)&cat, (size_t)K * P * d * 4));
    if (hipMalloc((void**)&bcat, (size_t)K * P * 4) != hipSuccess) { (void)hipFree(cat); return fail(e, VC_EHIP, "hipMalloc failed"); }
    auto drop = [&]() { (void)hipDeviceSynchronize(); (void)hipFree(cat); (void)hipFree(bcat); };
    if ((rc = dalloc(e, &e->bh1, (size_t)K * P))) { drop(); return rc; }
    if ((rc = dalloc(e, &e->wg_h1, (size_t)K * P))) { drop(); return rc; }
    for (int k = 0; k < K; ++k) {
      const std::string pre = "predict_layer." + std::to_string(k) + ".";
      if ((rc = need(e, pre + "0.weight", {P, d}, &t))) { drop(); return rc; }
      if (hipMemcpy(cat + (size_t)k * P * d, t->dev, (size_t)P * d * 4, hipMemcpyDeviceToDevice) != hipSuccess) { drop(); return fail(e, VC_EHIP, "hipMemcpy of %s0.weight failed", pre.c_str()); }
      if ((rc = need(e, pre + "0.bias", {P}, &t))) { drop(); return rc; }
      if (hipMemcpy(bcat + (size_t)k * P, t->dev, (size_t)P * 4, hipMemcpyDeviceToDevice) != hipSuccess) { drop(); return fail(e, VC_EHIP, "hipMemcpy of %s0.bias failed", pre.c_str()); }
    }
    const int KW = e->dtype == VC_DTYPE_BF16 ? 32 : 16;
    const long units = (long)(K * P / 16) * (d / KW) * 64;
    if ((rc = dalloc(e, &e->Wh1, (size_t)units))) { drop(); return rc; }
    hipError_t pe_ = vc_launch_pack(cat, tg->dev, e->Wh1, K * P, d, e->dtype, 16, 0);
    if (pe_ == hipSuccess) pe_ = vc_launch_fold_vecs(cat, tg->dev, tbe->dev, bcat, e->wg_h1, e->bh1, K * P, d, e->dtype, 0);
    drop();
    HIPCHK(e, pe_);
    const long gunits = (long)((V + 15) / 16) * (P / KW) * 64;
    e->wh2_group_stride = gunits;
    if ((rc = dalloc(e, &e->Wh2, (size_t)gunits * K))) return rc;
    if ((rc = dalloc(e, &e->bh2, (size_t)K * V))) return rc;
    for (int k = 0; k < K; ++k) {
      const std::string pre = "predict_layer." + std::to_string(k) + ".";
      if ((rc = need(e, pre + "2.weight", {V, P}, &t))) return rc;
      HIPCHK(e, vc_launch_pack(t->dev, nullptr, e->Wh2 + (size_t)k * gunits, V, P, e->dtype, 16, 0));
      if ((rc = need(e, pre + "2.bias", {V}, &t))) return rc;
      HIPCHK(e, hipMemcpy(e->bh2 + (size_t)k * V, t->dev, (size_t)V * 4, hipMemcpyDeviceToDevice));
    }
  }
  HIPCHK(e, hipDeviceSynchronize());
  for (auto& kv : e->raw) if (kv.second.dev) (void)hipFree(kv.second.dev);
  e->raw.clear();
  // ---- launch plans
  e->p_qkv = make_plan(3 * d, d, e->dtype, false, nullptr, VC_TH_QKV);
  e->p_qkv16 = make_plan(3 * d, d, e->dtype, false, nullptr, 16);
  e->p_o = make_plan(d, d, e->dtype, true, "VC_KSPLIT_O");
  e->p_f1 = make_plan(4 * d, d, e->dtype, false, nullptr);
  e->p_f2 = make_plan(d, 4 * d, e->dtype, true, "VC_KSPLIT_F");
  e->p_h1 = make_plan(K * P, d, e->dtype, false, nullptr);
  e->p_h2 = make_plan(V, P, e->dtype, false, nullptr);
  // ---- scratch arenas
  e->emb_cap = std::max(e->S_max, 8 * VC_MAX_ROWS);
  if ((rc = dalloc(e, &e->emb, (size_t)e->emb_cap * d))) return rc;
  if ((rc = dalloc(e, &e->pre_row_seq, (size_t)e->emb_cap + VC_MAX_ROWS))) return rc;
  if ((rc = dalloc(e, &e->pre_row_pos, (size_t)e->emb_cap + VC_MAX_ROWS))) return rc;
  if ((rc = dalloc(e, &e->hA, (size_t)VC_MAX_ROWS * d))) return rc;
  if ((rc = dalloc(e, &e->hB, (size_t)VC_MAX_ROWS * d))) return rc;
  if ((rc = dalloc(e, &e->q, (size_t)VC_MAX_ROWS * d))) return rc;
  if ((rc = dalloc(e, &e->parts, (size_t)VC_MAX_KSPLIT * VC_SLAB_ROWS * d))) return rc;
  if ((rc = dalloc(e, &e->att_o, (size_t)VC_ROWS * e->H * VC_MAX_NSPLIT * e->hd))) return rc;     // split partials: decode-size passes only
  if ((rc = dalloc(e, &e->att_ml, (size_t)VC_ROWS * e->H * VC_MAX_NSPLIT * 2))) return rc;
  char* tmp;
  if ((rc = dalloc(e, &tmp, (size_t)VC_MAX_ROWS * 4 * d * e->esz))) return rc;
  e->act = tmp;
  if ((rc = dalloc(e, &tmp, (size_t)VC_MAX_ROWS * d * e->esz))) return rc;
  e->xn = tmp;
  if ((rc = dalloc(e, &tmp, (size_t)e->NS * K * P * e->esz))) return rc;
  e->hh = tmp;
  if ((rc = dalloc(e, &e->logits, (size_t)e->NS * K * V))) return rc;
  if ((rc = dalloc(e, &e->dec_h, (size_t)e->NS * d))) return rc;
  if ((rc = dalloc(e, &e->dec_row_seq, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->dec_row_pos, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->logit_row, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->st, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->st_fin, (size_t)e->NS))) return rc;
  {
    char* t2;
    if ((rc = dalloc(e, &t2, (size_t)VC_ROWS * d * e->esz))) return rc;
    e->hqA = t2;
    if ((rc = dalloc(e, &t2, (size_t)VC_ROWS * d * e->esz))) return rc;
    e->hqB = t2;
    if ((rc = dalloc(e, &e->row_mu[0], (size_t)VC_ROWS))) return rc;
    if ((rc = dalloc(e, &e->row_mu[1], (size_t)VC_ROWS))) return rc;
    if ((rc = dalloc(e, &e->mu_zero, (size_t)VC_ROWS))) return rc;
    HIPCHK(e, hipMemset(e->row_mu[0], 0, VC_ROWS * 4));
    HIPCHK(e, hipMemset(e->row_mu[1], 0, VC_ROWS * 4));
    HIPCHK(e, hipMemset(e->mu_zero, 0, VC_ROWS * 4));
  }
  if ((rc = dalloc(e, &e->n_active, (size_t)4))) return rc;
  if ((rc = dalloc(e, &e->step_ctr, (size_t)4))) return rc;
  HIPCHK(e, hipMemset(e->step_ctr, 0, 4));
  if ((rc = dalloc(e, &e->err_flag, (size_t)4))) return rc;
  if ((rc = dalloc(e, &e->one, (size_t)4))) return rc;
  if ((rc = dalloc(e, &e->share_len, (size_t)4))) return rc;
  HIPCHK(e, hipMemset(e->share_len, 0, 16));
  if ((rc = dalloc(e, &e->samp, (size_t)e->NS * (VC_MAX_CODEBOOKS + 2)))) return rc;
  if ((rc = dalloc(e, &e->cond, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->amax, (size_t)e->NS))) return rc;
  if ((rc = dalloc(e, &e->dbg_ts, (size_t)64))) return rc;
  HIPCHK(e, hipMemset(e->dbg_ts, 0, 64 * 8));
  e->gen_cap = e->S_max;
  if ((rc = dalloc(e, &e->gen, (size_t)e->B_max * e->gen_cap * K))) return rc;
  HIPCHK(e, hipMemset(e->err_flag, 0, 16));
  HIPCHK(e, hipMemset(e->n_active, 0, 16));
  { const int one[4] = {1, 1, 1, 1}; HIPCHK(e, hipMemcpy(e->one, one, 16, hipMemcpyHostToDevice)); }
  HIPCHK(e, hipHostMalloc((void**)&e->h_st, sizeof(SeqState) * e->NS));
  HIPCHK(e, hipHostMalloc((void**)&e->h_st2, sizeof(SeqState) * e->NS));
  HIPCHK(e, hipHostMalloc((void**)&e->h_flag, 64));
  memset(e->h_flag, 0, 64);
  HIPCHK(e, hipHostMalloc((void**)&e->h_dyn, sizeof(SampleDyn)));
  if ((rc = dalloc(e, &e->d_dyn, (size_t)1))) return rc;
  for (auto& ev : e->ev_pace) HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (auto& ev : e->ev) HIPCHK(e, hipEventCreate(&ev));
  // a blocking stream: implicitly ordered after work the caller queued on the null stream
  HIPCHK(e, hipStreamCreate(&e->own_stream));
  // ---- options: the VC_* environment variables preset them, vc_set_option changes them at run time
  for (const auto& kv : {std::make_pair("VC_NT", "nt"), std::make_pair("VC_PREFILL_ROWS", "prefill_rows"), 
                         std::make_pair("VC_GRAPH_STEPS", "graph_steps"),
                         
                         std::make_pair("VC_FINISHED_ROWS", "finished_rows"),
                         std::make_pair("VC_TILE_ATTN", "tile_attn"),
                         std::make_pair("VC_FR_ONE", "fr_one"), std::make_pair("VC_QKV_P8", "qkv_p8"), std::make_pair("VC_FR_PAIR", "fr_pair"), std::make_pair("VC_QKV16", "qkv16"), std::make_pair("VC_WIDE_HEADS", "wide_heads"), std::make_pair("VC_WIDE_GEMM", "wide_gemm"), std::make_pair("VC_WD_STAGE", "wd_stage"), std::make_pair("VC_SHRINK", "shrink"), std::make_pair("VC_ATTN_FAST", "attn_fast"), std::make_pair("VC_ATT_P16", "att_p16"), std::make_pair("VC_HQ", "hq")})
    if (const char* v = getenv(kv.first)) {
      // (VC_NT was a boolean through round 3 - 1 = on, the default; it is a per-matrix bit mask now: the legacy "1" keeps meaning "on")
      if (std::string(kv.first) == "VC_NT" && std::string(v) == "1") v = "63";
      if ((rc = apply_option(e, kv.second, v))) return rc;
    }
  refresh_opt_state(e);
  HIPCHK(e, hipDeviceSynchronize());
  e->finalized = true;
  return VC_OK;
}

// ------------------------------------------------------------------------------------- TTS
namespace {

struct TtsJob { const int64_t* x; int Lx; const int64_t* y; int T; };

// Shared by vc_tts (one prompt, n_samples >= 1) and vc_tts_multi (B prompts).
int tts_run(vc_engine* e, const std::vector<TtsJob>& jobs, int n_samples, const vc_sample_cfg* sc,
            const int64_t* forced, int n_forced, float* logits_out, int logit_steps, int* steps_out,
            hipStream_t s, int shared_prefix = 0) {
  const int K = e->K;
  const bool grouped = n_samples > 1;
  const int B = grouped ? n_samples : (int)jobs.size();
  if (B > e->B_max) return fail(e, VC_ECAP, "%d sequences requested but max_seqs is %d", B, e->B_max);
  int max_steps = 0;
  for (int b = 0; b < (int)jobs.size(); ++b) {
    const TtsJob& j = jobs[b];
    if (j.Lx < 1 || j.T < 0) return fail(e, VC_EINVAL, "empty text or negative prompt length");
    const int n_cols = j.T + 1;                           // T+K columns minus the K-1 dropped ones (:967)
    const int cap_len = j.Lx * (e->cfg.encodec_sr / 5);
    int steps = std::max(0, cap_len - n_cols + 1) + K;
    // The reference grows its cache as it goes and normally stops at the terminator long before the
    // length cap (voicecraft.py:1041-1045), so the worst case is not required to fit: the step budget
    // is clamped to the room there is, the in-kernel capacity guard (vc_tokens.hip advance_phase) ends a
    // sequence that really runs out of positions, and only that case is reported (VC_ECAP, below).
    const int room = e->S_max - (j.Lx + n_cols) - 1;
    if (room < K + 1)
      return fail(e, VC_ECAP, "sequence %d: the prompt alone takes %d of max_positions %d", b, j.Lx + n_cols, e->S_max);
    steps = std::min(steps, room);
    max_steps = std::max(max_steps, steps);
  }
  max_steps = std::min(max_steps, e->gen_cap);
  // the device word share_len is non-zero only inside this call, whichever way it ends (a later vc_eval_forward or
  // vc_edit would otherwise read positions below it from sequence 0's cache)
  struct ShareGuard {
    vc_engine* e; hipStream_t s; bool armed = false;
    ~ShareGuard() { if (armed) (void)hipMemsetAsync(e->share_len, 0, sizeof(int), s); }
  } share_guard{e, s};
  // (a first call of this shape / option state captures its decode graphs HERE, ahead of both timers - ADVICE r05: between ev[0] and
  // ev[1] the host time of capture + instantiation showed up as prefill time of that first call)
  {
    SampleArgs sa0 = make_sample_args(e, B, 1);
    if (int rc0 = decode_loop(e, sa0, B, 1, grouped, sc, max_steps, nullptr, s, true)) return rc0;
  }
  HIPCHK(e, hipEventRecord(e->ev[0], s));
  // ---- prompts + ONE prefill over all of them; best-of-N prefills once and replicates the cache
  {
    std::vector<PromptArgs> pas(jobs.size());
    std::vector<int> slots(jobs.size());
    for (int b = 0; b < (int)jobs.size(); ++b) {
      const TtsJob& j = jobs[b];
      fill_prompt_common(e, pas[b], j.x, j.Lx, j.y, j.T);
      pas[b].n_seg = 1; pas[b].n_cols = j.T + 1;
      pas[b].seg[0] = Segment{0, j.T + 1, 0, j.T, -1, -1};
      pas[b].skip = (b > 0) ? shared_prefix : 0;          // the shared text prefix is prefilled once, in sequence 0
      pas[b].x_shared = jobs[0].x;                        // ... and checked on the device to be the same text (prompt_k)
      slots[b] = b;
      e->h_st[b] = init_state(e, j.Lx, j.T + 1, true, 1);
      e->h_st[b].slot = b;
    }
    e->h_flag[2] = shared_prefix;
    HIPCHK(e, hipMemcpyAsync(e->share_len, e->h_flag + 2, sizeof(int), hipMemcpyHostToDevice, s));
    share_guard.armed = shared_prefix > 0;
    int rc = prefill_batch(e, pas, slots, s);
    if (rc) return rc;
  }
  if (grouped) {
    const TtsJob& j = jobs[0];
    for (int l = 0; l < e->L; ++l) {
      const long stride = (long)e->H * e->S_max * e->hd;
      HIPCHK(e, vc_launch_copy_kv(e->layers[l].kc, stride, e->H, e->S_max, e->hd, j.Lx + j.T + 1, 0, 1, B - 1, e->dtype, s));
      HIPCHK(e, vc_launch_copy_kv(e->layers[l].vc, stride, e->H, e->S_max, e->hd, j.Lx + j.T + 1, 0, 1, B - 1, e->dtype, s));
    }
    for (int b = 1; b < B; ++b) {   // every sample starts from the same first-step logits
      HIPCHK(e, hipMemcpyAsync(e->logits + (size_t)b * K * e->V, e->logits, (size_t)K * e->V * 4, hipMemcpyDeviceToDevice, s));
      e->h_st[b] = e->h_st[0];
      e->h_st[b].slot = b;
    }
    for (int b = 0; b < B; ++b) e->h_st[b].group = 0;
  }
  HIPCHK(e, hipMemcpyAsync(e->st, e->h_st, sizeof(SeqState) * B, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemsetAsync(e->st_fin, 0, sizeof(SeqState) * B, s));
  e->h_flag[1] = B;
  e->h_flag[8] = B;
  e->h_flag[9] = B; e->h_flag[10] = B;        // the two slots of "sequences still live" (SampleArgs.host_live)
  e->host_ms[6] = 0;
  HIPCHK(e, hipMemcpyAsync(e->n_active, e->h_flag + 1, sizeof(int), hipMemcpyHostToDevice, s));
  int rc = push_sample_dyn(e, sc, forced, n_forced, logits_out, logit_steps, B, s);
  if (rc) return rc;
  rc = check_err_flag(e, s);   // also orders the pinned-buffer reuse
  if (rc) return rc;
  // ---- first sample comes from the prefill logits, then the decode loop
  SampleArgs sa = make_sample_args(e, B, 1);
  int steps_run = 0;
  HIPCHK(e, vc_launch_sample(sa, grouped, s));
  HIPCHK(e, hipEventRecord(e->ev[1], s));
  rc = decode_loop(e, sa, B, 1, grouped, sc, max_steps, &steps_run, s);
  if (rc) return rc;
  HIPCHK(e, hipEventRecord(e->ev[2], s));
  if (e->cur_rows == B) {
    HIPCHK(e, hipMemcpyAsync(e->h_st, e->st, sizeof(SeqState) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
  } else {      // the batch was re-packed onto fewer rows on the way: final states = the parked ones + the rows of the last layout, by slot
    const int Bc = e->cur_rows;
    HIPCHK(e, hipMemcpyAsync(e->h_st, e->st_fin, sizeof(SeqState) * B, hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipMemcpyAsync(e->h_st2, e->st, sizeof(SeqState) * Bc, hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
    for (int r = 0; r < Bc; ++r) {
      const int slot = e->h_st2[r].slot;
      if (slot >= 0 && slot < B) e->h_st[slot] = e->h_st2[r];
    }
    rc = check_err_flag(e, s);
    if (rc) return rc;
  }
  HIPCHK(e, hipEventElapsedTime(&e->ms[0], e->ev[0], e->ev[1]));
  HIPCHK(e, hipEventElapsedTime(&e->ms[1], e->ev[1], e->ev[2]));
  e->ms[2] = e->ms[0] + e->ms[1];
  if (steps_out) {   // steps really taken (the longest sequence), not the launched multiple of steps_per_graph
    int m = 0;
    for (int b = 0; b < B; ++b) m = std::max(m, e->h_st[b].total_steps);
    *steps_out = m;
  }
  (void)steps_run;
  return VC_OK;
}

int assemble_tts(vc_engine* e, const TtsJob& j, int slot, int64_t* res, int res_cap, int* gen_len, hipStream_t s) {
  const SeqState& st = e->h_st[slot];
  if (!st.done || st.span < 1)
    return fail(e, VC_ECAP, "generation ran out of room before it terminated (max_positions %d): raise max_positions", e->S_max);
  const int N = st.span_steps[0];
  const int Tg = N - e->K;                                  // voicecraft.py:1137
  if (Tg < 0) return fail(e, VC_ESTATE, "internal: span of %d steps", N);
  if (j.T + Tg > res_cap) return fail(e, VC_ECAP, "res capacity %d < %d", res_cap, j.T + Tg);
  AssembleArgs a;
  memset(&a, 0, sizeof a);
  a.y = j.y; a.gen = e->gen + (size_t)slot * e->gen_cap * e->K; a.K = e->K; a.T = j.T; a.res_cap = res_cap; a.res = res;
  a.n_piece = 0;
  if (j.T > 0) { a.kind[a.n_piece] = 0; a.src0[a.n_piece] = 0; a.len[a.n_piece] = j.T; a.dst0[a.n_piece] = 0; a.n_piece++; }
  if (Tg > 0) { a.kind[a.n_piece] = 1; a.src0[a.n_piece] = 0; a.len[a.n_piece] = Tg; a.dst0[a.n_piece] = j.T; a.n_piece++; }
  HIPCHK(e, vc_launch_assemble(a, s));
  *gen_len = Tg;
  return VC_OK;
}

}  // namespace

extern "C" int vc_tts(vc_engine* e, const int64_t* x_dev, int Lx, const int64_t* y_dev, int T,
                      const vc_sample_cfg* sc, int n_samples, const int64_t* forced_dev, int n_forced,
                      int64_t* res_dev, int res_cap, int* gen_len, float* logits_dev, int logit_steps,
                      int* n_steps, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (!x_dev || (!y_dev && T > 0) || !sc || !res_dev || !gen_len || n_samples < 1)
    return fail(e, VC_EINVAL, "null/invalid argument to vc_tts");
  hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
  std::vector<TtsJob> jobs{TtsJob{x_dev, Lx, y_dev, T}};
  rc = tts_run(e, jobs, n_samples, sc, forced_dev, n_forced, logits_dev, logit_steps, n_steps, s);
  if (rc) return rc;
  int slot = 0;
  if (n_samples > 1) {
    slot = -1;
    for (int b = 0; b < n_samples; ++b) if (e->h_st[b].kept && e->h_st[b].done && e->h_st[b].span >= 1) slot = b;
    if (slot < 0) return fail(e, VC_ECAP, "no sample terminated within the step budget");
  }
  rc = assemble_tts(e, jobs[0], slot, res_dev, res_cap, gen_len, s);
  if (rc) return rc;
  HIPCHK(e, hipStreamSynchronize(s));
  return VC_OK;
}

extern "C" int vc_tts_multi(vc_engine* e, int B, const int64_t* x_dev, const int32_t* x_off,
                            const int64_t* y_dev, const int32_t* y_off, const vc_sample_cfg* sc,
                            int shared_text_prefix,
                            const int64_t* forced_dev, int n_forced, int64_t* res_dev, int res_cap, int* gen_len,
                            float* logits_dev, int logit_steps, int* n_steps, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (B < 1 || !x_dev || !x_off || !y_dev || !y_off || !sc || !res_dev || !gen_len)
    return fail(e, VC_EINVAL, "null/invalid argument to vc_tts_multi");
  for (int b = 0; b < B; ++b)
    if (shared_text_prefix < 0 || shared_text_prefix >= x_off[b + 1] - x_off[b])
      return fail(e, VC_EINVAL, "shared_text_prefix %d must be shorter than every text (sequence %d has %d tokens)",
                  shared_text_prefix, b, x_off[b + 1] - x_off[b]);
  hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
  std::vector<TtsJob> jobs;
  for (int b = 0; b < B; ++b)
    jobs.push_back(TtsJob{x_dev + x_off[b], x_off[b + 1] - x_off[b], y_dev + (size_t)y_off[b] * e->K, y_off[b + 1] - y_off[b]});
  rc = tts_run(e, jobs, 1, sc, forced_dev, n_forced, logits_dev, logit_steps, n_steps, s, B > 1 ? shared_text_prefix : 0);
  if (rc) return rc;
  for (int b = 0; b < B; ++b) {
    rc = assemble_tts(e, jobs[b], b, res_dev + (size_t)b * e->K * res_cap, res_cap, &gen_len[b], s);
    if (rc) return rc;
  }
  HIPCHK(e, hipStreamSynchronize(s));
  return VC_OK;
}

// ------------------------------------------------------------------------------------- editing
extern "C" int vc_edit(vc_engine* e, const int64_t* x_dev, int Lx, const int64_t* y_dev, int T,
                       const int32_t* mask_intervals, int M, const int32_t* mask_values,
                       const vc_sample_cfg* sc, const int64_t* forced_dev, int n_forced,
                       int64_t* res_dev, int res_cap, int* res_len, float* logits_dev, int logit_steps,
                       int* n_steps, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (!x_dev || !y_dev || !mask_intervals || !mask_values || !sc || !res_dev || !res_len)
    return fail(e, VC_EINVAL, "null argument to vc_edit");
  if (M < 1 || M > e->cfg.max_n_spans || 2 * M + 1 > VC_MAX_SPANS * 2 + 1)
    return fail(e, VC_EINVAL, "number of spans %d outside [1,%d]", M, e->cfg.max_n_spans);
  if (Lx < 1 || T < 1) return fail(e, VC_EINVAL, "empty text or audio");
  hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
  const int K = e->K;
  const vc_model_cfg& c = e->cfg;
  if (c.eos > 0 && !c.reduced_eog) return fail(e, VC_EINVAL, "eos > 0 requires reduced_eog (voicecraft.py:244)");
  // non-mask intervals (voicecraft.py:620-628)
  std::vector<int> ns(M + 1), ne(M + 1);
  for (int i = 0; i <= M; ++i) {
    ns[i] = (i == 0) ? 0 : mask_intervals[2 * (i - 1) + 1];
    ne[i] = (i == M) ? T : mask_intervals[2 * i];
    if (ns[i] < 0 || ne[i] > T || ne[i] < ns[i]) return fail(e, VC_EINVAL, "mask intervals must be ordered, disjoint and inside [0,%d]", T);
  }
  for (int i = 0; i < M; ++i)
    if (mask_intervals[2 * i + 1] < mask_intervals[2 * i]) return fail(e, VC_EINVAL, "mask interval %d is reversed", i);
  for (int i = 0; i < 2 * M; ++i)
    if (mask_values[i] < 0 || mask_values[i] >= c.max_n_spans) return fail(e, VC_EINVAL, "mask value out of range");
  // segments of the prefill: every non-mask piece (delay-shifted, K extra columns), a mask
  // placeholder after each of them, and the first all-empty column of the first masked piece
  // (rearrange :239-252, shift :254-262, insert_mask :264-288, the cut at :672-679)
  PromptArgs pa;
  fill_prompt_common(e, pa, x_dev, Lx, y_dev, T);
  int col = 0, nseg = 0;
  for (int i = 0; i <= M; ++i) {
    int term = -1;
    if (c.eos > 0) term = (i == M) ? c.eos : -1;
    else if (c.reduced_eog) term = (i == M) ? c.eog : -1;
    else term = c.eog;
    const int n = (ne[i] - ns[i]) + (term >= 0 ? 1 : 0);
    if (n <= 0)   // the reference raises IndexError here (codebooks_patterns.py:174 on a zero-length piece)
      return fail(e, VC_EINVAL, "non-masked piece %d is empty (a span may not start at frame 0)", i);
    pa.seg[nseg++] = Segment{col, n + K, ns[i], ne[i] - ns[i], term, -1};
    col += n + K;
    pa.seg[nseg++] = Segment{col, 1, 0, 0, -1, mask_values[i]};
    col += 1;
  }
  pa.seg[nseg++] = Segment{col, 1, 0, 0, -1, -1};   // s = 0 of the first masked piece: all `empty`
  col += 1;
  pa.n_seg = nseg; pa.n_cols = col;
  const int cap_len = Lx * 10;
  int max_steps = std::max(0, cap_len - col + 1) + M * (K + 4) + 8;
  {   // as in tts_run: clamp to the room there is; running out of it is reported after the loop
    const int room = e->S_max - (Lx + col) - 3 * M - 1;
    if (room < M * (K + 1))
      return fail(e, VC_ECAP, "editing: the rearranged prompt alone takes %d of max_positions %d", Lx + col, e->S_max);
    max_steps = std::min(std::min(max_steps, room), e->gen_cap);
  }
  {   // decode graphs of a first call of this shape: captured ahead of both timers (as in tts_run)
    SampleArgs sa0 = make_sample_args(e, 1, (M > 1) ? 3 : 1);
    if ((rc = decode_loop(e, sa0, 1, (M > 1) ? 3 : 1, false, sc, max_steps, nullptr, s, true))) return rc;
  }
  HIPCHK(e, hipEventRecord(e->ev[0], s));
  {
    std::vector<PromptArgs> pas(1, pa);
    rc = prefill_batch(e, pas, std::vector<int>(1, 0), s);
    if (rc) return rc;
  }
  SeqState st = init_state(e, Lx, col, false, M);
  for (int i = 1; i < M; ++i) st.mask_value[i] = mask_values[M + i];   // more_mask_value (:676)
  e->h_st[0] = st;
  HIPCHK(e, hipMemcpyAsync(e->st, e->h_st, sizeof(SeqState), hipMemcpyHostToDevice, s));
  e->h_flag[1] = 1;
  e->h_flag[8] = 1;
  HIPCHK(e, hipMemcpyAsync(e->n_active, e->h_flag + 1, sizeof(int), hipMemcpyHostToDevice, s));
  rc = push_sample_dyn(e, sc, forced_dev, n_forced, logits_dev, logit_steps, 1, s);
  if (rc) return rc;
  rc = check_err_flag(e, s);
  if (rc) return rc;
  const int rps = (M > 1) ? 3 : 1;
  SampleArgs sa = make_sample_args(e, 1, rps);
  HIPCHK(e, vc_launch_sample(sa, false, s));
  HIPCHK(e, hipEventRecord(e->ev[1], s));
  int steps_run = 0;
  rc = decode_loop(e, sa, 1, rps, false, sc, max_steps, &steps_run, s);
  if (rc) return rc;
  HIPCHK(e, hipEventRecord(e->ev[2], s));
  HIPCHK(e, hipMemcpyAsync(e->h_st, e->st, sizeof(SeqState), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  HIPCHK(e, hipEventElapsedTime(&e->ms[0], e->ev[0], e->ev[1]));
  HIPCHK(e, hipEventElapsedTime(&e->ms[1], e->ev[1], e->ev[2]));
  e->ms[2] = e->ms[0] + e->ms[1];
  if (n_steps) *n_steps = e->h_st[0].total_steps;
  (void)steps_run;
  const SeqState& fs = e->h_st[0];
  if (!fs.done || fs.span < M)
    return fail(e, VC_ECAP, "editing ran out of room before it terminated (max_positions %d): raise max_positions", e->S_max);
  // res = nonmask_0, gen_0, nonmask_1, gen_1, ..., nonmask_M (voicecraft.py:890-898)
  AssembleArgs a;
  memset(&a, 0, sizeof a);
  a.y = y_dev; a.gen = e->gen; a.K = K; a.T = T; a.res_cap = res_cap; a.res = res_dev;
  int dst = 0, g0 = 0;
  for (int i = 0; i <= M; ++i) {
    if (ne[i] > ns[i]) { int p = a.n_piece++; a.kind[p] = 0; a.src0[p] = ns[i]; a.len[p] = ne[i] - ns[i]; a.dst0[p] = dst; dst += ne[i] - ns[i]; }
    if (i < M) {
      const int N = fs.span_steps[i], Tg = N - K;
      if (Tg < 0) return fail(e, VC_ESTATE, "internal: span of %d steps", N);
      if (Tg > 0) { int p = a.n_piece++; a.kind[p] = 1; a.src0[p] = g0; a.len[p] = Tg; a.dst0[p] = dst; dst += Tg; }
      g0 += N;
    }
  }
  if (dst > res_cap) return fail(e, VC_ECAP, "res capacity %d < %d", res_cap, dst);
  HIPCHK(e, vc_launch_assemble(a, s));
  HIPCHK(e, hipStreamSynchronize(s));
  *res_len = dst;
  return VC_OK;
}

// Layout of one utterance's training sequence (host only; also exported as vc_eval_layout so that it can be checked
// against the oracle without a GPU): the segment table prompt_k consumes (rearrange :239-252, shift :254-262,
// insert_mask :264-288: every non-masked piece, then every masked piece + eog, a placeholder after each piece but the
// last) and, per row of [text ; audio columns] and codebook, the target: piece column s, codebook q predicts piece token
// t = s - q (revert_pattern_logits with is_model_output, codebooks_patterns.py:209-215, :247-266).
// tgt: >= 0 index into the call's y ([frames][K], y_frame_off = first frame of this utterance); -1 none; <= -2 constant.
int eval_layout(const vc_model_cfg& c, int K, int Lx, int T, const int32_t* iv, int M, const int32_t* mv, int y_frame_off,
                Segment* seg, int* n_seg, int* n_cols, std::vector<int>* tgt, std::string* why) {
  char buf[160];
  if (Lx < 1 || T < 1) { *why = "empty text or audio"; return 1; }
  if (M < 1 || M > c.max_n_spans) { snprintf(buf, sizeof buf, "%d spans outside [1,%d]", M, c.max_n_spans); *why = buf; return 1; }
  for (int j = 0; j < M; ++j) {
    if (mv[j] < 0 || mv[j] >= c.max_n_spans) { *why = "mask value out of range"; return 1; }
    const int lo = j ? iv[2 * (j - 1) + 1] : 0;
    if (iv[2 * j] < lo || iv[2 * j + 1] < iv[2 * j] || iv[2 * j + 1] > T) {
      snprintf(buf, sizeof buf, "mask intervals must be ordered, disjoint and inside [0,%d]", T); *why = buf; return 1;
    }
  }
  struct Piece { int s, e, term; };
  std::vector<Piece> pieces;
  for (int j = 0; j <= M; ++j) {                   // non-masked pieces, the terminator as rearrange() places it
    const int ps = j ? iv[2 * (j - 1) + 1] : 0, pe = (j == M) ? T : iv[2 * j];
    int term;
    if (c.eos > 0) term = (j == M) ? c.eos : -1;
    else if (c.reduced_eog) term = (j == M) ? c.eog : -1;
    else term = c.eog;
    pieces.push_back(Piece{ps, pe, term});
  }
  for (int j = 0; j < M; ++j) pieces.push_back(Piece{iv[2 * j], iv[2 * j + 1], c.eog});     // masked pieces, always + eog
  if ((int)(2 * pieces.size()) > VC_MAX_SEG) { *why = "too many segments"; return 1; }
  int col = 0, ns = 0;
  tgt->assign((size_t)Lx * K, -1);                 // text rows carry no target
  for (size_t j = 0; j < pieces.size(); ++j) {
    const Piece& p = pieces[j];
    const int n = (p.e - p.s) + (p.term >= 0 ? 1 : 0);
    if (n <= 0) { snprintf(buf, sizeof buf, "piece %d is empty (the reference raises inside get_pattern)", (int)j); *why = buf; return 1; }
    seg[ns++] = Segment{col, n + K, p.s, p.e - p.s, p.term, -1};
    for (int sidx = 0; sidx < n + K; ++sidx)
      for (int q = 0; q < K; ++q) {
        const int t = sidx - q;
        int v = -1;
        if (t >= 0 && t < n) v = (t < p.e - p.s) ? (int)(((long)y_frame_off + p.s + t) * K + q) : -(p.term + 2);
        tgt->push_back(v);
      }
    col += n + K;
    if (j + 1 < pieces.size()) {                   // placeholder: values = emb_inds_use + emb_inds_use (:274)
      seg[ns++] = Segment{col, 1, 0, 0, -1, mv[j % M]};
      for (int q = 0; q < K; ++q) tgt->push_back(-1);
      col += 1;
    }
  }
  *n_seg = ns; *n_cols = col;
  return 0;
}

// ------------------------------------------------------------------------------------- training objective
// VoiceCraft.forward (voicecraft.py:472-559) with given mask intervals (validated on MI355X: tests/test_gpu_forward.py).
// The decoder pass is the prefill's (row stream over all utterances, block GEMM + tile attention); after every pass the
// heads run over its rows 16 at a time and ce_rows_k turns each group's logits into per-row terms straight away, so the
// [rows][K][V] logits of the whole batch never exist.
extern "C" int vc_eval_forward(vc_engine* e, int B, const int64_t* x_dev, const int32_t* x_off,
                               const int64_t* y_dev, const int32_t* y_off,
                               const int32_t* spans, const int32_t* span_off, const int32_t* mask_values,
                               double* nll_sum, int64_t* hits, int64_t* n_targets,
                               float* nll_dev, int32_t* tgt_dev, int64_t nll_cap, int64_t* n_rows_out, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (B < 1 || !x_dev || !x_off || !y_dev || !y_off || !spans || !span_off || !mask_values || !nll_sum || !hits || !n_targets)
    return fail(e, VC_EINVAL, "null/invalid argument to vc_eval_forward");
  if (B > e->B_max) return fail(e, VC_ECAP, "%d utterances, the engine was created for max_seqs %d", B, e->B_max);
  hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
  const int K = e->K;
  const vc_model_cfg& c = e->cfg;
  if (c.eos > 0 && !c.reduced_eog) return fail(e, VC_EINVAL, "eos > 0 requires reduced_eog (voicecraft.py:244)");
  if (!e->ce_tgt) {
    if ((rc = dalloc(e, &e->ce_tgt, (size_t)e->emb_cap * K))) return rc;
    if ((rc = dalloc(e, &e->ce_hit, (size_t)e->emb_cap * K))) return rc;
    if ((rc = dalloc(e, &e->ce_nll, (size_t)e->emb_cap * K))) return rc;
    if ((rc = dalloc(e, &e->ce_sum, (size_t)VC_MAX_CODEBOOKS))) return rc;
    if ((rc = dalloc(e, &e->ce_hits, (size_t)VC_MAX_CODEBOOKS))) return rc;
    if ((rc = dalloc(e, &e->ce_cnt, (size_t)VC_MAX_CODEBOOKS))) return rc;
  }
  // ---- per utterance: the segment table of its training sequence and the target of every (row, codebook)
  std::vector<PromptArgs> pas(B);
  std::vector<std::vector<int>> tgts(B);          // [rows_i][K]
  for (int i = 0; i < B; ++i) {
    const int Lx = x_off[i + 1] - x_off[i], T = y_off[i + 1] - y_off[i], M = span_off[i + 1] - span_off[i];
    PromptArgs& pa = pas[i];
    fill_prompt_common(e, pa, x_dev + x_off[i], std::max(Lx, 0), y_dev + (size_t)y_off[i] * K, std::max(T, 0));
    std::string why;
    if (eval_layout(c, K, Lx, T, spans + 2 * (size_t)span_off[i], M, mask_values + span_off[i], y_off[i], pa.seg, &pa.n_seg,
                    &pa.n_cols, &tgts[i], &why))
      return fail(e, VC_EINVAL, "utterance %d: %s", i, why.c_str());
    if (Lx + pa.n_cols > e->S_max) return fail(e, VC_ECAP, "utterance %d: %d positions, max_positions is %d", i, Lx + pa.n_cols, e->S_max);
  }
  HIPCHK(e, hipMemsetAsync(e->ce_sum, 0, sizeof(double) * VC_MAX_CODEBOOKS, s));
  HIPCHK(e, hipMemsetAsync(e->ce_hits, 0, sizeof(long long) * VC_MAX_CODEBOOKS, s));
  HIPCHK(e, hipMemsetAsync(e->ce_cnt, 0, sizeof(long long) * VC_MAX_CODEBOOKS, s));
  HIPCHK(e, hipEventRecord(e->ev[0], s));
  const int chunk = e->prefill_rows_per_pass;
  int64_t rows_out = 0;
  size_t i0 = 0;
  std::vector<int> tg_host;
  while (i0 < (size_t)B) {                         // groups of utterances that fit the row arena (as prefill_batch)
    size_t i1 = i0;
    int R = 0;
    auto rows_of = [](const PromptArgs& pa) { return ((pa.Lx + pa.n_cols) + 15) & ~15; };
    while (i1 < (size_t)B && (i1 == i0 || R + rows_of(pas[i1]) <= e->emb_cap)) { R += rows_of(pas[i1]); ++i1; }
    if (R > e->emb_cap) return fail(e, VC_ECAP, "an utterance of %d rows does not fit the row arena of %d rows", R, e->emb_cap);
    HIPCHK(e, hipMemsetAsync(e->pre_row_pos, 0xFF, (size_t)R * sizeof(int), s));
    HIPCHK(e, hipMemsetAsync(e->pre_row_seq, 0, (size_t)R * sizeof(int), s));
    HIPCHK(e, hipMemsetAsync(e->emb, 0, (size_t)R * e->d * sizeof(float), s));
    tg_host.assign((size_t)R * K, -1);
    int row0 = 0;
    for (size_t i = i0; i < i1; ++i) {
      PromptArgs& pa = pas[i];
      const int rows = pa.Lx + pa.n_cols;
      pa.seq = (int)(i - i0); pa.row0 = row0;     // a cache slot per utterance of the group
      pa.emb = e->emb; pa.row_seq = e->pre_row_seq; pa.row_pos = e->pre_row_pos; pa.err = e->err_flag;
      pa.logit_row = nullptr;
      HIPCHK(e, vc_launch_prompt(pa, s));
      if ((size_t)rows * K != tgts[i].size()) return fail(e, VC_ESTATE, "internal: target table of utterance %d", (int)i);
      std::copy(tgts[i].begin(), tgts[i].end(), tg_host.begin() + (size_t)row0 * K);
      row0 += (rows + 15) & ~15;
    }
    // the host table must outlive the copy: synchronous copy (a few hundred KB, once per group)
    HIPCHK(e, hipStreamSynchronize(s));
    HIPCHK(e, hipMemcpy(e->ce_tgt, tg_host.data(), (size_t)R * K * sizeof(int), hipMemcpyHostToDevice));
    for (int r0 = 0; r0 < R; r0 += chunk) {
      RowSrc rs{};
      rs.h_in = e->emb + (size_t)r0 * e->d;
      rs.row_seq = e->pre_row_seq + r0; rs.row_pos = e->pre_row_pos + r0;
      rs.n_rows = std::min(chunk, R - r0);
      if (rs.n_rows > VC_ROWS) { rs.nsplit = 1; rs.tiled = getenv("VC_NO_TILE_ATTN") ? 0 : 1; rc = prefill_rows(e, rs, s); }
      else { rs.nsplit = attn_nsplit(e, rs.n_rows); rc = forward_rows(e, rs, s); }
      if (rc) return rc;
      for (int g0 = 0; g0 < rs.n_rows; g0 += VC_ROWS) {
        const int n = std::min(VC_ROWS, rs.n_rows - g0);
        if ((rc = run_heads16(e, nullptr, n, g0, 0, nullptr, s))) return rc;
        CeArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.logits = e->logits; ca.y = y_dev; ca.tgt = e->ce_tgt + (size_t)(r0 + g0) * K;
        ca.nll = e->ce_nll + (size_t)(r0 + g0) * K; ca.hit = e->ce_hit + (size_t)(r0 + g0) * K;
        ca.n_rows = n; ca.K = K; ca.V = e->V; ca.err = e->err_flag;
        HIPCHK(e, vc_launch_ce(ca, s));
      }
    }
    CeReduceArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.nll = e->ce_nll; ra.hit = e->ce_hit; ra.tgt = e->ce_tgt; ra.n_rows = R; ra.K = K;
    ra.nll_sum = e->ce_sum; ra.hits = e->ce_hits; ra.count = e->ce_cnt;
    HIPCHK(e, vc_launch_ce_reduce(ra, s));
    if (nll_dev && tgt_dev) {                      // parity hook: the per-row terms of this group, appended
      if (rows_out + R > nll_cap) return fail(e, VC_ECAP, "nll capacity %lld < %lld rows", (long long)nll_cap, (long long)(rows_out + R));
      HIPCHK(e, hipMemcpyAsync(nll_dev + rows_out * K, e->ce_nll, (size_t)R * K * sizeof(float), hipMemcpyDeviceToDevice, s));
      HIPCHK(e, hipMemcpyAsync(tgt_dev + rows_out * K, e->ce_tgt, (size_t)R * K * sizeof(int), hipMemcpyDeviceToDevice, s));
    }
    rows_out += R;
    HIPCHK(e, hipStreamSynchronize(s));          // the next group reuses the arena, the target table and the cache slots
    i0 = i1;
  }
  HIPCHK(e, hipEventRecord(e->ev[1], s));
  rc = check_err_flag(e, s);
  if (rc) return rc;
  double h_sum[VC_MAX_CODEBOOKS];
  long long h_hits[VC_MAX_CODEBOOKS], h_cnt[VC_MAX_CODEBOOKS];
  HIPCHK(e, hipMemcpy(h_sum, e->ce_sum, sizeof h_sum, hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemcpy(h_hits, e->ce_hits, sizeof h_hits, hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemcpy(h_cnt, e->ce_cnt, sizeof h_cnt, hipMemcpyDeviceToHost));
  for (int k = 0; k < K; ++k) { nll_sum[k] = h_sum[k]; hits[k] = h_hits[k]; }
  *n_targets = h_cnt[0];
  if (n_rows_out) *n_rows_out = rows_out;
  HIPCHK(e, hipEventElapsedTime(&e->ms[0], e->ev[0], e->ev[1]));
  e->ms[1] = 0.f; e->ms[2] = e->ms[0];
  return VC_OK;
}

extern "C" int vc_eval_layout(const vc_model_cfg* cfg, int Lx, int T, const int32_t* spans, int M, const int32_t* mask_values,
                              int y_frame_off, int32_t* seg_out, int* n_seg, int* n_cols, int32_t* tgt_out, int64_t tgt_cap) {
  if (!cfg || !spans || !mask_values || !seg_out || !n_seg || !n_cols || !tgt_out) return VC_EINVAL;
  Segment seg[VC_MAX_SEG];
  std::vector<int> tgt;
  std::string why;
  if (eval_layout(*cfg, cfg->n_codebooks, Lx, T, spans, M, mask_values, y_frame_off, seg, n_seg, n_cols, &tgt, &why)) return VC_EINVAL;
  if ((int64_t)tgt.size() > tgt_cap) return VC_ECAP;
  for (int i = 0; i < *n_seg; ++i) {
    int32_t* o = seg_out + 6 * i;
    o[0] = seg[i].col0; o[1] = seg[i].ncols; o[2] = seg[i].src0; o[3] = seg[i].src_len; o[4] = seg[i].term; o[5] = seg[i].mask_value;
  }
  std::copy(tgt.begin(), tgt.end(), tgt_out);
  return VC_OK;
}

// ------------------------------------------------------------------------------------- hooks
extern "C" int vc_debug_read(vc_engine* e, const char* name, void* host_dst, int64_t nbytes) {
  int rc = check_ready(e);
  if (rc) return rc;
  const void* src = nullptr;
  const void* host_src = nullptr;
  int64_t avail = 0;
  const std::string n = name ? name : "";
  if (n == "logits") { src = e->logits; avail = (int64_t)e->NS * e->K * e->V * 4; }
  else if (n == "hA") { src = e->hA; avail = (int64_t)VC_ROWS * e->d * 4; }
  else if (n == "hB") { src = e->hB; avail = (int64_t)VC_ROWS * e->d * 4; }
  else if (n == "q") { src = e->q; avail = (int64_t)VC_ROWS * e->d * 4; }
  else if (n == "parts") { src = e->parts; avail = (int64_t)VC_MAX_KSPLIT * VC_SLAB_ROWS * e->d * 4; }
  else if (n == "emb") { src = e->emb; avail = (int64_t)e->emb_cap * e->d * 4; }
  else if (n == "dec_h") { src = e->dec_h; avail = (int64_t)e->NS * e->d * 4; }
  else if (n == "gen") { src = e->gen; avail = (int64_t)e->B_max * e->gen_cap * e->K * 4; }
  else if (n == "state") { src = e->st; avail = (int64_t)sizeof(SeqState) * e->NS; }
  else if (n == "kcache0") { src = e->layers[0].kc; avail = (int64_t)e->B_max * e->H * e->S_max * e->hd * e->esz; }
  else if (n == "vcache0") { src = e->layers[0].vc; avail = (int64_t)e->B_max * e->H * e->S_max * e->hd * e->esz; }
  else if (n == "pe") { src = e->pe; avail = (int64_t)e->S_max * e->d * 4; }
  else if (n == "sampler_ts") { src = e->dbg_ts; avail = 16 * 8; }
  else if (n == "kernel_ts") { src = e->dbg_ts; avail = 64 * 8; }
  else if (n == "host_ms") { host_src = e->host_ms; avail = 8 * 8; }
  else if (n == "options") {        // the option state as text (vc_set_option), NUL-padded
    memset(host_dst, 0, (size_t)nbytes);
    memcpy(host_dst, e->opt_state.c_str(), std::min<size_t>((size_t)nbytes > 0 ? (size_t)nbytes - 1 : 0, e->opt_state.size()));
    return VC_OK;
  }
  else if (n == "launch_counts") { host_src = vc_launch_counts; avail = VC_LC_N * 8; }     // process-wide census of kernel forms (vc_common.h)
  else return fail(e, VC_EINVAL, "unknown debug buffer '%s'", n.c_str());
  if (nbytes > avail) return fail(e, VC_ECAP, "debug buffer '%s' holds %lld bytes", n.c_str(), (long long)avail);
  if (host_src) { memcpy(host_dst, host_src, (size_t)nbytes); return VC_OK; }
  HIPCHK(e, hipDeviceSynchronize());
  HIPCHK(e, hipMemcpy(host_dst, src, (size_t)nbytes, hipMemcpyDeviceToHost));
  return VC_OK;
}

extern "C" int vc_last_timing(const vc_engine* e, float ms[3]) {
  if (!e || !ms) return VC_EINVAL;
  ms[0] = e->ms[0]; ms[1] = e->ms[1]; ms[2] = e->ms[2];
  return VC_OK;
}

// Times one kernel of the decode step in isolation (HIP events on the launch stream).
//   which = "ffn1" | "ffn2" | "qkv" | "oproj" : the rows-GEMM of layer (i % L), n_rows rows
//   which = "step" : one whole decode step (forward + heads, sampler excluded) for n_rows sequences
extern "C" int vc_bench_kernel(vc_engine* e, const char* which, int n_rows, int iters, float* avg_ms,
                               double* alg_bytes, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  if (!which || n_rows < 1 || iters < 1 || !avg_ms) return fail(e, VC_EINVAL, "bad argument to vc_bench_kernel");
  const bool pf = std::string(which).rfind("pf_", 0) == 0;      // prefill block GEMM: up to VC_MAX_ROWS rows
  const bool wide = std::string(which).rfind("wd_", 0) == 0;    // a linear layer of a wide decode step: 17..VC_MAX_SEQS rows
  if (wide && n_rows <= VC_ROWS) return fail(e, VC_EINVAL, "vc_bench_kernel: '%s' takes 17..%d rows", which, VC_MAX_SEQS);
  if (n_rows > (pf ? VC_MAX_ROWS : wide ? VC_MAX_SEQS : VC_ROWS)) return fail(e, VC_EINVAL, "vc_bench_kernel: %d rows exceed %d", n_rows, pf ? VC_MAX_ROWS : wide ? VC_MAX_SEQS : VC_ROWS);
  if (pf && (n_rows > e->emb_cap || ((std::string(which) == "pf_attn" || std::string(which) == "pf_qkv") && n_rows > e->S_max)))
    return fail(e, VC_EINVAL, "vc_bench_kernel: %d rows exceed the prefill arena / cache", n_rows);
  hipStream_t s = stream ? (hipStream_t)stream : e->own_stream;
  const std::string w = which;
  const int d = e->d;
  // neutral rows: sequence 0, positions 0..n_rows-1 (position only matters to attention/caches)
  std::vector<int> seq(VC_MAX_SEQS, 0), pos(VC_MAX_SEQS, 0);
  for (int i = 0; i < VC_MAX_SEQS; ++i) pos[i] = i;
  if (std::string(which).rfind("attn", 0) == 0)          // attention: a 16 s utterance's worth of cached positions
    for (int i = 0; i < VC_ROWS; ++i) pos[i] = std::min(e->S_max - 1, 883);
  if (pf) {   // prefill row tables: sequence 0, positions 0..n_rows-1
    std::vector<int> pseq(n_rows, 0), ppos(n_rows);
    for (int i = 0; i < n_rows; ++i) ppos[i] = i;
    HIPCHK(e, hipMemcpyAsync(e->pre_row_seq, pseq.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(e->pre_row_pos, ppos.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipStreamSynchronize(s));                   // the staging vectors die with this scope
  }
  if (w == "wd_attn")      // every row its own sequence slot, at the last position of a 16 s utterance: n_rows caches streamed
    for (int i = 0; i < VC_MAX_SEQS; ++i) { seq[i] = i % std::max(1, e->B_max); pos[i] = std::min(e->S_max - 1, 883); }
  const int n_tab = wide ? std::min(e->NS, VC_MAX_SEQS) : VC_ROWS;      // (wide: rows 0..n_rows-1 of sequence 0, positions 0.. - needs max_positions >= n_rows)
  if (wide && (n_rows > e->NS || n_rows > e->S_max || n_rows > e->B_max)) return fail(e, VC_EINVAL, "vc_bench_kernel: %d rows exceed max_seqs / max_positions", n_rows);
  HIPCHK(e, hipMemcpyAsync(e->dec_row_seq, seq.data(), (size_t)n_tab * 4, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemcpyAsync(e->dec_row_pos, pos.data(), (size_t)n_tab * 4, hipMemcpyHostToDevice, s));
  if (wide) {
    HIPCHK(e, hipMemsetAsync(e->act, 0, (size_t)n_rows * 4 * d * e->esz, s));
    HIPCHK(e, hipMemsetAsync(e->xn, 0, (size_t)n_rows * d * e->esz, s));
  }
  HIPCHK(e, hipMemsetAsync(e->dec_h, 0, (size_t)VC_ROWS * d * 4, s));
  HIPCHK(e, hipMemsetAsync(e->hA, 0, (size_t)VC_ROWS * d * 4, s));
  HIPCHK(e, hipMemsetAsync(e->hB, 0, (size_t)VC_ROWS * d * 4, s));
  HIPCHK(e, hipMemsetAsync(e->parts, 0, (size_t)VC_MAX_KSPLIT * VC_SLAB_ROWS * d * 4, s));
  HIPCHK(e, hipMemsetAsync(e->act, 0, (size_t)VC_ROWS * 4 * d * e->esz, s));
  HIPCHK(e, hipMemsetAsync(e->xn, 0, (size_t)VC_ROWS * d * e->esz, s));
  HIPCHK(e, hipMemsetAsync(e->att_o, 0, (size_t)VC_ROWS * e->H * VC_MAX_NSPLIT * e->hd * 4, s));
  HIPCHK(e, hipMemsetAsync(e->att_ml, 0, (size_t)VC_ROWS * e->H * VC_MAX_NSPLIT * 2 * 4, s));
  HIPCHK(e, hipMemsetAsync(e->logit_row, 0, VC_ROWS * 4, s));
  RowSrc rs{};
  rs.h_in = e->dec_h; rs.row_seq = e->dec_row_seq; rs.row_pos = e->dec_row_pos;
  rs.n_rows = n_rows; rs.nsplit = attn_nsplit(e, n_rows); rs.nt = 1;
  if (n_rows >= 2 && n_rows <= fr_max_rows(e)) rs.nsplit = fr_nsplit(e, n_rows);
  bool hot = false;
  std::string w2 = w;
  if (w.size() > 4 && w.substr(w.size() - 4) == "_hot") { hot = true; w2 = w.substr(0, w.size() - 4); }
  auto one = [&](int i) -> int {
    Layer& ly = e->layers[hot ? 0 : i % e->L];   // _hot: the same 8-34 MB every launch (cache-resident)
    const std::string& w = w2;
    const bool split_ln = n_rows >= e->ln_split_rows;   // the engine then normalises in ln_rows_k and takes the plain prologue
    if (wide) {      // one linear layer of a 17..64-row step, in the form prefill_rows launches it there (options wide_gemm / wd_stage / qkv16)
      const bool wd = use_wd(e, n_rows);
      RowSrc rw = rs;
      rw.n_active = e->one; rw.nsplit = 1;
      if (w == "wd_ffn1") {
        GemmArgs g = base_args(e, rw, e->p_f1, 4 * d, d);
        g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.x_in = e->xn; g.x_ld = d; g.out = e->act; g.out_ld = 4 * d; g.mt = 2;
        if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_RELU, 1, 1, s));
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_RELU, 1, 1, s));
      } else if (w == "wd_ffn2") {
        GemmArgs g = base_args(e, rw, e->p_f2, d, 4 * d);
        g.Wp = ly.W2; nt_bit(e, g, NT_F2); g.x_in = e->act; g.x_ld = 4 * d; g.part_out = e->parts; g.mt = 2;
        if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_PART, wd_ksplit(e, d, 4 * d), 1, s));
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_f2.ksplit, 1, s));
      } else if (w == "wd_oproj") {
        GemmArgs g = base_args(e, rw, e->p_o, d, d);
        g.Wp = ly.Wo; nt_bit(e, g, NT_O); g.x_in = e->xn; g.x_ld = d; g.part_out = e->parts; g.mt = 2;
        if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_PART, wd_ksplit(e, d, d), 1, s));
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_o.ksplit, 1, s));
      } else if (w == "wd_qkv") {
        GemmArgs g = base_args(e, rw, e->p_qkv, 3 * d, d);
        g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.x_in = e->xn; g.x_ld = d; g.mt = 2;
        g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
        if (use_qkv16(e, n_rows, 2)) {
          g.Wp = ly.Wqkv16; g.n_tiles = e->p_qkv16.n_tiles; g.KT = e->p_qkv16.KT; g.nchunk = e->p_qkv16.nchunk;
          if (wd) HIPCHK(e, vc_launch_gemm_wd(g, e->dtype, EPI_QKV16, 1, 1, s));
          else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV16, 1, 1, s));
        } else {
          HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV, 1, 1, s));
        }
      } else if (w == "wd_ln") {      // the per-row LayerNorm launch in front of the FFN up-projection (h + bias + the out-projection's slabs)
        GemmArgs g = base_args(e, rw, e->p_f1, 4 * d, d);
        g.h_in = e->hA; g.h_out = e->hB; g.parts = e->parts; g.n_parts = wd ? wd_ksplit(e, d, d) : e->p_o.ksplit; g.prev_bias = ly.bo; g.has_prev_bias = 1;
        g.x_out = e->xn;
        HIPCHK(e, vc_launch_ln_rows(g, e->dtype, s));
      } else if (w == "wd_attn") {    // the decode attention of that step: n_rows rows of ONE sequence slot at its last position (timing only)
        AttnArgs a;
        memset(&a, 0, sizeof a);
        a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc; a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
        a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = 1; a.scale = 1.0f / sqrtf((float)e->hd);
        a.row_seq = rs.row_seq; a.row_pos = rs.row_pos; a.n_rows = n_rows; a.att_o = e->att_o; a.att_ml = e->att_ml;
        a.n_active = e->one; a.dbg_ts = e->dbg_ts; a.share_len = e->share_len; a.nt = attn_nt_for(e, n_rows); a.x_out = e->xn;
        a.fast = e->attn_fast;
        HIPCHK(e, vc_launch_attn(a, e->dtype, n_rows, s));
      } else {
        return fail(e, VC_EINVAL, "unknown kernel '%s'", which);
      }
      return VC_OK;
    }
    if (n_rows >= 2 && n_rows <= fr_max_rows(e) && (w == "ffn1" || w == "ffn2" || w == "qkv" || w == "oproj")) {
      // the forms a step of this many rows really launches (forward_rows_fr): finished rows in, finished rows out
      if (w == "ffn1") {
        GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
        g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.wg = ly.wg_1; g.h_in = e->hA; g.out = e->act; g.out_ld = 4 * d;
        g.mt = rs.n_rows > VC_FR_MAX_ROWS ? 4 : lnw_two(e, rs.n_rows) ? 3 : 0;
        if (hq_on(e)) { g.x_in = e->hqA; g.x_ld = d; g.row_mu = e->row_mu[0]; g.row_mu_out = e->row_mu[1]; }     // the form forward_rows_fr launches
        HIPCHK(e, vc_launch_gemm(g, e->dtype, hq_on(e) ? PRO_LNQ : PRO_LNW, EPI_RELU, 1, 1, s));
      } else if (w == "qkv") {
        GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
        g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.wg = ly.wg_qkv; g.h_in = e->hB; g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
        g.mt = rs.n_rows > VC_FR_MAX_ROWS ? 4 : lnw_two(e, rs.n_rows) ? 3 : 0;
        if (hq_on(e)) { g.x_in = e->hqB; g.x_ld = d; g.row_mu = e->row_mu[0]; g.row_mu_out = e->row_mu[1]; }
        if (qp_on(e, ly, n_rows)) { g.Wp = ly.Wqkv8; HIPCHK(e, vc_launch_gemm_qp(g, e->dtype, s)); }
        else HIPCHK(e, vc_launch_gemm(g, e->dtype, hq_on(e) ? PRO_LNQ : PRO_LNW, EPI_QKV, 1, 1, s));
      } else if (w == "ffn2") {
        GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
        g.Wp = ly.W28; g.bias = ly.b2; g.x_in = e->act; g.x_ld = 4 * d; g.h_in = e->hA; g.h_out = e->hB;
        if (e->fr_pair && vc_gemm_frp_ok(n_rows, d, 4 * d, e->dtype)) HIPCHK(e, vc_launch_gemm_frp(g, e->dtype, s));
        else HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_PLAIN, s));
      } else {
        GemmArgs g = base_args(e, rs, e->p_o, d, d);
        g.Wp = ly.Wo8; g.bias = ly.bo; g.h_in = e->hB; g.h_out = e->hA;
        if (fr_nsplit(e, n_rows) == 1) { g.x_in = e->xn; g.x_ld = d; HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_PLAIN, s)); }
        else { g.att_o = e->att_o; g.att_ml = e->att_ml; g.nsplit = fr_nsplit(e, n_rows); g.att_p16 = (e->att_p16 && e->dtype == VC_DTYPE_BF16) ? 1 : 0; HIPCHK(e, vc_launch_gemm_fr(g, e->dtype, PRO_ATT, s)); }
      }
      return VC_OK;
    }
    if (w == "ffn1") {
      GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
      g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.h_in = e->hA; g.h_out = e->hB; g.parts = e->parts; g.n_parts = e->p_o.ksplit;
      g.prev_bias = ly.bo; g.has_prev_bias = 1; g.wg = ly.wg_1; g.out = e->act; g.out_ld = 4 * d;
      if (split_ln) { g.x_in = e->xn; g.x_ld = d; HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_RELU, 1, 1, s)); }
      else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LN, EPI_RELU, 1, 1, s));
    } else if (w == "ffn2" && fd_one(e, n_rows)) {     // one row, finished by the producer (forward_rows, option fr_one)
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W28; g.bias = ly.b2; g.x_in = e->act; g.x_ld = 4 * d; g.h_in = e->hA; g.h_out = e->hB;
      HIPCHK(e, vc_launch_gemm_fr1(g, e->dtype, PRO_PLAIN, EPI_RES, s));
    } else if (w == "ffn2") {
      GemmArgs g = base_args(e, rs, e->p_f2, d, 4 * d);
      g.Wp = ly.W2; nt_bit(e, g, NT_F2); g.x_in = e->act; g.x_ld = 4 * d; g.part_out = e->parts;
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_PART, e->p_f2.ksplit, 1, s));
    } else if (w == "qkv") {
      GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
      g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.h_in = e->hB; g.h_out = e->hA; g.parts = e->parts; g.n_parts = e->p_f2.ksplit;
      g.prev_bias = ly.b2; g.has_prev_bias = 1; g.wg = ly.wg_qkv; g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
      if (fd_one(e, n_rows)) { g.h_out = nullptr; g.n_parts = 0; g.has_prev_bias = 0; }       // the row in hB is finished
      if (fd_one(e, n_rows) && e->qkv_p8 && ly.Wqkv8 && !split_ln) {                          // ... and the step runs the paired 8-channel form
        g.Wp = ly.Wqkv8;
        HIPCHK(e, vc_launch_gemm_fr1(g, e->dtype, PRO_LN, EPI_QKV, s));
        return VC_OK;
      }
      if (split_ln) { g.x_in = e->xn; g.x_ld = d; HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV, 1, 1, s)); }
      else HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_LN, EPI_QKV, 1, 1, s));
    } else if (w == "oproj") {
      GemmArgs g = base_args(e, rs, e->p_o, d, d);
      g.Wp = ly.Wo; nt_bit(e, g, NT_O); g.att_o = e->att_o; g.att_ml = e->att_ml; g.nsplit = rs.nsplit; g.part_out = e->parts;
      HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_ATT, EPI_PART, e->p_o.ksplit, 1, s));
    } else if (w == "attn") {
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc; a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
      a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = rs.nsplit; a.scale = 1.0f / sqrtf((float)e->hd);
      a.row_seq = rs.row_seq; a.row_pos = rs.row_pos; a.n_rows = rs.n_rows; a.att_o = e->att_o; a.att_ml = e->att_ml;
      a.n_active = e->one; a.dbg_ts = e->dbg_ts; a.share_len = e->share_len; a.nt = attn_nt_for(e, rs.n_rows);
      a.fast = e->attn_fast;
      a.part16 = (e->att_p16 && e->dtype == VC_DTYPE_BF16 && n_rows >= 2 && n_rows <= fr_max_rows(e) && rs.nsplit > 1) ? 1 : 0;    // as forward_rows_fr launches it
      HIPCHK(e, vc_launch_attn(a, e->dtype, rs.n_rows, s));
    } else if (w == "pf_ffn1") {      // the prefill pass's FFN up-projection on the MFMA block GEMM (X = xn, n_rows rows)
      GemmArgs g = base_args(e, rs, e->p_f1, 4 * d, d);
      g.Wp = ly.W1; nt_bit(e, g, NT_F1); g.bias = ly.b1; g.x_in = e->xn; g.x_ld = d; g.out = e->act; g.out_ld = 4 * d; g.mt = 1;
      const char* dbg_env = getenv("VC_BLK_DBG");        // tools/blk_probe.py: wrong results by design, timing only
      vc_blk_dbg_mask = dbg_env ? atoi(dbg_env) : 0;
      hipError_t le = vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_RELU, 1, 1, s);
      vc_blk_dbg_mask = 0;
      HIPCHK(e, le);
    } else if (w == "pf_qkv") {       // the prefill pass's QKV projection (X = xn; the image option "qkv16" selects)
      GemmArgs g = base_args(e, rs, e->p_qkv, 3 * d, d);
      g.Wp = ly.Wqkv; nt_bit(e, g, NT_QKV); g.bias = ly.bqkv; g.x_in = e->xn; g.x_ld = d; g.mt = 1;
      g.q_out = e->q; g.kcache = ly.kc; g.vcache = ly.vc;
      g.row_seq = e->pre_row_seq; g.row_pos = e->pre_row_pos;        // sequence 0, positions 0 .. n_rows - 1
      if (use_qkv16(e, n_rows, 1)) {
        g.Wp = ly.Wqkv16; g.n_tiles = e->p_qkv16.n_tiles; g.KT = e->p_qkv16.KT; g.nchunk = e->p_qkv16.nchunk;
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV16, 1, 1, s));
      } else {
        HIPCHK(e, vc_launch_gemm(g, e->dtype, PRO_PLAIN, EPI_QKV, 1, 1, s));
      }
    } else if (w == "pf_attn") {      // the prefill's MFMA tile attention: n_rows consecutive positions of sequence 0 (16-row tiles)
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.q = e->q; a.kcache = ly.kc; a.vcache = ly.vc; a.cache_seq_stride = (long)e->H * e->S_max * e->hd;
      a.S_max = e->S_max; a.H = e->H; a.hd = e->hd; a.hd_shift = e->hd == 32 ? 5 : e->hd == 64 ? 6 : 7; a.d = d; a.nsplit = 1; a.scale = 1.0f / sqrtf((float)e->hd);
      a.row_seq = e->pre_row_seq; a.row_pos = e->pre_row_pos; a.n_rows = n_rows; a.att_o = e->att_o; a.att_ml = e->att_ml;
      a.n_active = e->one; a.dbg_ts = e->dbg_ts; a.share_len = e->share_len; a.x_out = e->xn;
      // (the kernel a prompt of this many rows would get: prefill_batch's rule)
      if (e->tile_attn == 2 && e->dtype == VC_DTYPE_BF16 && e->hd == 128 && n_rows >= e->tile_attn_min_rows) HIPCHK(e, vc_launch_tile_attn64(a, s));
      else HIPCHK(e, vc_launch_tile_attn(a, e->dtype, s));
    } else if (w == "step") {
      int r = forward_rows(e, rs, s);
      if (r) return r;
      return run_heads(e, nullptr, n_rows, 0, nullptr, s);
    } else {
      return fail(e, VC_EINVAL, "unknown kernel '%s'", which);
    }
    return VC_OK;
  };
  for (int i = 0; i < 3; ++i) { rc = one(i); if (rc) return rc; }
  HIPCHK(e, hipEventRecord(e->ev[0], s));
  for (int i = 0; i < iters; ++i) { rc = one(i); if (rc) return rc; }
  HIPCHK(e, hipEventRecord(e->ev[1], s));
  HIPCHK(e, hipStreamSynchronize(s));
  float ms = 0;
  HIPCHK(e, hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
  *avg_ms = ms / (float)iters;
  if (alg_bytes) {
    const double es = e->esz;
    const std::string& w = w2;
    double b = 0;
    if (w == "wd_ffn1") b = 4.0 * d * d * es + n_rows * (d * es + 4.0 * d * es);
    else if (w == "wd_ffn2") b = 4.0 * d * d * es + n_rows * (4.0 * d * es + d * 4.0 * (use_wd(e, n_rows) ? wd_ksplit(e, d, 4 * d) : e->p_f2.ksplit));
    else if (w == "wd_qkv") b = 3.0 * d * d * es + n_rows * (d * es + d * 4.0 + 2.0 * d * es);
    else if (w == "wd_oproj") b = 1.0 * d * d * es + n_rows * (d * es + d * 4.0 * (use_wd(e, n_rows) ? wd_ksplit(e, d, d) : e->p_o.ksplit));
    else if (w == "wd_ln") b = n_rows * (d * 4.0 * (2 + (use_wd(e, n_rows) ? wd_ksplit(e, d, d) : e->p_o.ksplit)) + d * es);
    else if (w == "wd_attn") b = n_rows * 2.0 * d * es * (std::min(e->S_max - 1, 883) + 1);
    else if (w == "ffn1") b = 4.0 * d * d * es + n_rows * (d * 4.0 + 4.0 * d * es);
    else if (w == "ffn2") b = 4.0 * d * d * es + n_rows * (4.0 * d * es + d * 4.0);
    else if (w == "qkv") b = 3.0 * d * d * es + n_rows * (d * 4.0 + 3.0 * d * es);
    else if (w == "oproj") b = 1.0 * d * d * es + n_rows * (d * 4.0 * 2);
    else if (w == "attn") b = n_rows * 2.0 * d * es * (std::min(e->S_max - 1, 883) + 1);   // K and V of every cached position
    else if (w == "pf_qkv") b = 2.0 * n_rows * (double)d * 3.0 * d;                        // FLOPs
    else if (w == "pf_ffn1") b = 2.0 * n_rows * (double)d * 4.0 * d;                       // FLOPs, not bytes (MFMA roofline)
    else if (w == "pf_attn") b = 4.0 * d * ((double)n_rows * (n_rows + 1) / 2.0);          // FLOPs of Q K^T and P V under the causal mask
    else b = (double)e->L * (12.0 * d * d + 13.0 * d) * es + 2.0 * d * es +
             (double)e->K * ((double)d * e->P + e->P + (double)e->P * e->V + e->V) * es;
    *alg_bytes = b;
  }
  return VC_OK;
}
