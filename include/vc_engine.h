/*
 * vc_engine.h — C ABI of libvcengine.so, the MI355X (gfx950) engine for the
 * VoiceCraft token-infilling decode path.
 *
 * The reference (jasonppy/VoiceCraft) has no FFI: its boundary for this path is
 * three Python methods on an nn.Module plus two tokenizer methods.
 * Each entry point below names the reference interface it replaces; the Python
 * mirror of that interface lives in voicecraft_amd/engine.py and is the only
 * in-tree caller (ctypes).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative
 *     VC_E* code, with a human-readable message available from vc_last_error().
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller; the engine
 *     owns everything it allocates itself and frees it in vc_destroy().
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - token tensors are int64, matching the reference's LongTensors.
 *   - one engine per (process, GPU); not thread-safe (the reference is neither:
 *     inference_tts_scale.py:42 runs single-threaded under torch.no_grad()).
 */
#ifndef VC_ENGINE_H
#define VC_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_OK 0
#define VC_EINVAL (-1)   /* bad argument / shape (the reference would AssertionError) */
#define VC_ESTATE (-2)   /* call order (e.g. decode before vc_finalize_weights)       */
#define VC_EHIP (-3)     /* a HIP runtime call failed                                 */
#define VC_ECAP (-4)     /* a capacity given at vc_create / by the caller is too small */
#define VC_EMISSING (-5) /* a required weight tensor was never loaded                 */

#define VC_DTYPE_F32 0
#define VC_DTYPE_BF16 1
#define VC_DTYPE_I64 2

#define VC_MAX_CODEBOOKS 8
#define VC_MAX_SPANS 8
#define VC_MAX_SILENCE 8

typedef struct vc_engine vc_engine;

/* Model hyper-parameters = the fields of the reference's pickled `args` Namespace that
 * the inference path reads (models/voicecraft.py:105-185, config.py:55-84), plus the
 * capacities the engine sizes its HBM arenas with. */
typedef struct vc_model_cfg {
  int32_t d_model;          /* args.d_model                                   */
  int32_t nhead;            /* args.nhead                                     */
  int32_t num_layers;       /* args.num_decoder_layers                        */
  int32_t n_codebooks;      /* args.n_codebooks (K)                           */
  int32_t audio_vocab_size; /* args.audio_vocab_size (2048)                   */
  int32_t n_special;        /* args.n_special; V = audio_vocab_size+n_special */
  int32_t text_rows;        /* args.text_vocab_size + 1 (voicecraft.py:129)   */
  int32_t head_hidden;      /* audio_vocab_size // 2   (voicecraft.py:183)    */
  int32_t empty_token;      /* args.empty_token                               */
  int32_t eog;              /* args.eog                                       */
  int32_t audio_pad_token;  /* args.audio_pad_token                           */
  int32_t eos;              /* args.eos, or -1 when the checkpoint has none   */
  int32_t reduced_eog;      /* args.reduced_eog (voicecraft.py:240)           */
  int32_t encodec_sr;       /* args.encodec_sr (50)                           */
  int32_t max_n_spans;      /* args.max_n_spans: rows of mask_embedding       */
  int32_t max_seqs;         /* capacity: concurrent sequences (KV slots)      */
  int32_t max_positions;    /* capacity: cached positions per sequence        */
} vc_model_cfg;

/* Decode controls = the keyword arguments of inference_tts / inference
 * (models/voicecraft.py:908-920, :561-573). */
typedef struct vc_sample_cfg {
  int32_t top_k;            /* <=0: no top-k filter (reference default -100)  */
  float top_p;              /* >=1: no nucleus filter                         */
  float temperature;        /* 1.0: untouched                                 */
  int32_t stop_repetition;  /* <=0: silence-run penalty off                   */
  int32_t n_silence;        /* number of entries used in silence_tokens       */
  int32_t silence_tokens[VC_MAX_SILENCE];
  uint64_t seed;            /* Philox key; stream = (seed, sequence, step, codebook).  The draws follow the
                             * DISTRIBUTION of topk_sampling (voicecraft.py:71-86), not torch's generator stream; ties at
                             * the k-th value survive as in the reference; exact ties at the nucleus boundary are kept or
                             * dropped together (the reference's sort splits them arbitrarily) */
  int32_t use_graph;        /* 1: replay the decode step as a captured hipGraph */
  int32_t poll_every;       /* host polls the done flag every N steps (0 = default 16) */
  int32_t forced_mode;      /* parity hook, meaning of forced_dev: 0 = the step's FINAL tokens (teacher forcing
                             * for logits parity); 1 = the raw draws of topk_sampling (models/voicecraft.py:1033):
                             * the state machine - overrides, termination, best-of-N keep - runs on them */
} vc_sample_cfg;

/* ---- lifecycle ---------------------------------------------------------- */
int vc_create(const vc_model_cfg* cfg, int hip_device, vc_engine** out);
void vc_destroy(vc_engine* e);
const char* vc_last_error(const vc_engine* e); /* e may be NULL: last vc_create error */
const char* vc_version(void);

/* Run-time options of a finalized engine: launch-shape knobs of the decode step, the same ones the VC_* environment
 * variables preset when the engine is created (no reference counterpart: the reference has no such knobs).  Sixteen of them (round 6
 * pruned round 5's 22 to 14 - the prefetch roles and the knobs whose value every measurement had fixed are constants now, the K/V hint is a
 * second value of "nt" - and added two measurement arms, "att_p16" and "hq").  name / value:
 *   "nt"           "mask[,kv]"  mask: the weight matrices streamed with the non-temporal hint (1 QKV, 2 out-proj, 4 FFN-up, 8 FFN-down, 16 / 32
 *                  heads); kv 0 / 1 / 2 = the decode attention's K/V loads carry it never / always / from two rows per step (the separate
 *                  name "attn_nt" of rounds 4-5 is gone)
 *   "finished_rows" rows up to which a several-row decode step keeps whole residual rows instead of split-K slabs (0 = off, max 16);
 *                  "fr_pair" 1 = its FFN down-projection with two k-tiles per MFMA fragment at 2..8 rows;  "att_p16" 1 (default) = bf16 mode: the
 *                  split attention of 2..8-row steps leaves its partial outputs in bf16 for the out-projection's merge;  "hq" 1 (default) = the
 *                  producers of finished rows also store a copy of each row in the compute dtype, centred on the mean its previous LayerNorm
 *                  found, and the LayerNorm-folding consumers read that copy instead of the fp32 rows (widths whose row of the copy is whole
 *                  1 KB requests: d a multiple of 512 in bf16, of 256 in fp32; others keep the fp32 rows)
 *   "fr_one"       ONE-row steps: 1 (default) = the FFN down-projection finishes its row (no split-K slabs), 0 = off;  "qkv_p8" 1 = the
 *                  one-row QKV projection in the same paired form, 2 (default) = also the QKV projection of 2..8 finished rows behind the
 *                  centred copy ("hq"; rows_gemm_qp_k: up to 6 rows at d >= 2048, where 8 rows measured slower);  "attn_fast" 1 = decode attention without per-visit rescaling (bf16: hardware exp2)
 *   "tile_attn"    "k[,min_rows]"  prefill attention kernel (1: 16 query rows per wave; 2: 64 per workgroup, P in registers - bf16,
 *                  head_dim 128, calls whose longest prompt has at least min_rows rows)
 *   "qkv16"        1 = prefill passes and wide decode passes (17..64 rows) run the QKV projection on a 16-channel image of the folded matrix
 *                  instead of the one-row kernels' 12-channel tiles.  The image is packed for engines with max_seqs > 16 (or VC_QKV16=1 at
 *                  creation); without it the option stays 0 and wide steps fall back to the weight-stationary kernel
 *   "wide_heads"   1 (default) = decode steps of 17..64 rows run the prediction heads once on the wide-decode kernel instead of once per 16 rows
 *   "wide_gemm"    1 (default) = the linear layers of 17..64-row steps run on rows_gemm_wd_k / rows_gemm_wds_k (every row tile of the step in flight;
 *                  widths whose K does not split into 8 x {1, 2, 4, 8, 16} k-tiles, and 0, take the weight-stationary rows_gemm_mt_k of rounds 2-5);
 *                  "wd_stage" 1 (default) = that kernel takes X as whole cache lines through a wave-private LDS stage, 0 = MFMA fragments straight from L2
 *   "shrink"       1 (default) = a multi-utterance call re-packs its live sequences onto the rows of a narrower step (next power of two) as the
 *                  others retire; 0 = the step keeps its starting width until the longest sequence ends (rounds 1-5)
 *   "graph_steps"  decode steps captured per hipGraph;  "prefill_rows"  rows per prefill pass (16: decode kernels only)
 * What an option may change: nothing in the exact fp32 mode's greedy tokens (tests/test_gpu_options.py, test_gpu_one_row.py); in bf16
 * mode the forms that re-order sums or round at another place ("finished_rows", "fr_pair", "att_p16", "hq", "fr_one", "qkv_p8", "attn_fast",
 * "qkv16", "wide_heads", "wide_gemm") move head logits by bf16 rounding (tests allow 0.25 absolute), so top-k SAMPLED tokens can differ between option
 * states; the cache-policy / data-path / host-side options ("nt", "wd_stage", "graph_steps", "shrink") change no value.
 * The non-temporal mask "nt" has no bit for the finished-row producers (rows_gemm_fr_k, rows_gemm_fr2_k, row_gemm_fr1_k) and the
 * wide-decode kernels: they always stream with the hint.  Captured decode graphs are kept per option state and step width, so an
 * in-process A/B (bench.py --ab) pays for capture once per state.  Unknown names / malformed values: VC_EINVAL.
 * (Left the tree in round 6: the prefetch roles "attn_pf", "attn_pf_cut", "gemm_pf" - default-off, or inside the spread on the driver's
 * box -, and the knobs "ln_trim", "lnw_tiles", "fr_split_rows", "attn_blocks", "attn_blocks1", "ln_split_rows", "mt_tiles", whose measured
 * best value is a constant now) */
int vc_set_option(vc_engine* e, const char* name, const char* value);

/* ---- weights: replaces get_model()/load_state_dict (inference_tts_scale.py:107-125).
 * `key` is the reference state_dict key; data is fp32 (or int64 for
 * eog/eos, ignored).  `on_device` tells whether `data` is a host or device pointer.
 * Unknown keys (e.g. accuracy_metrics.*) are ignored and return VC_OK. */
int vc_load_tensor(vc_engine* e, const char* key, const void* data, int on_device,
                   int dtype, const int64_t* shape, int ndim);
/* Packs everything into the MFMA fragment layout in `compute_dtype`
 * (VC_DTYPE_BF16, or VC_DTYPE_F32 for the exact mode), builds the sinusoidal
 * position tables (models/modules/embedding.py:69-92) and frees the staging copies. */
int vc_finalize_weights(vc_engine* e, int compute_dtype);

/* ---- TTS: VoiceCraft.inference_tts (models/voicecraft.py:908-1153) and, with
 * n_samples > 1, VoiceCraft.inference_tts_batch (:1156-1439, best-of-N: the sample
 * whose first codebook terminates first is kept).
 *   x_dev      int64 [Lx]            phoneme ids
 *   y_dev      int64 [T][K]          prompt codes, time-major exactly as the caller's y[0]
 *   forced_dev int64 [n_forced][B][K] optional teacher-forcing trajectory (parity hook, see
 *                                    vc_sample_cfg.forced_mode; B = n_samples); NULL = sample
 *   res_dev    int64 [K][res_cap]    out: prompt followed by generated frames (row stride res_cap)
 *   gen_len    out (host): number of generated frames Tg; res holds T+Tg columns
 *   logits_dev float [logit_steps][B][K][V] optional: raw head outputs of the first steps
 *   n_steps    out (host, optional): decode steps taken (Tg + K)                       */
int vc_tts(vc_engine* e, const int64_t* x_dev, int Lx, const int64_t* y_dev, int T,
           const vc_sample_cfg* sc, int n_samples, const int64_t* forced_dev, int n_forced,
           int64_t* res_dev, int res_cap, int* gen_len, float* logits_dev, int logit_steps,
           int* n_steps, void* stream);

/* ---- multi-utterance TTS (SURVEY.md §8f-1 / BASELINE config 5): B independent
 * (x, y) pairs decoded as one batch; each row follows inference_tts exactly.
 *   x_dev int64 [sum Lx], y_dev int64 [sum T][K] concatenated; *_off host arrays [B+1].
 *   res_dev int64 [B][K][res_cap]; gen_len host [B].
 *   forced_dev / logits_dev: as in vc_tts, indexed by each sequence's own step count.
 *   shared_text_prefix: the first P text tokens are IDENTICAL in every sequence (the sentence-chained
 *     "Long TTS" of gradio_app.py:231-236, :249-313: every sentence is synthesised as
 *     [transcript of the voice prompt ; sentence] against the same audio prompt).  Text precedes audio in the
 *     sequence and attention is causal, so the K/V of those P positions do not depend on what follows: they
 *     are computed once (sequence 0) and every sequence's attention reads them there - exact, no copy.
 *     (The audio prompt's K/V DO depend on the whole text and cannot be shared.)  0 = nothing shared;
 *     the equality is verified on the device while the prompts are built: a text that differs from sequence 0's
 *     inside the prefix makes the call fail with VC_EINVAL before anything is decoded. */
int vc_tts_multi(vc_engine* e, int B, const int64_t* x_dev, const int32_t* x_off,
                 const int64_t* y_dev, const int32_t* y_off, const vc_sample_cfg* sc,
                 int shared_text_prefix,
                 const int64_t* forced_dev, int n_forced, int64_t* res_dev, int res_cap, int* gen_len,
                 float* logits_dev, int logit_steps, int* n_steps, void* stream);

/* ---- speech editing: VoiceCraft.inference (models/voicecraft.py:561-906).
 *   mask_intervals host int32 [M][2] (codec-frame units, as mask_interval[0])
 *   mask_values    host int32 [2M]   the reference's mask_value list (insert_mask, :264-288)
 *   res_dev int64 [K][res_cap]; res_len out (host): T' (voicecraft.py:900)            */
int vc_edit(vc_engine* e, const int64_t* x_dev, int Lx, const int64_t* y_dev, int T,
            const int32_t* mask_intervals, int M, const int32_t* mask_values,
            const vc_sample_cfg* sc, const int64_t* forced_dev, int n_forced,
            int64_t* res_dev, int res_cap, int* res_len, float* logits_dev, int logit_steps,
            int* n_steps, void* stream);

/* ---- the training objective, teacher-forced: VoiceCraft.forward (models/voicecraft.py:472-559) for B utterances
 * whose mask intervals are GIVEN (the reference draws them at random, :198-237): rearrange / delay-shift / mask
 * placeholders (:239-320), one non-cached decoder pass over [text ; rearranged audio] of every utterance (:501-512;
 * the reference pads the batch and masks the padding, the engine runs the rows unpadded), the K heads on every audio
 * position (:516), the placeholders dropped and the delay pattern reverted per piece (:374-404,
 * codebooks_patterns.py:247-266), then per codebook the cross-entropy SUM and the number of targets among the ten
 * largest logits (torchmetrics MulticlassAccuracy(top_k=10), :187-195, :540-541).
 *   x_dev int64 (all texts back to back), x_off host int32 [B+1]; y_dev int64 [frames][K] time-major (all utterances
 *   back to back), y_off host int32 [B+1] in frames
 *   spans host int32 [sum M_i][2] frame intervals, span_off host int32 [B+1]; mask_values host int32 [sum M_i]
 *   (the utterance's emb_inds_use, :270-273)
 *   out (host): nll_sum double [K] = sum of -log softmax(logits)[target]; hits int64 [K]; n_targets int64 (per codebook)
 *   nll_dev optional device float [nll_cap][K]: the per-row terms in the engine's row order (parity hook), with
 *   tgt_dev int32 [nll_cap][K] (>= 0 index into y_dev, -1 none, <= -2 constant token -(v+2)); n_rows_out = rows written */
int vc_eval_forward(vc_engine* e, int B, const int64_t* x_dev, const int32_t* x_off,
                    const int64_t* y_dev, const int32_t* y_off,
                    const int32_t* spans, const int32_t* span_off, const int32_t* mask_values,
                    double* nll_sum, int64_t* hits, int64_t* n_targets,
                    float* nll_dev, int32_t* tgt_dev, int64_t nll_cap, int64_t* n_rows_out, void* stream);

/* Host-only half of vc_eval_forward for ONE utterance (no engine, no GPU: checked against the oracle in the CPU tests):
 * the segment table of its training sequence, seg_out int32 [n_seg][6] = {first column, columns, first source frame,
 * source frames, appended terminator or -1, mask_embedding row or -1}, the number of audio columns, and the target
 * table tgt_out int32 [Lx + n_cols][K] as described above (y_frame_off = the utterance's first frame inside y_dev). */
int vc_eval_layout(const vc_model_cfg* cfg, int Lx, int T, const int32_t* spans, int M, const int32_t* mask_values,
                   int y_frame_off, int32_t* seg_out, int* n_seg, int* n_cols, int32_t* tgt_out, int64_t tgt_cap);

/* ---- delayed-codebook pattern (models/codebooks_patterns.py:151-176, :222-245 and
 * the un-shift at models/voicecraft.py:1125-1139).  Integer, bit-exact.  No engine needed.
 *   shift : z [B][K][T]  -> out [B][K][T+K]   out[q][s] = z[q][s-1-q] or `special`
 *   revert: s [B][K][S]  -> out [B][K][T]     out[q][t] = s[q][t+1+q] if t+1+q < S else `special`
 *   unshift: span [N][K] (step-major) -> out [K][N-K]  out[j][t] = span[j+t][j]         */
int vc_pattern_shift(const int64_t* z_dev, int B, int K, int T, int64_t special,
                     int64_t* out_dev, void* stream);
int vc_pattern_revert(const int64_t* s_dev, int B, int K, int S, int T, int64_t special,
                      int64_t* out_dev, void* stream);
int vc_pattern_unshift(const int64_t* span_dev, int N, int K, int64_t* out_dev, void* stream);

/* ---- parity/debug hooks (tests only) ------------------------------------ */
/* n_draws independent draws from one logits row [V] through the product sampler (temperature, top-k,
 * top-p, inverse-CDF on a Philox stream; models/voicecraft.py:26-86): out_dev int32 [n_draws].
 * Only top_k / top_p / temperature / seed of `sc` are read.  No engine needed. */
int vc_debug_sample(const float* logits_dev, int V, const vc_sample_cfg* sc, int n_draws,
                    int32_t* out_dev, void* stream);
/* Diagnosis of the BOX, not of the model (no engine needed): ONE thread walks `hops` dependent loads over a ring of `bytes` bytes and
 * stamps the per-XCD shader clock and the chip-wide 100 MHz counter around the walk.  res[0] = ns per dependent load as a lone
 * workgroup on an otherwise idle chip sees it, res[1] = the shader clock (MHz) that workgroup really ran at.  The one-sequence
 * sampler launch is exactly such a workgroup; bench.py prints both numbers (`box`) next to the sampler's time. */
int vc_box_probe(long long bytes, int hops, float res[2], void* stream);
/* Host-only (no engine, no GPU): how a decode pass of `rows` rows is launched for this model shape and compute dtype.
 * out[0] rows up to which the finished-row form applies; out[1] form of this pass (0 slabs + rows-GEMM, 1 finished rows, 2 wide
 * decode); out[2] attention splits; out[3] consumer kernel shape (GemmArgs.mt) or -1; out[4] / out[5] producer form of the
 * out-projection / FFN down-projection (1 one piece, 2 K in two halves, 0 = NOT LAUNCHABLE, -1 n/a); out[6] heads-1 folds finished
 * rows itself; out[7] the consumers' tile counts are even.
 * The default forms of rounds 5-6 (-1 where the row count does not take them): out[8] ONE row: the FFN down-projection finishes its row
 * (row_gemm_fr1_k launchable at this width); out[9] ... and the QKV projection runs in the same paired 8-channel form (2..8 finished rows: 2 = layers 1.. run it on rows_gemm_qp_k, 0 = on the 12-channel tiles); out[10] 2..8
 * finished rows: the FFN down-projection in the paired form (rows_gemm_frp_k); out[11] 17..64 rows: the layer runs on rows_gemm_wd_k
 * (1) or falls back to rows_gemm_mt_k (0); out[12] / out[13] its K slices for the out-projection / FFN down-projection; out[14] /
 * out[15] its k-tiles per wave for K = d / K = head_hidden.  tests/test_plan_cpu.py walks every model width with it. */
int vc_debug_plan(const vc_model_cfg* cfg, int compute_dtype, int rows, int32_t out[16]);
/* Copies a named internal device buffer to host memory. */
int vc_debug_read(vc_engine* e, const char* name, void* host_dst, int64_t nbytes);
/* Timing of the last vc_tts/vc_edit call, measured with HIP events on `stream`:
 * ms[0] = prompt build + prefill, ms[1] = decode loop, ms[2] = their sum.  The FIRST call of a shape / option state captures and
 * instantiates its decode graphs before either timer starts (host time, in vc_debug_read "host_ms" [1] / [2]): neither figure holds it. */
int vc_last_timing(const vc_engine* e, float ms[3]);
/* Average duration in ms of ONE kernel of the decode step - `which` = "qkv" | "attn" | "oproj" | "ffn1" (the FFN up-projection: since
 * round 5 the longest launch of a one-row layer, the kernel bench.py's roofline object quotes) | "ffn2", "<name>_hot" (the same layer
 * every launch: cache-resident), "pf_ffn1" / "pf_qkv" / "pf_attn" (prefill block GEMM / tile attention, FLOPs instead of bytes),
 * "wd_qkv" | "wd_oproj" | "wd_ffn1" | "wd_ffn2" | "wd_ln" | "wd_attn" (the launches of a WIDE decode step, n_rows in 17..64) or "step"
 * (a whole decode step of up to 16 rows without the sampler) - in the form a step of `n_rows` rows really launches, measured with HIP
 * events over `iters` back-to-back launches on `stream` (layers rotate: cold caches), and the algorithmic bytes (FLOPs) one launch moves. */
int vc_bench_kernel(vc_engine* e, const char* which, int n_rows, int iters, float* avg_ms,
                    double* alg_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VC_ENGINE_H */
