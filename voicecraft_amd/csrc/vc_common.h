// vc_common.h — internal types shared by the HIP translation units of libvcengine.so.
// gfx950 only (wave64, MFMA 16x16x32 bf16 / 16x16x4 f32).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vc_engine.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define VC_ROWS 16          // rows (token positions) of one MFMA tile = the N dimension of the rows-GEMM
#define VC_MAX_SEQS 64      // sequences one engine decodes together (passes of more than VC_ROWS rows run on the block GEMM)
#ifndef VC_MAX_ROWS
#define VC_MAX_ROWS 2048    // rows one forward pass may carry (prefill: eight 256-row tiles of the long-stream GEMM)
#endif
#define VC_SLAB_ROWS (VC_MAX_ROWS + 5)   // row stride of the split-K slabs: not a power of two, so the slabs of a row do not share a cache channel
#define VC_MAX_NSPLIT 8     // split-S factor cap of the decode attention (the out-projection prologue loads this many partials)
#define VC_MAX_KSPLIT 4     // cross-block split-K cap of the rows-GEMM (the LN prologue prefetches this many slabs)
#define VC_MAX_SEG 32       // prompt segments (2*spans+1 pieces + placeholders)
#ifndef VC_TH_QKV
#define VC_TH_QKV 12       // output channels per weight tile of the QKV projection (every other matrix: 16)
#endif
#define VC_VPL 34           // logits per lane in the sampler: V <= 64*34

// ---------------------------------------------------------------- element types
struct bf16_t { uint16_t u; };

__device__ __forceinline__ float bf16_to_f32(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// round-to-nearest-even (torch .to(bfloat16)): gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef __attribute__((ext_vector_type(2))) float vc_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 vc_bf16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const vc_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vc_bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename WT> struct WTr;
template <> struct WTr<float> {      // exact mode: v_mfma_f32_16x16x4_f32, 4 per 16-byte fragment
  static constexpr int EPL = 4;      // elements per lane per fragment
  static constexpr int KW = 16;      // K elements per fragment tile
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct WTr<bf16_t> {     // bf16 mode: v_mfma_f32_16x16x32_bf16, 8 per 16-byte fragment
  static constexpr int EPL = 8;
  static constexpr int KW = 32;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->u); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { p->u = f32_to_bf16(v); }
};

// D[n][m] += sum_k W[n][k] * X[m][k] for one 16-byte fragment pair.
// A operand (weights): lane l holds W[16*nt + (l&15)][KW*kt + EPL*(l>>4) + j]
// B operand (rows)   : lane l holds X[m = l&15   ][KW*kt + EPL*(l>>4) + j]
// D                  : lane l holds D[n = 4*(l>>4) + r][m = l&15], r = 0..3
__device__ __forceinline__ f32x4 mfma_frag(const uint4& w, const uint4& x, f32x4 acc, bf16_t*) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w),
                                                 __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_frag(const uint4& w, const uint4& x, f32x4 acc, float*) {
  // four K=4 steps; step j pairs element j of both fragments (a permutation of k shared by
  // both operands, so the dot product is unchanged).
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.x), __uint_as_float(x.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.y), __uint_as_float(x.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.z), __uint_as_float(x.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w.w), __uint_as_float(x.w), acc, 0, 0, 0);
  return acc;
}

// ---------------------------------------------------------------- wave helpers (wave64)
// Cross-lane reductions on the DPP path (no LDS crossbar): four steps reduce each 16-lane row
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row
// results are combined through v_readlane.  ~8 instructions against 6 dependent ds_bpermute.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
#define VC_DPP_QP_1032 0xB1
#define VC_DPP_QP_2301 0x4E
#define VC_DPP_ROW_HALF_MIRROR 0x141
#define VC_DPP_ROW_MIRROR 0x140

__device__ __forceinline__ float row_sum(float v) {        // every lane gets the sum of its 16-lane row
  v += dpp_f<VC_DPP_QP_1032>(v);
  v += dpp_f<VC_DPP_QP_2301>(v);
  v += dpp_f<VC_DPP_ROW_HALF_MIRROR>(v);
  v += dpp_f<VC_DPP_ROW_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {       // sum over aligned groups of 4 lanes
  v += dpp_f<VC_DPP_QP_1032>(v);
  v += dpp_f<VC_DPP_QP_2301>(v);
  return v;
}
__device__ __forceinline__ float half_row_sum(float v) {   // sum over aligned groups of 8 lanes
  v += dpp_f<VC_DPP_QP_1032>(v);
  v += dpp_f<VC_DPP_QP_2301>(v);
  v += dpp_f<VC_DPP_ROW_HALF_MIRROR>(v);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row_sum(v);
  return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) +
          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<VC_DPP_QP_1032>(v));
  v = fmaxf(v, dpp_f<VC_DPP_QP_2301>(v));
  v = fmaxf(v, dpp_f<VC_DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<VC_DPP_ROW_MIRROR>(v));
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)),
                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48))));
}
__device__ __forceinline__ int wave_sum_i(int v) {
  v += dpp_i<VC_DPP_QP_1032>(v);
  v += dpp_i<VC_DPP_QP_2301>(v);
  v += dpp_i<VC_DPP_ROW_HALF_MIRROR>(v);
  v += dpp_i<VC_DPP_ROW_MIRROR>(v);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, dpp_i<VC_DPP_QP_1032>(v));
  v = min(v, dpp_i<VC_DPP_QP_2301>(v));
  v = min(v, dpp_i<VC_DPP_ROW_HALF_MIRROR>(v));
  v = min(v, dpp_i<VC_DPP_ROW_MIRROR>(v));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---------------------------------------------------------------- in-kernel time stamps (diagnostic build)
// -DVC_KERNEL_TS: the first and the last workgroup of a launch record shader-clock stamps into
// a.dbg_ts[0..15] / [16..31] (tools/kernel_ts.py).  Compiled out of the product library.
#ifdef VC_KERNEL_TS
#define VC_KTS_DECL()                                                                           \
  long long kts_[16];                                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) kts_[i_] = 0;                               \
  const int kts_lin_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);          \
  const int kts_tot_ = gridDim.x * gridDim.y * gridDim.z;                                       \
  const int kts_slot_ = (kts_lin_ == 0) ? 0 : ((kts_lin_ == kts_tot_ - 1) ? 16 : -1);
#define VC_KTS(i)                                                                               \
  do { __builtin_amdgcn_sched_barrier(0); kts_[i] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VC_KTS_FLUSH()                                                                          \
  do {                                                                                          \
    if (a.dbg_ts && kts_slot_ >= 0 && threadIdx.x == 0) {                                       \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) a.dbg_ts[kts_slot_ + i_] = kts_[i_];    \
    }                                                                                           \
  } while (0)
#else
#define VC_KTS_DECL()
#define VC_KTS(i)
#define VC_KTS_FLUSH()
#endif

// ---------------------------------------------------------------- per-sequence decode state (device)
// Mirrors the Python locals of the reference's generation loop (models/voicecraft.py:994-1013,
// :1037-1067): codebook_eog -> n_eog (always a prefix), cur_num_gen, prev_token,
// consec_silence_count, y_input.shape[1] -> y_len, num_gen -> span_steps.
struct SeqState {
  int Lx;            // phoneme count (x_lens[0])
  int y_len;         // audio positions embedded so far
  int cur_num_gen;   // steps taken in the current span
  int n_eog;         // codebooks already terminated in the current span
  int prev_token;    // -1 = None
  int consec_silence;
  int span;          // index of the span being generated
  int n_spans;       // spans to generate (1 for TTS)
  int done;          // 1 once every span is finished (or the sample was dropped)
  int total_steps;   // rows written to the gen buffer
  int cap_len;       // y_len > cap_len forces termination (voicecraft.py:1042, :751)
  int min_gen;       // cur_num_gen <= min_gen suppresses the terminator on cb0 (TTS, :1024); -1 = off
  int term_token;    // eog_inference (TTS) or eog (editing)
  int kill_token;    // token whose logit is forced to -10000 on every codebook (:1091-1093, :816-818); -1 = none
  int group;         // best-of-N group id, -1 = independent
  int kept;          // 1 if this sample is the group's kept one (or independent)
  int span_steps[VC_MAX_SPANS];
  int mask_value[VC_MAX_SPANS];   // mask_embedding row inserted before span i (i >= 1)
  int slot;          // the sequence's index in the CALL (= its KV-cache slot, its rows of the gen / forced / logits_out buffers, its Philox
                     // stream); equal to its state's index until a wide batch is re-packed onto fewer rows (vc_tokens.hip repack_k)
};

// ---------------------------------------------------------------- kernel argument blocks
enum { PRO_LN = 0, PRO_PLAIN = 1, PRO_ATT = 2, PRO_LNW = 3, PRO_LNQ = 4 };      // PRO_LNW: LayerNorm fold of 2..8 FINISHED rows, one wave per row;
// PRO_LNQ (round 6): the same fold on the producers' CENTRED COPY of those rows in the compute dtype (GemmArgs.x_in = q = WT(h - c), c = row_mu[row]:
// the LayerNorm fold is exact for ANY centring constant, so the producer centres on the previous LayerNorm's mean of the row and the consumer
// reads half the bytes - no fp32 row, no conversion)
enum { EPI_QKV = 0, EPI_PART = 1, EPI_RELU = 2, EPI_GELU = 3, EPI_LOGITS = 4, EPI_RES = 5, EPI_QKV16 = 6 };   // EPI_RES: h_out = h_in + bias + W x (whole rows)
// EPI_QKV16: the QKV epilogue over the 16-channel image of the matrix (prefill and wide-decode passes, round 5): the 12-channel tiles of
// EPI_QKV leave a quarter of every MFMA's A lanes idle, which costs a many-row pass a quarter of its QKV launch
constexpr bool vc_is_qkv(int epi) { return epi == EPI_QKV || epi == EPI_QKV16; }
#define VC_TH_RES 8         // output channels per weight tile of the finished-row producers (rows_gemm_fr_k): d/8 workgroups
#define VC_FR_WAVES 8       // waves of a finished-row producer workgroup (each streams K/8 of its 8 channels in ONE burst)
#define VC_FR_MAX_ROWS 8    // finished-row passes of up to this many rows: one row per consumer wave, split attention merged by the
                            // out-projection, X of the FFN down-projection (rows x 4d elements) in LDS in one piece; 9..VC_ROWS rows: two
                            // rows per wave, unsplit attention, FFN down-projection through LDS in two halves (rows_gemm_fr2_k)

struct GemmArgs {
  // weights, packed [group][n_tile][k_tile][lane] x 16 bytes
  const uint4* Wp;
  const float* bias;        // [group][N] (may be null)
  int N, K;                 // logical out / in features per group
  int n_tiles, KT;          // ceil(N/16), K / KW
  int nchunk;               // k-chunks per block; a chunk = 4 waves * KTW tiles
  int r_lds;                // rows of X staged in LDS (<= VC_ROWS)
  int rows_cap;             // row stride of the split-K slabs: parts[s][rows_cap][N]
  int wd_stage;             // rows_gemm_wd_k launches: 1 = X through the wave-private LDS stage (rows_gemm_wds_k), 0 = fragments straight from L2 (option "wd_stage")
  int nt;                   // 1: stream the weights with non-temporal loads
  int mt;                   // 1: prefill pass (rows_gemm_blk_k): n_rows may reach VC_MAX_ROWS, plain prologue only; 2: wide decode pass
                            // (rows_gemm_mt_k); 3: PRO_LNW with two weight tiles per workgroup (rows_gemm_k<..., NTW = 2>)
  long w_group_stride;      // in uint4 units
  int bias_group_stride;
  // rows
  const int* row_seq;
  const int* row_pos;
  int n_rows;
  const int* n_active;      // never null: the launch is a no-op when *n_active == 0
  // prologue LN:   hn = h_in[src] + prev_bias + sum_s parts[s][r]; X = hn (un-normalised); h_out[r] = hn;
  //                the LayerNorm itself is folded into the weights and the epilogue (vc_gemm.hip, "LayerNorm fold")
  const float* h_in;
  float* h_out;
  const float* parts;       // [VC_MAX_KSPLIT][rows_cap][d], always readable; the first n_parts slabs are summed
  int n_parts;
  const float* prev_bias;   // always a readable [d] vector; added only when has_prev_bias
  int has_prev_bias;
  const float* wg;          // LN prologue: [group][N] row sums of the folded weights (W . gamma), see vc_gemm.hip
  const int* gather_rows;   // optional indirection on h_in/parts rows (logit rows for the heads)
  const float* row_mu;      // never null.  Finished-row producers: the constant the centred copy hq_out is centred on; PRO_LNQ consumers: the same constant
  float* row_mu_out;        // PRO_LNW / PRO_LNQ consumers: the row's mean (for the next producer's centred copy), written by the launch's first workgroup; may be null
  void* hq_out;             // finished-row producers: WT [rows][d] = h_out - row_mu[row] in the compute dtype (null = not written)
  int d;                    // row width of h / parts
  // prologue PLAIN: X = x_in[r][grp*x_group_stride + k]
  const void* x_in;
  int x_ld;
  int x_group_stride;
  int x_upr_shift;          // log2 of the 16-byte units per X row slice when that is a power of two, else -1 (set by the launcher)
  // prologue ATT: X = combine of attention split partials
  const float* att_o;       // [VC_ROWS][H][nsplit][hd] (fp32; bf16 when att_p16)
  int att_p16;              // rows_gemm_fr_k<PRO_ATT>, bf16 mode: the partials are bf16 (AttnArgs.part16 of the attention launch in front)
  const float* att_ml;      // [VC_ROWS][H][nsplit][2]
  int nsplit, H, hd;
  int hd_shift;             // log2(hd): head_dim is a power of two (vc_create), so head / element of a channel are a shift and a mask
  int att_q4_shift;         // log2 of (K slice / 4) when a power of two, else -1 (set by the launcher)
  // epilogues
  void* out;                // RELU/GELU: WT [r][out_ld]; LOGITS: float [r][group][N]
  int out_ld;
  int out_group_stride;
  float* q_out;             // QKV: [r][d]
  void* kcache;             // QKV: WT [seq][H][S_max][hd]
  void* vcache;
  long cache_seq_stride;    // H*S_max*hd
  int S_max;
  float* part_out;          // PART: [ksplit][rows_cap][N]
  void* x_out;              // ln_rows_k: normalised rows, WT [rows][d]
  long long* dbg_ts;        // shader-clock stamps (diagnostic builds with -DVC_KERNEL_TS only)
};

struct AttnArgs {
  const float* q;           // [VC_ROWS][d]
  const void* kcache;
  const void* vcache;
  long cache_seq_stride;
  int S_max, H, hd, d, nsplit;
  int hd_shift;             // log2(hd)
  float inv_nsplit;         // 1 / nsplit rounded up (set by the launcher)
  float scale;
  const int* row_seq;
  const int* row_pos;
  int n_rows;
  const int* n_active;      // never null
  const int* share_len;     // never null: cached positions [0, *share_len) of EVERY sequence are read from sequence 0's cache
                            // (text prefix shared by all utterances of a sentence-chained call, vc_tts_multi)
  float* att_o;
  float* att_ml;
  void* x_out;              // nsplit == 1 only: normalised output rows, WT [rows][d] (the out-projection then takes the plain prologue)
  long long* dbg_ts;        // diagnostic builds only
  int nt;                   // rows_attn_k: K/V rows requested with the non-temporal hint
  int fast;                 // rows_attn_k: 1 = the round-5 form (wave maximum before any exponential; bf16 mode: hardware exp2)
  int part16;               // rows_attn_k (bf16 mode, split passes): 1 = the partial acc[hd] is written as bf16 (read by rows_gemm_fr_k<.., P16>)
};

struct Segment {            // one run of columns of the rearranged audio sequence
  int col0, ncols;          // first column and number of columns it contributes
  int src0, src_len;        // source frames y[src0 .. src0+src_len)
  int term;                 // token appended after the source frames (eos/eog) or -1
  int mask_value;           // >= 0: this is a one-column mask placeholder using mask_embedding[mask_value]
};

struct PromptArgs {
  const int64_t* x;         // [Lx]
  const int64_t* y;         // [T][K]
  int Lx, T, K, d, V;
  int n_seg, n_cols;        // audio columns in the prefill
  Segment seg[VC_MAX_SEG];
  int empty_token;
  const float* text_emb;    // [text_rows][d]
  const float* audio_emb;   // [K][V][d]
  const float* mask_emb;    // [max_n_spans][d]
  const float* pe;          // [max_positions][d]
  float alpha_text, alpha_audio;
  int seq;                  // sequence slot
  int row0;                 // first row in emb/row tables this prompt occupies
  float* emb;               // [rows][d]
  int* row_seq;
  int* row_pos;
  int* err;                 // bit 0: an out-of-range token id was met; bit 1: the text differs from x_shared inside the shared prefix
  const int64_t* x_shared;  // skip > 0: sequence 0's text, whose first `skip` tokens this sequence claims to share (null = unchecked)
  int text_rows;
  int skip;                 // leading rows NOT emitted (their K/V are shared with sequence 0): the grid covers rows [skip, Lx + n_cols)
  int* logit_row;           // optional: *logit_row = logit_row_val (row of the last prefill group
  int logit_row_val;        //           whose hidden state feeds the heads)
};

// Per-call values of the sampler: they live in device memory (written once per call), so that the captured
// decode step does not depend on them and its hipGraphExec can be kept across calls (vc_engine.hip).
struct SampleDyn {
  int top_k;
  float top_p, temperature;
  int stop_repetition, n_silence;
  int silence[VC_MAX_SILENCE];
  int forced_mode;          // 0: the step's final tokens are replaced; 1: the raw draws are replaced (vc_sample_cfg.forced_mode)
  int n_forced;
  int logit_steps;
  int max_steps;            // rows of the gen buffer per sequence this call may fill
  int n_seq;                // sequences of the CALL: row stride of forced / logits_out (the step's row count may have shrunk since)
  uint64_t seed;
  const int64_t* forced;    // [n_forced][B][K] or null
  float* logits_out;        // [logit_steps][B][K][V] or null
  long long* dbg_ts;        // optional [16] shader-clock stamps of sequence 0 (diagnosis only)
};

struct SampleArgs {         // engine-constant part (kernel argument)
  const float* logits;      // [B][K][V]
  int B, K, V, d;
  int empty_token;
  int gen_stride;           // rows of the gen buffer per sequence
  const SampleDyn* dyn;
  SeqState* st;
  int* n_active;
  int* host_active;         // pinned host word (device-visible): set to 0 by the block that retires the last sequence
  int* host_live;           // pinned host word: sequences still live as the step's sampler launch found them (written by its first workgroup only:
                            // one writer, never out of order) - what the host's decode loop shrinks a wide batch by
  int* step_ctr;            // device word: decode steps sampled so far in this call (first workgroup only).  host_live has TWO slots and a step writes
  int graph_steps;          // slot (step_ctr / graph_steps) & 1: the host reads the slot of a batch of steps it has seen END, so what it re-packs by
                            // does not depend on how far the device has run ahead (same seed, same widths, same tokens in every run)
  int* samp;                // scratch [B][K + 2]
  int* gen;                 // [B][gen_stride][K]
  // next-step rows
  int rps;                  // rows per sequence slot (1, or 3 for editing)
  float* dec_h;             // [B*rps][d]
  int* row_seq;
  int* row_pos;
  int* logit_row;           // [B]
  const float* audio_emb;
  const float* mask_emb;
  const float* pe;
  float alpha_audio;
  int max_positions;
};

struct AssembleArgs {       // writes res [K][res_cap] from y and the generated spans
  const int64_t* y;         // [T][K]
  const int* gen;           // [max_steps][K] of the kept sequence
  int K, T, res_cap, n_piece;
  // piece i: kind 0 = copy y[src0..src0+len), kind 1 = un-shift gen rows [g0, g0+len+K)
  int kind[2 * VC_MAX_SPANS + 1];
  int src0[2 * VC_MAX_SPANS + 1];
  int len[2 * VC_MAX_SPANS + 1];
  int dst0[2 * VC_MAX_SPANS + 1];
  int64_t* res;
};

// ---------------------------------------------------------------- launchers (defined in the .hip files)
hipError_t vc_launch_pack(const float* src, const float* colscale, void* dst, int N, int K, int dtype, int th, hipStream_t s);
hipError_t vc_launch_fold_vecs(const float* W, const float* gamma, const float* beta, const float* bias, float* wg,
                               float* cb, int N, int K, int dtype, hipStream_t s);
hipError_t vc_launch_gemm(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, int groups,
                          hipStream_t s);
hipError_t vc_launch_gemm_blk(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, hipStream_t s);   // prefill passes (vc_gemm_pf.hip)
hipError_t vc_launch_gemm_wd(const GemmArgs& a, int dtype, int epi, int ksplit, int groups, hipStream_t s);  // wide decode steps, 17..64 rows (vc_gemm_wd.hip)
int vc_gemm_wd_kpw(int K, int dtype, int ksplit);   // k-tiles per wave of that kernel for a K slice, 0 = no form
size_t vc_gemm_lds_bytes(const GemmArgs& a, int dtype, int ksplit);
hipError_t vc_launch_gemm_fr(const GemmArgs& a, int dtype, int pro, hipStream_t s);   // finished-row producers (vc_gemm.hip)
size_t vc_gemm_fr_lds_bytes(int rows, int K, int dtype);
int vc_gemm_fr_form(int rows, int N, int K, int dtype, int pro, int nsplit);   // 0 none, 1 one piece, 2 K in two halves
hipError_t vc_launch_gemm_fr1(const GemmArgs& a, int dtype, int pro, int epi, hipStream_t s);   // one-row paired kernel (vc_gemm.hip row_gemm_fr1_k)
int vc_gemm_fr1_ok(int N, int K, int dtype, int nw);
hipError_t vc_launch_gemm_frp(const GemmArgs& a, int dtype, hipStream_t s);           // paired finished-row producer of 2..8-row passes (rows_gemm_frp_k)
int vc_gemm_frp_ok(int rows, int N, int K, int dtype);
hipError_t vc_launch_gemm_qp(const GemmArgs& a, int dtype, hipStream_t s);            // paired QKV consumer of 2..8 finished rows (rows_gemm_qp_k)
int vc_gemm_qp_ok(int rows, int N, int K, int dtype);
extern int vc_blk_dbg_mask;   // vc_gemm.hip: diagnostic mask of the prefill block GEMM, 0 in production
// Launch census (process-wide, host side): which kernel FORM each launcher picked.  Read through
// vc_debug_read("launch_counts") by the parity tests, which assert that the form a benchmarked shape runs on is the one
// they compared with the oracle.
enum { VC_LC_ROWS_GEMM = 0, VC_LC_MT2 = 1, VC_LC_MT4 = 2, VC_LC_BLK64 = 3, VC_LC_BLK128_SBS = 4, VC_LC_BLK128_2X2 = 5,
       VC_LC_BLK64_OCC2 = 6, VC_LC_LN_ROWS = 7, VC_LC_ROWS_ATTN = 8, VC_LC_TILE_ATTN = 9, VC_LC_ROWS_GEMM_FR = 10, VC_LC_BIG256 = 11, VC_LC_BIG128 = 12, VC_LC_ROW_GEMM_FR1 = 13, VC_LC_TILE_ATTN64 = 14, VC_LC_ROWS_GEMM_FRP = 15, VC_LC_WD = 16, VC_LC_ROWS_GEMM_QP = 17, VC_LC_N = 18 };
extern long long vc_launch_counts[VC_LC_N];
hipError_t vc_launch_ln_rows(const GemmArgs& a, int dtype, hipStream_t s);
hipError_t vc_launch_attn(const AttnArgs& a, int dtype, int rows_cap, hipStream_t s);
hipError_t vc_launch_tile_attn(const AttnArgs& a, int dtype, hipStream_t s);
hipError_t vc_launch_tile_attn64(const AttnArgs& a, hipStream_t s);   // bf16, head_dim 128, prompts on 64-row boundaries
hipError_t vc_launch_prompt(const PromptArgs& a, hipStream_t s);
hipError_t vc_launch_sample(const SampleArgs& a, bool grouped, hipStream_t s);
// Re-packs the LIVE sequences of a batch onto rows [0, n_live) of a narrower step (B_new <= B_old rows, n_live <= B_new; stable order):
// their states, next input rows and row tables move, retired sequences' final states go to st_fin[slot].  vc_tokens.hip repack_k.
struct RepackArgs {
  SeqState* st;             // [B_old] in, [B_new] out
  SeqState* st_fin;         // [max_seqs] final states by slot
  float* dec_h;             // [B_old][d]
  int* row_seq;
  int* row_pos;
  int* logit_row;
  int* err;                 // bit 2 raised when more than B_new sequences are still live (host logic error: nothing is moved)
  int B_old, B_new, d;
};
hipError_t vc_launch_repack(const RepackArgs& a, hipStream_t s);
hipError_t vc_launch_assemble(const AssembleArgs& a, hipStream_t s);

// ---- training objective, teacher-forced (vc_eval_forward; VoiceCraft.forward, models/voicecraft.py:472-559)
struct CeArgs {            // cross-entropy and top-10 membership of up to VC_ROWS logits rows against their targets
  const float* logits;     // [n_rows][K][V] raw head outputs
  const int64_t* y;        // the call's audio tokens, [frames][K] (all utterances back to back)
  const int* tgt;          // [n_rows][K]: >= 0 = index into y; -1 = no target here; <= -2 = the constant token -(v + 2)
  float* nll;              // [n_rows][K]: -log softmax(logits)[target]; 0 where there is no target
  int* hit;                // [n_rows][K]: 1 when fewer than 10 logits exceed the target's (torchmetrics top_k = 10)
  int n_rows, K, V;
  int* err;                // raised on a target id outside [0, V)
};
hipError_t vc_launch_ce(const CeArgs& a, hipStream_t s);
struct CeReduceArgs {      // per codebook: acc[k] += sum over rows, in a fixed order, in double
  const float* nll;
  const int* hit;
  const int* tgt;
  long n_rows;
  int K;
  double* nll_sum;         // [K]
  long long* hits;         // [K]
  long long* count;        // [K] number of targets
};
hipError_t vc_launch_ce_reduce(const CeReduceArgs& a, hipStream_t s);

hipError_t vc_launch_cast(const float* src, void* dst, long n, int dtype, hipStream_t s);
hipError_t vc_launch_copy_kv(void* cache, long seq_stride, int H, int S_max, int hd, int len,
                             int src_seq, int dst_seq0, int n_dst, int dtype, hipStream_t s);
