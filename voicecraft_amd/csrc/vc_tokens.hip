// vc_tokens.hip — the integer side of the path: delayed-codebook pattern kernels, prompt
// construction (+ embedding gather), the device-side sampler / end-of-generation state machine
// and the output assembly.  Everything here that produces token ids is bit-exact by contract.
#include <string.h>
#include <algorithm>
#include "vc_common.h"

// =============================================================== delayed pattern (bit-exact)
// Closed forms of Pattern.build_pattern_sequence / revert_pattern_sequence for the
// DelayedPatternProvider(delays = 0..K-1) layout (models/codebooks_patterns.py:117-176,:177-245,
// :336-352): sequence step s holds (t = s-1-q, q).  One thread per output element; a wave covers
// 64 consecutive steps of one codebook, so both the load and the store are coalesced.
__global__ void pattern_shift_k(const int64_t* __restrict__ z, int K, int T, int64_t special,
                                int64_t* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int S = T + K;
  const int s = (int)(idx % S);
  const long bq = idx / S;               // b*K + q
  const int q = (int)(bq % K);
  const int t = s - 1 - q;
  out[idx] = (t >= 0 && t < T) ? z[bq * T + t] : special;
}
__global__ void pattern_revert_k(const int64_t* __restrict__ sq, int K, int S, int T, int64_t special,
                                 int64_t* __restrict__ out, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int t = (int)(idx % T);
  const long bq = idx / T;
  const int q = (int)(bq % K);
  const int s = t + 1 + q;
  out[idx] = (s < S) ? sq[bq * S + s] : special;
}
// un-shift of a generated span (models/voicecraft.py:1125-1139): span is step-major [N][K];
// out[j][t] = span[j + t][j] for t in [0, N-K).
__global__ void pattern_unshift_k(const int64_t* __restrict__ span, int N, int K,
                                  int64_t* __restrict__ out) {
  const int W = N - K;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)K * W) return;
  const int j = (int)(idx / W), t = (int)(idx % W);
  out[idx] = span[(long)(j + t) * K + j];
}

extern "C" int vc_pattern_shift(const int64_t* z_dev, int B, int K, int T, int64_t special,
                                int64_t* out_dev, void* stream) {
  if ((!z_dev && T > 0) || !out_dev || B <= 0 || K <= 0 || T < 0) return VC_EINVAL;
  const long total = (long)B * K * (T + K);
  hipLaunchKernelGGL(pattern_shift_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, z_dev, K, T, special, out_dev, total);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_EHIP;
}
extern "C" int vc_pattern_revert(const int64_t* s_dev, int B, int K, int S, int T, int64_t special,
                                 int64_t* out_dev, void* stream) {
  if (B <= 0 || K <= 0 || S < 0 || T < 0 || S > T + K) return VC_EINVAL;
  const long total = (long)B * K * T;
  if (total == 0) return VC_OK;
  if (!s_dev || !out_dev) return VC_EINVAL;
  hipLaunchKernelGGL(pattern_revert_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, s_dev, K, S, T, special, out_dev, total);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_EHIP;
}
extern "C" int vc_pattern_unshift(const int64_t* span_dev, int N, int K, int64_t* out_dev,
                                  void* stream) {
  if (K <= 0 || N < K) return VC_EINVAL;
  const long total = (long)K * (N - K);
  if (total == 0) return VC_OK;                      // N == K: a span of only terminator steps holds no frame
  if (!span_dev || !out_dev) return VC_EINVAL;
  hipLaunchKernelGGL(pattern_unshift_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, span_dev, N, K, out_dev);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_EHIP;
}

// =============================================================== prompt build + embedding
// One block per prefill row.  Rows [0,Lx) are phonemes: text_embedding + alpha*pe (voicecraft.py:
// 950-951).  Rows [Lx, Lx+n_cols) are the rearranged, delay-shifted audio sequence of
// rearrange/shift/insert_mask/cat_y/embed_y (voicecraft.py:239-320; TTS: :957-985): column c falls
// in exactly one Segment; a shifted piece of content length n (source frames + optional terminator)
// contributes columns s = 0..ncols with token(q) = content[s-1-q] or `empty`; a mask placeholder
// column takes mask_embedding[mask_value] instead of the 4-table sum.  Audio positions restart at 0.
__global__ __launch_bounds__(256) void prompt_k(const PromptArgs a) {
  const int row = blockIdx.x;
  const int d = a.d;
  float* dst = a.emb + (long)(a.row0 + row) * d;
  if (threadIdx.x == 0) {
    a.row_seq[a.row0 + row] = a.seq;
    a.row_pos[a.row0 + row] = row;
    if (row == 0 && a.logit_row) *a.logit_row = a.logit_row_val;
  }
  if (row < a.Lx) {
    long tok = a.x[row];
    if (tok < 0 || tok >= a.text_rows) { if (threadIdx.x == 0) *a.err = 1; tok = 0; }
    const float* e = a.text_emb + tok * d;
    const float* pe = a.pe + (long)row * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = e[c] + a.alpha_text * pe[c];
    return;
  }
  const int col = row - a.Lx;
  int si = 0;
  for (int i = 0; i < a.n_seg; ++i)
    if (col >= a.seg[i].col0 && col < a.seg[i].col0 + a.seg[i].ncols) si = i;
  const Segment sg = a.seg[si];
  const float* pe = a.pe + (long)col * d;
  if (sg.mask_value >= 0) {
    const float* e = a.mask_emb + (long)sg.mask_value * d;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = e[c] + a.alpha_audio * pe[c];
    return;
  }
  const int s = col - sg.col0;
  const int n = sg.src_len + (sg.term >= 0 ? 1 : 0);
  const float* e[VC_MAX_CODEBOOKS];
  for (int q = 0; q < a.K; ++q) {
    const int i = s - 1 - q;
    long tok = a.empty_token;
    if (i >= 0 && i < n) tok = (i < sg.src_len) ? a.y[(long)(sg.src0 + i) * a.K + q] : sg.term;
    if (tok < 0 || tok >= a.V) { if (threadIdx.x == 0) *a.err = 1; tok = a.empty_token; }
    e[q] = a.audio_emb + ((long)q * a.V + tok) * d;
  }
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float v = e[0][c];
    for (int q = 1; q < a.K; ++q) v += e[q][c];       // stack(...).sum(dim=0): k = 0..K-1 in order
    dst[c] = v + a.alpha_audio * pe[c];
  }
}
hipError_t vc_launch_prompt(const PromptArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(prompt_k, dim3(a.Lx + a.n_cols), dim3(256), 0, s, a);
  return hipGetLastError();
}

// =============================================================== sampler + EOG state machine
// Philox4x32-10 (Salmon et al. 2011), one uniform per (sequence, step, codebook).
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t seq, uint32_t step, uint32_t cb) {
  uint32_t c0 = step, c1 = seq, c2 = cb, c3 = 0x5643u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);   // [0, 1)
}

__device__ __forceinline__ uint32_t fkey(float f) {   // order-preserving float -> uint
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ bool is_silence(const SampleArgs& a, int tok) {
  bool r = false;
#pragma unroll
  for (int i = 0; i < VC_MAX_SILENCE; ++i) r |= (i < a.n_silence && a.silence[i] == tok);   // static indices: no scratch copy
  return r;
}

// Phase 1 (one block per sequence, one wave per codebook): logit edits, top-k / top-p filter,
// categorical draw.  Restates sample_helper + topk_sampling + top_k_top_p_filtering
// (models/voicecraft.py:1018-1067, :71-86, :26-68).
// The row is held in registers (34 per lane); LDS is only the scratchpad where lane 0 applies the
// reference's point edits.  Measured alternatives (profiles/r01_sampler_notes.md): an all-LDS version
// with rolled loops spends 34 us in the 32-step threshold search alone (LDS latency per element).
// `sp` is the sequence state (in LDS); results go to xs (LDS in the fused kernel, HBM scratch when the
// keep decision needs the kernel boundary): xs[0..K) tokens, xs[K] arg-max of codebook 0, xs[K+1] cond.
#define VC_TS(i) do { if (a.dbg_ts && b == 0 && threadIdx.x == 0) a.dbg_ts[i] = clock64(); } while (0)
// The logits row of codebook k = wave does not depend on the sequence state: the kernels request it
// before anything else (v0), so that the state's round trip and the row's overlap.
__device__ __forceinline__ void preload_row(const SampleArgs& a, int b, float (&v0)[VC_VPL]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* row = a.logits + ((long)b * a.K + min(wave, a.K - 1)) * a.V;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) v0[j] = row[min(lane + 64 * j, a.V - 1)];   // all loads in flight together
}
// temperature -> top-k -> softmax -> top-p -> inverse-CDF draw for the row a wave holds in registers
// (v: edited logits, padding -inf; bv: their maximum; u: the uniform of this draw).  Wave-uniform result.
__device__ __forceinline__ int filter_draw(const SampleArgs& a, int b, float (&v)[VC_VPL], float bv, int V, float u) {
  const int lane = threadIdx.x & 63;
  // ---- temperature
  float mx = bv;
  if (a.temperature != 1.0f) {
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) v[j] = v[j] / a.temperature;
    mx = bv / a.temperature;                      // filters never remove the maximum
  }
  // ---- top-k: keep everything >= the k-th largest value (ties at the threshold survive).
  // Bitwise binary search for the k-th largest order-preserving key, on registers.
  if (a.top_k > 0) {
    const int kk = min(max(a.top_k, 1), V);
    uint32_t key[VC_VPL];
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) key[j] = fkey(v[j]);       // padding: key(-inf) = 0x007fffff, below every finite key
    uint32_t t = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = t | (1u << bit);
      int c = 0;
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) c += (key[j] >= cand) ? 1 : 0;
      const int cs = wave_sum_i(c);
      if (cs >= kk) {
        t = cand;
        if (cs == kk) break;        // exactly the kk largest keys are >= t already: the remaining bits cannot change the kept set
      }
    }
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) v[j] = (key[j] < t) ? -INFINITY : v[j];
  }
  VC_TS(3);
  // ---- softmax numerators (v now holds p >= 0; 0 = filtered out)
  float ps = 0.f;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) { v[j] = __expf(v[j] - mx); ps += v[j]; }        // exp(-inf) = 0
  float tot = wave_sum(ps);
  // ---- top-p: drop a token when the mass of the strictly larger ones already exceeds top_p.
  // p is monotone in the logit, and the bits of a non-negative float order like the float.
  if (a.top_p < 1.0f) {
    const float lim = a.top_p * tot;
    // t = the largest key whose strictly-larger mass still exceeds lim (key 0 always qualifies:
    // the whole row weighs tot > lim); exactly the keys <= t are dropped.
    uint32_t t = 0;
#pragma unroll 1
    for (int bit = 30; bit >= 0; --bit) {
      const uint32_t cand = t | (1u << bit);
      float mm = 0.f;
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) mm += (__float_as_uint(v[j]) > cand) ? v[j] : 0.f;
      if (wave_sum(mm) > lim) t = cand;
    }
    ps = 0.f;
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) { v[j] = (__float_as_uint(v[j]) <= t) ? 0.f : v[j]; ps += v[j]; }
    tot = wave_sum(ps);
  }
  VC_TS(4);
  // ---- categorical draw by inverse CDF, order = (lane, j)
  const float target = u * tot;
  float incl = ps;
  for (int off = 1; off < 64; off <<= 1) {
    const float o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  const float excl = incl - ps;
  const bool mine = (ps > 0.f) && (target >= excl) && (target < incl);
  const uint64_t ball = __ballot(mine);
  int src_lane;
  if (ball) src_lane = __ffsll((long long)ball) - 1;
  else {     // rounding put target at/after the end: take the last lane with mass
    const uint64_t nz = __ballot(ps > 0.f);
    src_lane = nz ? 63 - __clzll((long long)nz) : 0;
  }
  // every lane walks its own elements (cheap, branch-free); the owner's result is broadcast
  float acc = excl;
  int pick = -1, last = -1;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) {
    const bool nz = v[j] > 0.f;
    acc += v[j];
    last = nz ? lane + 64 * j : last;
    pick = (pick < 0 && nz && target < acc) ? lane + 64 * j : pick;
  }
  int tok = (pick >= 0) ? pick : last;
  tok = __shfl(tok, src_lane, 64);
  return tok;
}

__device__ __forceinline__ void sample_phase(const SampleArgs& a, int b, const SeqState* sp, int* xs, float* s_rows,
                                             const float (&v0)[VC_VPL]) {
  const SeqState st = *sp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (st.done) return;
  const int step = st.total_steps;
  const int V = a.V;
  const int VP = ((V + 63) >> 6) << 6;
  for (int k = wave; k < a.K; k += 4) {
    const float* row = a.logits + ((long)b * a.K + k) * V;
    float* sv = s_rows + k * VP;
    float v[VC_VPL];
    if (k == wave) {
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) v[j] = v0[j];
    } else {                                                                     // K > 4: later rows of this wave
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) v[j] = row[min(lane + 64 * j, V - 1)];
    }
    if (a.logits_out && step < a.logit_steps) {
      float* lo = a.logits_out + (((long)step * a.B + b) * a.K + k) * V;
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) if (lane + 64 * j < V) lo[lane + 64 * j] = v[j];
    }
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) if (lane + 64 * j < VP) sv[lane + 64 * j] = v[j];
    VC_TS(1);
    // ---- logit edits, in the reference's order; each touches one element, so lane 0 does them in LDS
    if (lane == 0) {
      const int term = st.term_token;
      if (st.kill_token >= 0) sv[st.kill_token] = -10000.f;
      const bool kill_tl = (st.n_eog == 0) ? (k >= 1) : (k > st.n_eog);      // [term],[empty] on later codebooks
      if (kill_tl) { sv[term] = -10000.f; sv[a.empty_token] = -10000.f; }
      if (st.n_eog == 0 && k == 0) {
        if (st.min_gen >= 0 && st.cur_num_gen <= st.min_gen) sv[term] = -10000.f;
        if (a.stop_repetition > 0 && st.prev_token >= 0 && is_silence(a, st.prev_token) &&
            st.consec_silence > a.stop_repetition) {
          const float f = (float)(st.consec_silence - (a.stop_repetition - 1));
          const float x = sv[st.prev_token];
          sv[st.prev_token] = (x < 0.f) ? x * f : x / f;
        }
      }
    }
    // one wave: its LDS accesses are ordered, no barrier needed
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) {
      const int i = lane + 64 * j;
      const float t = sv[min(i, VP - 1)];
      v[j] = (i < V) ? t : -INFINITY;
    }
    // ---- arg-max of the edited logits (first index on ties, as torch.argmax)
    float bv = -INFINITY;
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) bv = fmaxf(bv, v[j]);
    bv = wave_max(bv);
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = VC_VPL - 1; j >= 0; --j) bi = (v[j] == bv) ? lane + 64 * j : bi;     // padding is -inf, never equal
    bi = wave_min_i(bi);
    VC_TS(2);
    const float u = philox_uniform(a.seed, (uint32_t)b, (uint32_t)step, (uint32_t)k);
    int tok = filter_draw(a, b, v, bv, V, u);
    if (a.forced && a.forced_mode == 1 && step < a.n_forced)      // replay of recorded reference draws (parity tests)
      tok = (int)a.forced[((long)step * a.B + b) * a.K + k];
    if (lane == 0) {
      xs[k] = tok;
      if (k == 0) xs[a.K] = bi;
    }
  }
  VC_TS(5);
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    if (st.n_eog == 0) c = (xs[0] == st.term_token) || (xs[a.K] == st.term_token) || (st.y_len > st.cap_len);
    xs[a.K + 1] = c;
  }
}

// Phase 2 (one block per sequence): advance the state machine, log the step's tokens and build
// the next decode rows (embedding sum + position, voicecraft.py:1102-1116; span switch :838-858).
__device__ void advance_phase(const SampleArgs& a, int b, bool grouped, SeqState* sp, const int* xs) {
  __shared__ int s_tok[VC_MAX_CODEBOOKS];
  __shared__ int s_mode;     // 0: row inactive, 1: one new row, 3: span switch (three rows)
  __shared__ int s_ylen, s_mask, s_Lx;
  const int tid = threadIdx.x;
  const int K = a.K;
  if (tid == 0) {
    SeqState& st = *sp;                 // lives in LDS: the span arrays are indexed dynamically
    s_mode = 0;
    if (!st.done) {
      int tok[VC_MAX_CODEBOOKS];
#pragma unroll
      for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) tok[k] = (k < K) ? xs[k] : 0;
      int cond = xs[K + 1];
      bool drop = false;
      if (grouped && st.n_eog == 0) {
        // best-of-N (voicecraft.py:1296-1302): the LAST sample whose first codebook terminates is kept
        int keep = -1;
        for (int bb = 0; bb < a.B; ++bb)
          if (a.st[bb].group == st.group && a.samp[bb * (VC_MAX_CODEBOOKS + 2) + K + 1]) keep = bb;
        // (group ids are immutable and every member is still alive while n_eog == 0)
        if (keep >= 0 && keep != b) drop = true;
      }
      const int step = st.total_steps;
      const bool forced = a.forced && a.forced_mode == 0 && step < a.n_forced;
      if (st.n_eog == 0) {
        if (st.cur_num_gen < K - 1)
          for (int jj = 1; jj < K - st.cur_num_gen; ++jj) tok[K - jj] = a.empty_token;
        if (forced) {
          for (int k = 0; k < K; ++k) tok[k] = (int)a.forced[((long)step * a.B + b) * K + k];
          cond = (tok[0] == st.term_token);
        }
        if (cond) { tok[0] = st.term_token; st.n_eog = 1; }
        if (is_silence(a, tok[0]) && tok[0] == st.prev_token) st.consec_silence += 1;
        else st.consec_silence = 0;
        st.prev_token = tok[0];
      } else {
        for (int k = 0; k < st.n_eog; ++k) tok[k] = a.empty_token;
        tok[st.n_eog] = st.term_token;
        if (forced)
          for (int k = 0; k < K; ++k) tok[k] = (int)a.forced[((long)step * a.B + b) * K + k];
        st.n_eog += 1;
      }
      st.cur_num_gen += 1;
      if (step < a.max_steps)
        for (int k = 0; k < K; ++k) a.gen[((long)b * a.max_steps + step) * K + k] = tok[k];
      st.total_steps = step + 1;
      for (int k = 0; k < K; ++k) s_tok[k] = tok[k];
      s_ylen = st.y_len;
      s_Lx = st.Lx;
      if (drop) {
        st.done = 1; st.kept = 0;
        atomicSub(a.n_active, 1);
      } else if (st.n_eog == K) {            // span finished
        st.span_steps[st.span] = st.cur_num_gen;
        st.cur_num_gen = 0;
        st.n_eog = 0;
        st.span += 1;
        if (st.span >= st.n_spans || st.total_steps >= a.max_steps ||
            st.Lx + st.y_len + 3 > a.max_positions) {
          st.done = 1;
          atomicSub(a.n_active, 1);
        } else {
          s_mode = 3;
          s_mask = st.mask_value[st.span];
          st.y_len += 3;
          st.prev_token = -1;
          st.consec_silence = 0;
        }
      } else if (st.total_steps >= a.max_steps ||
                 st.Lx + st.y_len + 1 > a.max_positions) {   // capacity guard (never hit when sized right)
        st.done = 1;
        atomicSub(a.n_active, 1);
      } else {
        s_mode = 1;
        st.y_len += 1;
      }
    }
    const int r0 = b * a.rps;
    for (int i = 0; i < a.rps; ++i) {
      a.row_seq[r0 + i] = b;
      a.row_pos[r0 + i] = (i < s_mode) ? (s_Lx + s_ylen + i) : -1;
    }
    a.logit_row[b] = r0 + (s_mode == 3 ? 2 : 0);
  }
  __syncthreads();
  const int mode = s_mode;
  VC_TS(7);
  if (mode == 0) return;
  // next-step input row(s): sum of the K codebook embeddings of the emitted tokens + alpha*pe
  // (voicecraft.py:1102-1116).  All K gathers of a thread are requested together (float4 columns).
  const int d = a.d;
  const int ylen = s_ylen;
  float* h0 = a.dec_h + (long)(b * a.rps) * d;
  const float* pe0 = a.pe + (long)ylen * d;
  const int nq = d >> 2;
  for (int i = tid; i < nq; i += blockDim.x) {
    const float4 p = *reinterpret_cast<const float4*>(pe0 + i * 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 4) {            // four gathers in flight per pass; summed in order k = 0..K-1
      float4 e[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = (k0 + q < K) ? k0 + q : 0;
        e[q] = *reinterpret_cast<const float4*>(a.audio_emb + ((long)kk * a.V + s_tok[kk]) * d + i * 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k0 + q < K) {
          if (k0 + q == 0) v = e[q];
          else { v.x += e[q].x; v.y += e[q].y; v.z += e[q].z; v.w += e[q].w; }
        }
    }
    v.x += a.alpha_audio * p.x; v.y += a.alpha_audio * p.y; v.z += a.alpha_audio * p.z; v.w += a.alpha_audio * p.w;
    *reinterpret_cast<float4*>(h0 + i * 4) = v;
  }
  if (mode == 3) {   // span switch: [last token, mask_embedding[next span], all-empty column] (voicecraft.py:838-858)
    for (int i = tid; i < nq; i += blockDim.x) {
      const float4 mk = *reinterpret_cast<const float4*>(a.mask_emb + (long)s_mask * d + i * 4);
      const float4 p1 = *reinterpret_cast<const float4*>(pe0 + d + i * 4);
      const float4 p2 = *reinterpret_cast<const float4*>(pe0 + 2 * d + i * 4);
      float4 v = *reinterpret_cast<const float4*>(a.audio_emb + ((long)a.empty_token) * d + i * 4);
#pragma unroll 1
      for (int k = 1; k < K; ++k) {
        const float4 e = *reinterpret_cast<const float4*>(a.audio_emb + ((long)k * a.V + a.empty_token) * d + i * 4);
        v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
      }
      float4 o1, o2;
      o1.x = mk.x + a.alpha_audio * p1.x; o1.y = mk.y + a.alpha_audio * p1.y; o1.z = mk.z + a.alpha_audio * p1.z; o1.w = mk.w + a.alpha_audio * p1.w;
      o2.x = v.x + a.alpha_audio * p2.x; o2.y = v.y + a.alpha_audio * p2.y; o2.z = v.z + a.alpha_audio * p2.z; o2.w = v.w + a.alpha_audio * p2.w;
      *reinterpret_cast<float4*>(h0 + d + i * 4) = o1;
      *reinterpret_cast<float4*>(h0 + 2 * d + i * 4) = o2;
    }
  }
}

// The state words are requested (fetch_state) together with the "still active" word and the logits
// row, i.e. in ONE round trip, and only then parked in LDS (park_state).
__device__ __forceinline__ int fetch_state(const SampleArgs& a, int b) {
  constexpr int W = sizeof(SeqState) / 4;
  return reinterpret_cast<const int*>(a.st + b)[min((int)threadIdx.x, W - 1)];
}
__device__ __forceinline__ void park_state(SeqState* sp, int word) {
  constexpr int W = sizeof(SeqState) / 4;
  if (threadIdx.x < W) reinterpret_cast<int*>(sp)[threadIdx.x] = word;
  __syncthreads();
}
__device__ __forceinline__ void store_state(const SampleArgs& a, int b, const SeqState* sp) {
  constexpr int W = sizeof(SeqState) / 4;
  __syncthreads();
  if (threadIdx.x < W) reinterpret_cast<int*>(a.st + b)[threadIdx.x] = reinterpret_cast<const int*>(sp)[threadIdx.x];
}
extern __shared__ __attribute__((aligned(16))) float s_dyn[];   // K rows of 64*ceil(V/64) logits
__global__ __launch_bounds__(256) void sample_fused_k(const SampleArgs a) {
  __shared__ SeqState s_st;
  __shared__ int s_xs[VC_MAX_CODEBOOKS + 2];
  const int b = blockIdx.x;
  VC_TS(0);
  float v0[VC_VPL];
  preload_row(a, blockIdx.x, v0);
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  park_state(&s_st, sw);
  sample_phase(a, blockIdx.x, &s_st, s_xs, s_dyn, v0);
  __syncthreads();
  VC_TS(6);
  advance_phase(a, blockIdx.x, false, &s_st, s_xs);
  VC_TS(8);
  store_state(a, blockIdx.x, &s_st);
  VC_TS(9);
}
__global__ __launch_bounds__(256) void sample_only_k(const SampleArgs a) {
  __shared__ SeqState s_st;
  float v0[VC_VPL];
  preload_row(a, blockIdx.x, v0);
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  park_state(&s_st, sw);
  sample_phase(a, blockIdx.x, &s_st, a.samp + blockIdx.x * (VC_MAX_CODEBOOKS + 2), s_dyn, v0);
}
__global__ __launch_bounds__(256) void advance_only_k(const SampleArgs a) {
  __shared__ SeqState s_st;
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  park_state(&s_st, sw);
  advance_phase(a, blockIdx.x, true, &s_st, a.samp + blockIdx.x * (VC_MAX_CODEBOOKS + 2));
  store_state(a, blockIdx.x, &s_st);
}
hipError_t vc_launch_sample(const SampleArgs& a, bool grouped, hipStream_t s) {
  const size_t lds = (size_t)a.K * (((a.V + 63) >> 6) << 6) * sizeof(float);
  if (!grouped) {
    hipLaunchKernelGGL(sample_fused_k, dim3(a.B), dim3(256), lds, s, a);
  } else {
    // the keep decision reads every sample's cond flag, so it needs the kernel boundary
    hipLaunchKernelGGL(sample_only_k, dim3(a.B), dim3(256), lds, s, a);
    hipLaunchKernelGGL(advance_only_k, dim3(a.B), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// =============================================================== sampler test hook
// n_draws independent draws from ONE logits row through the product filter_draw (the distribution test
// of tests/test_gpu_sampler.py): draw i uses the Philox counter (seed, sequence i, step 0, codebook 0).
__global__ __launch_bounds__(256) void sample_test_k(const SampleArgs a, int n_draws, int* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n_draws) return;
  float v[VC_VPL];
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) {
    const int c = lane + 64 * j;
    const float t = a.logits[min(c, a.V - 1)];
    v[j] = (c < a.V) ? t : -INFINITY;
  }
  float bv = -INFINITY;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) bv = fmaxf(bv, v[j]);
  bv = wave_max(bv);
  const float u = philox_uniform(a.seed, (uint32_t)i, 0u, 0u);
  const int tok = filter_draw(a, 0, v, bv, a.V, u);
  if (lane == 0) out[i] = tok;
}
extern "C" int vc_debug_sample(const float* logits_dev, int V, const vc_sample_cfg* sc, int n_draws,
                               int32_t* out_dev, void* stream) {
  if (!logits_dev || !sc || !out_dev || V < 1 || V > 64 * VC_VPL || n_draws < 1) return VC_EINVAL;
  SampleArgs a;
  memset(&a, 0, sizeof a);
  a.logits = logits_dev; a.B = 1; a.K = 1; a.V = V;
  a.top_k = sc->top_k; a.top_p = sc->top_p; a.temperature = sc->temperature; a.seed = sc->seed;
  hipLaunchKernelGGL(sample_test_k, dim3((n_draws + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, n_draws, out_dev);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_EHIP;
}

// =============================================================== output assembly
// res = cat(non-mask pieces of y, un-shifted generated spans) (voicecraft.py:1141-1153, :890-898).
__global__ void assemble_k(const AssembleArgs a) {
  const int p = blockIdx.y;
  const int len = a.len[p];
  const long total = (long)a.K * len;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx / len), t = (int)(idx % len);
    int64_t v;
    if (a.kind[p] == 0) v = a.y[(long)(a.src0[p] + t) * a.K + j];
    else v = (int64_t)a.gen[(long)(a.src0[p] + j + t) * a.K + j];
    a.res[(long)j * a.res_cap + a.dst0[p] + t] = v;
  }
}
hipError_t vc_launch_assemble(const AssembleArgs& a, hipStream_t s) {
  if (a.n_piece <= 0) return hipSuccess;
  hipLaunchKernelGGL(assemble_k, dim3(8, a.n_piece), dim3(256), 0, s, a);
  return hipGetLastError();
}
