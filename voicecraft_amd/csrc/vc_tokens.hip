// vc_tokens.hip — the integer side of the path: delayed-codebook pattern kernels, prompt
// construction (+ embedding gather), the device-side sampler / end-of-generation state machine
// and the output assembly.  Everything here that produces token ids is bit-exact by contract.
#include <stddef.h>
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
  const int row = blockIdx.x + a.skip;        // row of the sequence = its cache position
  const int d = a.d;
  float* dst = a.emb + (long)(a.row0 + blockIdx.x) * d;
  if (threadIdx.x == 0) {
    a.row_seq[a.row0 + blockIdx.x] = a.seq;
    a.row_pos[a.row0 + blockIdx.x] = row;
    if (blockIdx.x == 0 && a.logit_row) *a.logit_row = a.logit_row_val;
  }
  // rows [0, skip) are not emitted: their K/V are read from sequence 0's cache, so the texts must agree there
  if (blockIdx.x == 0 && a.skip > 0 && a.x_shared) {
    bool bad = false;
    for (int i = threadIdx.x; i < a.skip; i += blockDim.x) bad |= (a.x[i] != a.x_shared[i]);
    if (bad) atomicOr(a.err, 2);
  }
  if (row < a.Lx) {
    long tok = a.x[row];
    if (tok < 0 || tok >= a.text_rows) { if (threadIdx.x == 0) atomicOr(a.err, 1); tok = 0; }
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
    if (tok < 0 || tok >= a.V) { if (threadIdx.x == 0) atomicOr(a.err, 1); tok = a.empty_token; }
    e[q] = a.audio_emb + ((long)q * a.V + tok) * d;
  }
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float v = e[0][c];
    for (int q = 1; q < a.K; ++q) v += e[q][c];       // stack(...).sum(dim=0): k = 0..K-1 in order
    dst[c] = v + a.alpha_audio * pe[c];
  }
}
hipError_t vc_launch_prompt(const PromptArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(prompt_k, dim3(a.Lx + a.n_cols - a.skip), dim3(256), 0, s, a);
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
__device__ __forceinline__ bool is_silence(const SampleDyn& dy, int tok) {
  bool r = false;
#pragma unroll
  for (int i = 0; i < VC_MAX_SILENCE; ++i) r |= (i < dy.n_silence && dy.silence[i] == tok);   // static indices: no scratch copy
  return r;
}

// inclusive prefix sum over the 64 lanes on the DPP path: Hillis-Steele inside each 16-lane row (row_shr
// 1, 2, 4, 8; lanes shifted in from outside the row read 0), then the three row totals through v_readlane.
__device__ __forceinline__ float wave_scan_incl(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));
  const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
  const float t1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float t2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
  const int row = (threadIdx.x & 63) >> 4;
  const float add = (row >= 1 ? t0 : 0.f) + (row >= 2 ? t1 : 0.f) + (row >= 3 ? t2 : 0.f);
  return v + add;
}

// ---- k-th largest key of the row a wave holds in registers (kk >= 1), exact.
// Bitwise binary search over all VC_VPL keys per lane: 32 steps of VC_VPL compares + a wave sum.
__device__ __forceinline__ uint32_t kth_largest_full(const uint32_t (&key)[VC_VPL], int kk) {
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
  return t;
}
// Fast path for kk <= 64 (the path's top_k is 40): every lane keeps the 5 largest of its keys (a branch-free
// max/min insertion), the search then runs on those 320 candidates with wave ballots (5 compares per step
// instead of 34), and ONE count over the whole row proves the result: if the row holds exactly as many keys
// >= t as the candidate set, no key outside the set reaches t, so t is the row's kk-th largest.  Otherwise
// (a lane owns more than 5 of the kk largest: ~0.3 % of random rows at kk = 40) the full search runs.
__device__ __forceinline__ uint32_t kth_largest(const uint32_t (&key)[VC_VPL], int kk) {
  if (kk > 64) return kth_largest_full(key, kk);
  uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) {
    uint32_t x = key[j], t_;
    t_ = max(m0, x); x = min(m0, x); m0 = t_;
    t_ = max(m1, x); x = min(m1, x); m1 = t_;
    t_ = max(m2, x); x = min(m2, x); m2 = t_;
    t_ = max(m3, x); x = min(m3, x); m3 = t_;
    m4 = max(m4, x);
  }
  uint32_t t = 0;
  int cu = 0;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = t | (1u << bit);
    const int c = __popcll(__ballot(m0 >= cand)) + __popcll(__ballot(m1 >= cand)) + __popcll(__ballot(m2 >= cand)) +
                  __popcll(__ballot(m3 >= cand)) + __popcll(__ballot(m4 >= cand));
    if (c >= kk) {
      t = cand; cu = c;
      if (c == kk) break;
    }
  }
  int ca = 0;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) ca += (key[j] >= t) ? 1 : 0;
  ca = wave_sum_i(ca);
  if (ca != cu) t = kth_largest_full(key, kk);
  return t;
}

// Phase 1 (one block per sequence, one wave per codebook): logit edits, top-k / top-p filter,
// categorical draw.  Restates sample_helper + topk_sampling + top_k_top_p_filtering
// (models/voicecraft.py:1018-1067, :71-86, :26-68).
// The row is held in registers (34 per lane); LDS is only the scratchpad where lane 0 applies the
// reference's point edits.  Measured alternatives (profiles/r01_sampler_notes.md): an all-LDS version
// with rolled loops spends 34 us in the 32-step threshold search alone (LDS latency per element).
// `sp` is the sequence state (in LDS); results go to xs (LDS in the fused kernel, HBM scratch when the
// keep decision needs the kernel boundary): xs[0..K) tokens, xs[K] arg-max of codebook 0, xs[K+1] cond.
#define VC_TS(i) do { if (dy.dbg_ts && b == 0 && threadIdx.x == 0) dy.dbg_ts[i] = clock64(); } while (0)
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
__device__ __forceinline__ int filter_draw(int top_k, float top_p, float temperature, float (&v)[VC_VPL], float bv,
                                           int V, float u) {
  const int lane = threadIdx.x & 63;
  // ---- temperature
  float mx = bv;
  if (temperature != 1.0f) {
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) v[j] = v[j] / temperature;
    mx = bv / temperature;                      // filters never remove the maximum
  }
  // ---- top-k: keep everything >= the k-th largest value (ties at the threshold survive, as
  // `logits < topk(...)[-1]` in the reference, voicecraft.py:38-44)
  if (top_k > 0) {
    const int kk = min(max(top_k, 1), V);
    uint32_t key[VC_VPL];
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) key[j] = fkey(v[j]);       // padding: key(-inf) = 0x007fffff, below every finite key
    const uint32_t t = kth_largest(key, kk);
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) v[j] = (key[j] < t) ? -INFINITY : v[j];
  }
  // ---- softmax numerators (v now holds p >= 0; 0 = filtered out)
  float ps = 0.f;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) { v[j] = __expf(v[j] - mx); ps += v[j]; }        // exp(-inf) = 0
  float tot = wave_sum(ps);
  // ---- top-p: drop a token when the mass of the strictly larger ones already exceeds top_p.
  // p is monotone in the logit, and the bits of a non-negative float order like the float.
  if (top_p < 1.0f) {
    const float lim = top_p * tot;
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
  // ---- categorical draw by inverse CDF, order = (lane, j)
  const float target = u * tot;
  const float incl = wave_scan_incl(ps);
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
  const int tok = (pick >= 0) ? pick : last;
  return __builtin_amdgcn_readlane(tok, __builtin_amdgcn_readfirstlane(src_lane));
}

__device__ __forceinline__ void sample_phase(const SampleArgs& a, const SampleDyn& dy, int b, const SeqState* sp, int* xs,
                                             float* s_rows, const float (&v0)[VC_VPL]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the fields this phase reads, fetched from LDS in one go
  const int st_done = sp->done, step = sp->total_steps, term = sp->term_token, kill = sp->kill_token, n_eog = sp->n_eog;
  const int min_gen = sp->min_gen, cur_num_gen = sp->cur_num_gen, prev_token = sp->prev_token, consec = sp->consec_silence;
  const int y_len = sp->y_len, cap_len = sp->cap_len, slot = sp->slot;
  if (st_done) return;
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
    if (dy.logits_out && step < dy.logit_steps) {
      float* lo = dy.logits_out + (((long)step * dy.n_seq + slot) * a.K + k) * V;
#pragma unroll
      for (int j = 0; j < VC_VPL; ++j) if (lane + 64 * j < V) lo[lane + 64 * j] = v[j];
    }
#pragma unroll
    for (int j = 0; j < VC_VPL; ++j) if (lane + 64 * j < VP) sv[lane + 64 * j] = v[j];
    VC_TS(1);
    // ---- logit edits, in the reference's order; each touches one element, so lane 0 does them in LDS
    if (lane == 0) {
      if (kill >= 0) sv[kill] = -10000.f;
      const bool kill_tl = (n_eog == 0) ? (k >= 1) : (k > n_eog);      // [term],[empty] on later codebooks
      if (kill_tl) { sv[term] = -10000.f; sv[a.empty_token] = -10000.f; }
      if (n_eog == 0 && k == 0) {
        if (min_gen >= 0 && cur_num_gen <= min_gen) sv[term] = -10000.f;
        if (dy.stop_repetition > 0 && prev_token >= 0 && is_silence(dy, prev_token) && consec > dy.stop_repetition) {
          const float f = (float)(consec - (dy.stop_repetition - 1));
          const float x = sv[prev_token];
          sv[prev_token] = (x < 0.f) ? x * f : x / f;
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
    if (k == 0) {                                   // only codebook 0's arg-max is ever looked at (voicecraft.py:1041)
#pragma unroll
      for (int j = VC_VPL - 1; j >= 0; --j) bi = (v[j] == bv) ? lane + 64 * j : bi;     // padding is -inf, never equal
      bi = wave_min_i(bi);
    }
    VC_TS(2);
    const float u = philox_uniform(dy.seed, (uint32_t)slot, (uint32_t)step, (uint32_t)k);
    int tok = filter_draw(dy.top_k, dy.top_p, dy.temperature, v, bv, V, u);
    if (dy.forced && dy.forced_mode == 1 && step < dy.n_forced)      // replay of recorded reference draws (parity tests)
      tok = (int)dy.forced[((long)step * dy.n_seq + slot) * a.K + k];
    if (lane == 0) {
      xs[k] = tok;
      if (k == 0) xs[a.K] = bi;
    }
  }
  VC_TS(5);
  __syncthreads();
  if (tid == 0) {
    int c = 0;
    if (n_eog == 0) c = (xs[0] == term) || (xs[a.K] == term) || (y_len > cap_len);
    xs[a.K + 1] = c;
  }
}

// Phase 2 (one block per sequence): advance the state machine, log the step's tokens and build
// the next decode rows (embedding sum + position, voicecraft.py:1102-1116; span switch :838-858).
__device__ void advance_phase(const SampleArgs& a, const SampleDyn& dy, int b, bool grouped, SeqState* sp, const int* xs) {
  __shared__ int s_tok[VC_MAX_CODEBOOKS];
  __shared__ int s_mode;     // 0: row inactive, 1: one new row, 3: span switch (three rows)
  __shared__ int s_ylen, s_mask, s_Lx;
  const int tid = threadIdx.x;
  const int K = a.K;
  if (tid == 0) {
    // scalar fields in registers (fetched from LDS in one go); only the span arrays are indexed dynamically
    const int done0 = sp->done, Lx = sp->Lx, term = sp->term_token, group = sp->group, n_spans = sp->n_spans, slot = sp->slot;
    int n_eog = sp->n_eog, cur = sp->cur_num_gen, prev = sp->prev_token, consec = sp->consec_silence;
    int span = sp->span, total = sp->total_steps, y_len = sp->y_len;
    int mode = 0, mask = 0, ylen_row = y_len;
    if (!done0) {
      int tok[VC_MAX_CODEBOOKS];
#pragma unroll
      for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) tok[k] = (k < K) ? xs[k] : 0;
      int cond = xs[K + 1];
      bool drop = false;
      if (grouped && n_eog == 0) {
        // best-of-N (voicecraft.py:1296-1302): the LAST sample whose first codebook terminates is kept
        int keep = -1;
        for (int bb = 0; bb < a.B; ++bb)
          if (a.st[bb].group == group && a.samp[bb * (VC_MAX_CODEBOOKS + 2) + K + 1]) keep = bb;
        // (group ids are immutable and every member is still alive while n_eog == 0)
        if (keep >= 0 && keep != b) drop = true;
      }
      const int step = total;
      const bool forced = dy.forced && dy.forced_mode == 0 && step < dy.n_forced;
      if (n_eog == 0) {
#pragma unroll
        for (int k = 1; k < VC_MAX_CODEBOOKS; ++k)             // the first K-1 steps: codebooks cur+1.. are still `empty`
          if (k < K && k > cur) tok[k] = a.empty_token;
        if (forced) {
#pragma unroll
          for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) if (k < K) tok[k] = (int)dy.forced[((long)step * dy.n_seq + slot) * K + k];
          cond = (tok[0] == term);
        }
        if (cond) { tok[0] = term; n_eog = 1; }
        consec = (is_silence(dy, tok[0]) && tok[0] == prev) ? consec + 1 : 0;
        prev = tok[0];
      } else {
#pragma unroll
        for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) {
          if (k < n_eog) tok[k] = a.empty_token;
          if (k == n_eog) tok[k] = term;
        }
        if (forced) {
#pragma unroll
          for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) if (k < K) tok[k] = (int)dy.forced[((long)step * dy.n_seq + slot) * K + k];
        }
        n_eog += 1;
      }
      cur += 1;
      // every token final and in its own register BEFORE the first store: consumed inside the predicated store blocks, the
      // (test-only) forced-token loads above make the compiler wait vmcnt(0) in each block - every store behind the last
#pragma unroll
      for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) asm volatile("" : "+v"(tok[k]));
      if (step < dy.max_steps) {
#pragma unroll
        for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) if (k < K) a.gen[((long)slot * a.gen_stride + step) * K + k] = tok[k];
      }
      total = step + 1;
#pragma unroll
      for (int k = 0; k < VC_MAX_CODEBOOKS; ++k) if (k < K) s_tok[k] = tok[k];
      int done = 0, kept = 1;
      if (drop) {
        done = 1; kept = 0;
      } else if (n_eog == K) {            // span finished
        sp->span_steps[span] = cur;
        cur = 0;
        n_eog = 0;
        span += 1;
        if (span >= n_spans || total >= dy.max_steps || Lx + y_len + 3 > a.max_positions) {
          done = 1;
        } else {
          mode = 3;
          mask = sp->mask_value[span];
          y_len += 3;
          prev = -1;
          consec = 0;
        }
      } else if (total >= dy.max_steps || Lx + y_len + 1 > a.max_positions) {   // capacity guard (never hit when sized right)
        done = 1;
      } else {
        mode = 1;
        y_len += 1;
      }
      if (done) {
        sp->done = 1;
        if (atomicSub(a.n_active, 1) == 1)     // the last live sequence: tell the host loop without a stream operation
          __hip_atomic_store(a.host_active, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (!kept) sp->kept = 0;
      sp->n_eog = n_eog; sp->cur_num_gen = cur; sp->prev_token = prev; sp->consec_silence = consec;
      sp->span = span; sp->total_steps = total; sp->y_len = y_len;
    }
    s_mode = mode; s_mask = mask; s_ylen = ylen_row; s_Lx = Lx;
    const int r0 = b * a.rps;
    for (int i = 0; i < a.rps; ++i) {
      a.row_seq[r0 + i] = slot;
      a.row_pos[r0 + i] = (i < mode) ? (Lx + ylen_row + i) : -1;
    }
    a.logit_row[b] = r0 + (mode == 3 ? 2 : 0);
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
  // two columns per thread and pass (d = 2048: the whole row in ONE round trip instead of two dependent ones)
  for (int i = tid; i < nq; i += 2 * blockDim.x) {
    const int i2 = i + blockDim.x;
    const bool two = i2 < nq;
    const int j2 = two ? i2 : i;
    const float4 pa = *reinterpret_cast<const float4*>(pe0 + i * 4), pb = *reinterpret_cast<const float4*>(pe0 + j2 * 4);
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 4) {            // eight gathers in flight per pass; summed in order k = 0..K-1
      float4 ea[4], eb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = (k0 + q < K) ? k0 + q : 0;
        const float* row = a.audio_emb + ((long)kk * a.V + s_tok[kk]) * d;
        ea[q] = *reinterpret_cast<const float4*>(row + i * 4);
        eb[q] = *reinterpret_cast<const float4*>(row + j2 * 4);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k0 + q < K) {
          if (k0 + q == 0) { va = ea[q]; vb = eb[q]; }
          else {
            va.x += ea[q].x; va.y += ea[q].y; va.z += ea[q].z; va.w += ea[q].w;
            vb.x += eb[q].x; vb.y += eb[q].y; vb.z += eb[q].z; vb.w += eb[q].w;
          }
        }
    }
    va.x += a.alpha_audio * pa.x; va.y += a.alpha_audio * pa.y; va.z += a.alpha_audio * pa.z; va.w += a.alpha_audio * pa.w;
    vb.x += a.alpha_audio * pb.x; vb.y += a.alpha_audio * pb.y; vb.z += a.alpha_audio * pb.z; vb.w += a.alpha_audio * pb.w;
    *reinterpret_cast<float4*>(h0 + i * 4) = va;
    if (two) *reinterpret_cast<float4*>(h0 + i2 * 4) = vb;
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
  const long long t_entry = clock64();
  float v0[VC_VPL];
  preload_row(a, blockIdx.x, v0);
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  const SampleDyn dy = *a.dyn;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {      // (one writer per step, steps in order)
    const int c = *a.step_ctr;
    *a.step_ctr = c + 1;
    __hip_atomic_store(a.host_live + ((c / a.graph_steps) & 1), active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  VC_TS(0);
  if (dy.dbg_ts && b < 8 && threadIdx.x == 0) {   // diagnosis: entry / exit clock of every block
    dy.dbg_ts[16 + 2 * b] = t_entry;
    dy.dbg_ts[32 + 2 * b] = wall_clock64();       // 100 MHz, one counter for the whole chip (clock64 is per XCD); read
  }                                               // after the state fetch (~1.3 us in), only when stamps are requested
  park_state(&s_st, sw);
  sample_phase(a, dy, blockIdx.x, &s_st, s_xs, s_dyn, v0);
  __syncthreads();
  VC_TS(6);
  advance_phase(a, dy, blockIdx.x, false, &s_st, s_xs);
  VC_TS(8);
  store_state(a, blockIdx.x, &s_st);
  VC_TS(9);
  if (dy.dbg_ts && b < 8 && threadIdx.x == 0) {
    const long long t_exit = clock64();
    dy.dbg_ts[17 + 2 * b] = t_exit;
    dy.dbg_ts[33 + 2 * b] = wall_clock64();
    // sums over ALL steps of a call (the stamps above keep the last step only): slots 48..54 = the seven phase
    // deltas of block 0 (stamps 0,1,2,5,6,7,8,9), 55 = block 0 entry -> exit, 56 = steps summed, 57 = the same
    // total for the last block, 58 = its steps.  Steps whose block took the early exits are left out.
    long long* acc = dy.dbg_ts + 48;
    if (b == 0) {
      const int idx[8] = {0, 1, 2, 5, 6, 7, 8, 9};
      bool ok = true;
      for (int i = 0; i < 7; ++i) ok = ok && dy.dbg_ts[idx[i + 1]] >= dy.dbg_ts[idx[i]] && dy.dbg_ts[idx[i]] >= t_entry;
      if (ok) {
        for (int i = 0; i < 7; ++i) acc[i] += dy.dbg_ts[idx[i + 1]] - dy.dbg_ts[idx[i]];
        acc[7] += t_exit - t_entry;
        acc[8] += 1;
      }
    }
    if (b == (int)gridDim.x - 1) { acc[9] += t_exit - t_entry; acc[10] += 1; }
  }
}
// (A twin of this kernel that parks SampleArgs / SampleDyn in LDS first - 4 scalar kernarg loads instead of 23, no argument
// re-loads in the middle of the dependent chain - was built and measured in round 4: +0.25 % per step at one sequence, +0.18 % at
// eight, in-process; profiles/r04e_bench_*sampler_lds*.  The sampler's 12-26 us at one sequence turned out to vary with the BOX
// (12.5 us and 26 us on two boxes for the same code, stamps in profiles/r04d_sampler_stamps_b1.log / r04e_*), not with this.)
__global__ __launch_bounds__(256) void sample_only_k(const SampleArgs a) {
  __shared__ SeqState s_st;
  float v0[VC_VPL];
  preload_row(a, blockIdx.x, v0);
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  const SampleDyn dy = *a.dyn;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  park_state(&s_st, sw);
  sample_phase(a, dy, blockIdx.x, &s_st, a.samp + blockIdx.x * (VC_MAX_CODEBOOKS + 2), s_dyn, v0);
}
__global__ __launch_bounds__(256) void advance_only_k(const SampleArgs a) {
  __shared__ SeqState s_st;
  const int sw = fetch_state(a, blockIdx.x);
  const int active = *a.n_active;
  const SampleDyn dy = *a.dyn;
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  const int b = blockIdx.x;
  park_state(&s_st, sw);
  advance_phase(a, dy, blockIdx.x, true, &s_st, a.samp + blockIdx.x * (VC_MAX_CODEBOOKS + 2));
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

// =============================================================== re-packing a wide batch onto fewer rows
// A multi-utterance call keeps one ROW of the decode step per sequence; sequences retire at different steps (their own terminator /
// length cap, models/voicecraft.py:1041-1045 per sequence), and a retired row idles in place.  Once few enough are left for a
// narrower launch form, the host's decode loop queues this kernel between two graphs: the live sequences move - in their order - to rows
// [0, n_live) (state, next input row, row tables), rows [n_live, B_new) become inactive fillers, and the final states of the retired
// ones are parked in st_fin[slot].  Everything per-SEQUENCE (KV cache, gen log, forced / logits_out rows, Philox stream) is
// indexed by SeqState.slot and does not move.  One workgroup; runs a handful of times per call.
__global__ __launch_bounds__(256) void repack_k(const RepackArgs a) {
  __shared__ SeqState s_st[VC_MAX_SEQS];
  __shared__ int s_src[VC_MAX_SEQS], s_pos[VC_MAX_SEQS], s_live;
  const int tid = threadIdx.x;
  constexpr int W = sizeof(SeqState) / 4;
  for (int i = tid; i < a.B_old * W; i += blockDim.x) reinterpret_cast<int*>(s_st)[i] = reinterpret_cast<const int*>(a.st)[i];
  if (tid < a.B_old) s_pos[tid] = a.row_pos[tid];
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int r = 0; r < a.B_old; ++r)
      if (!s_st[r].done) s_src[n++] = r;
    s_live = n;
    if (n > a.B_new) atomicOr(a.err, 4);
  }
  __syncthreads();
  const int n_live = s_live;
  if (n_live > a.B_new) return;
  // final states of the retired sequences, by slot
  for (int r = 0; r < a.B_old; ++r) {
    if (!s_st[r].done || s_st[r].slot < 0) continue;
    if (tid < W) reinterpret_cast<int*>(a.st_fin + s_st[r].slot)[tid] = reinterpret_cast<const int*>(s_st + r)[tid];
  }
  // states and row tables of the new layout
  for (int r = 0; r < a.B_new; ++r) {
    if (r < n_live) {
      if (tid < W) reinterpret_cast<int*>(a.st + r)[tid] = reinterpret_cast<const int*>(s_st + s_src[r])[tid];
      if (tid == 0) { a.row_seq[r] = s_st[s_src[r]].slot; a.row_pos[r] = s_pos[s_src[r]]; a.logit_row[r] = r; }
    } else {       // filler: a finished state that owns no sequence
      if (tid < W) reinterpret_cast<int*>(a.st + r)[tid] = (tid == (int)(offsetof(SeqState, done) / 4)) ? 1 : (tid == (int)(offsetof(SeqState, slot) / 4)) ? -1 : 0;
      if (tid == 0) { a.row_seq[r] = 0; a.row_pos[r] = -1; a.logit_row[r] = r; }
    }
  }
  // next input rows: source index >= destination and increasing, so walking the destinations in order never reads a row already overwritten
  const int nq = a.d >> 2;
  for (int r = 0; r < n_live; ++r) {
    const int src = s_src[r];
    if (src == r) continue;
    const float4* sp = reinterpret_cast<const float4*>(a.dec_h + (long)src * a.d);
    float4* dp = reinterpret_cast<float4*>(a.dec_h + (long)r * a.d);
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (tid < nq) v0 = sp[tid];
    if (tid + 256 < nq) v1 = sp[tid + 256];
    if (tid < nq) dp[tid] = v0;
    if (tid + 256 < nq) dp[tid + 256] = v1;
    __syncthreads();
  }
}
hipError_t vc_launch_repack(const RepackArgs& a, hipStream_t s) {
  if (a.B_old < 1 || a.B_old > VC_MAX_SEQS || a.B_new < 1 || a.B_new > a.B_old || a.d > 2048) return hipErrorInvalidValue;
  hipLaunchKernelGGL(repack_k, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

// =============================================================== sampler test hook
// n_draws independent draws from ONE logits row through the product filter_draw (the distribution test
// of tests/test_gpu_sampler.py): draw i uses the Philox counter (seed, sequence i, step 0, codebook 0).
__global__ __launch_bounds__(256) void sample_test_k(const float* __restrict__ logits, int V, int top_k, float top_p,
                                                     float temperature, uint64_t seed, int n_draws, int* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n_draws) return;
  float v[VC_VPL];
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) {
    const int c = lane + 64 * j;
    const float t = logits[min(c, V - 1)];
    v[j] = (c < V) ? t : -INFINITY;
  }
  float bv = -INFINITY;
#pragma unroll
  for (int j = 0; j < VC_VPL; ++j) bv = fmaxf(bv, v[j]);
  bv = wave_max(bv);
  const float u = philox_uniform(seed, (uint32_t)i, 0u, 0u);
  const int tok = filter_draw(top_k, top_p, temperature, v, bv, V, u);
  if (lane == 0) out[i] = tok;
}
extern "C" int vc_debug_sample(const float* logits_dev, int V, const vc_sample_cfg* sc, int n_draws,
                               int32_t* out_dev, void* stream) {
  if (!logits_dev || !sc || !out_dev || V < 1 || V > 64 * VC_VPL || n_draws < 1) return VC_EINVAL;
  hipLaunchKernelGGL(sample_test_k, dim3((n_draws + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits_dev, V,
                     sc->top_k, sc->top_p, sc->temperature, sc->seed, n_draws, out_dev);
  return hipGetLastError() == hipSuccess ? VC_OK : VC_EHIP;
}

// =============================================================== box probe (diagnosis)
// ONE thread walks a dependent chain of loads (next = ring[next]) and stamps both clocks around it: the per-XCD shader clock
// (clock64) and the chip-wide 100 MHz counter (wall_clock64).  hops / wall = the latency one dependent load costs a lone
// workgroup on an otherwise idle chip; shader clocks / wall = the clock that workgroup really ran at.  The sampler at one
// sequence is exactly such a workgroup (DESIGN 4.3 f: 12.5 us on one box, 26 us on another, same code): bench.py reports these
// numbers next to the sampler's time so that boxes can be told apart.
__global__ void box_probe_init_k(unsigned* ring, unsigned n_mask, unsigned stride) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= n_mask) ring[(size_t)i * 16] = ((i * 5u + stride) & n_mask);      // full period: multiplier = 1 mod 4, odd increment; one entry per 64 B
}
__global__ void box_probe_k(const unsigned* __restrict__ ring, unsigned start, int warm, int hops, long long* out) {
  if (threadIdx.x != 0) return;
  unsigned p = start;
  for (int i = 0; i < warm; ++i) p = ring[(size_t)p * 16];                   // untimed: a whole lap of a small ring (cache-resident walk), nothing of a large one
  const long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < hops; ++i) p = __builtin_nontemporal_load(ring + (size_t)p * 16);
  const long long c1 = clock64(), w1 = wall_clock64();
  out[0] = w1 - w0; out[1] = c1 - c0; out[2] = p;
}
// res[0] = ns per dependent load over a `bytes`-sized ring, res[1] = shader clock in MHz during the walk
extern "C" int vc_box_probe(long long bytes, int hops, float res[2], void* stream) {
  if (bytes < 4096 || hops < 1 || !res) return VC_EINVAL;
  unsigned n = 1;
  while ((long long)n * 2 * 64 <= bytes) n *= 2;
  unsigned* ring = nullptr;
  long long* out = nullptr;
  if (hipMalloc((void**)&ring, (size_t)n * 64) != hipSuccess) return VC_EHIP;
  if (hipMalloc((void**)&out, 64) != hipSuccess) { (void)hipFree(ring); return VC_EHIP; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(box_probe_init_k, dim3((n + 255) / 256), dim3(256), 0, s, ring, n - 1, 12345u | 1u);
  long long h[3] = {0, 0, 0};
  hipError_t e = hipSuccess;
  // a ring of up to 4 MB is walked once round untimed (every line then sits in the probing XCD's L2: the timed hops are cache hits);
  // a larger one is entered at a different place in each of the two launches (the first warms code and TLB, the second - reported -
  // touches lines nothing has read since the init kernel wrote them: memory latency)
  const bool small = (long long)n * 64 <= (4LL << 20);
  for (int rep = 0; rep < 2 && e == hipSuccess; ++rep) {
    hipLaunchKernelGGL(box_probe_k, dim3(1), dim3(64), 0, s, ring, small ? 0u : (rep ? n / 2 + 77u : 0u), small ? (int)n : 0, hops, out);
    e = hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  (void)hipFree(ring); (void)hipFree(out);
  if (e != hipSuccess || h[0] <= 0) return VC_EHIP;
  res[0] = (float)((double)h[0] * 10.0 / hops);              // 100 MHz ticks -> ns
  res[1] = (float)((double)h[1] / ((double)h[0] * 0.01));    // shader clocks per microsecond
  return VC_OK;
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
// ---- training objective (VoiceCraft.forward, models/voicecraft.py:535-543): F.cross_entropy term and top-10 membership
// of one logits row per block.  Two passes over the V logits of the row (maximum; sum of exponentials + number of
// logits above the target's), fp32 like torch's log_softmax.  A target tied with the 10th value counts as a hit
// (torch.topk picks ten indices among equals arbitrarily).
__global__ __launch_bounds__(256) void ce_rows_k(const CeArgs a) {
  __shared__ float s_f[4];
  __shared__ int s_i[4];
  const int r = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long o = (long)r * a.K + k;
  const int t = a.tgt[o];
  if (t == -1) {
    if (tid == 0) { a.nll[o] = 0.f; a.hit[o] = 0; }
    return;
  }
  long tok = t >= 0 ? a.y[t] : (long)(-(t + 2));
  if (tok < 0 || tok >= a.V) { if (tid == 0) *a.err = 1; tok = 0; }
  const float* row = a.logits + o * a.V;
  float m = -INFINITY;
  for (int i = tid; i < a.V; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) s_f[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_f[0], s_f[1]), fmaxf(s_f[2], s_f[3]));
  __syncthreads();
  const float tl = row[tok];
  float se = 0.f;
  int gt = 0;
  for (int i = tid; i < a.V; i += 256) {
    const float v = row[i];
    se += expf(v - m);
    gt += (v > tl) ? 1 : 0;
  }
  se = wave_sum(se);
  gt = wave_sum_i(gt);
  if (lane == 0) { s_f[wave] = se; s_i[wave] = gt; }
  __syncthreads();
  if (tid == 0) {
    const float tot = (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]);
    const int above = (s_i[0] + s_i[1]) + (s_i[2] + s_i[3]);
    a.nll[o] = (logf(tot) + m) - tl;
    a.hit[o] = above < 10 ? 1 : 0;
  }
}
hipError_t vc_launch_ce(const CeArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(ce_rows_k, dim3(a.n_rows, a.K), dim3(256), 0, s, a);
  return hipGetLastError();
}
// one block per codebook; thread i adds rows i, i + 256, ... in order, then a fixed tree: the same sum on every run
__global__ __launch_bounds__(256) void ce_reduce_k(const CeReduceArgs a) {
  __shared__ double s_d[256];
  __shared__ long long s_h[256], s_c[256];
  const int k = blockIdx.x, tid = threadIdx.x;
  double acc = 0.0;
  long long h = 0, c = 0;
  for (long r = tid; r < a.n_rows; r += 256) {
    const long o = r * a.K + k;
    if (a.tgt[o] != -1) { acc += (double)a.nll[o]; h += a.hit[o]; c += 1; }
  }
  s_d[tid] = acc; s_h[tid] = h; s_c[tid] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { s_d[tid] += s_d[tid + w]; s_h[tid] += s_h[tid + w]; s_c[tid] += s_c[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) { a.nll_sum[k] += s_d[0]; a.hits[k] += s_h[0]; a.count[k] += s_c[0]; }
}
hipError_t vc_launch_ce_reduce(const CeReduceArgs& a, hipStream_t s) {
  if (a.n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(ce_reduce_k, dim3(a.K), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t vc_launch_assemble(const AssembleArgs& a, hipStream_t s) {
  if (a.n_piece <= 0) return hipSuccess;
  hipLaunchKernelGGL(assemble_k, dim3(8, a.n_piece), dim3(256), 0, s, a);
  return hipGetLastError();
}
