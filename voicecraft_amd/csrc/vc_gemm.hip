// vc_gemm.hip — weight packing and the "rows GEMM": out[r][n] = sum_k W[n][k] * X[r][k] for up to
// 16 rows (token positions) per pass, streaming W once from HBM.
//
// Replaces every F.linear on the reference decode path (models/modules/activation.py:86,:637,
// models/modules/transformer.py:386-388, models/voicecraft.py:181-185,:1085) together with the
// F.layer_norm in front of it (transformer.py:73-75) and the residual adds (transformer.py:328-329).
//
// Layout: W is stored in HBM already in MFMA A-fragment order,
//   Wp[n_tile][k_tile][lane] = 16 bytes = W[16*n_tile + (lane&15)][KW*k_tile + EPL*(lane>>4) + 0..EPL)
// so one wave-wide 16-byte load is a contiguous 1 KiB burst that feeds one MFMA with no LDS round
// trip; the 16 rows of X are the B operand, staged once per block in LDS.  At batch 1 the kernel is a
// pure HBM stream (1 FLOP/byte); the MFMA is only the cheapest way to consume 1 KiB per instruction
// and makes rows 2..16 (batched decode, prefill groups, the 3-row span switch of editing) free.
//
// LayerNorm fold.  Every LayerNorm on the path feeds a linear layer (transformer.py:328-329, voicecraft.py
// :1084-1086), and LN(x) = (x - mean) * rstd * gamma + beta is affine in x up to two per-row scalars:
//   W LN(x) + c = rstd * ( (W . gamma) x  -  mean * rowsum(W . gamma) )  +  (W beta + c)
// so the engine stores W' = W . gamma (column-scaled at load time, pack_k), wg = rowsum(W') over the
// ROUNDED elements and cb = W beta + c (fold_vecs_k).  A decode GEMM then multiplies the CENTRED, un-scaled row
// xc = x - mean(x) (one block barrier for the mean; centring BEFORE the rounding to bf16 keeps the mantissa for the
// signal when the residual stream carries a large common offset - tests/test_gpu_model.py, offset case) and applies
// rstd and the small residual mean of the rounded row in its epilogue, from two wave sums computed in the shadow of
// the weight stream.  Multi-row passes normalise once per row in ln_rows_k (x_hat = (x - mean) * rstd, no affine)
// and use the same W' / cb through the plain prologue.
//
// Block = 4 waves sharing one 16-row output tile; the waves split the block's K range 4 ways and
// reduce through LDS.  Cross-block split-K (EPI_PART) leaves fp32 partial slabs that the NEXT
// kernel's LayerNorm prologue sums (launch-boundary reduce: no atomics, deterministic) - the form of ONE-row steps.
// Steps of 2..16 rows keep whole residual rows instead (finished-row form, round 4): rows_gemm_fr_k / rows_gemm_fr2_k produce
// them, the PRO_LNW prologue of rows_gemm_k folds their LayerNorm one wave per row.
#include <algorithm>
#include "vc_common.h"

// ------------------------------------------------------------------ packing
// th = output channels per tile: 16 (every lane of the A fragment), 12 (VC_TH_QKV: lanes with
// (lane & 15) >= 12 carry no weight and nothing is stored for them - a tile is 4 x 12 fragments) or 8 (VC_TH_RES: the second
// image of the out-projection / FFN-down matrices that the finished-row producers stream, d/8 whole-K tiles).
template <typename WT>
__global__ void pack_k(const float* __restrict__ src, const float* __restrict__ colscale, WT* __restrict__ dst,
                       int N, int K, int KT, long total, int th) {
  constexpr int EPL = WTr<WT>::EPL, KW = WTr<WT>::KW;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (tile, fragment slot)
  if (idx >= total) return;
  const int spt = 4 * th;                                   // slots per (n_tile, k_tile)
  const int slot = (int)(idx % spt);
  const long tile = idx / spt;
  const int kt = (int)(tile % KT);
  const int nt = (int)(tile / KT);
  const int n = nt * th + (slot % th);
  const int k = kt * KW + EPL * (slot / th);
  WT* d = dst + idx * EPL;
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    float v = (n < N) ? src[(long)n * K + k + j] : 0.0f;
    if (colscale) v *= colscale[k + j];
    WTr<WT>::st(d + j, v);
  }
}

hipError_t vc_launch_pack(const float* src, const float* colscale, void* dst, int N, int K, int dtype, int th,
                          hipStream_t s) {
  const int n_tiles = (N + th - 1) / th;
  if (dtype == VC_DTYPE_BF16) {
    const int KT = K / 32;
    long total = (long)n_tiles * KT * 4 * th;
    hipLaunchKernelGGL(pack_k<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, colscale,
                       (bf16_t*)dst, N, K, KT, total, th);
  } else {
    const int KT = K / 16;
    long total = (long)n_tiles * KT * 4 * th;
    hipLaunchKernelGGL(pack_k<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, colscale,
                       (float*)dst, N, K, KT, total, th);
  }
  return hipGetLastError();
}

// wg[n] = sum_k WT(W[n][k] * gamma[k]) (the rounded elements the MFMA will see), cb[n] = bias[n] + sum_k W[n][k] * beta[k].
// One wave per output channel.
template <typename WT>
__global__ __launch_bounds__(256) void fold_vecs_k(const float* __restrict__ W, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ bias,
                                                   float* __restrict__ wg, float* __restrict__ cb, int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float sg = 0.f, sb = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(long)n * K + k];
    WT r;
    WTr<WT>::st(&r, w * gamma[k]);
    sg += WTr<WT>::ld(&r);
    sb += w * beta[k];
  }
  sg = wave_sum(sg);
  sb = wave_sum(sb);
  if (lane == 0) { wg[n] = sg; cb[n] = bias[n] + sb; }
}
hipError_t vc_launch_fold_vecs(const float* W, const float* gamma, const float* beta, const float* bias, float* wg,
                               float* cb, int N, int K, int dtype, hipStream_t s) {
  if (dtype == VC_DTYPE_BF16)
    hipLaunchKernelGGL(fold_vecs_k<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, s, W, gamma, beta, bias, wg, cb, N, K);
  else
    hipLaunchKernelGGL(fold_vecs_k<float>, dim3((N + 3) / 4), dim3(256), 0, s, W, gamma, beta, bias, wg, cb, N, K);
  return hipGetLastError();
}

template <typename WT>
__global__ void cast_k(const float* __restrict__ src, WT* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) WTr<WT>::st(dst + i, src[i]);
}
hipError_t vc_launch_cast(const float* src, void* dst, long n, int dtype, hipStream_t s) {
  unsigned g = (unsigned)((n + 255) / 256);
  if (dtype == VC_DTYPE_BF16)
    hipLaunchKernelGGL(cast_k<bf16_t>, dim3(g), dim3(256), 0, s, src, (bf16_t*)dst, n);
  else
    hipLaunchKernelGGL(cast_k<float>, dim3(g), dim3(256), 0, s, src, (float*)dst, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------ rows GEMM

#include "vc_gemm_dev.h"

// Decode-step kernel.  The dependency chain of a launch is what the ~80 launches of a step pay for, so
// the kernel is written around it: (1) everything any later stage needs from HBM - epilogue operands,
// the first row's prologue operands, the "any sequence still active" word - is requested first, in the
// order it will be consumed (a wave's loads return in order); (2) the weight burst follows immediately
// behind from a wave-uniform base (SGPR addressing: no per-load VALU address arithmetic, no register
// reuse that would make the compiler wait mid-burst); (3) a scheduling barrier keeps the prologue
// arithmetic from being hoisted into the burst; (4) only then is anything waited for.
// NTW = 2 (PRO_LNW only): TWO weight tiles per workgroup, 8 waves - waves 0..3 stream tile 2x, waves 4..7 tile 2x + 1 - sharing ONE
// staged copy of the rows.  The finished-row LayerNorm prologue reads 8 rows x 8 KB of fp32 h per workgroup out of L2: with
// one tile per workgroup that is 33.5 MB per launch, as much as the weight stream itself (measured: QKV 6.9 -> 9.1 us, FFN-up
// 7.0 -> 9.5 us against the plain prologue); two tiles halve it, and each of the 8 waves folds exactly one row.
// NT: the weight stream uses non-temporal loads (decode: every byte is read exactly once per launch).  A TEMPLATE parameter, not
// the runtime flag it used to be: with both load arms in one kernel the compiler merged them and silently dropped the hint -
// through round 3 the compiled QKV, out-projection and heads-2 forms had no non-temporal load at all, the FFN forms 15 of 16.
// R2 (NTW = 2 only): every wave folds TWO rows (w and w + 8) - passes of 9..16 finished rows.
// NP (PRO_LN only): split-K slabs the prologue requests per row - 4 (any n_parts <= VC_MAX_KSPLIT, the round-1..4 form), 2 (behind
// the out-projection's two slabs) or 0 (the row in h_in is FINISHED: no slabs, no pending bias - layer 0, and behind a producer
// that wrote whole rows).  Every workgroup of a consumer launch used to pull h + 4 slabs = 40 KB out of L2 whatever n_parts said
// (unconditional loads, unused slabs discarded by a select): 512 workgroups x 40 KB = 20 MB through the CUs' vector memory
// pipes per launch, queued IN FRONT of the 25-33 MB weight burst (round 5: in-kernel stamps, profiles/r04b_kernel_stamps_*).
template <typename WT, int KTW, int PRO, int EPI, int NTW = 1, bool NT = true, bool R2 = false, int NP = VC_MAX_KSPLIT>
__global__ __launch_bounds__(256 * NTW) void rows_gemm_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr bool LNW = PRO == PRO_LNW || PRO == PRO_LNQ;       // finished-row consumers (fp32 rows / the producers' centred copy)
  static_assert(NTW == 1 || LNW, "two tiles per workgroup: finished-row consumers only");
  static_assert(NP == VC_MAX_KSPLIT || PRO == PRO_LN, "slab count: LayerNorm prologue of slab-form passes only");
  static_assert(NP == 0 || NP == 2 || NP == VC_MAX_KSPLIT, "slabs requested per row");
  static_assert(!R2 || NTW == 2, "two rows per wave: the two-tile form");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  VC_KTS_DECL();
  VC_KTS(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = (NTW == 1) ? wv : (wv & 3);    // K quarter of the tile
  const int tg = (NTW == 1) ? 0 : (wv >> 2);      // which of the workgroup's tiles
  const int nt = (NTW == 1) ? (int)blockIdx.x : (int)blockIdx.x * NTW + tg;
  const int ks = blockIdx.y, grp = blockIdx.z;
  const int n_rows = a.n_rows;                    // >= 1 (host contract)
  const int kt_blk = a.nchunk * 4 * KTW;          // k-tiles this block covers
  const int kt0 = ks * kt_blk;                    // first of them
  const int kblk = kt_blk * T::KW;                // K elements this block covers
  const int k0 = kt0 * T::KW;
  const int xs = kblk * (int)sizeof(WT) + 16;     // LDS row stride in bytes (+16: rotate bank slots)
  char* xl = smem;
  f32x4* red0 = reinterpret_cast<f32x4*>(smem + (size_t)a.r_lds * xs);
  f32x4* red = red0 + tg * 256;
  float* stat = reinterpret_cast<float*>(red0 + 256 * NTW);      // LN prologue: [row][wave][sum, sum of squares] of the centred, rounded row
  float* msum = stat + VC_ROWS * 4 * 2;                   // LN prologue: [row][wave] partial sums of the fp32 row (its mean)

  // Tile height: the QKV projection (N = 3d) uses 12-channel tiles, so that its 3d/12 = d/4 tiles are
  // a multiple of the CU count (512 workgroups of 48 KB at d = 2048 instead of 384 of 64 KB = 1.5 per
  // CU); lanes 12..15 of each fragment row group re-read slot 11 and are zeroed before the MFMA.
  constexpr int TH = (EPI == EPI_QKV) ? VC_TH_QKV : 16;
  constexpr int SPT = 4 * TH;                     // fragment slots per (n_tile, k_tile)
  const int active = *a.n_active;                 // scalar; looked at once the burst is on its way
  const int m = lane & 15;
  const int kg = lane >> 4;
  const int n = nt * TH + 4 * kg;
  const bool wvalid = m < TH;                     // this lane's fragment row exists
  const bool nvalid = 4 * kg < TH;                // this lane's 4 output channels exist
  const int wslot = kg * TH + min(m, TH - 1);
  float4 eb;
  int epos, eseq;
  epi_preload<WT, EPI>(a, (m < n_rows) ? m : 0, n, grp, eb, epos, eseq);
  float4 ewg = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (PRO == PRO_LN || LNW) ewg = wg_preload<WT, EPI>(a, n, grp);

  const uint4* wbase = a.Wp + (long)grp * a.w_group_stride + ((long)nt * a.KT + kt0 + wave * KTW) * SPT;   // wave-uniform
  uint4 wf[KTW];
#define VC_ISSUE_WEIGHTS(c_)                                                                     \
  {                                                                                              \
    const uint4* wb_ = wbase + (long)(c_) * (4 * KTW * SPT);                                      \
    if constexpr (NT) {                                                                          \
      _Pragma("unroll") for (int i = 0; i < KTW; ++i)                                            \
        wf[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb_ + i * SPT + wslot))); \
    } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < KTW; ++i) wf[i] = wb_[i * SPT + wslot];              \
    }                                                                                            \
  }
#define VC_BURST_OUT()                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                             \
  if (active == 0) return;   /* every sequence finished: the replayed step is a no-op */     \
  VC_KTS(1);

  // prologue: build the rows' X slice [n_rows][kblk] as WT in LDS.  Every load is unconditional and
  // branch-free (out-of-range lanes re-read a valid address, unused split slabs are read and discarded
  // by a select): a load inside a branch makes the compiler drain the whole queue (vmcnt(0)).
  if constexpr (PRO == PRO_LN) {
    // LayerNorm fold (see the top of this file): hn = h + prev_bias + sum(parts); X = hn - mean(hn), un-scaled; the
    // sum and sum of squares of the ROUNDED centred values the MFMA multiplies are reduced per wave here, in the
    // shadow of the weight stream, and combined in the epilogue (the residual mean of the rounded row and rstd).  The whole block works on one row at a time:
    // thread t owns float4 columns t and t+256.  Rows beyond the first are software-pipelined through two
    // register sets (the next row's loads fly during this row's arithmetic).
    const int d = a.d;
    const int nq = d >> 2;                               // float4 per row (<= 512)
    const bool on0 = tid < nq, on1 = tid + 256 < nq;
    const int c0 = on0 ? tid * 4 : 0, c1 = on1 ? (tid + 256) * 4 : 0;
    constexpr int NPA = NP > 0 ? NP : 1;
    float4 pb0 = make_float4(0.f, 0.f, 0.f, 0.f), pb1 = pb0;
    if constexpr (NP > 0) {
      pb0 = *reinterpret_cast<const float4*>(a.prev_bias + c0);
      pb1 = *reinterpret_cast<const float4*>(a.prev_bias + c1);
    }
    const bool use_pb = NP > 0 && a.has_prev_bias != 0;
    float4 x0A, x1A, p0A[NPA], p1A[NPA];
    float4 x0B, x1B, p0B[NPA], p1B[NPA];
#define VC_LOAD_ROW(S, r, G)   /* G: honour a.gather_rows (single-row launches only, host contract) */ \
    {                                                                                            \
      int sr_ = (r);                                                                             \
      if (G && a.gather_rows) sr_ = a.gather_rows[sr_];                                          \
      const float* hp_ = a.h_in + (long)sr_ * d;                                                 \
      x0##S = *reinterpret_cast<const float4*>(hp_ + c0);                                        \
      x1##S = *reinterpret_cast<const float4*>(hp_ + c1);                                        \
      _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_) {                                        \
        const float* pp_ = a.parts + ((long)(s_ * a.rows_cap + sr_)) * d;                        \
        p0##S[s_] = *reinterpret_cast<const float4*>(pp_ + c0);                                  \
        p1##S[s_] = *reinterpret_cast<const float4*>(pp_ + c1);                                  \
      }                                                                                          \
    }
#define VC_FINISH_ROW(S, r)                                                                      \
    {                                                                                            \
      float4 x0 = x0##S, x1 = x1##S;                                                             \
      if (use_pb) {                                                                              \
        x0.x += pb0.x; x0.y += pb0.y; x0.z += pb0.z; x0.w += pb0.w;                              \
        x1.x += pb1.x; x1.y += pb1.y; x1.z += pb1.z; x1.w += pb1.w;                              \
      }                                                                                          \
      _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_) {                                        \
        const bool u_ = s_ < a.n_parts;                                                          \
        x0.x += u_ ? p0##S[s_].x : 0.f; x0.y += u_ ? p0##S[s_].y : 0.f; x0.z += u_ ? p0##S[s_].z : 0.f; x0.w += u_ ? p0##S[s_].w : 0.f; \
        x1.x += u_ ? p1##S[s_].x : 0.f; x1.y += u_ ? p1##S[s_].y : 0.f; x1.z += u_ ? p1##S[s_].z : 0.f; x1.w += u_ ? p1##S[s_].w : 0.f; \
      }                                                                                          \
      const bool writer_ = a.h_out && grp == 0 && (((r) % (int)gridDim.x) == (int)blockIdx.x);   \
      if (writer_ && on0) *reinterpret_cast<float4*>(a.h_out + (long)(r) * d + c0) = x0;         \
      if (writer_ && on1) *reinterpret_cast<float4*>(a.h_out + (long)(r) * d + c1) = x1;         \
      /* the row is CENTRED before it is rounded to the MFMA's input type: a residual stream with a large common   */ \
      /* offset (|mean| >> sigma, usual in trained pre-LN stacks) would otherwise spend its bf16 mantissa on the   */ \
      /* offset and the rounding error would come out of the fold amplified by |mean| / sigma                      */ \
      float t_ = (on0 ? ((x0.x + x0.y) + (x0.z + x0.w)) : 0.f) + (on1 ? ((x1.x + x1.y) + (x1.z + x1.w)) : 0.f);  \
      t_ = wave_sum(t_);                                                                         \
      if (lane == 0) msum[(r) * 4 + wave] = t_;                                                  \
      __syncthreads();                                                                           \
      const float4 ms_ = *reinterpret_cast<const float4*>(msum + (r) * 4);                       \
      const float mu_ = ((ms_.x + ms_.y) + (ms_.z + ms_.w)) * (1.0f / (float)d);                 \
      WT* xr_ = reinterpret_cast<WT*>(xl + (size_t)(r) * xs);                                    \
      float s1_ = 0.f, s2_ = 0.f;                                                                \
      if (on0) {                                                                                 \
        const f32x4 y_ = {x0.x - mu_, x0.y - mu_, x0.z - mu_, x0.w - mu_};                       \
        const f32x4 q_ = store4r(xr_ + c0, y_);                                                  \
        s1_ += (q_[0] + q_[1]) + (q_[2] + q_[3]);                                                \
        s2_ += (q_[0] * q_[0] + q_[1] * q_[1]) + (q_[2] * q_[2] + q_[3] * q_[3]);                \
      }                                                                                          \
      if (on1) {                                                                                 \
        const f32x4 y_ = {x1.x - mu_, x1.y - mu_, x1.z - mu_, x1.w - mu_};                       \
        const f32x4 q_ = store4r(xr_ + c1, y_);                                                  \
        s1_ += (q_[0] + q_[1]) + (q_[2] + q_[3]);                                                \
        s2_ += (q_[0] * q_[0] + q_[1] * q_[1]) + (q_[2] * q_[2] + q_[3] * q_[3]);                \
      }                                                                                          \
      s1_ = wave_sum(s1_);                                                                       \
      s2_ = wave_sum(s2_);                                                                       \
      if (lane == 0) { stat[((r) * 4 + wave) * 2] = s1_; stat[((r) * 4 + wave) * 2 + 1] = s2_; } \
    }
    VC_LOAD_ROW(A, 0, 1);
    VC_ISSUE_WEIGHTS(0);
    VC_BURST_OUT();
    if (n_rows == 1) {
      VC_FINISH_ROW(A, 0);
    } else {
      int r = 0;
      for (;;) {
        VC_LOAD_ROW(B, min(r + 1, n_rows - 1), 0);
        VC_FINISH_ROW(A, r);
        if (++r >= n_rows) break;
        VC_LOAD_ROW(A, min(r + 1, n_rows - 1), 0);
        VC_FINISH_ROW(B, r);
        if (++r >= n_rows) break;
      }
    }
#undef VC_LOAD_ROW
#undef VC_FINISH_ROW
  } else if constexpr (PRO == PRO_LNW) {
    // LayerNorm fold of 2..8 FINISHED rows (h holds the whole residual, bias included: the finished-row producers below,
    // rows_gemm_fr_k): one WAVE per row - wave w takes rows w and w + 4 - so a row's mean and its two statistics are wave
    // reductions, the prologue has no block barrier per row and both rows of a wave are requested ahead of the weight
    // burst.  (PRO_LN walks the rows one after the other with the whole block: one dependent round trip + one barrier per
    // row in every workgroup - 20 us per GEMM at 8 rows - which is why several-row steps used a separate LayerNorm launch.)
    const int d = a.d;
    const int npl = d >> 8;                                 // float4 per lane and row: d / 4 / 64 = 1..8 (d % 256 == 0)
    const int rowA = (NTW == 1) ? wave : wv;                // (two tiles per workgroup: 8 waves, one row each - or two, R2)
    const int rowB = (NTW == 1) ? wave + 4 : wv + 8;
    constexpr bool TWO_ROWS = (NTW == 1) || R2;
    const float* hrA = a.h_in + (long)min(rowA, n_rows - 1) * d;
    const float* hrB = a.h_in + (long)min(rowB, n_rows - 1) * d;
    float4 xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = (min(j, npl - 1) * 64 + lane) * 4;
      xa[j] = *reinterpret_cast<const float4*>(hrA + c);
      if constexpr (TWO_ROWS) xb[j] = *reinterpret_cast<const float4*>(hrB + c);
    }
    VC_ISSUE_WEIGHTS(0);
    VC_BURST_OUT();
#define VC_LNW_ROW(X, r)                                                                         \
    if ((r) < n_rows) {                                                                          \
      float t_ = 0.f;                                                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
        if (j < npl) t_ += (X[j].x + X[j].y) + (X[j].z + X[j].w);                                \
      const float mu_ = wave_sum(t_) * (1.0f / (float)d);                                        \
      WT* xr_ = reinterpret_cast<WT*>(xl + (size_t)(r) * xs);                                    \
      float s1_ = 0.f, s2_ = 0.f;                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
        if (j < npl) {                                                                           \
          const f32x4 y_ = {X[j].x - mu_, X[j].y - mu_, X[j].z - mu_, X[j].w - mu_};             \
          const f32x4 q_ = store4r(xr_ + (j * 64 + lane) * 4, y_);                               \
          s1_ += (q_[0] + q_[1]) + (q_[2] + q_[3]);                                              \
          s2_ += (q_[0] * q_[0] + q_[1] * q_[1]) + (q_[2] * q_[2] + q_[3] * q_[3]);              \
        }                                                                                        \
      s1_ = wave_sum(s1_);                                                                       \
      s2_ = wave_sum(s2_);                                                                       \
      if (lane == 0) {      /* the epilogue sums four per-wave pairs: this row has one */          \
        *reinterpret_cast<float4*>(stat + (r) * 8) = make_float4(s1_, s2_, 0.f, 0.f);            \
        *reinterpret_cast<float4*>(stat + (r) * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);        \
        if (a.row_mu_out && blockIdx.x == 0 && grp == 0) a.row_mu_out[r] = mu_;      /* the next producer centres its copy of the row on it */ \
      }                                                                                          \
    }
    VC_LNW_ROW(xa, rowA);
    if constexpr (TWO_ROWS) { VC_LNW_ROW(xb, rowB); }
#undef VC_LNW_ROW
  } else if constexpr (PRO == PRO_LNQ) {
    // The same fold on the producer's centred copy q = WT(h - c) of the finished rows (c = a.row_mu[row], the mean the PREVIOUS LayerNorm of
    // the row found): W LN(h) = rstd (W' q - mean(q) rowsum(W')) + cb holds for any c, and |mean(q)| stays far below sigma because one
    // residual update moves a row's mean by little - the rounding of q keeps its mantissa for the signal, as with the exact mean.  A wave
    // copies its row(s) from HBM / L2 to LDS as they are (16 bytes per lane and request: HALF the bytes of the fp32 rows in bf16 mode, no
    // conversion) and sums the values and their squares on the way.
    const int d = a.d;
    constexpr int E16 = 16 / (int)sizeof(WT);                   // elements per 16-byte unit
    const int npl = (d * (int)sizeof(WT)) >> 10;                // units per lane and row: d sizeof(WT) / 16 / 64 = 1..8
    const int rowA = (NTW == 1) ? wave : wv;
    const int rowB = (NTW == 1) ? wave + 4 : wv + 8;
    constexpr bool TWO_ROWS = (NTW == 1) || R2;
    const char* qA = reinterpret_cast<const char*>(a.x_in) + (long)min(rowA, n_rows - 1) * d * (long)sizeof(WT);
    const char* qB = reinterpret_cast<const char*>(a.x_in) + (long)min(rowB, n_rows - 1) * d * (long)sizeof(WT);
    const float cA = a.row_mu[min(rowA, n_rows - 1)], cB = a.row_mu[min(rowB, n_rows - 1)];
    uint4 xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int u = (min(j, npl - 1) * 64 + lane) * 16;
      xa[j] = *reinterpret_cast<const uint4*>(qA + u);
      if constexpr (TWO_ROWS) xb[j] = *reinterpret_cast<const uint4*>(qB + u);
    }
    VC_ISSUE_WEIGHTS(0);
    VC_BURST_OUT();
#define VC_LNQ_ROW(X, r, c_)                                                                     \
    if ((r) < n_rows) {                                                                          \
      char* xr_ = xl + (size_t)(r) * xs;                                                         \
      float s1_ = 0.f, s2_ = 0.f;                                                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
        if (j < npl) {                                                                           \
          *reinterpret_cast<uint4*>(xr_ + (j * 64 + lane) * 16) = X[j];                          \
          float f_[E16];                                                                         \
          if constexpr (sizeof(WT) == 2) {                                                       \
            f_[0] = __uint_as_float(X[j].x << 16); f_[1] = __uint_as_float(X[j].x & 0xffff0000u); \
            f_[2] = __uint_as_float(X[j].y << 16); f_[3] = __uint_as_float(X[j].y & 0xffff0000u); \
            f_[4] = __uint_as_float(X[j].z << 16); f_[5] = __uint_as_float(X[j].z & 0xffff0000u); \
            f_[6] = __uint_as_float(X[j].w << 16); f_[7] = __uint_as_float(X[j].w & 0xffff0000u); \
          } else {                                                                               \
            f_[0] = __uint_as_float(X[j].x); f_[1] = __uint_as_float(X[j].y);                    \
            f_[2] = __uint_as_float(X[j].z); f_[3] = __uint_as_float(X[j].w);                    \
          }                                                                                      \
          _Pragma("unroll") for (int e_ = 0; e_ < E16; e_ += 4) {                                \
            s1_ += (f_[e_] + f_[e_ + 1]) + (f_[e_ + 2] + f_[e_ + 3]);                            \
            s2_ += (f_[e_] * f_[e_] + f_[e_ + 1] * f_[e_ + 1]) + (f_[e_ + 2] * f_[e_ + 2] + f_[e_ + 3] * f_[e_ + 3]); \
          }                                                                                      \
        }                                                                                        \
      s1_ = wave_sum(s1_);                                                                       \
      s2_ = wave_sum(s2_);                                                                       \
      if (lane == 0) {                                                                           \
        *reinterpret_cast<float4*>(stat + (r) * 8) = make_float4(s1_, s2_, 0.f, 0.f);            \
        *reinterpret_cast<float4*>(stat + (r) * 8 + 4) = make_float4(0.f, 0.f, 0.f, 0.f);        \
        if (a.row_mu_out && blockIdx.x == 0 && grp == 0) a.row_mu_out[r] = (c_) + s1_ * (1.0f / (float)d);   \
      }                                                                                          \
    }
    VC_LNQ_ROW(xa, rowA, cA);
    if constexpr (TWO_ROWS) { VC_LNQ_ROW(xb, rowB, cB); }
#undef VC_LNQ_ROW
  } else if constexpr (PRO == PRO_PLAIN) {
    // X rows copied as 16-byte units, flat index = row * upr + unit.  The first NB*256 units are
    // requested ahead of the weight burst (clamped, unconditional), the rest behind it.
    const int upr = kblk * (int)sizeof(WT) / 16;   // 16-byte units per row
    const int total = n_rows * upr;
    const char* src = reinterpret_cast<const char*>(a.x_in) + ((long)grp * a.x_group_stride + k0) * (long)sizeof(WT);
    const long rstride = (long)a.x_ld * (long)sizeof(WT);
    const int sh = a.x_upr_shift;                  // log2(upr) when upr is a power of two, else -1
#define VC_X_SPLIT(idx_, r_, u_)                                                                 \
    const int r_ = (sh >= 0) ? ((idx_) >> sh) : ((idx_) / upr);                                  \
    const int u_ = (idx_) - r_ * upr;
#define VC_X_LOAD(j, dst)                                                                        \
    {                                                                                            \
      const int i_ = min((j) * 256 + tid, total - 1);                                            \
      VC_X_SPLIT(i_, rr_, uu_)                                                                   \
      dst = *reinterpret_cast<const uint4*>(src + (long)rr_ * rstride + (long)uu_ * 16);         \
    }
#define VC_X_STORE(j, val)                                                                       \
    {                                                                                            \
      const int i_ = (j) * 256 + tid;                                                            \
      if (i_ < total) {                                                                          \
        VC_X_SPLIT(i_, rr_, uu_)                                                                 \
        *reinterpret_cast<uint4*>(xl + (size_t)rr_ * xs + (size_t)uu_ * 16) = val;               \
      }                                                                                          \
    }
    // (explicit scalars: an indexed array here is demoted to scratch memory by the compiler)
#define VC_X_LATE(NB)                                                                            \
      for (int i_ = NB * 256 + tid; i_ < total; i_ += 256) {                                     \
        VC_X_SPLIT(i_, rr_, uu_)                                                                 \
        *reinterpret_cast<uint4*>(xl + (size_t)rr_ * xs + (size_t)uu_ * 16) =                    \
            *reinterpret_cast<const uint4*>(src + (long)rr_ * rstride + (long)uu_ * 16);         \
      }
    if (total <= 4 * 256) {        // one decode row (up to 4 KB x 4 slices)
      uint4 x0, x1, x2, x3;
      VC_X_LOAD(0, x0); VC_X_LOAD(1, x1); VC_X_LOAD(2, x2); VC_X_LOAD(3, x3);
      VC_ISSUE_WEIGHTS(0);
      VC_BURST_OUT();
      VC_X_STORE(0, x0); VC_X_STORE(1, x1); VC_X_STORE(2, x2); VC_X_STORE(3, x3);
    } else {                       // batched decode / span switch: 16 rows of 4 KB in one round trip
      uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
      VC_X_LOAD(0, x0); VC_X_LOAD(1, x1); VC_X_LOAD(2, x2); VC_X_LOAD(3, x3);
      VC_X_LOAD(4, x4); VC_X_LOAD(5, x5); VC_X_LOAD(6, x6); VC_X_LOAD(7, x7);
      VC_X_LOAD(8, x8); VC_X_LOAD(9, x9); VC_X_LOAD(10, x10); VC_X_LOAD(11, x11);
      VC_X_LOAD(12, x12); VC_X_LOAD(13, x13); VC_X_LOAD(14, x14); VC_X_LOAD(15, x15);
      VC_ISSUE_WEIGHTS(0);
      VC_BURST_OUT();
      VC_X_STORE(0, x0); VC_X_STORE(1, x1); VC_X_STORE(2, x2); VC_X_STORE(3, x3);
      VC_X_STORE(4, x4); VC_X_STORE(5, x5); VC_X_STORE(6, x6); VC_X_STORE(7, x7);
      VC_X_STORE(8, x8); VC_X_STORE(9, x9); VC_X_STORE(10, x10); VC_X_STORE(11, x11);
      VC_X_STORE(12, x12); VC_X_STORE(13, x13); VC_X_STORE(14, x14); VC_X_STORE(15, x15);
      VC_X_LATE(16)
    }
#undef VC_X_LATE
#undef VC_X_LOAD
#undef VC_X_STORE
#undef VC_X_SPLIT
  } else {  // PRO_ATT: merge the split-S partials of the decode attention (softmax denominators)
    // item = (row, 4 columns of one head) with NS partials of (max, sum, acc[4]); a thread carries IB
    // items at a time, NS * IB = 8 (one decode row: 8 splits x 1 item; batched rows: fewer splits, more
    // items).  The first batch is requested before the weight burst, the exp/scale arithmetic runs
    // behind it; further batches are double-buffered.
    const int q4 = kblk >> 2;
    const int n_items = n_rows * q4;
    const int qsh = a.att_q4_shift;                 // log2(q4) when q4 is a power of two, else -1
#define VC_ATT_LOAD(S, NS, IB, base)                                                             \
    _Pragma("unroll") for (int ib_ = 0; ib_ < IB; ++ib_) {                                       \
      const int idx_ = (base) + ib_ * 256 + tid;                                                 \
      on##S[ib_] = idx_ < n_items;                                                               \
      const int ic_ = on##S[ib_] ? idx_ : 0;                                                     \
      r##S[ib_] = (qsh >= 0) ? (ic_ >> qsh) : (ic_ / q4);                                        \
      c##S[ib_] = k0 + (ic_ - r##S[ib_] * q4) * 4;                                               \
      const int h_ = c##S[ib_] >> a.hd_shift, e_ = c##S[ib_] & (a.hd - 1);                      \
      const float2* ml_ = reinterpret_cast<const float2*>(a.att_ml) + (long)(r##S[ib_] * a.H + h_) * a.nsplit; \
      const float* op_ = a.att_o + ((long)(r##S[ib_] * a.H + h_) * a.nsplit) * a.hd + e_;        \
      _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                        \
        const int se_ = (s_ < a.nsplit) ? s_ : 0;                                                \
        ml##S[ib_][s_] = ml_[se_];                                                               \
        os##S[ib_][s_] = *reinterpret_cast<const float4*>(op_ + (long)se_ * a.hd);               \
      }                                                                                          \
    }
#define VC_ATT_FINISH(S, NS, IB)                                                                 \
    _Pragma("unroll") for (int ib_ = 0; ib_ < IB; ++ib_) {                                       \
      float M_ = -INFINITY;                                                                      \
      _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                        \
        ml##S[ib_][s_].x = (s_ < a.nsplit) ? ml##S[ib_][s_].x : -INFINITY;                       \
        M_ = fmaxf(M_, ml##S[ib_][s_].x);                                                        \
      }                                                                                          \
      float L_ = 0.f;                                                                            \
      f32x4 o_ = {0.f, 0.f, 0.f, 0.f};                                                           \
      _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                        \
        const float w_ = (ml##S[ib_][s_].x == -INFINITY) ? 0.f : expf(ml##S[ib_][s_].x - M_);    \
        L_ += w_ * ml##S[ib_][s_].y;                                                             \
        o_[0] += w_ * os##S[ib_][s_].x; o_[1] += w_ * os##S[ib_][s_].y;                          \
        o_[2] += w_ * os##S[ib_][s_].z; o_[3] += w_ * os##S[ib_][s_].w;                          \
      }                                                                                          \
      const float inv_ = (L_ > 0.f) ? 1.0f / L_ : 0.f;                                           \
      o_[0] *= inv_; o_[1] *= inv_; o_[2] *= inv_; o_[3] *= inv_;                                \
      if (on##S[ib_]) store4(reinterpret_cast<WT*>(xl + (size_t)r##S[ib_] * xs) + (c##S[ib_] - k0), o_); \
    }
#define VC_ATT_PATH(NS, IB)                                                                      \
    {                                                                                            \
      float2 mlA[IB][NS], mlB[IB][NS];                                                           \
      float4 osA[IB][NS], osB[IB][NS];                                                           \
      int rA[IB], cA[IB], rB[IB], cB[IB];                                                        \
      bool onA[IB], onB[IB];                                                                     \
      VC_ATT_LOAD(A, NS, IB, 0);                                                                 \
      VC_ISSUE_WEIGHTS(0);                                                                       \
      VC_BURST_OUT();                                                                            \
      if (n_items <= IB * 256) {                                                                 \
        VC_ATT_FINISH(A, NS, IB);                                                                \
      } else {                                                                                   \
        int base = 0;                                                                            \
        for (;;) {                                                                               \
          VC_ATT_LOAD(B, NS, IB, base + IB * 256);   /* past the end: clamped loads, nothing stored */ \
          VC_ATT_FINISH(A, NS, IB);                                                              \
          if ((base += IB * 256) >= n_items) break;                                              \
          VC_ATT_LOAD(A, NS, IB, base + IB * 256);                                               \
          VC_ATT_FINISH(B, NS, IB);                                                              \
          if ((base += IB * 256) >= n_items) break;                                              \
        }                                                                                        \
      }                                                                                          \
    }
    if (a.nsplit > 4) VC_ATT_PATH(8, 1)
    else if (a.nsplit > 2) VC_ATT_PATH(4, 2)
    else VC_ATT_PATH(2, 4)
#undef VC_ATT_PATH
#undef VC_ATT_LOAD
#undef VC_ATT_FINISH
  }
#undef VC_BURST_OUT
  VC_KTS(2);
  __syncthreads();
  VC_KTS(3);

  // main loop: one ds_read_b128 + one MFMA per 1 KiB weight burst
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mrow = (m < a.r_lds) ? m : 0;
  const char* xrow = xl + (size_t)mrow * xs + (size_t)(lane >> 4) * 16;
  for (int c = 0; c < a.nchunk; ++c) {
    const int ktl = (c * 4 + wave) * KTW;
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
      const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)(ktl + i) * 64);
      if constexpr (TH < 16) {
        if (!wvalid) wf[i] = make_uint4(0u, 0u, 0u, 0u);
      }
      acc = mfma_frag(wf[i], xf, acc, (WT*)nullptr);
    }
    if (c + 1 < a.nchunk) VC_ISSUE_WEIGHTS(c + 1);   // refill the same registers; co-resident blocks cover the latency
  }
#undef VC_ISSUE_WEIGHTS
  VC_KTS(4);

  // 4-way in-block K reduction, then the epilogue on wave 0
  red[wave * 64 + lane] = acc;
  __syncthreads();
  VC_KTS(5);
  if (wave == 0 && m < n_rows && nvalid) {
    {
      const f32x4 a1 = red[64 + lane], a2 = red[128 + lane], a3 = red[192 + lane];
      acc = (acc + a1) + (a2 + a3);
    }
    if constexpr (PRO == PRO_LN || LNW) {   // LayerNorm fold on the centred row xc: y = rstd * (W'xc - mean(xc) * rowsum(W')) [+ cb in the epilogue]
      const float4 sa = *reinterpret_cast<const float4*>(stat + m * 8), sb = *reinterpret_cast<const float4*>(stat + m * 8 + 4);
      const float inv_d = 1.0f / (float)a.d;
      const float mean = ((sa.x + sa.z) + (sb.x + sb.z)) * inv_d;
      const float var = fmaxf(((sa.y + sa.w) + (sb.y + sb.w)) * inv_d - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);      // eps 1e-5 (transformer.py:30)
      acc[0] = rstd * (acc[0] - mean * ewg.x); acc[1] = rstd * (acc[1] - mean * ewg.y);
      acc[2] = rstd * (acc[2] - mean * ewg.z); acc[3] = rstd * (acc[3] - mean * ewg.w);
    }
    gemm_epilogue<WT, EPI>(a, acc, m, n, ks, grp, (int)gridDim.z, eb, epos, eseq);
  }
  VC_KTS(6);
  VC_KTS_FLUSH();
}

// ------------------------------------------------------------------ finished-row producers: decode passes of 2..8 rows (9..16: also rows_gemm_fr2_k below)
// A several-row decode step (config 5's per-GPU share: 8 utterances) spent 18 % of its kernel time in two 8-workgroup
// LayerNorm launches per layer: the out-projection and the FFN down-projection leave split-K slabs, the sum of h + bias +
// 4 slabs per row is too much to redo in every consumer workgroup (320 KB each), so a launch of its own did it once.  Here
// those two producers finish their rows instead: K is NOT split across workgroups - a workgroup owns 8 output channels
// (VC_TH_RES: d/8 = 256 workgroups at d = 2048) over the whole K, its 8 waves stream K/8 each in ONE burst (the FFN
// down-projection at d = 2048: 32 fragments of 512 B per wave, 128 KB per workgroup, everything requested at once), X of
// all rows is staged in LDS (8 rows x 8192 x 2 B = 128 KB), and the epilogue adds the residual and the bias and writes the
// row once.  The consumers then fold the LayerNorm of one FINISHED row per wave (PRO_LNW): no slabs, no LayerNorm launch.
// PRO_PLAIN: X = activations (FFN down); PRO_ATT: X = merge of the attention's split partials of ALL heads (out-projection;
// NS = partials per item the loads are unrolled for, rows x NS <= 16 so that one batch of loads covers every item).
// Bounded by rows x K x sizeof(WT) <= 128 KB of LDS: 8 rows in bf16, 4 in the exact fp32 mode.
template <typename WT, int KTW, int PRO, int NS, bool P16 = false>
__global__ __launch_bounds__(64 * VC_FR_WAVES) void rows_gemm_fr_k(const GemmArgs a) {
  static_assert(!P16 || (PRO == PRO_ATT && sizeof(WT) == 2), "bf16 attention partials: the merging out-projection in bf16 mode");
  using T = WTr<WT>;
  constexpr int NW = VC_FR_WAVES, NTHR = 64 * NW, TH = VC_TH_RES, SPT = 4 * TH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x;
  const int n_rows = a.n_rows;                    // 2..16 as far as rows x K fits LDS and 16 staging units per thread (host contract)
  const int K = a.K;
  const int xs = K * (int)sizeof(WT) + 16;        // LDS row stride (+16: rotate bank slots)
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)a.r_lds * xs);
  const int active = *a.n_active;
  const int m = lane & 15, kg = lane >> 4;
  const bool nvalid = 4 * kg < TH;
  const int n = nt * TH + (nvalid ? 4 * kg : 0);
  const int wslot = kg * TH + min(m, TH - 1);
  // epilogue operands first (a wave's loads return in order): the residual and the bias of the lane's four channels
  const float4 eres = *reinterpret_cast<const float4*>(a.h_in + (long)((m < n_rows) ? m : 0) * a.d + n);
  const float4 eb = *reinterpret_cast<const float4*>(a.bias + n);
  const float cmu = a.row_mu[(m < n_rows) ? m : 0];                      // centring constant of the row's copy in the compute dtype (hq_out)
  const uint4* wbase = a.Wp + ((long)nt * a.KT + wave * KTW) * SPT;      // wave-uniform
  uint4 wf[KTW];
#define VC_FR_WEIGHTS(c_)                                                                        \
  {                                                                                              \
    const uint4* wb_ = wbase + (long)(c_) * (NW * KTW * SPT);                                     \
    _Pragma("unroll") for (int i = 0; i < KTW; ++i)                                              \
      wf[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb_ + i * SPT + wslot))); \
  }
  if constexpr (PRO == PRO_PLAIN) {
    // X rows as 16-byte units, flat index = row * upr + unit; <= 16 units per thread (rows x K x sizeof <= 128 KB)
    const int upr = K * (int)sizeof(WT) / 16;
    const int total = n_rows * upr;
    const char* src = reinterpret_cast<const char*>(a.x_in);
    const long rstride = (long)a.x_ld * (long)sizeof(WT);
    const int sh = a.x_upr_shift;
    // (explicit scalars: an indexed array is demoted to scratch memory by the compiler)
    uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
#define VC_FRX_LOAD(j, dst)                                                                      \
    {                                                                                            \
      const int i_ = min((j) * NTHR + tid, total - 1);                                           \
      const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr);                                        \
      dst = *reinterpret_cast<const uint4*>(src + (long)r_ * rstride + (long)(i_ - r_ * upr) * 16); \
    }
#define VC_FRX_STORE(j, val)                                                                     \
    {                                                                                            \
      const int i_ = (j) * NTHR + tid;                                                           \
      if (i_ < total) {                                                                          \
        const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr);                                      \
        *reinterpret_cast<uint4*>(xl + (size_t)r_ * xs + (size_t)(i_ - r_ * upr) * 16) = val;    \
      }                                                                                          \
    }
    VC_FRX_LOAD(0, x0) VC_FRX_LOAD(1, x1) VC_FRX_LOAD(2, x2) VC_FRX_LOAD(3, x3)
    VC_FRX_LOAD(4, x4) VC_FRX_LOAD(5, x5) VC_FRX_LOAD(6, x6) VC_FRX_LOAD(7, x7)
    VC_FRX_LOAD(8, x8) VC_FRX_LOAD(9, x9) VC_FRX_LOAD(10, x10) VC_FRX_LOAD(11, x11)
    VC_FRX_LOAD(12, x12) VC_FRX_LOAD(13, x13) VC_FRX_LOAD(14, x14) VC_FRX_LOAD(15, x15)
    VC_FR_WEIGHTS(0);
    __builtin_amdgcn_sched_barrier(0);
    if (active == 0) return;
    VC_FRX_STORE(0, x0) VC_FRX_STORE(1, x1) VC_FRX_STORE(2, x2) VC_FRX_STORE(3, x3)
    VC_FRX_STORE(4, x4) VC_FRX_STORE(5, x5) VC_FRX_STORE(6, x6) VC_FRX_STORE(7, x7)
    VC_FRX_STORE(8, x8) VC_FRX_STORE(9, x9) VC_FRX_STORE(10, x10) VC_FRX_STORE(11, x11)
    VC_FRX_STORE(12, x12) VC_FRX_STORE(13, x13) VC_FRX_STORE(14, x14) VC_FRX_STORE(15, x15)
#undef VC_FRX_LOAD
#undef VC_FRX_STORE
  } else if constexpr (P16) {   // PRO_ATT on bf16 partials (round 6): item = (row, EIGHT columns of one head) - one 16-byte load per partial - NI items per thread
    constexpr int NI = 8 / NS;
    const int q8 = K >> 3;
    const int n_items = n_rows * q8;              // <= NI * NTHR (rows x NS <= 16, K <= 2048: host contract)
    const int qsh = a.att_q4_shift - 1;           // log2(K / 8) when K / 4 is a power of two, else < 0
    float2 ml[NI][NS];
    uint4 os[NI][NS];
    int ir[NI], ic[NI];
    bool on[NI];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      const int idx = ib * NTHR + tid;
      on[ib] = idx < n_items;
      const int i_ = on[ib] ? idx : 0;
      ir[ib] = (a.att_q4_shift >= 1) ? (i_ >> qsh) : (i_ / q8);
      ic[ib] = (i_ - ir[ib] * q8) * 8;
      const int h_ = ic[ib] >> a.hd_shift, e_ = ic[ib] & (a.hd - 1);
      const float2* mp = reinterpret_cast<const float2*>(a.att_ml) + (long)(ir[ib] * a.H + h_) * a.nsplit;
      const uint16_t* op = reinterpret_cast<const uint16_t*>(a.att_o) + ((long)(ir[ib] * a.H + h_) * a.nsplit) * a.hd + e_;
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const int se = (sp < a.nsplit) ? sp : 0;
        ml[ib][sp] = mp[se];
        os[ib][sp] = *reinterpret_cast<const uint4*>(op + (long)se * a.hd);
      }
    }
    VC_FR_WEIGHTS(0);
    __builtin_amdgcn_sched_barrier(0);
    if (active == 0) return;
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      float M = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        ml[ib][sp].x = (sp < a.nsplit) ? ml[ib][sp].x : -INFINITY;
        M = fmaxf(M, ml[ib][sp].x);
      }
      float L = 0.f;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const float w = (ml[ib][sp].x == -INFINITY) ? 0.f : expf(ml[ib][sp].x - M);
        L += w * ml[ib][sp].y;
        const uint4 u = os[ib][sp];
        o[0] += w * __uint_as_float(u.x << 16); o[1] += w * __uint_as_float(u.x & 0xffff0000u);
        o[2] += w * __uint_as_float(u.y << 16); o[3] += w * __uint_as_float(u.y & 0xffff0000u);
        o[4] += w * __uint_as_float(u.z << 16); o[5] += w * __uint_as_float(u.z & 0xffff0000u);
        o[6] += w * __uint_as_float(u.w << 16); o[7] += w * __uint_as_float(u.w & 0xffff0000u);
      }
      const float inv = (L > 0.f) ? 1.0f / L : 0.f;
      uint4 pk;
      pk.x = pack_bf16x2(o[0] * inv, o[1] * inv); pk.y = pack_bf16x2(o[2] * inv, o[3] * inv);
      pk.z = pack_bf16x2(o[4] * inv, o[5] * inv); pk.w = pack_bf16x2(o[6] * inv, o[7] * inv);
      if (on[ib]) *reinterpret_cast<uint4*>(reinterpret_cast<WT*>(xl + (size_t)ir[ib] * xs) + ic[ib]) = pk;
    }
  } else {   // PRO_ATT: item = (row, 4 columns of one head) with a.nsplit <= NS partials of (max, sum, acc[4]); NI items per thread
    constexpr int NI = 16 / NS;
    const int q4 = K >> 2;
    const int n_items = n_rows * q4;              // <= NI * NTHR (rows x NS <= 16, K <= 2048: host contract)
    const int qsh = a.att_q4_shift;
    float2 ml[NI][NS];
    float4 os[NI][NS];
    int ir[NI], ic[NI];
    bool on[NI];
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      const int idx = ib * NTHR + tid;
      on[ib] = idx < n_items;
      const int i_ = on[ib] ? idx : 0;
      ir[ib] = (qsh >= 0) ? (i_ >> qsh) : (i_ / q4);
      ic[ib] = (i_ - ir[ib] * q4) * 4;
      const int h_ = ic[ib] >> a.hd_shift, e_ = ic[ib] & (a.hd - 1);
      const float2* mp = reinterpret_cast<const float2*>(a.att_ml) + (long)(ir[ib] * a.H + h_) * a.nsplit;
      const float* op = a.att_o + ((long)(ir[ib] * a.H + h_) * a.nsplit) * a.hd + e_;
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const int se = (sp < a.nsplit) ? sp : 0;
        ml[ib][sp] = mp[se];
        os[ib][sp] = *reinterpret_cast<const float4*>(op + (long)se * a.hd);
      }
    }
    VC_FR_WEIGHTS(0);
    __builtin_amdgcn_sched_barrier(0);
    if (active == 0) return;
#pragma unroll
    for (int ib = 0; ib < NI; ++ib) {
      float M = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        ml[ib][sp].x = (sp < a.nsplit) ? ml[ib][sp].x : -INFINITY;
        M = fmaxf(M, ml[ib][sp].x);
      }
      float L = 0.f;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) {
        const float w = (ml[ib][sp].x == -INFINITY) ? 0.f : expf(ml[ib][sp].x - M);
        L += w * ml[ib][sp].y;
        o[0] += w * os[ib][sp].x; o[1] += w * os[ib][sp].y; o[2] += w * os[ib][sp].z; o[3] += w * os[ib][sp].w;
      }
      const float inv = (L > 0.f) ? 1.0f / L : 0.f;
      o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
      if (on[ib]) store4(reinterpret_cast<WT*>(xl + (size_t)ir[ib] * xs) + ic[ib], o);
    }
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mrow = (m < a.r_lds) ? m : 0;
  const char* xrow = xl + (size_t)mrow * xs + (size_t)kg * 16;
  for (int c = 0; c < a.nchunk; ++c) {
    const int ktl = (c * NW + wave) * KTW;
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
      const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)(ktl + i) * 64);
      acc = mfma_frag(wf[i], xf, acc, (WT*)nullptr);     // (fragment rows 8..15 repeat slot 7: they only reach D rows nobody reads)
    }
    if (c + 1 < a.nchunk) VC_FR_WEIGHTS(c + 1);          // exact fp32 mode at d = 2048: the registers hold half of a wave's share
  }
#undef VC_FR_WEIGHTS
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && m < n_rows && nvalid) {
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += red[w * 64 + lane];
    const f32x4 o = {eres.x + eb.x + acc[0], eres.y + eb.y + acc[1], eres.z + eb.z + acc[2], eres.w + eb.w + acc[3]};
    store4(a.h_out + (long)m * a.d + n, o);
    if (a.hq_out) { const f32x4 q = {o[0] - cmu, o[1] - cmu, o[2] - cmu, o[3] - cmu}; store4(reinterpret_cast<WT*>(a.hq_out) + (long)m * a.d + n, q); }
  }
}

// The FFN down-projection of 9..16 finished rows: X (rows x 4d elements) no longer fits the LDS of one workgroup, so K goes
// through it in TWO halves - the same LDS buffer twice - while the wave's weight fragments of BOTH halves are requested at the
// very top (2 x KTW fragments in registers: the stream never waits for the staging).  The second half's X rows are requested
// as soon as the first half has been parked (they return behind the weights: a wave's loads come back in order) and parked
// once every wave has finished reading the first half.  Same sums in the same order as rows_gemm_fr_k.
template <typename WT, int KTW>
__global__ __launch_bounds__(64 * VC_FR_WAVES) void rows_gemm_fr2_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NW = VC_FR_WAVES, NTHR = 64 * NW, TH = VC_TH_RES, SPT = 4 * TH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x;
  const int n_rows = a.n_rows;                    // <= 16 (host contract)
  const int Kh = a.K >> 1;                        // K elements per half = NW * KTW k-tiles
  const int xs = Kh * (int)sizeof(WT) + 16;
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)a.r_lds * xs);
  const int active = *a.n_active;
  const int m = lane & 15, kg = lane >> 4;
  const bool nvalid = 4 * kg < TH;
  const int n = nt * TH + (nvalid ? 4 * kg : 0);
  const int wslot = kg * TH + min(m, TH - 1);
  const float4 eres = *reinterpret_cast<const float4*>(a.h_in + (long)((m < n_rows) ? m : 0) * a.d + n);
  const float4 eb = *reinterpret_cast<const float4*>(a.bias + n);
  const float cmu = a.row_mu[(m < n_rows) ? m : 0];
  const uint4* wbase = a.Wp + ((long)nt * a.KT + wave * KTW) * SPT;      // wave-uniform; the second half is NW * KTW k-tiles further
  uint4 wfa[KTW], wfb[KTW];
  const int upr = Kh * (int)sizeof(WT) / 16;       // 16-byte units per row and half; rows x upr <= 16 x 512 (host contract)
  const int total = n_rows * upr;
  const char* src = reinterpret_cast<const char*>(a.x_in);
  const long rstride = (long)a.x_ld * (long)sizeof(WT);
  const long hoff = (long)Kh * (long)sizeof(WT);
  const int sh = a.x_upr_shift;
  uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
#define VC_FR2_LOAD(j, dst, off_)                                                                \
    {                                                                                            \
      const int i_ = min((j) * NTHR + tid, total - 1);                                           \
      const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr);                                        \
      dst = *reinterpret_cast<const uint4*>(src + (off_) + (long)r_ * rstride + (long)(i_ - r_ * upr) * 16); \
    }
#define VC_FR2_STORE(j, val)                                                                     \
    {                                                                                            \
      const int i_ = (j) * NTHR + tid;                                                           \
      if (i_ < total) {                                                                          \
        const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr);                                      \
        *reinterpret_cast<uint4*>(xl + (size_t)r_ * xs + (size_t)(i_ - r_ * upr) * 16) = val;    \
      }                                                                                          \
    }
#define VC_FR2_LOAD_ALL(off_)                                                                    \
    VC_FR2_LOAD(0, x0, off_) VC_FR2_LOAD(1, x1, off_) VC_FR2_LOAD(2, x2, off_) VC_FR2_LOAD(3, x3, off_)         \
    VC_FR2_LOAD(4, x4, off_) VC_FR2_LOAD(5, x5, off_) VC_FR2_LOAD(6, x6, off_) VC_FR2_LOAD(7, x7, off_)         \
    VC_FR2_LOAD(8, x8, off_) VC_FR2_LOAD(9, x9, off_) VC_FR2_LOAD(10, x10, off_) VC_FR2_LOAD(11, x11, off_)     \
    VC_FR2_LOAD(12, x12, off_) VC_FR2_LOAD(13, x13, off_) VC_FR2_LOAD(14, x14, off_) VC_FR2_LOAD(15, x15, off_)
#define VC_FR2_STORE_ALL()                                                                       \
    VC_FR2_STORE(0, x0) VC_FR2_STORE(1, x1) VC_FR2_STORE(2, x2) VC_FR2_STORE(3, x3)               \
    VC_FR2_STORE(4, x4) VC_FR2_STORE(5, x5) VC_FR2_STORE(6, x6) VC_FR2_STORE(7, x7)               \
    VC_FR2_STORE(8, x8) VC_FR2_STORE(9, x9) VC_FR2_STORE(10, x10) VC_FR2_STORE(11, x11)           \
    VC_FR2_STORE(12, x12) VC_FR2_STORE(13, x13) VC_FR2_STORE(14, x14) VC_FR2_STORE(15, x15)
  VC_FR2_LOAD_ALL(0L)
#pragma unroll
  for (int i = 0; i < KTW; ++i)
    wfa[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + i * SPT + wslot)));
#pragma unroll
  for (int i = 0; i < KTW; ++i)
    wfb[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)(NW * KTW + i) * SPT + wslot)));
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  VC_FR2_STORE_ALL()
  __syncthreads();
  VC_FR2_LOAD_ALL(hoff)                           // second half of the rows: on its way during the first half's MFMAs
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mrow = (m < a.r_lds) ? m : 0;
  const char* xrow = xl + (size_t)mrow * xs + (size_t)kg * 16 + (size_t)(wave * KTW) * 64;
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)i * 64);
    acc = mfma_frag(wfa[i], xf, acc, (WT*)nullptr);
  }
  __syncthreads();                                // every wave has read the first half
  VC_FR2_STORE_ALL()
  __syncthreads();
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)i * 64);
    acc = mfma_frag(wfb[i], xf, acc, (WT*)nullptr);
  }
#undef VC_FR2_LOAD
#undef VC_FR2_STORE
#undef VC_FR2_LOAD_ALL
#undef VC_FR2_STORE_ALL
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && m < n_rows && nvalid) {
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += red[w * 64 + lane];
    const f32x4 o = {eres.x + eb.x + acc[0], eres.y + eb.y + acc[1], eres.z + eb.z + acc[2], eres.w + eb.w + acc[3]};
    store4(a.h_out + (long)m * a.d + n, o);
    if (a.hq_out) { const f32x4 q = {o[0] - cmu, o[1] - cmu, o[2] - cmu, o[3] - cmu}; store4(reinterpret_cast<WT*>(a.hq_out) + (long)m * a.d + n, q); }
  }
}
template <typename WT, int KTW>
static hipError_t launch_fr2(const GemmArgs& a, size_t lds, hipStream_t s) {
  auto kern = rows_gemm_fr2_k<WT, KTW>;
  static size_t granted[16] = {0};     // per instantiation and device
  int dev = 0;
  if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
  if (lds > 64 * 1024 && dev >= 0 && dev < 16 && lds > granted[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    granted[dev] = lds;
  }
  ++vc_launch_counts[VC_LC_ROWS_GEMM_FR];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(64 * VC_FR_WAVES), lds, s, a);
  return hipGetLastError();
}

template <typename WT, int KTW, int PRO, int NS, bool P16 = false>
static hipError_t launch_fr_n(const GemmArgs& a, size_t lds, hipStream_t s) {
  auto kern = rows_gemm_fr_k<WT, KTW, PRO, NS, P16>;
  if (lds > 64 * 1024) {
    static size_t granted[16] = {0};   // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && lds > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = lds;
    }
  }
  ++vc_launch_counts[VC_LC_ROWS_GEMM_FR];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(64 * VC_FR_WAVES), lds, s, a);
  return hipGetLastError();
}
template <typename WT, int KTW>
static hipError_t launch_fr_ktw(const GemmArgs& a, int pro, size_t lds, hipStream_t s) {
  if (pro == PRO_PLAIN) return launch_fr_n<WT, KTW, PRO_PLAIN, 1>(a, lds, s);
  if constexpr (KTW <= 16) {           // the out-projection: K = d <= 2048
    if (pro == PRO_ATT) {
      if constexpr (sizeof(WT) == 2) {     // bf16 mode: the attention launch in front may have left bf16 partials (GemmArgs.att_p16)
        if (a.att_p16) {
          if (a.nsplit > 4) return launch_fr_n<WT, KTW, PRO_ATT, 8, true>(a, lds, s);
          if (a.nsplit > 2) return launch_fr_n<WT, KTW, PRO_ATT, 4, true>(a, lds, s);
          return launch_fr_n<WT, KTW, PRO_ATT, 2, true>(a, lds, s);
        }
      }
      if (a.nsplit > 4) return launch_fr_n<WT, KTW, PRO_ATT, 8>(a, lds, s);
      if (a.nsplit > 2) return launch_fr_n<WT, KTW, PRO_ATT, 4>(a, lds, s);
      return launch_fr_n<WT, KTW, PRO_ATT, 2>(a, lds, s);
    }
  }
  return hipErrorInvalidValue;
}
size_t vc_gemm_fr_lds_bytes(int rows, int K, int dtype) {
  const size_t esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  return (size_t)rows * ((size_t)K * esz + 16) + (size_t)VC_FR_WAVES * 64 * sizeof(f32x4);
}
// A finished-row producer pass: out-projection (PRO_ATT) or FFN down-projection (PRO_PLAIN) of 2..VC_FR_MAX_ROWS rows; a.Wp
// is the 8-channel-tile image of the matrix, a.h_in the residual rows, a.bias the layer's bias, a.h_out the finished rows.
// Which form a finished-row producer pass takes - 0: none (the launcher refuses it), 1: X in LDS in one piece (rows_gemm_fr_k),
// 2: K through LDS in two halves (rows_gemm_fr2_k, plain prologue only).  The ONE statement of the constraints: the launcher
// below and the engine's pass planning (vc_engine.hip fr_max_rows, checked for every model width by tests/test_plan_cpu.py
// through vc_debug_plan) both go through it.
int vc_gemm_fr_form(int rows, int N, int K, int dtype, int pro, int nsplit) {
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  if (rows < 1 || rows > VC_ROWS || N % VC_TH_RES != 0 || K % KW != 0) return 0;
  const int KT = K / KW;
  const long upr = (long)K * esz / 16, cap = 16L * 64 * VC_FR_WAVES;        // 16-byte units of X per row; units a workgroup stages per piece
  const size_t lds_max = 160 * 1024;
  if (pro == PRO_ATT) {        // the out-projection merging split partials: K = d <= 2048 (a wave's share <= 16 fragments), one batch of loads
    const int ns = nsplit > 4 ? 8 : nsplit > 2 ? 4 : 2;
    if (nsplit < 1 || nsplit > 8 || rows * ns > 16 || (long)rows * (K / 4) > 16L / ns * 64 * VC_FR_WAVES) return 0;
    if (KT % VC_FR_WAVES != 0 || KT / VC_FR_WAVES > 16) return 0;
    return vc_gemm_fr_lds_bytes(rows, K, dtype) <= lds_max ? 1 : 0;
  }
  if (pro != PRO_PLAIN) return 0;
  if (KT % VC_FR_WAVES == 0 && rows * upr <= cap && vc_gemm_fr_lds_bytes(rows, K, dtype) <= lds_max) return 1;
  const int k2 = KT / (2 * VC_FR_WAVES);
  if (KT % (2 * VC_FR_WAVES) == 0 && (k2 == 2 || k2 == 4 || k2 == 8 || k2 == 16) && rows * (upr / 2) <= cap &&
      vc_gemm_fr_lds_bytes(rows, K / 2, dtype) <= lds_max)
    return 2;
  return 0;
}

hipError_t vc_launch_gemm_fr(const GemmArgs& a0, int dtype, int pro, hipStream_t s) {
  const int form = vc_gemm_fr_form(a0.n_rows, a0.N, a0.K, dtype, pro, a0.nsplit);
  if (form == 0) return hipErrorInvalidValue;
  GemmArgs a = a0;
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16;
  a.n_tiles = a.N / VC_TH_RES;
  a.KT = a.K / KW;
  a.r_lds = a.n_rows;
  int ktw = 32;
  while (ktw > 1 && a.KT % (VC_FR_WAVES * ktw) != 0) ktw >>= 1;
  if (a.KT % (VC_FR_WAVES * ktw) != 0 || a.N % VC_TH_RES != 0 || a.n_rows < 1 || a.n_rows > VC_ROWS) return hipErrorInvalidValue;
  a.nchunk = a.KT / (VC_FR_WAVES * ktw);
  const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  const int upr = a.K * esz / 16, q4 = a.K / 4;
  a.x_upr_shift = -1; a.att_q4_shift = -1;
  for (int sft = 0; sft < 20; ++sft) {
    if ((1 << sft) == upr) a.x_upr_shift = sft;
    if ((1 << sft) == q4) a.att_q4_shift = sft;
  }
  if (form == 2) {
    // X does not fit the LDS of a workgroup in one piece: K in two halves (rows_gemm_fr2_k), one burst of weights per half
    const int ktw2 = a.KT / (2 * VC_FR_WAVES);
    if (a.KT % (2 * VC_FR_WAVES) != 0 || (long)a.n_rows * (upr / 2) > 16L * 64 * VC_FR_WAVES || a.n_rows > VC_ROWS) return hipErrorInvalidValue;
    a.nchunk = 2;
    a.x_upr_shift = -1;
    for (int sft = 0; sft < 20; ++sft) if ((1 << sft) == upr / 2) a.x_upr_shift = sft;
    const size_t lds2 = vc_gemm_fr_lds_bytes(a.n_rows, a.K / 2, dtype);
#define VC_FR2_CASE(K_) case K_: return (dtype == VC_DTYPE_BF16) ? launch_fr2<bf16_t, K_>(a, lds2, s) : launch_fr2<float, K_>(a, lds2, s);
    switch (ktw2) {
      VC_FR2_CASE(2) VC_FR2_CASE(4) VC_FR2_CASE(8) VC_FR2_CASE(16)
      default: return hipErrorInvalidValue;
    }
#undef VC_FR2_CASE
  }
  if (pro == PRO_ATT && (a.nsplit < 1 || a.n_rows * (a.nsplit > 4 ? 8 : a.nsplit > 2 ? 4 : 2) > 16 || (long)a.n_rows * q4 > 16L / (a.nsplit > 4 ? 8 : a.nsplit > 2 ? 4 : 2) * 64 * VC_FR_WAVES))
    return hipErrorInvalidValue;
  const size_t lds = vc_gemm_fr_lds_bytes(a.n_rows, a.K, dtype);
#define VC_FR_CASE(K_) case K_: return (dtype == VC_DTYPE_BF16) ? launch_fr_ktw<bf16_t, K_>(a, pro, lds, s) : launch_fr_ktw<float, K_>(a, pro, lds, s);
  switch (ktw) {
    VC_FR_CASE(1) VC_FR_CASE(2) VC_FR_CASE(4) VC_FR_CASE(8) VC_FR_CASE(16) VC_FR_CASE(32)
    default: return hipErrorInvalidValue;
  }
#undef VC_FR_CASE
}

// ------------------------------------------------------------------ finished-row producer of ONE-row decode steps (round 5)
// The one-row step kept split-K slabs through round 4: the FFN down-projection left 4 fp32 slabs and every one of the 512
// workgroups of the next launch (the following layer's QKV projection) re-summed h + bias + 4 slabs - 40 KB out of L2 per
// workgroup, requested in front of its own 48 KB weight burst.  rows_gemm_fr_k (above) finishes rows for 2..16-row steps, but
// its 8-channel tiles fill only half of every MFMA fragment (lanes 8..15 repeat slot 7): twice the vector-memory instructions per
// weight byte, which a one-row step - nothing but a weight stream - cannot afford.  With ONE row the other 15 columns of the B
// operand are free, so a fragment carries TWO k-tiles: A rows 0..7 = the tile's 8 channels over k-tile 2p, rows 8..15 = the same
// channels over k-tile 2p + 1 (the 8-channel image stores consecutive k-tiles of a tile back to back: one contiguous 1 KB
// burst per wave instruction, every lane's 16 bytes used); B column 0 = x over k-tile 2p, column 1 = x over k-tile 2p + 1.
// D[c][0] + D[c + 8][1] is channel c's partial sum; every other element of D is a cross term nobody reads.  A workgroup owns
// 8 channels over the whole K (d/8 workgroups), its 8 waves stream K/8 each in one burst, the row of X is staged once in LDS,
// and the epilogue adds residual + bias and writes the finished channels: the consumer's LayerNorm prologue reads ONE 8 KB row.
// NPW = fragment pairs per wave (K / KW / 16, rounded up; pairs beyond the matrix are zeroed).
// EXACT: the matrix has exactly NW * NPW fragment pairs per tile (every real model width): no clamped addresses, no zeroed fragments.
// NW waves per workgroup, each streaming NPW consecutive fragment pairs (KB) in one burst.
// PRO_PLAIN / EPI_RES (the FFN down-projection): X = a.x_in (WT [K]), h_out = h_in + bias + W x.
// PRO_LN / EPI_QKV (the QKV projection behind a FINISHED row, option "qkv_p8"): X = the row of h_in, centred and rounded (LayerNorm
//   fold, rows_gemm_k above), epilogue = rstd / mean correction + folded bias, q to a buffer, K / V into the cache.  The 12-channel
//   tiles of rows_gemm_k leave a quarter of every fragment's lanes idle: the QKV launch pays the vector-memory instructions of a
//   33.6 MB stream for 25.2 MB; here every lane's 16 bytes are weights (3d / 8 workgroups: three per CU at d = 2048).
// (A third form - the FFN up-projection on its ordinary 16-channel tiles with this kernel's lean code, LayerNorm fold of h + bias + two
// slabs - was built at the end of round 5 on the reading that straight-line code in front of the burst is paid in instruction fetch
// at every cold launch start: its first workgroup had the burst out after 2 000 clk instead of 2 756, and the step came out
// +0.84 % +- 0.44 at giga830M, +0.21 % +- 0.06 at giga330M in in-process A/Bs, profiles/r05g_ab_*.log.  Not carried: the FFN-up
// launch is bound by its stream, not by its prologue.)
template <typename WT, int NPW, bool EXACT, int NW, int PRO, int EPI>
__global__ __launch_bounds__(64 * NW) void row_gemm_fr1_k(const GemmArgs a) {
  using T = WTr<WT>;
  static_assert((PRO == PRO_PLAIN && EPI == EPI_RES) || (PRO == PRO_LN && EPI == EPI_QKV), "one-row paired forms");
  constexpr int NTHR = 64 * NW, TH = VC_TH_RES, SPT = 4 * TH;   // SPT = 16-byte units per (tile, k-tile) = 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  VC_KTS_DECL();
  VC_KTS(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x;
  const int K = a.K;
  const int npairs = a.KT >> 1;                    // KT even (host contract)
  char* xl = smem;                                 // the row: K elements of WT
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)K * sizeof(WT));     // [NW][4] partial quads
  float* stat = reinterpret_cast<float*>(red + NW * 4);                     // PRO_LN: [NW] row sums, then [NW][2] statistics of the rounded row
  const int active = *a.n_active;
  const int m = lane & 15, kg = lane >> 4;
  // epilogue operands first (a wave's loads return in order): for the two finishing threads, channels 4 t .. 4 t + 3 of the tile
  const int nfin = nt * TH + 4 * (tid & 1);
  float4 eres = make_float4(0.f, 0.f, 0.f, 0.f), ewg = eres;
  const float4 eb = *reinterpret_cast<const float4*>(a.bias + nfin);
  int epos = -1, eseq = 0;
  if constexpr (EPI == EPI_RES) eres = *reinterpret_cast<const float4*>(a.h_in + nfin);
  else { ewg = *reinterpret_cast<const float4*>(a.wg + nfin); epos = a.row_pos[0]; eseq = a.row_seq[0]; }
  // the operand row: PLAIN - 16-byte units of X (a fragment pair covers 64 of them); LN - float4 columns of the residual row
  // (a fragment pair = two k-tiles = 128 bytes of WT = 8 units; K = NW * NPW pairs when EXACT)
  constexpr int NXU = (NPW * 8 + 63) / 64;                               // 16-byte units of the WT row per thread
  constexpr int NXQ = ((sizeof(WT) == 2 ? 16 : 8) * NPW + 63) / 64;      // float4 columns of the fp32 row per thread
  static_assert(NXU <= 4 && (PRO != PRO_LN || NXQ <= 4), "the operand row fits four staging registers per thread");
  const int units = K * (int)sizeof(WT) / 16, nq = K >> 2;
  // (explicit scalars: an indexed array is demoted to scratch memory by the compiler)
  uint4 xu0 = make_uint4(0u, 0u, 0u, 0u), xu1 = xu0, xu2 = xu0, xu3 = xu0;
  float4 xq0 = make_float4(0.f, 0.f, 0.f, 0.f), xq1 = xq0, xq2 = xq0, xq3 = xq0;
  if constexpr (PRO == PRO_PLAIN) {
    const char* src = reinterpret_cast<const char*>(a.x_in);
#define VC_FR1_XLOAD(j, dst) if constexpr (NXU > (j)) dst = *reinterpret_cast<const uint4*>(src + (size_t)min(tid + (j) * NTHR, units - 1) * 16);
    VC_FR1_XLOAD(0, xu0) VC_FR1_XLOAD(1, xu1) VC_FR1_XLOAD(2, xu2) VC_FR1_XLOAD(3, xu3)
#undef VC_FR1_XLOAD
  } else {
#define VC_FR1_QLOAD(j, dst) if constexpr (NXQ > (j)) dst = *reinterpret_cast<const float4*>(a.h_in + (size_t)min(tid + (j) * NTHR, nq - 1) * 4);
    VC_FR1_QLOAD(0, xq0) VC_FR1_QLOAD(1, xq1) VC_FR1_QLOAD(2, xq2) VC_FR1_QLOAD(3, xq3)
#undef VC_FR1_QLOAD
  }
  // the wave's whole share of the weights in one burst: pair p of wave w = fragment pair (NPW w + p)
  const int wunit = (m >> 3) * SPT + kg * TH + (m & 7);
  const uint4* wbase = a.Wp + ((long)nt * a.KT + 2 * NPW * wave) * SPT;    // wave-uniform: a wave streams NPW consecutive KB
  uint4 wf[NPW];
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    const int gp = EXACT ? p : min(NPW * wave + p, npairs - 1) - NPW * wave;
    wf[p] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)gp * (2 * SPT) + wunit)));
  }
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  VC_KTS(1);
  if constexpr (PRO == PRO_PLAIN) {
#define VC_FR1_XPARK(j, val) if constexpr (NXU > (j)) { if (tid + (j) * NTHR < units) *reinterpret_cast<uint4*>(xl + (size_t)(tid + (j) * NTHR) * 16) = val; }
    VC_FR1_XPARK(0, xu0) VC_FR1_XPARK(1, xu1) VC_FR1_XPARK(2, xu2) VC_FR1_XPARK(3, xu3)
#undef VC_FR1_XPARK
  } else {
    // LayerNorm fold of the finished row (see the top of this file): centred BEFORE it is rounded, statistics of the rounded values
    float t = 0.f;
#define VC_FR1_QSUM(j, v) if constexpr (NXQ > (j)) { if (tid + (j) * NTHR < nq) t += (v.x + v.y) + (v.z + v.w); }
    VC_FR1_QSUM(0, xq0) VC_FR1_QSUM(1, xq1) VC_FR1_QSUM(2, xq2) VC_FR1_QSUM(3, xq3)
#undef VC_FR1_QSUM
    t = wave_sum(t);
    if (lane == 0) stat[wave] = t;
    __syncthreads();
    float mu = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) mu += stat[w];
    mu *= 1.0f / (float)K;
    float s1 = 0.f, s2 = 0.f;
    WT* xr = reinterpret_cast<WT*>(xl);
#define VC_FR1_QPARK(j, v)                                                                     \
    if constexpr (NXQ > (j)) {                                                                 \
      if (tid + (j) * NTHR < nq) {                                                             \
        const f32x4 y_ = {v.x - mu, v.y - mu, v.z - mu, v.w - mu};                             \
        const f32x4 q_ = store4r(xr + (size_t)(tid + (j) * NTHR) * 4, y_);                     \
        s1 += (q_[0] + q_[1]) + (q_[2] + q_[3]);                                               \
        s2 += (q_[0] * q_[0] + q_[1] * q_[1]) + (q_[2] * q_[2] + q_[3] * q_[3]);               \
      }                                                                                        \
    }
    VC_FR1_QPARK(0, xq0) VC_FR1_QPARK(1, xq1) VC_FR1_QPARK(2, xq2) VC_FR1_QPARK(3, xq3)
#undef VC_FR1_QPARK
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { stat[NW + 2 * wave] = s1; stat[NW + 2 * wave + 1] = s2; }
  }
  VC_KTS(2);
  __syncthreads();
  VC_KTS(3);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // B operand: column m = 0 reads x over k-tile 2 gp, column 1 over k-tile 2 gp + 1 (the other columns repeat these: cross terms)
  const char* xcol = xl + ((size_t)(m & 1) * T::KW + (size_t)kg * T::EPL) * sizeof(WT);
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    const int gpr = NPW * wave + p;
    const int gp = EXACT ? gpr : min(gpr, npairs - 1);
    const uint4 xf = *reinterpret_cast<const uint4*>(xcol + (size_t)gp * (2 * T::KW * sizeof(WT)));
    uint4 w = wf[p];
    if constexpr (!EXACT) { if (gpr >= npairs) w = make_uint4(0u, 0u, 0u, 0u); }
    acc = mfma_frag(w, xf, acc, (WT*)nullptr);
  }
  // D[n = 4 kg + r][m]: channels 0..3 / 4..7 over the even k-tiles sit in lanes (m = 0, kg = 0 / 1), over the odd ones in lanes
  // (m = 1, kg = 2 / 3): four quads per wave
  VC_KTS(4);
  if (m == (kg >> 1) && m < 2) red[wave * 4 + kg] = acc;
  __syncthreads();
  VC_KTS(5);
  if (tid < 2) {             // thread t finishes channels 4 t .. 4 t + 3: quads t and t + 2 of every wave, in a fixed order
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[w * 4 + tid] + red[w * 4 + tid + 2];
    if constexpr (EPI == EPI_RES) {
      const f32x4 o = {eres.x + eb.x + sum[0], eres.y + eb.y + sum[1], eres.z + eb.z + sum[2], eres.w + eb.w + sum[3]};
      store4(a.h_out + nfin, o);
    } else {
      float q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { q1 += stat[NW + 2 * w]; q2 += stat[NW + 2 * w + 1]; }
      const float inv_d = 1.0f / (float)K;
      const float mean = q1 * inv_d;
      const float var = fmaxf(q2 * inv_d - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);      // eps 1e-5 (transformer.py:30)
      sum[0] = rstd * (sum[0] - mean * ewg.x); sum[1] = rstd * (sum[1] - mean * ewg.y);
      sum[2] = rstd * (sum[2] - mean * ewg.z); sum[3] = rstd * (sum[3] - mean * ewg.w);
      gemm_epilogue<WT, EPI_QKV>(a, sum, 0, nfin, 0, 0, 1, eb, epos, eseq);
    }
  }
  VC_KTS(6);
  VC_KTS_FLUSH();
}
template <typename WT, int NPW, int NW, int PRO, int EPI>
static hipError_t launch_fr1_n(const GemmArgs& a, hipStream_t s) {
  auto kern = (a.KT == 2 * NW * NPW) ? row_gemm_fr1_k<WT, NPW, true, NW, PRO, EPI> : row_gemm_fr1_k<WT, NPW, false, NW, PRO, EPI>;
  const size_t lds = (size_t)a.K * sizeof(WT) + (size_t)NW * 4 * sizeof(f32x4) + (size_t)NW * 3 * sizeof(float);
  ++vc_launch_counts[VC_LC_ROW_GEMM_FR1];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(64 * NW), lds, s, a);
  return hipGetLastError();
}
// 1 when the one-row paired kernel can take an [N x K] matrix in this dtype with `nw` waves per workgroup (the engine's planning and
// the launcher agree on it)
int vc_gemm_fr1_ok(int N, int K, int dtype, int nw) {
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  if (N % VC_TH_RES != 0 || K % (2 * KW) != 0 || (nw != 4 && nw != 8)) return 0;
  const int npairs = K / KW / 2, npw = (npairs + nw - 1) / nw;
  int cap = 1;
  while (cap < npw) cap <<= 1;
  if (cap > (nw == 4 ? 16 : 32) || (long)K * esz > 64L * 1024) return 0;      // a wave's fragments fit its registers; the row <= 4 staging registers per thread
  return 1;
}
// ONE row.  pro / epi = PRO_PLAIN / EPI_RES: h_out[n] = h_in[n] + bias[n] + sum_k W[n][k] x[k] (a.x_in = the row, WT [K]; 8 waves);
// PRO_LN / EPI_QKV: the QKV projection of the finished row a.h_in (LayerNorm fold; 4 waves).  a.Wp = the 8-channel-tile image.
hipError_t vc_launch_gemm_fr1(const GemmArgs& a0, int dtype, int pro, int epi, hipStream_t s) {
  const bool qkv = pro == PRO_LN && epi == EPI_QKV;
  if (!qkv && !(pro == PRO_PLAIN && epi == EPI_RES)) return hipErrorInvalidValue;
  const int nw = qkv ? 4 : VC_FR_WAVES;        // (the QKV form with eight waves measured +1.9 % per step, profiles/r05*: four stay)
  if (!vc_gemm_fr1_ok(a0.N, a0.K, dtype, nw) || a0.n_rows != 1) return hipErrorInvalidValue;
  GemmArgs a = a0;
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16;
  a.n_tiles = a.N / VC_TH_RES;
  a.KT = a.K / KW;
  const int npairs = a.KT / 2, npw = (npairs + nw - 1) / nw;
  int cap = 1;
  while (cap < npw) cap <<= 1;
#define VC_FR1_CASE(P_) case P_:                                                                                        \
    if (qkv) return (dtype == VC_DTYPE_BF16) ? launch_fr1_n<bf16_t, P_, 4, PRO_LN, EPI_QKV>(a, s) : launch_fr1_n<float, P_, 4, PRO_LN, EPI_QKV>(a, s);   \
    return (dtype == VC_DTYPE_BF16) ? launch_fr1_n<bf16_t, P_, VC_FR_WAVES, PRO_PLAIN, EPI_RES>(a, s) : launch_fr1_n<float, P_, VC_FR_WAVES, PRO_PLAIN, EPI_RES>(a, s);
  switch (cap) {
    VC_FR1_CASE(1) VC_FR1_CASE(2) VC_FR1_CASE(4) VC_FR1_CASE(8) VC_FR1_CASE(16)
    case 32:
      if (qkv) return hipErrorInvalidValue;      // (K <= 2048 there: at most 16 pairs per wave of four)
      return (dtype == VC_DTYPE_BF16) ? launch_fr1_n<bf16_t, 32, VC_FR_WAVES, PRO_PLAIN, EPI_RES>(a, s) : launch_fr1_n<float, 32, VC_FR_WAVES, PRO_PLAIN, EPI_RES>(a, s);
    default: return hipErrorInvalidValue;
  }
#undef VC_FR1_CASE
}

// ------------------------------------------------------------------ finished-row producer of 2..8-row steps, paired (round 5)
// The same two-k-tiles-per-fragment trick for up to EIGHT rows: the B operand has 16 columns, so row r takes columns 2r (x over
// k-tile 2p) and 2r + 1 (x over k-tile 2p + 1); D[c][2r] + D[c + 8][2r + 1] is channel c of row r.  rows_gemm_fr_k (above) issues one
// 512-byte fragment per wave instruction for the same 8-channel tiles; here every instruction is a full KB - half the vector-memory
// instructions of the FFN down-projection of config 5's per-GPU share (8 rows).  X rows sit in LDS at r K + 16 (r & 3) + 128 (r >> 2)
// bytes, so that the sixteen 16-byte reads of a lane group (8 rows x 2 k-tile parities) fall into sixteen different bank quads.
// RMAX = rows the staging is unrolled for (4 or 8); K x sizeof(WT) / 16 is a power of two (host contract).
template <typename WT, int NPW, int RMAX>
__global__ __launch_bounds__(64 * VC_FR_WAVES) void rows_gemm_frp_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NW = VC_FR_WAVES, NTHR = 64 * NW, TH = VC_TH_RES, SPT = 4 * TH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x;
  const int n_rows = a.n_rows;                      // 2..RMAX (host contract)
  const int Kb = a.K * (int)sizeof(WT);             // bytes of one X row = NW * NPW * 128
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)RMAX * Kb + 256);      // [NW][RMAX][4]
  const int active = *a.n_active;
  const int m = lane & 15, kg = lane >> 4;
  // epilogue operands first: thread t < 2 RMAX finishes channels 4 (t & 1) .. of row t >> 1
  const int frow = min(tid >> 1, n_rows - 1);
  const int nfin = nt * TH + 4 * (tid & 1);
  const float4 eres = *reinterpret_cast<const float4*>(a.h_in + (long)frow * a.d + nfin);
  const float4 eb = *reinterpret_cast<const float4*>(a.bias + nfin);
  const float cmu = a.row_mu[frow];
  // X rows as 16-byte units, flat index = row * upr + unit; RMAX * NPW / 8 units per thread
  constexpr int NXU = RMAX * ((NPW * 8 + 63) / 64);
  const int ush = a.x_upr_shift;                    // log2(units per row)
  const int upr = 1 << ush, total = n_rows * upr;
  const char* src = reinterpret_cast<const char*>(a.x_in);
  const long rstride = (long)a.x_ld * (long)sizeof(WT);
  // (explicit scalars: an indexed array is demoted to scratch memory by the compiler)
  uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
  static_assert(NXU <= 16, "staging registers per thread");
#define VC_FRP_LOAD(j, dst)                                                                      \
  if constexpr (NXU > (j)) {                                                                     \
    const int i_ = min(tid + (j) * NTHR, total - 1);                                             \
    const int r_ = i_ >> ush;                                                                    \
    dst = *reinterpret_cast<const uint4*>(src + (long)r_ * rstride + (long)(i_ - (r_ << ush)) * 16); \
  }
#define VC_FRP_PARK(j, val)                                                                      \
  if constexpr (NXU > (j)) {                                                                     \
    const int i_ = tid + (j) * NTHR;                                                             \
    if (i_ < total) {                                                                            \
      const int r_ = i_ >> ush;                                                                  \
      *reinterpret_cast<uint4*>(xl + (size_t)r_ * Kb + 16 * (r_ & 3) + 128 * (r_ >> 2) + (size_t)(i_ - (r_ << ush)) * 16) = val; \
    }                                                                                            \
  }
  VC_FRP_LOAD(0, x0) VC_FRP_LOAD(1, x1) VC_FRP_LOAD(2, x2) VC_FRP_LOAD(3, x3) VC_FRP_LOAD(4, x4) VC_FRP_LOAD(5, x5) VC_FRP_LOAD(6, x6) VC_FRP_LOAD(7, x7)
  VC_FRP_LOAD(8, x8) VC_FRP_LOAD(9, x9) VC_FRP_LOAD(10, x10) VC_FRP_LOAD(11, x11) VC_FRP_LOAD(12, x12) VC_FRP_LOAD(13, x13) VC_FRP_LOAD(14, x14) VC_FRP_LOAD(15, x15)
  const int wunit = (m >> 3) * SPT + kg * TH + (m & 7);
  const uint4* wbase = a.Wp + ((long)nt * a.KT + 2 * NPW * wave) * SPT;
  uint4 wf[NPW];
#pragma unroll
  for (int p = 0; p < NPW; ++p)
    wf[p] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)p * (2 * SPT) + wunit)));
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  VC_FRP_PARK(0, x0) VC_FRP_PARK(1, x1) VC_FRP_PARK(2, x2) VC_FRP_PARK(3, x3) VC_FRP_PARK(4, x4) VC_FRP_PARK(5, x5) VC_FRP_PARK(6, x6) VC_FRP_PARK(7, x7)
  VC_FRP_PARK(8, x8) VC_FRP_PARK(9, x9) VC_FRP_PARK(10, x10) VC_FRP_PARK(11, x11) VC_FRP_PARK(12, x12) VC_FRP_PARK(13, x13) VC_FRP_PARK(14, x14) VC_FRP_PARK(15, x15)
#undef VC_FRP_LOAD
#undef VC_FRP_PARK
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int rl = min(m >> 1, n_rows - 1);           // (columns of rows the pass does not have repeat the last row: cross terms)
  const char* xcol = xl + (size_t)rl * Kb + 16 * (rl & 3) + 128 * (rl >> 2) + ((size_t)(m & 1) * T::KW + (size_t)kg * T::EPL) * sizeof(WT);
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    const int gp = NPW * wave + p;
    const uint4 xf = *reinterpret_cast<const uint4*>(xcol + (size_t)gp * (2 * T::KW * sizeof(WT)));
    acc = mfma_frag(wf[p], xf, acc, (WT*)nullptr);
  }
  // D[n = 4 kg + r][m]: row m >> 1; even columns hold the even k-tiles' channels 0..7 (kg 0 / 1), odd columns the odd k-tiles' (kg 2 / 3)
  if ((m & 1) == (kg >> 1) && (m >> 1) < n_rows) red[(wave * RMAX + (m >> 1)) * 4 + kg] = acc;
  __syncthreads();
  if (tid < 2 * n_rows) {
    const int r = tid >> 1, half = tid & 1;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[(w * RMAX + r) * 4 + half] + red[(w * RMAX + r) * 4 + half + 2];
    const f32x4 o = {eres.x + eb.x + sum[0], eres.y + eb.y + sum[1], eres.z + eb.z + sum[2], eres.w + eb.w + sum[3]};
    store4(a.h_out + (long)r * a.d + nfin, o);
    if (a.hq_out) { const f32x4 q = {o[0] - cmu, o[1] - cmu, o[2] - cmu, o[3] - cmu}; store4(reinterpret_cast<WT*>(a.hq_out) + (long)r * a.d + nfin, q); }
  }
}
template <typename WT, int NPW, int RMAX>
static hipError_t launch_frp_n(const GemmArgs& a, hipStream_t s) {
  auto kern = rows_gemm_frp_k<WT, NPW, RMAX>;
  const size_t lds = (size_t)RMAX * a.K * sizeof(WT) + 256 + (size_t)VC_FR_WAVES * RMAX * 4 * sizeof(f32x4);
  if (lds > 64 * 1024) {
    static size_t granted[16] = {0};   // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && lds > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = lds;
    }
  }
  ++vc_launch_counts[VC_LC_ROWS_GEMM_FRP];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles), dim3(64 * VC_FR_WAVES), lds, s, a);
  return hipGetLastError();
}
// 1 when the paired producer can take `rows` rows of an [N x K] matrix: whole fragment pairs per wave, a power-of-two unit count per
// row, X within the LDS of one workgroup
int vc_gemm_frp_ok(int rows, int N, int K, int dtype) {
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  if (rows < 2 || rows > 8 || N % VC_TH_RES != 0 || K % (2 * KW * VC_FR_WAVES) != 0) return 0;
  const int npw = K / (2 * KW * VC_FR_WAVES), upr = K * esz / 16, rmax = rows <= 4 ? 4 : 8;
  if ((npw & (npw - 1)) != 0 || npw > 32 || (upr & (upr - 1)) != 0) return 0;
  if (rmax * ((npw * 8 + 63) / 64) > 16) return 0;                                    // staging registers per thread
  return (size_t)rmax * K * esz + 256 + (size_t)VC_FR_WAVES * rmax * 64 <= 160 * 1024 ? 1 : 0;
}
hipError_t vc_launch_gemm_frp(const GemmArgs& a0, int dtype, hipStream_t s) {
  if (!vc_gemm_frp_ok(a0.n_rows, a0.N, a0.K, dtype)) return hipErrorInvalidValue;
  GemmArgs a = a0;
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  a.n_tiles = a.N / VC_TH_RES;
  a.KT = a.K / KW;
  const int npw = a.KT / (2 * VC_FR_WAVES), upr = a.K * esz / 16;
  a.x_upr_shift = 0;
  while ((1 << a.x_upr_shift) < upr) ++a.x_upr_shift;
  const bool r4 = a.n_rows <= 4;
#define VC_FRP_CASE(P_) case P_:                                                                                         \
    if (r4) return (dtype == VC_DTYPE_BF16) ? launch_frp_n<bf16_t, P_, 4>(a, s) : launch_frp_n<float, P_, 4>(a, s);       \
    return (dtype == VC_DTYPE_BF16) ? launch_frp_n<bf16_t, P_, 8>(a, s) : launch_frp_n<float, P_, 8>(a, s);
  switch (npw) {
    VC_FRP_CASE(1) VC_FRP_CASE(2) VC_FRP_CASE(4) VC_FRP_CASE(8) VC_FRP_CASE(16)
    case 32: if (r4) return (dtype == VC_DTYPE_BF16) ? launch_frp_n<bf16_t, 32, 4>(a, s) : launch_frp_n<float, 32, 4>(a, s);
             return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
  }
#undef VC_FRP_CASE
}

// ------------------------------------------------------------------ QKV projection of 2..8 finished rows, paired (round 6)
// The consumer the round-5 verdict asked for (item 3: "full-lane QKV tiles at 2..16 rows").  rows_gemm_k runs the QKV projection on the one-row
// kernels' 12-channel tiles: a quarter of every weight request's lanes is idle, the launch pays the vector-memory instructions of a 33.6 MB
// stream for 25.2 MB.  Here the 8-channel image of the one-row
// paired form (Wqkv8, option "qkv_p8") is read with two k-tiles per MFMA fragment exactly as rows_gemm_frp_k reads W2 - every lane's 16 bytes are
// weights -, a workgroup owns THREE consecutive tiles (24 channels: d / 8 workgroups, one per CU at d = 2048, the rows staged once for all three)
// and its 8 waves stream K / 8 of each tile in one burst.  The rows are the producers' centred copy q = WT(h - c) (PRO_LNQ, rows_gemm_k above):
// wave r copies row r to LDS as it is - whole 1 KB requests, no conversion - and sums its values and squares on the way; the epilogue applies
// rstd (acc - mean(q) rowsum(W')) + cb and the QKV scatter (q to its buffer, K / V into the caches).  B operand: row r takes columns 2r
// (k-tile 2p) and 2r + 1 (k-tile 2p + 1); D[c][2r] + D[c + 8][2r + 1] is channel c of row r, the other elements are cross terms nobody reads.
// NPW = fragment pairs per wave and tile = K sizeof(WT) / 1024 = the row's 16-byte units per lane; RMAX = 4 or 8 rows.
template <typename WT, int NPW, int RMAX>
__global__ __launch_bounds__(64 * VC_FR_WAVES) void rows_gemm_qp_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NW = VC_FR_WAVES, TH = VC_TH_RES, SPT = 4 * TH, NTQ = 3;
  constexpr int E16 = 16 / (int)sizeof(WT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_rows = a.n_rows;                      // 2..RMAX (host contract)
  const int Kb = a.K * (int)sizeof(WT);             // bytes of one row = NPW * 1024
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)RMAX * Kb + 256);      // [NW][RMAX][NTQ][4]
  float* stat = reinterpret_cast<float*>(red + NW * RMAX * NTQ * 4);         // [RMAX][2]: sum and sum of squares of the row's copy
  const int active = *a.n_active;
  const int m = lane & 15, kg = lane >> 4;
  // epilogue operands first (a wave's loads return in order): thread t < 6 n_rows finishes channels 4 (t & 1) .. of tile (t >> 1) % 3 of row t / 6
  const int frow = min(tid / (2 * NTQ), n_rows - 1);
  const int ftile = (tid >> 1) % NTQ;
  const int nfin = ((int)blockIdx.x * NTQ + ftile) * TH + 4 * (tid & 1);
  const float4 eb = *reinterpret_cast<const float4*>(a.bias + nfin);
  const float4 ewg = *reinterpret_cast<const float4*>(a.wg + nfin);
  const int epos = a.row_pos[frow], eseq = a.row_seq[frow];
  // wave r < n_rows: row r of the centred copy, NPW requests of 1 KB
  const int xrow = min(wave, n_rows - 1);
  const char* qsrc = reinterpret_cast<const char*>(a.x_in) + (long)xrow * a.x_ld * (long)sizeof(WT);
  const float cmu = a.row_mu[xrow];
  uint4 xq[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) xq[j] = *reinterpret_cast<const uint4*>(qsrc + (j * 64 + lane) * 16);
  // the wave's share of the three tiles in one burst: pair p of wave w = fragment pair (NPW w + p) of each tile
  const int wunit = (m >> 3) * SPT + kg * TH + (m & 7);
  const uint4* wbase = a.Wp + ((long)blockIdx.x * NTQ * a.KT + 2 * NPW * wave) * SPT;
  uint4 wf[NTQ][NPW];
#pragma unroll
  for (int t = 0; t < NTQ; ++t)
#pragma unroll
    for (int p = 0; p < NPW; ++p)
      wf[t][p] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + ((long)t * a.KT + 2 * p) * SPT + wunit)));
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  if (wave < n_rows) {
    char* xr = xl + (size_t)wave * Kb + 16 * (wave & 3) + 128 * (wave >> 2);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      *reinterpret_cast<uint4*>(xr + (j * 64 + lane) * 16) = xq[j];
      float f[E16];
      if constexpr (sizeof(WT) == 2) {
        f[0] = __uint_as_float(xq[j].x << 16); f[1] = __uint_as_float(xq[j].x & 0xffff0000u);
        f[2] = __uint_as_float(xq[j].y << 16); f[3] = __uint_as_float(xq[j].y & 0xffff0000u);
        f[4] = __uint_as_float(xq[j].z << 16); f[5] = __uint_as_float(xq[j].z & 0xffff0000u);
        f[6] = __uint_as_float(xq[j].w << 16); f[7] = __uint_as_float(xq[j].w & 0xffff0000u);
      } else {
        f[0] = __uint_as_float(xq[j].x); f[1] = __uint_as_float(xq[j].y);
        f[2] = __uint_as_float(xq[j].z); f[3] = __uint_as_float(xq[j].w);
      }
#pragma unroll
      for (int e = 0; e < E16; e += 4) {
        s1 += (f[e] + f[e + 1]) + (f[e + 2] + f[e + 3]);
        s2 += (f[e] * f[e] + f[e + 1] * f[e + 1]) + (f[e + 2] * f[e + 2] + f[e + 3] * f[e + 3]);
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      stat[2 * wave] = s1; stat[2 * wave + 1] = s2;
      if (a.row_mu_out && blockIdx.x == 0) a.row_mu_out[wave] = cmu + s1 * (1.0f / (float)a.K);      // the next producer centres its copy of the row on it
    }
  }
  __syncthreads();
  f32x4 acc[NTQ];
#pragma unroll
  for (int t = 0; t < NTQ; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int rl = min(m >> 1, n_rows - 1);           // (columns of rows the pass does not have repeat the last row: cross terms)
  const char* xcol = xl + (size_t)rl * Kb + 16 * (rl & 3) + 128 * (rl >> 2) + ((size_t)(m & 1) * T::KW + (size_t)kg * T::EPL) * sizeof(WT);
#pragma unroll
  for (int p = 0; p < NPW; ++p) {
    const int gp = NPW * wave + p;
    const uint4 xf = *reinterpret_cast<const uint4*>(xcol + (size_t)gp * (2 * T::KW * sizeof(WT)));
#pragma unroll
    for (int t = 0; t < NTQ; ++t) acc[t] = mfma_frag(wf[t][p], xf, acc[t], (WT*)nullptr);
  }
  // D[n = 4 kg + r][m]: row m >> 1; even columns hold the even k-tiles' channels 0..7 (kg 0 / 1), odd columns the odd k-tiles' (kg 2 / 3)
  if ((m & 1) == (kg >> 1) && (m >> 1) < n_rows) {
#pragma unroll
    for (int t = 0; t < NTQ; ++t) red[((wave * RMAX + (m >> 1)) * NTQ + t) * 4 + kg] = acc[t];
  }
  __syncthreads();
  if (tid < 2 * NTQ * n_rows) {
    const int half = tid & 1;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += red[((w * RMAX + frow) * NTQ + ftile) * 4 + half] + red[((w * RMAX + frow) * NTQ + ftile) * 4 + half + 2];
    const float inv_d = 1.0f / (float)a.K;
    const float mean = stat[2 * frow] * inv_d;
    const float var = fmaxf(stat[2 * frow + 1] * inv_d - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);      // eps 1e-5 (transformer.py:30)
    sum[0] = rstd * (sum[0] - mean * ewg.x); sum[1] = rstd * (sum[1] - mean * ewg.y);
    sum[2] = rstd * (sum[2] - mean * ewg.z); sum[3] = rstd * (sum[3] - mean * ewg.w);
    gemm_epilogue<WT, EPI_QKV>(a, sum, frow, nfin, 0, 0, 1, eb, epos, eseq);
  }
}
template <typename WT, int NPW, int RMAX>
static hipError_t launch_qp_n(const GemmArgs& a, hipStream_t s) {
  auto kern = rows_gemm_qp_k<WT, NPW, RMAX>;
  const size_t lds = (size_t)RMAX * a.K * sizeof(WT) + 256 + (size_t)VC_FR_WAVES * RMAX * 3 * 4 * sizeof(f32x4) + (size_t)RMAX * 2 * sizeof(float);
  if (lds > 64 * 1024) {
    static size_t granted[16] = {0};   // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && lds > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = lds;
    }
  }
  ++vc_launch_counts[VC_LC_ROWS_GEMM_QP];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles / 3), dim3(64 * VC_FR_WAVES), lds, s, a);
  return hipGetLastError();
}
// 1 when the paired QKV consumer can take `rows` rows of the centred copy through an [N x K] matrix: three 8-channel tiles per workgroup, a row of
// whole 1 KB wave requests (1, 2, 4 or 8 of them), everything within the LDS of one workgroup
int vc_gemm_qp_ok(int rows, int N, int K, int dtype) {
  const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  if (rows < 2 || rows > 8 || N % (3 * VC_TH_RES) != 0 || (K * esz) % 1024 != 0) return 0;
  const int npw = K * esz / 1024, rmax = rows <= 4 ? 4 : 8;
  if ((npw & (npw - 1)) != 0 || npw > 8) return 0;
  return (size_t)rmax * K * esz + 256 + (size_t)VC_FR_WAVES * rmax * 3 * 64 + (size_t)rmax * 8 <= 160 * 1024 ? 1 : 0;
}
// X = a.x_in (the producers' centred copy, WT [rows][x_ld]), a.row_mu its constants, a.Wp the 8-channel-tile image of the folded QKV matrix
hipError_t vc_launch_gemm_qp(const GemmArgs& a0, int dtype, hipStream_t s) {
  if (!vc_gemm_qp_ok(a0.n_rows, a0.N, a0.K, dtype) || a0.x_in == nullptr || a0.row_mu == nullptr) return hipErrorInvalidValue;
  GemmArgs a = a0;
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  a.n_tiles = a.N / VC_TH_RES;
  a.KT = a.K / KW;
  const int npw = a.K * esz / 1024;
  const bool r4 = a.n_rows <= 4;
#define VC_QP_CASE(P_) case P_:                                                                                         \
    if (r4) return (dtype == VC_DTYPE_BF16) ? launch_qp_n<bf16_t, P_, 4>(a, s) : launch_qp_n<float, P_, 4>(a, s);       \
    return (dtype == VC_DTYPE_BF16) ? launch_qp_n<bf16_t, P_, 8>(a, s) : launch_qp_n<float, P_, 8>(a, s);
  switch (npw) {
    VC_QP_CASE(1) VC_QP_CASE(2) VC_QP_CASE(4) VC_QP_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef VC_QP_CASE
}

// ------------------------------------------------------------------ wide decode passes: 17..64 rows
// (round 1's prefill kernel, kept for decode passes that carry more than one 16-row tile - 17..64 sequences: such a
// pass is still a weight stream, and here the weights are read from HBM exactly once at the decode kernel's rate.)
// Same weight fragments, but a workgroup holds NTW weight tiles (NTW x 4 waves) in registers and walks
// the pass's rows in tiles of 16, so W is streamed once per 128 rows instead of once per 16, and every
// 16-row X tile staged in LDS feeds NTW output tiles (the X traffic out of L2 - 64 KB per row tile per
// workgroup at d = 2048 - is what bounds this kernel, not HBM).  The next row tile is requested
// while the current one is in the MFMAs.  LayerNorm is hoisted into ln_rows_k (one block per row).
// (the weights are read from HBM exactly once per pass: non-temporal loads, compile-time - VC_MT_NT=0 builds the comparison twin)
#ifndef VC_MT_NT
#define VC_MT_NT 1
#endif
#if VC_MT_NT
#define VC_MT_LOADW(p_) __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p_)))
#else
#define VC_MT_LOADW(p_) (*(p_))
#endif
template <typename WT, int KTW, int PRO, int EPI, int NTW>
__global__ __launch_bounds__(256 * NTW) void rows_gemm_mt_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NT = 256 * NTW;          // threads
  constexpr int XP = 16 / NTW;           // X units (16 B) a thread carries for the next row tile (64 KB per workgroup)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (*a.n_active == 0) return;          // a replayed decode step after the last sequence retired
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wv >> 2, wave = wv & 3;  // weight tile of this workgroup, K quarter
  const int nt_raw = blockIdx.x * NTW + g;
  const bool tile_ok = nt_raw < a.n_tiles;
  const int nt = tile_ok ? nt_raw : a.n_tiles - 1;
  const int ks = blockIdx.y, grp = blockIdx.z;
  const int n_rows = a.n_rows;
  const int kt_blk = a.nchunk * 4 * KTW;
  const int kt0 = ks * kt_blk;
  const int kblk = kt_blk * T::KW;
  const int k0 = kt0 * T::KW;
  const int xs = kblk * (int)sizeof(WT) + 16;
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)VC_ROWS * xs) + g * 256;
  constexpr int TH = (EPI == EPI_QKV) ? VC_TH_QKV : 16;    // see rows_gemm_k
  constexpr int SPT = 4 * TH;
  const int m = lane & 15;
  const int kg = lane >> 4;
  const bool wvalid = m < TH, nvalid = 4 * kg < TH;
  const uint4* wp = a.Wp + (long)grp * a.w_group_stride + ((long)nt * a.KT) * SPT + (kg * TH + min(m, TH - 1));
  uint4 wf[KTW];
  {
    const int kt = kt0 + wave * KTW;
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
      wf[i] = VC_MT_LOADW(wp + (long)(kt + i) * SPT);
      if (TH < 16 && !wvalid) wf[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  const int n = nt * TH + 4 * kg;
  const int upr = kblk * (int)sizeof(WT) / 16;       // 16-byte units per X row slice
  const char* xsrc = reinterpret_cast<const char*>(a.x_in) + ((long)grp * a.x_group_stride + k0) * (long)sizeof(WT);
  const long rstride = (long)a.x_ld * (long)sizeof(WT);
  const int sh = a.x_upr_shift;
  uint4 xq0, xq1, xq2, xq3, xq4, xq5, xq6, xq7;   // explicit scalars: an indexed array is demoted to scratch
  // (the 1024-thread form runs at the 128-register cap: the reciprocal of the general row split below - a loop invariant the
  // compiler keeps in a VECTOR register although it is uniform - would be spilled across the row-tile loop; there the divisor
  // is made opaque at each use, so the few instructions are redone per tile instead)
  auto upr_div = [&]() { int u = upr; if constexpr (NTW >= 4) asm volatile("" : "+s"(u)); return u; };

  // X tile of rows [row0, row0 + nr): global -> registers (first XP * NT units) ...
#define VC_XQ_LOAD(j, dst)                                                                       \
  if constexpr ((j) < XP) {                                                                      \
    const int i_ = min((j) * NT + tid, total - 1);                                               \
    const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr_div());                                    \
    const int u_ = i_ - r_ * upr;                                                                \
    dst = *reinterpret_cast<const uint4*>(xsrc + (long)(row0 + r_) * rstride + (long)u_ * 16);   \
  }
#define VC_XQ_STORE(j, val)                                                                      \
  if constexpr ((j) < XP) {                                                                      \
    const int i_ = (j) * NT + tid;                                                               \
    if (i_ < total) {                                                                            \
      const int r_ = (sh >= 0) ? (i_ >> sh) : (i_ / upr_div());                                  \
      const int u_ = i_ - r_ * upr;                                                              \
      *reinterpret_cast<uint4*>(xl + (size_t)r_ * xs + (size_t)u_ * 16) = val;                   \
    }                                                                                            \
  }
  auto x_fetch = [&](int row0, int nr) {
    if constexpr (PRO == PRO_PLAIN) {
      const int total = nr * upr;
      VC_XQ_LOAD(0, xq0) VC_XQ_LOAD(1, xq1) VC_XQ_LOAD(2, xq2) VC_XQ_LOAD(3, xq3)
      VC_XQ_LOAD(4, xq4) VC_XQ_LOAD(5, xq5) VC_XQ_LOAD(6, xq6) VC_XQ_LOAD(7, xq7)
    }
  };
  // ... -> LDS (plus whatever did not fit into the XP slots, copied directly)
  auto x_park = [&](int row0, int nr) {
    if constexpr (PRO == PRO_PLAIN) {
      const int total = nr * upr;
      VC_XQ_STORE(0, xq0) VC_XQ_STORE(1, xq1) VC_XQ_STORE(2, xq2) VC_XQ_STORE(3, xq3)
      VC_XQ_STORE(4, xq4) VC_XQ_STORE(5, xq5) VC_XQ_STORE(6, xq6) VC_XQ_STORE(7, xq7)
      for (int i = XP * NT + tid; i < total; i += NT) {
        const int r = (sh >= 0) ? (i >> sh) : (i / upr_div());
        const int u = i - r * upr;
        *reinterpret_cast<uint4*>(xl + (size_t)r * xs + (size_t)u * 16) =
            *reinterpret_cast<const uint4*>(xsrc + (long)(row0 + r) * rstride + (long)u * 16);
      }
    } else {   // PRO_ATT (short passes whose attention was split): merge the partials in place
      const int q4 = kblk >> 2;
      for (int idx = tid; idx < nr * q4; idx += NT) {
        const int r = idx / q4, c = k0 + (idx - r * q4) * 4;
        const int h = c >> a.hd_shift, e = c & (a.hd - 1);
        const float2* ml = reinterpret_cast<const float2*>(a.att_ml) + (long)((row0 + r) * a.H + h) * a.nsplit;
        const float* op = a.att_o + ((long)((row0 + r) * a.H + h) * a.nsplit) * a.hd + e;
        float M = -INFINITY;
        for (int sp = 0; sp < a.nsplit; ++sp) M = fmaxf(M, ml[sp].x);
        float L = 0.f;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < a.nsplit; ++sp) {
          const float2 v = ml[sp];
          const float w = (v.x == -INFINITY) ? 0.f : expf(v.x - M);
          const float4 os = *reinterpret_cast<const float4*>(op + (long)sp * a.hd);
          L += w * v.y;
          o[0] += w * os.x; o[1] += w * os.y; o[2] += w * os.z; o[3] += w * os.w;
        }
        const float inv = (L > 0.f) ? 1.0f / L : 0.f;
        o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
        store4(reinterpret_cast<WT*>(xl + (size_t)r * xs) + (c - k0), o);
      }
    }
  };

  x_fetch(0, min(VC_ROWS, n_rows));
  x_park(0, min(VC_ROWS, n_rows));
  for (int row0 = 0; row0 < n_rows; row0 += VC_ROWS) {
    const int nr = min(VC_ROWS, n_rows - row0);
    const int row1 = row0 + VC_ROWS;
    const int nr1 = min(VC_ROWS, n_rows - row1);
    __syncthreads();                       // X(row0) is in LDS
    // the epilogue's operands (bias, the rows' cache slots) are requested HERE, ahead of the next X tile: asked for next
    // to the stores they are a dependent L2 round trip per 16-row tile on the critical path of every workgroup
    float4 eb = make_float4(0.f, 0.f, 0.f, 0.f);
    int epos = -1, eseq = 0;
    // (the 1024-thread form runs at the 128-register cap: there the six operand registers would spill across the MFMA loop,
    // so it asks for them after the loop, under the K-reduce barrier)
    if constexpr (NTW < 4) { if (wave == 0) epi_preload<WT, EPI>(a, row0 + min(m, nr - 1), n, grp, eb, epos, eseq); }
    if (nr1 > 0) x_fetch(row1, nr1);       // next tile on its way while this one is in the MFMAs
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int mrow = (m < nr) ? m : 0;
    const char* xrow = xl + (size_t)mrow * xs + (size_t)(lane >> 4) * 16;
    for (int c = 0; c < a.nchunk; ++c) {
      if (a.nchunk > 1) {       // several chunks: the registers only ever hold one of them
        const int kt = kt0 + (c * 4 + wave) * KTW;
#pragma unroll
        for (int i = 0; i < KTW; ++i) {
          wf[i] = VC_MT_LOADW(wp + (long)(kt + i) * SPT);
          if (TH < 16 && !wvalid) wf[i] = make_uint4(0u, 0u, 0u, 0u);
        }
      }
      const int ktl = (c * 4 + wave) * KTW;
#pragma unroll
      for (int i = 0; i < KTW; ++i) {
        const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)(ktl + i) * 64);
        acc = mfma_frag(wf[i], xf, acc, (WT*)nullptr);
      }
    }
    // (... and the channel index is made opaque per tile: otherwise the epilogue's loop-invariant address arithmetic - head
    // and element of the lane's channels, 64-bit products - is hoisted out of the row-tile loop and spilled across it)
    int n_e = n;
    if constexpr (NTW >= 4) {
      asm volatile("" : "+v"(n_e));
      if (wave == 0) epi_preload<WT, EPI>(a, row0 + min(m, nr - 1), n_e, grp, eb, epos, eseq);
    }
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && m < nr && nvalid && tile_ok) {
      const f32x4 a1 = red[64 + lane], a2 = red[128 + lane], a3 = red[192 + lane];
      acc = (acc + a1) + (a2 + a3);
      gemm_epilogue<WT, EPI>(a, acc, row0 + m, n_e, ks, grp, (int)gridDim.z, eb, epos, eseq);
    }
    __syncthreads();            // x and red are rewritten for the next tile
    if (nr1 > 0) x_park(row1, nr1);
  }
#undef VC_XQ_LOAD
#undef VC_XQ_STORE
}

// h_new = h + prev_bias + sum of split-K slabs ; x_hat = (h_new - mean) * rstd as WT (the affine part of the
// LayerNorm lives in the folded weights, see the top of this file).  One block per row.
template <typename WT>
__global__ __launch_bounds__(256) void ln_rows_k(const GemmArgs a) {
  __shared__ float s_sum[4], s_sq[4];
  const int active = *a.n_active;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d;
  const int nq = d >> 2;
  const bool on0 = tid < nq, on1 = tid + 256 < nq;
  const int c0 = on0 ? tid * 4 : 0, c1 = on1 ? (tid + 256) * 4 : 0;
  // every operand is requested before anything is waited for (one round trip): the row, the bias and
  // all VC_MAX_KSPLIT slabs (unused ones are read and discarded by a select)
  const float* hp = a.h_in + (long)r * d;
  float4 x0 = *reinterpret_cast<const float4*>(hp + c0), x1 = *reinterpret_cast<const float4*>(hp + c1);
  const float4 pb0 = *reinterpret_cast<const float4*>(a.prev_bias + c0), pb1 = *reinterpret_cast<const float4*>(a.prev_bias + c1);
  float4 p0[VC_MAX_KSPLIT], p1[VC_MAX_KSPLIT];
#pragma unroll
  for (int s = 0; s < VC_MAX_KSPLIT; ++s) {
    const float* pp = a.parts + ((long)(s * a.rows_cap + r)) * d;
    p0[s] = *reinterpret_cast<const float4*>(pp + c0);
    p1[s] = *reinterpret_cast<const float4*>(pp + c1);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  if (a.has_prev_bias) {
    x0.x += pb0.x; x0.y += pb0.y; x0.z += pb0.z; x0.w += pb0.w;
    x1.x += pb1.x; x1.y += pb1.y; x1.z += pb1.z; x1.w += pb1.w;
  }
#pragma unroll
  for (int s = 0; s < VC_MAX_KSPLIT; ++s) {
    const bool u = s < a.n_parts;
    x0.x += u ? p0[s].x : 0.f; x0.y += u ? p0[s].y : 0.f; x0.z += u ? p0[s].z : 0.f; x0.w += u ? p0[s].w : 0.f;
    x1.x += u ? p1[s].x : 0.f; x1.y += u ? p1[s].y : 0.f; x1.z += u ? p1[s].z : 0.f; x1.w += u ? p1[s].w : 0.f;
  }
  const float t0 = on0 ? ((x0.x + x0.y) + (x0.z + x0.w)) : 0.f, t1 = on1 ? ((x1.x + x1.y) + (x1.z + x1.w)) : 0.f;
  const float ws = wave_sum(t0 + t1);
  if (lane == 0) s_sum[wave] = ws;
  __syncthreads();
  const float mean = ((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) / (float)d;
  const float dx = x0.x - mean, dy = x0.y - mean, dz = x0.z - mean, dw = x0.w - mean;
  const float ex = x1.x - mean, ey = x1.y - mean, ez = x1.z - mean, ew = x1.w - mean;
  const float q0 = on0 ? ((dx * dx + dy * dy) + (dz * dz + dw * dw)) : 0.f, q1 = on1 ? ((ex * ex + ey * ey) + (ez * ez + ew * ew)) : 0.f;
  const float wq = wave_sum(q0 + q1);
  if (lane == 0) s_sq[wave] = wq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3])) / (float)d + 1e-5f);
  WT* xo = reinterpret_cast<WT*>(a.x_out) + (long)r * d;
  if (on0) {
    if (a.h_out) *reinterpret_cast<float4*>(a.h_out + (long)r * d + c0) = x0;
    f32x4 y = {dx * rstd, dy * rstd, dz * rstd, dw * rstd};
    store4(xo + c0, y);
  }
  if (on1) {
    if (a.h_out) *reinterpret_cast<float4*>(a.h_out + (long)r * d + c1) = x1;
    f32x4 y = {ex * rstd, ey * rstd, ez * rstd, ew * rstd};
    store4(xo + c1, y);
  }
}
hipError_t vc_launch_ln_rows(const GemmArgs& a, int dtype, hipStream_t s) {
  ++vc_launch_counts[VC_LC_LN_ROWS];
  // (round 3's prefetch role of this launch - option ln_pf - left the tree in round 5: it served the slab form of 3..16-row steps, which
  // the finished-row form replaced as the default in round 4, and at 17..64 rows it measured a loss, profiles/r03j_lpf32_ab.log)
  if (dtype == VC_DTYPE_BF16) hipLaunchKernelGGL(ln_rows_k<bf16_t>, dim3(a.n_rows), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(ln_rows_k<float>, dim3(a.n_rows), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ dispatch
size_t vc_gemm_lds_bytes(const GemmArgs& a, int dtype, int ksplit) {
  const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  const size_t xs = (size_t)(a.K / ksplit) * esz + 16;
  return (size_t)a.r_lds * xs + 4 * 64 * sizeof(f32x4) + VC_ROWS * 4 * 3 * sizeof(float);   // X rows, K-reduce area, LN statistics + row means
}

template <typename WT, int KTW, int PRO, int EPI, int NTW, bool NT, bool R2 = false, int NP = VC_MAX_KSPLIT>
static hipError_t launch_dec_nt(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  auto kern = rows_gemm_k<WT, KTW, PRO, EPI, NTW, NT, R2, NP>;
  const size_t lds = vc_gemm_lds_bytes(a, dtype, ksplit) + (size_t)(NTW - 1) * 4 * 64 * sizeof(f32x4);
  if (lds > 64 * 1024) {
    static size_t granted[16] = {0};   // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && lds > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = lds;
    }
  }
  GemmArgs b = a;
  {
    const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
    const int upr = (a.K / ksplit) * esz / 16;
    const int q4 = (a.K / ksplit) / 4;
    b.x_upr_shift = -1;
    b.att_q4_shift = -1;
    for (int sft = 0; sft < 20; ++sft) {
      if ((1 << sft) == upr) b.x_upr_shift = sft;
      if ((1 << sft) == q4) b.att_q4_shift = sft;
    }
  }
  ++vc_launch_counts[VC_LC_ROWS_GEMM];
  hipLaunchKernelGGL(kern, dim3(a.n_tiles / NTW, ksplit, groups), dim3(256 * NTW), lds, s, b);
  return hipGetLastError();
}

template <typename WT, int KTW, int PRO, int EPI, int NTW = 1>
static hipError_t launch_dec(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  if constexpr (PRO == PRO_LN && NTW == 1) {
    // slabs the LayerNorm prologue requests per row: as many as the pass really has (round 5, -1.7..-2.8 % per one-row step; the
    // trimmed forms exist with the non-temporal weight stream only - the comparison arm nt = 0 keeps the four-slab form)
    if (a.nt) {
      if (a.n_parts == 0 && !a.has_prev_bias) return launch_dec_nt<WT, KTW, PRO, EPI, NTW, true, false, 0>(a, dtype, ksplit, groups, s);
      if (a.n_parts <= 2) return launch_dec_nt<WT, KTW, PRO, EPI, NTW, true, false, 2>(a, dtype, ksplit, groups, s);
    }
  }
  if (a.nt) return launch_dec_nt<WT, KTW, PRO, EPI, NTW, true>(a, dtype, ksplit, groups, s);
  return launch_dec_nt<WT, KTW, PRO, EPI, NTW, false>(a, dtype, ksplit, groups, s);
}

template <typename WT, int KTW, int PRO, int EPI, int NTW>
static hipError_t launch_mt_n(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  auto kern = rows_gemm_mt_k<WT, KTW, PRO, EPI, NTW>;
  GemmArgs b = a;
  b.r_lds = VC_ROWS;
  {
    const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
    const int upr = (a.K / ksplit) * esz / 16;
    b.x_upr_shift = -1;
    for (int sft = 0; sft < 20; ++sft)
      if ((1 << sft) == upr) b.x_upr_shift = sft;
  }
  const size_t lds = vc_gemm_lds_bytes(b, dtype, ksplit) + (size_t)(NTW - 1) * 4 * 64 * sizeof(f32x4);
  if (lds > 64 * 1024) {
    static size_t granted[16] = {0};   // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && lds > granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = lds;
    }
  }
  ++vc_launch_counts[NTW == 4 ? VC_LC_MT4 : VC_LC_MT2];
  hipLaunchKernelGGL(kern, dim3((a.n_tiles + NTW - 1) / NTW, ksplit, groups), dim3(256 * NTW), lds, s, b);
  return hipGetLastError();
}

template <typename WT, int KTW, int PRO, int EPI>
static hipError_t launch_mt(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  if constexpr (PRO == PRO_LN) {
    return hipErrorInvalidValue;       // multi-tile passes take their LayerNorm from ln_rows_k
  } else {
    // two weight tiles per workgroup: twice the workgroups of the four-tile form of rounds 2-4 - every CU busy (-2.6 % / -3.0 % per
    // step at 32 / 64 rows, profiles/r05t_mt_tiles_ab.log; the four-tile form left the tree in round 6)
    return launch_mt_n<WT, KTW, PRO, EPI, 2>(a, dtype, ksplit, groups, s);
  }
}

template <typename WT, int KTW, int PRO, int EPI>
static hipError_t launch_one(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  if (a.mt == 2) return launch_mt<WT, KTW, PRO, EPI>(a, dtype, ksplit, groups, s);     // wide decode pass (17..64 rows)
  return launch_dec<WT, KTW, PRO, EPI>(a, dtype, ksplit, groups, s);
}

template <typename WT, int KTW>
static hipError_t launch_ktw(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, int groups,
                             hipStream_t s) {
  if (pro == PRO_LN && epi == EPI_QKV) return launch_one<WT, KTW, PRO_LN, EPI_QKV>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LN && epi == EPI_RELU) return launch_one<WT, KTW, PRO_LN, EPI_RELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LN && epi == EPI_GELU) return launch_one<WT, KTW, PRO_LN, EPI_GELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LNQ) {      // finished-row consumers on the producers' centred copy (round 6): the same three shapes as PRO_LNW below
    if (a.mt == 4 && a.n_tiles % 2 == 0 && groups == 1) {
      if (epi == EPI_QKV) return a.nt ? launch_dec_nt<WT, KTW, PRO_LNQ, EPI_QKV, 2, true, true>(a, dtype, ksplit, groups, s)
                                      : launch_dec_nt<WT, KTW, PRO_LNQ, EPI_QKV, 2, false, true>(a, dtype, ksplit, groups, s);
      if (epi == EPI_RELU) return a.nt ? launch_dec_nt<WT, KTW, PRO_LNQ, EPI_RELU, 2, true, true>(a, dtype, ksplit, groups, s)
                                       : launch_dec_nt<WT, KTW, PRO_LNQ, EPI_RELU, 2, false, true>(a, dtype, ksplit, groups, s);
    }
    if (a.n_rows > VC_FR_MAX_ROWS) return hipErrorInvalidValue;
    if (a.mt == 3 && a.n_tiles % 2 == 0 && groups == 1) {
      if (epi == EPI_QKV) return launch_dec<WT, KTW, PRO_LNQ, EPI_QKV, 2>(a, dtype, ksplit, groups, s);
      if (epi == EPI_RELU) return launch_dec<WT, KTW, PRO_LNQ, EPI_RELU, 2>(a, dtype, ksplit, groups, s);
    }
    if (epi == EPI_QKV) return launch_dec<WT, KTW, PRO_LNQ, EPI_QKV>(a, dtype, ksplit, groups, s);
    if (epi == EPI_RELU) return launch_dec<WT, KTW, PRO_LNQ, EPI_RELU>(a, dtype, ksplit, groups, s);
    if (epi == EPI_GELU) return launch_dec<WT, KTW, PRO_LNQ, EPI_GELU>(a, dtype, ksplit, groups, s);
    return hipErrorInvalidValue;
  }
  if (pro == PRO_LNW && a.mt == 4 && a.n_tiles % 2 == 0 && groups == 1) {     // ... and two rows per wave (9..16 finished rows)
    if (epi == EPI_QKV) return a.nt ? launch_dec_nt<WT, KTW, PRO_LNW, EPI_QKV, 2, true, true>(a, dtype, ksplit, groups, s)
                                    : launch_dec_nt<WT, KTW, PRO_LNW, EPI_QKV, 2, false, true>(a, dtype, ksplit, groups, s);
    if (epi == EPI_RELU) return a.nt ? launch_dec_nt<WT, KTW, PRO_LNW, EPI_RELU, 2, true, true>(a, dtype, ksplit, groups, s)
                                     : launch_dec_nt<WT, KTW, PRO_LNW, EPI_RELU, 2, false, true>(a, dtype, ksplit, groups, s);
  }
  if (pro == PRO_LNW && a.n_rows > VC_FR_MAX_ROWS) return hipErrorInvalidValue;   // (9..16 rows exist only in the two-rows-per-wave form above)
  if (pro == PRO_LNW && a.mt == 3 && a.n_tiles % 2 == 0 && groups == 1) {     // two tiles per workgroup (finished-row consumers, GemmArgs.mt)
    if (epi == EPI_QKV) return launch_dec<WT, KTW, PRO_LNW, EPI_QKV, 2>(a, dtype, ksplit, groups, s);
    if (epi == EPI_RELU) return launch_dec<WT, KTW, PRO_LNW, EPI_RELU, 2>(a, dtype, ksplit, groups, s);
  }
  if (pro == PRO_LNW && epi == EPI_QKV) return launch_dec<WT, KTW, PRO_LNW, EPI_QKV>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LNW && epi == EPI_RELU) return launch_dec<WT, KTW, PRO_LNW, EPI_RELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LNW && epi == EPI_GELU) return launch_dec<WT, KTW, PRO_LNW, EPI_GELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_ATT && epi == EPI_PART) return launch_one<WT, KTW, PRO_ATT, EPI_PART>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_PART) return launch_one<WT, KTW, PRO_PLAIN, EPI_PART>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_LOGITS) return launch_one<WT, KTW, PRO_PLAIN, EPI_LOGITS>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_QKV) return launch_one<WT, KTW, PRO_PLAIN, EPI_QKV>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_QKV16) {      // the 16-channel image: wide decode passes only here (prefill: vc_gemm_pf.hip)
    if (a.mt == 2) return launch_mt<WT, KTW, PRO_PLAIN, EPI_QKV16>(a, dtype, ksplit, groups, s);
    return hipErrorInvalidValue;
  }
  if (pro == PRO_PLAIN && epi == EPI_RELU) return launch_one<WT, KTW, PRO_PLAIN, EPI_RELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_GELU) return launch_one<WT, KTW, PRO_PLAIN, EPI_GELU>(a, dtype, ksplit, groups, s);
  return hipErrorInvalidValue;
}

template <typename WT>
static hipError_t launch_wt(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, int groups,
                            hipStream_t s) {
  // the caller fixed a.nchunk; recover KTW from KT = ksplit * nchunk * 4 * KTW
  const int ktw = a.KT / (ksplit * a.nchunk * 4);
  switch (ktw) {
    case 2: return launch_ktw<WT, 2>(a, dtype, pro, epi, ksplit, groups, s);
    case 4: return launch_ktw<WT, 4>(a, dtype, pro, epi, ksplit, groups, s);
    case 8: return launch_ktw<WT, 8>(a, dtype, pro, epi, ksplit, groups, s);
    case 16: return launch_ktw<WT, 16>(a, dtype, pro, epi, ksplit, groups, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t vc_launch_gemm(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, int groups,
                          hipStream_t s) {
  if (a.mt == 1) {     // prefill pass: block GEMM
    if (groups != 1) return hipErrorInvalidValue;
    return vc_launch_gemm_blk(a, dtype, pro, epi, ksplit, s);      // vc_gemm_pf.hip
  }
  if (dtype == VC_DTYPE_BF16) return launch_wt<bf16_t>(a, dtype, pro, epi, ksplit, groups, s);
  return launch_wt<float>(a, dtype, pro, epi, ksplit, groups, s);
}
