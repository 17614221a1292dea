// vc_gemm_wd.hip - the linear layers of WIDE decode steps (17..64 rows per step: 17..64 utterances decoded together).
//
// Replaces, for those steps, every F.linear of the reference's decoder layer and prediction heads
// (models/modules/activation.py:86, :637; models/modules/transformer.py:386-388; models/voicecraft.py:181-185) that the
// one-row / several-row kernels of vc_gemm.hip serve below 17 rows.
//
// Why a second kernel.  Through round 5 these steps ran on rows_gemm_mt_k (vc_gemm.hip): a workgroup held its weight tiles in
// registers and walked the step's 16-row tiles past them ONE AFTER THE OTHER - per row tile a global -> register -> LDS staging
// of 64 KB of X, three block barriers, 16 MFMAs per wave, a 4-way reduction and an epilogue on one wave.  At 64 rows an FFN
// launch took 17.4-17.8 us for 33.5 MB of weights (1.9 TB/s; profiles/r05s_b64_rocprof_kernel_stats.txt): four dependent
// row-tile phases behind the weight burst, not a stream.
//
// Here ALL row tiles of the step are in flight together:
//   * a workgroup = 8 waves owns TWO 16-channel weight tiles over one K slice (the whole K, or K / ksplit for the split-K
//     producers); the waves split that slice 8 ways: wave w owns k-tiles [w KPW, (w + 1) KPW) of BOTH tiles and of ALL RT row tiles;
//   * a weight fragment is needed by exactly one wave -> one non-temporal 1 KB burst HBM -> registers, as everywhere in this
//     engine (2 KPW fragments per wave: 16 KB at d = 2048, the whole workgroup's 128 KB requested before anything is waited for);
//   * X (the step's normalised rows / activations, RT x 16 rows x the wave's K share) comes out of L2 once per wave, without a block
//     barrier, in one of two forms:
//       rows_gemm_wd_k  - every B fragment (16 rows x 32 k) asked for exactly as the MFMA takes it: one 16-byte load per lane, L2 ->
//                         registers.  Simple, but a request touches SIXTEEN half cache lines, and such an X stream costs a CU as
//                         much per byte as its HBM weight stream (X alone at 64 rows: 9.6 us; the weights alone: 7.4 us);
//       rows_gemm_wds_k - the default: whole lines (a chunk = 2 k-tiles = 128 B per row, 8 rows per request) parked in a wave-private
//                         8 KB LDS stage with XOR-swizzled 16-byte slots and read back as fragments (a wave reads only what it wrote:
//                         LDS ordering within a wave, no barrier); two chunks of requests in flight ahead of the weights;
//   * per wave 2 x RT x KPW MFMAs (64 at 64 rows) into 2 x RT accumulators, then ONE block barrier: the 8 K-partials of every
//     (weight tile, row tile) pair meet in LDS (64 KB at RT = 4) and wave p finishes pair p - the sums in a fixed order, the
//     fused epilogue (bias / ReLU / exact-erf GELU / logits / split-K slab / q + K/V straight into the cache) on 8 waves at once.
//
// Measured on one layer of giga830M, isolated, bf16 (tools/wd_probe.py; profiles/r06a_wd_probe.log, r06c_wd_probe_staged.log), us per launch,
// rows_gemm_mt_k (two tiles) -> rows_gemm_wd_k -> rows_gemm_wds_k:
//   64 rows: FFN-up 14.8 -> 13.5 -> 9.7 | FFN-down 15.9 -> 12.0 -> 9.9 | QKV 13.6 -> 12.6 -> 9.1 | out-projection 9.4 -> 4.9 -> 4.5
//   32 rows: FFN-up 11.0 ->  9.0 -> 8.5 | FFN-down 12.5 ->  9.5 -> 8.7 | QKV 10.4 ->  8.0 -> 7.4 | out-projection 6.8 -> 4.0 -> 3.9
// (9.7 us for 33.5 MB = 3.5 TB/s = 0.43 of the HBM peak; the weight stream alone - X requests compiled out - takes 7.4 us = 0.57.)
// In process on one engine: 64-row step 1.830 -> 1.668 -> 1.509 ms (-8.8 %, then -9.5 %), 32-row step 1.245 -> 1.087 -> 1.045 ms.
//
// HBM-bound (weights read once per step): algorithmic bytes per launch = N K sizeof(WT) (+ rows x (K + N) activations), the same
// figure as the rows-GEMM's.  gfx950 only.
#include <algorithm>
#include "vc_common.h"
#include "vc_gemm_dev.h"

namespace {

template <typename WT> struct WdChunk { static constexpr int X = 8; };     // X fragments of one row tile a wave keeps in flight
template <> struct WdChunk<float> { static constexpr int X = 4; };         // (exact mode: twice the k-tiles per K, 4-float fragments)

template <typename WT, int KPW, int RT, int EPI>
__global__ __launch_bounds__(512) void rows_gemm_wd_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NW = 8;                      // waves per workgroup = K shares
  constexpr int NP = 2 * RT;                 // (weight tile, row tile) pairs of the workgroup
  constexpr int CK = KPW < WdChunk<WT>::X ? KPW : WdChunk<WT>::X;      // k-tiles per chunk (everything of a chunk is requested at once)
  static_assert(KPW % CK == 0, "k-tiles per wave: a whole number of chunks");
  static_assert(EPI != EPI_QKV, "the QKV projection of wide steps reads the 16-channel image (EPI_QKV16)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);                          // [NW][NP][64]

  const int active = *a.n_active;            // scalar; looked at once the first requests are out
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.y, grp = blockIdx.z;
  const int nt0 = blockIdx.x * 2;
  const int n_rows = a.n_rows;
  const int m = lane & 15, kg = lane >> 4;

  // ---- the pair this wave will FINISH (p = wv: weight tile p / RT, row tile p % RT): its epilogue operands are requested first
  const int pt = (wv < NP) ? wv / RT : 0, pr = (wv < NP) ? wv % RT : 0;
  const int n_fin = (nt0 + pt) * 16 + 4 * kg;
  const int mg_fin = pr * 16 + m;
  float4 eb;
  int epos, eseq;
  epi_preload<WT, EPI>(a, min(mg_fin, n_rows - 1), n_fin, grp, eb, epos, eseq);

  // ---- operand addresses: k-tiles [ktb, ktb + KPW) of the matrix; X rows 16 r + m (clamped: rows past the pass re-read its last row)
  const int ktb = (ks * NW + wv) * KPW;
  const uint4* wp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ntc = min(nt0 + t, a.n_tiles - 1);
    wp[t] = a.Wp + (long)grp * a.w_group_stride + ((long)ntc * a.KT + ktb) * 64 + lane;
  }
  const char* xp[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int row = min(r * 16 + m, n_rows - 1);
    xp[r] = reinterpret_cast<const char*>(a.x_in) +
            ((long)grp * a.x_group_stride + (long)row * a.x_ld + (long)ktb * T::KW + kg * T::EPL) * (long)sizeof(WT);
  }

  f32x4 acc[2][RT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int c0 = 0; c0 < KPW; c0 += CK) {
    uint4 xf[RT][CK], wf[2][CK];
    // X first: L2 hits, and a wave's loads return in order - behind the weight burst they would wait for HBM (the other orders
    // - weights first, per k-tile interleaved, row-tile major - measured within +-7 % of this one either way:
    // profiles/r06b_wd_probe_request_order.log)
#pragma unroll
    for (int j = 0; j < CK; ++j)
#pragma unroll
      for (int r = 0; r < RT; ++r)
        xf[r][j] = *reinterpret_cast<const uint4*>(xp[r] + (size_t)(c0 + j) * (T::KW * sizeof(WT)));
#pragma unroll
    for (int j = 0; j < CK; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        wf[t][j] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t] + (size_t)(c0 + j) * 64)));
    if (c0 == 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (active == 0) return;               // a replayed decode step after the last sequence retired
    }
#pragma unroll
    for (int j = 0; j < CK; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[t][r] = mfma_frag(wf[t][j], xf[r][j], acc[t][r], (WT*)nullptr);
  }

  // ---- the 8 K-partials of every pair meet in LDS; wave p finishes pair p
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) red[(wv * NP + t * RT + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  if (wv < NP) {
    f32x4 s[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) s[w] = red[(w * NP + wv) * 64 + lane];
    const f32x4 v = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (mg_fin < n_rows && nt0 + pt < a.n_tiles)
      gemm_epilogue<WT, EPI>(a, v, mg_fin, n_fin, ks, grp, (int)gridDim.z, eb, epos, eseq);
  }
}

// The same workgroup / wave decomposition, X through a wave-private LDS stage (the default form).  The direct form above asks for an X
// fragment the way the MFMA wants it - 16 rows x 64 B per instruction, i.e. SIXTEEN half cache lines - and its X stream costs as much
// per byte as the HBM weight stream (profiles/r06b_wd_probe_request_order.log: +4.5 us per launch for the 128 KB a CU pulls more at 64
// rows than at 32, whatever the request order).  Here a wave requests its share of X as WHOLE lines - a chunk = 2 k-tiles = 128 B per
// row, 8 rows per instruction - parks them in its own 8 KB of LDS (16-byte slots XOR-swizzled by the row so that both the row-major
// stores and the fragment reads - 16 rows x one slot - cover all banks) and reads the B fragments back: no barrier (a wave reads only
// what it wrote), two chunks of requests in flight ahead of the weights.
template <typename WT, int KPW, int RT, int EPI>
__global__ __launch_bounds__(512) void rows_gemm_wds_k(const GemmArgs a) {
  using T = WTr<WT>;
  constexpr int NW = 8, NP = 2 * RT;
  constexpr int NCH = KPW / 2;               // chunks of 2 k-tiles = 128 B per row (32 x 2 x 2 B = 16 x 2 x 4 B)
  constexpr int NQ = 2 * RT;                 // requests per chunk: 8 rows x 128 B each
  constexpr bool WALL = 2 * KPW <= 16;       // all of a wave's weight fragments fit its registers (bf16: always; exact mode up to 8 k-tiles)
  static_assert(KPW % 2 == 0 && T::KW * sizeof(WT) == 64, "a chunk is two k-tiles of 64 B per row");
  static_assert(EPI != EPI_QKV, "the QKV projection of wide steps reads the 16-channel image (EPI_QKV16)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = reinterpret_cast<f32x4*>(smem);                          // [NW][NP][64]
  const int active = *a.n_active;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* stage = smem + (size_t)NW * NP * 64 * sizeof(f32x4) + (size_t)wv * (RT * 16 * 128);      // this wave's [RT x 16 rows][128 B]
  const int ks = blockIdx.y, grp = blockIdx.z;
  const int nt0 = blockIdx.x * 2;
  const int n_rows = a.n_rows;
  const int m = lane & 15, kg = lane >> 4;
  const int pt = (wv < NP) ? wv / RT : 0, pr = (wv < NP) ? wv % RT : 0;
  const int n_fin = (nt0 + pt) * 16 + 4 * kg;
  const int mg_fin = pr * 16 + m;
  float4 eb;
  int epos, eseq;
  epi_preload<WT, EPI>(a, min(mg_fin, n_rows - 1), n_fin, grp, eb, epos, eseq);

  const int ktb = (ks * NW + wv) * KPW;
  const uint4* wp[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ntc = min(nt0 + t, a.n_tiles - 1);
    wp[t] = a.Wp + (long)grp * a.w_group_stride + ((long)ntc * a.KT + ktb) * 64 + lane;
  }
  // request q of a chunk: rows 8 q + (lane >> 3), 16-byte slot lane & 7 of the row's 128 B.
  // (Everything per request is a NAMED scalar, generated by the list macro below: arrays of fragments that are parked in LDS - xa[q] -
  // are demoted to scratch memory by the compiler, however constant the index; the one-row kernels of vc_gemm.hip hit the same wall.)
  const int lr = lane >> 3, lc = lane & 7;
  const char* xbase = reinterpret_cast<const char*>(a.x_in) + ((long)grp * a.x_group_stride + (long)ktb * T::KW) * (long)sizeof(WT) + lc * 16;
  const long xrs = (long)a.x_ld * (long)sizeof(WT);
#define VC_WDS_Q(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define VC_WDS_DECL(q) const char* xq##q = xbase + (long)min((q) * 8 + lr, n_rows - 1) * xrs; uint4 xa##q, xb##q;
  VC_WDS_Q(VC_WDS_DECL)
  char* st_w = stage + lr * 128 + ((lc ^ lr) << 4);                                   // + q * 1024
  const char* st_r0 = stage + m * 128 + ((kg ^ (m & 7)) << 4);                          // + r * 2048: slot kg (first k-tile of the chunk)
  const char* st_r1 = stage + m * 128 + (((kg | 4) ^ (m & 7)) << 4);                    // slot 4 + kg (second k-tile)

  f32x4 acc[2][RT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 wf[2][WALL ? KPW : 2];
#define VC_WDS_LA(q) if constexpr ((q) < NQ) xa##q = *reinterpret_cast<const uint4*>(xq##q + coff);
#define VC_WDS_LB(q) if constexpr ((q) < NQ) xb##q = *reinterpret_cast<const uint4*>(xq##q + coff);
#define VC_WDS_PA(q) if constexpr ((q) < NQ) *reinterpret_cast<uint4*>(st_w + (q) * 1024) = xa##q;
#define VC_WDS_PB(q) if constexpr ((q) < NQ) *reinterpret_cast<uint4*>(st_w + (q) * 1024) = xb##q;
#define VC_WDS_LW(j_, slot_) _Pragma("unroll") for (int t = 0; t < 2; ++t) \
    wf[t][slot_] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t] + (size_t)(j_) * 64)))
  { constexpr size_t coff = 0; VC_WDS_Q(VC_WDS_LA) }
  if constexpr (NCH > 1) { constexpr size_t coff = 128; VC_WDS_Q(VC_WDS_LB) }
  if constexpr (WALL) {
#pragma unroll
    for (int j = 0; j < KPW; ++j) VC_WDS_LW(j, j);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  // one chunk: park it (whole lines -> swizzled slots), re-arm its registers with the chunk after next, read the fragments back, multiply
#define VC_WDS_STEP(S_, c_)                                                                                        \
  {                                                                                                                \
    if constexpr (!WALL) { VC_WDS_LW(2 * (c_), 0); VC_WDS_LW(2 * (c_) + 1, 1); }                                   \
    VC_WDS_Q(VC_WDS_P##S_)                                                                                         \
    if constexpr ((c_) + 2 < NCH) { constexpr size_t coff = (size_t)((c_) + 2) * 128; VC_WDS_Q(VC_WDS_L##S_) }     \
    uint4 f0[RT], f1[RT];                                                                                          \
    _Pragma("unroll") for (int r = 0; r < RT; ++r) {                                                               \
      f0[r] = *reinterpret_cast<const uint4*>(st_r0 + r * 2048);                                                   \
      f1[r] = *reinterpret_cast<const uint4*>(st_r1 + r * 2048);                                                   \
    }                                                                                                              \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
      _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                               \
        acc[t][r] = mfma_frag(wf[t][WALL ? 2 * (c_) : 0], f0[r], acc[t][r], (WT*)nullptr);                         \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
      _Pragma("unroll") for (int r = 0; r < RT; ++r)                                                               \
        acc[t][r] = mfma_frag(wf[t][WALL ? 2 * (c_) + 1 : 1], f1[r], acc[t][r], (WT*)nullptr);                     \
  }
  static_assert(NCH <= 8, "chunks per wave");
  if constexpr (NCH > 0) VC_WDS_STEP(A, 0)
  if constexpr (NCH > 1) VC_WDS_STEP(B, 1)
  if constexpr (NCH > 2) VC_WDS_STEP(A, 2)
  if constexpr (NCH > 3) VC_WDS_STEP(B, 3)
  if constexpr (NCH > 4) VC_WDS_STEP(A, 4)
  if constexpr (NCH > 5) VC_WDS_STEP(B, 5)
  if constexpr (NCH > 6) VC_WDS_STEP(A, 6)
  if constexpr (NCH > 7) VC_WDS_STEP(B, 7)
#undef VC_WDS_STEP
#undef VC_WDS_LW
#undef VC_WDS_PA
#undef VC_WDS_PB
#undef VC_WDS_LA
#undef VC_WDS_LB
#undef VC_WDS_DECL
#undef VC_WDS_Q
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < RT; ++r) red[(wv * NP + t * RT + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  if (wv < NP) {
    f32x4 s[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) s[w] = red[(w * NP + wv) * 64 + lane];
    const f32x4 v = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    if (mg_fin < n_rows && nt0 + pt < a.n_tiles)
      gemm_epilogue<WT, EPI>(a, v, mg_fin, n_fin, ks, grp, (int)gridDim.z, eb, epos, eseq);
  }
}

template <typename WT, int KPW, int RT, int EPI>
hipError_t launch_wds(const GemmArgs& a, int ksplit, int groups, hipStream_t s) {
  auto kern = rows_gemm_wds_k<WT, KPW, RT, EPI>;
  const size_t lds = (size_t)8 * 2 * RT * 64 * sizeof(f32x4) + (size_t)8 * RT * 16 * 128;
  {
    static bool granted[16] = {false};       // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (lds >= 64 * 1024 && dev >= 0 && dev < 16 && !granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = true;
    }
  }
  ++vc_launch_counts[VC_LC_WD];
  hipLaunchKernelGGL(kern, dim3((a.n_tiles + 1) / 2, ksplit, groups), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename WT, int KPW, int RT, int EPI>
hipError_t launch_wd(const GemmArgs& a, int ksplit, int groups, hipStream_t s) {
  if constexpr (KPW >= 2) {
    if (a.wd_stage) return launch_wds<WT, KPW, RT, EPI>(a, ksplit, groups, s);      // X through the wave-private LDS stage (the default)
  }
  auto kern = rows_gemm_wd_k<WT, KPW, RT, EPI>;
  const size_t lds = (size_t)8 * 2 * RT * 64 * sizeof(f32x4);
  if (lds >= 64 * 1024) {
    static bool granted[16] = {false};       // per instantiation and device
    int dev = 0;
    if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
    if (dev >= 0 && dev < 16 && !granted[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted[dev] = true;
    }
  }
  ++vc_launch_counts[VC_LC_WD];
  hipLaunchKernelGGL(kern, dim3((a.n_tiles + 1) / 2, ksplit, groups), dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename WT, int KPW, int RT>
hipError_t launch_wd_epi(const GemmArgs& a, int epi, int ksplit, int groups, hipStream_t s) {
  switch (epi) {
    case EPI_QKV16: return launch_wd<WT, KPW, RT, EPI_QKV16>(a, ksplit, groups, s);
    case EPI_PART: return launch_wd<WT, KPW, RT, EPI_PART>(a, ksplit, groups, s);
    case EPI_RELU: return launch_wd<WT, KPW, RT, EPI_RELU>(a, ksplit, groups, s);
    case EPI_GELU: return launch_wd<WT, KPW, RT, EPI_GELU>(a, ksplit, groups, s);
    case EPI_LOGITS: return launch_wd<WT, KPW, RT, EPI_LOGITS>(a, ksplit, groups, s);
    default: return hipErrorInvalidValue;
  }
}

template <typename WT, int KPW>
hipError_t launch_wd_rt(const GemmArgs& a, int epi, int ksplit, int groups, hipStream_t s) {
  // row tiles in flight: 2 up to 32 rows, 4 beyond (a third / fourth tile without rows re-reads the pass's last row and stores nothing)
  if (a.n_rows <= 2 * VC_ROWS) return launch_wd_epi<WT, KPW, 2>(a, epi, ksplit, groups, s);
  return launch_wd_epi<WT, KPW, 4>(a, epi, ksplit, groups, s);
}

}  // namespace

// k-tiles per wave for this K slice, or 0 when the kernel has no form for it (the slice must split 8 ways into 1, 2, 4, 8 or - exact
// mode - 16 k-tiles).  16-channel tiles only.
int vc_gemm_wd_kpw(int K, int dtype, int ksplit) {
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16;
  if (ksplit < 1 || K % (KW * 8 * ksplit) != 0) return 0;
  const int kpw = K / (KW * 8 * ksplit);
  if (kpw == 1 || kpw == 2 || kpw == 4 || kpw == 8) return kpw;
  if (kpw == 16 && dtype == VC_DTYPE_F32) return kpw;
  return 0;
}

// Launch of one wide-decode linear layer: GemmArgs as vc_launch_gemm takes them (plain prologue: X = x_in), n_rows in 17..64,
// a.n_tiles 16-channel tiles, a.KT k-tiles in all; ksplit = K slices (EPI_PART: one slab each), groups = grid.z (heads-2).
hipError_t vc_launch_gemm_wd(const GemmArgs& a, int dtype, int epi, int ksplit, int groups, hipStream_t s) {
  if (a.n_rows < 1 || a.n_rows > 4 * VC_ROWS) return hipErrorInvalidValue;
  const int kpw = vc_gemm_wd_kpw(a.K, dtype, ksplit);
  if (dtype == VC_DTYPE_BF16) {
    switch (kpw) {
      case 1: return launch_wd_rt<bf16_t, 1>(a, epi, ksplit, groups, s);
      case 2: return launch_wd_rt<bf16_t, 2>(a, epi, ksplit, groups, s);
      case 4: return launch_wd_rt<bf16_t, 4>(a, epi, ksplit, groups, s);
      case 8: return launch_wd_rt<bf16_t, 8>(a, epi, ksplit, groups, s);
      default: return hipErrorInvalidValue;
    }
  }
  switch (kpw) {
    case 1: return launch_wd_rt<float, 1>(a, epi, ksplit, groups, s);
    case 2: return launch_wd_rt<float, 2>(a, epi, ksplit, groups, s);
    case 4: return launch_wd_rt<float, 4>(a, epi, ksplit, groups, s);
    case 8: return launch_wd_rt<float, 8>(a, epi, ksplit, groups, s);
    case 16: return launch_wd_rt<float, 16>(a, epi, ksplit, groups, s);
    default: return hipErrorInvalidValue;
  }
}
