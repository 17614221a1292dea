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
// Here ALL row tiles are in flight together and nothing is staged:
//   * a workgroup = 8 waves owns TWO 16-channel weight tiles over one K slice (the whole K, or K / ksplit for the split-K
//     producers); the waves split that slice 8 ways: wave w owns k-tiles [w KPW, (w + 1) KPW) of BOTH tiles and of ALL RT row tiles;
//   * a weight fragment is needed by exactly one wave -> one non-temporal 1 KB burst HBM -> registers, as everywhere in this
//     engine (2 KPW fragments per wave: 16 KB at d = 2048, the whole workgroup's 128 KB requested before anything is waited for);
//   * an X fragment (16 rows x 32 k) is 16 row segments of 64 B that the MFMA's B operand takes exactly as they lie in the
//     normalised-row / activation buffer -> one plain 16-byte load per lane, L2 -> registers, no LDS, no barrier; they are
//     requested AHEAD of the weights (a wave's loads return in order: the L2 hits must not queue behind the HBM burst);
//   * per wave 2 x RT x KPW MFMAs (64 at 64 rows) into 2 x RT accumulators, then ONE block barrier: the 8 K-partials of every
//     (weight tile, row tile) pair meet in LDS (64 KB at RT = 4) and wave p finishes pair p - the sums in a fixed order, the
//     fused epilogue (bias / ReLU / exact-erf GELU / logits / split-K slab / q + K/V straight into the cache) on 8 waves at once.
// Per launch a CU pulls RT x 16 rows x Kslice of X out of L2 (256 KB at 64 rows, d = 2048) next to its 128 KB of weights out
// of HBM: 64 MB of L2 reads per launch, a quarter of what the L2s deliver in the time the weights take.
//
// HBM-bound (weights read once per step): algorithmic bytes per launch = N K sizeof(WT) (+ rows x N outputs), the same
// figure as the rows-GEMM's.  gfx950 only.
#include <algorithm>
#include "vc_common.h"
#include "vc_gemm_dev.h"

namespace {

template <typename WT> struct WdChunk { static constexpr int X = 8; };     // X fragments of one row tile a wave keeps in flight
template <> struct WdChunk<float> { static constexpr int X = 4; };         // (exact mode: twice the k-tiles per K, 4-float fragments)

// ORD: issue order of a chunk's requests (measured in process, tools/wd_probe.py): 0 = X (k-tile major) then W; 1 = X (row-tile major:
// the 8 k-tiles of a row tile are 512 contiguous bytes per row) then W; 2 = per k-tile W then X; 3 = W then X (row-tile major)
template <typename WT, int KPW, int RT, int EPI, int ORD = 0>
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
#define VC_WD_LX(r_, j_) xf[r_][j_] = *reinterpret_cast<const uint4*>(xp[r_] + (size_t)(c0 + (j_)) * (T::KW * sizeof(WT)))
#define VC_WD_LW(t_, j_) wf[t_][j_] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t_] + (size_t)(c0 + (j_)) * 64)))
    if constexpr (ORD == 0) {
      // X first: L2 hits, and a wave's loads return in order - behind the weight burst they would wait for HBM
#pragma unroll
      for (int j = 0; j < CK; ++j)
#pragma unroll
        for (int r = 0; r < RT; ++r) VC_WD_LX(r, j);
#pragma unroll
      for (int j = 0; j < CK; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) VC_WD_LW(t, j);
    } else if constexpr (ORD == 1) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < CK; ++j) VC_WD_LX(r, j);
#pragma unroll
      for (int j = 0; j < CK; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) VC_WD_LW(t, j);
    } else if constexpr (ORD == 2) {
#pragma unroll
      for (int j = 0; j < CK; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) VC_WD_LW(t, j);
#pragma unroll
        for (int r = 0; r < RT; ++r) VC_WD_LX(r, j);
      }
    } else if constexpr (ORD == 3) {
#pragma unroll
      for (int j = 0; j < CK; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) VC_WD_LW(t, j);
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < CK; ++j) VC_WD_LX(r, j);
    } else {        // timing probes, WRONG results by design: 4 = no X requests (the weight stream alone), 5 = no weight requests (X alone)
#pragma unroll
      for (int j = 0; j < CK; ++j) {
#pragma unroll
        for (int r = 0; r < RT; ++r) { if constexpr (ORD == 5) VC_WD_LX(r, j); else xf[r][j] = make_uint4(lane, j, r, 1u); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { if constexpr (ORD == 4) VC_WD_LW(t, j); else wf[t][j] = make_uint4(lane, j, t, 1u); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // (the requests leave in the order written)
#undef VC_WD_LX
#undef VC_WD_LW
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

template <typename WT, int KPW, int RT, int EPI, int ORD = 0>
hipError_t launch_wd(const GemmArgs& a, int ksplit, int groups, hipStream_t s) {
  if constexpr (ORD == 0 && sizeof(WT) == 2 && KPW >= 2) {       // measurement builds: the request order is selectable (GemmArgs.mt_ntw = 1..3)
    if (a.mt_ntw == 1) return launch_wd<WT, KPW, RT, EPI, 1>(a, ksplit, groups, s);
    if (a.mt_ntw == 2) return launch_wd<WT, KPW, RT, EPI, 2>(a, ksplit, groups, s);
    if (a.mt_ntw == 3) return launch_wd<WT, KPW, RT, EPI, 3>(a, ksplit, groups, s);
    if constexpr (KPW == 8 && (EPI == EPI_RELU || EPI == EPI_PART)) {
      if (a.mt_ntw == 4) return launch_wd<WT, KPW, RT, EPI, 4>(a, ksplit, groups, s);
      if (a.mt_ntw == 5) return launch_wd<WT, KPW, RT, EPI, 5>(a, ksplit, groups, s);
    }
  }
  auto kern = rows_gemm_wd_k<WT, KPW, RT, EPI, ORD>;
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
