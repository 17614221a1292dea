// vc_gemm_pf.hip - the prefill GEMMs: out[M][N] = X W'^T on the MFMA for passes of up to VC_MAX_ROWS rows (rows_gemm_blk_k: 128-row
// tiles, rows_gemm_big_k: 256 x 256 tiles with both operands by LDS-DMA).  Same packed weights, same epilogues as the decode
// rows-GEMM of vc_gemm.hip (which dispatches here for GemmArgs.mt == 1); a translation unit of its own since round 5 so that the
// two halves compile in parallel.
#include "vc_gemm_dev.h"

// ------------------------------------------------------------------ prefill: block GEMM, up to VC_MAX_ROWS rows per pass
// out[M][N] = X[M][K] W'[N][K]^T on the MFMA for M >> 16 (prompt rows of one or several sequences).  The weights
// keep the decode layout - they are already MFMA A fragments in HBM, so a wave loads them straight into
// registers (16 bytes per lane per fragment, no LDS, no transposition); only X goes through LDS.
//   workgroup  = 4 waves as 2 (rows) x 2 (channels); tile = 128 rows x 4 weight tiles (64 channels; 48 for QKV):
//                a 231-row prompt still gives 256 workgroups on the two wide matrices; 68 KB of LDS and <= 256
//                registers, so two workgroups share a CU (one in the MFMAs while the other waits for memory)
//   wave       = 4 row tiles x 2 weight tiles: per k-tile 2 A fragments (global) + 4 B fragments (ds_read_b128)
//                feed 8 MFMAs
//   K pipeline = chunks of 4 k-tiles.  Weight fragments run through a ring of register sets (three on the 2 x 2 form:
//                requested two chunks = 16 KB per wave ahead of their MFMAs; two on the side-by-side form, whose
//                register file holds no third) - at a few hundred rows the pass is as much a weight stream (HBM) as
//                a GEMM.  X rows of the next chunk (8 x 16 B per thread) are requested BEFORE that chunk's weights
//                (a wave's loads return in order: waiting for X must not drain the weight ring) and parked in the
//                other LDS buffer during the chunk's last k-tile.  Inside a chunk the four instruction classes are
//                interleaved explicitly (VC_BLK_SCHED below): one wave per SIMD issues in order, so anything
//                issued back to back also RUNS back to back.
// LDS: 2 x 128 rows x (256 B + 16 B pad) = 68 KB; the pad rotates rows by 4 banks.  Epilogues are the decode ones
// (bias/ReLU, split-K slab, QKV with the cache scatter), LayerNorm comes from ln_rows_k + the folded weights.
// NTW = weight tiles per wave: 2 (64-channel workgroup tile, enough workgroups for one short prompt) or 4 (128
// channels: each X byte staged in LDS feeds twice as many MFMAs - the L2->LDS traffic of X is what bounds the
// 64-channel form once the grid is large enough).
int vc_blk_dbg_mask = 0;     // diagnostic mask of the block GEMM (see the kernel); only the kernel microbenchmark sets it
long long vc_launch_counts[VC_LC_N] = {0};
#define VC_BLK_M 128
#define VC_BLK_KT 4          // k-tiles per pipeline chunk
template <typename WT, int EPI, int NTW, int WM, int OCC>
__global__ __launch_bounds__(256, OCC) void rows_gemm_blk_k(const GemmArgs a) {
  constexpr int WN = 4 / WM;                                    // waves along the channels
  constexpr int MT = 8 / WM;                                    // 16-row tiles per wave
  using T = WTr<WT>;
  constexpr int TH = (EPI == EPI_QKV) ? VC_TH_QKV : 16;
  constexpr int SPT = 4 * TH;
  constexpr int ROWB = VC_BLK_KT * T::KW * (int)sizeof(WT);     // bytes of one X row per chunk (256)
  constexpr int XS = ROWB + 16;                                 // LDS row stride
  constexpr int UPR = ROWB / 16;                                // 16-byte units per row per chunk (16)
  constexpr int XPT = VC_BLK_M * UPR / 256;                     // units per thread per chunk (8)
  constexpr int RPJ = 256 / UPR;                                // rows between a thread's consecutive units (16)
  static_assert(XPT == 8, "the X staging below is written for 8 units per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (*a.n_active == 0) return;                                 // a replayed decode step after the last sequence retired
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (WM == 2) ? (wv >> 1) : 0, wn = (WM == 2) ? (wv & 1) : wv;   // wave's row part / channel part
  const int m = lane & 15, kg = lane >> 4;
  const int row_blk = blockIdx.y * VC_BLK_M;
  const int nt0 = (blockIdx.x * WN + wn) * NTW;                 // first weight tile of this wave
  const int ks = blockIdx.z;
  const int n_rows = a.n_rows;
  const int kt_blk = a.KT / (int)gridDim.z;                     // k-tiles this workgroup covers
  const int kt0 = ks * kt_blk;
  const int nck = kt_blk / VC_BLK_KT;
  const bool wvalid = m < TH;
  // diagnostic mask (VC_BLK_DBG, 0 in production): bit 0 = every chunk re-reads the weights of chunk 0, bit 1 = the X
  // of chunk 0 (wrong results; tells a launch bound by the weight stream from one bound by the X traffic)
  const int wlast = (a.att_q4_shift & 1) ? 0 : nck - 1, xlast = (a.att_q4_shift & 2) ? 0 : nck - 1;
  // weight fragments of the wave's 2 tiles: tile j is j * KT * SPT units further (n_tiles is a multiple of
  // VC_BLK_NT for every matrix of the path: d % 256 == 0)
  const uint4* wp0 = a.Wp + ((long)nt0 * a.KT + kt0) * SPT + (kg * TH + min(m, TH - 1));
  const long wtile = (long)a.KT * SPT;
  // X source: thread t copies the 16-byte units t, t+256, ... of a chunk's 128 x UPR unit grid, i.e. row
  // tid/UPR + RPJ j, unit tid%UPR.  Rows past n_rows are read from the (VC_MAX_ROWS-row) buffer and dropped
  // by the epilogue - rows never mix in a GEMM.
  const long rstride = (long)a.x_ld * (long)sizeof(WT);
  const char* xg0 = reinterpret_cast<const char*>(a.x_in) + (long)kt0 * T::KW * (long)sizeof(WT) +
                    (long)(row_blk + tid / UPR) * rstride + (tid % UPR) * 16;
  const int xl0 = (tid / UPR) * XS + (tid % UPR) * 16;
  constexpr bool RING3 = (WM == 2 && NTW == 2 && OCC == 1);     // a third weight set fits the register file
  uint4 w0[VC_BLK_KT][NTW], w1[VC_BLK_KT][NTW], w2[RING3 ? VC_BLK_KT : 1][NTW];     // the weight ring
  uint4 xr0, xr1, xr2, xr3, xr4, xr5, xr6, xr7;   // explicit scalars: an indexed array living across the loop is demoted to scratch
  f32x4 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (chunk indices past the end are clamped: a redundant load is cheaper than a branch around the burst)
#define VC_BLK_LOADW(W, c_)                                                                      \
  {                                                                                              \
    const long cw_ = (long)min((c_), wlast) * VC_BLK_KT * SPT;                                    \
    _Pragma("unroll") for (int kt_ = 0; kt_ < VC_BLK_KT; ++kt_)                                   \
      _Pragma("unroll") for (int j_ = 0; j_ < NTW; ++j_) {                                        \
        W[kt_][j_] = wp0[j_ * wtile + cw_ + kt_ * SPT];                                          \
      }                                                                                          \
  }
#define VC_BLK_LX1(j_) xr##j_ = *reinterpret_cast<const uint4*>(xc_ + (long)j_ * RPJ * rstride);
#define VC_BLK_LOADX(c_)                                                                         \
  {                                                                                              \
    const char* xc_ = xg0 + (long)min((c_), xlast) * ROWB;                                        \
    VC_BLK_LX1(0) VC_BLK_LX1(1) VC_BLK_LX1(2) VC_BLK_LX1(3) VC_BLK_LX1(4) VC_BLK_LX1(5) VC_BLK_LX1(6) VC_BLK_LX1(7) \
  }
#define VC_BLK_PX1(j_, buf_) *reinterpret_cast<uint4*>(smem + (buf_) * (VC_BLK_M * XS) + xl0 + j_ * RPJ * XS) = xr##j_;
#define VC_BLK_PARKX(c_, buf_)                                                                   \
  {                                                                                              \
    VC_BLK_PX1(0, buf_) VC_BLK_PX1(1, buf_) VC_BLK_PX1(2, buf_) VC_BLK_PX1(3, buf_)               \
    VC_BLK_PX1(4, buf_) VC_BLK_PX1(5, buf_) VC_BLK_PX1(6, buf_) VC_BLK_PX1(7, buf_)               \
  }
  // A wave is alone on its SIMD and issues in order: whatever it issues back to back - 16 global loads (16 clocks of
  // address path each), 32 ds_read_b128, 8 ds_write_b128 (13 clocks each), 64 MFMAs (16 clocks each) - runs back to
  // back, and a chunk costs the SUM of the four (measured: 2 800 clocks per chunk with every load an L1/L2 hit, the
  // MFMAs being 1 024 of them).  So the step is one scheduling region with an explicit interleave
  // (sched_group_barrier): the first k-tile's X fragments; then per NTW MFMAs one ds_read of the NEXT k-tile's
  // fragments (two register sets) and, during the first two k-tiles, the step's global loads (next chunk's X
  // first, then the weights AHEAD chunks on); the last k-tile's MFMAs carry the ds_writes that park the next X.
#define VC_BLK_LDX(dst_, kt_)                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                               \
    dst_[i_] = *reinterpret_cast<const uint4*>(xb_ + i_ * 16 * XS + (kt_) * 64);
#define VC_BLK_MM(W, src_, kt_)                                                                  \
  _Pragma("unroll") for (int j_ = 0; j_ < NTW; ++j_) {                                            \
    uint4 w_ = W[kt_][j_];                                                                       \
    if (TH < 16 && !wvalid) w_ = make_uint4(0u, 0u, 0u, 0u);                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < MT; ++i_)                                             \
      acc[i_][j_] = mfma_frag(w_, src_[i_], acc[i_][j_], (WT*)nullptr);                           \
  }
#define VC_BLK_COMPUTE(W, buf_)                                                                  \
  {                                                                                              \
    static_assert(VC_BLK_KT == 4, "unrolled by hand for 4 k-tiles per chunk");                   \
    const char* xb_ = smem + (buf_) * (VC_BLK_M * XS) + (wm * (16 * MT) + m) * XS + kg * 16;      \
    uint4 xa_[MT], xd_[MT];                                                                      \
    VC_BLK_LDX(xa_, 0)                                                                           \
    VC_BLK_LDX(xd_, 1)                                                                           \
    VC_BLK_MM(W, xa_, 0)                                                                         \
    VC_BLK_LDX(xa_, 2)                                                                           \
    VC_BLK_MM(W, xd_, 1)                                                                         \
    VC_BLK_LDX(xd_, 3)                                                                           \
    VC_BLK_MM(W, xa_, 2)                                                                         \
    VC_BLK_MM(W, xd_, 3)                                                                         \
  }
  // instruction classes of sched_group_barrier: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
#define VC_BLK_SCHED()                                                                           \
  {                                                                                              \
    constexpr int VPS_ = (8 + VC_BLK_KT * NTW) / (2 * MT);   /* global loads per slot, first two k-tiles */ \
    __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);                                          \
    _Pragma("unroll") for (int n_ = 0; n_ < 2 * MT; ++n_) {                                       \
      __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
      __builtin_amdgcn_sched_group_barrier(0x020, VPS_, 0);                                      \
    }                                                                                            \
    _Pragma("unroll") for (int n_ = 0; n_ < MT; ++n_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
    }                                                                                            \
    _Pragma("unroll") for (int n_ = 0; n_ < MT; ++n_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x200, 8 / MT, 0);                                    \
    }                                                                                            \
  }
  // one pipeline step: chunk c is computed from ring slot WC / LDS buffer BUF, chunk c+1's X and chunk c+2's
  // weights (into slot WN, free since chunk c-1) are requested first - X before W, see above
#define VC_BLK_STEP(c_, WC, WN, BUF, AHEAD)                                                      \
  {                                                                                              \
    VC_BLK_LOADX((c_) + 1)                                                                       \
    VC_BLK_LOADW(WN, (c_) + (AHEAD))                                                             \
    VC_BLK_COMPUTE(WC, BUF)                                                                      \
    VC_BLK_PARKX((c_) + 1, 1 - (BUF))                                                            \
    VC_BLK_SCHED()                                                                               \
    __syncthreads();                                                                             \
  }
  if constexpr (RING3) {        // ring of three sets: weights two chunks ahead
    VC_BLK_LOADX(0)
    VC_BLK_LOADW(w0, 0)
    VC_BLK_LOADW(w1, 1)
    VC_BLK_PARKX(0, 0)
    __syncthreads();
    for (int c = 0; c < nck; c += 6) {      // 6 = lcm(ring of 3, 2 LDS buffers)
      VC_BLK_STEP(c, w0, w2, 0, 2)
      if (c + 1 >= nck) break;
      VC_BLK_STEP(c + 1, w1, w0, 1, 2)
      if (c + 2 >= nck) break;
      VC_BLK_STEP(c + 2, w2, w1, 0, 2)
      if (c + 3 >= nck) break;
      VC_BLK_STEP(c + 3, w0, w2, 1, 2)
      if (c + 4 >= nck) break;
      VC_BLK_STEP(c + 4, w1, w0, 0, 2)
      if (c + 5 >= nck) break;
      VC_BLK_STEP(c + 5, w2, w1, 1, 2)
    }
  } else {                      // 128-channel tile: two sets (the register file holds no third), one chunk ahead
    VC_BLK_LOADX(0)
    VC_BLK_LOADW(w0, 0)
    VC_BLK_PARKX(0, 0)
    __syncthreads();
    for (int c = 0; c < nck; c += 2) {
      VC_BLK_STEP(c, w0, w1, 0, 1)
      if (c + 1 >= nck) break;
      VC_BLK_STEP(c + 1, w1, w0, 1, 1)
    }
  }
#undef VC_BLK_STEP
#undef VC_BLK_LOADW
#undef VC_BLK_LOADX
#undef VC_BLK_PARKX
#undef VC_BLK_LX1
#undef VC_BLK_PX1
#undef VC_BLK_COMPUTE
#undef VC_BLK_LDX
#undef VC_BLK_MM
#undef VC_BLK_SCHED
  // ---- epilogue (tile_epilogue: operands, then values, then the stores back to back)
  const bool nvalid = 4 * kg < TH;
  float4 ebias[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int nt = min(nt0 + j, a.n_tiles - 1);
    int p0, s0;
    epi_preload<WT, EPI>(a, 0, nt * TH + (nvalid ? 4 * kg : 0), 0, ebias[j], p0, s0);
  }
  tile_epilogue<WT, EPI, MT, NTW>(a, acc, ebias, row_blk + wm * (16 * MT) + m, nt0, TH, kg, ks, n_rows);
}

// ------------------------------------------------------------------ prefill, long row streams: 256 x 256 tiles, everything through LDS-DMA
// The 128 x 128 block GEMM above is bound by what its workgroups pull out of L2: every weight byte is fetched once per
// 128 rows and every X byte once per 128 channels - 268 MB per 512-row FFN-up launch, ~11 TB/s, 0.25 of the bf16 MFMA peak
// whatever the schedule (profiles/r02_blk_probe.log).  A pass of >= 768 rows (several prompts of a batch as one row
// stream, a long editing prompt) can afford a 256 x 256 tile on every CU, which halves the bytes per FLOP:
//   workgroup  8 waves side by side: wave w owns ALL 16 row tiles of the 256-row block and weight tiles 2w, 2w+1 of its
//              16 (256 channels; 192 for the 12-channel QKV tiles) - 32 accumulators of 16 x 16, two waves per SIMD
//   staging    BOTH operands by LDS-DMA (global_load_lds_dwordx4), one 1 KB MFMA fragment per instruction: the weights
//              already ARE fragments in HBM; an X fragment is 16 rows x 64 B, its lane order IS the B operand's, so the
//              LDS image is fragment-linear and every ds_read_b128 is conflict-free.  Nothing passes through registers on
//              the way in (the compiler would drain the DMA queue before every use of an ordinary load next to it).
//   pipeline   4 stages of one k-tile (16 X + 16 W fragments = 32 KB), three in flight: per k-tile a wave issues its 4
//              DMA instructions, waits with a COUNTED vmcnt for the stage issued three steps ago, meets the workgroup at
//              one raw s_barrier and feeds 32 MFMAs from 18 fragment reads
// Epilogues are the decode ones (gemm_epilogue).  bf16 only (the exact fp32 mode keeps the 128 x 128 kernel).
#define VC_BIG_M 256
#define VC_BIG_STAGES 4
#define VC_BIG_STAGE_BYTES 32768
template <int EPI, int NTW>      // NTW weight tiles per wave: 2 = 256-channel tile (the only form dispatched, see launch_blk)
__global__ __launch_bounds__(512) void rows_gemm_big_k(const GemmArgs a) {
  using WT = bf16_t;
  constexpr int TH = (EPI == EPI_QKV) ? VC_TH_QKV : 16;
  constexpr int SPT = 4 * TH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (*a.n_active == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int row_blk = blockIdx.y * VC_BIG_M;
  const int nt_blk = blockIdx.x * 8 * NTW;
  const int ks = blockIdx.z;
  const int kt_blk = a.KT / (int)gridDim.z, kt0 = ks * kt_blk;
  const bool wvalid = m < TH;
  // DMA sources of this wave: X row tiles 2 wv, 2 wv + 1 and its own NTW weight tiles of every stage
  const long rstride = (long)a.x_ld * 2;
  const char* xsrc[2];
  const uint4* wsrc[NTW];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = row_blk + (2 * wv + j) * 16 + m;            // (rows past n_rows exist in the VC_MAX_ROWS-row buffer; dropped by the epilogue)
    xsrc[j] = reinterpret_cast<const char*>(a.x_in) + (long)row * rstride + ((long)kt0 * 32 + 8 * kg) * 2;
  }
#pragma unroll
  for (int j = 0; j < NTW; ++j) wsrc[j] = a.Wp + ((long)(nt_blk + NTW * wv + j) * a.KT + kt0) * SPT + (kg * TH + min(m, TH - 1));
  auto issue = [&](int kt) {
    char* st = smem + (size_t)(kt & (VC_BIG_STAGES - 1)) * VC_BIG_STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[j] + (long)kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + (2 * wv + j) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + (long)kt * SPT),
                                       (__attribute__((address_space(3))) void*)(st + 16384 + (NTW * wv + j) * 1024), 16, 0, 0);
  };
  f32x4 acc[16][NTW];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Epilogue operands: the bias of the lane's channels is requested HERE, the cache slots of its 16 rows in one batch
  // right after the k-loop (32 more live registers across the loop spill).  Fetched inside the store loop each is a
  // dependent L2 round trip between two stores - 32 of them in a row cost the QKV form 25 us.
  const bool nvalid = 4 * kg < TH;
  float4 ebias[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int nt = min(nt_blk + NTW * wv + j, a.n_tiles - 1);
    int p0, s0;
    epi_preload<WT, EPI>(a, 0, nt * TH + (nvalid ? 4 * kg : 0), 0, ebias[j], p0, s0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int nck = kt_blk;
  const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  issue(0);
  if (nck > 1) issue(1);
  if (nck > 2) issue(2);
  for (int kt = 0; kt < nck; ++kt) {
    // the stage of this step was issued three steps ago: 2 + NTW DMA instructions per step, so it has landed once no more
    // than two steps' worth (one, none at the tail) are outstanding - a wave's requests land in order
    const int ahead = nck - 1 - kt;
    if (ahead >= 2) { if (NTW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else if (ahead == 1) { if (NTW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's share of the stage is in; the stage read last step is free
    asm volatile("" ::: "memory");
    if (kt + 3 < nck) issue(kt + 3);
    // Fragment reads in inline asm: next to an LDS-DMA in flight the compiler drains the whole DMA queue (vmcnt(0)) before
    // any LDS read it cannot prove disjoint from the DMA's destination - here the read stage and the three stages in flight
    // are disjoint by construction (kt & 3).  Reads of the second half of the row tiles fly during the first half's MFMAs.
    const unsigned sa = (unsigned)((kt & (VC_BIG_STAGES - 1)) * VC_BIG_STAGE_BYTES + lane * 16) + lds_base;
    const unsigned wa = sa + 16384u + (unsigned)(NTW * wv) * 1024u;
    u32x4 wf0, wf1, xa0, xa1, xa2, xa3, xa4, xa5, xa6, xa7, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7;   // (vector types: asm operands)
    asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:1024\n\t"      /* (NTW == 1: the second tile's slot is read and ignored) */
                 "ds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:1024\n\tds_read_b128 %4, %11 offset:2048\n\t"
                 "ds_read_b128 %5, %11 offset:3072\n\tds_read_b128 %6, %11 offset:4096\n\tds_read_b128 %7, %11 offset:5120\n\t"
                 "ds_read_b128 %8, %11 offset:6144\n\tds_read_b128 %9, %11 offset:7168\n\t"
                 : "=&v"(wf0), "=&v"(wf1), "=&v"(xa0), "=&v"(xa1), "=&v"(xa2), "=&v"(xa3), "=&v"(xa4), "=&v"(xa5), "=&v"(xa6), "=&v"(xa7)
                 : "v"(wa), "v"(sa) : "memory");
    asm volatile("ds_read_b128 %0, %8 offset:8192\n\tds_read_b128 %1, %8 offset:9216\n\tds_read_b128 %2, %8 offset:10240\n\t"
                 "ds_read_b128 %3, %8 offset:11264\n\tds_read_b128 %4, %8 offset:12288\n\tds_read_b128 %5, %8 offset:13312\n\t"
                 "ds_read_b128 %6, %8 offset:14336\n\tds_read_b128 %7, %8 offset:15360\n\t"
                 : "=&v"(xb0), "=&v"(xb1), "=&v"(xb2), "=&v"(xb3), "=&v"(xb4), "=&v"(xb5), "=&v"(xb6), "=&v"(xb7)
                 : "v"(sa) : "memory");
    // the first ten reads have returned once at most eight are outstanding (LDS returns in order)
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(wf0), "+v"(wf1), "+v"(xa0), "+v"(xa1), "+v"(xa2), "+v"(xa3), "+v"(xa4), "+v"(xa5), "+v"(xa6), "+v"(xa7) :: "memory");
    if (TH < 16 && !wvalid) { wf0 = u32x4{0u, 0u, 0u, 0u}; wf1 = u32x4{0u, 0u, 0u, 0u}; }
#define VC_BIG_MM(i_, x_)                                                                        \
    acc[i_][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf0), __builtin_bit_cast(bf16x8, x_), acc[i_][0], 0, 0, 0); \
    if constexpr (NTW == 2) acc[i_][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf1), __builtin_bit_cast(bf16x8, x_), acc[i_][1], 0, 0, 0);
    VC_BIG_MM(0, xa0) VC_BIG_MM(1, xa1) VC_BIG_MM(2, xa2) VC_BIG_MM(3, xa3)
    VC_BIG_MM(4, xa4) VC_BIG_MM(5, xa5) VC_BIG_MM(6, xa6) VC_BIG_MM(7, xa7)
    __builtin_amdgcn_sched_barrier(0);                  // the first half's MFMAs stay ahead of the wait for the second half's reads
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb0), "+v"(xb1), "+v"(xb2), "+v"(xb3), "+v"(xb4), "+v"(xb5), "+v"(xb6), "+v"(xb7) :: "memory");
    VC_BIG_MM(8, xb0) VC_BIG_MM(9, xb1) VC_BIG_MM(10, xb2) VC_BIG_MM(11, xb3)
    VC_BIG_MM(12, xb4) VC_BIG_MM(13, xb5) VC_BIG_MM(14, xb6) VC_BIG_MM(15, xb7)
#undef VC_BIG_MM
    asm volatile("" ::: "memory");
  }
  // ---- epilogue: lane holds channels n..n+3 of row (row tile i, m) for weight tile j
  tile_epilogue<WT, EPI, 16, NTW>(a, acc, ebias, row_blk + m, nt_blk + NTW * wv, TH, kg, ks, a.n_rows);
}

template <int EPI, int NTW>
static hipError_t launch_big(const GemmArgs& a, int ksplit, hipStream_t s) {
  auto kern = rows_gemm_big_k<EPI, NTW>;
  constexpr size_t lds = (size_t)VC_BIG_STAGES * VC_BIG_STAGE_BYTES;
  static size_t granted[16] = {0};
  int dev = 0;
  if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
  if (dev >= 0 && dev < 16 && granted[dev] < lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    granted[dev] = lds;
  }
  dim3 grid(a.n_tiles / (8 * NTW), (a.n_rows + VC_BIG_M - 1) / VC_BIG_M, ksplit);
  ++vc_launch_counts[NTW == 2 ? VC_LC_BIG256 : VC_LC_BIG128];
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  return hipGetLastError();
}

template <typename WT, int EPI, int NTW, int WM, int OCC = 1>
static hipError_t launch_blk_n(const GemmArgs& a, int ksplit, hipStream_t s) {
  auto kern = rows_gemm_blk_k<WT, EPI, NTW, WM, OCC>;
  constexpr int WN = 4 / WM;
  constexpr size_t lds = 2 * (size_t)VC_BLK_M * (VC_BLK_KT * WTr<WT>::KW * sizeof(WT) + 16);
  static size_t granted[16] = {0};                  // per instantiation and device
  int dev = 0;
  if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
  if (dev >= 0 && dev < 16 && granted[dev] < lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    granted[dev] = lds;
  }
  if ((a.KT / ksplit) % VC_BLK_KT != 0) return hipErrorInvalidValue;
  if (a.n_tiles % (WN * NTW) != 0) return hipErrorInvalidValue;
  dim3 grid(a.n_tiles / (WN * NTW), (a.n_rows + VC_BLK_M - 1) / VC_BLK_M, ksplit);
  GemmArgs b = a;
  b.att_q4_shift = vc_blk_dbg_mask;                     // 0 except inside vc_bench_kernel("pf_ffn1") under VC_BLK_DBG
  ++vc_launch_counts[(NTW == 2 && WM == 1) ? VC_LC_BLK128_SBS : (NTW == 4) ? VC_LC_BLK128_2X2 : (OCC == 2) ? VC_LC_BLK64_OCC2 : VC_LC_BLK64];
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, b);
  return hipGetLastError();
}
template <typename WT, int EPI>
static hipError_t launch_blk_e(const GemmArgs& a, int ksplit, hipStream_t s) {
  // 128-channel tiles once they still give every CU a workgroup, else 64-channel tiles
  const long wide = (long)(a.n_tiles / 8) * ((a.n_rows + VC_BLK_M - 1) / VC_BLK_M) * ksplit;
  static const int form = getenv("VC_BLK_FORM") ? atoi(getenv("VC_BLK_FORM")) : 1;
  if (a.n_tiles % 8 == 0 && wide >= 240) {
    // 128-channel tile.  form 1: four waves side by side, each owning ALL 8 row tiles and 2 weight tiles of its own -
    // no weight fragment is requested twice in a workgroup; form 0: 2 x 2 waves, 4 x 4 tiles each (weights requested
    // by both row halves)
    if (form == 2) return launch_blk_n<WT, EPI, 2, 2, 2>(a, ksplit, s);   // 64-channel tiles, two workgroups per CU
    if (form == 1) return launch_blk_n<WT, EPI, 2, 1>(a, ksplit, s);
    return launch_blk_n<WT, EPI, 4, 2>(a, ksplit, s);
  }
  return launch_blk_n<WT, EPI, 2, 2>(a, ksplit, s);
}
template <typename WT>
static hipError_t launch_blk(const GemmArgs& a, int pro, int epi, int ksplit, hipStream_t s) {
  if (pro != PRO_PLAIN) return hipErrorInvalidValue;       // LayerNorm comes from ln_rows_k, attention normalises itself
  // long row streams (bf16): 256 x 256 tiles once they give (nearly) every CU a workgroup
  if constexpr (sizeof(WT) == 2) {
    static const int big_off = getenv("VC_NO_BIG_GEMM") ? 1 : 0;
    const long wgs = (long)(a.n_tiles / 16) * ((a.n_rows + VC_BIG_M - 1) / VC_BIG_M) * ksplit;
    if (!big_off && a.n_rows > 512 && a.n_tiles % 16 == 0) {
      if (wgs >= 160) {                    // 256 x 256 tiles
        if (epi == EPI_QKV) return launch_big<EPI_QKV, 2>(a, ksplit, s);
        if (epi == EPI_QKV16) return launch_big<EPI_QKV16, 2>(a, ksplit, s);
        if (epi == EPI_PART) return launch_big<EPI_PART, 2>(a, ksplit, s);
        if (epi == EPI_RELU) return launch_big<EPI_RELU, 2>(a, ksplit, s);
      }
      // (256 x 128 tiles - NTW = 1, twice the workgroups - for passes of 513..1279 rows were measured: no gain over the 128 x 128
      // kernel, e.g. 800 rows 48.0 us either way, profiles/r03_pf_gemm_probe.log: only the weight re-reads shrink, the X re-reads
      // that make up the other half of the L2 traffic do not.  Not dispatched.)
    }
  }
  if (epi == EPI_QKV) return launch_blk_e<WT, EPI_QKV>(a, ksplit, s);
  if (epi == EPI_QKV16) return launch_blk_e<WT, EPI_QKV16>(a, ksplit, s);
  if (epi == EPI_PART) return launch_blk_e<WT, EPI_PART>(a, ksplit, s);
  if (epi == EPI_RELU) return launch_blk_e<WT, EPI_RELU>(a, ksplit, s);
  return hipErrorInvalidValue;
}

// prefill pass (GemmArgs.mt == 1): the block GEMMs
hipError_t vc_launch_gemm_blk(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, hipStream_t s) {
  if (dtype == VC_DTYPE_BF16) return launch_blk<bf16_t>(a, pro, epi, ksplit, s);
  return launch_blk<float>(a, pro, epi, ksplit, s);
}
