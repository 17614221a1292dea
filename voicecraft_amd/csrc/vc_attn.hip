// vc_attn.hip — single-query causal attention over the in-place KV cache.
//
// Replaces F.scaled_dot_product_attention with the rebuilt [B*H,S,S] -inf mask
// (models/modules/activation.py:634, models/voicecraft.py:419-447) and the torch.cat cache growth
// (activation.py:628-631, voicecraft.py:1081).  The combined mask of the reference is plain causal
// (SURVEY.md §0), so a row at cache position p simply attends to positions 0..p of its sequence; no
// mask tensor exists here.  K/V of the row itself were written by the QKV epilogue of the same pass.
//
// grid = (rows, heads, nsplit <= 8), 8 waves per block: split-S so that one decode row still covers the chip; every block
// leaves an un-normalised (max, sum, acc[hd]) partial that the out-projection's prologue merges.
// A cached row (hd elements) is spread over LPR = hd*sizeof/16 lanes with 16-byte loads, so a wave
// reads 64/LPR consecutive positions = one contiguous 1 KiB burst per K (and V) instruction.
#include <math.h>
#include "vc_common.h"

template <typename WT>
__device__ __forceinline__ void unpack16(const uint4& u, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
  f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <>
__device__ __forceinline__ void unpack16<bf16_t>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

#ifndef VC_ATT_WAVES
#define VC_ATT_WAVES 8
#endif

// ---------------------------------------------------------------- decode attention
// NT: K/V rows are requested with the non-temporal hint (every cached row is read exactly once per step and never again before
// the next step's 1.7 GB of weights have gone through the caches) - a template parameter so that the hint cannot be merged away
// (vc_gemm.hip rows_gemm_k), chosen per launch from AttnArgs.nt (option "nt", second value).
// FAST (round 5; option "attn_fast", default on): the in-kernel stamps of round 4 put half of the one-row launch's in-kernel time
// (3 916 of 8 016 clk) into "wave merge + block sync + final + store" - online-softmax bookkeeping, not K/V.  Here a wave takes the
// MAXIMUM of a batch's scores over all of its lanes before any exponential, so every lane of the wave carries the same running
// maximum: no rescaling per visit (one per batch of 4), and the position groups of a wave merge by plain additions (no
// exponentials, no multiplies).  In bf16 mode the exponentials are hardware exp2 (v_exp_f32, q pre-scaled by log2 e; the exported
// maximum is converted back to natural units for the out-projection's merge); the exact fp32 mode keeps expf.
// P16 (round 6; bf16 mode, split passes of 2..8 rows): the un-normalised partial acc[hd] leaves as bf16 instead of fp32 - the
// finished-row out-projection re-reads every head's partials in each of its 256 workgroups (rows x heads x splits x hd values: 131 KB
// per workgroup at 8 rows, four times its weights), and those bytes go through the CU's vector-memory path like any others; the merged
// row is rounded to bf16 for the MFMA anyway.  (max, sum) stay fp32.
template <typename WT, bool NT, bool FAST, bool P16 = false>
__global__ __launch_bounds__(64 * VC_ATT_WAVES) void rows_attn_k(const AttnArgs a) {
  static_assert(!P16 || sizeof(WT) == 2, "bf16 partials: bf16 mode only");
  constexpr int EPL = WTr<WT>::EPL;
  constexpr int NW = VC_ATT_WAVES;
  constexpr bool X2 = sizeof(WT) == 2;        // FAST: base-2 exponentials in bf16 mode
  auto ex = [](float x) -> float {
    if constexpr (FAST && X2) return __builtin_amdgcn_exp2f(x);
    else return expf(x);
  };
  __shared__ float s_m[NW], s_l[NW];
  __shared__ float s_o[NW][128];
  VC_KTS_DECL();
  VC_KTS(0);
  const int r = blockIdx.x, h = blockIdx.y, sp = blockIdx.z;     // grid.x == n_rows
  // ONE scalar round trip for everything the addresses depend on.  Written as plain C these four came out as VECTOR loads (the
  // compiler cannot prove them invariant) and `share` was sunk behind the branch below - two dependent vector round trips before the
  // first K/V request (seen in the ISA, round 5).  All four are wave-uniform words that earlier launches of the step wrote (the
  // scalar cache is invalidated at every kernel start): one batch of s_load, one wait.
  int active, pos, seq, share;
  asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(active), "=&s"(pos), "=&s"(seq), "=&s"(share)
               : "s"(a.n_active), "s"(a.row_pos + r), "s"(a.row_seq + r), "s"(a.share_len)
               : "memory");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hd = a.hd;
  constexpr int EPL_SH = EPL == 8 ? 3 : 2;
  const int lpr_sh = a.hd_shift - EPL_SH;   // (head_dim is a power of two: shifts, not the divisions the compiler would expand)
  const int LPR = 1 << lpr_sh;         // lanes per cached row: 4 (bf16, head_dim 32), 8, 16 or 32
  const int PPW = 64 >> lpr_sh;        // positions per wave per step
  const int sub = lane >> lpr_sh, li = lane & (LPR - 1);

  float m = -INFINITY, l = 0.f;
  float o[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) o[j] = 0.f;

  if (active != 0 && pos >= 0 && seq >= 0) {   // a finished step / an inactive row leaves an empty partial (seq is
                                               // tested only so that its load is not sunk behind the branch)
    VC_KTS(1);
    const int S = pos + 1;
    // ceil(S / nsplit) without the integer division the compiler expands into ~25 dependent instructions: a.inv_nsplit is 1 / nsplit
    // rounded UP, so the product never falls below the quotient; a chunk that came out one too large only leaves the last split fewer
    // positions (any chunk with chunk * nsplit >= S partitions [0, S))
    int chunk = (int)((float)(S + a.nsplit - 1) * a.inv_nsplit);
    if (chunk * a.nsplit < S) ++chunk;
    const int p0 = sp * chunk;
    const int p1 = min(S, p0 + chunk);
    const int step = 4 * NW * PPW;
    // q, then K and V of the first 4 visits per wave, are requested together (clamped,
    // unconditional); q is only touched once all of them are on their way.
    float4 qv[EPL / 4];
    {
      const float4* qp = reinterpret_cast<const float4*>(a.q + (long)r * a.d + h * hd + li * EPL);
#pragma unroll
      for (int j = 0; j < EPL / 4; ++j) qv[j] = qp[j];
    }
    uint4 ku[4], vu[4];
    int pp[4];
    // (the general form - refills of long chunks, calls with a shared text prefix - computes its per-lane 64-bit bases where it is
    // used: hoisted in front of the first batch they would sit in front of the lean path too)
#define VC_KV_LOADS(pb_)                                                     \
    const long own = (long)seq * a.cache_seq_stride;      /* positions below `share` live in sequence 0's cache */ \
    const long base = own + (long)h * a.S_max * hd + li * EPL;               \
    const WT* kb = reinterpret_cast<const WT*>(a.kcache) + base;             \
    const WT* vb = reinterpret_cast<const WT*>(a.vcache) + base;             \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                       \
      pp[it] = (pb_) + (it * NW + wave) * PPW + sub;                         \
      const long pc = max(min(pp[it], p1 - 1), 0);                           \
      const long po = pc * hd - ((pc < share) ? own : 0);                    \
      if constexpr (NT) {                                                    \
        ku[it] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + po))); \
        vu[it] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + po))); \
      } else {                                                               \
        ku[it] = *reinterpret_cast<const uint4*>(kb + po);                   \
        vu[it] = *reinterpret_cast<const uint4*>(vb + po);                   \
      }                                                                      \
    }
    // The FIRST batch - the one every launch waits for - from a wave-uniform base and one 32-bit offset per lane when no prefix is
    // shared (every call but the sentence-chained ones): scalar-base addressing, no 64-bit vector arithmetic, no select per visit.
    // (In-kernel stamps of the first round-5 build: 2 956 clk from "position known" to "loads issued" - the address code in front of
    // the requests is paid in instruction fetch at the cold start of the launch, profiles/r05f_kernel_stamps_giga830M.log.)
    if (__builtin_expect(share == 0, 1)) {       // (against the general form, in-process: -0.19 % +- 0.06 per step at giga830M, -0.38 % at giga330M, -0.40 % at 8 rows; profiles/r05g_ab_*.log)
      const char* kbu = reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.kcache) + (long)seq * a.cache_seq_stride + (long)h * a.S_max * hd);
      const char* vbu = reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.vcache) + (long)seq * a.cache_seq_stride + (long)h * a.S_max * hd);
      const unsigned lo = (unsigned)(li * EPL) * (unsigned)sizeof(WT);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        pp[it] = p0 + (it * NW + wave) * PPW + sub;
        const unsigned pc = (unsigned)max(min(pp[it], p1 - 1), 0);
        const unsigned off = ((pc << a.hd_shift) * (unsigned)sizeof(WT)) + lo;
        if constexpr (NT) {
          ku[it] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kbu + off)));
          vu[it] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vbu + off)));
        } else {
          ku[it] = *reinterpret_cast<const uint4*>(kbu + off);
          vu[it] = *reinterpret_cast<const uint4*>(vbu + off);
        }
      }
    } else {
      VC_KV_LOADS(p0)
    }
    __builtin_amdgcn_sched_barrier(0);
    VC_KTS(2);
    float q[EPL];
    const float qs = (FAST && X2) ? a.scale * 1.4426950408889634f : a.scale;
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
      q[4 * j] = qv[j].x * qs; q[4 * j + 1] = qv[j].y * qs;
      q[4 * j + 2] = qv[j].z * qs; q[4 * j + 3] = qv[j].w * qs;
    }
    VC_KTS(3);
    for (int pb = p0;;) {
      if constexpr (FAST) {
        float sc[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float kf[EPL];
          unpack16<WT>(ku[it], kf);
          float t = 0.f;
#pragma unroll
          for (int j = 0; j < EPL; ++j) t += q[j] * kf[j];
          if (LPR == 4) t = quad_sum(t);
          else if (LPR == 8) t = half_row_sum(t);
          else { t = row_sum(t); if (LPR == 32) t += __shfl_xor(t, 16, 64); }
          sc[it] = (pp[it] < p1) ? t : -INFINITY;
        }
        const float mn = fmaxf(m, wave_max(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]))));    // wave-uniform
        if (mn > -INFINITY) {
          const float corr = ex(m - mn);             // m = -inf before the first valid position -> 0
          float pw[4];
#pragma unroll
          for (int it = 0; it < 4; ++it) pw[it] = ex(sc[it] - mn);    // -inf (beyond the chunk) -> 0
          l = l * corr + ((pw[0] + pw[1]) + (pw[2] + pw[3]));
          float vf[4][EPL];
#pragma unroll
          for (int it = 0; it < 4; ++it) unpack16<WT>(vu[it], vf[it]);
#pragma unroll
          for (int j = 0; j < EPL; ++j)
            o[j] = o[j] * corr + ((pw[0] * vf[0][j] + pw[1] * vf[1][j]) + (pw[2] * vf[2][j] + pw[3] * vf[3][j]));
          m = mn;
        }
      } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float kf[EPL], vf[EPL];
        unpack16<WT>(ku[it], kf);
        unpack16<WT>(vu[it], vf);
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) sc += q[j] * kf[j];
        if (LPR == 4) sc = quad_sum(sc);
        else if (LPR == 8) sc = half_row_sum(sc);
        else { sc = row_sum(sc); if (LPR == 32) sc += __shfl_xor(sc, 16, 64); }
        if (pp[it] < p1) {
          const float mn = fmaxf(m, sc);
          const float corr = expf(m - mn);     // m = -inf on the first visit -> 0
          const float pe = expf(sc - mn);
          l = l * corr + pe;
#pragma unroll
          for (int j = 0; j < EPL; ++j) o[j] = o[j] * corr + pe * vf[j];
          m = mn;
        }
      }
      }
      pb += step;
      if (pb >= p1) break;
      { VC_KV_LOADS(pb) }
    }
#undef VC_KV_LOADS
    VC_KTS(4);
  }
  // merge the PPW position groups of this wave (same li, different sub)
  if constexpr (FAST) {       // one running maximum per wave: the groups merge by plain sums
    for (int off = LPR; off < 64; off <<= 1) {
      l += __shfl_xor(l, off, 64);
#pragma unroll
      for (int j = 0; j < EPL; ++j) o[j] += __shfl_xor(o[j], off, 64);
    }
  } else
  for (int off = LPR; off < 64; off <<= 1) {
    const float m2 = __shfl_xor(m, off, 64);
    const float l2 = __shfl_xor(l, off, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn);
    const float c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
    l = l * c1 + l2 * c2;
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      const float o2 = __shfl_xor(o[j], off, 64);
      o[j] = o[j] * c1 + o2 * c2;
    }
    m = mn;
  }
  VC_KTS(5);
  if (lane < LPR) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) s_o[wave][lane * EPL + j] = o[j];
    if (lane == 0) { s_m[wave] = m; s_l[wave] = l; }
  }
  __syncthreads();
  VC_KTS(6);
  if (tid < hd) {
    float M = s_m[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = (s_m[w] == -INFINITY) ? 0.f : ex(s_m[w] - M);
      L += c * s_l[w];
      O += c * s_o[w][tid];
    }
    if constexpr (FAST && X2) M *= 0.6931471805599453f;      // back to natural units: the out-projection merges with expf
    if (a.x_out) {      // unsplit pass (prefill): the block saw every position, so it normalises itself
      WTr<WT>::st(reinterpret_cast<WT*>(a.x_out) + (long)r * a.d + h * hd + tid, (L > 0.f) ? O / L : 0.f);
    } else {
      const long pi = ((long)(r * a.H + h) * a.nsplit + sp);
      if constexpr (P16) reinterpret_cast<uint16_t*>(a.att_o)[pi * hd + tid] = f32_to_bf16(O);
      else a.att_o[pi * hd + tid] = O;
      if (tid == 0) { a.att_ml[pi * 2] = M; a.att_ml[pi * 2 + 1] = L; }
    }
  }
  VC_KTS(7);
  VC_KTS_FLUSH();
}

hipError_t vc_launch_attn(const AttnArgs& a, int dtype, int rows_cap, hipStream_t s) {
  dim3 grid(rows_cap, a.H, a.nsplit);
  ++vc_launch_counts[VC_LC_ROWS_ATTN];
  AttnArgs b = a;
  b.inv_nsplit = nextafterf(1.0f / (float)a.nsplit, 2.0f);
#define VC_ATTN_GO(WT_, NT_, F_) hipLaunchKernelGGL((rows_attn_k<WT_, NT_, F_>), grid, dim3(64 * VC_ATT_WAVES), 0, s, b);
  if (dtype == VC_DTYPE_BF16 && a.part16 && !a.x_out) {      // bf16 partials (finished-row passes of 2..8 rows whose attention is split)
    if (a.fast) { if (a.nt) hipLaunchKernelGGL((rows_attn_k<bf16_t, true, true, true>), grid, dim3(64 * VC_ATT_WAVES), 0, s, b);
                  else hipLaunchKernelGGL((rows_attn_k<bf16_t, false, true, true>), grid, dim3(64 * VC_ATT_WAVES), 0, s, b); }
    else { if (a.nt) hipLaunchKernelGGL((rows_attn_k<bf16_t, true, false, true>), grid, dim3(64 * VC_ATT_WAVES), 0, s, b);
           else hipLaunchKernelGGL((rows_attn_k<bf16_t, false, false, true>), grid, dim3(64 * VC_ATT_WAVES), 0, s, b); }
    return hipGetLastError();
  }
  if (dtype == VC_DTYPE_BF16) {
    if (a.fast) { if (a.nt) VC_ATTN_GO(bf16_t, true, true) else VC_ATTN_GO(bf16_t, false, true) }
    else { if (a.nt) VC_ATTN_GO(bf16_t, true, false) else VC_ATTN_GO(bf16_t, false, false) }
  } else {
    if (a.fast) { if (a.nt) VC_ATTN_GO(float, true, true) else VC_ATTN_GO(float, false, true) }
    else { if (a.nt) VC_ATTN_GO(float, true, false) else VC_ATTN_GO(float, false, false) }
  }
#undef VC_ATTN_GO
  return hipGetLastError();
}

// ---------------------------------------------------------------- KV replication (best-of-N)
// inference_tts_batch repeats the prefilled cache batch_size times (voicecraft.py:1329-1343);
// here the prompt is prefilled once into slot src and its first `len` positions are copied.
__global__ void copy_kv_k(char* cache, long seq_stride_b, long head_stride_b, long len_b, int H,
                          int src_seq, int dst_seq0) {
  const int dst = dst_seq0 + blockIdx.z;
  const int h = blockIdx.y;
  const char* s = cache + (long)src_seq * seq_stride_b + (long)h * head_stride_b;
  char* d = cache + (long)dst * seq_stride_b + (long)h * head_stride_b;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < len_b;
       i += (long)gridDim.x * blockDim.x * 16)
    *reinterpret_cast<uint4*>(d + i) = *reinterpret_cast<const uint4*>(s + i);
}
hipError_t vc_launch_copy_kv(void* cache, long seq_stride, int H, int S_max, int hd, int len,
                             int src_seq, int dst_seq0, int n_dst, int dtype, hipStream_t s) {
  if (n_dst <= 0 || len <= 0) return hipSuccess;
  const long esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  hipLaunchKernelGGL(copy_kv_k, dim3(8, H, n_dst), dim3(256), 0, s, (char*)cache, seq_stride * esz,
                     (long)S_max * hd * esz, (long)len * hd * esz, H, src_seq, dst_seq0);
  return hipGetLastError();
}

// ---------------------------------------------------------------- prefill attention on the MFMA (flash style)
// A prefill pass lays every prompt's rows back to back, each prompt starting on a multiple of 16 rows (vc_engine.hip
// prefill_batch), so a tile of 16 consecutive rows belongs to ONE sequence and holds consecutive positions
// pos0 .. pos0+na-1 (na <= 16: the tail of a prompt's last tile is padding, row_pos = -1).  rows_attn_k walks the
// whole prefix once per ROW (2 GB of K/V out of L2 per layer for a 700-row prompt - that was 45 % of the prefill);
// here a workgroup owns (row tile, head), its NW waves take every NW-th key tile, and per key tile of KW keys
// (bf16: 32, fp32: 16):
//   S = Q K^T   MFMA: A = Q fragments (registers, pre-scaled), B = K straight from the cache (a cached row IS a B fragment)
//   online softmax on the D layout (lane = 4 query rows x 1 key): row max / row sum are 16-lane DPP reductions
//   O += P V    MFMA: A = P (re-laid out through a 1 KB wave-private LDS tile), B = V^T.  The contraction index of this
//               product is the KEY, which the cache stores as the slow index: bf16 keeps the V tile row-major in
//               wave-private LDS and reads it back through ds_read_b64_tr_b16; fp32 transposes while it copies.
// The partial (max, sum, O) sets of the waves are merged through LDS at the end.  No block barrier inside the key loop.
//
// Round 3 (the round-2 kernel left every key tile's requests un-overlapped: a 512-row pass = 4 dependent L2 round trips
// per wave = 26 us for 1 GFLOP; now 14.7 us, 800 rows 49 -> 25 us, 2048 rows 213 -> 90 us):
//   * the NEXT key tile's K fragments and V items are requested as soon as the current tile's have been consumed (after
//     Q K^T and the V^T copy), so they fly under the softmax and P V - in the SAME registers: a second register set
//     costs 300 VGPRs = one wave per SIMD, measured worse than the latency it hides,
//   * workgroups are numbered head-first (blockIdx.x = head): the hardware deals consecutive workgroups round-robin
//     over the 8 XCDs, so a head's K/V prefix (re-read by every row tile of that head) stays in ONE XCD's L2 instead of
//     being fetched into all eight; the longest row tiles are dealt first,
//   * the softmax runs in base 2 (Q pre-scaled by log2 e, v_exp_f32 directly).
//   * tiles the causal mask does not cut take a mask-free softmax, the accumulators are rescaled only when a running
//     maximum moved, a tile inside the prefix is addressed from ONE wave-uniform base.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename WT, int HD, int NW>
__global__ __launch_bounds__(64 * NW) void tile_attn_k(const AttnArgs a) {
  using T = WTr<WT>;
  constexpr int EPL = T::EPL, KW = T::KW;
  constexpr int NSUB = KW / 16, NKS = HD / KW, NDT = HD / 16;
  constexpr int SZ = (int)sizeof(WT);
  constexpr int VS = KW * SZ + 16;
  constexpr bool TR = SZ == 2;                       // bf16: V stays row-major in LDS and is read through ds_read_b64_tr_b16
  constexpr int SUBT = KW * 32 + 32;                 // bytes of one [KW keys][16 dims] sub-tile (+32: the 16 stores of a key hit 64 banks)
  constexpr int VBYTES = TR ? NDT * SUBT : HD * VS;
  constexpr int WAVE_LDS = VBYTES + 16 * VS;
  constexpr int NPK = 4 / SZ;
  constexpr int VITEMS = (KW / NPK) * (HD / EPL) / 64;
  struct KV { uint4 kf[NSUB][NKS]; uint4 vu[VITEMS][NPK]; };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int h = blockIdx.x, r0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 16;
  const int share = *a.share_len;
  const int rp = (r0 + m < a.n_rows) ? a.row_pos[r0 + m] : -1;
  const int pos0 = __builtin_amdgcn_readfirstlane(rp);
  const int seq = a.row_seq[r0];
  const int na = __popcll(__ballot(rp >= 0 && lane < 16));
  char* vt = smem + wave * WAVE_LDS;
  char* pt = vt + VBYTES;
  float mrun[4], lrun[4];
  f32x4 o[NDT];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (na > 0 && pos0 >= 0) {
    const int last = pos0 + na - 1;
    const long hbase = (long)h * a.S_max * HD;
    const WT* kc = reinterpret_cast<const WT*>(a.kcache) + hbase;
    const WT* vc = reinterpret_cast<const WT*>(a.vcache) + hbase;
    const long own = (long)seq * a.cache_seq_stride;
    constexpr int DG = HD / EPL;                           // 16-byte groups per cached row
    const unsigned koff = (unsigned)(m * HD + EPL * kg);   // this lane's K fragment inside a key tile (elements)
    const unsigned voff = (unsigned)((lane / DG) * NPK * HD + (lane % DG) * EPL);
    auto issue = [&](KV& t, int kt0) __attribute__((always_inline)) {
      // Common case: the tile lies inside the prefix and on one side of the shared-prefix boundary - ONE wave-uniform
      // base, every request a constant offset from it (scalar base + 32-bit lane offset + immediate).
      if (kt0 + KW - 1 <= last && (kt0 >= share || kt0 + KW <= share)) {
        const WT* kb = kc + ((kt0 < share) ? 0 : own) + (long)kt0 * HD;
        const WT* vb = vc + ((kt0 < share) ? 0 : own) + (long)kt0 * HD;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
            t.kf[sub][ks] = *reinterpret_cast<const uint4*>(kb + (koff + (unsigned)(sub * 16 * HD + KW * ks)));
#pragma unroll
        for (int it = 0; it < VITEMS; ++it)
#pragma unroll
          for (int q = 0; q < NPK; ++q)
            t.vu[it][q] = *reinterpret_cast<const uint4*>(vb + (voff + (unsigned)((it * (64 / DG) * NPK + q) * HD)));
        return;
      }
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const int kp = min(kt0 + sub * 16 + m, last);
        const WT* kr = kc + ((kp < share) ? 0 : own) + (long)kp * HD + EPL * kg;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) t.kf[sub][ks] = *reinterpret_cast<const uint4*>(kr + KW * ks);
      }
#pragma unroll
      for (int it = 0; it < VITEMS; ++it) {
        const int item = it * 64 + lane;
        const int kgrp = item / DG, dgrp = item - kgrp * DG;
#pragma unroll
        for (int q = 0; q < NPK; ++q) {
          const int kp = min(kt0 + kgrp * NPK + q, last);
          t.vu[it][q] = *reinterpret_cast<const uint4*>(vc + ((kp < share) ? 0 : own) + (long)kp * HD + dgrp * EPL);
        }
      }
    };
    const int kt_first = wave * KW;
    KV A;
    if (kt_first <= last) issue(A, kt_first);              // in flight while Q is fetched and converted
    uint4 qf[NKS];
    {
      const float qs = a.scale * 1.4426950408889634f;      // scores in log2 units
      const float* qp = a.q + (long)min(r0 + m, a.n_rows - 1) * a.d + h * HD + EPL * kg;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        WT tmp[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) T::st(&tmp[j], qp[KW * ks + j] * qs);
        qf[ks] = *reinterpret_cast<const uint4*>(tmp);
      }
    }
    f32x4 sc[NSUB];
    auto scores_and_vt = [&](const KV& t) __attribute__((always_inline)) {
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sc[sub] = mfma_frag(qf[ks], t.kf[sub][ks], sc[sub], (WT*)nullptr);
      }
      if constexpr (TR) {
        // V tile as NDT sub-tiles [key][16 dims] (32-byte rows): one 16-byte store per item; the transposition happens
        // in the read (ds_read_b64_tr_b16: lane i of a 16-lane group supplies the 8-byte chunk C_i, lane n receives
        // element n&3 of chunks 4j + (n>>2), j = 0..3 - measured, tools/tr_probe.hip)
        char* dst = vt + ((lane % DG) >> 1) * SUBT + ((lane % DG) & 1) * 16 + (lane / DG) * NPK * 32;
#pragma unroll
        for (int it = 0; it < VITEMS; ++it)
#pragma unroll
          for (int q = 0; q < NPK; ++q)
            *reinterpret_cast<uint4*>(dst + (it * (64 / DG) * NPK + q) * 32) = t.vu[it][q];
      } else {
#pragma unroll
        for (int it = 0; it < VITEMS; ++it) {
          const int item = it * 64 + lane;
          const int kgrp = item / DG, dgrp = item - kgrp * DG;
          char* dst = vt + (dgrp * EPL) * VS + kgrp * 4;
          const uint32_t a0[4] = {t.vu[it][0].x, t.vu[it][0].y, t.vu[it][0].z, t.vu[it][0].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t*>(dst + j * VS) = a0[j];
        }
      }
    };
    auto softmax_and_pv = [&](int kt0) __attribute__((always_inline)) {
      float p[NSUB][4], corr[4];
      if (kt0 + KW - 1 <= pos0) {
        // interior tile: every key is visible to every row (padding rows ride along; their output is zeroed at the store)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float mx = sc[0][r];
#pragma unroll
          for (int sub = 1; sub < NSUB; ++sub) mx = fmaxf(mx, sc[sub][r]);
          mx = fmaxf(mx, dpp_f<VC_DPP_QP_1032>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_QP_2301>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_ROW_HALF_MIRROR>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_ROW_MIRROR>(mx));
          const float mn = fmaxf(mrun[r], mx);
          corr[r] = fast_exp2(mrun[r] - mn);               // first tile: exp2(-inf) = 0
          float ps = 0.f;
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub) { p[sub][r] = fast_exp2(sc[sub][r] - mn); ps += p[sub][r]; }
          lrun[r] = lrun[r] * corr[r] + row_sum(ps);
          mrun[r] = mn;
        }
      } else {
        // edge tile: the causal mask cuts through it
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qrow = 4 * kg + r;
          const int qpos = (qrow < na) ? pos0 + qrow : -1;
          float mx = -INFINITY;
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub) {
            const int kp = kt0 + sub * 16 + m;
            p[sub][r] = (kp <= qpos) ? sc[sub][r] : -INFINITY;
            mx = fmaxf(mx, p[sub][r]);
          }
          mx = fmaxf(mx, dpp_f<VC_DPP_QP_1032>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_QP_2301>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_ROW_HALF_MIRROR>(mx));
          mx = fmaxf(mx, dpp_f<VC_DPP_ROW_MIRROR>(mx));
          const float mn = fmaxf(mrun[r], mx);
          const float msafe = (mn == -INFINITY) ? 0.f : mn;  // nothing visible yet: exp2(-inf - 0) = 0 everywhere
          corr[r] = (mrun[r] == mn) ? 1.f : fast_exp2(mrun[r] - msafe);
          float ps = 0.f;
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub) { p[sub][r] = fast_exp2(p[sub][r] - msafe); ps += p[sub][r]; }
          lrun[r] = lrun[r] * corr[r] + row_sum(ps);
          mrun[r] = mn;
        }
      }
      // the accumulators are rescaled only when some row's running maximum moved (rare after the first tiles)
      if (__any(corr[0] != 1.f || corr[1] != 1.f || corr[2] != 1.f || corr[3] != 1.f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { o[dt][0] *= corr[0]; o[dt][1] *= corr[1]; o[dt][2] *= corr[2]; o[dt][3] *= corr[3]; }
      }
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) T::st(reinterpret_cast<WT*>(pt + (4 * kg + r) * VS) + sub * 16 + m, p[sub][r]);
      const uint4 pf = *reinterpret_cast<const uint4*>(pt + m * VS + kg * 16);
      if constexpr (TR) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        const char* vrd = vt + (8 * kg + (m >> 2)) * 32 + (m & 3) * 8;       // keys 8 kg .. 8 kg + 3 of dims 4 (m & 3) ..: chunk of lane m
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vrd + dt * SUBT));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vrd + dt * SUBT + 128));   // keys 8 kg + 4 ..
          uint4 vf;
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          vf.x = l2.x; vf.y = l2.y; vf.z = h2.x; vf.w = h2.y;
          o[dt] = mfma_frag(pf, vf, o[dt], (WT*)nullptr);
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const uint4 vf = *reinterpret_cast<const uint4*>(vt + (dt * 16 + m) * VS + kg * 16);
          o[dt] = mfma_frag(pf, vf, o[dt], (WT*)nullptr);
        }
      }
    };
    if (kt_first <= last) {
      for (int kt0 = kt_first;;) {
        scores_and_vt(A);                                  // the last reads of this tile's K fragments and V items
        const int nx = kt0 + NW * KW;
        const bool more = nx <= last;
        if (more) issue(A, nx);                            // ... so the next tile's requests fly under the softmax and P V
        softmax_and_pv(kt0);
        if (!more) break;
        kt0 = nx;
      }
    }
  }
  __syncthreads();
  float* mo = reinterpret_cast<float*>(smem);            // [NW waves][16 rows][HD]
  float* mm = mo + NW * 16 * HD;                         // [NW][16][2]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qrow = 4 * kg + r;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) mo[(wave * 16 + qrow) * HD + dt * 16 + m] = o[dt][r];
    if (m == 0) { mm[(wave * 16 + qrow) * 2] = mrun[r]; mm[(wave * 16 + qrow) * 2 + 1] = lrun[r]; }
  }
  __syncthreads();
  for (int e = tid; e < 16 * HD; e += 64 * NW) {
    const int qrow = e / HD, dim = e - qrow * HD;
    if (r0 + qrow >= a.n_rows) continue;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) M = fmaxf(M, mm[(w * 16 + qrow) * 2]);
    const float Ms = (M == -INFINITY) ? 0.f : M;
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = fast_exp2(mm[(w * 16 + qrow) * 2] - Ms);
      L += c * mm[(w * 16 + qrow) * 2 + 1];
      O += c * mo[(w * 16 + qrow) * HD + dim];
    }
    T::st(reinterpret_cast<WT*>(a.x_out) + (long)(r0 + qrow) * a.d + h * HD + dim, (qrow < na && L > 0.f) ? O / L : 0.f);
  }
}

// ---------------------------------------------------------------- prefill attention, second form: 64 query rows per workgroup
// tile_attn_k gives a wave 16 query rows and a QUARTER of the keys: ~450 VALU instructions per 16 MFMAs (32 DPP steps of row
// reductions per 16 x 32 score tile, P through LDS, every wave staging its own V tile) and a cross-wave merge at the end -
// 0.03-0.08 of the bf16 MFMA peak.  Here (bf16, head_dim 128; prompts laid out on 64-row boundaries, vc_engine.hip prefill_batch):
//   workgroup = 64 consecutive rows of one sequence x one head; wave w owns row tile w for ALL key tiles: no merge.
//   key tiles of 64 keys; K and V of a tile are staged ONCE per workgroup (8 x 16 B per thread), double-buffered, one block
//     barrier per tile; the next tile's global loads fly during this tile's MFMAs.
//     K: fragment-linear (fragment (sub, ks) = 1 KB, lane l at 16 l: every ds_read_b128 conflict-free); wave w stages sub-tile w.
//     V: row-major [key][128 dims], the 16-byte slot index XOR-ed with ((key & 3) << 2 | (key >> 2) & 3), so that the transposed
//        reads of 4 keys x 4 lane groups spread over all banks; wave w stages keys 16 w .. 16 w + 15.
//   the score product is computed TRANSPOSED: S^T = K Q^T (A = K fragments, B = Q fragments).  Its D layout - lane (m, kg) holds
//     keys 4 kg .. 4 kg + 3 of every 16-key sub-tile for QUERY m - gives each lane ONE query row: running maximum and sum are one
//     scalar per lane, the row reductions two cross-lane steps (lanes m, m + 16, m + 32, m + 48), and the eight probabilities of two
//     sub-tiles, packed to bf16, ARE the B fragment of O^T += V^T P^T for the 32-key step whose k order is (4 kg .. 4 kg + 3 of the
//     first sub-tile, then of the second) - P never leaves the registers; the A fragments (V^T in the same key order) come from
//     the V tile through two ds_read_b64_tr_b16 each (semantics as in tile_attn_k, tools/tr_probe.hip).
//   per 16 x 64 score tile and wave: 32 MFMAs against ~100 VALU instructions and 48 LDS reads.
#define VC_TA2_KEYS 64
__global__ __launch_bounds__(256) void tile_attn64_k(const AttnArgs a) {
  using WT = bf16_t;
  constexpr int HD = 128, KEYS = VC_TA2_KEYS, NKS = HD / 32, NDT = HD / 16;
  constexpr int KBYTES = KEYS * HD * 2, STAGE = 2 * KBYTES;            // 16 KB of K fragments + 16 KB of V rows per stage
  extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages = 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kg = lane >> 4;
  const int h = blockIdx.x, r0 = ((int)gridDim.y - 1 - (int)blockIdx.y) * 64;      // longest blocks first
  const int share = *a.share_len;
  // the block's rows: consecutive positions of one sequence, padding (row_pos = -1) only at the end
  const int rp_blk = (r0 + lane < a.n_rows) ? a.row_pos[r0 + lane] : -1;
  const int na_total = __popcll(__ballot(rp_blk >= 0));
  const int pos_blk = __builtin_amdgcn_readfirstlane(rp_blk);           // position of the block's first row
  if (na_total == 0 || pos_blk < 0) {                                    // nothing but padding: zero rows, as tile_attn_k leaves them
    for (int e2 = tid; e2 < 64 * HD; e2 += 256) {
      const int row = r0 + e2 / HD;
      if (row < a.n_rows) WTr<WT>::st(reinterpret_cast<WT*>(a.x_out) + (long)row * a.d + h * HD + (e2 % HD), 0.f);
    }
    return;
  }
  const int seq = a.row_seq[r0];
  const int blk_last = pos_blk + na_total - 1;                           // last key any row of the block sees
  const int tpos0 = pos_blk + 16 * wave;                                 // first position of this wave's row tile
  const int na_w = max(0, min(16, na_total - 16 * wave));                // active rows of the tile
  const int t_last = tpos0 + na_w - 1;
  const int qpos = (m < na_w) ? tpos0 + m : -1;
  const long hbase = (long)h * a.S_max * HD;
  const WT* kc = reinterpret_cast<const WT*>(a.kcache) + hbase;
  const WT* vc = reinterpret_cast<const WT*>(a.vcache) + hbase;
  const long own = (long)seq * a.cache_seq_stride;
  // ---- staging roles of this thread: K sub-tile `wave` (fragment (wave, ks): key 16 wave + m, dims 32 ks + 8 kg), V keys
  //      16 wave + 4 i + (lane >> 4), slot' = lane & 15 holding the row's 16-byte piece slot' ^ v(key)
  // (explicit scalars and macros: arrays captured by a lambda are demoted to scratch memory here)
  uint4 sk0, sk1, sk2, sk3, sv0, sv1, sv2, sv3;
#define VC_TA2_LOADV(i_, dst_)                                                                   \
  {                                                                                              \
    const int key_ = 16 * wave + 4 * (i_) + (lane >> 4);   /* key index inside the tile */      \
    const int v_ = (((key_ & 3) << 2) | ((key_ >> 2) & 3));                                      \
    const int kp_ = min(kt0_ + key_, blk_last);                                                  \
    dst_ = *reinterpret_cast<const uint4*>(vc + ((kp_ < share) ? 0 : own) + (long)kp_ * HD + 8 * ((lane & 15) ^ v_)); \
  }
#define VC_TA2_STAGE_LOAD(kt0_arg)                                                               \
  {                                                                                              \
    const int kt0_ = (kt0_arg);                                                                  \
    const int kpk_ = min(kt0_ + 16 * wave + m, blk_last);                                        \
    const WT* kr_ = kc + ((kpk_ < share) ? 0 : own) + (long)kpk_ * HD + 8 * kg;                  \
    sk0 = *reinterpret_cast<const uint4*>(kr_); sk1 = *reinterpret_cast<const uint4*>(kr_ + 32); \
    sk2 = *reinterpret_cast<const uint4*>(kr_ + 64); sk3 = *reinterpret_cast<const uint4*>(kr_ + 96); \
    VC_TA2_LOADV(0, sv0) VC_TA2_LOADV(1, sv1) VC_TA2_LOADV(2, sv2) VC_TA2_LOADV(3, sv3)          \
  }
#define VC_TA2_STAGE_STORE(st_arg)                                                               \
  {                                                                                              \
    char* st_ = (st_arg);                                                                        \
    *reinterpret_cast<uint4*>(st_ + (wave * NKS + 0) * 1024 + lane * 16) = sk0;                  \
    *reinterpret_cast<uint4*>(st_ + (wave * NKS + 1) * 1024 + lane * 16) = sk1;                  \
    *reinterpret_cast<uint4*>(st_ + (wave * NKS + 2) * 1024 + lane * 16) = sk2;                  \
    *reinterpret_cast<uint4*>(st_ + (wave * NKS + 3) * 1024 + lane * 16) = sk3;                  \
    *reinterpret_cast<uint4*>(st_ + KBYTES + (wave * 4 + 0) * 1024 + lane * 16) = sv0;           \
    *reinterpret_cast<uint4*>(st_ + KBYTES + (wave * 4 + 1) * 1024 + lane * 16) = sv1;           \
    *reinterpret_cast<uint4*>(st_ + KBYTES + (wave * 4 + 2) * 1024 + lane * 16) = sv2;           \
    *reinterpret_cast<uint4*>(st_ + KBYTES + (wave * 4 + 3) * 1024 + lane * 16) = sv3;           \
  }
  static_assert(NKS == 4, "the staging macros are written for head_dim 128");
  VC_TA2_STAGE_LOAD(0)
  // ---- Q fragments of the wave's row tile, pre-scaled to log2 units (B operand of S^T)
  uint4 qf[NKS];
  {
    const float qs = a.scale * 1.4426950408889634f;
    const float* qp = a.q + (long)min(r0 + 16 * wave + m, a.n_rows - 1) * a.d + h * HD + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      WT tmp[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) WTr<WT>::st(&tmp[j], qp[32 * ks + j] * qs);
      qf[ks] = *reinterpret_cast<const uint4*>(tmp);
    }
  }
  float mrun = -INFINITY, lsum = 0.f;                                    // lsum: this lane's share of the row sum (its own keys)
  f32x4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  VC_TA2_STAGE_STORE(smem)
  __syncthreads();
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int vkey = (((m >> 2) & 3) << 2) | kg;                           // v(key) of the keys this lane addresses in the transposed reads
  const int nkt = blk_last / KEYS + 1;
  for (int t = 0; t < nkt; ++t) {
    const int kt0 = t * KEYS;
    char* st = smem + (t & 1) * STAGE;
    const bool more = t + 1 < nkt;
    if (more) VC_TA2_STAGE_LOAD(kt0 + KEYS)                              // in flight during this tile's MFMAs
    if (na_w > 0 && kt0 <= t_last) {
      // ---- S^T = K Q^T: lane (query m, kg) gets keys kt0 + 16 sub + 4 kg + r
      f32x4 sc[4];
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const uint4 kf = *reinterpret_cast<const uint4*>(st + (sub * NKS + ks) * 1024 + lane * 16);
          sc[sub] = mfma_frag(kf, qf[ks], sc[sub], (WT*)nullptr);
        }
      }
      // ---- online softmax of ONE query row per lane
      float mx = -INFINITY;
      if (kt0 + KEYS - 1 <= tpos0) {                                     // interior tile: every key visible to every row of the tile
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[sub][r]);
      } else {                                                           // the causal mask cuts through it
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kp = kt0 + 16 * sub + 4 * kg + r;
            sc[sub][r] = (kp <= qpos) ? sc[sub][r] : -INFINITY;
            mx = fmaxf(mx, sc[sub][r]);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(mrun, mx);
      const float msafe = (mn == -INFINITY) ? 0.f : mn;                  // nothing visible yet: exp2(-inf - 0) = 0 everywhere
      const float corr = (mrun == mn) ? 1.f : fast_exp2(mrun - msafe);
      float ps = 0.f;
      uint4 pf[2];
      {
        float p[4][4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) { p[sub][r] = fast_exp2(sc[sub][r] - msafe); ps += p[sub][r]; }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          pf[c].x = pack_bf16x2(p[2 * c][0], p[2 * c][1]);
          pf[c].y = pack_bf16x2(p[2 * c][2], p[2 * c][3]);
          pf[c].z = pack_bf16x2(p[2 * c + 1][0], p[2 * c + 1][1]);
          pf[c].w = pack_bf16x2(p[2 * c + 1][2], p[2 * c + 1][3]);
        }
      }
      lsum = lsum * corr + ps;
      mrun = mn;
      if (__any(corr != 1.f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) { o[dt][0] *= corr; o[dt][1] *= corr; o[dt][2] *= corr; o[dt][3] *= corr; }
      }
      // ---- O^T += V^T P^T: two 32-key steps; A = V^T fragment of (dim tile dt, step c) through two transposed reads
      const char* vb = st + KBYTES + (4 * kg + (m >> 2)) * (HD * 2) + (m & 1) * 8;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const int slot = ((2 * dt + ((m >> 1) & 1)) ^ vkey) * 16;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + (2 * c) * 16 * (HD * 2) + slot));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + (2 * c + 1) * 16 * (HD * 2) + slot));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          uint4 vf;
          vf.x = l2.x; vf.y = l2.y; vf.z = h2.x; vf.w = h2.y;
          o[dt] = mfma_frag(vf, pf[c], o[dt], (WT*)nullptr);
        }
      }
    }
    if (more) VC_TA2_STAGE_STORE(smem + ((t + 1) & 1) * STAGE)           // (that buffer's readers left it before the last barrier)
    __syncthreads();
  }
  // ---- the four lanes of a query add up their shares of the row sum; O^T[dim = 16 dt + 4 kg + r][query m] -> x_out[row][dim]
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  const int row = r0 + 16 * wave + m;
  if (row < a.n_rows) {
    const float inv = (m < na_w && lsum > 0.f) ? 1.0f / lsum : 0.f;
    WT* xo = reinterpret_cast<WT*>(a.x_out) + (long)row * a.d + h * HD + 4 * kg;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      uint2 u;
      u.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
      u.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<uint2*>(xo + 16 * dt) = u;
    }
  }
}

#undef VC_TA2_LOADV
#undef VC_TA2_STAGE_LOAD
#undef VC_TA2_STAGE_STORE
hipError_t vc_launch_tile_attn64(const AttnArgs& a, hipStream_t s) {
  if (!a.x_out || a.hd != 128) return hipErrorInvalidValue;
  ++vc_launch_counts[VC_LC_TILE_ATTN];
  ++vc_launch_counts[VC_LC_TILE_ATTN64];        // (also counted as a tile attention launch: the tests written before this slot read that one)
  hipLaunchKernelGGL(tile_attn64_k, dim3(a.H, (a.n_rows + 63) / 64), dim3(256), 4 * VC_TA2_KEYS * 128 * 2, s, a);
  return hipGetLastError();
}

template <typename WT, int HD, int NW>
static hipError_t launch_tile_attn(const AttnArgs& a, hipStream_t s) {
  constexpr int VS = WTr<WT>::KW * (int)sizeof(WT) + 16;
  constexpr size_t vbytes = sizeof(WT) == 2 ? (size_t)(HD / 16) * (WTr<WT>::KW * 32 + 32) : (size_t)HD * VS;   // as in the kernel
  constexpr size_t wave_lds = vbytes + (size_t)16 * VS;
  constexpr size_t merge = (size_t)(NW * 16 * HD + NW * 16 * 2) * sizeof(float);
  constexpr size_t lds = (NW * wave_lds > merge) ? NW * wave_lds : merge;
  static_assert(lds <= 64 * 1024, "tile_attn_k: raise the dynamic LDS limit (hipFuncSetAttribute) for this shape");
  hipLaunchKernelGGL((tile_attn_k<WT, HD, NW>), dim3(a.H, (a.n_rows + 15) / 16), dim3(64 * NW), lds, s, a);
  return hipGetLastError();
}

// Prefill passes whose rows come in 16-row tiles of one sequence each (vc_engine.hip prefill_batch).
hipError_t vc_launch_tile_attn(const AttnArgs& a, int dtype, hipStream_t s) {
  if (!a.x_out) return hipErrorInvalidValue;
  ++vc_launch_counts[VC_LC_TILE_ATTN];
  if (dtype == VC_DTYPE_BF16) {
    if (a.hd == 128) return launch_tile_attn<bf16_t, 128, 4>(a, s);
    if (a.hd == 64) return launch_tile_attn<bf16_t, 64, 4>(a, s);
    if (a.hd == 32) return launch_tile_attn<bf16_t, 32, 4>(a, s);
  } else {
    if (a.hd == 128) return launch_tile_attn<float, 128, 4>(a, s);
    if (a.hd == 64) return launch_tile_attn<float, 64, 4>(a, s);
    if (a.hd == 32) return launch_tile_attn<float, 32, 4>(a, s);
  }
  return hipErrorInvalidValue;
}
