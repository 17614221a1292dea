// vc_fused.hip - the decode attention and the out-projection of a ONE-row step as ONE launch (round 5, option "fuse_ao").
//
// The two launches are the pair furthest from its roofline (attention 4.9 us for 4.6 MB of K/V, out-projection 5.0 us for 8.4 MB):
// both are dependency chains, and between them sits a launch boundary (~1.2-1.5 us: end-of-kernel write-back, dispatch, argument
// fetch) behind which the out-projection's 8.4 MB of weights - which depend on nothing the attention computes - are only then
// requested.  Here the grid holds both roles: workgroups [0, n_attn) are the attention's (head, split) blocks, the rest are the
// out-projection's (tile, K-half) blocks.  An out-projection workgroup requests its weights at once, then ONE lane polls a
// device-scope arrival counter until every attention workgroup has published its partial, and only then reads the partials.
//
// Hand-off (MI355X_MICROARCH "handoff-flag", the drained write-through form): a producer stores its partial with 16-byte
// `sc0 sc1` (write-through) stores, drains them (`s_waitcnt vmcnt(0)`), and takes a ticket on the counter with a relaxed
// agent-scope atomic; a consumer polls with relaxed agent-scope loads and reads the partials with loads that bypass its L1
// (non-temporal).  No L2 of this launch can hold a stale copy of a partial line: every kernel start invalidates the non-local
// lines, write-through stores drop the line from the writer's L2, and a consumer first touches a line after the counter says
// it is complete.  Workgroups are dispatched in launch order, so every producer is resident before any consumer starts
// spinning (both roles together need 384 workgroups of 512 threads: two per CU fit).  A replayed step after the last sequence
// retired: both roles leave before the counter is touched.  The counter is zeroed by workgroup 0 of the QKV launch in front
// (GemmArgs.zero_word), which the stream orders before this launch and after the previous layer's.
#include <math.h>
#include "vc_gemm_dev.h"

struct FusedArgs {
  AttnArgs at;              // the attention's arguments (one row: grid.x = 1; nsplit = VC_MAX_NSPLIT)
  GemmArgs og;              // the out-projection's (PRO_ATT / EPI_PART form: Wp, N, K, KT, n_tiles, part_out, rows_cap, att_o, att_ml, nsplit, H, hd, hd_shift)
  int* sync;                // arrival counter, zero at launch
  int* err;                 // engine error word: bit 2 = a consumer gave up waiting (never expected)
  int n_attn;               // attention workgroups = H * nsplit
  int ksplit;               // K slices of the out-projection (2)
};

template <typename WT>
__device__ __forceinline__ void fz_unpack16(const uint4& u, float* f);
template <>
__device__ __forceinline__ void fz_unpack16<float>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <>
__device__ __forceinline__ void fz_unpack16<bf16_t>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

__device__ __forceinline__ void store16_wt(float* p, const f32x4 v) {      // 16-byte write-through store (past L2, line dropped there)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_wt(float* p, const vc_f32x2 v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

constexpr int FZ_NW = 8;          // waves per workgroup, both roles

// ---------------------------------------------------------------- role A: one (head, split) block of the decode attention
// (rows_attn_k<WT, false, FAST = true> of vc_attn.hip for ONE row, same sums in the same order; the partial is PUBLISHED)
template <typename WT>
__device__ __forceinline__ void fused_attn_role(const FusedArgs& fa, char* smem) {
  const AttnArgs& a = fa.at;
  constexpr int EPL = WTr<WT>::EPL;
  constexpr int NW = FZ_NW;
  constexpr bool X2 = sizeof(WT) == 2;
  auto ex = [](float x) -> float {
    if constexpr (X2) return __builtin_amdgcn_exp2f(x);
    else return expf(x);
  };
  float* s_m = reinterpret_cast<float*>(smem);
  float* s_l = s_m + NW;
  float* s_o = s_l + NW;                          // [NW][128]
  float* s_fin = s_o + NW * 128;                  // [128 + 2]
  const int h = (int)blockIdx.x % a.H, sp = (int)blockIdx.x / a.H;
  int active, pos, seq, share;
  asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(active), "=&s"(pos), "=&s"(seq), "=&s"(share)
               : "s"(a.n_active), "s"(a.row_pos), "s"(a.row_seq), "s"(a.share_len)
               : "memory");
  if (active == 0) return;                        // (the consumers leave on the same word: nobody waits)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hd = a.hd;
  constexpr int EPL_SH = EPL == 8 ? 3 : 2;
  const int lpr_sh = a.hd_shift - EPL_SH;
  const int LPR = 1 << lpr_sh;
  const int PPW = 64 >> lpr_sh;
  const int sub = lane >> lpr_sh, li = lane & (LPR - 1);
  float m = -INFINITY, l = 0.f;
  float o[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) o[j] = 0.f;
  if (pos >= 0 && seq >= 0) {
    const int S = pos + 1;
    int chunk = (int)((float)(S + a.nsplit - 1) * a.inv_nsplit);
    if (chunk * a.nsplit < S) ++chunk;
    const int p0 = sp * chunk;
    const int p1 = min(S, p0 + chunk);
    const int step = 4 * NW * PPW;
    float4 qv[EPL / 4];
    {
      const float4* qp = reinterpret_cast<const float4*>(a.q + h * hd + li * EPL);
#pragma unroll
      for (int j = 0; j < EPL / 4; ++j) qv[j] = qp[j];
    }
    uint4 ku[4], vu[4];
    int pp[4];
    // one code path for the addresses: a shared text prefix is a feature of several-sequence calls, which never take this launch
    const long own = (long)seq * a.cache_seq_stride;
    const char* kbu = reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.kcache) + own + (long)h * a.S_max * hd);
    const char* vbu = reinterpret_cast<const char*>(reinterpret_cast<const WT*>(a.vcache) + own + (long)h * a.S_max * hd);
    const unsigned lo = (unsigned)(li * EPL) * (unsigned)sizeof(WT);
#define FZ_KV_LOADS(pb_)                                                     \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                       \
      pp[it] = (pb_) + (it * NW + wave) * PPW + sub;                         \
      const unsigned pc = (unsigned)max(min(pp[it], p1 - 1), 0);             \
      const long off = ((long)(pc << a.hd_shift) * (long)sizeof(WT)) + lo - ((int)pc < share ? own * (long)sizeof(WT) : 0L); \
      ku[it] = *reinterpret_cast<const uint4*>(kbu + off);                   \
      vu[it] = *reinterpret_cast<const uint4*>(vbu + off);                   \
    }
    FZ_KV_LOADS(p0)
    __builtin_amdgcn_sched_barrier(0);
    float q[EPL];
    const float qs = X2 ? a.scale * 1.4426950408889634f : a.scale;
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
      q[4 * j] = qv[j].x * qs; q[4 * j + 1] = qv[j].y * qs;
      q[4 * j + 2] = qv[j].z * qs; q[4 * j + 3] = qv[j].w * qs;
    }
    for (int pb = p0;;) {
      float sc[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        float kf[EPL];
        fz_unpack16<WT>(ku[it], kf);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) t += q[j] * kf[j];
        if (LPR == 4) t = quad_sum(t);
        else if (LPR == 8) t = half_row_sum(t);
        else { t = row_sum(t); if (LPR == 32) t += __shfl_xor(t, 16, 64); }
        sc[it] = (pp[it] < p1) ? t : -INFINITY;
      }
      const float mn = fmaxf(m, wave_max(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]))));
      if (mn > -INFINITY) {
        const float corr = ex(m - mn);
        float pw[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) pw[it] = ex(sc[it] - mn);
        l = l * corr + ((pw[0] + pw[1]) + (pw[2] + pw[3]));
        float vf[4][EPL];
#pragma unroll
        for (int it = 0; it < 4; ++it) fz_unpack16<WT>(vu[it], vf[it]);
#pragma unroll
        for (int j = 0; j < EPL; ++j)
          o[j] = o[j] * corr + ((pw[0] * vf[0][j] + pw[1] * vf[1][j]) + (pw[2] * vf[2][j] + pw[3] * vf[3][j]));
        m = mn;
      }
      pb += step;
      if (pb >= p1) break;
      FZ_KV_LOADS(pb)
    }
#undef FZ_KV_LOADS
  }
  for (int off = LPR; off < 64; off <<= 1) {
    l += __shfl_xor(l, off, 64);
#pragma unroll
    for (int j = 0; j < EPL; ++j) o[j] += __shfl_xor(o[j], off, 64);
  }
  if (lane < LPR) {
#pragma unroll
    for (int j = 0; j < EPL; ++j) s_o[wave * 128 + lane * EPL + j] = o[j];
    if (lane == 0) { s_m[wave] = m; s_l[wave] = l; }
  }
  __syncthreads();
  if (tid < hd) {
    float M = s_m[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, s_m[w]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float c = (s_m[w] == -INFINITY) ? 0.f : ex(s_m[w] - M);
      L += c * s_l[w];
      O += c * s_o[w * 128 + tid];
    }
    if constexpr (X2) M *= 0.6931471805599453f;      // natural units for the merge below (expf)
    s_fin[tid] = O;
    if (tid == 0) { s_fin[128] = M; s_fin[129] = L; }
  }
  __syncthreads();
  // publish: 16-byte write-through stores, drained; then the ticket
  const long pi = ((long)h * a.nsplit + sp);
  if (tid < (hd >> 2)) store16_wt(a.att_o + pi * hd + 4 * tid, *reinterpret_cast<const f32x4*>(s_fin + 4 * tid));
  if (tid == 0) store8_wt(a.att_ml + pi * 2, *reinterpret_cast<const vc_f32x2*>(s_fin + 128));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) (void)__hip_atomic_fetch_add(fa.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------- role B: one (tile, K slice) block of the out-projection
// (rows_gemm_k<WT, KTW, PRO_ATT, EPI_PART> of vc_gemm.hip for ONE row and 8 splits, eight waves; same merge arithmetic)
template <typename WT, int KTW>
__device__ __forceinline__ void fused_oproj_role(const FusedArgs& fa, char* smem) {
  using T = WTr<WT>;
  const GemmArgs& a = fa.og;
  constexpr int NW = FZ_NW, SPT = 64;               // 16-channel tiles: 64 fragment slots per (tile, k-tile)
  const int j = (int)blockIdx.x - fa.n_attn;
  const int nt = j % a.n_tiles, ks = j / a.n_tiles;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt_blk = NW * KTW;                      // k-tiles of this K slice
  const int kblk = kt_blk * T::KW, k0 = ks * kblk;
  char* xl = smem;                                  // the merged slice of x: kblk elements of WT
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)kblk * sizeof(WT));      // [NW][64]
  int* s_flag = reinterpret_cast<int*>(red + NW * 64);
  const int active = *a.n_active;
  // the weights depend on nothing this launch computes: requested at once (a wave's whole share in one burst)
  const uint4* wbase = a.Wp + ((long)nt * a.KT + (long)ks * kt_blk + wave * KTW) * SPT + lane;
  uint4 wf[KTW];
#pragma unroll
  for (int i = 0; i < KTW; ++i)
    wf[i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase + (long)i * SPT)));
  __builtin_amdgcn_sched_barrier(0);
  if (active == 0) return;
  // wait for every attention workgroup's partial: one lane polls, the others sleep at the barrier
  if (tid == 0) {
    // (bounded: ~30 ms of polling.  The launch order guarantees the producers are resident, so the bound is never met - it is there so
    // that a wrong assumption about the hardware shows up as an error flag (bit 2 of the engine's error word), not as a hung GPU)
    int spins = 0;
    while (__hip_atomic_load(fa.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < fa.n_attn && spins < (1 << 14)) { __builtin_amdgcn_s_sleep(2); ++spins; }
    if (spins >= (1 << 14) && fa.err) atomicOr(fa.err, 4);
    *s_flag = 1;
  }
  __syncthreads();
  asm volatile("" ::: "memory");
  // merge the 8 split partials of the slice's columns: item = 4 columns of one head; kblk / 4 items, one per thread
  const int n_items = kblk >> 2;
  if (tid < n_items) {
    const int c = k0 + tid * 4;
    const int hh = c >> a.hd_shift, e = c & (a.hd - 1);
    const float2* mlp = reinterpret_cast<const float2*>(a.att_ml) + (long)hh * VC_MAX_NSPLIT;
    const float* op = a.att_o + ((long)hh * VC_MAX_NSPLIT) * a.hd + e;
    float2 ml[VC_MAX_NSPLIT];
    float4 os[VC_MAX_NSPLIT];
#pragma unroll
    for (int s_ = 0; s_ < VC_MAX_NSPLIT; ++s_) {     // past this CU's L1 (the lines were written through by other CUs during this launch)
      ml[s_] = __builtin_bit_cast(float2, __builtin_nontemporal_load(reinterpret_cast<const vc_f32x2*>(mlp + s_)));
      os[s_] = __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(op + (long)s_ * a.hd)));
    }
    float M = -INFINITY;
#pragma unroll
    for (int s_ = 0; s_ < VC_MAX_NSPLIT; ++s_) M = fmaxf(M, ml[s_].x);
    float L = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s_ = 0; s_ < VC_MAX_NSPLIT; ++s_) {
      const float w = (ml[s_].x == -INFINITY) ? 0.f : expf(ml[s_].x - M);
      L += w * ml[s_].y;
      o[0] += w * os[s_].x; o[1] += w * os[s_].y; o[2] += w * os[s_].z; o[3] += w * os[s_].w;
    }
    const float inv = (L > 0.f) ? 1.0f / L : 0.f;
    o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
    store4(reinterpret_cast<WT*>(xl) + tid * 4, o);
  }
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const char* xrow = xl + (size_t)(lane >> 4) * 16;   // one row: every B column reads the row (the other 15 columns are cross terms)
#pragma unroll
  for (int i = 0; i < KTW; ++i) {
    const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)(wave * KTW + i) * 64);
    acc = mfma_frag(wf[i], xf, acc, (WT*)nullptr);
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && (lane & 15) == 0) {                // column 0 holds channels 4 kg .. 4 kg + 3
#pragma unroll
    for (int w = 1; w < NW; ++w) acc += red[w * 64 + lane];
    const int n = nt * 16 + 4 * (lane >> 4);
    store4(a.part_out + ((long)ks * a.rows_cap) * a.N + n, acc);
  }
}

template <typename WT, int KTW>
__global__ __launch_bounds__(64 * FZ_NW) void attn_oproj_k(const FusedArgs fa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < fa.n_attn) fused_attn_role<WT>(fa, smem);
  else fused_oproj_role<WT, KTW>(fa, smem);
}

// 1 when the fused launch can take this shape: 8 attention splits, two K slices of 8 waves x KTW whole k-tiles, grids that keep the
// tile -> XCD rule
int vc_fused_ao_ok(int d, int H, int nsplit, int ksplit, int dtype) {
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16;
  if (nsplit != VC_MAX_NSPLIT || ksplit < 1 || ksplit > 2 || d % 16 != 0 || d % H != 0) return 0;
  const int hd = d / H;
  if (hd != 32 && hd != 64 && hd != 128) return 0;
  const int kt = d / KW / ksplit;
  if (kt % FZ_NW != 0) return 0;
  const int ktw = kt / FZ_NW;
  if (ktw != 1 && ktw != 2 && ktw != 4 && ktw != 8) return 0;
  if ((d / ksplit) / 4 > 64 * FZ_NW) return 0;        // one merge item per thread
  return ((H * nsplit) % 8 == 0) ? 1 : 0;
}

hipError_t vc_launch_fused_ao(const AttnArgs& at, const GemmArgs& og, int ksplit, int* sync, int* err, int dtype, hipStream_t s) {
  if (!vc_fused_ao_ok(og.K, at.H, at.nsplit, ksplit, dtype) || at.n_rows != 1 || !sync) return hipErrorInvalidValue;
  FusedArgs fa;
  fa.at = at; fa.og = og; fa.sync = sync; fa.err = err; fa.n_attn = at.H * at.nsplit; fa.ksplit = ksplit;
  fa.at.inv_nsplit = nextafterf(1.0f / (float)at.nsplit, 2.0f);
  const int KW = dtype == VC_DTYPE_BF16 ? 32 : 16, esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  const int ktw = og.K / KW / ksplit / FZ_NW;
  const size_t lds_a = (size_t)(2 * FZ_NW + FZ_NW * 128 + 132) * sizeof(float);
  const size_t lds_b = (size_t)(og.K / ksplit) * esz + (size_t)FZ_NW * 64 * sizeof(f32x4) + 16;
  const size_t lds = lds_a > lds_b ? lds_a : lds_b;
  const dim3 grid(fa.n_attn + og.n_tiles * ksplit), block(64 * FZ_NW);
  ++vc_launch_counts[VC_LC_FUSED_AO];
#define VC_FZ_GO(WT_, K_) hipLaunchKernelGGL((attn_oproj_k<WT_, K_>), grid, block, lds, s, fa)
  if (dtype == VC_DTYPE_BF16) {
    switch (ktw) { case 1: VC_FZ_GO(bf16_t, 1); break; case 2: VC_FZ_GO(bf16_t, 2); break; case 4: VC_FZ_GO(bf16_t, 4); break; case 8: VC_FZ_GO(bf16_t, 8); break; default: return hipErrorInvalidValue; }
  } else {
    switch (ktw) { case 1: VC_FZ_GO(float, 1); break; case 2: VC_FZ_GO(float, 2); break; case 4: VC_FZ_GO(float, 4); break; case 8: VC_FZ_GO(float, 8); break; default: return hipErrorInvalidValue; }
  }
#undef VC_FZ_GO
  return hipGetLastError();
}
