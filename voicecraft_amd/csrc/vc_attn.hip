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

#define VC_ATT_WAVES 8
template <typename WT>
__global__ __launch_bounds__(64 * VC_ATT_WAVES) void rows_attn_k(const AttnArgs a) {
  constexpr int EPL = WTr<WT>::EPL;
  constexpr int NW = VC_ATT_WAVES;
  __shared__ float s_m[NW], s_l[NW];
  __shared__ float s_o[NW][128];
  VC_KTS_DECL();
  VC_KTS(0);
  const int r = blockIdx.x, h = blockIdx.y, sp = blockIdx.z;     // grid.x == n_rows
  // one scalar round trip for everything the addresses depend on (no early exit in between: a
  // branch would let the compiler serialise these three loads)
  const int active = *a.n_active;
  const int pos = a.row_pos[r];
  const int seq = a.row_seq[r];
  const int share = *a.share_len;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hd = a.hd;
  const int LPR = hd / EPL;            // lanes per cached row: 4 (bf16, head_dim 32), 8, 16 or 32
  const int PPW = 64 / LPR;            // positions per wave per step
  const int sub = lane / LPR, li = lane - sub * LPR;

  float m = -INFINITY, l = 0.f;
  float o[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) o[j] = 0.f;

  if (active != 0 && pos >= 0 && seq >= 0) {   // a finished step / an inactive row leaves an empty partial (seq is
                                               // tested only so that its load is not sunk behind the branch)
    VC_KTS(1);
    const int S = pos + 1;
    const int chunk = (S + a.nsplit - 1) / a.nsplit;
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
    const long hbase = (long)h * a.S_max * hd + li * EPL;
    const long base = (long)seq * a.cache_seq_stride + hbase;
    const WT* kb = reinterpret_cast<const WT*>(a.kcache) + base;
    const WT* vb = reinterpret_cast<const WT*>(a.vcache) + base;
    const long own = (long)seq * a.cache_seq_stride;      // positions below `share` live in sequence 0's cache
    uint4 ku[4], vu[4];
    int pp[4];
#define VC_KV_LOADS(pb_)                                                     \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                       \
      pp[it] = (pb_) + (it * NW + wave) * PPW + sub;                         \
      const long pc = max(min(pp[it], p1 - 1), 0);                           \
      const long po = pc * hd - ((pc < share) ? own : 0);                    \
      ku[it] = *reinterpret_cast<const uint4*>(kb + po);                     \
      vu[it] = *reinterpret_cast<const uint4*>(vb + po);                     \
    }
    VC_KV_LOADS(p0);
    __builtin_amdgcn_sched_barrier(0);
    VC_KTS(2);
    float q[EPL];
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
      q[4 * j] = qv[j].x * a.scale; q[4 * j + 1] = qv[j].y * a.scale;
      q[4 * j + 2] = qv[j].z * a.scale; q[4 * j + 3] = qv[j].w * a.scale;
    }
    VC_KTS(3);
    for (int pb = p0;;) {
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
      pb += step;
      if (pb >= p1) break;
      VC_KV_LOADS(pb);
    }
#undef VC_KV_LOADS
    VC_KTS(4);
  }
  // merge the PPW position groups of this wave (same li, different sub)
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
      const float c = (s_m[w] == -INFINITY) ? 0.f : expf(s_m[w] - M);
      L += c * s_l[w];
      O += c * s_o[w][tid];
    }
    if (a.x_out) {      // unsplit pass (prefill): the block saw every position, so it normalises itself
      WTr<WT>::st(reinterpret_cast<WT*>(a.x_out) + (long)r * a.d + h * hd + tid, (L > 0.f) ? O / L : 0.f);
    } else {
      const long pi = ((long)(r * a.H + h) * a.nsplit + sp);
      a.att_o[pi * hd + tid] = O;
      if (tid == 0) { a.att_ml[pi * 2] = M; a.att_ml[pi * 2 + 1] = L; }
    }
  }
  VC_KTS(7);
  VC_KTS_FLUSH();
}

hipError_t vc_launch_attn(const AttnArgs& a, int dtype, int rows_cap, hipStream_t s) {
  dim3 grid(rows_cap, a.H, a.nsplit);
  if (dtype == VC_DTYPE_BF16) hipLaunchKernelGGL(rows_attn_k<bf16_t>, grid, dim3(64 * VC_ATT_WAVES), 0, s, a);
  else hipLaunchKernelGGL(rows_attn_k<float>, grid, dim3(64 * VC_ATT_WAVES), 0, s, a);
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
