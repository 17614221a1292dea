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
// Block = 4 waves sharing one 16-row output tile; the waves split the block's K range 4 ways and
// reduce through LDS.  Cross-block split-K (EPI_PART) leaves fp32 partial slabs that the NEXT
// kernel's LayerNorm prologue sums (launch-boundary reduce: no atomics, deterministic).
#include "vc_common.h"

// ------------------------------------------------------------------ packing
template <typename WT>
__global__ void pack_k(const float* __restrict__ src, WT* __restrict__ dst, int N, int K, int KT,
                       long total) {
  constexpr int EPL = WTr<WT>::EPL, KW = WTr<WT>::KW;
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (tile, lane)
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long tile = idx >> 6;
  const int kt = (int)(tile % KT);
  const int nt = (int)(tile / KT);
  const int n = nt * 16 + (lane & 15);
  const int k = kt * KW + EPL * (lane >> 4);
  WT* d = dst + idx * EPL;
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    float v = (n < N) ? src[(long)n * K + k + j] : 0.0f;
    WTr<WT>::st(d + j, v);
  }
}

hipError_t vc_launch_pack(const float* src, void* dst, int N, int K, int dtype, hipStream_t s) {
  const int n_tiles = (N + 15) / 16;
  if (dtype == VC_DTYPE_BF16) {
    const int KT = K / 32;
    long total = (long)n_tiles * KT * 64;
    hipLaunchKernelGGL(pack_k<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src,
                       (bf16_t*)dst, N, K, KT, total);
  } else {
    const int KT = K / 16;
    long total = (long)n_tiles * KT * 64;
    hipLaunchKernelGGL(pack_k<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src,
                       (float*)dst, N, K, KT, total);
  }
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
#define VC_LN_MAXV4 8      // d <= 64 lanes * 4 floats * 8 = 2048 (checked in vc_create)

__device__ __forceinline__ void store4(float* p, const f32x4& v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v) {
  uint2 u;
  u.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  u.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = u;
}

template <typename WT, int KTW, int PRO, int EPI>
__global__ __launch_bounds__(256) void rows_gemm_k(const GemmArgs a) {
  using T = WTr<WT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (a.n_active && *a.n_active == 0) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockIdx.x, ks = blockIdx.y, grp = blockIdx.z;
  const int n_rows = a.n_rows_ptr ? *a.n_rows_ptr : a.n_rows;
  const int kt_blk = a.nchunk * 4 * KTW;          // k-tiles this block covers
  const int kt0 = ks * kt_blk;                    // first of them
  const int kblk = kt_blk * T::KW;                // K elements this block covers
  const int k0 = kt0 * T::KW;
  const int xs = kblk * (int)sizeof(WT) + 16;     // LDS row stride in bytes (+16: rotate bank slots)
  char* xl = smem;
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)a.r_lds * xs);

  // (1) put this wave's first weight burst in flight before anything else: it does not depend on X.
  const uint4* wp = a.Wp + (long)grp * a.w_group_stride + ((long)nt * a.KT) * 64 + lane;
  uint4 wf[KTW];
  {
    const int kt = kt0 + wave * KTW;
#pragma unroll
    for (int i = 0; i < KTW; ++i) wf[i] = wp[(long)(kt + i) * 64];
  }

  // (2) prologue: build the rows' X slice [n_rows][kblk] as WT in LDS.
  if constexpr (PRO == PRO_LN) {
    // one wave per row: two-pass LayerNorm in registers (eps 1e-5, transformer.py:30)
    const int d = a.d;
    const int nv4 = d >> 8;                       // float4 per lane
    for (int r = wave; r < n_rows; r += 4) {
      const int sr = a.gather_rows ? a.gather_rows[r] : r;
      const float* hp = a.h_in + (long)sr * d;
      float4 v[VC_LN_MAXV4];
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < VC_LN_MAXV4; ++i) {
        if (i < nv4) {
          const int c = (i * 64 + lane) * 4;
          float4 x = *reinterpret_cast<const float4*>(hp + c);
          if (a.prev_bias) {
            const float4 b = *reinterpret_cast<const float4*>(a.prev_bias + c);
            x.x += b.x; x.y += b.y; x.z += b.z; x.w += b.w;
          }
          for (int s = 0; s < a.n_parts; ++s) {
            const float4 p = *reinterpret_cast<const float4*>(a.parts + ((long)(s * VC_ROWS + sr)) * d + c);
            x.x += p.x; x.y += p.y; x.z += p.z; x.w += p.w;
          }
          v[i] = x;
          sum += (x.x + x.y) + (x.z + x.w);
        }
      }
      const float mean = wave_sum(sum) / (float)d;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < VC_LN_MAXV4; ++i) {
        if (i < nv4) {
          const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
          sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
      }
      const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + 1e-5f);
      const bool writer = a.h_out && grp == 0 && ((r % (int)gridDim.x) == (int)blockIdx.x);
#pragma unroll
      for (int i = 0; i < VC_LN_MAXV4; ++i) {
        if (i < nv4) {
          const int c = (i * 64 + lane) * 4;
          if (writer) *reinterpret_cast<float4*>(a.h_out + (long)r * d + c) = v[i];
          const float4 g = *reinterpret_cast<const float4*>(a.ln_w + c);
          const float4 b = *reinterpret_cast<const float4*>(a.ln_b + c);
          f32x4 y;
          y[0] = (v[i].x - mean) * rstd * g.x + b.x;
          y[1] = (v[i].y - mean) * rstd * g.y + b.y;
          y[2] = (v[i].z - mean) * rstd * g.z + b.z;
          y[3] = (v[i].w - mean) * rstd * g.w + b.w;
          store4(reinterpret_cast<WT*>(xl + (size_t)r * xs) + c, y);
        }
      }
    }
  } else if constexpr (PRO == PRO_PLAIN) {
    const int upr = kblk * (int)sizeof(WT) / 16;  // 16-byte units per row
    const char* src = reinterpret_cast<const char*>(a.x_in);
    for (int idx = tid; idx < n_rows * upr; idx += 256) {
      const int r = idx / upr, u = idx - r * upr;
      const long off = ((long)r * a.x_ld + (long)grp * a.x_group_stride + k0) * (long)sizeof(WT) + (long)u * 16;
      *reinterpret_cast<uint4*>(xl + (size_t)r * xs + (size_t)u * 16) = *reinterpret_cast<const uint4*>(src + off);
    }
  } else {  // PRO_ATT: merge the split-S partials of the decode attention (softmax denominators)
    const int q4 = kblk >> 2;
    for (int idx = tid; idx < n_rows * q4; idx += 256) {
      const int r = idx / q4, c = k0 + (idx - r * q4) * 4;
      const int h = c / a.hd, e = c - h * a.hd;
      const float* ml = a.att_ml + ((long)(r * a.H + h) * a.nsplit) * 2;
      const float* op = a.att_o + ((long)(r * a.H + h) * a.nsplit) * a.hd + e;
      float M = -INFINITY;
      for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, ml[2 * s]);
      float L = 0.f;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < a.nsplit; ++s) {
        const float ms = ml[2 * s];
        if (ms == -INFINITY) continue;
        const float w = expf(ms - M);
        L += w * ml[2 * s + 1];
        const float4 os = *reinterpret_cast<const float4*>(op + (long)s * a.hd);
        o[0] += w * os.x; o[1] += w * os.y; o[2] += w * os.z; o[3] += w * os.w;
      }
      const float inv = (L > 0.f) ? 1.0f / L : 0.f;
      o[0] *= inv; o[1] *= inv; o[2] *= inv; o[3] *= inv;
      store4(reinterpret_cast<WT*>(xl + (size_t)r * xs) + (c - k0), o);
    }
  }
  __syncthreads();

  // (3) main loop: one ds_read_b128 + one MFMA per 1 KiB weight burst
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int mrow = ((lane & 15) < a.r_lds) ? (lane & 15) : 0;
  const char* xrow = xl + (size_t)mrow * xs + (size_t)(lane >> 4) * 16;
  for (int c = 0; c < a.nchunk; ++c) {
    const int ktl = (c * 4 + wave) * KTW;
#pragma unroll
    for (int i = 0; i < KTW; ++i) {
      const uint4 xf = *reinterpret_cast<const uint4*>(xrow + (size_t)(ktl + i) * 64);
      acc = mfma_frag(wf[i], xf, acc, (WT*)nullptr);
    }
    if (c + 1 < a.nchunk) {   // refill the same registers; co-resident blocks cover the latency
      const int kt = kt0 + ((c + 1) * 4 + wave) * KTW;
#pragma unroll
      for (int i = 0; i < KTW; ++i) wf[i] = wp[(long)(kt + i) * 64];
    }
  }

  // (4) 4-way in-block K reduction, then the epilogue on wave 0
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  {
    const f32x4 a1 = red[64 + lane], a2 = red[128 + lane], a3 = red[192 + lane];
    acc = (acc + a1) + (a2 + a3);
  }
  const int m = lane & 15;
  const int n = nt * 16 + 4 * (lane >> 4);
  if (m >= n_rows) return;

  if constexpr (EPI == EPI_PART) {
    if (n < a.N) store4(a.part_out + ((long)(ks * VC_ROWS + m)) * a.N + n, acc);
  } else if constexpr (EPI == EPI_QKV) {
    if (n >= a.N) return;
    const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
    acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w;
    const int d = a.d;
    if (n < d) {
      store4(a.q_out + (long)m * d + n, acc);
    } else {
      const int which = (n - d) / d;
      const int c = (n - d) - which * d;
      const int h = c / a.hd, e = c - h * a.hd;
      const int pos = a.row_pos[m];
      if (pos >= 0) {
        WT* base = reinterpret_cast<WT*>(which ? a.vcache : a.kcache) +
                   (long)a.row_seq[m] * a.cache_seq_stride + ((long)h * a.S_max + pos) * a.hd + e;
        store4(base, acc);
      }
    }
  } else if constexpr (EPI == EPI_RELU || EPI == EPI_GELU) {
    if (n >= a.N) return;
    const float4 b = *reinterpret_cast<const float4*>(a.bias + (long)grp * a.bias_group_stride + n);
    acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (EPI == EPI_RELU) acc[j] = fmaxf(acc[j], 0.f);
      else acc[j] = 0.5f * acc[j] * (1.0f + erff(acc[j] * 0.70710678118654752440f));  // nn.GELU() exact erf
    }
    store4(reinterpret_cast<WT*>(a.out) + (long)m * a.out_ld + (long)grp * a.out_group_stride + n, acc);
  } else {  // EPI_LOGITS: float [row][group][N], N need not be a multiple of 4
    float* o = reinterpret_cast<float*>(a.out) + ((long)m * gridDim.z + grp) * a.N;
    const float* b = a.bias + (long)grp * a.bias_group_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < a.N) o[n + j] = acc[j] + b[n + j];
  }
}

// ------------------------------------------------------------------ dispatch
size_t vc_gemm_lds_bytes(const GemmArgs& a, int dtype, int ksplit) {
  const int esz = dtype == VC_DTYPE_BF16 ? 2 : 4;
  const size_t xs = (size_t)(a.K / ksplit) * esz + 16;
  return (size_t)a.r_lds * xs + 4 * 64 * sizeof(f32x4);
}

template <typename WT, int KTW, int PRO, int EPI>
static hipError_t launch_one(const GemmArgs& a, int dtype, int ksplit, int groups, hipStream_t s) {
  auto kern = rows_gemm_k<WT, KTW, PRO, EPI>;
  const size_t lds = vc_gemm_lds_bytes(a, dtype, ksplit);
  if (lds > 64 * 1024) {
    static size_t granted = 0;   // per instantiation
    if (lds > granted) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      granted = lds;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.n_tiles, ksplit, groups), dim3(256), lds, s, a);
  return hipGetLastError();
}

template <typename WT, int KTW>
static hipError_t launch_ktw(const GemmArgs& a, int dtype, int pro, int epi, int ksplit, int groups,
                             hipStream_t s) {
  if (pro == PRO_LN && epi == EPI_QKV) return launch_one<WT, KTW, PRO_LN, EPI_QKV>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LN && epi == EPI_RELU) return launch_one<WT, KTW, PRO_LN, EPI_RELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_LN && epi == EPI_GELU) return launch_one<WT, KTW, PRO_LN, EPI_GELU>(a, dtype, ksplit, groups, s);
  if (pro == PRO_ATT && epi == EPI_PART) return launch_one<WT, KTW, PRO_ATT, EPI_PART>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_PART) return launch_one<WT, KTW, PRO_PLAIN, EPI_PART>(a, dtype, ksplit, groups, s);
  if (pro == PRO_PLAIN && epi == EPI_LOGITS) return launch_one<WT, KTW, PRO_PLAIN, EPI_LOGITS>(a, dtype, ksplit, groups, s);
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
  if (dtype == VC_DTYPE_BF16) return launch_wt<bf16_t>(a, dtype, pro, epi, ksplit, groups, s);
  return launch_wt<float>(a, dtype, pro, epi, ksplit, groups, s);
}
