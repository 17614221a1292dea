// vc_gemm_dev.h - device helpers shared by the decode GEMMs (vc_gemm.hip) and the prefill block GEMMs (vc_gemm_pf.hip).
#pragma once
#include "vc_common.h"

__device__ __forceinline__ void store4(float* p, const f32x4& v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const f32x4& v) {
  uint2 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

// store and return the values as the MFMA will read them back (identity in fp32 mode)
__device__ __forceinline__ f32x4 store4r(float* p, const f32x4& v) { store4(p, v); return v; }
__device__ __forceinline__ f32x4 store4r(bf16_t* p, const f32x4& v) {
  uint2 u;
  u.x = pack_bf16x2(v[0], v[1]);
  u.y = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
  const f32x4 r = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                   __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  return r;
}

// Operands of the fused epilogue that live in HBM (bias, the row's cache slot).  Requested by every
// lane at the very top of the kernel - a dependent load at the END of a 10 us kernel is a full
// round trip on the critical path.  mg < n_rows (host contract), n is clamped here.
template <typename WT, int EPI>
__device__ __forceinline__ void epi_preload(const GemmArgs& a, int mg, int n, int grp, float4& b, int& pos, int& seq);
// LN prologue: the row sums of the folded weights, same indexing as the bias
template <typename WT, int EPI>
__device__ __forceinline__ float4 wg_preload(const GemmArgs& a, int n, int grp) {
  const int nc = (n < a.N) ? n : 0;
  if constexpr (vc_is_qkv(EPI)) return *reinterpret_cast<const float4*>(a.wg + nc);
  else return *reinterpret_cast<const float4*>(a.wg + (long)grp * a.bias_group_stride + nc);
}
template <typename WT, int EPI>
__device__ __forceinline__ void epi_preload(const GemmArgs& a, int mg, int n, int grp, float4& b, int& pos, int& seq) {
  b = make_float4(0.f, 0.f, 0.f, 0.f);
  pos = -1;
  seq = 0;
  const int nc = (n < a.N) ? n : 0;
  if constexpr (vc_is_qkv(EPI)) {
    b = *reinterpret_cast<const float4*>(a.bias + nc);
    pos = a.row_pos[mg];
    seq = a.row_seq[mg];
  } else if constexpr (EPI == EPI_RELU || EPI == EPI_GELU) {
    b = *reinterpret_cast<const float4*>(a.bias + (long)grp * a.bias_group_stride + nc);
  } else if constexpr (EPI == EPI_LOGITS) {   // N need not be a multiple of 4
    const float* bp = a.bias + (long)grp * a.bias_group_stride;
    const int last = a.N - 1;
    b.x = bp[min(n, last)]; b.y = bp[min(n + 1, last)]; b.z = bp[min(n + 2, last)]; b.w = bp[min(n + 3, last)];
  }
}

// fused epilogue of one lane: output channels n..n+3 of row mg (mg = global row index of the pass)
template <typename WT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 acc, int mg, int n, int ks, int grp, int ngroups,
                                              const float4& b, int pos, int seq) {
  if constexpr (EPI == EPI_PART) {
    if (n < a.N) store4(a.part_out + ((long)(ks * a.rows_cap + mg)) * a.N + n, acc);
  } else if constexpr (vc_is_qkv(EPI)) {
    if (n < a.N) {
      acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w;
      const int d = a.d;
      if (n < d) {
        store4(a.q_out + (long)mg * d + n, acc);
      } else {
        const int which = (n - d) >= d ? 1 : 0;          // n < 3 d: K or V
        const int c = (n - d) - which * d;
        const int h = c >> a.hd_shift, e = c & (a.hd - 1);
        if (pos >= 0) {
          WT* base = reinterpret_cast<WT*>(which ? a.vcache : a.kcache) +
                     (long)seq * a.cache_seq_stride + ((long)h * a.S_max + pos) * a.hd + e;
          store4(base, acc);
        }
      }
    }
  } else if constexpr (EPI == EPI_RELU || EPI == EPI_GELU) {
    if (n < a.N) {
      acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (EPI == EPI_RELU) acc[j] = fmaxf(acc[j], 0.f);
        else acc[j] = 0.5f * acc[j] * (1.0f + erff(acc[j] * 0.70710678118654752440f));  // nn.GELU() exact erf
      }
      store4(reinterpret_cast<WT*>(a.out) + (long)mg * a.out_ld + (long)grp * a.out_group_stride + n, acc);
    }
  } else {  // EPI_LOGITS: float [row][group][N]
    float* o = reinterpret_cast<float*>(a.out) + ((long)mg * ngroups + grp) * a.N;
    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n + j < a.N) o[n + j] = acc[j] + bb[j];
  }
}

// Epilogue of the block GEMMs (rows_gemm_blk_k, rows_gemm_big_k): the lane holds channels n .. n+3 (n = nt TH + 4 kg) of
// MT x NTW (row tile, weight tile) pairs, rows row0 + 16 i.  THREE passes: every operand from HBM (the rows' cache slots)
// is requested in one batch; every value is finished (bias, activation, bf16 packing) and PINNED in its own registers;
// then the stores go out back to back.  Written as one loop, the compiler sinks the arithmetic into the predicated store
// blocks and re-uses the data registers - and on gfx950 a store's data registers may only be overwritten once the store
// has COMPLETED (it emits s_waitcnt vmcnt(0) before every store: each of a lane's 8-32 stores waits out the previous
// one's round trip; 25 us of the 256 x 256 QKV launch, 10 us of every other form).  Same results, same addresses as
// gemm_epilogue (the decode kernels keep that one: a single store per lane).
template <typename WT, int EPI, int MT, int NTW>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& a, f32x4 (&acc)[MT][NTW], const float4 (&ebias)[NTW],
                                              int row0, int nt0, int TH, int kg, int ks, int n_rows) {
  static_assert(vc_is_qkv(EPI) || EPI == EPI_PART || EPI == EPI_RELU, "block GEMM epilogues");
  constexpr bool PACK = sizeof(WT) == 2;
  int n[NTW];
  bool ok[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    n[j] = (nt0 + j) * TH + 4 * kg;
    ok[j] = 4 * kg < TH && nt0 + j < a.n_tiles && n[j] < a.N;
  }
  long rowoff[MT], choff[NTW];
  bool isq[NTW], isv[NTW];
  if constexpr (vc_is_qkv(EPI)) {
    int pos[MT], seq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mg = min(row0 + 16 * i, n_rows - 1);
      pos[i] = a.row_pos[mg];
      seq[i] = a.row_seq[mg];
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {               // which matrix / head / element the lane's four channels are (once, not per store)
      const int d = a.d;
      isq[j] = n[j] < d;
      const int c0 = max(n[j] - d, 0);
      isv[j] = c0 >= d;
      const int c = c0 - (isv[j] ? d : 0);
      const int h = c >> a.hd_shift;
      choff[j] = (long)h * a.S_max * a.hd + (c & (a.hd - 1));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) rowoff[i] = (pos[i] >= 0) ? (long)seq[i] * a.cache_seq_stride + (long)pos[i] * a.hd : -1;
  }
  // ---- values
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      f32x4 v = acc[i][j];
      if constexpr (EPI != EPI_PART) { v[0] += ebias[j].x; v[1] += ebias[j].y; v[2] += ebias[j].z; v[3] += ebias[j].w; }
      if constexpr (EPI == EPI_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if constexpr (PACK && EPI != EPI_PART) {    // bf16 outputs ride in the first two registers (q of the QKV form stays fp32)
        const float p0 = __uint_as_float(pack_bf16x2(v[0], v[1])), p1 = __uint_as_float(pack_bf16x2(v[2], v[3]));
        const bool keep = vc_is_qkv(EPI) && isq[j];
        v[0] = keep ? v[0] : p0;
        v[1] = keep ? v[1] : p1;
      }
      asm volatile("" : "+v"(v));
      acc[i][j] = v;
    }
  // ---- stores
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mg = row0 + 16 * i;
    if (mg >= n_rows) continue;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if (!ok[j]) continue;
      const f32x4 v = acc[i][j];
      if constexpr (EPI == EPI_PART) {
        *reinterpret_cast<f32x4*>(a.part_out + ((long)(ks * a.rows_cap + mg)) * a.N + n[j]) = v;
      } else if constexpr (vc_is_qkv(EPI)) {
        if (isq[j]) {
          *reinterpret_cast<f32x4*>(a.q_out + (long)mg * a.d + n[j]) = v;
        } else if (rowoff[i] >= 0) {
          WT* dst = reinterpret_cast<WT*>(isv[j] ? a.vcache : a.kcache) + rowoff[i] + choff[j];
          if constexpr (PACK) *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1]));
          else *reinterpret_cast<f32x4*>(dst) = v;
        }
      } else {
        WT* dst = reinterpret_cast<WT*>(a.out) + (long)mg * a.out_ld + n[j];
        if constexpr (PACK) *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(v[0]), __float_as_uint(v[1]));
        else *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
}

