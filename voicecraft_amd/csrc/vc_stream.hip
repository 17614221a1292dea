// vc_stream.hip — the batch-1 decode step as ONE persistent launch over all decoder layers ("stream engine").
//
// Replaces, for one sequence and one row per step, the 5 launches per layer of forward_rows (vc_engine.hip): LayerNorm +
// QKV, attention over the in-place cache, out-projection + residual, LayerNorm + FFN up-projection + ReLU, FFN
// down-projection + residual (models/modules/transformer.py:266-343, activation.py:513-652) for ALL layers of a step.
// A step at batch 1 is a pure weight stream (1 FLOP per byte) cut into ~80 dependent launches; what those launches
// pay for is the dependency chain at every boundary (drain, dispatch, first round trip).  Here (MI355X_MICROARCH.md,
// "Persistent kernels" price list: engine-vs-launches, prefetch-credit, allgather, ldsdma-fill):
//
//   grid        G = d / 8 workgroups, one per CU (d = 2048: 256), resident together for the whole step
//   workgroup   wave 0 = LOADER: streams this CU's share of every layer's weights (d = 2048: 384 KB per layer) with
//               LDS-DMA (global_load_lds_dwordx4, non-temporal) into a ring of 8 x 16 KB LDS slots, running AHEAD of the
//               consumers through every dependency edge - the weights of an op do not depend on its input;
//               waves 1..4 = CONSUMERS: wait for a slot, multiply it against the op's input vector held in LDS
//               (v_dot2c_f32_bf16, fp32 accumulation), reduce, run the op's epilogue and PUBLISH their outputs
//   hand-offs   every value that crosses CUs travels as an 8-byte granule {value bits, epoch tag} written with ONE
//               relaxed agent-scope store and polled with relaxed agent-scope loads (value and tag cannot tear, no
//               fence, no flag): the consumers of an op sweep the granules of its input into LDS, then meet at a
//               consumer-only barrier (an LDS counter - the loader never takes part)
//   per CU      QKV 24 channels, attention = (head, 1/NS of the positions), out-projection 8, FFN-up 32, FFN-down 8
//               channels (K = d, d, d, 4d) - the same shape at every model width because G scales with d
//   per layer   six edges: residual -> QKV, q (+ the new k, v) -> attention, partials -> head leader, merged heads ->
//               out-projection, residual' -> FFN-up, activations -> FFN-down
//
// The KV cache keeps its layout (written here with plain stores for LATER steps; the position written in this step
// reaches the attention of this step as granules).  bf16 weights and inputs, fp32 accumulation, fp32 residual stream and
// statistics - the arithmetic of the launch path (rows_gemm_k / rows_attn_k) in a different summation order.
// Every wait is bounded: a wait that gives up raises ctl[1] and the whole grid drains (the host reports VC_EHIP).
#include "vc_common.h"
#include "vc_stream.h"

#define SG_CONS 4                      // consumer waves
#define SG_THREADS ((SG_CONS + 1) * 64)
#define SG_SLOT 16384                  // bytes: [4 consumers][4 rows][64 lanes][16 B]
#define SG_RING 8
#define SG_LDS_XB (SG_RING * SG_SLOT)  // op input as bf16 (up to 4d elements = 16 KB); also the head leader's partial table
#define SG_LDS_HB (SG_LDS_XB + 16384)  // residual row, fp32 [d]
#define SG_LDS_MISC (SG_LDS_HB + 8192)
#define SG_LDS_TOTAL (SG_LDS_MISC + 8192)       // 163 840 = all of a CU's LDS
#define SG_SPIN_LDS (1 << 21)          // bound of an LDS wait (~0.1 s)
#define SG_SPIN_GLOBAL (1 << 17)       // bound of a granule sweep (~0.1 s)

// misc region (byte offsets from SG_LDS_MISC)
#define SG_M_FILL 0                    // u32: slots landed so far (loader)
#define SG_M_ABORT 4                   // u32: set by any wave that gives up
#define SG_M_CBAR 8                    // u32: consumer barrier counter
#define SG_M_DONE 16                   // u32[4]: slots released, per consumer
#define SG_M_Q 128                     // float[128]: q of this CU's head
#define SG_M_KNEW 640                  // float[128]
#define SG_M_VNEW 1152                 // float[128]
#define SG_M_WML 1664                  // float[4][2]: per-wave (max, sum)
#define SG_M_WO 1696                   // float[4][128]: per-wave un-normalised output
#define SG_M_END 3744

typedef __attribute__((ext_vector_type(2))) __bf16 sg_bf16x2;

// ---------------------------------------------------------------- small device helpers
__device__ __forceinline__ unsigned sg_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ unsigned sg_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sg_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct SgCtx {
  char* smem;
  unsigned* fill;
  unsigned* abort_w;
  unsigned* cbar;
  unsigned* done;
  unsigned* ctl;
  unsigned bar_target;                 // consumer barrier: next value of cbar to wait for
  unsigned tag;
  int cw, lane, tid_c;                 // consumer index 0..3, lane, consumer-thread index 0..255
};

__device__ __forceinline__ void sg_give_up(SgCtx& c, unsigned code) {
  sg_st(c.abort_w, 1u);
  __hip_atomic_store(c.ctl + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// true = condition met; false = the workgroup is aborting
#define SG_WAIT_LDS(c_, cond_, code_)                                                            \
  ([&]() -> bool {                                                                               \
    for (int it_ = 0; it_ < SG_SPIN_LDS; ++it_) {                                                \
      if (cond_) return true;                                                                    \
      if ((it_ & 63) == 63) {                                                                    \
        if (sg_ld((c_).abort_w)) return false;                                                   \
        if ((it_ & 4095) == 4095 && __hip_atomic_load((c_).ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sg_st((c_).abort_w, 1u); return false; } \
      }                                                                                          \
      __builtin_amdgcn_s_sleep(1);                                                               \
    }                                                                                            \
    sg_give_up((c_), (code_));                                                                   \
    return false;                                                                                \
  })()

// consumer-only barrier (4 waves): LDS writes before it are visible to every consumer after it
__device__ __forceinline__ bool sg_cbarrier(SgCtx& c) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  c.bar_target += SG_CONS;
  if (c.lane == 0) __hip_atomic_fetch_add(c.cbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned target = c.bar_target;
  const bool ok = SG_WAIT_LDS(c, (int)(sg_ld(c.cbar) - target) >= 0, 0x10u);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok;
}

__device__ __forceinline__ void sg_publish(unsigned long long* g, unsigned tag, unsigned bits) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sweep n granules (n <= 4096) into LDS through `put(idx, low 32 bits)`; every consumer lane takes idx = tid_c + 256 p.
// Returns false when the wait gave up.
template <typename PUT>
__device__ __forceinline__ bool sg_gather(SgCtx& c, const unsigned long long* g, int n, PUT put) {
  const int per = (n + 255) >> 8;
  for (int it = 0; it < SG_SPIN_GLOBAL; ++it) {
    unsigned long long v[16];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int idx = c.tid_c + 256 * p;
      v[p] = (p < per && idx < n) ? __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : ((unsigned long long)c.tag << 32);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) ok = ok && ((unsigned)(v[p] >> 32) == c.tag);
    if (ok) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int idx = c.tid_c + 256 * p;
        if (p < per && idx < n) put(idx, (unsigned)v[p]);
      }
      return true;
    }
    if ((it & 15) == 15 && sg_ld(c.abort_w)) return false;
    if ((it & 1023) == 1023 && __hip_atomic_load(c.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sg_st(c.abort_w, 1u); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  sg_give_up(c, 0x20u);
  return false;
}

__device__ __forceinline__ float sg_dot8(const uint4& w, const uint4& x, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.x), __builtin_bit_cast(sg_bf16x2, x.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.y), __builtin_bit_cast(sg_bf16x2, x.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.z), __builtin_bit_cast(sg_bf16x2, x.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.w), __builtin_bit_cast(sg_bf16x2, x.w), acc, false);
  return acc;
}

// One op of the stream for this consumer wave: NC channels of `rows_per_ch` 1 KB rows each, starting at global slot
// `gslot0`; acc[j] = sum_k W[channel j][k] * X[k] (X = bf16 vector in LDS at xb, chunk (64 i + lane) for row i of a channel).
template <int NC>
__device__ __forceinline__ bool sg_consume(SgCtx& c, int gslot0, int rows_per_ch, int op_slots, const char* xb, float (&acc)[NC]) {
  int r = 0;
  const char* ring = c.smem;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    float s = 0.f;
    for (int i = 0; i < rows_per_ch; ++i, ++r) {
      const int slot = gslot0 + (r >> 2), row4 = r & 3;
      if (row4 == 0) ok = ok && SG_WAIT_LDS(c, (int)(sg_ld(c.fill) - (unsigned)slot) > 0, 0x30u);
      if (!ok) break;
      const uint4 wv = *reinterpret_cast<const uint4*>(ring + (size_t)(slot & (SG_RING - 1)) * SG_SLOT + c.cw * 4096 + row4 * 1024 + c.lane * 16);
      const uint4 xv = *reinterpret_cast<const uint4*>(xb + (size_t)(i * 64 + c.lane) * 16);
      s = sg_dot8(wv, xv, s);
      if (row4 == 3) {     // this consumer is through with the slot (its reads have been executed: LDS ops of a wave run in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (c.lane == 0) sg_st(c.done + c.cw, (unsigned)(slot + 1));
      }
    }
    acc[j] = wave_sum(s);
  }
  if (!ok) return false;
  // padding rows of the op's last slot (narrow models only) and nothing else may remain
  const int last = gslot0 + op_slots - 1;
  if (r & 3) {
    ok = SG_WAIT_LDS(c, (int)(sg_ld(c.fill) - (unsigned)last) > 0, 0x31u);
    if (ok && c.lane == 0) sg_st(c.done + c.cw, (unsigned)(last + 1));
  }
  return ok;
}

// LayerNorm statistics of the fp32 row in LDS (every consumer wave computes them for the whole row, no exchange) and the
// CENTRED row rounded to bf16 into xb (each wave writes its quarter).  mean_c / rstd as rows_gemm_k's LN prologue defines
// them: statistics of the rounded centred values.
__device__ __forceinline__ void sg_ln_center(SgCtx& c, const float* hb, int d, char* xb, float& mean_c, float& rstd) {
  float t = 0.f;
  for (int i = c.lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(hb + i);
    t += (v.x + v.y) + (v.z + v.w);
  }
  const float mu = wave_sum(t) / (float)d;
  float s1 = 0.f, s2 = 0.f;
  const int q0 = c.cw * (d >> 2), q1 = q0 + (d >> 2);
  for (int i = c.lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(hb + i);
    uint2 u;
    u.x = pack_bf16x2(v.x - mu, v.y - mu);
    u.y = pack_bf16x2(v.z - mu, v.w - mu);
    const float r0 = __uint_as_float(u.x << 16), r1 = __uint_as_float(u.x & 0xffff0000u);
    const float r2 = __uint_as_float(u.y << 16), r3 = __uint_as_float(u.y & 0xffff0000u);
    s1 += (r0 + r1) + (r2 + r3);
    s2 += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
    if (i >= q0 && i < q1) *reinterpret_cast<uint2*>(xb + (size_t)i * 2) = u;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  mean_c = s1 / (float)d;
  const float var = fmaxf(s2 / (float)d - mean_c * mean_c, 0.f);
  rstd = 1.0f / sqrtf(var + 1e-5f);      // eps 1e-5 (transformer.py:30)
}

__device__ __forceinline__ void sg_unpack8(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ float sg_bf16_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// ---------------------------------------------------------------- the kernel
__global__ __launch_bounds__(SG_THREADS) void stream_step_k(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cu = blockIdx.x;
  if (*a.n_active == 0) return;                       // a replayed step after the sequence retired: no-op for the whole grid
  char* misc = smem + SG_LDS_MISC;
  unsigned* fill = reinterpret_cast<unsigned*>(misc + SG_M_FILL);
  unsigned* abort_w = reinterpret_cast<unsigned*>(misc + SG_M_ABORT);
  unsigned* cbar = reinterpret_cast<unsigned*>(misc + SG_M_CBAR);
  unsigned* done = reinterpret_cast<unsigned*>(misc + SG_M_DONE);
  if (tid < 64) reinterpret_cast<unsigned*>(misc)[tid] = 0u;      // fill, abort, cbar, done[]
  __syncthreads();                                                  // the only block-wide barrier: before the roles split
  const int total_slots = a.L * a.spl;

  if (wave == 0) {
    // ------------------------------------------------------------ LOADER
    // LDS flag traffic of this wave goes through inline asm: the compiler would otherwise wait for every pending LDS-DMA
    // (vmcnt(0)) before an LDS read it cannot prove disjoint from the DMA's destination.
    const unsigned done_addr = sg_lds_addr(done), fill_addr = sg_lds_addr(fill), abort_addr = sg_lds_addr(abort_w);
    const char* src0 = reinterpret_cast<const char*>(a.Ws) + (size_t)lane * 16;
    for (int g = 0; g < total_slots; ++g) {
      if (g >= SG_RING) {          // the slot's previous content must have been released by all four consumers
        const unsigned need = (unsigned)(g - SG_RING + 1);
        bool ok = false;
        for (int it = 0; it < SG_SPIN_LDS; ++it) {
          uint4 dv;
          unsigned ab;
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(dv), "=&v"(ab) : "v"(done_addr), "v"(abort_addr) : "memory");
          if (ab) return;
          const unsigned mn = min(min(dv.x, dv.y), min(dv.z, dv.w));
          if ((int)(mn - need) >= 0) { ok = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) {
          asm volatile("ds_write_b32 %0, %1" :: "v"(abort_addr), "v"(1u) : "memory");
          __hip_atomic_store(a.ctl + 1, 0x40u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          return;
        }
      }
      const int layer = g / a.spl, s = g - layer * a.spl;
      const char* src = src0 + (((size_t)layer * a.G + cu) * a.spl + s) * (size_t)SG_SLOT;
      char* dst = smem + (size_t)(g & (SG_RING - 1)) * SG_SLOT;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2 /* nt */);
      // three slots in flight at most: once <= 32 loads are outstanding, slot g - 2 has landed (a wave's loads land in order)
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      if (g >= 2) asm volatile("ds_write_b32 %0, %1" :: "v"(fill_addr), "v"((unsigned)(g - 1)) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(fill_addr), "v"((unsigned)total_slots) : "memory");
    return;
  }

  // -------------------------------------------------------------- CONSUMERS
  SgCtx c;
  c.smem = smem; c.fill = fill; c.abort_w = abort_w; c.cbar = cbar; c.done = done; c.ctl = a.ctl;
  c.bar_target = 0; c.cw = wave - 1; c.lane = lane; c.tid_c = (wave - 1) * 64 + lane;
  c.tag = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  char* xb = smem + SG_LDS_XB;
  float* hb = reinterpret_cast<float*>(smem + SG_LDS_HB);
  float* qbuf = reinterpret_cast<float*>(misc + SG_M_Q);
  float* knew = reinterpret_cast<float*>(misc + SG_M_KNEW);
  float* vnew = reinterpret_cast<float*>(misc + SG_M_VNEW);
  float* wml = reinterpret_cast<float*>(misc + SG_M_WML);
  float* wo = reinterpret_cast<float*>(misc + SG_M_WO);
  const int d = a.d, hd = a.hd;
  const int pos = a.row_pos[0], seq = a.row_seq[0];
  if (pos < 0 || pos >= a.S_max) return;              // inactive row (uniform over the grid)
  const int S = pos + 1;
  const int hh = cu / a.NS, sp = cu - hh * a.NS;       // attention role: head, split
  const int chunk = (S + a.NS - 1) / a.NS;
  const int p0 = sp * chunk, p1 = min(S, p0 + chunk);
  const bool owner = (p0 < S) && (p1 == S);           // this split holds the position written in this step
  const int LPR = hd / 8, PPW = 64 / LPR;             // lanes per cached row (bf16), positions per wave per visit
  const int sub = lane / LPR, li = lane - sub * LPR;

  // layer 0 input: the new token's embedding row
  for (int i = c.tid_c; i < d; i += 256) hb[i] = a.h_in[i];
  if (!sg_cbarrier(c)) return;

  for (int l = 0; l < a.L; ++l) {
    const StreamLayerDev ly = a.layers[l];
    unsigned long long* gl = a.gran + (size_t)l * a.gran_layer_stride;
    const int g0 = l * a.spl;
    // ================================================= A. LayerNorm 1 + QKV
    {
      float mean_c, rstd;
      sg_ln_center(c, hb, d, xb, mean_c, rstd);
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 0) * 4 * d + i] = hb[i];
      if (!sg_cbarrier(c)) return;
      float acc[6];
      if (!sg_consume<6>(c, g0, a.rpc, a.sq, xb, acc)) return;
      if (lane < 6) {
        float y = acc[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) y = (lane == j) ? acc[j] : y;
        const int n = 24 * cu + 6 * c.cw + lane;
        y = rstd * (y - mean_c * ly.wg_qkv[n]) + ly.b_qkv[n];
        if (n < d) {
          sg_publish(gl + sg_off_q(a) + n, c.tag, __float_as_uint(y));
        } else {
          const int which = (n - d) / d;
          const int cc = (n - d) - which * d;
          const int h = cc / hd, e = cc - h * hd;
          bf16_t* base = reinterpret_cast<bf16_t*>(which ? ly.vc : ly.kc) + (size_t)seq * a.cache_seq_stride + ((size_t)h * a.S_max + pos) * hd + e;
          base->u = f32_to_bf16(y);
          sg_publish(gl + sg_off_kv(a) + (n - d), c.tag, __float_as_uint(y));
        }
      }
    }
    // ================================================= B. attention: (head hh, split sp) of the cached positions
    {
      bool ok = true;
      if (c.tid_c < hd) {
        // q of the head; on the owner split also the k and v of the position written in this step
        for (int it = 0; it < SG_SPIN_GLOBAL; ++it) {
          const unsigned long long vq = __hip_atomic_load(gl + sg_off_q(a) + hh * hd + c.tid_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned long long vk = (unsigned long long)c.tag << 32, vv = vk;
          if (owner) {
            vk = __hip_atomic_load(gl + sg_off_kv(a) + hh * hd + c.tid_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vv = __hip_atomic_load(gl + sg_off_kv(a) + d + hh * hd + c.tid_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          ok = (unsigned)(vq >> 32) == c.tag && (unsigned)(vk >> 32) == c.tag && (unsigned)(vv >> 32) == c.tag;
          if (ok) {
            qbuf[c.tid_c] = __uint_as_float((unsigned)vq) * a.scale;
            knew[c.tid_c] = sg_bf16_round(__uint_as_float((unsigned)vk));      // what a later step will read back from the cache
            vnew[c.tid_c] = sg_bf16_round(__uint_as_float((unsigned)vv));
            break;
          }
          if ((it & 15) == 15 && sg_ld(c.abort_w)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && !sg_ld(c.abort_w)) sg_give_up(c, 0x21u);
      }
      if (!sg_cbarrier(c)) return;
      if (sg_ld(c.abort_w)) return;
      float q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = qbuf[li * 8 + j];
      float m = -INFINITY, lsum = 0.f, o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
      const int p1c = min(p1, S - 1);                   // cached positions of this split: [p0, p1c); position S-1 comes as granules
      const bf16_t* kb = reinterpret_cast<const bf16_t*>(ly.kc) + (size_t)seq * a.cache_seq_stride + (size_t)hh * a.S_max * hd + li * 8;
      const bf16_t* vb = reinterpret_cast<const bf16_t*>(ly.vc) + (size_t)seq * a.cache_seq_stride + (size_t)hh * a.S_max * hd + li * 8;
      auto visit = [&](float sc, const float* vf, bool on) {
        if (LPR == 4) sc = quad_sum(sc);
        else if (LPR == 8) sc = half_row_sum(sc);
        else { sc = row_sum(sc); if (LPR == 32) sc += __shfl_xor(sc, 16, 64); }
        if (on) {
          const float mn = fmaxf(m, sc);
          const float corr = expf(m - mn);              // m = -inf on the first visit -> 0
          const float pe = expf(sc - mn);
          lsum = lsum * corr + pe;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = o[j] * corr + pe * vf[j];
          m = mn;
        }
      };
      for (int pb = p0; pb < p1c; pb += 4 * SG_CONS * PPW) {
        uint4 ku[4], vu[4];
        int pp[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          pp[t] = pb + (t * SG_CONS + c.cw) * PPW + sub;
          const size_t pc = (size_t)max(min(pp[t], p1c - 1), 0);
          ku[t] = *reinterpret_cast<const uint4*>(kb + pc * hd);
          vu[t] = *reinterpret_cast<const uint4*>(vb + pc * hd);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float kf[8], vf[8];
          sg_unpack8(ku[t], kf);
          sg_unpack8(vu[t], vf);
          float sc = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) sc += q[j] * kf[j];
          visit(sc, vf, pp[t] < p1c);
        }
      }
      if (owner) {                                       // the position of this step: first position group of consumer 0
        float kf[8], vf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { kf[j] = knew[li * 8 + j]; vf[j] = vnew[li * 8 + j]; }
        float sc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sc += q[j] * kf[j];
        visit(sc, vf, c.cw == 0 && sub == 0);
      }
      // merge the position groups of the wave, then the four consumers
      for (int off = LPR; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(lsum, off, 64);
        const float mn = fmaxf(m, m2);
        const float c1 = (m == -INFINITY) ? 0.f : expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        lsum = lsum * c1 + l2 * c2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float o2 = __shfl_xor(o[j], off, 64); o[j] = o[j] * c1 + o2 * c2; }
        m = mn;
      }
      if (lane < LPR) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wo[c.cw * 128 + lane * 8 + j] = o[j];
        if (lane == 0) { wml[c.cw * 2] = m; wml[c.cw * 2 + 1] = lsum; }
      }
      if (!sg_cbarrier(c)) return;
      unsigned long long* pg = gl + sg_off_p(a) + (size_t)cu * (hd + 2);
      if (c.tid_c < hd + 2) {
        float M = wml[0];
#pragma unroll
        for (int w = 1; w < SG_CONS; ++w) M = fmaxf(M, wml[w * 2]);
        float Ls = 0.f, O = 0.f;
#pragma unroll
        for (int w = 0; w < SG_CONS; ++w) {
          const float cf = (wml[w * 2] == -INFINITY) ? 0.f : expf(wml[w * 2] - M);
          Ls += cf * wml[w * 2 + 1];
          if (c.tid_c < hd) O += cf * wo[w * 128 + c.tid_c];
        }
        const float val = (c.tid_c < hd) ? O : (c.tid_c == hd ? M : Ls);
        sg_publish(pg + c.tid_c, c.tag, __float_as_uint(val));
      }
      // ---- head leader (split 0): merge the NS partials of the head, publish the normalised head output
      if (sp == 0) {
        float* pt = reinterpret_cast<float*>(xb);        // [NS][hd + 2]
        const int n = a.NS * (hd + 2);
        if (!sg_gather(c, gl + sg_off_p(a) + (size_t)hh * a.NS * (hd + 2), n, [&](int idx, unsigned bits) { pt[idx] = __uint_as_float(bits); })) return;
        if (!sg_cbarrier(c)) return;
        if (c.tid_c < hd) {
          float M = -INFINITY;
          for (int s2 = 0; s2 < a.NS; ++s2) M = fmaxf(M, pt[s2 * (hd + 2) + hd]);
          float Ls = 0.f, O = 0.f;
          for (int s2 = 0; s2 < a.NS; ++s2) {
            const float ms = pt[s2 * (hd + 2) + hd];
            const float cf = (ms == -INFINITY) ? 0.f : expf(ms - M);
            Ls += cf * pt[s2 * (hd + 2) + hd + 1];
            O += cf * pt[s2 * (hd + 2) + c.tid_c];
          }
          sg_publish(gl + sg_off_o(a) + hh * hd + c.tid_c, c.tag, __float_as_uint((Ls > 0.f) ? O / Ls : 0.f));
        }
        if (!sg_cbarrier(c)) return;                     // pt (= xb) is rewritten below
      }
    }
    // ================================================= C. out-projection + residual
    {
      uint16_t* x16 = reinterpret_cast<uint16_t*>(xb);
      if (!sg_gather(c, gl + sg_off_o(a), d, [&](int idx, unsigned bits) { x16[idx] = f32_to_bf16(__uint_as_float(bits)); })) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 1) * 4 * d + i] = bf16_to_f32(x16[i]);
      float acc[2];
      if (!sg_consume<2>(c, g0 + a.sq, a.rpc, a.so, xb, acc)) return;
      if (lane < 2) {
        const int n = 8 * cu + 2 * c.cw + lane;
        const float y = (lane == 0 ? acc[0] : acc[1]) + ly.b_o[n];
        sg_publish(gl + sg_off_h2(a) + n, c.tag, __float_as_uint(hb[n] + y));
      }
    }
    // ================================================= D. LayerNorm 2 + FFN up-projection + ReLU
    {
      if (!sg_cbarrier(c)) return;                       // every consumer has read hb[n] above before it is overwritten
      if (!sg_gather(c, gl + sg_off_h2(a), d, [&](int idx, unsigned bits) { hb[idx] = __uint_as_float(bits); })) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 2) * 4 * d + i] = hb[i];
      float mean_c, rstd;
      sg_ln_center(c, hb, d, xb, mean_c, rstd);
      if (!sg_cbarrier(c)) return;
      float acc[8];
      if (!sg_consume<8>(c, g0 + a.sq + a.so, a.rpc, a.s1, xb, acc)) return;
      // lanes 0..3 publish the wave's 8 activations as 4 granules of two bf16
      if (lane < 4) {
        float y0 = acc[0], y1 = acc[1];
#pragma unroll
        for (int j = 1; j < 4; ++j) { y0 = (lane == j) ? acc[2 * j] : y0; y1 = (lane == j) ? acc[2 * j + 1] : y1; }
        const int n = 32 * cu + 8 * c.cw + 2 * lane;
        y0 = fmaxf(rstd * (y0 - mean_c * ly.wg_1[n]) + ly.b_1[n], 0.f);
        y1 = fmaxf(rstd * (y1 - mean_c * ly.wg_1[n + 1]) + ly.b_1[n + 1], 0.f);
        sg_publish(gl + sg_off_a(a) + (n >> 1), c.tag, pack_bf16x2(y0, y1));
      }
    }
    // ================================================= E. FFN down-projection + residual
    {
      if (!sg_cbarrier(c)) return;                       // xb (LN2 output) fully consumed by every wave before it is rewritten
      unsigned* x32 = reinterpret_cast<unsigned*>(xb);
      if (!sg_gather(c, gl + sg_off_a(a), 2 * d, [&](int idx, unsigned bits) { x32[idx] = bits; })) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < 4 * d; i += 256) a.dbg[((size_t)l * 5 + 3) * 4 * d + i] = bf16_to_f32(reinterpret_cast<uint16_t*>(xb)[i]);
      float acc[2];
      if (!sg_consume<2>(c, g0 + a.sq + a.so + a.s1, 4 * a.rpc, a.s2, xb, acc)) return;
      const bool last = (l == a.L - 1);
      if (lane < 2) {
        const int n = 8 * cu + 2 * c.cw + lane;
        const float y = hb[n] + (lane == 0 ? acc[0] : acc[1]) + ly.b_2[n];
        if (last) a.h_out[n] = y;
        else sg_publish(a.gran + (size_t)(l + 1) * a.gran_layer_stride + sg_off_h(a) + n, c.tag, __float_as_uint(y));
        if (a.dbg) a.dbg[((size_t)l * 5 + 4) * 4 * d + n] = y;
      }
      if (!last) {
        if (!sg_cbarrier(c)) return;                     // hb[n] read above, xb consumed
        unsigned long long* gn = a.gran + (size_t)(l + 1) * a.gran_layer_stride;
        if (!sg_gather(c, gn + sg_off_h(a), d, [&](int idx, unsigned bits) { hb[idx] = __uint_as_float(bits); })) return;
        if (!sg_cbarrier(c)) return;
      }
    }
  }
  // the step is over for this workgroup; workgroup 0 moves the epoch on (every workgroup read it long ago: this point
  // lies behind six all-to-all edges per layer)
  if (cu == 0 && c.cw == 0 && lane == 0) {
    unsigned nt = c.tag + 1u;
    if (nt == 0u) nt = 1u;
    __hip_atomic_store(a.ctl, nt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------- weight packing into the stream layout
// dst unit (16 B) index within a layer: (((cu * spl + slot) * 4 + cw) * 4 + row4) * 64 + lane.  The op of a slot and the
// row inside the op follow from the slot index; row r of an op for consumer cw = channel j = r / rows_per_ch, k-chunk
// group i = r % rows_per_ch; the unit holds W[n][8 (64 i + lane) .. + 8) (scaled by colscale[k] for LN-folded matrices).
struct StreamPackArgs {
  const float *Wqkv, *Wo, *W1, *W2;    // fp32 [3d][d], [d][d], [4d][d], [d][4d]
  const float *g1, *g2;                // LayerNorm gammas folded into Wqkv / W1 (columns)
  uint4* dst;                          // this layer's [G][spl][1024]
  int d, G, rpc, sq, so, s1, s2, spl;
};
__global__ __launch_bounds__(256) void stream_pack_k(const StreamPackArgs a) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)a.G * a.spl * 1024;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const int row4 = (int)((idx >> 6) & 3);
  const int cw = (int)((idx >> 8) & 3);
  const long sl = idx >> 10;
  const int slot = (int)(sl % a.spl), cu = (int)(sl / a.spl);
  const float* W; const float* cs = nullptr;
  int K, nc, rpch, s_in_op, nbase;
  if (slot < a.sq) { W = a.Wqkv; cs = a.g1; K = a.d; nc = 6; rpch = a.rpc; s_in_op = slot; nbase = 24 * cu + 6 * cw; }
  else if (slot < a.sq + a.so) { W = a.Wo; K = a.d; nc = 2; rpch = a.rpc; s_in_op = slot - a.sq; nbase = 8 * cu + 2 * cw; }
  else if (slot < a.sq + a.so + a.s1) { W = a.W1; cs = a.g2; K = a.d; nc = 8; rpch = a.rpc; s_in_op = slot - a.sq - a.so; nbase = 32 * cu + 8 * cw; }
  else { W = a.W2; K = 4 * a.d; nc = 2; rpch = 4 * a.rpc; s_in_op = slot - a.sq - a.so - a.s1; nbase = 8 * cu + 2 * cw; }
  const int r = s_in_op * 4 + row4;
  const int j = r / rpch, i = r - j * rpch;
  uint4 out = make_uint4(0u, 0u, 0u, 0u);
  if (j < nc) {
    const int n = nbase + j;
    const int k = 8 * (64 * i + lane);
    const float* src = W + (long)n * K + k;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[e] * (cs ? cs[k + e] : 1.0f);
    out.x = pack_bf16x2(v[0], v[1]); out.y = pack_bf16x2(v[2], v[3]);
    out.z = pack_bf16x2(v[4], v[5]); out.w = pack_bf16x2(v[6], v[7]);
  }
  a.dst[idx] = out;
}

hipError_t vc_stream_pack_layer(const float* Wqkv, const float* Wo, const float* W1, const float* W2, const float* g1,
                                const float* g2, void* dst, int d, int G, hipStream_t s) {
  StreamPackArgs p;
  p.Wqkv = Wqkv; p.Wo = Wo; p.W1 = W1; p.W2 = W2; p.g1 = g1; p.g2 = g2; p.dst = reinterpret_cast<uint4*>(dst);
  p.d = d; p.G = G; p.rpc = d / 512;
  p.sq = (6 * p.rpc + 3) / 4; p.so = (2 * p.rpc + 3) / 4; p.s1 = 2 * p.rpc; p.s2 = 2 * p.rpc;
  p.spl = p.sq + p.so + p.s1 + p.s2;
  const long total = (long)G * p.spl * 1024;
  hipLaunchKernelGGL(stream_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  return hipGetLastError();
}

size_t vc_stream_layer_bytes(int d, int G) {
  const int rpc = d / 512;
  const int spl = (6 * rpc + 3) / 4 + (2 * rpc + 3) / 4 + 4 * rpc;
  return (size_t)G * spl * SG_SLOT;
}

hipError_t vc_stream_launch(const StreamArgs& a, hipStream_t s) {
  static bool granted[16] = {false};
  int dev = 0;
  if (hipError_t ge = hipGetDevice(&dev); ge != hipSuccess) return ge;
  if (dev >= 0 && dev < 16 && !granted[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_step_k), hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_TOTAL);
    if (e != hipSuccess) return e;
    granted[dev] = true;
  }
  ++vc_launch_counts[VC_LC_PERSIST];
  hipLaunchKernelGGL(stream_step_k, dim3(a.G), dim3(SG_THREADS), SG_LDS_TOTAL, s, a);
  return hipGetLastError();
}
