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
#define SG_LDS_XB (SG_RING * SG_SLOT)  // FFN-down input as bf16 (4d elements = 16 KB); during attention the head leader's partial table
#define SG_LDS_HB (SG_LDS_XB + 16384)  // residual row, fp32 [d]
#define SG_LDS_MISC (SG_LDS_HB + 8192)
#define SG_LDS_TOTAL (SG_LDS_MISC + 8192)       // 163 840 = all of a CU's LDS
#define SG_SPIN_LDS (1 << 21)          // bound of an LDS wait (~0.1 s)
#define SG_SPIN_GLOBAL (1 << 17)       // bound of a granule sweep (~0.1 s)
#ifndef SG_EDGE_DELAY
#define SG_EDGE_DELAY 40               // s_sleep units (64 clocks) before the first sweep of an all-to-all edge
#endif

// misc region (byte offsets from SG_LDS_MISC)
#define SG_M_FILL 0                    // u32: slots landed so far (loader)
#define SG_M_ABORT 4                   // u32: set by any wave that gives up
#define SG_M_CBAR 8                    // u32: consumer barrier counter
#define SG_M_DONE 16                   // u32[4]: slots released, per consumer
#define SG_M_GATH 32                   // u32: consumers are sweeping granules
#define SG_M_Q 128                     // float[128]: q of this CU's head
#define SG_M_KNEW 640                  // float[128]
#define SG_M_VNEW 1152                 // float[128]
#define SG_M_WML 1664                  // float[4][2]: per-wave (max, sum)
#define SG_M_WO 1696                   // float[4][128]: per-wave un-normalised output
#define SG_M_XS 4096                   // 4 KB: input of the K = d ops (bf16 [d]); the 16 KB buffer at SG_LDS_XB holds the FFN-down input
#define SG_M_END 8192

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
  unsigned* gath;                      // LDS word: 1 while the consumers sweep granules (the loader keeps one fill in flight then)
  unsigned* ctl;
  unsigned bar_target;                 // consumer barrier: next value of cbar to wait for
  unsigned tag;
  int cw, lane, tid_c;                 // consumer index 0..3, lane, consumer-thread index 0..255
  int sweeps;                          // granule sweeps so far (diagnostics)
};

__device__ __forceinline__ void sg_give_up(unsigned* abort_w, unsigned* ctl, unsigned code) {
  sg_st(abort_w, 1u);
  __hip_atomic_store(ctl + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Slow path of every LDS wait (kept out of line: the fast path is one LDS read at the call site): spins until
// (int)(*p - target) >= 0.  true = condition met; false = the workgroup is aborting (this wait gave up, or another did).
__device__ __attribute__((noinline)) bool sg_spin_ge(const unsigned* p, unsigned target, unsigned* abort_w, unsigned* ctl, unsigned code) {
  for (int it = 0; it < SG_SPIN_LDS; ++it) {
    if ((int)(sg_ld(p) - target) >= 0) return true;
    if ((it & 63) == 63) {
      if (sg_ld(abort_w)) return false;
      if ((it & 4095) == 4095 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sg_st(abort_w, 1u); return false; }
    }
    __builtin_amdgcn_s_sleep(1);
  }
  sg_give_up(abort_w, ctl, code);
  return false;
}
__device__ __forceinline__ bool sg_wait_ge(SgCtx& c, const unsigned* p, unsigned target, unsigned code) {
  if ((int)(sg_ld(p) - target) >= 0) return true;
  return sg_spin_ge(p, target, c.abort_w, c.ctl, code);
}

// consumer-only barrier (4 waves): LDS writes before it are visible to every consumer after it
__device__ __forceinline__ bool sg_cbarrier(SgCtx& c) {
  // only LDS traffic is ordered here (a workgroup-scope fence would also drain the wave's global stores and loads:
  // granules in flight, prefetched K/V rows): the wave's LDS operations execute in order, so its writes are in LDS
  // once the counter it bumps afterwards is
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  c.bar_target += SG_CONS;
  if (c.lane == 0) __hip_atomic_fetch_add(c.cbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const bool ok = sg_wait_ge(c, c.cbar, c.bar_target, 0x10u);
  asm volatile("" ::: "memory");
  return ok;
}

__device__ __forceinline__ void sg_publish(unsigned long long* g, unsigned tag, unsigned bits) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sweep n granules (n <= 4096) into LDS; every consumer lane takes idx = tid_c + 256 p.  MODE 0: the 32 data bits as they
// are (fp32 values, or two bf16) into dst32[idx]; MODE 1: an fp32 value rounded to bf16 into dst16[idx].
// Returns false when the wait gave up.
template <int MODE>
__device__ __attribute__((noinline)) bool sg_gather_fn(const unsigned long long* g, int n, unsigned tag, int tid_c, void* dst,
                                                      unsigned* abort_w, unsigned* ctl) {
#ifdef SG_DIAG_GATHER_DIV      // timing experiment only (wrong results): sweep a fraction of the granules
  n = (n >= 1024) ? n / SG_DIAG_GATHER_DIV : n;
#endif
  const int per = (n + 255) >> 8;
  for (int it = 0; it < SG_SPIN_GLOBAL; ++it) {
    // every sweep asks for everything, unconditionally and back to back (16 loads in flight per lane): a sweep that re-asks
    // only for the granules still missing was measured slower (+1.4 us per edge) - its loads sit behind per-lane branches
    unsigned long long v[16];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int idx = tid_c + 256 * p;
      v[p] = (p < per && idx < n) ? __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : ((unsigned long long)tag << 32);
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) ok = ok && ((unsigned)(v[p] >> 32) == tag);
    if (ok) {
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int idx = tid_c + 256 * p;
        if (p < per && idx < n) {
          if (MODE == 0) reinterpret_cast<unsigned*>(dst)[idx] = (unsigned)v[p];
          else reinterpret_cast<uint16_t*>(dst)[idx] = f32_to_bf16(__uint_as_float((unsigned)v[p]));
        }
      }
      return true;
    }
    if ((it & 15) == 15 && sg_ld(abort_w)) return false;
    if ((it & 1023) == 1023 && __hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { sg_st(abort_w, 1u); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  sg_give_up(abort_w, ctl, 0x20u);
  return false;
}
// n granules, `share` consecutive ones per producer
template <int MODE>
__device__ __forceinline__ bool sg_gather(SgCtx& c, const unsigned long long* g, int n, void* dst, int share = 0) {
  // share > 0: an all-to-all edge.  A full sweep moves n x 8 B per CU through the fabric (8 MB over the chip for the
  // activations) - the fabric the weight stream uses - and cannot succeed before the slowest producer's stores are
  // visible (~1 us after its last dot product, producers finish within ~1.5 us of each other): the first sweep is held back.
  if (share > 0) __builtin_amdgcn_s_sleep(SG_EDGE_DELAY);
  return sg_gather_fn<MODE>(g, n, c.tag, c.tid_c, dst, c.abort_w, c.ctl);
}

__device__ __forceinline__ float sg_dot8(const uint4& w, const uint4& x, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.x), __builtin_bit_cast(sg_bf16x2, x.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.y), __builtin_bit_cast(sg_bf16x2, x.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.z), __builtin_bit_cast(sg_bf16x2, x.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sg_bf16x2, w.w), __builtin_bit_cast(sg_bf16x2, x.w), acc, false);
  return acc;
}

// One op of the stream for this consumer wave: NC channels of ROWS (a power of two) 1 KB rows each, starting at global
// slot gslot0.  Returns, in lane j (j < NC), sum_k W[channel j][k] * X[k]; X = bf16 vector in LDS at xb, row i of a
// channel multiplies chunk (64 i + lane).  A slot (4 rows) is read with four ds_read_b128 issued together (plus the four
// X chunks it meets), the next slot's reads are issued before this slot's dot products (two register sets), and a slot is
// released to the loader as soon as its dot products are through - their operands ARE the reads, and the LDS executes a
// wave's operations in order.  ok = false when a wait gave up.
template <int NC, int ROWS>
__device__ __forceinline__ float sg_consume(SgCtx& c, int gslot0, const char* xb, bool& ok) {
  constexpr int R = NC * ROWS;                       // rows of this wave in the op
  constexpr int SL = (R + 3) / 4;                    // slots of the op
  static_assert((ROWS & (ROWS - 1)) == 0, "rows per channel: a power of two");
  const char* ring = c.smem + c.cw * 4096 + c.lane * 16;
  const char* xl = xb + (size_t)c.lane * 16;
  float y = 0.f, sacc = 0.f;
  unsigned filled = sg_ld(c.fill);
  uint4 wa[4], wb[4], xa[4], xq[4];
  auto fetch = [&](int sl, uint4 (&w)[4], uint4 (&x)[4]) -> bool {
    const int slot = gslot0 + sl;
    if ((int)(filled - (unsigned)slot) <= 0) {        // (one LDS read covers every slot the loader is ahead by)
      if (!sg_wait_ge(c, c.fill, (unsigned)slot + 1u, 0x30u)) return false;
      filled = sg_ld(c.fill);
    }
    const char* wp = ring + (size_t)(slot & (SG_RING - 1)) * SG_SLOT;
    const char* xp = xl + (size_t)((sl * 4) & (ROWS - 1)) * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w[q] = *reinterpret_cast<const uint4*>(wp + q * 1024);
      x[q] = *reinterpret_cast<const uint4*>(xp + (size_t)(q & (ROWS - 1)) * 1024);
    }
    return true;
  };
  auto compute = [&](int sl, const uint4 (&w)[4], const uint4 (&x)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = sl * 4 + q;
      if (r < R) {
        sacc = sg_dot8(w[q], x[q], sacc);
        if ((r & (ROWS - 1)) == ROWS - 1) {           // last row of channel r / ROWS: its sum goes to lane (r / ROWS)
          const float t = wave_sum(sacc);
          y = (c.lane == r / ROWS) ? t : y;
          sacc = 0.f;
        }
      }
    }
    asm volatile("" ::: "memory");                   // the release below stays behind the reads it releases (compiler order)
    if (c.lane == 0) sg_st(c.done + c.cw, (unsigned)(gslot0 + sl + 1));
  };
  ok = fetch(0, wa, xa);
  for (int sl = 0; sl < SL && ok; sl += 2) {
    if (sl + 1 < SL) ok = fetch(sl + 1, wb, xq);
    if (!ok) break;
    compute(sl, wa, xa);
    if (sl + 1 >= SL) break;
    if (sl + 2 < SL) ok = fetch(sl + 2, wa, xa);
    if (!ok) break;
    compute(sl + 1, wb, xq);
  }
  return y;
}

// LayerNorm statistics of the fp32 row in LDS (every consumer wave computes them for the whole row, no exchange) and the
// CENTRED row rounded to bf16 into xb.  One pass over LDS: the row lives in registers
// (NV float4 per lane).  mean_c / rstd as rows_gemm_k's LN prologue defines them: statistics of the rounded centred values.
template <int NV>
__device__ __forceinline__ void sg_ln_center(SgCtx& c, const float* hb, char* xb, float& mean_c, float& rstd) {
  constexpr int d = NV * 256;
  const float* hp = hb + c.lane * 4;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 v = *reinterpret_cast<const float4*>(hp + i * 256);
    t += (v.x + v.y) + (v.z + v.w);
  }
  const float mu = wave_sum(t) * (1.0f / (float)d);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {          // (read again rather than kept: NV float4 per lane would be live across the sum)
    const float4 v = *reinterpret_cast<const float4*>(hp + i * 256);
    uint2 u;
    u.x = pack_bf16x2(v.x - mu, v.y - mu);
    u.y = pack_bf16x2(v.z - mu, v.w - mu);
    const float r0 = __uint_as_float(u.x << 16), r1 = __uint_as_float(u.x & 0xffff0000u);
    const float r2 = __uint_as_float(u.y << 16), r3 = __uint_as_float(u.y & 0xffff0000u);
    s1 += (r0 + r1) + (r2 + r3);
    s2 += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
    // EVERY consumer stores the whole row (identical values): a wave reads back what it wrote itself, no barrier in between
    *reinterpret_cast<uint2*>(xb + (size_t)(i * 256 + c.lane * 4) * 2) = u;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  mean_c = s1 * (1.0f / (float)d);
  const float var = fmaxf(s2 * (1.0f / (float)d) - mean_c * mean_c, 0.f);
  rstd = 1.0f / sqrtf(var + 1e-5f);      // eps 1e-5 (transformer.py:30)
}

__device__ __forceinline__ void sg_unpack8(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ float sg_bf16_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// phase stamps (100 MHz wall clock) of consumer 0 of three workgroups (first, second, last) when a.ts is set
#define SG_TS(i_)                                                                                \
  do {                                                                                           \
    if (a.ts && ts_sel >= 0 && c.cw == 0 && lane == 0) a.ts[((size_t)ts_sel * a.L + l) * 16 + (i_)] = (long long)wall_clock64(); \
  } while (0)

// ---------------------------------------------------------------- the kernel (RPC = d / 512: 1 KB rows per K = d channel)
template <int RPC>
__global__ __launch_bounds__(SG_THREADS) void stream_step_k(const StreamArgs a) {
  constexpr int d = 512 * RPC;
  constexpr int SQ = (6 * RPC + 3) / 4, SO = (2 * RPC + 3) / 4, S1 = 2 * RPC, S2 = 2 * RPC, SPL = SQ + SO + S1 + S2;
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
  const int total_slots = a.L * SPL;

  if (wave == 0) {
    // ------------------------------------------------------------ LOADER
    // LDS flag traffic of this wave goes through inline asm: the compiler would otherwise wait for every pending LDS-DMA
    // (vmcnt(0)) before an LDS read it cannot prove disjoint from the DMA's destination.
    const unsigned done_addr = sg_lds_addr(done), fill_addr = sg_lds_addr(fill), abort_addr = sg_lds_addr(abort_w);
    const char* src0 = reinterpret_cast<const char*>(a.Ws) + (size_t)lane * 16;
    int landed = 0;                // slots declared landed so far (what `fill` holds)
    for (int g = 0; g < total_slots; ++g) {
      if (g >= SG_RING) {          // the slot's previous content must have been released by all four consumers
        const unsigned need = (unsigned)(g - SG_RING + 1);
        bool ok = false;
        for (int it = 0; it < SG_SPIN_LDS; ++it) {
          if (it == 1) {           // the ring is full: nothing to issue, so everything issued is declared as soon as it has landed
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed = g;
            asm volatile("ds_write_b32 %0, %1" :: "v"(fill_addr), "v"((unsigned)landed) : "memory");
          }
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
      const int layer = g / SPL, sidx = g - layer * SPL;
      const char* src = src0 + (((size_t)layer * a.G + cu) * SPL + sidx) * (size_t)SG_SLOT;
      char* dst = smem + (size_t)(g & (SG_RING - 1)) * SG_SLOT;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2 /* nt */);
      // three slots in flight at most: once <= 32 loads are outstanding, slot g - 2 has landed (a wave's loads land in order).
      // While the consumers of this CU sweep granules the loader keeps ONE fill in flight: their polls share the CU's
      // memory pipeline with the DMA burst (MI355X_MICROARCH "gather-pass": 0.3-0.65 us with the own DMA quiet, 1.0-1.7 behind it)
      // (Keeping ONE fill in flight while the CU's consumers sweep granules - the guide's "gather-pass" remedy - was measured
      // here: the activation edge 6.1 -> 5.6 us, but the FFN-down weights then arrive late, its consume 1.7 -> 2.6 us and the
      // edge behind it 5.1 -> 6.0 us: a net loss of 1.3 us per layer, profiles/r03_stream_probe_*.log.)
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      landed = max(landed, g - 1);
      asm volatile("ds_write_b32 %0, %1" :: "v"(fill_addr), "v"((unsigned)landed) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(fill_addr), "v"((unsigned)total_slots) : "memory");
    return;
  }

  // -------------------------------------------------------------- CONSUMERS
  SgCtx c;
  c.smem = smem; c.fill = fill; c.abort_w = abort_w; c.cbar = cbar; c.done = done; c.ctl = a.ctl;
  c.gath = reinterpret_cast<unsigned*>(misc + SG_M_GATH);
  c.bar_target = 0; c.cw = wave - 1; c.lane = lane; c.tid_c = (wave - 1) * 64 + lane; c.sweeps = 0;
  c.tag = __hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  char* xb = smem + SG_LDS_XB;                         // FFN-down input; the head leader's partial table during attention
  char* xs = misc + SG_M_XS;                           // input of the K = d ops
  float* hb = reinterpret_cast<float*>(smem + SG_LDS_HB);
  float* qbuf = reinterpret_cast<float*>(misc + SG_M_Q);
  float* knew = reinterpret_cast<float*>(misc + SG_M_KNEW);
  float* vnew = reinterpret_cast<float*>(misc + SG_M_VNEW);
  float* wml = reinterpret_cast<float*>(misc + SG_M_WML);
  float* wo = reinterpret_cast<float*>(misc + SG_M_WO);
  const int hd = a.hd;
  const int pos = a.row_pos[0], seq = a.row_seq[0];
  if (pos < 0 || pos >= a.S_max) return;              // inactive row (uniform over the grid)
  const int S = pos + 1;
  const int hh = cu / a.NS, sp = cu - hh * a.NS;       // attention role: head, split
  const int chunk = (S + a.NS - 1) / a.NS;
  const int p0 = sp * chunk, p1 = min(S, p0 + chunk);
  const bool owner = (p0 < S) && (p1 == S);           // this split holds the position written in this step
  const int p1c = min(p1, S - 1);                     // cached positions of this split: [p0, p1c); position S-1 comes as granules
  const int LPR = hd / 8, PPW = 64 / LPR;             // lanes per cached row (bf16), positions per wave per visit
  const int sub = lane / LPR, li = lane - sub * LPR;
  const int ts_sel = (cu == 0) ? 0 : (cu == 1) ? 1 : (cu == a.G - 1) ? 2 : -1;
  // channels of this wave (the lanes that carry a channel; the others compute on a clamped copy)
  const int nq = 24 * cu + 6 * c.cw + min(lane, 5);           // QKV: lanes 0..5
  const int no = 8 * cu + 2 * c.cw + min(lane, 1);            // out-projection / FFN-down: lanes 0..1
  const int n1 = 32 * cu + 8 * c.cw + min(lane, 7);           // FFN-up: lanes 0..7

  // layer 0 input: the new token's embedding row
  for (int i = c.tid_c; i < d; i += 256) hb[i] = a.h_in[i];
  if (!sg_cbarrier(c)) return;

  // epilogue constants of a layer (and its cache pointers) are requested ONE LAYER AHEAD: a dependent global load at the
  // start or the end of an op would sit on the critical path of the edge behind it
  StreamLayerDev ly = a.layers[0];
  float c_wgq = ly.wg_qkv[nq], c_bq = ly.b_qkv[nq], c_bo = ly.b_o[no], c_b2 = ly.b_2[no], c_wg1 = ly.wg_1[n1], c_b1 = ly.b_1[n1];
  for (int l = 0; l < a.L; ++l) {
    SG_TS(0);
    unsigned long long* gl = a.gran + (size_t)l * a.gran_layer_stride;
    const int g0 = l * SPL;
    const StreamLayerDev lyn = a.layers[min(l + 1, a.L - 1)];
    const float n_wgq = lyn.wg_qkv[nq], n_bq = lyn.b_qkv[nq], n_bo = lyn.b_o[no], n_b2 = lyn.b_2[no], n_wg1 = lyn.wg_1[n1], n_b1 = lyn.b_1[n1];
    const float hres = hb[no];                           // this wave's own residual values (lanes 0..1), before hb is rewritten
    // ================================================= A. LayerNorm 1 + QKV
    {
      float mean_c, rstd;
      sg_ln_center<2 * RPC>(c, hb, xs, mean_c, rstd);
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 0) * 4 * d + i] = hb[i];
      SG_TS(1);
      bool okc;
      float y = sg_consume<6, RPC>(c, g0, xs, okc);
      if (!okc) return;
      SG_TS(2);
      if (lane < 6) {
        y = rstd * (y - mean_c * c_wgq) + c_bq;
        if (nq < d) {
          sg_publish(gl + sg_off_q(a) + nq, c.tag, __float_as_uint(y));
        } else {
          const int which = (nq - d) / d;
          const int cc = (nq - d) - which * d;
          const int h = cc / hd, e = cc - h * hd;
          bf16_t* base = reinterpret_cast<bf16_t*>(which ? ly.vc : ly.kc) + (size_t)seq * a.cache_seq_stride + ((size_t)h * a.S_max + pos) * hd + e;
          sg_publish(gl + sg_off_kv(a) + (nq - d), c.tag, __float_as_uint(y));
          base->u = f32_to_bf16(y);
        }
      }
    }
    // K and V of this split's cached positions do not depend on this step: requested before q exists
    const bf16_t* kb = reinterpret_cast<const bf16_t*>(ly.kc) + (size_t)seq * a.cache_seq_stride + (size_t)hh * a.S_max * hd + li * 8;
    const bf16_t* vb = reinterpret_cast<const bf16_t*>(ly.vc) + (size_t)seq * a.cache_seq_stride + (size_t)hh * a.S_max * hd + li * 8;
    uint4 ku[4], vu[4];
    int pp[4];
#define SG_KV_LOADS(pb_)                                                     \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                          \
      pp[t] = (pb_) + (t * SG_CONS + c.cw) * PPW + sub;                      \
      const size_t pc = (size_t)max(min(pp[t], p1c - 1), 0);                 \
      ku[t] = *reinterpret_cast<const uint4*>(kb + pc * hd);                 \
      vu[t] = *reinterpret_cast<const uint4*>(vb + pc * hd);                 \
    }
    // ================================================= B. attention: (head hh, split sp) of the cached positions
    {
      SG_KV_LOADS(p0);                                   // in flight while q crosses the chip
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
        if (!ok && !sg_ld(c.abort_w)) sg_give_up(c.abort_w, c.ctl, 0x21u);
      }
      if (!sg_cbarrier(c)) return;
      if (sg_ld(c.abort_w)) return;
      SG_TS(3);
      float q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = qbuf[li * 8 + j];
      float m = -INFINITY, lsum = 0.f, o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
      auto visit = [&](float sc, const float* vf, bool on) {
        if (LPR == 4) sc = quad_sum(sc);
        else if (LPR == 8) sc = half_row_sum(sc);
        else { sc = row_sum(sc); if (LPR == 32) sc += __shfl_xor(sc, 16, 64); }
        if (on) {
          const float mn = fmaxf(m, sc);
          const float corr = __expf(m - mn);            // m = -inf on the first visit -> 0
          const float pe = __expf(sc - mn);
          lsum = lsum * corr + pe;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = o[j] * corr + pe * vf[j];
          m = mn;
        }
      };
      for (int pb = p0;;) {
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
        pb += 4 * SG_CONS * PPW;
        if (pb >= p1c) break;
        SG_KV_LOADS(pb);
      }
#undef SG_KV_LOADS
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
        const float c1 = (m == -INFINITY) ? 0.f : __expf(m - mn), c2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
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
          const float cf = (wml[w * 2] == -INFINITY) ? 0.f : __expf(wml[w * 2] - M);
          Ls += cf * wml[w * 2 + 1];
          if (c.tid_c < hd) O += cf * wo[w * 128 + c.tid_c];
        }
        const float val = (c.tid_c < hd) ? O : (c.tid_c == hd ? M : Ls);
        sg_publish(pg + c.tid_c, c.tag, __float_as_uint(val));
      }
      SG_TS(4);
      // ---- head leader (split 0): merge the NS partials of the head, publish the normalised head output
      if (sp == 0) {
        float* pt = reinterpret_cast<float*>(xb);        // [NS][hd + 2]
        const int n = a.NS * (hd + 2);
        if (!sg_gather<0>(c, gl + sg_off_p(a) + (size_t)hh * a.NS * (hd + 2), n, pt)) return;
        if (!sg_cbarrier(c)) return;
        if (c.tid_c < hd) {
          float M = -INFINITY;
          for (int s2 = 0; s2 < a.NS; ++s2) M = fmaxf(M, pt[s2 * (hd + 2) + hd]);
          float Ls = 0.f, O = 0.f;
          for (int s2 = 0; s2 < a.NS; ++s2) {
            const float ms = pt[s2 * (hd + 2) + hd];
            const float cf = (ms == -INFINITY) ? 0.f : __expf(ms - M);
            Ls += cf * pt[s2 * (hd + 2) + hd + 1];
            O += cf * pt[s2 * (hd + 2) + c.tid_c];
          }
          sg_publish(gl + sg_off_o(a) + hh * hd + c.tid_c, c.tag, __float_as_uint((Ls > 0.f) ? O / Ls : 0.f));
        }
      }
      SG_TS(5);
    }
    // ================================================= C. out-projection + residual
    {
      uint16_t* x16 = reinterpret_cast<uint16_t*>(xs);
      if (!sg_gather<1>(c, gl + sg_off_o(a), d, x16, hd)) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 1) * 4 * d + i] = bf16_to_f32(x16[i]);
      SG_TS(6);
      bool okc;
      const float yo = sg_consume<2, RPC>(c, g0 + SQ, xs, okc);
      if (!okc) return;
      SG_TS(7);
      // the residual value of a channel is only ever needed by the wave that owns the channel: it stays in a register
      const float h2 = hres + yo + c_bo;
      if (lane < 2) sg_publish(gl + sg_off_h2(a) + no, c.tag, __float_as_uint(h2));
      // ================================================= D. LayerNorm 2 + FFN up-projection + ReLU
      if (!sg_gather<0>(c, gl + sg_off_h2(a), d, hb, 8)) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < d; i += 256) a.dbg[((size_t)l * 5 + 2) * 4 * d + i] = hb[i];
      SG_TS(8);
      float mean_c, rstd;
      sg_ln_center<2 * RPC>(c, hb, xs, mean_c, rstd);
      SG_TS(9);
      float y1 = sg_consume<8, RPC>(c, g0 + SQ + SO, xs, okc);
      if (!okc) return;
      SG_TS(10);
      if (a.ts && l == a.L / 2 && c.cw == 0 && lane == 0) a.ts[(size_t)3 * a.L * 16 + cu * 4 + 0] = (long long)wall_clock64();
      // lanes 0..7 hold the wave's 8 activations; the even lanes publish them as 4 granules of two bf16
      y1 = fmaxf(rstd * (y1 - mean_c * c_wg1) + c_b1, 0.f);
      const float y1n = __shfl_down(y1, 1, 64);
      if (lane < 8 && !(lane & 1)) sg_publish(gl + sg_off_a(a) + (n1 >> 1), c.tag, pack_bf16x2(y1, y1n));
      // ================================================= E. FFN down-projection + residual
      unsigned* x32 = reinterpret_cast<unsigned*>(xb);
      if (!sg_gather<0>(c, gl + sg_off_a(a), 2 * d, x32, 16)) return;
      if (!sg_cbarrier(c)) return;
      if (a.dbg) for (int i = c.tid_c; i < 4 * d; i += 256) a.dbg[((size_t)l * 5 + 3) * 4 * d + i] = bf16_to_f32(reinterpret_cast<uint16_t*>(xb)[i]);
      SG_TS(11);
      if (a.ts && l == a.L / 2 && c.cw == 0 && lane == 0) a.ts[(size_t)3 * a.L * 16 + cu * 4 + 1] = (long long)wall_clock64();
      const float y2 = sg_consume<2, 4 * RPC>(c, g0 + SQ + SO + S1, xb, okc);
      if (!okc) return;
      SG_TS(12);
      if (a.ts && l == a.L / 2 && c.cw == 0 && lane == 0) a.ts[(size_t)3 * a.L * 16 + cu * 4 + 2] = (long long)wall_clock64();
      const bool last = (l == a.L - 1);
      const float y = h2 + y2 + c_b2;
      if (lane < 2) {
        if (last) a.h_out[no] = y;
        else sg_publish(a.gran + (size_t)(l + 1) * a.gran_layer_stride + sg_off_h(a) + no, c.tag, __float_as_uint(y));
        if (a.dbg) a.dbg[((size_t)l * 5 + 4) * 4 * d + no] = y;
      }
      if (!last) {
        unsigned long long* gn = a.gran + (size_t)(l + 1) * a.gran_layer_stride;
        if (!sg_gather<0>(c, gn + sg_off_h(a), d, hb, 8)) return;
        if (!sg_cbarrier(c)) return;
      }
      SG_TS(13);
      if (a.ts && l == a.L / 2 && c.cw == 0 && lane == 0) a.ts[(size_t)3 * a.L * 16 + cu * 4 + 3] = (long long)wall_clock64();
      ly = lyn;
      c_wgq = n_wgq; c_bq = n_bq; c_bo = n_bo; c_b2 = n_b2; c_wg1 = n_wg1; c_b1 = n_b1;
      if (a.ts && ts_sel >= 0 && c.cw == 0 && lane == 0) a.ts[((size_t)ts_sel * a.L + l) * 16 + 14] = (long long)c.sweeps;
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
    for (const void* k : {reinterpret_cast<const void*>(stream_step_k<1>), reinterpret_cast<const void*>(stream_step_k<2>),
                          reinterpret_cast<const void*>(stream_step_k<4>)}) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SG_LDS_TOTAL);
      if (e != hipSuccess) return e;
    }
    granted[dev] = true;
  }
  ++vc_launch_counts[VC_LC_PERSIST];
  if (a.rpc == 4) hipLaunchKernelGGL(stream_step_k<4>, dim3(a.G), dim3(SG_THREADS), SG_LDS_TOTAL, s, a);
  else if (a.rpc == 2) hipLaunchKernelGGL(stream_step_k<2>, dim3(a.G), dim3(SG_THREADS), SG_LDS_TOTAL, s, a);
  else if (a.rpc == 1) hipLaunchKernelGGL(stream_step_k<1>, dim3(a.G), dim3(SG_THREADS), SG_LDS_TOTAL, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ================================================================ weight prefetcher for the launch path
// A decode step on the launch path (forward_rows + heads) is 4 L + 2 weight-streaming launches; each starts cold: its
// weights come from HBM behind a fresh dispatch (FFN-up 8.7 us), while the same launch takes 5.0 us when its weights
// already sit in the 256 MB Infinity Cache ("_hot" rows of bench.py).  This kernel keeps them there: ONE persistent
// launch per call on a side stream, one wave per CU, streaming the weight matrices of the step in launch order with
// LDS-DMA into a scrap LDS tile (no registers, nothing is waited for except to bound the queue), a few matrices AHEAD
// of the compute launches.  The pace comes from the device: every decode GEMM launch stores its matrix index to `prog`
// when it starts (GemmArgs.progress); matrix q (counted from the call's first launch) may be requested once q < launches + ahead.  It ends when the last
// sequence retires (*n_active == 0), when the host raises *stop, or after a bounded number of idle polls.
// (Round 1 forked one prefetch launch per layer with event fork/join inside the graph: 17 forks per step cost more
// than the cold misses they removed - 1.26 ms per step, profiles/r01_bench_prefetch_sidestream_on.json.log.)
__global__ __launch_bounds__(64) void weight_prefetch_k(const PrefetchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // 16 KB scrap
  const int lane = threadIdx.x, cu = blockIdx.x;
  const unsigned per_step = (unsigned)a.n_seg;
  // the launches publish "matrix i of the step is being read" (i + 1; 0 = the call has not started a step yet); the
  // prefetcher unwraps that into a count of launches since the call began: a drop of the index is a new step
  unsigned last = 0, base = 0, cur = 0, idle = 0;
  const bool direct = a.ahead < 0;                       // microbenchmark form: request matrix (-ahead - 1) once and leave
  for (unsigned q = direct ? (unsigned)(-a.ahead - 1) : 0u;; ++q) {
    if (direct && q != (unsigned)(-a.ahead - 1)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    // ---- pace: matrix q (counted from the call's first launch) may be requested once q < cur + ahead
    for (unsigned it = 0; !direct; ++it) {
      const unsigned v = __hip_atomic_load(a.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v < last) base += per_step;
      last = v;
      cur = base + v;                                    // launches started so far
      if ((int)(q - cur) < a.ahead) break;
      if (*(volatile const int*)a.n_active == 0) return;
      if ((it & 255) == 255 && __hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
      if (++idle > (1u << 20)) return;                  // ~2 s of waiting: the caller is gone
      __builtin_amdgcn_s_sleep(40);                     // ~1 us: a launch lasts 4-9 us
    }
    const PrefetchSeg sg = a.segs[q % per_step];
    // this workgroup's share of the matrix, in 16 KB pieces
    const size_t share = (sg.bytes / (size_t)a.G + 16383) & ~(size_t)16383;
    const size_t b0 = (size_t)cu * share;
    for (size_t off = 0; off < share && b0 + off + 16384 <= sg.bytes; off += 16384) {
      const char* src = sg.ptr + b0 + off + (size_t)lane * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + i * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(32)" ::: "memory");     // at most three pieces in flight per CU
    }
  }
}

hipError_t vc_prefetch_launch(const PrefetchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(weight_prefetch_k, dim3(a.G), dim3(64), 16384, s, a);
  return hipGetLastError();
}
