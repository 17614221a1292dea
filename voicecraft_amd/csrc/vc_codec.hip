// vc_codec.hip — EnCodec (SEANet encoder/decoder + LSTM + residual VQ) on gfx950.
//
// Replaces AudioTokenizer.encode/.decode (data/tokenizer.py:127-133 -> audiocraft EncodecModel,
// not vendored; architecture restated from transformers.EncodecModel at the VoiceCraft codec shape,
// SURVEY.md §8c).  Everything is fp32: RVQ codes come out of an arg-min, so the convolutions are
// kept at full precision and run on the fp32 MFMA (v_mfma_f32_16x16x4_f32, an exact fma chain).
//
// Layout: activations are CHANNELS-LAST [time][channel].  A 1-D convolution is then an implicit
// GEMM  out[t][co] = sum_{k,ci} W[co][ci][k] * x[t*s + k*d - pad][ci]  whose K index is ordered
// (tap k, channel ci): one 16-wide K tile is 16 consecutive channels of ONE input position, so
// both MFMA operands are plain 16-byte loads — weights from a pre-packed fragment image, inputs
// straight from the activation row (no im2col, no LDS staging).  A transposed convolution with
// kernel = 2*stride is `stride` such GEMMs (one per output phase r) with two taps each:
//   out[t*s + r - pl][co] = sum_ci W[ci][co][r]*x[t][ci] + W[ci][co][r+s]*x[t-1][ci].
// ELU always precedes a convolution in SEANet, so every epilogue can write the raw tensor (for
// the residual skip) and/or its ELU (for the next convolution) — each element is activated once.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/vc_codec.h"
#include "vc_common.h"

// ============================================================================ kernels
struct ConvArgs {
  const float* x;        // [L_in][Ci]
  const float4* Wp;      // [phase][co_tile][k_tile][lane]
  const float* bias;     // [Co]
  const float* res;      // optional residual [L_dst][Co]
  float* out_raw;        // optional [L_dst][Co]
  float* out_elu;        // optional [L_dst][Co]
  int L_in, T, Ci, Co, Kw;
  int s_in, dil, pad, reflect;
  int L_ext;             // reflect padding of an input shorter than its padding: the input is first zero-extended to L_ext positions (EncodecConv1d._pad1d), else = L_in
  int s_out, o_off, L_dst;
  long w_phase_stride;   // float4 units
  int nphase;            // grid.z = nphase * batch: phase = z % nphase, batch item = z / nphase
  long x_bstride, o_bstride;   // floats between consecutive batch items of x and of out/res
};

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }

// block = 4 waves along t; a wave owns 32 output channels x 32 positions (2x2 MFMA tiles).
__global__ __launch_bounds__(256) void conv_gemm_k(const ConvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, m = lane & 15;
  const int phase = blockIdx.z % a.nphase, bi = blockIdx.z / a.nphase;
  const float* xbase = a.x + (long)bi * a.x_bstride;
  const long obase = (long)bi * a.o_bstride;
  const int t0 = (blockIdx.x * 4 + wave) * 32;
  const int cot0 = blockIdx.y * 2;                       // first 16-channel tile
  if (t0 >= a.T) return;
  const int ktpk = a.Ci >> 4;                            // k-tiles per tap
  const int KT = a.Kw * ktpk;
  const float4* wp0 = a.Wp + (long)phase * a.w_phase_stride + ((long)cot0 * KT) * 64 + lane;
  const float4* wp1 = wp0 + (long)KT * 64;
  const bool two = (cot0 + 1) * 16 < a.Co;               // Co may be a single 16-tile multiple of 32? (always 32-multiples here)
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // input position of output position t for tap k: (pointer, valid) - explicit scalars (an indexed pair of pointers is
  // demoted to scratch memory by the compiler)
  auto tap = [&](int ni, int k, const float*& xp, bool& ok) {
    const int t = min(t0 + ni * 16 + m, a.T - 1);
    int p = t * a.s_in + k * a.dil - a.pad;
    bool v = true;
    if (a.reflect) {
      if (p < 0) p = -p;
      if (p >= a.L_ext) p = 2 * (a.L_ext - 1) - p;
      v = p < a.L_in;                                    // the zero extension of an input shorter than its padding
      p = max(0, min(p, a.L_in - 1));
    } else {
      v = (p >= 0 && p < a.L_in);
      p = max(0, min(p, a.L_in - 1));
    }
    ok = v;
    xp = xbase + (long)p * a.Ci + 4 * g;
  };
  for (int k = 0; k < a.Kw; ++k) {
    const float *xp0, *xp1;
    bool ok0, ok1;
    tap(0, k, xp0, ok0);
    tap(1, k, xp1, ok1);
    const float4* w0 = wp0 + (long)k * ktpk * 64;
    const float4* w1 = (two ? wp1 : wp0) + (long)k * ktpk * 64;      // (a select between a load and a zero constant puts the constant in scratch)
#pragma unroll 2
    for (int c = 0; c < ktpk; ++c) {
      const float4 wa = w0[(long)c * 64];
      float4 wb = w1[(long)c * 64];
      if (!two) wb = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xa = *reinterpret_cast<const float4*>(xp0 + c * 16);
      float4 xb = *reinterpret_cast<const float4*>(xp1 + c * 16);
      if (!ok0) xa = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!ok1) xb = make_float4(0.f, 0.f, 0.f, 0.f);
      const uint4 uwa = __builtin_bit_cast(uint4, wa), uwb = __builtin_bit_cast(uint4, wb);
      const uint4 uxa = __builtin_bit_cast(uint4, xa), uxb = __builtin_bit_cast(uint4, xb);
      acc[0][0] = mfma_frag(uwa, uxa, acc[0][0], (float*)nullptr);
      acc[0][1] = mfma_frag(uwa, uxb, acc[0][1], (float*)nullptr);
      acc[1][0] = mfma_frag(uwb, uxa, acc[1][0], (float*)nullptr);
      acc[1][1] = mfma_frag(uwb, uxb, acc[1][1], (float*)nullptr);
    }
  }
  // epilogue: lane holds channels co..co+3 of position t
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int co = (cot0 + mi) * 16 + 4 * g;
    if (co >= a.Co) continue;
    const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int t = t0 + ni * 16 + m;
      const int o = t * a.s_out + a.o_off + phase;
      if (t >= a.T || o < 0 || o >= a.L_dst) continue;
      float4 v = make_float4(acc[mi][ni][0] + b.x, acc[mi][ni][1] + b.y, acc[mi][ni][2] + b.z, acc[mi][ni][3] + b.w);
      const long off = obase + (long)o * a.Co + co;
      if (a.res) {
        const float4 r = *reinterpret_cast<const float4*>(a.res + off);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      if (a.out_raw) *reinterpret_cast<float4*>(a.out_raw + off) = v;
      if (a.out_elu) *reinterpret_cast<float4*>(a.out_elu + off) = make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));
    }
  }
}

// weights -> fragment image.  conv: W[Co][Ci][Kw]; transposed (phase r, taps r and r+s): W[Ci][Co][2s]
__global__ void conv_pack_k(const float* __restrict__ W, float4* __restrict__ Wp, int Co, int Ci, int Kw,
                            int transposed, int stride, int phase, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long tile = idx >> 6;
  const int ktpk = Ci >> 4;
  const int KT = Kw * ktpk;
  const int kt = (int)(tile % KT), cot = (int)(tile / KT);
  const int k = kt / ktpk, cit = kt - k * ktpk;
  const int co = cot * 16 + (lane & 15);
  const int ci = cit * 16 + 4 * (lane >> 4);
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (co >= Co) v[j] = 0.f;
    else if (!transposed) v[j] = W[((long)co * Ci + ci + j) * Kw + k];
    else v[j] = W[((long)(ci + j) * Co + co) * (2 * stride) + phase + k * stride];
  }
  Wp[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

// first layer: 1 -> Co channels, reflect padding (Ci = 1 has no 16-wide K tile)
__global__ void conv_first_k(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                             float* __restrict__ out_raw, float* __restrict__ out_elu, int L, int Co, int Kw, int pad,
                             int reflect, int B, int L_ext) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = Co >> 2;
  if (idx >= (long)B * L * cq) return;
  const long tg = idx / cq;                              // b * L + t
  const int t = (int)(tg % L), co = (int)(idx % cq) * 4;
  const float* xb = x + (tg - t);
  float4 v = *reinterpret_cast<const float4*>(bias + co);
  for (int k = 0; k < Kw; ++k) {
    int p = t + k - pad;
    bool ok = true;
    if (reflect) {
      if (p < 0) p = -p;
      if (p >= L_ext) p = 2 * (L_ext - 1) - p;
      ok = p < L;
    } else {
      ok = (p >= 0 && p < L);
    }
    p = max(0, min(p, L - 1));
    const float xv = ok ? xb[p] : 0.f;
    v.x += W[(co + 0) * Kw + k] * xv; v.y += W[(co + 1) * Kw + k] * xv;
    v.z += W[(co + 2) * Kw + k] * xv; v.w += W[(co + 3) * Kw + k] * xv;
  }
  const long off = tg * Co + co;
  if (out_raw) *reinterpret_cast<float4*>(out_raw + off) = v;
  if (out_elu) *reinterpret_cast<float4*>(out_elu + off) = make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));
}

// last layer: Ci -> 1 channel.  W is [1][Ci][Kw]; x is the ELU'd [L][Ci].
__global__ void conv_last_k(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                            float* __restrict__ out, int L, int Ci, int Kw, int pad, int reflect, int L_ext) {
  extern __shared__ float s_w[];                         // [Kw][Ci]
  for (int i = threadIdx.x; i < Kw * Ci; i += blockDim.x) { const int k = i / Ci, ci = i - k * Ci; s_w[i] = W[ci * Kw + k]; }
  __syncthreads();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  x += (long)blockIdx.y * L * Ci;                         // batch item
  out += (long)blockIdx.y * L;
  float acc = bias[0];
  for (int k = 0; k < Kw; ++k) {
    int p = t + k - pad;
    if (reflect) {
      if (p < 0) p = -p;
      if (p >= L_ext) p = 2 * (L_ext - 1) - p;
      if (p >= L) continue;                               // zero extension (input shorter than its padding)
    } else if (p < 0 || p >= L) {
      continue;
    }
    p = max(0, min(p, L - 1));
    const float4* xr = reinterpret_cast<const float4*>(x + (long)p * Ci);
    const float4* wr = reinterpret_cast<const float4*>(s_w + k * Ci);
    for (int c = 0; c < (Ci >> 2); ++c) {
      const float4 a = xr[c], w = wr[c];
      acc += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    }
  }
  out[t] = acc;
}

// ---- LSTM recurrence, one launch per time step (nn.LSTM gate order i,f,g,o; rows re-ordered so the
// four gates of hidden unit u are rows 4u..4u+3).  gates = G[t] (input projection + both biases,
// computed for all t by one implicit GEMM) + W_hh h_{t-1}.
struct LstmArgs {
  const float* Whh;      // [4H][H], rows permuted (unit-major)
  const float* G;        // [T][4H] permuted the same way
  const float* h_prev;   // [H]
  float* c;              // [H] in/out
  float* h_out;          // [H]: row t of the layer's output sequence
  const float* skip;     // optional [H]: x_t, added to h for the block output (EncodecLSTM: lstm(x) + x)
  float* out_raw;        // optional [H]
  float* out_elu;        // optional [H]
  int H;
};
__global__ __launch_bounds__(256) void lstm_step_k(const LstmArgs a) {
  // one wave per hidden unit: its four gate rows (4 x H floats, contiguous after the permutation) are
  // streamed with 16-byte loads, h_{t-1} comes from LDS, each dot product ends in a DPP wave sum.
  extern __shared__ float s_h[];                         // h_{t-1}
  const int H = a.H;
  for (int i = threadIdx.x; i < H; i += blockDim.x) s_h[i] = a.h_prev[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;                   // hidden unit
  if (u >= H) return;
  const int nq = H >> 2;                                 // float4 per row
  const float4* w = reinterpret_cast<const float4*>(a.Whh + (long)(4 * u) * H);
  const float4* hv = reinterpret_cast<const float4*>(s_h);
  float g4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < nq; i += 64) {
    const float4 hh = hv[i];
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) {
      const float4 ww = w[(long)gate * nq + i];
      g4[gate] += (ww.x * hh.x + ww.y * hh.y) + (ww.z * hh.z + ww.w * hh.w);
    }
  }
#pragma unroll
  for (int gate = 0; gate < 4; ++gate) g4[gate] = wave_sum(g4[gate]);
  if (lane != 0) return;
  const float4 gi = *reinterpret_cast<const float4*>(a.G + 4 * u);
  const float ig = 1.f / (1.f + expf(-(g4[0] + gi.x)));
  const float fg = 1.f / (1.f + expf(-(g4[1] + gi.y)));
  const float gg = tanhf(g4[2] + gi.z);
  const float og = 1.f / (1.f + expf(-(g4[3] + gi.w)));
  const float c = fg * a.c[u] + ig * gg;
  const float h = og * tanhf(c);
  a.c[u] = c;
  a.h_out[u] = h;
  if (a.skip) {
    const float y = h + a.skip[u];
    if (a.out_raw) a.out_raw[u] = y;
    if (a.out_elu) a.out_elu[u] = elu1(y);
  }
}

// ---- the same recurrence as a two-layer wavefront: launch k advances layer 0 to step k and layer 1
// to step k-1 (grid.y = layer), so a T-frame sequence takes T+1 launches instead of 2T, and layer 1's
// input projection (W_ih h0_t) is folded into its step instead of being a separate GEMM over all T.
// Every operand of a step is requested up front (one round trip): the 4 gate rows of W_hh (and W_ih),
// h_{t-1} (and the lower layer's h_t) straight from L2 - no LDS staging, no block barrier.
struct LstmWaveArgs {
  const float* Whh[2];   // [4H][H], unit-major rows
  const float* Wih1;     // layer 1: [4H][H], unit-major rows
  const float* b1;       // layer 1: b_ih + b_hh, unit-major
  const float* G0;       // layer 0: [B][T][4H] input projection + both biases
  float* hs[2];          // [B][T][H] hidden sequences
  float* c[2];           // [B][H] cell states
  const float* hzero;    // [H] zeros
  const float* skip;     // [B][T][H] block input (EncodecLSTM: lstm(x) + x)
  float* out_raw;        // optional [B][T][H]
  float* out_elu;        // optional [B][T][H]
  int H, T, k, B;        // B sequences advance together: the step's weights are read ONCE for all of them
};
template <int NQ>   // H = 256 * NQ
__global__ __launch_bounds__(256) void lstm_wave_k(const LstmWaveArgs a) {
  const int n = blockIdx.y;
  const int t = a.k - n;
  if (t < 0 || t >= a.T) return;
  const int H = a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;                   // hidden unit (grid.x = H / 4)
  const int nq = H >> 2;
  const long TH = (long)a.T * H;
  const float4* whh = reinterpret_cast<const float4*>(a.Whh[n] + (long)(4 * u) * H);
  float4 w[4][NQ], wi[4][NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g][j] = whh[(long)g * nq + lane + 64 * j];
  }
  float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n == 1) {
    const float4* wih = reinterpret_cast<const float4*>(a.Wih1 + (long)(4 * u) * H);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) wi[g][j] = wih[(long)g * nq + lane + 64 * j];
    }
    gb = *reinterpret_cast<const float4*>(a.b1 + 4 * u);
  }
  for (int b = 0; b < a.B; ++b) {
    const float4* hp = reinterpret_cast<const float4*>(t ? a.hs[n] + b * TH + (long)(t - 1) * H : a.hzero);
    float4 hv[NQ], xv[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) hv[j] = hp[lane + 64 * j];
    float4 gi = gb;
    if (n == 0) {
      gi = *reinterpret_cast<const float4*>(a.G0 + (b * (long)a.T + t) * 4 * H + 4 * u);
    } else {
      const float4* xp = reinterpret_cast<const float4*>(a.hs[0] + b * TH + (long)t * H);
#pragma unroll
      for (int j = 0; j < NQ; ++j) xv[j] = xp[lane + 64 * j];
    }
    const float c_prev = a.c[n][(long)b * H + u];
    const float sk = (n == 1 && a.skip) ? a.skip[b * TH + (long)t * H + u] : 0.f;
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        g4[g] += (w[g][j].x * hv[j].x + w[g][j].y * hv[j].y) + (w[g][j].z * hv[j].z + w[g][j].w * hv[j].w);
    }
    if (n == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          acc += (wi[g][j].x * xv[j].x + wi[g][j].y * xv[j].y) + (wi[g][j].z * xv[j].z + wi[g][j].w * xv[j].w);
        g4[g] += acc;
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) g4[g] = wave_sum(g4[g]);
    if (lane == 0) {
      const float ig = 1.f / (1.f + expf(-(g4[0] + gi.x)));
      const float fg = 1.f / (1.f + expf(-(g4[1] + gi.y)));
      const float gg = tanhf(g4[2] + gi.z);
      const float og = 1.f / (1.f + expf(-(g4[3] + gi.w)));
      const float c = fg * c_prev + ig * gg;
      const float h = og * tanhf(c);
      a.c[n][(long)b * H + u] = c;
      a.hs[n][b * TH + (long)t * H + u] = h;
      if (n == 1 && a.skip) {
        const float y = h + sk;
        if (a.out_raw) a.out_raw[b * TH + (long)t * H + u] = y;
        if (a.out_elu) a.out_elu[b * TH + (long)t * H + u] = elu1(y);
      }
    }
  }
}

// ---- the recurrence as ONE persistent cooperative launch (the default for hidden 512 / 1024; lstm_wave_k is the fallback).
// A wave keeps its unit's gate rows (16 KB of W_hh, layer 1 also 16 KB of W_ih) in registers for the whole sequence
// instead of re-reading 48 MB of weights per step, and hidden values travel between workgroups as 8-byte granules
// {h bits, epoch} written and read with ONE relaxed agent-scope atomic each (value and tag cannot tear and need no
// fence; MI355X_MICROARCH "data-tagged hand-off").  A granule is valid when its tag equals the call's epoch, so nothing
// is cleared between calls.  A workgroup is 8 waves = 8 units (2 x H/8 workgroups: one per CU at H = 1024); its waves
// poll ONE EIGHTH of the hidden vector each (H/8 granules, one or two per lane), park the values in LDS and meet at one
// block barrier per step; every wave then reads the whole vector from LDS (2 MB of L2 traffic per poll round; a first
// form in which every wave polled the whole vector moved 24 MB and lost to the launches: 14.0 against 10.4 us per step,
// profiles/r02_lstm_persist_probe.log).  The LDS vector is double-buffered by step parity, so one barrier per step is
// enough.  The arithmetic (order of every sum) is lstm_wave_k's: codes AND waveform are bit-identical to it
// (profiles/r03_next_round.log).  Launched cooperatively (every workgroup resident: a wait can only be for a workgroup
// that is running); every wait is bounded: a wait that gives up raises an LDS flag BEFORE the barrier, the whole
// workgroup leaves together and the host reports err instead of hanging.
// Measured (16 s of audio): 4.8 us per step against 8.8 us for one launch per step - encode 9.4 -> 6.2 ms, decode 8.1 -> 5.3 ms.
struct LstmPersistArgs {
  const float* Whh[2];
  const float* Wih1;
  const float* b1;
  const float* G0;
  unsigned long long* hg[2];   // [B][T][H] granules of the two hidden sequences
  float* c[2];                 // [B][H] cell states (private to the wave that owns the unit)
  const float* skip;
  float* out_raw;
  float* out_elu;
  int H, T, B;
  unsigned epoch;              // != 0, differs from call to call
  int* err;
};
#define VC_LSTM_SPIN_LIMIT 400000
#define VC_LSTM_BG 8            // clips advanced per hand-off round (LDS: 2 parities x BG x {h, x} x H floats = 128 KB at H = 1024)
template <int NQ>   // H = 256 * NQ
__global__ __launch_bounds__(512) void lstm_persist_k(const LstmPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_hx[];          // [parity][clip][h_prev | x][H]
  __shared__ int s_abort;
  const int H = a.H;
  const int per_layer = H >> 3;                          // workgroups per layer
  const int n = blockIdx.x / per_layer;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = (blockIdx.x - n * per_layer) * 8 + wave;  // hidden unit
  const int nq = H >> 2;
  const long TH = (long)a.T * H;
  if (threadIdx.x == 0) s_abort = 0;
  const float4* whh = reinterpret_cast<const float4*>(a.Whh[n] + (long)(4 * u) * H);
  float4 w[4][NQ], wi[4][NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
#pragma unroll
    for (int g = 0; g < 4; ++g) w[g][j] = whh[(long)g * nq + lane + 64 * j];
  }
  float4 gb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n == 1) {
    const float4* wih = reinterpret_cast<const float4*>(a.Wih1 + (long)(4 * u) * H);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) wi[g][j] = wih[(long)g * nq + lane + 64 * j];
    }
    gb = *reinterpret_cast<const float4*>(a.b1 + 4 * u);
  } else {
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) wi[g][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // this wave's slice of a hidden vector: granules [wave * H/8, (wave + 1) * H/8) = GPL consecutive granules per lane
  // (this form accepts H = 512 or 1024: one or two per lane)
  constexpr int GPL = NQ / 2;
  static_assert(NQ == 2 || NQ == 4, "lstm_persist_k: hidden 512 or 1024");
  const int s0 = wave * (H >> 3) + GPL * lane;
  // The clips of a batch advance together, VC_LSTM_BG at a time: ONE hand-off round (poll, park, barrier) per step serves
  // every clip of the group, and the register-resident weights are reused for all of them.
  const int bg = min(VC_LSTM_BG, a.B);                    // clips per group = what the launcher sized the LDS for
  for (int b0 = 0; b0 < a.B; b0 += VC_LSTM_BG) {
    const int nb = min(VC_LSTM_BG, a.B - b0);
    for (int t = 0; t < a.T; ++t) {
      float* sbuf = s_hx + (size_t)(t & 1) * (bg * 2 * H);
      // ---- bounded wait for this wave's slices of h_{t-1} (own layer) and, on layer 1, of the lower layer's h_t
      bool ok = true;
      if (t || n == 1) {
        ok = false;
        for (int spins = 0; spins < VC_LSTM_SPIN_LIMIT && !ok; ++spins) {
          bool all = true;
#pragma unroll 1
          for (int bb = 0; bb < nb; ++bb) {
            const long bo = (long)(b0 + bb) * TH;
            float* sh = sbuf + (size_t)bb * 2 * H;
#pragma unroll
            for (int q = 0; q < GPL; ++q) {
              if (t) {
                const unsigned long long g0 = __hip_atomic_load(a.hg[n] + bo + (long)(t - 1) * H + s0 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && (unsigned)(g0 >> 32) == a.epoch;
                sh[s0 + q] = __uint_as_float((unsigned)g0);
              }
              if (n == 1) {
                const unsigned long long g1 = __hip_atomic_load(a.hg[0] + bo + (long)t * H + s0 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && (unsigned)(g1 >> 32) == a.epoch;
                sh[H + s0 + q] = __uint_as_float((unsigned)g1);
              }
            }
          }
          ok = __all(all);                                 // (values parked by an unsuccessful round are overwritten by the next)
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
      }
      if (!ok && lane == 0) {
        s_abort = 1;
        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __syncthreads();
      if (s_abort) return;                               // every wave sees the flag behind the same barrier
#pragma unroll 1
      for (int bb = 0; bb < nb; ++bb) {
        const int b = b0 + bb;
        const float* sh = sbuf + (size_t)bb * 2 * H;
        float4 hv[NQ], xv[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          hv[j] = t ? reinterpret_cast<const float4*>(sh)[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
          xv[j] = (n == 1) ? reinterpret_cast<const float4*>(sh + H)[lane + 64 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 gi = gb;
        if (n == 0) gi = *reinterpret_cast<const float4*>(a.G0 + (b * (long)a.T + t) * 4 * H + 4 * u);
        const float c_prev = t ? a.c[n][(long)b * H + u] : 0.f;
        const float sk = (n == 1 && a.skip) ? a.skip[b * TH + (long)t * H + u] : 0.f;
        float g4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            g4[g] += (w[g][j].x * hv[j].x + w[g][j].y * hv[j].y) + (w[g][j].z * hv[j].z + w[g][j].w * hv[j].w);
        }
        if (n == 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NQ; ++j)
              acc += (wi[g][j].x * xv[j].x + wi[g][j].y * xv[j].y) + (wi[g][j].z * xv[j].z + wi[g][j].w * xv[j].w);
            g4[g] += acc;
          }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) g4[g] = wave_sum(g4[g]);
        if (lane == 0) {
          const float ig = 1.f / (1.f + expf(-(g4[0] + gi.x)));
          const float fg = 1.f / (1.f + expf(-(g4[1] + gi.y)));
          const float gg = tanhf(g4[2] + gi.z);
          const float og = 1.f / (1.f + expf(-(g4[3] + gi.w)));
          const float c = fg * c_prev + ig * gg;
          const float h = og * tanhf(c);
          a.c[n][(long)b * H + u] = c;
          __hip_atomic_store(a.hg[n] + b * TH + (long)t * H + u, ((unsigned long long)a.epoch << 32) | (unsigned long long)__float_as_uint(h),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n == 1 && a.skip) {
            const float y = h + sk;
            if (a.out_raw) a.out_raw[b * TH + (long)t * H + u] = y;
            if (a.out_elu) a.out_elu[b * TH + (long)t * H + u] = elu1(y);
          }
        }
      }
    }
    __syncthreads();   // the next group starts on parity 0 again: nobody may still be reading this group's last buffers
  }
}

// ---- residual VQ (EncodecResidualVectorQuantizer.encode/.decode).  One block per frame.
// dist = -(|r|^2 - 2 r.e + |e|^2), arg-max with the lowest index on ties (torch.max), residual update.
__global__ __launch_bounds__(256) void rvq_encode_k(const float* __restrict__ z, const float* __restrict__ Et,
                                                    const float* __restrict__ E, const float* __restrict__ e2,
                                                    int64_t* __restrict__ codes, int T, int D, int C, int Q) {
  extern __shared__ float s_r[];                         // [D] residual, then scratch
  __shared__ float s_bv[4];
  __shared__ int s_bi[4];
  __shared__ int s_best;
  const int tg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;     // grid.x = B * T frames
  const int t = tg % T;
  codes += (long)(tg / T) * Q * T;                        // codes [B][Q][T]
  for (int i = tid; i < D; i += blockDim.x) s_r[i] = z[(long)tg * D + i];
  __syncthreads();
  for (int qz = 0; qz < Q; ++qz) {
    float r2 = 0.f;
    for (int i = 0; i < D; ++i) r2 += s_r[i] * s_r[i];
    const float* et = Et + (long)qz * D * C;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = tid; c < C; c += blockDim.x) {
      float dot = 0.f;
      for (int i = 0; i < D; ++i) dot += s_r[i] * et[(long)i * C + c];
      const float dist = -(r2 - 2.f * dot + e2[qz * C + c]);
      if (dist > bv) { bv = dist; bi = c; }              // c ascending: first maximum wins
    }
    const float wm = wave_max(bv);
    const int wi = wave_min_i(bv == wm ? bi : 0x7fffffff);
    if (lane == 0) { s_bv[wave] = wm; s_bi[wave] = wi; }
    __syncthreads();
    if (tid == 0) {
      float m = s_bv[0]; int b = s_bi[0];
      for (int w = 1; w < 4; ++w) if (s_bv[w] > m || (s_bv[w] == m && s_bi[w] < b)) { m = s_bv[w]; b = s_bi[w]; }
      s_best = b;
      codes[(long)qz * T + t] = b;
    }
    __syncthreads();
    const float* eb = E + ((long)qz * C + s_best) * D;
    for (int i = tid; i < D; i += blockDim.x) s_r[i] -= eb[i];
    __syncthreads();
  }
}
// The same search on the fp32 MFMA: a workgroup owns 16 frames, stage q is the [2048 codes] x [16 frames] product
// E_q R^T (A = codebook rows straight from HBM/L2 - a row IS an A fragment -, B = the residual rows from LDS, loaded
// once per stage), every wave takes every 4th 16-code tile, a lane keeps the best (distance, index) of the 4 codes x
// 1 frame it sees per tile, and the 16 candidates per frame (4 lane groups x 4 waves) are settled through LDS with
// the lowest index winning ties (torch.max returns the first maximum).  dist is evaluated exactly as the reference
// writes it, -((|r|^2 - 2 r.e) + |e|^2).  One block per frame with scalar dot products took 1.5 ms of a 16 s encode.
__global__ __launch_bounds__(256) void rvq_encode_mfma_k(const float* __restrict__ z, const float* __restrict__ E,
                                                         const float* __restrict__ e2, int64_t* __restrict__ codes,
                                                         int BT, int T, int D, int C, int Q) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int RS = D * 4 + 16;                               // LDS row stride of the residual tile (bytes)
  float* s_r2 = reinterpret_cast<float*>(sm + 16 * RS);    // [16] |r|^2
  float* s_bv = s_r2 + 16;                                 // [4 waves][4 groups][16 frames]
  int* s_bi = reinterpret_cast<int*>(s_bv + 256);
  int* s_best = s_bi + 256;                                // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int f0 = blockIdx.x * 16;
  const int nks = D >> 4;                                  // 16-dim k-steps (<= 16)
  for (int i = tid; i < 16 * (D >> 2); i += 256) {         // residual := latent rows (frames past the end: zeros)
    const int fr = i / (D >> 2), c4 = i - fr * (D >> 2);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f0 + fr < BT) v = *reinterpret_cast<const float4*>(z + (long)(f0 + fr) * D + 4 * c4);
    *reinterpret_cast<float4*>(sm + fr * RS + 16 * c4) = v;
  }
  __syncthreads();
  for (int qz = 0; qz < Q; ++qz) {
    if (tid < 16) {                                        // |r|^2 in ascending dim order, as the scalar kernel did
      const float* rr = reinterpret_cast<const float*>(sm + tid * RS);
      float r2 = 0.f;
      for (int i = 0; i < D; ++i) r2 += rr[i] * rr[i];
      s_r2[tid] = r2;
    }
    uint4 rf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      if (ks < nks) rf[ks] = *reinterpret_cast<const uint4*>(sm + m * RS + (16 * ks + 4 * kg) * 4);
    __syncthreads();
    const float r2 = s_r2[m];
    const float* Eq = E + (long)qz * C * D;
    const float* e2q = e2 + qz * C;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int ct = wave; ct < (C >> 4); ct += 4) {
      const float* er = Eq + (long)(ct * 16 + m) * D + 4 * kg;     // A fragment: code ct*16 + m, dims 16 ks + 4 kg ..
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        if (ks < nks) {
          const uint4 ef = *reinterpret_cast<const uint4*>(er + 16 * ks);
          acc = mfma_frag(ef, rf[ks], acc, (float*)nullptr);        // D[code 4 kg + r][frame m]
        }
      const float4 ee = *reinterpret_cast<const float4*>(e2q + ct * 16 + 4 * kg);
      const float en[4] = {ee.x, ee.y, ee.z, ee.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dist = -((r2 - 2.f * acc[r]) + en[r]);
        if (dist > bv) { bv = dist; bi = ct * 16 + 4 * kg + r; }   // codes ascend within a lane: the first maximum stays
      }
    }
    s_bv[(wave * 4 + kg) * 16 + m] = bv;
    s_bi[(wave * 4 + kg) * 16 + m] = bi;
    __syncthreads();
    if (tid < 16) {
      float best = -INFINITY;
      int idx = 0x7fffffff;
      for (int g = 0; g < 16; ++g) {
        const float v = s_bv[g * 16 + tid];
        const int ix = s_bi[g * 16 + tid];
        if (v > best || (v == best && ix < idx)) { best = v; idx = ix; }
      }
      s_best[tid] = idx;
      const int fg = f0 + tid;
      if (fg < BT) codes[((long)(fg / T) * Q + qz) * T + (fg % T)] = idx;
    }
    __syncthreads();
    for (int i = tid; i < 16 * D; i += 256) {              // residual update
      const int fr = i / D, dd = i - fr * D;
      float* rr = reinterpret_cast<float*>(sm + fr * RS);
      rr[dd] -= Eq[(long)s_best[fr] * D + dd];
    }
    __syncthreads();
  }
}
__global__ void rvq_decode_k(const int64_t* __restrict__ codes, const float* __restrict__ E, float* __restrict__ out,
                             int T, int D, int C, int Q, int* err, int B) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * T * D) return;
  const long tg = idx / D;                                // b * T + t
  const int t = (int)(tg % T), i = (int)(idx % D);
  codes += (tg / T) * (long)Q * T;                        // codes [B][Q][T]
  float v = 0.f;
  for (int q = 0; q < Q; ++q) {                          // quantized_out = 0 + q0 + q1 + ... in this order
    long c = codes[(long)q * T + t];
    if (c < 0 || c >= C) { *err = 1; c = 0; }
    v = v + E[((long)q * C + c) * D + i];
  }
  out[idx] = v;
}
__global__ void transpose_k(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {   // [R][Cc] -> [Cc][R]
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)R * Cc) return;
  const int r = (int)(idx / Cc), c = (int)(idx % Cc);
  dst[(long)c * R + r] = src[idx];
}
__global__ void rownorm2_k(const float* __restrict__ E, float* __restrict__ e2, int rows, int D) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int i = 0; i < D; ++i) s += E[(long)r * D + i] * E[(long)r * D + i];
  e2[r] = s;
}
// unit-major row permutation of LSTM matrices/biases: dst row 4u+g <- src row g*H+u
__global__ void lstm_perm_k(const float* __restrict__ src, float* __restrict__ dst, int H, int cols) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)4 * H * cols) return;
  const int row = (int)(idx / cols), col = (int)(idx % cols);
  const int u = row >> 2, g = row & 3;
  dst[idx] = src[((long)g * H + u) * cols + col];
}
__global__ void add_vec_k(float* __restrict__ a, const float* __restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}

// ============================================================================ host side
namespace {
std::string g_codec_create_err;

struct Conv {
  int Ci = 0, Co = 0, Kw = 0, stride = 1, transposed = 0;
  float* w_raw = nullptr;   // device fp32, reference layout
  float* bias = nullptr;
  float4* Wp = nullptr;     // packed (all phases)
  long phase_stride = 0;
};
struct ResUnit {            // EncodecResnetBlock: ELU, conv(k = residual_kernel_size, dilation), ELU, conv(k = 1) + shortcut
  Conv c3, c1, sc;
  int dil = 1;
  bool has_sc = false;
};
struct Lstm {
  int H = 0, layers = 0;
  std::vector<float*> Whh;   // permuted [4H][H]
  std::vector<Conv> Wih;     // as 1x1 "convs" H -> 4H with the summed, permuted bias
  std::vector<float*> WihP;  // the same matrices as plain permuted rows (wavefront kernel, layer 1)
  std::vector<float*> bP;    // permuted b_ih + b_hh
};
}  // namespace

struct vc_codec {
  vc_codec_cfg cfg{};
  int device = 0;
  bool finalized = false;
  std::string err;
  std::map<std::string, std::pair<std::vector<int64_t>, float*>> raw;
  std::vector<void*> allocs;
  int hop = 1;
  // encoder
  Conv enc_first, enc_last;
  std::vector<std::vector<ResUnit>> enc_res;   // [stage][residual layer]
  std::vector<Conv> enc_down;
  Lstm enc_lstm;
  // decoder
  Conv dec_first, dec_last;
  std::vector<Conv> dec_up;
  std::vector<std::vector<ResUnit>> dec_res;
  Lstm dec_lstm;
  // quantizer
  float *E = nullptr, *Et = nullptr, *e2 = nullptr;
  // activations
  float *A_raw = nullptr, *C_raw = nullptr, *S_raw = nullptr, *A_elu = nullptr, *B_elu = nullptr, *H_elu = nullptr, *latent = nullptr;
  int B_max = 1;
  float *G = nullptr, *HS0 = nullptr, *HS1 = nullptr, *cstate = nullptr, *hzero = nullptr;
  unsigned long long* hgran = nullptr;   // persistent LSTM: 2 x [B_max][T_max][H] granules, allocated on first use
  unsigned lstm_epoch = 0;
  int persist_ok = -1;                   // -1 not probed, 0 the cooperative grid does not fit, 1 usable
  bool persist_used = false;             // a persistent launch is in flight / unchecked (check_lstm_flag)
  int last_lstm_persist = 0;             // 1: the last LSTM ran as the persistent launch, 0: launch per step
  int* err_flag = nullptr;
  int* h_flag = nullptr;
  int T_max = 0;
  int last_T = 0;
  hipEvent_t ev[2]{};
  float last_ms = 0, last_lstm_ms = 0;
  hipEvent_t ev_l[2]{};
  hipStream_t own_stream = nullptr;
};

namespace {
int cfail(vc_codec* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_codec_create_err = buf;
  return code;
}
#define CCHK(c, call)                                                                    \
  do {                                                                                   \
    hipError_t _e = (call);                                                              \
    if (_e != hipSuccess) return cfail(c, VC_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

template <typename T>
int calloc_dev(vc_codec* c, T** p, size_t n) {
  void* v = nullptr;
  hipError_t e = hipMalloc(&v, std::max<size_t>(n * sizeof(T), 16));
  if (e != hipSuccess) return cfail(c, VC_EHIP, "hipMalloc(%zu) failed: %s", n * sizeof(T), hipGetErrorString(e));
  c->allocs.push_back(v);
  *p = reinterpret_cast<T*>(v);
  return VC_OK;
}

int get_raw(vc_codec* c, const std::string& key, std::vector<int64_t> shape, float** out) {
  auto it = c->raw.find(key);
  if (it == c->raw.end()) return cfail(c, VC_EMISSING, "codec weight '%s' was never loaded", key.c_str());
  if (it->second.first != shape) {
    std::string got, exp;
    for (auto v : it->second.first) got += std::to_string(v) + ",";
    for (auto v : shape) exp += std::to_string(v) + ",";
    return cfail(c, VC_EINVAL, "codec weight '%s' has shape [%s], expected [%s]", key.c_str(), got.c_str(), exp.c_str());
  }
  *out = it->second.second;
  return VC_OK;
}

int make_conv(vc_codec* c, const std::string& prefix, int Ci, int Co, int Kw, int stride, int transposed, Conv* cv, bool pack) {
  cv->Ci = Ci; cv->Co = Co; cv->Kw = Kw; cv->stride = stride; cv->transposed = transposed;
  int rc;
  if (transposed) { if ((rc = get_raw(c, prefix + ".weight", {Ci, Co, Kw}, &cv->w_raw))) return rc; }
  else { if ((rc = get_raw(c, prefix + ".weight", {Co, Ci, Kw}, &cv->w_raw))) return rc; }
  if ((rc = get_raw(c, prefix + ".bias", {Co}, &cv->bias))) return rc;
  if (!pack) return VC_OK;
  if (Ci % 16 || Co % 16) return cfail(c, VC_EINVAL, "%s: channels must be multiples of 16 (%d -> %d)", prefix.c_str(), Ci, Co);
  const int cot = (Co + 31) / 32 * 2;                    // tiles allocated in pairs (a wave owns two)
  if (!transposed) {
    const long KT = (long)Kw * (Ci / 16);
    const long total = (long)cot * KT * 64;
    if ((rc = calloc_dev(c, &cv->Wp, (size_t)total))) return rc;
    cv->phase_stride = total;
    hipLaunchKernelGGL(conv_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, cv->w_raw, cv->Wp, Co, Ci, Kw, 0, 1, 0, total);
  } else {
    if (Kw != 2 * stride) return cfail(c, VC_EINVAL, "%s: transposed conv needs kernel = 2*stride", prefix.c_str());
    const long KT = 2L * (Ci / 16);
    const long total = (long)cot * KT * 64;
    if ((rc = calloc_dev(c, &cv->Wp, (size_t)total * stride))) return rc;
    cv->phase_stride = total;
    for (int r = 0; r < stride; ++r)
      hipLaunchKernelGGL(conv_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, cv->w_raw, cv->Wp + (long)r * total, Co, Ci, 2, 1, stride, r, total);
  }
  CCHK(c, hipGetLastError());
  return VC_OK;
}

int make_lstm(vc_codec* c, const std::string& prefix, int H, int layers, Lstm* L) {
  L->H = H; L->layers = layers;
  int rc;
  for (int n = 0; n < layers; ++n) {
    const std::string sfx = "_l" + std::to_string(n);
    float *wih, *whh, *bih, *bhh;
    if ((rc = get_raw(c, prefix + ".weight_ih" + sfx, {4 * H, H}, &wih))) return rc;
    if ((rc = get_raw(c, prefix + ".weight_hh" + sfx, {4 * H, H}, &whh))) return rc;
    if ((rc = get_raw(c, prefix + ".bias_ih" + sfx, {4 * H}, &bih))) return rc;
    if ((rc = get_raw(c, prefix + ".bias_hh" + sfx, {4 * H}, &bhh))) return rc;
    float *pwhh, *pwih, *pb, *pb2;
    if ((rc = calloc_dev(c, &pwhh, (size_t)4 * H * H))) return rc;
    if ((rc = calloc_dev(c, &pwih, (size_t)4 * H * H))) return rc;
    if ((rc = calloc_dev(c, &pb, (size_t)4 * H))) return rc;
    if ((rc = calloc_dev(c, &pb2, (size_t)4 * H))) return rc;
    const long tot = (long)4 * H * H;
    hipLaunchKernelGGL(lstm_perm_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, whh, pwhh, H, H);
    hipLaunchKernelGGL(lstm_perm_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, wih, pwih, H, H);
    hipLaunchKernelGGL(lstm_perm_k, dim3((unsigned)((4 * H + 255) / 256)), dim3(256), 0, 0, bih, pb, H, 1);
    hipLaunchKernelGGL(lstm_perm_k, dim3((unsigned)((4 * H + 255) / 256)), dim3(256), 0, 0, bhh, pb2, H, 1);
    hipLaunchKernelGGL(add_vec_k, dim3((unsigned)((4 * H + 255) / 256)), dim3(256), 0, 0, pb, pb2, 4 * H);
    L->Whh.push_back(pwhh);
    L->WihP.push_back(pwih);
    L->bP.push_back(pb);
    Conv cv;
    cv.Ci = H; cv.Co = 4 * H; cv.Kw = 1; cv.stride = 1; cv.transposed = 0; cv.w_raw = pwih; cv.bias = pb;
    const long KT = H / 16;
    const long total = (long)(4 * H / 16) * KT * 64;
    if ((rc = calloc_dev(c, &cv.Wp, (size_t)total))) return rc;
    cv.phase_stride = total;
    hipLaunchKernelGGL(conv_pack_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, pwih, cv.Wp, 4 * H, H, 1, 0, 1, 0, total);
    L->Wih.push_back(cv);
  }
  CCHK(c, hipGetLastError());
  return VC_OK;
}

// out positions of a strided / plain convolution, padding as EncodecConv1d does (transformers modeling_encodec:
// padding_total = (Kw - 1) * dilation + 1 - stride; non-causal: right = total // 2, left = total - right;
// causal: everything on the left; the "extra" right padding that completes the last frame is padded the same way).
// x / out / res hold B items of L_in (T) positions back to back.
int run_conv(vc_codec* c, const Conv& cv, const float* x, int L_in, const float* res, float* out_raw, float* out_elu,
             int* L_out, hipStream_t s, int B = 1, int dil = 1) {
  const int pt = (cv.Kw - 1) * dil + 1 - cv.stride;
  const int pl = c->cfg.causal ? pt : pt - pt / 2;
  const int T = (L_in + cv.stride - 1) / cv.stride;
  const int prt = (T - 1) * cv.stride + (cv.Kw - 1) * dil + 1 - pl - L_in;      // right padding incl. the extra part
  // an input not longer than its reflect padding is zero-extended to max(pad) + 1 positions first, reflected, and the
  // extension cut off again (EncodecConv1d._pad1d = audiocraft pad1d): clips of 1-3 frames
  const int L_ext = (c->cfg.pad_reflect && L_in <= std::max(pl, prt)) ? std::max(pl, prt) + 1 : L_in;
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.Wp = cv.Wp; a.bias = cv.bias; a.res = res; a.out_raw = out_raw; a.out_elu = out_elu;
  a.L_in = L_in; a.T = T; a.Ci = cv.Ci; a.Co = cv.Co; a.Kw = cv.Kw;
  a.s_in = cv.stride; a.dil = dil; a.pad = pl; a.reflect = c->cfg.pad_reflect; a.L_ext = L_ext;
  a.s_out = 1; a.o_off = 0; a.L_dst = T; a.w_phase_stride = cv.phase_stride;
  a.nphase = 1; a.x_bstride = (long)L_in * cv.Ci; a.o_bstride = (long)T * cv.Co;
  hipLaunchKernelGGL(conv_gemm_k, dim3((T + 127) / 128, (cv.Co + 31) / 32, B), dim3(256), 0, s, a);
  CCHK(c, hipGetLastError());
  *L_out = T;
  return VC_OK;
}
// EncodecConvTranspose1d: full length (L-1)*s + 2s, trimmed by (left = total - right; right = total // 2, or
// everything when causal with trim_right_ratio = 1)
int run_convT(vc_codec* c, const Conv& cv, const float* x, int L_in, float* out_raw, float* out_elu, int* L_out, hipStream_t s,
              int B = 1) {
  const int st = cv.stride;
  const int pt = cv.Kw - st;
  const int pr = c->cfg.causal ? pt : pt / 2, pl = pt - pr;
  const int L_dst = L_in * st;
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.Wp = cv.Wp; a.bias = cv.bias; a.out_raw = out_raw; a.out_elu = out_elu;
  a.L_in = L_in; a.T = L_in + 1; a.Ci = cv.Ci; a.Co = cv.Co; a.Kw = 2;
  a.s_in = 1; a.dil = -1; a.pad = 0; a.reflect = 0;
  a.s_out = st; a.o_off = -pl; a.L_dst = L_dst; a.w_phase_stride = cv.phase_stride;
  a.nphase = st; a.x_bstride = (long)L_in * cv.Ci; a.o_bstride = (long)L_dst * cv.Co;
  hipLaunchKernelGGL(conv_gemm_k, dim3((a.T + 127) / 128, (cv.Co + 31) / 32, st * B), dim3(256), 0, s, a);
  CCHK(c, hipGetLastError());
  *L_out = L_dst;
  return VC_OK;
}

// A stage's residual units (EncodecResnetBlock x num_residual_layers).  In: raw tensor in *raw, its ELU in *elu.
// Out: *elu holds the ELU of the last unit's output (what the following ELU + conv consumes); *raw its raw value
// only when want_raw.  Buffers ping-pong between (A_raw, A_elu) and (C_raw, B_elu).
int run_res_units(vc_codec* c, const std::vector<ResUnit>& units, float** raw, float** elu, int L, bool want_raw,
                  hipStream_t s, int B) {
  int rc, Lo;
  for (size_t j = 0; j < units.size(); ++j) {
    const ResUnit& u = units[j];
    float* nraw = (*raw == c->A_raw) ? c->C_raw : c->A_raw;
    float* nelu = (*elu == c->A_elu) ? c->B_elu : c->A_elu;
    const bool keep_raw = want_raw || j + 1 < units.size();
    if ((rc = run_conv(c, u.c3, *elu, L, nullptr, nullptr, c->H_elu, &Lo, s, B, u.dil))) return rc;
    const float* res = *raw;
    if (u.has_sc) {                                       // shortcut = 1x1 conv of the block input (no activation)
      if ((rc = run_conv(c, u.sc, *raw, L, nullptr, c->S_raw, nullptr, &Lo, s, B))) return rc;
      res = c->S_raw;
    }
    if ((rc = run_conv(c, u.c1, c->H_elu, L, res, keep_raw ? nraw : nullptr, nelu, &Lo, s, B))) return rc;
    *raw = nraw; *elu = nelu;
  }
  return VC_OK;
}

// a linear layer over every position as a 1x1 implicit GEMM (the LSTM input projection)
int run_conv1x1(vc_codec* c, const Conv& cv, const float* x, int T, float* out_raw, hipStream_t s, int B) {
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.x = x; a.Wp = cv.Wp; a.bias = cv.bias; a.out_raw = out_raw;
  a.L_in = T; a.T = T; a.Ci = cv.Ci; a.Co = cv.Co; a.Kw = 1;
  a.s_in = 1; a.dil = 1; a.pad = 0; a.reflect = 0;
  a.s_out = 1; a.o_off = 0; a.L_dst = T; a.w_phase_stride = cv.phase_stride;
  a.nphase = 1; a.x_bstride = (long)T * cv.Ci; a.o_bstride = (long)T * cv.Co;
  hipLaunchKernelGGL(conv_gemm_k, dim3((T + 127) / 128, (cv.Co + 31) / 32, B), dim3(256), 0, s, a);
  CCHK(c, hipGetLastError());
  return VC_OK;
}

// persistent LSTM: after a synchronise, report a wait that gave up (word 1 of err_flag)
int check_lstm_flag(vc_codec* c) {
  if (!c->persist_used) return VC_OK;
  c->persist_used = false;
  int w = 0;
  CCHK(c, hipMemcpy(&w, c->err_flag + 1, 4, hipMemcpyDeviceToHost));
  if (w) {
    (void)hipMemset(c->err_flag + 1, 0, 4);
    return cfail(c, VC_EHIP, "persistent LSTM: a hand-off wait exceeded its bound (workgroups not co-resident?)");
  }
  return VC_OK;
}
// EncodecLSTM: y = lstm(x) + x over [B][T][H]; x is raw, the block output is written raw and/or ELU'd
int run_lstm(vc_codec* c, const Lstm& L, const float* x, int T, float* out_raw, float* out_elu, hipStream_t s, int B = 1) {
  const int H = L.H;
  const float* in = x;
  float* seq[2] = {c->HS0, c->HS1};
  if (L.layers == 2 && H % 256 == 0 && H <= 1024 && !getenv("VC_LSTM_SEQUENTIAL")) {
    // two-layer wavefront: T + 1 launches (lstm_wave_k)
    int rc = run_conv1x1(c, L.Wih[0], x, T, c->G, s, B);                          // layer 0: G = x W_ih^T + b_ih + b_hh
    if (rc) return rc;
    // one persistent cooperative launch (lstm_persist_k) when its 2 x H/8 workgroups are all resident at once;
    // VC_LSTM_WAVE=1 forces the launch-per-step wavefront (the reference form of the tests)
    if ((H == 512 || H == 1024) && !getenv("VC_LSTM_WAVE")) {
      const void* kern = H == 512 ? (const void*)lstm_persist_k<2> : (const void*)lstm_persist_k<4>;
      const long n_wg = 2L * (H / 8);
      if (c->persist_ok < 0) {
        int per_cu = 0, coop = 0;
        hipDeviceProp_t prop;
        CCHK(c, hipGetDeviceProperties(&prop, c->device));
        CCHK(c, hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, c->device));
        const size_t lds_max = (size_t)2 * VC_LSTM_BG * 2 * H * sizeof(float);
        CCHK(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        CCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds_max));
        c->persist_ok = (coop && (long)per_cu * prop.multiProcessorCount >= n_wg) ? 1 : 0;
      }
      if (c->persist_ok) {
        if (!c->hgran) {
          int rc2 = calloc_dev(c, &c->hgran, (size_t)2 * c->B_max * c->T_max * H);
          if (rc2) return rc2;
          CCHK(c, hipMemsetAsync(c->hgran, 0, (size_t)2 * c->B_max * c->T_max * H * 8, s));
        }
        if (++c->lstm_epoch == 0) c->lstm_epoch = 1;
        LstmPersistArgs pa;
        memset(&pa, 0, sizeof pa);
        pa.Whh[0] = L.Whh[0]; pa.Whh[1] = L.Whh[1]; pa.Wih1 = L.WihP[1]; pa.b1 = L.bP[1]; pa.G0 = c->G;
        pa.hg[0] = c->hgran; pa.hg[1] = c->hgran + (size_t)c->B_max * c->T_max * H;
        pa.c[0] = c->cstate; pa.c[1] = c->cstate + (size_t)B * H;
        pa.skip = x; pa.out_raw = out_raw; pa.out_elu = out_elu; pa.H = H; pa.T = T; pa.B = B;
        pa.epoch = c->lstm_epoch; pa.err = c->err_flag + 1;      // word 1: a bounded wait gave up
        void* kargs[] = {&pa};
        const size_t lds = (size_t)2 * std::min(B, VC_LSTM_BG) * 2 * H * sizeof(float);
        CCHK(c, hipEventRecord(c->ev_l[0], s));
        CCHK(c, hipLaunchCooperativeKernel(kern, dim3((unsigned)n_wg), dim3(512), kargs, (unsigned)lds, s));
        CCHK(c, hipEventRecord(c->ev_l[1], s));
        c->persist_used = true;
        c->last_lstm_persist = 1;
        return VC_OK;
      }
    }
    c->last_lstm_persist = 0;
    CCHK(c, hipMemsetAsync(c->cstate, 0, (size_t)2 * B * H * 4, s));
    LstmWaveArgs a;
    memset(&a, 0, sizeof a);
    a.Whh[0] = L.Whh[0]; a.Whh[1] = L.Whh[1]; a.Wih1 = L.WihP[1]; a.b1 = L.bP[1]; a.G0 = c->G;
    a.hs[0] = seq[0]; a.hs[1] = seq[1]; a.c[0] = c->cstate; a.c[1] = c->cstate + (size_t)B * H; a.hzero = c->hzero;
    a.skip = x; a.out_raw = out_raw; a.out_elu = out_elu; a.H = H; a.T = T; a.B = B;
    const dim3 grid(H / 4, 2);
    CCHK(c, hipEventRecord(c->ev_l[0], s));
    for (int k = 0; k <= T; ++k) {
      a.k = k;
      if (H == 256) hipLaunchKernelGGL(lstm_wave_k<1>, grid, dim3(256), 0, s, a);
      else if (H == 512) hipLaunchKernelGGL(lstm_wave_k<2>, grid, dim3(256), 0, s, a);
      else if (H == 768) hipLaunchKernelGGL(lstm_wave_k<3>, grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL(lstm_wave_k<4>, grid, dim3(256), 0, s, a);
    }
    CCHK(c, hipGetLastError());
    CCHK(c, hipEventRecord(c->ev_l[1], s));
    return VC_OK;
  }
  if (B != 1) return cfail(c, VC_EINVAL, "batched LSTM needs the two-layer wavefront (2 layers, hidden a multiple of 256 <= 1024)");
  for (int n = 0; n < L.layers; ++n) {
    int rc = run_conv1x1(c, L.Wih[n], in, T, c->G, s, 1);                        // G = in W_ih^T + b_ih + b_hh
    if (rc) return rc;
    CCHK(c, hipMemsetAsync(c->cstate, 0, (size_t)H * 4, s));
    float* hs = seq[n & 1];
    const bool last = (n == L.layers - 1);
    for (int t = 0; t < T; ++t) {
      LstmArgs a;
      memset(&a, 0, sizeof a);
      a.Whh = L.Whh[n]; a.G = c->G + (size_t)t * 4 * H; a.h_prev = t ? hs + (size_t)(t - 1) * H : c->hzero;
      a.c = c->cstate; a.h_out = hs + (size_t)t * H; a.H = H;
      if (last) {
        a.skip = x + (size_t)t * H;
        a.out_raw = out_raw ? out_raw + (size_t)t * H : nullptr;
        a.out_elu = out_elu ? out_elu + (size_t)t * H : nullptr;
      }
      hipLaunchKernelGGL(lstm_step_k, dim3((H + 3) / 4), dim3(256), (size_t)H * 4, s, a);
    }
    CCHK(c, hipGetLastError());
    in = hs;
  }
  return VC_OK;
}
}  // namespace

extern "C" const char* vc_codec_last_error(const vc_codec* c) { return c ? c->err.c_str() : g_codec_create_err.c_str(); }

extern "C" int vc_codec_create(const vc_codec_cfg* cfg, int hip_device, vc_codec** out) {
  if (!cfg || !out) return cfail(nullptr, VC_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->n_ratios < 1 || cfg->n_ratios > VC_CODEC_MAX_RATIOS) return cfail(nullptr, VC_EINVAL, "n_ratios out of range");
  if (cfg->n_filters % 32 || cfg->hidden % 16) return cfail(nullptr, VC_EINVAL, "n_filters must be a multiple of 32, hidden of 16");
  if (cfg->compress != 2) return cfail(nullptr, VC_EINVAL, "compress must be 2");
  if (cfg->n_q < 1 || cfg->n_q > VC_MAX_CODEBOOKS) return cfail(nullptr, VC_EINVAL, "n_q out of range");
  if (cfg->max_samples < 1) return cfail(nullptr, VC_EINVAL, "max_samples must be positive");
  if (cfg->num_residual_layers < 1 || cfg->num_residual_layers > 4) return cfail(nullptr, VC_EINVAL, "num_residual_layers must be in [1,4]");
  if (cfg->dilation_growth_rate < 1 || cfg->dilation_growth_rate > 4) return cfail(nullptr, VC_EINVAL, "dilation_growth_rate must be in [1,4]");
  if (cfg->max_batch < 1 || cfg->max_batch > 64) return cfail(nullptr, VC_EINVAL, "max_batch must be in [1,64]");
  hipError_t e = hipSetDevice(hip_device);
  if (e != hipSuccess) return cfail(nullptr, VC_EHIP, "hipSetDevice(%d): %s", hip_device, hipGetErrorString(e));
  vc_codec* c = new vc_codec();
  c->cfg = *cfg; c->device = hip_device;
  c->hop = 1;
  for (int i = 0; i < cfg->n_ratios; ++i) c->hop *= cfg->ratios[i];
  *out = c;
  return VC_OK;
}

extern "C" void vc_codec_destroy(vc_codec* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (auto& kv : c->raw) if (kv.second.second) (void)hipFree(kv.second.second);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->h_flag) (void)hipHostFree(c->h_flag);
  for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_l) if (ev) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

extern "C" int vc_codec_load_tensor(vc_codec* c, const char* key, const void* data, int on_device,
                                    const int64_t* shape, int ndim) {
  if (!c || !key || !data || ndim < 1 || ndim > 3) return cfail(c, VC_EINVAL, "bad argument to vc_codec_load_tensor");
  if (c->finalized) return cfail(c, VC_ESTATE, "codec already finalized");
  CCHK(c, hipSetDevice(c->device));
  std::vector<int64_t> sh(shape, shape + ndim);
  long n = 1;
  for (auto v : sh) n *= v;
  if (n <= 0) return cfail(c, VC_EINVAL, "empty tensor '%s'", key);
  auto it = c->raw.find(key);
  if (it != c->raw.end()) { (void)hipFree(it->second.second); c->raw.erase(it); }
  float* d = nullptr;
  CCHK(c, hipMalloc((void**)&d, (size_t)n * 4));
  CCHK(c, hipMemcpy(d, data, (size_t)n * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  c->raw[key] = {sh, d};
  return VC_OK;
}

extern "C" int vc_codec_finalize(vc_codec* c) {
  if (!c) return VC_EINVAL;
  if (c->finalized) return cfail(c, VC_ESTATE, "codec already finalized");
  CCHK(c, hipSetDevice(c->device));
  const vc_codec_cfg& g = c->cfg;
  const int F = g.n_filters, R = g.n_ratios;
  int rc;
  // ---- encoder: module indices as transformers.EncodecEncoder enumerates them
  const int NR = g.num_residual_layers;
  auto make_unit = [&](const std::string& rb, int ch, int j, ResUnit* u) -> int {
    int r;
    u->dil = 1;
    for (int q = 0; q < j; ++q) u->dil *= g.dilation_growth_rate;              // dilation_growth_rate ** j
    if ((r = make_conv(c, rb + "block.1.conv", ch, ch / g.compress, g.residual_kernel_size, 1, 0, &u->c3, true))) return r;
    if ((r = make_conv(c, rb + "block.3.conv", ch / g.compress, ch, 1, 1, 0, &u->c1, true))) return r;
    u->has_sc = g.conv_shortcut != 0;
    if (u->has_sc && (r = make_conv(c, rb + "shortcut.conv", ch, ch, 1, 1, 0, &u->sc, true))) return r;
    return VC_OK;
  };
  int idx = 0;
  if ((rc = make_conv(c, "encoder.layers.0.conv", 1, F, g.kernel_size, 1, 0, &c->enc_first, false))) return rc;
  idx = 1;
  int ch = F;
  c->enc_res.assign(R, std::vector<ResUnit>(NR)); c->enc_down.resize(R);
  for (int i = 0; i < R; ++i) {
    const int ratio = g.ratios[R - 1 - i];               // reversed(upsampling_ratios)
    for (int j = 0; j < NR; ++j) {
      if ((rc = make_unit("encoder.layers." + std::to_string(idx) + ".", ch, j, &c->enc_res[i][j]))) return rc;
      idx += 1;                                          // one module per resblock
    }
    idx += 1;                                            // ELU
    if ((rc = make_conv(c, "encoder.layers." + std::to_string(idx) + ".conv", ch, ch * 2, ratio * 2, ratio, 0, &c->enc_down[i], true))) return rc;
    idx += 1;
    ch *= 2;
  }
  if ((rc = make_lstm(c, "encoder.layers." + std::to_string(idx) + ".lstm", ch, g.lstm_layers, &c->enc_lstm))) return rc;
  idx += 2;                                              // LSTM, ELU
  if ((rc = make_conv(c, "encoder.layers." + std::to_string(idx) + ".conv", ch, g.hidden, g.last_kernel_size, 1, 0, &c->enc_last, true))) return rc;
  const int top = ch;
  // ---- decoder
  if ((rc = make_conv(c, "decoder.layers.0.conv", g.hidden, top, g.kernel_size, 1, 0, &c->dec_first, true))) return rc;
  if ((rc = make_lstm(c, "decoder.layers.1.lstm", top, g.lstm_layers, &c->dec_lstm))) return rc;
  idx = 2;
  ch = top;
  c->dec_up.resize(R); c->dec_res.assign(R, std::vector<ResUnit>(NR));
  for (int i = 0; i < R; ++i) {
    const int ratio = g.ratios[i];
    idx += 1;                                            // ELU
    if ((rc = make_conv(c, "decoder.layers." + std::to_string(idx) + ".conv", ch, ch / 2, ratio * 2, ratio, 1, &c->dec_up[i], true))) return rc;
    idx += 1;
    for (int j = 0; j < NR; ++j) {
      if ((rc = make_unit("decoder.layers." + std::to_string(idx) + ".", ch / 2, j, &c->dec_res[i][j]))) return rc;
      idx += 1;
    }
    ch /= 2;
  }
  idx += 1;                                              // ELU
  if ((rc = make_conv(c, "decoder.layers." + std::to_string(idx) + ".conv", F, 1, g.last_kernel_size, 1, 0, &c->dec_last, false))) return rc;
  // ---- codebooks
  const int D = g.hidden, C = g.codebook_size, Q = g.n_q;
  if ((rc = calloc_dev(c, &c->E, (size_t)Q * C * D))) return rc;
  if ((rc = calloc_dev(c, &c->Et, (size_t)Q * C * D))) return rc;
  if ((rc = calloc_dev(c, &c->e2, (size_t)Q * C))) return rc;
  for (int q = 0; q < Q; ++q) {
    float* src;
    if ((rc = get_raw(c, "quantizer.layers." + std::to_string(q) + ".codebook.embed", {C, D}, &src))) return rc;
    CCHK(c, hipMemcpy(c->E + (size_t)q * C * D, src, (size_t)C * D * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(transpose_k, dim3((unsigned)(((long)C * D + 255) / 256)), dim3(256), 0, 0, src, c->Et + (size_t)q * C * D, C, D);
  }
  hipLaunchKernelGGL(rownorm2_k, dim3((Q * C + 255) / 256), dim3(256), 0, 0, c->E, c->e2, Q * C, D);
  CCHK(c, hipGetLastError());
  // ---- activation arenas
  const size_t N = (size_t)g.max_samples;
  const size_t NB = (size_t)g.max_batch;
  c->B_max = g.max_batch;
  c->T_max = (int)((N + c->hop - 1) / c->hop) + 1;
  const size_t big = NB * std::max(N * F, (size_t)c->T_max * top) + 1024;
  if ((rc = calloc_dev(c, &c->A_raw, big))) return rc;
  if ((rc = calloc_dev(c, &c->A_elu, big))) return rc;
  if ((rc = calloc_dev(c, &c->B_elu, big))) return rc;
  if ((rc = calloc_dev(c, &c->H_elu, big))) return rc;
  if (NR > 1 && (rc = calloc_dev(c, &c->C_raw, big))) return rc;
  if (g.conv_shortcut && (rc = calloc_dev(c, &c->S_raw, big))) return rc;
  if ((rc = calloc_dev(c, &c->latent, NB * c->T_max * D))) return rc;
  if ((rc = calloc_dev(c, &c->G, NB * c->T_max * 4 * top))) return rc;
  if ((rc = calloc_dev(c, &c->HS0, NB * c->T_max * top))) return rc;
  if ((rc = calloc_dev(c, &c->HS1, NB * c->T_max * top))) return rc;
  if ((rc = calloc_dev(c, &c->cstate, NB * 2 * top))) return rc;
  if ((rc = calloc_dev(c, &c->hzero, (size_t)top))) return rc;
  if ((rc = calloc_dev(c, &c->err_flag, (size_t)4))) return rc;
  CCHK(c, hipMemset(c->hzero, 0, (size_t)top * 4));
  CCHK(c, hipMemset(c->err_flag, 0, 16));
  CCHK(c, hipHostMalloc((void**)&c->h_flag, 64));
  for (auto& ev : c->ev) CCHK(c, hipEventCreate(&ev));
  for (auto& ev : c->ev_l) CCHK(c, hipEventCreate(&ev));
  CCHK(c, hipStreamCreate(&c->own_stream));
  CCHK(c, hipDeviceSynchronize());
  // the packed images are all that is needed of the conv weights; first/last convs and the LSTM keep their raw rows
  c->finalized = true;
  return VC_OK;
}

extern "C" int vc_codec_encode_batch(vc_codec* c, const float* wav_dev, int B, int n_samples, int64_t* codes_dev,
                                     int codes_cap, int* n_frames, void* stream) {
  if (!c || !c->finalized) return cfail(c, VC_ESTATE, "codec not finalized");
  if (!wav_dev || !codes_dev || !n_frames) return cfail(c, VC_EINVAL, "null argument to vc_codec_encode");
  if (B < 1 || B > c->B_max) return cfail(c, VC_ECAP, "batch %d outside [1, %d]", B, c->B_max);
  if (n_samples < 1 || n_samples > c->cfg.max_samples)
    return cfail(c, VC_ECAP, "n_samples %d outside [1, %d]", n_samples, c->cfg.max_samples);
  CCHK(c, hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
  const vc_codec_cfg& g = c->cfg;
  const int F = g.n_filters;
  CCHK(c, hipEventRecord(c->ev[0], s));
  int L = n_samples;
  {
    const long tot = (long)B * L * (F / 4);
    const int pt = g.kernel_size - 1;
    const int pad = g.causal ? pt : pt - pt / 2;
    const int mp = std::max(pad, pt - pad);
    hipLaunchKernelGGL(conv_first_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, wav_dev, c->enc_first.w_raw,
                       c->enc_first.bias, c->A_raw, c->A_elu, L, F, g.kernel_size, pad, g.pad_reflect, B,
                       (g.pad_reflect && L <= mp) ? mp + 1 : L);
  }
  int rc, Lo;
  const int R = g.n_ratios;
  float *raw = c->A_raw, *elu = c->A_elu;
  for (int i = 0; i < R; ++i) {
    if ((rc = run_res_units(c, c->enc_res[i], &raw, &elu, L, false, s, B))) return rc;
    const bool last = (i == R - 1);
    float* nelu = (elu == c->A_elu) ? c->B_elu : c->A_elu;
    // the down-sampling conv writes the next stage's raw input (A_raw) and, unless the LSTM follows, its ELU
    if ((rc = run_conv(c, c->enc_down[i], elu, L, nullptr, c->A_raw, last ? nullptr : nelu, &Lo, s, B))) return rc;
    raw = c->A_raw; elu = nelu;
    L = Lo;
  }
  const int T = L;
  if (T > codes_cap) return cfail(c, VC_ECAP, "codes capacity %d < %d frames", codes_cap, T);
  if ((rc = run_lstm(c, c->enc_lstm, c->A_raw, T, nullptr, c->B_elu, s, B))) return rc;
  if ((rc = run_conv(c, c->enc_last, c->B_elu, T, nullptr, c->latent, nullptr, &Lo, s, B))) return rc;
  if (g.hidden % 16 == 0 && g.hidden <= 256 && g.codebook_size % 16 == 0 && !getenv("VC_RVQ_SCALAR")) {
    const size_t lds = (size_t)16 * (g.hidden * 4 + 16) + (16 + 256 + 256 + 16) * 4;
    hipLaunchKernelGGL(rvq_encode_mfma_k, dim3((B * T + 15) / 16), dim3(256), lds, s, c->latent, c->E, c->e2, codes_dev, B * T, T,
                       g.hidden, g.codebook_size, g.n_q);
  } else {
    hipLaunchKernelGGL(rvq_encode_k, dim3(B * T), dim3(256), (size_t)g.hidden * 4, s, c->latent, c->Et, c->E, c->e2, codes_dev, T,
                       g.hidden, g.codebook_size, g.n_q);
  }
  CCHK(c, hipGetLastError());
  CCHK(c, hipEventRecord(c->ev[1], s));
  CCHK(c, hipStreamSynchronize(s));
  CCHK(c, hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
  if ((rc = check_lstm_flag(c))) return rc;
  c->last_T = T;
  *n_frames = T;
  return VC_OK;
}
extern "C" int vc_codec_encode(vc_codec* c, const float* wav_dev, int n_samples, int64_t* codes_dev, int codes_cap,
                               int* n_frames, void* stream) {
  return vc_codec_encode_batch(c, wav_dev, 1, n_samples, codes_dev, codes_cap, n_frames, stream);
}

extern "C" int vc_codec_decode_batch(vc_codec* c, const int64_t* codes_dev, int B, int T, float* wav_dev, int wav_cap,
                                     void* stream) {
  if (!c || !c->finalized) return cfail(c, VC_ESTATE, "codec not finalized");
  if (!codes_dev || !wav_dev) return cfail(c, VC_EINVAL, "null argument to vc_codec_decode");
  if (B < 1 || B > c->B_max) return cfail(c, VC_ECAP, "batch %d outside [1, %d]", B, c->B_max);
  if (T < 1 || T > c->T_max - 1) return cfail(c, VC_ECAP, "T %d outside [1, %d]", T, c->T_max - 1);
  if ((long)T * c->hop > wav_cap) return cfail(c, VC_ECAP, "wav capacity %d < %ld", wav_cap, (long)T * c->hop);
  CCHK(c, hipSetDevice(c->device));
  hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
  const vc_codec_cfg& g = c->cfg;
  CCHK(c, hipEventRecord(c->ev[0], s));
  {
    const long tot = (long)B * T * g.hidden;
    hipLaunchKernelGGL(rvq_decode_k, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, codes_dev, c->E, c->latent, T, g.hidden,
                       g.codebook_size, g.n_q, c->err_flag, B);
  }
  int rc, Lo, L = T;
  if ((rc = run_conv(c, c->dec_first, c->latent, L, nullptr, c->A_raw, nullptr, &Lo, s, B))) return rc;
  if ((rc = run_lstm(c, c->dec_lstm, c->A_raw, L, nullptr, c->B_elu, s, B))) return rc;
  float* elu = c->B_elu;
  for (int i = 0; i < g.n_ratios; ++i) {
    float* nelu = (elu == c->A_elu) ? c->B_elu : c->A_elu;
    if ((rc = run_convT(c, c->dec_up[i], elu, L, c->A_raw, nelu, &Lo, s, B))) return rc;
    L = Lo;
    float* raw = c->A_raw;
    elu = nelu;
    if ((rc = run_res_units(c, c->dec_res[i], &raw, &elu, L, false, s, B))) return rc;
  }
  {
    const int Ci = g.n_filters, Kw = g.last_kernel_size;
    const int pt = Kw - 1;
    const int pad = g.causal ? pt : pt - pt / 2;
    const int mp = std::max(pad, pt - pad);
    hipLaunchKernelGGL(conv_last_k, dim3((L + 255) / 256, B), dim3(256), (size_t)Kw * Ci * 4, s, elu, c->dec_last.w_raw,
                       c->dec_last.bias, wav_dev, L, Ci, Kw, pad, g.pad_reflect, (g.pad_reflect && L <= mp) ? mp + 1 : L);
  }
  CCHK(c, hipGetLastError());
  CCHK(c, hipEventRecord(c->ev[1], s));
  CCHK(c, hipMemcpyAsync(c->h_flag, c->err_flag, 4, hipMemcpyDeviceToHost, s));
  CCHK(c, hipStreamSynchronize(s));
  CCHK(c, hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
  if (*c->h_flag) {
    (void)hipMemset(c->err_flag, 0, 4);   // already on the error path
    return cfail(c, VC_EINVAL, "code index outside [0, %d)", g.codebook_size);
  }
  return check_lstm_flag(c);
}
extern "C" int vc_codec_decode(vc_codec* c, const int64_t* codes_dev, int T, float* wav_dev, int wav_cap, void* stream) {
  return vc_codec_decode_batch(c, codes_dev, 1, T, wav_dev, wav_cap, stream);
}

extern "C" int vc_codec_debug_latent(vc_codec* c, float* host_dst, int64_t n_floats) {
  if (!c || !c->finalized || !host_dst) return VC_EINVAL;
  if (n_floats > (int64_t)c->B_max * c->T_max * c->cfg.hidden) return cfail(c, VC_ECAP, "latent holds %lld floats", (long long)c->B_max * c->T_max * c->cfg.hidden);
  CCHK(c, hipDeviceSynchronize());
  CCHK(c, hipMemcpy(host_dst, c->latent, (size_t)n_floats * 4, hipMemcpyDeviceToHost));
  return VC_OK;
}

extern "C" int vc_codec_last_ms(const vc_codec* c, float* ms) {
  if (!c || !ms) return VC_EINVAL;
  *ms = c->last_ms;
  return VC_OK;
}
extern "C" int vc_codec_last_lstm_ms(vc_codec* c, float* ms, double* bytes_per_step) {
  if (!c || !ms) return VC_EINVAL;
  if (hipEventElapsedTime(&c->last_lstm_ms, c->ev_l[0], c->ev_l[1]) != hipSuccess) c->last_lstm_ms = 0;
  *ms = c->last_lstm_ms;
  if (bytes_per_step) *bytes_per_step = 3.0 * 4.0 * c->enc_lstm.H * c->enc_lstm.H * 4.0;    // W_hh0, W_hh1, W_ih1 in fp32
  return VC_OK;
}
