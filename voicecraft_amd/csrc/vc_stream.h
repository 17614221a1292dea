// vc_stream.h — argument blocks of the persistent batch-1 decode step (vc_stream.hip), shared with the engine.
#pragma once
#include "vc_common.h"

struct StreamLayerDev {                // per layer, device memory
  const float *wg_qkv, *b_qkv;         // LN-folded QKV: row sums of W.gamma, W beta + b   [3d]
  const float* b_o;                    // [d]
  const float *wg_1, *b_1;             // LN-folded FFN-up                                  [4d]
  const float* b_2;                    // [d]
  void *kc, *vc;                       // KV cache of the layer (bf16 [seq][H][S_max][hd])
};

struct StreamArgs {
  const uint4* Ws;                     // stream weights [L][G][spl][1024] x 16 B
  const StreamLayerDev* layers;
  unsigned long long* gran;            // granule arena, per layer: see the SG_G_* offsets below
  long gran_layer_stride;
  unsigned* ctl;                       // [0] epoch (bumped by workgroup 0 at the end of a step), [1] error
  const float* h_in;                   // row 0 of dec_h: the new token's embedding
  float* h_out;                        // row 0 of hB: the last layer's finished residual (feeds the heads)
  const int* n_active;
  const int* row_pos;
  const int* row_seq;
  int d, H, hd, L, G, NS, S_max;
  long cache_seq_stride;
  int rpc;                             // 1 KB rows per K = d channel (d / 512)
  int sq, so, s1, s2, spl;             // slots per op and per layer
  float scale;
  long long* ts;                       // optional [3][L][16] phase stamps (100 MHz) of workgroups 0, 1 and G-1 (tools/stream_probe.py)
  float* dbg;                          // optional [L][5][4d] dump of the op outputs as the kernel saw its INPUTS (parity bisect)
};
// granule offsets inside a layer's arena (in granules)
__host__ __device__ inline long sg_off_h(const StreamArgs& a) { return 0; }                                   // [d]  residual entering the layer (layers >= 1)
__host__ __device__ inline long sg_off_q(const StreamArgs& a) { return a.d; }                                 // [d]
__host__ __device__ inline long sg_off_kv(const StreamArgs& a) { return 2L * a.d; }                           // [2d] new k, new v
__host__ __device__ inline long sg_off_p(const StreamArgs& a) { return 4L * a.d; }                            // [G][hd + 2] attention partials
__host__ __device__ inline long sg_off_o(const StreamArgs& a) { return 4L * a.d + (long)a.G * (a.hd + 2); }  // [d]  merged heads
__host__ __device__ inline long sg_off_h2(const StreamArgs& a) { return sg_off_o(a) + a.d; }                  // [d]  residual after attention
__host__ __device__ inline long sg_off_a(const StreamArgs& a) { return sg_off_h2(a) + a.d; }                  // [2d] activations, two bf16 per granule
__host__ __device__ inline long sg_gran_per_layer(const StreamArgs& a) { return sg_off_a(a) + 2L * a.d; }


hipError_t vc_stream_pack_layer(const float* Wqkv, const float* Wo, const float* W1, const float* W2, const float* g1,
                                const float* g2, void* dst, int d, int G, hipStream_t s);
size_t vc_stream_layer_bytes(int d, int G);
hipError_t vc_stream_launch(const StreamArgs& a, hipStream_t s);

// ---- weight prefetcher of the launch path (weight_prefetch_k)
struct PrefetchSeg { const char* ptr; size_t bytes; };   // one weight matrix (packed image), in launch order
struct PrefetchArgs {
  const PrefetchSeg* segs;   // device array [n_seg]: the 4 L + 2 matrices of a decode step
  int n_seg;
  unsigned* prog;            // index + 1 of the matrix the running launch reads (stored by the GEMM kernels; 0 before the first)
  const int* n_active;       // 0 once the last sequence retired
  const int* stop;           // pinned host word: raised by the host when the call ends early
  int ahead;                 // matrices the prefetcher may run ahead of the launches
  int G;                     // workgroups (one per CU)
};
hipError_t vc_prefetch_launch(const PrefetchArgs& a, hipStream_t s);
