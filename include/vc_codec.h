/*
 * vc_codec.h — C ABI of the EnCodec (SEANet + LSTM + RVQ) encode/decode path in libvcengine.so.
 *
 * Replaces AudioTokenizer.encode / .decode of the reference (data/tokenizer.py:127-133), which call
 * audiocraft's EncodecModel (audiocraft @ c5157b5, not vendored in the reference tree).  The
 * architecture restated here is the one `transformers.EncodecModel` implements for the VoiceCraft
 * codec shape: 16 kHz mono, 64 base filters, strides 2/4/5/8 (hop 320, 50 Hz),
 * 1 residual unit per stage, 2-layer LSTM with skip, weight-normalised convolutions, ELU,
 * reflect padding, non-causal, RVQ of n_q x codebook_size x hidden.
 *
 * Conventions are those of vc_engine.h: plain C, opaque handle, device pointers, a hipStream_t as
 * void*, 0 / negative VC_E* return codes with vc_codec_last_error().
 */
#ifndef VC_CODEC_H
#define VC_CODEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_CODEC_MAX_RATIOS 8

typedef struct vc_codec vc_codec;

typedef struct vc_codec_cfg {
  int32_t sample_rate;          /* 16000 (informational)                               */
  int32_t n_filters;            /* 64                                                  */
  int32_t n_ratios;             /* 4                                                   */
  int32_t ratios[VC_CODEC_MAX_RATIOS]; /* decoder (upsampling) order: 8,5,4,2           */
  int32_t hidden;               /* 128: latent / codebook dimension                    */
  int32_t n_q;                  /* 4 quantizers (K)                                    */
  int32_t codebook_size;        /* 2048                                                */
  int32_t lstm_layers;          /* 2                                                   */
  int32_t kernel_size;          /* 7: first conv                                       */
  int32_t last_kernel_size;     /* 7                                                   */
  int32_t residual_kernel_size; /* 3                                                   */
  int32_t compress;             /* 2: residual unit hidden = dim / compress            */
  int32_t max_samples;          /* capacity: longest waveform (samples) per call       */
  /* Architecture switches of the SEANet stacks.  Which values the reference's checkpoint
   * (audiocraft encodec_4cb2048_giga.th, data/tokenizer.py:109-110) was trained with cannot be read from the
   * reference tree, so they are configuration, named as in transformers.EncodecConfig: */
  int32_t causal;               /* use_causal_conv: all padding on the left, transposed convs trimmed on the right */
  int32_t pad_reflect;          /* pad_mode: 1 = "reflect", 0 = "constant" (zeros)     */
  int32_t conv_shortcut;        /* use_conv_shortcut: 1x1 conv on the residual path (else identity) */
  int32_t num_residual_layers;  /* residual units per stage (1)                        */
  int32_t dilation_growth_rate; /* unit j dilates its first conv by rate**j (2)        */
  int32_t max_batch;            /* capacity: clips per vc_codec_encode_batch / decode_batch call */
} vc_codec_cfg;

int vc_codec_create(const vc_codec_cfg* cfg, int hip_device, vc_codec** out);
void vc_codec_destroy(vc_codec* c);
const char* vc_codec_last_error(const vc_codec* c);

/* Effective fp32 tensors (weight norm already folded: w = g * v / ||v||), keyed like the
 * transformers.EncodecModel modules:
 *   encoder.layers.{i}.conv.{weight,bias}            Conv1d        [Co][Ci][Kw]
 *   encoder.layers.{i}.block.{1,3}.conv.{weight,bias}
 *   {encoder,decoder}.layers.{i}.lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{n}
 *   decoder.layers.{i}.conv.{weight,bias}            Conv1d or ConvTranspose1d [Ci][Co][Kw]
 *   quantizer.layers.{q}.codebook.embed              [codebook_size][hidden]               */
int vc_codec_load_tensor(vc_codec* c, const char* key, const void* data, int on_device,
                         const int64_t* shape, int ndim);
int vc_codec_finalize(vc_codec* c);

/* AudioTokenizer.encode (data/tokenizer.py:127-129): wav fp32 [n_samples] -> codes int64 [K][T],
 * T = ceil(n_samples / hop).  *n_frames receives T. */
int vc_codec_encode(vc_codec* c, const float* wav_dev, int n_samples, int64_t* codes_dev,
                    int codes_cap, int* n_frames, void* stream);
/* AudioTokenizer.decode (data/tokenizer.py:131-133): codes int64 [K][T] -> wav fp32 [hop*T]. */
int vc_codec_decode(vc_codec* c, const int64_t* codes_dev, int T, float* wav_dev, int wav_cap,
                    void* stream);
/* Bulk forms: the reference's dataset encoder pushes a zero-padded batch [B,1,N] through ONE model.encode call
 * (data/phonemize_encodec_encode_hf.py:186-206) and AudioTokenizer.encode/decode take [B,...] tensors.
 *   encode_batch: wav fp32 [B][n_samples] (already padded to a common length) -> codes int64 [B][K][T]
 *   decode_batch: codes int64 [B][K][T] -> wav fp32 [B][hop*T]
 * Every item gives bit for bit what the single-clip call gives on the same (padded) row: the batch is an extra
 * grid dimension of the convolutions and the LSTM advances all B sequences per launch (one read of the
 * recurrence weights for the whole batch).  codes_cap / wav_cap are per item. */
int vc_codec_encode_batch(vc_codec* c, const float* wav_dev, int B, int n_samples, int64_t* codes_dev,
                          int codes_cap, int* n_frames, void* stream);
int vc_codec_decode_batch(vc_codec* c, const int64_t* codes_dev, int B, int T, float* wav_dev, int wav_cap,
                          void* stream);
/* Test hooks: the latent before quantisation ([T][hidden], channels-last) of the last encode,
 * and its timing (HIP events on the stream). */
int vc_codec_debug_latent(vc_codec* c, float* host_dst, int64_t n_floats);
int vc_codec_last_ms(const vc_codec* c, float* ms);
/* Duration of the LSTM recurrence (T+1 wavefront launches) of the last call and the weight bytes one launch reads. */
int vc_codec_last_lstm_ms(vc_codec* c, float* ms, double* bytes_per_step);

#ifdef __cplusplus
}
#endif
#endif /* VC_CODEC_H */
