/*
 * talkshow_hip_debug.h — measurement, tuning and test aids of libtalkshow_hip.so.
 *
 * NOT part of the drop-in surface (include/talkshow_hip.h): nothing here has a counterpart in the reference, no product code under
 * nets/ or evaluation/ calls these, and they may change between rounds.  Clients: tools/ (profiling / A-B scripts) and tests/
 * (host-side layout and launch-plan checks, tile-shape agreement, the shader-clock sampler).  Same conventions as the main header
 * (0 on success unless stated, ts_last_error() for the message).
 */
#ifndef TALKSHOW_HIP_DEBUG_H
#define TALKSHOW_HIP_DEBUG_H

#include "talkshow_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Like ts_stream_create, but kernels of this stream only run on compute units [cu_first, cu_first+cu_count) of the
 * device's CU-mask index space (hipExtStreamCreateWithCUMask).  No reference counterpart. */
int ts_stream_create_cus(ts_ctx *ctx, int cu_first, int cu_count, void **out_stream);
/* Tuning aid: with TS_SKINNY_TRACE=1 the chain kernel stamps the device wall clock (100 MHz) at five points; this reads
 * (and resets) the records, 6 uint64 each.  Returns the number of records or -1. */
int ts_debug_skinny_trace(unsigned long long *out, int max_records);
/* Measurement aid: launches a one-wave kernel on `stream` that, every window_us for n windows, writes three uint64 to dev_out
 * (device memory, 3 n values): wall-clock ticks (100 MHz) since its start, ticks of this window, shader-clock cycles of this
 * window — the clock the chip actually sustains while other streams load it (tools/conv_clock.py).  No reference counterpart. */
int ts_debug_clock_sample(unsigned long long *dev_out, int n, int window_us, void *stream);
/* Host-only helper (no GPU needed): how `conv_gemm_f32` launches an (M rows x N columns, `groups` problems) layer that takes 128 x 128
 * tiles — out4 = {row blocks tiled 128 x 128, row blocks tiled 64 x 128, workgroups of the first band, workgroups in all}.  Returns 1 if the
 * layer is launched in two bands (more than one round of 512 resident workgroups, not a whole number of rounds), 0 for a plain grid,
 * -1 on a bad argument.  No reference counterpart. */
int ts_debug_conv_bands(int M, int N, int groups, int *out4);
/* Host-only: the tile {row tile, column tile} workgroup `bid` of conv_gemm_split's XCD-aware 1-D grid works on (MT x NT tiles, column groups of
 * `gw`); 1 = out2 filled, 0 = that workgroup has no tile, -1 = bad argument. */
int ts_debug_split_tile(int bid, int MT, int NT, int gw, int *out2);
/* Host-only helper (no GPU needed): the TILED copy of a row-major weight matrix W[N][ldw] (K columns used) that the
 * PixelCNN chain kernel multiplies with — every 16-column x 16-k operand fragment one contiguous KB in lane order
 * (DESIGN.md §3/§4); epi 0 = linear column order, 1 = gate (8 tanh channels + their 8 sigmoid partners per tile,
 * gateD channels per half).  out holds ceil(N/16) * (K/16) * 256 floats.  K % 16 == 0.  No reference counterpart. */
int ts_debug_tile_weights(const float *W, int N, int K, long ldw, int epi, int gateD, float *out);

/* Tuning / roofline entry (not part of the drop-in surface): a stride-1 conv layer (K = 1 or 3, Cin % 32 == 0)
 * whose weights are ALREADY packed on the device as [round128(Cout)][K*Cin] (tap-major, k contiguous), launched
 * `iters` times between two HIP events recorded on `stream`; tile: 0 = production heuristic, 1 = 128x128,
 * 2 = 64x64, 3 = 128x64, 4 = 64x128, 5 = 64x64 with 64-deep K chunks, 6 = 160x128, 7 = 96x128; 31 / 39 / 33 = the LDS-DMA ring engine's
 * 128x128 tile with 4 / 8 waves and its 96x128 tile (conv_gemm_ring.hip), 35 / 36 = 39 / 33 with the tiles dealt to the XCDs in blocks that
 * share operands, 37 = bands (128 x 128 + 64 x 128 tiles) + dealt tiles, 38 = whole tiles + a stream-K band (deterministic; not bit-identical
 * with the others).  *ms_out = mean launch duration in milliseconds. */
int ts_op_conv1d_timed(ts_ctx *ctx, const float *x_dev, int B, int Lin, int Cin, const float *w_packed_dev,
                       const float *bias_dev, int Cout, int K, int tile, int iters, float *out_dev, float *ms_out,
                       void *stream);
/* The same for a strided convolution without padding + GELU — the shape of the wav2vec2 feature convolutions
 * (out[t] = sum_k W_k x[stride t + k], HF Wav2Vec2FeatureEncoder; K <= 4): out_dev is (B, (Lin - K) / stride + 1, Cout). */
int ts_op_conv1d_strided_timed(ts_ctx *ctx, const float *x_dev, int B, int Lin, int Cin, const float *w_packed_dev,
                               const float *bias_dev, int Cout, int K, int stride, int tile, int iters, float *out_dev,
                               float *ms_out, void *stream);

/* Tuning / test entry for conv_taps48.hip (the wav2vec2 positional convolution, HF Wav2Vec2PositionalConvEmbedding + the encoder's residual
 * add): grouped convolution with G groups of 48 channels in and out, `ntap` taps -ntap / 2 .. ntap - ntap / 2 - 1 with zero padding, out = GELU(conv +
 * bias) + res.  x_dev / res_dev (optional) / out_dev: (B, T, G * 48); w_dev: [G][48][ntap * 48] (tap-major, a tap's 48 input channels contiguous);
 * bias_dev: [G * 48].  `iters` launches between two HIP events; *ms_out = mean launch duration in milliseconds. */
int ts_op_conv_taps48_timed(ts_ctx *ctx, const float *x_dev, int B, int T, int G, int ntap, const float *w_dev, const float *bias_dev,
                            const float *res_dev, int iters, float *out_dev, float *ms_out, void *stream);

/* Host-only helper (no GPU needed): the tile plan conv_gemm_f32's LDS-DMA ring engine gives a layer of `groups` problems of M rows x N columns —
 * 128 (128 x 128 tiles), 96 (96 x 128 tiles) or 64 (bands: 128 x 128 tiles for the whole rounds of 512 resident workgroups, 64 x 128 tiles for
 * the rows that are left) — by tile count: a last round at most half full costs half a round (csrc/conv_gemm_ring.hip::conv_gemm_ring_pick).
 * -1 on a bad argument.  No reference counterpart. */
int ts_debug_conv_ring_pick(int M, int N, int groups);
/* Host-only helper (no GPU needed): the stream-K plan of the ring engine for `groups` problems of M rows x N columns x K (csrc/conv_gemm_ring.hip:
 * whole 128 x 128 tiles for the row tiles that fill whole units of 256 tiles, the rows after them as one list of (tile, 32-k stage) iterations
 * cut into equal runs).  1 = out6 = {row tiles kept whole, row tiles in the band, dealt ids of the whole-tile region, band workgroups,
 * stages per tile, the plan the layer gets by cost: 8 = this one, 9 / 3 / 7 = a whole-tile plan}; 0 = no stream-K plan for this shape
 * (whole units, under one unit, runs under 4 stages); -1 = bad argument.  No reference counterpart. */
int ts_debug_conv_sk_plan(int M, int N, int K, int groups, int *out6);
/* Host-only: the run of band workgroup q (0 <= q < band_workgroups, a multiple of 8) over a stream-K band of `band_tiles` tiles x `stages`
 * stages, in band-iteration units (tile * stages + stage): out4 = {first iteration, one past the last, the XCD (= q % 8) whose tiles
 * [xcd * band_tiles / 8, (xcd + 1) * band_tiles / 8) the run lies in, the run index the kernel's owner search finds for the first iteration
 * (= q / 8)}.  0 on success, -1 on a bad argument.  No reference counterpart. */
/* 1 if the current device passed the stream-K band's hardware check (workgroup ids of equal residue mod 8 share an XCD: probed once by
 * ts_ctx_create), 0 if not or if no context was created yet: then no layer gets a stream-K plan. */
int ts_debug_conv_sk_supported(void);
int ts_debug_conv_sk_run(int band_tiles, int stages, int band_workgroups, int q, int *out4);

/* Test aid: out[i] = the chain kernels' gate activation tanh(v[i]) * sigmoid(p[i]) as they compute it (v_exp_f32 / v_rcp_f32 form,
 * csrc/kernels.h::gate_act; reference: GatedActivation, gated_pixelcnn_v2.py:16-22) on n device floats. */
int ts_debug_gate_act(const float *v_dev, const float *p_dev, float *out_dev, long n, void *stream);

/* Test aid: how many captured hipGraphs the PixelCNN keeps for `stream` right now (whole-call graphs of repeated shapes + the chunk
 * graphs that serve first-time shapes of any length; bounded, least recently used out first), or -1. */
int ts_debug_pixelcnn_graphs(ts_pixelcnn *pix, void *stream);

/* Tuning entry (not part of the drop-in surface): `iters` DEPENDENT skinny_gemm launches replayed from one hipGraph;
 * *us_out = microseconds per launch.  gate != 0: N = 2K with the tanh*sigmoid epilogue; debug: unused. */
int ts_debug_skinny_chain(ts_ctx *ctx, int M, int K, int gate, int iters, int debug, float *us_out);

#ifdef __cplusplus
}
#endif
#endif /* TALKSHOW_HIP_DEBUG_H */
