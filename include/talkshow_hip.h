/*
 * talkshow_hip.h — C ABI of the MI355X-native speech -> SMPL-X body-motion inference path.
 *
 * The reference (yhw-yhw/TalkSHOW) has no FFI: its boundary for this path is the Python surface of
 * package `nets`.  This header is what a host language binds underneath that surface;
 * our own `nets/` package (same names / signatures as the reference's) is the first client, through
 * ctypes (talkshow_amd/_lib.py).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; ts_last_error() gives the message
 *     (thread-local).  Nothing throws, nothing aborts.
 *   - "dev" pointers are HIP device pointers on the context's device; "host" pointers are CPU memory.
 *   - all activations are fp32, time-major / channel-last ("NLC"): x[b][t][c]; code indices are int64
 *     exactly as the reference's torch.int64 latents.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Calls enqueue work
 *     and return; the caller synchronises (torch.cuda.synchronize() / hipStreamSynchronize).
 *   - weights are handed over ONCE as the reference's own state_dict (name, host pointer, shape); BatchNorm
 *     folding, mask-A zeroing, tap/segment packing and upload happen inside the library.
 *
 * Threads (the reference is single-threaded, default stream: SURVEY.md §8b)
 *   - ONE HOST THREAD PER STREAM AT A TIME.  Handles (ts_ctx, ts_convnet, ts_vqvae, ts_pixelcnn, ts_face, ts_mfcc, ts_smplx) hold
 *     read-only weights plus one scratch arena and one hipGraph cache PER STREAM; several host threads may call into the same
 *     handles concurrently as long as each thread uses its own stream (the per-stream maps are mutex-guarded; a stream's arena and
 *     graphs are only touched by the thread driving that stream).  Two threads on the same stream at the same time is a data race.
 *   - create / destroy / ts_face_set_arith / ts_prof_enable of a handle must not run concurrently with calls on that handle.
 *   - ts_pixelcnn_stream sessions belong to the thread that steps them.
 *   - ts_last_error() is thread-local.  ts_stream_destroy(stream) drops that stream's arenas in every live handle: no call on the
 *     stream may be in progress.
 *   Exercised by tests/test_gpu_threads.py (three host threads, one stream each, bit-equal to the serial run).
 */
#ifndef TALKSHOW_HIP_H
#define TALKSHOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ts_ctx ts_ctx;
typedef struct ts_convnet ts_convnet;     /* AudioEncoder            nets/spg/vqvae_1d.py:11-34          */
typedef struct ts_vqvae ts_vqvae;         /* VQVAE                   nets/spg/vqvae_1d.py:152-208        */
typedef struct ts_pixelcnn ts_pixelcnn;   /* GatedPixelCNN           nets/spg/gated_pixelcnn_v2.py:90-177 */
typedef struct ts_face ts_face;           /* s2g_face.Generator      nets/spg/s2g_face.py:142-224        */
typedef struct ts_mfcc ts_mfcc;
typedef struct ts_smplx ts_smplx;         /* smplx.SMPLX forward (third-party) scripts/demo.py:122-152, get_j.py */           /* get_mfcc_ta front-end   data_utils/utils.py:148-231         */

/* One entry of a reference state_dict: key name as the reference spells it (an optional "module." prefix is
 * accepted and stripped, nets/smplx_body_pixel.py:119-126), fp32 host data, shape.  int64 buffers
 * (num_batches_tracked) may be passed with data == NULL; they are ignored. */
typedef struct ts_tensor {
    const char *name;
    const float *data;
    int32_t ndim;
    int64_t shape[4];
} ts_tensor;

/* ---- context ------------------------------------------------------------------------------------------ */
int ts_ctx_create(int device, ts_ctx **out);
void ts_ctx_destroy(ts_ctx *ctx);
const char *ts_last_error(void);
/* library / build identification, e.g. "talkshow_hip 0.1 gfx950" */
const char *ts_version(void);
/* Non-blocking HIP streams for keeping several independent batches in flight on one GPU (the library keeps one
 * scratch arena per stream; weights are shared).  *out is a hipStream_t. */
int ts_stream_create(ts_ctx *ctx, void **out);
int ts_stream_destroy(ts_ctx *ctx, void *stream);

/* ---- AudioEncoder(in_dim=64, num_hiddens, num_residual_layers, ·)  — vqvae_1d.py:11-34 ------------------ */
int ts_audioenc_create(ts_ctx *ctx, const ts_tensor *sd, int n, int in_dim, int num_hiddens,
                       int num_residual_layers, ts_convnet **out);
void ts_convnet_destroy(ts_convnet *net);
/* AudioEncoder.forward (vqvae_1d.py:27-34): mfcc_dev (B,T,in_dim) -> feat_dev (B,H,num_hiddens), H = T//4
 * (two k4/s2/p1 convolutions: L -> floor(L/2)). */
int ts_audioenc_forward(ts_convnet *net, const float *mfcc_dev, int B, int T, float *feat_dev, void *stream);

/* ---- VQVAE(in_dim, embedding_dim, num_embeddings, num_hiddens, num_residual_layers, ·) — vqvae_1d.py:152 --
 * num_embeddings == 0 builds the quantiser-free auto-encoder `vqvae_1d.AE` (vqvae_1d.py:211-235), the FGD feature
 * extractor of nets/body_ae.py: same Encoder / Decoder, no vq_layer keys; its checkpoint's unused frame_enc / GRU
 * entries (Decoder(ae=True), never touched by forward) are accepted and ignored. */
int ts_vqvae_create(ts_ctx *ctx, const ts_tensor *sd, int n, int in_dim, int embedding_dim, int num_embeddings,
                    int num_hiddens, int num_residual_layers, ts_vqvae **out);
void ts_vqvae_destroy(ts_vqvae *vq);
/* VQVAE.encode (vqvae_1d.py:196-199) + VectorQuantizerEMA eval branch (vqvae_modules.py:274-286,311-323):
 * poses_dev (B,T,in_dim) -> z_dev (B,H,embedding_dim) [may be NULL], latents_dev (B,H) int64,
 * quantized_dev (B,H,embedding_dim) [may be NULL].  For an auto-encoder handle (num_embeddings == 0) this is
 * AE.encode (vqvae_1d.py:233-235): z_dev is required, the other two outputs must be NULL. */
int ts_vqvae_encode(ts_vqvae *vq, const float *poses_dev, int B, int T, float *z_dev, int64_t *latents_dev,
                    float *quantized_dev, void *stream);
/* VQVAE.decode(latents=...) (vqvae_1d.py:201-208): latents_dev (B,H) int64 -> recon written into
 * out_dev[b][t][out_col0 + c], c < in_dim, row stride out_ld floats (so body and hand decoders can write
 * the two halves of one (B,4H,129) buffer — the torch.cat of smplx_body_pixel.py:285). */
int ts_vqvae_decode(ts_vqvae *vq, const int64_t *latents_dev, int B, int H, float *out_dev, int out_ld,
                    int out_col0, void *stream);
/* Decoder.forward on CONTINUOUS latents (AE.forward eval branch, vqvae_1d.py:225-229; also VQVAE.decode(e=...)):
 * z_dev (B,H,embedding_dim) -> recon, same output addressing as ts_vqvae_decode. */
int ts_vqvae_decode_z(ts_vqvae *vq, const float *z_dev, int B, int H, float *out_dev, int out_ld, int out_col0,
                      void *stream);
/* body + hand decode in lockstep into one (B,4H,body_dim+hand_dim) buffer (the two decode calls + torch.cat of
 * smplx_body_pixel.py:282-285). */
int ts_vqvae_decode_pair(ts_vqvae *vq_body, ts_vqvae *vq_hand, const int64_t *lat_body_dev, const int64_t *lat_hand_dev,
                         int B, int H, float *out_dev, void *stream);
/* VQVAE.forward, eval branch (vqvae_1d.py:184-189): encode -> quantise -> decode in one call. */
int ts_vqvae_forward(ts_vqvae *vq, const float *poses_dev, int B, int T, int64_t *latents_dev, float *out_dev,
                     int out_ld, int out_col0, void *stream);

/* ---- GatedPixelCNN(input_dim, dim, n_layers, n_classes, audio=True, bh_model=True) ---------------------- */
int ts_pixelcnn_create(ts_ctx *ctx, const ts_tensor *sd, int n, int input_dim, int dim, int n_layers,
                       int n_classes, int aud_dim, ts_pixelcnn **out);
void ts_pixelcnn_destroy(ts_pixelcnn *pix);

#define TS_SAMPLE_GREEDY 0      /* argmax(logits), ties -> lowest index (the harness of SURVEY.md §0.3)           */
#define TS_SAMPLE_UNIFORMS 1    /* inverse-CDF draw from softmax(logits) with caller-supplied uniforms (B,H,2)    */
#define TS_SAMPLE_PHILOX 2      /* same draw, uniforms from Philox4x32-10(seed; clip index, position)             */
#define TS_TEACHER_FORCED 3     /* do not sample: positions are read from codes_dev (GatedPixelCNN.forward)       */

/* GatedPixelCNN.generate (gated_pixelcnn_v2.py:152-177), computed incrementally (row cache) instead of the
 * reference's full-grid recompute per position; same arithmetic per position.
 *   label_dev (B,) int64 speaker class; aud_dev (B,H,aud_dim) per-row audio features (the reference's
 *   (B,aud_dim,H,2) tensor is this repeated over the 2 columns, smplx_body_pixel.py:274);
 *   codes_dev (B,H,2) int64: output (input when mode == TS_TEACHER_FORCED);
 *   uniforms_dev (B,H,2) fp32 for TS_SAMPLE_UNIFORMS else NULL; seed / clip_index0 for TS_SAMPLE_PHILOX
 *   (clip b draws from subsequence clip_index0 + b, so results do not depend on how clips are sharded);
 *   logits_dev optional (B,H,2,input_dim) fp32: logits of every position as the reference's forward gives them.
 *   pre_codes_dev / pre_aud_dev / H0: optional continuity prefix (gated_pixelcnn_v2.py:158-165): H0 rows of
 *   already generated codes (B,H0,2) and their audio features (B,H0,aud_dim); pass NULL, NULL, 0 otherwise.
 *   Philox position rule (since round 3): the counter word of a code is its ABSOLUTE grid position (row * 2 + column) with
 *   the prefix rows counted — the first generated row of a call with H0 prefix rows draws at positions 2 H0, 2 H0 + 1 — so a
 *   clip generated as head + prefix-continued tail (or in ts_pixelcnn_stream_step chunks) draws exactly what one call over
 *   all its rows draws.  (Rounds 1-2 restarted at position 0 behind a prefix: sampled codes of H0 > 0 calls differ from those
 *   builds for the same seed.)  uniforms_dev always holds the H GENERATED rows only, (B,H,2). */
int ts_pixelcnn_generate(ts_pixelcnn *pix, const int64_t *label_dev, const float *aud_dev, int B, int H, int mode,
                         const float *uniforms_dev, uint64_t seed, int64_t clip_index0, int64_t *codes_dev,
                         float *logits_dev, const int64_t *pre_codes_dev, const float *pre_aud_dev, int H0,
                         void *stream);

/* hipGraph policy of ts_pixelcnn_generate / ts_body_pixel_infer (no counterpart in the reference, which launches eagerly): a one-shot
 * call (H0 == 0) replays graphs captured per (stream, B, H, mode).  A shape runs on small length-independent CHUNK graphs (8 code rows
 * each) until it is hot — its third sighting among the last 16 one-shot calls on the stream — and then gets one whole-call graph; at
 * most 24 unpinned graphs per stream are kept (8 whole-call + 16 chunk / streaming-step graphs), least recently used of its class out
 * first, destroyed behind an event (no host-side wait).  A serving host
 * that knows its pass shapes calls ts_pixelcnn_prepare once per (stream, shape): the whole-call graph is captured there (nothing
 * runs), pinned (never evicted; at most 12 per stream) and the first real call is already one replay. */
int ts_pixelcnn_prepare(ts_pixelcnn *pix, int B, int H, int mode, void *stream);
/* hipGraphs captured + instantiated on `stream` since the handle was created, or -1 (a serving loop checks that this stands still
 * once it is warm; bench.py asserts it over its timed regions). */
long ts_pixelcnn_graph_captures(ts_pixelcnn *pix, void *stream);

/* GatedPixelCNN(input_dim, dim, n_layers, n_classes, audio, bh_model=False) — the single-stack form (gated_pixelcnn_v2.py:37-42,
 * 80-85,147-150): vertical kernels one column wide, out_v = horiz_resid(gate(vert_stack(x_v) + class)) [+ x_v], logits from x_v; the
 * grid's W columns never mix, so the W codes of a row are drawn together.  Same state_dict keys as the reference module (the
 * vert_to_horiz / horiz_stack / fusion_h tensors it also holds are not read).  audio != 0: embedding_aud + fusion_v in front of
 * layer 1.  Not used by config/body_pixel.json; eager launches, no tuning. */
typedef struct ts_pixelcnn_v ts_pixelcnn_v;
int ts_pixelcnn_v_create(ts_ctx *ctx, const ts_tensor *sd, int n, int input_dim, int dim, int n_layers, int n_classes, int audio,
                         int aud_dim, ts_pixelcnn_v **out);
void ts_pixelcnn_v_destroy(ts_pixelcnn_v *pix);
/* generate / forward of that form: label_dev (B,), aud_dev (B,H,aud_dim) or NULL (audio == 0) — ONE audio row per code row: the
 * reference's (B, aud_dim, H, W) map with all W columns equal, as its caller builds it (smplx_body_pixel.py:274); a map whose columns
 * differ has no counterpart here (the Python layer raises NotImplementedError instead of dropping columns) —, grid (H, W) with W a power of two;
 * codes_dev (B,H,W) int64 out (in for TS_TEACHER_FORCED), logits_dev optional (B,H,W,input_dim), uniforms_dev (B,H,W) for
 * TS_SAMPLE_UNIFORMS; Philox position of (row, column) = (H0 + row) * W + column; prefix as in ts_pixelcnn_generate
 * (pre_codes_dev (B,H0,W), pre_aud_dev (B,H0,aud_dim)). */
int ts_pixelcnn_v_generate(ts_pixelcnn_v *pix, const int64_t *label_dev, const float *aud_dev, int B, int H, int W, int mode,
                           const float *uniforms_dev, uint64_t seed, int64_t clip_index0, int64_t *codes_dev, float *logits_dev,
                           const int64_t *pre_codes_dev, const float *pre_aud_dev, int H0, void *stream);

/* Launch count and algorithmic flops (2*M*N*K over every skinny_gemm launch) of the hipGraph captured for
 * (B, H, mode) on `stream` — what one replay executes; used by bench.py for the roofline line. */
int ts_pixelcnn_graph_stats(ts_pixelcnn *pix, void *stream, int B, int H, int mode, int64_t *launches, double *flops);

/* ---- s2g_face.Generator over the wav2vec2-base encoder (nets/spg/s2g_face.py:142-224, nets/spg/wav2vec.py:73-143) ---- */
/* state_dict of the reference Generator (keys "audio_encoder.*", "audio_feature_map.*", "audio_middle.*", "decoder.*",
 * "final_out.*"; the positional conv's weight norm is accepted under both the transformers>=4.3x names
 * "...conv.parametrizations.weight.original0/1" and the 4.22-era "...conv.weight_g/_v"). */
/* num_classes == 0 builds Generator(identity=False) (what smplx_face.py:37-45 constructs when convert_to_6d is set): no id_mlp keys,
 * first_net over the 256 audio channels alone, a 6-wide jaw head -> ts_face_generate writes (B,frames,106) and ignores id_dev. */
int ts_face_create(ts_ctx *ctx, const ts_tensor *sd, int n, int n_layers, int num_classes, ts_face **out);
void ts_face_destroy(ts_face *face);
/* Generator.forward, eval (s2g_face.py:196-224; TrainWrapper.generate / infer_on_audio, smplx_face.py:169-238):
 * wav_dev (B,N) fp32 16 kHz samples, id_dev (B,num_classes) fp32 one-hot or all-zero (smplx_face.py:205-208),
 * frames = N*30//16000 normally -> out_dev (B,frames,103) = [jaw(3) | expression(100)];
 * hidden_dev optional (B,frames,768): the wav2vec2 last_hidden_state (parity tests). */
int ts_face_generate(ts_face *face, const float *wav_dev, int B, int N, int frames, const float *id_dev, float *out_dev,
                     float *hidden_dev, void *stream);
/* OPT-IN arithmetic plan of the generator's GEMMs (no counterpart in the reference, which runs fp32 throughout): 0 = fp32 MFMA,
 * the default and the path every parity claim is made on; 3 / 6 = split-bf16: each fp32 operand becomes 2 / 3 bf16 terms and a
 * product 3 / 6 exact bf16 products accumulated in fp32 (csrc/conv_gemm_split.hip; measured error vs the reference golden in
 * DESIGN.md).  The first feature convolution, attention, LayerNorms and soft-max stay fp32.  Never applies to the body path.
 * The first selection of the 3-product plan also writes each layer's weights as bf16 plane images (one extra copy of the weights in HBM;
 * synchronises the device once). */
int ts_face_set_arith(ts_face *face, int bf16_products);

/* ---- audio front-end on the device: get_mfcc_ta (data_utils/utils.py:148-231) = torchaudio Resample(sr_in -> sr_out)
 * + MFCC(n_mfcc=64, n_fft=2048, n_mels=256, hop = 734 (fps 30) | 1467 (fps 15), mel_scale='htk').  torchaudio is
 * third-party and absent from the image: its published definitions are restated; parity against it is unpinned. ---- */
int ts_mfcc_create(ts_ctx *ctx, int sr_in, int sr_out, int fps, ts_mfcc **out);
void ts_mfcc_destroy(ts_mfcc *m);
/* frames for N input samples: floor(ceil(N * sr_out / sr_in) / hop) + 1 */
int ts_mfcc_num_frames(const ts_mfcc *m, long N);
/* wav_dev (B,N) mono fp32 at sr_in (multi-channel files: resample each channel, then average, as utils.py:150-154 —
 * resampling is linear, so averaging first is the same signal) -> feat_dev (B,T,64), T = ts_mfcc_num_frames(m, N). */
int ts_mfcc_forward(ts_mfcc *m, const float *wav_dev, int B, long N, float *feat_dev, void *stream);

/* ---- whole wrappers --------------------------------------------------------------------------------------- */
/* s2g_body_pixel.TrainWrapper.infer_on_audio after the MFCC front-end (smplx_body_pixel.py:272-285):
 * mfcc_dev (B,T,64), ids_dev (B,) int64 -> codes_dev (B,H,2) int64, poses_dev (B,4H,body_dim+hand_dim). */
int ts_body_pixel_infer(ts_convnet *audioenc, ts_pixelcnn *pix, ts_vqvae *vq_body, ts_vqvae *vq_hand,
                        const float *mfcc_dev, const int64_t *ids_dev, int B, int T, int mode,
                        const float *uniforms_dev, uint64_t seed, int64_t clip_index0, int64_t *codes_dev,
                        float *poses_dev, void *stream);
/* s2g_body_vq.TrainWrapper.infer_on_audio(initial_pose=gt) core (smplx_body_vq.py:254-281):
 * poses_dev (B,T,body_dim+hand_dim) in c_index order -> recon_dev same shape, codes_dev (B,H,2) int64.
 * Either output may be NULL: recon_dev == NULL is the encode-only form (VQVAE.encode of both parts, the latents
 * `s2g_body_pixel.__call__` builds, smplx_body_pixel.py:193-203).  Body and hand networks run in lockstep
 * (same-shape layers go out as one grouped launch). */
int ts_body_vq_infer(ts_vqvae *vq_body, ts_vqvae *vq_hand, const float *poses_dev, int B, int T,
                     int64_t *codes_dev, float *recon_dev, void *stream);

/* ---- single operators (kernel-level parity tests call these; weights given in the reference's layouts) ------ */
/* nn.Conv1d / nn.ConvTranspose1d (+ optional fused activation: 0 none, 1 LeakyReLU(0.2), 2 ReLU) on NLC data:
 * x_dev (B,Lin,Cin); w_host (Cout,Cin,K) or, transposed, (Cin,Cout,K); bias_host (Cout) or NULL;
 * out_dev (B,Lout,Cout).  Supported: K in {1,3} stride 1 pad (K-1)/2; K=4 stride 2 pad 1 (both directions). */
int ts_op_conv1d(ts_ctx *ctx, const float *x_dev, int B, int Lin, int Cin, const float *w_host,
                 const float *bias_host, int Cout, int K, int stride, int pad, int transposed, int act,
                 float *out_dev, void *stream);
/* VectorQuantizerEMA.get_code_indices (vqvae_modules.py:311-319): x_dev (M,dim), codebook_dev (ncode,dim)
 * -> idx_dev (M) int64 = argmin_j (|x|^2 + |e_j|^2 - 2 x.e_j), ties -> lowest j. */
int ts_op_vq_argmin(ts_ctx *ctx, const float *x_dev, int M, const float *codebook_dev, int ncode, int dim,
                    int64_t *idx_dev, void *stream);
/* F.linear on few rows (the per-position GEMM of the PixelCNN chain): x_dev (M,K), w_host (N,K), bias_host (N)
 * -> out_dev (M,N); relu optional. */
int ts_op_linear(ts_ctx *ctx, const float *x_dev, int M, int K, const float *w_host, const float *bias_host,
                 int N, int relu, float *out_dev, void *stream);
/* the per-position sampler: logits_dev (B,V) -> idx_dev (B) int64; mode TS_SAMPLE_GREEDY or
 * TS_SAMPLE_UNIFORMS (uniforms_dev (B)). */
int ts_op_sample(ts_ctx *ctx, const float *logits_dev, int B, int V, int mode, const float *uniforms_dev,
                 int64_t *idx_dev, void *stream);
/* the same draw in TS_SAMPLE_PHILOX mode (softmax + multinomial(1) of gated_pixelcnn_v2.py:173-176 with the library's own
 * random stream): row b draws with u = Philox4x32-10(key = seed; counter = (position, clip_index0 + b, 0)) >> 8 * 2^-24,
 * exactly what ts_pixelcnn_generate uses for clip clip_index0 + b at grid position row * 2 + column. */
int ts_op_sample_philox(ts_ctx *ctx, const float *logits_dev, int B, int V, uint64_t seed, int64_t clip_index0,
                        uint32_t position, int64_t *idx_dev, void *stream);

/* Output assembly the callers do after both generators (scripts/demo.py:207-229 + data_utils/lower_body.py:68-87
 * `part2full`): body_dev (B,Tb,129) body+hand poses, face_dev (B,Tf,103) jaw(3)+expression(100) -> out_dev (B,Tf,265).
 * The body is aligned to the face length (last frame repeated, or trimmed); lower_pose33_host = the 33 fixed
 * lower-body values part2full inserts (`lower_pose`, or zeros with [6:9] = global orientation when stand=True). */
int ts_assemble_full(ts_ctx *ctx, const float *body_dev, int Tb, const float *face_dev, int Tf, int B,
                     const float *lower_pose33_host, float *out_dev, void *stream);

/* ---- instrumentation ---------------------------------------------------------------------------------------- */
/* Per-kernel-family device time of the calls made on this context since the last reset, measured with HIP
 * events on the launch stream when enabled (adds synchronisation: benchmarking / profiling only).
 * families: 0 conv_gemm, 1 skinny_gemm (PixelCNN chain), 2 vq / sampling / glue.  ms_out[3], launches_out[3],
 * flops_out[3] (algorithmic 2*M*N*K of the GEMM launches; 0 for family 2). */
int ts_prof_enable(ts_ctx *ctx, int on);
int ts_prof_read(ts_ctx *ctx, double *ms_out, int64_t *launches_out, double *flops_out, int reset);
/* the same with n_families (1..4) entries per array; family 3 = the face generator's fused attention kernel (flops = 4 T^2 64 per
 * (clip, head) and layer), which ts_prof_read's three families leave out */
int ts_prof_read_n(ts_ctx *ctx, int n_families, double *ms_out, int64_t *launches_out, double *flops_out, int reset);

/* stage 1 of ts_mfcc_forward alone (get_mfcc_sepa, data_utils/utils.py:234-263, resamples the whole clip and then takes
 * the MFCC of two parts): wav_dev (B,N) at sr_in -> out_dev (B, ts_mfcc_resampled_len(m, N)) at sr_out. */
long ts_mfcc_resampled_len(const ts_mfcc *m, long n_samples);
int ts_mfcc_resample(ts_mfcc *m, const float *wav_dev, int B, long n_samples, float *out_dev, void *stream);
/* librosa.load(path, sr=16000) of the face front-end (data_utils/utils.py:194; librosa ~= 0.9.2 -> resampy 'kaiser_best'):
 * band-limited interpolation with a Kaiser-windowed sinc table (64 zero crossings, 512 samples per crossing, rolloff
 * 0.9475937167399596, beta 14.769656459379492), output length ceil(N * sr_out / sr_in) (librosa's fix_length).
 * wav_dev (B,N) mono -> out_dev (B, ts_resample_kaiser_len(N, sr_in, sr_out)).  Third-party arithmetic: PARITY UNPINNED. */
long ts_resample_kaiser_len(long n_samples, int sr_in, int sr_out);
int ts_resample_kaiser(ts_ctx *ctx, const float *wav_dev, int B, long n_samples, int sr_in, int sr_out, float *out_dev,
                       void *stream);

/* ---- streaming generation -----------------------------------------------------------------------
 * Replaces the pre_latents / pre_audio prefix of GatedPixelCNN.generate (gated_pixelcnn_v2.py:158-165) and its caller
 * (smplx_body_pixel.py:260-269,291-304), which re-run the whole prefix for every chunk: a session keeps the row cache
 * (one previous row per layer, the layer-0 partial sums, the last code rows) on the device, so a step costs the same
 * whatever the history length and the state is O(1) (the receptive field is 17 code rows).  A clip generated in chunks
 * is bit-identical to the same clip generated by one ts_pixelcnn_generate call (greedy; and stochastic, since the
 * Philox position of a code is its absolute (row, column)).
 * label_dev (B,) int64 is fixed for the session; max_chunk_rows bounds Hc of every step (buffers are sized once). */
typedef struct ts_pixelcnn_stream ts_pixelcnn_stream;
int ts_pixelcnn_stream_open(ts_pixelcnn *pix, const int64_t *label_dev, int B, int max_chunk_rows, ts_pixelcnn_stream **out);
/* aud_dev (B,Hc,aud_dim): audio-encoder rows of the next Hc code rows -> codes_dev (B,Hc,2) int64.  mode: GREEDY,
 * UNIFORMS (uniforms_dev (B,Hc,2)) or PHILOX (seed, clip_index0 as in ts_pixelcnn_generate). */
int ts_pixelcnn_stream_step(ts_pixelcnn_stream *st, const float *aud_dev, int Hc, int mode, const float *uniforms_dev,
                            uint64_t seed, int64_t clip_index0, int64_t *codes_dev, void *stream);
/* code rows generated so far */
int64_t ts_pixelcnn_stream_rows(const ts_pixelcnn_stream *st);
void ts_pixelcnn_stream_close(ts_pixelcnn_stream *st);

/* ---- batched SMPL-X joints / vertices ----------------------------------------------------------------
 * Replaces the per-frame float64 CPU calls of smplx.SMPLX.forward in scripts/demo.py:122-152 (get_vertices) and
 * data_utils/get_j.py:20-50 (get_joints).  Third-party arithmetic (smplx ~= 0.1.28, requirements.txt:5; package and
 * licensed model file absent): PARITY UNPINNED — the published LBS algorithm restated; fp32 on the device.
 * Model arrays are HOST float32 / int32 in the package's layouts: v_template (V,3), shapedirs (V,3,n_betas+n_expr) =
 * cat(shapedirs, expr_dirs), posedirs ((J-1)*9, V*3), J_regressor (J,V), parents (J), lbs_weights (V,J), pose_mean (J*3)
 * in the package's joint order; pose_src_offset[j] = column of a pose row where joint j's axis-angle starts (TalkSHOW's
 * 265-d rows: global_orient 9, body 12.., jaw 0, eyes 3 / 6, hands 75.. / 120.., get_j.py:21-30); extra_idx = vertex ids
 * of vertex_joint_selector, lmk_faces (n_lmk,3) = faces_tensor[lmk_faces_idx], lmk_bary (n_lmk,3).
 * with_vertices != 0 also uploads the full-mesh blend-shape matrix (V*3 x 896 floats, 112 MB for the real model). */
int ts_smplx_create(ts_ctx *ctx, int V, int J, int n_betas, int n_expr, const float *v_template, const float *shapedirs,
                    const float *posedirs, const float *J_regressor, const int32_t *parents, const float *lbs_weights,
                    const float *pose_mean, const int32_t *pose_src_offset, int n_extra, const int32_t *extra_idx, int n_lmk,
                    const int32_t *lmk_faces, const float *lmk_bary, int with_vertices, ts_smplx **out);
void ts_smplx_destroy(ts_smplx *m);
/* J + n_extra + n_lmk (127 for the reference's model) */
int ts_smplx_num_joints(const ts_smplx *m);
/* rows_dev (N,row_ld): pose rows, expression coefficients at columns [expr_off, expr_off + n_expr); betas_dev (n_betas) shared
 * by all rows, or (N,n_betas) when betas_per_row != 0 -> joints_dev (N, num_joints, 3) and, if not NULL, verts_dev (N,V,3). */
int ts_smplx_forward(ts_smplx *m, const float *betas_dev, int betas_per_row, const float *rows_dev, int row_ld, int expr_off,
                     int64_t N, float *joints_dev, float *verts_dev, void *stream);

/* ---- evaluation on the device --------------------------------------------------------------------
 * The reference computes its metrics on the CPU after the hot path (scripts/test_body.py:113-194); these are the
 * reductions behind them, float64 accumulation, deterministic (fixed-order partial sums, no float atomics).
 * Outputs are device doubles. */
/* evaluation/FGD.py:131-146 (np.mean / np.cov inputs of the Frechet distance): feat_dev (n,D) float32, D in {32,64,128}
 * -> stats_dev[0..D) = sum over rows, stats_dev[D + i*D + j] = sum over rows of x_i x_j  (D + D*D doubles). */
int ts_eval_feat_stats(ts_ctx *ctx, const float *feat_dev, int64_t n, int D, double *stats_dev, void *stream);
/* evaluation/FGD.py:153-158 (feat_dist numerator): sum over all n elements of |a - b|. */
int ts_eval_l1_total(ts_ctx *ctx, const float *a_dev, const float *b_dev, int64_t n, double *out_dev, void *stream);
/* scripts/test_body.py:98-110 body_loss on joints: gt_dev (T,J,3), prs_dev (B,T,J,3) ->
 * out3_dev = { sum_t sum_b sum_{j<J_lvd} | |v_pr| - |v_gt| | over t < T_lvd-1   (LVD numerator, metrics.py:73-84),
 *              sum_{b,t,j} |gt - pr|_2                                           ("error" numerator),
 *              sum_{t,j} | var_b(pr) |_2  (unbiased variance over the B samples) ("diverse" numerator) }. */
int ts_eval_body_loss(ts_ctx *ctx, const float *gt_dev, const float *prs_dev, int B, int T, int J, int J_lvd, int T_lvd,
                      double *out3_dev, void *stream);
/* evaluation/metrics.py:96-109 diversity: kps_dev (bs, L) -> sum over pairs i<j of sum_k |kps_i[k] - kps_j[k]|. */
int ts_eval_diversity(ts_ctx *ctx, const float *kps_dev, int bs, int64_t L, double *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TALKSHOW_HIP_H */
