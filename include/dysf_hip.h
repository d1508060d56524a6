/*
 * dysf_hip.h -- C ABI of the MI355X (gfx950) dysfluency feature-extraction library (libdysf_hip.so).
 *
 * The reference (kishormb/Recognizing-Speech-Dysfluencies-in-Stuttering) has no FFI: its hot path sits
 * behind plain Python functions.  Each entry point below names the reference interface it replaces;
 * the Python binding a maintainer would add is shown in INTEGRATION.md.
 *
 *   reference interface                                      replaced by
 *   -------------------------------------------------------  ----------------------------------------
 *   module constants TARGET_SR / MFCC_N   pipeline1.py:77-86  DysfParams + dysf_params_preset()
 *                    N_FFT / HOP_LEN      main.py:621-630
 *   extract_audio_features(y, sr)         pipeline1.py:206-239 dysf_extract_batch{,_dev}  (pooled rows)
 *   extract_features(y, sr, transcript)   pipeline1.py:257-265 same + 5 zero text features appended by
 *                                                              the caller (transcript is always "")
 *   load_audio() on PCM_16 WAV            pipeline1.py:100-106 int16 PCM accepted directly (x/32768)
 *   "except Exception -> zeros" convention pipeline1.py:213-214,237-239  per-clip status codes; non-OK
 *                                                              clips produce an all-zero row
 *   librosa.estimate_tuning (inside chroma_stft, p1:227)      optional per-clip tuning index output
 *
 * Conventions: every function returns 0 on success or a negative DYSF_E_* code and never throws;
 * dysf_last_error() returns a thread-local message for the last failure.  The library never retains
 * or frees caller buffers.  A handle owns its device tables and workspace and must not be used from
 * two threads at once; use one handle per GPU / stream.
 */
#ifndef DYSF_HIP_H
#define DYSF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYSF_ABI_VERSION 2

/* per-clip status (int32) */
#define DYSF_OK            0
#define DYSF_TOO_SHORT     1 /* fewer than delta_width frames: librosa.feature.delta raises (p1:217)   */
#define DYSF_NONFINITE     2 /* NaN/Inf sample: librosa.util.valid_audio raises                         */
#define DYSF_EMPTY         3 /* zero-length clip                                                        */
#define DYSF_TOO_LONG      4 /* device entry point only: the clip has more samples than the caller's max_samples */

/* function return codes */
#define DYSF_E_INVALID    (-1) /* bad argument / unsupported parameter combination */
#define DYSF_E_HIP        (-2) /* a HIP runtime call failed (message has the HIP error string) */
#define DYSF_E_NOMEM      (-3)

/* PCM element types */
#define DYSF_PCM_I16 0 /* int16, scaled by 1/32768 on the fly (soundfile / librosa.load semantics) */
#define DYSF_PCM_F32 1 /* float32 in [-1, 1) exactly as the reference passes `y` */

typedef struct DysfParams {
  int32_t sr;          /* sample rate the clips are at (no resampling on this path)            */
  int32_t n_fft;       /* 256, 512, 1024 or 2048                                               */
  int32_t hop;         /* hop length in samples                                                */
  int32_t n_mels;      /* Slaney mel bands, <= 128                                             */
  int32_t n_mfcc;      /* DCT-II (ortho) rows kept, <= 48                                      */
  float   fmin;        /* mel lower edge (Hz)                                                  */
  float   fmax;        /* mel upper edge (Hz); <= 0 means sr / 2                               */
  float   top_db;      /* power_to_db clamp below the clip-global max; < 0 disables the clamp  */
  float   amin;        /* power_to_db floor (1e-10)                                            */
  int32_t delta_width; /* must be 9 (librosa default; the only width the reference uses)       */
  int32_t chroma;      /* 1: append chroma_stft mean/std (12 + 12) to the pooled row           */
  int32_t rich;        /* 1: append main.py:738-754's descriptors to the pooled row: spectral-contrast mean (7) */
                       /*    and std (7), mean zero-crossing rate, mean RMS, mean spectral centroid (17 floats); */
                       /*    2048 must be a whole number of hops (frame_length of librosa's zcr / rms)            */
  int32_t parity;      /* 1: reproduce the reference's float rounding steps where they decide results: float64 FFT */
                       /*    rounded to complex64 and NumPy's complex abs (librosa.stft keeps a float64 product of  */
                       /*    frame and window), NumPy's pairwise float32 mean / std (pipeline1.py:220-221).  The     */
                       /*    tuning histogram of chroma_stft (p1:227) is a near-tie in most clips and follows the    */
                       /*    reference only with this set; 0 = float32 FFT, float64 statistics (faster)             */
  int32_t reserved[3];
} DysfParams;

#define DYSF_PRESET_REFERENCE_149 0 /* pipeline1.py / main1.py: 2048/512/128 mel/20 mfcc + chroma      */
#define DYSF_PRESET_BENCH_512     1 /* BASELINE.json configs[1]: 512/160/64 mel/40 mfcc, no chroma     */
#define DYSF_PRESET_MAIN_286      2 /* main.py:621-630, 718-763: 512/256/128 mel/40 mfcc + chroma + rich  */

typedef struct DysfHandle DysfHandle;

int32_t     dysf_abi_version(void);
const char* dysf_last_error(void);

/* Fill *p with a named preset. */
int32_t dysf_params_preset(int32_t preset, DysfParams* p);

/* Build all tables (window, FFT twiddles, banded mel pack, DCT pack, 100 chroma banks) in float64 on
 * the host, round to float32 and upload to `device`.  device < 0 builds host tables only (no HIP
 * call is made; used by CPU-side table tests).                                                        */
int32_t dysf_create(const DysfParams* p, int32_t device, DysfHandle** out);
void    dysf_destroy(DysfHandle* h);

int32_t dysf_pooled_dim(const DysfHandle* h);                   /* 6*n_mfcc (+24 with chroma)      */
int32_t dysf_frame_rows(const DysfHandle* h);                   /* 3*n_mfcc                         */
int64_t dysf_num_frames(const DysfHandle* h, int64_t n_samples);/* 1 + n_samples / hop              */

/* Debug / test access to the host copies of the tables.  which: 0 window[n_fft], 1 mel[n_mels*n_bins]
 * (dense, row-major), 2 dct[n_mfcc*n_mels], 3 chroma bank[12*n_bins] for tuning index `arg` (0..99).
 * Returns the number of floats written (<= cap) or a negative error.                                  */
int64_t dysf_debug_table(const DysfHandle* h, int32_t which, int32_t arg, float* dst, int64_t cap);

/*
 * Batched extraction, DEVICE buffers (the hot path; everything stays in HBM).
 *   d_pcm      concatenated clips, element type pcm_type; 16-byte aligned
 *   d_offsets  [n_clips + 1] sample offsets into d_pcm (clip i = [off[i], off[i+1]))
 *   max_samples  an upper bound on the longest clip (host-side sizing of workspace / launch)
 *   d_pooled   [n_clips, pooled_dim] float32, or NULL
 *   d_frames   [n_clips, 3*n_mfcc, frame_stride] float32 (MFCC, delta, delta2 rows; columns >= T_i are
 *              zero-filled), or NULL
 *   d_status   [n_clips] int32, or NULL
 *   d_tuning   [n_clips] int32 tuning-histogram index 0..99 (tuning = -0.5 + 0.01*idx), or NULL
 *   stream     hipStream_t (NULL = default stream).  The call only enqueues work (at most a small memset and two to four kernels).
 */
int32_t dysf_extract_batch_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type,
                               const int64_t* d_offsets, int32_t n_clips, int64_t max_samples,
                               float* d_pooled, float* d_frames, int64_t frame_stride,
                               int32_t* d_status, int32_t* d_tuning, void* stream);

/*
 * Batched extraction, HOST buffers: copies PCM to the device, runs dysf_extract_batch_dev, copies the
 * requested outputs back and synchronises.  Same argument meaning with host pointers; `offsets` is a
 * host array.  This is what a per-clip `extract_features(y, sr)` call binds to (n_clips = 1).
 */
int32_t dysf_extract_batch(DysfHandle* h, const void* pcm, int32_t pcm_type, const int64_t* offsets,
                           int32_t n_clips, float* pooled, float* frames, int64_t frame_stride,
                           int32_t* status, int32_t* tuning);

/*
 * Per-file audio metrics of pipeline1.py:151-186 (the columns of output_results/per_file_analysis.csv), batched, DEVICE
 * buffers: d_metrics [n_clips, 3] float32 = { snr_db (p1:151-165), spectral_flatness_mean (p1:168-174),
 * high_freq_energy_ratio (p1:177-186) }.  The handle must have librosa's STFT defaults (n_fft 2048, hop 512, chroma on:
 * REFERENCE_149), which is what spectral_flatness_mean transforms with.  Same input arguments as dysf_extract_batch_dev.
 * A clip shorter than 25 ms has snr_db 0; an empty or non-finite clip has all three 0 (the reference's except -> 0.0).
 */
int32_t dysf_metrics_batch_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, const int64_t* d_offsets,
                               int32_t n_clips, int64_t max_samples, float* d_metrics, void* stream);

/* Kernel-level test hook: power spectrogram of one clip, device buffers.
 * d_S is [T, n_bins] float32 (frame-major).                                                           */
int32_t dysf_debug_stft_power_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, int64_t n_samples,
                                  float* d_S, void* stream);

/* Same for the un-clamped log-mel spectrogram (10*log10(max(amin, mel))); d_logmel is [T, n_mels]. */
int32_t dysf_debug_logmel_dev(DysfHandle* h, const void* d_pcm, int32_t pcm_type, int64_t n_samples,
                              float* d_logmel, void* stream);

/* Test hook (chroma presets): the piptrack peak list (magnitude, histogram bin) of clip 0 of the most recent launch,
 * in append order.  Returns the number of peaks found (which may exceed cap).                                   */
int64_t dysf_debug_peaks(DysfHandle* h, float* mag, int32_t* bin, int64_t cap);

/* Measurement hook (bench.py): when enabled, every dysf_extract_batch_dev records HIP events on the launch
 * stream around its kernels; a later call with ms3 != NULL waits for them and returns the durations in ms of
 * {STFT+mel kernel, MFCC kernel, chroma kernel} of the most recent launch.                                   */
int32_t dysf_debug_timing(DysfHandle* h, int32_t enable, float* ms3);

/*
 * RandomForest inference: replaces scaler.transform + rf.predict / rf.predict_proba of the
 * single-clip predictor, main1.py:987-989 (model of pipeline1.py:496).  The forest is given as flat node arrays
 * (sklearn tree_.feature / threshold / children_left / children_right with tree-local child indices rebased to the
 * flat array, leaf rows of tree_.value normalised to class fractions); children must follow their parent.
 *   d_x      [n, n_features] float32 UNscaled feature rows (device); scaling by (mean, scale) is done in-kernel
 *            with StandardScaler's float32 rounding steps
 *   d_proba  [n, n_classes] float64 or NULL;  d_pred [n] int32 (first arg-max) or NULL
 */
typedef struct DysfForest DysfForest;
const char* dysf_forest_last_error(void);
int32_t dysf_forest_create(int32_t device, int32_t n_features, int32_t n_classes, int32_t n_trees, int64_t n_nodes,
                           const int32_t* feature, const double* threshold, const int32_t* left, const int32_t* right,
                           const double* value, const int32_t* root, const double* mean, const double* scale,
                           DysfForest** out);
void    dysf_forest_destroy(DysfForest* f);
int32_t dysf_forest_predict_dev(DysfForest* f, const float* d_x, int32_t n, double* d_proba, int32_t* d_pred,
                                void* stream);

/*
 * RBF support-vector classifier inference: replaces scaler.transform + SVC.predict /
 * SVC.predict_proba / decision_function(ovo) of pipeline1.py:498, 512-514 (main1.py:865 soft-votes it with the
 * forest and the MLP).  The model is given as scikit-learn exposes the fitted libsvm model:
 *   support_vectors [n_sv, n_features] (grouped by class), dual_coef [n_classes-1, n_sv], intercept [n_pairs]
 *   (= -rho), n_support [n_classes], prob_a / prob_b [n_pairs] (Platt coefficients; both NULL for a model fitted
 *   without probability=True), gamma (the resolved value, SVC._gamma), mean / scale [n_features] of the scaler.
 *   d_x         [n, n_features] float32 UNscaled feature rows (device)
 *   d_decision  [n, n_pairs] float64 pair decision values in the order (0,1), (0,2), ..., (1,2), ... or NULL
 *   d_proba     [n, n_classes] float64 (pairwise-coupled Platt probabilities) or NULL
 *   d_pred      [n] int32 (one-vs-one vote, first maximum -- what SVC.predict returns) or NULL
 */
typedef struct DysfSvc DysfSvc;
const char* dysf_svc_last_error(void);
int32_t dysf_svc_create(int32_t device, int32_t n_features, int32_t n_classes, int32_t n_sv,
                        const double* support_vectors, const double* dual_coef, const double* intercept,
                        const int32_t* n_support, const double* prob_a, const double* prob_b, double gamma,
                        const double* mean, const double* scale, DysfSvc** out);
void    dysf_svc_destroy(DysfSvc* m);
int32_t dysf_svc_predict_dev(DysfSvc* m, const float* d_x, int32_t n, double* d_decision, double* d_proba,
                             int32_t* d_pred, void* stream);

/*
 * MLP head as one kernel: replaces scaler.transform + MLPClassifier.predict / predict_proba of pipeline1.py:470-473,
 * 497, 512-514 for rows that are already on the device.  dims = {n_features, hidden..., n_outputs} (n_layers + 1
 * entries: n_features 1..512, every layer width 1..256; up to 4 layers); weights[l] is coefs_[l] ([dims[l], dims[l+1]] row-major float64),
 * biases[l] is intercepts_[l]; ReLU between layers (the MLPClassifier default), soft-max on the last layer.
 *   d_x [n, n_features] float32 UNscaled rows;  d_logits / d_proba [n, n_outputs] float32 or NULL;
 *   d_pred [n] int32 (first arg-max of the logits) or NULL.  float32 arithmetic.
 */
typedef struct DysfMlp DysfMlp;
const char* dysf_mlp_last_error(void);
int32_t dysf_mlp_create(int32_t device, int32_t n_layers, const int32_t* dims, const double* const* weights,
                        const double* const* biases, const double* mean, const double* scale, DysfMlp** out);
void    dysf_mlp_destroy(DysfMlp* m);
int32_t dysf_mlp_forward_dev(DysfMlp* m, const float* d_x, int32_t n, float* d_logits, float* d_proba,
                             int32_t* d_pred, void* stream);

/*
 * Rational sample-rate conversion: the resampling half of librosa.load(path, sr=16000),
 * pipeline1.py:100-106, for batches resident on the device.  Polyphase FIR as scipy.signal.resample_poly defines it:
 * out[n] = sum_k taps[(n + n_pre_remove)*down - k*up] * x[k], zero extension, len(out) = ceil(len(x)*up/down).
 * taps: float64, front-padded (rsd_amd/resample.py: design_taps builds resample_poly's default Kaiser(5.0) filter).
 *   d_pcm / pcm_type / d_in_offsets  as for dysf_extract_batch_dev (offsets are a DEVICE array here)
 *   d_out          float32 samples of all clips back to back;  d_out_offsets [n_clips+1] int64 (device) with
 *                  out_offsets[i+1] - out_offsets[i] = dysf_resampler_out_len(len_i)
 *   max_out_len    an upper bound on any single clip's output length (sizes the launch)
 */
typedef struct DysfResampler DysfResampler;
const char* dysf_resampler_last_error(void);
int32_t dysf_resampler_create(int32_t device, int32_t up, int32_t down, const double* taps, int32_t n_taps,
                              int32_t n_pre_remove, DysfResampler** out);
void    dysf_resampler_destroy(DysfResampler* r);
int64_t dysf_resampler_out_len(const DysfResampler* r, int64_t n_in);
int32_t dysf_resample_batch_dev(DysfResampler* r, const void* d_pcm, int32_t pcm_type, const int64_t* d_in_offsets,
                                int32_t n_clips, float* d_out, const int64_t* d_out_offsets, int64_t max_out_len,
                                void* stream);

/*
 * Benchmark CNN head of BASELINE.json configs[2] (SURVEY 8d "Config 3"; the reference itself has no CNN):
 * Conv2d(1->32, 3x3, pad 1) -> ReLU -> MaxPool 2x2 -> Conv2d(32->64, 3x3, pad 1) -> ReLU -> global average pool ->
 * Linear(64 -> n_classes) on the frame-level rows [n, rows, row_stride] (cols valid columns per row) that
 * dysf_extract_batch_dev writes.  Weights in PyTorch layout: conv1_w [32,1,3,3], conv2_w [64,32,3,3], fc_w [n_classes,64].
 * conv1 runs in float32; conv2 on the float16 MFMA with float32 accumulation (the precision configs[4] asks for).
 *   d_logits [n, n_classes] float32
 */
#define DYSF_CNN_F16 0 /* conv2 on v_mfma_f32_16x16x32_f16: float16 operands, float32 accumulation (BASELINE configs[4]) */
#define DYSF_CNN_F32 1 /* conv2 on v_mfma_f32_16x16x4_f32: the float32 module of configs[2], exact fmaf chains          */
typedef struct DysfCnn DysfCnn;
const char* dysf_cnn_last_error(void);
int32_t dysf_cnn_create(int32_t device, int32_t n_classes, int32_t precision, const float* conv1_w, const float* conv1_b,
                        const float* conv2_w, const float* conv2_b, const float* fc_w, const float* fc_b,
                        DysfCnn** out);
void    dysf_cnn_destroy(DysfCnn* m);
/* cols = the widest clip's valid columns; d_cols = [n] int32 valid columns of every clip on the device, or NULL when all
 * clips have `cols` (ragged batches, BASELINE configs[4]: each clip is convolved, pooled and averaged over its own width).
 * A handle owns one partial-sum workspace: use one handle per stream (work queued on one stream is ordered).            */
int32_t dysf_cnn_forward_dev(DysfCnn* m, const float* d_frames, int32_t n, int32_t rows, int32_t cols,
                             const int32_t* d_cols, int64_t row_stride, float* d_logits, void* stream);

/*
 * Audio cleaning stage ahead of the feature path, clean_audio_and_cache pipeline1.py:126-146:
 *     y_clean = nr.reduce_noise(y=y, sr=sr)       p1:139  (main.py:655 / main1.py:604 pass prop_decrease=0.8)
 *     y_clean = librosa.util.normalize(y_clean)   p1:140
 *     sf.write(out_path, y_clean, sr)             p1:141  (PCM_16)
 * noisereduce's default non-stationary spectral gate (n_fft 1024, hop 256, 30 000 samples of zero padding), in float64:
 * scipy.signal.stft -> |X| smoothed by filtfilt([b], [1, b - 1]) -> sigmoid mask -> triangle blur (fftconvolve "same")
 * -> scipy.signal.istft -> float32; then peak normalisation and libsndfile's clipping PCM_16 conversion.
 * dysf_clean_create takes nr.reduce_noise's keyword values (defaults: prop_decrease 1.0, time_constant_s 2.0,
 * freq_mask_smooth_hz 500, time_mask_smooth_ms 50, thresh_n_mult_nonstationary 2, sigmoid_slope_nonstationary 10,
 * padding 30000).  dysf_clean_batch_dev:
 *   d_pcm / pcm_type / d_offsets / n_clips / max_samples  as for dysf_extract_batch_dev (offsets on the DEVICE);
 *   total_samples   offsets[n_clips] (sizes the internal float32 signal when d_denoised is NULL)
 *   d_denoised      float32, nr.reduce_noise's return value          } each may be NULL; all are indexed like the
 *   d_normalized    float32, after librosa.util.normalize            } input samples (d_offsets)
 *   d_pcm16         int16, the samples sf.write stores               }
 *   d_status        [n_clips] DYSF_OK, DYSF_EMPTY, DYSF_TOO_LONG (the clip has more samples than max_samples: left
 *                   untouched) or DYSF_NONFINITE (the gate produced a non-finite sample, e.g. an all-zero clip:
 *                   librosa.util.normalize raises there and the reference writes no file)
 * Clips beyond noisereduce's chunk_size (600 000 samples) are gated as noisereduce gates them: in full 600 000-sample
 * spans, each padded with its 30 000 neighbouring samples and filtered on its own (main1.py's 126.5 s sidebar uploads
 * took this path).  A handle owns its workspaces: use one handle per stream (work queued on one stream is ordered).
 */
typedef struct DysfCleaner DysfCleaner;
const char* dysf_clean_last_error(void);
int32_t dysf_clean_create(int32_t device, int32_t sr, double prop_decrease, double time_constant_s,
                          double freq_mask_smooth_hz, double time_mask_smooth_ms, double thresh_n_mult,
                          double sigmoid_slope, int32_t padding, DysfCleaner** out);
void    dysf_clean_destroy(DysfCleaner* c);
int32_t dysf_clean_batch_dev(DysfCleaner* c, const void* d_pcm, int32_t pcm_type, const int64_t* d_offsets, int32_t n_clips,
                             int64_t max_samples, int64_t total_samples, float* d_denoised, float* d_normalized,
                             int16_t* d_pcm16, int32_t* d_status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYSF_HIP_H */
