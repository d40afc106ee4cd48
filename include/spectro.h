/*
 * spectro.h -- C ABI of libspectro.so, the MI355X (gfx950) STFT / analyser engine.
 *
 * This is the drop-in boundary for the one hot path of amilajack/spectrogram:
 * the Web Audio AnalyserNode that its draw loop polls once per animation frame.
 * The reference has no FFI of its own (it is browser JavaScript); the "binding"
 * it would use is an N-API addon whose object is handed to
 * AnalyserView.setAnalyserNode() (3D/visualizer.js:567-569).  Every entry point
 * below names the reference interface it stands in for (file:line under
 * /root/reference/src/javascripts).  Plain pointers and sizes only -- no torch,
 * no C++ types.  INTEGRATION.md shows the N-API / ctypes stubs on top of it.
 *
 * Threading: a handle is not thread-safe; use one handle per stream/thread
 * (the reference polls from the single JS main thread, UI/spectrogram.js:153-161).
 * Errors: every int-returning call returns SG_OK (0) or a negative sg_status;
 * sg_last_error() gives the thread-local message.  Nothing here falls back to a
 * CPU path: without a usable HIP device the calls fail with SG_ERR_HIP.
 */
#ifndef SPECTRO_H
#define SPECTRO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_VERSION 400 /* 0.4.0: + sg_runtime_info, sg_config.downmix_mode; 0.3.1: + sg_debug_twiddle_triplets; 0.3.0: sg_config.logfreq_align / history_samples, sg_set_fft_size, sg_debug_set */

typedef enum sg_status {
  SG_OK = 0,
  SG_ERR_INVALID = -1,     /* bad argument / configuration */
  SG_ERR_NOMEM = -2,       /* host or device allocation failed */
  SG_ERR_HIP = -3,         /* HIP runtime error (no device, launch failure, ...) */
  SG_ERR_UNSUPPORTED = -4, /* valid request this build has no path for (e.g. more log-frequency columns than LDS holds) */
  SG_ERR_STATE = -5        /* call order (e.g. poll before stream_begin) */
} sg_status;

typedef enum sg_window {
  SG_WINDOW_HANN = 0,     /* north_star default; periodic 0.5 - 0.5 cos(2 pi n/N) */
  SG_WINDOW_BLACKMAN = 1, /* AnalyserNode: alpha = 0.16, periodic */
  SG_WINDOW_RECT = 2
} sg_window;

typedef enum sg_scale { SG_SCALE_NONE = 0, SG_SCALE_INV_N = 1 } sg_scale;

typedef enum sg_out_kind {
  SG_OUT_POWER = 0, /* float32 |X[k]|^2 */
  SG_OUT_MAG = 1,   /* float32 |X[k]|   */
  SG_OUT_DB = 2,    /* float32 max(10 log10 |X[k]|^2, floor_db) */
  SG_OUT_DB_U8 = 3  /* uint8   clip(floor(255 (dB - min_db)/(max_db - min_db)), 0, 255):
                       the Uint8Array contract of getByteFrequencyData
                       (3D/visualizer.js:301,358) */
} sg_out_kind;

/* Sample format of the batch input (the step before the hot path in the reference is
 * decodeAudioData / a MediaStream, util/util.js:2-28, UI/player.js:110-115, which hand the
 * analyser float samples; recorded audio is usually 16-bit PCM). */
typedef enum sg_pcm_format {
  SG_PCM_F32 = 0, /* float32 in [-1, 1) */
  SG_PCM_S16 = 1, /* int16, value / 32768; converted inside the framing load (sg_stft) or on the way into the
                     history (sg_push): pass the int16 buffer through the `pcm` pointer, strides count samples */
  SG_PCM_S24 = 2  /* packed little-endian 24-bit PCM (3 bytes per sample, WAV "PCM 24"), value / 8388608;
                     converted inside the framing load (sg_stft) or on the way into the history (sg_push; the fused
                     sg_push_poll_* runs as push + poll for it); strides count samples */
} sg_pcm_format;

/* Where the log-frequency columns sample the row (sg_config.logfreq_cols).  The reference has both conventions:
 * its default display, the 3-D sonogram (3D/visualizer.js:52), looks the row up in the VERTEX shader with the mesh's
 * texture coordinate u = x / (sonogram3DWidth - 1), x = 0 .. 255 (3D/visualizer.js:54,213-214,
 * sonogram-vertex.shader:51-52): u runs from 0 to 1 inclusive, column 0 samples x = 1/256 and the last column x = 1,
 * which CLAMP_TO_EDGE (:313) turns into the last texel.  Its 2-D sonogram looks it up per FRAGMENT
 * (sonogram-fragment.shader:16), i.e. at pixel centres u = (c + 0.5) / cols. */
typedef enum sg_logfreq_align {
  SG_LOGFREQ_PIXEL_CENTRE = 0, /* u = (c + 0.5) / cols        (fragment shader; the C ABI's default) */
  SG_LOGFREQ_VERTEX = 1        /* u = c / (cols - 1), c = 0 .. cols - 1  (the 3-D sonogram's mesh; default of the
                                  JS / Python Analyser objects, which stand in for the reference's default mode) */
} sg_logfreq_align;

/* How cfg.downmix = K interleaved input channels become the one analysed channel. */
typedef enum sg_downmix_mode {
  SG_DOWNMIX_MEAN = 0,    /* x = (1/K) sum_c x_c: what the reference's `mix` GainNode feeds the analyser from a mono or stereo
                             source (UI/player.js:25,110-131) -- and the default */
  SG_DOWNMIX_SPEAKERS = 1 /* the W3C "speakers" down-mix to mono an AnalyserNode applies to its input (channel order L R C LFE
                             SL SR): K = 2: 0.5 (L + R); K = 4: 0.25 (L + R + SL + SR); K = 6: sqrt(1/2) (L + R) + C +
                             0.5 (SL + SR), LFE dropped; any other K has no speaker layout and is mixed "discrete": channel 0
                             alone.  Equals SG_DOWNMIX_MEAN for K = 1, 2, 4 (bit for bit: those run the same code). */
} sg_downmix_mode;

/* pointer-space flags for sg_stft / sg_push / sg_poll_* */
#define SG_PTR_HOST 0u
#define SG_IN_DEVICE 1u  /* input pointer is device memory  */
#define SG_OUT_DEVICE 2u /* output pointer is device memory */
#define SG_PTR_DEVICE (SG_IN_DEVICE | SG_OUT_DEVICE)
#define SG_IN_PINNED 4u  /* sg_push: the host input is page-locked (sg_alloc_pinned): a kernel ordered on the stream
                            reads it straight over PCIe into the history (no DMA set-up: ~25 us per hop saved on a
                            2 MB push) and sg_push does not wait for it.  The buffer may be rewritten only once that
                            kernel has run: after a synchronise of the stream, after an event recorded behind the
                            push has completed, or after a poll on the same stream that WAITS has returned (host
                            output without SG_OUT_PINNED).  A poll with SG_OUT_DEVICE or SG_OUT_PINNED only
                            enqueues and is no such point. */
#define SG_OUT_PINNED 8u /* sg_poll_u8 / sg_poll_f32: `out` is page-locked host memory (sg_alloc_pinned): the fused
                            poll kernel stores the rows straight into it over PCIe and the call does not wait;
                            synchronise the stream before reading them.
                            sg_stft: SG_IN_PINNED / SG_OUT_PINNED likewise -- the kernel reads the samples and
                            stores the rows in place (input and output cross PCIe at the same time instead of
                            one staged copy after the other); with both sides device or pinned the call only
                            enqueues */

/*
 * Engine configuration.  Mirrors the AnalyserNode attributes the reference
 * touches -- fftSize (UI/player.js:10), smoothingTimeConstant (UI/player.js:11,
 * 3D/visualizer.js:351,357,362), minDecibels/maxDecibels (never set there, so
 * the platform defaults -100/-30) -- plus the batch parameters BASELINE.json's
 * configs add (hop, window, output kind).
 */
typedef struct sg_config {
  uint32_t struct_size; /* = sizeof(sg_config); set by sg_config_default */
  int32_t fft_size;     /* N: power of two, 32 .. 32768 (AnalyserNode range) */
  int32_t hop;          /* >= 1; batch mode only */
  int32_t window;       /* sg_window */
  int32_t scale;        /* sg_scale: SG_SCALE_INV_N divides X by N (AnalyserNode) */
  int32_t out_kind;     /* sg_out_kind, batch mode */
  int32_t nyquist;      /* 0: N/2 bins = frequencyBinCount (3D/visualizer.js:299); 1: N/2+1 */
  float min_db;         /* AnalyserNode.minDecibels, default -100 */
  float max_db;         /* AnalyserNode.maxDecibels, default  -30 */
  float floor_db;       /* lower clamp of the dB outputs, default -200 */
  float smoothing;      /* AnalyserNode.smoothingTimeConstant: poll mode, and in sg_stft the same
                           recurrence X^ = tau X^_prev + (1 - tau)|X| along the frames of a channel */
  int32_t device_id;    /* HIP device ordinal; -1 = current device */
  int32_t pcm_format;   /* sg_pcm_format of sg_stft's input */
  int32_t downmix;      /* 0/1 = off.  K > 1: every analysed sample is the mean of K interleaved input
                           samples, x[i] = (1/K) sum_c pcm[(i K + c) * sample_stride] -- the mono down-mix
                           an AnalyserNode applies to a multi-channel input ("speakers" interpretation:
                           stereo -> 0.5 (L + R)) before windowing; fused into the framing load (sg_stft)
                           and applied by sg_push on the way into the history (poll mode) */
  int32_t logfreq_cols; /* 0 = off.  > 0: fuse the log-frequency column lookup the reference's shaders
                           apply to each analyser row (sonogram-vertex.shader:51-52,
                           sonogram-fragment.shader:16-19 with LINEAR / CLAMP_TO_EDGE sampling,
                           3D/visualizer.js:313-316): column c of `cols` samples the N/2-bin row at
                           normalised x = 256^(u-1), u = (c + 0.5)/cols, linearly interpolated.  Batch
                           output becomes [frames][cols] of the configured out_kind's values: float32, or
                           for SG_OUT_DB_U8 the blend of the two byte texels rounded to the nearest byte
                           (uint8, what a LINEAR-filtered UNSIGNED_BYTE texture fetch quantises to).
                           Poll mode: the row a frequency poll hands out -- and the history ring keeps --
                           is these `cols` columns of the analyser row instead of its N/2 bins (bytes:
                           the blend of the two floored byte texels, rounded; floats: the blend of the
                           two dB texels clamped at floor_db), so a 512-channel bank moves cols bytes per
                           channel per hop instead of N/2.
                           cols * 8 bytes of table must fit in LDS behind one workgroup's exchange
                           buffers: sg_create answers SG_ERR_UNSUPPORTED otherwise (e.g. > 2048 columns
                           at fft_size 16384). */
  int32_t logfreq_align; /* sg_logfreq_align: where column c samples the row -- pixel centres (default), or the
                           vertices of the reference's 3-D sonogram mesh, u = c / (cols - 1) */
  int32_t history_samples; /* poll mode: samples of history kept per channel, 0 = fft_size.  An AnalyserNode keeps its
                           most recent 32768 samples whatever fftSize is, so that a larger fftSize set later
                           (sg_set_fft_size) transforms audio that arrived before the change; the Analyser objects
                           pass 32768.  Must be >= fft_size; device memory is 2 * history_samples floats per channel */
  int32_t downmix_mode; /* sg_downmix_mode: how cfg.downmix = K channels are mixed (batch framing load and sg_push alike) */
} sg_config;

typedef struct sg_handle sg_handle;

/* ---- lifetime ---------------------------------------------------------- */

/* Fills *cfg with: N=1024, hop=256, Hann, no scale, POWER, N/2 bins,
 * -100/-30/-200 dB, smoothing 0, current device. */
int sg_config_default(sg_config* cfg);

/* Replaces `context.createAnalyser()` + the attribute writes that follow it
 * (UI/player.js:7,10,11).  Uploads twiddle and window tables; no CPU path. */
int sg_create(const sg_config* cfg, sg_handle** out);
int sg_destroy(sg_handle* h);

/* Thread-local text of the last failure on this thread ("" if none). */
const char* sg_last_error(void);
int sg_version(void);

/* ---- shape helpers ----------------------------------------------------- */

/* frames = floor((n_samples - N)/hop) + 1, 0 if n_samples < N (no padding). */
int64_t sg_num_frames(const sg_handle* h, int64_t n_samples);
/* = AnalyserNode.frequencyBinCount (3D/visualizer.js:299,301), +1 with nyquist; = logfreq_cols when set. */
int32_t sg_num_bins(const sg_handle* h);
/* 4 for the float outputs, 1 for SG_OUT_DB_U8 (bins or log-frequency columns alike). */
int32_t sg_out_elem_size(const sg_handle* h);
int sg_get_config(const sg_handle* h, sg_config* out);

/* ---- batch STFT (BASELINE.json configs 2, 3, 5) ------------------------ */

/*
 * n_channels independent channels; sample i of channel c is
 * pcm[c * channel_stride + i * sample_stride] (strides in floats; interleaved
 * stereo is channel_stride = 1, sample_stride = 2).  Frame f / bin k of
 * channel c is written to out[c * out_channel_stride + f * bins + k] (stride
 * in output elements; 0 means frames * bins).  `flags` says which of the two
 * pointers are device memory.  `hip_stream` is a hipStream_t (NULL = default
 * stream).  With device pointers on both sides the call only enqueues work on
 * that stream; with a host pointer on either side it stages through
 * handle-owned device memory and returns after the stream has drained.
 * Only out[c * out_channel_stride + f * bins + k] is written: caller bytes
 * between the channels of a padded output are left untouched on both paths.
 * With cfg.downmix = K > 1, n_samples counts ANALYSED (down-mixed) samples: the
 * call reads n_samples * K * sample_stride input elements per channel.
 * Streams: a handle is not thread-safe, and its calls are meant for ONE stream at a time.  Calls that only enqueue
 * (device / page-locked buffers on both sides) may be issued on different streams one after the other: the handle's
 * staging buffer is guarded -- a call that has to rewrite it while an earlier asynchronous call on another stream may
 * still read it makes its own stream wait for that call (a device-side event wait, no host stall), and so is the |X|
 * scratch of two-pass batch smoothing (cfg.smoothing > 0 with few channels) -- but the tables, the poll-mode history and
 * the smoothing state are not: keep the calls of one handle ordered by the caller.
 *
 * Stands in for N calls of analyser.getByteFrequencyData() (3D/visualizer.js:358)
 * over a recorded signal, i.e. the loop UI/spectrogram.js:153-161 runs at 60 Hz.
 */
int sg_stft(sg_handle* h, const float* pcm, int64_t n_channels, int64_t n_samples,
            int64_t channel_stride, int64_t sample_stride, void* out,
            int64_t out_channel_stride, uint32_t flags, void* hip_stream);

/* ---- poll mode: an AnalyserNode bank (config 4 and the N-API shim) ------ */

/*
 * Allocates device-resident history (the last H = cfg.history_samples, or N, samples per channel) and
 * smoothing state for n_channels analysers.  Replaces the sample ring the
 * platform fills behind `mix.connect(analyser)` (UI/player.js:25).  The history is a ring as well: 2 H floats per
 * channel, every sample written twice (at its ring position and H floats above it), so that the newest N <= H samples
 * always lie contiguous for the transform and a push never moves old samples -- every hop costs the same.
 */
int sg_stream_begin(sg_handle* h, int32_t n_channels);

/*
 * Appends n_new samples per channel; pcm is [n_channels][n_new] with
 * channel_stride floats between channels (0 = n_new).  Only the new samples
 * cross PCIe.  A native object cannot sit in a Web Audio graph, so this feed
 * replaces the graph edge `mix.connect(analyser)` (UI/player.js:25).
 * With cfg.downmix = K > 1 a channel's row holds n_new * K interleaved floats
 * (channel_stride 0 = n_new * K) and every analysed sample is their mean over K:
 * the mono mix the reference's `mix` GainNode performs on whatever the source
 * delivers -- a stereo file or microphone (UI/player.js:25,110-115,125-131).
 * With cfg.pcm_format = SG_PCM_S16 `pcm` points at int16 samples (a 16-bit capture
 * device: half the bytes across PCIe), with SG_PCM_S24 at packed 24-bit samples; strides count samples.
 */
int sg_push(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride,
            uint32_t flags, void* hip_stream);

/*
 * analyser.getByteFrequencyData(Uint8Array) -- 3D/visualizer.js:352,358.
 * One frame per channel over the latest N samples (zeros before the first N
 * have arrived): Blackman or configured window, scale, smoothing with
 * cfg.smoothing, 20 log10, byte map.  Writes min(len, bins) bytes per channel
 * to out[c * bins_stride ...]; excess caller bytes are left untouched, like the
 * platform does.  len is per channel; bins_stride 0 = len.  `bins` = N/2, or
 * cfg.logfreq_cols when that is set (the row is then the log-frequency columns).
 * One kernel launch per poll (transform with the analyser epilogue fused).  Polling again with no
 * sg_push in between hands out the same block's data: the smoother is not advanced a second time
 * (an AnalyserNode computes one frequency frame per render quantum, however often it is read).
 */
int sg_poll_u8(sg_handle* h, uint8_t* out, int64_t len, int64_t bins_stride,
               uint32_t flags, void* hip_stream);
/*
 * sg_push followed by sg_poll_u8, as one call -- and, when the block can be read where it lies, as ONE LAUNCH: with
 * SG_IN_PINNED or SG_IN_DEVICE input, no down-mix and 0 < n_new <= fft_size (even), the fused transform takes the
 * newest n_new samples of every channel straight from the caller's block (page-locked host memory is read in place over
 * PCIe while the older samples come from the device-resident history) and files them in the history on its way; any
 * other shape runs the two calls in sequence with the same results.  This is the per-animation-frame step of the
 * reference as a whole: the audio that arrived since the last frame reaches the analyser, then
 * getByteFrequencyData(freqByteData) (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358).  Config 4 (512 channels,
 * N = 4096, 1024 new samples per hop): one launch of 59 us instead of a 56 us push and a 29 us poll.  The input
 * buffer follows the SG_IN_PINNED rule of sg_push; flags carry the input and the output side (e.g.
 * SG_IN_PINNED | SG_OUT_PINNED).
 * Pageable host buffers (SG_PTR_HOST on the input side) of up to 256 KiB per call -- an analyser object polled from
 * JavaScript or Python: one channel, a few KB per animation frame -- take the one-launch path as well, through page-locked
 * staging the handle owns (copy in, one launch, one synchronise, copy out); the call is synchronous like every
 * pageable call.  Larger pageable blocks run sg_push + sg_poll_*.
 */
int sg_push_poll_u8(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, uint8_t* out, int64_t len,
                    int64_t bins_stride, uint32_t flags, void* hip_stream);
/* ... followed by getFloatFrequencyData */
int sg_push_poll_f32(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, float* out, int64_t len,
                     int64_t bins_stride, uint32_t flags, void* hip_stream);
/* analyser.getFloatFrequencyData(Float32Array): dB values, -inf for silence (with cfg.logfreq_cols: the
 * blend of dB texels clamped at cfg.floor_db). */
int sg_poll_f32(sg_handle* h, float* out, int64_t len, int64_t bins_stride,
                uint32_t flags, void* hip_stream);
/* Waits until everything this handle's calls have enqueued on `hip_stream` has finished (hipStreamSynchronize on the
 * handle's device).  The reference's getByteFrequencyData is synchronous (3D/visualizer.js:358: the bytes are there when
 * it returns); a caller that polls with SG_OUT_PINNED / pushes with SG_IN_PINNED gets the same guarantee from this call
 * without including any HIP header (examples/stream_latency.c). */
int sg_synchronize(sg_handle* h, void* hip_stream);

/* analyser.getByteTimeDomainData(Uint8Array) -- 3D/visualizer.js:363:
 * the first `len` (<= N) of the latest N samples as clip(floor(128 (1 + x)), 0, 255). */
int sg_poll_time_u8(sg_handle* h, uint8_t* out, int64_t len, int64_t stride,
                    uint32_t flags, void* hip_stream);
/* analyser.getFloatTimeDomainData(Float32Array): the same samples as floats (the fourth getter of an AnalyserNode;
 * the reference itself only reads the byte form, 3D/visualizer.js:363).  int16 pushes come back as x / 32768. */
int sg_poll_time_f32(sg_handle* h, float* out, int64_t len, int64_t stride,
                     uint32_t flags, void* hip_stream);
/*
 * The spectrogram history the reference keeps as a 1024 x 256 ALPHA / UNSIGNED_BYTE texture:
 * every frame texSubImage2D()s the byte row at `yoffset` and sets yoffset = (yoffset + 1) % 256
 * (3D/visualizer.js:60,293-331,390-417; TEXTURE_WRAP_T = REPEAT, :314).  sg_ring_begin allocates
 * a zeroed device ring [n_channels][height][N/2] ([..][cfg.logfreq_cols] when set); from then on every
 * sg_poll_u8 also deposits its full row there and advances the offset, with no extra transfer.
 */
int sg_ring_begin(sg_handle* h, int32_t height);
/* pinned != 0: every poll writes row 0 and the offset stays 0 -- what the reference does outside its sonogram
 * modes (`yoffset = 0` in the 2-D frequency / waveform modes, 3D/visualizer.js:385-390).  0 restores the ring. */
int sg_ring_set_pinned(sg_handle* h, int32_t pinned);
/* Size of the whole ring in bytes (n_channels * height * row, row = N/2 or cfg.logfreq_cols): what sg_ring_read
 * fills with channel_stride 0. */
int64_t sg_ring_bytes(const sg_handle* h);
/* Row index the next poll will overwrite (= the oldest row; the reference's `yoffset`). */
int32_t sg_ring_offset(const sg_handle* h);
/* Copies the ring unrolled, oldest row first: out[c * channel_stride + r * row + k],
 * r = 0 .. height-1 (channel_stride in bytes, 0 = height * row). */
int sg_ring_read(sg_handle* h, uint8_t* out, int64_t channel_stride, uint32_t flags, void* hip_stream);

/* ---- PCM ingest: the step before the hot path (util/util.js:2-28 decodeAudioData) ------------- */

/* Page-locked host memory for sg_push / sg_stft staging (true async DMA). */
int sg_alloc_pinned(size_t bytes, void** out);
int sg_free_pinned(void* ptr);

typedef struct sg_wav_info {
  int32_t pcm_format;      /* sg_pcm_format of the sample bytes, -1 if this build cannot ingest them */
  int32_t channels;        /* interleaved */
  int32_t sample_rate;
  int32_t bits_per_sample;
  int32_t block_align;     /* bytes per frame of `channels` samples */
  int64_t data_offset;     /* byte offset of the first sample in the file image */
  int64_t data_bytes;
  int64_t n_frames;        /* samples per channel = data_bytes / block_align */
} sg_wav_info;

/*
 * Parses a RIFF/WAVE header held in memory (host only, no HIP): PCM 16, PCM 24 and IEEE float 32, plain or
 * WAVE_FORMAT_EXTENSIBLE.  The samples then go to sg_stft as they lie in the file: pcm = bytes + data_offset,
 * a handle created with info.pcm_format, channel_stride 1 / sample_stride = channels for one channel of an
 * interleaved file, or cfg.downmix = channels for the AnalyserNode mono mix.  Stands in for what
 * decodeAudioData (util/util.js:10-17) hands the reference for uncompressed files; MP3 stays out of scope.
 */
int sg_wav_parse(const void* bytes, size_t len, sg_wav_info* out);

/* ---- frequency axes (host helpers; plain arithmetic, no device) ---------------------------------- */

/* The reference's legend scale, Util.setLogScale / lin2log / log2lin (util/util.js:32-43): y = a exp(b x),
 * fitted through (x1, y1) and (x2, y2).  The page sets setLogScale(20, 20, 20000, 20000)
 * (UI/spectrogram.js:77) and maps canvas height to Hz with it (yToFreq / freqToY, :191-221).  Checked
 * against the reference's own code run under node (tests/golden/ref_util_logscale.json). */
typedef struct sg_log_scale { double a, b; } sg_log_scale;
int sg_axis_log_scale(double x1, double y1, double x2, double y2, sg_log_scale* out);
double sg_axis_lin2log(const sg_log_scale* s, double x);
double sg_axis_log2lin(const sg_log_scale* s, double y);
/* Centre frequency of bin k (fractional bins allowed): k * sample_rate / fft_size. */
double sg_bin_hz(int32_t fft_size, double bin, double sample_rate);
/* Where column `col` of `cols` log-frequency columns samples the N/2-bin row (fractional bin index, before
 * CLAMP_TO_EDGE), and the frequency there: the mapping cfg.logfreq_cols applies
 * (x = 256^(u - 1); u = (col + 0.5) / cols for SG_LOGFREQ_PIXEL_CENTRE, col / (cols - 1) for SG_LOGFREQ_VERTEX;
 * sonogram-vertex.shader:51-52 over the mesh of 3D/visualizer.js:213-214). */
double sg_logfreq_col_bin(int32_t fft_size, int32_t cols, double col, int32_t align /* sg_logfreq_align */);
double sg_logfreq_col_hz(int32_t fft_size, int32_t cols, double col, double sample_rate, int32_t align);

/* analyser.fftSize = n on a live analyser (UI/player.js:10 sets it once, before any audio): new tables, the smoothing
 * state (and the history ring's rows, whose length is fftSize / 2) start from zero, the SAMPLE HISTORY IS KEPT -- the next
 * poll transforms the latest n samples that were pushed before the change (zeros beyond what cfg.history_samples
 * retains).  n must be a power of two in [32, 32768] and, in poll mode, <= cfg.history_samples. */
int sg_set_fft_size(sg_handle* h, int32_t fft_size);

/* analyser.smoothingTimeConstant = tau (3D/visualizer.js:351,357,362). */
int sg_set_smoothing(sg_handle* h, float tau);
/* analyser.minDecibels / maxDecibels setters. */
int sg_set_db_range(sg_handle* h, float min_db, float max_db);

/* Diagnostics (no reference counterpart; the tests use it to assert WHICH kernel path a call took).  How the
 * last sg_stft / frequency poll of this handle was launched:
 *   load_variant  0 generic loads (any stride / format / hop; 5: the same for plain rows -- unit stride, float32, no down-mix),
 *                 2 / 4 register sliding window (hop = N/4 / N/2),
 *                 1 / 3 LDS sample ring (any other hop <= N - 4 at unit stride; 2 T / 4 T granules per frame),
 *                 + 8 for int16 input, + 16 for packed int24 input
 *   run_len       frames per run (a run = consecutive frames of one channel walked by one frame slot)
 *   single_round  1 when the launch was cut into one exact dispatch round with remaining-work priorities */
typedef struct sg_launch_info {
  int32_t load_variant;
  int32_t run_len;
  int32_t single_round;
  int32_t reserved;
  int64_t total_runs;
} sg_launch_info;
int sg_last_launch(const sg_handle* h, sg_launch_info* out);

/* Tuning / test switches of one handle (no reference counterpart; nothing here changes a bit of the output, and the
 * library never reads the environment).  The parity suite uses them to force every load variant and run geometry
 * through the same inputs; tools/ uses them for A/B measurements.  Keys:
 *   "run_len"            frames per run, 0 = the launcher's choice
 *   "no_lring"           1: generic loads where the LDS sample ring would run
 *   "lring_all"          1: the 4-row ring variant at every ring size (default: only where it measured faster)
 *   "no_single_round"    1: 16-64-frame runs also for launches one dispatch round could hold
 *   "no_fused_smoothing" 1: batch smoothing through the |X| scratch + chunked scan even with many channels
 *   "no_pinned_direct"   1: page-locked buffers go through the copy engine instead of being read / written in place
 *   "no_gen_unit"        1: plain rows (unit stride, float32, no down-mix) on the generic loads run through the general instantiation
 *                           (load variant 0) instead of the one that carries the plain 8-byte loads alone (variant 5; same bits)
 *   "no_lffast"          1: 256-column log-frequency rows at fft_size 1024 / 2048 run through the general log-frequency kernel
 *                           instead of the instantiation that carries the consecutive-column path alone (same bits)
 *   "extra_lds"          bytes of dynamic LDS added to every transform launch (throttles occupancy for A/B runs)
 *   "pinned_direct_generic" 1: sg_stft reads page-locked input in place even when the generic loads will re-read the
 *                        overlap over PCIe (measurement of why it does not: tools/bench_host.py)
 * Unknown key: SG_ERR_INVALID. */
int sg_debug_set(sg_handle* h, const char* key, int64_t value);

/* Which HIP runtime this library is bound to in the running process (no reference counterpart; needs no device).  A
 * process holds ONE libamdhip64: the first one loaded serves every library that names it -- /opt/rocm's when libspectro.so
 * is loaded by itself (its DT_NEEDED + RUNPATH), a PyTorch wheel's own copy when torch was imported, or preloaded, first
 * (spectrogram_amd/_lib.py states the binding's policy).  library_path is where the runtime's own
 * entry point lives (dladdr), runtime_version / driver_version are hipRuntimeGetVersion / hipDriverGetVersion of THAT
 * library, built_hip_version the HIP_VERSION of the hipcc that compiled libspectro.so -- so a code object built by ROCm 7.2
 * running on a wheel's 7.0 runtime is visible instead of silent. */
typedef struct sg_runtime {
  int32_t runtime_version;   /* e.g. 70253211 */
  int32_t driver_version;
  int32_t built_hip_version; /* HIP_VERSION at compile time */
  int32_t reserved;
  char library_path[512];    /* "" if the loader does not say */
} sg_runtime;
int sg_runtime_info(sg_runtime* out);

/* The twiddle triplets the kernels of fft_size >= 4096 derive four of a radix-8 pass's seven twiddles from (no reference
 * counterpart; pure host arithmetic, needs no device -- exported so that the CPU suite can hold the choice to its
 * claim, tests/test_twiddles.py).  For pass ps = 1, 2, ... (twiddle index k < 8^ps, twiddles W^r = exp(-2 pi i k r /
 * 8^(ps+1))) and every k, in that order: (Re W^1, Im W^1, Re W^2, Im W^2, Re W^4, Im W^4) as float32, each component the
 * nearest float or one of its two neighbours, chosen so that W^3 = W^1 W^2, W^5 = W^1 W^4, W^6 = W^2 W^4,
 * W^7 = W^3 W^4 -- each formed as (fl(a.x w.x), fl(a.y w.x)) followed by one fused multiply-add per component, the
 * kernel's sequence -- together with the three stored values have the least squared error against the exact twiddles.
 * Returns the number of floats the table has (6 per k; 0 for fft_size < 4096) and writes min(that, cap) of them. */
int64_t sg_debug_twiddle_triplets(int32_t fft_size, float* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRO_H */
