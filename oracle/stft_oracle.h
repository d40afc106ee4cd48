/*
 * stft_oracle.h -- CPU oracle for the STFT / analyser hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may link or call it, and there only as the checker.
 *
 * PARITY UNPINNED.  The reference (amilajack/spectrogram) contains no FFT,
 * window, framing, magnitude or dB code; it delegates the whole transform to
 * the browser's Web Audio AnalyserNode (created UI/player.js:7, configured
 * UI/player.js:10-11, polled 3D/visualizer.js:352,358,363).  That platform
 * code is a third-party dependency with no pinned version (it is a browser
 * built-in, absent from package.json / yarn.lock) and the reference has no
 * tests, golden vectors or fixtures for it.  This file therefore restates
 *   (1) the pipeline BASELINE.json's north_star prescribes: Hann window ->
 *       hop-overlapped framing -> radix-2 Cooley-Tukey FFT -> |X|^2 /
 *       log-magnitude per hop, and
 *   (2) the published W3C Web Audio AnalyserNode algorithm for the byte
 *       encoding the reference's consumer expects (Blackman alpha=0.16,
 *       1/N scale, tau smoothing, 20*log10, [minDb,maxDb] -> [0,255]),
 * in float64, and is cross-checked in the build container against
 * numpy.fft.rfft / scipy.signal.get_window (oracle/make_golden.py) and
 * against analytic known-answer tests (tests/test_oracle.py).
 */
#ifndef STFT_ORACLE_H
#define STFT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same numeric values as include/spectro.h */
enum { SO_WINDOW_HANN = 0, SO_WINDOW_BLACKMAN = 1, SO_WINDOW_RECT = 2 };
enum { SO_SCALE_NONE = 0, SO_SCALE_INV_N = 1 };
enum { SO_OUT_POWER = 0, SO_OUT_MAG = 1, SO_OUT_DB = 2, SO_OUT_DB_U8 = 3 };

typedef struct so_config {
  int fft_size;     /* N, power of two, 32..32768 (AnalyserNode range) */
  int hop;          /* >= 1 */
  int window;       /* SO_WINDOW_* */
  int scale;        /* SO_SCALE_* */
  int out_kind;     /* SO_OUT_* */
  int nyquist;      /* 0: N/2 bins (frequencyBinCount, visualizer.js:301); 1: N/2+1 */
  double min_db;    /* byte mapping lower edge  (AnalyserNode default -100) */
  double max_db;    /* byte mapping upper edge  (AnalyserNode default  -30) */
  double floor_db;  /* clamp for the dB outputs so log(0) stays finite      */
} so_config;

/* periodic window of length n into w[0..n) */
int so_window(int kind, int n, double* w);

/* in-place iterative radix-2 decimation-in-time FFT, forward sign e^{-2 pi i kn/N} */
int so_fft_radix2(double* re, double* im, int n);

/* frames = floor((n_samples - N)/hop) + 1, or 0 when n_samples < N (no padding) */
int64_t so_num_frames(int64_t n_samples, int fft_size, int hop);
int so_num_bins(const so_config* cfg);

/*
 * One channel.  pcm[i * sample_stride], i in [0, n_samples).
 * out_f64: frames x bins doubles (POWER / MAG / DB, and the un-floored byte
 * value for DB_U8); out_u8: frames x bins bytes, only written for DB_U8
 * (may be NULL otherwise).  Returns the number of frames or < 0 on error.
 */
int64_t so_stft(const so_config* cfg, const float* pcm, int64_t n_samples,
                int64_t sample_stride, double* out_f64, uint8_t* out_u8);

/* Same pipeline with float32 storage and arithmetic: the "per-frame FFT a
 * front-end would run" -- used ONLY as the timed CPU baseline in bench.py. */
int64_t so_stft_f32(const so_config* cfg, const float* pcm, int64_t n_samples,
                    int64_t sample_stride, float* out_f32);

/* oracle/stft_oracle.js on Float32Array, restated: float32 storage, every stored value one double expression rounded
 * once (what a JS engine does).  POWER, N/2 bins only.  This is the "float32 radix-2 JS FFT" of north_star: the
 * yardstick the stated tolerance is anchored to (the HIP kernel's error against the float64 oracle must not exceed this
 * function's on the same frames; tests/test_parity_gpu.py).  Bit-equal to the JS file under node
 * (tests/test_oracle.py). */
int64_t so_stft_f32js(const so_config* cfg, const float* pcm, int64_t n_samples,
                      int64_t sample_stride, float* out_f32);

/*
 * AnalyserNode poll (W3C Web Audio): one frame over the latest N samples.
 * smooth_state[bins] carries X^[k] between polls and is updated in place.
 * Any of db_out / byte_out may be NULL.
 */
int so_analyser_poll(int fft_size, const float* last_n, double tau,
                     double min_db, double max_db, double* smooth_state,
                     double* db_out, uint8_t* byte_out);

/* getByteTimeDomainData: b = clamp(floor(128 * (1 + x)), 0, 255) */
int so_time_domain_bytes(const float* x, int n, uint8_t* out);

/*
 * Log-frequency column resampling the reference's shaders apply to a row of
 * bins (sonogram-vertex.shader:51-52, sonogram-fragment.shader:16-19, with
 * LINEAR/CLAMP_TO_EDGE sampling, visualizer.js:313-316): column c of n_cols
 * samples the row at normalised x = 256^(u-1), u = (c + 0.5)/n_cols.
 */
int so_logfreq_resample(const double* row, int n_bins, int n_cols, double* out);
/* ... or, align = 1, at the vertices of the reference's 3-D sonogram mesh: u = c / (n_cols - 1), c = 0 .. n_cols - 1
 * (3D/visualizer.js:54,213-214: texCoord.x = x / (sonogram3DWidth - 1)); align = 0 is the function above. */
int so_logfreq_resample_at(const double* row, int n_bins, int n_cols, int align, double* out);

#ifdef __cplusplus
}
#endif
#endif
