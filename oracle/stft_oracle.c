/*
 * stft_oracle.c -- float64 CPU oracle (see stft_oracle.h: TEST INFRASTRUCTURE
 * ONLY, PARITY UNPINNED).
 *
 * Deliberately the plainest possible algorithm -- textbook iterative radix-2
 * Cooley-Tukey on a full complex N-point transform, no real-input packing, no
 * high-radix passes -- so that it shares no structure (and no bugs) with the
 * HIP kernels it checks.
 */
#include "stft_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

/* Window definitions.
 * Hann (north_star): w[n] = 0.5 - 0.5 cos(2 pi n / N), periodic.
 * Blackman (W3C AnalyserNode, alpha = 0.16): a0 = 0.42, a1 = 0.5, a2 = 0.08,
 * w[n] = a0 - a1 cos(2 pi n / N) + a2 cos(4 pi n / N), periodic.
 * The reference never picks a window itself (UI/player.js:7-11 only sets
 * fftSize and smoothingTimeConstant), so both are provided. */
int so_window(int kind, int n, double* w) {
  if (n <= 0 || !w) return -1;
  for (int i = 0; i < n; ++i) {
    double a = 2.0 * M_PI * (double)i / (double)n;
    switch (kind) {
      case SO_WINDOW_HANN: w[i] = 0.5 - 0.5 * cos(a); break;
      case SO_WINDOW_BLACKMAN: w[i] = 0.42 - 0.5 * cos(a) + 0.08 * cos(2.0 * a); break;
      case SO_WINDOW_RECT: w[i] = 1.0; break;
      default: return -1;
    }
  }
  return 0;
}

/* Iterative radix-2 DIT: bit-reversal permutation, then log2(n) butterfly
 * stages with twiddles e^{-2 pi i j / len}. */
int so_fft_radix2(double* re, double* im, int n) {
  if (!is_pow2(n)) return -1;
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1;
    for (int j = 0; j < half; ++j) {
      double a = -2.0 * M_PI * (double)j / (double)len;
      double wr = cos(a), wi = sin(a);
      for (int i = j; i < n; i += len) {
        int k = i + half;
        double xr = re[k] * wr - im[k] * wi;
        double xi = re[k] * wi + im[k] * wr;
        re[k] = re[i] - xr; im[k] = im[i] - xi;
        re[i] += xr;        im[i] += xi;
      }
    }
  }
  return 0;
}

int64_t so_num_frames(int64_t n_samples, int fft_size, int hop) {
  if (fft_size <= 0 || hop <= 0 || n_samples < fft_size) return 0;
  return (n_samples - fft_size) / hop + 1;
}

int so_num_bins(const so_config* cfg) {
  return cfg->fft_size / 2 + (cfg->nyquist ? 1 : 0);
}

static int check_cfg(const so_config* c) {
  if (!c || !is_pow2(c->fft_size) || c->fft_size < 4 || c->hop < 1) return -1;
  if (c->window < 0 || c->window > SO_WINDOW_RECT) return -1;
  if (c->scale < 0 || c->scale > SO_SCALE_INV_N) return -1;
  if (c->out_kind < 0 || c->out_kind > SO_OUT_DB_U8) return -1;
  if (c->out_kind == SO_OUT_DB_U8 && !(c->max_db > c->min_db)) return -1;
  return 0;
}

/* Byte mapping of the AnalyserNode contract: b = floor(255/(max-min) * (Y - min))
 * clipped to [0,255]; this is what lands in the reference's Uint8Array
 * (3D/visualizer.js:301,358) and then in its 1024x256 texture (:399-409). */
static uint8_t db_to_byte(double y, double min_db, double max_db) {
  double v = floor(255.0 / (max_db - min_db) * (y - min_db));
  if (!(v > 0.0)) return 0; /* also catches NaN */
  if (v > 255.0) return 255;
  return (uint8_t)v;
}

int64_t so_stft(const so_config* cfg, const float* pcm, int64_t n_samples,
                int64_t sample_stride, double* out_f64, uint8_t* out_u8) {
  if (check_cfg(cfg) || !pcm || !out_f64 || sample_stride < 1) return -1;
  const int n = cfg->fft_size;
  const int bins = so_num_bins(cfg);
  const int64_t frames = so_num_frames(n_samples, n, cfg->hop);
  double* w = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  if (!w) return -2;
  double* re = w + n;
  double* im = re + n;
  so_window(cfg->window, n, w);
  const double s = cfg->scale == SO_SCALE_INV_N ? 1.0 / (double)n : 1.0;
  for (int64_t f = 0; f < frames; ++f) {
    const float* x = pcm + f * (int64_t)cfg->hop * sample_stride;
    for (int i = 0; i < n; ++i) {
      re[i] = w[i] * (double)x[(int64_t)i * sample_stride];
      im[i] = 0.0;
    }
    so_fft_radix2(re, im, n);
    double* o = out_f64 + f * bins;
    for (int k = 0; k < bins; ++k) {
      double xr = re[k] * s, xi = im[k] * s;
      double p = xr * xr + xi * xi;
      switch (cfg->out_kind) {
        case SO_OUT_POWER: o[k] = p; break;
        case SO_OUT_MAG: o[k] = sqrt(p); break;
        default: {
          double y = p > 0.0 ? 10.0 * log10(p) : -INFINITY;
          if (y < cfg->floor_db) y = cfg->floor_db;
          o[k] = y;
          if (cfg->out_kind == SO_OUT_DB_U8 && out_u8)
            out_u8[f * bins + k] = db_to_byte(y, cfg->min_db, cfg->max_db);
        }
      }
    }
  }
  free(w);
  return frames;
}

/* float32 twin used only as the timed CPU baseline ("the per-frame FFT a JS
 * front-end would run", BASELINE.md section 4, fallback B2). */
static void fft_radix2_f32(float* re, float* im, int n, const float* twr, const float* twi) {
  for (int i = 1, j = 0; i < n; ++i) {
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      float t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len) {
      for (int j = 0; j < half; ++j) {
        float wr = twr[j * step], wi = twi[j * step];
        int a = i + j, b = a + half;
        float xr = re[b] * wr - im[b] * wi;
        float xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr; im[b] = im[a] - xi;
        re[a] += xr;        im[a] += xi;
      }
    }
  }
}

int64_t so_stft_f32(const so_config* cfg, const float* pcm, int64_t n_samples,
                    int64_t sample_stride, float* out) {
  if (check_cfg(cfg) || !pcm || !out || sample_stride < 1) return -1;
  if (cfg->out_kind == SO_OUT_DB_U8) return -1;
  const int n = cfg->fft_size;
  const int bins = so_num_bins(cfg);
  const int64_t frames = so_num_frames(n_samples, n, cfg->hop);
  double* wd = (double*)malloc(sizeof(double) * (size_t)n);
  float* buf = (float*)malloc(sizeof(float) * 5 * (size_t)n);
  if (!wd || !buf) { free(wd); free(buf); return -2; }
  float *w = buf, *re = buf + n, *im = re + n, *twr = im + n, *twi = twr + n;
  so_window(cfg->window, n, wd);
  const float s = cfg->scale == SO_SCALE_INV_N ? 1.0f / (float)n : 1.0f;
  for (int i = 0; i < n; ++i) {
    w[i] = (float)wd[i] * s;
    double a = -2.0 * M_PI * (double)i / (double)n;
    twr[i] = (float)cos(a); twi[i] = (float)sin(a);
  }
  for (int64_t f = 0; f < frames; ++f) {
    const float* x = pcm + f * (int64_t)cfg->hop * sample_stride;
    for (int i = 0; i < n; ++i) { re[i] = w[i] * x[(int64_t)i * sample_stride]; im[i] = 0.0f; }
    fft_radix2_f32(re, im, n, twr, twi);
    float* o = out + f * bins;
    for (int k = 0; k < bins; ++k) {
      float p = re[k] * re[k] + im[k] * im[k];
      if (cfg->out_kind == SO_OUT_POWER) o[k] = p;
      else if (cfg->out_kind == SO_OUT_MAG) o[k] = sqrtf(p);
      else {
        float y = p > 0.0f ? 10.0f * log10f(p) : -INFINITY;
        o[k] = y < (float)cfg->floor_db ? (float)cfg->floor_db : y;
      }
    }
  }
  free(wd); free(buf);
  return frames;
}

/* The float32 radix-2 JS FFT north_star names as the comparison ("<= 1e-4 max relative error vs the JS
 * reference"): oracle/stft_oracle.js run on Float32Array, restated statement by statement.  A JS engine evaluates every
 * expression in double and rounds when the value is stored into the Float32Array, so: storage float, each stored value
 * one double expression rounded once.  tests/test_oracle.py holds this function to the JS file itself under node, bit
 * for bit.  It is the accuracy yardstick of the stated tolerance, never a product path. */
int64_t so_stft_f32js(const so_config* cfg, const float* pcm, int64_t n_samples,
                      int64_t sample_stride, float* out) {
  if (check_cfg(cfg) || !pcm || !out || sample_stride < 1) return -1;
  if (cfg->out_kind != SO_OUT_POWER || cfg->nyquist) return -1; /* stftPower() emits N/2 power bins */
  const int n = cfg->fft_size, bins = n / 2;
  const int64_t frames = so_num_frames(n_samples, n, cfg->hop);
  double* wd = (double*)malloc(sizeof(double) * (size_t)n);
  float* buf = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  if (!wd || !buf) { free(wd); free(buf); return -2; }
  float *wt = buf, *re = buf + n, *im = re + n, *cs = im + n, *sn = cs + n / 2;
  so_window(cfg->window, n, wd);
  const double scale = cfg->scale == SO_SCALE_INV_N ? 1.0 / (double)n : 1.0;
  for (int i = 0; i < n; ++i) wt[i] = (float)(wd[i] * scale);
  for (int i = 0; i < n / 2; ++i) {
    cs[i] = (float)cos(2.0 * M_PI * (double)i / (double)n);
    sn[i] = (float)(-sin(2.0 * M_PI * (double)i / (double)n));
  }
  for (int64_t f = 0; f < frames; ++f) {
    const float* x = pcm + f * (int64_t)cfg->hop * sample_stride;
    for (int i = 0; i < n; ++i) { re[i] = (float)((double)wt[i] * (double)x[(int64_t)i * sample_stride]); im[i] = 0.0f; }
    for (int i = 1, j = 0; i < n; ++i) {
      int bit = n >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) {
        float t = re[i]; re[i] = re[j]; re[j] = t;
        t = im[i]; im[i] = im[j]; im[j] = t;
      }
    }
    for (int len = 2; len <= n; len <<= 1) {
      const int half = len >> 1, step = n / len;
      for (int i = 0; i < n; i += len) {
        for (int j = 0, k = 0; j < half; ++j, k += step) {
          const double wr = cs[k], wi = sn[k];
          const int a = i + j, b = a + half;
          const double xr = (double)re[b] * wr - (double)im[b] * wi, xi = (double)re[b] * wi + (double)im[b] * wr;
          re[b] = (float)((double)re[a] - xr); im[b] = (float)((double)im[a] - xi);
          re[a] = (float)((double)re[a] + xr); im[a] = (float)((double)im[a] + xi);
        }
      }
    }
    float* o = out + f * bins;
    for (int k = 0; k < bins; ++k) o[k] = (float)((double)re[k] * (double)re[k] + (double)im[k] * (double)im[k]);
  }
  free(wd); free(buf);
  return frames;
}

/* AnalyserNode poll, W3C Web Audio "FFT windowing and smoothing over time":
 *   x^[n] = w[n] x[n]  (Blackman, alpha 0.16)
 *   X[k]  = (1/N) sum_n x^[n] e^{-2 pi i k n / N},  k = 0 .. N/2-1
 *   X^[k] = tau X^_prev[k] + (1 - tau) |X[k]|
 *   Y[k]  = 20 log10 X^[k]
 *   b[k]  = clip(floor(255/(max-min) (Y[k]-min)), 0, 255)
 * Call sites being served: 3D/visualizer.js:351-352 (tau=0.75),
 * :357-358 (tau=0, the default sonogram mode), UI/player.js:10-11. */
int so_analyser_poll(int fft_size, const float* last_n, double tau,
                     double min_db, double max_db, double* smooth_state,
                     double* db_out, uint8_t* byte_out) {
  if (!is_pow2(fft_size) || fft_size < 4 || !last_n || !smooth_state) return -1;
  const int n = fft_size, bins = n / 2;
  double* w = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  if (!w) return -2;
  double *re = w + n, *im = re + n;
  so_window(SO_WINDOW_BLACKMAN, n, w);
  for (int i = 0; i < n; ++i) { re[i] = w[i] * (double)last_n[i]; im[i] = 0.0; }
  so_fft_radix2(re, im, n);
  for (int k = 0; k < bins; ++k) {
    double mag = sqrt(re[k] * re[k] + im[k] * im[k]) / (double)n;
    double sm = tau * smooth_state[k] + (1.0 - tau) * mag;
    if (!isfinite(sm)) sm = 0.0; /* spec: NaN/inf in the smoothed value resets to 0 */
    smooth_state[k] = sm;
    double y = sm > 0.0 ? 20.0 * log10(sm) : -INFINITY;
    if (db_out) db_out[k] = y;
    if (byte_out) byte_out[k] = db_to_byte(y, min_db, max_db);
  }
  free(w);
  return 0;
}

int so_time_domain_bytes(const float* x, int n, uint8_t* out) {
  if (!x || !out || n < 0) return -1;
  for (int i = 0; i < n; ++i) {
    double v = floor(128.0 * (1.0 + (double)x[i]));
    out[i] = (uint8_t)(v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
  }
  return 0;
}

/* GL_LINEAR / CLAMP_TO_EDGE sampling of a 1-D row at normalised x in [0,1]:
 * texel centres sit at (i + 0.5)/n. */
static double sample_linear(const double* row, int n, double x) {
  double p = x * (double)n - 0.5;
  double fl = floor(p);
  double fr = p - fl;
  int i0 = (int)fl, i1 = i0 + 1;
  if (i0 < 0) i0 = 0;
  if (i1 < 0) i1 = 0;
  if (i0 > n - 1) i0 = n - 1;
  if (i1 > n - 1) i1 = n - 1;
  return row[i0] * (1.0 - fr) + row[i1] * fr;
}

/* align 0: pixel centres u = (c + 0.5) / n_cols (sonogram-fragment.shader:16); align 1: the vertices of the reference's
 * 3-D sonogram mesh, u = x / (sonogram3DWidth - 1), x = 0 .. width - 1 (3D/visualizer.js:213-214), whose vertex shader
 * does the lookup (sonogram-vertex.shader:51-52): u = 1 gives x = 1, which CLAMP_TO_EDGE (:313) makes the last texel. */
int so_logfreq_resample_at(const double* row, int n_bins, int n_cols, int align, double* out) {
  if (!row || !out || n_bins < 1 || n_cols < 1 || align < 0 || align > 1) return -1;
  for (int c = 0; c < n_cols; ++c) {
    double u = align == 1 ? (n_cols > 1 ? (double)c / (double)(n_cols - 1) : 0.0) : ((double)c + 0.5) / (double)n_cols;
    double x = pow(256.0, u - 1.0);
    out[c] = sample_linear(row, n_bins, x);
  }
  return 0;
}

int so_logfreq_resample(const double* row, int n_bins, int n_cols, double* out) {
  return so_logfreq_resample_at(row, n_bins, n_cols, 0, out);
}
