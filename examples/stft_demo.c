/*
 * stft_demo.c -- the C ABI of libspectro.so on its own: no Python, no torch, no node.
 *
 *   gcc -O2 -Iinclude examples/stft_demo.c -o stft_demo -Lspectrogram_amd/lib -lspectro \
 *       -Wl,-rpath,$PWD/spectrogram_amd/lib -Wl,-rpath,/opt/rocm/lib -lm
 *
 * Part 1 is what the reference's draw loop does with its analyser (UI/player.js:7-11,
 * 3D/visualizer.js:357-358): create, fftSize 2048, smoothing 0, feed audio, poll bytes.
 * Part 2 is the batch form over a recorded signal (BASELINE.json configs[1] shape, 1 s of it).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "spectro.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != SG_OK) {                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sg_last_error());      \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(void) {
  const double two_pi = 6.283185307179586;
  const int sr = 48000;

  /* ---- 1. analyser poll, as the reference drives it */
  sg_config cfg;
  CHECK(sg_config_default(&cfg));
  cfg.fft_size = 2048;                 /* UI/player.js:10 */
  cfg.hop = 2048;
  cfg.window = SG_WINDOW_BLACKMAN;     /* AnalyserNode numerics */
  cfg.scale = SG_SCALE_INV_N;
  cfg.smoothing = 0.f;                 /* UI/player.js:11 */
  sg_handle* an = NULL;
  CHECK(sg_create(&cfg, &an));
  CHECK(sg_stream_begin(an, 1));
  float* tone = (float*)malloc(sizeof(float) * 2048);
  const double f_tone = 3000.0;
  for (int i = 0; i < 2048; ++i) tone[i] = (float)(0.5 * sin(two_pi * f_tone * i / sr));
  CHECK(sg_push(an, tone, 2048, 0, SG_PTR_HOST, NULL));
  uint8_t bytes[1024];                 /* Uint8Array(frequencyBinCount), 3D/visualizer.js:301 */
  CHECK(sg_poll_u8(an, bytes, 1024, 0, SG_PTR_HOST, NULL));
  int peak = 0;
  for (int k = 1; k < 1024; ++k) if (bytes[k] > bytes[peak]) peak = k;
  const int expect = (int)(f_tone / sr * 2048 + 0.5);
  printf("analyser: %d bins, peak byte %d at bin %d (expected bin %d)\n", sg_num_bins(an), bytes[peak], peak, expect);
  if (abs(peak - expect) > 1) { fprintf(stderr, "peak in the wrong bin\n"); return 2; }
  CHECK(sg_destroy(an));

  /* ---- 2. batch STFT of one second of a sweep, host buffers */
  CHECK(sg_config_default(&cfg));      /* N = 1024, hop = 256, Hann, |X|^2 */
  sg_handle* h = NULL;
  CHECK(sg_create(&cfg, &h));
  float* pcm = (float*)malloc(sizeof(float) * sr);
  for (int i = 0; i < sr; ++i) {
    const double t = (double)i / sr;
    pcm[i] = (float)(0.5 * sin(two_pi * (20.0 * t + (20000.0 - 20.0) * t * t / 2.0)));
  }
  const int64_t frames = sg_num_frames(h, sr);
  const int bins = sg_num_bins(h);
  float* out = (float*)malloc(sizeof(float) * (size_t)frames * bins);
  CHECK(sg_stft(h, pcm, 1, sr, sr, 1, out, 0, SG_PTR_HOST, NULL));
  /* the sweep's peak bin must rise monotonically-ish from first to last frame */
  int first = 0, last = 0;
  for (int k = 1; k < bins; ++k) {
    if (out[k] > out[first]) first = k;
    if (out[(frames - 1) * bins + k] > out[(frames - 1) * bins + last]) last = k;
  }
  printf("batch: %lld frames x %d bins, peak bin %d -> %d\n", (long long)frames, bins, first, last);
  if (!(frames == 184 && last > first + 300)) { fprintf(stderr, "unexpected sweep spectrum\n"); return 3; }
  CHECK(sg_destroy(h));
  free(tone); free(pcm); free(out);
  printf("ok\n");
  return 0;
}
