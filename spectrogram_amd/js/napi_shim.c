/*
 * napi_shim.c -- thin N-API addon (spectro.node) over the C ABI of libspectro.so.
 *
 * This is the binding the reference's front-end would load: analyser.js wraps these calls in
 * an object with the members AnalyserView / Player touch on `context.createAnalyser()`
 * (UI/player.js:7,10,11,25,26; 3D/visualizer.js:299,301,351-363,567-569).  Buffers are the
 * caller's typed arrays, written in place and never retained past the call
 * (3D/visualizer.js:301 allocates the Uint8Array once and reuses it every frame).
 * No arithmetic here -- every number comes from the HIP kernels behind include/spectro.h.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/spectro.h"

#define NAPI_OK(call)                                              \
  do {                                                             \
    if ((call) != napi_ok) {                                       \
      napi_throw_error(env, NULL, "N-API call failed: " #call);    \
      return NULL;                                                 \
    }                                                              \
  } while (0)

static napi_value throw_sg(napi_env env, int rc) {
  const char* msg = sg_last_error();
  if (rc == SG_ERR_INVALID) napi_throw_range_error(env, "SG_ERR_INVALID", msg);
  else napi_throw_error(env, "SG_ERR", msg);
  return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
    napi_throw_type_error(env, NULL, "not enough arguments");
    return -1;
  }
  return 0;
}

/* The external wraps a box, not the handle itself, so that destroy() can release the device memory at once
 * (fftSize re-plans would otherwise leave the old handle's history / state / ring to the GC finaliser). */
typedef struct handle_box {
  sg_handle* h;
  int32_t n_channels, downmix; /* kept beside the handle: push() checks the typed array against them */
} handle_box;

static handle_box* get_box(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, NULL, "expected a native analyser handle");
    return NULL;
  }
  return (handle_box*)p;
}

static sg_handle* get_handle(napi_env env, napi_value v) {
  handle_box* b = get_box(env, v);
  if (!b) return NULL;
  if (!b->h) {
    napi_throw_error(env, "SG_ERR_STATE", "analyser handle was destroyed");
    return NULL;
  }
  return b->h;
}

/* typed array of the wanted type -> (data, length); throws TypeError otherwise */
static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* len) {
  bool is = false;
  napi_typedarray_type t;
  napi_value ab;
  size_t off;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is ||
      napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
    napi_throw_type_error(env, NULL, want == napi_uint8_array ? "expected a Uint8Array" : "expected a Float32Array");
    return -1;
  }
  return 0;
}

static void finalize_handle(napi_env env, void* data, void* hint) {
  (void)env; (void)hint;
  handle_box* b = (handle_box*)data;
  if (!b) return;
  if (b->h) sg_destroy(b->h);
  free(b);
}

/* destroy(handle): frees the engine now; later calls on the handle throw, the finaliser only frees the box */
static napi_value Destroy(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  handle_box* b = get_box(env, a[0]);
  if (!b) return NULL;
  if (b->h) {
    sg_destroy(b->h);
    b->h = NULL;
  }
  return NULL;
}

/* create(fftSize, window, nChannels, smoothing, minDb, maxDb[, logfreqCols, downmix, logfreqAlign, downmixMode]) -> external */
static napi_value Create(napi_env env, napi_callback_info info) {
  napi_value a[10];
  size_t argc = 10;
  if (napi_get_cb_info(env, info, &argc, a, NULL, NULL) != napi_ok || argc < 6) {
    napi_throw_type_error(env, NULL, "not enough arguments");
    return NULL;
  }
  int32_t fft, window, nch, lf_cols = 0, downmix = 0, lf_align = SG_LOGFREQ_PIXEL_CENTRE, downmix_mode = SG_DOWNMIX_MEAN;
  double tau, mn, mx;
  if (argc > 6) NAPI_OK(napi_get_value_int32(env, a[6], &lf_cols));
  if (argc > 7) NAPI_OK(napi_get_value_int32(env, a[7], &downmix));
  if (argc > 8) NAPI_OK(napi_get_value_int32(env, a[8], &lf_align));
  if (argc > 9) NAPI_OK(napi_get_value_int32(env, a[9], &downmix_mode));
  NAPI_OK(napi_get_value_int32(env, a[0], &fft));
  NAPI_OK(napi_get_value_int32(env, a[1], &window));
  NAPI_OK(napi_get_value_int32(env, a[2], &nch));
  NAPI_OK(napi_get_value_double(env, a[3], &tau));
  NAPI_OK(napi_get_value_double(env, a[4], &mn));
  NAPI_OK(napi_get_value_double(env, a[5], &mx));
  sg_config cfg;
  sg_config_default(&cfg);
  cfg.fft_size = fft;
  cfg.hop = fft;
  cfg.window = window;
  cfg.scale = SG_SCALE_INV_N; /* AnalyserNode divides by N */
  cfg.out_kind = SG_OUT_MAG;
  cfg.smoothing = (float)tau;
  cfg.min_db = (float)mn;
  cfg.max_db = (float)mx;
  cfg.logfreq_cols = lf_cols; /* rows of log-frequency columns instead of N/2 bins (sonogram-vertex.shader:51-52) */
  cfg.downmix = downmix;      /* push() takes K interleaved channels per analyser (the `mix` node, UI/player.js:25) */
  cfg.logfreq_align = lf_align; /* pixel centres, or the vertices of the 3-D sonogram's mesh (3D/visualizer.js:213-214) */
  cfg.downmix_mode = downmix_mode; /* the mean, or the W3C "speakers" mono mix */
  cfg.history_samples = 32768;  /* an AnalyserNode keeps its latest 32768 samples whatever fftSize is (setFftSize) */
  sg_handle* h = NULL;
  int rc = sg_create(&cfg, &h);
  if (rc) return throw_sg(env, rc);
  rc = sg_stream_begin(h, nch);
  if (rc) {
    sg_destroy(h);
    return throw_sg(env, rc);
  }
  handle_box* box = (handle_box*)malloc(sizeof *box);
  if (!box) {
    sg_destroy(h);
    napi_throw_error(env, NULL, "out of memory");
    return NULL;
  }
  box->h = h;
  box->n_channels = nch;
  box->downmix = downmix > 1 ? downmix : 1;
  napi_value ext;
  if (napi_create_external(env, box, finalize_handle, NULL, &ext) != napi_ok) {
    sg_destroy(h);
    free(box);
    napi_throw_error(env, NULL, "napi_create_external failed");
    return NULL;
  }
  return ext;
}

/* push(handle, Float32Array[nChannels * nNew * downmix], nNew) */
static napi_value Push(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  const handle_box* box = get_box(env, a[0]);
  void* data; size_t len; int32_t n_new;
  if (get_typed(env, a[1], napi_float32_array, &data, &len)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[2], &n_new));
  /* sg_push reads n_channels rows of n_new * downmix floats: all of them must lie inside the typed array */
  if (n_new < 0 || (size_t)n_new * (size_t)box->n_channels * (size_t)box->downmix > len) {
    napi_throw_range_error(env, NULL, "push: nNew * channels * downmix exceeds the array");
    return NULL;
  }
  int rc = sg_push(h, (const float*)data, n_new, 0, SG_PTR_HOST, NULL); /* dense rows: n_new * downmix floats */
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* pushPollU8 / pushPollF32(handle, Float32Array pcm, nNew, typedArray out, nChannels): one animation frame of the
 * reference as one call -- the audio that arrived since the last frame, then getByteFrequencyData / getFloatFrequencyData
 * (UI/spectrogram.js:153-161 -> 3D/visualizer.js:346-358).  sg_push_poll_* runs it as ONE launch when the block is an even
 * number of samples up to one window (small pageable blocks go through page-locked staging the handle owns). */
static napi_value push_poll_common(napi_env env, napi_callback_info info, napi_typedarray_type t) {
  napi_value a[5];
  if (get_args(env, info, 5, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  const handle_box* box = get_box(env, a[0]);
  void* pcm; size_t plen; int32_t n_new, nch;
  void* out; size_t olen;
  if (get_typed(env, a[1], napi_float32_array, &pcm, &plen)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[2], &n_new));
  if (get_typed(env, a[3], t, &out, &olen)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[4], &nch));
  if (nch < 1) nch = 1;
  if (n_new < 0 || (size_t)n_new * (size_t)box->n_channels * (size_t)box->downmix > plen) {
    napi_throw_range_error(env, NULL, "pushPoll: nNew * channels * downmix exceeds the array");
    return NULL;
  }
  int rc;
  if (olen == 0) rc = sg_push(h, (const float*)pcm, n_new, 0, SG_PTR_HOST, NULL); /* nothing to fill: the samples still arrive */
  else if (t == napi_uint8_array)
    rc = sg_push_poll_u8(h, (const float*)pcm, n_new, 0, (uint8_t*)out, (int64_t)(olen / (size_t)nch), (int64_t)(olen / (size_t)nch), SG_PTR_HOST, NULL);
  else
    rc = sg_push_poll_f32(h, (const float*)pcm, n_new, 0, (float*)out, (int64_t)(olen / (size_t)nch), (int64_t)(olen / (size_t)nch), SG_PTR_HOST, NULL);
  if (rc) return throw_sg(env, rc);
  return NULL;
}
static napi_value PushPollU8(napi_env env, napi_callback_info info) { return push_poll_common(env, info, napi_uint8_array); }
static napi_value PushPollF32(napi_env env, napi_callback_info info) { return push_poll_common(env, info, napi_float32_array); }

typedef int (*poll_fn)(sg_handle*, void*, int64_t, int64_t, uint32_t, void*);
static int poll_u8(sg_handle* h, void* o, int64_t n, int64_t s, uint32_t f, void* st) { return sg_poll_u8(h, (uint8_t*)o, n, s, f, st); }
static int poll_f32(sg_handle* h, void* o, int64_t n, int64_t s, uint32_t f, void* st) { return sg_poll_f32(h, (float*)o, n, s, f, st); }
static int poll_time(sg_handle* h, void* o, int64_t n, int64_t s, uint32_t f, void* st) { return sg_poll_time_u8(h, (uint8_t*)o, n, s, f, st); }
static int poll_time_f32(sg_handle* h, void* o, int64_t n, int64_t s, uint32_t f, void* st) { return sg_poll_time_f32(h, (float*)o, n, s, f, st); }

/* poll*(handle, typedArray, nChannels): array length is nChannels * perChannel */
static napi_value poll_common(napi_env env, napi_callback_info info, napi_typedarray_type t, poll_fn fn) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  void* data; size_t len; int32_t nch;
  if (get_typed(env, a[1], t, &data, &len)) return NULL;
  NAPI_OK(napi_get_value_int32(env, a[2], &nch));
  if (nch < 1) nch = 1;
  if (len == 0) return NULL; /* nothing to fill: like the platform */
  int rc = fn(h, data, (int64_t)(len / (size_t)nch), (int64_t)(len / (size_t)nch), SG_PTR_HOST, NULL);
  if (rc) return throw_sg(env, rc);
  return NULL; /* undefined, like getByteFrequencyData */
}
static napi_value PollU8(napi_env env, napi_callback_info info) { return poll_common(env, info, napi_uint8_array, poll_u8); }
static napi_value PollF32(napi_env env, napi_callback_info info) { return poll_common(env, info, napi_float32_array, poll_f32); }
static napi_value PollTimeU8(napi_env env, napi_callback_info info) { return poll_common(env, info, napi_uint8_array, poll_time); }
static napi_value PollTimeF32(napi_env env, napi_callback_info info) { return poll_common(env, info, napi_float32_array, poll_time_f32); }

static napi_value SetSmoothing(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  double tau;
  NAPI_OK(napi_get_value_double(env, a[1], &tau));
  int rc = sg_set_smoothing(h, (float)tau);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* setFftSize(handle, n): analyser.fftSize = n on a live analyser -- the sample history stays, smoothing restarts */
static napi_value SetFftSize(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  int32_t n;
  NAPI_OK(napi_get_value_int32(env, a[1], &n));
  int rc = sg_set_fft_size(h, n);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

static napi_value SetDbRange(napi_env env, napi_callback_info info) {
  napi_value a[3];
  if (get_args(env, info, 3, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  double mn, mx;
  NAPI_OK(napi_get_value_double(env, a[1], &mn));
  NAPI_OK(napi_get_value_double(env, a[2], &mx));
  int rc = sg_set_db_range(h, (float)mn, (float)mx);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* ---- the history texture AnalyserView keeps (3D/visualizer.js:60,293-331,390-417), held on the device ---- */

/* ringBegin(handle, height): allocate the [channels][height][bins] byte ring; every later pollU8 deposits its row */
static napi_value RingBegin(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  int32_t height;
  NAPI_OK(napi_get_value_int32(env, a[1], &height));
  int rc = sg_ring_begin(h, height);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* ringSetPinned(handle, bool): row 0 only, yoffset stays 0 (the reference's 2-D modes, 3D/visualizer.js:385-390) */
static napi_value RingSetPinned(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  bool pinned;
  NAPI_OK(napi_get_value_bool(env, a[1], &pinned));
  int rc = sg_ring_set_pinned(h, pinned ? 1 : 0);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* ringOffset(handle) -> the reference's `yoffset` */
static napi_value RingOffset(napi_env env, napi_callback_info info) {
  napi_value a[1];
  if (get_args(env, info, 1, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  int32_t off = sg_ring_offset(h);
  if (off < 0) return throw_sg(env, off);
  napi_value v;
  NAPI_OK(napi_create_int32(env, off, &v));
  return v;
}

/* ringRead(handle, Uint8Array[channels * height * bins]): the ring unrolled, oldest row first, into the caller's array */
static napi_value RingRead(napi_env env, napi_callback_info info) {
  napi_value a[2];
  if (get_args(env, info, 2, a)) return NULL;
  sg_handle* h = get_handle(env, a[0]);
  if (!h) return NULL;
  void* data; size_t len;
  if (get_typed(env, a[1], napi_uint8_array, &data, &len)) return NULL;
  int64_t expect = sg_ring_bytes(h);
  if (expect < 0) return throw_sg(env, (int)expect);
  if ((int64_t)len != expect) { napi_throw_range_error(env, NULL, "ringRead expects channels * height * rowLength bytes"); return NULL; }
  int rc = sg_ring_read(h, (uint8_t*)data, 0, SG_PTR_HOST, NULL);
  if (rc) return throw_sg(env, rc);
  return NULL;
}

/* stft(Float32Array pcm, fftSize, hop, window, scale, outKind) -> Float32Array | Uint8Array, mono batch */
static napi_value Stft(napi_env env, napi_callback_info info) {
  napi_value a[6];
  if (get_args(env, info, 6, a)) return NULL;
  void* data; size_t len;
  if (get_typed(env, a[0], napi_float32_array, &data, &len)) return NULL;
  int32_t v[5];
  for (int i = 0; i < 5; ++i) NAPI_OK(napi_get_value_int32(env, a[i + 1], &v[i]));
  sg_config cfg;
  sg_config_default(&cfg);
  cfg.fft_size = v[0]; cfg.hop = v[1]; cfg.window = v[2]; cfg.scale = v[3]; cfg.out_kind = v[4];
  sg_handle* h = NULL;
  int rc = sg_create(&cfg, &h);
  if (rc) return throw_sg(env, rc);
  const int64_t frames = sg_num_frames(h, (int64_t)len);
  const int bins = sg_num_bins(h), esz = sg_out_elem_size(h);
  napi_value ab, out;
  void* obuf = NULL;
  if (napi_create_arraybuffer(env, (size_t)(frames * bins * esz), &obuf, &ab) != napi_ok) { sg_destroy(h); return NULL; }
  rc = sg_stft(h, (const float*)data, 1, (int64_t)len, (int64_t)len, 1, obuf, 0, SG_PTR_HOST, NULL);
  sg_destroy(h);
  if (rc) return throw_sg(env, rc);
  NAPI_OK(napi_create_typedarray(env, esz == 1 ? napi_uint8_array : napi_float32_array, (size_t)(frames * bins), ab, 0, &out));
  return out;
}

static napi_value Version(napi_env env, napi_callback_info info) {
  (void)info;
  napi_value v;
  NAPI_OK(napi_create_int32(env, sg_version(), &v));
  return v;
}

/* runtimeInfo() -> { libraryPath, runtimeVersion, driverVersion, builtWithHipVersion }: which libamdhip64 the process is on
 * (sg_runtime_info).  Under node nothing loads a runtime before libspectro.so does: the system's, by its RUNPATH. */
static napi_value RuntimeInfo(napi_env env, napi_callback_info info) {
  (void)info;
  sg_runtime rt;
  int rc = sg_runtime_info(&rt);
  if (rc) return throw_sg(env, rc);
  napi_value o, v;
  NAPI_OK(napi_create_object(env, &o));
  NAPI_OK(napi_create_string_utf8(env, rt.library_path, NAPI_AUTO_LENGTH, &v));
  NAPI_OK(napi_set_named_property(env, o, "libraryPath", v));
  NAPI_OK(napi_create_int32(env, rt.runtime_version, &v));
  NAPI_OK(napi_set_named_property(env, o, "runtimeVersion", v));
  NAPI_OK(napi_create_int32(env, rt.driver_version, &v));
  NAPI_OK(napi_set_named_property(env, o, "driverVersion", v));
  NAPI_OK(napi_create_int32(env, rt.built_hip_version, &v));
  NAPI_OK(napi_set_named_property(env, o, "builtWithHipVersion", v));
  return o;
}

static napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor d[] = {
      {"create", 0, Create, 0, 0, 0, napi_default, 0},       {"push", 0, Push, 0, 0, 0, napi_default, 0},
      {"pollU8", 0, PollU8, 0, 0, 0, napi_default, 0},       {"pollF32", 0, PollF32, 0, 0, 0, napi_default, 0},
      {"pollTimeU8", 0, PollTimeU8, 0, 0, 0, napi_default, 0}, {"pollTimeF32", 0, PollTimeF32, 0, 0, 0, napi_default, 0}, {"setSmoothing", 0, SetSmoothing, 0, 0, 0, napi_default, 0},
      {"setDbRange", 0, SetDbRange, 0, 0, 0, napi_default, 0}, {"stft", 0, Stft, 0, 0, 0, napi_default, 0},
      {"version", 0, Version, 0, 0, 0, napi_default, 0},       {"destroy", 0, Destroy, 0, 0, 0, napi_default, 0},
      {"ringBegin", 0, RingBegin, 0, 0, 0, napi_default, 0},   {"ringSetPinned", 0, RingSetPinned, 0, 0, 0, napi_default, 0},
      {"ringOffset", 0, RingOffset, 0, 0, 0, napi_default, 0}, {"ringRead", 0, RingRead, 0, 0, 0, napi_default, 0},
      {"setFftSize", 0, SetFftSize, 0, 0, 0, napi_default, 0}, {"runtimeInfo", 0, RuntimeInfo, 0, 0, 0, napi_default, 0},
      {"pushPollU8", 0, PushPollU8, 0, 0, 0, napi_default, 0}, {"pushPollF32", 0, PushPollF32, 0, 0, 0, napi_default, 0},
  };
  if (napi_define_properties(env, exports, sizeof d / sizeof d[0], d) != napi_ok) return NULL;
  return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
