// shim_driver.cpp -- TEST INFRASTRUCTURE ONLY: drives the C ABI of include/spectro.h against the host side of the product
// (csrc/spectro_api.hip, compiled host-only with AddressSanitizer + UBSan) on top of tests/host/hip_stub.cpp and
// tests/host/launch_stub.cpp -- no GPU, no transform, every buffer with exact bounds.  What it pins down is the code between
// the caller and the kernels: argument checks, staging growth and reuse, padded outputs, the history ring's positions under
// ragged pushes (read back sample for sample), the fused push + poll's bookkeeping, the byte ring's row order, the
// transactional sg_set_fft_size under injected allocation failures, and that a destroyed handle leaves nothing behind.
//   make -C oracle check-shim-asan
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/spectro.h"

extern "C" {
long sg_stub_errors(void);
long sg_stub_violations(void);
long sg_stub_launch_seq(void);
long sg_stub_live_device_blocks(void);
long sg_stub_live_host_blocks(void);
void sg_stub_fail_malloc_at(int nth);
void sg_stub_set_device_count(int n);
int sg_stub_current_device(void);
long sg_stub_blocks_on_device(int d);
}
extern "C" int hipSetDevice(int d);   // the caller's side of the HIP runtime (hip_stub.cpp): 0 = hipSuccess

static long g_checks = 0;
#define CHECK(cond)                                                                          \
  do {                                                                                       \
    ++g_checks;                                                                              \
    if (!(cond)) {                                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s   (last error: %s)\n", __FILE__, __LINE__, #cond, sg_last_error()); \
      std::exit(1);                                                                          \
    }                                                                                        \
  } while (0)

static uint64_t g_rng = 0x5EED0006ULL;
static uint32_t rnd() {
  g_rng = g_rng * 6364136223846793005ULL + 1442695040888963407ULL;
  return (uint32_t)(g_rng >> 33);
}
static int rnd_in(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

static sg_config cfg_of(int n, int hop) {
  sg_config c;
  CHECK(sg_config_default(&c) == SG_OK);
  c.fft_size = n;
  c.hop = hop;
  return c;
}

static void test_configs() {
  sg_handle* h = nullptr;
  sg_config c = cfg_of(1024, 256);
  sg_stub_set_device_count(0);
  CHECK(sg_create(&c, &h) == SG_ERR_HIP && h == nullptr);       // no device: no CPU path, a loud error
  sg_stub_set_device_count(1);
  CHECK(sg_create(nullptr, &h) == SG_ERR_INVALID && sg_create(&c, nullptr) == SG_ERR_INVALID);
  c.struct_size -= 4;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  for (int bad : {0, 16, 31, 33, 1000, 65536, -1024}) {
    c = cfg_of(bad, 8);
    CHECK(sg_create(&c, &h) == SG_ERR_INVALID && h == nullptr);
  }
  c = cfg_of(1024, 0);
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.window = 7;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.downmix = 65;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.downmix_mode = 2;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.smoothing = 1.5f;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.logfreq_cols = 64; c.nyquist = 1;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(1024, 256); c.history_samples = 512;
  CHECK(sg_create(&c, &h) == SG_ERR_INVALID);
  c = cfg_of(32768, 8192); c.logfreq_cols = 2049;               // the column table must fit in LDS behind the exchange buffers
  CHECK(sg_create(&c, &h) == SG_ERR_UNSUPPORTED);
  c.logfreq_cols = 2048;
  CHECK(sg_create(&c, &h) == SG_OK && sg_num_bins(h) == 2048 && sg_destroy(h) == SG_OK);
  // every size builds its tables (the derived twiddle triplets of N >= 4096 included) and frees them again
  for (int l2 = 5; l2 <= 15; ++l2) {
    c = cfg_of(1 << l2, 1 << (l2 - 2));
    c.downmix = l2 == 9 ? 6 : 0;
    c.downmix_mode = SG_DOWNMIX_SPEAKERS;
    CHECK(sg_create(&c, &h) == SG_OK);
    CHECK(sg_num_bins(h) == (1 << l2) / 2 && sg_num_frames(h, (1 << l2) - 1) == 0 && sg_num_frames(h, (1 << l2)) == 1);
    CHECK(sg_destroy(h) == SG_OK);
  }
  CHECK(sg_stub_live_device_blocks() == 0);
  sg_runtime rt;
  CHECK(sg_runtime_info(&rt) == SG_OK && rt.built_hip_version > 0 && sg_runtime_info(nullptr) == SG_ERR_INVALID);
}

// batch calls from pageable memory: every load variant the launcher picks, padded outputs, staging growth and reuse
static void test_batch() {
  struct Shape { int n, hop, fmt, downmix, sstride, cols; float tau; int n_ch; };
  const Shape shapes[] = {{1024, 256, 0, 0, 1, 0, 0.f, 3},   {1024, 512, 1, 0, 1, 0, 0.f, 2},    {1024, 300, 0, 0, 1, 0, 0.f, 2},
                          {2048, 735, 2, 0, 1, 0, 0.f, 2},   {256, 64, 0, 0, 2, 0, 0.f, 4},       {512, 100, 1, 3, 1, 0, 0.f, 2},
                          {1024, 256, 0, 0, 1, 256, 0.f, 2}, {1024, 256, 0, 0, 1, 0, 0.7f, 2},    {2048, 512, 0, 0, 1, 128, 0.7f, 3},
                          {256, 64, 0, 0, 1, 0, 0.5f, 8200}, {256, 64, 0, 0, 1, 64, 0.5f, 8200}, {32768, 8192, 0, 0, 1, 300, 0.f, 1},
                          {4096, 4100, 0, 6, 1, 0, 0.f, 2}};
  for (const Shape& s : shapes) {
    for (int out_kind = 0; out_kind < 4; out_kind += 3) {
      sg_config c = cfg_of(s.n, s.hop);
      c.pcm_format = s.fmt;
      c.downmix = s.downmix;
      c.downmix_mode = s.downmix == 6 ? SG_DOWNMIX_SPEAKERS : SG_DOWNMIX_MEAN;
      c.logfreq_cols = s.cols;
      c.smoothing = s.tau;
      c.out_kind = out_kind;
      sg_handle* h = nullptr;
      CHECK(sg_create(&c, &h) == SG_OK);
      const int k = s.downmix > 1 ? s.downmix : 1;
      const size_t ssz = s.fmt == 1 ? 2 : s.fmt == 2 ? 3 : 4;
      for (int frames : {3, 0, 17, 1}) {                         // grows, shrinks, grows: the staging buffers are reused
        const int64_t ns = frames == 0 ? s.n - 1 : s.n + (int64_t)(frames - 1) * s.hop + 5;
        const int64_t cs = ns * k * s.sstride + 7;               // padded rows
        std::vector<unsigned char> in((size_t)(s.n_ch * cs) * ssz, 1);
        const int bins = sg_num_bins(h), esz = sg_out_elem_size(h);
        const int64_t nf = sg_num_frames(h, ns), ocs = nf * bins + 11;   // padded output
        CHECK(nf == frames);
        std::vector<unsigned char> outv((size_t)(s.n_ch * ocs + 1) * esz, 0xEE);
        const long seq0 = sg_stub_launch_seq();
        CHECK(sg_stft(h, (const float*)in.data(), s.n_ch, ns, cs, s.sstride, outv.data(), ocs, SG_PTR_HOST, nullptr) == SG_OK);
        CHECK(sg_stub_violations() == 0 && sg_stub_errors() == 0);
        if (frames == 0) {
          CHECK(sg_stub_launch_seq() == seq0);                   // ragged / empty input: nothing launched, nothing written
          for (unsigned char b : outv) CHECK(b == 0xEE);
          continue;
        }
        CHECK(sg_stub_launch_seq() == seq0 + 1);
        const unsigned char want = s.tau > 0.f && s.n_ch < 1000 ? 0x5A : (unsigned char)((seq0 + 1) & 0xff);   // two-pass smoothing: the scan's rows
        for (int ch = 0; ch < s.n_ch; ch += (s.n_ch > 64 ? 997 : 1)) {
          const unsigned char* row = outv.data() + (size_t)(ch * ocs) * esz;
          for (int64_t i = 0; i < nf * bins * esz; ++i) CHECK(row[i] == want);
          for (int64_t i = nf * bins * esz; i < ocs * esz; ++i) CHECK(row[i] == 0xEE);   // the gaps of a padded layout stay untouched
        }
        sg_launch_info li;
        CHECK(sg_last_launch(h, &li) == SG_OK && li.run_len >= 1 && li.total_runs >= s.n_ch);
      }
      // argument checks of the call itself
      float x[8] = {0};
      CHECK(sg_stft(h, x, -1, 10, 10, 1, x, 0, 0, nullptr) == SG_ERR_INVALID);
      CHECK(sg_stft(h, x, 1, 10, 10, 0, x, 0, 0, nullptr) == SG_ERR_INVALID);
      CHECK(sg_stft(h, nullptr, 1, 4 * s.n * k, 4 * s.n * k, 1, x, 0, 0, nullptr) == SG_ERR_INVALID);
      CHECK(sg_stft(h, x, 1, 4 * s.n * k, 4 * s.n * k, 1, x, 0, SG_IN_PINNED, nullptr) == SG_ERR_INVALID);   // not page-locked
      CHECK(sg_stft(nullptr, x, 1, 1, 1, 1, x, 0, 0, nullptr) == SG_ERR_INVALID);
      CHECK(sg_destroy(h) == SG_OK);
      CHECK(sg_stub_live_device_blocks() == 0);
    }
  }
}

// calls that only enqueue, issued on two "streams": the guards of the staging buffer and of the two-pass smoothing scratch
static void test_async_guards() {
  sg_config c = cfg_of(1024, 300);
  c.smoothing = 0.5f;
  c.out_kind = SG_OUT_MAG;
  sg_handle* h = nullptr;
  CHECK(sg_create(&c, &h) == SG_OK);
  const int n_ch = 3;
  const int64_t ns = 1024 + 40 * 300, frames = sg_num_frames(h, ns);
  void *pin = nullptr, *pout = nullptr;
  CHECK(sg_alloc_pinned((size_t)(n_ch * ns * 2) * 4, &pin) == SG_OK && sg_alloc_pinned((size_t)(n_ch * frames * 512) * 4, &pout) == SG_OK);
  std::memset(pin, 0, (size_t)(n_ch * ns * 2) * 4);
  for (int round = 0; round < 4; ++round) {
    void* stream = (void*)(uintptr_t)(1 + (round & 1));        // alternating streams: each call waits for the other's event
    // interleaved rows (sample_stride 2): page-locked input is staged, the call returns without waiting
    CHECK(sg_stft(h, (const float*)pin, n_ch, ns, 2 * ns, 2, pout, 0, SG_IN_PINNED | SG_OUT_PINNED, stream) == SG_OK);
    CHECK(sg_stub_errors() == 0 && sg_stub_violations() == 0);
  }
  CHECK(sg_synchronize(h, (void*)(uintptr_t)1) == SG_OK);
  // a larger call while the earlier one "is in flight": growing the buffers first drains the stream that uses them
  const int64_t ns2 = 3 * ns;
  void* pin2 = nullptr;
  CHECK(sg_alloc_pinned((size_t)(n_ch * ns2 * 2) * 4, &pin2) == SG_OK);
  std::memset(pin2, 0, (size_t)(n_ch * ns2 * 2) * 4);
  std::vector<float> big((size_t)(n_ch * sg_num_frames(h, ns2) * 512));
  CHECK(sg_stft(h, (const float*)pin2, n_ch, ns2, 2 * ns2, 2, big.data(), 0, SG_IN_PINNED, (void*)(uintptr_t)2) == SG_OK);
  CHECK(sg_stub_errors() == 0);
  CHECK(sg_destroy(h) == SG_OK);
  CHECK(sg_free_pinned(pin) == SG_OK && sg_free_pinned(pin2) == SG_OK && sg_free_pinned(pout) == SG_OK);
  CHECK(sg_stub_live_device_blocks() == 0 && sg_stub_live_host_blocks() == 0);
}

// the history ring under ragged pushes, every feed format, read back sample for sample; polls, the fused call, the byte ring
static void test_poll_mode(int n, int hist, int n_ch, int fmt, int k) {
  sg_config c = cfg_of(n, n);
  c.window = SG_WINDOW_BLACKMAN;
  c.scale = SG_SCALE_INV_N;
  c.pcm_format = fmt;
  c.downmix = k;
  c.history_samples = hist;
  sg_handle* h = nullptr;
  CHECK(sg_create(&c, &h) == SG_OK);
  uint8_t by[8];
  CHECK(sg_poll_u8(h, by, 8, 0, 0, nullptr) == SG_ERR_STATE && sg_push(h, (const float*)by, 1, 0, 0, nullptr) == SG_ERR_STATE);
  CHECK(sg_ring_begin(h, 4) == SG_ERR_STATE && sg_stream_begin(h, 0) == SG_ERR_INVALID);
  CHECK(sg_stream_begin(h, n_ch) == SG_OK);
  const int H = hist ? hist : n, kk = k > 1 ? k : 1, m = n / 2;
  const size_t ssz = fmt == 1 ? 2 : fmt == 2 ? 3 : 4;
  const float in_scale = fmt == 1 ? 1.f / 32768.f : fmt == 2 ? 1.f / 8388608.f : 1.f;
  std::vector<std::vector<float>> model(n_ch, std::vector<float>((size_t)n, 0.f));   // the latest N analysed samples per channel
  const int ring_h = 5;
  CHECK(sg_ring_begin(h, ring_h) == SG_OK && sg_ring_bytes(h) == (int64_t)n_ch * ring_h * m);
  std::vector<long> ring_rows;                              // sequence numbers of the polls that wrote ring rows, oldest first
  for (int step = 0; step < 60; ++step) {
    const int n_new = step % 7 == 0 ? rnd_in(H, 3 * H) : (step % 3 == 0 ? 2 * rnd_in(1, n / 2) : rnd_in(1, n + 3));
    const int64_t cs = (int64_t)n_new * kk + rnd_in(0, 5);
    std::vector<unsigned char> blk((size_t)(n_ch * cs) * ssz);
    std::vector<std::vector<float>> mono(n_ch, std::vector<float>((size_t)n_new));
    for (int ch = 0; ch < n_ch; ++ch)
      for (int t = 0; t < n_new; ++t) {
        float a = 0.f;
        for (int j = 0; j < kk; ++j) {
          const int64_t idx = ch * cs + (int64_t)t * kk + j;
          float v;
          if (fmt == 0) { v = (float)((int)(rnd() % 2001) - 1000) / 1000.f; ((float*)blk.data())[idx] = v; }
          else if (fmt == 1) { const short q = (short)(rnd() & 0xffff); ((short*)blk.data())[idx] = q; v = (float)q; }
          else { const int q = (int)(rnd() & 0xffffff) - (1 << 23); unsigned char* b = blk.data() + 3 * idx; b[0] = q & 0xff; b[1] = (q >> 8) & 0xff; b[2] = (q >> 16) & 0xff; v = (float)q; }
          a += v;
        }
        mono[ch][t] = kk > 1 ? a * (1.0f / (float)kk) : a;
      }
    // feed: plain push, or the one-call form (pageable: staged through the handle's page-locked buffers when small)
    const bool one_call = step % 2 == 1;
    std::vector<uint8_t> bytes((size_t)n_ch * (m + 3), 0xEE);
    const int len = step % 5 == 0 ? m / 2 : m;               // a caller's array may be shorter than the row
    long wrote_seq = -1;
    if (one_call) {
      CHECK(sg_push_poll_u8(h, (const float*)blk.data(), n_new, cs, bytes.data(), len, m + 3, SG_PTR_HOST, nullptr) == SG_OK);
      wrote_seq = sg_stub_launch_seq();
    } else {
      CHECK(sg_push(h, (const float*)blk.data(), n_new, cs, SG_PTR_HOST, nullptr) == SG_OK);
    }
    CHECK(sg_stub_errors() == 0 && sg_stub_violations() == 0);
    for (int ch = 0; ch < n_ch; ++ch) {
      std::vector<float>& md = model[ch];
      if (n_new >= n) md.assign(mono[ch].end() - n, mono[ch].end());
      else {
        md.erase(md.begin(), md.begin() + n_new);
        md.insert(md.end(), mono[ch].begin(), mono[ch].end());
      }
    }
    if (one_call) {
      for (int ch = 0; ch < n_ch; ++ch) {
        for (int i = 0; i < len; ++i) CHECK(bytes[(size_t)ch * (m + 3) + i] == (uint8_t)(wrote_seq & 0xff));
        for (int i = len; i < m + 3; ++i) CHECK(bytes[(size_t)ch * (m + 3) + i] == 0xEE);   // excess caller bytes untouched
      }
      ring_rows.push_back(wrote_seq);
    }
    // getFloatTimeDomainData: the latest N samples, exactly, whatever the ring position
    std::vector<float> td((size_t)n_ch * n, -7.f);
    CHECK(sg_poll_time_f32(h, td.data(), n, 0, SG_PTR_HOST, nullptr) == SG_OK);
    for (int ch = 0; ch < n_ch; ++ch)
      for (int i = 0; i < n; ++i) CHECK(td[(size_t)ch * n + i] == in_scale * model[ch][i]);
    if (step % 4 == 2) {                                       // a plain poll: new transform if dirty, else a re-emit of the state
      const long before = sg_stub_launch_seq();
      CHECK(sg_poll_u8(h, bytes.data(), m, m + 3, SG_PTR_HOST, nullptr) == SG_OK);
      const long s1 = sg_stub_launch_seq();
      CHECK(s1 == before + (one_call ? 0 : 1));                // nothing pushed since the fused call's poll: no second transform
      ring_rows.push_back(one_call ? wrote_seq : s1);
      CHECK(bytes[0] == (uint8_t)((one_call ? wrote_seq : s1) & 0xff));
      std::vector<uint8_t> tb((size_t)n_ch * 16);
      CHECK(sg_poll_time_u8(h, tb.data(), 16, 0, SG_PTR_HOST, nullptr) == SG_OK);
    }
    // the byte ring, unrolled oldest row first: the last ring_h polls in order
    std::vector<uint8_t> img((size_t)sg_ring_bytes(h));
    CHECK(sg_ring_read(h, img.data(), 0, SG_PTR_HOST, nullptr) == SG_OK);
    const int have = (int)ring_rows.size();
    for (int r = 0; r < ring_h; ++r) {
      const int idx = have - ring_h + r;
      const uint8_t want = idx < 0 ? 0 : (uint8_t)(ring_rows[idx] & 0xff);
      for (int ch = 0; ch < n_ch; ++ch) CHECK(img[((size_t)ch * ring_h + r) * m] == want && img[((size_t)ch * ring_h + r) * m + m - 1] == want);
    }
    CHECK(sg_ring_offset(h) == have % ring_h);
  }
  // argument checks around the polls
  CHECK(sg_poll_u8(h, nullptr, 4, 0, 0, nullptr) == SG_ERR_INVALID && sg_poll_u8(h, by, -1, 0, 0, nullptr) == SG_ERR_INVALID);
  CHECK(sg_poll_u8(h, by, 8, 4, 0, nullptr) == SG_ERR_INVALID);                      // stride < len
  CHECK(sg_poll_u8(h, by, 8, 0, SG_OUT_PINNED, nullptr) == SG_ERR_INVALID);          // not page-locked
  CHECK(sg_push(h, nullptr, 4, 0, 0, nullptr) == SG_ERR_INVALID && sg_push(h, (const float*)by, -1, 0, 0, nullptr) == SG_ERR_INVALID);
  CHECK(sg_push(h, (const float*)by, 8, 4, 0, nullptr) == SG_ERR_INVALID);           // channel_stride < n_new
  CHECK(sg_ring_begin(h, 0) == SG_ERR_INVALID && sg_ring_set_pinned(h, 1) == SG_OK && sg_ring_offset(h) == 0);
  // a refused fused call leaves the history where it was
  std::vector<float> before((size_t)n_ch * n), after((size_t)n_ch * n);
  CHECK(sg_poll_time_f32(h, before.data(), n, 0, 0, nullptr) == SG_OK);
  std::vector<unsigned char> blk((size_t)n_ch * 64 * kk * ssz, 0);
  CHECK(sg_push_poll_u8(h, (const float*)blk.data(), 64, 0, nullptr, m, 0, 0, nullptr) != SG_OK || fmt == 2 || kk > 1);
  if (!(fmt == 2 || kk > 1)) {
    CHECK(sg_poll_time_f32(h, after.data(), n, 0, 0, nullptr) == SG_OK && before == after);
  }
  CHECK(sg_destroy(h) == SG_OK);
  CHECK(sg_stub_live_device_blocks() == 0 && sg_stub_live_host_blocks() == 0 && sg_stub_errors() == 0);
}

// analyser.fftSize = n on a live handle: the samples stay, and a failed allocation leaves the handle exactly as it was
static void test_set_fft_size() {
  sg_config c = cfg_of(1024, 1024);
  c.history_samples = 32768;
  sg_handle* h = nullptr;
  CHECK(sg_create(&c, &h) == SG_OK && sg_stream_begin(h, 2) == SG_OK && sg_ring_begin(h, 3) == SG_OK);
  std::vector<float> x(2 * 40000);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (float)(i % 977) / 977.f;
  CHECK(sg_push(h, x.data(), 40000, 0, 0, nullptr) == SG_OK);
  auto tail_ok = [&](int n) {
    std::vector<float> td((size_t)2 * n);
    CHECK(sg_poll_time_f32(h, td.data(), n, 0, 0, nullptr) == SG_OK);
    // history keeps 32768 of the 40000: the newest n of channel c are x[c * 40000 + 40000 - n ...]
    for (int ch = 0; ch < 2; ++ch)
      for (int i = 0; i < n; ++i) CHECK(td[(size_t)ch * n + i] == x[(size_t)ch * 40000 + 40000 - n + i]);
  };
  tail_ok(1024);
  for (int n : {4096, 32768, 64, 2048}) {
    CHECK(sg_set_fft_size(h, n) == SG_OK && sg_num_bins(h) == n / 2 && sg_ring_bytes(h) == (int64_t)2 * 3 * (n / 2));
    tail_ok(n);
    std::vector<uint8_t> by((size_t)2 * (n / 2));
    CHECK(sg_poll_u8(h, by.data(), n / 2, 0, 0, nullptr) == SG_OK && sg_stub_violations() == 0);
  }
  CHECK(sg_set_fft_size(h, 1000) == SG_ERR_INVALID && sg_set_fft_size(h, 65536) == SG_ERR_INVALID && sg_num_bins(h) == 1024);
  // every allocation of the swap fails once: the handle keeps fft_size 2048, its tables, state and ring, and leaks nothing
  const long blocks = sg_stub_live_device_blocks();
  for (int nth = 0; nth < 8; ++nth) {
    sg_stub_fail_malloc_at(nth);
    const int rc = sg_set_fft_size(h, 8192);
    sg_stub_fail_malloc_at(-1);
    if (rc == SG_OK) {                                       // fewer than nth + 1 allocations in the swap: it went through
      CHECK(sg_num_bins(h) == 4096 && sg_set_fft_size(h, 2048) == SG_OK);
    } else {
      CHECK((rc == SG_ERR_NOMEM || rc == SG_ERR_HIP) && sg_num_bins(h) == 1024);
    }
    CHECK(sg_stub_live_device_blocks() == blocks);
    tail_ok(2048);
    std::vector<uint8_t> by(2 * 1024);
    CHECK(sg_poll_u8(h, by.data(), 1024, 0, 0, nullptr) == SG_OK && sg_stub_violations() == 0 && sg_stub_errors() == 0);
  }
  sg_config got;
  CHECK(sg_get_config(h, &got) == SG_OK && got.fft_size == 2048 && got.history_samples == 32768);
  // a batch handle (no stream) changes size too
  sg_handle* b = nullptr;
  sg_config cb = cfg_of(512, 128);
  CHECK(sg_create(&cb, &b) == SG_OK && sg_set_fft_size(b, 4096) == SG_OK && sg_num_frames(b, 4096) == 1 && sg_destroy(b) == SG_OK);
  CHECK(sg_destroy(h) == SG_OK && sg_destroy(nullptr) == SG_OK);
  CHECK(sg_stub_live_device_blocks() == 0);
}

static void test_debug_and_misc() {
  sg_config c = cfg_of(1024, 256);
  sg_handle* h = nullptr;
  CHECK(sg_create(&c, &h) == SG_OK);
  CHECK(sg_debug_set(h, "run_len", 8) == SG_OK && sg_debug_set(h, "run_len", -1) == SG_ERR_INVALID);
  CHECK(sg_debug_set(h, "no_such_key", 1) == SG_ERR_INVALID && sg_debug_set(h, nullptr, 1) == SG_ERR_INVALID);
  CHECK(sg_debug_set(h, "extra_lds", 1 << 20) == SG_ERR_INVALID && sg_debug_set(h, "extra_lds", 4096) == SG_OK);
  CHECK(sg_set_smoothing(h, 2.f) == SG_ERR_INVALID && sg_set_smoothing(h, 0.3f) == SG_OK);
  CHECK(sg_set_db_range(h, -30.f, -100.f) == SG_ERR_INVALID && sg_set_db_range(h, -90.f, -20.f) == SG_OK);
  std::vector<float> x(1024 + 20 * 256, 0.25f), out((size_t)21 * 512);
  CHECK(sg_stft(h, x.data(), 1, (int64_t)x.size(), (int64_t)x.size(), 1, out.data(), 0, 0, nullptr) == SG_OK);
  sg_launch_info li;
  CHECK(sg_last_launch(h, &li) == SG_OK && li.run_len == 8 && li.total_runs == 3);
  float trip[12];
  CHECK(sg_debug_twiddle_triplets(4096, trip, 12) == 6 * (8 + 64) && sg_debug_twiddle_triplets(1024, trip, 12) == 0);
  CHECK(sg_debug_twiddle_triplets(1000, trip, 12) == SG_ERR_INVALID);
  CHECK(sg_destroy(h) == SG_OK && sg_stub_live_device_blocks() == 0);
}

// Several devices in one process (spectrogram_amd/multidev.py; bench.py's ranks each see their own, but an application holding
// eight analysers does this): a handle lives on cfg.device_id whatever device the caller has current; every allocation, copy,
// event and launch of every entry happens on the handle's device (the stubs count anything else as misuse: memory belongs
// to the device it was allocated on, no peer access), and the caller's current device is what it was when the call returns
// -- on the error paths too.  No 8-GPU node is available to the builder: this is the CPU evidence for device_id > 0.
static void test_multi_device() {
  const int ndev = 8;
  sg_stub_set_device_count(ndev);
  CHECK(sg_stub_live_device_blocks() == 0);
  sg_config bad = cfg_of(1024, 256);
  bad.device_id = ndev;
  sg_handle* none = nullptr;
  CHECK(hipSetDevice(5) == 0 && sg_create(&bad, &none) == SG_ERR_INVALID && none == nullptr && sg_stub_current_device() == 5);
  sg_handle* hs[ndev];
  std::vector<float> x(2 * (2048 + 9 * 512));
  for (size_t i = 0; i < x.size(); ++i) x[i] = (float)(i % 613) / 613.f - 0.5f;
  for (int d = 0; d < ndev; ++d) {
    const int caller = (d + 3) % ndev;                 // never the handle's own device
    CHECK(hipSetDevice(caller) == 0);
    sg_config c = cfg_of(2048, 512);
    c.device_id = d;
    c.smoothing = d & 1 ? 0.6f : 0.f;                  // odd devices: the two-pass smoothing scratch and its event guard
    c.logfreq_cols = d == 2 ? 128 : 0;
    c.history_samples = 4096;                          // room for the live fftSize change below
    const long before = sg_stub_live_device_blocks();
    CHECK(sg_create(&c, &hs[d]) == SG_OK && sg_stub_current_device() == caller);
    CHECK(sg_stub_blocks_on_device(d) == sg_stub_live_device_blocks() - before && sg_stub_blocks_on_device(d) > 0);
  }
  // interleaved use: batch transforms from pageable memory (staging grows on the handle's device), poll mode, the byte ring,
  // a live fftSize change, the fused push + poll -- each from a caller sitting on some other device
  for (int round = 0; round < 3; ++round)
    for (int d = ndev - 1; d >= 0; --d) {
      const int caller = (d + 1 + round) % ndev == d ? (d + 2) % ndev : (d + 1 + round) % ndev;
      CHECK(hipSetDevice(caller) == 0);
      sg_handle* h = hs[d];
      const int row = sg_num_bins(h);
      const int64_t ns = (int64_t)x.size() / 2, fr = sg_num_frames(h, ns);
      std::vector<float> out((size_t)2 * fr * row);
      CHECK(sg_stft(h, x.data(), 2, ns, ns, 1, out.data(), 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      CHECK(sg_stft(h, x.data(), 2, ns - 1, ns, 1, out.data(), 0, 0, nullptr) == SG_OK);   // a second shape through the same staging
      CHECK(sg_stft(h, nullptr, 2, ns, ns, 1, out.data(), 0, 0, nullptr) == SG_ERR_INVALID && sg_stub_current_device() == caller);
      if (round == 0) CHECK(sg_stream_begin(h, 2) == SG_OK && sg_ring_begin(h, 4) == SG_OK && sg_stub_current_device() == caller);
      CHECK(sg_push(h, x.data(), 700 + 13 * d, 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      std::vector<uint8_t> by((size_t)2 * row);
      std::vector<float> fl((size_t)2 * row), td((size_t)2 * 2048);
      CHECK(sg_poll_u8(h, by.data(), row, 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      CHECK(sg_poll_f32(h, fl.data(), row, 0, 0, nullptr) == SG_OK);                        // no push in between: the re-emit kernel
      CHECK(sg_poll_time_f32(h, td.data(), 64, 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      CHECK(sg_push_poll_u8(h, x.data(), 736, 0, by.data(), row, 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      std::vector<uint8_t> img((size_t)sg_ring_bytes(h));
      CHECK(sg_ring_read(h, img.data(), 0, 0, nullptr) == SG_OK && sg_stub_current_device() == caller);
      if (round == 1) {
        CHECK(sg_set_fft_size(h, 4096) == SG_OK && sg_stub_current_device() == caller && sg_set_fft_size(h, 2048) == SG_OK);
      }
      CHECK(sg_stub_errors() == 0 && sg_stub_violations() == 0);
    }
  // everything a handle owns sits on its device and goes with it
  for (int d = 0; d < ndev; ++d) {
    CHECK(hipSetDevice((d + 4) % ndev) == 0);
    CHECK(sg_stub_blocks_on_device(d) > 0 && sg_destroy(hs[d]) == SG_OK && sg_stub_blocks_on_device(d) == 0);
    CHECK(sg_stub_current_device() == (d + 4) % ndev);
  }
  CHECK(sg_stub_live_device_blocks() == 0 && sg_stub_live_host_blocks() == 0 && sg_stub_errors() == 0 && sg_stub_violations() == 0);
  CHECK(hipSetDevice(0) == 0);
  sg_stub_set_device_count(1);
}

int main() {
  test_configs();
  test_batch();
  test_async_guards();
  test_poll_mode(256, 0, 1, 0, 0);
  test_poll_mode(256, 0, 3, 1, 0);
  test_poll_mode(512, 512 + 37, 2, 2, 0);
  test_poll_mode(128, 32768, 2, 0, 2);
  test_poll_mode(1024, 4096, 1, 2, 3);
  test_set_fft_size();
  test_debug_and_misc();
  test_multi_device();
  CHECK(sg_stub_errors() == 0 && sg_stub_violations() == 0);
  std::printf("shim driver ok: %ld checks, %ld stub launches, no stub errors, nothing left allocated\n", g_checks, sg_stub_launch_seq());
  return 0;
}
