// wav_fuzz.cpp -- host-only sanitizer driver for spectrogram_amd/csrc/wav.cpp (no HIP, no GPU).
// Built by `make -C oracle check-host-asan` with -fsanitize=address,undefined: parses every fixture under
// tests/golden/wav, then every prefix of each fixture, then seeded single-byte and 32-bit corruptions of
// the header bytes, each from a heap buffer of exactly the stated length so any over-read trips ASan.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spectro.h"

static char g_msg[512];
int sg_set_error(int code, const char* fmt, ...) {  // the library's definition lives in spectro_api.hip
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_msg, sizeof g_msg, fmt, ap);
  va_end(ap);
  return code;
}

static std::vector<unsigned char> slurp(const std::string& path) {
  std::vector<unsigned char> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return v;
  unsigned char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

static int parse_exact(const unsigned char* p, size_t n, sg_wav_info* info) {
  unsigned char* heap = (unsigned char*)malloc(n ? n : 1);  // exact size: over-reads are caught
  memcpy(heap, p, n);
  const int rc = sg_wav_parse(heap, n, info);
  if (rc == SG_OK) {
    // whatever it reports must lie inside the image
    if (info->data_offset < 0 || info->data_bytes < 0 || (uint64_t)info->data_offset + (uint64_t)info->data_bytes > n ||
        info->block_align <= 0 || info->n_frames * info->block_align > info->data_bytes) {
      printf("inconsistent result: off=%lld bytes=%lld frames=%lld align=%d len=%zu\n", (long long)info->data_offset,
             (long long)info->data_bytes, (long long)info->n_frames, info->block_align, n);
      exit(1);
    }
  }
  free(heap);
  return rc;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/wav";
  const char* names[] = {"s16_mono", "s16_stereo_list", "s24_stereo", "s24_mono_ext", "f32_stereo", "s16_mono_streamed",
                         "u8_mono_unsupported"};
  long cases = 0, ok = 0;
  uint32_t rng = 12345;
  for (const char* nm : names) {
    const std::vector<unsigned char> img = slurp(dir + "/" + nm + ".wav");
    if (img.empty()) { printf("missing fixture %s\n", nm); return 1; }
    sg_wav_info info;
    const int rc = parse_exact(img.data(), img.size(), &info);
    const bool expect_ok = strstr(nm, "unsupported") == nullptr;
    if ((rc == SG_OK) != expect_ok) { printf("%s: rc=%d (%s)\n", nm, rc, g_msg); return 1; }
    ++cases;
    const size_t hdr = img.size() < 128 ? img.size() : 128;
    for (size_t n = 0; n <= hdr; ++n) { ok += parse_exact(img.data(), n, &info) == SG_OK; ++cases; }
    std::vector<unsigned char> m(img.begin(), img.begin() + (img.size() < 512 ? img.size() : 512));
    for (int it = 0; it < 4000; ++it) {
      std::vector<unsigned char> c = m;
      rng = rng * 1664525u + 1013904223u;
      const size_t pos = (rng >> 8) % hdr;
      rng = rng * 1664525u + 1013904223u;
      if (it & 1) c[pos] = (unsigned char)(rng >> 16);
      else if (pos + 4 <= c.size()) { const uint32_t v = (it & 2) ? 0xFFFFFFFFu : (rng >> 3); memcpy(&c[pos], &v, 4); }
      ok += parse_exact(c.data(), c.size(), &info) == SG_OK;
      ++cases;
    }
  }
  sg_wav_info info;
  if (sg_wav_parse(nullptr, 10, &info) == SG_OK) return 1;
  printf("wav_fuzz: %ld cases, %ld parsed, no sanitizer report\n", cases, ok);
  return 0;
}
