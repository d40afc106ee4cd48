// wav.cpp -- RIFF/WAVE header parse for the PCM ingest step in front of the hot path.
//
// The reference gets its samples from decodeAudioData / a MediaStream (util/util.js:2-28,
// UI/player.js:110-115); recorded audio arrives as WAV far more often than as raw floats, so the
// library tells a caller where the sample bytes are and which sg_pcm_format they have, and the
// conversion itself happens inside the framing load of the STFT kernel.  Host code only, no HIP:
// `make -C oracle check-host-asan` builds this file with -fsanitize=address,undefined and feeds it
// truncated / hostile headers.
#include "../../include/spectro.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>

int sg_set_error(int code, const char* fmt, ...);  // spectro_api.hip (thread-local message)

namespace {

uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

}  // namespace

extern "C" int sg_wav_parse(const void* bytes, size_t len, sg_wav_info* out) {
  if (!bytes || !out) return sg_set_error(SG_ERR_INVALID, "sg_wav_parse: NULL argument");
  std::memset(out, 0, sizeof *out);
  out->pcm_format = -1;
  const unsigned char* b = static_cast<const unsigned char*>(bytes);
  if (len < 12 || std::memcmp(b, "RIFF", 4) != 0 || std::memcmp(b + 8, "WAVE", 4) != 0)
    return sg_set_error(SG_ERR_INVALID, "not a RIFF/WAVE file");
  size_t pos = 12;
  bool have_fmt = false;
  uint16_t tag = 0;
  while (pos + 8 <= len) {
    const unsigned char* ck = b + pos;
    const uint32_t sz = rd32(ck + 4);
    const size_t body = pos + 8;
    if (std::memcmp(ck, "fmt ", 4) == 0) {
      if (sz < 16 || body + 16 > len) return sg_set_error(SG_ERR_INVALID, "truncated fmt chunk");
      tag = rd16(b + body);
      out->channels = rd16(b + body + 2);
      out->sample_rate = (int32_t)rd32(b + body + 4);
      out->block_align = rd16(b + body + 12);
      out->bits_per_sample = rd16(b + body + 14);
      if (tag == 0xFFFE) {  // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        if (sz < 40 || body + 26 > len) return sg_set_error(SG_ERR_INVALID, "truncated extensible fmt chunk");
        tag = rd16(b + body + 24);
      }
      have_fmt = true;
    } else if (std::memcmp(ck, "data", 4) == 0) {
      if (!have_fmt) return sg_set_error(SG_ERR_INVALID, "data chunk before fmt chunk");
      out->data_offset = (int64_t)body;
      size_t avail = len - body;
      // a streamed file may carry 0 or 0xFFFFFFFF here: take what is present
      out->data_bytes = (int64_t)((sz == 0 || sz == 0xFFFFFFFFu || (size_t)sz > avail) ? avail : (size_t)sz);
      break;
    }
    const size_t adv = (size_t)sz + (sz & 1u);  // chunks are word aligned
    if (adv > len - body) break;
    pos = body + adv;
  }
  if (!have_fmt) return sg_set_error(SG_ERR_INVALID, "no fmt chunk");
  if (out->data_offset == 0) return sg_set_error(SG_ERR_INVALID, "no data chunk");
  if (out->channels < 1 || out->sample_rate < 1) return sg_set_error(SG_ERR_INVALID, "bad channel count / sample rate");
  const int bytes_per_sample = out->bits_per_sample / 8;
  if (tag == 1 && out->bits_per_sample == 16) out->pcm_format = SG_PCM_S16;
  else if (tag == 1 && out->bits_per_sample == 24) out->pcm_format = SG_PCM_S24;
  else if (tag == 3 && out->bits_per_sample == 32) out->pcm_format = SG_PCM_F32;
  else
    return sg_set_error(SG_ERR_UNSUPPORTED, "WAV format tag %u with %d bits per sample: this build ingests PCM 16, PCM 24 and float 32",
                        (unsigned)tag, (int)out->bits_per_sample);
  if (out->block_align != out->channels * bytes_per_sample)
    return sg_set_error(SG_ERR_INVALID, "block_align %d != channels %d x %d bytes", (int)out->block_align, (int)out->channels, bytes_per_sample);
  out->n_frames = out->data_bytes / out->block_align;
  return SG_OK;
}
