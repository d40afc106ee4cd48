// axis.cpp -- frequency-axis helpers beside the log-frequency epilogue (host only, no HIP).
//
// Two mappings of the reference, restated:
//  * its legend scale: Util.setLogScale(x1, y1, x2, y2) / lin2log / log2lin (util/util.js:32-43), set up as
//    setLogScale(20, 20, 20000, 20000) (UI/spectrogram.js:77) and used by yToFreq / freqToY (:191-221);
//    pinned by tests/golden/ref_util_logscale.json, which is the reference's own code run under node;
//  * its shaders' column lookup x = 256^(u - 1) over the N/2-bin row (sonogram-vertex.shader:51-52), the
//    mapping sg_config.logfreq_cols bakes into the STFT epilogue: column (pixel centre, or mesh vertex as in the
//    reference's 3-D sonogram, 3D/visualizer.js:213-214) -> bin position -> Hz.
#include "../../include/spectro.h"

#include <cmath>

int sg_set_error(int code, const char* fmt, ...);  // spectro_api.hip (thread-local message)

extern "C" {

int sg_axis_log_scale(double x1, double y1, double x2, double y2, sg_log_scale* out) {
  if (!out || x1 == x2 || !(y1 > 0.0) || !(y2 > 0.0))
    return sg_set_error(SG_ERR_INVALID, "sg_axis_log_scale: needs an output, x1 != x2 and positive y1, y2");
  out->b = std::log(y1 / y2) / (x1 - x2);
  out->a = y1 / std::exp(out->b * x1);
  return SG_OK;
}

double sg_axis_lin2log(const sg_log_scale* s, double x) { return s->a * std::exp(s->b * x); }

double sg_axis_log2lin(const sg_log_scale* s, double y) { return std::log(y / s->a) / s->b; }

double sg_bin_hz(int32_t fft_size, double bin, double sample_rate) { return bin * sample_rate / (double)fft_size; }

double sg_logfreq_col_bin(int32_t fft_size, int32_t cols, double col, int32_t align) {
  // pixel centres (fragment shader, sonogram-fragment.shader:16) or the vertices of the 3-D sonogram's mesh,
  // u = x / (sonogram3DWidth - 1) (3D/visualizer.js:213-214)
  const double u = align == SG_LOGFREQ_VERTEX ? (cols > 1 ? col / (double)(cols - 1) : 0.0) : (col + 0.5) / (double)cols;
  return std::pow(256.0, u - 1.0) * (double)(fft_size / 2) - 0.5;  // texel centres sit at (i + 0.5) / (N/2)
}

double sg_logfreq_col_hz(int32_t fft_size, int32_t cols, double col, double sample_rate, int32_t align) {
  double b = sg_logfreq_col_bin(fft_size, cols, col, align);
  const double last = (double)(fft_size / 2 - 1);
  if (b < 0.0) b = 0.0;  // CLAMP_TO_EDGE (3D/visualizer.js:313)
  if (b > last) b = last;
  return sg_bin_hz(fft_size, b, sample_rate);
}

}  // extern "C"
