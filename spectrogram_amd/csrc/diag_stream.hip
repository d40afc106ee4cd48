// diag_stream.hip -- libsgdiag.so: what this box streams for the STFT path's traffic mix.
//
// Measurement aid for bench.py's roofline object only (not part of the C ABI in include/spectro.h,
// never on the product path): a grid-stride kernel in which every wave reads `rd` KiB and writes
// `wr` KiB per unit with 16-byte accesses, i.e. the best-shaped stream of the same read : write
// ratio as one STFT frame (hop * 4 bytes in, N/2 * 4 bytes out).  bench.py reports its rate as
// `measured_stream_GBps` beside the 8 TB/s spec peak.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

template <int RD_F4, int WR_F4>  // float4 per lane per unit
__global__ __launch_bounds__(256) void stream_mix(const f4* __restrict__ in, f4* __restrict__ out, long long units) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long u = w0; u < units; u += nw) {
    float acc = 0.f;
    const f4* s = in + u * (RD_F4 * 64);
#pragma unroll
    for (int i = 0; i < RD_F4; ++i) {
      const f4 v = __builtin_nontemporal_load(s + lane + 64 * i);
      acc += v.x + v.y + v.z + v.w;
    }
    f4* d = out + u * (WR_F4 * 64);
#pragma unroll
    for (int i = 0; i < WR_F4; ++i) __builtin_nontemporal_store(f4{acc, acc + 1.f, acc + 2.f, acc + 3.f}, d + lane + 64 * i);
  }
}

// the same stream with one (shader clock, 100 MHz clock) stamp pair per workgroup at its start and at its end
template <int RD_F4, int WR_F4>
__global__ __launch_bounds__(256) void stream_mix_stamped(const f4* __restrict__ in, f4* __restrict__ out, long long units,
                                                          unsigned long long* __restrict__ stamps) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long u = w0; u < units; u += nw) {
    float acc = 0.f;
    const f4* s = in + u * (RD_F4 * 64);
#pragma unroll
    for (int i = 0; i < RD_F4; ++i) {
      const f4 v = __builtin_nontemporal_load(s + lane + 64 * i);
      acc += v.x + v.y + v.z + v.w;
    }
    f4* d = out + u * (WR_F4 * 64);
#pragma unroll
    for (int i = 0; i < WR_F4; ++i) __builtin_nontemporal_store(f4{acc, acc + 1.f, acc + 2.f, acc + 3.f}, d + lane + 64 * i);
  }
  if (threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long* st = stamps + 4 * (long long)blockIdx.x;
    st[0] = r0;
    st[1] = __builtin_amdgcn_s_memrealtime();
    st[2] = c0;
    st[3] = __builtin_amdgcn_s_memtime();
  }
}

// One wave that does nothing but read the two clocks: `segments` times, the shader-clock ticks (s_memtime) that pass
// during `seg_ticks` ticks of the constant 100 MHz clock (s_memrealtime).  Sleeps between the reads: it takes one
// wave slot and next to no issue cycles from whatever else runs on its SIMD.
__global__ __launch_bounds__(64) void clock_probe(unsigned long long* __restrict__ out, int segments, unsigned long long seg_ticks,
                                                  unsigned long long* __restrict__ span) {
  if (threadIdx.x != 0) return;
  if (blockIdx.x == 0) span[0] = __builtin_amdgcn_s_memrealtime();
  out += 2LL * segments * blockIdx.x;
  for (int s = 0; s < segments; ++s) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r;
    do {
      __builtin_amdgcn_s_sleep(16);
      r = __builtin_amdgcn_s_memrealtime();
    } while (r - r0 < seg_ticks);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[2 * s] = c1 - c0;
    out[2 * s + 1] = r - r0;
  }
  if (blockIdx.x == 0) span[1] = __builtin_amdgcn_s_memrealtime();
  span[2 + blockIdx.x] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 15u;  // HW_REG_XCC_ID[3:0]: the XCD this wave ran on
}

__global__ void realtime_stamp(unsigned long long* __restrict__ out) { *out = __builtin_amdgcn_s_memrealtime(); }

// 16 bytes per lane from `src` to `dst`: one of the two is page-locked host memory mapped into the device's address space
__global__ __launch_bounds__(256) void link_copy(const f4* __restrict__ src, f4* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace

extern "C" {

// tools/clock_probe_check.py: the 100 MHz clock at this point of `stream`, into device memory
int sg_diag_realtime_stamp(void* stream, void* d_out) {
  hipLaunchKernelGGL(realtime_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)d_out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The shader clock the chip holds WHILE other work runs: `waves` one-wave workgroups on a stream of their own, started
// behind `wait_event` (a hipEvent_t recorded in the caller's stream of back-to-back launches; NULL: at once), each
// sampling `segments` stretches of `seg_us` microseconds.  Blocks until the probe has finished.  mhz[0..2] = median,
// min, max over all waves' segments; mhz[3..4] = the 100 MHz clock at wave 0's start and end; mhz[5..12] = the
// median on each of the 8 XCDs (HW_REG_XCC_ID of the wave; 0 where no wave landed).  The caller keeps its
// own launches queued for at least segments * seg_us behind the event.
int sg_diag_clock_probe(void* wait_event, int waves, int segments, int seg_us, double* mhz) {
  if (!mhz || waves < 1 || waves > 1024 || segments < 1 || segments > 4096 || seg_us < 1 || seg_us > 100000) return -1;
  unsigned long long* d = nullptr;
  hipStream_t side = nullptr;
  hipError_t e = hipMalloc((void**)&d, (size_t)waves * segments * 16 + 16 + (size_t)waves * 8);
  if (e != hipSuccess) return -(int)e;
  e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  if (e != hipSuccess) { (void)hipFree(d); return -(int)e; }
  if (wait_event) e = hipStreamWaitEvent(side, (hipEvent_t)wait_event, 0);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(clock_probe, dim3(waves), dim3(64), 0, side, d, segments, (unsigned long long)seg_us * 100ULL,
                       d + 2LL * waves * segments);
    e = hipGetLastError();
  }
  // copied back on the probe's own stream: a null-stream hipMemcpy would wait for the caller's queued launches first
  const size_t nseg = (size_t)waves * segments * 2;
  std::vector<unsigned long long> h(nseg + 2 + waves);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost, side);
  if (e == hipSuccess) e = hipStreamSynchronize(side);
  (void)hipStreamDestroy(side);
  (void)hipFree(d);
  if (e != hipSuccess) return -(int)e;
  std::vector<double> r;
  for (int i = 0; i < waves * segments; ++i)
    if (h[2 * i + 1] > 0) r.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  if (r.empty()) return -3;
  std::sort(r.begin(), r.end());
  mhz[0] = r[r.size() / 2];
  mhz[1] = r.front();
  mhz[2] = r.back();
  mhz[3] = (double)h[nseg];  // when wave 0 ran: the 100 MHz clock at its start and end
  mhz[4] = (double)h[nseg + 1];
  for (int x = 0; x < 8; ++x) {  // median per XCD (0: no wave landed there)
    std::vector<double> rx;
    for (int w = 0; w < waves; ++w)
      if ((int)h[nseg + 2 + w] == x)
        for (int i = 0; i < segments; ++i)
          if (h[2 * ((size_t)w * segments + i) + 1] > 0)
            rx.push_back((double)h[2 * ((size_t)w * segments + i)] / (double)h[2 * ((size_t)w * segments + i) + 1] * 100.0);
    std::sort(rx.begin(), rx.end());
    mhz[5 + x] = rx.empty() ? 0.0 : rx[rx.size() / 2];
  }
  return 0;
}

// In-kernel clock of the stream kernel: `seconds` of back-to-back launches over `total_bytes`, then one stamped
// launch; *mhz = median over its workgroups of (shader-clock ticks) / (100 MHz ticks) x 100, *gbps = the rate of the
// un-stamped launches (HIP events around the whole run).
int sg_diag_stream_clock(long long total_bytes, double seconds, double* mhz, double* gbps) {
  if (!mhz || !gbps || total_bytes < (3 << 20)) return -1;
  const long long units = total_bytes / 3072;
  const int blocks = 8192;
  f4 *in = nullptr, *out = nullptr;
  unsigned long long* st = nullptr;
  if (hipMalloc((void**)&in, (size_t)units * 1024) != hipSuccess) return -2;
  if (hipMalloc((void**)&out, (size_t)units * 2048) != hipSuccess) { (void)hipFree(in); return -2; }
  if (hipMalloc((void**)&st, (size_t)blocks * 32) != hipSuccess) { (void)hipFree(in); (void)hipFree(out); return -2; }
  (void)hipMemset(in, 0, (size_t)units * 1024);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  // one launch timed alone to size the run
  (void)hipEventRecord(a, 0);
  hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float t1 = 1.f;
  (void)hipEventElapsedTime(&t1, a, b);
  const int n = std::max(8, (int)(seconds * 1e3 / std::max(t1, 1e-3f)));
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
  (void)hipEventRecord(b, 0);
  hipLaunchKernelGGL((stream_mix_stamped<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units, st);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipError_t e = hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(in);
  (void)hipFree(out);
  (void)hipFree(st);
  if (e != hipSuccess) return -(int)e;
  std::vector<double> r;
  for (int i = 0; i < blocks; ++i) {
    const double dr = (double)(h[4 * i + 1] - h[4 * i]), dc = (double)(h[4 * i + 3] - h[4 * i + 2]);
    if (dr > 50.0) r.push_back(dc / dr * 100.0);  // workgroups that ran for more than 0.5 us
  }
  if (r.empty()) return -3;
  std::sort(r.begin(), r.end());
  *mhz = r[r.size() / 2];
  *gbps = (double)units * 3072.0 * n / (double)ms / 1e6;
  return 0;
}


// Streams `units` x (1 KiB read + 2 KiB written) per launch over caller-sized device buffers it allocates
// itself (in_bytes + out_bytes must exceed the 256 MiB Infinity Cache to mean anything); median of
// `reps` launches after 3 warm-ups.  Returns 0 and the rate in *gbps, or a negative hipError_t.
int sg_diag_stream_rate(long long total_bytes, int reps, double* gbps) {
  if (!gbps || total_bytes < (3 << 20) || reps < 1) return -1;
  const long long units = total_bytes / 3072;
  f4 *in = nullptr, *out = nullptr;
  hipError_t e = hipMalloc((void**)&in, (size_t)units * 1024);
  if (e != hipSuccess) return -(int)e;
  e = hipMalloc((void**)&out, (size_t)units * 2048);
  if (e != hipSuccess) { (void)hipFree(in); return -(int)e; }
  (void)hipMemset(in, 0, (size_t)units * 1024);
  (void)hipMemset(out, 0, (size_t)units * 2048);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int blocks = 8192;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    ms.push_back(t);
  }
  e = hipGetLastError();
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(in);
  (void)hipFree(out);
  if (e != hipSuccess) return -(int)e;
  std::sort(ms.begin(), ms.end());
  *gbps = (double)units * 3072.0 / (double)ms[ms.size() / 2] / 1e6;
  return 0;
}

// What the host link gives a KERNEL that reads (writes) `bytes` of page-locked host memory in place -- the way config 4's
// hop reaches the device (sg_push / sg_push_poll_* with SG_IN_PINNED read 2.1 MB per hop, polls with SG_OUT_PINNED write
// 1.05 MB): median of `reps` launches after 5 warm-ups, the host touching the buffer between launches as a capture
// thread would.  bench.py prices `streaming.roofline` (bound "pcie") with it at the hop's own transfer sizes.
int sg_diag_link_rate(long long bytes, int reps, double* read_gbps, double* write_gbps) {
  if (!read_gbps || !write_gbps || bytes < 4096 || bytes > (1LL << 30) || reps < 1) return -1;
  const long long n = bytes / 16;
  void* host = nullptr;
  f4* dev = nullptr;
  hipError_t e = hipHostMalloc(&host, (size_t)n * 16, hipHostMallocDefault);
  if (e != hipSuccess) return -(int)e;
  e = hipMalloc((void**)&dev, (size_t)n * 16);
  if (e != hipSuccess) { (void)hipHostFree(host); return -(int)e; }
  void* hd = nullptr;
  e = hipHostGetDevicePointer(&hd, host, 0);
  if (e != hipSuccess) { (void)hipFree(dev); (void)hipHostFree(host); return -(int)e; }
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const unsigned nb = (unsigned)((n + 255) / 256);
  double out[2] = {0.0, 0.0};
  for (int dir = 0; dir < 2; ++dir) {
    std::vector<float> ms;
    for (int r = 0; r < reps + 5; ++r) {
      for (long long i = 0; i < n * 4; i += 1024) ((float*)host)[i] = (float)(i + r);
      (void)hipEventRecord(a, 0);
      if (dir == 0) hipLaunchKernelGGL(link_copy, dim3(nb), dim3(256), 0, 0, (const f4*)hd, dev, n);
      else hipLaunchKernelGGL(link_copy, dim3(nb), dim3(256), 0, 0, (const f4*)dev, (f4*)hd, n);
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      float t = 0.f;
      (void)hipEventElapsedTime(&t, a, b);
      if (r >= 5) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    out[dir] = (double)n * 16.0 / (double)ms[ms.size() / 2] / 1e6;
  }
  e = hipGetLastError();
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(dev);
  (void)hipHostFree(host);
  if (e != hipSuccess) return -(int)e;
  *read_gbps = out[0];
  *write_gbps = out[1];
  return 0;
}

}  // extern "C"
