// hip_stub.cpp -- TEST INFRASTRUCTURE ONLY (never part of the package): a host-memory stand-in for the slice of the HIP
// runtime that the host side of libspectro.so calls, so that csrc/spectro_api.hip -- argument checks, staging growth,
// history-ring positions, the transactional sg_set_fft_size, the byte ring -- can run under AddressSanitizer + UBSan on a
// machine without a GPU (`make -C oracle check-shim-asan`; GPU ASan is not available on the pool).  "Device" and
// page-locked memory are malloc blocks kept in a registry: every buffer the library touches has exact bounds for ASan,
// a free of something that was never allocated or a copy through a stale pointer is an error, and what is still
// allocated when a handle is gone is a leak the driver can see.  Kernels are replaced by tests/host/launch_stub.cpp.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace {
std::map<void*, size_t> g_dev, g_host;   // live "device" / page-locked blocks
std::map<void*, int> g_owner;             // the device that was current when a "device" block was allocated
long g_errors = 0;                        // misuse seen by the stub (the drivers assert 0)
int g_fail_malloc_at = -1;                // the n-th hipMalloc from now fails (transaction tests); -1: never
int g_device_count = 1;
int g_device = 0;
hipError_t g_last = hipSuccess;
struct StubEvent { int magic; long records; int device; };

hipError_t fail(hipError_t e, const char* what) {
  std::fprintf(stderr, "hip_stub: %s\n", what);
  ++g_errors;
  g_last = e;
  return e;
}
}  // namespace

extern "C" {

// ---- knobs and counters for the drivers
long sg_stub_errors(void) { return g_errors; }
long sg_stub_live_device_blocks(void) { return (long)g_dev.size(); }
long sg_stub_live_host_blocks(void) { return (long)g_host.size(); }
long sg_stub_live_device_bytes(void) {
  long n = 0;
  for (auto& kv : g_dev) n += (long)kv.second;
  return n;
}
void sg_stub_fail_malloc_at(int nth) { g_fail_malloc_at = nth; }
void sg_stub_set_device_count(int n) { g_device_count = n; if (g_device >= n) g_device = 0; }
int sg_stub_current_device(void) { return g_device; }
// multi-device model (one process driving several GPUs: spectrogram_amd/multidev.py, several N-API analysers): memory
// belongs to the device that was current at its hipMalloc, an event to the device it was created on.  No peer access is
// ever enabled by the library, so a kernel may touch only blocks of the device it is launched on (launch_stub.cpp asks
// through sg_stub_owner_of) and an event is recorded / waited for on its own device only.
int sg_stub_owner_of(const void* p) {   // owning device of the "device" block p lies in; -1: page-locked / unknown memory
  auto it = g_dev.upper_bound(const_cast<void*>(p));
  if (it == g_dev.begin()) return -1;
  --it;
  const char* b = (const char*)it->first;
  if ((const char*)p < b || (const char*)p > b + it->second) return -1;
  return g_owner[it->first];
}
long sg_stub_blocks_on_device(int d) {
  long n = 0;
  for (auto& kv : g_owner) n += kv.second == d;
  return n;
}
// the block a "device" pointer lies in: bytes from p to the block's end, or -1 (launch_stub checks kernel arguments with it)
long sg_stub_bytes_from(const void* p) {
  for (auto& m : {&g_dev, &g_host}) {
    auto it = m->upper_bound(const_cast<void*>(p));
    if (it == m->begin()) continue;
    --it;
    const char* b = (const char*)it->first;
    if ((const char*)p >= b && (const char*)p <= b + it->second) return (long)(b + it->second - (const char*)p);
  }
  return -1;
}

hipError_t hipGetDeviceCount(int* n) {
  *n = g_device_count;
  return g_device_count > 0 ? hipSuccess : (g_last = hipErrorNoDevice);
}
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= g_device_count) return g_last = hipErrorInvalidDevice;
  g_device = d;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev) {
  (void)dev;
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
  return hipSuccess;
}
hipError_t hipRuntimeGetVersion(int* v) { *v = 1; return hipSuccess; }
hipError_t hipDriverGetVersion(int* v) { *v = 1; return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorOutOfMemory: return "out of memory (stub)";
    case hipErrorInvalidValue: return "invalid value (stub)";
    case hipErrorNoDevice: return "no ROCm-capable device is detected (stub)";
    default: return "error (stub)";
  }
}

hipError_t hipMalloc(void** p, size_t n) {
  if (g_fail_malloc_at >= 0 && g_fail_malloc_at-- == 0) {
    *p = nullptr;
    return g_last = hipErrorOutOfMemory;
  }
  *p = std::malloc(n ? n : 1);
  if (!*p) return g_last = hipErrorOutOfMemory;
  std::memset(*p, 0xCD, n);   // "device" memory starts as garbage, never as zeros
  g_dev[*p] = n;
  g_owner[*p] = g_device;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  auto it = g_dev.find(p);
  if (it == g_dev.end()) return fail(hipErrorInvalidValue, "hipFree of a pointer hipMalloc never returned (or freed twice)");
  g_dev.erase(it);
  g_owner.erase(p);
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) {
  (void)flags;
  *p = std::malloc(n ? n : 1);
  if (!*p) return g_last = hipErrorOutOfMemory;
  g_host[*p] = n;
  return hipSuccess;
}
hipError_t hipHostFree(void* p) {
  if (!p) return hipSuccess;
  auto it = g_host.find(p);
  if (it == g_host.end()) return fail(hipErrorInvalidValue, "hipHostFree of a pointer hipHostMalloc never returned");
  g_host.erase(it);
  std::free(p);
  return hipSuccess;
}
// page-locked memory is mapped at its own address; a pageable pointer is refused like the runtime refuses it
hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned flags) {
  (void)flags;
  for (auto& kv : g_host)
    if ((char*)hp >= (char*)kv.first && (char*)hp < (char*)kv.first + kv.second) {
      *dp = hp;
      return hipSuccess;
    }
  *dp = nullptr;
  return g_last = hipErrorInvalidValue;
}
// (HIP would serve a copy that touches another device's memory through unified addressing; the library never means to --
// every entry switches to its handle's device first -- so the stub counts it as misuse)
static bool foreign(const void* p) { const int o = sg_stub_owner_of(p); return o >= 0 && o != g_device; }
hipError_t hipMemset(void* p, int v, size_t n) {
  if (foreign(p)) return fail(hipErrorInvalidValue, "hipMemset on memory of a device that is not current");
  std::memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) {
  (void)k;
  if (foreign(d) || foreign(s)) return fail(hipErrorInvalidValue, "hipMemcpy touching memory of a device that is not current");
  std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) {
  (void)k; (void)st;
  if (foreign(d) || foreign(s)) return fail(hipErrorInvalidValue, "hipMemcpyAsync touching memory of a device that is not current");
  std::memcpy(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st) {
  (void)k; (void)st;
  if (w > dp || w > sp) return fail(hipErrorInvalidValue, "hipMemcpy2DAsync: width exceeds a pitch");
  if (foreign(d) || foreign(s)) return fail(hipErrorInvalidValue, "hipMemcpy2DAsync touching memory of a device that is not current");
  for (size_t r = 0; r < h; ++r) std::memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags) {
  (void)flags;
  StubEvent* ev = (StubEvent*)std::malloc(sizeof(StubEvent));
  ev->magic = 0x5EED;
  ev->records = 0;
  ev->device = g_device;
  *e = (hipEvent_t)ev;
  return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
  StubEvent* ev = (StubEvent*)e;
  if (!ev || ev->magic != 0x5EED) return fail(hipErrorInvalidValue, "hipEventDestroy of something that is not a live event");
  ev->magic = 0;
  std::free(ev);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  (void)s;
  StubEvent* ev = (StubEvent*)e;
  if (!ev || ev->magic != 0x5EED) return fail(hipErrorInvalidValue, "hipEventRecord on something that is not a live event");
  if (ev->device != g_device) return fail(hipErrorInvalidHandle, "hipEventRecord: the event belongs to another device than the current one");
  ++ev->records;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags) {
  (void)s; (void)flags;
  StubEvent* ev = (StubEvent*)e;
  if (!ev || ev->magic != 0x5EED) return fail(hipErrorInvalidValue, "hipStreamWaitEvent on something that is not a live event");
  if (ev->records == 0) return fail(hipErrorInvalidValue, "hipStreamWaitEvent on an event that was never recorded");
  if (ev->device != g_device) return fail(hipErrorInvalidHandle, "hipStreamWaitEvent: the event belongs to another device than the current one");
  return hipSuccess;
}

}  // extern "C"
