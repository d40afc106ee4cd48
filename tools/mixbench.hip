// mixbench.hip -- do VALU, LDS and VMEM instruction streams of DIFFERENT waves on one SIMD overlap?
// A 512- or 768-thread workgroup puts waves w, w+4 (, w+8) on the same SIMD.  Each group of 4 waves
// gets a role (packed-f32 VALU / LDS write+read / global stores / global loads); the kernel is timed
// with every role alone and with the roles together.  together ~= max(alone)  -> the streams overlap,
// together ~= sum(alone) -> one issue path (diagnostic for docs/EXPERIMENTS.md section 5.1; not part of the product).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/mixbench tools/mixbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

enum { R_NONE = 0, R_VALU = 1, R_LDS = 2, R_STORE = 3, R_LOAD = 4, R_VALU1 = 5, R_STORE_B = 6, R_STORE_X2 = 7, R_STORE_X4 = 8, R_STORE_NT = 9, R_STORE_X4U = 10 };

struct Roles { int r[3]; };

__global__ __launch_bounds__(768) void k(Roles roles, int iters, float* gbuf, size_t gstride, float* sink) {
  extern __shared__ f2 lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = roles.r[wave >> 2];
  if (role == R_VALU) {
    f2 a = {1.0f + lane * 1e-9f, 1.0f}, b = {1.0000001f, 0.9999999f}, c = {0.f, 1e-9f};
    f2 a1 = a + 1.f, a2 = a + 2.f, a3 = a + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    f2 r = a + a1 + a2 + a3;
    if (r.x == 12345.678f) sink[threadIdx.x] = r.x + r.y;
  } else if (role == R_VALU1) {
    float a = 1.0f + lane * 1e-9f, b = 1.0000001f, c = 1e-9f, a1 = a + 1.f, a2 = a + 2.f, a3 = a + 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    }
    if (a + a1 + a2 + a3 == 12345.678f) sink[threadIdx.x] = a;
  } else if (role == R_LDS) {
    f2* base = lds + wave * 640 + lane;   // 8 x 64 points + pad per wave
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f2{(float)lane, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) base[64 * i + (i >> 2)] = v[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = base[64 * (7 - i) + ((7 - i) >> 2)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (v[0].x == 12345.678f) sink[threadIdx.x] = v[3].y;
  } else if (role == R_STORE) {
    float* p = gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) p[64 * i + (it & 15) * 1024] = (float)i;  // wraps in 64 KB: L2 absorbs it
    }
  } else if (role == R_STORE_NT) {
    float* p = gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) __builtin_nontemporal_store((float)i, p + 64 * i + (it & 15) * 1024);
    }
  } else if (role == R_STORE_B) {
    unsigned char* p = reinterpret_cast<unsigned char*>(gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) p[64 * i + (it & 15) * 1024] = (unsigned char)i;
    }
  } else if (role == R_STORE_X2) {
    f2* p = reinterpret_cast<f2*>(gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride) + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) p[64 * i + (it & 7) * 1024] = f2{(float)i, 1.f};
    }
  } else if (role == R_STORE_X4 || role == R_STORE_X4U) {
    float* p0 = gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride + 4 * lane + (role == R_STORE_X4U ? 1 : 0);
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef f4 f4u __attribute__((aligned(4)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f4u*>(p0 + 256 * i + (it & 3) * 4096) = f4{(float)i, 1.f, 2.f, 3.f};
    }
  } else if (role == R_LOAD) {
    const f2* p = reinterpret_cast<const f2*>(gbuf + ((size_t)blockIdx.x * 12 + wave) * gstride) + lane;
    f2 acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      f2 t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = p[64 * i + (it & 15) * 256];  // wraps: L2 hits
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += t[i];
    }
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc.y;
  }
}

static float run(Roles r, int threads, int iters, float* gbuf, size_t gstride, float* sink) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int lds_bytes = 12 * 640 * 8;
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds_bytes, 0, r, iters, gbuf, gstride, sink);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), lds_bytes, 0, r, iters, gbuf, gstride, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  const int iters = 2000;
  const size_t gstride = 20 * 1024;  // floats per wave stream (the streams wrap: issue cost, not HBM)
  float *gbuf, *sink;
  CK(hipMalloc(&gbuf, gstride * 4 * 12 * 256));
  CK(hipMemset(gbuf, 0, gstride * 4 * 12 * 256));
  CK(hipMalloc(&sink, 4096));
  const char* nm[] = {"-", "VALUpk", "LDS", "STORE", "LOAD", "VALU1", "ST_b8", "ST_x2", "ST_x4", "ST_nt", "ST_x4u"};
  auto line = [&](int threads, int r0, int r1, int r2) {
    Roles r{{r0, r1, r2}};
    const float ms = run(r, threads, iters, gbuf, gstride, sink);
    printf("threads=%3d roles=[%-6s %-6s %-6s]  %.3f ms\n", threads, nm[r0], nm[r1], nm[r2], ms);
  };
  // per-iteration instruction counts: VALUpk 64 pk_fma; VALU1 128 fma; LDS 16 ds_write_b64 + 16 ds_read_b64;
  // STORE 16 global_store_dword (256 B each); LOAD 4 global_load_dwordx2 (512 B each)
  for (int role : {R_VALU, R_VALU1, R_LDS, R_STORE, R_LOAD}) line(512, role, R_NONE, R_NONE);
  for (int role : {R_STORE_B, R_STORE_X2, R_STORE_X4, R_STORE_X4U, R_STORE_NT}) line(512, role, R_NONE, R_NONE);
  for (int role : {R_STORE_B, R_STORE, R_STORE_X4}) line(512, role, role, R_NONE);
  for (int role : {R_STORE_B, R_STORE, R_STORE_X4}) line(768, role, role, role);
  line(768, R_VALU, R_VALU, R_STORE_X4);
  line(768, R_VALU, R_VALU, R_STORE);
  line(512, R_VALU, R_VALU, 0);
  line(512, R_VALU, R_LDS, 0);
  line(512, R_VALU1, R_LDS, 0);
  line(512, R_VALU, R_STORE, 0);
  line(512, R_VALU, R_LOAD, 0);
  line(512, R_LDS, R_LDS, 0);
  line(512, R_LDS, R_STORE, 0);
  line(512, R_STORE, R_STORE, 0);
  line(768, R_VALU, R_LDS, R_STORE);
  line(768, R_VALU, R_VALU, R_VALU);
  line(768, R_VALU, R_VALU, R_LDS);
  line(768, R_VALU, R_LDS, R_LDS);
  line(768, R_LDS, R_LDS, R_LDS);
  return 0;
}
