// TEST INFRASTRUCTURE ONLY — a host stand-in for <hip/hip_runtime.h>, just large enough to compile the lane-per-env
// kernels of mettagrid_amd/csrc (construction, world update) for x86 and run them one work-item after another under
// AddressSanitizer / UndefinedBehaviorSanitizer (GPU sanitizers are not available on the MI355X pool).
// The observation kernel is wavefront-cooperative (DPP, ballot, LDS atomics) and is NOT emulated: a library built
// with this header produces no observations and no rewards, so it cannot stand in for the product.  See
// tests/cpu_emu/README.md.
#ifndef MGX_CPU_EMU_HIP_RUNTIME_H_
#define MGX_CPU_EMU_HIP_RUNTIME_H_
#ifndef MGX_CPU_EMU
#error "this header is only for the MGX_CPU_EMU sanitizer build"
#endif
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__
#define __constant__
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint4 { uint32_t x, y, z, w; };
struct int4 { int32_t x, y, z, w; };
struct char2 { signed char x, y; };
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline char2 make_char2(signed char a, signed char b) { return char2{a, b}; }

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef void* hipModule_t;      // run-time code objects are not part of the sanitizer build (mgx_attach_code refuses)
typedef void* hipFunction_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipStreamNonBlocking = 1 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipFuncAttributes { size_t localSizeBytes = 0; };
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipModuleUnload(hipModule_t) { return hipSuccess; }
static inline hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void**, void**) { return hipErrorInvalidValue; }
template <class T> static inline hipError_t hipMemcpyToSymbolAsync(T& sym, const void* s, size_t n, size_t, int, hipStream_t) { memcpy(&sym, s, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T& sym, const void* s, size_t n) { memcpy(&sym, s, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* d, T& sym, size_t n) { memcpy(d, &sym, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { a->localSizeBytes = 0; return hipSuccess; }
enum { hipEventDisableTiming = 2 };

// one work-item after another; kernels launched through this shim have no cross-lane communication
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                       \
  do {                                                                                                  \
    dim3 _g = (grid), _b = (block);                                                                     \
    gridDim = _g; blockDim = _b;                                                                        \
    for (unsigned _bz = 0; _bz < _g.z; _bz++) for (unsigned _by = 0; _by < _g.y; _by++)                  \
      for (unsigned _bx = 0; _bx < _g.x; _bx++) {                                                       \
        blockIdx = dim3(_bx, _by, _bz);                                                                 \
        for (unsigned _tx = 0; _tx < _b.x; _tx++) { threadIdx = dim3(_tx, 0, 0); kernel(__VA_ARGS__); } \
      }                                                                                                 \
  } while (0)

static inline void __syncthreads() {}
static inline void __threadfence() {}
static inline void __threadfence_system() {}
enum { hipHostMallocMapped = 2 };
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 1; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline double __dsqrt_rn(double a) { return std::sqrt(a); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __mul24(int a, int b) { return a * b; }
static inline unsigned long long clock64() { return 0; }
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; *p = o < v ? o : v; return o; }
#define MGX_EMU_MINMAX(T)                                   \
  static inline T min(T a, T b) { return b < a ? b : a; } \
  static inline T max(T a, T b) { return a < b ? b : a; }
MGX_EMU_MINMAX(int) MGX_EMU_MINMAX(unsigned) MGX_EMU_MINMAX(long) MGX_EMU_MINMAX(unsigned long) MGX_EMU_MINMAX(long long)
MGX_EMU_MINMAX(unsigned long long) MGX_EMU_MINMAX(float) MGX_EMU_MINMAX(double)
static inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
static inline unsigned max(unsigned a, int b) { return a < (unsigned)b ? (unsigned)b : a; }
// wavefront intrinsics: referenced by the (never launched) observation kernel only
static inline int __builtin_amdgcn_update_dpp(int, int x, int, int, int, bool) { return x; }
static inline int __builtin_amdgcn_readlane(int x, int) { return x; }
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
template <class T> static inline T __shfl(T v, int) { return v; }
#endif
