// sk_rt.h -- the one place this library talks to the HIP runtime from.
//
// Every allocation, copy, fill, launch and wait of libstrelka_amd.so goes through `skrt::`.  In a caller process of its own
// (the default) each is the HIP call of the same name, inline.  With $STRELKA_AMD_BROKER=1 the process is a CLIENT of the per-GPU
// broker (csrc/sk_rt.hip, `sk_broker`): it never creates a GPU context; the calls are written as records into a ring in shared memory
// and executed, in order, on a stream of the client's own inside the ONE server process that holds the device's context.  Why: the
// kernel driver gives a device's compute work eight address-space slots, one per process; a ninth caller process is time-sliced however
// idle the device is (profiles/r05_v12) -- the reference's workflow runs one caller process per core
// (PY/strelkaSharedOptions.py:153-161), far more than eight.  Threads of one process do not share that limit.
//
// What makes the split thin:
//   * kernels are named by the offset of their host stub in libstrelka_amd.so (the server has loaded the same file) and their
//     arguments travel as the bytes hipLaunchKernel would be given: `launch` knows every parameter's type, nothing is registered by hand;
//   * page-locked host memory (`hostMalloc`) is a shared-memory segment mapped at the SAME virtual address in client and server and
//     page-locked there: pointers into it mean the same thing on both sides and in kernel arguments (the realignment job's and the
//     staged calls' mirrors are read and written by kernels directly);
//   * copies from / to the caller's pageable arrays go through a page-locked staging segment of the same kind (copied out at the next
//     wait, which is when an asynchronous copy to pageable memory is complete in HIP as well);
//   * device pointers are the server's; the client never dereferences them.
// Not carried: events (the timing entry points), `*_dev` entry points on a stream of the caller's (a client has exactly one stream).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>

namespace skrt
{
// true in a broker client (decided once, at the library's first use: $STRELKA_AMD_BROKER)
extern bool g_remote;
inline bool remote() { return g_remote; }
void set_remote(bool on);

// the remote halves (sk_rt.hip)
hipError_t r_malloc(void** p, size_t bytes);
hipError_t r_free(void* p);
hipError_t r_host_malloc(void** p, size_t bytes);
hipError_t r_host_free(void* p);
hipError_t r_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t r_memset_async(void* dst, int value, size_t bytes, hipStream_t st);
hipError_t r_stream_synchronize(hipStream_t st);
hipError_t r_get_last_error();
hipError_t r_func_set_attribute(const void* fn, hipFuncAttribute attr, int value);
void r_launch(const void* fn, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st, void** args, const uint32_t* sizes, int n_args);
hipStream_t r_stream(); // the client's one stream (a handle, not a HIP object)
const char* r_error_text(hipError_t e);
int r_connect(int device, std::string* why); // to the device's broker, starting it when nobody listens (sk_init)
void r_disconnect();
int r_device_count(std::string* why);
bool r_host_backend(); // the broker runs the no-GPU test backend
void r_wake_hint();
void r_kick();

inline hipError_t malloc_(void** p, const size_t bytes) { return g_remote ? r_malloc(p, bytes) : hipMalloc(p, bytes); }
template <typename T> inline hipError_t malloc_(T** p, const size_t bytes) { return malloc_(reinterpret_cast<void**>(p), bytes); }
inline hipError_t free_(void* p) { return g_remote ? r_free(p) : hipFree(p); }
inline hipError_t hostMalloc(void** p, const size_t bytes) { return g_remote ? r_host_malloc(p, bytes) : hipHostMalloc(p, bytes, hipHostMallocDefault); }
inline hipError_t hostFree(void* p) { return g_remote ? r_host_free(p) : hipHostFree(p); }
inline hipError_t memcpyAsync(void* dst, const void* src, const size_t bytes, const hipMemcpyKind kind, hipStream_t st)
{
    return g_remote ? r_memcpy_async(dst, src, bytes, kind, st) : hipMemcpyAsync(dst, src, bytes, kind, st);
}
inline hipError_t memsetAsync(void* dst, const int value, const size_t bytes, hipStream_t st)
{
    return g_remote ? r_memset_async(dst, value, bytes, st) : hipMemsetAsync(dst, value, bytes, st);
}
inline hipError_t streamSynchronize(hipStream_t st) { return g_remote ? r_stream_synchronize(st) : hipStreamSynchronize(st); }
// the synchronous forms: in a client, the same on its one stream followed by the wait
inline hipError_t memcpy_(void* dst, const void* src, const size_t bytes, const hipMemcpyKind kind)
{
    if (!g_remote) return hipMemcpy(dst, src, bytes, kind);
    const hipError_t e = r_memcpy_async(dst, src, bytes, kind, r_stream());
    return e != hipSuccess ? e : r_stream_synchronize(r_stream());
}
inline hipError_t memset_(void* dst, const int value, const size_t bytes)
{
    if (!g_remote) return hipMemset(dst, value, bytes);
    const hipError_t e = r_memset_async(dst, value, bytes, r_stream());
    return e != hipSuccess ? e : r_stream_synchronize(r_stream());
}
inline hipError_t getLastError() { return g_remote ? r_get_last_error() : hipGetLastError(); }
inline hipError_t setDevice(const int device) { return g_remote ? hipSuccess : hipSetDevice(device); } // (a client has no device of its own)
inline hipError_t funcSetAttribute(const void* fn, const hipFuncAttribute attr, const int value)
{
    return g_remote ? r_func_set_attribute(fn, attr, value) : hipFuncSetAttribute(fn, attr, value);
}
/// "work is coming": lets a broker client's server thread wake up beside the caller's packing instead of after it (nothing otherwise)
inline void wakeHint()
{
    if (g_remote) r_wake_hint();
}
/// "run what has been submitted": an entry point that returns to its caller with work in flight (sk_pileup_stream_push_begin) says so --
/// a broker client's server thread is otherwise woken when the client waits (sk_rt.hip, ring_commit); a process with a context of its own
/// has rung the device's doorbell with every launch already
inline void kick()
{
    if (g_remote) r_kick();
}
inline const char* errorString(const hipError_t e) { return (g_remote && r_error_text(e)[0]) ? r_error_text(e) : hipGetErrorString(e); }

namespace detail
{
template <typename Tuple, size_t... I> inline void launch_tuple(const void* fn, const dim3 grid, const dim3 block, const size_t lds, hipStream_t st, Tuple& vals, std::index_sequence<I...>)
{
    void* argv[sizeof...(I) ? sizeof...(I) : 1] = { static_cast<void*>(&std::get<I>(vals))... };
    if (!g_remote) {
        (void)hipLaunchKernel(fn, grid, block, argv, lds, st); // (an error stays with the runtime: skrt::getLastError())
        return;
    }
    const uint32_t sizes[sizeof...(I) ? sizeof...(I) : 1] = { uint32_t(sizeof(std::tuple_element_t<I, Tuple>))... };
    r_launch(fn, grid, block, lds, st, argv, sizes, int(sizeof...(I)));
}
}

/// kernel<<<grid, block, lds_bytes, st>>>(args...): the arguments are converted to the kernel's parameter types here, so that what
/// travels (to the runtime or to the broker) is exactly the parameter block
template <typename... KA, typename... A>
inline void launch(void (*kernel)(KA...), const dim3 grid, const dim3 block, const size_t lds_bytes, hipStream_t st, A&&... args)
{
    static_assert(sizeof...(KA) == sizeof...(A), "skrt::launch: argument count differs from the kernel's parameter count");
    std::tuple<std::remove_cv_t<KA>...> vals{ static_cast<std::remove_cv_t<KA>>(std::forward<A>(args))... };
    detail::launch_tuple(reinterpret_cast<const void*>(kernel), grid, block, lds_bytes, st, vals, std::index_sequence_for<KA...>{});
}
} // namespace skrt

#define SK_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) skrt::launch(kernel, dim3(grid), dim3(block), lds_bytes, stream, __VA_ARGS__)
