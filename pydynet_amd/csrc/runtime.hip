// Device runtime of libpdnhip.so: device selection, a caching HBM allocator, host<->device copies,
// streams, events, and the RCCL communicator used by data-parallel training.
//
// Replaces the runtime half of the reference's only device seam -- everything CuPy does for
// pydynet/cuda.py:16-32,89-99 besides arithmetic: cp.cuda.Device(id).use(), cupy's memory pool
// behind every xp.zeros / xp.array (core/tensor.py:80,90), `.get()` / `xp.asarray` transfers
// (tensor.py:385-403) -- and adds what the reference lacks entirely: a gradient all-reduce over
// xGMI.  Plain C ABI (include/pdn_hip.h): pointers and sizes only.
//
// Allocator design (MI355X: 288 GB of HBM3E per GPU, one process per GPU):
//   * every tape node of a training step allocates an output; hipMalloc/hipFree cost tens of
//     microseconds and hipFree synchronises the device, so blocks are cached per device in
//     exact-size-class free lists and handed out again without touching the driver;
//   * size classes: multiples of 512 B up to 64 KiB, then 8 classes per octave up to 64 MiB, then
//     multiples of 2 MiB -- a training loop requests the same sizes every step, so steady state is
//     100 % cache hits with <= 12.5 % internal slack on mid-size blocks and < 2 MiB on large ones;
//   * reuse is stream-ordered: the front end enqueues ALL compute on one stream per device, so a
//     block freed by the host (Python refcount) may be handed to the next request at once -- the
//     new user's kernels are queued behind the old user's.  Buffers touched by the communication
//     stream (the flat gradient buffer) live for the whole run and are never recycled mid-flight;
//   * on hipErrorOutOfMemory the cache is released to the driver and the request retried once.
#include <dlfcn.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

struct Block {
  size_t bytes;  // size class
  int device;
  int pool;      // 0 = the device's general cache, > 0 = a hipGraph's private pool
};

struct Pool {
  std::unordered_map<size_t, std::vector<void*>> free_lists;
  int64_t in_use = 0, reserved = 0, peak = 0, device_allocs = 0, requests = 0, hits = 0;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;   // every block obtained from hipMalloc (in use or cached)
std::map<int, Pool> g_pools;               // general cache per device
// Private pools: buffers a captured hipGraph refers to by address must never be handed to anyone else
// while the graph lives, so everything allocated while a pool is ACTIVE comes from -- and returns to --
// that pool's own free lists (frees inside the capture are reused inside it: the captured stream
// order keeps that safe), and the pool's blocks only rejoin the general cache when it is destroyed.
std::map<int, Pool> g_private;             // pool id -> pool
std::map<int, int> g_private_device;
int g_active_pool = 0, g_next_pool = 1;
std::map<int, hipStream_t> g_streams;      // the per-device compute stream

size_t size_class(size_t n) {
  if (n == 0) n = 1;
  if (n <= (64u << 10)) return (n + 511) & ~size_t(511);
  if (n >= (size_t(64) << 20)) return (n + ((size_t(2) << 20) - 1)) & ~((size_t(2) << 20) - 1);
  size_t p = size_t(1) << (63 - __builtin_clzll(n));   // largest power of two <= n
  size_t step = p >> 3;
  return (n + step - 1) / step * step;
}

int release_cached_locked(int device) {
  Pool& pool = g_pools[device];
  for (auto& kv : pool.free_lists) {
    for (void* p : kv.second) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) {
        pdn_set_error("hipFree: %s", hipGetErrorString(e));
        return (int)e;
      }
      pool.reserved -= (int64_t)kv.first;
      g_live.erase(p);
    }
    kv.second.clear();
  }
  return 0;
}

// ---- RCCL, bound lazily so the library loads (and the CPU build check passes) without it --------
typedef struct { char internal[128]; } rccl_unique_id;
typedef void* rccl_comm;
struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(rccl_unique_id*) = nullptr;
  int (*CommInitRank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
  int (*CommDestroy)(rccl_comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
} g_rccl;

// ncclDataType_t / ncclRedOp_t values of rccl.h (stable ABI since NCCL 2)
enum { RCCL_INT8 = 0, RCCL_FLOAT32 = 7 };
enum { RCCL_SUM = 0, RCCL_MAX = 2 };

int load_rccl() {
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    pdn_set_error("cannot load librccl.so: %s", dlerror());
    return PDN_EUNSUPPORTED;
  }
#define PDN_SYM(field, name)                                  \
  *(void**)(&g_rccl.field) = dlsym(h, name);                  \
  if (!g_rccl.field) {                                        \
    pdn_set_error("librccl.so lacks %s", name);               \
    return PDN_EUNSUPPORTED;                                  \
  }
  PDN_SYM(GetUniqueId, "ncclGetUniqueId")
  PDN_SYM(CommInitRank, "ncclCommInitRank")
  PDN_SYM(CommDestroy, "ncclCommDestroy")
  PDN_SYM(AllReduce, "ncclAllReduce")
  PDN_SYM(Broadcast, "ncclBroadcast")
  PDN_SYM(AllGather, "ncclAllGather")
  PDN_SYM(GroupStart, "ncclGroupStart")
  PDN_SYM(GroupEnd, "ncclGroupEnd")
  PDN_SYM(GetErrorString, "ncclGetErrorString")
#undef PDN_SYM
  g_rccl.handle = h;
  return 0;
}

#define PDN_RCCL(call)                                                             \
  do {                                                                             \
    int _r = (call);                                                               \
    if (_r != 0) {                                                                 \
      pdn_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call,                     \
                    g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl");   \
      return 1000 + _r;                                                            \
    }                                                                              \
  } while (0)

}  // namespace

extern "C" {

// ---- device ---------------------------------------------------------------------------------
int pdn_set_device(int device) {
  PDN_HIP(hipSetDevice(device));
  return 0;
}

int pdn_get_device(int* device) {
  PDN_CHECK_ARG(device != nullptr, "pdn_get_device: null output");
  PDN_HIP(hipGetDevice(device));
  return 0;
}

int pdn_device_synchronize(void) {
  PDN_HIP(hipDeviceSynchronize());
  return 0;
}

// ---- allocator --------------------------------------------------------------------------------
int pdn_malloc(void** ptr, int64_t bytes) {
  PDN_CHECK_ARG(ptr != nullptr && bytes >= 0, "pdn_malloc: bad arguments");
  int device = 0;
  PDN_HIP(hipGetDevice(&device));
  const size_t cls = size_class((size_t)bytes);
  std::lock_guard<std::mutex> lock(g_mu);
  const int pid = g_active_pool;
  Pool& pool = pid ? g_private[pid] : g_pools[device];
  pool.requests++;
  auto it = pool.free_lists.find(cls);
  void* p = nullptr;
  if (it != pool.free_lists.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
    pool.hits++;
  } else {
    hipError_t e = hipMalloc(&p, cls);
    if (e == hipErrorOutOfMemory) {
      (void)hipGetLastError();
      int rc = release_cached_locked(device);
      if (rc) return rc;
      e = hipMalloc(&p, cls);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      pdn_set_error("pdn_malloc(%lld bytes) on device %d: %s (in use %lld MiB, cached %lld MiB)",
                    (long long)bytes, device, hipGetErrorString(e), (long long)(pool.in_use >> 20),
                    (long long)((pool.reserved - pool.in_use) >> 20));
      return (int)e;
    }
    pool.reserved += (int64_t)cls;
    pool.device_allocs++;
    g_live[p] = Block{cls, device, pid};
  }
  pool.in_use += (int64_t)cls;
  if (pool.in_use > pool.peak) pool.peak = pool.in_use;
  *ptr = p;
  return 0;
}

int pdn_free(void* ptr) {
  if (!ptr) return 0;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_live.find(ptr);
  if (it == g_live.end()) {
    pdn_set_error("pdn_free: %p was not allocated by pdn_malloc", ptr);
    return PDN_EINVAL;
  }
  Pool& pool = it->second.pool ? g_private[it->second.pool] : g_pools[it->second.device];
  pool.free_lists[it->second.bytes].push_back(ptr);
  pool.in_use -= (int64_t)it->second.bytes;
  return 0;
}

/* ---- private pools + hipGraph capture / replay of a whole training step ------------------------- */
int pdn_pool_create(int* pool) {
  PDN_CHECK_ARG(pool != nullptr, "pdn_pool_create: null output");
  int device = 0;
  PDN_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_mu);
  *pool = g_next_pool++;
  g_private[*pool];
  g_private_device[*pool] = device;
  return 0;
}

/* pool > 0: allocations come from that pool until pdn_pool_activate(0) */
int pdn_pool_activate(int pool) {
  std::lock_guard<std::mutex> lock(g_mu);
  PDN_CHECK_ARG(pool == 0 || g_private.count(pool), "pdn_pool_activate: unknown pool %d", pool);
  g_active_pool = pool;
  return 0;
}

/* hands the pool's blocks back to the device's general cache (blocks still in use follow when freed) */
int pdn_pool_destroy(int pool) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_private.find(pool);
  PDN_CHECK_ARG(pool > 0 && it != g_private.end(), "pdn_pool_destroy: unknown pool %d", pool);
  if (g_active_pool == pool) g_active_pool = 0;
  Pool& gen = g_pools[g_private_device[pool]];
  for (auto& kv : it->second.free_lists)
    for (void* p : kv.second) gen.free_lists[kv.first].push_back(p);          // cached blocks
  gen.reserved += it->second.reserved;
  gen.in_use += it->second.in_use;                                              // blocks still held by arrays
  if (gen.in_use > gen.peak) gen.peak = gen.in_use;
  for (auto& kv : g_live)
    if (kv.second.pool == pool) kv.second.pool = 0;
  g_private.erase(it);
  g_private_device.erase(pool);
  return 0;
}

int pdn_pool_stats(int pool, int64_t* in_use, int64_t* reserved, int64_t* device_allocs) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_private.find(pool);
  PDN_CHECK_ARG(it != g_private.end(), "pdn_pool_stats: unknown pool %d", pool);
  if (in_use) *in_use = it->second.in_use;
  if (reserved) *reserved = it->second.reserved;
  if (device_allocs) *device_allocs = it->second.device_allocs;
  return 0;
}

/* capture everything enqueued on `stream` between begin and end into an executable graph */
int pdn_graph_begin_capture(void* stream) {
  PDN_CHECK_ARG(stream != nullptr, "pdn_graph_begin_capture: the null stream cannot be captured");
  PDN_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
  return 0;
}

int pdn_graph_end_capture(void* stream, void** graph_exec, int* n_nodes) {
  PDN_CHECK_ARG(stream && graph_exec, "pdn_graph_end_capture: null argument");
  hipGraph_t g = nullptr;
  PDN_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  if (n_nodes) {
    size_t n = 0;
    PDN_HIP(hipGraphGetNodes(g, nullptr, &n));
    *n_nodes = (int)n;
  }
  hipGraphExec_t e = nullptr;
  hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (rc != hipSuccess) {
    pdn_set_error("hipGraphInstantiate: %s", hipGetErrorString(rc));
    return (int)rc;
  }
  *graph_exec = (void*)e;
  return 0;
}

int pdn_graph_launch(void* graph_exec, void* stream) {
  PDN_CHECK_ARG(graph_exec != nullptr, "pdn_graph_launch: null graph");
  PDN_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return 0;
}

int pdn_graph_destroy(void* graph_exec) {
  if (graph_exec) PDN_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}

int pdn_empty_cache(void) {
  int device = 0;
  PDN_HIP(hipGetDevice(&device));
  PDN_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lock(g_mu);
  return release_cached_locked(device);
}

int pdn_mem_stats(int device, int64_t* in_use, int64_t* reserved, int64_t* peak_in_use,
                  int64_t* device_allocs, int64_t* requests, int64_t* cache_hits) {
  std::lock_guard<std::mutex> lock(g_mu);
  Pool& pool = g_pools[device];
  if (in_use) *in_use = pool.in_use;
  if (reserved) *reserved = pool.reserved;
  if (peak_in_use) *peak_in_use = pool.peak;
  if (device_allocs) *device_allocs = pool.device_allocs;
  if (requests) *requests = pool.requests;
  if (cache_hits) *cache_hits = pool.hits;
  return 0;
}

// ---- copies -----------------------------------------------------------------------------------
// Host buffers are pageable NumPy memory.  h2d returns once the source may be reused (the copy is
// ordered on `stream` and completed before returning); d2h returns with the data on the host.
int pdn_memcpy_h2d(void* dst, const void* src_host, int64_t bytes, void* stream) {
  if (bytes == 0) return 0;
  PDN_CHECK_ARG(dst && src_host && bytes > 0, "pdn_memcpy_h2d: bad arguments");
  PDN_HIP(hipMemcpyAsync(dst, src_host, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  PDN_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int pdn_memcpy_d2h(void* dst_host, const void* src, int64_t bytes, void* stream) {
  if (bytes == 0) return 0;
  PDN_CHECK_ARG(dst_host && src && bytes > 0, "pdn_memcpy_d2h: bad arguments");
  PDN_HIP(hipMemcpyAsync(dst_host, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  PDN_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// Pinned host memory + a device -> host copy that does NOT synchronise: the read-back of a small result (the token of a
// decode step) can be queued on its own stream behind an event while the compute stream already runs the next step;
// the host waits on the copy's event only (pdn_event_synchronize).
int pdn_host_alloc(void** out, int64_t bytes) {
  PDN_CHECK_ARG(out && bytes > 0, "pdn_host_alloc: bad arguments");
  PDN_HIP(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
  return 0;
}

// Coherent pinned host memory the GPU can write directly: *host_ptr for the CPU, *device_ptr for kernels.  A kernel
// that leaves a few bytes there (system-scope store) hands them to the host with no copy command and no event: the
// host polls the location (the token of a decode step, llm/llama.py).  Freed with pdn_host_free(host_ptr).
int pdn_host_alloc_mapped(void** host_ptr, void** device_ptr, int64_t bytes) {
  PDN_CHECK_ARG(host_ptr && device_ptr && bytes > 0, "pdn_host_alloc_mapped: bad arguments");
  PDN_HIP(hipHostMalloc(host_ptr, (size_t)bytes, hipHostMallocCoherent | hipHostMallocMapped));
  PDN_HIP(hipHostGetDevicePointer(device_ptr, *host_ptr, 0));
  return 0;
}

int pdn_host_free(void* ptr) {
  if (ptr) PDN_HIP(hipHostFree(ptr));
  return 0;
}

int pdn_memcpy_d2h_async(void* dst_pinned_host, const void* src, int64_t bytes, void* stream) {
  if (bytes == 0) return 0;
  PDN_CHECK_ARG(dst_pinned_host && src && bytes > 0, "pdn_memcpy_d2h_async: bad arguments");
  PDN_HIP(hipMemcpyAsync(dst_pinned_host, src, (size_t)bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}

int pdn_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream) {
  if (bytes == 0) return 0;
  PDN_CHECK_ARG(dst && src && bytes > 0, "pdn_memcpy_d2d: bad arguments");
  PDN_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int pdn_memset(void* dst, int byte_value, int64_t bytes, void* stream) {
  if (bytes == 0) return 0;
  PDN_CHECK_ARG(dst && bytes > 0, "pdn_memset: bad arguments");
  PDN_HIP(hipMemsetAsync(dst, byte_value, (size_t)bytes, (hipStream_t)stream));
  return 0;
}

// ---- streams and events -------------------------------------------------------------------------
// Streams are created non-blocking: they never synchronise implicitly with the legacy null stream,
// so the communication stream overlaps compute and a compute stream can be captured into a hipGraph.
int pdn_stream_create(void** stream, int high_priority) {
  PDN_CHECK_ARG(stream != nullptr, "pdn_stream_create: null output");
  hipStream_t s;
  if (high_priority) {
    int lo = 0, hi = 0;
    PDN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    PDN_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
  } else {
    PDN_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  }
  *stream = (void*)s;
  return 0;
}

int pdn_stream_destroy(void* stream) {
  if (stream) PDN_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

/* the compute stream of the current device (created on first use, lives for the process) */
int pdn_compute_stream(void** stream) {
  PDN_CHECK_ARG(stream != nullptr, "pdn_compute_stream: null output");
  int device = 0;
  PDN_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_streams.find(device);
  if (it == g_streams.end()) {
    hipStream_t s;
    PDN_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    it = g_streams.emplace(device, s).first;
  }
  *stream = (void*)it->second;
  return 0;
}

int pdn_stream_wait_event(void* stream, void* event) {
  PDN_CHECK_ARG(event != nullptr, "pdn_stream_wait_event: null event");
  PDN_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return 0;
}

int pdn_event_create(void** event, int timing) {
  PDN_CHECK_ARG(event != nullptr, "pdn_event_create: null output");
  hipEvent_t e;
  // Ordering events order streams of ONE device: a device-scope release is all they need.  (The
  // default system-scope release makes every later kernel of the step pay a cache write-back: a
  // forced single-rank data-parallel step ran 61.5 -> 69 ms with default events, measured.)
  PDN_HIP(hipEventCreateWithFlags(&e, timing ? hipEventDefault : (hipEventDisableTiming | hipEventReleaseToDevice)));
  *event = (void*)e;
  return 0;
}

int pdn_event_record(void* event, void* stream) {
  PDN_CHECK_ARG(event != nullptr, "pdn_event_record: null event");
  PDN_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return 0;
}

int pdn_event_synchronize(void* event) {
  PDN_CHECK_ARG(event != nullptr, "pdn_event_synchronize: null event");
  PDN_HIP(hipEventSynchronize((hipEvent_t)event));
  return 0;
}

int pdn_event_elapsed_ms(void* start, void* stop, float* ms) {
  PDN_CHECK_ARG(start && stop && ms, "pdn_event_elapsed_ms: null argument");
  PDN_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return 0;
}

int pdn_event_destroy(void* event) {
  if (event) PDN_HIP(hipEventDestroy((hipEvent_t)event));
  return 0;
}

// ---- RCCL communicator (one rank per process, one process per GPU) --------------------------------
int pdn_comm_unique_id(char* id128) {
  PDN_CHECK_ARG(id128 != nullptr, "pdn_comm_unique_id: null output");
  int rc = load_rccl();
  if (rc) return rc;
  rccl_unique_id id;
  PDN_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  return 0;
}

int pdn_comm_init(void** comm, int rank, int world, const char* id128) {
  PDN_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "pdn_comm_init: bad arguments");
  int rc = load_rccl();
  if (rc) return rc;
  rccl_unique_id id;
  memcpy(id.internal, id128, 128);
  rccl_comm c = nullptr;
  PDN_RCCL(g_rccl.CommInitRank(&c, world, id, rank));
  *comm = c;
  return 0;
}

int pdn_comm_destroy(void* comm) {
  if (!comm) return 0;
  PDN_CHECK_ARG(g_rccl.handle != nullptr, "pdn_comm_destroy: RCCL is not loaded");
  PDN_RCCL(g_rccl.CommDestroy((rccl_comm)comm));
  return 0;
}

/* in place; op 0 = sum, 1 = max */
int pdn_comm_allreduce_f32(void* comm, float* buf, int64_t n, int op, void* stream) {
  PDN_CHECK_ARG(comm && buf && n >= 0 && (op == 0 || op == 1), "pdn_comm_allreduce_f32: bad arguments");
  if (n == 0) return 0;
  PDN_RCCL(g_rccl.AllReduce(buf, buf, (size_t)n, RCCL_FLOAT32, op == 0 ? RCCL_SUM : RCCL_MAX,
                            (rccl_comm)comm, (hipStream_t)stream));
  return 0;
}

int pdn_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream) {
  PDN_CHECK_ARG(comm && buf && bytes >= 0, "pdn_comm_broadcast: bad arguments");
  if (bytes == 0) return 0;
  PDN_RCCL(g_rccl.Broadcast(buf, buf, (size_t)bytes, RCCL_INT8, root, (rccl_comm)comm, (hipStream_t)stream));
  return 0;
}

/* recv holds world * bytes_per_rank bytes, rank r's contribution at offset r * bytes_per_rank */
int pdn_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
  PDN_CHECK_ARG(comm && send && recv && bytes_per_rank >= 0, "pdn_comm_allgather: bad arguments");
  if (bytes_per_rank == 0) return 0;
  PDN_RCCL(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, RCCL_INT8, (rccl_comm)comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
