// Gradient exchange over RCCL behind the C ABI: nst_comm_{unique_id,init,allreduce_bucket,fence,broadcast,destroy}.
//
// What it replaces: the Horovod calls of the reference's data-parallel step -- hvd.DistributedOptimizer's averaged all-reduce of
// every gradient (neurst/training/hvd_utils.py:46-62) and the rank-0 broadcast of the initial variables
// (neurst/exps/trainer.py:285, hvd.callbacks.BroadcastGlobalVariablesCallback).  One process drives one GPU; a communicator
// is one RCCL rank plus a library-held communication stream and two events:
//
//   producer streams --(event)--> communication stream: ncclAllReduce(sum, in place) --(event)--> consumer stream
//
// so that neither the compute stream nor the weight-gradient stream ever waits for a bucket, and the host never blocks: every
// call is asynchronous.  The 1/N of hvd.Average is NOT applied here (the fused Adam kernel scales by it, nst_adam_update).
//
// RCCL is bound at run time (dlopen): a process that already carries librccl.so.1 (PyTorch-ROCm links its own copy) must not
// get a second one, and a single-GPU deployment needs none at all.  NST_RCCL_PATH overrides the library looked for.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>

#include <mutex>

#include "nst_common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* (*GetLastError)(ncclComm_t) = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool bind(void* h, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(h, name));
  return fn != nullptr;
}

// returns NULL and sets the error string when RCCL cannot be bound
const RcclApi* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return &g_rccl;
  void* h = nullptr;
  const char* override_path = getenv("NST_RCCL_PATH");
  if (override_path && override_path[0]) {
    h = dlopen(override_path, RTLD_NOW | RTLD_GLOBAL);
  } else {
    // the copy this process already carries, if any (same soname whoever shipped it)
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!h) {
    nst_set_error("nst_comm: librccl.so.1 not found (%s); set NST_RCCL_PATH", dlerror());
    return nullptr;
  }
  RcclApi a;
  a.handle = h;
  const bool ok = bind(h, "ncclGetUniqueId", a.GetUniqueId) && bind(h, "ncclCommInitRank", a.CommInitRank) &&
                  bind(h, "ncclCommDestroy", a.CommDestroy) && bind(h, "ncclAllReduce", a.AllReduce) &&
                  bind(h, "ncclBroadcast", a.Broadcast) && bind(h, "ncclGetErrorString", a.GetErrorString);
  bind(h, "ncclGetLastError", a.GetLastError);   // optional (newer RCCL)
  if (!ok) {
    nst_set_error("nst_comm: librccl.so.1 lacks an entry point (%s)", dlerror());
    return nullptr;
  }
  if (getenv("NST_COMM_DEBUG")) {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(a.AllReduce), &info) && info.dli_fname) fprintf(stderr, "[nst_comm] RCCL bound from %s\n", info.dli_fname);
  }
  g_rccl = a;
  return &g_rccl;
}

constexpr uint32_t COMM_MAGIC = 0x4e535443u;   // "NSTC"

struct Comm {
  uint32_t magic;
  int rank, world, device;
  ncclComm_t nccl;
  hipStream_t stream;        // the communication stream (library-held)
  hipEvent_t ready, done;    // producers -> communication stream, communication stream -> consumer
  int64_t buckets, bytes;    // issued since the last fence (introspection)
};

Comm* as_comm(void* p) {
  Comm* c = reinterpret_cast<Comm*>(p);
  return (c && c->magic == COMM_MAGIC) ? c : nullptr;
}

int nccl_fail(const RcclApi* api, const Comm* c, const char* what, ncclResult_t r) {
  const char* detail = (api->GetLastError && c) ? api->GetLastError(c->nccl) : "";
  nst_set_error("nst_comm: %s failed: %s %s", what, api->GetErrorString(r), detail ? detail : "");
  return NST_ERR_LAUNCH;
}

bool nccl_dtype(int dtype, ncclDataType_t* out) {
  switch (dtype) {
    case NST_F32: *out = ncclFloat32; return true;
    case NST_BF16: *out = ncclBfloat16; return true;
    case NST_COMM_F16: *out = ncclFloat16; return true;
    case NST_COMM_U8: *out = ncclUint8; return true;
    default: return false;
  }
}

}  // namespace

extern "C" int nst_comm_unique_id(void* id, size_t id_bytes) {
  NST_CHECK_ARG(id != nullptr, "nst_comm_unique_id: NULL buffer");
  NST_CHECK_ARG(id_bytes >= NST_COMM_UNIQUE_ID_BYTES, "nst_comm_unique_id: the buffer holds %zu bytes, %d needed", id_bytes,
                NST_COMM_UNIQUE_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == NST_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
  const RcclApi* api = rccl();
  if (!api) return NST_ERR_UNSUPPORTED;
  ncclUniqueId u;
  const ncclResult_t r = api->GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail(api, nullptr, "ncclGetUniqueId", r);
  memcpy(id, &u, sizeof(u));
  return NST_OK;
}

extern "C" int nst_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out) {
  NST_CHECK_ARG(comm_out != nullptr, "nst_comm_init: NULL output");
  *comm_out = nullptr;
  NST_CHECK_ARG(id != nullptr && id_bytes >= NST_COMM_UNIQUE_ID_BYTES, "nst_comm_init: the unique id needs %d bytes",
                NST_COMM_UNIQUE_ID_BYTES);
  NST_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "nst_comm_init: rank %d of %d", rank, world);
  const RcclApi* api = rccl();
  if (!api) return NST_ERR_UNSUPPORTED;
  Comm* c = new Comm();
  c->magic = COMM_MAGIC;
  c->rank = rank;
  c->world = world;
  c->buckets = c->bytes = 0;
  if (hipGetDevice(&c->device) != hipSuccess) {
    nst_set_error("nst_comm_init: no current HIP device: %s", hipGetErrorString(hipGetLastError()));
    delete c;
    return NST_ERR_LAUNCH;
  }
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  const ncclResult_t r = api->CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    const int rc = nccl_fail(api, nullptr, "ncclCommInitRank", r);
    delete c;
    return rc;
  }
  // the communication stream stays in the DEFAULT priority class: the host runs the step on a high-priority stream and its
  // weight-gradient stream on a low-priority one (nst_stream_create), so this stream -- whose waits for those two must never
  // sit in front of their kernels in a shared hardware queue -- cannot meet either of them in a queue
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
    nst_set_error("nst_comm_init: stream / event creation failed: %s", hipGetErrorString(hipGetLastError()));
    api->CommDestroy(c->nccl);
    delete c;
    return NST_ERR_LAUNCH;
  }
  *comm_out = c;
  return NST_OK;
}

extern "C" int nst_comm_info(void* comm, int* rank, int* world, int64_t* buckets_since_fence, int64_t* bytes_since_fence) {
  Comm* c = as_comm(comm);
  NST_CHECK_ARG(c != nullptr, "nst_comm_info: not a communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (buckets_since_fence) *buckets_since_fence = c->buckets;
  if (bytes_since_fence) *bytes_since_fence = c->bytes;
  return NST_OK;
}

// the communication stream waits for everything queued so far on the producers
static int wait_for_producers(Comm* c, void* const* producers, int nproducers) {
  for (int i = 0; i < nproducers; ++i) {
    NST_CHECK_HIP(hipEventRecord(c->ready, (hipStream_t)producers[i]));
    NST_CHECK_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
  }
  return NST_OK;
}

extern "C" int nst_comm_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, void* const* producers, int nproducers) {
  Comm* c = as_comm(comm);
  NST_CHECK_ARG(c != nullptr, "nst_comm_allreduce_bucket: not a communicator");
  NST_CHECK_ARG(count >= 0 && (buf != nullptr || count == 0), "nst_comm_allreduce_bucket: NULL bucket of %lld elements",
                (long long)count);
  NST_CHECK_ARG(nproducers >= 0 && nproducers <= 8 && (producers != nullptr || nproducers == 0),
                "nst_comm_allreduce_bucket: %d producer streams (0..8)", nproducers);
  ncclDataType_t dt;
  NST_CHECK_ARG(nccl_dtype(dtype, &dt), "nst_comm_allreduce_bucket: dtype %d", dtype);
  if (count == 0) return NST_OK;
  const RcclApi* api = rccl();
  if (!api) return NST_ERR_UNSUPPORTED;
  const int rc = wait_for_producers(c, producers, nproducers);
  if (rc != NST_OK) return rc;
  const ncclResult_t r = api->AllReduce(buf, buf, (size_t)count, dt, ncclSum, c->nccl, c->stream);
  if (r != ncclSuccess) return nccl_fail(api, c, "ncclAllReduce", r);
  c->buckets += 1;
  c->bytes += count * (dtype == NST_F32 ? 4 : dtype == NST_COMM_U8 ? 1 : 2);
  return NST_OK;
}

extern "C" int nst_comm_fence(void* comm, void* consumer) {
  Comm* c = as_comm(comm);
  NST_CHECK_ARG(c != nullptr, "nst_comm_fence: not a communicator");
  NST_CHECK_HIP(hipEventRecord(c->done, c->stream));
  NST_CHECK_HIP(hipStreamWaitEvent((hipStream_t)consumer, c->done, 0));
  c->buckets = c->bytes = 0;
  return NST_OK;
}

extern "C" int nst_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream) {
  Comm* c = as_comm(comm);
  NST_CHECK_ARG(c != nullptr, "nst_comm_broadcast: not a communicator");
  NST_CHECK_ARG(count >= 0 && (buf != nullptr || count == 0), "nst_comm_broadcast: NULL buffer of %lld elements", (long long)count);
  NST_CHECK_ARG(root >= 0 && root < c->world, "nst_comm_broadcast: root %d of %d ranks", root, c->world);
  ncclDataType_t dt;
  NST_CHECK_ARG(nccl_dtype(dtype, &dt), "nst_comm_broadcast: dtype %d", dtype);
  if (count == 0) return NST_OK;
  const RcclApi* api = rccl();
  if (!api) return NST_ERR_UNSUPPORTED;
  // in order with the caller's stream on both sides: its earlier writes are sent, its later reads see the root's values
  void* producers[1] = {stream};
  const int rc = wait_for_producers(c, producers, 1);
  if (rc != NST_OK) return rc;
  const ncclResult_t r = api->Broadcast(buf, buf, (size_t)count, dt, root, c->nccl, c->stream);
  if (r != ncclSuccess) return nccl_fail(api, c, "ncclBroadcast", r);
  NST_CHECK_HIP(hipEventRecord(c->done, c->stream));
  NST_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, c->done, 0));
  return NST_OK;
}

extern "C" int nst_comm_destroy(void* comm) {
  if (comm == nullptr) return NST_OK;
  Comm* c = as_comm(comm);
  NST_CHECK_ARG(c != nullptr, "nst_comm_destroy: not a communicator");
  const RcclApi* api = rccl();
  int rc = NST_OK;
  if (hipStreamSynchronize(c->stream) != hipSuccess) {
    nst_set_error("nst_comm_destroy: the communication stream failed: %s", hipGetErrorString(hipGetLastError()));
    rc = NST_ERR_LAUNCH;
  }
  if (api) {
    const ncclResult_t r = api->CommDestroy(c->nccl);
    if (r != ncclSuccess && rc == NST_OK) rc = nccl_fail(api, nullptr, "ncclCommDestroy", r);
  }
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->stream);
  c->magic = 0;
  delete c;
  return rc;
}
