// Gradient exchange over RCCL (xGMI): the C-ABI side of SURVEY 8(b)'s `st_allreduce_f32`.
//
// One communicator per process (one process per GPU).  RCCL is resolved at run time with dlsym so that
// libspeecht_hip.so has no link-time dependency on it and -- more importantly -- binds to the RCCL
// instance that is already loaded in the process (PyTorch bundles one); a second copy would bring a
// second set of IPC handles and proxy threads.  The host side (speecht_amd/data_parallel.py) moves the
// 128-byte unique id from rank 0 to the other ranks over whatever bootstrap channel it has.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "st_common.h"

namespace {

struct RcclApi {
  ncclResult_t (*get_unique_id)(ncclUniqueId*);
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*comm_destroy)(ncclComm_t);
  ncclResult_t (*comm_count)(const ncclComm_t, int*);
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*group_start)();
  ncclResult_t (*group_end)();
  const char* (*error_string)(ncclResult_t);
  bool ready = false;
};

RcclApi g_rccl;

template <typename F>
bool bind(void* handle, const char* name, F& slot) {
  void* sym = handle ? dlsym(handle, name) : nullptr;
  if (!sym) sym = dlsym(RTLD_DEFAULT, name);
  slot = reinterpret_cast<F>(sym);
  return sym != nullptr;
}

int load_rccl() {
  if (g_rccl.ready) return ST_OK;
  // prefer an instance that is already mapped (RTLD_NOLOAD), fall back to the loader's search path
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  bool ok = bind(h, "ncclGetUniqueId", g_rccl.get_unique_id) & bind(h, "ncclCommInitRank", g_rccl.comm_init_rank) &
            bind(h, "ncclCommDestroy", g_rccl.comm_destroy) & bind(h, "ncclCommCount", g_rccl.comm_count) & bind(h, "ncclAllReduce", g_rccl.all_reduce) &
            bind(h, "ncclGroupStart", g_rccl.group_start) & bind(h, "ncclGroupEnd", g_rccl.group_end) &
            bind(h, "ncclGetErrorString", g_rccl.error_string);
  if (!ok) {
    st::set_error("RCCL is not available in this process (dlopen/dlsym librccl.so: %s)", dlerror());
    return ST_ECOMM;
  }
  g_rccl.ready = true;
  return ST_OK;
}

int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return ST_OK;
  st::set_error("%s: %s", what, g_rccl.error_string(r));
  return ST_ECOMM;
}

}  // namespace

extern "C" {

int st_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int st_comm_unique_id(void* id_out, size_t id_bytes) {
  ST_REQUIRE(id_out && id_bytes >= sizeof(ncclUniqueId), "st_comm_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
  if (int e = load_rccl()) return e;
  ncclUniqueId id;
  if (int e = rccl_check(g_rccl.get_unique_id(&id), "ncclGetUniqueId")) return e;
  memcpy(id_out, &id, sizeof(id));
  return ST_OK;
}

int st_comm_init(const void* id, size_t id_bytes, int rank, int world, void** comm_out) {
  ST_REQUIRE(id && id_bytes >= sizeof(ncclUniqueId) && comm_out, "st_comm_init: null/short argument");
  ST_REQUIRE(world >= 1 && rank >= 0 && rank < world, "st_comm_init: rank %d outside world %d", rank, world);
  if (int e = load_rccl()) return e;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  if (int e = rccl_check(g_rccl.comm_init_rank(&comm, world, uid, rank), "ncclCommInitRank")) return e;
  *comm_out = comm;
  return ST_OK;
}

int st_comm_destroy(void* comm) {
  if (!comm) return ST_OK;
  if (int e = load_rccl()) return e;
  return rccl_check(g_rccl.comm_destroy(reinterpret_cast<ncclComm_t>(comm)), "ncclCommDestroy");
}

int st_comm_count(void* comm, int* count) {
  ST_REQUIRE(comm && count, "st_comm_count: null argument");
  if (int e = load_rccl()) return e;
  return rccl_check(g_rccl.comm_count(reinterpret_cast<ncclComm_t>(comm), count), "ncclCommCount");
}

int st_allreduce_f32(void* comm, float* buf, size_t n, void* stream) {
  ST_REQUIRE(comm && (buf || n == 0), "st_allreduce_f32: null argument");
  if (n == 0) return ST_OK;
  if (int e = load_rccl()) return e;
  return rccl_check(g_rccl.all_reduce(buf, buf, n, ncclFloat32, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                      st::as_stream(stream)),
                    "ncclAllReduce");
}

int st_allreduce_buckets_f32(void* comm, float* base, const size_t* starts, const size_t* counts, int n_buckets,
                             void* stream) {
  ST_REQUIRE(comm && base && starts && counts && n_buckets >= 0, "st_allreduce_buckets_f32: null argument");
  if (n_buckets == 0) return ST_OK;
  if (int e = load_rccl()) return e;
  // one group => one launch for several slices of the flat gradient buffer
  if (int e = rccl_check(g_rccl.group_start(), "ncclGroupStart")) return e;
  int status = ST_OK;
  for (int i = 0; i < n_buckets && status == ST_OK; ++i)
    if (counts[i])
      status = rccl_check(g_rccl.all_reduce(base + starts[i], base + starts[i], counts[i], ncclFloat32, ncclSum,
                                            reinterpret_cast<ncclComm_t>(comm), st::as_stream(stream)),
                          "ncclAllReduce");
  int end = rccl_check(g_rccl.group_end(), "ncclGroupEnd");
  return status != ST_OK ? status : end;
}

}  // extern "C"
