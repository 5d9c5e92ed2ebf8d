// glb_comm.hip - the path's ONE collective behind the C ABI (include/glb.h: glb_comm_*, glb_allgather_f32), for integrators
// that bind the library without PyTorch: the all-gather of the shards' log-weights that README.md:108-110's normalisation
// needs once the particles are split over GPUs.  A thin layer over RCCL (ncclAllGather over xGMI); the
// Python host keeps using torch.distributed ("nccl" = the same RCCL, one initialisation per process) - this is the same
// call for a caller that has no torch.  RCCL is taken from the process at run time (dlopen: the copy PyTorch brought if
// there is one, else the system's), so the library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "../../include/glb.h"
#include "glb_common.hpp"

namespace {

struct Id {
  char bytes[GLB_COMM_ID_BYTES];
};
using GetUniqueId = int (*)(Id *);
using CommInitRank = int (*)(void **, int, Id, int);
using CommDestroy = int (*)(void *);
using AllGather = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
using GetErrorString = const char *(*)(int);

struct Rccl {
  GetUniqueId get_id = nullptr;
  CommInitRank init = nullptr;
  CommDestroy destroy = nullptr;
  AllGather all_gather = nullptr;
  GetErrorString err = nullptr;
  bool ok = false;
};

const Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1"})  // (already in the process - PyTorch's - if there is one)
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    r.get_id = (GetUniqueId)dlsym(h, "ncclGetUniqueId");
    r.init = (CommInitRank)dlsym(h, "ncclCommInitRank");
    r.destroy = (CommDestroy)dlsym(h, "ncclCommDestroy");
    r.all_gather = (AllGather)dlsym(h, "ncclAllGather");
    r.err = (GetErrorString)dlsym(h, "ncclGetErrorString");
    r.ok = r.get_id && r.init && r.destroy && r.all_gather;
  });
  return r;
}

int rccl_fail(const Rccl &r, int rc, const char *what) {
  return glb::api_fail(GLB_EHIP, "%s: RCCL error %d (%s)", what, rc, r.err ? r.err(rc) : "?");
}

constexpr int kNcclFloat = 7;  // ncclFloat32

}  // namespace

extern "C" {

int glb_comm_unique_id(void *out_id) {
  if (!out_id) return glb::api_fail(GLB_EINVAL, "null pointer");
  const Rccl &r = rccl();
  if (!r.ok) return glb::api_fail(GLB_EUNSUPPORTED, "RCCL (librccl.so) is not available in this process");
  const int rc = r.get_id((Id *)out_id);
  return rc ? rccl_fail(r, rc, "ncclGetUniqueId") : GLB_OK;
}

int glb_comm_init(const void *id, int32_t rank, int32_t world, void **out_comm) {
  if (!id || !out_comm || world <= 0 || rank < 0 || rank >= world) return glb::api_fail(GLB_EINVAL, "bad arguments");
  const Rccl &r = rccl();
  if (!r.ok) return glb::api_fail(GLB_EUNSUPPORTED, "RCCL (librccl.so) is not available in this process");
  const int rc = r.init(out_comm, world, *(const Id *)id, rank);
  return rc ? rccl_fail(r, rc, "ncclCommInitRank") : GLB_OK;
}

int glb_allgather_f32(void *comm, const float *send, int64_t n, float *recv, void *hip_stream) {
  if (!comm || !send || !recv || n < 0) return glb::api_fail(GLB_EINVAL, "bad arguments");
  const Rccl &r = rccl();
  if (!r.ok) return glb::api_fail(GLB_EUNSUPPORTED, "RCCL (librccl.so) is not available in this process");
  const int rc = r.all_gather(send, recv, (size_t)n, kNcclFloat, comm, (hipStream_t)hip_stream);
  return rc ? rccl_fail(r, rc, "ncclAllGather") : GLB_OK;
}

int glb_comm_destroy(void *comm) {
  if (!comm) return glb::api_fail(GLB_EINVAL, "null pointer");
  const Rccl &r = rccl();
  if (!r.ok) return glb::api_fail(GLB_EUNSUPPORTED, "RCCL (librccl.so) is not available in this process");
  const int rc = r.destroy(comm);
  return rc ? rccl_fail(r, rc, "ncclCommDestroy") : GLB_OK;
}

}  // extern "C"
