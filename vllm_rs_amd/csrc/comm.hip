// comm.hip — tensor-parallel all-reduce over RCCL (xGMI).  Restates the AllReduce CustomOp1 of
// src/models/layers/distributed.rs:325-396 (`comm.all_reduce(src, dst, Sum)`, bf16/f16 only) and
// the bootstrap of src/runner/runner.rs:80-89 (`Comm::from_rank(dev, rank, world, id)`) with the
// 128-byte unique id shipped in MessageType::Init (src/runner/mod.rs:25-27).
#include <rccl/rccl.h>
#include <string.h>

#include "common.cuh"

struct VraComm {
  ncclComm_t comm;
  int rank, world;
};
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId must be 128 bytes");

extern "C" int32_t vra_comm_unique_id(uint8_t h_id_out[128]) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) {
    vra_set_error("ncclGetUniqueId: %s", ncclGetErrorString(r));
    return -1;
  }
  memcpy(h_id_out, &id, 128);
  return 0;
}
extern "C" void* vra_comm_create(const uint8_t h_id[128], int32_t rank, int32_t world_size, int32_t device) {
  if (world_size == 1) {  // dummy Comm of distributed.rs:14-32
    VraComm* c = new VraComm{nullptr, 0, 1};
    return c;
  }
  if (hipSetDevice(device) != hipSuccess) {
    vra_set_error("vra_comm_create: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  ncclUniqueId id;
  memcpy(&id, h_id, 128);
  ncclComm_t comm;
  ncclResult_t r = ncclCommInitRank(&comm, world_size, id, rank);
  if (r != ncclSuccess) {
    vra_set_error("ncclCommInitRank: %s", ncclGetErrorString(r));
    return nullptr;
  }
  return new VraComm{comm, rank, world_size};
}
extern "C" void vra_comm_destroy(void* c) {
  VraComm* vc = static_cast<VraComm*>(c);
  if (!vc) return;
  if (vc->comm) ncclCommDestroy(vc->comm);
  delete vc;
}
extern "C" int32_t vra_comm_rank(const void* c) { return c ? static_cast<const VraComm*>(c)->rank : 0; }
extern "C" int32_t vra_comm_world_size(const void* c) { return c ? static_cast<const VraComm*>(c)->world : 1; }
extern "C" void vra_all_reduce(void* c, const void* src, void* dst, int64_t numel, int32_t dtype, int64_t stream) {
  VraComm* vc = static_cast<VraComm*>(c);
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16 || dtype == VRA_F32, "vra_all_reduce: bad dtype");
  if (!vc || vc->world == 1) {
    if (src != dst) (void)hipMemcpyAsync(dst, src, (size_t)numel * (dtype == VRA_F32 ? 4 : 2), hipMemcpyDeviceToDevice, as_stream(stream));
    return;
  }
  ncclDataType_t dt = dtype == VRA_BF16 ? ncclBfloat16 : (dtype == VRA_F16 ? ncclFloat16 : ncclFloat32);
  ncclResult_t r = ncclAllReduce(src, dst, (size_t)numel, dt, ncclSum, vc->comm, as_stream(stream));
  if (r != ncclSuccess) vra_set_error("ncclAllReduce: %s", ncclGetErrorString(r));
}
