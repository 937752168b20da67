// comm.hip — tensor-parallel all-reduce.  Restates the AllReduce CustomOp1 of
// src/models/layers/distributed.rs:325-396 (`comm.all_reduce(src, dst, Sum)`, bf16/f16) and the
// bootstrap of src/runner/runner.rs:80-89 (`Comm::from_rank(dev, rank, world, id)`) with the
// 128-byte unique id shipped in MessageType::Init (src/runner/mod.rs:25-27).
//
// Two transports behind one communicator:
//  * RCCL (`ncclAllReduce` on the compute stream) — ring / tree over xGMI, used for large messages
//    (prefill chunks: [8192, H] = 128 MiB at H = 8192);
//  * a ONE-SHOT exchange for the decode-sized messages SURVEY §5 / §8(e) name ([T, 8192] bf16 =
//    16 KiB at bs 1 ... 512 KiB at bs 32, 160 of them per forward): every rank exposes an exchange
//    region through a hipIpcMemHandle, publishes its partial into its OWN region (write-through
//    16-byte stores), raises one flag per slice in every peer's region, and each rank then reads all
//    W partials of a slice and sums them in RANK ORDER in f32 — so every rank computes bit-identical
//    results (Appendix A21: every rank samples) and the result does not depend on arrival order.
//    One launch, no second barrier (the slots are double-buffered by the slice's own launch count,
//    which lives in device memory: the launch is hipGraph-replayable), the `+ bias` and `+ residual`
//    of TensorParallelRowLinear::forward / the decoder layer (distributed.rs:438-455, llama.rs:126,130)
//    fused behind the sum.  xGMI is point-to-point: W-1 concurrent 16..512 KiB peer reads per rank are
//    what the links are good at, a ring of 2(W-1) latency-bound hops is not.
//    The same code path serves W ranks that are processes sharing ONE GPU (how the TP product path is
//    tested on a single-GPU box): all accesses to exchange memory are system scope (sc0 sc1).
#include <rccl/rccl.h>
#include <string.h>

#include <type_traits>

#include "common.cuh"

#define OS_MAX_WORLD 8
#define OS_MAX_WG 128                    // slices per launch
#define OS_CAP (64 * 1024)               // bytes per slice and slot
#define OS_FLAG_STRIDE 16                // u32 per flag: one 64-byte line each
#define OS_FLAGS_BYTES (OS_MAX_WORLD * OS_MAX_WG * OS_FLAG_STRIDE * 4)
#define OS_SLOT_BYTES ((size_t)OS_MAX_WG * OS_CAP)
#define OS_REGION_BYTES (OS_FLAGS_BYTES + 2 * OS_SLOT_BYTES)

struct VraComm {
  ncclComm_t comm = nullptr;  // RCCL transport (may be absent: ranks sharing one GPU cannot form an RCCL communicator)
  int rank = 0, world = 1, device = 0;
  // one-shot transport
  unsigned char* local = nullptr;               // own exchange region
  unsigned char* peer[OS_MAX_WORLD] = {nullptr};  // every rank's region as mapped here (peer[rank] == local)
  bool opened[OS_MAX_WORLD] = {false};
  uint32_t* epochs = nullptr;  // [OS_MAX_WG] launches each slice index has taken part in (+ 1 error word behind them)
  bool ipc_ready = false;
  size_t oneshot_max = 0;  // messages up to this many bytes take the one-shot path when RCCL is also present
};
static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId must be 128 bytes");
static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t must be 64 bytes");

struct OneShotArgs {
  const void* src;
  void* dst;
  const void* bias;      // [cols] or null
  const void* residual;  // [n] or null (may alias dst)
  int64_t n;             // elements of this launch
  int per;               // elements per slice (multiple of 8)
  int cols;              // row length (bias broadcast)
  int rank, world;
  unsigned char* peer[OS_MAX_WORLD];
  uint32_t* epochs;
  unsigned long long timeout_ticks;  // bound of the peer wait, 100 MHz ticks
};

// KIND: 0 bf16, 1 f16, 2 f32 (no fused epilogue for f32)
template <int KIND>
__global__ __launch_bounds__(256) void oneshot_all_reduce_kernel(const OneShotArgs a) {
  constexpr int ES = KIND == 2 ? 4 : 2;   // element bytes
  constexpr int EPV = 16 / ES;            // elements per 16-byte access
  __shared__ uint32_t s_epoch;
  const int tid = threadIdx.x, j = blockIdx.x;
  if (tid == 0) {  // private memory, stream-ordered: plain accesses
    const uint32_t e = a.epochs[j] + 1u;
    a.epochs[j] = e;
    s_epoch = e;
  }
  __syncthreads();
  const uint32_t e = s_epoch;
  const size_t slot_off = OS_FLAGS_BYTES + (size_t)(e & 1u) * OS_SLOT_BYTES + (size_t)j * OS_CAP;
  const int64_t base = (int64_t)j * a.per;
  const int cnt = (int)min((int64_t)a.per, a.n - base);  // > 0 by construction of the grid
  const int nvec = (cnt + EPV - 1) / EPV;               // whole 16-byte accesses (the caller guarantees n % EPV == 0)
  const unsigned char* srcb = static_cast<const unsigned char*>(a.src) + base * ES;

  // ---- 1. publish this rank's partial of slice j into its own region (system scope, write-through)
  {
    const __amdgpu_buffer_rsrc_t mine = __builtin_amdgcn_make_buffer_rsrc(a.peer[a.rank] + slot_off, 0, OS_CAP, 0x00020000);
    for (int v = tid; v < nvec; v += 256) {
      const u32x4 x = *reinterpret_cast<const u32x4*>(srcb + (size_t)v * 16);
      __builtin_amdgcn_raw_buffer_store_b128(x, mine, v * 16, 0, 17);  // sc0 sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged by memory before the flag goes out
  }
  __syncthreads();
  // ---- 2. raise flag (rank, j) in every rank's flag array; 3. wait for all W flags of slice j in OUR array
  if (tid < a.world) {
    uint32_t* theirs = reinterpret_cast<uint32_t*>(a.peer[tid]) + ((size_t)a.rank * OS_MAX_WG + j) * OS_FLAG_STRIDE;
    __hip_atomic_store(theirs, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* ours = reinterpret_cast<const uint32_t*>(a.peer[a.rank]) + ((size_t)tid * OS_MAX_WG + j) * OS_FLAG_STRIDE;
    const uint64_t t0 = wall_clock64();
    // a peer is at most one launch ahead of us (it cannot pass ITS wait for launch e+1 without our flag).
    // The wait is bounded (never hang the device on a lost peer), and once one wait of this communicator has timed out the
    // error word stays set until the host has taken it: later launches do not wait again, so a dead peer costs ONE bound.
    const bool dead = __hip_atomic_load(a.epochs + OS_MAX_WG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    while (!dead && (int32_t)(__hip_atomic_load(ours, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - e) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > a.timeout_ticks) {
        // the first waiter to give up leaves what it saw (slice, peer, epoch expected, flag read) behind the error word
        if (__hip_atomic_exchange(a.epochs + OS_MAX_WG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          a.epochs[OS_MAX_WG + 1] = (uint32_t)j, a.epochs[OS_MAX_WG + 2] = (uint32_t)tid, a.epochs[OS_MAX_WG + 3] = e;
          a.epochs[OS_MAX_WG + 4] = __hip_atomic_load(ours, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
    }
  }
  __syncthreads();
  // ---- 4. sum the W partials of slice j in rank order (f32), fused epilogue, store
  __amdgpu_buffer_rsrc_t rs[OS_MAX_WORLD];
#pragma unroll
  for (int r = 0; r < OS_MAX_WORLD; r++)
    rs[r] = __builtin_amdgcn_make_buffer_rsrc(a.peer[r < a.world ? r : 0] + slot_off, 0, OS_CAP, 0x00020000);
  unsigned char* dstb = static_cast<unsigned char*>(a.dst) + base * ES;
  for (int v = tid; v < nvec; v += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < OS_MAX_WORLD; r++) {
      if (r < a.world) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs[r], v * 16, 0, 17);  // sc0 sc1
        if (KIND == 2) {
#pragma unroll
          for (int i = 0; i < 4; i++) acc[i] += __uint_as_float(x[i]);
        } else {
          float f[8];
          if (KIND == 0) unpack8<BF16>(x, f);
          else unpack8<F16>(x, f);
#pragma unroll
          for (int i = 0; i < 8; i++) acc[i] += f[i];
        }
      }
    }
    u32x4 o;
    if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = __float_as_uint(acc[i]);
    } else {
      using DT = typename std::conditional<KIND == 0, BF16, F16>::type;
      // all_reduce output rounds to the storage dtype; `+ bias` and `+ residual` are separate rounded ops in the reference
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = rnd_dt<DT>(acc[i]);
      if (a.bias) {
        const int64_t el = base + (int64_t)v * 8;
        const u32x4 b = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(a.bias) + (size_t)(el % a.cols) * 2);
        float f[8];
        unpack8<DT>(b, f);
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = rnd_dt<DT>(acc[i] + f[i]);
      }
      if (a.residual) {
        const u32x4 rr = *reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(a.residual) + (size_t)(base + (int64_t)v * 8) * 2);
        float f[8];
        unpack8<DT>(rr, f);
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = acc[i] + f[i];
      }
      o = pack8<DT>(acc);
    }
    *reinterpret_cast<u32x4*>(dstb + (size_t)v * 16) = o;
  }
}

// epilogue of the RCCL path: out = round(round(x + bias) + residual), 8 elements per thread
template <class DT>
__global__ void bias_residual_kernel(const void* x, void* out, const void* bias, const void* residual, int64_t nvec, int cols) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8<DT>(reinterpret_cast<const u32x4*>(x)[v], f);
    if (bias) {
      unpack8<DT>(*reinterpret_cast<const u32x4*>(static_cast<const unsigned char*>(bias) + (size_t)((v * 8) % cols) * 2), g);
#pragma unroll
      for (int i = 0; i < 8; i++) f[i] = rnd_dt<DT>(f[i] + g[i]);
    }
    if (residual) {
      unpack8<DT>(reinterpret_cast<const u32x4*>(residual)[v], g);
#pragma unroll
      for (int i = 0; i < 8; i++) f[i] = f[i] + g[i];
    }
    reinterpret_cast<u32x4*>(out)[v] = pack8<DT>(f);
  }
}

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
// Comm::from_rank (runner.rs:80-89).  world_size 1 also creates a real RCCL communicator (a one-rank group).
extern "C" void* vra_comm_create(const uint8_t h_id[128], int32_t rank, int32_t world_size, int32_t device) {
  if (world_size < 1 || world_size > 64 || rank < 0 || rank >= world_size) {
    vra_set_error("vra_comm_create: bad rank %d / world %d", rank, world_size);
    return nullptr;
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
  VraComm* c = new VraComm();
  c->comm = comm;
  c->rank = rank;
  c->world = world_size;
  c->device = device;
  return c;
}

// ---- one-shot transport: (1) every rank allocates its exchange region and exports a 64-byte handle; the launcher
// gathers the W handles (the same hand-off as the 128-byte id, runner/mod.rs:25-121) and (2) every rank maps its peers.
// `comm` may be an RCCL communicator from vra_comm_create (hybrid: one-shot below `oneshot_max_bytes`, RCCL above) or
// NULL (one-shot only: ranks that share a GPU).
extern "C" void* vra_comm_ipc_begin(void* comm, int32_t rank, int32_t world_size, int32_t device, uint8_t h_handle_out[64]) {
  if (world_size < 1 || world_size > OS_MAX_WORLD || rank < 0 || rank >= world_size) {
    vra_set_error("vra_comm_ipc_begin: bad rank %d / world %d (one-shot transport: world <= %d)", rank, world_size, OS_MAX_WORLD);
    return nullptr;
  }
  VraComm* c = static_cast<VraComm*>(comm);
  if (c && (c->rank != rank || c->world != world_size)) {
    vra_set_error("vra_comm_ipc_begin: rank/world differ from the communicator's");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    vra_set_error("vra_comm_ipc_begin: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  const bool own = c == nullptr;
  if (own) {
    c = new VraComm();
    c->rank = rank, c->world = world_size, c->device = device;
  }
  auto fail = [&](const char* what, hipError_t e) -> void* {
    vra_set_error("vra_comm_ipc_begin: %s: %s", what, hipGetErrorString(e));
    if (c->local) (void)hipFree(c->local);
    if (c->epochs) (void)hipFree(c->epochs);
    c->local = nullptr, c->epochs = nullptr;
    if (own) delete c;
    return nullptr;
  };
  // The region is polled by kernels on OTHER devices while this device's kernels write it: fine-grained (uncached) device
  // memory, so that a peer's loads over xGMI never see a line parked in this device's L2 and stores become visible without a
  // kernel boundary.  (Plain hipMalloc memory is coarse-grained: coherent at kernel boundaries only — it worked for ranks that
  // share ONE device, which is all a single-GPU test box can run.)  Falls back to hipMalloc where the runtime refuses the flag.
  hipError_t e = hipExtMallocWithFlags((void**)&c->local, OS_REGION_BYTES, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c->local = nullptr;
    e = hipExtMallocWithFlags((void**)&c->local, OS_REGION_BYTES, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c->local = nullptr;
    e = hipMalloc((void**)&c->local, OS_REGION_BYTES);
  }
  if (e != hipSuccess) return fail("hipMalloc(exchange region)", e);
  if ((e = hipMemset(c->local, 0, OS_REGION_BYTES)) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipMalloc((void**)&c->epochs, (OS_MAX_WG + 16) * 4)) != hipSuccess) return fail("hipMalloc(epochs)", e);
  if ((e = hipMemset(c->epochs, 0, (OS_MAX_WG + 16) * 4)) != hipSuccess) return fail("hipMemset", e);
  if ((e = hipDeviceSynchronize()) != hipSuccess) return fail("sync", e);
  hipIpcMemHandle_t h;
  if ((e = hipIpcGetMemHandle(&h, c->local)) != hipSuccess) return fail("hipIpcGetMemHandle (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", e);
  memcpy(h_handle_out, &h, 64);
  c->peer[rank] = c->local;
  return c;
}
extern "C" int32_t vra_comm_ipc_connect(void* comm, const uint8_t* h_all_handles, int64_t oneshot_max_bytes) {
  VraComm* c = static_cast<VraComm*>(comm);
  if (!c || !c->local) {
    vra_set_error("vra_comm_ipc_connect: communicator has no exchange region (vra_comm_ipc_begin first)");
    return -1;
  }
  (void)hipSetDevice(c->device);
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, h_all_handles + (size_t)r * 64, 64);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      vra_set_error("vra_comm_ipc_connect: hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
      return -1;
    }
    c->peer[r] = static_cast<unsigned char*>(p);
    c->opened[r] = true;
  }
  c->oneshot_max = oneshot_max_bytes > 0 ? (size_t)oneshot_max_bytes : (size_t)1 << 20;
  c->ipc_ready = true;
  return 0;
}
extern "C" void vra_comm_destroy(void* c) {
  VraComm* vc = static_cast<VraComm*>(c);
  if (!vc) return;
  if (vc->local || vc->comm) (void)hipSetDevice(vc->device);
  for (int r = 0; r < OS_MAX_WORLD; r++)
    if (vc->opened[r]) (void)hipIpcCloseMemHandle(vc->peer[r]);
  if (vc->local) (void)hipFree(vc->local);
  if (vc->epochs) (void)hipFree(vc->epochs);
  if (vc->comm) ncclCommDestroy(vc->comm);
  delete vc;
}
extern "C" int32_t vra_comm_rank(const void* c) { return c ? static_cast<const VraComm*>(c)->rank : 0; }
extern "C" int32_t vra_comm_world_size(const void* c) { return c ? static_cast<const VraComm*>(c)->world : 1; }
// 1 if a one-shot exchange gave up waiting for a peer since the last call (results of that launch are invalid)
extern "C" int32_t vra_comm_take_error(void* c) {
  VraComm* vc = static_cast<VraComm*>(c);
  if (!vc || !vc->epochs) return 0;
  uint32_t v = 0;
  if (hipMemcpy(&v, vc->epochs + OS_MAX_WG, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  if (v) (void)hipMemset(vc->epochs + OS_MAX_WG, 0, 4);
  return v != 0;
}

extern "C" int32_t vra_comm_error_detail(void* c, uint32_t h_out[4]) {
  VraComm* vc = static_cast<VraComm*>(c);
  if (!vc || !vc->epochs || !h_out) return -1;
  return hipMemcpy(h_out, vc->epochs + OS_MAX_WG + 1, 16, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

// device word behind vra_comm_take_error, for callers that fold the check into their own device-to-host copy (NULL: no one-shot transport)
extern "C" uint32_t* vra_comm_error_word(void* c) {
  VraComm* vc = static_cast<VraComm*>(c);
  return vc && vc->epochs ? vc->epochs + OS_MAX_WG : nullptr;
}

// bound of a one-shot peer wait: 4 s, or VRA_COMM_TIMEOUT_S seconds (ranks that share one GPU with more processes than the
// hardware scheduler runs concurrently can be descheduled for seconds: tests/test_gpu_tp.py raises it)
static unsigned long long oneshot_timeout_ticks() {
  static const char* e = getenv("VRA_COMM_TIMEOUT_S");
  const long sec = e && atol(e) > 0 ? atol(e) : 4;
  return (unsigned long long)sec * 100000000ull;
}
static void launch_oneshot(VraComm* vc, const void* src, void* dst, const void* bias, const void* residual, int64_t numel, int cols,
                           int dtype, hipStream_t st) {
  const int es = dtype == VRA_F32 ? 4 : 2;
  const int64_t max_el = (int64_t)OS_MAX_WG * OS_CAP / es;
  for (int64_t off = 0; off < numel; off += max_el) {  // messages above 8 MiB go out in several launches
    const int64_t n = numel - off < max_el ? numel - off : max_el;
    const int64_t bytes = n * es;
    int grid = (int)((bytes + 8191) / 8192);  // >= 8 KiB per slice: decode messages of 16 KiB use 2 workgroups
    if (grid > OS_MAX_WG) grid = OS_MAX_WG;
    if (grid < 1) grid = 1;
    int64_t per = (n + grid - 1) / grid;
    per = (per + 7) / 8 * 8;
    grid = (int)((n + per - 1) / per);
    OneShotArgs a;
    a.src = static_cast<const unsigned char*>(src) + off * es;
    a.dst = static_cast<unsigned char*>(dst) + off * es;
    a.bias = bias;
    a.residual = residual ? static_cast<const unsigned char*>(residual) + off * es : nullptr;
    a.n = n;
    a.per = (int)per;
    a.cols = cols > 0 ? cols : 8;
    a.rank = vc->rank, a.world = vc->world;
    for (int r = 0; r < OS_MAX_WORLD; r++) a.peer[r] = vc->peer[r < vc->world ? r : 0];
    a.epochs = vc->epochs;
    a.timeout_ticks = oneshot_timeout_ticks();
    if (dtype == VRA_BF16) oneshot_all_reduce_kernel<0><<<grid, 256, 0, st>>>(a);
    else if (dtype == VRA_F16) oneshot_all_reduce_kernel<1><<<grid, 256, 0, st>>>(a);
    else oneshot_all_reduce_kernel<2><<<grid, 256, 0, st>>>(a);
  }
}

static void all_reduce_impl(VraComm* vc, void* src, void* dst, const void* bias, const void* residual, int64_t numel, int cols, int dtype,
                            hipStream_t st) {
  if (numel == 0) return;
  const int es = dtype == VRA_F32 ? 4 : 2;
  const bool vec_ok = numel % (16 / es) == 0 && (!bias || cols % 8 == 0) && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
  const bool oneshot = vc->ipc_ready && vec_ok && (!vc->comm || (size_t)numel * es <= vc->oneshot_max);
  if (oneshot) {
    launch_oneshot(vc, src, dst, bias, residual, numel, cols, dtype, st);
    return;
  }
  VRA_CHECK_ARG(vc->comm != nullptr, "vra_all_reduce: message shape needs the RCCL transport, which this communicator lacks");
  ncclDataType_t dt = dtype == VRA_BF16 ? ncclBfloat16 : (dtype == VRA_F16 ? ncclFloat16 : ncclFloat32);
  const bool epi = bias || residual;
  // with an epilogue the sum is formed IN PLACE in `src` (the residual may alias dst), the epilogue kernel writes dst
  ncclResult_t r = ncclAllReduce(src, epi ? src : dst, (size_t)numel, dt, ncclSum, vc->comm, st);
  if (r != ncclSuccess) {
    vra_set_error("ncclAllReduce: %s", ncclGetErrorString(r));
    return;
  }
  if (epi) {
    VRA_CHECK_ARG(numel % 8 == 0 && cols % 8 == 0, "vra_all_reduce: fused epilogue needs multiples of 8 elements");
    const int64_t nvec = numel / 8;
    const int grid = (int)(nvec / 256 + 1 > 2048 ? 2048 : nvec / 256 + 1);
    if (dtype == VRA_BF16) bias_residual_kernel<BF16><<<grid, 256, 0, st>>>(src, dst, bias, residual, nvec, cols);
    else bias_residual_kernel<F16><<<grid, 256, 0, st>>>(src, dst, bias, residual, nvec, cols);
  }
}
// TensorParallelRowLinear::forward + the decoder layer's residual add in one call (distributed.rs:438-455, llama.rs:126,130):
// dst = all_reduce_sum(partial); dst = round(dst + bias) [bias: [cols] or NULL]; dst = dst + residual [or NULL; may alias dst].
// `partial` [rows, cols] is CLOBBERED when the RCCL transport runs with an epilogue (the sum is formed in place).
extern "C" void vra_all_reduce_fused(void* c, void* partial, void* dst, const void* bias, const void* residual, int64_t rows, int32_t cols,
                                     int32_t dtype, int64_t stream) {
  VraComm* vc = static_cast<VraComm*>(c);
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_all_reduce_fused: bf16/f16 only (distributed.rs:340-381)");
  VRA_CHECK_ARG(vc != nullptr, "vra_all_reduce_fused: null communicator");
  VRA_CHECK_ARG(rows >= 0 && cols > 0, "vra_all_reduce_fused: bad shape");
  all_reduce_impl(vc, partial, dst, bias, residual, rows * cols, cols, dtype, as_stream(stream));
}
extern "C" void vra_all_reduce(void* c, const void* src, void* dst, int64_t numel, int32_t dtype, int64_t stream) {
  VraComm* vc = static_cast<VraComm*>(c);
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16 || dtype == VRA_F32, "vra_all_reduce: bad dtype");
  VRA_CHECK_ARG(vc != nullptr, "vra_all_reduce: null communicator");
  all_reduce_impl(vc, const_cast<void*>(src), dst, nullptr, nullptr, numel, 8, dtype, as_stream(stream));
}
