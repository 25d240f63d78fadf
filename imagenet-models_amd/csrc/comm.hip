// RCCL side of the C ABI (include/gaext.h, "gradient exchange"): the data-parallel gradient reduction of
// /root/reference/GA/train.py:514 (NativeDDP's bucket reducer) as explicit collectives on caller-owned buffers.
//
//  * one communicator per process (one process per GPU), created from a 128-byte unique id that rank 0 makes and the host
//    side hands to the other ranks (torch.distributed's store, a file, MPI -- not this library's business);
//  * ga_allreduce_bucket: in-place sum of one contiguous slice of the flat fp32 gradient buffer.  Wire format fp32
//    (ncclAllReduce on the slice itself) or bf16 (pack -> all-reduce -> unpack through a caller workspace: half the xGMI
//    bytes; the sum is then a bf16 sum, so it is an option, not the default);
//  * ga_reduce_scatter_bucket / ga_allgather_bucket: the two halves of the all-reduce exposed separately, for an optimizer
//    that updates only its 1/world shard between them (ZeRO-1 layout: same wire bytes, 1/world of the optimizer traffic);
//  * everything is enqueued on the hipStream_t passed in; nothing synchronises, nothing allocates.
//
// librccl is resolved at run time (dlopen): libgaext.so keeps no link-time dependency on it, loads on machines without it,
// and shares the copy PyTorch has already mapped when there is one.
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace {

typedef void* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9 };

struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {      // a copy that is already mapped (PyTorch's) first
            r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            if (r.h) break;
        }
        for (int i = 0; !r.h && i < 3; ++i) r.h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!r.h) return;
#define GA_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.h, name))
        GA_SYM(GetUniqueId, "ncclGetUniqueId");
        GA_SYM(CommInitRank, "ncclCommInitRank");
        GA_SYM(CommDestroy, "ncclCommDestroy");
        GA_SYM(AllReduce, "ncclAllReduce");
        GA_SYM(ReduceScatter, "ncclReduceScatter");
        GA_SYM(AllGather, "ncclAllGather");
        GA_SYM(Broadcast, "ncclBroadcast");
        GA_SYM(GetErrorString, "ncclGetErrorString");
#undef GA_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.Broadcast;
    });
    return r;
}

int need_rccl(const char* what) {
    if (!rccl().ok) {
        ga_set_error("%s: librccl could not be loaded (%s)", what, rccl().h ? "missing symbols" : "dlopen failed");
        return GA_ERR_UNSUPPORTED;
    }
    return GA_OK;
}

int check_nccl(int rc, const char* what) {
    if (rc != kNcclSuccess) {
        ga_set_error("%s: RCCL error %d (%s)", what, rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
        return GA_ERR_HIP;
    }
    return GA_OK;
}

// fp32 -> bf16 wire image and back (8 elements per thread; tails element-wise)
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8(src + i, v);
            store8(dst + i, v);
        } else {
            for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
        }
    }
}
__global__ __launch_bounds__(256) void unpack_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n, float scale) {
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            float v[8];
            load8(src + i, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= scale;
            store8(dst + i, v);
        } else {
            for (long j = i; j < n; ++j) dst[j] = bf2f(src[j]) * scale;
        }
    }
}
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ p, long n, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] *= scale;
}
int blocks_for(long n, int per) { return (int)std::max<long>(1, std::min<long>(2048, (n + per - 1) / per)); }

}  // namespace

struct ga_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int ga_comm_unique_id(void* id128) {
    GA_REQUIRE(id128, "ga_comm_unique_id: null");
    if (int rc = need_rccl("ga_comm_unique_id")) return rc;
    ncclUniqueId id;
    if (int rc = check_nccl(rccl().GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    memcpy(id128, id.internal, 128);
    return GA_OK;
}

extern "C" int ga_comm_init(ga_comm_t* out, int rank, int world, const void* id128) {
    GA_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "ga_comm_init: bad arguments (rank %d of %d)", rank, world);
    if (int rc = need_rccl("ga_comm_init")) return rc;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t c = nullptr;
    if (int rc = check_nccl(rccl().CommInitRank(&c, world, id, rank), "ncclCommInitRank")) return rc;
    *out = new ga_comm{c, rank, world};
    return GA_OK;
}

extern "C" int ga_comm_destroy(ga_comm_t c) {
    if (!c) return GA_OK;
    int rc = GA_OK;
    if (rccl().ok) rc = check_nccl(rccl().CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

extern "C" int ga_comm_info(ga_comm_t c, int* rank, int* world) {
    GA_REQUIRE(c, "ga_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return GA_OK;
}

extern "C" size_t ga_allreduce_workspace(int64_t n, int wire_dtype) { return wire_dtype == GA_BF16 ? (size_t)n * 2 : 0; }

extern "C" int ga_allreduce_bucket(ga_comm_t c, float* grads, int64_t n, int wire_dtype, float scale, void* workspace,
                                   size_t ws_bytes, ga_stream_t stream) {
    GA_REQUIRE(c && grads && n > 0, "ga_allreduce_bucket: bad arguments");
    GA_REQUIRE(wire_dtype == GA_F32 || wire_dtype == GA_BF16, "ga_allreduce_bucket: wire dtype %d", wire_dtype);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (wire_dtype == GA_F32) {
        if (int rc = check_nccl(rccl().AllReduce(grads, grads, (size_t)n, kNcclFloat32, kNcclSum, c->comm, s), "ncclAllReduce")) return rc;
        if (scale != 1.0f) hipLaunchKernelGGL(scale_f32_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, grads, (long)n, scale);
    } else {
        GA_REQUIRE(workspace && ws_bytes >= (size_t)n * 2 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(grads) & 15) == 0,
                   "ga_allreduce_bucket: the bf16 wire needs a 16-byte aligned workspace of %ld bytes (ga_allreduce_workspace)", (long)n * 2);
        bf16_t* w = static_cast<bf16_t*>(workspace);
        hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, s, grads, w, (long)n);
        if (int rc = check_nccl(rccl().AllReduce(w, w, (size_t)n, kNcclBfloat16, kNcclSum, c->comm, s), "ncclAllReduce(bf16)")) return rc;
        hipLaunchKernelGGL(unpack_bf16_kernel, dim3(blocks_for(n, 2048)), dim3(256), 0, s, w, grads, (long)n, scale);
    }
    return ga_check_launch("ga_allreduce_bucket");
}

extern "C" int ga_reduce_scatter_bucket(ga_comm_t c, const float* grads, float* shard, int64_t n_per_rank, float scale,
                                        ga_stream_t stream) {
    GA_REQUIRE(c && grads && shard && n_per_rank > 0, "ga_reduce_scatter_bucket: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (int rc = check_nccl(rccl().ReduceScatter(grads, shard, (size_t)n_per_rank, kNcclFloat32, kNcclSum, c->comm, s), "ncclReduceScatter"))
        return rc;
    if (scale != 1.0f) hipLaunchKernelGGL(scale_f32_kernel, dim3(blocks_for(n_per_rank, 256)), dim3(256), 0, s, shard, (long)n_per_rank, scale);
    return ga_check_launch("ga_reduce_scatter_bucket");
}

extern "C" int ga_allgather_bucket(ga_comm_t c, const float* shard, float* full, int64_t n_per_rank, ga_stream_t stream) {
    GA_REQUIRE(c && shard && full && n_per_rank > 0, "ga_allgather_bucket: bad arguments");
    return check_nccl(rccl().AllGather(shard, full, (size_t)n_per_rank, kNcclFloat32, c->comm, reinterpret_cast<hipStream_t>(stream)),
                      "ncclAllGather");
}

extern "C" int ga_comm_broadcast(ga_comm_t c, void* buf, int64_t n_f32, int root, ga_stream_t stream) {
    GA_REQUIRE(c && buf && n_f32 > 0 && root >= 0 && root < c->world, "ga_comm_broadcast: bad arguments");
    return check_nccl(rccl().Broadcast(buf, buf, (size_t)n_f32, kNcclFloat32, root, c->comm, reinterpret_cast<hipStream_t>(stream)),
                      "ncclBroadcast");
}
