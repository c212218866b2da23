// api_rccl.cpp -- the sharded corpus query with its one exchange step inside the library: every rank scans its
// shard and an RCCL all-reduce (ncclUint64, ncclMax) of the packed (score, ~index) key picks the global best
// match, lowest index winning ties -- the strict '<' of LBAudioDetectiveTests/LBAudioDetectiveTests.m:80-83
// across shards.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a host that never calls these entry points needs no RCCL,
// and a process that already carries an RCCL (PyTorch ships its own copy under the same soname) keeps exactly
// one -- the communicator handed in and the ncclAllReduce called here must come from the same library.
#include "internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

const OSStatus kLBAudioDetectiveCollectiveError = 0x7263636C;   // 'rccl'

namespace lbad {
namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy the process already uses, if any; else the installed one (RUNPATH of this library: /opt/rocm/lib)
        r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!r.handle) r.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            fprintf(stderr, "lbaudiodetective: cannot load librccl.so.1: %s\n", dlerror());
            return;
        }
        auto sym = [&](const char* name) { return dlsym(r.handle, name); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.AllReduce;
        if (!r.ok) fprintf(stderr, "lbaudiodetective: librccl.so.1 lacks a symbol this library binds\n");
    });
    return r;
}

OSStatus nccl_status(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return noErr;
    Rccl& r = rccl();
    fprintf(stderr, "lbaudiodetective: %s failed: %s\n", what, r.GetErrorString ? r.GetErrorString(e) : "RCCL error");
    return kLBAudioDetectiveCollectiveError;
}

// pinned landing area of the reduced keys, one per device, only growing
struct KeyOut {
    std::mutex lock;
    unsigned long long* h = nullptr;
    unsigned long long* d = nullptr;
    size_t cap = 0;
};
KeyOut g_keys[kMaxDevices];

}  // namespace
}  // namespace lbad

extern "C" {

OSStatus LBAudioDetectiveCommGetUniqueId(void* outUniqueId) {
    if (!outUniqueId) return kLBAudioDetectiveArgumentInvalid;
    lbad::Rccl& r = lbad::rccl();
    if (!r.ok) return kLBAudioDetectiveCollectiveError;
    ncclUniqueId id;
    OSStatus st = lbad::nccl_status(r.GetUniqueId(&id), "ncclGetUniqueId");
    if (st == noErr) std::memcpy(outUniqueId, &id, sizeof(id));
    return st;
}

OSStatus LBAudioDetectiveCommInitRank(void** outComm, SInt32 inNumberOfRanks, const void* inUniqueId, SInt32 inRank) {
    if (!outComm || !inUniqueId || inNumberOfRanks < 1 || inRank < 0 || inRank >= inNumberOfRanks)
        return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::device_ready()) return kLBAudioDetectiveDeviceUnavailable;
    lbad::Rccl& r = lbad::rccl();
    if (!r.ok) return kLBAudioDetectiveCollectiveError;
    ncclUniqueId id;
    std::memcpy(&id, inUniqueId, sizeof(id));
    ncclComm_t comm = nullptr;
    OSStatus st = lbad::nccl_status(r.CommInitRank(&comm, inNumberOfRanks, id, inRank), "ncclCommInitRank");
    if (st == noErr) *outComm = comm;
    return st;
}

OSStatus LBAudioDetectiveCommDestroy(void* inComm) {
    if (!inComm) return noErr;
    lbad::Rccl& r = lbad::rccl();
    if (!r.ok) return kLBAudioDetectiveCollectiveError;
    return lbad::nccl_status(r.CommDestroy(static_cast<ncclComm_t>(inComm)), "ncclCommDestroy");
}

OSStatus LBAudioDetectiveCorpusQueryBatchSharded(LBAudioDetectiveCorpusRef inCorpus,
                                                 const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                 UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                                 SInt64* outIndices, Float32* outScores) {
    if (inCount == 0 || !inComm) return kLBAudioDetectiveArgumentInvalid;
    lbad::Rccl& r = lbad::rccl();
    if (!r.ok) return kLBAudioDetectiveCollectiveError;
    const int dev = lbad::current_device();
    if (dev < 0 || dev >= lbad::kMaxDevices) return kLBAudioDetectiveDeviceUnavailable;
    lbad::KeyOut& k = lbad::g_keys[dev];
    std::lock_guard<std::mutex> guard(k.lock);
    if (k.cap < inCount) {
        unsigned long long *nd = nullptr, *nh = nullptr;
        LBAD_HIP(hipMalloc(reinterpret_cast<void**>(&nd), (size_t)inCount * 8));
        if (hipHostMalloc(reinterpret_cast<void**>(&nh), (size_t)inCount * 8, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(nd);
            return kLBAudioDetectiveDeviceError;
        }
        if (k.d) (void)hipFree(k.d);
        if (k.h) (void)hipHostFree(k.h);
        k.d = nd; k.h = nh; k.cap = inCount;
    }
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    // From here on the other ranks are (or will be) waiting in the exchange: a rank whose OWN scan cannot run -- bad
    // arguments, a shard whose indices do not fit the key's 32 bits, a failed launch -- still takes part, with keys of
    // zero ("nothing found here"), and reports its error afterwards.  Returning early would hang the other ranks.
    OSStatus local = noErr;
    if (!inCorpus || !inQueries) local = kLBAudioDetectiveArgumentInvalid;
    else if (inIndexBase + LBAudioDetectiveCorpusGetCount(inCorpus) > 0x100000000ull) local = kLBAudioDetectiveArgumentInvalid;
    else
        local = inCount == 1
            ? LBAudioDetectiveCorpusQueryKeyDevice(inCorpus, inQueries[0], inRange, inIndexBase, k.d, stream)
            : LBAudioDetectiveCorpusQueryBatchKeysDevice(inCorpus, inQueries, inCount, inRange, inIndexBase, k.d, stream);
    if (local != noErr) LBAD_HIP(hipMemsetAsync(k.d, 0, (size_t)inCount * 8, stream));
    // the one exchange step: MAX over ranks of the unsigned 64-bit keys, in place, on the caller's stream
    OSStatus st = lbad::nccl_status(r.AllReduce(k.d, k.d, inCount, ncclUint64, ncclMax, static_cast<ncclComm_t>(inComm), stream),
                                    "ncclAllReduce");
    if (st != noErr) return st;
    if (local != noErr) {
        (void)hipStreamSynchronize(stream);
        return local;
    }
    LBAD_HIP(hipMemcpyAsync(k.h, k.d, (size_t)inCount * 8, hipMemcpyDeviceToHost, stream));
    LBAD_HIP(hipStreamSynchronize(stream));
    for (UInt32 i = 0; i < inCount; ++i)
        LBAudioDetectiveCorpusDecodeKey(k.h[i], outIndices ? outIndices + i : NULL, outScores ? outScores + i : NULL);
    return noErr;
}

OSStatus LBAudioDetectiveCorpusQuerySharded(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                            UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                            SInt64* outIndex, Float32* outScore) {
    return LBAudioDetectiveCorpusQueryBatchSharded(inCorpus, &inQuery, 1, inRange, inIndexBase, inComm, inStream, outIndex,
                                                   outScore);
}

}  // extern "C"
