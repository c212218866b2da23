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

#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

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

// The keys of one exchange: at most kKeysPerExchange queries are reduced at a time (larger batches run as several
// exchanges, cut the same way on every rank).  The buffer of a call lives in the CORPUS (api_corpus.cpp allocates it
// with the corpus, so a sharded query never allocates); a rank that has no corpus to offer takes part with this
// module-level block of zero keys -- "nothing found here" -- whose contents nobody reads afterwards.
constexpr uint32_t kKeysPerExchange = LBAD_SHARD_KEYS;
__device__ unsigned long long g_no_keys[kKeysPerExchange];

std::atomic<uint32_t> g_exchange_timeout_ms{60000};

OSStatus rccl_all_reduce_max(void* context, UInt64* keys, UInt32 count, void* stream) {
    Rccl& r = rccl();
    if (!r.ok || !context) return kLBAudioDetectiveCollectiveError;
    return nccl_status(r.AllReduce(keys, keys, count, ncclUint64, ncclMax, static_cast<ncclComm_t>(context),
                                   static_cast<hipStream_t>(stream)), "ncclAllReduce");
}

// wait for `stream` (or, stream == nullptr, for `event`) without hanging for ever on a peer that never joined the exchange
OSStatus wait_with_deadline(hipStream_t stream, hipEvent_t event = nullptr) {
    const uint32_t limit = g_exchange_timeout_ms.load();
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        const hipError_t e = event ? hipEventQuery(event) : hipStreamQuery(stream);
        if (e == hipSuccess) return noErr;
        if (e != hipErrorNotReady) return hip_status(e, "sharded query", __LINE__);
        if (limit && (spins & 0x3FF) == 0 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit)) {
            fprintf(stderr, "lbaudiodetective: the exchange of a sharded query did not finish within %u ms (a rank missing?)\n", limit);
            return kLBAudioDetectiveCollectiveError;
        }
        if (spins > 2000) std::this_thread::yield();
    }
}

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

OSStatus LBAudioDetectiveCommGetInfo(void* inComm, SInt32* outNumberOfRanks, SInt32* outRank) {
    if (!inComm) return kLBAudioDetectiveArgumentInvalid;
    lbad::Rccl& r = lbad::rccl();
    if (!r.ok) return kLBAudioDetectiveCollectiveError;
    int count = 0, rank = 0;
    OSStatus st = lbad::nccl_status(r.CommCount(static_cast<ncclComm_t>(inComm), &count), "ncclCommCount");
    if (st == noErr) st = lbad::nccl_status(r.CommUserRank(static_cast<ncclComm_t>(inComm), &rank), "ncclCommUserRank");
    if (st != noErr) return st;
    if (outNumberOfRanks) *outNumberOfRanks = count;
    if (outRank) *outRank = rank;
    return noErr;
}

void LBAudioDetectiveSetExchangeTimeout(UInt32 inMilliseconds) { lbad::g_exchange_timeout_ms.store(inMilliseconds); }

// The sharded query, exchange step as a parameter.  Nothing between "the other ranks may already be waiting" and the
// exchange can return: the key buffer exists since the corpus was made, every local failure -- bad arguments, a shard
// whose indices do not fit the key's 32 bits, a failed launch, even a failed memset -- turns this rank's keys into zeros
// ("nothing found here") or leaves them as they are, the exchange runs, and the status comes back afterwards.
OSStatus LBAudioDetectiveCorpusQueryBatchShardedWith(LBAudioDetectiveCorpusRef inCorpus,
                                                     const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                     UInt32 inRange, UInt64 inIndexBase,
                                                     LBAudioDetectiveAllReduceMaxFn inAllReduce, void* inContext,
                                                     void* inStream, SInt64* outIndices, Float32* outScores) {
    if (inCount == 0 || !inAllReduce) return kLBAudioDetectiveArgumentInvalid;      // (the same on every rank, or a caller's bug)
    hipStream_t stream = static_cast<hipStream_t>(inStream);
    OSStatus first_error = noErr;
    // the key block is ONE per corpus: two threads with sharded queries on the same corpus take turns (round-4 advice)
    std::unique_lock<std::mutex> keys_lock;
    if (inCorpus) keys_lock = std::unique_lock<std::mutex>(inCorpus->shard_lock);
    bool block_busy = false;          // the corpus' key block may still be written by a call that gave up: hands off
    if (inCorpus && inCorpus->shard_stale) {
        // an earlier call gave up waiting: its copy and its collective may still be queued behind the block -- they must
        // have drained before the block is written again.  What is awaited is an EVENT recorded behind that work when the
        // call gave up (round-5 advice: the caller may have destroyed the stream since; an event outlives it), with the
        // same deadline.  Still stuck: this call fails on every rank alike, and -- so that the ranks stay in step -- still
        // joins the exchange, with zero keys from the spare block and WITHOUT a local scan into the busy one.
        OSStatus drained = inCorpus->shard_stale_event ? lbad::wait_with_deadline(nullptr, inCorpus->shard_stale_event) : noErr;
        if (drained != noErr) { first_error = drained; block_busy = true; }
        else inCorpus->shard_stale = false;
    }
    for (UInt32 done = 0; done < inCount; done += lbad::kKeysPerExchange) {
        const UInt32 n = inCount - done < lbad::kKeysPerExchange ? inCount - done : lbad::kKeysPerExchange;
        unsigned long long* keys = inCorpus && !block_busy ? LBAudioDetectiveCorpusShardKeysDevice(inCorpus) : nullptr;
        unsigned long long* host = inCorpus && !block_busy ? LBAudioDetectiveCorpusShardKeysHost(inCorpus) : nullptr;
        OSStatus local = noErr;
        if (block_busy) local = first_error;
        else if (!inCorpus || !inQueries || !keys || !host) local = kLBAudioDetectiveArgumentInvalid;
        else if (inIndexBase + LBAudioDetectiveCorpusGetCount(inCorpus) > 0x100000000ull) local = kLBAudioDetectiveArgumentInvalid;
        else
            local = n == 1 ? LBAudioDetectiveCorpusQueryKeyDevice(inCorpus, inQueries[done], inRange, inIndexBase, keys, stream)
                           : LBAudioDetectiveCorpusQueryBatchKeysDevice(inCorpus, inQueries + done, n, inRange, inIndexBase, keys, stream);
        if (!keys) {
            void* p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(lbad::g_no_keys)) == hipSuccess) keys = static_cast<unsigned long long*>(p);
        }
        if (local != noErr && keys) (void)hipMemsetAsync(keys, 0, (size_t)n * 8, stream);      // best effort: the exchange runs either way
        // the one exchange step: MAX over ranks of the unsigned 64-bit keys, in place, on the caller's stream
        OSStatus st = keys ? inAllReduce(inContext, reinterpret_cast<UInt64*>(keys), n, stream) : kLBAudioDetectiveDeviceError;
        if (st == noErr && local == noErr)
            st = lbad::hip_status(hipMemcpyAsync(host, keys, (size_t)n * 8, hipMemcpyDeviceToHost, stream), "keys D2H", __LINE__);
        if (st == noErr) {
            st = lbad::wait_with_deadline(stream);
            if (st != noErr && inCorpus && !block_busy) {
                // remember WHERE the stuck work ends, not the stream it sits on
                inCorpus->shard_stale = true;
                if (!inCorpus->shard_stale_event &&
                    hipEventCreateWithFlags(&inCorpus->shard_stale_event, hipEventDisableTiming) != hipSuccess) inCorpus->shard_stale_event = nullptr;
                if (inCorpus->shard_stale_event) (void)hipEventRecord(inCorpus->shard_stale_event, stream);
            }
        }
        if (st == noErr && local == noErr)
            for (UInt32 i = 0; i < n; ++i)
                LBAudioDetectiveCorpusDecodeKey(host[i], outIndices ? outIndices + done + i : NULL, outScores ? outScores + done + i : NULL);
        if (first_error == noErr) first_error = local != noErr ? local : st;
        if (st != noErr) break;            // the exchange itself failed: the communicator is not usable any more
    }
    return first_error;
}

OSStatus LBAudioDetectiveCorpusQueryBatchSharded(LBAudioDetectiveCorpusRef inCorpus,
                                                 const LBAudioDetectiveFingerprintRef* inQueries, UInt32 inCount,
                                                 UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                                 SInt64* outIndices, Float32* outScores) {
    if (inCount == 0 || !inComm) return kLBAudioDetectiveArgumentInvalid;
    if (!lbad::rccl().ok) return kLBAudioDetectiveCollectiveError;
    return LBAudioDetectiveCorpusQueryBatchShardedWith(inCorpus, inQueries, inCount, inRange, inIndexBase, lbad::rccl_all_reduce_max,
                                                       inComm, inStream, outIndices, outScores);
}

OSStatus LBAudioDetectiveCorpusQuerySharded(LBAudioDetectiveCorpusRef inCorpus, LBAudioDetectiveFingerprintRef inQuery,
                                            UInt32 inRange, UInt64 inIndexBase, void* inComm, void* inStream,
                                            SInt64* outIndex, Float32* outScore) {
    return LBAudioDetectiveCorpusQueryBatchSharded(inCorpus, &inQuery, 1, inRange, inIndexBase, inComm, inStream, outIndex,
                                                   outScore);
}

}  // extern "C"
