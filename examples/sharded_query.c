/* The sharded best-match query from a plain C host, one process per GPU, no Python and no torch: the corpus-compare
 * path of BASELINE.json configs[3] (10 M reference fingerprints in contiguous index shards, RCCL all-reduce of the
 * best-match key over xGMI) behind liblbaudiodetective.so alone.
 *
 * build:  gcc -std=c99 -Iinclude examples/sharded_query.c -Llbaudiodetective_amd/lib -llbaudiodetective \
 *             -Wl,-rpath,$PWD/lbaudiodetective_amd/lib -o /tmp/sharded_query
 * run:    for r in 0 1 2 3 4 5 6 7; do /tmp/sharded_query $r 8 /tmp/lbad.id 10000000 & done; wait
 *         (one rank per GPU of the node; rank 0 writes the 128-byte RCCL id to the file, the others read it)
 *
 * Every rank prints the same global (index, score): entry 7 777 777 with 7 of 100 sign pairs flipped per
 * sub-fingerprint is planted as the query, the lowest index wins ties across shards (LBAudioDetectiveTests.m:80).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "LBAudioDetective.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        OSStatus st_ = (call);                                                   \
        if (st_ != noErr) {                                                      \
            fprintf(stderr, "rank %d: %s -> OSStatus %d\n", rank, #call, (int)st_); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static int exchange_id(const char* path, int rank, unsigned char* id) {
    if (rank == 0) {
        char tmp[1024];
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, LBAD_COMM_UNIQUE_ID_BYTES, f) != LBAD_COMM_UNIQUE_ID_BYTES) return -1;
        fclose(f);
        return rename(tmp, path);                          /* appears atomically */
    }
    for (int tries = 0; tries < 6000; ++tries) {            /* up to a minute */
        FILE* f = fopen(path, "rb");
        if (f) {
            size_t n = fread(id, 1, LBAD_COMM_UNIQUE_ID_BYTES, f);
            fclose(f);
            if (n == LBAD_COMM_UNIQUE_ID_BYTES) return 0;
        }
        usleep(10000);
    }
    return -1;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s rank n_ranks id_file total_entries\n", argv[0]);
        return 2;
    }
    const int rank = atoi(argv[1]), n_ranks = atoi(argv[2]);
    const UInt64 total = strtoull(argv[4], NULL, 10);
    const UInt32 seed = 0x4C424145u, per = 5, length = 200;
    const UInt64 planted = 7777777ull % total;

    const int n_devices = LBAudioDetectiveDeviceCount();
    CHECK(n_devices > 0 ? LBAudioDetectiveDeviceSet(rank % n_devices) : kLBAudioDetectiveDeviceUnavailable);
    unsigned char id[LBAD_COMM_UNIQUE_ID_BYTES];
    if (rank == 0) CHECK(LBAudioDetectiveCommGetUniqueId(id));
    if (exchange_id(argv[3], rank, id) != 0) {
        fprintf(stderr, "rank %d: no RCCL id through %s\n", rank, argv[3]);
        return 1;
    }
    void* comm = NULL;
    CHECK(LBAudioDetectiveCommInitRank(&comm, n_ranks, id, rank));
    {   /* what RCCL itself says about the communicator: every rank joined, and this one is who it thinks it is */
        SInt32 joined = 0, me = -1;
        CHECK(LBAudioDetectiveCommGetInfo(comm, &joined, &me));
        if (joined != n_ranks || me != rank) {
            fprintf(stderr, "rank %d: the communicator reports %d rank(s), this one as %d\n", rank, (int)joined, (int)me);
            return 3;
        }
    }

    /* this rank's contiguous index range, sizes differing by at most one */
    const UInt64 base = total / n_ranks, extra = total % n_ranks;
    const UInt64 begin = rank * base + ((UInt64)rank < extra ? (UInt64)rank : extra);
    const UInt64 count = base + ((UInt64)rank < extra ? 1 : 0);

    LBAudioDetectiveCorpusRef corpus = LBAudioDetectiveCorpusNew(length, per, count ? count : 1);
    if (!corpus) return 1;
    const UInt64 chunk = 1u << 20;
    void* d_packed = NULL;
    CHECK(LBAudioDetectiveDeviceMalloc(&d_packed, chunk * per * LBAD_PACKED_BYTES));
    for (UInt64 at = 0; at < count; at += chunk) {
        const UInt64 n = count - at < chunk ? count - at : chunk;
        CHECK(LBAudioDetectiveSynthCorpusDevice(seed, begin + at, n, per, length, d_packed, NULL));
        CHECK(LBAudioDetectiveCorpusAppendPackedDevice(corpus, d_packed, n, NULL));
        CHECK(LBAudioDetectiveDeviceSynchronize());
    }

    /* the query: the planted entry with pairs 0, 13, 26, ... swapped (7 or 8 of 100 per sub-fingerprint); every rank builds the same one */
    UInt32 words[5 * LBAD_PACKED_WORDS];
    CHECK(LBAudioDetectiveSynthCorpusDevice(seed, planted, 1, per, length, d_packed, NULL));
    CHECK(LBAudioDetectiveDeviceCopyOut(words, d_packed, sizeof words));
    LBAudioDetectiveFingerprintRef query = LBAudioDetectiveFingerprintNew(length);
    for (UInt32 s = 0; s < per; ++s) {
        Boolean b[200];
        LBAudioDetectiveUnpackSubfingerprint(words + s * LBAD_PACKED_WORDS, length, b);
        for (UInt32 p = 0; p < 100; p += 13) {
            Boolean t = b[2 * p];
            b[2 * p] = b[2 * p + 1];
            b[2 * p + 1] = t;
        }
        LBAudioDetectiveFingerprintAddSubfingerprint(query, b);
    }

    SInt64 index = -1;
    Float32 score = 0.0f;
    CHECK(LBAudioDetectiveCorpusQuerySharded(corpus, query, 0, begin, comm, NULL, &index, &score));
    printf("rank %d of %d: entries [%llu, %llu) -> best match index %lld score %.6f (planted %llu)\n", rank, n_ranks,
           (unsigned long long)begin, (unsigned long long)(begin + count), (long long)index, score,
           (unsigned long long)planted);

    LBAudioDetectiveFingerprintDispose(query);
    LBAudioDetectiveDeviceFree(d_packed);
    LBAudioDetectiveCorpusDispose(corpus);
    CHECK(LBAudioDetectiveCommDestroy(comm));
    return index == (SInt64)planted ? 0 : 3;
}
