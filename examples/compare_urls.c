/* The upstream README snippet (README.md:11-17) as a plain C program against liblbaudiodetective.so:
 *
 *     LBAudioDetectiveRef detective = LBAudioDetectiveNew();
 *     LBAudioDetectiveCompareAudioURLs(detective, url1, url2, 0, &match);
 *
 * build:  gcc -Iinclude examples/compare_urls.c -Llbaudiodetective_amd/lib -llbaudiodetective \
 *             -Wl,-rpath,$PWD/lbaudiodetective_amd/lib -o /tmp/compare_urls
 * run:    /tmp/compare_urls a.caf b.caf [upstream-hop]
 */
#include <stdio.h>
#include <string.h>

#include "LBAudioDetective.h"

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s file1 file2 [upstream-hop]\n%s\n", argv[0], LBAudioDetectiveVersionString());
        return 2;
    }
    LBAudioDetectiveRef detective = LBAudioDetectiveNew();
    if (argc > 3 && strcmp(argv[3], "upstream-hop") == 0) LBAudioDetectiveSetFileHopMode(detective, 1);

    LBAudioDetectiveFingerprintRef fp = NULL;
    OSStatus status = LBAudioDetectiveProcessAudioURL(detective, argv[1], &fp);
    if (status != noErr) {
        fprintf(stderr, "cannot fingerprint %s: OSStatus %d\n", argv[1], (int)status);
        LBAudioDetectiveDispose(detective);
        return 1;
    }
    printf("%s: %u sub-fingerprints of %u Booleans\n", argv[1],
           (unsigned)LBAudioDetectiveFingerprintGetNumberOfSubfingerprints(fp),
           (unsigned)LBAudioDetectiveFingerprintGetSubfingerprintLength(fp));
    LBAudioDetectiveFingerprintDispose(fp);

    Float32 match = 0.0f;
    status = LBAudioDetectiveCompareAudioURLs(detective, argv[1], argv[2], 0, &match);
    if (status != noErr) {
        fprintf(stderr, "compare failed: OSStatus %d\n", (int)status);
        LBAudioDetectiveDispose(detective);
        return 1;
    }
    printf("match %.4f\n", match);
    LBAudioDetectiveDispose(detective);
    return 0;
}
