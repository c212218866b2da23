/* An Objective-C host shaped like the reference's callers (LBAudioDetectiveTests.m:57-91 and the README snippet):
 * NSURL* goes straight into the two file calls.  tests/test_capi.py compiles it with `clang -x objective-c
 * -fsyntax-only` (with and without ARC, and as Objective-C++) against include/ -- no Apple SDK is needed, the
 * header forward-declares NSURL and reaches the path through objc_msgSend. */
#import "LBAudioDetective.h"

static Float32 best_match(LBAudioDetectiveRef detective, NSURL* original, NSURL* first, NSURL* second, int* which) {
    Float32 maxMatch = 0.0f;
    for (int i = 0; i < 2; ++i) {
        Float32 match = 0.0f;
        LBAudioDetectiveCompareAudioURLs(detective, original, i ? second : first, 0, &match);
        if (maxMatch < match) {
            maxMatch = match;
            *which = i;
        }
    }
    return maxMatch;
}

int main(void) {
    LBAudioDetectiveRef detective = LBAudioDetectiveNew();
    LBAudioDetectiveFingerprintRef fingerprint = NULL;
    NSURL* url = (NSURL*)0;
    int which = -1;
    OSStatus status = LBAudioDetectiveProcessAudioURL(detective, url, &fingerprint);
    (void)best_match(detective, url, url, url, &which);
    LBAudioDetectiveFingerprintDispose(fingerprint);
    LBAudioDetectiveDispose(detective);
    return (int)status;
}
