/* Source-compatibility shim: upstream callers #import "LBAudioDetectiveFingerprint.h". */
#include "lbaudiodetective.h"
