/* Source-compatibility shim: upstream callers #import "LBAudioDetectiveFrame.h". */
#include "lbaudiodetective.h"
