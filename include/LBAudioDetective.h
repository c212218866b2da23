/* Source-compatibility shim: upstream callers #import "LBAudioDetective.h". */
#include "lbaudiodetective.h"
