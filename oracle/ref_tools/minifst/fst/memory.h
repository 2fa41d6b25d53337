// TEST INFRASTRUCTURE: see fstlib.h in this directory (MemoryPool lives there).
#include "fst/fstlib.h"
