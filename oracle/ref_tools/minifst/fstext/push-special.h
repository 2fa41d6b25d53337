// TEST INFRASTRUCTURE: declaration only (never called by the oracle tools; the programs link with --unresolved-symbols=ignore-all).
#ifndef K3_MINIFST_PUSH_SPECIAL_H_
#define K3_MINIFST_PUSH_SPECIAL_H_
#include "fst/fstlib.h"
namespace fst { template <class F> void PushSpecial(F *fst, float delta); }
#endif
