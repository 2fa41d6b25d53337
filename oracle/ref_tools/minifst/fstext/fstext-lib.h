// TEST INFRASTRUCTURE: stands in for the reference's fstext/fstext-lib.h (which pulls in all of Kaldi's OpenFst extensions); the
// decoder only needs the lattice weight types, and those come from the REFERENCE's own header.
#ifndef K3_MINIFST_FSTEXT_LIB_H_
#define K3_MINIFST_FSTEXT_LIB_H_
#include "fst/fstlib.h"
#include "fstext/lattice-weight.h"
#endif
