// Kernels for the algebra with 4 generators, negative-signature mask 0u.
#define CSMPN_ALG_N 4
#define CSMPN_ALG_NEG 0u
#define CSMPN_ALG_TAG n4
#include "alg_inst.inc"
