// Kernels for the algebra with 2 generators, negative-signature mask 0u.
#define CSMPN_ALG_N 2
#define CSMPN_ALG_NEG 0u
#define CSMPN_ALG_TAG n2
#include "alg_inst.inc"
