// Kernels for the algebra with 4 generators, negative-signature mask 0x8u.
#define CSMPN_ALG_N 4
#define CSMPN_ALG_NEG 0x8u
#define CSMPN_ALG_TAG n4m
#include "alg_inst.inc"
