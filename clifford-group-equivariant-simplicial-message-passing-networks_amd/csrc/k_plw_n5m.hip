// Wide parity-lane kernels for the algebra with 5 generators, negative-signature mask 0x10u.
#define CSMPN_ALG_N 5
#define CSMPN_ALG_NEG 0x10u
#define CSMPN_ALG_TAG n5m
#include "plw_inst.inc"
