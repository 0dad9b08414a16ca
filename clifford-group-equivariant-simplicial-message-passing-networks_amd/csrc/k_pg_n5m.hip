// 16-row-tile MFMA-mixing kernels for the algebra with 5 generators, negative-signature mask 16u (Cl(4,1)).
#define CSMPN_ALG_N 5
#define CSMPN_ALG_NEG 16u
#define CSMPN_ALG_TAG n5m
#include "pg_inst.inc"
