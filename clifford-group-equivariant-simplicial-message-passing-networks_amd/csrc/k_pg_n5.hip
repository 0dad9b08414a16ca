// 16-row-tile MFMA-mixing kernels for the algebra with 5 generators, negative-signature mask 0u.
#define CSMPN_ALG_N 5
#define CSMPN_ALG_NEG 0u
#define CSMPN_ALG_TAG n5
#include "pg_inst.inc"
