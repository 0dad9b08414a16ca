// channel-MFMA kernels for the algebra with 3 generators, negative-signature mask 0u.
#define CSMPN_ALG_N 3
#define CSMPN_ALG_NEG 0u
#define CSMPN_ALG_TAG n3
#include "cm_inst.inc"
