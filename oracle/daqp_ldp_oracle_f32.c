/* TEST INFRASTRUCTURE ONLY -- binary32 build of the CPU parity oracle: the same source compiled with
 * `real` = float and fmaf, exported as oracle_*_f32 (see daqp_ldp_oracle.c). */
#define ORACLE_F32 1
#include "daqp_ldp_oracle.c"
