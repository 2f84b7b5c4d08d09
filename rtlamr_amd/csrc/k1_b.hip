#define AMR_K1_UNIT launch_k1_b
#define AMR_K1_CASES(X) X(64) X(72)
#ifndef AMR_K1T_DUMP_C
#define AMR_K1T_DUMP_HERE 1     // diagnostic builds: this unit (chip 64 / 72) holds the timeline dump, or k1_c.hip (-DAMR_K1T_DUMP_C=1)
#endif
#include "k1_launch.inc"
