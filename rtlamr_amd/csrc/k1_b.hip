#define AMR_K1_UNIT launch_k1_b
#define AMR_K1_CASES(X) X(64) X(72)
#define AMR_K1T_DUMP_HERE 1     // diagnostic builds: this unit (chip 64 / 72) holds the timeline dump
#include "k1_launch.inc"
