#define AMR_K1_UNIT launch_k1_b
#define AMR_K1_CASES(X) X(64) X(72)
#include "k1_launch.inc"
