#define AMR_K1_UNIT launch_k1_a
#define AMR_K1_CASES(X) X(8) X(32) X(40) X(48) X(56)
#include "k1_launch.inc"
