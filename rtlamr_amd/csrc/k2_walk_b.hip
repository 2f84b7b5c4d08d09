#define AMR_K2W_UNIT launch_k2_walk_b
#define AMR_K2W_SLS(X) X(80) X(96)
#include "k2_walk_launch.inc"
