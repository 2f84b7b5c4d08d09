#define AMR_K2W_UNIT launch_k2_walk_a
#define AMR_K2W_SLS(X) X(16) X(64)
#include "k2_walk_launch.inc"
