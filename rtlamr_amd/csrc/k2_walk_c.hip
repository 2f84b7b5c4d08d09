#define AMR_K2W_UNIT launch_k2_walk_c
#define AMR_K2W_SLS(X) X(112) X(128)
#include "k2_walk_launch.inc"
