#define AMR_K2W_UNIT launch_k2_walk_d
#define AMR_K2W_SLS(X) X(144)
#include "k2_walk_launch.inc"
