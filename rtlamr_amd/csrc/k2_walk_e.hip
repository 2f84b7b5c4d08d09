#define AMR_K2W_UNIT launch_k2_walk_e
#define AMR_K2W_SLS(X) X(160) X(176)
#include "k2_walk_launch.inc"
