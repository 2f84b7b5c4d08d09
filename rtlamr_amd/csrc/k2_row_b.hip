#define AMR_K2R_UNIT launch_k2_row_b
#define AMR_K2R_SLS(X) X(64) X(80) X(96)
#include "k2_row_launch.inc"
