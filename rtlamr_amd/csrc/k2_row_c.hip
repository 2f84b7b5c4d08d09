#define AMR_K2R_UNIT launch_k2_row_c
#define AMR_K2R_SLS(X) X(112) X(128)
#include "k2_row_launch.inc"
