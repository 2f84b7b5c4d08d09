#define AMR_K2R_UNIT launch_k2_row_a
#define AMR_K2R_CLEANUP_UNIT 1
#define AMR_K2R_SLS(X) X(144) X(16)
#include "k2_row_launch.inc"
