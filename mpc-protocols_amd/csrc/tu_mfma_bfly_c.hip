#define BF_NAME launch_mfma_bfly_c
#define BF_LO 10
#define BF_COUNT 3
#include "tu_mfma_bfly.inc"
