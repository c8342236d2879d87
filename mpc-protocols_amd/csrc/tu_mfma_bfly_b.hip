#define BF_NAME launch_mfma_bfly_b
#define BF_LO 6
#define BF_COUNT 4
#include "tu_mfma_bfly.inc"
