#define BF_NAME launch_mfma_bfly_d
#define BF_LO 13
#define BF_COUNT 4
#include "tu_mfma_bfly.inc"
