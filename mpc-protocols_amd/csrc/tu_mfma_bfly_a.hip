#define BF_NAME launch_mfma_bfly_a
#define BF_LO 2
#define BF_COUNT 4
#include "tu_mfma_bfly.inc"
