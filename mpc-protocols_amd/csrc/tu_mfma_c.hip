#define MF_NAME launch_mfma_rows_c
#define MF_LO 10
#define MF_COUNT 3
#include "tu_mfma.inc"
