#define MF_NAME launch_mfma_rows_d
#define MF_LO 13
#define MF_COUNT 3
#include "tu_mfma.inc"
