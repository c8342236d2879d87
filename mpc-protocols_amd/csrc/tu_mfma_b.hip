#define MF_NAME launch_mfma_rows_b
#define MF_LO 6
#define MF_COUNT 4
#include "tu_mfma.inc"
