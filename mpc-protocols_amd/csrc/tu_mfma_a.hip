#define MF_NAME launch_mfma_rows_a
#define MF_LO 2
#define MF_COUNT 4
#include "tu_mfma.inc"
