#define SMX_TU_TH 27
#include "tu_fast_tall.inc"
