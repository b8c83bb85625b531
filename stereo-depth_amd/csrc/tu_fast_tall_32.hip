#define SMX_TU_TH 32
#include "tu_fast_tall.inc"
