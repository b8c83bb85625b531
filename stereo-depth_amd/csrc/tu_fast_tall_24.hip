#define SMX_TU_TH 24
#include "tu_fast_tall.inc"
