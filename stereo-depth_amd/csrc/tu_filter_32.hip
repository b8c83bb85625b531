#define SMX_TU_TH 32
#include "tu_filter.inc"
