#define SMX_TU_TH 27
#include "tu_filter.inc"
