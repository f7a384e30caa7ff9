// k_multi<R, INIT, 1, NT> for every tile width (see qsv_kmulti_inst.h)
#include "qsv_kmulti_inst.h"
QSV_KMULTI_FOR_MODE(QSV_KMULTI_DEFINE, 1)
