// the general kernel k_multi<R, INIT, 0, NT> at R = 0, 1, 2 (see qsv_kmulti_inst.h)
#include "qsv_kmulti_inst.h"
QSV_KMULTI_FOR_R_LOW(QSV_KMULTI_DEFINE, 0, 0)
QSV_KMULTI_FOR_R_LOW(QSV_KMULTI_DEFINE, 0, 1)
QSV_KMULTI_FOR_R_LOW(QSV_KMULTI_DEFINE, 0, 2)
