// the general kernel k_multi<3, INIT, 0, NT> (see qsv_kmulti_inst.h)
#include "qsv_kmulti_inst.h"
QSV_KMULTI_FOR_R_HIGH(QSV_KMULTI_DEFINE, 0, 3)
