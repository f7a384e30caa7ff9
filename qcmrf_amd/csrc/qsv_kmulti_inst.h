// qsv_kmulti_inst.h -- the k_multi instantiations libqsv launches, one list per kernel MODE, so that each
// mode compiles in its own translation unit (qsv_kmulti_m1/2.hip; the general kernel, most of the build,
// one per tile width: qsv_kmulti_m0_r*.hip), built in parallel, while qsv.hip only declares them (extern template).
//   X(R, INIT, MODE, NT); the non-temporal form exists for the tile widths large shards use (R >= 3)
#pragma once
#include "qsv_kmulti.h"

#define QSV_KMULTI_ARGS cplx*, uint64_t, BitIns, RegPos, LanePos, const MultiOp*, const MultiSlot*, int, const cplx*, int, \
                        uint64_t, double, unsigned int, double*, uint64_t
#define QSV_KMULTI_FOR_R_LOW(X, M, R) X(R, false, M, false) X(R, true, M, false)
#define QSV_KMULTI_FOR_R_HIGH(X, M, R) X(R, false, M, false) X(R, true, M, false) X(R, false, M, true) X(R, true, M, true)
#define QSV_KMULTI_FOR_MODE(X, M)                                                                      \
  QSV_KMULTI_FOR_R_LOW(X, M, 0) QSV_KMULTI_FOR_R_LOW(X, M, 1) QSV_KMULTI_FOR_R_LOW(X, M, 2)           \
  QSV_KMULTI_FOR_R_HIGH(X, M, 3) QSV_KMULTI_FOR_R_HIGH(X, M, 4) QSV_KMULTI_FOR_R_HIGH(X, M, 5) QSV_KMULTI_FOR_R_HIGH(X, M, 6)
// the general kernel (MODE 0) holds at most 2^5 amplitudes per lane, and each width is a translation unit of its own
#define QSV_GENERAL_MAXR 5
#define QSV_KMULTI_FOR_GENERAL(X)                                                                      \
  QSV_KMULTI_FOR_R_LOW(X, 0, 0) QSV_KMULTI_FOR_R_LOW(X, 0, 1) QSV_KMULTI_FOR_R_LOW(X, 0, 2)           \
  QSV_KMULTI_FOR_R_HIGH(X, 0, 3) QSV_KMULTI_FOR_R_HIGH(X, 0, 4) QSV_KMULTI_FOR_R_HIGH(X, 0, 5)
#define QSV_KMULTI_DEFINE(R, I, M, N) template __global__ void k_multi<R, I, M, N>(QSV_KMULTI_ARGS);
#define QSV_KMULTI_DECLARE(R, I, M, N) extern template __global__ void k_multi<R, I, M, N>(QSV_KMULTI_ARGS);
