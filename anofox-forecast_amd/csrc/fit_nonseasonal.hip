// ETS kernels without a seasonal component (10 specs).
#include "fit_units.hpp"
#define ANOFOX_UNIT_NAME nonseasonal
#define ANOFOX_UNIT_VARIANTS 0
#define ANOFOX_UNIT_SPECS(X) X(0) X(3) X(6) X(9) X(12) X(15) X(18) X(21) X(24) X(27)
#include "fit_unit_impl.inc"
