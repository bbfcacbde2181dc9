// ETS kernels, additive error with a multiplicative trend and/or season.
#include "fit_units.hpp"
#define ANOFOX_UNIT_NAME seasonal_gen_a
#define ANOFOX_UNIT_VARIANTS 1
#define ANOFOX_UNIT_SPECS(X) X(10) X(13) X(2) X(5) X(8) X(11) X(14)
#include "fit_unit_impl.inc"
