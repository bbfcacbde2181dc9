// ETS kernels, multiplicative error with multiplicative seasonality.
#include "fit_units.hpp"
#define ANOFOX_UNIT_NAME seasonal_gen_m
#define ANOFOX_UNIT_VARIANTS 1
#define ANOFOX_UNIT_SPECS(X) X(17) X(20) X(23) X(26) X(29)
#include "fit_unit_impl.inc"
