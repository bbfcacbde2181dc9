// ETS kernels, additive class with additive seasonality: ANA, AAA, AAdA.
#include "fit_units.hpp"
#define ANOFOX_UNIT_NAME seasonal_add
#define ANOFOX_UNIT_VARIANTS 2
#define ANOFOX_UNIT_SPECS(X) X(1) X(4) X(7)
#include "fit_unit_impl.inc"
