// ETS fit kernels, additive class with additive seasonality: ANA, AAA, AAdA.
#include "fit_units.hpp"
namespace anofox {
FitLaunchers fit_unit_seasonal_add(int spec_id, int m)
{
    switch (spec_id) {
        ANOFOX_SEASONAL_CASE12(1) ANOFOX_SEASONAL_CASE12(4) ANOFOX_SEASONAL_CASE12(7)
    default: return FitLaunchers{nullptr, nullptr, nullptr};
    }
}
} // namespace anofox
