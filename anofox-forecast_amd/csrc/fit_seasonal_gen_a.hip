// ETS fit kernels, additive error with a multiplicative trend and/or season.
#include "fit_units.hpp"
namespace anofox {
FitLaunchers fit_unit_seasonal_gen_a(int spec_id, int m)
{
    switch (spec_id) {
        ANOFOX_SEASONAL_CASE(10) ANOFOX_SEASONAL_CASE(13)
        ANOFOX_SEASONAL_CASE(2) ANOFOX_SEASONAL_CASE(5) ANOFOX_SEASONAL_CASE(8)
        ANOFOX_SEASONAL_CASE(11) ANOFOX_SEASONAL_CASE(14)
    default: return FitLaunchers{nullptr, nullptr, nullptr};
    }
}
} // namespace anofox
