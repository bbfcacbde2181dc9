// ETS fit kernels, multiplicative error with multiplicative seasonality.
#include "fit_units.hpp"
namespace anofox {
FitLaunchers fit_unit_seasonal_gen_m(int spec_id, int m)
{
    switch (spec_id) {
        ANOFOX_SEASONAL_CASE(17) ANOFOX_SEASONAL_CASE(20) ANOFOX_SEASONAL_CASE(23)
        ANOFOX_SEASONAL_CASE(26) ANOFOX_SEASONAL_CASE(29)
    default: return FitLaunchers{nullptr, nullptr, nullptr};
    }
}
} // namespace anofox
