// ETS fit kernels without a seasonal component (10 specs).
#include "fit_units.hpp"
namespace anofox {
FitLaunchers fit_unit_nonseasonal(int spec_id, int m)
{
    (void)m;
    switch (spec_id) {
        ANOFOX_NONSEASONAL_CASE(0) ANOFOX_NONSEASONAL_CASE(3) ANOFOX_NONSEASONAL_CASE(6)
        ANOFOX_NONSEASONAL_CASE(9) ANOFOX_NONSEASONAL_CASE(12) ANOFOX_NONSEASONAL_CASE(15)
        ANOFOX_NONSEASONAL_CASE(18) ANOFOX_NONSEASONAL_CASE(21) ANOFOX_NONSEASONAL_CASE(24)
        ANOFOX_NONSEASONAL_CASE(27)
    default: return FitLaunchers{nullptr, nullptr, nullptr};
    }
}
} // namespace anofox
