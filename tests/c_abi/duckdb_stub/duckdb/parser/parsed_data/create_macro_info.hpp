// TEST INFRASTRUCTURE: see duckdb.hpp at the stub's root (declaration-only stand-in, -fsyntax-only checks of binding/*.cpp)
#pragma once
#include "duckdb.hpp"
