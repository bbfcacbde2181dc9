// TEST INFRASTRUCTURE: see ../duckdb.hpp (declaration-only stand-in for the DuckDB API; syntax check of the binding only)
#pragma once
#include "duckdb.hpp"
