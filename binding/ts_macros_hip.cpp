// ts_macros_hip.cpp -- `ts_forecast_by` (and its alias `anofox_fcst_ts_forecast_by`) on the batch route of the MI355X backend.
//
// What the reference ships (src/macros/ts_macros.cpp:575-591): the table macro
//     ts_forecast_by(source, group_col, date_col, target_col, method, horizon, frequency, params := MAP{})
// expands to GROUP BY group_col + two ordered LIST() aggregates + unnest(_ts_forecast_scalar(...)): route A, ONE
// anofox_ts_forecast call per group (src/scalar_functions/ts_forecast_scalar.cpp:475-482).  The table-in-out function
// `_ts_forecast_native` (route B, src/table_functions/ts_forecast_native.cpp:806-821) is registered by the reference too, but no
// shipped macro reaches it (its header comment, :18-21, still says "used by ts_forecast_by macro").
//
// What this file registers: the SAME macro name, the SAME positional parameters and the SAME default (`params := MAP{}`), with a
// body that hands the three columns to `_ts_forecast_native` -- with this backend's binding (binding/ts_forecast_native_hip.cpp)
// that is ONE anofox_ts_forecast_batch call for the whole statement instead of N per-group calls.  Output columns are the shipped
// macro's (ts_macros.cpp:577): <group_col>, forecast_step, ds, yhat, yhat_lower, yhat_upper, model_name:
//   * `_ts_forecast_native` names its first and third output column after the input table's first and second column
//     (ts_forecast_native.cpp:426-450), so the sub-select passes the group expression through unchanged and aliases the date
//     column to `ds` -- the name the shipped macro's unnest(recursive := true) gives the STRUCT field;
//   * `target_col::DOUBLE` is the shipped macro's cast (ts_macros.cpp:582; VARCHAR targets, ts_varchar_edge_cases.test:56-67);
//   * `source::VARCHAR` through query_table: a bare identifier or a quoted string (ts_table_macro_aliases.test:25);
//   * `frequency::VARCHAR`: an integer literal `1` is a frequency too (ts_integer_frequency.test:137); the table function's
//     parameter is VARCHAR (ts_forecast_native.cpp:811) and DuckDB >= 0.10 no longer casts INTEGER to VARCHAR implicitly.
// Differences a user of the shipped macro can observe (both are route B's own behaviour in the reference, not this backend's):
//   * groups come in first-appearance order (ts_forecast_native.cpp:586) where the hash aggregate of route A gives them in no
//     particular order; rows with a NULL date are dropped (:505) where route A sorts them first as date 0;
//   * route B's bind-time validations apply (:357-399: confidence range, 'model' only with ETS, 'window' only with SMA,
//     'seasonal_periods' only for multi-seasonal models) -- ts_forecast_ets_model.test:98-102 expects exactly that from
//     ts_forecast_by.
//
// Two ways to adopt it:
//   (1) source patch: replace the body text of the `ts_forecast_by` entry in ts_macros.cpp:576-591 by kForecastByBody below and
//       leave the table, CreateTableMacro and RegisterTsTableMacros (:2130-2196) as they are -- both names are registered by the
//       existing loop (:2184-2196);
//   (2) no edit of ts_macros.cpp: add this file to EXTENSION_SOURCES and call RegisterTsForecastByBatchMacros(loader) AFTER
//       RegisterTsTableMacros(loader) (src/anofox_forecast_extension.cpp:159); the entries replace the ones just registered.
// A session can do the same without any C++ (the DDL is kForecastByDDL with CREATE OR REPLACE).
//
// Needs DuckDB's headers: compiled on the integration side; here it is type-checked against a declaration-only stand-in
// (tests/test_abi_cpu.py::test_duckdb_binding_parses).
#include "duckdb.hpp"
#include "duckdb/parser/parser.hpp"
#include "duckdb/parser/parsed_data/create_macro_info.hpp"
#include "duckdb/parser/statement/create_statement.hpp"

namespace duckdb {

// The macro body (what replaces ts_macros.cpp:576-591 under adoption path (1)).
static const char *const kForecastByBody = R"SQL(
SELECT * FROM _ts_forecast_native(
    (SELECT group_col, date_col AS ds, target_col::DOUBLE AS y FROM query_table(source::VARCHAR)),
    horizon,
    frequency::VARCHAR,
    method,
    params
)
)SQL";

// The same definition as DDL; %s = the macro's name.  The parameter list is ts_macros.cpp:575 verbatim.
static const char *const kForecastByDDL =
    "CREATE OR REPLACE TEMPORARY MACRO %s(source, group_col, date_col, target_col, method, horizon, frequency, params := MAP{}) AS TABLE %s";

static unique_ptr<CreateMacroInfo> ForecastByMacro(const string &name) {
    Parser parser;
    parser.ParseQuery(StringUtil::Format(kForecastByDDL, name, string(kForecastByBody)));
    if (parser.statements.size() != 1 || parser.statements[0]->type != StatementType::CREATE_STATEMENT) {
        throw InternalException("ts_forecast_by (HIP backend): the macro definition did not parse to one CREATE statement");
    }
    auto &create = parser.statements[0]->Cast<CreateStatement>();
    if (!create.info || create.info->type != CatalogType::TABLE_MACRO_ENTRY) {
        throw InternalException("ts_forecast_by (HIP backend): the macro definition is not a table macro");
    }
    auto info = unique_ptr_cast<CreateInfo, CreateMacroInfo>(std::move(create.info));
    info->schema = DEFAULT_SCHEMA;
    info->temporary = true;
    info->internal = true;                                        // as the reference registers its macros (ts_macros.cpp:2160-2163)
    info->on_conflict = OnCreateConflict::REPLACE_ON_CONFLICT;    // path (2): takes the place of the route A entry
    FunctionDescription doc;
    doc.description = "Generates forecasts for multiple time series grouped by one or more keys (one GPU batch per statement). "
                      "Returns point forecasts with prediction intervals.";
    doc.examples.push_back("SELECT * FROM ts_forecast_by('sales', product_id, date, qty, 'AutoETS', 12, '1d')");
    doc.categories.push_back("time-series");
    doc.categories.push_back("forecasting");
    info->descriptions.push_back(std::move(doc));
    return info;
}

// Both names, as RegisterTsTableMacros registers every macro (ts_macros.cpp:2184-2196; the alias is tested by
// ts_table_macro_aliases.test:23-25).
void RegisterTsForecastByBatchMacros(ExtensionLoader &loader) {
    auto primary = ForecastByMacro("ts_forecast_by");
    loader.RegisterFunction(*primary);
    auto alias = ForecastByMacro("anofox_fcst_ts_forecast_by");
    alias->alias_of = "ts_forecast_by";
    loader.RegisterFunction(*alias);
}

} // namespace duckdb
