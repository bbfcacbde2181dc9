// ts_forecast_native_hip.cpp -- DuckDB table-in-out binding of the MI355X backend.
//
// Drop-in for src/table_functions/ts_forecast_native.cpp of DataZooDE/anofox-forecast: it registers the same internal function
// `_ts_forecast_native(TABLE, horizon, frequency, method, params)` (route B), with the same output schema
// (ts_forecast_native.cpp:426-450) and the same bind-time validations (:357-399).  NOTE: the reference's SHIPPED ts_forecast_by
// macro does not reach this function -- its text expands to GROUP BY + `_ts_forecast_scalar` (src/macros/ts_macros.cpp:576-591,
// route A); binding/ts_macros_hip.cpp is the macro definition that sends ts_forecast_by (and its alias) here, and
// binding/ts_forecast_scalar_hip.cpp batches route A per DataChunk for the unchanged macro text.  Against the reference's
// `_ts_forecast_native`:
//   * collection (:476-553) appends every DataChunk as plain columns to the library's columnar ingest
//     (include/anofox_fcst_hip.h block 4) instead of boxing every row into a std::map<string, GroupData>;
//   * finalize (:559-740) makes ONE call, anofox_ts_forecast_batch (block 2), for all groups -- the library shards the
//     series ranges over the devices of ANOFOX_HIP_DEVICES / anofox_hip_set_devices -- instead of one anofox_ts_forecast call
//     per group from one thread;
//   * emission (:746-799) is unchanged in behaviour: <= STANDARD_VECTOR_SIZE rows per call, groups in first-appearance order.
// The collect -> barrier -> single-thread finalize protocol is the reference's (docs/table-in-out-parallel-execution.md).
//
// Build: add this file to EXTENSION_SOURCES in CMakeLists.txt IN PLACE OF src/table_functions/ts_forecast_native.cpp, add
// <this repo>/include to the include path and link libanofox_fcst_hip.so.  It needs DuckDB's headers
// and the extension's own ts_fill_gaps_native.hpp (frequency / date helpers, ts_fill_gaps_native.cpp:21-102); neither exists in
// the build image of this repository, so the file is compiled on the integration side only.
#include "ts_forecast_native.hpp"
#include "ts_fill_gaps_native.hpp"      // ParseFrequencyWithType, DateToMicroseconds, MicrosecondsToDate, GetGroupKey, DateColumnType
#include "anofox_fcst_hip.h"            // block 1 is layout-identical to anofox_fcst_ffi.h: do not include both
#include "duckdb/common/exception.hpp"
#include "duckdb/common/string_util.hpp"
#include "duckdb/main/config.hpp"

#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>

namespace duckdb {

namespace {

// ------------------------------------------------------------------------------------------------ parameters
struct HipForecastBind : public TableFunctionData {
    ForecastOptions options;                 // what every group is forecast with (filled once: the block is shared by the batch)
    string method = "AutoETS";
    ParsedFrequency frequency {86400, false, FrequencyType::FIXED};
    DateColumnType date_kind = DateColumnType::TIMESTAMP;
    LogicalType date_type = LogicalType::TIMESTAMP;
    LogicalType group_type = LogicalType::VARCHAR;
};

// MAP{'k': 'v'} / STRUCT / NULL -> key -> text (route A's per-row tolerance: values arrive as strings or typed fields,
// ts_forecast_scalar.cpp:85-107)
std::unordered_map<string, string> ParamTexts(const Value &params) {
    std::unordered_map<string, string> out;
    if (params.IsNull()) {
        return out;
    }
    const auto id = params.type().id();
    if (id == LogicalTypeId::MAP) {
        for (auto &entry : MapValue::GetChildren(params)) {
            auto &kv = StructValue::GetChildren(entry);
            if (kv.size() == 2 && !kv[0].IsNull() && !kv[1].IsNull()) {
                out[kv[0].ToString()] = kv[1].ToString();
            }
        }
    } else if (id == LogicalTypeId::STRUCT) {
        auto &fields = StructType::GetChildTypes(params.type());
        auto &children = StructValue::GetChildren(params);
        for (idx_t i = 0; i < children.size(); i++) {
            if (!children[i].IsNull()) {
                out[fields[i].first] = children[i].ToString();
            }
        }
    }
    return out;
}

void CopyText(char *dst, size_t cap, const string &src) {
    std::memset(dst, 0, cap);
    std::strncpy(dst, src.c_str(), cap - 1);
}

unique_ptr<FunctionData> HipForecastBindFn(ClientContext &, TableFunctionBindInput &input, vector<LogicalType> &return_types,
                                           vector<string> &names) {
    auto bind = make_uniq<HipForecastBind>();
    if (input.input_table_types.size() != 3) {
        throw InvalidInputException("_ts_forecast_native expects a table with 3 columns: group, date, value");
    }
    int64_t horizon = input.inputs.size() >= 2 ? input.inputs[1].GetValue<int64_t>() : 7;
    if (input.inputs.size() >= 3) {
        bind->frequency = ParseFrequencyWithType(input.inputs[2].ToString());       // an integer literal works too (ts_integer_frequency.test:137)
    }
    if (input.inputs.size() >= 4 && !input.inputs[3].IsNull()) {
        bind->method = input.inputs[3].ToString();
    }
    std::unordered_map<string, string> params;
    if (input.inputs.size() >= 5) {
        params = ParamTexts(input.inputs[4]);
    }
    static const std::unordered_set<string> known = {"model", "seasonal_period", "seasonal_periods", "confidence_level", "window",
                                                     "model_pool", "laplace_variant", "laplace_seasonal_batch_init"};
    string unknown;
    for (auto &kv : params) {
        if (!known.count(kv.first)) {
            unknown += (unknown.empty() ? "'" : ", '") + kv.first + "'";
        }
    }
    if (!unknown.empty()) {
        throw InvalidInputException("Unknown parameter(s): %s. Valid parameters are: model, seasonal_period, seasonal_periods, "
                                    "confidence_level, window, model_pool, laplace_variant, laplace_seasonal_batch_init", unknown);
    }
    auto text = [&](const char *key) { auto it = params.find(key); return it == params.end() ? string() : it->second; };
    auto number = [&](const char *key, double fallback) {
        auto it = params.find(key);
        if (it == params.end()) {
            return fallback;
        }
        try { return std::stod(it->second); } catch (...) { return fallback; }
    };
    auto integer = [&](const char *key, int64_t fallback) {
        auto it = params.find(key);
        if (it == params.end()) {
            return fallback;
        }
        try { return (int64_t)std::stoll(it->second); } catch (...) { return fallback; }
    };
    const string model_spec = text("model");
    const int64_t seasonal_period = (int64_t)number("seasonal_period", 0);
    const double confidence = number("confidence_level", 0.90);
    const int64_t window = (int64_t)number("window", 0);
    const string seasonal_periods = text("seasonal_periods");
    // bind-time validations of the reference (ts_forecast_native.cpp:357-399)
    if (!params.empty()) {
        if (confidence <= 0.0 || confidence >= 1.0) {
            throw InvalidInputException("Invalid confidence_level: %.2f. Must be between 0.0 and 1.0 (exclusive). Common values: 0.80 (80%%), "
                                        "0.90 (90%%), 0.95 (95%%), 0.99 (99%%)", confidence);
        }
        if (!model_spec.empty() && bind->method != "ETS") {
            throw InvalidInputException("Parameter 'model' (value: '%s') is only valid when method='ETS'. Current method is '%s'. Remove the "
                                        "'model' parameter or change method to 'ETS'.", model_spec, bind->method);
        }
        if (window != 0) {
            if (bind->method != "SMA") {
                throw InvalidInputException("Parameter 'window' is only valid when method='SMA'. Current method is '%s'. Remove the 'window' "
                                            "parameter or change method to 'SMA'.", bind->method);
            }
            if (window < 1) {
                throw InvalidInputException("Parameter 'window' must be a positive integer. Got %lld.", (long long)window);
            }
        }
        static const std::unordered_set<string> multi = {"MFLES", "AutoMFLES", "MSTL", "AutoMSTL", "TBATS", "AutoTBATS"};
        if (!seasonal_periods.empty() && !multi.count(bind->method)) {
            throw InvalidInputException("Parameter 'seasonal_periods' is only valid for multi-seasonal models (MFLES, AutoMFLES, MSTL, AutoMSTL, "
                                        "TBATS, AutoTBATS). Current method is '%s'.", bind->method);
        }
    }
    // the option block, filled the way the reference's bindings fill it (ts_forecast_scalar.cpp:439-468)
    ForecastOptions &o = bind->options;
    std::memset(&o, 0, sizeof o);
    CopyText(o.model, sizeof o.model, bind->method);
    CopyText(o.ets_model, sizeof o.ets_model, model_spec);
    o.horizon = (int)horizon;
    o.confidence_level = confidence;
    o.seasonal_period = (int)seasonal_period;
    o.auto_detect_seasonality = seasonal_period == 0 && seasonal_periods.empty();
    o.window = (int)window;
    CopyText(o.seasonal_periods_str, sizeof o.seasonal_periods_str, seasonal_periods);
    CopyText(o.model_pool, sizeof o.model_pool, text("model_pool"));
    CopyText(o.laplace_variant, sizeof o.laplace_variant, text("laplace_variant"));
    // an integer, non-zero = on, anything unparsable = the default 0 -- ParseInt64FromParams(...) != 0, ts_forecast_native.cpp:182-200, 353-354
    // ('1' enables it, 'true' does not: the reference's answer for both spellings)
    o.laplace_seasonal_batch_init = integer("laplace_seasonal_batch_init", 0) != 0;

    bind->group_type = input.input_table_types[0];
    bind->date_type = input.input_table_types[1];
    switch (bind->date_type.id()) {
    case LogicalTypeId::DATE: bind->date_kind = DateColumnType::DATE; break;
    case LogicalTypeId::TIMESTAMP:
    case LogicalTypeId::TIMESTAMP_TZ: bind->date_kind = DateColumnType::TIMESTAMP; break;
    case LogicalTypeId::INTEGER: bind->date_kind = DateColumnType::INTEGER; break;
    case LogicalTypeId::BIGINT: bind->date_kind = DateColumnType::BIGINT; break;
    default:
        throw InvalidInputException("Date column must be DATE, TIMESTAMP, INTEGER, or BIGINT, got: %s", bind->date_type.ToString());
    }
    const auto &in_names = input.input_table_names;
    names = {in_names.size() > 0 ? in_names[0] : "id", "forecast_step", in_names.size() > 1 ? in_names[1] : "date",
             "yhat", "yhat_lower", "yhat_upper", "model_name"};
    return_types = {bind->group_type, LogicalType::INTEGER, bind->date_type, LogicalType::DOUBLE, LogicalType::DOUBLE,
                    LogicalType::DOUBLE, LogicalType::VARCHAR};
    return std::move(bind);
}

// ------------------------------------------------------------------------------------------------ state
struct HipForecastLocal : public LocalTableFunctionState {
    bool collecting = false, done_collecting = false;
    bool owns_finalize = false;      // this operator instance claimed the emission (the reference's lstate.owns_finalize, ts_forecast_native.cpp:575-586)
};

struct HipForecastGlobal : public GlobalTableFunctionState {
    idx_t MaxThreads() const override { return 999999; }
    ~HipForecastGlobal() override {
        for (auto &r : results) {
            anofox_free_forecast_result(&r);
        }
        if (ingest) {
            anofox_hip_ingest_destroy(ingest);
        }
    }
    AnofoxHipIngest *ingest = anofox_hip_ingest_create();
    // dictionary of group values: text key -> dense id (the ingest sees ids only); ids are first-appearance order
    std::mutex dict_mutex;
    std::unordered_map<string, int64_t> id_of;
    vector<Value> value_of;
    // finalize: one owner, after every collector has arrived
    std::atomic<bool> claimed {false};
    std::atomic<idx_t> collectors {0}, collectors_done {0};
    bool forecast_done = false;
    size_t n_groups = 0;
    vector<ForecastResult> results;
    vector<AnofoxError> errors;
    size_t emit_group = 0, emit_step = 0;
};

unique_ptr<GlobalTableFunctionState> HipForecastInitGlobal(ClientContext &, TableFunctionInitInput &) {
    return make_uniq<HipForecastGlobal>();
}
unique_ptr<LocalTableFunctionState> HipForecastInitLocal(ExecutionContext &, TableFunctionInitInput &, GlobalTableFunctionState *) {
    return make_uniq<HipForecastLocal>();
}

int64_t DateCellToMicros(const UnifiedVectorFormat &col, idx_t idx, DateColumnType kind) {
    switch (kind) {
    case DateColumnType::DATE: return DateToMicroseconds(UnifiedVectorFormat::GetData<date_t>(col)[idx]);
    case DateColumnType::TIMESTAMP: return UnifiedVectorFormat::GetData<timestamp_t>(col)[idx].value;
    case DateColumnType::INTEGER: return UnifiedVectorFormat::GetData<int32_t>(col)[idx];
    default: return UnifiedVectorFormat::GetData<int64_t>(col)[idx];
    }
}

// ------------------------------------------------------------------------------------------------ collect
OperatorResultType HipForecastInOut(ExecutionContext &, TableFunctionInput &data, DataChunk &input, DataChunk &output) {
    auto &bind = data.bind_data->Cast<HipForecastBind>();
    auto &g = data.global_state->Cast<HipForecastGlobal>();
    auto &l = data.local_state->Cast<HipForecastLocal>();
    if (!l.collecting) {
        l.collecting = true;
        g.collectors++;
    }
    const idx_t n = input.size();
    UnifiedVectorFormat dates, values;
    input.data[1].ToUnifiedFormat(n, dates);
    input.data[2].ToUnifiedFormat(n, values);
    Vector as_double(LogicalType::DOUBLE);
    if (input.data[2].GetType().id() != LogicalTypeId::DOUBLE) {            // (the macro casts target_col::DOUBLE already, ts_macros.cpp:582)
        VectorOperations::Cast(input.data[2], as_double, n);
        as_double.ToUnifiedFormat(n, values);
    }
    // plain columns for the ingest: group id, date in microseconds (+ validity), value (+ validity)
    vector<int64_t> key(n), micros(n);
    vector<double> val(n);
    vector<uint64_t> date_ok((n + 63) / 64, 0), val_ok((n + 63) / 64, 0);
    {
        // group value -> dense id, one short critical section per chunk (the reference holds its mutex for the map insertions too)
        std::lock_guard<std::mutex> lock(g.dict_mutex);
        for (idx_t i = 0; i < n; i++) {
            Value gv = input.data[0].GetValue(i);
            auto ins = g.id_of.emplace(GetGroupKey(gv), (int64_t)g.value_of.size());
            if (ins.second) {
                g.value_of.push_back(std::move(gv));
            }
            key[i] = ins.first->second;
        }
    }
    for (idx_t i = 0; i < n; i++) {
        const idx_t di = dates.sel->get_index(i), vi = values.sel->get_index(i);
        if (dates.validity.RowIsValid(di)) {                                 // rows with a NULL date are dropped by the ingest (:505)
            micros[i] = DateCellToMicros(dates, di, bind.date_kind);
            date_ok[i / 64] |= 1ull << (i % 64);
        }
        if (values.validity.RowIsValid(vi)) {                                // a NULL target is an invalid slot: interpolated by the packer
            val[i] = UnifiedVectorFormat::GetData<double>(values)[vi];
            val_ok[i / 64] |= 1ull << (i % 64);
        }
    }
    AnofoxError err;
    if (!anofox_hip_ingest_append(g.ingest, key.data(), micros.data(), date_ok.data(), val.data(), val_ok.data(), n, &err)) {
        throw InvalidInputException("ts_forecast_by: %s", err.message);
    }
    output.SetCardinality(0);
    return OperatorResultType::NEED_MORE_INPUT;
}

// ------------------------------------------------------------------------------------------------ forecast dates
int64_t ForecastDateMicros(int64_t last, int64_t step, const HipForecastBind &bind) {
    const auto &f = bind.frequency;
    if (f.type == FrequencyType::FIXED) {
        int64_t unit;
        if (bind.date_kind == DateColumnType::INTEGER || bind.date_kind == DateColumnType::BIGINT) {
            unit = f.seconds;                                                // integer "dates": the frequency is in the column's own units
        } else {
            unit = f.is_raw ? f.seconds * 86400LL * 1000000LL : f.seconds * 1000000LL;     // a bare integer counts days on calendar columns
        }
        return last + unit * step;
    }
    // calendar steps: whole months, the day of month clamped to the target month's length (ts_forecast_scalar.cpp:250-292)
    int32_t y, m, d;
    Date::Convert(MicrosecondsToDate(last), y, m, d);
    const int64_t months = step * f.seconds * (f.type == FrequencyType::QUARTERLY ? 3 : (f.type == FrequencyType::YEARLY ? 12 : 1));
    int64_t total = (int64_t)y * 12 + (m - 1) + months;
    int32_t ny = (int32_t)(total / 12), nm = (int32_t)(total % 12) + 1;
    if (nm < 1) {
        nm += 12;
        ny -= 1;
    }
    return DateToMicroseconds(Date::FromDate(ny, nm, std::min(d, Date::MonthDays(ny, nm))));
}

Value DateValue(int64_t micros, const HipForecastBind &bind) {
    switch (bind.date_kind) {
    case DateColumnType::DATE: return Value::DATE(MicrosecondsToDate(micros));
    case DateColumnType::TIMESTAMP:
        return bind.date_type.id() == LogicalTypeId::TIMESTAMP_TZ ? Value::TIMESTAMPTZ(timestamp_tz_t(micros)) : Value::TIMESTAMP(timestamp_t(micros));
    case DateColumnType::INTEGER: return Value::INTEGER((int32_t)micros);
    default: return Value::BIGINT(micros);
    }
}

// ------------------------------------------------------------------------------------------------ finalize
OperatorFinalizeResultType HipForecastFinalize(ExecutionContext &, TableFunctionInput &data, DataChunk &output) {
    auto &bind = data.bind_data->Cast<HipForecastBind>();
    auto &g = data.global_state->Cast<HipForecastGlobal>();
    auto &l = data.local_state->Cast<HipForecastLocal>();
    if (l.collecting && !l.done_collecting) {
        l.done_collecting = true;
        g.collectors_done++;
    }
    // one thread emits everything: after source exhaustion all threads share one batch index, so rows from several threads would
    // collide in PhysicalBatchInsert (docs/table-in-out-parallel-execution.md:58-77)
    // Ownership lives in the LOCAL state, as in the reference (ts_forecast_native.cpp:575-586): HAVE_MORE_OUTPUT re-entries come
    // back with the same local state, and a later statement whose global state happens to be allocated at a freed one's address
    // can never inherit a claim (a thread_local keyed by the global state's address could).
    if (!l.owns_finalize) {
        bool expected = false;
        if (!g.claimed.compare_exchange_strong(expected, true)) {
            return OperatorFinalizeResultType::FINISHED;
        }
        l.owns_finalize = true;
        while (g.collectors_done.load() < g.collectors.load()) {
            std::this_thread::yield();
        }
    }
    if (!g.forecast_done) {
        AnofoxError err;
        size_t t_max = 0;
        if (!anofox_hip_ingest_finish(g.ingest, &g.n_groups, &t_max, &err)) {
            throw InvalidInputException("ts_forecast_by: %s", err.message);
        }
        g.results.resize(g.n_groups);
        g.errors.resize(g.n_groups);
        for (auto &r : g.results) {
            std::memset(&r, 0, sizeof r);
        }
        AnofoxError batch_error;
        // ONE call for every group (replaces the per-group loop of ts_forecast_native.cpp:588-740): sorted by date and masked by the
        // ingest, NULLs interpolated, packed, sharded over the configured devices, fitted and forecast by the library
        const bool ok = anofox_ts_forecast_batch(anofox_hip_ingest_values(g.ingest), anofox_hip_ingest_validity(g.ingest),
                                                 anofox_hip_ingest_lengths(g.ingest), g.n_groups, &bind.options, nullptr, g.results.data(),
                                                 g.errors.data(), &batch_error);
        // error policy of the reference (ts_forecast_native.cpp:666-672): INVALID_MODEL / INVALID_INPUT abort the statement, anything
        // else drops the group's rows
        if (!ok && (batch_error.code == INVALID_MODEL || batch_error.code == INVALID_INPUT)) {
            throw InvalidInputException(batch_error.message);
        }
        if (!ok) {
            throw InternalException("ts_forecast_by (HIP backend): %s", batch_error.message);
        }
        for (size_t s = 0; s < g.n_groups; s++) {
            if (g.errors[s].code == INVALID_MODEL || g.errors[s].code == INVALID_INPUT) {
                throw InvalidInputException(g.errors[s].message);
            }
        }
        g.forecast_done = true;
    }
    // emission: groups in first-appearance order (ingest order == dictionary order), <= STANDARD_VECTOR_SIZE rows per call
    const int64_t *group_ids = anofox_hip_ingest_group_keys(g.ingest);
    const int64_t *last_dates = anofox_hip_ingest_last_dates(g.ingest);
    idx_t row = 0;
    while (g.emit_group < g.n_groups && row < STANDARD_VECTOR_SIZE) {
        const ForecastResult &r = g.results[g.emit_group];
        if (g.errors[g.emit_group].code != SUCCESS || g.emit_step >= r.n_forecasts) {
            g.emit_group++;
            g.emit_step = 0;
            continue;
        }
        const size_t h = g.emit_step;
        output.SetValue(0, row, g.value_of[(size_t)group_ids[g.emit_group]]);
        output.SetValue(1, row, Value::INTEGER((int32_t)(h + 1)));
        output.SetValue(2, row, DateValue(ForecastDateMicros(last_dates[g.emit_group], (int64_t)h + 1, bind), bind));
        output.SetValue(3, row, Value::DOUBLE(r.point_forecasts[h]));
        output.SetValue(4, row, Value::DOUBLE(r.lower_bounds[h]));
        output.SetValue(5, row, Value::DOUBLE(r.upper_bounds[h]));
        output.SetValue(6, row, Value(string(r.model_name)));
        row++;
        g.emit_step++;
    }
    output.SetCardinality(row);
    return g.emit_group < g.n_groups ? OperatorFinalizeResultType::HAVE_MORE_OUTPUT : OperatorFinalizeResultType::FINISHED;
}

} // namespace

// ------------------------------------------------------------------------------------------------ registration
// Same name and signature as RegisterTsForecastNativeFunction (ts_forecast_native.cpp:806-821), so the macro text and
// anofox_forecast_extension.cpp:155 stay as they are.
void RegisterTsForecastNativeFunction(ExtensionLoader &loader) {
    TableFunction fn("_ts_forecast_native", {LogicalType::TABLE, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::ANY},
                     nullptr, HipForecastBindFn, HipForecastInitGlobal, HipForecastInitLocal);
    fn.in_out_function = HipForecastInOut;
    fn.in_out_function_final = HipForecastFinalize;
    loader.RegisterFunction(fn);

    // SET anofox_hip_arima_method = 'css' | 'css-ml' -- the caller-visible estimation method of AutoARIMA (header block 2);
    // SET anofox_hip_devices = '0,1,2,3' | 'all' | '' -- the devices the batch entry shards over
    auto &config = DBConfig::GetConfig(loader.GetDatabaseInstance());
    config.AddExtensionOption("anofox_hip_arima_method", "AutoARIMA estimation of the selected model: 'css' (default) or 'css-ml' (exact-likelihood refit)",
                              LogicalType::VARCHAR, Value("css"), [](ClientContext &, SetScope, Value &v) {
                                  const string s = StringUtil::Lower(v.ToString());
                                  if (s != "css" && s != "css-ml") {
                                      throw InvalidInputException("anofox_hip_arima_method must be 'css' or 'css-ml'");
                                  }
                                  anofox_hip_set_default_arima_method(s == "css-ml" ? ANOFOX_ARIMA_CSS_ML : ANOFOX_ARIMA_CSS);
                              });
    config.AddExtensionOption("anofox_hip_devices", "GPUs the forecast batch is sharded over: '0,1,2,3', 'all', or '' for the current device",
                              LogicalType::VARCHAR, Value(""), [](ClientContext &, SetScope, Value &v) {
                                  const string s = StringUtil::Lower(v.ToString());
                                  vector<int> devs;
                                  if (s == "all") {
                                      for (int i = 0; i < anofox_hip_device_count(); i++) {
                                          devs.push_back(i);
                                      }
                                  } else {
                                      for (auto &part : StringUtil::Split(s, ',')) {
                                          if (!part.empty()) {
                                              devs.push_back(std::stoi(part));
                                          }
                                      }
                                  }
                                  if (!anofox_hip_set_devices(devs.data(), devs.size())) {
                                      throw InvalidInputException("anofox_hip_devices names a device that is not visible");
                                  }
                              });
}

} // namespace duckdb
