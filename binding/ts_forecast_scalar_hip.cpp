// ts_forecast_scalar_hip.cpp -- `_ts_forecast_scalar` of the MI355X backend: one GPU batch per DataChunk.
//
// Drop-in for src/scalar_functions/ts_forecast_scalar.cpp of DataZooDE/anofox-forecast (route A: the scalar the SHIPPED
// ts_forecast_by macro calls once per group under GROUP BY, src/macros/ts_macros.cpp:576-591).  Same function name, argument
// types, bind (the date LIST's child type decides the `ds` field's type, ts_forecast_scalar.cpp:164-210), the same per-row
// tolerance (every row may carry its own horizon / frequency / method / params, :405-436), the same option block (:439-468),
// the same forecast dates (:250-292) and the same error policy (:484-490).  What changes is the call shape:
//
//   reference  : for each of the <= 2,048 rows of the chunk: anofox_ts_forecast(...)                       (:475-482)
//   this file  : decode all rows, group them by their option block (under the macro every row has the same one), ONE
//                anofox_ts_forecast_batch per distinct block with per-row horizons (include/anofox_fcst_hip.h block 2),
//                then write the LIST(STRUCT) rows straight into the result vector's child columns.
//
// With the link-time drop-in alone a chunk costs 2,048 one-series GPU runs (~0.9 ms each with eight
// workers coalescing, profiles/r05_single_call_latency.txt: 27 s for the 30,490 M5 series); with this file it is 15 batch calls
// of 2,048 series, shared by DuckDB's worker threads (the library is re-entrant: concurrent batches run side by side on their own
// stream sets).  The whole-statement batch of route B (binding/ts_macros_hip.cpp + ts_forecast_native_hip.cpp) is still the
// faster integration -- one launch schedule for all groups and no LIST() materialisation; this file is what makes the shipped
// macro text, and any user SQL that calls _ts_forecast_scalar directly, run at batch speed.
//
// Decoding differs from the reference in cost only: dates are read through the child vector's unified format instead of one
// boxed `Value` per element (:351-361 boxes T values per group -- 58 M for M5), and the index sort (:363-366) is skipped when the
// list is already ordered, which `LIST(... ORDER BY date_col)` guarantees under the macro; an unordered list takes a stable sort
// (the reference's std::sort leaves the order of equal dates unspecified).
//
// Needs DuckDB's headers and the extension's ts_fill_gaps_native.hpp: compiled on the integration side (replace
// src/scalar_functions/ts_forecast_scalar.cpp in EXTENSION_SOURCES by this file, add <this repo>/include, link
// libanofox_fcst_hip.so); type-checked here by tests/test_abi_cpu.py::test_duckdb_binding_parses.
#include "anofox_forecast_extension.hpp"   // void RegisterTsForecastScalarFunction(ExtensionLoader &)
#include "ts_fill_gaps_native.hpp"          // ParseFrequencyWithType, DateToMicroseconds, MicrosecondsToDate, DateColumnType
#include "anofox_fcst_hip.h"                // block 1 is layout-identical to anofox_fcst_ffi.h (shared include guard)
#include "duckdb/common/exception.hpp"
#include "duckdb/common/types/date.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/function/scalar_function.hpp"
#include "duckdb/planner/expression/bound_function_expression.hpp"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <unordered_set>

namespace duckdb {

namespace {

// ------------------------------------------------------------------------------------------------ bind
struct BatchScalarBind : public FunctionData {
    DateColumnType date_kind = DateColumnType::DATE;
    unique_ptr<FunctionData> Copy() const override {
        auto c = make_uniq<BatchScalarBind>();
        c->date_kind = date_kind;
        return std::move(c);
    }
    bool Equals(const FunctionData &other) const override {
        return date_kind == other.Cast<BatchScalarBind>().date_kind;
    }
};

unique_ptr<FunctionData> BatchScalarBindFn(ClientContext &, ScalarFunction &fn, vector<unique_ptr<Expression>> &args) {
    auto bind = make_uniq<BatchScalarBind>();
    const LogicalType &dates = args[0]->return_type;
    if (dates.id() != LogicalTypeId::LIST) {
        throw InvalidInputException("_ts_forecast_scalar: the first argument must be a LIST of dates, got: %s", dates.ToString());
    }
    const LogicalType &date_type = ListType::GetChildType(dates);
    switch (date_type.id()) {
    case LogicalTypeId::DATE: bind->date_kind = DateColumnType::DATE; break;
    case LogicalTypeId::TIMESTAMP:
    case LogicalTypeId::TIMESTAMP_TZ: bind->date_kind = DateColumnType::TIMESTAMP; break;
    case LogicalTypeId::INTEGER: bind->date_kind = DateColumnType::INTEGER; break;
    case LogicalTypeId::BIGINT: bind->date_kind = DateColumnType::BIGINT; break;
    default:
        throw InvalidInputException("Date list must contain DATE, TIMESTAMP, INTEGER, or BIGINT, got: %s", date_type.ToString());
    }
    // LIST(STRUCT(forecast_step, ds, yhat, yhat_lower, yhat_upper, model_name)): the field names the macro's
    // unnest(recursive := true) turns into its output columns (ts_macros.cpp:577)
    child_list_t<LogicalType> fields;
    fields.emplace_back("forecast_step", LogicalType::INTEGER);
    fields.emplace_back("ds", date_type);
    fields.emplace_back("yhat", LogicalType::DOUBLE);
    fields.emplace_back("yhat_lower", LogicalType::DOUBLE);
    fields.emplace_back("yhat_upper", LogicalType::DOUBLE);
    fields.emplace_back("model_name", LogicalType::VARCHAR);
    fn.return_type = LogicalType::LIST(LogicalType::STRUCT(std::move(fields)));
    return std::move(bind);
}

// ------------------------------------------------------------------------------------------------ parameters
// One row's MAP / STRUCT as the option block's fields (ts_forecast_scalar.cpp:85-158, 423-436).  Values arrive as text or as
// typed fields; a number that does not parse keeps the default, exactly like the reference's try / catch around stoll / stod.
struct RowParams {
    string model_spec, seasonal_periods, model_pool, laplace_variant;
    int64_t seasonal_period = 0, window = 0;
    double confidence = 0.90;
    bool laplace_batch_init = false;
};

const std::unordered_set<string> &KnownKeys() {
    static const std::unordered_set<string> keys = {"model", "seasonal_period", "seasonal_periods", "confidence_level", "window",
                                                    "model_pool", "laplace_variant", "laplace_seasonal_batch_init"};
    return keys;
}

RowParams DecodeParams(const Value &params) {
    RowParams p;
    if (params.IsNull()) {
        return p;
    }
    vector<std::pair<string, string>> texts;       // (key, value text) of the non-NULL entries, in the caller's order
    string unknown;
    auto see = [&](const string &key, const Value &v) {
        if (!KnownKeys().count(key)) {
            unknown += (unknown.empty() ? "'" : ", '") + key + "'";
        } else if (!v.IsNull()) {
            texts.emplace_back(key, v.ToString());
        }
    };
    if (params.type().id() == LogicalTypeId::MAP) {
        for (auto &entry : MapValue::GetChildren(params)) {
            auto &kv = StructValue::GetChildren(entry);
            see(kv[0].ToString(), kv[1]);
        }
    } else if (params.type().id() == LogicalTypeId::STRUCT) {
        auto &names = StructType::GetChildTypes(params.type());
        auto &vals = StructValue::GetChildren(params);
        for (idx_t i = 0; i < vals.size(); i++) {
            see(names[i].first, vals[i]);
        }
    }
    if (!unknown.empty()) {
        throw InvalidInputException("Unknown parameter(s): %s. Valid parameters are: model, seasonal_period, seasonal_periods, "
                                    "confidence_level, window, model_pool, laplace_variant, laplace_seasonal_batch_init", unknown);
    }
    auto text = [&](const char *key) {
        for (auto &kv : texts) {
            if (kv.first == key) {
                return kv.second;          // the first entry of a key wins, as in the reference's linear search
            }
        }
        return string();
    };
    auto whole = [&](const char *key, int64_t fallback) {
        const string s = text(key);
        if (s.empty()) {
            return fallback;
        }
        try { return (int64_t)std::stoll(s); } catch (...) { return fallback; }
    };
    p.model_spec = text("model");
    p.seasonal_period = whole("seasonal_period", 0);
    const string conf = text("confidence_level");
    if (!conf.empty()) {
        try { p.confidence = std::stod(conf); } catch (...) { p.confidence = 0.90; }
    }
    p.window = whole("window", 0);
    p.seasonal_periods = text("seasonal_periods");
    p.model_pool = text("model_pool");
    p.laplace_variant = text("laplace_variant");
    p.laplace_batch_init = whole("laplace_seasonal_batch_init", 0) != 0;
    return p;
}

void PutText(char *dst, size_t cap, const string &src) {
    std::strncpy(dst, src.c_str(), cap - 1);       // the block is zeroed before: NUL-terminated, truncated like :441-467
}

// The option block WITHOUT the horizon (per row: it travels in the batch entry's horizons[] array).
ForecastOptions MakeOptions(const string &method, const RowParams &p) {
    ForecastOptions o;
    std::memset(&o, 0, sizeof o);
    PutText(o.model, sizeof o.model, method);
    PutText(o.ets_model, sizeof o.ets_model, p.model_spec);
    o.confidence_level = p.confidence;
    o.seasonal_period = (int)p.seasonal_period;
    o.auto_detect_seasonality = p.seasonal_period == 0 && p.seasonal_periods.empty();
    o.window = (int)p.window;
    PutText(o.seasonal_periods_str, sizeof o.seasonal_periods_str, p.seasonal_periods);
    PutText(o.model_pool, sizeof o.model_pool, p.model_pool);
    PutText(o.laplace_variant, sizeof o.laplace_variant, p.laplace_variant);
    o.laplace_seasonal_batch_init = p.laplace_batch_init;
    return o;
}

// ------------------------------------------------------------------------------------------------ one decoded row
struct Row {
    idx_t at = 0;                       // row of the chunk
    vector<double> values;              // sorted by date; 0.0 in NULL slots
    vector<uint64_t> valid;             // DuckDB bitmask over `values`
    int64_t last_date = 0;              // microseconds (or the integer column's own unit)
    int horizon = 7;
    ParsedFrequency frequency {86400, false, FrequencyType::FIXED};
    size_t block = 0;                   // index into the chunk's distinct option blocks
    size_t slot = 0;                    // index inside that block's batch
};

struct Block {                          // rows that share one option block = one batch call
    ForecastOptions options;
    vector<size_t> rows;                // indices into the decoded rows
    vector<ForecastResult> results;
    vector<AnofoxError> errors;
};

int64_t DateAt(const UnifiedVectorFormat &col, idx_t idx, DateColumnType kind) {
    switch (kind) {
    case DateColumnType::DATE: return DateToMicroseconds(UnifiedVectorFormat::GetData<date_t>(col)[idx]);
    case DateColumnType::TIMESTAMP: return UnifiedVectorFormat::GetData<timestamp_t>(col)[idx].value;
    case DateColumnType::INTEGER: return UnifiedVectorFormat::GetData<int32_t>(col)[idx];
    default: return UnifiedVectorFormat::GetData<int64_t>(col)[idx];
    }
}

// last + step * frequency; calendar frequencies add whole months and clamp the day of month (ts_forecast_scalar.cpp:250-292)
int64_t ForecastDate(int64_t last, int64_t step, const ParsedFrequency &f, DateColumnType kind) {
    if (f.type == FrequencyType::FIXED) {
        int64_t unit = f.seconds;                                          // integer columns: the column's own unit
        if (kind == DateColumnType::DATE || kind == DateColumnType::TIMESTAMP) {
            unit = f.is_raw ? f.seconds * 86400LL * 1000000LL : f.seconds * 1000000LL;       // a bare integer counts days
        }
        return last + unit * step;
    }
    int32_t y, m, d;
    Date::Convert(MicrosecondsToDate(last), y, m, d);
    const int64_t months = step * f.seconds * (f.type == FrequencyType::QUARTERLY ? 3 : (f.type == FrequencyType::YEARLY ? 12 : 1));
    const int64_t total = (int64_t)y * 12 + (m - 1) + months;
    int32_t ny = (int32_t)(total / 12), nm = (int32_t)(total % 12) + 1;
    if (nm < 1) {
        nm += 12;
        ny -= 1;
    }
    return DateToMicroseconds(Date::FromDate(ny, nm, std::min(d, Date::MonthDays(ny, nm))));
}

struct BlockCleanup {                   // result arrays are the callee's malloc()s: released on every way out, exceptions included
    vector<Block> &blocks;
    ~BlockCleanup() {
        for (auto &b : blocks) {
            for (auto &r : b.results) {
                anofox_free_forecast_result(&r);
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------ execute
void BatchScalarExecute(DataChunk &args, ExpressionState &state, Vector &result) {
    const auto &bind = state.expr.Cast<BoundFunctionExpression>().bind_info->Cast<BatchScalarBind>();
    const idx_t count = args.size();
    result.SetVectorType(VectorType::FLAT_VECTOR);

    UnifiedVectorFormat date_lists, value_lists, horizons, frequencies, methods, params;
    args.data[0].ToUnifiedFormat(count, date_lists);
    args.data[1].ToUnifiedFormat(count, value_lists);
    args.data[2].ToUnifiedFormat(count, horizons);
    args.data[3].ToUnifiedFormat(count, frequencies);
    args.data[4].ToUnifiedFormat(count, methods);
    args.data[5].ToUnifiedFormat(count, params);
    UnifiedVectorFormat date_cells, value_cells;                           // the lists' child vectors, decoded once per chunk
    ListVector::GetEntry(args.data[0]).ToUnifiedFormat(ListVector::GetListSize(args.data[0]), date_cells);
    ListVector::GetEntry(args.data[1]).ToUnifiedFormat(ListVector::GetListSize(args.data[1]), value_cells);
    const auto *date_entries = UnifiedVectorFormat::GetData<list_entry_t>(date_lists);
    const auto *value_entries = UnifiedVectorFormat::GetData<list_entry_t>(value_lists);
    const auto *value_data = UnifiedVectorFormat::GetData<double>(value_cells);

    // ---- pass 1: decode every row; rows of equal (method, params) share an option block -------------------------------------
    vector<Row> rows;
    rows.reserve(count);
    vector<Block> blocks;
    BlockCleanup cleanup {blocks};
    vector<bool> is_null(count, false);
    // Under the macro `method` and `params` are constants: every row maps to the same cell, decoded once.
    idx_t seen_method_cell = (idx_t)-1, seen_params_cell = (idx_t)-1, seen_freq_cell = (idx_t)-1;
    bool seen_method_valid = false, seen_params_valid = false, have_block = false;
    size_t current_block = 0;
    ParsedFrequency current_freq {86400, false, FrequencyType::FIXED};
    bool have_freq = false;
    vector<int64_t> micros;
    vector<uint32_t> order;

    for (idx_t r = 0; r < count; r++) {
        const idx_t dl = date_lists.sel->get_index(r), vl = value_lists.sel->get_index(r);
        if (!date_lists.validity.RowIsValid(dl) || !value_lists.validity.RowIsValid(vl) || value_entries[vl].length == 0) {
            is_null[r] = true;                                             // a NULL list or an empty group: NULL row (:328-349)
            continue;
        }
        const list_entry_t values_at = value_entries[vl], dates_at = date_entries[dl];
        const idx_t n = values_at.length;
        Row row;
        row.at = r;

        // dates -> microseconds (NULL date = 0, :356-357), order by date
        micros.resize(n);
        bool ordered = true;
        for (idx_t i = 0; i < n; i++) {
            const idx_t cell = date_cells.sel->get_index(dates_at.offset + i);
            micros[i] = (i < dates_at.length && date_cells.validity.RowIsValid(cell)) ? DateAt(date_cells, cell, bind.date_kind) : 0;
            ordered = ordered && (i == 0 || micros[i - 1] <= micros[i]);
        }
        order.resize(n);
        std::iota(order.begin(), order.end(), 0u);
        if (!ordered) {
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return micros[a] < micros[b]; });
        }
        row.values.resize(n);
        row.valid.assign((n + 63) / 64, 0);
        for (idx_t i = 0; i < n; i++) {
            const idx_t cell = value_cells.sel->get_index(values_at.offset + order[i]);
            if (value_cells.validity.RowIsValid(cell)) {
                row.values[i] = value_data[cell];
                row.valid[i / 64] |= 1ull << (i % 64);
            }
        }
        row.last_date = micros[order[n - 1]];

        // horizon, frequency (per row; the reference's bind-data defaults are 7 and one day, :36-39)
        const idx_t hc = horizons.sel->get_index(r);
        if (horizons.validity.RowIsValid(hc)) {
            row.horizon = UnifiedVectorFormat::GetData<int32_t>(horizons)[hc];
        }
        const idx_t fc = frequencies.sel->get_index(r);
        if (frequencies.validity.RowIsValid(fc)) {
            if (!have_freq || fc != seen_freq_cell) {
                current_freq = ParseFrequencyWithType(UnifiedVectorFormat::GetData<string_t>(frequencies)[fc].GetString());
                seen_freq_cell = fc;
                have_freq = true;
            }
            row.frequency = current_freq;
        }

        // method + params -> option block; re-decoded only when the row points at another cell
        const idx_t mc = methods.sel->get_index(r), pc = params.sel->get_index(r);
        const bool m_valid = methods.validity.RowIsValid(mc), p_valid = params.validity.RowIsValid(pc);
        if (!have_block || mc != seen_method_cell || pc != seen_params_cell || m_valid != seen_method_valid || p_valid != seen_params_valid) {
            const string method = m_valid ? UnifiedVectorFormat::GetData<string_t>(methods)[mc].GetString() : string("AutoETS");
            const RowParams decoded = p_valid ? DecodeParams(args.data[5].GetValue(r)) : RowParams();
            const ForecastOptions o = MakeOptions(method, decoded);
            current_block = blocks.size();
            for (size_t b = 0; b < blocks.size(); b++) {
                if (std::memcmp(&blocks[b].options, &o, sizeof o) == 0) {       // both zero-filled before: padding compares equal
                    current_block = b;
                    break;
                }
            }
            if (current_block == blocks.size()) {
                blocks.emplace_back();
                blocks.back().options = o;
            }
            seen_method_cell = mc;
            seen_params_cell = pc;
            seen_method_valid = m_valid;
            seen_params_valid = p_valid;
            have_block = true;
        }
        row.block = current_block;
        row.slot = blocks[current_block].rows.size();
        blocks[current_block].rows.push_back(rows.size());
        rows.push_back(std::move(row));
    }

    // ---- pass 2: ONE batch call per distinct option block (replaces the per-row call of :475-482) ----------------------------
    for (auto &b : blocks) {
        const size_t n = b.rows.size();
        vector<const double *> vptr(n);
        vector<const uint64_t *> mptr(n);
        vector<size_t> lens(n);
        vector<int> hz(n);
        for (size_t i = 0; i < n; i++) {
            const Row &row = rows[b.rows[i]];
            vptr[i] = row.values.data();
            mptr[i] = row.valid.data();
            lens[i] = row.values.size();
            hz[i] = row.horizon;
        }
        b.results.resize(n);
        b.errors.resize(n);
        std::memset(b.results.data(), 0, n * sizeof(ForecastResult));
        std::memset(b.errors.data(), 0, n * sizeof(AnofoxError));
        b.options.horizon = n ? hz[0] : 0;                                 // informational: horizons[] decides per series
        AnofoxError batch_error;
        std::memset(&batch_error, 0, sizeof batch_error);
        if (!anofox_ts_forecast_batch(vptr.data(), mptr.data(), lens.data(), n, &b.options, hz.data(), b.results.data(),
                                      b.errors.data(), &batch_error)) {
            // a batch-level failure: an option error is uniform over the block (what every row's own call would have said), anything
            // else (no device, NULL pointers) is not a per-series condition and must not silently turn 2,048 groups into NULLs
            if (batch_error.code == INVALID_MODEL || batch_error.code == INVALID_INPUT) {
                throw InvalidInputException(string(batch_error.message));
            }
            throw InternalException("_ts_forecast_scalar (HIP backend): %s", string(batch_error.message));
        }
    }

    // ---- pass 3: the reference's error policy (:484-490), first failing row in chunk order, then the output -----------------
    idx_t total = 0;
    for (auto &row : rows) {
        const AnofoxError &e = blocks[row.block].errors[row.slot];
        if (e.code == INVALID_MODEL || e.code == INVALID_INPUT) {
            throw InvalidInputException(string(e.message));
        }
        if (e.code == SUCCESS) {
            total += blocks[row.block].results[row.slot].n_forecasts;
        }
    }
    ListVector::Reserve(result, total);
    auto *out_entries = FlatVector::GetData<list_entry_t>(result);
    auto &fields = StructVector::GetEntries(ListVector::GetEntry(result));
    auto *out_step = FlatVector::GetData<int32_t>(*fields[0]);
    auto *out_yhat = FlatVector::GetData<double>(*fields[2]);
    auto *out_lower = FlatVector::GetData<double>(*fields[3]);
    auto *out_upper = FlatVector::GetData<double>(*fields[4]);
    auto *out_name = FlatVector::GetData<string_t>(*fields[5]);
    Vector &ds = *fields[1];
    idx_t at = 0;
    for (idx_t r = 0; r < count; r++) {
        out_entries[r].offset = 0;                 // (a NULL row's entry is defined too: consumers may look at it before the validity bit)
        out_entries[r].length = 0;
        if (is_null[r]) {
            FlatVector::SetNull(result, r, true);
        }
    }
    for (auto &row : rows) {
        const Block &b = blocks[row.block];
        if (b.errors[row.slot].code != SUCCESS) {
            FlatVector::SetNull(result, row.at, true);                     // any other failure: this group yields no rows
            continue;
        }
        const ForecastResult &res = b.results[row.slot];
        out_entries[row.at].offset = at;
        out_entries[row.at].length = res.n_forecasts;
        const string_t name = StringVector::AddString(*fields[5], res.model_name);
        for (size_t i = 0; i < res.n_forecasts; i++, at++) {
            const int64_t when = ForecastDate(row.last_date, (int64_t)i + 1, row.frequency, bind.date_kind);
            out_step[at] = (int32_t)(i + 1);
            switch (bind.date_kind) {
            case DateColumnType::DATE: FlatVector::GetData<date_t>(ds)[at] = MicrosecondsToDate(when); break;
            case DateColumnType::TIMESTAMP: FlatVector::GetData<timestamp_t>(ds)[at] = timestamp_t(when); break;
            case DateColumnType::INTEGER: FlatVector::GetData<int32_t>(ds)[at] = (int32_t)when; break;
            default: FlatVector::GetData<int64_t>(ds)[at] = when; break;
            }
            out_yhat[at] = res.point_forecasts[i];
            out_lower[at] = res.lower_bounds[i];
            out_upper[at] = res.upper_bounds[i];
            out_name[at] = name;
        }
    }
    ListVector::SetListSize(result, at);
}

} // namespace

// ------------------------------------------------------------------------------------------------ registration
// Same name, argument types and NULL handling as the reference's registration (ts_forecast_scalar.cpp:529-547), so the shipped
// macro text and anofox_forecast_extension.cpp:116 stay as they are.
void RegisterTsForecastScalarFunction(ExtensionLoader &loader) {
    ScalarFunction fn("_ts_forecast_scalar",
                      {LogicalType::LIST(LogicalType::ANY), LogicalType::LIST(LogicalType::DOUBLE), LogicalType::INTEGER,
                       LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::ANY},
                      LogicalType::LIST(LogicalType::ANY), BatchScalarExecute, BatchScalarBindFn);
    fn.null_handling = FunctionNullHandling::SPECIAL_HANDLING;             // NULL lists reach the function and yield NULL rows
    loader.RegisterFunction(fn);
}

} // namespace duckdb
