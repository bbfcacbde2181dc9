// duckdb.hpp -- TEST INFRASTRUCTURE.  A minimal DECLARATION-ONLY stand-in for the DuckDB C++ API (v1.x), written from DuckDB's public
// interface as binding/ts_forecast_native_hip.cpp, ts_forecast_scalar_hip.cpp and ts_macros_hip.cpp use it.  It exists for ONE purpose:
// `g++ -fsyntax-only` of those files (tests/test_abi_cpu.py test_duckdb_binding_parses), because DuckDB's headers are not in this image.  Nothing here is built, linked
// or shipped; no reference build is made with it; signatures follow duckdb/src/include (types.hpp, value.hpp, vector.hpp,
// data_chunk.hpp, table_function.hpp, scalar_function.hpp, parser.hpp, create_macro_info.hpp, exception.hpp, string_util.hpp, config.hpp).  A type-check against this file says the binding is
// well-formed C++ against THESE declarations -- the real check is the extension build on the integration side.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#define STANDARD_VECTOR_SIZE 2048

namespace duckdb {
using std::string;
using idx_t = uint64_t;
template <class T> using vector = std::vector<T>;
template <class T> using unique_ptr = std::unique_ptr<T>;
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> unique_ptr<T> make_uniq(A &&...a) { return unique_ptr<T>(new T(std::forward<A>(a)...)); }
template <class T> using child_list_t = vector<std::pair<string, T>>;

struct date_t { int32_t days; };
struct timestamp_t { int64_t value; timestamp_t() = default; explicit timestamp_t(int64_t v) : value(v) {} };
struct timestamp_tz_t : public timestamp_t { timestamp_tz_t() = default; explicit timestamp_tz_t(int64_t v) : timestamp_t(v) {} };

enum class LogicalTypeId : uint8_t { INVALID, BOOLEAN, INTEGER, BIGINT, DOUBLE, VARCHAR, DATE, TIMESTAMP, TIMESTAMP_TZ, STRUCT, MAP, LIST, TABLE, ANY };
struct LogicalType {
    LogicalType();
    LogicalType(LogicalTypeId id); // NOLINT
    LogicalTypeId id() const;
    string ToString() const;
    bool operator==(const LogicalType &o) const;
    static const LogicalType INTEGER, BIGINT, DOUBLE, VARCHAR, DATE, TIMESTAMP, TIMESTAMP_TZ, TABLE, ANY;
    static LogicalType LIST(const LogicalType &child);
    static LogicalType STRUCT(child_list_t<LogicalType> children);
};
struct StructType { static const child_list_t<LogicalType> &GetChildTypes(const LogicalType &type); };
struct ListType { static const LogicalType &GetChildType(const LogicalType &type); };
struct list_entry_t { uint64_t offset; uint64_t length; };
struct string_t { string GetString() const; };

class Value {
public:
    Value();
    Value(const char *s);       // NOLINT
    Value(string s);            // NOLINT
    bool IsNull() const;
    const LogicalType &type() const;
    string ToString() const;
    template <class T> T GetValue() const;
    static Value INTEGER(int32_t v);
    static Value BIGINT(int64_t v);
    static Value DOUBLE(double v);
    static Value DATE(date_t v);
    static Value TIMESTAMP(timestamp_t v);
    static Value TIMESTAMPTZ(timestamp_tz_t v);
};
struct StructValue { static const vector<Value> &GetChildren(const Value &v); };
struct MapValue { static const vector<Value> &GetChildren(const Value &v); };

class Exception : public std::runtime_error { public: explicit Exception(const string &m) : std::runtime_error(m) {} };
class InvalidInputException : public Exception {
public:
    explicit InvalidInputException(const string &msg);
    template <class... A> explicit InvalidInputException(const string &msg, A... params);
};
class InternalException : public Exception {
public:
    explicit InternalException(const string &msg);
    template <class... A> explicit InternalException(const string &msg, A... params);
};

struct StringUtil {
    static string Lower(const string &s);
    static vector<string> Split(const string &s, char delimiter);
    template <class... A> static string Format(const string fmt, A... params);
};

struct SelectionVector { idx_t get_index(idx_t i) const; };
struct ValidityMask { bool RowIsValid(idx_t i) const; };
struct UnifiedVectorFormat {
    const SelectionVector *sel;
    const uint8_t *data;
    ValidityMask validity;
    template <class T> static const T *GetData(const UnifiedVectorFormat &f) { return reinterpret_cast<const T *>(f.data); }
};
enum class VectorType : uint8_t { FLAT_VECTOR, FSST_VECTOR, CONSTANT_VECTOR, DICTIONARY_VECTOR, SEQUENCE_VECTOR };
class Vector {
public:
    explicit Vector(LogicalType type);
    const LogicalType &GetType() const;
    void SetVectorType(VectorType type);
    void ToUnifiedFormat(idx_t count, UnifiedVectorFormat &out);
    Value GetValue(idx_t index) const;
};
struct VectorOperations { static void Cast(Vector &source, Vector &result, idx_t count); };
struct FlatVector {
    template <class T> static T *GetData(Vector &v);
    static void SetNull(Vector &v, idx_t idx, bool is_null);
};
struct ListVector {
    static Vector &GetEntry(Vector &list);
    static idx_t GetListSize(const Vector &list);
    static void Reserve(Vector &list, idx_t required_capacity);
    static void SetListSize(Vector &list, idx_t size);
};
struct StructVector { static vector<unique_ptr<Vector>> &GetEntries(Vector &v); };
struct StringVector {
    static string_t AddString(Vector &v, const char *data);
    static string_t AddString(Vector &v, const string &data);
};
class DataChunk {
public:
    vector<Vector> data;
    idx_t size() const;
    void SetCardinality(idx_t n);
    void SetValue(idx_t col, idx_t row, const Value &v);
};
struct Date {
    static void Convert(date_t d, int32_t &year, int32_t &month, int32_t &day);
    static date_t FromDate(int32_t year, int32_t month, int32_t day);
    static int32_t MonthDays(int32_t year, int32_t month);
};

class ClientContext;
class ExecutionContext;
class DatabaseInstance;
struct FunctionData {
    virtual ~FunctionData() = default;
    virtual unique_ptr<FunctionData> Copy() const;
    virtual bool Equals(const FunctionData &other) const;
    template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
    template <class T> const T &Cast() const { return reinterpret_cast<const T &>(*this); }
};
struct TableFunctionData : public FunctionData {};
struct GlobalTableFunctionState {
    virtual ~GlobalTableFunctionState() = default;
    virtual idx_t MaxThreads() const { return 1; }
    template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
};
struct LocalTableFunctionState {
    virtual ~LocalTableFunctionState() = default;
    template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
};
struct TableFunctionBindInput {
    vector<Value> &inputs;
    vector<LogicalType> &input_table_types;
    vector<string> &input_table_names;
};
struct TableFunctionInitInput {};
struct TableFunctionInput {
    const FunctionData *bind_data;
    LocalTableFunctionState *local_state;
    GlobalTableFunctionState *global_state;
};
enum class OperatorResultType : uint8_t { NEED_MORE_INPUT, HAVE_MORE_OUTPUT, FINISHED, BLOCKED };
enum class OperatorFinalizeResultType : uint8_t { HAVE_MORE_OUTPUT, FINISHED };

typedef unique_ptr<FunctionData> (*table_function_bind_t)(ClientContext &, TableFunctionBindInput &, vector<LogicalType> &, vector<string> &);
typedef unique_ptr<GlobalTableFunctionState> (*table_function_init_global_t)(ClientContext &, TableFunctionInitInput &);
typedef unique_ptr<LocalTableFunctionState> (*table_function_init_local_t)(ExecutionContext &, TableFunctionInitInput &, GlobalTableFunctionState *);
typedef void (*table_function_t)(ClientContext &, TableFunctionInput &, DataChunk &);
typedef OperatorResultType (*table_in_out_function_t)(ExecutionContext &, TableFunctionInput &, DataChunk &, DataChunk &);
typedef OperatorFinalizeResultType (*table_in_out_function_final_t)(ExecutionContext &, TableFunctionInput &, DataChunk &);
class TableFunction {
public:
    TableFunction(string name, vector<LogicalType> arguments, table_function_t function, table_function_bind_t bind = nullptr,
                  table_function_init_global_t init_global = nullptr, table_function_init_local_t init_local = nullptr);
    table_in_out_function_t in_out_function;
    table_in_out_function_final_t in_out_function_final;
};

// ---- scalar functions (scalar_function.hpp, expression.hpp, bound_function_expression.hpp, expression_executor_state.hpp)
class Expression {
public:
    virtual ~Expression() = default;
    LogicalType return_type;
    template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
    template <class T> const T &Cast() const { return reinterpret_cast<const T &>(*this); }
};
class BoundFunctionExpression : public Expression {
public:
    unique_ptr<FunctionData> bind_info;
};
struct ExpressionState { const Expression &expr; };
enum class FunctionNullHandling : uint8_t { DEFAULT_NULL_HANDLING, SPECIAL_HANDLING };
class ScalarFunction;
typedef void (*scalar_function_t)(DataChunk &, ExpressionState &, Vector &);
typedef unique_ptr<FunctionData> (*bind_scalar_function_t)(ClientContext &, ScalarFunction &, vector<unique_ptr<Expression>> &);
class ScalarFunction {
public:
    ScalarFunction(string name, vector<LogicalType> arguments, LogicalType return_type, scalar_function_t function,
                   bind_scalar_function_t bind = nullptr);
    LogicalType return_type;
    FunctionNullHandling null_handling;
};

// ---- parser and macro catalog entries (parser.hpp, sql_statement.hpp, create_statement.hpp, create_info.hpp, create_macro_info.hpp)
#define DEFAULT_SCHEMA "main"
enum class StatementType : uint8_t { INVALID_STATEMENT, SELECT_STATEMENT, CREATE_STATEMENT };
enum class CatalogType : uint8_t { INVALID, MACRO_ENTRY, TABLE_MACRO_ENTRY };
enum class OnCreateConflict : uint8_t { ERROR_ON_CONFLICT, IGNORE_ON_CONFLICT, REPLACE_ON_CONFLICT, ALTER_ON_CONFLICT };
struct FunctionDescription {
    string description;
    vector<string> examples;
    vector<string> categories;
};
struct CreateInfo {
    virtual ~CreateInfo() = default;
    CatalogType type;
    string schema;
    OnCreateConflict on_conflict;
    bool temporary;
    bool internal;
};
struct CreateFunctionInfo : public CreateInfo {
    string name;
    string alias_of;
    vector<FunctionDescription> descriptions;
};
struct CreateMacroInfo : public CreateFunctionInfo {};
class SQLStatement {
public:
    virtual ~SQLStatement() = default;
    StatementType type;
    template <class T> T &Cast() { return reinterpret_cast<T &>(*this); }
};
class CreateStatement : public SQLStatement {
public:
    unique_ptr<CreateInfo> info;
};
class Parser {
public:
    Parser();
    void ParseQuery(const string &query);
    vector<unique_ptr<SQLStatement>> statements;
};
template <class S, class T> unique_ptr<T> unique_ptr_cast(unique_ptr<S> src) { return unique_ptr<T>(static_cast<T *>(src.release())); }

class ExtensionLoader {
public:
    void RegisterFunction(TableFunction function);
    void RegisterFunction(ScalarFunction function);
    void RegisterFunction(CreateMacroInfo &function);
    DatabaseInstance &GetDatabaseInstance();
};
class Extension {                                  // extension.hpp: what anofox_forecast_extension.hpp derives from
public:
    virtual ~Extension() = default;
    virtual void Load(ExtensionLoader &loader) = 0;
    virtual std::string Name() = 0;
    virtual std::string Version() const { return ""; }
};
enum class SetScope : uint8_t { AUTOMATIC, LOCAL, SESSION, GLOBAL };
typedef void (*set_option_callback_t)(ClientContext &context, SetScope scope, Value &parameter);
struct DBConfig {
    static DBConfig &GetConfig(DatabaseInstance &db);
    void AddExtensionOption(const string &name, string description, LogicalType parameter, const Value &default_value = Value(),
                            set_option_callback_t function = nullptr);
};
} // namespace duckdb
