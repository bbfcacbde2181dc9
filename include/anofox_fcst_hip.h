/*
 * anofox_fcst_hip.h -- C ABI of the MI355X batch-forecasting backend.
 *
 * This is the drop-in boundary for the `ts_forecast_by` hot path of
 * DataZooDE/anofox-forecast.  The first block mirrors, byte for byte, the part
 * of the reference's cbindgen header that the path touches; the second block is
 * this backend's additive batch / device-resident interface.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   ErrorCode            src/include/anofox_fcst_ffi.h:72-84
 *   AnofoxError          src/include/anofox_fcst_ffi.h:269-272
 *   ForecastOptions      src/include/anofox_fcst_ffi.h:1036-1095   (184 bytes)
 *   ForecastResult       src/include/anofox_fcst_ffi.h:1100-1145   (144 bytes)
 *   anofox_ts_forecast   src/include/anofox_fcst_ffi.h:2332-2337
 *                        (Rust body crates/anofox-fcst-ffi/src/lib.rs:3344-3550)
 *   anofox_free_forecast_result  anofox_fcst_ffi.h:2886 (lib.rs:5900-5930)
 *   anofox_fcst_version  anofox_fcst_ffi.h:3058
 *
 * Plain C, SysV x86-64, no torch / HIP types in any signature: a device stream
 * is passed as an opaque `void *` (a hipStream_t), device buffers as `void *`.
 */
#ifndef ANOFOX_FCST_HIP_H
#define ANOFOX_FCST_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Block 1: the reference's ABI for this path (layout-identical)              */
/* ------------------------------------------------------------------------- */

#ifndef ANOFOX_FCST_FFI_H /* the reference header defines the same names */

typedef enum ErrorCode {
    SUCCESS = 0,
    NULL_POINTER = 1,
    INVALID_INPUT = 2,
    COMPUTATION_ERROR = 3,
    ALLOCATION_ERROR = 4,
    INVALID_MODEL = 5,
    INSUFFICIENT_DATA = 6,
    INVALID_DATE_FORMAT = 7,
    INVALID_FREQUENCY = 8,
    PANIC_CAUGHT = 9,
    INTERNAL_ERROR = 10,
} ErrorCode;

typedef struct AnofoxError {
    enum ErrorCode code;
    char message[256]; /* NUL-terminated, truncated at 255 */
} AnofoxError;

typedef struct ForecastOptions {
    char model[32];                 /* method name, exact or alias            */
    char ets_model[8];              /* "AAA", "MNM", "AAdA"...; "" = no spec   */
    int horizon;
    double confidence_level;
    int seasonal_period;            /* 0 = not given                          */
    bool auto_detect_seasonality;
    bool include_fitted;
    bool include_residuals;
    int window;                     /* SMA window, 0 = default                */
    char seasonal_periods_str[64];  /* "[24, 168]"; multi-seasonal models only */
    char model_pool[32];            /* AutoETS pool; "" = complete            */
    char laplace_variant[16];       /* unused by this backend                 */
    bool laplace_seasonal_batch_init;
} ForecastOptions;

typedef struct ForecastResult {
    double *point_forecasts;        /* malloc'd, n_forecasts                  */
    double *lower_bounds;
    double *upper_bounds;
    double *fitted_values;          /* malloc'd, n_fitted, or NULL            */
    double *residuals;
    size_t n_forecasts;
    size_t n_fitted;
    char model_name[64];
    double aic;                     /* always NaN on this path                */
    double bic;
    double mse;                     /* NaN unless fitted values requested     */
} ForecastResult;

/*
 * Fit + forecast ONE series.  `values[length]` is sorted by date by the caller;
 * `validity` is a DuckDB bitmask (bit i%64 of word i/64, 1 = valid) or NULL.
 * Returns false and fills `out_error` (may be NULL) on failure; on success the
 * callee malloc()s the result arrays, released by anofox_free_forecast_result.
 * Runs on the GPU as a batch of one; there is no CPU fallback.
 */
bool anofox_ts_forecast(const double *values,
                        const uint64_t *validity,
                        size_t length,
                        const struct ForecastOptions *options,
                        struct ForecastResult *out_result,
                        struct AnofoxError *out_error);

void anofox_free_forecast_result(struct ForecastResult *result);

const char *anofox_fcst_version(void);

#endif /* ANOFOX_FCST_FFI_H */

/* ------------------------------------------------------------------------- */
/* Block 2: batch entry (host buffers) -- additive, same per-series semantics */
/* ------------------------------------------------------------------------- */

/*
 * Fit + forecast `n_series` independent series with ONE shared option block in
 * one GPU pass.  Replaces the serial per-group loop of the reference binding
 * (src/table_functions/ts_forecast_native.cpp:586-740, one anofox_ts_forecast
 * call per group).  `horizons` may be NULL (= options->horizon for all) or give
 * a per-series horizon (cross-validation folds, ts_cv_forecast_native.cpp:676).
 * Per-series failures land in out_errors[i] with out_results[i] zeroed; the
 * return value is false only for batch-level failures (NULL pointers, no GPU).
 * INVALID_MODEL / INVALID_INPUT are uniform across the batch and are also
 * reported through `out_batch_error` (may be NULL).
 */
bool anofox_ts_forecast_batch(const double *const *values,
                              const uint64_t *const *validity,
                              const size_t *lengths,
                              size_t n_series,
                              const struct ForecastOptions *options,
                              const int *horizons,
                              struct ForecastResult *out_results,
                              struct AnofoxError *out_errors,
                              struct AnofoxError *out_batch_error);

/*
 * Multi-device execution of the batch entry.  The reference's finalize loop is ONE process walking all groups
 * (src/table_functions/ts_forecast_native.cpp:559-800); series are independent, so anofox_ts_forecast_batch shards
 * contiguous series-id ranges [g * ceil(N / G), (g + 1) * ceil(N / G)) over the G devices named here: one host thread,
 * stream set and allocator cache per device, each range packed into its device's own pinned staging block, results
 * written straight into the caller's arrays (host side: no RCCL involved).  Default: the caller's current device only.
 * `devices` = ordinal list (a device may be listed more than once: its ranges then run as separate batches on it);
 * n_devices = 0 restores the default.  The environment variable ANOFOX_HIP_DEVICES ("0,1,2,3" or "all"), read once at
 * the first batch call, gives the initial list.  Batches smaller than `min_series_per_device` (default 2,048) per
 * device use fewer devices.  Returns false (nothing changed) when an ordinal is not a visible device.
 */
bool anofox_hip_set_devices(const int *devices, size_t n_devices);
size_t anofox_hip_get_devices(int *devices, size_t capacity);      /* returns the number of devices in use (0 = default) */
void anofox_hip_set_min_series_per_device(size_t min_series);
/* The range of shard `shard` of `n_shards`: [shard * ceil(N / n_shards), ...) clipped to N. */
void anofox_hip_shard_range(size_t n_series, size_t n_shards, size_t shard, size_t *begin, size_t *end);

/*
 * AutoARIMA estimation method -- a caller-visible choice (the reference's `AutoARIMAConfig::default()`,
 * forecast.rs:1447-1455, exposes none, and its measured cost, benchmark/README.md:55, leaves no room for an exact-likelihood
 * refit: DESIGN.md section 3).  ANOFOX_ARIMA_CSS (default): the selected model keeps the conditional-sum-of-squares
 * estimates of the search.  ANOFOX_ARIMA_CSS_ML: the selected model is re-estimated on the exact Gaussian likelihood
 * (Kalman filter of the Harvey state space, evaluated through the Chandrasekhar recursions: BASELINE.json's "Kalman
 * kernel").  The process default applies to anofox_ts_forecast / anofox_ts_forecast_batch and to batches created afterwards;
 * anofox_hip_batch_set_arima_method (block 3) overrides it per batch.
 */
enum { ANOFOX_ARIMA_CSS = 0, ANOFOX_ARIMA_CSS_ML = 1 };
bool anofox_hip_set_default_arima_method(int method);

/*
 * The library keeps idle device blocks (per device at most ANOFOX_HIP_CACHE_GB, default a quarter of the device, oldest
 * evicted first), pinned staging blocks (ANOFOX_HIP_PINNED_CACHE_GB, default 2), stream / event sets and up to 32 parked
 * single-series batches for re-use (a batch of the M5 shape is ~300 hipMalloc calls = 0.4 s without them).  This gives all of
 * it back: call it when the host wants the memory (another allocator in the process is short of HBM) or before unloading.
 * Safe while other threads run batches: only idle resources are touched.
 */
void anofox_hip_release_caches(void);

/* ------------------------------------------------------------------------- */
/* Block 3: device-resident batch (series block already in HBM)               */
/* ------------------------------------------------------------------------- */

typedef struct AnofoxHipBatch AnofoxHipBatch; /* opaque plan + HBM workspace */

/* Per-run measurements, filled by anofox_hip_batch_run. */
typedef struct AnofoxHipStats {
    uint64_t n_series;
    uint64_t t_max;
    uint64_t n_problems;        /* (series, candidate spec) optimiser problems */
    uint64_t total_passes;      /* streamed passes over a series (sum over problems) */
    uint64_t max_passes;        /* max over series of its summed passes          */
    uint64_t total_evals;       /* objective evaluations (>= passes)             */
    uint64_t algorithmic_bytes; /* sum_s 8*T_s*(P_s+1) + 24*h, SURVEY.md 8(d)    */
    double   fit_kernel_ms;     /* HIP-event time of the dominant (fit) kernel(s) */
    double   total_device_ms;   /* HIP-event time of the whole run on its stream  */
    uint32_t fit_kernel_launches;
    uint32_t y_storage;         /* what the ETS fit of the run streamed (round 6; the field was `reserved`, always 0): 0 the fp64 block,
                                   1 / 2 a float / uint16_t copy of it -- made when EVERY observation of the batch survives that type
                                   exactly (counts), so the fp64 arithmetic sees the same numbers; algorithmic_bytes keeps SURVEY.md's
                                   unit (8 bytes per observation and pass) whatever was streamed */
    uint64_t total_iters;       /* Nelder-Mead iterations summed over the ETS problems (each problem counts from 1: the initial
                                   simplex) -- the passes a one-pass-per-iteration schedule needs; 0 for models without spec slots */
    uint64_t min_pass_bytes;    /* sum over problems of 8*T_s*(iterations + 1 final pass) + 24*h per series: the algorithmic bytes
                                   of that schedule ("if the kernel evaluates k simplex vertices in one pass, that
                                   is one pass"), independent of which driver ran which round                                  */
} AnofoxHipStats;

/*
 * Lane-level efficiency of the Nelder-Mead round kernels of the last run (round 5).  A wave streams a pass as long as ONE of its 64
 * lanes still evaluates a trial point; a lane whose problem has converged, or that has used the round's budget, idles until the wave
 * leaves.  wave_passes = passes streamed by waves, live_lane_passes = lane-passes that evaluated a point of a running problem (the
 * four-lanes-per-problem and one-wave-per-problem drivers count every lane that evaluates a speculative point), so
 * live_lane_passes / (64 * wave_passes) is the share of issued lanes that did work.  Classes: 0 additive-class specs, 1 general-class
 * specs, 2 damped multiplicative-trend specs (b^phi every step).  Per slot: the spec id (error * 15 + trend * 3 + season; trend
 * 0 N, 1 A, 2 Ad, 3 M, 4 Md) and its two counters.  Versioned by size: pass sizeof(AnofoxHipLaneStats) of the header you compiled
 * against; the library writes min(struct_size, its own size) bytes and stores that number in `struct_size` (an older caller never
 * sees bytes it did not allocate -- what anofox_hip_batch_stats, whose struct grew in round 4, cannot promise).
 */
typedef struct AnofoxHipLaneStats {
    uint64_t struct_size;
    uint64_t wave_passes[3];
    uint64_t live_lane_passes[3];
    uint32_t n_slots;
    uint32_t reserved;
    int32_t  slot_spec_id[30];
    uint64_t slot_wave_passes[30];
    uint64_t slot_live_lane_passes[30];
} AnofoxHipLaneStats;

/* Device selection; returns number of visible devices or -1. */
int anofox_hip_device_count(void);
int anofox_hip_set_device(int device);

/*
 * Create a plan for `n_series` series of at most `t_max` observations.
 * Validates the option block exactly like anofox_ts_forecast would (model name,
 * ETS notation, pool, seasonal_period compatibility).  The packed layout is the
 * time-major block Y[t * ld + s] (fp64), ld = n_series rounded up to 64.
 */
bool anofox_hip_batch_create(size_t n_series, size_t t_max,
                             const struct ForecastOptions *options,
                             AnofoxHipBatch **out_batch,
                             struct AnofoxError *out_error);
void anofox_hip_batch_destroy(AnofoxHipBatch *batch);

size_t anofox_hip_batch_ld(const AnofoxHipBatch *batch);
size_t anofox_hip_batch_n_series(const AnofoxHipBatch *batch);

/*
 * ETS(spec) with GIVEN smoothing parameters (BASELINE.json configs[1]: "ETS(A,A,A) fixed smoothing params"): the batch
 * must have been created for model "ETS" with an explicit `ets_model`.  No optimiser runs: every series takes ONE
 * streamed pass (initial states as in the fitted path, then filter + forecast with these parameters).  Parameters are in
 * the model's own terms, as anofox_hip_batch_inspect reports them: 0 < alpha < 1, 0 <= beta <= alpha,
 * 0 <= gamma <= 1 - alpha, 0 < phi <= 1; the ones the spec does not have are ignored.  What the external crate's
 * `ETS::new(spec, m)` + explicit parameters would be for forecast.rs:1357-1367; additive, no reference entry exists.
 */
bool anofox_hip_batch_set_fixed_params(AnofoxHipBatch *batch, double alpha, double beta, double gamma, double phi,
                                       struct AnofoxError *out_error);

/* AutoARIMA estimation method of this batch: ANOFOX_ARIMA_CSS / ANOFOX_ARIMA_CSS_ML (see block 2). */
bool anofox_hip_batch_set_arima_method(AnofoxHipBatch *batch, int method, struct AnofoxError *out_error);

/* Host series -> HBM block (NULL interpolation, imputation.rs:61-114, then pack + H2D). */
bool anofox_hip_batch_pack_host(AnofoxHipBatch *batch,
                                const double *const *values,
                                const uint64_t *const *validity,
                                const size_t *lengths,
                                struct AnofoxError *out_error);

/*
 * Adopt a block that is already in HBM: `d_y` is [t_max x ld] fp64 time-major,
 * `d_len` is int32[n_series].  No copy; the caller keeps both alive.  The block
 * holds no NULLs.  With auto_detect_seasonality and seasonal_period 0 the periods
 * are detected here, on the device, as the host packer has them detected.
 */
bool anofox_hip_batch_set_device_block(AnofoxHipBatch *batch,
                                       const void *d_y, size_t ld,
                                       const void *d_len,
                                       struct AnofoxError *out_error);

/*
 * The seasonal period every series of the packed / adopted block runs with: the caller's, or -- auto_detect_seasonality with
 * seasonal_period 0 -- the lag of the strongest autocorrelation peak (seasonality.rs:323-377 detect_seasonality_first, 1 when
 * there is none), found by detect_period_kernel on the resident block.  False before a block is set.
 */
bool anofox_hip_batch_periods(const AnofoxHipBatch *batch, int32_t *out_periods);

/*
 * Self test of the kernels' reciprocal (csrc/det_math.hpp dm_recip: v_rcp_f64 + two Newton steps + one correction, the division
 * expansion without range scaling and fix-up) against the compiled IEEE division on `n_operands` generated operands covering
 * [2^-1000, 2^1000] in both signs: *out_mismatches = operands whose two quotients differ in any bit (the parity contract needs 0:
 * oracle/ets.c divides), *out_first_bad (may be NULL) one such operand.  False when no device could run it.
 */
bool anofox_hip_selftest_recip(uint64_t n_operands, uint64_t seed, uint64_t *out_mismatches, double *out_first_bad);

/* Lane-level efficiency counters of the last run (waits for it); false before a run. */
bool anofox_hip_batch_lane_stats(AnofoxHipBatch *batch, AnofoxHipLaneStats *out, size_t struct_size);

/* Asynchronous fit + forecast, ordered on `stream` (a hipStream_t).  NULL is NOT the null stream: it means the batch's own non-blocking stream, which the null stream does not wait for -- wait for the batch (anofox_hip_batch_stats / _fetch, or a device-wide wait) before reading its results from another stream. */
bool anofox_hip_batch_run(AnofoxHipBatch *batch, void *stream,
                          struct AnofoxError *out_error);

/*
 * Run several batches side by side, one host thread each -- the device-resident counterpart of the multi-device batch
 * entry: create one batch per device (anofox_hip_set_device(d) before each create; a batch remembers its device and every
 * entry point makes it current for its own duration), adopt each device's block, run them together.  out_errors may be
 * NULL; returns false if any run failed.
 */
bool anofox_hip_batch_run_many(AnofoxHipBatch *const *batches, size_t n_batches, struct AnofoxError *out_errors);

/* Waits for the run, then fills `out_stats`. */
bool anofox_hip_batch_stats(AnofoxHipBatch *batch, AnofoxHipStats *out_stats);

/*
 * Device result pointers (valid until destroy): yhat/lower/upper are
 * [n_series x horizon] fp64 row-major; model_code int32[n_series] (see
 * anofox_hip_model_name); status int32[n_series] (ErrorCode per series).
 */
bool anofox_hip_batch_device_results(AnofoxHipBatch *batch,
                                     void **d_yhat, void **d_lower, void **d_upper,
                                     void **d_model_code, void **d_status);

/* D2H + per-series malloc'd results with reference ownership rules. */
bool anofox_hip_batch_fetch(AnofoxHipBatch *batch,
                            struct ForecastResult *out_results,
                            struct AnofoxError *out_errors);

/*
 * Fit-state snapshot of a batch that has been run (what ts_forecast_inspect_by /
 * ts_forecast_explain_by read out of a fitted model, forecast.rs:1739-1885, 1899-2017).  For AutoETS and ETS(spec):
 * the smoothing parameters in the model's own terms (beta = alpha beta*, gamma = gamma* (1 - alpha); NaN where the
 * spec has no such component), AIC / AICc / BIC, SSE, the final level and growth, optionally the final seasonal states
 * by phase (`seasonal`, row stride `seasonal_stride` >= period) and the one-step fitted values (`fitted`, [n_series x
 * t_max] row-major, NaN past a series' length).  A second streamed pass per candidate spec on the device produces
 * them; series that took the fallback chain, and every other model, report NaN (AutoARIMA reports its AICc; its orders
 * are in model_code).  The batch must have one seasonal period.
 */
typedef struct AnofoxHipInspection {
    int32_t model_code;      /* as in anofox_hip_batch_device_results */
    int32_t status;          /* ErrorCode of the series */
    int32_t seasonal_period;
    int32_t reserved;
    double alpha, beta, gamma, phi;
    double aic, aicc, bic, sse;
    double level, trend;     /* final states l_T, b_T */
} AnofoxHipInspection;
bool anofox_hip_batch_inspect(AnofoxHipBatch *batch, AnofoxHipInspection *out,
                              double *fitted, double *seasonal, size_t seasonal_stride,
                              struct AnofoxError *out_error);

/* Render a device model_code to the reference's model_name text (<= 63 chars). */
void anofox_hip_model_name(const struct ForecastOptions *options, int32_t model_code,
                           char out_name[64]);
/*
 * The same for a series of a batch: an AutoARIMA code is rendered with the period THAT SERIES was fitted with -- the given one,
 * or the one detect_period_kernel found on a resident block (anofox_hip_batch_periods) -- i.e. exactly the name
 * anofox_hip_batch_fetch writes ("AutoARIMA(p,d,q)(P,D,Q)[m]", forecast.rs:1469-1493).  anofox_hip_model_name only has the option
 * block, whose seasonal_period is 0 when periods are detected.
 */
void anofox_hip_batch_model_name(const AnofoxHipBatch *batch, size_t series, int32_t model_code, char out_name[64]);

/* ------------------------------------------------------------------------- */
/* Block 4: columnar ingest for the table-in-out caller (route B)              */
/* ------------------------------------------------------------------------- */
/*
 * Replaces the collection loop of _ts_forecast_native
 * (src/table_functions/ts_forecast_native.cpp:476-610: per-row GetValue into a
 * std::map<string, GroupData> under a mutex, per-group sort at finalize).
 * The binding appends each DataChunk as plain columns; `group_key` is the
 * dictionary id (or hash) of the group value -- the binding keeps id -> Value.
 * Rules kept: rows with a NULL date are dropped (:505); a NULL target is an
 * invalid slot (interpolated by the packer, imputation.rs:61-114); groups come
 * out in first-appearance order (:586); rows of a group are stably sorted by date.
 * append is thread-safe; validity bitmasks use DuckDB's layout (bit i%64 of word
 * i/64, 1 = valid) and may be NULL (= all valid).
 */
typedef struct AnofoxHipIngest AnofoxHipIngest;
AnofoxHipIngest *anofox_hip_ingest_create(void);
void anofox_hip_ingest_destroy(AnofoxHipIngest *ingest);
bool anofox_hip_ingest_append(AnofoxHipIngest *ingest,
                              const int64_t *group_key, const int64_t *date,
                              const uint64_t *date_valid,
                              const double *value, const uint64_t *value_valid,
                              size_t n_rows, struct AnofoxError *out_error);
bool anofox_hip_ingest_finish(AnofoxHipIngest *ingest, size_t *n_groups, size_t *t_max,
                              struct AnofoxError *out_error);
/* Valid after finish, owned by the ingest: [n_groups] each. */
const int64_t *anofox_hip_ingest_group_keys(const AnofoxHipIngest *ingest);
const int64_t *anofox_hip_ingest_last_dates(const AnofoxHipIngest *ingest);
const size_t *anofox_hip_ingest_lengths(const AnofoxHipIngest *ingest);
const double *const *anofox_hip_ingest_values(const AnofoxHipIngest *ingest);
const uint64_t *const *anofox_hip_ingest_validity(const AnofoxHipIngest *ingest);
/* Series of a finished ingest -> the batch's HBM block (batch created with n_groups, t_max). */
bool anofox_hip_batch_pack_ingest(AnofoxHipBatch *batch, const AnofoxHipIngest *ingest,
                                  struct AnofoxError *out_error);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* ANOFOX_FCST_HIP_H */
