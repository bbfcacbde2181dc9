// host_api.hip -- the C-ABI host layer of the MI355X backend (include/anofox_fcst_hip.h).
//
// Replaces crates/anofox-fcst-ffi (lib.rs:3344-3550 anofox_ts_forecast, :5900-5930 free) and the
// wrapper decisions of crates/anofox-fcst-core/src/forecast.rs:512-733.  One batch = one
// time-major fp64 block in HBM; the per-series fit/forecast arithmetic runs ONLY on the GPU
// (kernels.hip, fit_*.hip).  There is no CPU fallback: without a HIP device every entry point
// fails with INTERNAL_ERROR.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <unordered_map>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "classic_device.hpp"
#include "host_semantics.hpp"
#include "kernels.hpp"

using namespace anofox;

namespace {

constexpr int TINY_BATCH_PROBLEMS = 1024;   // (series x specs) up to which every problem runs on a wave of its own in one launch

#include "host_resources.hpp"     // HipFail / HIPCHECK, the tunables, DeviceGuard, the device / pinned / stream caches, the device list

struct Plan {
    ModelType model;
    std::string ets_notation;   // explicit ETS spec ("" = none)
    int ets_spec_id = -1;
    int pool = 0;
    std::string static_name;    // model_name when model_code == 0
    double z = 1.645;
};

thread_local unsigned tl_host_thread_share = 1;   // > 1 while this thread runs one of several device shards of a batch call (the packer takes its share of the host threads)
std::atomic<int> g_default_arima_method{0};      // ANOFOX_ARIMA_CSS (anofox_hip_set_default_arima_method)
// The estimation method a CALL runs under is fixed when the call enters the library and handed to every host thread that works for
// it (shard, part and merged-batch workers; the leader of a coalesced group runs under the method its members were matched on) --
// a concurrent anofox_hip_set_default_arima_method never changes a call half way.  -1: no call in force, the process default.
std::atomic<int> g_arima_runs{0};       // AutoARIMA searches in flight (launch_arima shortens its launches when it shares the device)
thread_local int tl_arima_method = -1;
static int arima_method_in_force() { return tl_arima_method >= 0 ? tl_arima_method : g_default_arima_method.load(); }
struct ArimaMethodScope {
    int saved;
    explicit ArimaMethodScope(int m) : saved(tl_arima_method) { tl_arima_method = m; }
    ~ArimaMethodScope() { tl_arima_method = saved; }
};

} // namespace

struct AnofoxHipBatch {
    int dev = 0;                       // the device the batch lives on (the caller's current device when it was created): every
                                       // entry point makes it current for its own duration, so a host thread may drive batches of
                                       // several devices
    size_t n = 0, t_max = 0, ld = 0;
    int h = 0;
    ForecastOptions opt;
    Plan plan;
    bool has_block = false, owns_y = false;
    double *d_y = nullptr;
    int32_t *d_len = nullptr;          // lengths with unusable series zeroed
    bool owns_len = false;
    std::vector<int32_t> h_len;        // true lengths
    bool quiesced = true;              // nothing of this batch is in flight (set by the fetch's wait, cleared by a run): destroy need not wait
    std::vector<int32_t> h_period;     // per-series period
    std::vector<int32_t> h_base_status;
    std::vector<int32_t> h_slot_spec;
    double *h_stage = nullptr;         // pinned staging copy of the time-major block (packer output, H2D source)
    size_t h_stage_elems = 0;
    std::vector<double> h_clean;       // interpolated host copy (only when fitted values requested)
    std::vector<size_t> h_clean_off;
    // prep outputs
    double *d_mean = nullptr, *d_sd = nullptr, *d_fig_add = nullptr, *d_fig_mul = nullptr, *d_l0 = nullptr, *d_b0 = nullptr;
    uint32_t *d_flags = nullptr;
    int fig_m = 0;
    // spec slots
    int n_slots_cap = 0;
    double *d_aicc = nullptr, *d_yhat_slots = nullptr;
    int32_t *d_status_slots = nullptr, *d_evals_slots = nullptr, *d_iters_slots = nullptr, *d_passes_slots = nullptr, *d_slot_spec = nullptr;
    // compact copies of the series block for the ETS round / final kernels (round 6, ets_device.hpp YT_*): a float block and a uint16_t
    // block of t_max x ld cells each, remade by every run that fits (the caller may have rewritten a resident block), and the count of
    // observations that do not survive either round trip.  y_type: what THIS run's fit streams (YT_F64 unless a counter is zero).
    float *d_yc32 = nullptr;
    unsigned short *d_yc16 = nullptr;
    size_t yc_cells = 0;
    unsigned int *d_misfit = nullptr;
    unsigned int h_misfit[2] = {1u, 1u};
    int y_type = 0;
    unsigned long long *d_wave_trace = nullptr;    // developer instrument (tune wave_trace): 4 + 4 x WAVE_TRACE_RECORDS words, see FitArgs::wave_trace
    unsigned long long *d_lane_stats = nullptr;    // [n_slots_cap x 2] wave passes / live lane passes of every spec's round kernels (zeroed per run)
    // outputs
    double *d_yhat = nullptr, *d_lo = nullptr, *d_hi = nullptr;
    int32_t *d_model_code = nullptr, *d_status = nullptr, *d_detail = nullptr, *d_passes_total = nullptr, *d_evals_total = nullptr;
    uint32_t *d_mask = nullptr;
    int32_t *d_len_group = nullptr;
    int32_t *d_count = nullptr;
    // strictly positive series of the current group, dense: round 0 of the specs with a multiplicative component runs on it
    int32_t *d_pos_map = nullptr, *d_pos_cnt = nullptr, *d_notpos = nullptr;
    double *d_ypos = nullptr;
    bool use_pos = false;
    bool none_pos = false;             // the group has no strictly positive series: multiplicative specs are retired without a launch chain
    // what the inspection pass needs from the last fit: the final-kernel arguments of every spec, in launch order
    std::vector<anofox::FitArgs> insp_args;
    std::vector<anofox::FitLaunchers> insp_fns;
    std::vector<int> insp_spec, insp_stream;
    bool insp_ok = false;
    int insp_m = 1;
    int32_t live_pos = -1, live_all = -1;   // usable strictly positive / usable series of the current group (-1: not counted)
    // AutoARIMA workspace
    int arima_method = 0;            // ANOFOX_ARIMA_CSS: the search's own estimates; ANOFOX_ARIMA_CSS_ML: exact-likelihood refit of the
                                     // selected model (anofox_hip_batch_set_arima_method / anofox_hip_set_default_arima_method)
    size_t ar_ws_bytes = 0;
    double *ar_w = nullptr, *ar_wmean = nullptr, *ar_wsd = nullptr, *ar_l0 = nullptr, *ar_l1 = nullptr, *ar_x = nullptr, *ar_aicc = nullptr;
    int32_t *ar_wlen = nullptr, *ar_d = nullptr, *ar_D = nullptr, *ar_order = nullptr, *ar_status = nullptr, *ar_evals = nullptr, *ar_passes = nullptr, *ar_models = nullptr;
    // streams / events (borrowed from the process-wide pool: `sset`)
    StreamSet *sset = nullptr;
    hipStream_t own_stream = nullptr;
    hipStream_t aux[N_AUX_STREAMS] = {};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_fit0 = nullptr, ev_fit1 = nullptr, ev_fork = nullptr;
    hipEvent_t ev_join[N_AUX_STREAMS] = {};
    // per aux stream: gathered block of the running problems, ping-pong column maps + counts, parked NM state
    struct Lane {
        double *ybuf = nullptr;          // allocated when a run first needs it (launch_fit_slots: only the specs that have anything to fit)
        size_t ybuf_cols = 0;            // columns it holds (<= ld)
        int32_t *map[2] = {nullptr, nullptr};
        int32_t *cnt = nullptr;          // [2]
        anofox::NmStateBuf st{};
        double *nm_scratch = nullptr;    // where the lanes' simplices rest between passes (kernels that keep them out of LDS: RoundTraits::PARK)
        size_t nm_scratch_doubles = 0;
    } lanes[N_AUX_STREAMS];
    hipStream_t last_stream = nullptr;
    bool ran = false, timed_fit = false;
    uint32_t fit_launches = 0;
    uint64_t n_problems = 0;
    Tunables tun;            // the environment knobs, read when the batch was created
    int seq_rounds = 3;      // rounds run by the sequential Nelder-Mead driver before switching to the speculative one
    int seq_rounds_env = -1; // tune seq_rounds override (-1 = decide from the number of live problems)
    int spec_below_md = 8192; // same, for the damped multiplicative-trend specs (their pass is ~10x longer: the stragglers matter more)
    // the SES / Holt / Holt-Winters / SeasonalES family on the round kernels: its own Nelder-Mead state, status and compaction lists
    // (the spec lanes keep the optima an inspection call re-reads)
    anofox::NmStateBuf classic_st{};
    int32_t *d_classic_status = nullptr, *classic_map[2] = {nullptr, nullptr}, *classic_cnt = nullptr;
    double *d_classic_ybuf = nullptr;
    size_t classic_ybuf_cols = 0;
    std::vector<void *> retired;  // blocks a run outgrew while kernels may still read them: handed back when the batch is destroyed
                                  // (freeing them on the spot needs a device-wide synchronisation, which couples every host thread's batch)
    int merged_m_max = 0;         // ... and its largest period (sizes)
    int32_t *d_m_col = nullptr;   // merged batch of several seasonal periods (auto-detected): period of every column, constant within 64 columns
    int spec2_below_md = 2048; // per spec: the last problems run one per wave, two iterations per pass (0 = never); damped multiplicative
                               // trend.  Measured on the 30-spec M5 batch (tools/spec2_sweep.sh): 0 / 256 / 1024 / 2048 / 4096 / 8192 -> 579 / 575 / 565 / 562 / 594 / 736 ms
    int spec2_below = 1024;    // same, other specs (single-spec ETS(A,A,A) fit: 22.5 -> 18.2 ms; all specs at 2048 / 4096: 588 / 673 ms)
    int spec_below = 8192;   // per spec: switch to the speculative driver once this few problems are still running
    bool use_gather = false; // rebuild a dense block of the running problems between rounds (else index y by series)
    double gather_budget = 0.0;  // bytes the gather blocks of one run may take together (55 % of the device): a block holds ld columns, or
                                 // fewer when the live specs' blocks would not fit (the gather then starts once that few problems still
                                 // run; until then the rounds index y by series)
    bool sd_in_final = false;        // this run's one-spec ETS batch: the final pass computes d_sd (prep_kernel ran one sweep)
    double *d_ring = nullptr;        // seasonal rings of periods above the LDS limit: one area per (candidate spec, workgroup)
    size_t ring_elems = 0;
    double *d_prep_scratch = nullptr;   // prep kernel's window ring + per-phase accumulators for such periods
    size_t prep_scratch_elems = 0;
    // BASELINE config 2: ETS(spec) with GIVEN smoothing parameters -- no optimiser, one streamed pass per series
    bool fixed_params = false;
    double fixed_x[4] = {0.0, 0.0, 0.0, 0.0};   // optimiser coordinates (alpha, beta*, gamma*, phi) of the given parameters
};

namespace {

// ---------------------------------------------------------------------------------------------
// option validation shared by every entry point (forecast.rs:512-565, 1278-1314, 1524-1556)
// ---------------------------------------------------------------------------------------------
bool make_plan(const ForecastOptions *o, Plan &p, AnofoxError *err)
{
    std::string mname = cstr_field(o->model, sizeof o->model);
    if (!parse_model(mname, p.model)) {
        set_error(err, INVALID_MODEL, "Invalid model: Unknown model: '" + mname + "'");
        return false;
    }
    if (o->horizon < 0) { set_error(err, PANIC_CAUGHT, "Panic in Rust code"); return false; }
    if (!o->auto_detect_seasonality && o->seasonal_period > 1 && is_non_seasonal_model(p.model)) {
        char buf[512];
        std::snprintf(buf, sizeof buf,
                      "Invalid input: Model '%s' does not use seasonal_period (got %d). For seasonal forecasting, use: "
                      "SeasonalNaive, HoltWinters, SeasonalES, AutoETS, AutoMFLES, AutoMSTL, or AutoTBATS.",
                      model_name(p.model), o->seasonal_period);
        set_error(err, INVALID_INPUT, buf);
        return false;
    }
    p.static_name = model_name(p.model);
    p.z = z_for_confidence(o->confidence_level);
    switch (p.model) {
    case M_Naive: case M_SeasonalNaive: case M_SMA: case M_RandomWalkDrift: case M_ARIMA:
    case M_SES: case M_SESOptimized: case M_Holt: case M_HoltWinters: case M_SeasonalES: case M_SeasonalESOptimized:
        break;
    case M_ETS: {
        p.ets_notation = cstr_field(o->ets_model, sizeof o->ets_model);
        if (!p.ets_notation.empty()) {
            const std::string &nt = p.ets_notation;
            if (!valid_ets_notation(nt)) {
                set_error(err, INVALID_INPUT,
                          "Invalid input: Invalid ETS model specification '" + nt +
                              "'. Expected 3 or 4 character notation: Error(A/M) + Trend(A/M/N) + Seasonal(A/M/N), with optional 'd' "
                              "for damped trend. Examples: 'AAA' (additive), 'MNM' (multiplicative error, no trend), 'AAdA' (additive "
                              "damped trend). Valid characters: A=Additive, M=Multiplicative, N=None, d=Damped.");
                return false;
            }
            p.ets_spec_id = spec_id_from_notation(nt);
            if (!spec_is_valid(p.ets_spec_id)) {
                set_error(err, INVALID_INPUT,
                          "Invalid input: ETS model '" + nt +
                              "' is an unstable combination (multiplicative error with additive components). Try one of: 'AAA', 'ANA', "
                              "'AAdA', 'MNM', 'MAM', 'MAdM', 'MMM', 'MMdM', or use 'AutoETS' for automatic selection.");
                return false;
            }
            p.static_name = "ETS(" + nt + ")";
        }
        break;
    }
    case M_AutoARIMA:
        break;
    case M_AutoETS: {
        std::string pool = cstr_field(o->model_pool, sizeof o->model_pool);
        p.pool = 0;
        if (!pool.empty()) {
            p.pool = parse_model_pool(pool);
            if (p.pool < 0) {
                set_error(err, INVALID_INPUT,
                          "Invalid input: Unknown model_pool '" + pool +
                              "'. Valid options: complete, no_multiplicative_trend, damped_trend_only, match_error_seasonal, reduced");
                return false;
            }
        }
        break;
    }
    default:
        set_error(err, INTERNAL_ERROR,
                  std::string("Internal error: model '") + model_name(p.model) + "' is not implemented by the HIP backend");
        return false;
    }
    return true;
}

// imputation.rs:61-114
void fill_nulls_interpolate(const double *values, const uint64_t *validity, size_t n, double *out)
{
    auto valid = [&](size_t i) { return validity == nullptr || ((validity[i / 64] >> (i % 64)) & 1ull); };
    long first = -1, last = -1;
    for (size_t i = 0; i < n; i++)
        if (valid(i)) { if (first < 0) first = (long)i; last = (long)i; }
    for (size_t i = 0; i < n; i++) out[i] = std::nan("");
    if (first < 0) return;
    for (long i = 0; i < first; i++) out[i] = values[first];
    for (size_t i = (size_t)last + 1; i < n; i++) out[i] = values[last];
    long prev = first;
    double pv = values[first];
    out[first] = pv;
    for (long i = first + 1; i <= last; i++) {
        if (valid((size_t)i)) {
            double v = values[i];
            long gap = i - prev;
            if (gap > 1) {
                double slope = (v - pv) / (double)gap;
                for (long j = 1; j < gap; j++) out[prev + j] = pv + slope * (double)j;
            }
            out[i] = v;
            prev = i;
            pv = v;
        }
    }
}

// a batch borrows its streams and events from the process-wide pool (and hands them back while it is parked, see pool_give)
void batch_attach_streams(AnofoxHipBatch *b)
{
    if (b->sset) return;
    b->sset = stream_set_take();
    b->own_stream = b->sset->own;
    for (int i = 0; i < N_AUX_STREAMS; i++) { b->aux[i] = b->sset->aux[i]; b->ev_join[i] = b->sset->ev_join[i]; }
    b->ev_start = b->sset->ev_start; b->ev_stop = b->sset->ev_stop; b->ev_fit0 = b->sset->ev_fit0; b->ev_fit1 = b->sset->ev_fit1;
    b->ev_fork = b->sset->ev_fork;
}
void batch_detach_streams(AnofoxHipBatch *b)           // the caller has synchronised the set's streams (or nothing was ever launched on them)
{
    if (!b->sset) return;
    stream_set_give(b->sset);
    b->sset = nullptr; b->own_stream = nullptr; b->last_stream = nullptr; b->ran = false;
    for (auto &q : b->aux) q = nullptr;
}

void free_batch_buffers(AnofoxHipBatch *b)
{
    auto F = [](void *p) { dev_free(p, true); };             // the caller has synchronised the batch's streams
    if (b->owns_y) F(b->d_y);
    pin_free(b->h_stage);
    if (b->owns_len) F(b->d_len);
    F(b->d_mean); F(b->d_sd); F(b->d_fig_add); F(b->d_fig_mul); F(b->d_l0); F(b->d_b0); F(b->d_flags);
    F(b->d_aicc); F(b->d_yhat_slots); F(b->d_status_slots); F(b->d_evals_slots); F(b->d_iters_slots);
    F(b->d_passes_slots); F(b->d_slot_spec); F(b->d_lane_stats); F(b->d_wave_trace); F(b->d_yc32); F(b->d_yc16); F(b->d_misfit);
    F(b->d_yhat); F(b->d_lo); F(b->d_hi); F(b->d_model_code); F(b->d_status); F(b->d_detail);
    F(b->classic_st.sim); F(b->classic_st.fs); F(b->classic_st.phase); F(b->classic_st.evals); F(b->classic_st.iters); F(b->classic_st.passes); F(b->classic_st.done);
    F(b->d_classic_ybuf); F(b->d_classic_status); F(b->classic_map[0]); F(b->classic_map[1]); F(b->classic_cnt);
    for (void *p : b->retired) F(p);
    b->retired.clear();
    F(b->d_m_col); F(b->d_ring); F(b->d_prep_scratch);
    F(b->d_passes_total); F(b->d_evals_total); F(b->d_mask); F(b->d_len_group); F(b->d_count); F(b->d_pos_map); F(b->d_pos_cnt); F(b->d_notpos); F(b->d_ypos);
    F(b->ar_w); F(b->ar_wmean); F(b->ar_wsd); F(b->ar_l0); F(b->ar_l1); F(b->ar_x); F(b->ar_aicc); F(b->ar_wlen); F(b->ar_d); F(b->ar_D);
    F(b->ar_order); F(b->ar_status); F(b->ar_evals); F(b->ar_passes); F(b->ar_models);
    batch_detach_streams(b);
    for (auto &l : b->lanes) {
        F(l.ybuf); l.ybuf_cols = 0; F(l.map[0]); F(l.map[1]); F(l.cnt);
        F(l.st.sim); F(l.st.fs); F(l.st.phase); F(l.st.evals); F(l.st.iters); F(l.st.passes); F(l.st.done);
        F(l.nm_scratch); l.nm_scratch_doubles = 0;
    }
}

int max_slots_for(const Plan &p)
{
    if (p.model == M_AutoETS) return 30;
    if (p.model == M_ETS && p.ets_spec_id >= 0) return 1;
    return 0;
}

void alloc_common(AnofoxHipBatch *b)
{
    const size_t n = b->n, ld = b->ld, h = (size_t)std::max(b->h, 0);
    b->d_mean = dalloc<double>(ld);
    b->d_sd = dalloc<double>(ld);
    b->d_flags = dalloc<uint32_t>(ld);
    b->d_yhat = dalloc<double>(n * h);
    b->d_lo = dalloc<double>(n * h);
    b->d_hi = dalloc<double>(n * h);
    b->d_model_code = dalloc<int32_t>(ld);
    b->d_status = dalloc<int32_t>(ld);
    b->d_detail = dalloc<int32_t>(ld);
    b->d_passes_total = dalloc<int32_t>(ld);
    b->d_evals_total = dalloc<int32_t>(ld);
    b->d_mask = dalloc<uint32_t>(ld);
    b->d_len_group = dalloc<int32_t>(ld);
    b->d_count = dalloc<int32_t>(2);
    if (b->plan.model == M_AutoETS) {
        b->d_pos_map = dalloc<int32_t>(ld); b->d_pos_cnt = dalloc<int32_t>(2); b->d_notpos = dalloc<int32_t>(ld);
    }
    if (b->plan.model == M_AutoARIMA) {
        const size_t T = std::max<size_t>(b->t_max, 1);
        b->ar_ws_bytes = arima_workspace_bytes((int)b->n, (int)T);
        b->ar_w = (double *)dalloc<char>(b->ar_ws_bytes);     // search workspace: differenced rows, candidate cache, queues
        HIPCHECK(hipMemset(b->ar_w, 0, b->ar_ws_bytes));
        b->ar_wmean = dalloc<double>(ld); b->ar_wsd = dalloc<double>(ld); b->ar_l0 = dalloc<double>(ld); b->ar_l1 = dalloc<double>(ld);
        b->ar_x = dalloc<double>(6 * ld); b->ar_aicc = dalloc<double>(ld);
        b->ar_wlen = dalloc<int32_t>(ld); b->ar_d = dalloc<int32_t>(ld); b->ar_D = dalloc<int32_t>(ld); b->ar_order = dalloc<int32_t>(5 * ld);
        b->ar_status = dalloc<int32_t>(ld); b->ar_evals = dalloc<int32_t>(ld); b->ar_passes = dalloc<int32_t>(ld); b->ar_models = dalloc<int32_t>(ld);
        HIPCHECK(hipMemset(b->ar_wlen, 0, ld * sizeof(int32_t)));
    }
    HIPCHECK(hipMemset(b->d_model_code, 0, ld * sizeof(int32_t)));
    HIPCHECK(hipMemset(b->d_detail, 0, ld * sizeof(int32_t)));
    HIPCHECK(hipMemset(b->d_passes_total, 0, ld * sizeof(int32_t)));
    HIPCHECK(hipMemset(b->d_evals_total, 0, ld * sizeof(int32_t)));
    HIPCHECK(hipMemset(b->d_mask, 0, ld * sizeof(uint32_t)));
    b->n_slots_cap = max_slots_for(b->plan);
    if (b->n_slots_cap > 0) {
        const size_t S = (size_t)b->n_slots_cap;
        b->d_l0 = dalloc<double>(9 * ld);
        b->d_b0 = dalloc<double>(9 * ld);
        b->d_aicc = dalloc<double>(S * ld);
        b->d_yhat_slots = dalloc<double>(S * n * h);
        b->d_status_slots = dalloc<int32_t>(S * ld);
        b->d_evals_slots = dalloc<int32_t>(S * ld);
        b->d_iters_slots = dalloc<int32_t>(S * ld);
        b->d_passes_slots = dalloc<int32_t>(S * ld);
        b->d_slot_spec = dalloc<int32_t>(S);
        b->d_lane_stats = dalloc<unsigned long long>(2 * S);
        HIPCHECK(hipMemset(b->d_lane_stats, 0, 2 * S * sizeof(unsigned long long)));
        const int n_lanes = std::min<int>(N_AUX_STREAMS, b->n_slots_cap);
        for (int q = 0; q < n_lanes; q++) {
            auto &l = b->lanes[q];
            l.map[0] = dalloc<int32_t>(ld);
            l.map[1] = dalloc<int32_t>(ld);
            l.cnt = dalloc<int32_t>(3);
            l.st.sim = dalloc<double>(20 * ld);
            l.st.fs = dalloc<double>(5 * ld);
            l.st.phase = dalloc<int32_t>(ld);
            l.st.evals = dalloc<int32_t>(ld);
            l.st.iters = dalloc<int32_t>(ld);
            l.st.passes = dalloc<int32_t>(ld);
            l.st.done = dalloc<int32_t>(ld);
        }
    }
    batch_attach_streams(b);
}

void ensure_fig(AnofoxHipBatch *b, int m)
{
    if (m <= b->fig_m) return;
    if (b->d_fig_add) b->retired.push_back(b->d_fig_add);
    if (b->d_fig_mul) b->retired.push_back(b->d_fig_mul);
    b->d_fig_add = nullptr; b->d_fig_mul = nullptr; b->fig_m = 0;      // (an allocation below may throw: the retired blocks must not be freed twice)
    b->d_fig_add = dalloc<double>((size_t)m * b->ld);
    b->d_fig_mul = dalloc<double>((size_t)m * b->ld);
    b->fig_m = m;
}

// HBM scratch for seasonal periods whose rings do not fit LDS (ETS_LDS_PERIOD): grown on demand, kept for the batch
double *ensure_ring(AnofoxHipBatch *b, size_t elems)
{
    if (b->ring_elems < elems) {
        if (b->d_ring) b->retired.push_back(b->d_ring);
        b->d_ring = nullptr; b->ring_elems = 0;
        b->d_ring = dalloc<double>(elems);
        b->ring_elems = elems;
    }
    return b->d_ring;
}
double *ensure_prep_scratch(AnofoxHipBatch *b, size_t elems)
{
    if (b->prep_scratch_elems < elems) {
        if (b->d_prep_scratch) b->retired.push_back(b->d_prep_scratch);
        b->d_prep_scratch = nullptr; b->prep_scratch_elems = 0;
        b->d_prep_scratch = dalloc<double>(elems);
        b->prep_scratch_elems = elems;
    }
    return b->d_prep_scratch;
}

// Seasonal periods of the columns of a resident time-major block (kernels.hip detect_period_kernel): 0 = no autocorrelation peak.
void detect_periods_block(const double *d_y, size_t ld, const int32_t *d_len, size_t n, size_t t_rows, int32_t *h_out, hipStream_t st)
{
    if (n == 0) return;
    int32_t *d_per = dalloc<int32_t>(n);
    const size_t sc = detect_scratch_doubles((int)n, (int)t_rows);
    double *d_sc = sc ? dalloc<double>(sc) : nullptr;
    try {
        launch_detect_periods(d_y, ld, d_len, (int)n, (int)t_rows, d_sc, d_per, nullptr, st);
        LAUNCHCHECK("seasonal period detection");
        HIPCHECK(hipMemcpyAsync(h_out, d_per, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
    } catch (...) { dev_free(d_per); dev_free(d_sc); throw; }
    dev_free(d_per, true);
    dev_free(d_sc, true);
}

// ... of host series (the batch entry with params := MAP{}, before the batch is split by period): packed 64 series per tile into
// pinned blocks of ~128 MB, two in flight, so that the packer threads work on one block while the other is copied and scanned
std::vector<int> detect_periods_host_series(const double *const *values, const uint64_t *const *validity, const size_t *lengths, size_t n_series)
{
    std::vector<int> out(n_series, 0);
    size_t t_all = 0;
    for (size_t s = 0; s < n_series; s++) t_all = std::max(t_all, lengths[s]);
    if (t_all < 4 || n_series == 0) return out;
    const size_t n_pad = (n_series + 63) / 64 * 64;
    const size_t cols = std::min(n_pad, std::max<size_t>(64, (size_t)(134217728.0 / (8.0 * (double)t_all)) / 64 * 64));
    struct Buf { double *h = nullptr, *d = nullptr, *sc = nullptr; int32_t *h_len = nullptr, *d_len = nullptr, *h_per = nullptr, *d_per = nullptr;
                 size_t s0 = 0, cnt = 0; bool busy = false; hipEvent_t done = nullptr; } buf[2];
    hipStream_t st = nullptr;
    auto release = [&]() {
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (auto &b : buf) {
            pin_free(b.h); pin_free(b.h_len); dev_free(b.d, true); dev_free(b.sc, true); dev_free(b.d_len, true); dev_free(b.d_per, true);
            if (b.done) (void)hipEventDestroy(b.done);
        }
    };
    try {
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        const size_t n_chunks = (n_series + cols - 1) / cols;
        for (int k = 0; k < (n_chunks > 1 ? 2 : 1); k++) {
            HIPCHECK(hipEventCreateWithFlags(&buf[k].done, hipEventDisableTiming));
            buf[k].h = (double *)pin_alloc_bytes(t_all * cols * sizeof(double));
            buf[k].h_len = (int32_t *)pin_alloc_bytes(2 * cols * sizeof(int32_t));
            buf[k].h_per = buf[k].h_len + cols;
            buf[k].d = dalloc<double>(t_all * cols);
            buf[k].d_len = dalloc<int32_t>(cols);
            buf[k].d_per = dalloc<int32_t>(cols);
            const size_t sc = detect_scratch_doubles((int)cols, (int)t_all);
            if (sc) buf[k].sc = dalloc<double>(sc);
        }
        auto collect = [&](Buf &b) {
            if (!b.busy) return;
            HIPCHECK(hipEventSynchronize(b.done));                // this block's copy back (the other block may still be in flight)
            for (size_t j = 0; j < b.cnt; j++) out[b.s0 + j] = b.h_per[j];
            b.busy = false;
        };
        unsigned n_thr = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
        if (tl_host_thread_share > 1) n_thr = std::max(1u, n_thr / tl_host_thread_share);
        for (size_t ck = 0; ck < n_chunks; ck++) {
            Buf &b = buf[ck & 1];
            collect(b);
            b.s0 = ck * cols;
            b.cnt = std::min(cols, n_series - b.s0);
            size_t T = 0;
            for (size_t j = 0; j < b.cnt; j++) T = std::max(T, lengths[b.s0 + j]);
            T = std::max<size_t>(T, 1);
            const size_t ld = (b.cnt + 63) / 64 * 64, n_tiles = ld / 64;
            auto do_tiles = [&](size_t tile0, size_t tile1) {
                std::vector<double> clean(64 * T);
                for (size_t tile = tile0; tile < tile1; tile++) {
                    size_t len_of[64];
                    for (size_t j = 0; j < 64; j++) {
                        const size_t c = tile * 64 + j;
                        const size_t len = c < b.cnt ? lengths[b.s0 + c] : 0;
                        len_of[j] = len;
                        if (len) fill_nulls_interpolate(values[b.s0 + c], validity ? validity[b.s0 + c] : nullptr, len, clean.data() + j * T);
                        b.h_len[c] = len >= 3 ? (int32_t)len : 0;
                    }
                    for (size_t t = 0; t < T; t++) {
                        double *row = b.h + t * ld + tile * 64;
                        for (size_t j = 0; j < 64; j++) row[j] = t < len_of[j] ? clean[j * T + t] : 0.0;
                    }
                }
            };
            const unsigned thr = (unsigned)std::min<size_t>(n_thr, n_tiles);
            if (thr <= 1) do_tiles(0, n_tiles);
            else {
                std::atomic<bool> failed{false};
                parallel_shares(thr, [&](unsigned k) { try { do_tiles(n_tiles * k / thr, n_tiles * (k + 1) / thr); } catch (...) { failed = true; } });
                if (failed) throw HipFail{"period detection: packer thread failed (out of host memory)", true};
            }
            HIPCHECK(hipMemcpyAsync(b.d, b.h, T * ld * sizeof(double), hipMemcpyHostToDevice, st));
            HIPCHECK(hipMemcpyAsync(b.d_len, b.h_len, ld * sizeof(int32_t), hipMemcpyHostToDevice, st));
            launch_detect_periods(b.d, ld, b.d_len, (int)b.cnt, (int)T, b.sc, b.d_per, nullptr, st);
            LAUNCHCHECK("seasonal period detection");
            HIPCHECK(hipMemcpyAsync(b.h_per, b.d_per, b.cnt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHECK(hipEventRecord(b.done, st));
            b.busy = true;
        }
        collect(buf[0]);
        collect(buf[1]);
    } catch (...) { release(); throw; }
    release();
    return out;
}

// Series that cannot be forecast at all (forecast.rs:516-525) and per-series periods.
void finalize_lengths(AnofoxHipBatch *b)
{
    const size_t n = b->n;
    b->h_base_status.assign(n, 0);
    std::vector<int32_t> eff(b->ld, 0);
    for (size_t s = 0; s < n; s++) {
        if (b->h_len[s] < 3) b->h_base_status[s] = INSUFFICIENT_DATA;
        else eff[s] = b->h_len[s];
    }
    if (!b->d_len) { b->d_len = dalloc<int32_t>(b->ld); b->owns_len = true; }
    HIPCHECK(hipMemcpy(b->d_len, eff.data(), b->ld * sizeof(int32_t), hipMemcpyHostToDevice));
}

// auto-detected periods of a batch's own block (after finalize_lengths: series shorter than three observations read as empty)
void batch_detect_periods(AnofoxHipBatch *b)
{
    std::vector<int32_t> per(b->n, 0);
    batch_attach_streams(b);
    detect_periods_block(b->d_y, b->ld, b->d_len, b->n, std::max<size_t>(b->t_max, 1), per.data(), b->own_stream);
    b->h_period.assign(b->n, 1);
    for (size_t s = 0; s < b->n; s++) b->h_period[s] = per[s] > 0 ? per[s] : 1;
}

// ---------------------------------------------------------------------------------------------
// the pipeline for one group of series sharing a seasonal period
// ---------------------------------------------------------------------------------------------
void run_classic(AnofoxHipBatch *b, int kind, const int32_t *d_len, int m, int optimized, double alpha, int min_len,
                 const uint32_t *mask, uint32_t want, int32_t code, bool write_code, hipStream_t st)
{
    ClassicArgs a{};
    a.y = b->d_y; a.ld = b->ld; a.len = d_len; a.n_series = (int)b->n;
    a.m = m; a.h = b->h; a.optimized = optimized; a.fixed_alpha = alpha;
    a.mask = mask; a.want = want; a.min_len = min_len;
    a.yhat = b->d_yhat; a.status = b->d_detail; a.passes = b->d_passes_total;
    a.model_code = code; a.model_code_out = write_code ? b->d_model_code : nullptr;
    a.ring_scratch = nullptr;
    a.m_col = (kind == CK_HW || kind == CK_SEASONAL_ES) ? b->d_m_col : nullptr;
    const bool optimise = kind == CK_HOLT || kind == CK_HW || optimized != 0;
    if (optimise) {
        // The optimised members of the family run on the ETS round kernels (ClassicCfg<KIND>, fit_classic.hip): resumable rounds
        // with compaction, the drivers picked from the device-side count, then one final pass.  One chain on `st`.
        const size_t n = b->n, ld = b->ld;
        if (!b->classic_st.sim) {
            b->classic_st.sim = dalloc<double>(12 * ld); b->classic_st.fs = dalloc<double>(4 * ld);
            b->classic_st.phase = dalloc<int32_t>(ld); b->classic_st.evals = dalloc<int32_t>(ld); b->classic_st.iters = dalloc<int32_t>(ld);
            b->classic_st.passes = dalloc<int32_t>(ld); b->classic_st.done = dalloc<int32_t>(ld);
            b->d_classic_status = dalloc<int32_t>(ld);
            b->classic_map[0] = dalloc<int32_t>(ld); b->classic_map[1] = dalloc<int32_t>(ld); b->classic_cnt = dalloc<int32_t>(3);
        }
        // dense re-gather between rounds: the spec lanes' block when there is one (free by now), else the family's own
        double *ybuf = (b->n_slots_cap > 0 && b->use_gather) ? b->lanes[0].ybuf : nullptr;
        size_t ybuf_cols = ybuf ? b->lanes[0].ybuf_cols : ld;
        if (!ybuf) {
            // the family's own block: up to 32 GiB of columns (the 1M x 1,024 block is 8.2 GB), fewer if the batch is larger still
            const double per_col = (double)std::max<size_t>(b->t_max, 1) * 8.0;
            const size_t cols = (size_t)std::min((double)ld, 32.0 * 1073741824.0 / per_col) / 64 * 64;
            if (cols >= 1024 || cols >= ld) {
                if (!b->d_classic_ybuf) { b->d_classic_ybuf = dalloc<double>(std::max<size_t>(b->t_max, 1) * cols); b->classic_ybuf_cols = cols; }
                ybuf = b->d_classic_ybuf;
                ybuf_cols = b->classic_ybuf_cols;
            }
        }
        const bool seasonal = kind == CK_HW || kind == CK_SEASONAL_ES;
        FitArgs f{};
        f.y = b->d_y; f.ld = ld; f.len = d_len; f.n_series = (int)n; f.t_rows = (int)std::max<size_t>(b->t_max, 1);
        f.m = seasonal ? std::max(m, 1) : 1; f.h = b->h;
        f.flags = b->d_flags; f.fig = nullptr; f.fig_ld = ld;
        f.status = b->d_classic_status; f.st = b->classic_st;
        f.mask = mask; f.want = want; f.min_len = min_len;
        f.m_col = a.m_col;
        const bool merged = f.m_col != nullptr;
        FitLaunchers fns = classic_fit_launcher(kind, (merged && f.m == 7) ? 8 : f.m);          // (7 has a register-ring variant: one period only)
        if (!fns.round_seq || !fns.round_spec || !fns.round_spec2 || !fns.round_auto) throw HipFail{"no round kernel for this model"};
        if (f.m > ETS_LDS_PERIOD) {
            const size_t wg = merged ? n : std::max<size_t>((n + 15) / 16, std::min<size_t>(n, 1024));
            f.ring_scratch = ensure_ring(b, wg * (size_t)f.m * 64u);
            f.ring_scratch_doubles = wg * (size_t)f.m * 64u;
        }
        static const int BUDGET[] = {24, 24, 24, 24, 48, 48, 96, 192, 1024};
        // a handful of series (one call per group from the scalar binding, the coalesced calls of a few workers): every problem
        // gets a WAVE from the start -- two iterations per pass -- and runs to completion in ONE launch instead of nine rounds of
        // compaction + gather + fit (one series through Holt-Winters: 8.6 ms of launches)
        const bool tiny = n <= TINY_BATCH_PROBLEMS;
        const int n_rounds = tiny ? 1 : (merged ? (n > 4096 ? 4 : 3) : (int)(sizeof BUDGET / sizeof BUDGET[0]));
        for (int r = 0; r < n_rounds; r++) {
            f.first_round = (r == 0);
            f.spec_below = -1; f.spec2_below = -1;
            f.gathered = 0;
            f.y_round = b->d_y; f.ld_round = ld; f.series_of = nullptr; f.n_active = nullptr;
            if (tiny) {
                f.budget = 1 << 30; f.budget_seq = f.budget;
                fns.round_spec2(f, st);
            } else if (merged) {
                // several periods in one block: every launch sweeps all columns in place (see launch_fit_slots)
                f.budget = r == 0 ? 64 : (r == 1 ? 128 : 256); f.budget_seq = f.budget;
                (r < n_rounds - 1 ? fns.round_spec : fns.round_spec2)(f, st);
            } else if (r == 0) {
                const bool seq0 = n >= 4u * 65536u;           // the chip is full with one lane per problem
                f.budget = seq0 ? (BUDGET[0] * 7) / 4 : BUDGET[0];
                (seq0 ? fns.round_seq : fns.round_spec)(f, st);
            } else {
                int32_t **map = b->classic_map, *cnt = b->classic_cnt;
                const int32_t *prev_map = (r == 1) ? nullptr : map[(r - 1) & 1];
                const int32_t *prev_cnt = (r == 1) ? nullptr : cnt + ((r - 1) % 3);
                if (r == 1) HIPCHECK(hipMemsetAsync(cnt, 0, 3 * sizeof(int32_t), st));
                launch_compact(prev_map, prev_cnt, (int)n, b->classic_st.done, map[r & 1], cnt + (r % 3), st, cnt + ((r + 1) % 3));
                f.series_of = map[r & 1]; f.n_active = cnt + (r % 3);
                if (ybuf) {
                    launch_gather_columns(b->d_y, ld, map[r & 1], cnt + (r % 3), (int)n, (int)b->t_max, ybuf, ybuf_cols, st, (int)ybuf_cols);
                    f.y_round = ybuf; f.ld_round = ybuf_cols; f.gathered = 1; f.gather_cap = (int)ybuf_cols;
                }
                f.spec_below = 8192; f.spec2_below = 1024;
                f.budget = BUDGET[r]; f.budget_seq = (BUDGET[r] * 7) / 4;
                fns.round_auto(f, st);
            }
            b->fit_launches++;
        }
        launch_classic_final(kind, f, a, st);
        return;
    }
    if ((kind == CK_HW || kind == CK_SEASONAL_ES) && m > 48)       // CLASSIC_LDS_PERIOD: four candidate rings per lane
        a.ring_scratch = ensure_ring(b, (size_t)((b->n + 63) / 64) * 4u * (size_t)m * 64u);
    launch_classic(kind, a, st);
}

// Start of a run, ONE launch (it was a status upload plus six fills: a fifth of the device time of the fixed-parameter step):
// usable series start as "not computed" and only a kernel turns that into success, the forecasts start as NaN (all bits set, as
// the 0xff fill wrote them), the per-series totals and the model codes as zero.
__global__ void seed_outputs_kernel(int n, int ld, size_t nh, const int32_t *len, int32_t *status, double *yhat, double *lo, double *hi,
                                    int32_t *passes_total, int32_t *evals_total, int32_t *model_code)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double nanv = __longlong_as_double(-1LL);
    if (i < nh) { yhat[i] = nanv; lo[i] = nanv; hi[i] = nanv; }
    if (i < (size_t)ld) { passes_total[i] = 0; evals_total[i] = 0; model_code[i] = 0; }
    if (i < (size_t)n) status[i] = len[i] > 0 ? (int32_t)STATUS_NOT_COMPUTED : (int32_t)INSUFFICIENT_DATA;      // (finalize_lengths: a series too short to forecast reads as empty)
}

__global__ void fill_i32_kernel(int n, int32_t *dst, int32_t v)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n) dst[s] = v;
}

// d_detail holds the per-series fit status of the last model stage; map it into ErrorCodes.
__global__ void finish_status_kernel(int n, const int32_t *len, const int32_t *detail, int32_t *status)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (len[s] <= 0) return;                 // not in this group / unusable: keep base status
    status[s] = detail[s] == FIT_OK ? 0 : COMPUTATION_ERROR;
}

__global__ void explicit_select_kernel(int n, int h, const int32_t *len, const int32_t *fit_status, const double *yhat_slot,
                                       const int32_t *passes, const int32_t *evals, double *yhat, int32_t *detail,
                                       int32_t *passes_total, int32_t *evals_total)
{
    // one thread per forecast value (a thread per series copied its h values 8 h bytes apart from its neighbour's: 14 us for the M5 batch)
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t hh = (size_t)(h > 0 ? h : 1);
    const size_t s = idx / hh;
    const int i = (int)(idx - s * hh);
    if (s >= (size_t)n || len[s] <= 0) return;
    const int32_t fs = fit_status[s];
    if (i == 0) {
        detail[s] = fs;
        passes_total[s] = passes[s];
        evals_total[s] = evals[s];
    }
    if (fs == FIT_OK && i < h) yhat[idx] = yhat_slot[idx];
}

// count[0] = usable strictly positive series, count[1] = usable series (one workgroup)
__global__ void count_positive_kernel(int n, const int32_t *len, const uint32_t *flags, int32_t *count)
{
    __shared__ int sp[1024], su[1024];
    int p = 0, u = 0;
    for (int s = threadIdx.x; s < n; s += 1024)
        if (len[s] > 0) { u++; if (flags[s] & SF_POSITIVE) p++; }
    sp[threadIdx.x] = p; su[threadIdx.x] = u;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) { sp[threadIdx.x] += sp[threadIdx.x + o]; su[threadIdx.x] += su[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { count[0] = sp[0]; count[1] = su[0]; }
}

// notpos[s] = 1 unless series s is usable and strictly positive
__global__ void mark_nonpositive_kernel(int n, const int32_t *len, const uint32_t *flags, int32_t *notpos)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s < n) notpos[s] = (len[s] > 0 && (flags[s] & SF_POSITIVE)) ? 0 : 1;
}

// per (spec with a multiplicative component): the series the dense first round never visits are retired here
__global__ void retire_nonpositive_kernel(int n, const int32_t *len, const int32_t *notpos, int32_t *status, int32_t *done, int32_t *passes,
                                          int32_t *evals, int32_t *iters, double *aicc = nullptr, int32_t *f_passes = nullptr,
                                          int32_t *f_evals = nullptr, int32_t *f_iters = nullptr)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n || !notpos[s]) return;
    status[s] = len[s] > 0 ? FIT_NONPOSITIVE : FIT_SKIPPED;
    done[s] = 1; passes[s] = 0; evals[s] = 0; iters[s] = 0;
    // a spec without a launch chain: there is no final kernel to leave these behind
    if (aicc && len[s] > 0) { aicc[s] = __builtin_huge_val(); f_passes[s] = 0; f_evals[s] = 0; f_iters[s] = 0; }
}

// AutoETS fallback plan (forecast.rs:1327-1336): 1 Holt-Winters, 2 Holt, 3 SES(0.3)
__global__ void fallback_plan_kernel(int n, const int32_t *len, int period, uint32_t *mask, int32_t *detail, const int32_t *m_col)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (m_col) period = m_col[s];                  // merged batch: the column's own period
    const int L = len[s];
    if (L <= 0) { mask[s] = 0; return; }
    if (mask[s] == 0) { detail[s] = FIT_OK; return; }
    mask[s] = (period > 1 && L >= 2 * period) ? 1u : (L >= 10 ? 2u : 3u);
    detail[s] = FIT_PERIOD;      // until the planned stage fits the series (Holt-Winters with a period above ETS_MAX_PERIOD never runs)
}

// HoltWinters: 1 = two full seasons or more (seasonal fit), 2 = shorter (Holt); an unsupported period fails the long series
__global__ void holt_winters_plan_kernel(int n, const int32_t *len, int m, uint32_t *mask, int32_t *detail, const int32_t *m_col)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    if (m_col) m = m_col[s] > 2 ? m_col[s] : 2;    // merged batch: the column's own period
    const int L = len[s];
    mask[s] = L <= 0 ? 0u : (L >= 2 * m ? 1u : 2u);
    if (L > 0) detail[s] = FIT_PERIOD;        // overwritten by the stage that fits the series
}

// Fixed-parameter ETS (BASELINE config 2): what the first fit round would have decided about admissibility, and the given
// parameters parked where the final pass reads the optimum (vertex 0 of the simplex).  dim = number of coordinates.
__global__ void ets_fixed_setup_kernel(int n, size_t ld, const int32_t *len, const uint32_t *flags, int seasonal, int m, int n_param,
                                       int need_positive, int dim, double x0, double x1, double x2, double x3, int32_t *status,
                                       double *sim, int32_t *evals, int32_t *iters, int32_t *passes, int32_t *done)
{
    int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int L = len[s];
    int st = FIT_OK;
    if (L <= 0) st = FIT_SKIPPED;
    else if (seasonal && L < 2 * m) st = FIT_SHORT;
    else if (L < n_param + 2) st = FIT_SHORT;
    else if (need_positive && !(flags[s] & SF_POSITIVE)) st = FIT_NONPOSITIVE;
    status[s] = st;
    const double x[4] = {x0, x1, x2, x3};
    for (int i = 0; i < dim; i++) sim[(size_t)i * ld + s] = x[i];
    evals[s] = 0; iters[s] = 0; passes[s] = 0; done[s] = 1;
}

// Compact storage of the series block (round 6; ets_device.hpp YT_*, kernels.hip compact_block_kernel).  The optimiser streams every
// series hundreds of times, so a batch of counts -- the M5 shape: integers, or anything else whose every observation survives the
// round trip through float or uint16_t bit for bit -- is streamed from a 4- or 2-byte copy of the block: half / a quarter of the bytes
// of a pass and of the registers the staged rows take (30,490 x 1,913, 25 specs: 450 -> 396 ms per step with the float copy, same
// passes, same bits).  begin: one sweep of the block makes both copies and counts the observations that do NOT survive (+ 0.15 ms for
// 467 MB); decide (after the host has synchronised the stream): a copy is used only when its counter is zero, so ONE inexact
// observation anywhere keeps the whole batch on the fp64 block.  Not for given parameters (one pass: the sweep would cost more than
// it saves) or a handful of series.
// tune compact: -1 auto (narrowest exact type; batches of at least 65,536 cells), 0 never, 1 float at most, 2 narrowest exact type (1 / 2: any size).
bool compact_storage_begin(AnofoxHipBatch *b, const int32_t *d_len, hipStream_t st)
{
    b->y_type = YT_F64;
    if (b->tun.compact == 0 || b->fixed_params) return false;
    if (b->tun.compact < 0 && b->n * (size_t)std::max<size_t>(b->t_max, 1) < (size_t)1 << 16) return false;      // (auto: not for a handful of short series)
    const size_t cells = b->ld * std::max<size_t>(b->t_max, 1);
    if (b->yc_cells < cells) {
        if (b->d_yc32) b->retired.push_back(b->d_yc32);
        if (b->d_yc16) b->retired.push_back(b->d_yc16);
        b->d_yc32 = nullptr; b->d_yc16 = nullptr; b->yc_cells = 0;
        try {
            b->d_yc32 = dalloc<float>(cells);
            b->d_yc16 = dalloc<unsigned short>(cells);
        } catch (const HipFail &f) {
            if (!f.oom) throw;
            if (b->d_yc32) { dev_free(b->d_yc32, true); b->d_yc32 = nullptr; }      // no room beside a co-resident allocator: the fp64 block serves
            return false;
        }
        b->yc_cells = cells;
    }
    if (!b->d_misfit) b->d_misfit = dalloc<unsigned int>(2);
    HIPCHECK(hipMemsetAsync(b->d_misfit, 0, 2 * sizeof(unsigned int), st));
    launch_compact_block(b->d_y, b->ld, d_len, (int)b->n, (int)std::max<size_t>(b->t_max, 1), b->d_yc32, b->d_yc16, b->d_misfit, st);
    b->h_misfit[0] = b->h_misfit[1] = 1u;
    HIPCHECK(hipMemcpyAsync(b->h_misfit, b->d_misfit, sizeof b->h_misfit, hipMemcpyDeviceToHost, st));
    return true;
}
// `m`: the batch's seasonal period (1: none).  The uint16_t kernels exist for the variants without a period and with the weekly ring
// in registers (fit_units.hpp ets_u16_variant); any other period -- and a merged batch of several -- streams the float copy.
void compact_storage_decide(AnofoxHipBatch *b, bool pending, hipStream_t st, int m)
{
    b->y_type = YT_F64;
    if (!pending) return;
    HIPCHECK(hipStreamSynchronize(st));          // (already drained when the caller has just read its own counters)
    const bool u16_kernels = b->d_m_col == nullptr && (m <= 1 || m == 7);
    if (b->h_misfit[1] == 0u && b->tun.compact != 1 && u16_kernels) b->y_type = YT_U16;
    else if (b->h_misfit[0] == 0u) b->y_type = YT_F32;
}

void launch_fit_slots(AnofoxHipBatch *b, const std::vector<int> &specs, const int32_t *d_len, int m, bool skip_constant,
                      hipStream_t st)
{
    const size_t n = b->n, ld = b->ld;
    // Round budgets (streamed passes per launch).  Geometric, so every round retires roughly half of the
    // still-running problems and the compaction + gather in between stays a few percent of the passes.
    // iteration budgets of the rounds (the last one runs everything left to completion); ANOFOX_HIP_TUNE budgets=... overrides
    const std::vector<int> &BUDGET = b->tun.budgets;
    const int n_rounds = (int)BUDGET.size();
    const int n_lanes = std::min<int>(N_AUX_STREAMS, b->n_slots_cap);
    // fork: aux streams wait for everything queued on `st` so far
    HIPCHECK(hipEventRecord(b->ev_fit0, st));
    // several periods in one block (columns grouped by period in blocks of 64): the round kernels read the period PER LANE, so the
    // batch runs the ordinary schedule -- compaction and the dense re-gather pack survivors of different periods into one wave;
    // the final pass sweeps the original blocks (one period each)
    const bool merged = b->d_m_col != nullptr;
    // what the kernels stream: the fp64 block, or a compact copy of it (compact_storage_decide: every observation survives the narrower type)
    const int yt = b->y_type;
    const void *ybase = yt == YT_F32 ? (const void *)b->d_yc32 : (yt == YT_U16 ? (const void *)b->d_yc16 : (const void *)b->d_y);
    const int ybytes = yt == YT_F32 ? 4 : (yt == YT_U16 ? 2 : 8);
    const int lds_limit = merged ? ETS_MERGED_LDS_PERIOD : ETS_LDS_PERIOD;      // largest period whose seasonal ring stays in LDS
    // ONE spec (ETS with an explicit model, fitted or with given parameters): its launches go on the run's own stream -- no fork, no
    // join.  A cross-stream event wait is cheap when the host has just synchronised and expensive inside a pipeline of runs: twenty
    // fixed-parameter steps enqueued back to back took 1.4 ms each with the fork / join against 0.5 ms with a host wait in between.
    const bool inline_stream = specs.size() == 1;
    const int n_fork = inline_stream ? 0 : n_lanes;      // streams that carry work: one per spec
    auto spec_stream = [&](int idx) -> hipStream_t { return inline_stream ? st : b->aux[idx]; };
    for (int i = 0; i < n_fork; i++) HIPCHECK(hipStreamWaitEvent(b->aux[i], b->ev_fit0, 0));
    // enqueue order: most expensive specs first, dealt round-robin over the streams, so the long
    // multiplicative / damped / seasonal fits start together instead of queueing behind each other
    std::vector<size_t> order(specs.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    // (relative wave-time of one fitted problem of each spec, measured on the M5 shape with the wave trace -- profiles/r06_wave_residency_base.txt:
    //  resident wave-milliseconds per spec over a 30,490-series step, in thousands; the damped multiplicative-trend specs lead, and
    //  among them the additive-season one needs the most iterations, not the one with the longest step)
    auto cost = [&](int id) {
        static const int measured[30] = {/* A,N,* */ 3, 8, 14, /* A,A,* */ 7, 13, 27, /* A,Ad,* */ 17, 29, 41, /* A,M,* */ 14, 26, 22, /* A,Md,* */ 63, 92, 84,
                                         /* M,N,* */ 7, 0, 15, /* M,A,* */ 18, 0, 29, /* M,Ad,* */ 33, 0, 48, /* M,M,* */ 16, 0, 28, /* M,Md,* */ 65, 0, 95};
        return (id >= 0 && id < 30 && measured[id] > 0) ? measured[id] : 1;
    };
    // expected work of a spec on THIS batch: a spec with a multiplicative component only runs on the strictly positive series
    auto work = [&](int id) {
        const double live = (b->live_all >= 0 && spec_has_mult(id)) ? (double)b->live_pos : (b->live_all >= 0 ? (double)b->live_all : 1.0);
        return (double)cost(id) * (live + 1.0);
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return work(specs[x]) > work(specs[y]); });
    // one stream per spec (the runtime maps them onto the hardware queues; two explicit pairing schemes -- dedicated
    // queues for the heaviest specs, heaviest-with-lightest -- both measured 14 % slower on the 30-spec batch)
    std::vector<int> stream_of(order.size());
    for (size_t oi = 0; oi < order.size(); oi++) stream_of[oi] = (int)(oi % (size_t)n_lanes);
    std::vector<FitArgs> args(order.size());
    std::vector<FitLaunchers> fns(order.size());
    for (size_t oi = 0; oi < order.size(); oi++) {
        const size_t k = order[oi];
        const int id = specs[k];
        const int q = (int)(oi % (size_t)n_lanes);
        auto &lane = b->lanes[q];
        const int se = spec_season(id), ti = spec_trend_idx(id);
        const int tt = ti == 0 ? 0 : (ti <= 2 ? 1 : 2);
        FitArgs &a = args[oi];
        a = FitArgs{};
        a.y = (const double *)ybase; a.ld = ld; a.len = d_len; a.n_series = (int)n; a.t_rows = (int)std::max<size_t>(b->t_max, 1);
        a.m = se != 0 ? m : 1; a.h = b->h;
        a.l0 = b->d_l0 + (size_t)(se * 3 + tt) * ld;
        a.b0 = b->d_b0 + (size_t)(se * 3 + tt) * ld;
        a.fig = se == 1 ? b->d_fig_add : (se == 2 ? b->d_fig_mul : nullptr);
        a.fig_ld = ld;
        a.flags = b->d_flags;
        a.need_positive = spec_has_mult(id) ? 1 : 0;
        a.skip_constant = skip_constant ? 1 : 0;
        a.n_param = spec_n_param(id, a.m);
        a.aicc = b->d_aicc + k * ld;
        a.yhat = b->d_yhat_slots + k * n * (size_t)b->h;
        a.status = b->d_status_slots + k * ld;
        a.evals = b->d_evals_slots + k * ld;
        a.iters = b->d_iters_slots + k * ld;
        a.passes = b->d_passes_slots + k * ld;
        a.st = lane.st;
        a.ring_scratch = nullptr;
        a.m_col = b->d_m_col;
        if (b->sd_in_final && order.size() == 1) { a.mean = b->d_mean; a.sd_out = b->d_sd; }
        a.lane_stats = b->d_lane_stats ? b->d_lane_stats + 2 * k : nullptr;
        a.wave_trace = b->d_wave_trace;
        fns[oi] = ets_fit_launcher(id, (merged && se != 0) ? (a.m > lds_limit ? ETS_PERLANE_HBM : ETS_PERLANE_LDS) : a.m, yt);
        if (!fns[oi].round_seq || !fns[oi].round_spec || !fns[oi].round_auto || !fns[oi].final) throw HipFail{"no kernel for ETS spec id " + std::to_string(id)};
    }
    // Periods above the LDS limit keep the seasonal ring of every lane in HBM: one area per (spec, workgroup of the widest
    // launch = the speculative driver's 16 problems per workgroup)
    {
        size_t n_long = 0;
        for (size_t oi = 0; oi < order.size(); oi++) if (args[oi].m > lds_limit) n_long++;
        if (n_long) {
            // (a tiny batch runs EVERY problem on a wave of its own whatever the thresholds say: found in round 6 as the intermittent
            //  "Memory access fault by GPU" of test_two_level_speculation_is_bit_identical[20] -- 24 series, spec2_below = 20: 24 one-wave
            //  workgroups on a scratch sized for 20, four rings written past its end; never with the default thresholds, which are
            //  larger than any tiny batch.  docs/history/r06_round_log.md section E)
            const bool tiny_batch = (uint64_t)n * order.size() <= (uint64_t)TINY_BATCH_PROBLEMS;
            const size_t wg = std::max<size_t>((n + 15) / 16, std::min<size_t>(n, tiny_batch ? n : (size_t)std::max(b->spec2_below, b->spec2_below_md)));
            const size_t per_spec = wg * (size_t)m * 64u;
            if ((double)n_long * (double)per_spec * 8.0 > 64.0 * 1073741824.0)
                throw HipFail{"seasonal period " + std::to_string(m) + " on " + std::to_string(n) + " series needs more than 64 GiB of ring scratch: shard the batch"};
            double *base = ensure_ring(b, n_long * per_spec);
            size_t k = 0;
            for (size_t oi = 0; oi < order.size(); oi++) if (args[oi].m > lds_limit) { args[oi].ring_scratch = base + (k++) * per_spec; args[oi].ring_scratch_doubles = per_spec; }
        }
    }
    // Specs whose round kernels keep the Nelder-Mead simplex out of LDS: a global scratch per stream, one slice per workgroup of the
    // widest launch (four lanes per problem for every series, or one wave per problem for the last spec2_below / a tiny batch)
    {
        const bool tiny_batch = (uint64_t)n * order.size() <= (uint64_t)TINY_BATCH_PROBLEMS;
        const size_t wg = std::max<size_t>((n + 15) / 16, std::min<size_t>(n, tiny_batch ? n : (size_t)std::max(b->spec2_below, b->spec2_below_md)));
        for (size_t oi = 0; oi < order.size(); oi++) {
            if (!fns[oi].nm_scratch_per_wg) continue;
            auto &lane = b->lanes[oi % (size_t)n_lanes];
            const size_t need = (wg + 8) * fns[oi].nm_scratch_per_wg;         // (+ the waves that round a launch up to whole workgroups)
            if (lane.nm_scratch_doubles < need) {
                if (lane.nm_scratch) b->retired.push_back(lane.nm_scratch);
                lane.nm_scratch = nullptr; lane.nm_scratch_doubles = 0;
                lane.nm_scratch = dalloc<double>(need);
                lane.nm_scratch_doubles = need;
            }
            args[oi].nm_scratch = lane.nm_scratch; args[oi].nm_scratch_doubles = lane.nm_scratch_doubles;
        }
    }
    // Round-major submission: round r of every spec is enqueued before round r+1 of any, each spec on its
    // own stream, so that all specs advance together and the hardware queues never hold a long spec behind
    // another one.  Early rounds run the sequential driver (least arithmetic while problems outnumber
    // lanes), late rounds the speculative one (shortest critical path for the stragglers).
    // specs that are inadmissible for EVERY series of the group (a multiplicative component, no strictly positive series: the raw
    // M5 counts) get their per-series outputs from one small kernel and no launch chain at all -- 19 of the 25 chains of the
    // intermittent batch used to be compaction / gather / round launches over empty lists, queueing in front of the live ones
    std::vector<char> dead(order.size(), 0);
    if (b->none_pos && !b->fixed_params)
        for (size_t oi = 0; oi < order.size(); oi++)
            if (args[oi].need_positive) {
                dead[oi] = 1;
                const FitArgs &a = args[oi];
                hipLaunchKernelGGL(retire_nonpositive_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, spec_stream(stream_of[oi]), (int)n, d_len,
                                   b->d_notpos, a.status, a.st.done, a.st.passes, a.st.evals, a.st.iters, a.aicc, a.passes, a.evals, a.iters);
            }
    // additive-class specs, one lane per problem: four trial points per pass (K4).  Their pass is memory bound (6-10 instructions per
    // 8-byte load): against the sequential driver K4 is one pass per iteration instead of ~1.7, against four lanes per problem it is
    // the same bytes through a quarter of the load instructions (512 instead of 128 bytes each)
    std::vector<char> k4(order.size(), 0);
    {
        bool all_additive = true;           // (-1: only when the general-class specs see under half of the series)
        for (size_t oi = 0; oi < order.size(); oi++)
            if (!dead[oi] && spec_has_mult(specs[order[oi]]) && !(b->live_all > 0 && b->live_pos >= 0 && 2 * (int64_t)b->live_pos < (int64_t)b->live_all))
                all_additive = false;
        // ... and enough of them to fill the chip one lane per problem: two K4 waves per SIMD
        int64_t additive_live = 0;
        for (size_t oi = 0; oi < order.size(); oi++)
            if (!dead[oi] && !spec_has_mult(specs[order[oi]])) additive_live += b->live_all >= 0 ? (int64_t)b->live_all : (int64_t)n;
        // ... or (round 5) the batch oversubscribes the chip whatever its mix: with the general-class steps a third shorter (error-correction
        // form) the 25-spec batch of strictly positive series is no longer bound by fp64 issue alone, and the additive specs' ~41 % fewer
        // passes are worth their arithmetic -- 477.6 -> 450.9 ms on the 30,490-series batch, same box (profiles/r05_tune_sweep.txt).  One
        // lane per problem fills two waves per SIMD from 131,072 live problems on; below four times that the chains' latency decides
        int64_t all_live = 0;
        for (size_t oi = 0; oi < order.size(); oi++)
            if (!dead[oi]) all_live += (spec_has_mult(specs[order[oi]]) && b->live_pos >= 0) ? (int64_t)b->live_pos : (b->live_all >= 0 ? (int64_t)b->live_all : (int64_t)n);
        // (round 6) ... as long as the run streams the fp64 block.  K4 buys 41 % fewer passes with four times the arithmetic: a trade for a
        // pass bound by its BYTES.  Over a compact copy (2 or 4 bytes per observation) the additive pass is bound by its instructions like
        // every other, and K4 loses: intermittent M5 batch 58.0 -> 53.8 ms, 125k x 1,024 99.9 -> 75.5 ms, 125k x 256 with all 25 specs
        // 148-156 -> 141-142 ms, the 25-spec M5 batch unchanged (profiles/r06_k4_shapes.txt)
        const bool on = b->tun.k4 > 0 || (b->tun.k4 < 0 && yt == YT_F64 && ((all_additive && additive_live >= 131072) || all_live >= 524288));
        for (size_t oi = 0; oi < order.size(); oi++)
            k4[oi] = on && !dead[oi] && !b->fixed_params && !spec_has_mult(specs[order[oi]]) && fns[oi].round_k4 && fns[oi].round_auto_k4;
    }
    // Mid-size batches (round 6): fewer than 524,288 live problems (below that the first rounds run four lanes per problem for EVERY
    // spec) but so many that four lanes for everybody oversubscribe the resident lanes more than eight times -- the 15,245-series
    // shard of a 2-GPU M5 job: 381k problems, 1.5 M lanes on 131k.  There the first round's lanes are dealt by expected work: specs
    // in cost order get four lanes per problem while the total stays under fill_target x the resident lanes, the rest start one
    // lane per problem (the least arithmetic) and switch to four lanes when 1 / fill_late_div of their problems are left; one wave
    // per problem at 1 / fill_s2_div.  Measured (profiles/r06_shard_policy.txt, shard of a 2-rank job alone on one GPU): 314 -> 280 ms.
    // SMALLER batches keep four lanes for everybody: the 7,623- and 3,812-series shards (4 / 8 ranks) are bound by their longest
    // chain from the first round, and dealing them fewer lanes makes it longer (204 -> 246-320 ms, 173 -> 187-215 ms) -- the chip
    // absorbs three to six times its resident lanes of speculative work better than a chain absorbs 1.7 passes per iteration.
    // Same trajectories whatever the driver (test_schedule_variants_are_bit_identical).
    std::vector<char> fill_spec4(order.size(), 0);
    bool fill_mode = false;
    {
        int64_t lanes = 0;
        std::vector<int64_t> live_of(order.size(), 0);
        for (size_t oi = 0; oi < order.size(); oi++) {
            if (dead[oi]) continue;
            live_of[oi] = (spec_has_mult(specs[order[oi]]) && b->live_pos >= 0) ? (int64_t)b->live_pos : (b->live_all >= 0 ? (int64_t)b->live_all : (int64_t)n);
            lanes += live_of[oi];
        }
        const int64_t resident = 2 * 1024 * 64;
        const bool tiny_batch = (uint64_t)n * order.size() <= (uint64_t)TINY_BATCH_PROBLEMS;
        fill_mode = b->tun.fill_policy > 0 && !tiny_batch && !b->fixed_params && order.size() > 1 && b->seq_rounds_env < 0 && b->seq_rounds == 0 &&
                    lanes * 4 > 8 * resident;
        if (fill_mode) {
            const int64_t target = resident * (int64_t)b->tun.fill_target / 100;
            for (size_t oi = 0; oi < order.size(); oi++) {                  // `order` is by expected work, most first
                if (dead[oi] || k4[oi]) continue;
                if (lanes + 3 * live_of[oi] <= target) { fill_spec4[oi] = 1; lanes += 3 * live_of[oi]; }
            }
        }
    }
    if (b->use_gather && !b->fixed_params) {
        // gather blocks of the specs that have something to fit: as many columns as the spec can ever have running (the strictly
        // positive series for a spec with a multiplicative component), scaled down together if that exceeds the budget
        const size_t T = std::max<size_t>(b->t_max, 1);
        std::vector<size_t> need(order.size(), 0);
        double total = 0.0;
        for (size_t oi = 0; oi < order.size(); oi++) {
            if (dead[oi]) continue;
            size_t c = (b->use_pos && args[oi].need_positive && b->live_pos >= 0) ? (size_t)b->live_pos : n;
            need[oi] = std::min(ld, (c + 63) / 64 * 64);
            total += (double)need[oi] * (double)T * 8.0;
        }
        const double scale = total > b->gather_budget ? b->gather_budget / total : 1.0;
        for (size_t oi = 0; oi < order.size(); oi++) {
            if (dead[oi]) continue;
            auto &lane = b->lanes[oi % (size_t)n_lanes];
            size_t cols = scale < 1.0 ? (size_t)((double)need[oi] * scale) / 64 * 64 : need[oi];
            if (b->tun.gather_cols > 0) cols = std::min<size_t>(need[oi], (size_t)b->tun.gather_cols / 64 * 64);
            if (cols < need[oi] && cols < 1024 && b->tun.gather_cols <= 0) cols = 0;      // (a block for a few hundred stragglers is not worth its launches)
            if (lane.ybuf && lane.ybuf_cols >= cols) continue;                            // (a larger block from an earlier run serves as well)
            if (lane.ybuf) {
                HIPCHECK(hipDeviceSynchronize());                                          // an earlier run of this batch may still read the old block
                dev_free(lane.ybuf, true);
                lane.ybuf = nullptr; lane.ybuf_cols = 0;
            }
            while (cols >= 64) {
                try { lane.ybuf = dalloc<double>(T * cols); lane.ybuf_cols = cols; break; }
                catch (const HipFail &f) {
                    if (!f.oom) throw;
                    cols = cols / 2 / 64 * 64;                 // a co-resident allocator holds memory: half the columns, in the end none
                    if (cols < 1024) cols = 0;
                }
            }
        }
    }
    if (b->fixed_params) {
        // given smoothing parameters: no rounds at all -- admissibility + parameters, then the final pass below
        for (size_t oi = 0; oi < order.size(); oi++) {
            const int id = specs[order[oi]];
            FitArgs &a = args[oi];
            // coordinates present in this spec, in optimiser order (alpha, [beta*], [gamma*], [phi])
            double x[4] = {0.0, 0.0, 0.0, 0.0};
            int d = 0;
            x[d++] = b->fixed_x[0];
            if (spec_trend_idx(id) != 0) x[d++] = b->fixed_x[1];
            if (spec_season(id) != 0) x[d++] = b->fixed_x[2];
            if (spec_trend_idx(id) == 2 || spec_trend_idx(id) == 4) x[d++] = b->fixed_x[3];
            hipLaunchKernelGGL(ets_fixed_setup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, spec_stream(stream_of[oi]), (int)n, ld, d_len,
                               b->d_flags, spec_season(id) != 0 ? 1 : 0, a.m, a.n_param, a.need_positive, d, x[0], x[1], x[2], x[3], a.status,
                               a.st.sim, a.st.evals, a.st.iters, a.st.passes, a.st.done);
        }
        LAUNCHCHECK("ETS fixed-parameter setup");
    }
    // a handful of series: one launch per spec, every problem on a wave of its own (two iterations per pass) to completion -- one
    // series through AutoETS was 12 rounds x 3 launches x 25 specs
    const bool tiny = (uint64_t)n * order.size() <= (uint64_t)TINY_BATCH_PROBLEMS;
    auto enqueue_round = [&](const int r, const size_t oi) {
            if (dead[oi]) return;
            const int q = (int)(oi % (size_t)n_lanes);
            auto &lane = b->lanes[q];
            hipStream_t sq = spec_stream(stream_of[oi]);
            FitArgs &a = args[oi];
            const bool spec_mode = r >= b->seq_rounds;
            a.first_round = (r == 0);
            a.wave_trace_tag = ((unsigned long long)specs[order[oi]] << 32) | ((unsigned long long)r << 16);
            a.wave_prio = (int)oi < b->tun.prio_top ? std::max(1, 3 - (int)oi) : 0;
            a.spec_below = -1; a.spec2_below = -1;
            a.gathered = 0;
            if (tiny) {
                a.y_round = b->d_y; a.ld_round = ld; a.series_of = nullptr; a.n_active = nullptr;
                a.budget = 1 << 30; a.budget_seq = a.budget;
                fns[oi].round_spec2(a, sq);
                b->fit_launches++;
                return;
            }
            if (r == 0 && b->use_pos && a.need_positive) {
                // mixed batch: this spec is admissible for the strictly positive series only -- its first round runs on
                // their dense list (built once per group) instead of sweeping every wave for a few live lanes
                hipLaunchKernelGGL(retire_nonpositive_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sq, (int)n, d_len, b->d_notpos,
                                   a.status, lane.st.done, lane.st.passes, lane.st.evals, lane.st.iters);
                a.series_of = b->d_pos_map; a.n_active = b->d_pos_cnt;
                if (b->d_ypos) { a.y_round = b->d_ypos; a.ld_round = ld; a.gathered = 1; a.gather_cap = (int)ld; }
                else { a.y_round = b->d_y; a.ld_round = ld; }
            } else if (r == 0) {
                a.y_round = a.y; a.ld_round = ld; a.series_of = nullptr; a.n_active = nullptr;
            } else {
                const int32_t *prev_map = (r == 1) ? nullptr : lane.map[(r - 1) & 1];
                const int32_t *prev_cnt = (r == 1) ? nullptr : lane.cnt + ((r - 1) % 3);
                if (r == 1) HIPCHECK(hipMemsetAsync(lane.cnt, 0, 3 * sizeof(int32_t), sq));      // then the counters rotate: no more memsets
                launch_compact(prev_map, prev_cnt, (int)n, lane.st.done, lane.map[r & 1], lane.cnt + (r % 3), sq, lane.cnt + ((r + 1) % 3));
                a.series_of = lane.map[r & 1]; a.n_active = lane.cnt + (r % 3);
                if (b->use_gather && lane.ybuf) {
                    launch_gather_columns(a.y, ld, lane.map[r & 1], lane.cnt + (r % 3), (int)n, (int)b->t_max, lane.ybuf, lane.ybuf_cols, sq,
                                          (int)lane.ybuf_cols, ybytes);
                    a.y_round = lane.ybuf; a.ld_round = lane.ybuf_cols; a.gathered = 1; a.gather_cap = (int)lane.ybuf_cols;
                } else {
                    a.y_round = a.y; a.ld_round = ld;
                }
            }
            const int s2 = (spec_trend_idx(specs[order[oi]]) == 4) ? b->spec2_below_md : b->spec2_below;
            if (fill_mode && !k4[oi]) {
                const int64_t live_i = (spec_has_mult(specs[order[oi]]) && b->live_pos >= 0) ? (int64_t)b->live_pos : (b->live_all >= 0 ? (int64_t)b->live_all : (int64_t)n);
                const int s2v = (int)std::max<int64_t>(32, std::min<int64_t>(s2 > 0 ? s2 : 32, live_i / std::max(1, b->tun.fill_s2_div)));
                a.budget = BUDGET[r]; a.budget_seq = (BUDGET[r] * 7) / 4;
                if (r == 0) {
                    if (fill_spec4[oi]) fns[oi].round_spec(a, sq);
                    else { a.budget = a.budget_seq; fns[oi].round_seq(a, sq); }
                } else {
                    a.spec_below = fill_spec4[oi] ? 0x7fffffff : (int)std::max<int64_t>(64, live_i / std::max(1, b->tun.fill_late_div));
                    a.spec2_below = s2v;
                    fns[oi].round_auto(a, sq);
                }
                b->fit_launches++;
                return;
            }
            if (k4[oi]) {
                // additive spec of a memory-bound run: one lane per problem with four trial points per pass wherever four LANES per
                // problem would otherwise run -- the same bytes per iteration through a quarter of the load instructions (a wave of
                // four-lane groups moves 128 bytes per load, and it is the CU's address pipeline that such a run saturates) -- as
                // long as the spec still has more problems than `spec_below`; fewer are a latency problem again: four lanes, then
                // one wave per problem
                a.budget = BUDGET[r]; a.budget_seq = BUDGET[r];
                const int64_t live = b->live_all >= 0 ? b->live_all : (int64_t)n;
                // The most expensive spec (`k4_top` of them) switches to four lanes per problem EARLIER, at `k4_top_below` live problems
                // (20,480; every other spec at spec_below = 8,192): its chain ends the step, alone on the chip for the last 10-25 ms of
                // the intermittent M5 batch with 477 one-lane waves on 1,024 SIMDs -- four lanes per problem are the same arithmetic on
                // four times the waves.  70.6-71.7 -> 66.1-68.1 ms on that batch (thresholds 14,336-24,576: the same; 12,288 and 32,768:
                // no gain -- at 32,768 the first round already runs four lanes while all six chains still fill the chip), 125k x 1,024:
                // 138-142 -> 132-139 ms, batches with general-class specs: unchanged (profiles/r04_ab_experiments.txt).
                const int sb = ((int)oi < b->tun.k4_top && b->tun.k4_top_below > 0) ? b->tun.k4_top_below : b->spec_below;
                if (r == 0) (live > sb ? fns[oi].round_k4 : fns[oi].round_spec)(a, sq);
                else {
                    a.spec_below = sb; a.spec2_below = s2 > 0 ? s2 : -1;
                    fns[oi].round_auto_k4(a, sq);
                }
            } else if (r == 0 || b->seq_rounds_env >= 0 || b->seq_rounds == 0) {
                // first round (no device count yet) or a forced schedule: the host picks the driver
                a.budget = spec_mode ? BUDGET[r] : (BUDGET[r] * 7) / 4;     // ~1.7 passes per iteration when sequential
                if (spec_mode && r > 0 && s2 > 0) {
                    // ... but the last s2 problems still go one per wave (device-side count), in the same launch
                    a.spec_below = 0x7fffffff; a.spec2_below = s2; a.budget_seq = a.budget;
                    fns[oi].round_auto(a, sq);
                } else
                    (spec_mode ? fns[oi].round_spec : fns[oi].round_seq)(a, sq);
            } else {
                // later rounds: the device-side count of running problems picks the driver -- sequential (least arithmetic)
                // while this spec still fills >= 1/8 of the chip, then speculative (one pass per iteration), then two-level
                // speculative (one problem per wave, two iterations per pass)
                a.spec_below = (spec_trend_idx(specs[order[oi]]) == 4) ? b->spec_below_md : b->spec_below;
                a.spec2_below = s2 > 0 ? s2 : -1;
                if ((int)oi < b->tun.top_boost_n) {         // (experiment: the chains that end the step get their parallelism earlier)
                    a.spec_below = (int)std::min<int64_t>(0x7fffffff, (int64_t)a.spec_below * b->tun.top_boost_pct / 100);
                    if (a.spec2_below > 0) a.spec2_below = (int)std::min<int64_t>(0x7fffffff, (int64_t)a.spec2_below * b->tun.top_boost_pct / 100);
                }
                // ONE launch: a launch whose workgroups only find out that another driver owns the round still has to be dispatched
                a.budget_seq = (BUDGET[r] * 7) / 4;
                a.budget = BUDGET[r];
                fns[oi].round_auto(a, sq);
            }
            b->fit_launches++;
    };
    const int total_rounds = b->fixed_params ? 0 : (tiny ? 1 : n_rounds);
    // Head start of the damped multiplicative-trend chains (tune dm_head_rounds, round 5): their fits are the step's critical path -- alone
    // on the chip one of them takes 185-280 ms of a 450 ms step (profiles/r05_step_anatomy.txt) -- and while all 25 chains start together
    // their early, full rounds share every SIMD with the cheap specs' waves.  With a head start their first rounds are enqueued alone,
    // the other specs' streams wait for an event recorded behind them, and fill the chip once those chains thin out.
    int head = 0;
    if (!tiny && !inline_stream && total_rounds > 0 && b->tun.dm_head_rounds > 0) {
        std::vector<size_t> dm, rest;
        for (size_t oi = 0; oi < order.size(); oi++)
            if (!dead[oi]) (spec_trend_idx(specs[order[oi]]) == 4 ? dm : rest).push_back(oi);
        if (!dm.empty() && !rest.empty()) {
            head = std::min(b->tun.dm_head_rounds, total_rounds);
            for (int r = 0; r < head; r++) for (size_t oi : dm) enqueue_round(r, oi);
            LAUNCHCHECK("ETS fit head rounds");
            for (size_t oi : dm) HIPCHECK(hipEventRecord(b->ev_join[stream_of[oi]], spec_stream(stream_of[oi])));
            for (size_t oi : rest) for (size_t od : dm) HIPCHECK(hipStreamWaitEvent(spec_stream(stream_of[oi]), b->ev_join[stream_of[od]], 0));
        }
    }
    for (int r = 0; r < total_rounds; r++) {
        for (size_t oi = 0; oi < order.size(); oi++) {
            if (r < head && spec_trend_idx(specs[order[oi]]) == 4) continue;      // (already enqueued)
            enqueue_round(r, oi);
        }
        LAUNCHCHECK("ETS fit round");
    }
    // fixed parameters: the one streamed pass IS the workload -- the fit events bracket exactly that launch, on its stream
    if (b->fixed_params && order.size() == 1) HIPCHECK(hipEventRecord(b->ev_fit0, spec_stream(stream_of[0])));
    for (size_t oi = 0; oi < order.size(); oi++) if (!dead[oi]) fns[oi].final(args[oi], spec_stream(stream_of[oi]));
    if (b->fixed_params && order.size() == 1) HIPCHECK(hipEventRecord(b->ev_fit1, spec_stream(stream_of[0])));
    LAUNCHCHECK("ETS final pass");
    for (int i = 0; i < n_fork; i++) {
        HIPCHECK(hipEventRecord(b->ev_join[i], b->aux[i]));
        HIPCHECK(hipStreamWaitEvent(st, b->ev_join[i], 0));
    }
    if (!(b->fixed_params && order.size() == 1)) HIPCHECK(hipEventRecord(b->ev_fit1, st));
    b->timed_fit = true;
    b->n_problems += (uint64_t)specs.size() * n;
    b->insp_args = args; b->insp_fns = fns; b->insp_stream = stream_of; b->insp_m = m;
    b->insp_spec.resize(order.size());
    for (size_t oi = 0; oi < order.size(); oi++) b->insp_spec[oi] = specs[order[oi]];
    b->insp_ok = true;
}

void run_group(AnofoxHipBatch *b, int period, const int32_t *d_len, hipStream_t st)
{
    const Plan &p = b->plan;
    const size_t n = b->n, ld = b->ld;
    const int blocks256 = (int)((n + 255) / 256);
    // sd_in_final: the batch has ONE candidate spec, whose final pass carries the second sweep of the intervals' sd (FitArgs::sd_out)
    b->sd_in_final = false;
    auto prep = [&](int m, bool states, bool sd_in_final = false, int skip_types = 0) {
        PrepArgs a{};
        a.skip_sd = sd_in_final ? 1 : 0;
        a.skip_types = skip_types;
        b->sd_in_final = sd_in_final;
        a.y = b->d_y; a.ld = ld; a.len = d_len; a.n_series = (int)n; a.m = m;
        a.m_col = (states && m >= 2) ? b->d_m_col : nullptr;
        a.mean = b->d_mean; a.sd = b->d_sd; a.flags = b->d_flags;
        a.scratch = nullptr;
        if (states) {
            if (m >= 2) ensure_fig(b, m);
            a.fig_add = b->d_fig_add; a.fig_mul = b->d_fig_mul; a.l0 = b->d_l0; a.b0 = b->d_b0;
            a.t_rows = (int)std::max<size_t>(b->t_max, 1);
            if (m >= 2 && m <= ETS_MAX_PERIOD && !(m == 7 && !a.m_col)) {       // season_figures_kernel: series longer than its LDS use a scratch
                const size_t sc = season_scratch_doubles((int)n, a.t_rows, m);
                if (sc) a.scratch = ensure_prep_scratch(b, sc);
            }
        }
        launch_prep(a, st);
    };
    auto simple = [&](int kind, int per, int window) {
        SimpleArgs a{};
        a.y = b->d_y; a.ld = ld; a.len = d_len; a.n_series = (int)n;
        a.kind = kind; a.h = b->h; a.period = per; a.window = window;
        a.yhat = b->d_yhat; a.status = b->d_status;
        launch_simple(a, st);
    };
    auto finish = [&]() { hipLaunchKernelGGL(finish_status_kernel, dim3(blocks256), dim3(256), 0, st, (int)n, d_len, b->d_detail, b->d_status); };

    // A period the kernels cannot hold fails the group's series loudly (COMPUTATION_ERROR naming the cap) -- never a silent
    // non-seasonal fit.  The reference takes any period (forecast.rs:528-537); 2,048 covers every calendar period.
    const bool uses_period = p.model == M_AutoETS || p.model == M_HoltWinters || p.model == M_SeasonalES || p.model == M_SeasonalESOptimized ||
                             (p.model == M_ETS && (p.ets_spec_id < 0 || spec_season(p.ets_spec_id) != 0));
    const bool arima_period = p.model == M_AutoARIMA && period > ETS_MAX_PERIOD;      // seasonal ARIMA terms: LDS rings up to m = 24, an HBM scratch ring up to 2,048
    if ((uses_period && period > ETS_MAX_PERIOD) || arima_period) {
        prep(1, false);
        HIPCHECK(hipMemsetAsync(b->d_detail, 0, ld * sizeof(int32_t), st));
        hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks256), dim3(256), 0, st, (int)n, b->d_detail, (int32_t)FIT_PERIOD);
        finish();
        LAUNCHCHECK("period cap");
        return;
    }

    switch (p.model) {
    case M_Naive: prep(1, false); simple(SK_NAIVE, 1, 0); break;
    case M_SeasonalNaive: prep(1, false); simple(SK_SEASONAL_NAIVE, period, 0); break;
    case M_SMA: prep(1, false); simple(SK_SMA, 1, b->opt.window > 0 ? b->opt.window : std::max(period, 3)); break;
    case M_RandomWalkDrift: prep(1, false); simple(SK_DRIFT, 1, 0); break;
    case M_ARIMA: prep(1, false); simple(SK_TOY_ARIMA, 1, 0); break;
    case M_SES: prep(1, false); run_classic(b, CK_SES, d_len, 1, 0, 0.3, 0, nullptr, 0, 0, false, st); finish(); break;
    case M_SESOptimized: prep(1, false); run_classic(b, CK_SES, d_len, 1, 1, 0.0, 0, nullptr, 0, 0, false, st); finish(); break;
    case M_Holt: prep(1, false); run_classic(b, CK_HOLT, d_len, 1, 1, 0.0, 0, nullptr, 0, 0, false, st); finish(); break;
    case M_HoltWinters: {
        int m = std::max(period, 2);
        prep(1, false);
        // fewer than two seasons -> Holt's linear trend under the same name (the crate's fallback, pinned by
        // test/sql/ts_forecast_exp_smoothing.test:498-503 and ts_forecast_params.test:203-207)
        hipLaunchKernelGGL(holt_winters_plan_kernel, dim3(blocks256), dim3(256), 0, st, (int)n, d_len, m, b->d_mask, b->d_detail, (const int32_t *)b->d_m_col);
        if (m <= ETS_MAX_PERIOD) run_classic(b, CK_HW, d_len, m, 1, 0.0, 2 * m, b->d_mask, 1u, 0, false, st);
        run_classic(b, CK_HOLT, d_len, 1, 1, 0.0, 0, b->d_mask, 2u, 0, false, st);
        finish();
        break;
    }
    case M_SeasonalES: case M_SeasonalESOptimized: {
        int m = std::max(period, 2);
        prep(1, false);
        if (m > ETS_MAX_PERIOD) HIPCHECK(hipMemsetAsync(b->d_detail, 0xff, ld * sizeof(int32_t), st));
        else run_classic(b, CK_SEASONAL_ES, d_len, m, p.model == M_SeasonalESOptimized ? 1 : 0, 0.1, m, nullptr, 0, 0, false, st);
        finish();
        break;
    }
    case M_ETS:
        if (p.ets_spec_id >= 0) {
            int id = p.ets_spec_id;
            int m = 1;
            if (spec_season(id) != 0 && period > 1) m = period;
            else id = id - spec_season(id);             // forecast.rs:1347-1351: no usable period -> non-seasonal
            // (one candidate spec: its final pass carries the intervals' sd, and only its own season type is prepared)
            prep(m, true, !(spec_season(id) != 0 && m > ETS_MAX_PERIOD), spec_season(id) == 2 ? 0 : 2);
            if (spec_season(id) != 0 && m > ETS_MAX_PERIOD) {
                HIPCHECK(hipMemsetAsync(b->d_detail, 0x04, ld * sizeof(int32_t), st));   // != FIT_OK: unsupported period
                finish();
                break;
            }
            std::vector<int> specs{id};
            compact_storage_decide(b, compact_storage_begin(b, d_len, st), st, m);
            launch_fit_slots(b, specs, d_len, m, false, st);
            hipLaunchKernelGGL(explicit_select_kernel, dim3((unsigned)((n * (size_t)std::max(b->h, 1) + 255) / 256)), dim3(256), 0, st, (int)n, b->h, d_len, b->d_status_slots,
                               b->d_yhat_slots, b->d_passes_slots, b->d_evals_slots, b->d_yhat, b->d_detail, b->d_passes_total,
                               b->d_evals_total);
            finish();
        } else {
            prep(1, false);
            // default chain decided by length only: mark every usable series for fallback
            HIPCHECK(hipMemsetAsync(b->d_mask, 0x01, ld * sizeof(uint32_t), st));
            hipLaunchKernelGGL(fallback_plan_kernel, dim3(blocks256), dim3(256), 0, st, (int)n, d_len, period, b->d_mask, b->d_detail, (const int32_t *)nullptr);
            if (period > 1 && period <= ETS_MAX_PERIOD)
                run_classic(b, CK_HW, d_len, period, 1, 0.0, 2 * period, b->d_mask, 1u, 0, false, st);
            run_classic(b, CK_HOLT, d_len, 1, 1, 0.0, 0, b->d_mask, 2u, 0, false, st);
            run_classic(b, CK_SES, d_len, 1, 0, 0.3, 0, b->d_mask, 3u, 0, false, st);
            finish();
        }
        break;
    case M_AutoETS: {
        int m = (period > 1 && period <= ETS_MAX_PERIOD) ? period : 1;
        prep(m, true);
        const bool compact_pending = compact_storage_begin(b, d_len, st);
        std::vector<int> specs;
        for (int id = 0; id < 30; id++) {
            if (spec_season(id) != 0 && m <= 1) continue;
            if (!spec_is_valid(id) || !pool_allows(p.pool, id)) continue;
            specs.push_back(id);
        }
        // slot order = spec-id order (first minimum wins ties in the selection)
        b->h_slot_spec.assign(specs.begin(), specs.end());
        HIPCHECK(hipMemcpyAsync(b->d_slot_spec, b->h_slot_spec.data(), b->h_slot_spec.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (b->seq_rounds_env < 0) {
            // Choose the Nelder-Mead driver of the early rounds from the number of LIVE problems: strictly
            // positive series run every spec, the others only the additive ones.  One 4-byte read-back.
            hipLaunchKernelGGL(count_positive_kernel, dim3(1), dim3(1024), 0, st, (int)n, d_len, b->d_flags, b->d_count);
            int32_t cnt[2] = {0, 0};
            HIPCHECK(hipMemcpyAsync(cnt, b->d_count, sizeof cnt, hipMemcpyDeviceToHost, st));
            HIPCHECK(hipStreamSynchronize(st));             // (the compact copies' misfit counters have arrived too)
            size_t n_add = 0;
            for (int id : specs) if (!spec_has_mult(id)) n_add++;
            const double live = (double)cnt[0] * (double)specs.size() + (double)(cnt[1] - cnt[0]) * (double)n_add;
            b->seq_rounds = live >= 8.0 * 65536.0 ? 4 : 0;         // >= 8 problems per SIMD lane-slot: VALU-bound, go sequential
            b->use_pos = cnt[0] > 0 && (double)cnt[0] < 0.7 * (double)cnt[1];
            b->live_pos = cnt[0]; b->live_all = cnt[1];
        } else {
            b->use_pos = false;
            b->live_pos = b->live_all = -1;
        }
        compact_storage_decide(b, compact_pending, st, m);
        b->none_pos = b->live_all >= 0 && b->live_pos == 0;       // no strictly positive series at all: the 19 specs with a multiplicative component have nothing to fit
        if (b->use_pos || b->none_pos)
            hipLaunchKernelGGL(mark_nonpositive_kernel, dim3((unsigned)blocks256), dim3(256), 0, st, (int)n, d_len, b->d_flags, b->d_notpos);
        if (b->use_pos) {
            launch_compact(nullptr, nullptr, (int)n, b->d_notpos, b->d_pos_map, b->d_pos_cnt, st);
            if (b->use_gather) {
                if (!b->d_ypos) b->d_ypos = dalloc<double>(std::max<size_t>(b->t_max, 1) * ld);
                // (the strictly positive columns, in the storage type the fit streams)
                const int yt = b->y_type;
                launch_gather_columns(yt == YT_F32 ? (const void *)b->d_yc32 : (yt == YT_U16 ? (const void *)b->d_yc16 : (const void *)b->d_y), ld, b->d_pos_map,
                                      b->d_pos_cnt, (int)n, (int)b->t_max, b->d_ypos, ld, st, (int)ld, yt == YT_F32 ? 4 : (yt == YT_U16 ? 2 : 8));
            }
        }
        launch_fit_slots(b, specs, d_len, m, true, st);
        SelectArgs sa{};
        sa.n_series = (int)n; sa.h = b->h; sa.n_slots = (int)specs.size(); sa.ld = ld; sa.len = d_len;
        sa.aicc = b->d_aicc; sa.yhat_slots = b->d_yhat_slots; sa.slot_spec = b->d_slot_spec;
        sa.status_slots = b->d_status_slots; sa.passes_slots = b->d_passes_slots; sa.evals_slots = b->d_evals_slots;
        sa.yhat = b->d_yhat; sa.model_code = b->d_model_code; sa.status = b->d_detail; sa.fallback_mask = b->d_mask;
        sa.passes_total = b->d_passes_total; sa.evals_total = b->d_evals_total;
        launch_select(sa, st);
        hipLaunchKernelGGL(fallback_plan_kernel, dim3(blocks256), dim3(256), 0, st, (int)n, d_len, period, b->d_mask, b->d_detail, (const int32_t *)b->d_m_col);
        if (period > 1 && period <= ETS_MAX_PERIOD)
            run_classic(b, CK_HW, d_len, period, 1, 0.0, 2 * period, b->d_mask, 1u, 0, true, st);
        run_classic(b, CK_HOLT, d_len, 1, 1, 0.0, 0, b->d_mask, 2u, 0, true, st);
        run_classic(b, CK_SES, d_len, 1, 0, 0.3, 0, b->d_mask, 3u, 0, true, st);
        finish();
        break;
    }
    case M_AutoARIMA: {
        prep(1, false);
        ArimaArgs aa{};
        aa.y = b->d_y; aa.ld = ld; aa.len = d_len; aa.n_series = (int)n;
        aa.m = period > 1 ? period : 1;                        // <= 2,048 here: longer periods failed loudly above
        aa.long_scratch = nullptr;
        aa.m_col = (b->d_m_col && aa.m > 24) ? b->d_m_col : nullptr;      // merged batch of long periods: `period` is the largest
        if (aa.m > 24) aa.long_scratch = ensure_ring(b, arima_long_scratch_doubles((int)n, aa.m, arima_max_fit_waves()));
        aa.h = b->h;
        aa.ws = b->ar_w; aa.ws_bytes = b->ar_ws_bytes; aa.t_max = (int)std::max<size_t>(b->t_max, 1); aa.wlen = b->ar_wlen; aa.d = b->ar_d; aa.D = b->ar_D; aa.wmean = b->ar_wmean; aa.wsd = b->ar_wsd;
        aa.last_d0 = b->ar_l0; aa.last_d1 = b->ar_l1; aa.order = b->ar_order; aa.xbest = b->ar_x; aa.aicc = b->ar_aicc;
        aa.status = b->d_detail; aa.evals = b->d_evals_total; aa.passes = b->d_passes_total; aa.models = b->ar_models;
        aa.yhat = b->d_yhat; aa.model_code = b->d_model_code;
        aa.ml_refit = b->arima_method == ANOFOX_ARIMA_CSS_ML ? 1 : 0;
        aa.trace = b->tun.arima_trace;
        aa.lookahead = b->tun.arima_lookahead; aa.lookahead_depth = b->tun.arima_lookahead_depth; aa.spec_factor = b->tun.arima_spec_factor; aa.refit_budget = b->tun.arima_refit_budget; aa.queue_sort = b->tun.arima_queue_sort;
        aa.prep_lanes = b->tun.arima_prep_lanes < 1 ? 1 : (b->tun.arima_prep_lanes > 64 ? 64 : b->tun.arima_prep_lanes);
        aa.concurrent = &g_arima_runs; aa.shared_chunk_rounds = b->tun.arima_shared_chunk_rounds;
        HIPCHECK(hipEventRecord(b->ev_fit0, st));
        struct RunCount { RunCount() { g_arima_runs.fetch_add(1); } ~RunCount() { g_arima_runs.fetch_sub(1); } } run_count;
        try { b->fit_launches += launch_arima(aa, st); }
        catch (const std::exception &e) { throw HipFail{e.what()}; }
        HIPCHECK(hipEventRecord(b->ev_fit1, st));
        b->timed_fit = true;
        b->n_problems += n;
        finish();
        break;
    }
    default: throw HipFail{"model not implemented"};
    }
    LAUNCHCHECK(model_name(p.model));
}

void run_batch(AnofoxHipBatch *b, hipStream_t st)
{
    const size_t n = b->n, ld = b->ld;
    (void)hipGetLastError();     // the launch checks below must only see THIS run's errors (an earlier, already reported failure is sticky)
    b->fit_launches = 0;
    b->n_problems = 0;
    b->timed_fit = false;
    b->insp_ok = false;
    b->quiesced = false;         // (from here on, also when a launch below fails)
    b->last_stream = st;
    HIPCHECK(hipEventRecord(b->ev_start, st));
    if (b->d_lane_stats) HIPCHECK(hipMemsetAsync(b->d_lane_stats, 0, 2 * (size_t)b->n_slots_cap * sizeof(unsigned long long), st));
    if (!b->tun.wave_trace.empty() && b->n_slots_cap > 0) {
        constexpr size_t WAVE_TRACE_RECORDS = 4u << 20;
        if (!b->d_wave_trace) b->d_wave_trace = dalloc<unsigned long long>(4 + 4 * WAVE_TRACE_RECORDS);
        const unsigned long long head[4] = {0ull, (unsigned long long)WAVE_TRACE_RECORDS, 0ull, 0ull};
        HIPCHECK(hipMemcpyAsync(b->d_wave_trace, head, sizeof head, hipMemcpyHostToDevice, st));
        HIPCHECK(hipStreamSynchronize(st));        // (`head` is a stack array)
    }
    {
        const size_t nh = n * (size_t)std::max(b->h, 0);
        const size_t cover = std::max<size_t>(std::max(nh, ld), 1);
        hipLaunchKernelGGL(seed_outputs_kernel, dim3((unsigned)((cover + 255) / 256)), dim3(256), 0, st, (int)n, (int)ld, nh, (const int32_t *)b->d_len, b->d_status,
                           b->d_yhat, b->d_lo, b->d_hi, b->d_passes_total, b->d_evals_total, b->d_model_code);
        LAUNCHCHECK("output seeding");
    }
    // group series by seasonal period (all equal unless auto-detection ran).  Detection gives ~140 distinct periods per
    // thousand M5-like series and every group is one run of the whole pipeline, so series are grouped by the period the
    // model actually USES: none for the non-seasonal models.
    auto used_period = [&](int period) {
        switch (b->plan.model) {
        case M_Naive: case M_RandomWalkDrift: case M_ARIMA: case M_SES: case M_SESOptimized: case M_Holt: return 1;
        case M_AutoARIMA:
            // a DETECTED period goes to the seasonal search exactly like an explicit one (forecast.rs:528-537 hands it to
            // forecast_auto_arima, :1448-1452 passes any period > 1 to with_seasonal_period): used up to 2,048 (rings in LDS up to
            // 24, in HBM scratch above), failing loudly beyond.  Rounds 2-3 made detected periods above 24 non-seasonal -- a
            // product limit the oracle had been written to share; both are gone.
            return period > ETS_MAX_PERIOD ? ETS_MAX_PERIOD + 1 : (period > 1 ? period : 1);
        default: return period;
        }
    };
    std::map<int, std::vector<size_t>> groups;
    bool uniform = true;
    for (size_t s = 1; s < n; s++) if (used_period(b->h_period[s]) != used_period(b->h_period[0])) { uniform = false; break; }
    if (b->d_m_col) run_group(b, b->merged_m_max, b->d_len, st);        // several periods, one run: the kernels read the period per column
    else if (uniform) run_group(b, n ? used_period(b->h_period[0]) : 1, b->d_len, st);
    else {
        for (size_t s = 0; s < n; s++) groups[used_period(b->h_period[s])].push_back(s);
        std::vector<int32_t> eff(ld);
        for (auto &g : groups) {
            std::fill(eff.begin(), eff.end(), 0);
            for (size_t s : g.second) if (b->h_base_status[s] == 0) eff[s] = b->h_len[s];
            HIPCHECK(hipStreamSynchronize(st));   // d_len_group is reused between groups
            HIPCHECK(hipMemcpy(b->d_len_group, eff.data(), ld * sizeof(int32_t), hipMemcpyHostToDevice));
            run_group(b, g.first, b->d_len_group, st);
        }
    }
    IntervalArgs ia{};
    ia.n_series = (int)n; ia.h = b->h; ia.yhat = b->d_yhat; ia.sd = b->d_sd; ia.status = b->d_status; ia.z = b->plan.z;
    ia.lower = b->d_lo; ia.upper = b->d_hi;
    launch_intervals(ia, st);
    LAUNCHCHECK("intervals");
    HIPCHECK(hipEventRecord(b->ev_stop, st));
    b->last_stream = st;
    b->ran = true;
    b->quiesced = false;
}

std::string series_error_message(const AnofoxHipBatch *b, size_t s, int code, int detail)
{
    char buf[512];
    if (code == INTERNAL_ERROR) return "Internal error: the device never computed this series (kernel did not run)";
    if (code == INSUFFICIENT_DATA) {
        int got = b->h_len[s];
        std::snprintf(buf, sizeof buf, "Insufficient data: need at least %d observations, got %d", got == 0 ? 1 : 3, got);
        return buf;
    }
    const char *why = detail == FIT_SHORT ? "not enough observations for this model"
                      : detail == FIT_NONPOSITIVE ? "multiplicative components require strictly positive data"
                      : detail == FIT_NONFINITE ? "likelihood is not finite"
                      : "unsupported seasonal period (periods above 2048 are not supported)";
    switch (b->plan.model) {
    case M_ETS:
        if (b->plan.ets_spec_id >= 0) {
            std::snprintf(buf, sizeof buf, "Computation error: ETS model '%s' failed to fit: Computation error: Failed to fit ETS model: %s",
                          b->plan.ets_notation.c_str(), why);
            return buf;
        }
        // fallthrough
    default:
        std::snprintf(buf, sizeof buf, "Computation error: %s fit failed: %s", model_name(b->plan.model), why);
        return buf;
    }
}

void fitted_values_host(const double *y, size_t n, ModelType model, size_t period, double *f)
{
    // forecast.rs:2593-2643
    if (model == M_Naive) {
        f[0] = y[0];
        for (size_t i = 1; i < n; i++) f[i] = y[i - 1];
    } else if (model == M_SeasonalNaive) {
        size_t p = std::min(std::max<size_t>(period, 1), n);
        for (size_t i = 0; i < p; i++) f[i] = y[0];
        for (size_t i = p; i < n; i++) f[i] = y[i - p];
    } else {
        double level = y[0];
        f[0] = level;
        for (size_t i = 1; i < n; i++) { f[i] = level; level = 0.3 * y[i] + (1.0 - 0.3) * level; }
    }
}

// The candidate specs of a batch run on concurrent streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
// (default 4: the 25-spec grid then runs ~2x slower; 16 is the measured best).  The library does NOT touch the process
// environment: the host sets the variable before its first HIP call (lib.py and bench.py do).
bool device_ready(AnofoxError *err)
{
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) {
        set_error(err, INTERNAL_ERROR, "Internal error: no HIP device available (the MI355X backend has no CPU fallback)");
        return false;
    }
    return true;
}

// Route A (the shipped macro) calls anofox_ts_forecast once per group from every DuckDB worker thread
// (ts_forecast_scalar.cpp:298-523).  Creating a device batch costs ~30 ms of allocator / stream calls that serialise across
// threads -- far more than a single-series fit -- so idle single-series batches are kept in a small process-wide pool, keyed
// by the option block and the device: a call takes a matching batch (or creates one with room for 2x its length), re-packs
// it, runs, fetches and gives it back.  Entries are never destroyed at process exit (the HIP runtime may be gone by then).
struct PooledBatch { ForecastOptions opt; int device; AnofoxHipBatch *b; };
std::mutex g_pool_mu;
std::vector<PooledBatch> g_pool;
constexpr size_t POOL_MAX_IDLE = 32;

AnofoxHipBatch *pool_take(const ForecastOptions &o, size_t length, int device)
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (size_t i = 0; i < g_pool.size(); i++)
        if (g_pool[i].device == device && g_pool[i].b->t_max >= length && std::memcmp(&g_pool[i].opt, &o, sizeof o) == 0) {
            AnofoxHipBatch *b = g_pool[i].b;
            g_pool.erase(g_pool.begin() + (long)i);
            return b;                                   // (the caller attaches a stream set again)
        }
    return nullptr;
}

void pool_give(const ForecastOptions &o, int device, AnofoxHipBatch *b)
{
    // a parked batch keeps its device blocks but not its 33 streams: an early single-series call would otherwise sit on the
    // process's first (favourably mapped, high-priority) stream set for as long as it stays in the pool
    batch_detach_streams(b);
    AnofoxHipBatch *evict = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        if (g_pool.size() >= POOL_MAX_IDLE) { evict = g_pool.front().b; g_pool.erase(g_pool.begin()); }
        g_pool.push_back(PooledBatch{o, device, b});
    }
    if (evict) anofox_hip_batch_destroy(evict);
}

void pool_drain(std::vector<AnofoxHipBatch *> &out)
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (auto &e : g_pool) out.push_back(e.b);
    g_pool.clear();
}

} // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *anofox_fcst_version(void) { return "0.1.0-hip-gfx950"; }

int anofox_hip_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return -1;
    return cnt;
}

int anofox_hip_set_device(int device) { return hipSetDevice(device) == hipSuccess ? 0 : -1; }

void anofox_hip_release_caches(void)
{
    // parked single-series batches first (their blocks go back to the caches), then the caches themselves
    std::vector<AnofoxHipBatch *> parked;
    pool_drain(parked);
    for (AnofoxHipBatch *b : parked) anofox_hip_batch_destroy(b);
    dev_cache_release_all();
    pin_cache_release_all();
    stream_pool_release_all();
}

bool anofox_hip_set_default_arima_method(int method)
{
    if (method != ANOFOX_ARIMA_CSS && method != ANOFOX_ARIMA_CSS_ML) return false;
    g_default_arima_method.store(method);
    return true;
}

bool anofox_hip_batch_set_arima_method(AnofoxHipBatch *b, int method, AnofoxError *out_error)
{
    if (out_error) { out_error->code = SUCCESS; std::memset(out_error->message, 0, sizeof out_error->message); }
    if (!b) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (method != ANOFOX_ARIMA_CSS && method != ANOFOX_ARIMA_CSS_ML) {
        set_error(out_error, INVALID_INPUT, "Invalid input: unknown ARIMA estimation method (0 = CSS, 1 = CSS-ML)");
        return false;
    }
    b->arima_method = method;
    return true;
}

bool anofox_hip_batch_create(size_t n_series, size_t t_max, const ForecastOptions *options, AnofoxHipBatch **out_batch,
                             AnofoxError *out_error)
{
    if (out_error) { out_error->code = SUCCESS; std::memset(out_error->message, 0, sizeof out_error->message); }
    if (!options || !out_batch) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    Plan plan;
    if (!make_plan(options, plan, out_error)) return false;
    if (!device_ready(out_error)) return false;
    AnofoxHipBatch *b = new AnofoxHipBatch();
    try {
        HIPCHECK(hipGetDevice(&b->dev));
        b->n = n_series;
        b->t_max = t_max;
        b->ld = (n_series + 63) / 64 * 64;
        if (b->ld == 0) b->ld = 64;
        // the streamed loads address a block of <= 32 rows with 32-bit offsets (buffer loads): 33 rows must stay under 4 GiB
        if ((double)b->ld * 8.0 * 33.0 >= 4294967296.0) throw HipFail{"batch too wide for one launch (more than ~16 million series): shard it"};
        b->h = options->horizon;
        b->opt = *options;
        b->plan = plan;
        b->tun = Tunables::from_env();
        if (b->tun.seq_rounds >= 0) { b->seq_rounds_env = b->tun.seq_rounds; b->seq_rounds = b->seq_rounds_env; }
        // dense re-gather of the running problems between rounds (up to 2x on a large batch) costs one block copy per
        // candidate spec (sized below)
        {
            // one block per candidate spec THAT HAS ANYTHING TO FIT, allocated by the first run that needs it (launch_fit_slots): together
            // at most 55 % of the device (158 GB of the MI355X's 288).  The 1M x 1,024 stress configuration on ONE GPU: 6 admissible specs
            // on intermittent counts -> six full blocks of 8.2 GB; with all 25 specs live, blocks of 772k of the 1M columns (1.34 s per
            // step against 2.75 s without any gather, which is what the old all-or-nothing 96 GiB rule gave)
            size_t free_b = 0, total_b = 0;
            b->gather_budget = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? std::max(96.0 * 1073741824.0, 0.55 * (double)total_b) : 96.0 * 1073741824.0;
            b->use_gather = true;
        }
        if (b->tun.gather >= 0) b->use_gather = b->tun.gather != 0;
        b->spec_below = b->tun.spec_below; b->spec_below_md = b->tun.spec_below_md;
        b->spec2_below = b->tun.spec2_below; b->spec2_below_md = b->tun.spec2_below_md;
        b->arima_method = arima_method_in_force();
        alloc_common(b);
    } catch (const HipFail &f) {
        report_hip_failure(out_error, f);
        free_batch_buffers(b);
        delete b;
        return false;
    } catch (const std::exception &e) {
        set_error(out_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        free_batch_buffers(b);
        delete b;
        return false;
    }
    *out_batch = b;
    return true;
}

void anofox_hip_batch_destroy(AnofoxHipBatch *b)
{
    if (!b) return;
    DeviceGuard guard(b->dev);
    // wait for this batch's own work only (other batches may be running from other host threads)
    const bool timing = b->tun.timing;
    const auto t0 = std::chrono::steady_clock::now();
    // (a wait on an idle stream is not free either: with more streams than hardware queues it queues behind whatever other
    //  batches run on the shared queue -- 200-550 ms per destroy measured beside two running batches -- so a batch whose last run
    //  has been waited for skips them: its auxiliary streams had joined the run's stream before that wait returned)
    if (!b->quiesced) {
        if (b->last_stream) (void)hipStreamSynchronize(b->last_stream);
        if (b->own_stream) (void)hipStreamSynchronize(b->own_stream);
        for (auto &q : b->aux) if (q) (void)hipStreamSynchronize(q);
    }
    const auto t1 = std::chrono::steady_clock::now();
    const size_t n = b->n;
    free_batch_buffers(b);
    delete b;
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        if (std::chrono::duration<double, std::milli>(t2 - t0).count() > 20.0)
            std::fprintf(stderr, "[anofox-hip] destroy of a batch of %zu: stream waits %.1f ms, buffers %.1f ms\n", n,
                         std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
}

size_t anofox_hip_batch_ld(const AnofoxHipBatch *b) { return b ? b->ld : 0; }
size_t anofox_hip_batch_n_series(const AnofoxHipBatch *b) { return b ? b->n : 0; }

bool anofox_hip_batch_periods(const AnofoxHipBatch *b, int32_t *out_periods)
{
    if (!b || !out_periods || !b->has_block || b->h_period.size() < b->n) return false;
    for (size_t s = 0; s < b->n; s++) out_periods[s] = b->h_period[s];
    return true;
}

bool anofox_hip_batch_set_fixed_params(AnofoxHipBatch *b, double alpha, double beta, double gamma, double phi, AnofoxError *out_error)
{
    if (out_error) { out_error->code = SUCCESS; std::memset(out_error->message, 0, sizeof out_error->message); }
    if (!b) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (!(b->plan.model == M_ETS && b->plan.ets_spec_id >= 0)) {
        set_error(out_error, INVALID_INPUT, "Invalid input: fixed smoothing parameters need model 'ETS' with an explicit ets_model");
        return false;
    }
    const int id = b->plan.ets_spec_id;
    const bool has_trend = spec_trend_idx(id) != 0, has_season = spec_season(id) != 0;
    const bool damped = spec_trend_idx(id) == 2 || spec_trend_idx(id) == 4;
    // the model's own parameters: 0 < alpha < 1, 0 <= beta <= alpha, 0 <= gamma <= 1 - alpha, 0 < phi <= 1
    bool ok = alpha > 0.0 && alpha < 1.0;
    if (has_trend) ok = ok && beta >= 0.0 && beta <= alpha;
    if (has_season) ok = ok && gamma >= 0.0 && gamma <= 1.0 - alpha;
    if (damped) ok = ok && phi > 0.0 && phi <= 1.0;
    if (!ok) {
        set_error(out_error, INVALID_INPUT,
                  "Invalid input: smoothing parameters out of range (0 < alpha < 1, 0 <= beta <= alpha, 0 <= gamma <= 1 - alpha, 0 < phi <= 1)");
        return false;
    }
    // optimiser coordinates of the recursion (ets_device.hpp ets_unpack): beta = alpha beta*, gamma = gamma* (1 - alpha)
    b->fixed_x[0] = alpha;
    b->fixed_x[1] = has_trend ? beta / alpha : 0.0;
    b->fixed_x[2] = has_season ? gamma / (1.0 - alpha) : 0.0;
    b->fixed_x[3] = damped ? phi : 1.0;
    b->fixed_params = true;
    return true;
}

bool anofox_hip_batch_pack_host(AnofoxHipBatch *b, const double *const *values, const uint64_t *const *validity,
                                const size_t *lengths, AnofoxError *out_error)
{
    if (!b || !values || !lengths) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    DeviceGuard guard(b->dev);
    try {
        const size_t n = b->n, ld = b->ld, T = std::max<size_t>(b->t_max, 1);
        for (size_t s = 0; s < n; s++)
            if (lengths[s] > b->t_max) throw HipFail{"series longer than the plan's t_max"};
        // The packer is the host half of the batch entry and easily costs more than the fit: it runs on all host threads, 64 series
        // (one 512-byte row segment of the time-major block) per tile, into pinned staging.  The block goes over in column chunks
        // of ~512 MB through TWO staging buffers (kept by the batch): while one chunk is copied -- a pitched copy into its columns of
        // the device block, on the batch's own stream (the legacy default stream of a synchronous hipMemcpy is shared by every host
        // thread of the process) -- the packer threads fill the other, and no staging block of the size of the batch is ever pinned
        // (8.2 GB for the 1M x 1,024 configuration).
        constexpr size_t TILE = 64;
        const size_t n_tiles = ld / TILE;
        // (512 MB chunks: the M5 block -- 467 MB -- stays one contiguous copy, which measured faster than four pitched ones: 13 against 20 ms)
        const size_t chunk_tiles = std::min(n_tiles, std::max<size_t>(1, (size_t)(536870912.0 / (8.0 * (double)T)) / TILE));
        const size_t chunk_cols = chunk_tiles * TILE;
        const size_t n_chunks = (n_tiles + chunk_tiles - 1) / chunk_tiles;
        const size_t n_bufs = n_chunks > 1 ? 2 : 1;
        const auto tp0 = std::chrono::steady_clock::now();
        double fill_ms = 0.0;
        if (b->h_stage_elems < n_bufs * T * chunk_cols) {
            pin_free(b->h_stage);
            b->h_stage = nullptr; b->h_stage_elems = 0;
            b->h_stage = (double *)pin_alloc_bytes(n_bufs * T * chunk_cols * sizeof(double));
            b->h_stage_elems = n_bufs * T * chunk_cols;
        }
        b->h_len.assign(n, 0);
        b->h_period.assign(n, 1);
        const bool keep = b->opt.include_fitted || b->opt.include_residuals;
        if (keep) {
            b->h_clean_off.assign(n + 1, 0);
            for (size_t s = 0; s < n; s++) b->h_clean_off[s + 1] = b->h_clean_off[s] + lengths[s];
            b->h_clean.assign(b->h_clean_off[n], 0.0);
        }
        const bool detect = b->opt.auto_detect_seasonality && b->opt.seasonal_period == 0;
        // tiles [tile0, tile1) of the chunk that starts at tile `base` -> stage (row stride `cols` doubles)
        auto do_tiles = [&](double *stage, size_t cols, size_t base, size_t tile0, size_t tile1) {
            std::vector<double> clean(TILE * T);
            for (size_t tile = tile0; tile < tile1; tile++) {
                const size_t s0 = tile * TILE;
                size_t len_of[TILE];
                for (size_t j = 0; j < TILE; j++) {
                    const size_t s = s0 + j;
                    const size_t len = s < n ? lengths[s] : 0;
                    len_of[j] = len;
                    double *c = clean.data() + j * T;
                    if (len) fill_nulls_interpolate(values[s], validity ? validity[s] : nullptr, len, c);
                    if (s < n) {
                        b->h_len[s] = (int32_t)len;
                        b->h_period[s] = (!detect && b->opt.seasonal_period > 0) ? b->opt.seasonal_period : 1;      // (detected periods: below, on the device)
                        if (keep && len) std::memcpy(b->h_clean.data() + b->h_clean_off[s], c, len * sizeof(double));
                    }
                }
                for (size_t t = 0; t < T; t++) {
                    double *row = stage + t * cols + (tile - base) * TILE;
                    for (size_t j = 0; j < TILE; j++) row[j] = t < len_of[j] ? clean[j * T + t] : 0.0;
                }
            }
        };
        const auto tp1 = std::chrono::steady_clock::now();
        unsigned n_thr = std::thread::hardware_concurrency();
        if (tl_host_thread_share > 1) n_thr = std::max(1u, n_thr / tl_host_thread_share);     // one of several device shards packing side by side
        if (b->tun.pack_threads > 0) n_thr = (unsigned)b->tun.pack_threads;
        n_thr = std::max(1u, std::min(n_thr, 32u));
        if (!b->d_y || !b->owns_y) { b->d_y = dalloc<double>(T * ld); b->owns_y = true; }
        batch_attach_streams(b);
        hipEvent_t copied[2] = {nullptr, nullptr};
        struct EvGuard { hipEvent_t (&e)[2]; ~EvGuard() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } ev_guard{copied};
        for (size_t k = 0; k < n_bufs; k++) HIPCHECK(hipEventCreateWithFlags(&copied[k], hipEventDisableTiming));
        try {
            for (size_t ck = 0; ck < n_chunks; ck++) {
                const size_t base = ck * chunk_tiles, tiles = std::min(chunk_tiles, n_tiles - base), cols = tiles * TILE;
                double *stage = b->h_stage + (ck % n_bufs) * T * chunk_cols;
                if (ck >= n_bufs) HIPCHECK(hipEventSynchronize(copied[ck % n_bufs]));       // the copy that last read this buffer
                const unsigned thr = (unsigned)std::min<size_t>(n_thr, tiles);
                const auto tf0 = std::chrono::steady_clock::now();
                if (thr <= 1 || tiles < 8) do_tiles(stage, cols, base, base, base + tiles);
                else {
                    std::vector<std::string> fails(thr);
                    parallel_shares(thr, [&](unsigned k) {
                        try { do_tiles(stage, cols, base, base + tiles * k / thr, base + tiles * (k + 1) / thr); }
                        catch (const std::exception &e) { try { fails[k] = e.what(); } catch (...) { fails[k].assign(1, '?'); } }
                        catch (...) { try { fails[k] = "packer thread failed"; } catch (...) { fails[k].assign(1, '?'); } }
                    });
                    for (auto &f : fails) if (!f.empty()) throw HipFail{f};
                }
                fill_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count();
                HIPCHECK(hipMemcpy2DAsync(b->d_y + base * TILE, ld * sizeof(double), stage, cols * sizeof(double), cols * sizeof(double), T,
                                          hipMemcpyHostToDevice, b->own_stream));
                HIPCHECK(hipEventRecord(copied[ck % n_bufs], b->own_stream));
            }
            HIPCHECK(hipStreamSynchronize(b->own_stream));
        } catch (...) { (void)hipStreamSynchronize(b->own_stream); throw; }       // nothing may still read the staging buffers
        if (b->tun.timing) {
            const auto tp2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[anofox-hip] pack of %zu columns: staging + bookkeeping %.1f ms, %u threads filled %zu tiles in %.1f ms, device block + copies %.1f ms\n", ld,
                         std::chrono::duration<double, std::milli>(tp1 - tp0).count(), n_thr, n_tiles, fill_ms,
                         std::chrono::duration<double, std::milli>(tp2 - tp1).count() - fill_ms);
        }
        finalize_lengths(b);
        if (detect) batch_detect_periods(b);
        b->has_block = true;
    } catch (const HipFail &f) {
        report_hip_failure(out_error, f);
        return false;
    } catch (const std::exception &e) {
        set_error(out_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        return false;
    }
    return true;
}

bool anofox_hip_batch_set_device_block(AnofoxHipBatch *b, const void *d_y, size_t ld, const void *d_len, AnofoxError *out_error)
{
    if (!b || !d_y || !d_len) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (ld != b->ld) { set_error(out_error, INVALID_INPUT, "Invalid input: leading dimension must equal anofox_hip_batch_ld()"); return false; }
    DeviceGuard guard(b->dev);
    try {
        if (b->owns_y && b->d_y) { dev_free(b->d_y); }
        b->d_y = (double *)d_y;
        b->owns_y = false;
        b->h_len.assign(b->n, 0);
        HIPCHECK(hipMemcpy(b->h_len.data(), d_len, b->n * sizeof(int32_t), hipMemcpyDeviceToHost));
        b->h_period.assign(b->n, b->opt.seasonal_period > 0 ? b->opt.seasonal_period : 1);
        finalize_lengths(b);
        if (b->opt.auto_detect_seasonality && b->opt.seasonal_period == 0) {
            // the block holds no NULLs: nothing to interpolate.  The scan runs now, on the batch's own stream: whatever stream the
            // caller filled the block on must have finished (the run itself is ordered by the stream the caller passes to it)
            HIPCHECK(hipDeviceSynchronize());
            batch_detect_periods(b);
        }
        b->has_block = true;
    } catch (const HipFail &f) {
        report_hip_failure(out_error, f);
        return false;
    } catch (const std::exception &e) {
        set_error(out_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        return false;
    }
    return true;
}

bool anofox_hip_batch_run(AnofoxHipBatch *b, void *stream, AnofoxError *out_error)
{
    if (!b) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (!b->has_block) { set_error(out_error, INVALID_INPUT, "Invalid input: batch has no series block"); return false; }
    DeviceGuard guard(b->dev);
    try {
        run_batch(b, stream ? (hipStream_t)stream : b->own_stream);
    } catch (const HipFail &f) {
        report_hip_failure(out_error, f);
        return false;
    } catch (const std::exception &e) {
        set_error(out_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        return false;
    }
    return true;
}

bool anofox_hip_batch_stats(AnofoxHipBatch *b, AnofoxHipStats *out)
{
    if (!b || !out || !b->ran) return false;
    DeviceGuard guard(b->dev);
    std::memset(out, 0, sizeof *out);
    if (hipEventSynchronize(b->ev_stop) != hipSuccess) return false;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, b->ev_start, b->ev_stop) == hipSuccess) out->total_device_ms = ms;
    if (b->timed_fit && hipEventElapsedTime(&ms, b->ev_fit0, b->ev_fit1) == hipSuccess) out->fit_kernel_ms = ms;
    std::vector<int32_t> passes(b->n), evals(b->n);
    if (hipMemcpy(passes.data(), b->d_passes_total, b->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
    if (hipMemcpy(evals.data(), b->d_evals_total, b->n * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
    out->n_series = b->n;
    out->t_max = b->t_max;
    out->n_problems = b->n_problems;
    out->fit_kernel_launches = b->fit_launches;
    out->y_storage = (uint32_t)(b->insp_ok ? b->y_type : 0);
    for (size_t s = 0; s < b->n; s++) {
        uint64_t p = (uint64_t)std::max(passes[s], 0);
        out->total_passes += p;
        out->max_passes = std::max<uint64_t>(out->max_passes, p);
        out->total_evals += (uint64_t)std::max(evals[s], 0);
        out->algorithmic_bytes += 8ull * (uint64_t)b->h_len[s] * p + 24ull * (uint64_t)std::max(b->h, 0);
    }
    // the one-pass-per-iteration count of the same run (the spec slots of the last fit: iterations per problem, + 1 final pass each)
    if (b->insp_ok && b->d_iters_slots && b->d_status_slots && !b->fixed_params) {
        const size_t n_used = b->insp_spec.size(), ld = b->ld;
        std::vector<int32_t> it(n_used * ld), stt(n_used * ld);
        if (hipMemcpy(it.data(), b->d_iters_slots, it.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(stt.data(), b->d_status_slots, stt.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (size_t s = 0; s < b->n; s++) {
            uint64_t q = 0;
            for (size_t k = 0; k < n_used; k++)
                if (stt[k * ld + s] == anofox::FIT_OK) q += (uint64_t)std::max(it[k * ld + s], 0) + 1;
            out->total_iters += q;
            out->min_pass_bytes += 8ull * (uint64_t)b->h_len[s] * q + 24ull * (uint64_t)std::max(b->h, 0);
        }
    }
    return true;
}

bool anofox_hip_selftest_recip(uint64_t n_operands, uint64_t seed, uint64_t *out_mismatches, double *out_first_bad)
{
    if (!out_mismatches) return false;
    AnofoxError err;
    if (!device_ready(&err)) return false;
    double fb = 0.0;
    const unsigned long long r = anofox::recip_selftest(n_operands, seed, &fb, nullptr);
    if (r == ~0ull) return false;
    *out_mismatches = r;
    if (out_first_bad) *out_first_bad = fb;
    return true;
}

bool anofox_hip_batch_lane_stats(AnofoxHipBatch *b, AnofoxHipLaneStats *out, size_t struct_size)
{
    if (!b || !out || !b->ran || struct_size < sizeof(uint64_t)) return false;
    DeviceGuard guard(b->dev);
    AnofoxHipLaneStats r;
    std::memset(&r, 0, sizeof r);
    r.struct_size = sizeof r;
    if (hipEventSynchronize(b->ev_stop) != hipSuccess) return false;
    if (b->d_wave_trace && !b->tun.wave_trace.empty()) {
        unsigned long long head[4] = {0, 0, 0, 0};
        if (hipMemcpy(head, b->d_wave_trace, sizeof head, hipMemcpyDeviceToHost) != hipSuccess) return false;
        const size_t used = (size_t)std::min(head[0], head[1]);
        std::vector<unsigned long long> rec(4 * used);
        if (used && hipMemcpy(rec.data(), b->d_wave_trace + 4, rec.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (FILE *f = std::fopen(b->tun.wave_trace.c_str(), "wb")) {
            std::fwrite(head, sizeof head, 1, f);
            if (used) std::fwrite(rec.data(), sizeof(unsigned long long), rec.size(), f);
            std::fclose(f);
        }
    }
    if (b->insp_ok && b->d_lane_stats && !b->fixed_params) {
        std::vector<unsigned long long> c(2 * (size_t)b->n_slots_cap);
        if (hipMemcpy(c.data(), b->d_lane_stats, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return false;
        // insp_spec[oi] = spec id in launch order; the counters are indexed by the slot k the spec was given (order[oi] = k): rebuild k
        // from h_slot_spec (slot -> spec id)
        for (size_t k = 0; k < b->h_slot_spec.size() && k < (size_t)b->n_slots_cap; k++) {
            const int id = b->h_slot_spec[k];
            const int cls = !spec_has_mult(id) ? 0 : (spec_trend_idx(id) == 4 ? 2 : 1);
            r.wave_passes[cls] += c[2 * k];
            r.live_lane_passes[cls] += c[2 * k + 1];
            if (k < 30) { r.slot_spec_id[k] = id; r.slot_wave_passes[k] = c[2 * k]; r.slot_live_lane_passes[k] = c[2 * k + 1]; r.n_slots = (uint32_t)(k + 1); }
        }
    }
    // the caller's struct may be older (smaller) or newer (larger) than this library's: copy what both know, report what was written
    const size_t nbytes = std::min(struct_size, sizeof r);
    r.struct_size = nbytes;
    std::memcpy(out, &r, nbytes);
    return true;
}

bool anofox_hip_batch_device_results(AnofoxHipBatch *b, void **d_yhat, void **d_lower, void **d_upper, void **d_model_code,
                                     void **d_status)
{
    if (!b) return false;
    if (d_yhat) *d_yhat = b->d_yhat;
    if (d_lower) *d_lower = b->d_lo;
    if (d_upper) *d_upper = b->d_hi;
    if (d_model_code) *d_model_code = b->d_model_code;
    if (d_status) *d_status = b->d_status;
    return true;
}

void anofox_hip_model_name(const ForecastOptions *options, int32_t model_code, char out_name[64])
{
    out_name[0] = 0;
    if (model_code >= 1000000) { auto_arima_name(model_code, options ? options->seasonal_period : 1, out_name); return; }
    if (model_code >= 100 && model_code < 130) { auto_ets_name(model_code - 100, out_name); return; }
    Plan p;
    AnofoxError e;
    if (options && make_plan(options, p, &e)) {
        std::snprintf(out_name, 64, "%s", p.static_name.c_str());
    }
}

void anofox_hip_batch_model_name(const AnofoxHipBatch *b, size_t series, int32_t model_code, char out_name[64])
{
    out_name[0] = 0;
    if (!b) return;
    // the period the series was fitted with (given, or detected on the resident block): what anofox_hip_batch_fetch names it with
    if (model_code >= 1000000) {
        const int period = (series < b->h_period.size()) ? b->h_period[series] : (b->opt.seasonal_period > 0 ? b->opt.seasonal_period : 1);
        auto_arima_name(model_code, period, out_name);
        return;
    }
    anofox_hip_model_name(&b->opt, model_code, out_name);
}

bool anofox_hip_batch_fetch(AnofoxHipBatch *b, ForecastResult *out_results, AnofoxError *out_errors)
{
    if (!b || !out_results || !b->ran) return false;
    DeviceGuard guard(b->dev);
    const size_t n = b->n, h = (size_t)std::max(b->h, 0);
    if (hipStreamSynchronize(b->last_stream) != hipSuccess) return false;
    b->quiesced = true;
    // results come back through ONE pinned block (a copy into pageable memory is staged by the runtime at a fraction of the link
    // speed: 672 MB of forecasts for 1M series), on the stream of the run
    const size_t nh = n * h;
    struct PinBlock { void *p = nullptr; ~PinBlock() { pin_free(p); } } pin;
    try { pin.p = pin_alloc_bytes(std::max<size_t>(3 * nh * sizeof(double) + 3 * n * sizeof(int32_t), 64)); }
    catch (...) { return false; }
    double *const yhat = (double *)pin.p, *const lo = yhat + nh, *const hi = lo + nh;
    int32_t *const status = (int32_t *)(hi + nh), *const code = status + n, *const detail = code + n;
    bool ok = true;
    hipStream_t cs = b->last_stream;
    if (nh) {
        ok &= hipMemcpyAsync(yhat, b->d_yhat, nh * sizeof(double), hipMemcpyDeviceToHost, cs) == hipSuccess;
        ok &= hipMemcpyAsync(lo, b->d_lo, nh * sizeof(double), hipMemcpyDeviceToHost, cs) == hipSuccess;
        ok &= hipMemcpyAsync(hi, b->d_hi, nh * sizeof(double), hipMemcpyDeviceToHost, cs) == hipSuccess;
    }
    ok &= hipMemcpyAsync(status, b->d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost, cs) == hipSuccess;
    ok &= hipMemcpyAsync(code, b->d_model_code, n * sizeof(int32_t), hipMemcpyDeviceToHost, cs) == hipSuccess;
    ok &= hipMemcpyAsync(detail, b->d_detail, n * sizeof(int32_t), hipMemcpyDeviceToHost, cs) == hipSuccess;
    ok &= hipStreamSynchronize(cs) == hipSuccess;
    if (!ok) return false;
    // the per-series result records (three allocations each, the reference's ownership contract): on several host threads for a large
    // batch -- 1M series spent 0.47 s here on one
    auto do_range = [&](size_t s_lo, size_t s_hi) {
    for (size_t s = s_lo; s < s_hi; s++) {
        ForecastResult &r = out_results[s];
        std::memset(&r, 0, sizeof r);
        if (out_errors) { out_errors[s].code = SUCCESS; std::memset(out_errors[s].message, 0, sizeof out_errors[s].message); }
        if (status[s] == STATUS_NOT_COMPUTED) status[s] = INTERNAL_ERROR;
        if (status[s] != 0) {
            if (out_errors) set_error(&out_errors[s], status[s], series_error_message(b, s, status[s], detail[s]));
            continue;
        }
        r.n_forecasts = h;
        if (h) {
            r.point_forecasts = (double *)std::malloc(h * sizeof(double));
            r.lower_bounds = (double *)std::malloc(h * sizeof(double));
            r.upper_bounds = (double *)std::malloc(h * sizeof(double));
            if (!r.point_forecasts || !r.lower_bounds || !r.upper_bounds) {
                std::free(r.point_forecasts); std::free(r.lower_bounds); std::free(r.upper_bounds);
                std::memset(&r, 0, sizeof r);
                if (out_errors) set_error(&out_errors[s], ALLOCATION_ERROR, "Failed to allocate point forecasts");
                continue;
            }
            std::memcpy(r.point_forecasts, &yhat[s * h], h * sizeof(double));
            std::memcpy(r.lower_bounds, &lo[s * h], h * sizeof(double));
            std::memcpy(r.upper_bounds, &hi[s * h], h * sizeof(double));
        }
        if (code[s] >= 1000000) auto_arima_name(code[s], b->h_period[s], r.model_name);
        else if (code[s] >= 100) auto_ets_name(code[s] - 100, r.model_name);
        else std::snprintf(r.model_name, 64, "%s", b->plan.static_name.c_str());
        r.aic = std::nan(""); r.bic = std::nan(""); r.mse = std::nan("");
        if ((b->opt.include_fitted || b->opt.include_residuals) && !b->h_clean_off.empty()) {
            const size_t len = (size_t)b->h_len[s];
            const double *y = b->h_clean.data() + b->h_clean_off[s];
            double *f = (double *)std::malloc(len * sizeof(double));
            fitted_values_host(y, len, b->plan.model, (size_t)b->h_period[s], f);
            double sse = 0.0;
            for (size_t i = 0; i < len; i++) { double d = y[i] - f[i]; sse += d * d; }
            r.mse = sse / (double)len;
            if (b->opt.include_residuals) {
                r.residuals = (double *)std::malloc(len * sizeof(double));
                for (size_t i = 0; i < len; i++) r.residuals[i] = y[i] - f[i];
            }
            if (b->opt.include_fitted) { r.fitted_values = f; r.n_fitted = len; }
            else std::free(f);
        }
    }
    };
    unsigned n_thr = (unsigned)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), 16, n / 8192});
    if (tl_host_thread_share > 1) n_thr = std::max(1u, n_thr / tl_host_thread_share);
    if (n_thr <= 1) do_range(0, n);
    else {
        std::atomic<bool> failed{false};
        parallel_shares(n_thr, [&](unsigned k) { try { do_range(n * k / n_thr, n * (k + 1) / n_thr); } catch (...) { failed = true; } });
        if (failed) return false;
    }
    return true;
}

void anofox_free_forecast_result(ForecastResult *r)
{
    if (!r) return;
    std::free(r->point_forecasts); r->point_forecasts = nullptr;
    std::free(r->lower_bounds); r->lower_bounds = nullptr;
    std::free(r->upper_bounds); r->upper_bounds = nullptr;
    std::free(r->fitted_values); r->fitted_values = nullptr;
    std::free(r->residuals); r->residuals = nullptr;
}

bool anofox_hip_batch_inspect(AnofoxHipBatch *b, AnofoxHipInspection *out, double *fitted, double *seasonal, size_t seasonal_stride,
                              AnofoxError *out_error)
{
    if (!b || !out) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (!b->ran) { set_error(out_error, INVALID_INPUT, "Invalid input: the batch has not been run"); return false; }
    DeviceGuard guard(b->dev);
    const size_t n = b->n, ld = b->ld, T = std::max<size_t>(b->t_max, 1);
    try {
        HIPCHECK(hipStreamSynchronize(b->last_stream));
        b->quiesced = false;             // the inspection passes below run on the auxiliary streams
        std::vector<int32_t> code(n), status(n);
        HIPCHECK(hipMemcpy(code.data(), b->d_model_code, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(status.data(), b->d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t s = 0; s < n; s++) {
            AnofoxHipInspection &o = out[s];
            o.model_code = code[s]; o.status = status[s]; o.seasonal_period = b->h_period.empty() ? 1 : b->h_period[s];
            o.alpha = o.beta = o.gamma = o.phi = o.aic = o.aicc = o.bic = o.sse = o.level = o.trend = std::nan("");
        }
        if (fitted) for (size_t i = 0; i < n * T; i++) fitted[i] = std::nan("");
        const bool ets = (b->plan.model == M_AutoETS || (b->plan.model == M_ETS && b->plan.ets_spec_id >= 0)) && b->insp_ok;
        if (ets) {
            for (size_t s = 1; s < n; s++)
                if (b->h_period[s] != b->h_period[0]) throw HipFail{"inspection needs one seasonal period for the whole batch"};
            const int m = b->insp_m;
            const size_t rows = (size_t)(2 + std::max(m, 1));
            double *d_fit = dalloc<double>(T * ld), *d_states = dalloc<double>(rows * ld), *d_info = dalloc<double>(8 * ld);
            std::vector<double> nanv(std::max(T, rows) * ld, std::nan(""));
            HIPCHECK(hipMemcpy(d_fit, nanv.data(), T * ld * sizeof(double), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(d_states, nanv.data(), rows * ld * sizeof(double), hipMemcpyHostToDevice));
            HIPCHECK(hipMemcpy(d_info, nanv.data(), 8 * ld * sizeof(double), hipMemcpyHostToDevice));
            hipStream_t st = b->last_stream;
            for (size_t oi = 0; oi < b->insp_args.size(); oi++) {
                FitArgs a = b->insp_args[oi];
                a.insp_sel = b->d_model_code;
                a.insp_code = b->plan.model == M_AutoETS ? 100 + b->insp_spec[oi] : 0;
                a.insp_fitted = d_fit; a.insp_states = d_states; a.insp_info = d_info;
                b->insp_fns[oi].final(a, st);
            }
            HIPCHECK(hipStreamSynchronize(st));
            std::vector<double> info(8 * ld), states(rows * ld), fit;
            HIPCHECK(hipMemcpy(info.data(), d_info, 8 * ld * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(states.data(), d_states, rows * ld * sizeof(double), hipMemcpyDeviceToHost));
            if (fitted) { fit.resize(T * ld); HIPCHECK(hipMemcpy(fit.data(), d_fit, T * ld * sizeof(double), hipMemcpyDeviceToHost)); }
            dev_free(d_fit, true); dev_free(d_states, true); dev_free(d_info, true);       // the stream was synchronised above
            for (size_t s = 0; s < n; s++) {
                AnofoxHipInspection &o = out[s];
                o.alpha = info[0 * ld + s]; o.beta = info[1 * ld + s]; o.gamma = info[2 * ld + s]; o.phi = info[3 * ld + s];
                o.aic = info[4 * ld + s]; o.aicc = info[5 * ld + s]; o.bic = info[6 * ld + s]; o.sse = info[7 * ld + s];
                o.level = states[0 * ld + s]; o.trend = states[1 * ld + s];
                if (seasonal) for (int j = 0; j < m && (size_t)j < seasonal_stride; j++) seasonal[s * seasonal_stride + j] = states[(size_t)(2 + j) * ld + s];
                if (fitted) { const size_t len = (size_t)std::max(b->h_len[s], 0); for (size_t t = 0; t < len; t++) fitted[s * T + t] = fit[t * ld + s]; }
            }
        } else if (b->plan.model == M_AutoARIMA) {
            std::vector<double> aicc(n);
            std::vector<int32_t> ord(5 * ld), wlen(n);
            HIPCHECK(hipMemcpy(aicc.data(), b->ar_aicc, n * sizeof(double), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(ord.data(), b->ar_order, 5 * ld * sizeof(int32_t), hipMemcpyDeviceToHost));
            HIPCHECK(hipMemcpy(wlen.data(), b->ar_wlen, n * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (size_t s = 0; s < n; s++)
                if (code[s] >= 1000000) {
                    const double k = 1.0 + ord[0 * ld + s] + ord[1 * ld + s] + ord[2 * ld + s] + ord[3 * ld + s] + ord[4 * ld + s];
                    const double nn = (double)wlen[s];
                    out[s].aicc = aicc[s];
                    out[s].aic = aicc[s] - 2.0 * k * (k + 1.0) / (nn - k - 1.0);
                    out[s].bic = out[s].aic - 2.0 * k + k * std::log(nn);
                    out[s].reserved = ord[4 * ld + s];          // 1: the model has a constant
                }
        }
    } catch (const HipFail &f) {
        report_hip_failure(out_error, f);
        return false;
    } catch (const std::exception &e) {
        set_error(out_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        return false;
    }
    return true;
}

// one plan, one device batch: create, pack, run, fetch (the whole of the batch entry unless periods are auto-detected)
static bool forecast_batch_uniform(const double *const *values, const uint64_t *const *validity, const size_t *lengths, size_t n_series,
                                   const ForecastOptions *options, const int *horizons, ForecastResult *out_results,
                                   AnofoxError *out_errors, AnofoxError *out_batch_error)
{
    ForecastOptions opt = *options;
    int hmax = options->horizon;
    if (horizons) for (size_t s = 0; s < n_series; s++) hmax = std::max(hmax, horizons[s]);
    opt.horizon = hmax;
    size_t t_max = 0;
    for (size_t s = 0; s < n_series; s++) t_max = std::max(t_max, lengths[s]);
    AnofoxHipBatch *b = nullptr;
    AnofoxError e;
    const auto t0 = std::chrono::steady_clock::now();
    if (!anofox_hip_batch_create(n_series, t_max, &opt, &b, &e)) {
        if (e.code == INVALID_INPUT) {
            // an option block the core rejects (seasonal_period on a non-seasonal model, bad ETS notation or pool) is a
            // PER-SERIES error there, raised after the length checks (forecast.rs:516-565): a series that is too short
            // reports InsufficientData first, and a batch made only of such series does not abort the statement
            if (out_errors)
                for (size_t s = 0; s < n_series; s++) {
                    if (lengths[s] == 0) set_error(&out_errors[s], INSUFFICIENT_DATA, "Insufficient data: need at least 1 observations, got 0");
                    else if (lengths[s] < 3)
                        set_error(&out_errors[s], INSUFFICIENT_DATA, "Insufficient data: need at least 3 observations, got " + std::to_string(lengths[s]));
                    else out_errors[s] = e;
                }
            return true;
        }
        if (out_batch_error) *out_batch_error = e;
        if (out_errors) for (size_t s = 0; s < n_series; s++) out_errors[s] = e;
        return false;
    }
    const bool timing = b->tun.timing;     // ANOFOX_HIP_TIMING: phase times of the batch entry on stderr
    const auto t1 = std::chrono::steady_clock::now();
    bool ok = anofox_hip_batch_pack_host(b, values, validity, lengths, &e);
    const auto t2 = std::chrono::steady_clock::now();
    ok = ok && anofox_hip_batch_run(b, nullptr, &e);
    if (ok && timing) (void)hipStreamSynchronize(b->last_stream);
    const auto t3 = std::chrono::steady_clock::now();
    ok = ok && anofox_hip_batch_fetch(b, out_results, out_errors);
    const auto t4 = std::chrono::steady_clock::now();
    if (!ok) {
        if (e.code == SUCCESS) set_error(&e, INTERNAL_ERROR, "Internal error: device batch failed");
        if (out_batch_error) *out_batch_error = e;
    } else if (horizons) {
        // per-series horizon: forecasts are prefix-consistent, so truncate (ts_cv_forecast_native.cpp:676-677)
        for (size_t s = 0; s < n_series; s++)
            if (out_results[s].point_forecasts && horizons[s] >= 0 && (size_t)horizons[s] < out_results[s].n_forecasts) {
                out_results[s].n_forecasts = (size_t)horizons[s];
                if (horizons[s] == 0) {
                    std::free(out_results[s].point_forecasts); std::free(out_results[s].lower_bounds); std::free(out_results[s].upper_bounds);
                    out_results[s].point_forecasts = out_results[s].lower_bounds = out_results[s].upper_bounds = nullptr;
                }
            }
    }
    anofox_hip_batch_destroy(b);
    if (timing) {
        auto ms = [](auto a, auto z) { return std::chrono::duration<double, std::milli>(z - a).count(); };
        std::fprintf(stderr, "[anofox-hip] batch of %zu: create %.1f ms, pack + H2D %.1f ms, run %.1f ms, fetch %.1f ms, destroy %.1f ms\n", n_series,
                     ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, std::chrono::steady_clock::now()));
    }
    return ok;
}

// the batch entry on the calling thread's current device
static bool forecast_batch_one_device(const double *const *values, const uint64_t *const *validity, const size_t *lengths, size_t n_series,
                                      const ForecastOptions *options, const int *horizons, ForecastResult *out_results,
                                      AnofoxError *out_errors, AnofoxError *out_batch_error)
{
    // Per-series period detection (params := MAP{}): every distinct period a model uses is its own run of the pipeline,
    // latency bound and tiny (~140 distinct periods per thousand M5-like series).  Detect the periods here, split the batch
    // by used period and run the parts side by side from a few host threads, each part as an ordinary batch that names
    // its period -- forecast() sees the same period either way (forecast.rs:527-539), so the results are unchanged.
    Plan plan;
    AnofoxError pe;
    const int method_in_force = arima_method_in_force();       // ... for the worker threads below
    const bool detect = options->auto_detect_seasonality && options->seasonal_period == 0 && n_series >= 2;
    if (!detect || !make_plan(options, plan, &pe))
        return forecast_batch_uniform(values, validity, lengths, n_series, options, horizons, out_results, out_errors, out_batch_error);
    try {
        EvictionDeferral park_evictions;                 // several batches run side by side below
        auto used_period = [&](int period) {
            switch (plan.model) {
            case M_Naive: case M_RandomWalkDrift: case M_ARIMA: case M_SES: case M_SESOptimized: case M_Holt: return 1;
            case M_AutoARIMA: return period > ETS_MAX_PERIOD ? ETS_MAX_PERIOD + 1 : (period > 1 ? period : 1);      // (forecast.rs:528-537, 1448-1452: any detected period is seasonal)
            default: return period;
            }
        };
        std::vector<int> period(n_series, 1);
        {
            const auto td0 = std::chrono::steady_clock::now();
            const std::vector<int> raw = detect_periods_host_series(values, validity, lengths, n_series);
            for (size_t s = 0; s < n_series; s++) period[s] = lengths[s] < 3 ? 1 : used_period(raw[s] > 0 ? raw[s] : 1);
            if (Tunables::from_env().timing)
                std::fprintf(stderr, "[anofox-hip] period detection of %zu series: %.1f ms\n", n_series,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count());
        }
        std::map<int, std::vector<size_t>> groups;
        for (size_t s = 0; s < n_series; s++) groups[period[s]].push_back(s);
        std::vector<std::pair<int, std::vector<size_t>>> parts(groups.begin(), groups.end());
        std::sort(parts.begin(), parts.end(), [](const auto &x, const auto &y) { return x.second.size() > y.second.size(); });
        std::mutex err_mu;
        bool all_ok = true;
        AnofoxError first_err;
        first_err.code = SUCCESS;
        auto run_part = [&](const std::pair<int, std::vector<size_t>> &part) {
            const std::vector<size_t> &idx = part.second;
            const size_t k = idx.size();
            std::vector<const double *> v(k);
            std::vector<const uint64_t *> m(k);
            std::vector<size_t> len(k);
            std::vector<int> hz(k);
            for (size_t j = 0; j < k; j++) {
                v[j] = values[idx[j]]; m[j] = validity ? validity[idx[j]] : nullptr; len[j] = lengths[idx[j]];
                if (horizons) hz[j] = horizons[idx[j]];
            }
            ForecastOptions o = *options;
            o.auto_detect_seasonality = false;
            o.seasonal_period = part.first > 1 ? part.first : 0;           // 0 with detection off: period 1
            std::vector<ForecastResult> res(k);
            std::vector<AnofoxError> errs(k);
            for (auto &e : errs) { e.code = SUCCESS; e.message[0] = 0; }
            AnofoxError be;
            const bool ok = forecast_batch_uniform(v.data(), validity ? m.data() : nullptr, len.data(), k, &o, horizons ? hz.data() : nullptr,
                                                   res.data(), errs.data(), &be);
            for (size_t j = 0; j < k; j++) {
                out_results[idx[j]] = res[j];
                if (out_errors) out_errors[idx[j]] = errs[j];
            }
            if (!ok) {
                std::lock_guard<std::mutex> lock(err_mu);
                if (all_ok) first_err = be;
                all_ok = false;
            }
        };
        // AutoETS: the small parts with periods 2..48 run as ONE batch whose columns are grouped by period in blocks of 64 (the
        // kernels read the period per block): 138 tiny batches x 25 spec chains on 16 hardware queues were latency bound end to end
        const Tunables tun = Tunables::from_env();
        // AutoARIMA (round 4): every detected period is seasonal now (forecast.rs:528-537, 1448-1452) -- ~400 distinct ones on the M5
        // shape, each a latency-bound batch of its own (13 sweeps with a host round trip apiece: 8.2 s for the 30,490 series).  The
        // periods above 24 (ring of seasonal lags in the HBM scratch) run as ONE batch with the period per series (arima.hip pass
        // variant 6); the periods up to 24 keep their compile-time / LDS-ring kernels, one batch each.
        const bool arima_merge = tun.merge_periods && plan.model == M_AutoARIMA;
        const bool do_merge = arima_merge || (tun.merge_periods && (plan.model == M_AutoETS || plan.model == M_HoltWinters || plan.model == M_SeasonalES || plan.model == M_SeasonalESOptimized ||
                                                (plan.model == M_ETS && plan.ets_spec_id >= 0 && spec_season(plan.ets_spec_id) != 0)));
            // one merged batch per ring class (seasonal ring in LDS up to 64, in an HBM scratch above; the scratch is sized by the
            // class's largest period, hence a few classes)
            using Part = std::pair<int, std::vector<size_t>>;
            std::mutex merged_mu;
            auto run_merged = [&](std::vector<Part> &take) -> bool {
                std::sort(take.begin(), take.end(), [](const Part &x, const Part &y) { return x.first < y.first; });
                static const double dummy_v = 0.0;
                std::vector<const double *> v;
                std::vector<const uint64_t *> mk;
                std::vector<size_t> len, src;                       // src: index of the caller's series, or SIZE_MAX for padding
                std::vector<int32_t> mcol;
                size_t t_cap = 0;
                int h_cap = options->horizon, m_max = 0;
                for (const auto &part : take) {
                    for (size_t idx : part.second) {
                        v.push_back(values[idx]); mk.push_back(validity ? validity[idx] : nullptr); len.push_back(lengths[idx]); src.push_back(idx);
                        mcol.push_back(part.first);
                        t_cap = std::max(t_cap, lengths[idx]);
                        if (horizons) h_cap = std::max(h_cap, horizons[idx]);
                    }
                    while (v.size() % 64) { v.push_back(&dummy_v); mk.push_back(nullptr); len.push_back(0); src.push_back(SIZE_MAX); mcol.push_back(part.first); }
                    m_max = std::max(m_max, part.first);
                }
                const size_t nc = v.size();
                ForecastOptions o = *options;
                o.auto_detect_seasonality = false;
                o.seasonal_period = m_max;
                o.horizon = h_cap;
                AnofoxHipBatch *mb = nullptr;
                AnofoxError be;
                be.code = SUCCESS; be.message[0] = 0;
                std::vector<ForecastResult> res(nc);
                std::vector<AnofoxError> errs(nc);
                for (size_t j = 0; j < nc; j++) { std::memset(&res[j], 0, sizeof(ForecastResult)); errs[j].code = SUCCESS; errs[j].message[0] = 0; }
                const auto tm0 = std::chrono::steady_clock::now();
                bool ok = anofox_hip_batch_create(nc, t_cap, &o, &mb, &be);
                auto tm1 = std::chrono::steady_clock::now(), tm2 = tm1, tm3 = tm1;
                if (ok) {
                    ok = anofox_hip_batch_pack_host(mb, v.data(), validity ? mk.data() : nullptr, len.data(), &be);
                    tm2 = std::chrono::steady_clock::now();
                    if (ok) {
                        try {
                            mcol.resize(mb->ld, mcol.empty() ? 1 : mcol.back());
                            mb->d_m_col = dalloc<int32_t>(mb->ld);
                            HIPCHECK(hipMemcpy(mb->d_m_col, mcol.data(), mb->ld * sizeof(int32_t), hipMemcpyHostToDevice));
                            mb->merged_m_max = m_max;
                            for (size_t j = 0; j < nc; j++) mb->h_period[j] = mcol[j];     // what the host-side fitted values use
                        } catch (const HipFail &f) { report_hip_failure(&be, f); ok = false; }
                    }
                    ok = ok && anofox_hip_batch_run(mb, nullptr, &be);
                    tm3 = std::chrono::steady_clock::now();
                    ok = ok && anofox_hip_batch_fetch(mb, res.data(), errs.data());
                    anofox_hip_batch_destroy(mb);
                }
                if (tun.timing) {
                    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
                    std::fprintf(stderr, "[anofox-hip] merged batch: %zu parts with periods %d..%d in %zu columns: %.1f ms (create %.1f, pack %.1f, run %.1f, fetch + destroy %.1f)\n",
                                 take.size(), take.front().first, m_max, nc, ms(tm0, std::chrono::steady_clock::now()), ms(tm0, tm1), ms(tm1, tm2), ms(tm2, tm3),
                                 ms(tm3, std::chrono::steady_clock::now()));
                }
                if (!ok) {
                    if (be.code == SUCCESS) set_error(&be, INTERNAL_ERROR, "Internal error: device batch failed");
                    for (size_t j = 0; j < nc; j++) anofox_free_forecast_result(&res[j]);
                    std::lock_guard<std::mutex> lock(merged_mu);
                    if (out_batch_error && out_batch_error->code == SUCCESS) *out_batch_error = be;
                    return false;
                }
                for (size_t j = 0; j < nc; j++) {
                    if (src[j] == SIZE_MAX) { anofox_free_forecast_result(&res[j]); continue; }
                    ForecastResult &r = res[j];
                    const int hs = horizons ? horizons[src[j]] : options->horizon;      // forecasts are prefix-consistent: truncate
                    if (r.point_forecasts && hs >= 0 && (size_t)hs < r.n_forecasts) {
                        r.n_forecasts = (size_t)hs;
                        if (hs == 0) {
                            std::free(r.point_forecasts); std::free(r.lower_bounds); std::free(r.upper_bounds);
                            r.point_forecasts = r.lower_bounds = r.upper_bounds = nullptr;
                        }
                    }
                    out_results[src[j]] = r;
                    if (out_errors) out_errors[src[j]] = errs[j];
                }
                return true;
            };
            // (the LDS ring of a round kernel is sized by the batch's largest period, for every wave: periods up to 64 in one batch
            //  left three waves per CU -- a tenth of the uniform batch's rate per series -- so a merged batch keeps periods above 16
            //  in the HBM ring, which streams like y; more, finer classes only multiply the streams that share the hardware queues)
            static const int CLASS_HI[] = {ETS_MERGED_LDS_PERIOD, ETS_MAX_PERIOD};
            std::vector<Part> keep;
            std::vector<std::vector<Part>> take(sizeof CLASS_HI / sizeof CLASS_HI[0]);
            for (auto &part : parts) {
                int c = -1;
                if (arima_merge) { if (part.first > 24 && part.first <= ETS_MAX_PERIOD && part.second.size() < 2048) c = 1; }
                else if (do_merge && part.first >= 2 && part.second.size() < 2048)
                    for (size_t k = 0; k < take.size(); k++) if (part.first <= CLASS_HI[k]) { c = (int)k; break; }
                (c >= 0 ? take[(size_t)c] : keep).push_back(std::move(part));
            }
            // The HBM-ring class is cut where its ring scratch (workgroups of the widest launch x largest period x 512 B per seasonal
            // spec; the widest launch is the speculative driver's 16 problems per workgroup, or one wave per problem for the last
            // 2,048) would pass 2 GiB per spec: many rare long periods otherwise ask for tens of GB.
            {
                std::vector<std::vector<Part>> cut;
                auto scratch_of = [](size_t cols, int m_hi) { return (double)std::max<size_t>((cols + 15) / 16, std::min<size_t>(cols, 2048)) * (double)m_hi * 512.0; };
                for (auto &cls : take) {
                    std::sort(cls.begin(), cls.end(), [](const Part &x, const Part &y) { return x.first < y.first; });
                    std::vector<Part> cur;
                    size_t cols = 0;
                    for (auto &part : cls) {
                        const size_t pc = (part.second.size() + 63) / 64 * 64;
                        if (!cur.empty() && part.first > ETS_MERGED_LDS_PERIOD && scratch_of(cols + pc, part.first) > 2.0 * 1073741824.0) {
                            cut.push_back(std::move(cur));
                            cur.clear();
                            cols = 0;
                        }
                        cols += pc;
                        cur.push_back(std::move(part));
                    }
                    if (!cur.empty()) cut.push_back(std::move(cur));
                }
                take = std::move(cut);
            }
            // the classes side by side (each is latency bound by its slowest fit and far from filling the chip)
            std::vector<std::vector<Part> *> todo;
            for (auto &cls : take) {
                if (cls.size() >= 2) todo.push_back(&cls);
                else for (auto &part : cls) keep.push_back(std::move(part));
            }
            std::atomic<bool> merged_ok{true};
            int cur_dev_m = 0;
            (void)hipGetDevice(&cur_dev_m);
            // one host thread per LDS-ring batch, one for the HBM-ring batches (one after the other: they share the ring scratch)
            auto run_cls = [&](std::vector<std::vector<Part> *> group) {
                ArimaMethodScope method_scope(method_in_force);
                try {
                    (void)hipSetDevice(cur_dev_m);
                    for (auto *cls : group) if (!run_merged(*cls)) merged_ok = false;
                } catch (...) { merged_ok = false; }                // nothing may leave a worker thread
            };
            // ... and beside the parts that stay separate (below); joined before this function returns, whatever the way out
            std::vector<std::thread> cls_threads;
            struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } cls_joiner{cls_threads};
            std::vector<std::vector<Part> *> grp_hbm;
            // (a class whose thread cannot be created -- EAGAIN under the host's thread limit -- runs here, before the parts)
            auto start_cls = [&](std::vector<std::vector<Part> *> group) {
                try { cls_threads.emplace_back(run_cls, group); }
                catch (const std::system_error &) { run_cls(group); }
            };
            for (auto *cls : todo) {
                if (cls->back().first <= ETS_MERGED_LDS_PERIOD) start_cls(std::vector<std::vector<Part> *>{cls});
                else grp_hbm.push_back(cls);
            }
            if (!grp_hbm.empty()) start_cls(grp_hbm);
            parts = std::move(keep);
            std::sort(parts.begin(), parts.end(), [](const Part &x, const Part &y) { return x.second.size() > y.second.size(); });
        if (tun.timing) {
            std::string desc;
            for (auto &part : parts) desc += " " + std::to_string(part.first) + "x" + std::to_string(part.second.size());
            std::fprintf(stderr, "[anofox-hip] remaining parts (period x series):%s\n", desc.c_str());
        }
        // the big parts fill the chip on their own and hold the most memory: one at a time on this thread (below), beside the small
        // ones, which run side by side on the workers (a small part is a chain of latency-bound launches: it hardly slows a big one)
        size_t first_small = 0;
        while (first_small < parts.size() && parts[first_small].second.size() >= 2048) first_small++;
        // A worker keeps ONE device batch for all its parts (creating and destroying a batch costs ~30 ms of allocator calls
        // that serialise across threads -- as much as the part's own run): capacity = the largest small part, short parts are
        // padded with empty series, the period is set before every pack.
        const size_t cap = first_small < parts.size() ? parts[first_small].second.size() : 0;
        size_t t_cap = 0;
        int h_cap = options->horizon;
        for (size_t s = 0; s < n_series; s++) { t_cap = std::max(t_cap, lengths[s]); if (horizons) h_cap = std::max(h_cap, horizons[s]); }
        std::atomic<size_t> next{first_small};
        const auto t_small0 = std::chrono::steady_clock::now();
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);                      // the device is a per-thread setting: the workers inherit the caller's
        auto work_body = [&]() {
            (void)hipSetDevice(cur_dev);
            AnofoxHipBatch *wb = nullptr;
            static const double dummy = 0.0;
            std::vector<const double *> v(cap, &dummy);
            std::vector<const uint64_t *> m(cap, nullptr);
            std::vector<size_t> len(cap, 0);
            std::vector<ForecastResult> res(cap);
            std::vector<AnofoxError> errs(cap);
            for (size_t g = next.fetch_add(1); g < parts.size(); g = next.fetch_add(1)) {
                const std::vector<size_t> &idx = parts[g].second;
                const size_t k = idx.size();
                AnofoxError be;
                be.code = SUCCESS; be.message[0] = 0;
                if (!wb) {
                    ForecastOptions o = *options;
                    o.auto_detect_seasonality = false;
                    o.seasonal_period = parts[g].first > 1 ? parts[g].first : 0;
                    o.horizon = h_cap;
                    if (!anofox_hip_batch_create(cap, t_cap, &o, &wb, &be)) { wb = nullptr; run_part(parts[g]); continue; }   // the plain path reports it
                }
                wb->opt.seasonal_period = parts[g].first > 1 ? parts[g].first : 0;
                for (size_t j = 0; j < cap; j++) {
                    if (j < k) { v[j] = values[idx[j]]; m[j] = validity ? validity[idx[j]] : nullptr; len[j] = lengths[idx[j]]; }
                    else { v[j] = &dummy; m[j] = nullptr; len[j] = 0; }
                    std::memset(&res[j], 0, sizeof(ForecastResult));
                    errs[j].code = SUCCESS; errs[j].message[0] = 0;
                }
                const auto tw0 = std::chrono::steady_clock::now();
                const bool ok = anofox_hip_batch_pack_host(wb, v.data(), validity ? m.data() : nullptr, len.data(), &be) &&
                                anofox_hip_batch_run(wb, nullptr, &be) && anofox_hip_batch_fetch(wb, res.data(), errs.data());
                if (tun.timing)
                    std::fprintf(stderr, "[anofox-hip] small part: period %d, %zu series: %.1f ms (started at %.1f ms)\n", parts[g].first, k,
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count(),
                                 std::chrono::duration<double, std::milli>(tw0 - t_small0).count());
                if (!ok) {
                    if (be.code == SUCCESS) set_error(&be, INTERNAL_ERROR, "Internal error: device batch failed");
                    for (size_t j = 0; j < cap; j++) anofox_free_forecast_result(&res[j]);      // whatever the fetch had allocated
                    std::lock_guard<std::mutex> lock(err_mu);
                    if (all_ok) first_err = be;
                    all_ok = false;
                    continue;
                }
                for (size_t j = 0; j < k; j++) {
                    ForecastResult &r = res[j];
                    const int hs = horizons ? horizons[idx[j]] : options->horizon;      // forecasts are prefix-consistent: truncate
                    if (r.point_forecasts && hs >= 0 && (size_t)hs < r.n_forecasts) {
                        r.n_forecasts = (size_t)hs;
                        if (hs == 0) {
                            std::free(r.point_forecasts); std::free(r.lower_bounds); std::free(r.upper_bounds);
                            r.point_forecasts = r.lower_bounds = r.upper_bounds = nullptr;
                        }
                    }
                    out_results[idx[j]] = r;
                    if (out_errors) out_errors[idx[j]] = errs[j];
                }
            }
            if (wb) anofox_hip_batch_destroy(wb);
        };
        auto work = [&]() {
            ArimaMethodScope method_scope(method_in_force);
            try { work_body(); }
            catch (const std::exception &e) {           // nothing may leave a worker thread
                std::lock_guard<std::mutex> lock(err_mu);
                if (all_ok) set_error(&first_err, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
                all_ok = false;
            } catch (...) {
                std::lock_guard<std::mutex> lock(err_mu);
                if (all_ok) set_error(&first_err, INTERNAL_ERROR, "Internal error: worker failed");
                all_ok = false;
            }
        };
        // (a small part is a chain of latency-bound launches with host round trips in between -- AutoARIMA: ~13 sweeps of advance /
        //  read-back / fit, ~0.2 s whatever its size -- so the number of parts in flight is what the call's duration divides by)
        const size_t part_threads = (size_t)std::max(1, tun.part_threads);
        const unsigned n_thr = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), part_threads, parts.size() - first_small}));
        std::vector<std::thread> pool;
        struct PoolJoiner { std::vector<std::thread> &t; ~PoolJoiner() { for (auto &x : t) if (x.joinable()) x.join(); } } pool_joiner{pool};
        const bool have_big = first_small > 0;
        // (workers pull parts from a shared counter: the ones that cannot be created are not missed, this thread takes what is left)
        for (unsigned i = have_big ? 0 : 1; i < n_thr; i++) {
            try { pool.emplace_back(work); } catch (const std::system_error &) { break; }
        }
        if (have_big) {
            // the big parts: up to four side by side (one is throttled by whatever else runs; one after the other they were the
            // critical path of the call), each on its own thread, largest first
            std::atomic<size_t> next_big{0};
            auto big_work = [&]() {
                ArimaMethodScope method_scope(method_in_force);
                (void)hipSetDevice(cur_dev);
                try {
                    for (size_t g = next_big.fetch_add(1); g < first_small; g = next_big.fetch_add(1)) run_part(parts[g]);
                } catch (...) {                          // nothing may leave a worker thread
                    std::lock_guard<std::mutex> lock(err_mu);
                    if (all_ok) set_error(&first_err, INTERNAL_ERROR, "Internal error: worker failed (out of host memory)");
                    all_ok = false;
                }
            };
            const size_t n_big_thr = std::min<size_t>(first_small, 4);
            for (size_t i = 1; i < n_big_thr; i++) {
                try { pool.emplace_back(big_work); } catch (const std::system_error &) { break; }
            }
            big_work();
            work();                          // (nothing left unless the small parts' workers could not be started)
        } else work();
        for (auto &t : pool) t.join();
        for (auto &t : cls_threads) if (t.joinable()) t.join();
        if (!merged_ok) {
            if (out_batch_error && out_batch_error->code == SUCCESS) set_error(out_batch_error, INTERNAL_ERROR, "Internal error: device batch failed");
            return false;
        }
        if (!all_ok && out_batch_error) *out_batch_error = first_err;
        return all_ok;
    } catch (const std::exception &e) {
        set_error(out_batch_error, INTERNAL_ERROR, std::string("Internal error: ") + e.what());
        return false;
    }
}

// ---------------------------------------------------------------------------------------------
// Multi-device batch entry (north star: "series-id ranges shard trivially across the 8 GPUs of one node", behind ts_forecast_by;
// replaces the ONE-thread finalize loop of ts_forecast_native.cpp:559-800 with one host thread per device)
// ---------------------------------------------------------------------------------------------
bool anofox_hip_set_devices(const int *devices, size_t n_devices)
{
    std::vector<int> v;
    if (n_devices) {
        if (!devices) return false;
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess) return false;
        for (size_t i = 0; i < n_devices; i++) { if (devices[i] < 0 || devices[i] >= cnt) return false; v.push_back(devices[i]); }
    }
    DeviceList &d = device_list();
    std::lock_guard<std::mutex> lock(d.mu);
    d.env_read = true;
    d.devs = v;
    return true;
}

size_t anofox_hip_get_devices(int *devices, size_t capacity)
{
    const std::vector<int> v = devices_in_use(nullptr);
    for (size_t i = 0; i < v.size() && i < capacity && devices; i++) devices[i] = v[i];
    return v.size();
}

void anofox_hip_set_min_series_per_device(size_t min_series)
{
    DeviceList &d = device_list();
    std::lock_guard<std::mutex> lock(d.mu);
    d.min_series = std::max<size_t>(min_series, 1);
}

// contiguous ranges [g * ceil(N / G), (g + 1) * ceil(N / G)) -- SURVEY.md section 8(e), the same rule as dist.shard_range
void anofox_hip_shard_range(size_t n_series, size_t n_shards, size_t shard, size_t *begin, size_t *end)
{
    const size_t per = n_shards ? (n_series + n_shards - 1) / n_shards : n_series;
    const size_t lo = std::min(n_series, shard * per), hi = std::min(n_series, lo + per);
    if (begin) *begin = lo;
    if (end) *end = hi;
}

bool anofox_ts_forecast_batch(const double *const *values, const uint64_t *const *validity, const size_t *lengths, size_t n_series,
                              const ForecastOptions *options, const int *horizons, ForecastResult *out_results,
                              AnofoxError *out_errors, AnofoxError *out_batch_error)
{
    if (out_batch_error) { out_batch_error->code = SUCCESS; std::memset(out_batch_error->message, 0, sizeof out_batch_error->message); }
    if (!values || !lengths || !options || !out_results) { set_error(out_batch_error, NULL_POINTER, "Null pointer argument"); return false; }
    for (size_t s = 0; s < n_series; s++) {
        std::memset(&out_results[s], 0, sizeof(ForecastResult));
        if (!values[s]) { set_error(out_batch_error, NULL_POINTER, "Null pointer argument"); return false; }
    }
    size_t min_series = 2048;
    const std::vector<int> devs = devices_in_use(&min_series);
    // as many shards as there are listed devices, but none smaller than min_series (a shard that does not fill its device
    // finishes no sooner than a larger one: the fit is bound by its slowest problems)
    const size_t G = devs.empty() ? 1 : std::max<size_t>(1, std::min(devs.size(), n_series / std::max<size_t>(min_series, 1)));
    if (devs.empty()) {
        const auto tb0 = std::chrono::steady_clock::now();
        const bool r = forecast_batch_one_device(values, validity, lengths, n_series, options, horizons, out_results, out_errors, out_batch_error);
        if (Tunables::from_env().timing)
            std::fprintf(stderr, "[anofox-hip] anofox_ts_forecast_batch: %zu series in %.1f ms\n", n_series,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
        return r;
    }
    if (G == 1) {
        DeviceGuard guard(devs[0]);
        return forecast_batch_one_device(values, validity, lengths, n_series, options, horizons, out_results, out_errors, out_batch_error);
    }
    // one host thread per shard: its device made current, its own batch (stream set, allocator cache entries and pinned
    // staging block of that device), its slice of the caller's arrays -- the shards share nothing but the option block
    std::vector<AnofoxError> berr(G);
    std::vector<char> ok(G, 1);
    const bool timing = std::getenv("ANOFOX_HIP_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int method_in_force = arima_method_in_force();
    auto shard = [&](size_t g) {
        ArimaMethodScope method_scope(method_in_force);
        berr[g].code = SUCCESS; berr[g].message[0] = 0;
        try {
            size_t lo = 0, hi = 0;
            anofox_hip_shard_range(n_series, G, g, &lo, &hi);
            if (hi <= lo) return;
            if (hipSetDevice(devs[g]) != hipSuccess) { set_error(&berr[g], INTERNAL_ERROR, "Internal error: cannot select device " + std::to_string(devs[g])); ok[g] = 0; return; }
            tl_host_thread_share = (unsigned)G;
            ok[g] = forecast_batch_one_device(values + lo, validity ? validity + lo : nullptr, lengths + lo, hi - lo, options, horizons ? horizons + lo : nullptr,
                                              out_results + lo, out_errors ? out_errors + lo : nullptr, &berr[g]) ? 1 : 0;
            tl_host_thread_share = 1;
        } catch (const std::exception &e) {           // nothing may leave a worker thread
            set_error(&berr[g], INTERNAL_ERROR, std::string("Internal error: ") + e.what()); ok[g] = 0;
        } catch (...) { set_error(&berr[g], INTERNAL_ERROR, "Internal error: device shard failed"); ok[g] = 0; }
    };
    int caller_dev = 0;
    (void)hipGetDevice(&caller_dev);
    parallel_shares((unsigned)G, [&](unsigned g) { shard((size_t)g); });       // (a shard that finds no thread runs after shard 0, on this one)
    (void)hipSetDevice(caller_dev);
    if (timing)
        std::fprintf(stderr, "[anofox-hip] batch of %zu over %zu device shards: %.1f ms\n", n_series, G,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    bool all_ok = true;
    for (size_t g = 0; g < G; g++)
        if (!ok[g] || berr[g].code != SUCCESS) {
            // the option errors (INVALID_MODEL / INVALID_INPUT) are uniform across the shards: the first one speaks for the batch
            if (out_batch_error && out_batch_error->code == SUCCESS) *out_batch_error = berr[g];
            if (!ok[g]) all_ok = false;
        }
    return all_ok;
}

// Block 3 counterpart: run several device-resident batches -- typically one per device, each created after
// anofox_hip_set_device(d) -- side by side, one host thread per batch (a run synchronises its stream a few times, so one thread
// cannot drive several devices concurrently).  Returns false if any run failed; out_errors (may be NULL) is per batch.
bool anofox_hip_batch_run_many(AnofoxHipBatch *const *batches, size_t n_batches, AnofoxError *out_errors)
{
    if (!batches && n_batches) return false;
    std::vector<char> ok(n_batches, 1);
    auto one = [&](size_t i) {
        AnofoxError e;
        e.code = SUCCESS; e.message[0] = 0;
        try { ok[i] = anofox_hip_batch_run(batches[i], nullptr, &e) ? 1 : 0; }
        catch (...) { ok[i] = 0; set_error(&e, INTERNAL_ERROR, "Internal error: batch run failed"); }
        if (out_errors) out_errors[i] = e;
    };
    parallel_shares((unsigned)n_batches, [&](unsigned i) { one((size_t)i); });
    for (size_t i = 0; i < n_batches; i++) if (!ok[i]) return false;
    return true;
}

// one series on a parked one-series device batch (or a new one): pack, run, fetch, park again
static bool forecast_one_pooled(const double *values, const uint64_t *validity, size_t length, const ForecastOptions *options, const ForecastOptions &key,
                                ForecastResult *out_result, AnofoxError *out_error)
{
    AnofoxError e;
    e.code = SUCCESS;
    std::memset(e.message, 0, sizeof e.message);
    AnofoxError se;
    se.code = SUCCESS;
    std::memset(se.message, 0, sizeof se.message);
    const double *vals[1] = {values};
    const uint64_t *valid[1] = {validity};
    size_t lens[1] = {length};
    ForecastResult r;
    std::memset(&r, 0, sizeof r);
    // a pooled single-series batch: the option block is the key, the capacity covers series up to twice this length
    int device = 0;
    (void)hipGetDevice(&device);
    AnofoxHipBatch *b = pool_take(key, length, device);
    if (b) {
        try { batch_attach_streams(b); }
        catch (const HipFail &f) { report_hip_failure(out_error, f); anofox_hip_batch_destroy(b); return false; }
        b->arima_method = arima_method_in_force();      // a parked batch predates the method of this call
        b->tun = Tunables::from_env();
    }
    if (!b) {
        size_t cap = 256;
        while (cap < 2 * length) cap *= 2;
        if (!anofox_hip_batch_create(1, cap, options, &b, &e)) {
            if (out_error) *out_error = e;                     // INVALID_INPUT etc.: the series is long enough, so the option error is its error
            return false;
        }
    }
    const bool ok = anofox_hip_batch_pack_host(b, vals, validity ? valid : nullptr, lens, &e) && anofox_hip_batch_run(b, nullptr, &e) &&
                    anofox_hip_batch_fetch(b, &r, &se);
    if (!ok) {
        if (e.code == SUCCESS) set_error(&e, INTERNAL_ERROR, "Internal error: device batch failed");
        anofox_hip_batch_destroy(b);                            // a batch that failed is not reused
        if (out_error) *out_error = e;
        return false;
    }
    pool_give(key, device, b);
    if (se.code != SUCCESS) { if (out_error) *out_error = se; return false; }
    *out_result = r;
    return true;
}

// Route A coalescing (ts_forecast_scalar.cpp:298-523: every DuckDB worker thread calls anofox_ts_forecast once per group of its
// chunk, concurrently): a one-series fit is latency bound -- 5 ms for AutoETS, the chip all but idle -- so calls that are inside the
// library AT THE SAME TIME with an equal option block join ONE multi-series batch.  The first caller of a key opens a group and
// leads it: it waits until every call currently inside the library has joined or at most ANOFOX_HIP_COALESCE_US (default 1,000 us;
// a lone caller does not wait at all), closes the group and runs it through the batch entry; the followers sleep on the group and
// pick up their own result and their own error (per-series isolation is the batch entry's: out_errors).  Results are those of
// single calls bit for bit -- a series' result does not depend on the batch it is in (test_concurrent_single_series_calls).
struct CoalesceReq { const double *values; const uint64_t *validity; size_t length; ForecastResult res; AnofoxError err; bool ok = false; };
struct CoalesceGroup {
    ForecastOptions key; int device = 0; int method = 0;
    std::vector<CoalesceReq *> reqs;
    bool closed = false, done = false;
    std::condition_variable cv;
};
static std::mutex g_co_mu;
static std::vector<std::shared_ptr<CoalesceGroup>> g_co_open;
static int g_co_inflight = 0;                     // calls inside anofox_ts_forecast past their argument checks (under g_co_mu)
static int g_co_recent_width = 1;                 // the largest number of concurrent calls seen in the last 20 ms, and when: workers that call
static std::chrono::steady_clock::time_point g_co_recent_time;   // again and again arrive a few microseconds apart, so the first one back waits for its peers
constexpr size_t COALESCE_MAX = 256;

bool anofox_ts_forecast(const double *values, const uint64_t *validity, size_t length, const ForecastOptions *options,
                        ForecastResult *out_result, AnofoxError *out_error)
{
    if (out_error) { out_error->code = SUCCESS; std::memset(out_error->message, 0, sizeof out_error->message); }
    if (!values || !options || !out_result) { set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    // argument errors come first, exactly like the reference (lib.rs:3368-3380, forecast.rs:514-565)
    {
        std::string mname = cstr_field(options->model, sizeof options->model);
        ModelType mt;
        if (!parse_model(mname, mt)) { set_error(out_error, INVALID_MODEL, "Invalid model: Unknown model: '" + mname + "'"); return false; }
        if (options->horizon < 0) { set_error(out_error, PANIC_CAUGHT, "Panic in Rust code"); return false; }
        if (length == 0) { set_error(out_error, INSUFFICIENT_DATA, "Insufficient data: need at least 1 observations, got 0"); return false; }
        if (length < 3) {
            set_error(out_error, INSUFFICIENT_DATA, "Insufficient data: need at least 3 observations, got " + std::to_string(length));
            return false;
        }
    }
    // the option block is the key (every byte of it: callers memset it first, ts_forecast_scalar.cpp:440)
    ForecastOptions key;                       // normalised copy: padding and the bytes behind a string's NUL do not take part
    std::memset(&key, 0, sizeof key);
    auto copy_str = [](char *dst, const char *src, size_t cap) { for (size_t i = 0; i + 1 < cap && src[i]; i++) dst[i] = src[i]; };
    copy_str(key.model, options->model, sizeof key.model);
    copy_str(key.ets_model, options->ets_model, sizeof key.ets_model);
    copy_str(key.seasonal_periods_str, options->seasonal_periods_str, sizeof key.seasonal_periods_str);
    copy_str(key.model_pool, options->model_pool, sizeof key.model_pool);
    copy_str(key.laplace_variant, options->laplace_variant, sizeof key.laplace_variant);
    key.horizon = options->horizon; key.confidence_level = options->confidence_level; key.seasonal_period = options->seasonal_period;
    key.auto_detect_seasonality = options->auto_detect_seasonality; key.include_fitted = options->include_fitted;
    key.include_residuals = options->include_residuals; key.window = options->window;
    key.laplace_seasonal_batch_init = options->laplace_seasonal_batch_init;

    const double window_us = ProcessTunables::get().coalesce_us;
    if (!(window_us > 0.0)) return forecast_one_pooled(values, validity, length, options, key, out_result, out_error);
    int device = 0;
    (void)hipGetDevice(&device);
    const int method = g_default_arima_method.load();
    CoalesceReq req;
    req.values = values; req.validity = validity; req.length = length;
    std::memset(&req.res, 0, sizeof req.res);
    req.err.code = SUCCESS; req.err.message[0] = 0;
    std::shared_ptr<CoalesceGroup> grp;
    bool leader = false;
    {
        std::unique_lock<std::mutex> lock(g_co_mu);
        g_co_inflight++;
        const auto now = std::chrono::steady_clock::now();
        const bool recent = now - g_co_recent_time < std::chrono::milliseconds(20);
        if (g_co_inflight > 1 && (g_co_inflight >= g_co_recent_width || !recent)) { g_co_recent_width = g_co_inflight; g_co_recent_time = now; }
        else if (g_co_inflight > 1 && recent) g_co_recent_time = now;
        for (auto &g : g_co_open)
            if (!g->closed && g->device == device && g->method == method && g->reqs.size() < COALESCE_MAX && std::memcmp(&g->key, &key, sizeof key) == 0) { grp = g; break; }
        if (grp) {
            grp->reqs.push_back(&req);
            grp->cv.notify_all();                                  // the leader re-checks "has everyone inside joined?"
            grp->cv.wait(lock, [&] { return grp->done; });
            g_co_inflight--;
        } else {
            leader = true;
            grp = std::make_shared<CoalesceGroup>();
            grp->key = key; grp->device = device; grp->method = method;
            grp->reqs.push_back(&req);
            g_co_open.push_back(grp);
            const int expect = recent ? std::max(g_co_inflight, g_co_recent_width) : g_co_inflight;
            if (expect > 1) {
                // other calls are inside the library right now, or were a moment ago: give them the window to join (those of another
                // key never will: the window bounds what their presence costs; a caller that has always been alone never waits)
                const auto deadline = now + std::chrono::nanoseconds((long long)(window_us * 1000.0));
                grp->cv.wait_until(lock, deadline, [&] { return (int)grp->reqs.size() >= std::max(expect, g_co_inflight) || grp->reqs.size() >= COALESCE_MAX; });
            }
            grp->closed = true;
            g_co_open.erase(std::find(g_co_open.begin(), g_co_open.end(), grp));
        }
    }
    if (!leader) {
        if (!req.ok) { if (out_error) *out_error = req.err; return false; }
        *out_result = req.res;
        return true;
    }
    // ---- the leader runs the group (its members are blocked on the group; their request records live on their stacks) ----
    // Whatever happens below -- a refused allocation of the argument vectors included -- every member gets an answer and is woken:
    // the guard fills the requests nobody answered with INTERNAL_ERROR, marks the group done and gives the in-flight count back.
    const size_t k = grp->reqs.size();
    char answered[COALESCE_MAX] = {0};          // (no allocation outside the try block; k <= COALESCE_MAX)
    struct LeaderGuard {
        std::shared_ptr<CoalesceGroup> &grp;
        char *answered;
        ~LeaderGuard()
        {
            for (size_t j = 0; j < grp->reqs.size() && j < COALESCE_MAX; j++)
                if (!answered[j]) {
                    CoalesceReq *q = grp->reqs[j];
                    q->ok = false;
                    set_error(&q->err, INTERNAL_ERROR, "Internal error: coalesced batch failed");
                }
            std::lock_guard<std::mutex> lock(g_co_mu);
            grp->done = true;
            g_co_inflight--;
            grp->cv.notify_all();
        }
    };
    bool my_ok = false;
    try {
        LeaderGuard guard{grp, answered};
        ArimaMethodScope method_scope(grp->method);        // the method the members were matched on, not a re-read of the process default
        if (k == 1) {
            my_ok = forecast_one_pooled(values, validity, length, options, key, &req.res, &req.err);
            req.ok = my_ok;
            answered[0] = 1;
        } else {
            std::vector<const double *> v(k);
            std::vector<const uint64_t *> m(k);
            std::vector<size_t> len(k);
            bool any_mask = false;
            for (size_t j = 0; j < k; j++) { v[j] = grp->reqs[j]->values; m[j] = grp->reqs[j]->validity; len[j] = grp->reqs[j]->length; any_mask |= m[j] != nullptr; }
            std::vector<ForecastResult> res(k);
            std::vector<AnofoxError> errs(k);
            for (size_t j = 0; j < k; j++) { std::memset(&res[j], 0, sizeof(ForecastResult)); errs[j].code = SUCCESS; errs[j].message[0] = 0; }
            AnofoxError be;
            be.code = SUCCESS; be.message[0] = 0;
            bool ok = false;
            try {
                ok = forecast_batch_one_device(v.data(), any_mask ? m.data() : nullptr, len.data(), k, options, nullptr, res.data(), errs.data(), &be);
            } catch (...) { ok = false; }
            if (!ok && be.code == SUCCESS) set_error(&be, INTERNAL_ERROR, "Internal error: device batch failed");
            for (size_t j = 0; j < k; j++) {
                CoalesceReq *q = grp->reqs[j];
                if (!ok) { anofox_free_forecast_result(&res[j]); q->ok = false; q->err = be; }
                else if (errs[j].code != SUCCESS) { anofox_free_forecast_result(&res[j]); q->ok = false; q->err = errs[j]; }
                else { q->ok = true; q->res = res[j]; }
                answered[j] = 1;
            }
            my_ok = req.ok;
        }
    } catch (...) {
        // (the guard has answered and woken everyone; nothing leaves the C ABI)
        my_ok = false;
        if (req.err.code == SUCCESS) set_error(&req.err, INTERNAL_ERROR, "Internal error: coalesced batch failed (out of host memory)");
    }
    if (!my_ok) { if (out_error) *out_error = req.err; return false; }
    *out_result = req.res;
    return true;
}

} // extern "C"
