// ingest.hip -- columnar ingest for route B.  Host code only.
//
// Replaces the collection side of _ts_forecast_native (src/table_functions/ts_forecast_native.cpp:476-610): the
// reference appends row by row (`GetValue` boxing per cell) into a std::map<string, GroupData> under a global mutex
// and sorts every group by date at finalize.  Here a chunk of rows arrives as plain columns (dictionary id of the group
// value, date in microseconds or raw integers, target, two optional validity bitmasks in DuckDB's layout), is appended
// to a flat row log with one dictionary probe per row, and `finish` turns the log into series with one stable counting
// sort by group (first-appearance order, ts_forecast_native.cpp:586) and a date sort only for groups that arrived out of
// order.  Same rules as the reference: rows with a NULL date are dropped (`:505`), a NULL target is an invalid slot that
// the packer interpolates (imputation.rs:61-114), equal dates keep arrival order.  `anofox_hip_batch_pack_ingest` then
// hands the series to the existing packer (interpolation, period detection, time-major block, H2D).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/anofox_fcst_hip.h"
#include "host_semantics.hpp"

struct AnofoxHipIngest {
    std::mutex mu;
    std::unordered_map<int64_t, uint32_t> dict;      // group key -> group index (first appearance)
    std::vector<int64_t> keys;                       // [n_groups]
    // row log
    std::vector<uint32_t> r_gid;
    std::vector<int64_t> r_date;
    std::vector<double> r_val;
    std::vector<uint8_t> r_ok;
    // after finish
    bool finished = false;
    size_t t_max = 0;
    std::vector<size_t> len, off;                    // [n_groups], [n_groups + 1]
    std::vector<int64_t> last_date;                  // [n_groups]
    std::vector<double> val;                         // group-major, date order
    std::vector<uint64_t> mask;                      // per group ceil(len / 64) words, DuckDB layout
    std::vector<size_t> mask_off;
    std::vector<const double *> vptr;
    std::vector<const uint64_t *> mptr;
};

extern "C" {

AnofoxHipIngest *anofox_hip_ingest_create(void)
{
    try { return new AnofoxHipIngest(); } catch (...) { return nullptr; }
}

void anofox_hip_ingest_destroy(AnofoxHipIngest *g) { delete g; }

bool anofox_hip_ingest_append(AnofoxHipIngest *g, const int64_t *group_key, const int64_t *date, const uint64_t *date_valid,
                              const double *value, const uint64_t *value_valid, size_t n_rows, AnofoxError *out_error)
{
    if (!g || (n_rows && (!group_key || !date || !value))) { anofox::set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    try {
        std::lock_guard<std::mutex> lock(g->mu);
        if (g->finished) { anofox::set_error(out_error, INVALID_INPUT, "Invalid input: ingest already finished"); return false; }
        g->r_gid.reserve(g->r_gid.size() + n_rows);
        g->r_date.reserve(g->r_date.size() + n_rows);
        g->r_val.reserve(g->r_val.size() + n_rows);
        g->r_ok.reserve(g->r_ok.size() + n_rows);
        for (size_t i = 0; i < n_rows; i++) {
            if (date_valid && !((date_valid[i >> 6] >> (i & 63)) & 1ull)) continue;          // NULL date: row dropped
            auto it = g->dict.find(group_key[i]);
            uint32_t gid;
            if (it == g->dict.end()) {
                gid = (uint32_t)g->keys.size();
                g->dict.emplace(group_key[i], gid);
                g->keys.push_back(group_key[i]);
            } else gid = it->second;
            const bool ok = !value_valid || ((value_valid[i >> 6] >> (i & 63)) & 1ull);
            g->r_gid.push_back(gid);
            g->r_date.push_back(date[i]);
            g->r_val.push_back(ok ? value[i] : 0.0);
            g->r_ok.push_back(ok ? 1 : 0);
        }
    } catch (const std::exception &e) {
        anofox::set_error(out_error, ALLOCATION_ERROR, std::string("Allocation error: ") + e.what());
        return false;
    }
    return true;
}

bool anofox_hip_ingest_finish(AnofoxHipIngest *g, size_t *n_groups, size_t *t_max, AnofoxError *out_error)
{
    if (!g) { anofox::set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    try {
        std::lock_guard<std::mutex> lock(g->mu);
        if (!g->finished) {
            const size_t G = g->keys.size(), R = g->r_gid.size();
            g->len.assign(G, 0);
            for (size_t i = 0; i < R; i++) g->len[g->r_gid[i]]++;
            g->off.assign(G + 1, 0);
            for (size_t k = 0; k < G; k++) g->off[k + 1] = g->off[k] + g->len[k];
            // stable counting sort of the row log by group
            std::vector<size_t> cur(g->off.begin(), g->off.end() - 1), idx(R);
            for (size_t i = 0; i < R; i++) idx[cur[g->r_gid[i]]++] = i;
            g->val.resize(R);
            g->last_date.assign(G, 0);
            g->mask_off.assign(G + 1, 0);
            for (size_t k = 0; k < G; k++) g->mask_off[k + 1] = g->mask_off[k] + (g->len[k] + 63) / 64;
            g->mask.assign(g->mask_off[G], 0);
            g->t_max = 0;
            for (size_t k = 0; k < G; k++) {
                size_t *lo = idx.data() + g->off[k], *hi = idx.data() + g->off[k + 1];
                bool sorted = true;
                for (size_t *p = lo; p + 1 < hi; p++)
                    if (g->r_date[p[1]] < g->r_date[p[0]]) { sorted = false; break; }
                if (!sorted) std::stable_sort(lo, hi, [&](size_t a, size_t b) { return g->r_date[a] < g->r_date[b]; });
                uint64_t *mk = g->mask.data() + g->mask_off[k];
                for (size_t j = 0; j < g->len[k]; j++) {
                    const size_t r = lo[j];
                    g->val[g->off[k] + j] = g->r_val[r];
                    if (g->r_ok[r]) mk[j >> 6] |= 1ull << (j & 63);
                }
                if (g->len[k]) g->last_date[k] = g->r_date[hi[-1]];
                g->t_max = std::max(g->t_max, g->len[k]);
            }
            g->vptr.resize(G); g->mptr.resize(G);
            for (size_t k = 0; k < G; k++) { g->vptr[k] = g->val.data() + g->off[k]; g->mptr[k] = g->mask.data() + g->mask_off[k]; }
            std::vector<uint32_t>().swap(g->r_gid); std::vector<int64_t>().swap(g->r_date);
            std::vector<double>().swap(g->r_val); std::vector<uint8_t>().swap(g->r_ok);
            g->finished = true;
        }
        if (n_groups) *n_groups = g->keys.size();
        if (t_max) *t_max = g->t_max;
    } catch (const std::exception &e) {
        anofox::set_error(out_error, ALLOCATION_ERROR, std::string("Allocation error: ") + e.what());
        return false;
    }
    return true;
}

const int64_t *anofox_hip_ingest_group_keys(const AnofoxHipIngest *g) { return (g && g->finished) ? g->keys.data() : nullptr; }
const int64_t *anofox_hip_ingest_last_dates(const AnofoxHipIngest *g) { return (g && g->finished) ? g->last_date.data() : nullptr; }
const size_t *anofox_hip_ingest_lengths(const AnofoxHipIngest *g) { return (g && g->finished) ? g->len.data() : nullptr; }
const double *const *anofox_hip_ingest_values(const AnofoxHipIngest *g) { return (g && g->finished) ? g->vptr.data() : nullptr; }
const uint64_t *const *anofox_hip_ingest_validity(const AnofoxHipIngest *g) { return (g && g->finished) ? g->mptr.data() : nullptr; }

bool anofox_hip_batch_pack_ingest(AnofoxHipBatch *batch, const AnofoxHipIngest *g, AnofoxError *out_error)
{
    if (!batch || !g) { anofox::set_error(out_error, NULL_POINTER, "Null pointer argument"); return false; }
    if (!g->finished) { anofox::set_error(out_error, INVALID_INPUT, "Invalid input: ingest not finished"); return false; }
    // the packer indexes values / validity / lengths for every series of the batch: the batch must have been created for
    // exactly the ingested groups
    if (anofox_hip_batch_n_series(batch) != g->keys.size()) {
        anofox::set_error(out_error, INVALID_INPUT, "Invalid input: the batch was created for " + std::to_string(anofox_hip_batch_n_series(batch)) +
                                                        " series, the ingest holds " + std::to_string(g->keys.size()) + " groups");
        return false;
    }
    return anofox_hip_batch_pack_host(batch, g->vptr.data(), g->mptr.data(), g->len.data(), out_error);
}

} // extern "C"
