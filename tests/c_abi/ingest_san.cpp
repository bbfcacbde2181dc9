// Sanitizer harness for the columnar ingest (C-ABI block 4), CPU only: csrc/ingest.hip is plain host C++, so it is
// compiled here with g++ -fsanitize=address,undefined and driven with random chunks against the reference's own
// collection rule restated naively (ts_forecast_native.cpp:502-548, 586-600: map of groups in first-appearance order,
// rows with a NULL date dropped, NULL targets kept as invalid slots, stable sort by date at finalize).
//   g++ -std=c++17 -g -fsanitize=address,undefined -fno-sanitize-recover=all -x c++ csrc/ingest.hip -x c++ ingest_san.cpp
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "../../include/anofox_fcst_hip.h"

struct Seen { const double *const *values; const uint64_t *const *validity; const size_t *lengths; };
static Seen g_seen;

// The packer lives in host_api.hip (needs the HIP runtime); this harness only records what the ingest hands to it.
extern "C" bool anofox_hip_batch_pack_host(AnofoxHipBatch *, const double *const *values, const uint64_t *const *validity,
                                           const size_t *lengths, AnofoxError *)
{
    g_seen = Seen{values, validity, lengths};
    return true;
}

// ... and the batch it was "created" for: the harness names the series count the ingest is checked against
static size_t g_batch_n = 0;
extern "C" size_t anofox_hip_batch_n_series(const AnofoxHipBatch *) { return g_batch_n; }

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "FAIL %s:%d %s (seed %u)\n", __FILE__, __LINE__, #c, seed); return 1; } } while (0)

int main(int argc, char **argv)
{
    const unsigned rounds = argc > 1 ? (unsigned)std::atoi(argv[1]) : 200;
    for (unsigned seed = 1; seed <= rounds; seed++) {
        std::mt19937_64 rng(seed);
        const size_t n_groups = 1 + rng() % 40, n_chunks = rng() % 6;
        struct Row { int64_t date; double v; bool ok; size_t arrival; };
        std::map<int64_t, std::vector<Row>> ref;
        std::vector<int64_t> order;
        AnofoxHipIngest *g = anofox_hip_ingest_create();
        CHECK(g != nullptr);
        AnofoxError err;
        size_t arrival = 0;
        const bool shuffled = rng() % 2;
        for (size_t c = 0; c < n_chunks; c++) {
            const size_t n = rng() % 300;                                   // includes empty chunks
            std::vector<int64_t> key(n), date(n);
            std::vector<double> val(n);
            std::vector<uint64_t> dmask((n + 63) / 64, 0), vmask((n + 63) / 64, 0);
            const bool with_dmask = rng() % 2, with_vmask = rng() % 2;
            for (size_t i = 0; i < n; i++) {
                key[i] = (int64_t)(rng() % n_groups) * 7919 - 3;
                date[i] = shuffled ? (int64_t)(rng() % 50) - 10 : (int64_t)arrival + (int64_t)i;   // ties and negatives when shuffled
                val[i] = (double)(rng() % 1000) / 8.0;
                const bool d_ok = !with_dmask || rng() % 10 != 0, v_ok = !with_vmask || rng() % 5 != 0;
                if (d_ok) dmask[i >> 6] |= 1ull << (i & 63);
                if (v_ok) vmask[i >> 6] |= 1ull << (i & 63);
                if (!d_ok) continue;
                if (!ref.count(key[i])) order.push_back(key[i]);
                ref[key[i]].push_back(Row{date[i], v_ok ? val[i] : 0.0, v_ok, arrival + i});
            }
            arrival += n;
            CHECK(anofox_hip_ingest_append(g, n ? key.data() : nullptr, n ? date.data() : nullptr, with_dmask ? dmask.data() : nullptr,
                                           n ? val.data() : nullptr, with_vmask ? vmask.data() : nullptr, n, &err));
        }
        CHECK(anofox_hip_ingest_values(g) == nullptr);                      // nothing is visible before finish
        size_t G = 99, tmax = 99;
        CHECK(anofox_hip_ingest_finish(g, &G, &tmax, &err));
        CHECK(anofox_hip_ingest_finish(g, &G, &tmax, &err));                // idempotent
        CHECK(!anofox_hip_ingest_append(g, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &err) && err.code == INVALID_INPUT);
        CHECK(G == order.size());
        const int64_t *keys = anofox_hip_ingest_group_keys(g), *last = anofox_hip_ingest_last_dates(g);
        const size_t *len = anofox_hip_ingest_lengths(g);
        const double *const *vals = anofox_hip_ingest_values(g);
        const uint64_t *const *masks = anofox_hip_ingest_validity(g);
        size_t want_tmax = 0;
        for (size_t k = 0; k < G; k++) {
            CHECK(keys[k] == order[k]);
            auto rows = ref[order[k]];
            std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.date < b.date; });
            CHECK(len[k] == rows.size());
            want_tmax = std::max(want_tmax, rows.size());
            CHECK(last[k] == rows.back().date);
            for (size_t j = 0; j < rows.size(); j++) {
                CHECK(vals[k][j] == rows[j].v);
                CHECK((((masks[k][j >> 6]) >> (j & 63)) & 1ull) == (rows[j].ok ? 1ull : 0ull));
            }
        }
        CHECK(tmax == want_tmax);
        g_batch_n = G + 1;                                    // a batch created for another number of series is refused
        CHECK(!anofox_hip_batch_pack_ingest((AnofoxHipBatch *)(uintptr_t)0x10, g, &err) && err.code == INVALID_INPUT);
        g_batch_n = G;
        CHECK(anofox_hip_batch_pack_ingest((AnofoxHipBatch *)(uintptr_t)0x10, g, &err));
        CHECK(g_seen.values == vals && g_seen.validity == masks && g_seen.lengths == len);
        CHECK(!anofox_hip_batch_pack_ingest(nullptr, g, &err) && err.code == NULL_POINTER);
        anofox_hip_ingest_destroy(g);
    }
    anofox_hip_ingest_destroy(nullptr);
    std::printf("OK %u rounds\n", rounds);
    return 0;
}
