// tunables.hpp — every switch of the library the environment can set, parsed in one place (not where it is used).
//
// Product switches choose between mechanisms that give the same results (an A/B run, a fallback forced on for the parity
// suite): they are documented in INTEGRATION.md.  Diagnostic switches (dumps, per-wave clock stamps, timing-only kernels)
// exist only in builds with -DRSREG_DIAG (RSREG_CXXFLAGS=-DRSREG_DIAG; `python -c "import rsreg_amd.lib as l; print(l.build_diag())"`):
// the shipped library has no switch that writes a file or changes a result.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>

namespace rsreg {

struct Tunables {
    // ---- target index
    double cell_cap = 0.014;               // RSREG_CELL_CAP: metres; a few D435i pixel pitches at 1-2 m (swept on MI355X: DESIGN.md §5)
    long long dense_max_cells = 1ll << 28; // RSREG_DENSE_MAX_CELLS (RSREG_FORCE_HASH=1: 0): 1 GiB of cell starts at most (out of 288 GB)
    bool keys64 = false;                   // RSREG_KEYS64=1: 64-bit sort keys whatever the grid
    bool full_table = false;               // RSREG_FULL_TABLE=1: the sort-based build writes every table entry
    bool far_rows = false;                 // RSREG_FAR_ROWS: the row search beyond ring 1 for every gate (default: only unbounded / wide gates)
    bool wide_cells = true;                // RSREG_NO_WIDE_CELLS=1: cells never larger than the gate
    bool adaptive_cell = true;             // RSREG_NO_ADAPTIVE_CELL=1: the brick-hash build never refines its cell size
    bool box_cache = true;                 // RSREG_NO_BOX_CACHE=1: a cloud handle's bounding box is measured by every build / load
    bool count_sort = true;                // RSREG_COUNT_SORT=0: the index by sorting (k_dense_keys, radix sort, k_dense_compact) instead of counting (cellsort.hpp)
    bool one_side_worker = false;          // RSREG_ONE_SIDE_WORKER=1: the side jobs of a context (filters, edge extractions of the frames ahead) on one thread as in rounds 3-5 (default: two, alternating)
    bool nbr_from_table = true;            // RSREG_NO_NBR_FROM_TABLE=1: the counting build makes the occupancy words for a small source too (default: left out when the gate fits into ring 1, the search reads them off the cell table)
    bool cc_apart = false;                 // RSREG_CC_APART=1: the counting build's scatter / occupancy words / small cells / crowded cells as four launches (rounds 5; default: two)
    bool scan_apart = false;               // RSREG_SCAN_APART=1: sort-based build: flag, scan, scatter as three launches
    // ---- source
    bool sort_small = false;               // RSREG_SORT_SMALL=1: sources of <= 65 536 points are put into spatial order too
    size_t plain_source_max = 65536;       // RSREG_PLAIN_SOURCE_MAX
    int morton_bits = 23;                  // RSREG_MORTON_BITS: bits of the source's Morton keys (+ the invalid bit: three digit passes)
    bool worker = true;                    // RSREG_NO_WORKER=1: the source load's host side on the caller's thread
    bool seed = true;                      // RSREG_NO_SEED=1: searches never start from the previous iteration's match
    bool restart_apart = false;            // RSREG_RESTART_APART=1: guess * source as a launch of its own
    bool scan_target = true;               // RSREG_NO_SCAN: an index even for a handful of source points
    // ---- tile schedule of the fused search kernel
    bool sched = true;                     // RSREG_SCHED=0
    double sched_f4 = 0.0, sched_f2 = 0.10;   // RSREG_SCHED_F4 / _F2: fractions of the tiles searched by 4 / 2 lanes per query
    uint32_t sched_min_tiles = 1024;       // RSREG_SCHED_MIN_TILES
    int sched_at = 1;                      // RSREG_SCHED_AT: the launch that is timed
    bool sched_xcd = true;                 // RSREG_SCHED_XCD=0
    uint32_t sched_xcd_deal = 32;          // RSREG_SCHED_XCD_DEAL
    bool sched_keep = true;                // RSREG_SCHED_KEEP=0: every alignment times a launch of its own and builds its own schedule
    // ---- clouds, NDT
    long long cloud_pool_mb = 4096;        // RSREG_CLOUD_POOL_MB
    bool upload_wait_staged = false;       // RSREG_UPLOAD_WAIT_STAGED=1
    bool ndt_watch = true;                 // RSREG_NDT_NO_WATCH: hipStreamSynchronize instead of watching the stamped pass number
    bool ndt_one_launch = false;           // RSREG_NDT_ONE_LAUNCH=1: a derivative pass as ONE launch whose last workgroup adds the slabs (k_ndt_pass_reduce; same bits, 8 us per pass SLOWER: DESIGN.md §5f) instead of k_ndt_pass + k_ndt_final_reduce
    bool ndt_resident_ls = false;          // RSREG_NDT_RESIDENT_LS=1: all the passes of a line search in one launch (k_ndt_line_search; same bits, not faster: DESIGN.md §5e)
#ifdef RSREG_DIAG
    const char *dump_seed = nullptr, *wave_times = nullptr, *edge_dump = nullptr;   // RSREG_DUMP_SEED, RSREG_WAVE_TIMES, RSREG_EDGE_DUMP: files
    bool wave_times_light = false, dump_nn_ms = false, sched_verbose = false, grid_stats = false;
    uint32_t debug_skip = 0;               // RSREG_DEBUG_SKIP: results WRONG (timing experiments)
#endif
};

inline Tunables tunables_from_environment()
{
    Tunables v;
    auto on = [](const char *n) { const char *e = std::getenv(n); return e && e[0] == '1'; };
    auto off = [](const char *n) { const char *e = std::getenv(n); return e && e[0] == '0'; };
    auto set = [](const char *n) { return std::getenv(n) != nullptr; };
    if (const char *e = std::getenv("RSREG_CELL_CAP")) { const double c = std::atof(e); if (c > 0) v.cell_cap = c; }
    if (const char *e = std::getenv("RSREG_DENSE_MAX_CELLS")) v.dense_max_cells = std::atoll(e);
    if (on("RSREG_FORCE_HASH")) v.dense_max_cells = 0;
    v.keys64 = on("RSREG_KEYS64");
    v.full_table = on("RSREG_FULL_TABLE");
    v.far_rows = set("RSREG_FAR_ROWS");
    v.wide_cells = !on("RSREG_NO_WIDE_CELLS");
    v.adaptive_cell = !on("RSREG_NO_ADAPTIVE_CELL");
    v.box_cache = !on("RSREG_NO_BOX_CACHE");
    v.count_sort = !off("RSREG_COUNT_SORT");
    v.cc_apart = on("RSREG_CC_APART");
    v.nbr_from_table = !on("RSREG_NO_NBR_FROM_TABLE");
    v.one_side_worker = on("RSREG_ONE_SIDE_WORKER");
    v.scan_apart = on("RSREG_SCAN_APART");
    v.sort_small = on("RSREG_SORT_SMALL");
    if (const char *e = std::getenv("RSREG_PLAIN_SOURCE_MAX")) v.plain_source_max = (size_t)std::atoll(e);
    if (const char *e = std::getenv("RSREG_MORTON_BITS")) v.morton_bits = std::max(6, std::min(31, std::atoi(e)));
    v.worker = !on("RSREG_NO_WORKER");
    v.seed = !on("RSREG_NO_SEED");
    v.restart_apart = on("RSREG_RESTART_APART");
    v.scan_target = !set("RSREG_NO_SCAN");
    v.sched = !off("RSREG_SCHED");
    if (const char *e = std::getenv("RSREG_SCHED_F4")) v.sched_f4 = std::atof(e);
    if (const char *e = std::getenv("RSREG_SCHED_F2")) v.sched_f2 = std::atof(e);
    if (const char *e = std::getenv("RSREG_SCHED_MIN_TILES")) v.sched_min_tiles = (uint32_t)std::atoll(e);
    if (const char *e = std::getenv("RSREG_SCHED_AT")) v.sched_at = std::atoi(e);
    v.sched_f4 = std::min(std::max(v.sched_f4, 0.0), 1.0);
    v.sched_f2 = std::min(std::max(v.sched_f2, 0.0), 1.0 - v.sched_f4);   // (every tile at most once: up to 4 workgroups per tile)
    v.sched_xcd = !off("RSREG_SCHED_XCD");
    if (const char *e = std::getenv("RSREG_SCHED_XCD_DEAL")) v.sched_xcd_deal = (uint32_t)std::atoi(e);
    v.sched_keep = !off("RSREG_SCHED_KEEP");
    if (const char *e = std::getenv("RSREG_CLOUD_POOL_MB")) v.cloud_pool_mb = std::max(0ll, std::atoll(e));
    v.upload_wait_staged = on("RSREG_UPLOAD_WAIT_STAGED");
    v.ndt_watch = !set("RSREG_NDT_NO_WATCH");
    v.ndt_one_launch = on("RSREG_NDT_ONE_LAUNCH");
    v.ndt_resident_ls = on("RSREG_NDT_RESIDENT_LS");
#ifdef RSREG_DIAG
    v.dump_seed = std::getenv("RSREG_DUMP_SEED");
    v.wave_times = std::getenv("RSREG_WAVE_TIMES");
    v.edge_dump = std::getenv("RSREG_EDGE_DUMP");
    v.wave_times_light = set("RSREG_WAVE_TIMES_LIGHT");
    v.dump_nn_ms = set("RSREG_DUMP_NN_MS");
    v.sched_verbose = set("RSREG_SCHED_VERBOSE");
    v.grid_stats = set("RSREG_GRID_STATS");
    if (const char *e = std::getenv("RSREG_DEBUG_SKIP")) v.debug_skip = (uint32_t)std::atoi(e);
#endif
    return v;
}

// What the environment says about every switch above, as one string: two readings are compared by this, not by the bytes
// of two structs (whose padding is indeterminate).
inline std::string tunables_signature()
{
    static const char *const names[] = {
        "RSREG_CELL_CAP", "RSREG_DENSE_MAX_CELLS", "RSREG_FORCE_HASH", "RSREG_KEYS64", "RSREG_FULL_TABLE", "RSREG_FAR_ROWS", "RSREG_NO_WIDE_CELLS",
        "RSREG_NO_ADAPTIVE_CELL", "RSREG_NO_BOX_CACHE", "RSREG_COUNT_SORT", "RSREG_SCAN_APART", "RSREG_CC_APART", "RSREG_SORT_SMALL",
        "RSREG_PLAIN_SOURCE_MAX", "RSREG_MORTON_BITS", "RSREG_NO_WORKER", "RSREG_NO_SEED", "RSREG_RESTART_APART", "RSREG_NO_SCAN", "RSREG_SCHED",
        "RSREG_SCHED_F4", "RSREG_SCHED_F2", "RSREG_SCHED_MIN_TILES", "RSREG_SCHED_AT", "RSREG_SCHED_XCD", "RSREG_SCHED_XCD_DEAL", "RSREG_SCHED_KEEP",
        "RSREG_CLOUD_POOL_MB", "RSREG_UPLOAD_WAIT_STAGED", "RSREG_NDT_NO_WATCH", "RSREG_NDT_RESIDENT_LS", "RSREG_NDT_ONE_LAUNCH",
#ifdef RSREG_DIAG
        "RSREG_DUMP_SEED", "RSREG_WAVE_TIMES", "RSREG_EDGE_DUMP", "RSREG_WAVE_TIMES_LIGHT", "RSREG_DUMP_NN_MS", "RSREG_SCHED_VERBOSE", "RSREG_GRID_STATS",
        "RSREG_DEBUG_SKIP",
#endif
    };
    std::string sig;
    for (const char *n : names) {
        const char *e = std::getenv(n);
        sig += e ? e : "\x01";
        sig += '\0';
    }
    return sig;
}

// The switches are published as a pointer to an immutable struct: a reader (a context's worker threads among them) holds a
// reference to a complete set that never changes under it; a new reading is a new struct swapped in (the old ones stay
// allocated: a few hundred bytes per change of the environment, which only A/B tools ever make).
struct TunablesBox {
    std::atomic<const Tunables *> cur{nullptr};
    std::mutex mu;
    std::string sig;
};

inline TunablesBox &tunables_box()
{
    static TunablesBox box;
    return box;
}

// the switches as the environment had them when the process first asked -- or when a context was last created
// (rsreg_ctx_create looks again, so that one process can compare settings run by run; nothing is published when the
// environment has not changed.  A change applies to every context of the process from then on: set the environment
// before the first context is created, or between runs -- not while contexts are at work)
inline const Tunables &tunables()
{
    TunablesBox &b = tunables_box();
    const Tunables *t = b.cur.load(std::memory_order_acquire);
    if (t) return *t;
    std::lock_guard<std::mutex> lk(b.mu);
    t = b.cur.load(std::memory_order_relaxed);
    if (!t) {
        b.sig = tunables_signature();
        t = new Tunables(tunables_from_environment());
        b.cur.store(t, std::memory_order_release);
    }
    return *t;
}

inline void tunables_refresh()
{
    (void)tunables();
    TunablesBox &b = tunables_box();
    std::lock_guard<std::mutex> lk(b.mu);
    std::string now = tunables_signature();
    if (now == b.sig) return;
    b.sig = std::move(now);
    b.cur.store(new Tunables(tunables_from_environment()), std::memory_order_release);
}

}  // namespace rsreg
